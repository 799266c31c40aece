import sys, json, os, re
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from efgh_amd import synthetic as syn, ops
from efgh_amd.losses import EFGHCriterion
from efgh_amd.nets import EFGHBackbone
from oracle import efgh_oracle as O
RAW, NPTS = (128, 256), 2048
man = json.load(open('/root/repo/tests/golden/state_dict_manifest.json'))
args_c, args_g = syn.default_args(RAW, 'cpu'), syn.default_args(RAW, 'cuda')
b = syn.make_batch(RAW, NPTS, 1)
T = torch.from_numpy
cpu = [T(b[k]) for k in ('pc', 'img', 'calib', 'A')]
P = syn.synthetic_state_dict(man['state_dict'], 1)
for k in man['parameters']: P[k].requires_grad_(True)
pred_o = O.forward(P, *cpu, args_c, train=True)
only = sys.argv[1] if len(sys.argv) > 1 else 'total'
L_o, _ = O.compute_loss(cpu[0], {k: T(v) for k, v in b['gt'].items()}, pred_o, args_c)
L_o[only].backward()
h_img_o = pred_o['h_img'].detach().cuda()
ops.rotate_nearest_u8 = lambda img, rot, **kw: (h_img_o, ops.nchw_to_nhwc(h_img_o, 4))
m = EFGHBackbone(args_g); m.load_state_dict(syn.synthetic_state_dict(man['state_dict'], 1)); m = m.cuda().train()
crit = EFGHCriterion(args_g)
gpu = [t.cuda() for t in cpu]
pred = m(*gpu)
L, _ = crit.compute_loss(*gpu, {k: T(v) for k, v in b['gt'].items()}, pred)
L[only].backward()
params = dict(m.named_parameters())
for k in man['parameters']:
    g_o, g = P[k].grad, params[k].grad
    if g_o is None and g is None: continue
    if g_o is None: g_o = torch.zeros_like(P[k])
    g = torch.zeros_like(g_o) if g is None else g.cpu()
    err = float((g - g_o).norm() / (g_o.norm() + 1e-20))
    if err > 1e-3: print('%-40s err %.2e  norm %.3e' % (k, err, float(g_o.norm())))
for k in ('g_depth', 'g_mask', 'g_trs', 'f_score'):
    print(k, float((pred[k].detach().cpu() - pred_o[k].detach()).abs().max() / pred_o[k].detach().abs().max()))
