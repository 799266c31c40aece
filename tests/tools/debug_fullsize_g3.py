"""full size: G gradients of the depth+mask loss terms, HIP vs oracle"""
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from efgh_amd import ops, synthetic as syn
from efgh_amd.losses import EFGHCriterion
from efgh_amd.nets import EFGHBackbone
from oracle import efgh_oracle as O
RAW, NPTS = (768, 2560), 131072
manifest = json.load(open('tests/golden/state_dict_manifest.json'))
b = syn.make_batch(RAW, NPTS, 1)
T = torch.from_numpy
cpu = [T(b[k]) for k in ('pc', 'img', 'calib', 'A')]
skip = re.compile(r'(features\.\d+|conv_gn_\d|conv_hrzn_\d|E\.bcn5\.blur_conv\.2)\.bias$')
torch.set_num_threads(32)
P = syn.synthetic_state_dict(manifest['state_dict'], 1)
gnames = [k for k in manifest['parameters'] if k.startswith('G.') and not skip.search(k)]
for k in manifest['parameters']:
    P[k].requires_grad_(True)
pred_o = O.forward(P, *cpu, syn.default_args(RAW, 'cpu'), train=True)
L_o, _ = O.compute_loss(cpu[0], {k: T(v) for k, v in b['gt'].items()}, pred_o, syn.default_args(RAW, 'cpu'))
go = {}
for term in ('g_depth', 'g_mask', 'g_trs'):
    gr = torch.autograd.grad(L_o[term], [P[k] for k in gnames], retain_graph=True, allow_unused=True)
    go[term] = [None if g is None else g.double() for g in gr]
h_img_o = pred_o['h_img'].detach().cuda()
ops.rotate_nearest_u8 = lambda img, rot, **kw: (h_img_o, ops.nchw_to_nhwc(h_img_o, 4))
m = EFGHBackbone(syn.default_args(RAW, 'cuda'))
m.load_state_dict(syn.synthetic_state_dict(manifest['state_dict'], 1))
m = m.cuda().train()
gpu = [t.cuda() for t in cpu]
pred = m(*gpu)
L, _ = EFGHCriterion(syn.default_args(RAW, 'cuda')).compute_loss(*gpu, {k: T(v) for k, v in b['gt'].items()}, pred)
params = dict(m.named_parameters())
for term in ('g_trs',):
    gr = torch.autograd.grad(L[term], [params[k] for k in gnames], retain_graph=True, allow_unused=True)
    for k, a, c in list(zip(gnames, gr, go[term]))[-40:]:
        if a is not None and c is not None: print('   %-36s |oracle| %.3e rel %.3e' % (k, float(c.norm()), float((a.cpu().double() - c).norm() / max(float(c.norm()), 1e-300))))
    num = den = 0.0
    worst = (0, '')
    for k, a, c in zip(gnames, gr, go[term]):
        if a is None or c is None: continue
        d = float((a.cpu().double() - c).pow(2).sum()); n = float(c.pow(2).sum())
        num += d; den += n
        if n > 0 and (d / n) ** 0.5 > worst[0]: worst = ((d / n) ** 0.5, k)
    print(term, 'G grad rel err HIP vs oracle: %.3e   worst layer %.3e %s' % ((num / max(den, 1e-300)) ** 0.5, worst[0], worst[1]))
