"""full-size (config S, B=1) gradient comparison: HIP (Winograd on/off) vs oracle, and the oracle against itself with another thread count"""
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


def oracle(threads):
    torch.set_num_threads(threads)
    P = syn.synthetic_state_dict(manifest['state_dict'], 1)
    for k in manifest['parameters']:
        P[k].requires_grad_(True)
    pred = O.forward(P, *cpu, syn.default_args(RAW, 'cpu'), train=True)
    L, _ = O.compute_loss(cpu[0], {k: T(v) for k, v in b['gt'].items()}, pred, syn.default_args(RAW, 'cpu'))
    L['total'].backward()
    return pred, {k: P[k].grad.double() for k in manifest['parameters']}


def rel(ga, gb):
    num = {n: 0.0 for n in 'EHFG'}; den = dict(num)
    for k in manifest['parameters']:
        if skip.search(k): continue
        num[k[0]] += float((ga[k] - gb[k]).pow(2).sum()); den[k[0]] += float(gb[k].pow(2).sum())
    return {n: '%.2e' % ((num[n] / max(den[n], 1e-300)) ** 0.5) for n in 'EHFG'}


pred_o, g32 = oracle(32)
_, g4 = oracle(4)
print('oracle 4 threads vs oracle 32 threads:', rel(g4, g32))
h_img_o = pred_o['h_img'].detach().cuda()
ops.rotate_nearest_u8 = lambda img, rot, **kw: (h_img_o, ops.nchw_to_nhwc(h_img_o, 4))

runs = []
for it in range(2):
    m = EFGHBackbone(syn.default_args(RAW, 'cuda'))
    m.load_state_dict(syn.synthetic_state_dict(manifest['state_dict'], 1))
    m = m.cuda().train()
    gpu = [t.cuda() for t in cpu]
    pred = m(*gpu)
    L, _ = EFGHCriterion(syn.default_args(RAW, 'cuda')).compute_loss(*gpu, {k: T(v) for k, v in b['gt'].items()}, pred)
    L['total'].backward()
    runs.append({k: p.grad.cpu().double() for k, p in m.named_parameters()})
print('HIP run 1 vs oracle:', rel(runs[0], g32), ' HIP run 2 vs oracle:', rel(runs[1], g32), ' HIP run 1 vs run 2:', rel(runs[0], runs[1]))
rows = []
for k in manifest['parameters']:
    if k[0] != 'G' or skip.search(k): continue
    d = float((runs[0][k] - g32[k]).norm()); n = float(g32[k].norm()); d12 = float((runs[0][k] - runs[1][k]).norm())
    rows.append((d, n, d12, k))
rows.sort(reverse=True)
tot = sum(r[1] ** 2 for r in rows) ** 0.5
print('G total oracle grad norm %.3e' % tot)
for d, n, d12, k in rows[:14]:
    print('%-44s |hip-oracle| %.3e  |oracle| %.3e  rel %.2e   |hip1-hip2| %.3e' % (k, d, n, d / max(n, 1e-30), d12))
