"""per-layer census of one full-size training step (debug aid): which layers carry the BatchNorm / activation traffic, which kernel
serves their data gradient, and what produced their input"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efgh_amd import synthetic as syn
from efgh_amd.losses import EFGHCriterion
from efgh_amd.nets import EFGHBackbone, fn as FN
from efgh_amd.train import Trainer
raw = (768, 2560)
args = syn.default_args(raw, 'cuda')
torch.manual_seed(0)
tr = Trainer(EFGHBackbone(args).cuda(), EFGHCriterion(args), lr=1e-4)
b = syn.make_batch(raw, 131072, 8)
inp = [torch.from_numpy(b[k]).cuda() for k in ('pc', 'img', 'calib', 'A')]
gt = {k: torch.from_numpy(v).cuda() for k, v in b['gt'].items()}
tr.step(*inp, gt)
FN.TRACE = []
tr.step(*inp, gt)
torch.cuda.synchronize()
rows = FN.TRACE
FN.TRACE = None
agg = collections.OrderedDict()
for r in rows:
    kind = 'wino2d' if r['wino2d'] else 'wino' if r['wino'] else 'c4' if r['c4'] else 'custom' if r['custom'] else 'gemm%d' % r['mode']
    key = (r['M'], r['C'], r['N'], r['T'], kind, r['bn'], r['res'], r['pool'], r['x_from'], r['dx'])
    agg[key] = agg.get(key, 0) + 1
print('M C N T kernel bn res pool x_from dx | layers | act MB (out) | in MB')
for k, n in sorted(agg.items(), key=lambda kv: -kv[0][0] * kv[0][2] * kv[1]):
    print(*k, '|', n, '| %.0f | %.0f' % (k[0] * k[2] * 4 / 1e6, k[0] * k[1] * 4 / 1e6))
