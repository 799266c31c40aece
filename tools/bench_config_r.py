"""BASELINE configs[0] (the reference's shipped RELLIS configuration: raw 900x1600, 65 536 points, batch 1) on the GPU box:
eval forward and one training step, for the comparison with the oracle timings in SURVEY 8(d)"""
import sys
import time
sys.path.insert(0, '/root/repo')
import torch
from efgh_amd import synthetic as syn
from efgh_amd.losses import EFGHCriterion
from efgh_amd.nets import EFGHBackbone
from efgh_amd.train import Trainer

raw, n = (900, 1600), 65536
args = syn.default_args(raw, 'cuda')
torch.manual_seed(0)
m = EFGHBackbone(args).cuda()
b = syn.make_batch(raw, n, 1)
inp = [torch.from_numpy(b[k]).cuda() for k in ('pc', 'img', 'calib', 'A')]
gt = {k: torch.from_numpy(v).cuda() for k, v in b['gt'].items()}


def timeit(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


m.eval()
with torch.no_grad():
    print('eval forward, batch 1: %.1f ms' % timeit(lambda: m(*inp)))
tr = Trainer(m, EFGHCriterion(args), lr=1e-4)
print('training step, batch 1: %.1f ms' % timeit(lambda: tr.step(*inp, gt)))
