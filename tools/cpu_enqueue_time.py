"""host time to ENQUEUE one training step (no synchronisation inside) against the GPU time of the step: how far the CPU runs ahead"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efgh_amd import synthetic as syn
from efgh_amd.losses import EFGHCriterion
from efgh_amd.nets import EFGHBackbone
from efgh_amd.train import Trainer
raw = (768, 2560)
args = syn.default_args(raw, 'cuda')
torch.manual_seed(0)
tr = Trainer(EFGHBackbone(args).cuda(), EFGHCriterion(args), lr=1e-4)
b = syn.make_batch(raw, 131072, 8)
inp = [torch.from_numpy(b[k]).cuda() for k in ('pc', 'img', 'calib', 'A')]
gt = {k: torch.from_numpy(v).cuda() for k, v in b['gt'].items()}
for _ in range(3):
    tr.step(*inp, gt)
torch.cuda.synchronize()
cpu, tot = [], []
for _ in range(5):
    t0 = time.perf_counter()
    tr.step(*inp, gt)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    cpu.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
print('host enqueue %.1f ms (min %.1f), step until idle %.1f ms' % (sum(cpu) / len(cpu), min(cpu), sum(tot) / len(tot)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); tr.step(*inp, gt); pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(22)
