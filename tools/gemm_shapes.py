"""per-shape table of the generic implicit-GEMM launches (k_gather_gemm / k_gather_wgrad, every mode) of one full-size training step
on ONE stream: launches, ms per step, TFLOP/s - where the `roofline_gemm` / `roofline_wgrad` families spend their time"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efgh_amd import ops, synthetic as syn
from efgh_amd.losses import EFGHCriterion
from efgh_amd.nets import EFGHBackbone, efghbackbone as bb
from efgh_amd.train import Trainer
raw = (768, 2560)
args = syn.default_args(raw, 'cuda')
torch.manual_seed(0)
tr = Trainer(EFGHBackbone(args).cuda(), EFGHCriterion(args), lr=1e-4)
b = syn.make_batch(raw, 131072, 8)
inp = [torch.from_numpy(b[k]).cuda() for k in ('pc', 'img', 'calib', 'A')]
gt = {k: torch.from_numpy(v).cuda() for k, v in b['gt'].items()}
bb.SIDE_STREAM, ops.WGRAD_SIDE = False, False
tr.step(*inp, gt)
ops.PROFILE, ops.PROFILE_WGRAD = [], []
ops.PROFILE_WINO, ops.PROFILE_WINO_WGRAD, ops.PROFILE_THIN, ops.PROFILE_WINO2D, ops.PROFILE_WINO2D_GEMM = [], [], [], [], []
tr.step(*inp, gt)
torch.cuda.synchronize()
for name, recs in (('k_gather_gemm', ops.PROFILE), ('k_gather_wgrad', ops.PROFILE_WGRAD), ('thin / c4 / n4', ops.PROFILE_THIN)):
    agg = collections.OrderedDict()
    for r in recs:
        shape = r[3] if len(r) > 3 else None
        ms = r[0].elapsed_time(r[1])
        a = agg.setdefault(shape, [0, 0.0, 0.0])
        a[0] += 1; a[1] += ms; a[2] += r[2]
    tot = sum(v[1] for v in agg.values())
    print('== %s: %d launches, %.2f ms per step' % (name, sum(v[0] for v in agg.values()), tot))
    print('   (mode, M, N, T, C)                      launches      ms   TFLOP/s')
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print('   %-40s %5d %8.3f %8.1f' % (k, v[0], v[1], v[2] / v[1] / 1e9))
