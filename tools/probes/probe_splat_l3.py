"""level-3 splat (C = 132) timing anomaly probe: per-call event timing of back-to-back launches"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from efgh_amd import lattice, ops, synthetic as syn
SC = (1.0, 0.75, 0.5, 0.25, 0.125)
pc = torch.from_numpy(np.stack([syn.lidar_sweep(131072, b) for b in range(8)])).cuda()
for _ in range(2):
    lv = lattice.build_pyramid_batched(pc, SC)
for l, cf in ((2, 64), (3, 128), (4, 256)):
    d = lv[l]
    feat = torch.randn(d.n_in, cf, device='cuda')
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(13)]
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(12):
        ops.splat_fwd(d, feat, cf)
        ev[i + 1].record()
    torch.cuda.synchronize()
    print('level', l, 'n', d.n_in, 'H', d.H, [round(ev[i].elapsed_time(ev[i + 1]) * 1e3) for i in range(12)])
    # with preallocated outputs: is it the allocator?
    import time
    t0 = time.perf_counter()
    for i in range(12):
        ops.splat_fwd(d, feat, cf)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print('   host time per call %.1f us' % ((t1 - t0) / 12 * 1e6))
