"""BCL index + splat pipeline alone at the bench sizes (batch B, 131072 points): wall time per pyramid build, per splat level.
Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from efgh_amd import lattice, ops, synthetic as syn

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--points', type=int, default=131072)
ap.add_argument('--iters', type=int, default=20)
a = ap.parse_args()
SCALES = (1.0, 0.75, 0.5, 0.25, 0.125)
pc = torch.from_numpy(np.stack([syn.lidar_sweep(a.points, b) for b in range(a.batch)])).cuda()
for _ in range(3):
    lv = lattice.build_pyramid_batched(pc, SCALES)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    lv = lattice.build_pyramid_batched(pc, SCALES)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
by = sum(d.n_in * 76.0 + d.H * 136.0 for d in lv)
print('pyramid build: %.1f us wall per build (B=%d), H=%s, %.1f MB algorithmic -> %.1f GB/s' %
      (dt * 1e6, a.batch, [d.H for d in lv], by / 1e6, by / dt / 1e9))
cfs = [32, 32, 64, 128, 256]
for l, d in enumerate(lv):
    feat = torch.randn(d.n_in, cfs[l], device='cuda')
    for _ in range(5):
        s, w = ops.splat_fwd(d, feat, cfs[l])
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        s, w = ops.splat_fwd(d, feat, cfs[l])
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        s, w = ops.splat_fwd(d, feat, cfs[l])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.iters * 1e3
    C = cfs[l] + 4
    by = d.n_in * (4 * C + 48.0) + d.H * (4 * C + 4.0)
    g = torch.randn(d.H, C, device='cuda')
    gf = torch.empty(d.n_in, cfs[l], device='cuda')
    e0.record()
    for _ in range(a.iters):
        ops.splat_bwd(d, g, w, cfs[l], gf)
    e1.record()
    torch.cuda.synchronize()
    usb = e0.elapsed_time(e1) / a.iters * 1e3
    src = torch.randn(d.H, 15 * C, device='cuda')
    e0.record()
    for _ in range(a.iters):
        ops.neighbor_gather_adjoint(d, src, C)
    e1.record()
    torch.cuda.synchronize()
    usg = e0.elapsed_time(e1) / a.iters * 1e3
    print('level %d: n=%d H=%d C=%d  splat %.1f us [best single %.1f] (%.0f GB/s algorithmic)  splat bwd %.1f us (%.0f GB/s)  gather adjoint %.1f us' %
          (l, d.n_in, d.H, C, us, best, by / us / 1e3, usb, by / usb / 1e3, usg))
