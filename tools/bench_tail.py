"""the pyramid build with the one-launch tail (k_lat_tail) against the per-level kernels, GPU time between events, several batch
sizes: python tools/bench_tail.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from efgh_amd import lattice, synthetic as syn

SCALES = (1.0, 0.75, 0.5, 0.25, 0.125)
for B in (1, 4, 8, 16, 32, 64):
    pc = torch.from_numpy(np.stack([syn.lidar_sweep(131072, b) for b in range(B)])).cuda()
    res = {}
    for tail in (False, True):
        lattice.TAIL = tail
        lattice._SIZES.clear(); lattice._PER_SAMPLE.clear(); lattice._NO_TAIL.clear()
        for _ in range(3):
            lv = lattice.build_pyramid_batched(pc, SCALES)
        torch.cuda.synchronize()
        lattice.PROFILE = []
        for _ in range(6):
            lv = lattice.build_pyramid_batched(pc, SCALES)
        torch.cuda.synchronize()
        ms = sorted(p[0].elapsed_time(p[1]) for p in lattice.PROFILE)[len(lattice.PROFILE) // 2]
        lattice.PROFILE = None
        res[tail] = (ms, [x._mode[0] for x in lv])
    print('B=%2d  per-level kernels %.3f ms | tail %.3f ms %s' % (B, res[False][0], res[True][0], res[True][1]), flush=True)
