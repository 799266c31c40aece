"""level-0 timing of the partitioned lattice build at the bench sizes for several (buckets, slots) plans (debug aid);
run under `rocprofv3 --kernel-trace` for the per-kernel split"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from efgh_amd import lattice, _C, synthetic as syn
SCALES = (1.0, 0.75, 0.5, 0.25, 0.125)
pc = torch.from_numpy(np.stack([syn.lidar_sweep(131072, b) for b in range(8)])).cuda()
L = _C.lib()
for _ in range(3):
    lv = lattice.build_pyramid_batched(pc, SCALES)
torch.cuda.synchronize()
n = pc.shape[0] * pc.shape[2]
pts = pc.permute(1, 0, 2).reshape(3, n).contiguous()
st = _C.stream_ptr()
modes = [lattice._plan(L, n, 313000)]
for dbg in [int(x) for x in (sys.argv[1:] or ['0'])]:
    for mode in modes:
        ts = []
        for it in range(6):
            d = lattice._level_arrays(L, pc.device, n, 313000, 8, mode)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lattice._launch_build(L, d, pts, n, None, None, pc.shape[2], 8, 1.0, st)
            lattice._launch_neighbors(L, d, 8, 313000, st)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print('dbg', dbg, 'mode', mode, 'level-0 build + neighbours: %.1f us (min of %s)' % (min(ts), [round(t) for t in ts]),
              'info', d.info[:3].tolist())
