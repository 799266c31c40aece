"""the partitioned lattice build on a spatially COHERENT sweep (ground plane + walls: neighbouring beams hit continuous
surfaces, so lattice cells near the sensor hold hundreds of points) - how long do the vertex lists get, do buckets overflow,
how fast is the build compared with the random-range bench scene (debug aid)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from efgh_amd import lattice, _C, synthetic as syn


coherent_sweep = syn.coherent_sweep


SC = (1.0, 0.75, 0.5, 0.25, 0.125)
B = 8
for name, gen in (('random ranges (bench scene)', lambda b: syn.lidar_sweep(131072, b)), ('coherent scene', lambda b: coherent_sweep(131072, b))):
    pc = torch.from_numpy(np.stack([gen(b) for b in range(B)])).cuda()
    lattice._SIZES.clear()
    for _ in range(3):
        lv = lattice.build_pyramid_batched(pc, SC)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        lv = lattice.build_pyramid_batched(pc, SC)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(name, ': %.0f us per pyramid' % (dt * 1e6))
    for l, d in enumerate(lv):
        ln = d.vseg[:d.H, 1]
        print('   level %d: n=%d H=%d mode=%s max list %d, mean %.1f, lists > 2048: %d' % (
            l, d.n_in, d.H, d._mode, int(ln.max()), float(ln.float().mean()), int((ln > 2048).sum())))
    # splat on this scene (long lists are walked by ONE lane group per vertex)
    from efgh_amd import ops
    cfs = [32, 32, 64, 128, 256]
    for l, d in enumerate(lv):
        feat = torch.randn(d.n_in, cfs[l], device='cuda')
        for _ in range(3):
            sp, w = ops.splat_fwd(d, feat, cfs[l])
        g = torch.randn(d.H, cfs[l] + 4, device='cuda')
        gf = torch.empty(d.n_in, cfs[l], device='cuda')
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        for _ in range(10):
            sp, w = ops.splat_fwd(d, feat, cfs[l])
        e1.record()
        for _ in range(10):
            ops.splat_bwd(d, g, w, cfs[l], gf)
        e2.record()
        torch.cuda.synchronize()
        print('   level %d splat %.1f us, splat bwd %.1f us' % (l, e0.elapsed_time(e1) * 100, e1.elapsed_time(e2) * 100))
