"""the small-channel kernels (smallc.hip) against the generic implicit-GEMM kernels on F's up-sampling shapes and the 1x1 64 -> 32 layer
(and its data gradient 32 -> 64) at full resolution (batch 8)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from efgh_amd import ops
from efgh_amd.nets import layers as L
shapes = [(8, 376, 1281, 16, 16, 3), (8, 190, 637, 32, 32, 3), (8, 188, 640, 16, 16, 3), (8, 94, 322, 32, 32, 3),
          (8, 384, 1280, 64, 32, 1), (8, 384, 1280, 32, 64, 1)]
for (B, H, W, ci, co, k) in shapes:
    torch.manual_seed(0)
    conv = nn.Conv2d(ci, co, k, 1, k // 2, bias=False).cuda()
    T = k * k
    x = torch.randn(B, H, W, ci, device='cuda')
    g = torch.randn(B, H, W, co, device='cuda')
    taps = ([t // 3 - 1 for t in range(9)], [t % 3 - 1 for t in range(9)]) if k == 3 else ([0], [0])
    geom = (B, H, W, H, W, 1, 1, taps[0], taps[1], H, W, 1, 1, 0, 0)
    res = {}
    for sc in (False, True):
        ops.USE_SMALLC = sc
        ctx = L.Ctx(False)
        with torch.no_grad():
            for _ in range(3):
                y = L.conv2d(ctx, x, conv, None)
            e0, e1, e2, e3 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
            e0.record()
            for _ in range(10):
                y = L.conv2d(ctx, x, conv, None)
            e1.record()
            dWp = torch.empty((co, T, ci), device='cuda')
            for _ in range(2):
                ops.gather_wgrad(x, ci, ci, T, co, B * H * W, g, co, dWp, mode=1, geom=geom)
            e2.record()
            for _ in range(10):
                ops.gather_wgrad(x, ci, ci, T, co, B * H * W, g, co, dWp, mode=1, geom=geom)
            e3.record()
            torch.cuda.synchronize()
        res[sc] = (e0.elapsed_time(e1) / 10, e2.elapsed_time(e3) / 10, y, dWp.clone())
    ops.USE_SMALLC = True
    fl = 2.0 * B * H * W * co * ci * T
    by = B * H * W * (ci + co) * 4.0
    print('B%d %dx%d %d->%d (%.1f GFLOP, %.0f MB): conv %.3f -> %.3f ms (%.1f TF, %.2f TB/s), wgrad %.3f -> %.3f ms (%.1f TF) | rel diff %.1e / %.1e' % (
        B, H, W, ci, co, fl / 1e9, by / 1e6, res[False][0], res[True][0], fl / res[True][0] / 1e9, by / res[True][0] / 1e9,
        res[False][1], res[True][1], fl / res[True][1] / 1e9,
        float((res[True][2] - res[False][2]).abs().max() / res[False][2].abs().max()),
        float((res[True][3] - res[False][3]).norm() / res[False][3].norm())))
