"""thin (<= 4 channel) layers: VALU kernels vs the MFMA gather-GEMM / wgrad on the same shapes (GPU box)"""
import sys
sys.path.insert(0, '/root/repo')
import torch
import torch.nn as nn
from efgh_amd import ops
from efgh_amd.nets import layers as L

torch.set_grad_enabled(False)


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (B, H, W, ci, co) in [(4, 384, 5119, 3, 64), (4, 384, 1280, 3, 64), (8, 384, 5119, 3, 64)]:
    torch.manual_seed(0)
    conv = nn.Conv2d(ci, co, 3, 1, 1, bias=False).cuda()
    x = torch.randn(B, H, W, 4, device='cuda'); x[..., 3] = 0
    g = torch.randn(B, H, W, co, device='cuda')
    geom = (B, H, W, H, W, 1, 1, [t // 3 - 1 for t in range(9)], [t % 3 - 1 for t in range(9)], H, W, 1, 1, 0, 0)
    ctx = L.Ctx(False)
    out = {}
    for thin in (True, False):
        ops.USE_THIN = thin
        tf = timeit(lambda: L.conv2d(ctx, x, conv, None))
        y = L.conv2d(ctx, x, conv, None)
        dWp = torch.empty((co, 9, 4), device='cuda')
        tw = timeit(lambda: ops.gather_wgrad(x, 4, 4, 9, co, B * H * W, g, co, dWp, mode=1, geom=geom))
        out[thin] = (tf, tw, y, dWp.clone())
    gb_f = B * H * W * (co + 4) * 4 / 1e9
    print('B%d %dx%d %d->%d  fwd: thin %.3f ms  mfma %.3f ms (HBM floor %.2f ms) | wgrad: thin %.3f ms  mfma %.3f ms  | diff %.1e %.1e' % (
        B, H, W, ci, co, out[True][0], out[False][0], gb_f / 4.5, out[True][1], out[False][1],
        (out[True][2] - out[False][2]).abs().max().item(), ((out[True][3] - out[False][3]).norm() / out[False][3].norm()).item()))
