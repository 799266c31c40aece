"""micro-benchmark: Winograd F(4,3) kernel vs the direct gather-GEMM on the 3x3 layers of config S (GPU box)"""
import sys
sys.path.insert(0, '/root/repo')
import torch
import torch.nn as nn
from efgh_amd import ops
from efgh_amd.nets import layers as L

torch.set_grad_enabled(False)
import os
ops.USE_WINO2D = os.environ.get('EFGH_WINO2D', '1') != '0'
shapes = [  # (B, H, W, Cin, Cout)
    (4, 384, 1280, 64, 64), (4, 192, 640, 128, 128), (4, 96, 320, 256, 256), (4, 48, 160, 512, 512),
    (4, 24, 80, 512, 512), (1, 384, 5119, 64, 64), (4, 96, 1279, 256, 256), (4, 192, 640, 64, 128),
]
for (B, H, W, ci, co) in shapes:
    torch.manual_seed(0)
    conv = nn.Conv2d(ci, co, 3, 1, 1, bias=False).cuda()
    x = torch.randn(B, H, W, ci, device='cuda').clamp_min(0)
    ctx = L.Ctx(False)
    res = {}
    for wino in (False, True):
        ops.USE_WINO = wino
        for _ in range(2):
            y = L.conv2d(ctx, x, conv, None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 5
        for _ in range(n):
            y = L.conv2d(ctx, x, conv, None)
        e1.record(); torch.cuda.synchronize()
        res[wino] = (e0.elapsed_time(e1) / n, y)
    fl = 2.0 * B * H * W * co * ci * 9
    err = (res[True][1] - res[False][1]).abs().max().item() / res[False][1].abs().max().item()
    print('B%d %dx%d %d->%d : direct %.3f ms %.1f TF | wino %.3f ms %.1f TF (algorithmic)  x%.2f  rel diff %.1e' % (
        B, H, W, ci, co, res[False][0], fl / res[False][0] / 1e9, res[True][0], fl / res[True][0] / 1e9,
        res[False][0] / res[True][0], err))

print('--- weight gradient')
for (B, H, W, ci, co) in shapes:
    x = torch.randn(B, H, W, ci, device='cuda').clamp_min(0)
    g = torch.randn(B, H, W, co, device='cuda')
    geom = (B, H, W, H, W, 1, 1, [t // 3 - 1 for t in range(9)], [t % 3 - 1 for t in range(9)], H, W, 1, 1, 0, 0)
    res = {}
    for wino in (False, True):
        ops.USE_WINO_WGRAD = wino
        dWp = torch.empty((co, 9, ci), device='cuda')
        for _ in range(2):
            ops.gather_wgrad(x, ci, ci, 9, co, B * H * W, g, co, dWp, mode=1, geom=geom)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 5
        for _ in range(n):
            ops.gather_wgrad(x, ci, ci, 9, co, B * H * W, g, co, dWp, mode=1, geom=geom)
        e1.record(); torch.cuda.synchronize()
        res[wino] = (e0.elapsed_time(e1) / n, dWp)
    fl = 2.0 * B * H * W * co * ci * 9
    err = (res[True][1] - res[False][1]).norm().item() / res[False][1].norm().item()
    print('B%d %dx%d %d->%d : direct %.3f ms %.1f TF | wino %.3f ms %.1f TF (algorithmic)  x%.2f  rel diff %.1e' % (
        B, H, W, ci, co, res[False][0], fl / res[False][0] / 1e9, res[True][0], fl / res[True][0] / 1e9,
        res[False][0] / res[True][0], err))
