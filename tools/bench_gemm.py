"""micro-benchmark of the gather-GEMM on representative conv shapes (run on the GPU box)"""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
import torch.nn as nn
from efgh_amd import ops
from efgh_amd.nets import layers as L

torch.set_grad_enabled(False)
shapes = [  # (B, H, W, Cin, Cout, k, s)
    (4, 384, 1280, 64, 64, 3, 1), (4, 192, 640, 128, 128, 3, 1), (4, 96, 320, 256, 256, 3, 1),
    (4, 48, 160, 512, 512, 3, 1), (4, 96, 1279, 256, 256, 3, 1), (4, 384, 1280, 64, 128, 3, 2),
]
for (B, H, W, ci, co, k, s) in shapes:
    torch.manual_seed(0)
    conv = nn.Conv2d(ci, co, k, s, k // 2, bias=False).cuda()
    x = torch.randn(B, H, W, ci, device='cuda')
    ctx = L.Ctx(False)
    for _ in range(2):
        y = L.conv2d(ctx, x, conv, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 5
    for _ in range(n):
        y = L.conv2d(ctx, x, conv, None)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * co * ci * k * k
    print('B%d %dx%d %d->%d k%d s%d : %.3f ms  %.1f TFLOP/s' % (B, H, W, ci, co, k, s, ms, fl / ms / 1e9))
