"""k_wino43 at the training-step shapes of the 64- / 128-channel layers (batch 8), one library per process: run once per build
(EFGH_LIB=<path to libefgh_hip.so>) and compare.  python tools/bench_wino_ab.py [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from efgh_amd import ops
from efgh_amd.nets import layers as L

torch.set_grad_enabled(False)
ops.USE_WINO2D = False
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tot = 0.0
for (B, H, W, ci, co) in [(8, 384, 1280, 64, 64), (8, 192, 640, 128, 128), (8, 192, 640, 64, 128), (2, 384, 5119, 64, 64), (8, 192, 2559, 64, 128)]:
    torch.manual_seed(0)
    conv = nn.Conv2d(ci, co, 3, 1, 1, bias=False).cuda()
    x = torch.randn(B, H, W, ci, device='cuda').clamp_min(0)
    ctx = L.Ctx(False)
    for _ in range(2):
        y = L.conv2d(ctx, x, conv, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        y = L.conv2d(ctx, x, conv, None)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tot += ms
    fl = 2.0 * B * H * W * co * ci * 9
    print('B%d %dx%d %d->%d : %.3f ms  %.1f TF algorithmic (%.1f executed)  checksum %r' % (
        B, H, W, ci, co, ms, fl / ms / 1e9, fl / ms / 2e9, float(y.double().sum())))
print('sum %.3f ms' % tot)
