"""the 64 -> 4 channel 3x3 layer (data gradient of the range trunk's input convolution, 15.7 M pixels at batch 8): k_n4_conv3x3_c64"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from efgh_amd import ops
from efgh_amd.nets import layers as L
B, H, W = 8, 384, 5119
conv = nn.Conv2d(64, 3, 3, 1, 1, bias=False).cuda()
x = torch.randn(B, H, W, 64, device='cuda')
ctx = L.Ctx(False)
with torch.no_grad():
    for _ in range(2):
        y = L.conv2d(ctx, x, conv, None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        y = L.conv2d(ctx, x, conv, None)
    e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
by = B * H * W * 68 * 4.0
print('B%d %dx%d 64->4: %.3f ms (%.2f TB/s on %.2f GB, %.1f TFLOP/s algorithmic)' % (B, H, W, ms, by / ms / 1e9, by / 1e9, 2.0 * B * H * W * 64 * 9 * 4 / ms / 1e9))
