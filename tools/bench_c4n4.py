"""the 4 -> 4 channel 3x3 stencil layers behind G's transposed heads at full raw resolution (batch 8): forward and weight gradient"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from efgh_amd import ops
from efgh_amd.nets import layers as L
B, H, W = 8, 768, 2560
conv = nn.Conv2d(2, 2, 3, 1, 1, bias=True).cuda()
x = torch.randn(B, H, W, 4, device='cuda')
g = torch.randn(B, H, W, 4, device='cuda')
geom = (B, H, W, H, W, 1, 1, [t // 3 - 1 for t in range(9)], [t % 3 - 1 for t in range(9)], H, W, 1, 1, 0, 0)
ctx = L.Ctx(False)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
with torch.no_grad():
    for _ in range(2):
        y = L.conv2d(ctx, x, conv, None)
    ev[0].record()
    for _ in range(10):
        y = L.conv2d(ctx, x, conv, None)
    ev[1].record()
    dWp = torch.empty((4, 9, 4), device='cuda')
    for _ in range(2):
        ops.gather_wgrad(x, 4, 4, 9, 4, B * H * W, g, 4, dWp, mode=1, geom=geom)
    ev[2].record()
    for _ in range(10):
        ops.gather_wgrad(x, 4, 4, 9, 4, B * H * W, g, 4, dWp, mode=1, geom=geom)
    ev[3].record()
    torch.cuda.synchronize()
by = B * H * W * 32.0
f, w = ev[0].elapsed_time(ev[1]) / 10, ev[2].elapsed_time(ev[3]) / 10
print('B%d %dx%d 4->4: conv %.3f ms (%.2f TB/s), wgrad %.3f ms (%.2f TB/s)' % (B, H, W, f, by / f / 1e9, w, by / w / 1e9))
