"""one 3x3 layer through the Winograd forward kernel, N launches (for rocprofv3 --pmc passes):
python tools/bench_wino_one.py B H W Cin Cout [iters]"""
import sys
sys.path.insert(0, '/root/repo')
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from efgh_amd import ops
from efgh_amd.nets import layers as L

torch.set_grad_enabled(False)
B, H, W, ci, co = [int(v) for v in sys.argv[1:6]]
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 20
conv = nn.Conv2d(ci, co, 3, 1, 1, bias=False).cuda()
x = torch.randn(B, H, W, ci, device='cuda').clamp_min(0)
ctx = L.Ctx(False)
for _ in range(3):
    y = L.conv2d(ctx, x, conv, None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    y = L.conv2d(ctx, x, conv, None)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
fl = 2.0 * B * H * W * co * ci * 9
print('B%d %dx%d %d->%d: %.3f ms, %.1f TF algorithmic, %.1f TF executed' % (B, H, W, ci, co, ms, fl / ms / 1e9, fl / ms / 2e9))
