"""k_wino43 on one layer shape (64->64, 128->128) at equal pixel counts but different row lengths (L2 locality probe)"""
import sys
sys.path.insert(0, '/root/repo')
import torch, torch.nn as nn
from efgh_amd import ops
from efgh_amd.nets import layers as L
torch.set_grad_enabled(False)
for ci in (64, 128):
    for (B, H, W) in [(4, 1536, 320), (4, 384, 1280), (4, 96, 5120), (4, 24, 20480), (1, 384, 5120), (16, 96, 1280)]:
        if ci == 128: H //= 2; W //= 2
        conv = nn.Conv2d(ci, ci, 3, 1, 1, bias=False).cuda()
        x = torch.randn(B, H, W, ci, device='cuda').clamp_min(0)
        ctx = L.Ctx(False)
        for _ in range(2): y = L.conv2d(ctx, x, conv, None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): y = L.conv2d(ctx, x, conv, None)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print('%d->%d B%d %dx%d : %.3f ms %.1f TF' % (ci, ci, B, H, W, ms, 2.0 * B * H * W * ci * ci * 9 / ms / 1e9))
