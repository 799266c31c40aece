import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
from efgh_amd import ops
from efgh_amd.nets import layers as L, fn as FN
ops.TLS.train_step = True
c1, b1 = nn.Conv2d(128, 128, 3, 1, 1, bias=False).cuda(), nn.BatchNorm2d(128).cuda()
c2, b2 = nn.Conv2d(128, 128, 3, 1, 1, bias=False).cuda(), nn.BatchNorm2d(128).cuda()
x = torch.randn(2, 24, 40, 128, device='cuda', requires_grad=True)
ctx = L.Ctx(True)
print('consumer ok', L.lazy_consumer_ok(ctx, c2, 2, 24, 40), ctx.grad, ctx.train, ops.LAZY_ACT)
y1 = L.conv2d(ctx, x, c1, b1, L.ACT_RELU, defer_act=True)
print('lazy attr', getattr(y1, '_efgh_lazy', None))
y1, al = L.conv2d(ctx, x, c1, b1, L.ACT_RELU, defer_act=True, skip_out=True)
print('lazy attr (skip_out)', getattr(y1, '_efgh_lazy', None))
y2 = L.conv2d(ctx, y1, c2, b2, L.ACT_RELU, residual=al)
print(ops.LAZY_HITS)
y2.sum().backward()
print(ops.LAZY_HITS)
