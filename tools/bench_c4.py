"""micro-benchmark: the 4-channel input layers of config S at batch 8 on efgh_c4_conv3x3 / efgh_c4_wgrad vs the generic
implicit-GEMM kernels (GPU box)"""
import sys
sys.path.insert(0, '/root/repo')
import torch
import torch.nn as nn
from efgh_amd import ops
from efgh_amd.nets import layers as L

torch.set_grad_enabled(False)
shapes = [  # (B, H, W, Cout, stride)
    (8, 384, 5120, 64, 1), (8, 384, 1280, 64, 1), (8, 768, 2560, 32, 2), (8, 768, 2560, 128, 2),
]


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


for (B, H, W, co, s) in shapes:
    torch.manual_seed(0)
    conv = nn.Conv2d(4, co, 3, s, 1, bias=False).cuda()
    bn = nn.BatchNorm2d(co).cuda().train()
    x = torch.randn(B, H, W, 4, device='cuda')
    ho, wo = (H - 1) // s + 1, (W - 1) // s + 1
    g = torch.randn(B, ho, wo, co, device='cuda')
    taps = ([t // 3 - 1 for t in range(9)], [t % 3 - 1 for t in range(9)])
    geom = (B, H, W, ho, wo, s, s, taps[0], taps[1], ho, wo, 1, 1, 0, 0)
    M = B * ho * wo
    dWp = torch.empty((co, 9, 4), device='cuda')
    res = {}
    for c4 in (False, True):
        ops.USE_C4 = c4
        tf = timeit(lambda: L.conv2d(L.Ctx(False), x, conv, None))
        tw = timeit(lambda: ops.gather_wgrad(x, 4, 4, 9, co, M, g, co, dWp, mode=1, geom=geom))
        res[c4] = (tf, tw, dWp.clone())
    ops.USE_C4 = True
    err = (res[True][2] - res[False][2]).norm().item() / res[False][2].norm().item()
    ob = M * co * 4 / 1e9
    print('B%d %dx%d 4->%d s%d : fwd generic %.3f ms (%.2f TB/s out) | c4 %.3f ms (%.2f TB/s)  x%.2f || wgrad generic %.3f ms | c4 %.3f ms '
          '(%.2f TB/s in) x%.2f  rel diff %.1e' % (B, H, W, co, s, res[False][0], ob / res[False][0], res[True][0], ob / res[True][0],
                                                 res[False][0] / res[True][0], res[False][1], res[True][1], ob / res[True][1],
                                                 res[False][1] / res[True][1], err))
