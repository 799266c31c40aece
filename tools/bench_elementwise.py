"""HBM rate of the BatchNorm/activation streaming passes (GPU box)"""
import sys
sys.path.insert(0, '/root/repo')
import torch
from efgh_amd import ops

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


for M, C in [(8 * 384 * 5119, 64), (8 * 384 * 1280, 64), (8 * 96 * 1279, 256), (8 * 24 * 319, 512)]:
    x = torch.randn(M, C, device='cuda'); y = torch.empty_like(x); dy = torch.randn(M, C, device='cuda')
    sc = torch.rand(C, device='cuda') + 0.5; sh = torch.randn(C, device='cuda')
    mean = torch.randn(C, device='cuda') * 0.1; invstd = torch.rand(C, device='cuda') + 0.5
    t = timeit(lambda: ops.scale_shift_act(x, C, sc, sh, y, C, M, C, ops.ACT_RELU))
    gb = M * C * 4 / 1e9
    G = ops.bwd_groups(M)
    part = torch.empty((G, 2, C), device='cuda', dtype=torch.float64); s1 = torch.empty(C, device='cuda'); s2 = torch.empty(C, device='cuda')
    m1 = torch.empty(C, device='cuda', dtype=torch.float64); m2 = torch.empty(C, device='cuda', dtype=torch.float64)
    tr = timeit(lambda: ops.act_bn_bwd_reduce(dy, C, None, C, x, C, mean, invstd, M, C, ops.ACT_RELU, 0.0, part, s1, s2, m1, m2,
                                              pscale=sc, pshift=sh))
    draw = torch.empty_like(x)
    ta = timeit(lambda: ops.act_bn_bwd_apply(dy, C, None, C, x, C, mean, invstd, sc, m1, m2, M, C, ops.ACT_RELU, 0.0, draw, C,
                                             None, C, pscale=sc, pshift=sh))
    print('M=%d C=%d (%.2f GB/tensor): scale_shift_act %.3f ms %.2f TB/s | bwd_reduce %.3f ms %.2f TB/s | bwd_apply %.3f ms %.2f TB/s' % (
        M, C, gb, t, 2 * gb / t, tr, 2 * gb / tr, ta, 3 * gb / ta))
