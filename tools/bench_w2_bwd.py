"""round 6: the fused transforms against the passes they replace, at the layer shapes of a batch-8 training step.
   forward:  efgh_scale_shift_act + efgh_wino2d_input        vs  efgh_wino2d_input_act
   backward: efgh_act_bn_bwd_apply + efgh_wino2d_input + efgh_wino2d_dy   vs  efgh_wino2d_bwd_transforms
GPU box: python tools/bench_w2_bwd.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from efgh_amd import _C, ops
from efgh_amd._C import c_float, c_int32, c_int64, ptr

torch.set_grad_enabled(False)
L = _C.lib()


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for B, H, W, C in [(8, 96, 320, 256), (8, 48, 160, 512), (8, 96, 1280, 256), (8, 48, 640, 512), (8, 192, 640, 128)]:
    st = _C.stream_ptr()
    T = L.efgh_wino2d_tiles(c_int32(B), c_int32(H), c_int32(W))
    M = B * H * W
    raw = torch.randn(B, H, W, C, device='cuda')
    dy = torch.randn(B, H, W, C, device='cuda')
    y = torch.empty_like(raw)
    draw = torch.empty_like(raw)
    dres = torch.empty_like(raw)
    V = torch.empty(T, 36, C, device='cuda')
    G = torch.empty(T, 36, C, device='cuda')
    sc, sf = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')
    mean, invstd, coef = torch.randn(C, device='cuda') * 0.1, torch.rand(C, device='cuda') + 0.5, torch.rand(C, device='cuda') + 0.5
    m1, m2 = torch.randn(C, device='cuda', dtype=torch.float64) * 0.01, torch.randn(C, device='cuda', dtype=torch.float64) * 0.01
    bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (M * C // 32,), device='cuda', dtype=torch.int32)
    t_ssa = timeit(lambda: ops.scale_shift_act(raw, C, sc, sf, y, C, M, C, 1, 0.0))
    t_in = timeit(lambda: ops.wino2d_input(y, 0, C, C, B, H, W, V))
    t_ina = timeit(lambda: ops.wino2d_input(raw, 0, C, C, B, H, W, V, ops.LazyAct(sc, sf, 1, 0.0)))
    t_app = timeit(lambda: ops.act_bn_bwd_apply(dy, C, None, C, raw, C, mean, invstd, coef, m1, m2, M, C, 1, 0.0, draw, C, None, C, pscale=sc, pshift=sf))
    t_appb = timeit(lambda: ops.act_bn_bwd_apply(dy, C, bits, 0, raw, C, mean, invstd, coef, m1, m2, M, C, 1, 0.0, draw, C, dres, C))
    t_dy = timeit(lambda: _C.check(L.efgh_wino2d_dy(ptr(draw), c_int64(C), c_int32(C), c_int32(B), c_int32(H), c_int32(W), ptr(G), st)))

    def fused(b):
        _C.check(L.efgh_wino2d_bwd_transforms(ptr(dy), c_int64(C), ptr(raw), c_int64(C), ptr(bits if b else None), ptr(None if b else sc),
                                              ptr(None if b else sf), ptr(mean), ptr(invstd), ptr(coef), ptr(m1), ptr(m2), c_int32(C), c_int32(B),
                                              c_int32(H), c_int32(W), c_int32(1), c_float(0.0), ptr(V), ptr(G), ptr(dres if b else None), c_int64(C), st))
    t_f, t_fb = timeit(lambda: fused(False)), timeit(lambda: fused(True))
    # pooled layer (vgg.py:69-83): reduce stays; apply + input + dy vs the pooled fused transforms
    dyp = torch.randn(B, H // 2, W // 2, C, device='cuda')
    t_pool3 = timeit(lambda: ops.pool_bn_bwd(dyp, raw, mean, invstd, coef, sc, sf, 1, 0.0))
    t_poolf = timeit(lambda: ops.pool_bn_bwd(dyp, raw, mean, invstd, coef, sc, sf, 1, 0.0, transforms=True))
    gb = raw.numel() * 4 / 1e9
    print('   pooled: reduce + apply %.0f us (+ input %.0f + dy %.0f = %.0f us)  vs  reduce + fused transforms %.0f us'
          % (t_pool3 * 1e3, t_in * 1e3, t_dy * 1e3, (t_pool3 + t_in + t_dy) * 1e3, t_poolf * 1e3))
    print('B=%d %dx%d C=%d (%.3f GB) fwd: act %.0f + input %.0f = %.0f us  vs input_act %.0f us (%.2f TB/s) | bwd: apply %.0f + input %.0f + dy %.0f = %.0f us vs '
          'fused %.0f us (%.2f TB/s) | residual: apply %.0f + .. = %.0f us vs fused %.0f us (%.2f TB/s)'
          % (B, H, W, C, gb, t_ssa * 1e3, t_in * 1e3, (t_ssa + t_in) * 1e3, t_ina * 1e3, 3.25 * gb / t_ina, t_app * 1e3, t_in * 1e3, t_dy * 1e3,
             (t_app + t_in + t_dy) * 1e3, t_f * 1e3, 6.5 * gb / t_f, t_appb * 1e3, (t_appb + t_in + t_dy) * 1e3, t_fb * 1e3, 7.5 * gb / t_fb))
