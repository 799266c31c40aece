"""HBM rate of the 2-D Winograd transform passes (k_w2_input / k_w2_output / k_w2_dy) at the layer shapes of a batch-8 training step.
GPU box: python tools/bench_w2.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from efgh_amd import _C, ops
from efgh_amd._C import c_int32, c_int64, ptr

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


for B, H, W, C in [(8, 96, 320, 256), (8, 48, 160, 512), (8, 96, 1280, 256), (8, 48, 640, 512), (8, 192, 640, 128), (4, 96, 320, 256)]:
    st = _C.stream_ptr()
    T = L.efgh_wino2d_tiles(c_int32(B), c_int32(H), c_int32(W))
    x = torch.randn(B, H, W, C, device='cuda')
    V = torch.empty(T, 36, C, device='cuda')
    y = torch.empty(B, H, W, C, device='cuda')
    rows = L.efgh_wino2d_stats_rows(c_int32(B), c_int32(H), c_int32(W), c_int32(C))
    stats = torch.empty(rows, 2, C, device='cuda')
    ti = timeit(lambda: _C.check(L.efgh_wino2d_input(ptr(x), c_int64(C), c_int32(C), c_int32(B), c_int32(H), c_int32(W), ptr(V), st)))
    td = timeit(lambda: _C.check(L.efgh_wino2d_dy(ptr(x), c_int64(C), c_int32(C), c_int32(B), c_int32(H), c_int32(W), ptr(V), st)))

    def out(with_stats, act, res):
        d = _C.GemmDesc()
        d.mode, d.B, d.Hin, d.Win, d.Hv, d.Wv, d.Ho, d.Wo = 1, B, H, W, H, W, H, W
        d.sh = d.sw = d.osh = d.osw = 1
        d.C, d.T, d.N, d.M = C, 9, C, B * H * W
        for t in range(9):
            d.dh[t], d.dw[t] = t // 3 - 1, t % 3 - 1
        d.out, d.ldo, d.act = y.data_ptr(), C, act
        if with_stats:
            d.stats = stats.data_ptr()
        if res:
            d.residual, d.ldr = x.data_ptr(), C
        return timeit(lambda: _C.check(L.efgh_wino2d_output(ptr(V), ctypes.byref(d), st)))
    to_plain, to_stats, to_res = out(False, 1, False), out(True, 0, False), out(False, 1, True)
    gb = x.numel() * 4 / 1e9
    print('B=%d %dx%d C=%d (%.3f GB): input %.1f us %.2f TB/s | dy %.1f us %.2f TB/s | output(relu) %.1f us %.2f TB/s | output(stats) %.1f us | '
          'output(res+relu) %.1f us' % (B, H, W, C, gb, ti * 1e3, 3.25 * gb / ti, td * 1e3, 3.25 * gb / td, to_plain * 1e3, 3.25 * gb / to_plain,
                                        to_stats * 1e3, to_res * 1e3))
