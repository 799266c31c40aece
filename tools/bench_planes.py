"""The 36 alpha planes of the 2-D Winograd layers at their training-step shapes (batch 8, config S), in the tile-major layout the
layers use ([tiles][36][C]): LDS-DMA staged kernels (planes.hip, 2- and 3-slot rings) against k_gather_gemm<0> / k_gather_wgrad<0>,
alternating in one process; results compared bit for bit.  Run on the GPU box:  python tools/bench_planes.py [reps]"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efgh_amd import _C, ops
from efgh_amd._C import c_int32, c_int64, ptr

torch.set_grad_enabled(False)
L = _C.lib()
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 5


def desc(V, W, out, T2, C, N):
    g = _C.GemmDesc()
    g.A, g.lda, g.C, g.T, g.mode = V.data_ptr(), 36 * C, C, 1, 0
    g.W, g.N, g.M = (W.data_ptr() if W is not None else 0), N, T2
    if out is not None:
        g.out, g.ldo = out.data_ptr(), 36 * N
    g.nbatch, g.batch_stride_a, g.batch_stride_w, g.batch_stride_out = 36, C, N * C, N
    return g


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS


st = _C.stream_ptr
# (tiles, C, N): G / F / H layers of a batch-8 training step (8*H*W/16 tiles) + one ragged count
shapes = [(8 * 96 * 320 // 16, 256, 256), (8 * 48 * 160 // 16, 512, 512), (8 * 48 * 160 // 16, 256, 512), (8 * 96 * 1279 // 16 + 7, 256, 256),
          (8 * 48 * 639 // 16 + 3, 512, 512), (8 * 192 * 640 // 16, 128, 128), (8 * 24 * 80 // 16, 512, 512)]
tot = {'gemm_old': 0.0, 'gemm_dma2': 0.0, 'gemm_dma3': 0.0, 'wgrad_old': 0.0, 'wgrad_dma2': 0.0, 'wgrad_dma3': 0.0}
for (T2, C, N) in shapes:
    V = torch.randn(T2, 36, C, device='cuda')
    U = torch.randn(36, N, C, device='cuda')
    Gy = torch.randn(T2, 36, N, device='cuda')
    o_old, o_new = torch.empty(T2, 36, N, device='cuda'), torch.empty(T2, 36, N, device='cuda')
    g_old, g_new = desc(V, U, o_old, T2, C, N), desc(V, U, o_new, T2, C, N)
    assert L.efgh_plane_gemm_supported(ctypes.byref(g_new))
    fl = 2.0 * 36 * T2 * C * N
    t_old = timed(lambda: _C.check(L.efgh_gather_gemm(ctypes.byref(g_old), st())))
    res = []
    for nbuf in (2, 3):
        o_new.zero_()
        t = timed(lambda: _C.check(L.efgh_plane_gemm(ctypes.byref(g_new), c_int32(nbuf), st())))
        res.append((t, bool(torch.equal(o_old, o_new))))
        tot['gemm_dma%d' % nbuf] += t
    tot['gemm_old'] += t_old
    line = 'tiles %6d C %3d N %3d  gemm: old %.3f ms %5.1f TF | dma2 %.3f ms %5.1f TF equal %s | dma3 %.3f ms %5.1f TF equal %s' % (
        T2, C, N, t_old, fl / t_old / 1e9, res[0][0], fl / res[0][0] / 1e9, res[0][1], res[1][0], fl / res[1][0] / 1e9, res[1][1])
    # weight gradient
    if C % 128 == 0 and N % 128 == 0:
        S_old, S_new = torch.empty(36, N, C, device='cuda'), torch.empty(36, N, C, device='cuda')
        w = desc(V, None, None, T2, C, N)
        ws_old = ops._scratch(L.efgh_gather_wgrad_workspace(ctypes.byref(w)), V.device)
        t_old = timed(lambda: _C.check(L.efgh_gather_wgrad_batched(ctypes.byref(w), ptr(Gy), c_int64(36 * N), c_int64(N), ptr(S_old),
                                                                   c_int64(N * C), ptr(ws_old), st())))
        ws_new = torch.empty(max(1, L.efgh_plane_wgrad_workspace(ctypes.byref(w))), device='cuda')
        res = []
        for nbuf in (2, 3):
            S_new.zero_()
            t = timed(lambda: _C.check(L.efgh_plane_wgrad_batched(ctypes.byref(w), ptr(Gy), c_int64(36 * N), c_int64(N), ptr(S_new),
                                                                  c_int64(N * C), ptr(ws_new), c_int32(nbuf), st())))
            res.append((t, bool(torch.equal(S_old, S_new)), float((S_old - S_new).abs().max() / S_old.abs().max())))
            tot['wgrad_dma%d' % nbuf] += t
        tot['wgrad_old'] += t_old
        line += '\n' + ' ' * 29 + 'wgrad: old %.3f ms %5.1f TF | dma2 %.3f ms %5.1f TF equal %s (%.1e) | dma3 %.3f ms %5.1f TF equal %s' % (
            t_old, fl / t_old / 1e9, res[0][0], fl / res[0][0] / 1e9, res[0][1], res[0][2], res[1][0], fl / res[1][0] / 1e9, res[1][1])
    print(line, flush=True)
print('sum over shapes (ms):', {k: round(v, 3) for k, v in tot.items()})
