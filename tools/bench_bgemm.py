"""batched plain GEMM through k_gather_gemm (mode 0, T = 1): the contraction a 2-D Winograd F(4x4,3x3) would run (36 alpha)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efgh_amd import ops

torch.set_grad_enabled(False)
for (tiles, C, N) in [(8 * 96 * 320 // 16, 256, 256), (8 * 48 * 160 // 16, 512, 512), (8 * 192 * 640 // 16, 128, 128),
                      (8 * 384 * 1280 // 16, 64, 64), (4 * 96 * 1280 // 16, 256, 256)]:
    nb = 36
    A = torch.randn(nb, tiles, C, device='cuda')
    W = torch.randn(nb, N, C, device='cuda')
    out = torch.empty(nb, tiles, N, device='cuda')
    for _ in range(2):
        ops.gather_gemm(A, C, C, 1, W, N, tiles, out, N, mode=0, batch=(nb, tiles * C, N * C, tiles * N))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 5
    for _ in range(n):
        ops.gather_gemm(A, C, C, 1, W, N, tiles, out, N, mode=0, batch=(nb, tiles * C, N * C, tiles * N))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * nb * tiles * C * N
    ref = torch.einsum('tc,nc->tn', A[3, :64].double(), W[3].double())
    err = float((out[3, :64].double() - ref).abs().max() / ref.abs().max())
    print('tiles %d C %d N %d x36: %.3f ms  %.1f TF executed (= %.1f TF direct-form equivalent)  err %.1e' % (
        tiles, C, N, ms, fl / ms / 1e9, fl * 4 / ms / 1e9, err))
