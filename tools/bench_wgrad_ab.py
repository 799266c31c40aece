"""k_gather_wgrad (mode 1) at training-step shapes, one library per process (EFGH_LIB=...): python tools/bench_wgrad_ab.py [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efgh_amd import ops

torch.set_grad_enabled(False)
ops.USE_WINO_WGRAD = False
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
tot = 0.0
# (B, Hin, Win, C, N, k, stride)
for (B, H, W, C, N, k, s) in [(8, 384, 1280, 64, 128, 3, 2), (8, 192, 640, 128, 256, 3, 2), (8, 96, 320, 256, 512, 3, 2), (8, 384, 1280, 64, 128, 1, 2),
                              (8, 96, 320, 256, 512, 1, 2), (8, 48, 160, 512, 512, 1, 1), (8, 192, 640, 128, 128, 3, 1), (8, 384, 1280, 64, 64, 3, 1)]:
    Ho, Wo = H // s, W // s
    T = k * k
    dh = [t // k - k // 2 for t in range(T)]
    dw = [t % k - k // 2 for t in range(T)]
    geom = (B, H, W, Ho, Wo, s, s, dh, dw, Ho, Wo, 1, 1, 0, 0)
    M = B * Ho * Wo
    torch.manual_seed(0)
    A = torch.randn(B * H * W, C, device='cuda')
    G = torch.randn(M, N, device='cuda')
    dW = torch.empty(N, T, C, device='cuda')
    for _ in range(2):
        ops.gather_wgrad(A, C, C, T, N, M, G, N, dW, mode=1, geom=geom)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gather_wgrad(A, C, C, T, N, M, G, N, dW, mode=1, geom=geom)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tot += ms
    fl = 2.0 * M * N * T * C
    print('B%d %dx%d %d->%d k%d s%d : %.3f ms  %.1f TF  checksum %r' % (B, H, W, C, N, k, s, ms, fl / ms / 1e9, float(dW.double().abs().sum())), flush=True)
print('sum %.3f ms' % tot)
