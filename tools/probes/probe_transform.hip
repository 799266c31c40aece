// Probe: what bounds a "36 loads -> math -> 36 stores per thread" kernel (the Winograd input transform) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// x [B*H*W][C], tiles of 4x4 outputs (6x6 inputs, overlapping), V [T][36][C]; MATH = VALU ops of fake math per value
template <int MATH, int NLOAD>
__global__ void __launch_bounds__(256) k_t(const float *__restrict__ x, int C, int H, int W, int TW, float *__restrict__ V) {
    const int q2 = C / 2, idx = blockIdx.x * 256 + threadIdx.x;
    const int tx = idx / q2, q = idx % q2;
    if (tx >= TW) return;
    const int rowt = blockIdx.y, TH = H / 4, ty = rowt % TH, b = rowt / TH;
    const long long t = (long long)rowt * TW + tx;
    float2 d[36];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int i = r * 6 + c;
            if (i >= NLOAD) { d[i] = make_float2(1.f, 2.f); continue; }
            const int y = min(max(4 * ty - 1 + r, 0), H - 1), xx = min(max(4 * tx - 1 + c, 0), W - 1);
            d[i] = *reinterpret_cast<const float2 *>(x + (((long long)b * H + y) * W + xx) * C + q * 2);
        }
#pragma unroll
    for (int m = 0; m < MATH; ++m)
#pragma unroll
        for (int i = 0; i < 36; ++i) { d[i].x = d[i].x * 1.0001f + d[(i + 7) % 36].y; d[i].y = d[i].y * 0.9999f - d[(i + 11) % 36].x; }
    float *vp = V + t * 36 * C + q * 2;
#pragma unroll
    for (int i = 0; i < 36; ++i) *reinterpret_cast<float2 *>(vp + i * C) = d[i];
}

int main() {
    const int B = 4, H = 96, W = 320, C = 256, TW = W / 4, TH = H / 4;
    const long long T = (long long)B * TH * TW;
    float *x, *V; CK(hipMalloc(&x, (size_t)B * H * W * C * 4)); CK(hipMalloc(&V, (size_t)T * 36 * C * 4));
    CK(hipMemset(x, 0, (size_t)B * H * W * C * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    dim3 grid((TW * (C / 2) + 255) / 256, B * TH);
    auto run = [&](const char *name, auto f) {
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%-56s %8.1f us\n", name, best * 1e3);
    };
    run("36 loads, no math, 36 stores", [&] { k_t<0, 36><<<grid, 256>>>(x, C, H, W, TW, V); });
    run("36 loads, 72 VALU, 36 stores", [&] { k_t<1, 36><<<grid, 256>>>(x, C, H, W, TW, V); });
    run("36 loads, 288 VALU, 36 stores", [&] { k_t<4, 36><<<grid, 256>>>(x, C, H, W, TW, V); });
    run("36 loads, 576 VALU, 36 stores", [&] { k_t<8, 36><<<grid, 256>>>(x, C, H, W, TW, V); });
    run("0 loads, no math, 36 stores", [&] { k_t<0, 0><<<grid, 256>>>(x, C, H, W, TW, V); });
    run("0 loads, 576 VALU, 36 stores", [&] { k_t<8, 0><<<grid, 256>>>(x, C, H, W, TW, V); });
    run("16 loads (no overlap), no math, 36 stores", [&] { k_t<0, 16><<<grid, 256>>>(x, C, H, W, TW, V); });
    printf("bytes: x %.1f MB, V %.1f MB\n", B * H * W * C * 4 / 1e6, T * 36 * C * 4 / 1e6);
    return 0;
}
