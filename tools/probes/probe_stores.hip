// Probe: store bandwidth of the patterns the Winograd transform kernels use (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o probe_stores probe_stores.hip && ./probe_stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_fill4(float4 *p, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void k_fill2(float2 *p, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = make_float2(1.f, (float)i);
}
// thread = (tile, channel pair): 36 stores of 8 B at plane stride `ps` floats, tile stride `ts` floats
__global__ void k_tile36(float *p, long long tiles, int C, long long ps, long long ts) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int q = (int)(idx % (C / 2));
    const long long t = idx / (C / 2);
    if (t >= tiles) return;
    float *b = p + t * ts + q * 2;
#pragma unroll
    for (int a = 0; a < 36; ++a) *reinterpret_cast<float2 *>(b + a * ps) = make_float2((float)a, (float)q);
}
// the same bytes, but one thread writes ONE plane of FOUR consecutive tiles... (16 B per lane)
__global__ void k_tile36_f4(float *p, long long tiles, int C, long long ps, long long ts) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int q = (int)(idx % (C / 4));
    const long long t = idx / (C / 4);
    if (t >= tiles) return;
    float *b = p + t * ts + q * 4;
#pragma unroll
    for (int a = 0; a < 36; ++a) *reinterpret_cast<float4 *>(b + a * ps) = make_float4((float)a, (float)q, 0.f, 1.f);
}

int main() {
    const long long tiles = 7680 * 4; const int C = 256;
    const long long n = tiles * 36 * C;       // floats (283 MB)
    float *buf; CK(hipMalloc(&buf, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto f) {
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%-58s %8.1f us  %6.2f TB/s\n", name, best * 1e3, n * 4.0 / (best * 1e-3) / 1e12);
    };
    run("contiguous fill, 16 B per lane", [&] { k_fill4<<<8192, 256>>>((float4 *)buf, n / 4); });
    run("contiguous fill, 8 B per lane", [&] { k_fill2<<<8192, 256>>>((float2 *)buf, n / 2); });
    run("(tile, pair) x 36 planes, plane-major [36][T][C], 8 B", [&] { k_tile36<<<(unsigned)(tiles * (C / 2) / 256), 256>>>(buf, tiles, C, tiles * C, C); });
    run("(tile, pair) x 36 planes, tile-major [T][36][C], 8 B", [&] { k_tile36<<<(unsigned)(tiles * (C / 2) / 256), 256>>>(buf, tiles, C, C, 36LL * C); });
    run("(tile, quad) x 36 planes, plane-major, 16 B", [&] { k_tile36_f4<<<(unsigned)(tiles * (C / 4) / 256), 256>>>(buf, tiles, C, tiles * C, C); });
    run("(tile, quad) x 36 planes, tile-major, 16 B", [&] { k_tile36_f4<<<(unsigned)(tiles * (C / 4) / 256), 256>>>(buf, tiles, C, C, 36LL * C); });
    return 0;
}
