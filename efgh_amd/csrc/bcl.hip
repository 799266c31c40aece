// BCL splat (K3): scatter-add of bary (x) feat onto lattice vertices + density normalisation.
// Replaces SparseSum / the ones-splat of nets/bilateralNN.py:6-40,179-211.
//   splat[h][c] = ( sum_{(p,r): off[r][p]==h} bary[r][p]*feat[p][c] ) / ( sum bary[r][p] + 1e-5 )
// v1: fp32 global atomics, shaped as contiguous row segments (one vertex row of C floats per
// group of C/4.. lanes); the sum order, hence the last bits, is not fixed run to run.
#include "common.h"

namespace {
constexpr int TPB = 256;

__global__ void __launch_bounds__(TPB)
k_splat_add(const float *__restrict__ feat, long long ldf, int C, const float *__restrict__ bary,
            const int *__restrict__ off, int n, float *__restrict__ splat, float *__restrict__ wsum) {
    const int c4n = C >> 2;
    long long total = (long long)n * c4n;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        int p = (int)(i / c4n), cq = (int)(i - (long long)p * c4n), c = cq * 4;
        float4 f = *reinterpret_cast<const float4 *>(feat + (long long)p * ldf + c);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float b = bary[(long long)r * n + p];
            int h = off[(long long)r * n + p];
            float *dst = splat + (long long)h * C + c;
            atomicAdd(dst + 0, b * f.x); atomicAdd(dst + 1, b * f.y);
            atomicAdd(dst + 2, b * f.z); atomicAdd(dst + 3, b * f.w);
            if (cq == 0) atomicAdd(wsum + h, b);
        }
    }
}

__global__ void __launch_bounds__(TPB)
k_splat_norm(float *__restrict__ splat, const float *__restrict__ wsum, int H, int C) {
    const int c4n = C >> 2;
    long long total = (long long)H * c4n;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        int h = (int)(i / c4n);
        float nrm = 1.0f / (wsum[h] + 1e-5f);
        float4 *p = reinterpret_cast<float4 *>(splat) + i;
        float4 v = *p;
        v.x *= nrm; v.y *= nrm; v.z *= nrm; v.w *= nrm;
        *p = v;
    }
}

__global__ void __launch_bounds__(TPB)
k_splat_bwd(const float *__restrict__ gsplat, const float *__restrict__ wsum, int C,
            const float *__restrict__ bary, const int *__restrict__ off, int n, float *__restrict__ gfeat,
            long long ldg) {
    const int c4n = C >> 2;
    long long total = (long long)n * c4n;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        int p = (int)(i / c4n), c = (int)(i - (long long)p * c4n) * 4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int h = off[(long long)r * n + p];
            float w = bary[(long long)r * n + p] * (1.0f / (wsum[h] + 1e-5f));
            float4 g = *reinterpret_cast<const float4 *>(gsplat + (long long)h * C + c);
            a.x += w * g.x; a.y += w * g.y; a.z += w * g.z; a.w += w * g.w;
        }
        *reinterpret_cast<float4 *>(gfeat + (long long)p * ldg + c) = a;
    }
}

int grid_for(long long total) {
    long long g = (total + TPB - 1) / TPB;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}
}  // namespace

extern "C" int efgh_splat_fwd(const float *feat, int64_t ldf, int32_t C, const float *bary, const int32_t *off,
                              int32_t n, int32_t H, float *splat, float *wsum, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(feat && bary && off && splat && wsum && n > 0 && H > 0 && C > 0 && C % 4 == 0 && ldf % 4 == 0);
    if (hipMemsetAsync(splat, 0, (size_t)H * C * 4, st) != hipSuccess ||
        hipMemsetAsync(wsum, 0, (size_t)H * 4, st) != hipSuccess) {
        efgh_set_error("splat: memset failed");
        return EFGH_E_LAUNCH;
    }
    k_splat_add<<<grid_for((long long)n * (C / 4)), TPB, 0, st>>>(feat, ldf, C, bary, off, n, splat, wsum);
    k_splat_norm<<<grid_for((long long)H * (C / 4)), TPB, 0, st>>>(splat, wsum, H, C);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_splat_bwd(const float *gsplat, const float *wsum, int32_t C, const float *bary,
                              const int32_t *off, int32_t n, int32_t H, float *gfeat, int64_t ldg, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(gsplat && wsum && bary && off && gfeat && n > 0 && H > 0 && C % 4 == 0 && ldg % 4 == 0);
    k_splat_bwd<<<grid_for((long long)n * (C / 4)), TPB, 0, st>>>(gsplat, wsum, C, bary, off, n, gfeat, ldg);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}
