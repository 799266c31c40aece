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


// ---- CSR form of the splat (no floating-point atomics) ---------------------------------------------------------
// The scatter-add above spends its time in same-address fp32 atomics (17 points per vertex on level 0).  Inverting
// `off` once per level - count per vertex, exclusive scan, fill - turns the splat into a gather: one lane group per
// vertex walks its (point, remainder) list, sums bary * feat rows in registers and writes the normalised row once.
__global__ void __launch_bounds__(TPB)
k_csr_count(const int *__restrict__ off, long long n4, int *__restrict__ cnt) {
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n4; i += (long long)gridDim.x * TPB)
        atomicAdd(&cnt[off[i]], 1);
}

// exclusive scan of cnt[0..H) in three phases (1024 elements per block)
__global__ void __launch_bounds__(TPB)
k_scan_local(const int *__restrict__ in, int H, int *__restrict__ out, int *__restrict__ block_sum) {
    __shared__ int wsum_[TPB / 64];
    const int base = blockIdx.x * 1024 + threadIdx.x * 4;
    int v[4], s = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[q] = base + q < H ? in[base + q] : 0; s += v[q]; }
    int incl = s;                                  // inclusive scan of the per-thread sums across the block
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o); if ((threadIdx.x & 63) >= o) incl += t; }
    if ((threadIdx.x & 63) == 63) wsum_[threadIdx.x >> 6] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum_[w];
    int run = woff + incl - s;
#pragma unroll
    for (int q = 0; q < 4; ++q) { if (base + q < H) out[base + q] = run; run += v[q]; }
    if (threadIdx.x == TPB - 1) block_sum[blockIdx.x] = woff + incl;
}

__global__ void __launch_bounds__(TPB)
k_scan_blocksums(int *__restrict__ block_sum, int nb, int *__restrict__ total) {
    __shared__ int buf[TPB];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += TPB) {
        const int i = base + threadIdx.x;
        const int v = i < nb ? block_sum[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < TPB; o <<= 1) {
            int t = threadIdx.x >= o ? buf[threadIdx.x - o] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nb) block_sum[i] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == TPB - 1) carry += buf[TPB - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ void __launch_bounds__(TPB)
k_scan_add(int *__restrict__ out, int H, const int *__restrict__ block_sum, const int *__restrict__ total) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i < H) out[i] += block_sum[i >> 10];
    if (i == 0) out[H] = *total;
}

__global__ void __launch_bounds__(TPB)
k_csr_fill(const int *__restrict__ off, long long n4, const int *__restrict__ start, int *__restrict__ fill,
           int *__restrict__ list) {
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n4; i += (long long)gridDim.x * TPB) {
        const int h = off[i];
        list[start[h] + atomicAdd(&fill[h], 1)] = (int)i;           // entry id = r*n + p
    }
}

// The fill above places the entries of a vertex in arrival order (int atomics), which varies run to run.  Sorting every
// segment by entry id (one wave per vertex, rank = number of smaller ids) fixes the summation order of the gather, so the
// whole E branch - and with it the forward pass - is bit-reproducible.
__global__ void __launch_bounds__(TPB)
k_csr_sort(const int *__restrict__ start, const int *__restrict__ list, int H, int *__restrict__ sorted) {
    const int lane = threadIdx.x & 63;
    const long long h = (long long)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (h >= H) return;
    const int e0 = start[h], n = start[h + 1] - e0;
    for (int i = lane; i < n; i += 64) {
        const int id = list[e0 + i];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += list[e0 + j] < id ? 1 : 0;      // ids are distinct
        sorted[e0 + rank] = id;
    }
}

// One wave per vertex: G lanes across the channels (lane g owns channels 4g.. and 4(g+G)..) x S = 64/G entry slots, so
// S list entries of the vertex are in flight at once (the walk is a chain of dependent loads: list -> bary, feat row);
// the S partial sums are folded with shuffles at the end.
template <int G>
__global__ void __launch_bounds__(TPB)
k_splat_gather(const float *__restrict__ feat, long long ldf, int C, const float *__restrict__ bary, int n,
               const int *__restrict__ start, const int *__restrict__ list, int H, float *__restrict__ splat,
               float *__restrict__ wsum) {
    constexpr int S = 64 / G;
    const int c4n = C >> 2;
    const int lane = threadIdx.x & 63, g = lane % G, sl = lane / G;
    const long long h = (long long)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (h >= H) return;
    const int e0 = start[h], e1 = start[h + 1];
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    float w = 0.f;
    // two entries per slot and trip (both ids first, then both rows): halves the dependent-load chain
    for (int e = e0 + sl; e < e1; e += 2 * S) {
        const bool two = e + S < e1;
        const int id0 = list[e], id1 = two ? list[e + S] : id0;
        const float b0 = bary[id0], b1 = two ? bary[id1] : 0.f;
        const float4 *r0 = reinterpret_cast<const float4 *>(feat + (long long)(id0 % n) * ldf);
        const float4 *r1 = reinterpret_cast<const float4 *>(feat + (long long)(id1 % n) * ldf);
        w += b0 + b1;
        if (g < c4n) {
            const float4 f0 = r0[g], f1 = r1[g];
            a0.x += b0 * f0.x + b1 * f1.x; a0.y += b0 * f0.y + b1 * f1.y; a0.z += b0 * f0.z + b1 * f1.z; a0.w += b0 * f0.w + b1 * f1.w;
        }
        if (g + G < c4n) {
            const float4 f0 = r0[g + G], f1 = r1[g + G];
            a1.x += b0 * f0.x + b1 * f1.x; a1.y += b0 * f0.y + b1 * f1.y; a1.z += b0 * f0.z + b1 * f1.z; a1.w += b0 * f0.w + b1 * f1.w;
        }
    }
#pragma unroll
    for (int o = G; o < 64; o <<= 1) {
        w += __shfl_xor(w, o);
        a0.x += __shfl_xor(a0.x, o); a0.y += __shfl_xor(a0.y, o); a0.z += __shfl_xor(a0.z, o); a0.w += __shfl_xor(a0.w, o);
        a1.x += __shfl_xor(a1.x, o); a1.y += __shfl_xor(a1.y, o); a1.z += __shfl_xor(a1.z, o); a1.w += __shfl_xor(a1.w, o);
    }
    if (sl != 0) return;
    const float nrm = 1.0f / (w + 1e-5f);
    float4 *dst = reinterpret_cast<float4 *>(splat + h * C);
    if (g < c4n) dst[g] = make_float4(a0.x * nrm, a0.y * nrm, a0.z * nrm, a0.w * nrm);
    if (g + G < c4n) dst[g + G] = make_float4(a1.x * nrm, a1.y * nrm, a1.z * nrm, a1.w * nrm);
    if (g == 0) wsum[h] = w;
}


// ---- adjoint of the neighbour gather (bilateralNN.py:240-242) as a CSR gather: dst[h][c] = sum over (m, t) with
// table[m][t] == h of src[m][t*C + c].  Same inversion as the splat, over the [M][16] neighbour table (entries < 0 skipped).
__global__ void __launch_bounds__(TPB)
k_tcsr_count(const int *__restrict__ table, long long total, int T, int *__restrict__ cnt) {
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const int h = table[i];
        if ((int)(i & 15) < T && h >= 0) atomicAdd(&cnt[h], 1);
    }
}

__global__ void __launch_bounds__(TPB)
k_tcsr_fill(const int *__restrict__ table, long long total, int T, const int *__restrict__ start, int *__restrict__ fill,
            int *__restrict__ list) {
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const int h = table[i];
        if ((int)(i & 15) < T && h >= 0) list[start[h] + atomicAdd(&fill[h], 1)] = (int)i;      // entry id = m*16 + t
    }
}

template <int G>
__global__ void __launch_bounds__(TPB)
k_table_gather_add(const float *__restrict__ src, int T, int C, const int *__restrict__ start,
                   const int *__restrict__ list, int H, float *__restrict__ dst) {
    constexpr int S = 64 / G;
    const int c4n = C >> 2;
    const int lane = threadIdx.x & 63, g = lane % G, sl = lane / G;
    const long long h = (long long)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (h >= H) return;
    const int e0 = start[h], e1 = start[h + 1];
    const long long ld = (long long)T * C;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    for (int e = e0 + sl; e < e1; e += S) {
        const int id = list[e];
        const float4 *row = reinterpret_cast<const float4 *>(src + (long long)(id >> 4) * ld + (long long)(id & 15) * C);
        if (g < c4n) { float4 f = row[g]; a0.x += f.x; a0.y += f.y; a0.z += f.z; a0.w += f.w; }
        if (g + G < c4n) { float4 f = row[g + G]; a1.x += f.x; a1.y += f.y; a1.z += f.z; a1.w += f.w; }
    }
#pragma unroll
    for (int o = G; o < 64; o <<= 1) {
        a0.x += __shfl_xor(a0.x, o); a0.y += __shfl_xor(a0.y, o); a0.z += __shfl_xor(a0.z, o); a0.w += __shfl_xor(a0.w, o);
        a1.x += __shfl_xor(a1.x, o); a1.y += __shfl_xor(a1.y, o); a1.z += __shfl_xor(a1.z, o); a1.w += __shfl_xor(a1.w, o);
    }
    if (sl != 0) return;
    float4 *d = reinterpret_cast<float4 *>(dst + h * C);
    if (g < c4n) d[g] = a0;
    if (g + G < c4n) d[g + G] = a1;
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

extern "C" int64_t efgh_splat_csr_workspace_ints(int32_t n, int32_t H) {
    return (int64_t)H + 1 /*start*/ + H /*fill*/ + 8LL * n /*list, sorted list*/ + (H + 1023) / 1024 + 2 /*block sums, total*/;
}

// off [4][n] (vertex of every (remainder, point)) -> CSR: start [H+1], list [4n] (entry ids r*n + p grouped by vertex)
extern "C" int efgh_splat_csr_build(const int32_t *off, int32_t n, int32_t H, int32_t *ws, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(off && ws && n > 0 && H > 0);
    int *start = ws, *fill = ws + H + 1, *list = fill + H, *sorted = list + 4LL * n, *bsum = sorted + 4LL * n;
    const int nb = (H + 1023) / 1024;
    int *total = bsum + nb;
    if (hipMemsetAsync(start, 0, (size_t)(2 * (long long)H + 1) * 4, st) != hipSuccess) {
        efgh_set_error("splat csr: memset failed");
        return EFGH_E_LAUNCH;
    }
    const long long n4 = 4LL * n;
    k_csr_count<<<grid_for(n4), TPB, 0, st>>>(off, n4, fill);           // counts into `fill`, scanned into `start`
    k_scan_local<<<nb, TPB, 0, st>>>(fill, H, start, bsum);
    k_scan_blocksums<<<1, TPB, 0, st>>>(bsum, nb, total);
    k_scan_add<<<cdiv(H, TPB), TPB, 0, st>>>(start, H, bsum, total);
    if (hipMemsetAsync(fill, 0, (size_t)H * 4, st) != hipSuccess) {
        efgh_set_error("splat csr: memset failed");
        return EFGH_E_LAUNCH;
    }
    k_csr_fill<<<grid_for(n4), TPB, 0, st>>>(off, n4, start, fill, list);
    k_csr_sort<<<cdiv(H, TPB / 64), TPB, 0, st>>>(start, list, H, sorted);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_splat_gather(const float *feat, int64_t ldf, int32_t C, const float *bary, int32_t n, int32_t H,
                                 const int32_t *ws, float *splat, float *wsum, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(feat && bary && ws && splat && wsum && n > 0 && H > 0 && C > 0 && C % 4 == 0 && C <= 512 && ldf % 4 == 0);
    const int *start = ws, *list = ws + 2LL * H + 1 + 4LL * n;      // the sorted copy
    const int c4n = C / 4;
    const int grid = cdiv(H, TPB / 64);
    if (c4n <= 16) k_splat_gather<16><<<grid, TPB, 0, st>>>(feat, ldf, C, bary, n, start, list, H, splat, wsum);
    else if (c4n <= 32) k_splat_gather<32><<<grid, TPB, 0, st>>>(feat, ldf, C, bary, n, start, list, H, splat, wsum);
    else k_splat_gather<64><<<grid, TPB, 0, st>>>(feat, ldf, C, bary, n, start, list, H, splat, wsum);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int64_t efgh_table_csr_workspace_ints(int64_t M) {
    return 2 * M + 1 + 16 * M + (M + 1023) / 1024 + 2;
}

// dst [M][C] = adjoint of the gather through table [M][16] (first T columns) applied to src [M][T*C]; ws as above
extern "C" int efgh_table_gather_add(const float *src, const int32_t *table, int64_t M, int32_t T, int32_t C, int32_t *ws,
                                     float *dst, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(src && table && ws && dst && M > 0 && M < (1 << 27) && T > 0 && T <= 16 && C > 0 && C % 4 == 0 && C <= 512);
    const int H = (int)M;
    int *start = ws, *fill = ws + H + 1, *list = fill + H, *bsum = list + 16LL * M;
    const int nb = (H + 1023) / 1024;
    int *total = bsum + nb;
    if (hipMemsetAsync(start, 0, (size_t)(2 * (long long)H + 1) * 4, st) != hipSuccess) {
        efgh_set_error("table csr: memset failed");
        return EFGH_E_LAUNCH;
    }
    const long long tot = 16LL * M;
    k_tcsr_count<<<grid_for(tot), TPB, 0, st>>>(table, tot, T, fill);
    k_scan_local<<<nb, TPB, 0, st>>>(fill, H, start, bsum);
    k_scan_blocksums<<<1, TPB, 0, st>>>(bsum, nb, total);
    k_scan_add<<<cdiv(H, TPB), TPB, 0, st>>>(start, H, bsum, total);
    if (hipMemsetAsync(fill, 0, (size_t)H * 4, st) != hipSuccess) {
        efgh_set_error("table csr: memset failed");
        return EFGH_E_LAUNCH;
    }
    k_tcsr_fill<<<grid_for(tot), TPB, 0, st>>>(table, tot, T, start, fill, list);
    const int c4n = C / 4, grid = cdiv(H, TPB / 64);
    if (c4n <= 16) k_table_gather_add<16><<<grid, TPB, 0, st>>>(src, T, C, start, list, H, dst);
    else if (c4n <= 32) k_table_gather_add<32><<<grid, TPB, 0, st>>>(src, T, C, start, list, H, dst);
    else k_table_gather_add<64><<<grid, TPB, 0, st>>>(src, T, C, start, list, H, dst);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}
