// BCL splat (K3) and the adjoints of the BCL's two index operations, as GATHERS - no floating-point atomics on the path.
// Replaces SparseSum / the ones-splat of nets/bilateralNN.py:6-40,179-211 and the autograd of the neighbour gather :240-242.
//   splat[h][c] = ( sum_{(p,r): off[p][r]==h} bary[p][r]*x[p][c] ) / ( sum bary[p][r] + 1e-5 )
// with x[p] = [el_minus_gr[p] (4 channels, written by the lattice build) | feat[p] (Cf channels)]: the level's input row of
// enet.py:113,119,... is never concatenated, the two parts are read where they lie (feature rows stay 128-B multiples).
//
// The inverse of `off` (vertex -> sorted list of its flat positions f = 4p + r) is a by-product of the lattice build
// (lattice.hip: list, vseg), so the splat is one kernel: a wave per vertex, the whole list (<= 64 entries per trip) and its
// weights fetched by one load each, then lanes = (entry, 16-byte chunk) pairs so that every row of the list is in flight at
// once; partial sums are folded across the entry slots with shuffles and the normalised row is written once.
#include "common.h"

namespace {
constexpr int TPB = 256;

// blocks of consecutive vertices / points are dealt to the XCDs as contiguous bands (blocks b and b+8 share an XCD under
// round-robin dispatch): vertices are numbered sample-major, so an XCD's L2 sees the rows of one or two samples only.
// Speed only - any placement gives the same result.
__device__ __forceinline__ int64_t xcd_band_block(int64_t b, int64_t nblocks) {
    const int64_t per = (nblocks + 7) / 8;
    const int64_t x = b & 7, j = b >> 3;
    const int64_t r = x * per + j;
    return (j < per && r < nblocks) ? r : -1;
}

__device__ __forceinline__ void fma4(float4 &a, float b, const float4 &v) {
    a.x += b * v.x; a.y += b * v.y; a.z += b * v.z; a.w += b * v.w;
}

// LPV lanes per vertex (64: one vertex per wave; 32: two - short lists leave a whole wave mostly idle and the kernel is bound
// by the number of vertices in flight x the three dependent round trips vseg -> list/bary -> rows).  Four trips of row loads
// are issued before the first FMA.
template <int LPV>
__global__ void __launch_bounds__(TPB)
k_splat_gather(const float4 *__restrict__ emg, const float4 *__restrict__ feat, int64_t ldf4, int cf4,
               const float *__restrict__ bary, const int *__restrict__ list, const int2 *__restrict__ vseg, int H,
               float4 *__restrict__ splat, float *__restrict__ wsum, int normalize) {
    constexpr int VPB = TPB / LPV;                             // vertices per block
    const int64_t nblocks = ((int64_t)H + VPB - 1) / VPB;
    const int64_t blk = xcd_band_block(blockIdx.x, nblocks);
    if (blk < 0) return;
    const int lane = threadIdx.x & 63, lv = lane & (LPV - 1), gb = lane - lv;
    const int64_t hh = blk * VPB + threadIdx.x / LPV;
    const bool have = hh < H;
    const int h = have ? (int)hh : 0;
    const int he = emg ? 1 : 0;
    const int CH = he + cf4;                                   // 16-byte chunks per entry row
    const int EPT = CH >= LPV ? 1 : LPV / CH;                  // entries in flight per trip
    const int e = lv / CH, c = lv - e * CH;                    // this lane: entry slot e, chunk c (and c + 64 when CH > 64)
    const bool active = e < EPT;
    const int cc = active ? c : 0;
    int2 seg = have ? vseg[h] : make_int2(0, 0);
    int lmax = seg.y;
    if (LPV <= 32) lmax = max(lmax, __shfl_xor(lmax, 32));     // every group of the wave runs the same shuffle rounds
    if (LPV <= 16) lmax = max(lmax, __shfl_xor(lmax, 16));
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    float w = 0.f;
    for (int base = 0; base < lmax; base += LPV) {
        const int m = min(LPV, max(seg.y - base, 0));
        const int mm = min(LPV, lmax - base);
        const int f_l = lv < m ? list[seg.x + base + lv] : 0;
        const float b_l = lv < m ? bary[f_l] : 0.f;
        for (int t = 0; t < mm; t += 4 * EPT) {
            float4 v[4];
            float bq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                bq[u] = 0.f;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t + u * EPT >= mm) continue;                          // (wave-uniform: no trip without an entry)
                const int j = t + u * EPT + e;
                const int f = __shfl(f_l, gb + (j & (LPV - 1)));          // (all lanes take part in both shuffles)
                const float b = __shfl(b_l, gb + (j & (LPV - 1)));
                const bool ok = active && j < m;
                // no branch around the load: lanes without an entry read row 0 with weight 0, so the four loads of a macro
                // trip are independent and all in flight before the first FMA
                const int p = ok ? f >> 2 : 0;
                bq[u] = ok ? b : 0.f;
                const float4 *src = (he && cc == 0) ? emg + p : feat + ((int64_t)p * ldf4 + (cc - he));
                v[u] = *src;
                if (LPV == 64 && c + 64 < CH && ok) fma4(a1, b, feat[(int64_t)p * ldf4 + (c + 64 - he)]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { fma4(a0, bq[u], v[u]); w += bq[u]; }
        }
    }
    // fold the entry slots: lane c (< CH, slot 0) adds lanes c + k*CH of its group
    for (int k = 1; k < EPT; ++k) {
        const int src = gb + ((lv + k * CH) & (LPV - 1));
        const float4 o = make_float4(__shfl(a0.x, src), __shfl(a0.y, src), __shfl(a0.z, src), __shfl(a0.w, src));
        const float ow = __shfl(w, src);
        if (e == 0) { a0.x += o.x; a0.y += o.y; a0.z += o.z; a0.w += o.w; w += ow; }
    }
    if (e != 0 || !have) return;
    // w of slot 0 was accumulated by every lane of the slot identically
    const float nrm = normalize ? 1.0f / (w + 1e-5f) : 1.0f;          // (use_norm = False, bilateralNN.py:196: the plain sparse sum)
    float4 *dst = splat + (int64_t)h * CH;
    dst[c] = make_float4(a0.x * nrm, a0.y * nrm, a0.z * nrm, a0.w * nrm);
    if (LPV == 64 && c + 64 < CH) dst[c + 64] = make_float4(a1.x * nrm, a1.y * nrm, a1.z * nrm, a1.w * nrm);
    if (c == 0) wsum[h] = w;
}

// backward of the splat w.r.t. the feature part of the rows (el_minus_gr carries no gradient, generate_data.py:119):
//   gfeat[p][c] = sum_r bary[p][r] / (wsum[off[p][r]] + 1e-5) * gsplat[off[p][r]][coff + c]
__global__ void __launch_bounds__(TPB)
k_splat_bwd(const float4 *__restrict__ gsplat, int c4, int coff4, const float *__restrict__ wsum, int cf4,
            const float4 *__restrict__ bary, const int4 *__restrict__ off, int n, float4 *__restrict__ gfeat, int64_t ldg4, int normalize) {
    const int64_t total = (int64_t)n * cf4;
    const int64_t nblocks = (total + TPB - 1) / TPB;
    const int64_t blk = xcd_band_block(blockIdx.x, nblocks);
    if (blk < 0) return;
    const int64_t i = blk * TPB + threadIdx.x;
    if (i >= total) return;
    const int p = (int)(i / cf4), c = (int)(i - (int64_t)p * cf4);
    const int4 o = off[p];
    const float4 b = bary[p];
    const int oo[4] = {o.x, o.y, o.z, o.w};
    const float bb[4] = {b.x, b.y, b.z, b.w};
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float w = normalize ? bb[r] * (1.0f / (wsum[oo[r]] + 1e-5f)) : bb[r];
        fma4(a, w, gsplat[(int64_t)oo[r] * c4 + coff4 + c]);
    }
    gfeat[(int64_t)p * ldg4 + c] = a;
}

// ---- adjoint of the neighbour gather (bilateralNN.py:240-242): dst[h][c] = sum over (m, t) with table[m][t] == h of
// src[m][t*C + c].  The exact part of the neighbour relation is symmetric - table[m][t] == h  <=>  table[h][inv(t)] == m,
// inv(t) = index of the negated offset (generate_data.py:44-52: t <-> 15 - t for t >= 1) - so the adjoint is a gather through
// the SAME table; the aliased hits of key2int (lattice.hip k_neighbors: marked in column 15, listed in alist) are excluded
// here and added by k_table_alias_add.
template <int G>
__global__ void __launch_bounds__(TPB)
k_table_gather_t(const float *__restrict__ src, int C, const int *__restrict__ table, int H, float *__restrict__ dst) {
    constexpr int S = 64 / G;
    const int c4n = C >> 2;
    const int lane = threadIdx.x & 63, g = lane % G, sl = lane / G;
    const int64_t nblocks = ((int64_t)H + TPB / 64 - 1) / (TPB / 64);
    const int64_t blk = xcd_band_block(blockIdx.x, nblocks);
    if (blk < 0) return;
    const int h = (int)(blk * (TPB / 64) + (threadIdx.x >> 6));
    if (h >= H) return;
    const int tv = lane < 16 ? table[(int64_t)h * 16 + lane] : -1;
    const int amask = __shfl(tv, 15);
    const int64_t ld = 15LL * C;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    for (int t0 = 0; t0 < 15; t0 += S) {             // (uniform trip count: every lane takes part in the shuffle)
        const int t = t0 + sl;
        const int m = __shfl(tv, t & 15);
        if (t >= 15 || m < 0 || (amask >> t & 1)) continue;
        const int ti = t == 0 ? 0 : 15 - t;
        const float4 *row = reinterpret_cast<const float4 *>(src + (int64_t)m * ld + (int64_t)ti * C);
        if (g < c4n) { float4 f = row[g]; a0.x += f.x; a0.y += f.y; a0.z += f.z; a0.w += f.w; }
        if (g + G < c4n) { float4 f = row[g + G]; a1.x += f.x; a1.y += f.y; a1.z += f.z; a1.w += f.w; }
    }
#pragma unroll
    for (int o = G; o < 64; o <<= 1) {
        a0.x += __shfl_xor(a0.x, o); a0.y += __shfl_xor(a0.y, o); a0.z += __shfl_xor(a0.z, o); a0.w += __shfl_xor(a0.w, o);
        a1.x += __shfl_xor(a1.x, o); a1.y += __shfl_xor(a1.y, o); a1.z += __shfl_xor(a1.z, o); a1.w += __shfl_xor(a1.w, o);
    }
    if (sl != 0) return;
    float4 *d = reinterpret_cast<float4 *>(dst + (int64_t)h * C);
    if (g < c4n) d[g] = a0;
    if (g + G < c4n) d[g + G] = a1;
}

// ---- the aliased neighbour hits (a handful per level; k_neighbors / k_lat_nbr list them in arrival order, which varies from run
// to run).  Both kernels below apply them in a FIXED order without atomics: every block loads the records, ranks them by their
// (entry, target) key in LDS, and owns the targets with target % gridDim.x == blockIdx.x - it walks the sorted records and adds the
// ones that land on its own targets, one after the other.
constexpr int ALIAS_LDS = 4096;

__device__ __forceinline__ int load_sorted_alias(const int2 *__restrict__ alist, const int *__restrict__ n_alias, int alias_cap,
                                                 int2 *sorted) {
    const int na = min(min(*n_alias, alias_cap), ALIAS_LDS);
    for (int i = threadIdx.x; i < na; i += TPB) {
        const int2 a = alist[i];
        int r = 0;
        for (int j = 0; j < na; ++j) {              // (records are distinct: an entry is listed once)
            const int2 o = alist[j];
            r += (o.x < a.x || (o.x == a.x && o.y < a.y)) ? 1 : 0;
        }
        sorted[r] = a;
    }
    __syncthreads();
    return na;
}

// dst[target][c] += src[m][t*C + c]   (adjoint of the neighbour gather on an explicit [H][15 C] intermediate)
__global__ void __launch_bounds__(TPB)
k_table_alias_add(const float *__restrict__ src, int C, const int2 *__restrict__ alist, const int *__restrict__ n_alias,
                  int alias_cap, float *__restrict__ dst) {
    __shared__ int2 sorted[ALIAS_LDS];
    const int na = load_sorted_alias(alist, n_alias, alias_cap, sorted);
    for (int k = 0; k < na; ++k) {
        const int2 a = sorted[k];
        if (a.y % (int)gridDim.x != (int)blockIdx.x) continue;
        const int m = a.x >> 4, t = a.x & 15;
        for (int c = threadIdx.x; c < C; c += TPB) dst[(int64_t)a.y * C + c] += src[((int64_t)m * 15 + t) * C + c];
        __syncthreads();                            // (the next record may hit the same target)
    }
}

// dx[target][c] += sum_n dy[m][n] * w[(n*C + c)*15 + t]   (the same adjoint when the gather and the convolution are one GEMM)
__global__ void __launch_bounds__(TPB)
k_blur_dgrad_alias(const float *__restrict__ dy, int64_t ldy, int N, const float *__restrict__ w, int C,
                   const int2 *__restrict__ alist, const int *__restrict__ n_alias, int alias_cap, float *__restrict__ dx,
                   int64_t ldx) {
    __shared__ int2 sorted[ALIAS_LDS];
    const int na = load_sorted_alias(alist, n_alias, alias_cap, sorted);
    for (int k = 0; k < na; ++k) {
        const int2 a = sorted[k];
        if (a.y % (int)gridDim.x != (int)blockIdx.x) continue;
        const int m = a.x >> 4, t = a.x & 15;
        const float *row = dy + (int64_t)m * ldy;
        for (int c = threadIdx.x; c < C; c += TPB) {
            float acc = 0.f;
            for (int n = 0; n < N; ++n) acc += row[n] * w[((int64_t)n * C + c) * 15 + t];
            dx[(int64_t)a.y * ldx + c] += acc;
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int efgh_splat_gather(const float *emg, const float *feat, int64_t ldf, int32_t Cf, const float *bary,
                                 const int32_t *list, const int32_t *vseg, int32_t H, int32_t avg_len,
                                 int32_t lanes_per_vertex, int32_t normalize, float *splat, float *wsum, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(feat && bary && list && vseg && splat && wsum && H > 0 && Cf > 0 && Cf % 4 == 0 && Cf <= 508 && ldf % 4 == 0);
    const int CH = (emg ? 1 : 0) + Cf / 4;
    // two vertices per wave when a row fits half a wave and the lists are short (avg_len = 4n/H entries per vertex)
    // two vertices per wave when two rows fit a wave's lanes and the lists are short: the kernel is bound by the vertices in
    // flight x the dependent round trips, not by the lanes (measured, level 1 of the bench scene: 46 vs 60 us)
    int lpv = lanes_per_vertex;
    if (lpv == 0) lpv = (CH <= 32 && avg_len <= 32) ? 32 : 64;
    EFGH_CHECK_ARG(lpv == 64 || (lpv == 32 && CH <= 32) || (lpv == 16 && CH <= 16));
    const int vpb = TPB / lpv;
    const int64_t nblocks = ((int64_t)H + vpb - 1) / vpb;
    const int64_t grid = (nblocks + 7) / 8 * 8;
    if (lpv == 16)
        k_splat_gather<16><<<(unsigned)grid, TPB, 0, st>>>((const float4 *)emg, (const float4 *)feat, ldf / 4, Cf / 4, bary, list,
                                                           (const int2 *)vseg, H, (float4 *)splat, wsum, normalize);
    else if (lpv == 32)
        k_splat_gather<32><<<(unsigned)grid, TPB, 0, st>>>((const float4 *)emg, (const float4 *)feat, ldf / 4, Cf / 4, bary, list,
                                                           (const int2 *)vseg, H, (float4 *)splat, wsum, normalize);
    else
        k_splat_gather<64><<<(unsigned)grid, TPB, 0, st>>>((const float4 *)emg, (const float4 *)feat, ldf / 4, Cf / 4, bary, list,
                                                           (const int2 *)vseg, H, (float4 *)splat, wsum, normalize);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_splat_bwd(const float *gsplat, int32_t C, int32_t coff, const float *wsum, int32_t Cf, const float *bary,
                              const int32_t *off, int32_t n, float *gfeat, int64_t ldg, int32_t normalize, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(gsplat && wsum && bary && off && gfeat && n > 0 && C % 4 == 0 && coff % 4 == 0 && Cf % 4 == 0 && Cf > 0 &&
                   coff + Cf <= C && ldg % 4 == 0);
    const int64_t nblocks = ((int64_t)n * (Cf / 4) + TPB - 1) / TPB;
    const int64_t grid = (nblocks + 7) / 8 * 8;
    k_splat_bwd<<<(unsigned)grid, TPB, 0, st>>>((const float4 *)gsplat, C / 4, coff / 4, wsum, Cf / 4, (const float4 *)bary,
                                                (const int4 *)off, n, (float4 *)gfeat, ldg / 4, normalize);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_table_gather_transposed(const float *src, const int32_t *table, int32_t H, int32_t C, const int32_t *alist,
                                            const int32_t *n_alias, int32_t alias_cap, float *dst, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(src && table && dst && alist && n_alias && H > 0 && C > 0 && C % 4 == 0 && C <= 512 && alias_cap > 0 && alias_cap <= ALIAS_LDS);
    const int64_t nblocks = ((int64_t)H + TPB / 64 - 1) / (TPB / 64);
    const unsigned grid = (unsigned)((nblocks + 7) / 8 * 8);
    const int c4n = C / 4;
    if (c4n <= 16) k_table_gather_t<16><<<grid, TPB, 0, st>>>(src, C, table, H, dst);
    else if (c4n <= 32) k_table_gather_t<32><<<grid, TPB, 0, st>>>(src, C, table, H, dst);
    else k_table_gather_t<64><<<grid, TPB, 0, st>>>(src, C, table, H, dst);
    k_table_alias_add<<<64, TPB, 0, st>>>(src, C, (const int2 *)alist, n_alias, alias_cap, dst);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_blur_dgrad_alias(const float *dy, int64_t ldy, int32_t N, const float *w, int32_t C, const int32_t *alist,
                                     const int32_t *n_alias, int32_t alias_cap, float *dx, int64_t ldx, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(dy && w && alist && n_alias && dx && N > 0 && C > 0 && alias_cap > 0 && alias_cap <= ALIAS_LDS);
    k_blur_dgrad_alias<<<64, TPB, 0, st>>>(dy, ldy, N, w, C, (const int2 *)alist, n_alias, alias_cap, dx, ldx);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}
