// Weight-gradient gather-GEMM on fp32 MFMA (backward of gemm.hip w.r.t. W):
//
//     dWp[n][t*C + c] += sum_m  G[orow(m)][n] * A[row(m,t)][c]
//
// i.e. the gradient of every Conv2d / ConvTranspose2d / Conv1d / Linear / BCL blur weight of the
// reference (autograd of nets/vgg.py:77, resnet.py:22-30, net_utils.py:35-98, bilateralNN.py:103-135),
// in the packed [N][T][C] layout of the forward kernel; efgh_unpack_weight() scatters it back to
// the reference's (out,in,kh,kw) / (in,out,kh,kw) / (out,in) layouts.
//
// The contraction runs over rows m (pixels / vertices), which is huge, so the launch is split over m
// (gridDim.z chunks) and partial tiles are combined with fp32 global atomics (128-B contiguous
// segments per wave instruction).  Tile: 128 (n) x 128 (k) per block, 32 rows of m per step; the MFMA
// contraction index is m, so fragments are plain ds_read_b32 of [m][n] / [m][k] LDS images
// (lanes 0-31: consecutive columns, lanes 32-63: the next row -> conflict free).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TK = 128, TM = 32, LD = 128;

struct WArgs {
    const float *A; int64_t lda;
    int C, T, K;
    unsigned magicC;
    int Hin, Win, Hv, Wv, sh, sw;
    unsigned long long dhpack, dwpack;
    int Ho, Wo, osh, osw, oh0, ow0;
    const int *table;
    const float *G; int64_t ldg;     // upstream gradient rows [orow][ldg], first N used
    int N;
    long long M; int mchunk;
    float *dW;                       // [N][K] (one row chunk only) or the partial planes [zs][nbatch][N][K] (zstride apart)
    long long zstride;
    unsigned kt, nt;
    long long bsA, bsG, bsD;         // batched launch (gridDim.y): element strides of A, G, dW per batch entry
};

// TN = 128: waves 2(n) x 2(k), each 64x64;  TN = 64 (layers with <= 64 outputs): waves 1 x 4, each 64(n) x 32(k)
template <int MODE, int TN>
__global__ void __launch_bounds__(256)
k_gather_wgrad(const WArgs p0) {
    WArgs p = p0;
    p.A += (long long)blockIdx.y * p0.bsA; p.G += (long long)blockIdx.y * p0.bsG; p.dW += (long long)blockIdx.y * p0.bsD;
    constexpr int TJ = TN == 128 ? 2 : 1;        // k tiles per wave
    constexpr int NGQ = TN == 128 ? 4 : 2;       // G float4 slots per thread
    __shared__ __attribute__((aligned(16))) float Gs[TM * TN];
    __shared__ __attribute__((aligned(16))) float As[TM * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = TN == 128 ? (wave >> 1) : 0, wk = TN == 128 ? (wave & 1) : wave;
    const int l31 = lane & 31, lh = lane >> 5;
    // XCD-aware order (see k_wino_wgrad): all (k, n) blocks of one m-chunk run back to back on one XCD and share its L2
    const unsigned nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const unsigned lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
    const unsigned per_z = p.kt * p.nt, bz = lin / per_z, rem = lin - bz * per_z;
    const unsigned by = rem / p.kt, bx = rem - by * p.kt;
    const int k0 = bx * TK, n0 = by * TN;
    const long long mbeg = (long long)bz * p.mchunk;
    long long mend = mbeg + p.mchunk;
    if (mend > p.M) mend = p.M;
    if (mbeg >= mend) return;
    p.dW += (long long)bz * p.zstride;           // this row chunk's partial plane (combined in a fixed order by k_fold_splits)

    // staging: thread -> rows (tid>>5) + 8q, columns (tid&31)*4 .. +3
    const int r0 = tid >> 5, c4 = (tid & 31) * 4;
    const int kk = k0 + c4;
    const bool kin = kk < p.K;
    const int t = (int)(((unsigned long long)kk * p.magicC) >> 32);
    const int c = kk - t * p.C;
    const int dh = (int)((p.dhpack >> (4 * (t & 15))) & 15) - 8, dw = (int)((p.dwpack >> (4 * (t & 15))) & 15) - 8;
    // G staging: TN/4 float4 columns per row
    const int gc4 = (tid % (TN / 4)) * 4, gr0 = tid / (TN / 4);
    constexpr int GRS = 256 / (TN / 4);          // rows covered per G slot pass (8 or 16)
    const bool nin = (n0 + gc4) < p.N;

    // mode 1: every staged row carries its position (b, i, j) in the virtual output grid, the pointer to its gradient row and the
    // pointer to the tap's input pixel.  A step advances all three by TM pixels of the same grid row - three adds; only when j runs
    // past the row end (once per Wv / TM steps, all rows of a step within one or two steps of each other) are they re-derived with
    // the 64-bit products that used to be paid per row and step (~90 multiply instructions per step next to 64 MFMAs)
    struct RowIt { int j, i; long long b; const float *g, *a; bool rowok; };
    auto point = [&](RowIt &it, int gcol) {                          // pointers and row validity from (b, i, j)
        const long long orow = (it.b * p.Ho + (it.i * p.osh + p.oh0)) * p.Wo + (it.j * p.osw + p.ow0);
        it.g = p.G + orow * p.ldg + gcol;
        const int ih = it.i * p.sh + dh;
        it.rowok = (unsigned)ih < (unsigned)p.Hin;
        it.a = p.A + ((it.b * p.Hin + ih) * p.Win + (it.j * p.sw + dw)) * p.lda + c;
    };
    auto place = [&](RowIt &it, long long m, int gcol) {             // from the row number (start of the chunk)
        it.j = (int)(m % p.Wv); const long long r = m / p.Wv;
        it.i = (int)(r % p.Hv); it.b = r / p.Hv;
        point(it, gcol);
    };
    const long long gstep = (long long)TM * p.osw * p.ldg, astep = (long long)TM * p.sw * p.lda;
    const bool narrow = p.Wv < 4 * TM;           // grid rows of less than four steps wrap so often that the test is not worth it
    auto advance = [&](RowIt &it, int gcol) {
        it.j += TM; it.g += gstep; it.a += astep;
        if (narrow || it.j >= p.Wv) {
            while (it.j >= p.Wv) { it.j -= p.Wv; ++it.i; }
            while (it.i >= p.Hv) { it.i -= p.Hv; ++it.b; }
            point(it, gcol);
        }
    };
    RowIt rit[4], git[NGQ];
    if (MODE == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) place(rit[q], mbeg + r0 + 8 * q, n0 + c4);
        if (TN != 128) {
#pragma unroll
            for (int q = 0; q < NGQ; ++q) place(git[q], mbeg + gr0 + GRS * q, n0 + gc4);
        }
    }
    float4 rg[NGQ], ra[4];
    auto load_step = [&](long long ms) {
        if (TN != 128) {                         // G rows are staged on their own (row, column) grid
#pragma unroll
            for (int q = 0; q < NGQ; ++q) {
                long long m = ms + gr0 + GRS * q;
                float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < mend && nin) g = *reinterpret_cast<const float4 *>(MODE == 1 ? git[q].g : p.G + m * p.ldg + n0 + gc4);
                rg[q] = g;
                if (MODE == 1) advance(git[q], n0 + gc4);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            long long m = ms + r0 + 8 * q;
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f), a = g;
            if (m < mend) {
                if (MODE == 1) {
                    if (TN == 128 && nin) g = *reinterpret_cast<const float4 *>(rit[q].g);
                    if (kin && rit[q].rowok && (unsigned)(rit[q].j * p.sw + dw) < (unsigned)p.Win)
                        a = *reinterpret_cast<const float4 *>(rit[q].a);
                } else {
                    long long arow = -1;
                    if (MODE == 0) arow = m;
                    else if (kin) arow = p.table[m * 16 + t];
                    if (TN == 128 && nin) g = *reinterpret_cast<const float4 *>(p.G + m * p.ldg + n0 + c4);
                    if (kin && arow >= 0) a = *reinterpret_cast<const float4 *>(p.A + arow * p.lda + c);
                }
            }
            if (TN == 128) rg[q] = g;
            ra[q] = a;
            if (MODE == 1) advance(rit[q], n0 + c4);
        }
    };

    f32x16 acc[2][TJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_step(mbeg);
    for (long long ms = mbeg; ms < mend; ms += TM) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4 *>(&As[(r0 + 8 * q) * LD + c4]) = ra[q];
#pragma unroll
        for (int q = 0; q < NGQ; ++q)
            *reinterpret_cast<float4 *>(&Gs[(gr0 + GRS * q) * TN + gc4]) = rg[q];
        __syncthreads();
        if (ms + TM < mend) load_step(ms + TM);
#pragma unroll
        for (int mm = 0; mm < TM / 2; ++mm) {
            float g[2], a[TJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) g[i] = Gs[(mm * 2 + lh) * TN + (wn * 2 + i) * 32 + l31];
#pragma unroll
            for (int j = 0; j < TJ; ++j) a[j] = As[(mm * 2 + lh) * LD + (wk * TJ + j) * 32 + l31];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[i], a[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // D[row = n][col = k]; lanes run along k (contiguous in dW)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int k = k0 + (wk * TJ + j) * 32 + l31;
            if (k >= p.K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + (wn * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (n < p.N) p.dW[(long long)n * p.K + k] = acc[i][j][r];
            }
        }
}

// dst[i] = sum_z part[z][i]: the row-chunk partials of a weight gradient combined in a FIXED order (round 1 used fp32 atomics on a
// zeroed buffer: run-to-run different sums).  The planes are small (N*K floats) and there can be hundreds of them, so the sum over
// z is itself spread over 16 lanes per element: lane l adds planes l, l+16, ... in order, an LDS tree (fixed shape) adds the lanes.
// dst may be plane 0 of `part` (every element is read and written by one workgroup only, reads before the barrier).
// UNPACK (round 5): the folded element goes straight to its place in the reference layout, W.flat[n*sn + c*sc + tap[t]*st] (what
// k_unpack_weight did in a second launch from the packed [N][T][Cp] plane) - same sums, same order, one launch and one plane less
typedef efgh_wgrad_out_desc FoldUnpackArgs;

template <bool UNPACK>
__global__ void __launch_bounds__(256) k_fold_splits(const float4 *part, int zs, long long total4, float4 *dst, const FoldUnpackArgs u) {
    __shared__ float4 red[256];
    const int e = threadIdx.x & 15, zl = threadIdx.x >> 4;
    const long long i = (long long)blockIdx.x * 16 + e;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (i < total4) {
        int z = zl;
        for (; z + 16 < zs; z += 32) {
            const float4 u = part[(long long)z * total4 + i], v = part[(long long)(z + 16) * total4 + i];
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
        }
        if (z < zs) { const float4 u = part[(long long)z * total4 + i]; a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w; }
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    red[threadIdx.x] = a;
    __syncthreads();
    for (int h = 8; h; h >>= 1) {
        if (zl < h) {
            const float4 o = red[threadIdx.x + h * 16];
            float4 m = red[threadIdx.x];
            m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
            red[threadIdx.x] = m;
        }
        __syncthreads();
    }
    if (zl == 0 && i < total4) {
        if (!UNPACK) { dst[i] = red[e]; return; }
        const float4 v4 = red[e];
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
        const long long idx = 4 * i;                       // packed index of the quad's first element: (n, t, c), Cp % 4 == 0
        const int c0 = (int)(idx % u.Cp); const long long r = idx / u.Cp;
        const int t = (int)(r % u.T), n = (int)(r / u.T);
        if (n >= u.N) return;
        int ti = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) if (q == t) ti = u.taps[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (c0 + q >= u.C) continue;
            float *d = &u.W[n * u.sn + (c0 + q) * u.sc + ti * u.st];
            *d = u.accumulate ? *d + v[q] : v[q];
        }
    }
}

// W.flat[n*sn + c*sc + tap[t]*st] = Wp[n][t][c]   (inverse of k_pack_weight; Wp may be padded to ldn/ldc)
__global__ void k_unpack_weight(const float *__restrict__ Wp, float *__restrict__ W, int N, int T, int C,
                                int Cp, long long sn, long long sc, long long st, const int4 taps0,
                                const int4 taps1, const int4 taps2, const int4 taps3, int accumulate) {
    const int tp[16] = {taps0.x, taps0.y, taps0.z, taps0.w, taps1.x, taps1.y, taps1.z, taps1.w,
                        taps2.x, taps2.y, taps2.z, taps2.w, taps3.x, taps3.y, taps3.z, taps3.w};
    long long total = (long long)N * T * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int c = (int)(i % C); long long r = i / C;
        int t = (int)(r % T); int n = (int)(r / T);
        int ti = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) if (q == t) ti = tp[q];
        float v = Wp[((long long)n * T + t) * Cp + c];
        float *dst = &W[n * sn + c * sc + ti * st];
        *dst = accumulate ? *dst + v : v;
    }
}

// scatter-add of rows through a neighbour table: dst[table[m][t]][c] += src[m][t*C + c]  (t < T)
__global__ void __launch_bounds__(256)
k_table_scatter_add(const float *__restrict__ src, const int *__restrict__ table, long long M, int T, int C,
                    float *__restrict__ dst) {
    const int c4n = C >> 2;
    long long total = M * T * c4n;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        int cq = (int)(i % c4n); long long r = i / c4n;
        int t = (int)(r % T); long long m = r / T;
        int row = table[m * 16 + t];
        if (row < 0) continue;
        float4 v = *reinterpret_cast<const float4 *>(src + (m * T + t) * C + cq * 4);
        float *d = dst + (long long)row * C + cq * 4;
        atomicAdd(d + 0, v.x); atomicAdd(d + 1, v.y); atomicAdd(d + 2, v.z); atomicAdd(d + 3, v.w);
    }
}

}  // namespace

// (also used by efgh_wino_wgrad, wino.hip, and the dedicated kernels' weight gradients)
bool efgh_launch_fold_splits(const float *part, int zs, long long total, float *dst, hipStream_t st, const efgh_wgrad_out_desc *out) {
    const long long total4 = total / 4;
    const unsigned grid = (unsigned)((total4 + 15) / 16);
    if (out && out->W && out->T >= 1 && out->T <= 16 && out->Cp >= out->C && out->Cp % 4 == 0) {
        const long long row = (long long)out->T * out->Cp;
        if (total % row == 0 && total / row >= out->N && total / row < out->N + 4) {      // (Np = N rounded up to 4 rows)
            k_fold_splits<true><<<grid, 256, 0, st>>>((const float4 *)part, zs, total4, (float4 *)dst, *out);
            return true;
        }
    }
    FoldUnpackArgs none = {};
    k_fold_splits<false><<<grid, 256, 0, st>>>((const float4 *)part, zs, total4, (float4 *)dst, none);
    return false;
}

static int gather_wgrad_impl(const efgh_gemm_desc *d, const float *G, int64_t ldg, float *dWp, float *workspace, int nbatch,
                             int64_t bsG, int64_t bsD, const efgh_wgrad_out_desc *out, void *stream_);


// the rows m are cut into `zs` chunks that fill whole rounds of resident workgroups (efgh_round_chunks, common.h; round 5 - rounds
// 1-4 aimed at ~1024 workgroups whatever the layer).  The occupancy is asked of the runtime.  Chunks are multiples of TM.
static long long wgrad_chunks(const efgh_gemm_desc *d, int nbatch, long long *chunk_out) {
    const int K = d->T * d->C;
    const int TN = d->N <= 64 ? 64 : 128;
    const long long kt = (K + TK - 1) / TK, nt = (d->N + TN - 1) / TN;
    static const int occ128[3] = {efgh_wg_per_cu((const void *)k_gather_wgrad<0, 128>, 256, 0), efgh_wg_per_cu((const void *)k_gather_wgrad<1, 128>, 256, 0),
                                  efgh_wg_per_cu((const void *)k_gather_wgrad<2, 128>, 256, 0)};
    static const int occ64[3] = {efgh_wg_per_cu((const void *)k_gather_wgrad<0, 64>, 256, 0), efgh_wg_per_cu((const void *)k_gather_wgrad<1, 64>, 256, 0),
                                 efgh_wg_per_cu((const void *)k_gather_wgrad<2, 64>, 256, 0)};
    const int mode = d->mode >= 0 && d->mode <= 2 ? d->mode : 0;
    return efgh_round_chunks(d->M, kt * nt * nbatch, TN == 128 ? occ128[mode] : occ64[mode], TM, 256, (double)nbatch * d->N * K, chunk_out);
}

/* floats of scratch efgh_gather_wgrad(_batched) needs for this problem (0: a single row chunk writes dWp directly) */
extern "C" int64_t efgh_gather_wgrad_workspace(const efgh_gemm_desc *d) {
    if (!d || d->C <= 0 || d->T < 1 || d->N < 1 || d->M < 1) return 0;
    const int nbatch = d->nbatch > 1 ? d->nbatch : 1;
    const long long zs = wgrad_chunks(d, nbatch, nullptr);
    return zs > 1 ? zs * nbatch * (int64_t)d->N * d->T * d->C : 0;
}

extern "C" int efgh_gather_wgrad(const efgh_gemm_desc *d, const float *G, int64_t ldg, float *dWp, float *workspace,
                                 const efgh_wgrad_out_desc *out, void *stream_) {
    return gather_wgrad_impl(d, G, ldg, dWp, workspace, 1, 0, 0, out, stream_);
}

/* the same contraction for d->nbatch independent problems in ONE launch (mode 0): problem b reads A + b*d->batch_stride_a and
 * G + b*batch_stride_g and writes dWp + b*batch_stride_dw  (the 36 alpha planes of the 2-D Winograd weight gradient) */
extern "C" int efgh_gather_wgrad_batched(const efgh_gemm_desc *d, const float *G, int64_t ldg, int64_t batch_stride_g,
                                         float *dWp, int64_t batch_stride_dw, float *workspace, void *stream_) {
    EFGH_CHECK_ARG(d && d->mode == 0 && d->nbatch >= 1 && d->nbatch <= 65535);
    EFGH_CHECK_ARG(d->batch_stride_a % 4 == 0 && batch_stride_g % 4 == 0 && batch_stride_dw == (int64_t)d->N * d->C);
    return gather_wgrad_impl(d, G, ldg, dWp, workspace, d->nbatch, batch_stride_g, batch_stride_dw, nullptr, stream_);
}

static int gather_wgrad_impl(const efgh_gemm_desc *d, const float *G, int64_t ldg, float *dWp, float *workspace, int nbatch,
                             int64_t bsG, int64_t bsD, const efgh_wgrad_out_desc *out, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(d && d->A && G && dWp);
    EFGH_CHECK_ARG(d->C > 0 && d->C % 4 == 0 && d->T >= 1 && d->T <= 16 && d->N >= 1 && d->M >= 1);
    EFGH_CHECK_ARG(d->lda % 4 == 0 && ldg % 4 == 0 && d->N % 4 == 0);
    EFGH_CHECK_ARG((((uintptr_t)d->A) & 15) == 0 && (((uintptr_t)G) & 15) == 0 && (((uintptr_t)dWp) & 15) == 0);
    EFGH_CHECK_ARG((int64_t)d->T * d->C < 65536 && d->mode >= 0 && d->mode <= 2);
    WArgs a;
    a.A = d->A; a.lda = d->lda; a.C = d->C; a.T = d->T; a.K = d->T * d->C;
    a.magicC = (unsigned)((0x100000000ULL + d->C - 1) / d->C);
    a.Hin = d->Hin; a.Win = d->Win; a.Hv = d->Hv; a.Wv = d->Wv; a.sh = d->sh; a.sw = d->sw;
    a.dhpack = 0; a.dwpack = 0;
    for (int t = 0; t < 16; ++t) {
        int dh = t < d->T ? d->dh[t] : 0, dw = t < d->T ? d->dw[t] : 0;
        a.dhpack |= (unsigned long long)((dh + 8) & 15) << (4 * t);
        a.dwpack |= (unsigned long long)((dw + 8) & 15) << (4 * t);
    }
    a.Ho = d->Ho; a.Wo = d->Wo; a.osh = d->osh; a.osw = d->osw; a.oh0 = d->oh0; a.ow0 = d->ow0;
    a.table = d->table; a.G = G; a.ldg = ldg; a.N = d->N; a.M = d->M;
    a.bsA = nbatch > 1 ? d->batch_stride_a : 0; a.bsG = bsG; a.bsD = bsD;
    if (d->mode == 1) EFGH_CHECK_ARG(d->M == (int64_t)d->B * d->Hv * d->Wv && d->osh >= 1 && d->osw >= 1);
    if (d->mode == 2) EFGH_CHECK_ARG(d->table != nullptr);
    const int TN = a.N <= 64 ? 64 : 128;
    const int kt = (a.K + TK - 1) / TK, nt = (a.N + TN - 1) / TN;
    long long chunk = 0;
    const long long zs = wgrad_chunks(d, nbatch, &chunk);
    a.mchunk = (int)chunk;
    EFGH_CHECK_ARG(zs * kt * nt < 0x7fffffffLL);
    a.kt = (unsigned)kt; a.nt = (unsigned)nt;
    // every row chunk writes its own [nbatch][N][K] plane with plain stores (each element of a plane is written by exactly one
    // workgroup); k_fold_splits then adds the planes in chunk order: no memset, no atomics, bit-reproducible
    const long long plane = (long long)nbatch * a.N * a.K;
    EFGH_CHECK_ARG(zs == 1 || (workspace && (((uintptr_t)workspace) & 15) == 0));
    a.dW = zs > 1 ? workspace : dWp;
    a.zstride = zs > 1 ? plane : 0;
    const dim3 grid((unsigned)(zs * kt * nt), (unsigned)nbatch);
    if (TN == 128) {
        if (d->mode == 0) k_gather_wgrad<0, 128><<<grid, 256, 0, st>>>(a);
        else if (d->mode == 1) k_gather_wgrad<1, 128><<<grid, 256, 0, st>>>(a);
        else k_gather_wgrad<2, 128><<<grid, 256, 0, st>>>(a);
    } else {
        if (d->mode == 0) k_gather_wgrad<0, 64><<<grid, 256, 0, st>>>(a);
        else if (d->mode == 1) k_gather_wgrad<1, 64><<<grid, 256, 0, st>>>(a);
        else k_gather_wgrad<2, 64><<<grid, 256, 0, st>>>(a);
    }
    bool wrote = false;
    if (zs > 1) wrote = efgh_launch_fold_splits(workspace, (int)zs, plane, dWp, st, nbatch == 1 ? out : nullptr);       // (N % 4 == 0)
    EFGH_CHECK_LAUNCH();
    return wrote ? EFGH_WROTE_OUT : EFGH_OK;
}

extern "C" int efgh_unpack_weight(const float *Wp, float *W, int32_t N, int32_t T, int32_t C, int32_t Cp,
                                  int64_t sn, int64_t sc, int64_t stt, const int32_t *tapidx, int32_t accumulate,
                                  void *stream_) {
    EFGH_CHECK_ARG(Wp && W && N > 0 && T > 0 && T <= 16 && C > 0 && Cp >= C);
    int tp[16];
    for (int t = 0; t < 16; ++t) tp[t] = (tapidx && t < T) ? tapidx[t] : (t < T ? t : 0);
    long long total = (long long)N * T * C;
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    k_unpack_weight<<<grid, 256, 0, (hipStream_t)stream_>>>(Wp, W, N, T, C, Cp, sn, sc, stt,
                                                           make_int4(tp[0], tp[1], tp[2], tp[3]),
                                                           make_int4(tp[4], tp[5], tp[6], tp[7]),
                                                           make_int4(tp[8], tp[9], tp[10], tp[11]),
                                                           make_int4(tp[12], tp[13], tp[14], tp[15]), accumulate);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_table_scatter_add(const float *src, const int32_t *table, int64_t M, int32_t T, int32_t C,
                                      float *dst, void *stream_) {
    EFGH_CHECK_ARG(src && table && dst && M > 0 && T > 0 && T <= 16 && C % 4 == 0);
    long long total = M * T * (C / 4);
    long long g = (total + 255) / 256;
    k_table_scatter_add<<<(int)(g > 16384 ? 16384 : g), 256, 0, (hipStream_t)stream_>>>(src, table, M, T, C, dst);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

/* debugging aid (tools/probes/occupancy.py): the occupancy figures the chunking uses, as text */
extern "C" int efgh_debug_occupancy(char *out, int32_t cap) {
    EFGH_CHECK_ARG(out && cap > 64);
    snprintf(out, cap, "k_gather_wgrad<0,128> %d  <1,128> %d  <2,128> %d  <0,64> %d  <1,64> %d  <2,64> %d workgroups per CU",
             efgh_wg_per_cu((const void *)k_gather_wgrad<0, 128>, 256, 0), efgh_wg_per_cu((const void *)k_gather_wgrad<1, 128>, 256, 0),
             efgh_wg_per_cu((const void *)k_gather_wgrad<2, 128>, 256, 0), efgh_wg_per_cu((const void *)k_gather_wgrad<0, 64>, 256, 0),
             efgh_wg_per_cu((const void *)k_gather_wgrad<1, 64>, 256, 0), efgh_wg_per_cu((const void *)k_gather_wgrad<2, 64>, 256, 0));
    return EFGH_OK;
}
