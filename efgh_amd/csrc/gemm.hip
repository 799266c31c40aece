// Gather-GEMM on fp32 MFMA (v_mfma_f32_32x32x2_f32) for gfx950 — K4 / K6 / K8 of SURVEY.md §2a.
//
//   out[m][n] = act( scale[n]*(sum_{t<T,c<C} A[row(m,t)][c] * W[n][t*C+c] + bias[n]) + shift[n]
//                    + residual[m][n] )
//
// One kernel family serves every dense contraction of the reference:
//   Conv2d 3x3 / 1x1 / (1,2), stride 1|2          nets/vgg.py:77, nets/resnet.py:22-30, net_utils.py:49
//   ConvTranspose2d k3 s2 (as 4 output-parity classes)               nets/net_utils.py:72-79
//   Conv1d k1 / Linear                                               nets/enet.py:85-97, hnet.py:19-31
//   neighbour gather + Conv2d(C,C',(15,1))  (BCL blur)               nets/bilateralNN.py:240-246
// A rows are gathered on the fly (implicit GEMM): `row(m,t)` is an input pixel (mode 1), a lattice
// neighbour (mode 2) or m itself (mode 0); out-of-image / missing rows contribute zeros.
//
// Tiling: 256 threads = 4 waves; block tile BM x BN x 32; each wave owns (BM/WM) x (BN/WN) as
// 32x32 MFMA tiles with fp32 accumulators.  Global -> registers -> LDS staging (16-B vectors
// along the channel axis, 128-B rows), LDS rows padded to 36 floats so that both the
// ds_write_b128 of the staging pass and the ds_read_b128 fragment reads are bank-conflict free.
// One ds_read_b128 feeds four MFMAs: lanes 0-31 hold k = g*8+j, lanes 32-63 hold k = g*8+4+j.
// Numerics: exact fp32 products, fp32 accumulation (a k-ordered fma chain per output).
#include "common.h"
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;       // fp32 path: floats per LDS row
struct KArgs {
    const float *A; int64_t lda;
    int C, T, K;
    unsigned magicC;          // ceil(2^32 / C)
    int Hin, Win, Hv, Wv, sh, sw;
    unsigned long long dhpack, dwpack;   // 16 taps x 4 bits, value + 8
    int Ho, Wo, osh, osw, oh0, ow0;
    const int *table;
    const float *W; int N;
    long long M; const int *M_dev;
    const float *bias, *scale, *shift, *residual; int64_t ldr;
    int act; float slope;
    float *out; int64_t ldo;
    float *stats;
    unsigned nbx;              // n-tiles per m-tile (set by the launcher)
    unsigned nbatch;
    long long bsA, bsW, bsO;   // batched launch: per-problem element strides (blockIdx.y = problem)
    int bsT;                   // ... and neighbour-table COLUMN offset per problem (mode 2: problem z takes taps z*bsT ..)
    int tam;                   // mode 2: taps marked in column 15 of the table row read zeros
};

// epilogue shared by the register-staged and the DMA-staged kernels: bias, BatchNorm scale / shift, residual, activation, per-tile
// column statistics.  `scratch`: >= 2 * WM * BN floats of LDS nobody reads any more (the column statistics of the waves)
template <int MODE, int BM, int BN, int WM, int WN>
__device__ __forceinline__ void gemm_epilogue(const KArgs &p, const long long M, const long long m0, const int n0, const unsigned tile_m,
                                              f32x16 (&acc)[BM / WM / 32][BN / WN / 32], const long long *rowout, float *scratch,
                                              const int wm, const int wn, const int l31, const int lh, const int tid) {
    constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
    float *As = scratch;
    // ---- epilogue -----------------------------------------------------------------------
    // rows outside, columns inside: the output row index, its two 64-bit row addresses and the mask are formed once per row (16 MI
    // of them per lane) instead of once per element, the per-column terms once per column block
    float *ssum = As;            // [WM][BN] column sums, [WM][BN] sums of squares (LDS reuse)
    float *ssq = As + WM * BN;
    float bi[NI], sc[NI], sf[NI], s1[NI], s2[NI];
    bool cok[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int col = n0 + (wn * NI + j) * 32 + l31;
        cok[j] = col < p.N;
        bi[j] = (p.bias && cok[j]) ? p.bias[col] : 0.f;
        sc[j] = (p.scale && cok[j]) ? p.scale[col] : 1.f;
        sf[j] = (p.shift && cok[j]) ? p.shift[col] : 0.f;
        s1[j] = 0.f; s2[j] = 0.f;
    }
    const int colb = n0 + wn * NI * 32 + l31;
    const float neg = p.act == 1 ? 0.f : (p.act == 2 ? p.slope : 1.f);      // act(v) = max(v, 0) + neg * min(v, 0)
    // two workgroup-uniform switches pick one of four straight-line bodies: with / without a residual operand (its NI loads of a row
    // are issued together, ahead of the arithmetic), and the whole tile inside M x N (no row mask, no per-element predicate) or not
    auto rows = [&](auto res_c, auto full_c) {
        constexpr bool RES = decltype(res_c)::value, FULL = decltype(full_c)::value;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (wm * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const long long orow = rowout[rl];
                if (!FULL && orow < 0) continue;
                float *op = p.out + orow * p.ldo + colb;
                float rv[NI];
                if (RES) {
                    const float *rp = p.residual + orow * p.ldr + colb;
#pragma unroll
                    for (int j = 0; j < NI; ++j) rv[j] = (FULL || cok[j]) ? rp[j * 32] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    float v = acc[i][j][r] + bi[j];
                    if (FULL || cok[j]) { s1[j] += v; s2[j] = fmaf(v, v, s2[j]); }      // (explicit fma: the rounding must not follow the vectoriser)
                    v = v * sc[j] + sf[j];
                    if (RES) v += rv[j];
                    v = act_neg(v, neg);
                    if (FULL || cok[j]) st_out(op + j * 32, v);
                }
            }
        }
    };
    const bool fulln = n0 + BN <= p.N && m0 + BM <= M;      // the whole tile exists: no row mask, no column predicate
    if (fulln && !p.bias && !p.scale && !p.shift && !p.residual && !p.stats && p.act == 0) {
        // bare products (the 36 planes of a 2-D Winograd layer, the GEMM + col2im heads): one store per element
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (wm * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float *op = p.out + rowout[rl] * p.ldo + colb;
#pragma unroll
                for (int j = 0; j < NI; ++j) st_out(op + j * 32, acc[i][j][r]);
            }
        }
    } else if (p.residual) {
        if (fulln) rows(std::true_type{}, std::true_type{});
        else rows(std::true_type{}, std::false_type{});
    } else {
        if (fulln) rows(std::false_type{}, std::true_type{});
        else rows(std::false_type{}, std::false_type{});
    }
    if (p.stats) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int coll = (wn * NI + j) * 32 + l31;
            s1[j] += __shfl_xor(s1[j], 32); s2[j] += __shfl_xor(s2[j], 32);
            if (lh == 0) { ssum[wm * BN + coll] = s1[j]; ssq[wm * BN + coll] = s2[j]; }
        }
    }
    if (p.stats) {
        __syncthreads();
        for (int c = tid; c < BN; c += 256) {
            if (n0 + c < p.N) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) { a += ssum[w * BN + c]; b += ssq[w * BN + c]; }
                p.stats[((long long)tile_m * 2 + 0) * p.N + n0 + c] = a;
                p.stats[((long long)tile_m * 2 + 1) * p.N + n0 + c] = b;
            }
        }
    }
}

// exact fp32 MFMA (v_mfma_f32_32x32x2_f32): fp32 operands, products and accumulation
template <int MODE, int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(256, 3)
k_gather_gemm(const KArgs p_in) {
    constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
    constexpr int NA = BM / 32, NB = BN / 32;
    __shared__ __attribute__((aligned(16))) float As[BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDS_LD];
    __shared__ long long rowout[BM];          // output row (pixel) index per tile row, -1 = masked

    KArgs p = p_in;
    if (blockIdx.y) {              // batched launch: shift the operand pointers to problem blockIdx.y
        p.A += blockIdx.y * p.bsA; p.W += blockIdx.y * p.bsW; p.out += blockIdx.y * p.bsO;
        if (MODE == 2) p.table += blockIdx.y * p.bsT;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;
    long long M = p.M;
    if (p.M_dev) { long long md = *p.M_dev; M = md < M ? md : M; }
    // XCD-aware tile order: the dispatcher deals consecutive workgroups round-robin over the 8 XCDs, so
    // workgroup b and b+8 share an L2.  Give every XCD one contiguous band of output tiles (n fastest,
    // then m), so that vertically adjacent image tiles - which re-read each other's halo rows - hit the
    // same L2 instead of each fetching them from the fabric.  Pure speed: any placement is correct.
    const unsigned nbx = p.nbx, nblk = gridDim.x;
    const unsigned q8 = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const unsigned lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
    const unsigned tile_m = lin / nbx, tile_n = lin - tile_m * nbx;
    const long long m0 = (long long)tile_m * BM;
    const int n0 = tile_n * BN;
    if (m0 >= M) return;

    // ---- per-thread row state for the staging loads --------------------------------------
    // mode 0: abase = row offset (floats) | mode 1: abase = centre pixel offset (floats), amask bit t = tap t in bounds
    // mode 2: abase = m*16 (neighbour-table row)
    const int arow = tid >> 3, kv = (tid & 7) * 4;
    long long abase[NA];
    unsigned amask[NA];
#pragma unroll
    for (int q = 0; q < NA; ++q) {
        long long m = m0 + q * 32 + arow;
        const bool ok = m < M;
        abase[q] = 0;
        amask[q] = 0;
        if (ok) {
            if (MODE == 0) { abase[q] = m * p.lda; amask[q] = 1; }
            else if (MODE == 2) {
                abase[q] = m * 16;
                amask[q] = 0xFFFFu;
                if (p.tam) amask[q] = (0x7FFFu & ~(unsigned)p_in.table[m * 16 + 15]) >> (blockIdx.y * p.bsT);
            }
            else if (MODE == 3) {            // row stack: tap t = image row t, window of C floats starting at pixel j
                long long b = m / p.Wv; int j = (int)(m - b * p.Wv);
                abase[q] = b * p.Hin * p.Win + j; amask[q] = 1;
            } else {
                int j = (int)(m % p.Wv); long long r = m / p.Wv;
                int i = (int)(r % p.Hv); long long b = r / p.Hv;
                const int ih0 = i * p.sh, iw0 = j * p.sw;
                abase[q] = ((b * p.Hin + ih0) * p.Win + iw0) * p.lda;
                unsigned mk = 0;
                for (int t = 0; t < p.T; ++t) {
                    int ih = ih0 + (int)((p.dhpack >> (4 * t)) & 15) - 8, iw = iw0 + (int)((p.dwpack >> (4 * t)) & 15) - 8;
                    if ((unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win) mk |= 1u << t;
                }
                amask[q] = mk;
            }
        }
    }
    for (int r = tid; r < BM; r += 256) {
        long long m = m0 + r, o = -1;
        if (m < M) {
            if (MODE == 1) {
                int j = (int)(m % p.Wv); long long rr = m / p.Wv;
                int i = (int)(rr % p.Hv); long long b = rr / p.Hv;
                o = (b * p.Ho + (i * p.osh + p.oh0)) * p.Wo + (j * p.osw + p.ow0);
            } else o = m;
        }
        rowout[r] = o;
    }
    long long bbase[NB]; bool bval[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        int n = n0 + q * 32 + arow;
        bval[q] = n < p.N;
        bbase[q] = (long long)n * p.K;
    }

    float4 ra[NA], rb[NB];
    auto load_chunk = [&](int k0) {
        const int kk = k0 + kv;
        const bool kin = kk < p.K;
        int t = (int)(((unsigned long long)kk * p.magicC) >> 32);
        int c = kk - t * p.C;
        if (MODE == 1) {
            // branch-free: out-of-image taps read a valid dummy address (row 0) and are zeroed by select
            const int dh = (int)((p.dhpack >> (4 * (t & 15))) & 15) - 8, dw = (int)((p.dwpack >> (4 * (t & 15))) & 15) - 8;
            const long long dl = ((long long)dh * p.Win + dw) * p.lda + c;      // (abase already carries the factor lda: one product
#pragma unroll                                                               //  per chunk instead of one per row)
            for (int q = 0; q < NA; ++q) {
                const bool ok = kin && ((amask[q] >> t) & 1u);
                const long long off = ok ? abase[q] + dl : 0;
                float4 v = *reinterpret_cast<const float4 *>(p.A + off);
                ra[q] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else if (MODE == 3) {
            const long long delta = (long long)t * p.Win;
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                const bool ok = kin && amask[q];
                float4 v = *reinterpret_cast<const float4 *>(p.A + (ok ? (abase[q] + delta) * p.lda + c : 0));
                ra[q] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kin && (MODE == 2 ? ((amask[q] >> t) & 1u) : amask[q])) {
                    if (MODE == 0) {
                        v = *reinterpret_cast<const float4 *>(p.A + abase[q] + kk);
                    } else {
                        int r = p.table[abase[q] + t];
                        if (r >= 0) v = *reinterpret_cast<const float4 *>(p.A + (long long)r * p.lda + c);
                    }
                }
                ra[q] = v;
            }
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const bool ok = bval[q] && kin;
            float4 v = *reinterpret_cast<const float4 *>(p.W + (ok ? bbase[q] + kk : 0));
            rb[q] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunks = (p.K + BK - 1) / BK;
    load_chunk(0);
    for (int ch = 0; ch < nchunks; ++ch) {
#pragma unroll
        for (int q = 0; q < NA; ++q)
            *reinterpret_cast<float4 *>(&As[(q * 32 + arow) * LDS_LD + kv]) = ra[q];
#pragma unroll
        for (int q = 0; q < NB; ++q)
            *reinterpret_cast<float4 *>(&Bs[(q * 32 + arow) * LDS_LD + kv]) = rb[q];
        __syncthreads();
        if (ch + 1 < nchunks) load_chunk((ch + 1) * BK);
        const int kleft = p.K - ch * BK;          // a ragged last chunk (K = 36 for the 4-channel input convs) skips its
#pragma unroll                                    // all-zero groups of 8
        for (int g = 0; g < 4; ++g) {
            if (g * 8 >= kleft) continue;
            float4 a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i)
                a[i] = *reinterpret_cast<const float4 *>(
                    &As[((wm * MI + i) * 32 + l31) * LDS_LD + g * 8 + lh * 4]);
#pragma unroll
            for (int j = 0; j < NI; ++j)
                b[j] = *reinterpret_cast<const float4 *>(
                    &Bs[((wn * NI + j) * 32 + l31) * LDS_LD + g * 8 + lh * 4]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    gemm_epilogue<MODE, BM, BN, WM, WN>(p, M, m0, n0, tile_m, acc, rowout, As, wm, wn, l31, lh, tid);
}

// ---- the same contraction with LDS-DMA staging (round 5): BM = 128, 2 x 2 waves, modes 0 and 1 with C % 32 == 0 (a 32-deep step
// never straddles a tap).  Staging as in planes.hip: every wave moves four 1-KiB pieces (8 rows x 128 B) of A and BN / 32 of W per
// step with `global_load_lds_dwordx4` into a two-slot ring, the 16-B quad q of row r sits at slot q ^ ((r >> 1) & 7) (conflict-free
// ds_read_b128 without row padding), ONE barrier per step; rows past M and taps outside the image fetch a zero page (the source
// address is per lane).  No VGPR staging, no ds_write pass, no second barrier - the register-staged kernel spends a fifth of its
// issue slots on them at K = 128 ... 576.  Same products in the same k order: bit-identical outputs
// (tests/test_gpu_ops.py::test_dma_gemm_equals_register_staged).
__device__ __attribute__((aligned(128))) float g_gemm_zero[32];

constexpr int DMA_BM = 128;

template <int MODE, int BN>
__global__ void __launch_bounds__(256, 2)
k_gather_gemm_dma(const KArgs p_in) {
    constexpr int BM = DMA_BM, WM = 2, WN = 2, MI = 2, NI = BN / WN / 32, NBP = BN / 32;      // NBP: W pieces per wave and step
    constexpr int STAGE = (BM + BN) * BK;
    extern __shared__ __attribute__((aligned(1024))) float ring[];          // 2 x STAGE floats
    __shared__ long long rowout[BM];

    KArgs p = p_in;
    if (blockIdx.y) { p.A += blockIdx.y * p.bsA; p.W += blockIdx.y * p.bsW; p.out += blockIdx.y * p.bsO; }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, lh = lane >> 5;
    const unsigned ring_lds = lds_addr(ring);
    const long long M = p.M;
    const unsigned nbx = p.nbx, nblk = gridDim.x;
    const unsigned q8 = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const unsigned lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
    const unsigned tile_m = lin / nbx, tile_n = lin - tile_m * nbx;
    const long long m0 = (long long)tile_m * BM;
    const int n0 = tile_n * BN;
    if (m0 >= M) return;

    // ---- DMA roles: wave w moves A pieces 4w .. 4w+3 and W pieces NBP w .. ; lane l -> row 8 piece + (l >> 3), slot l & 7
    long long abase[4]; unsigned amask[4];
    const float *srcW[NBP];
    const int sub = lane >> 3;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = (wave * 4 + q) * 8 + sub;
        const int kq = (lane & 7) ^ ((r >> 1) & 7);
        const long long m = m0 + r;
        abase[q] = kq * 4; amask[q] = 0;
        if (m < M) {
            if (MODE == 0) { abase[q] += m * p.lda; amask[q] = 1; }
            else {
                const int j = (int)(m % p.Wv); const long long rr = m / p.Wv;
                const int i = (int)(rr % p.Hv); const long long b = rr / p.Hv;
                const int ih0 = i * p.sh, iw0 = j * p.sw;
                abase[q] += ((b * p.Hin + ih0) * p.Win + iw0) * p.lda;
                unsigned mk = 0;
                for (int t = 0; t < p.T; ++t) {
                    const int ih = ih0 + (int)((p.dhpack >> (4 * t)) & 15) - 8, iw = iw0 + (int)((p.dwpack >> (4 * t)) & 15) - 8;
                    if ((unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win) mk |= 1u << t;
                }
                amask[q] = mk;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NBP; ++q) {
        const int r = (wave * NBP + q) * 8 + sub;
        const int kq = (lane & 7) ^ ((r >> 1) & 7);
        int n = n0 + r;
        if (n >= p.N) n = p.N - 1;               // (columns past N are never stored)
        srcW[q] = p.W + (long long)n * p.K + kq * 4;
    }
    for (int r = tid; r < BM; r += 256) {
        const long long m = m0 + r;
        long long o = -1;
        if (m < M) {
            if (MODE == 1) {
                const int j = (int)(m % p.Wv); const long long rr = m / p.Wv;
                const int i = (int)(rr % p.Hv); const long long b = rr / p.Hv;
                o = (b * p.Ho + (i * p.osh + p.oh0)) * p.Wo + (j * p.osw + p.ow0);
            } else o = m;
        }
        rowout[r] = o;
    }
    const float *zsrc = g_gemm_zero + (lane & 7) * 4;
    auto issue = [&](int chunk, int buf) {
        const unsigned base = ring_lds + (unsigned)(buf * STAGE) * 4u;
        const int k0 = chunk * BK;
        const int t = (int)(((unsigned long long)k0 * p.magicC) >> 32);        // (uniform: C % 32 == 0, the step lies inside tap t)
        const int c = k0 - t * p.C;
        long long dl = c;
        if (MODE == 1) {
            const int dh = (int)((p.dhpack >> (4 * (t & 15))) & 15) - 8, dw = (int)((p.dwpack >> (4 * (t & 15))) & 15) - 8;
            dl += ((long long)dh * p.Win + dw) * p.lda;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool ok = MODE == 0 ? amask[q] != 0 : ((amask[q] >> t) & 1u) != 0;
            dma16(ok ? p.A + abase[q] + dl : zsrc, base + (unsigned)((wave * 4 + q) * 1024));
        }
#pragma unroll
        for (int q = 0; q < NBP; ++q) dma16(srcW[q] + k0, base + (unsigned)(BM * BK * 4 + (wave * NBP + q) * 1024));
    };

    int offA[MI], offW[NI], swA[MI], swW[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) { const int ra = (wm * MI + i) * 32 + l31; offA[i] = ra * BK; swA[i] = (ra >> 1) & 7; }
#pragma unroll
    for (int j = 0; j < NI; ++j) { const int rw = (wn * NI + j) * 32 + l31; offW[j] = BM * BK + rw * BK; swW[j] = (rw >> 1) & 7; }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nch = p.K / BK;
    issue(0, 0);
    for (int ch = 0; ch < nch; ++ch) {
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();            // step ch is in LDS (every wave waited for its own pieces); step ch-1's slot is free
        asm volatile("" ::: "memory");
        if (ch + 1 < nch) issue(ch + 1, (ch + 1) & 1);
        const float *buf = ring + (ch & 1) * STAGE;
        float4 a[2][MI], b[2][NI];
        auto frag = [&](int g, float4 (&fa)[MI], float4 (&fb)[NI]) {
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const float4 *>(buf + offA[i] + (((g * 2 + lh) ^ swA[i]) << 2));
#pragma unroll
            for (int j = 0; j < NI; ++j) fb[j] = *reinterpret_cast<const float4 *>(buf + offW[j] + (((g * 2 + lh) ^ swW[j]) << 2));
        };
        frag(0, a[0], b[0]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) frag(g + 1, a[(g + 1) & 1], b[(g + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const float4 (&ca)[MI] = a[g & 1];
            const float4 (&cb)[NI] = b[g & 1];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[i].x, cb[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[i].y, cb[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[i].z, cb[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[i].w, cb[j].w, acc[i][j], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();            // (the epilogue's statistics scratch is the ring: every wave is done reading it)
    gemm_epilogue<MODE, BM, BN, WM, WN>(p, M, m0, n0, tile_m, acc, rowout, ring, wm, wn, l31, lh, tid);
}

__global__ void k_pack_weight(const float *__restrict__ W, float *__restrict__ Wp, int N, int T, int C, int Nw, int Cw,
                              long long sn, long long sc, long long st, const int4 taps0,
                              const int4 taps1, const int4 taps2, const int4 taps3) {
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
        Wp[i] = (n < Nw && c < Cw) ? W[n * sn + c * sc + ti * st] : 0.f;      // (N, C) may be the zero-padded extents of (Nw, Cw)
    }
}

// every packed layout of every weight in ONE launch (after an optimizer step all of them are stale): workgroup -> job by binary
// search over the prefix sums of the jobs' output sizes
__global__ void __launch_bounds__(256) k_pack_weight_batched(const efgh_pack_job *__restrict__ jobs, const long long *__restrict__ prefix,
                                                             int njobs, long long total) {
    __shared__ int job_s;
    for (long long i0 = (long long)blockIdx.x * 256; i0 < total; i0 += (long long)gridDim.x * 256) {
        // one search per workgroup (the job of its first element); only the few workgroups that straddle a job boundary let their
        // later threads step forward from there
        if (threadIdx.x == 0) {
            int lo = 0, hi = njobs - 1;                  // largest j with prefix[j] <= i0
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (prefix[mid] <= i0) lo = mid; else hi = mid - 1; }
            job_s = lo;
        }
        __syncthreads();
        int jb = job_s;
        __syncthreads();
        const long long i = i0 + threadIdx.x;
        if (i >= total) continue;
        while (jb + 1 < njobs && prefix[jb + 1] <= i) ++jb;
        const efgh_pack_job &j = jobs[jb];
        const long long e = i - prefix[jb];
        int n, t, c;
        if (e < 0x7fffffffLL) {                          // (every layout of this network: 32-bit divisions)
            const unsigned eu = (unsigned)e, r = eu / (unsigned)j.Cp;
            c = (int)(eu - r * (unsigned)j.Cp); n = (int)(r / (unsigned)j.T); t = (int)(r - (unsigned)n * (unsigned)j.T);
        } else {
            c = (int)(e % j.Cp); const long long r = e / j.Cp;
            t = (int)(r % j.T); n = (int)(r / j.T);
        }
        j.Wp[e] = (n < j.N && c < j.C) ? j.W[n * j.sn + c * j.sc + j.taps[t] * j.st] : 0.f;
    }
}

// the same through an LDS tile (8 rows n x 32 channels c x all T taps per workgroup): the flat kernel above reads W with one lane
// per output element - for the usual (out, in, kh, kw) weight that is a 36-byte stride between lanes (18 cache lines per wave
// load), for the transposed data-gradient layouts one line per lane - and ran at 1.2 TB/s (0.65 ms per training step for the
// 2 x 191 MB of layouts).  Here the reads walk W along its contiguous axis (whichever of sn / sc is smaller: 288-1152 contiguous
// bytes per row of the tile), the writes leave as 128-byte rows of Wp.  Same values, same zero padding.
constexpr int PK_TN = 8, PK_TC = 32;
__global__ void __launch_bounds__(256) k_pack_weight_tiled(const efgh_pack_job *__restrict__ jobs, const long long *__restrict__ tile_prefix,
                                                           int njobs) {
    __shared__ float tile[PK_TN * 16 * (PK_TC + 1)];
    __shared__ int job_s;
    if (threadIdx.x == 0) {
        int lo = 0, hi = njobs - 1;                  // largest j with tile_prefix[j] <= blockIdx.x
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tile_prefix[mid] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1; }
        job_s = lo;
    }
    __syncthreads();
    const efgh_pack_job &j = jobs[job_s];
    const int T = j.T, Np = j.Np, Cp = j.Cp;
    const int ctiles = (Cp + PK_TC - 1) / PK_TC;
    const int tl = (int)((long long)blockIdx.x - tile_prefix[job_s]);
    const int n0 = (tl / ctiles) * PK_TN, c0 = (tl % ctiles) * PK_TC;
    const int nelem = PK_TN * PK_TC * T;
    const bool c_inner = j.sc <= j.sn;               // W contiguous along (c, t) for a fixed n - or along (n, t) for a fixed c
    for (int e = threadIdx.x; e < nelem; e += 256) {
        const int t = e % T, r = e / T;
        int n, c;
        if (c_inner) { c = r % PK_TC; n = r / PK_TC; }
        else { n = r % PK_TN; c = r / PK_TN; }
        const int gn = n0 + n, gc = c0 + c;
        float v = 0.f;
        if (gn < j.N && gc < j.C) v = j.W[gn * j.sn + gc * j.sc + j.taps[t] * j.st];
        tile[(n * T + t) * (PK_TC + 1) + c] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nelem; e += 256) {
        const int c = e % PK_TC, r = e / PK_TC;      // r = n * T + t
        const int n = r / T, t = r - n * T;
        const int gn = n0 + n, gc = c0 + c;
        if (gn < Np && gc < Cp) j.Wp[((long long)gn * T + t) * Cp + gc] = tile[r * (PK_TC + 1) + c];
    }
}

__global__ void k_fold_planes(const float4 *__restrict__ part, int S, long long M, int N4, const float4 *__restrict__ bias, int act,
                              float slope, float *__restrict__ out, long long ldo) {
    const long long total = M * N4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / N4; const int n4 = (int)(i - m * N4);
        float4 a = part[i];
        for (int z = 1; z < S; ++z) { const float4 b = part[(long long)z * total + i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
        if (bias) { const float4 b = bias[n4]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
        if (act == 1) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
        else if (act == 2) {
            a.x = a.x > 0.f ? a.x : a.x * slope; a.y = a.y > 0.f ? a.y : a.y * slope;
            a.z = a.z > 0.f ? a.z : a.z * slope; a.w = a.w > 0.f ? a.w : a.w * slope;
        }
        *reinterpret_cast<float4 *>(out + m * ldo + n4 * 4) = a;
    }
}

__global__ void k_pad_vec(const float *__restrict__ v, int n, float *__restrict__ out, int np, float fill) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < np) out[i] = i < n ? v[i] : fill;
}

template <int MODE, int BM, int BN, int WM, int WN>
void launch(const KArgs &a0, hipStream_t st) {
    KArgs a = a0;
    a.nbx = (unsigned)((a.N + BN - 1) / BN);
    const long long nby = (a.M + BM - 1) / BM;
    k_gather_gemm<MODE, BM, BN, WM, WN><<<dim3((unsigned)(a.nbx * nby), a.nbatch), 256, 0, st>>>(a);
}

std::atomic<unsigned long long> g_dma_raised[4];
bool g_use_dma = true;          // efgh_gather_gemm_set_dma (tests, A/B runs)

// the launches the DMA-staged instances serve: modes 0 / 1, more than 32 outputs, every 32-deep step inside one tap, no device-side
// row count, 16-byte aligned rows
bool dma_ok(const KArgs &a, int mode) {
    return g_use_dma && (mode == 0 || mode == 1) && a.N > 32 && a.C % 32 == 0 && !a.M_dev && a.lda % 4 == 0 && a.K % BK == 0
           && (((uintptr_t)a.A) & 15) == 0 && (((uintptr_t)a.W) & 15) == 0;
}

template <int MODE, int BN>
int launch_dma(const KArgs &a0, hipStream_t st) {
    KArgs a = a0;
    a.nbx = (unsigned)((a.N + BN - 1) / BN);
    const long long nby = (a.M + DMA_BM - 1) / DMA_BM;
    const size_t lds = (size_t)2 * (DMA_BM + BN) * BK * sizeof(float);
    if (!efgh_raise_lds_once(g_dma_raised[MODE * 2 + (BN == 128)], (const void *)k_gather_gemm_dma<MODE, BN>, (int)lds)) {
        efgh_set_error("%s:%d: cannot raise the dynamic LDS limit of k_gather_gemm_dma", __FILE__, __LINE__);
        return EFGH_E_LAUNCH;
    }
    k_gather_gemm_dma<MODE, BN><<<dim3((unsigned)(a.nbx * nby), a.nbatch), 256, lds, st>>>(a);
    return EFGH_OK;
}

template <int MODE>
int dispatch(const KArgs &a, hipStream_t st) {
    if constexpr (MODE == 0 || MODE == 1) {
        if (dma_ok(a, MODE)) return a.N > 64 ? launch_dma<MODE, 128>(a, st) : launch_dma<MODE, 64>(a, st);
    }
    if (a.N > 64) launch<MODE, 128, 128, 2, 2>(a, st);
    else if (a.N > 32) launch<MODE, 128, 64, 2, 2>(a, st);
    else launch<MODE, 256, 32, 4, 1>(a, st);
    return EFGH_OK;
}

int tile_m(int N) { return N > 32 ? 128 : 256; }

}  // namespace

extern "C" int32_t efgh_gather_gemm_grid_m(int64_t M, int32_t N) {
    int bm = tile_m(N);
    return (int32_t)((M + bm - 1) / bm);
}

static int fill_args(const efgh_gemm_desc *d, KArgs &a) {
    EFGH_CHECK_ARG(d && d->A && d->W && d->out);
    EFGH_CHECK_ARG(d->C > 0 && d->C % 4 == 0 && d->T >= 1 && (d->T <= 16 || d->mode == 3));
    EFGH_CHECK_ARG(d->N >= 1 && d->M >= 1 && d->lda % 4 == 0);
    EFGH_CHECK_ARG((((uintptr_t)d->A) & 15) == 0 && (((uintptr_t)d->W) & 15) == 0);
    EFGH_CHECK_ARG(d->mode >= 0 && d->mode <= 3);
    EFGH_CHECK_ARG((int64_t)d->T * d->C < 65536);
    a.A = d->A; a.lda = d->lda; a.C = d->C; a.T = d->T; a.K = d->T * d->C;
    a.magicC = (unsigned)((0x100000000ULL + d->C - 1) / d->C);
    a.Hin = d->Hin; a.Win = d->Win; a.Hv = d->Hv; a.Wv = d->Wv; a.sh = d->sh; a.sw = d->sw;
    a.dhpack = 0; a.dwpack = 0;
    for (int t = 0; t < 16; ++t) {
        int dh = t < d->T ? d->dh[t] : 0, dw = t < d->T ? d->dw[t] : 0;
        if (d->mode == 1) EFGH_CHECK_ARG(dh >= -8 && dh <= 7 && dw >= -8 && dw <= 7);
        a.dhpack |= (unsigned long long)((dh + 8) & 15) << (4 * t);
        a.dwpack |= (unsigned long long)((dw + 8) & 15) << (4 * t);
    }
    a.Ho = d->Ho; a.Wo = d->Wo; a.osh = d->osh; a.osw = d->osw; a.oh0 = d->oh0; a.ow0 = d->ow0;
    a.table = d->table; a.W = d->W; a.N = d->N; a.M = d->M; a.M_dev = d->M_dev;
    a.bias = d->bias; a.scale = d->scale; a.shift = d->shift; a.residual = d->residual; a.ldr = d->ldr;
    a.act = d->act; a.slope = d->slope; a.out = d->out; a.ldo = d->ldo; a.stats = d->stats;
    a.nbatch = d->nbatch > 1 ? (unsigned)d->nbatch : 1u;
    a.bsA = d->batch_stride_a; a.bsW = d->batch_stride_w; a.bsO = d->batch_stride_out; a.bsT = d->batch_stride_table;
    a.tam = d->mode == 2 ? d->table_alias_mask : 0;
    if (a.nbatch > 1) EFGH_CHECK_ARG(!d->stats && !d->residual && a.nbatch <= 65535 && a.bsA % 4 == 0 && a.bsW % 4 == 0);
    if (d->mode == 1) {
        EFGH_CHECK_ARG(d->B > 0 && d->Hin > 0 && d->Win > 0 && d->Hv > 0 && d->Wv > 0);
        EFGH_CHECK_ARG(d->M == (int64_t)d->B * d->Hv * d->Wv);
        EFGH_CHECK_ARG(d->osh >= 1 && d->osw >= 1 && d->Ho > 0 && d->Wo > 0);
    }
    if (d->mode == 2) EFGH_CHECK_ARG(d->table != nullptr && d->batch_stride_table >= 0 && (int64_t)d->batch_stride_table * (a.nbatch - 1) + d->T <= 16);
    if (d->mode == 0) EFGH_CHECK_ARG(d->T == 1);
    if (d->mode == 3) EFGH_CHECK_ARG(d->B > 0 && d->Hin == d->T && d->Win > 0 && d->Wv > 0 && d->M == (int64_t)d->B * d->Wv);
    return EFGH_OK;
}

extern "C" int efgh_gather_gemm(const efgh_gemm_desc *d, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    KArgs a;
    int rc = fill_args(d, a);
    if (rc != EFGH_OK) return rc;
    if (d->mode == 0) rc = dispatch<0>(a, st);
    else if (d->mode == 1) rc = dispatch<1>(a, st);
    else if (d->mode == 2) rc = dispatch<2>(a, st);
    else rc = dispatch<3>(a, st);
    if (rc != EFGH_OK) return rc;
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

/* tests / A-B runs: 0 = every launch on the register-staged kernel, 1 = eligible launches on the DMA-staged instances (default).
 * Returns the previous setting.  Process-wide; not meant to be flipped while another thread launches. */
extern "C" int efgh_gather_gemm_set_dma(int32_t on) {
    const int old = g_use_dma ? 1 : 0;
    g_use_dma = on != 0;
    return old;
}

extern "C" int efgh_pack_weight_batched(const efgh_pack_job *jobs_dev, const int64_t *prefix_dev, int32_t njobs, int64_t total,
                                        void *stream_) {
    EFGH_CHECK_ARG(jobs_dev && prefix_dev && njobs > 0 && total > 0);
    long long g = (total + 255) / 256;
    k_pack_weight_batched<<<(int)(g > 16384 ? 16384 : g), 256, 0, (hipStream_t)stream_>>>(jobs_dev, (const long long *)prefix_dev, njobs,
                                                                                         total);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_pack_weight_batched_tiled(const efgh_pack_job *jobs_dev, const int64_t *tile_prefix_dev, int32_t njobs, int64_t ntiles,
                                              void *stream_) {
    EFGH_CHECK_ARG(jobs_dev && tile_prefix_dev && njobs > 0 && ntiles > 0 && ntiles < 0x7fffffffLL);
    k_pack_weight_tiled<<<(unsigned)ntiles, 256, 0, (hipStream_t)stream_>>>(jobs_dev, (const long long *)tile_prefix_dev, njobs);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_fold_planes(const float *part, int32_t S, int64_t M, int32_t N, const float *bias, int32_t act, float slope,
                                float *out, int64_t ldo, void *stream_) {
    EFGH_CHECK_ARG(part && out && S >= 1 && M > 0 && N > 0 && N % 4 == 0 && ldo % 4 == 0 && ldo >= N);
    EFGH_CHECK_ARG((((uintptr_t)part) & 15) == 0 && (((uintptr_t)out) & 15) == 0 && (!bias || (((uintptr_t)bias) & 15) == 0));
    const long long total = M * (N / 4);
    long long g = (total + 255) / 256;
    k_fold_planes<<<(int)(g > 8192 ? 8192 : g), 256, 0, (hipStream_t)stream_>>>((const float4 *)part, S, M, N / 4, (const float4 *)bias,
                                                                              act, slope, out, ldo);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_pad_vec(const float *v, int32_t n, float *out, int32_t np, float fill, void *stream_) {
    EFGH_CHECK_ARG(v && out && n > 0 && np >= n);
    k_pad_vec<<<cdiv(np, 256), 256, 0, (hipStream_t)stream_>>>(v, n, out, np, fill);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_pack_weight(const float *W, float *Wp, int32_t N, int32_t T, int32_t C, int64_t sn,
                                int64_t sc, int64_t stt, const int32_t *tapidx, void *stream_) {
    return efgh_pack_weight_padded(W, Wp, N, T, C, N, C, sn, sc, stt, tapidx, stream_);
}

extern "C" int efgh_pack_weight_padded(const float *W, float *Wp, int32_t N, int32_t T, int32_t C, int32_t Np, int32_t Cp,
                                       int64_t sn, int64_t sc, int64_t stt, const int32_t *tapidx, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(W && Wp && N > 0 && T > 0 && T <= 16 && C > 0 && Np >= N && Cp >= C);
    int tp[16];
    for (int t = 0; t < 16; ++t) tp[t] = (tapidx && t < T) ? tapidx[t] : (t < T ? t : 0);
    long long total = (long long)Np * T * Cp;
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    k_pack_weight<<<grid, 256, 0, st>>>(W, Wp, Np, T, Cp, N, C, sn, sc, stt, make_int4(tp[0], tp[1], tp[2], tp[3]),
                                        make_int4(tp[4], tp[5], tp[6], tp[7]),
                                        make_int4(tp[8], tp[9], tp[10], tp[11]),
                                        make_int4(tp[12], tp[13], tp[14], tp[15]));
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}
