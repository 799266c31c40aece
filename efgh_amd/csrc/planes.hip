// Batched plain GEMMs of the 2-D Winograd layers with LDS-DMA staging (round 5) - the 36 alpha planes of F(4x4,3x3):
//
//   forward / data gradient   M_a[tile][n]   = sum_c     V_a[tile][c] * U_a[n][c]          (k_plane_gemm)
//   weight gradient           dU_a[n][c]     = sum_tile  G_a[tile][n] * V_a[tile][c]       (k_plane_wgrad)
//
// i.e. the contraction of every 3x3 / stride-1 convolution with >= 256 (weight gradient: >= 128) channels of nets/vgg.py:77 and
// nets/resnet.py:22-30 (H's and F's VGG blocks 3-5, G's deep ResNet stages), 65 ms of a 297-ms single-stream training step when
// they ran on k_gather_gemm<0> / k_gather_wgrad<0> (profiles/r04_train_kernel_stats_single_stream.csv).  Those kernels stage
// global -> VGPR -> ds_write with two workgroup barriers per 32-deep step; these rows need no transform between the load and the
// LDS image (contiguous fp32 rows), so here they go global -> LDS directly (`global_load_lds_dwordx4`, 1 KiB per wave
// instruction), the staging costs no VGPRs and no ds_write pass, and a step has ONE barrier:
//
//     wait for step s's DMA (counted vmcnt) -> barrier -> issue step s+NBUF-1's DMA into the buffer step s-1 just released
//     -> fragments of step s from LDS -> 64 MFMAs
//
// The destination of an LDS-DMA is wave-uniform base + lane * 16 (no per-lane scatter, no row padding), so the LDS images are
// linear and the bank-conflict-free layout is obtained by permuting which SOURCE 16 bytes a lane fetches:
//   k_plane_gemm:  image [row][32 floats] (128-B rows); the 16-B quad q of row r is stored at slot q ^ ((r >> 1) & 7) - the 16
//                  lanes of a ds_read_b128 phase (rows r .. r+15, one quad) then cover all 16 slots of the 256-B bank row;
//   k_plane_wgrad: image [m][128 floats], linear.  The contraction index is the ROW here, so a lane's MFMA operand is one float
//                  of a row; a lane reads TWO adjacent floats per ds_read_b64 (256 B contiguous per half-wave: conflict free)
//                  and feeds them to two MFMA tiles whose rows / columns interleave (tile q holds n = 2 i + q): half the LDS
//                  instructions of a ds_read_b32 per operand, and the epilogue stores 8 bytes per lane.
// Arithmetic: v_mfma_f32_32x32x2_f32, exact fp32 products and accumulation, the same k order as the kernels they replace
// (bit-identical results: tests/test_gpu_ops.py::test_plane_gemm_equals_gather_gemm).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int STAGE = (BM + BN) * BK;            // floats per ring slot: 32 KiB

struct PArgs {
    const float *A; long long lda;
    const float *W; long long ldw;
    float *out; long long ldo;
    long long M; int N, K;
    long long bsA, bsW, bsO;
    unsigned nbx;
};

template <int NBUF>
__global__ void __launch_bounds__(256, NBUF == 2 ? 2 : 1) k_plane_gemm(const PArgs p0) {
    extern __shared__ __attribute__((aligned(1024))) float ring[];          // NBUF x STAGE floats
    PArgs p = p0;
    p.A += (long long)blockIdx.y * p0.bsA; p.W += (long long)blockIdx.y * p0.bsW; p.out += (long long)blockIdx.y * p0.bsO;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
    const unsigned ring_lds = lds_addr(ring);
    // XCD-aware tile order (as k_gather_gemm): every XCD owns one contiguous band of output tiles
    const unsigned nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const unsigned lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
    const unsigned tile_m = lin / p.nbx, tile_n = lin - tile_m * p.nbx;
    const long long m0 = (long long)tile_m * BM;
    const int n0 = tile_n * BN;

    // ---- DMA roles: wave w moves A pieces 4w .. 4w+3 (rows 32w .. 32w+31) and W pieces 4w .. 4w+3 of every step.
    // piece = 8 rows x 128 B; lane l -> row 8 piece + (l >> 3), LDS slot (l & 7), source quad (l & 7) ^ ((row >> 1) & 7)
    const float *srcA[4], *srcW[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = (wave * 4 + q) * 8 + (lane >> 3);
        const int kq = (lane & 7) ^ ((r >> 1) & 7);
        long long m = m0 + r;
        if (m >= p.M) m = p.M - 1;                 // rows past the end re-read the last one (their outputs are never stored)
        srcA[q] = p.A + m * p.lda + kq * 4;
        srcW[q] = p.W + (long long)(n0 + r) * p.ldw + kq * 4;
    }
    auto issue = [&](int chunk, int buf) {
        const unsigned base = ring_lds + (unsigned)(buf * STAGE + wave * 4 * 256) * 4u;        // a piece is 256 floats = 1 KiB
        const int ko = chunk * BK;
#pragma unroll
        for (int q = 0; q < 4; ++q) dma16(srcA[q] + ko, base + q * 1024u);
#pragma unroll
        for (int q = 0; q < 4; ++q) dma16(srcW[q] + ko, base + (unsigned)(BM * BK) * 4u + q * 1024u);
    };

    // ---- fragment addresses: row (wm*2+i)*32 + l31 of A, (wn*2+j)*32 + l31 of W; quad g*2+lh stored at (g*2+lh) ^ ((row>>1)&7)
    int offA[2], offW[2], swA[2], swW[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = (wm * 2 + i) * 32 + l31, rw = (wn * 2 + i) * 32 + l31;
        offA[i] = ra * BK; swA[i] = (ra >> 1) & 7;
        offW[i] = BM * BK + rw * BK; swW[i] = (rw >> 1) & 7;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nch = p.K / BK;
#pragma unroll
    for (int s = 0; s < NBUF - 1; ++s)
        if (s < nch) issue(s, s);
    for (int ch = 0; ch < nch; ++ch) {
        // step ch's eight pieces of this wave have landed when at most the younger steps' pieces are outstanding
        if (NBUF == 2 || ch + 1 >= nch) wait_vm<0>();
        else wait_vm<8>();
        __builtin_amdgcn_s_barrier();           // everyone's pieces of step ch are in LDS; everyone is done reading step ch-1's slot
        asm volatile("" ::: "memory");
        if (ch + NBUF - 1 < nch) issue(ch + NBUF - 1, (ch + NBUF - 1) % NBUF);
        const float *buf = ring + (ch % NBUF) * STAGE;
        // fragments of group g+1 are fetched while the 16 MFMAs of group g run (two register sets; sched_barrier pins the order -
        // left alone the scheduler issues the next group's reads two MFMAs before they are needed and exposes the LDS latency
        // four times per step)
        float4 a[2][2], b[2][2];
        auto frag = [&](int g, float4 (&fa)[2], float4 (&fb)[2]) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const float4 *>(buf + offA[i] + (((g * 2 + lh) ^ swA[i]) << 2));
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const float4 *>(buf + offW[j] + (((g * 2 + lh) ^ swW[j]) << 2));
        };
        frag(0, a[0], b[0]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) frag(g + 1, a[(g + 1) & 1], b[(g + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const float4 (&ca)[2] = a[g & 1];
            const float4 (&cb)[2] = b[g & 1];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[i].x, cb[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[i].y, cb[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[i].z, cb[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[i].w, cb[j].w, acc[i][j], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the fragment reads of this step are retired before the next barrier)
    }

    // ---- epilogue: bare products, one 128-B row segment per half-wave and store.  A workgroup-uniform switch picks the body: a
    // whole tile walks its rows with one pointer add per row (register r -> row (r & 3) + 8 (r >> 2) + 4 lh: steps of 1, 1, 1, 5);
    // the last, ragged tile of a plane tests every row
    const int colb = n0 + wn * 64 + l31;
    if (m0 + BM <= p.M) {
        const long long ld5 = 5 * p.ldo;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float *op = p.out + (m0 + (wm * 2 + i) * 32 + 4 * lh) * p.ldo + colb;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st_out(op, acc[i][0][r]);
                st_out(op + 32, acc[i][1][r]);
                op += (r & 3) == 3 ? ld5 : p.ldo;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= p.M) continue;
                float *op = p.out + m * p.ldo + colb;
                st_out(op, acc[i][0][r]);
                st_out(op + 32, acc[i][1][r]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
struct PWArgs {
    const float *A; long long lda;       // [M][K]  (K = channels of the input side)
    const float *G; long long ldg;       // [M][N]
    float *dW;                           // [N][K] or the partial planes [zs][nbatch][N][K]
    long long M; int mchunk; int N, K;
    long long zstride;
    unsigned kt, nt;
    long long bsA, bsG, bsD;
};

constexpr int TM = 32, TK = 128, TN = 128;
constexpr int WSTAGE = TM * (TN + TK);            // 32 KiB per ring slot

__device__ __attribute__((aligned(512))) float g_plane_zero[128];          // 512 B of zeros: the source of rows past the end of a row chunk

template <int NBUF>
__global__ void __launch_bounds__(256, NBUF == 2 ? 2 : 1) k_plane_wgrad(const PWArgs p0) {
    extern __shared__ __attribute__((aligned(1024))) float ring[];
    PWArgs p = p0;
    p.A += (long long)blockIdx.y * p0.bsA; p.G += (long long)blockIdx.y * p0.bsG; p.dW += (long long)blockIdx.y * p0.bsD;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wk = wave & 1, l31 = lane & 31, lh = lane >> 5;
    const unsigned ring_lds = lds_addr(ring);
    const unsigned nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const unsigned lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
    const unsigned per_z = p.kt * p.nt, bz = lin / per_z, rem = lin - bz * per_z;
    const unsigned by = rem / p.kt, bx = rem - by * p.kt;
    const int k0 = bx * TK, n0 = by * TN;
    const long long mbeg = (long long)bz * p.mchunk;
    long long mend = mbeg + p.mchunk;
    if (mend > p.M) mend = p.M;
    if (mbeg >= mend) return;
    p.dW += (long long)bz * p.zstride;

    // ---- DMA roles: wave w moves G pieces 4w .. 4w+3 (rows 8w .. 8w+7) and A pieces 4w .. 4w+3 of every step.
    // piece = 2 rows x 512 B; lane l -> row 2 piece + (l >> 5), quad (l & 31) of that row
    const int prow = wave * 8 + (lane >> 5);                  // + 2 q
    const int cq = (lane & 31) * 4;
    const float *gcol = p.G + n0 + cq, *acol = p.A + k0 + cq;
    const float *zsrc = g_plane_zero + (lane & 31) * 4;
    auto issue = [&](long long ms, int buf) {
        const unsigned base = ring_lds + (unsigned)(buf * WSTAGE + wave * 4 * 256) * 4u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long long m = ms + prow + 2 * q;
            dma16(m < mend ? gcol + m * p.ldg : zsrc, base + q * 1024u);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long long m = ms + prow + 2 * q;
            dma16(m < mend ? acol + m * p.lda : zsrc, base + (unsigned)(TM * TN) * 4u + q * 1024u);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nst = (int)((mend - mbeg + TM - 1) / TM);
#pragma unroll
    for (int s = 0; s < NBUF - 1; ++s)
        if (s < nst) issue(mbeg + (long long)s * TM, s);
    // fragment columns: lane (l31, lh) reads floats 2 l31, 2 l31 + 1 of its wave's 64-column range in row 2 mm + lh
    const int gx = wn * 64 + 2 * l31, ax = TM * TN + wk * 64 + 2 * l31;
    for (int s = 0; s < nst; ++s) {
        if (NBUF == 2 || s + 1 >= nst) wait_vm<0>();
        else wait_vm<8>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + NBUF - 1 < nst) issue(mbeg + (long long)(s + NBUF - 1) * TM, (s + NBUF - 1) % NBUF);
        const float *Gs = ring + (s % NBUF) * WSTAGE + lh * 128;
        float2 g[2], a[2];
        g[0] = *reinterpret_cast<const float2 *>(Gs + gx);
        a[0] = *reinterpret_cast<const float2 *>(Gs + ax);
#pragma unroll
        for (int mm = 0; mm < TM / 2; ++mm) {
            if (mm + 1 < TM / 2) {              // the next row pair's operands while this one's four MFMAs run
                g[(mm + 1) & 1] = *reinterpret_cast<const float2 *>(Gs + (mm + 1) * 256 + gx);
                a[(mm + 1) & 1] = *reinterpret_cast<const float2 *>(Gs + (mm + 1) * 256 + ax);
            }
            __builtin_amdgcn_sched_barrier(0);
            const float2 cg = g[mm & 1], ca = a[mm & 1];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cg.x, ca.x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cg.x, ca.y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cg.y, ca.x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cg.y, ca.y, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // tile (q, q'): D row i <-> n = 2 i + q, column j <-> k = 2 j + q' of the wave's 64 x 64 block; lanes run along k: 8 bytes each
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        float *dp = p.dW + (long long)(n0 + wn * 64 + q + 8 * lh) * p.K + k0 + wk * 64 + 2 * l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2);                 // (+ 4 lh: in dp)
            *reinterpret_cast<float2 *>(dp + (long long)(2 * i) * p.K) = make_float2(acc[q][0][r], acc[q][1][r]);
        }
    }
}

std::atomic<unsigned long long> g_raised[4];

int plane_nbuf() { return 2; }          // two 32-KiB slots, two workgroups per CU: 7 % / 3 % faster than a three-slot ring with one (tools/bench_planes.py)

bool gemm_ok(const efgh_gemm_desc *d) {
    if (!d || d->mode != 0 || d->T != 1 || !d->A || !d->W || !d->out) return false;
    if (d->bias || d->scale || d->shift || d->residual || d->stats || d->act != 0 || d->M_dev) return false;
    if (d->C % BK || d->N % BN || d->M < 1 || d->lda % 4 || d->ldo < d->N) return false;
    if ((((uintptr_t)d->A) | ((uintptr_t)d->W)) & 15) return false;
    const int nb = d->nbatch > 1 ? d->nbatch : 1;
    if (nb > 65535) return false;
    if (nb > 1 && (d->batch_stride_a % 4 || d->batch_stride_w % 4)) return false;
    return true;
}

}  // namespace


extern "C" int efgh_plane_gemm_supported(const efgh_gemm_desc *d) { return gemm_ok(d) ? 1 : 0; }

/* out[b][m][n] = sum_k A[b][m][k] * W[b][n][k] for d->nbatch problems (mode 0, bare products): the LDS-DMA staged form of
 * efgh_gather_gemm for the launches efgh_plane_gemm_supported accepts; bit-identical results.  nbuf: ring slots (2 or 3; 0 = default) */
extern "C" int efgh_plane_gemm(const efgh_gemm_desc *d, int32_t nbuf, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(gemm_ok(d));
    if (nbuf == 0) nbuf = plane_nbuf();
    EFGH_CHECK_ARG(nbuf == 2 || nbuf == 3);
    PArgs a;
    a.A = d->A; a.lda = d->lda; a.W = d->W; a.ldw = d->C; a.out = d->out; a.ldo = d->ldo;
    a.M = d->M; a.N = d->N; a.K = d->C;
    const int nb = d->nbatch > 1 ? d->nbatch : 1;
    a.bsA = nb > 1 ? d->batch_stride_a : 0; a.bsW = nb > 1 ? d->batch_stride_w : 0; a.bsO = nb > 1 ? d->batch_stride_out : 0;
    a.nbx = (unsigned)(d->N / BN);
    const long long nby = (d->M + BM - 1) / BM;
    EFGH_CHECK_ARG(a.nbx * nby < 0x7fffffffLL);
    const dim3 grid((unsigned)(a.nbx * nby), (unsigned)nb);
    const size_t lds = (size_t)nbuf * STAGE * sizeof(float);
    if (nbuf == 2) {
        if (!efgh_raise_lds_once(g_raised[0], (const void *)k_plane_gemm<2>, (int)lds)) {
            efgh_set_error("%s:%d: cannot raise the dynamic LDS limit of k_plane_gemm", __FILE__, __LINE__);
            return EFGH_E_LAUNCH;
        }
        k_plane_gemm<2><<<grid, 256, lds, st>>>(a);
    } else {
        if (!efgh_raise_lds_once(g_raised[1], (const void *)k_plane_gemm<3>, (int)lds)) {
            efgh_set_error("%s:%d: cannot raise the dynamic LDS limit of k_plane_gemm", __FILE__, __LINE__);
            return EFGH_E_LAUNCH;
        }
        k_plane_gemm<3><<<grid, 256, lds, st>>>(a);
    }
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

// row chunks that fill rounds of the 512 resident workgroups (efgh_round_chunks, common.h): a 512 x 512-channel layer has 16 (n, c)
// blocks x 36 planes = 576 workgroups per chunk - one chunk alone runs a round and an eighth
static long long plane_wgrad_chunks(const efgh_gemm_desc *d, int nbatch, long long *chunk_out) {
    const long long blocks = (long long)(d->C / TK) * (d->N / TN) * nbatch;
    static const int occ = efgh_wg_per_cu((const void *)k_plane_wgrad<2>, 256, (size_t)2 * WSTAGE * sizeof(float));
    return efgh_round_chunks(d->M, blocks, occ, TM, 256, (double)nbatch * d->N * d->C, chunk_out);
}

static bool wgrad_ok(const efgh_gemm_desc *d, int64_t ldg) {
    if (!d || d->mode != 0 || d->T != 1 || !d->A) return false;
    if (d->C % TK || d->N % TN || d->M < 1 || d->lda % 4 || ldg % 4) return false;
    if (((uintptr_t)d->A) & 15) return false;
    const int nb = d->nbatch > 1 ? d->nbatch : 1;
    return nb <= 65535 && (nb == 1 || d->batch_stride_a % 4 == 0);
}

extern "C" int efgh_plane_wgrad_supported(const efgh_gemm_desc *d, int64_t ldg) { return wgrad_ok(d, ldg) ? 1 : 0; }

/* floats of scratch efgh_plane_wgrad_batched needs (0: a single row chunk writes dWp directly) */
extern "C" int64_t efgh_plane_wgrad_workspace(const efgh_gemm_desc *d) {
    if (!wgrad_ok(d, 4)) return 0;
    const int nb = d->nbatch > 1 ? d->nbatch : 1;
    const long long zs = plane_wgrad_chunks(d, nb, nullptr);
    return zs > 1 ? zs * nb * (int64_t)d->N * d->C : 0;
}

/* dWp[b][n][c] = sum_m G[b][m][n] * A[b][m][c]: the LDS-DMA staged form of efgh_gather_wgrad_batched (same arguments, same
 * row-chunk partials folded in chunk order; the chunks are sized to fill rounds of resident workgroups) for the launches
 * efgh_plane_wgrad_supported accepts */
extern "C" int efgh_plane_wgrad_batched(const efgh_gemm_desc *d, const float *G, int64_t ldg, int64_t batch_stride_g, float *dWp,
                                        int64_t batch_stride_dw, float *workspace, int32_t nbuf, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(wgrad_ok(d, ldg) && G && dWp && (((uintptr_t)G) & 15) == 0 && (((uintptr_t)dWp) & 15) == 0);
    const int nb = d->nbatch > 1 ? d->nbatch : 1;
    EFGH_CHECK_ARG(batch_stride_g % 4 == 0 && (nb == 1 || batch_stride_dw == (int64_t)d->N * d->C));
    if (nbuf == 0) nbuf = plane_nbuf();
    EFGH_CHECK_ARG(nbuf == 2 || nbuf == 3);
    PWArgs a;
    a.A = d->A; a.lda = d->lda; a.G = G; a.ldg = ldg; a.M = d->M; a.N = d->N; a.K = d->C;
    a.bsA = nb > 1 ? d->batch_stride_a : 0; a.bsG = nb > 1 ? batch_stride_g : 0; a.bsD = nb > 1 ? batch_stride_dw : 0;
    long long chunk = 0;
    const long long zs = plane_wgrad_chunks(d, nb, &chunk);
    a.mchunk = (int)chunk;
    a.kt = (unsigned)(d->C / TK); a.nt = (unsigned)(d->N / TN);
    EFGH_CHECK_ARG(zs * a.kt * a.nt < 0x7fffffffLL);
    const long long plane = (long long)nb * a.N * a.K;
    EFGH_CHECK_ARG(zs == 1 || (workspace && (((uintptr_t)workspace) & 15) == 0));
    a.dW = zs > 1 ? workspace : dWp;
    a.zstride = zs > 1 ? plane : 0;
    const dim3 grid((unsigned)(zs * a.kt * a.nt), (unsigned)nb);
    const size_t lds = (size_t)nbuf * WSTAGE * sizeof(float);
    if (nbuf == 2) {
        if (!efgh_raise_lds_once(g_raised[2], (const void *)k_plane_wgrad<2>, (int)lds)) {
            efgh_set_error("%s:%d: cannot raise the dynamic LDS limit of k_plane_wgrad", __FILE__, __LINE__);
            return EFGH_E_LAUNCH;
        }
        k_plane_wgrad<2><<<grid, 256, lds, st>>>(a);
    } else {
        if (!efgh_raise_lds_once(g_raised[3], (const void *)k_plane_wgrad<3>, (int)lds)) {
            efgh_set_error("%s:%d: cannot raise the dynamic LDS limit of k_plane_wgrad", __FILE__, __LINE__);
            return EFGH_E_LAUNCH;
        }
        k_plane_wgrad<3><<<grid, 256, lds, st>>>(a);
    }
    if (zs > 1) efgh_launch_fold_splits(workspace, (int)zs, plane, dWp, st);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}
