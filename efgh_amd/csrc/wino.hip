// Winograd F(4,3) along the image row for the 3x3 / stride 1 / pad 1 convolutions on fp32 MFMA (gfx950).
//
// Every VGG layer and most ResNet-18 layers of the H/F/G nets (nets/vgg.py:77, nets/resnet.py:22-30) and
// their data gradients are "same" 3x3 convolutions: ~85 % of the contraction FLOPs of the hot path.
// For 4 consecutive output pixels of a row and one kernel row kh
//     y[0..3] = A^T ( (G w[kh][0..2]) .* (B^T d[0..5]) )            (Lavin & Gray, F(4,3))
// needs 6 products instead of 12; with the kernel row folded into the contraction axis the layer becomes
// six GEMMs  M_a[tile][n] = sum_{kh,c} V_a[tile][kh][c] * U_a[n][kh][c]  (a = 0..5) of depth 3C: half the
// MFMA work of the direct form, all in fp32 (products and accumulation; no reduced-precision operand).
// Rounding: the transforms add ~2-3x the error of a k-ordered fp32 dot product (measured 3e-6 relative at
// C = 512), far inside the 1e-4 parity budget.
//
// One workgroup (256 threads = 2 x 2 waves) owns 64 tiles (= 256 output pixels, tiles are consecutive in
// (b, y, x/4) order) x 64 output channels; every wave holds ALL six M_a accumulators of its 32 x 32 block
// (96 VGPRs), so the inverse transform A^T runs on registers in the epilogue.  Per 16-channel chunk:
// global -> registers (6 input pixels x float4 per thread, 6 x float4 of U), B^T on the VALU, ds_write_b128
// into V[a][tile][16] / U[a][n][16] (64-B rows with XOR-swizzled quads: conflict-free b128 writes AND reads), then
// 48 v_mfma_f32_32x32x2_f32 per wave.  Epilogue = that of k_gather_gemm (bias, BN scale/shift, residual,
// activation, per-tile column statistics).
#include "common.h"
#include <type_traits>


namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// out-of-image taps read this zero page (address select) instead of being zeroed after the load
__device__ __attribute__((aligned(16))) float g_zero_page[16];

constexpr int TM = 64, TN = 64, KC = 16, LD = KC;     // 64-B LDS rows, 16-B quads XOR-swizzled by (row >> 2) & 3

struct WinoArgs {
    const float *A; long long lda; int C;
    int B, H, W, TW;                 // TW = ceil(W / 4) tiles per image row
    const float *U; int N;           // U [3C/16][6][N][16]
    long long Mt;                    // B * H * TW
    const float *bias, *scale, *shift, *residual; long long ldr;
    int act; float slope;
    float *out; long long ldo;
    float *stats;
    unsigned nbx;
    // stats_mode 1: BatchNorm-backward column sums of the written value (see efgh_gemm_desc)
    int smode;
    const float *bn_raw; long long bn_ldraw;
    const float *bn_y; long long bn_ldy;
    const float *bn_psc, *bn_psh, *bn_mean, *bn_invstd;
    int bn_act; float bn_slope;
};

// BNM: the epilogue also takes the BatchNorm-backward column sums of the written value (stats_mode 1).  A template parameter, not a
// run-time flag: the flag alone cost the plain kernel 9 % (measured) through the unrolled 16 x 4 epilogue.
// HPOOL (inference, a layer followed by MaxPool2d(2,2)): a tile's four pixels are two horizontal halves of pooling windows - the
// epilogue writes max(v0, v1), max(v2, v3) into a map of HALF the width ([B][H][W/2][N]); the vertical half of the window is a
// separate pass over that map (a row pair is two workgroups here).  Half the output bytes, half the pooling pass's input.
template <bool BNM, bool HPOOL = false>
__global__ void __launch_bounds__(256, 2) k_wino43(const WinoArgs p) {
    __shared__ __attribute__((aligned(16))) float Vs[6 * TM * LD];
    __shared__ __attribute__((aligned(16))) float Us[6 * TN * LD];
    __shared__ long long tpix[TM];       // first output pixel of the tile, -1 = no such tile
    __shared__ int tcnt[TM];             // valid pixels in the tile (ragged right edge)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
    // XCD-aware tile order (see k_gather_gemm): one contiguous band of tiles per XCD
    const unsigned nbx = p.nbx, nblk = gridDim.x;
    const unsigned q8 = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const unsigned lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
    const unsigned tile_m = lin / nbx, tile_n = lin - tile_m * nbx;
    const long long m0 = (long long)tile_m * TM;
    const int n0 = tile_n * TN;
    if (m0 >= p.Mt) return;

    // ---- staging state: thread = (tile st, channel quad cq) ---------------------------------
    const int st = tid >> 2, cq = (tid & 3) * 4;
    // LDS image: row = tile (or channel n), four 16-B quads per row stored at quad ^ ((row >> 2) & 3): the eight
    // lanes of a ds_write_b128 group (2 rows x 4 quads) and the sixteen rows of a ds_read_b128 group (one quad
    // each) then fall on distinct bank slots without padding
    const int cqs = ((tid & 3) ^ ((st >> 2) & 3)) * 4;
    long long pix0 = 0;
    unsigned cmask = 0, rmask = 0;
    {
        const long long t = m0 + st;
        if (t < p.Mt) {
            const int xt = (int)(t % p.TW); const long long r = t / p.TW;
            const int y = (int)(r % p.H); const long long b = r / p.H;
            const int x0 = 4 * xt - 1;
#pragma unroll
            for (int q = 0; q < 6; ++q) if ((unsigned)(x0 + q) < (unsigned)p.W) cmask |= 1u << q;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) if ((unsigned)(y + kh - 1) < (unsigned)p.H) rmask |= 1u << kh;
            pix0 = (b * p.H + y) * p.W + x0;
        }
    }
    if (tid < TM) {
        const long long t = m0 + tid;
        long long o = -1; int cnt = 0;
        if (t < p.Mt) {
            const int xt = (int)(t % p.TW); const long long r = t / p.TW;
            o = r * p.W + 4 * xt;                       // r = b*H + y
            cnt = p.W - 4 * xt; cnt = cnt > 4 ? 4 : cnt;
            if (HPOOL) {                                // pooled pixels of the tile in the half-width map (floor: an odd last column has none)
                const int wp = p.W >> 1;
                o = r * wp + 2 * xt;
                cnt = wp - 2 * xt; cnt = cnt > 2 ? 2 : cnt;
                if (cnt <= 0) { o = -1; cnt = 0; }
            }
        }
        tpix[tid] = o; tcnt[tid] = cnt;
    }
    const long long ustride = (long long)p.N * KC;      // one alpha slab of a chunk
    const float *ubase = p.U + (long long)(n0 + st) * KC + cq;

    // (named registers, not arrays: keeps the prefetched chunk out of scratch memory)
    float4 ra0, ra1, ra2, ra3, ra4, ra5, ru0, ru1, ru2, ru3, ru4, ru5;
    // per-thread part of the address once; per chunk only workgroup-uniform offsets (kernel row, channel block)
    const float *abase = p.A + (pix0 * p.lda + cq);
    const float *zpage = g_zero_page;
#define EFGH_LDA(q, dst)                                                                              \
    {                                                                                                \
        const float *src = (rok && ((cmask >> q) & 1u)) ? arow + (long long)q * p.lda : zpage;        \
        dst = *reinterpret_cast<const float4 *>(src);                                                \
    }
    // the 12 loads of the next chunk are issued two at a time between the MFMA groups of the current one (a burst of
    // 12 right after the barrier backs up the address path and delays the first MFMAs of every wave)
    const float *arow = abase, *urow = ubase;
    bool rok = false;
#define EFGH_CHUNK_ADDR(chv)                                                                          \
    {                                                                                                \
        const int ch_ = (chv);                                                                       \
        /* channel block fastest: consecutive chunks use the two 64-B halves of the same 128-B lines */ \
        const int kh = ch_ / ccn, cc = ch_ - kh * ccn;                                               \
        rok = (rmask >> kh) & 1u;                                                                    \
        arow = abase + ((long long)(kh - 1) * p.W * p.lda + cc * KC);                                \
        urow = ubase + (long long)(cc * 3 + kh) * 6 * ustride;                                       \
    }
#define EFGH_LOAD_PAIR(q, da, du)                                                                     \
    {                                                                                                \
        EFGH_LDA(q, da)                                                                              \
        du = *reinterpret_cast<const float4 *>(urow + q * ustride);                                  \
    }

    f32x16 acc[6];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    const int ccn = p.C / KC, nchunks = 3 * ccn;
    EFGH_CHUNK_ADDR(0)
    EFGH_LOAD_PAIR(0, ra0, ru0) EFGH_LOAD_PAIR(1, ra1, ru1) EFGH_LOAD_PAIR(2, ra2, ru2)
    EFGH_LOAD_PAIR(3, ra3, ru3) EFGH_LOAD_PAIR(4, ra4, ru4) EFGH_LOAD_PAIR(5, ra5, ru5)
    for (int ch = 0; ch < nchunks; ++ch) {
        {   // B^T d on the four channels of this thread, then one ds_write_b128 per alpha
            float4 v0, v1, v2, v3, v4, v5;
#define EFGH_BT(e)                                                                                   \
            {                                                                                        \
                const float d0 = ra0.e, d1 = ra1.e, d2 = ra2.e, d3 = ra3.e, d4 = ra4.e, d5 = ra5.e;    \
                const float p42 = d4 - 4.f * d2, p31 = d3 - 4.f * d1;                                \
                const float q42 = d4 - d2, q31 = 2.f * (d3 - d1);                                    \
                v0.e = 4.f * d0 - 5.f * d2 + d4;                                                     \
                v1.e = p42 + p31;                                                                    \
                v2.e = p42 - p31;                                                                    \
                v3.e = q42 + q31;                                                                    \
                v4.e = q42 - q31;                                                                    \
                v5.e = 4.f * d1 - 5.f * d3 + d5;                                                     \
            }
            EFGH_BT(x) EFGH_BT(y) EFGH_BT(z) EFGH_BT(w)
#undef EFGH_BT
#define EFGH_ST(a, vv, uu)                                                                            \
            *reinterpret_cast<float4 *>(&Vs[(a * TM + st) * LD + cqs]) = vv;                         \
            *reinterpret_cast<float4 *>(&Us[(a * TN + st) * LD + cqs]) = uu;
            EFGH_ST(0, v0, ru0) EFGH_ST(1, v1, ru1) EFGH_ST(2, v2, ru2) EFGH_ST(3, v3, ru3) EFGH_ST(4, v4, ru4) EFGH_ST(5, v5, ru5)
#undef EFGH_ST
        }
        __syncthreads();
        const bool more = ch + 1 < nchunks;
        if (more) EFGH_CHUNK_ADDR(ch + 1)
        {   // fragments of alpha a+1 are fetched while the eight MFMAs of alpha a run (two register sets)
            const int sw = (l31 >> 2) & 3, q0 = ((2 * lh) ^ sw) * 4, q1 = ((2 * lh + 1) ^ sw) * 4;
            const float *va = &Vs[(wm * 32 + l31) * LD];
            const float *ub = &Us[(wn * 32 + l31) * LD];
            float4 fa[2][2], fb[2][2];
            fa[0][0] = *reinterpret_cast<const float4 *>(va + q0); fa[0][1] = *reinterpret_cast<const float4 *>(va + q1);
            fb[0][0] = *reinterpret_cast<const float4 *>(ub + q0); fb[0][1] = *reinterpret_cast<const float4 *>(ub + q1);
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                const int cur = a & 1, nxt = cur ^ 1;
                if (a < 5) {
                    fa[nxt][0] = *reinterpret_cast<const float4 *>(va + (a + 1) * TM * LD + q0);
                    fa[nxt][1] = *reinterpret_cast<const float4 *>(va + (a + 1) * TM * LD + q1);
                    fb[nxt][0] = *reinterpret_cast<const float4 *>(ub + (a + 1) * TN * LD + q0);
                    fb[nxt][1] = *reinterpret_cast<const float4 *>(ub + (a + 1) * TN * LD + q1);
                }
                if (more) {
                    if (a == 0) EFGH_LOAD_PAIR(0, ra0, ru0)
                    if (a == 1) EFGH_LOAD_PAIR(1, ra1, ru1)
                    if (a == 2) EFGH_LOAD_PAIR(2, ra2, ru2)
                    if (a == 3) EFGH_LOAD_PAIR(3, ra3, ru3)
                    if (a == 4) EFGH_LOAD_PAIR(4, ra4, ru4)
                    if (a == 5) EFGH_LOAD_PAIR(5, ra5, ru5)
                }
                __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ahead of the MFMAs (the scheduler would sink it)
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][0].x, fb[cur][0].x, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][0].y, fb[cur][0].y, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][0].z, fb[cur][0].z, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][0].w, fb[cur][0].w, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][1].x, fb[cur][1].x, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][1].y, fb[cur][1].y, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][1].z, fb[cur][1].z, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][1].w, fb[cur][1].w, acc[a], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: A^T on registers, then the k_gather_gemm epilogue --------------------------
    float *ssum = Vs, *ssq = Vs + 2 * TN;
    const int coll = wn * 32 + l31, col = n0 + coll;
    const float bi = p.bias ? p.bias[col] : 0.f;
    const float sc = p.scale ? p.scale[col] : 1.f;
    const float sf = p.shift ? p.shift[col] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    constexpr bool bnm = BNM;
    float b_psc = 0.f, b_psh = 0.f, b_mu = 0.f, b_is = 0.f;
    if (bnm) {
        if (!p.bn_y) { b_psc = p.bn_psc[col]; b_psh = p.bn_psh[col]; }
        b_mu = p.bn_mean[col]; b_is = p.bn_invstd[col];
    }
    // per tile row: the pixel index, the row addresses and the count of valid pixels are formed once; a workgroup-uniform switch picks
    // the body with / without a residual operand (its four loads of a tile are issued together, ahead of the arithmetic)
    const float neg = p.act == 1 ? 0.f : (p.act == 2 ? p.slope : 1.f);      // act(v) = max(v, 0) + neg * min(v, 0)
    auto rows = [&](auto res_c, auto full_c) {
        constexpr bool RES = decltype(res_c)::value, FULL = decltype(full_c)::value;     // FULL: every tile of the workgroup exists
#pragma unroll                                                                            // and has its four pixels
        for (int r = 0; r < 16; ++r) {
            const int rl = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const long long opix = tpix[rl];
            const int cnt = FULL ? 4 : tcnt[rl];
            if (!FULL && opix < 0) continue;
            float *op = p.out + opix * p.ldo + col;
            float rv[4] = {0.f, 0.f, 0.f, 0.f};
            if (RES) {
                const float *rp = p.residual + opix * p.ldr + col;
#pragma unroll
                for (int i = 0; i < 4; ++i) if (FULL || i < cnt) rv[i] = rp[i * p.ldr];
            }
            const float m0_ = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r], m4 = acc[4][r], m5 = acc[5][r];
            const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
            const float yv[4] = {m0_ + s12 + s34, d12 + 2.f * d34, s12 + 4.f * s34, d12 + 8.f * d34 + m5};
            if (HPOOL) {                     // (no residual, no statistics: an inference epilogue)
                float w[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { float v = yv[i] + bi; v = v * sc + sf; w[i] = act_neg(v, neg); }
                st_out(op, fmaxf(w[0], w[1]));
                if (FULL || cnt > 1) st_out(op + p.ldo, fmaxf(w[2], w[3]));
                continue;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (!FULL && i >= cnt) continue;
                float v = yv[i] + bi;
                if (!bnm) { s1 += v; s2 = fmaf(v, v, s2); }
                v = v * sc + sf;
                if (RES) v += rv[i];
                v = act_neg(v, neg);
                st_out(op + i * p.ldo, v);
                if (bnm) {
                    const float rw = p.bn_raw[(opix + i) * p.bn_ldraw + col];
                    const float yy = p.bn_y ? p.bn_y[(opix + i) * p.bn_ldy + col] : rw * b_psc + b_psh;
                    const float g = p.bn_act == 1 ? (yy > 0.f ? v : 0.f) : p.bn_act == 2 ? (yy > 0.f ? v : v * p.bn_slope) : v;
                    s1 += g; s2 += g * ((rw - b_mu) * b_is);
                }
            }
        }
    };
    const bool full = m0 + TM <= p.Mt && (p.W & 3) == 0;
    if (p.residual) {
        if (full) rows(std::true_type{}, std::true_type{});
        else rows(std::true_type{}, std::false_type{});
    } else {
        if (full) rows(std::false_type{}, std::true_type{});
        else rows(std::false_type{}, std::false_type{});
    }
    if (p.stats) {
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        if (lh == 0) { ssum[wm * TN + coll] = s1; ssq[wm * TN + coll] = s2; }
        __syncthreads();
        if (tid < TN) {
            p.stats[((long long)tile_m * 2 + 0) * p.N + n0 + tid] = ssum[tid] + ssum[TN + tid];
            p.stats[((long long)tile_m * 2 + 1) * p.N + n0 + tid] = ssq[tid] + ssq[TN + tid];
        }
    }
}

// U[(cc*3 + kh)*6 + a][n][ci] = sum_kw G[a][kw] * Wp[n][kh*3 + kw][cc*16 + ci]      (Wp: packed [N][9][C])
__global__ void k_wino_pack(const float *__restrict__ Wp, float *__restrict__ U, int N, int C) {
    const long long total = 3LL * C * N;        // (ch, n, ci) triples, six outputs each (wino_pack_item, common.h)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
        wino_pack_item(Wp, U, N, C, i);
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient of the same layers in Winograd form (F(3,4): three taps out of a 4-pixel gradient tile):
//     dW[n][kh][kw][c] = sum_alpha A3^T[kw][alpha] * S_alpha[n][kh][c],
//     S_alpha[n][kh][c] = sum_tiles (G4 dy)_alpha[tile][n] * (B^T x)_alpha[tile][kh][c]
// six GEMMs contracted over the tiles (18 instead of 36 products per tile, tap row and channel pair).  One
// workgroup = 64 n x 64 (kh, c) columns x all six alpha; the tiles are split over gridDim.z and the partial
// S blocks left as one partial per unit (folded in a fixed order); k_wino_wgrad_finish3 writes the packed [N][9][C] layout.
constexpr int TT = 16;            // tiles per contraction step

struct WinoWArgs {
    const float *A; long long lda; int C;
    const float *G; long long ldg; int N;
    int B, H, W, TW;
    long long Mt; long long tchunk;
    float *S;                      // [units][3][N][3C]: one partial per unit (plain stores; folded in unit order, then k_wino_wgrad_finish3)
    unsigned kt, nt;               // column / row blocks of S
};

// ---- every operand row is read ONCE.  (Rounds 1-2 gave each (n, kh, c) block its own workgroup: the three kernel rows kh of a
// tile range were three workgroups that each read the gradient tiles and their own input row y + kh - 1 - 3.9 GB from HBM per
// launch for 2.0 GB of operands; that kernel, k_wino_wgrad, was removed in round 5.)  ONE workgroup of 12 waves owns a strip of 16 tiles (64 pixels) and walks DOWN the image rows:
// per row it transforms one gradient tile row (G4 dy) and one new input row (B^T x) into LDS; the transformed input rows stay in a
// ring of three, so the row staged for y + 1 is kernel row 2 of this step, row 1 of the next and row 0 of the one after.  Wave group
// kh (4 waves, 32 n x 32 c x 6 alpha each, as before) contracts the gradient image with ring slot y + kh - 1.  Per step the workgroup
// loads 40 KB for 3 x the MFMA work of one kernel row's contraction step.
constexpr int WR_WAVES = 12;

__global__ void __launch_bounds__(64 * WR_WAVES, 1) k_wino_wgrad_rows(const WinoWArgs p, int strips, int chunks, int rh, int ct) {
    __shared__ __attribute__((aligned(16))) float Gs[6 * TT * 64];
    __shared__ __attribute__((aligned(16))) float Vr[3][6 * TT * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave >> 2, wn = (wave >> 1) & 1, wc = wave & 1, l31 = lane & 31, lh = lane >> 5;
    const int unit = blockIdx.x, strip = unit % strips, ur = unit / strips, chk = ur % chunks;
    const long long b = ur / chunks;
    const int y0 = chk * rh, y1 = y0 + rh < p.H ? y0 + rh : p.H;
    const int c0 = (blockIdx.y % ct) * 64, n0 = (blockIdx.y / ct) * 64;
    if (y0 >= y1) return;
    // staging: waves 0-3 the input row, waves 4-7 the gradient row; thread = (tile ts of the strip, channel quad q4)
    const int role = wave >> 2, st = tid & 255, ts = st >> 4, q4 = (st & 15) * 4;
    const int xt = strip * TT + ts, x0 = 4 * xt - 1, xg = 4 * xt;
    const bool tv = xt < p.TW;
    const float *zpage = g_zero_page;
    float4 r0, r1, r2, r3, r4, r5;                   // raw pixels of the row being staged next (role 0: six of x; role 1: four of dy)
    auto load_x = [&](int r) {
        const bool rok = tv && (unsigned)r < (unsigned)p.H;
        const float *xrow = p.A + (((b * p.H + r) * p.W + x0) * p.lda + c0 + q4);
#define EFGH_LDX(q, dst) dst = *reinterpret_cast<const float4 *>((rok && (unsigned)(x0 + q) < (unsigned)p.W) ? xrow + (long long)q * p.lda : zpage);
        EFGH_LDX(0, r0) EFGH_LDX(1, r1) EFGH_LDX(2, r2) EFGH_LDX(3, r3) EFGH_LDX(4, r4) EFGH_LDX(5, r5)
#undef EFGH_LDX
    };
    auto load_g = [&](int r) {
        const float *grow = p.G + (((b * p.H + r) * p.W + xg) * p.ldg + n0 + q4);
#define EFGH_LDG(i, dst) dst = *reinterpret_cast<const float4 *>((tv && (xg + i) < p.W) ? grow + (long long)i * p.ldg : zpage);
        EFGH_LDG(0, r0) EFGH_LDG(1, r1) EFGH_LDG(2, r2) EFGH_LDG(3, r3)
#undef EFGH_LDG
    };
    auto stage_x = [&](float *V) {
        float4 v0, v1, v2, v3, v4, v5;
#define EFGH_TR(e)                                                                                    \
        {                                                                                            \
            const float d0 = r0.e, d1 = r1.e, d2 = r2.e, d3 = r3.e, d4 = r4.e, d5 = r5.e;            \
            const float p42 = d4 - 4.f * d2, p31 = d3 - 4.f * d1;                                    \
            const float q42 = d4 - d2, q31 = 2.f * (d3 - d1);                                        \
            v0.e = 4.f * d0 - 5.f * d2 + d4;                                                         \
            v1.e = p42 + p31; v2.e = p42 - p31; v3.e = q42 + q31; v4.e = q42 - q31;                  \
            v5.e = 4.f * d1 - 5.f * d3 + d5;                                                         \
        }
        EFGH_TR(x) EFGH_TR(y) EFGH_TR(z) EFGH_TR(w)
#undef EFGH_TR
#define EFGH_STW(a, vv) *reinterpret_cast<float4 *>(&V[(a * TT + ts) * 64 + q4]) = vv;
        EFGH_STW(0, v0) EFGH_STW(1, v1) EFGH_STW(2, v2) EFGH_STW(3, v3) EFGH_STW(4, v4) EFGH_STW(5, v5)
    };
    auto stage_g = [&]() {
        float4 u0, u1, u2, u3, u4, u5;
#define EFGH_TR(e)                                                                                    \
        {                                                                                            \
            const float g0 = r0.e, g1 = r1.e, g2 = r2.e, g3 = r3.e;                                  \
            const float ev = g0 + g2, od = g1 + g3, e2 = g0 + 4.f * g2, o2 = 2.f * g1 + 8.f * g3;     \
            u0.e = 0.25f * g0;                                                                       \
            u1.e = (ev + od) * (-1.f / 6.f); u2.e = (ev - od) * (-1.f / 6.f);                        \
            u3.e = (e2 + o2) * (1.f / 24.f); u4.e = (e2 - o2) * (1.f / 24.f);                        \
            u5.e = g3;                                                                               \
        }
        EFGH_TR(x) EFGH_TR(y) EFGH_TR(z) EFGH_TR(w)
#undef EFGH_TR
        float *V = Gs;
        EFGH_STW(0, u0) EFGH_STW(1, u1) EFGH_STW(2, u2) EFGH_STW(3, u3) EFGH_STW(4, u4) EFGH_STW(5, u5)
#undef EFGH_STW
    };

    f32x16 acc[6];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    // ring slot of input row r: (r - (y0 - 1)) % 3.  Prologue: rows y0 - 1 and y0
    if (role == 0) {
        load_x(y0 - 1); stage_x(Vr[0]);
        load_x(y0); stage_x(Vr[1]);
        load_x(y0 + 1);
    } else if (role == 1) {
        load_g(y0);
    }
    int s0 = 0;                                      // slot of row y - 1
    for (int y = y0; y < y1; ++y) {
        const int s2 = s0 == 0 ? 2 : s0 - 1;         // slot of row y + 1 = (s0 + 2) % 3
        if (role == 0) stage_x(Vr[s2]);
        else if (role == 1) stage_g();
        __syncthreads();
        const bool more = y + 1 < y1;
        if (more) {                                  // the next row's pixels: in flight during the MFMAs
            if (role == 0) load_x(y + 2);
            else if (role == 1) load_g(y + 1);
        }
        {   // contraction index = tile: lane half lh takes tile 2s + lh; operands are single dwords of the images
            const int sk = s0 + kg >= 3 ? s0 + kg - 3 : s0 + kg;       // slot of row y + kg - 1
            const float *gp = &Gs[lh * 64 + wn * 32 + l31];
            const float *vp = &Vr[sk][lh * 64 + wc * 32 + l31];
            float fg[2][8], fv[2][8];
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) { fg[0][s8] = gp[s8 * 128]; fv[0][s8] = vp[s8 * 128]; }
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                const int cur = a & 1, nxt = cur ^ 1;
                if (a < 5) {
#pragma unroll
                    for (int s8 = 0; s8 < 8; ++s8) {
                        fg[nxt][s8] = gp[(a + 1) * TT * 64 + s8 * 128];
                        fv[nxt][s8] = vp[(a + 1) * TT * 64 + s8 * 128];
                    }
                }
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8)
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fg[cur][s8], fv[cur][s8], acc[a], 0, 0, 0);
            }
        }
        __syncthreads();
        s0 = s0 == 2 ? 0 : s0 + 1;
    }
    // A3^T is applied HERE, per unit (it commutes with the sum over units): three kw planes per unit instead of six alpha planes -
    // half the bytes the fold has to read.  D[row = n][col = c]: lanes run along c (contiguous in S)
    const long long K3 = 3LL * p.C;
    float *sb = p.S + (((long long)unit * 3) * p.N + n0 + wn * 32) * K3 + kg * p.C + c0 + wc * 32 + l31;
    const long long plane = (long long)p.N * K3;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int nl = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float s0 = acc[0][r], s1 = acc[1][r], s2 = acc[2][r], s3 = acc[3][r], s4 = acc[4][r], s5 = acc[5][r];
        sb[nl * K3] = s0 + s1 + s2 + s3 + s4;
        sb[plane + nl * K3] = (s1 - s2) + 2.f * (s3 - s4);
        sb[2 * plane + nl * K3] = (s1 + s2) + 4.f * (s3 + s4) + s5;
    }
}

// the folded kw planes [3][N][3C] -> packed [N][9][C]
// (on, oc, ot): strides of the output - the packed layout (9 C, 1, C) or the reference layout of an armed unpack descriptor
__global__ void k_wino_wgrad_finish3(const float *__restrict__ S, float *__restrict__ dWp, int N, int C, long long on, long long oc,
                                     long long ot, int Nr, int Cr, int accumulate) {
    const long long K3 = 3LL * C, total = (long long)N * K3;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int kc = (int)(i % K3); const long long n = i / K3;
        const int kh = kc / C, c = kc - kh * C;
        if (n >= Nr || c >= Cr) continue;
        float *o = dWp + n * on + c * oc + (kh * 3) * ot;
        if (accumulate) { o[0] += S[i]; o[ot] += S[total + i]; o[2 * ot] += S[2 * total + i]; }
        else { o[0] = S[i]; o[ot] = S[total + i]; o[2 * ot] = S[2 * total + i]; }
    }
}

bool supported(const efgh_gemm_desc *d) {
    if (!d || d->mode != 1 || d->T != 9 || d->C % KC || d->N % TN || d->M_dev || d->nbatch > 1) return false;
    if (d->sh != 1 || d->sw != 1 || d->osh != 1 || d->osw != 1 || d->oh0 || d->ow0) return false;
    if (d->Hv != d->Hin || d->Wv != d->Win || d->Ho != d->Hin || d->Wo != d->Win) return false;
    for (int t = 0; t < 9; ++t) if (d->dh[t] != t / 3 - 1 || d->dw[t] != t % 3 - 1) return false;
    return d->lda % 4 == 0;
}

}  // namespace

extern "C" int efgh_wino_supported(const efgh_gemm_desc *d) { return supported(d) ? 1 : 0; }

extern "C" int32_t efgh_wino_grid_m(int32_t B, int32_t H, int32_t W) {
    const long long mt = (long long)B * H * ((W + 3) / 4);
    return (int32_t)((mt + TM - 1) / TM);
}

extern "C" int efgh_wino_pack(const float *Wp, float *U, int32_t N, int32_t C, void *stream_) {
    EFGH_CHECK_ARG(Wp && U && N > 0 && C > 0 && C % KC == 0);
    const long long total = 3LL * C * N;
    long long g = (total + 255) / 256;
    k_wino_pack<<<(int)(g > 8192 ? 8192 : g), 256, 0, (hipStream_t)stream_>>>(Wp, U, N, C);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_wino_conv3x3(const efgh_gemm_desc *d, const float *U, void *stream_) {
    EFGH_CHECK_ARG(supported(d) && U && d->A && d->out);
    EFGH_CHECK_ARG((((uintptr_t)d->A) & 15) == 0 && (((uintptr_t)U) & 15) == 0);
    EFGH_CHECK_ARG(d->B > 0 && d->M == (int64_t)d->B * d->Hin * d->Win);
    WinoArgs a;
    a.A = d->A; a.lda = d->lda; a.C = d->C;
    a.B = d->B; a.H = d->Hin; a.W = d->Win; a.TW = (d->Win + 3) / 4;
    a.U = U; a.N = d->N; a.Mt = (long long)d->B * d->Hin * a.TW;
    a.bias = d->bias; a.scale = d->scale; a.shift = d->shift; a.residual = d->residual; a.ldr = d->ldr;
    a.act = d->act; a.slope = d->slope; a.out = d->out; a.ldo = d->ldo; a.stats = d->stats;
    a.smode = d->stats ? d->stats_mode : 0;
    if (a.smode == 1) EFGH_CHECK_ARG(d->bn_raw && d->bn_mean && d->bn_invstd && (d->bn_y || (d->bn_pscale && d->bn_pshift)));
    a.bn_raw = d->bn_raw; a.bn_ldraw = d->bn_ldraw; a.bn_y = d->bn_y; a.bn_ldy = d->bn_ldy;
    a.bn_psc = d->bn_pscale; a.bn_psh = d->bn_pshift; a.bn_mean = d->bn_mean; a.bn_invstd = d->bn_invstd;
    a.bn_act = d->bn_act; a.bn_slope = d->bn_slope;
    a.nbx = (unsigned)(d->N / TN);
    const long long nby = (a.Mt + TM - 1) / TM;
    EFGH_CHECK_ARG(a.nbx * nby < 0x7fffffffLL);
    if (a.smode == 1) k_wino43<true><<<(unsigned)(a.nbx * nby), 256, 0, (hipStream_t)stream_>>>(a);
    else k_wino43<false><<<(unsigned)(a.nbx * nby), 256, 0, (hipStream_t)stream_>>>(a);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_wino_conv3x3_hpool(const efgh_gemm_desc *d, const float *U, void *stream_) {
    EFGH_CHECK_ARG(supported(d) && U && d->A && d->out && !d->residual && !d->stats && d->Win >= 2);
    EFGH_CHECK_ARG((((uintptr_t)d->A) & 15) == 0 && (((uintptr_t)U) & 15) == 0);
    EFGH_CHECK_ARG(d->B > 0 && d->M == (int64_t)d->B * d->Hin * d->Win);
    WinoArgs a;
    a.A = d->A; a.lda = d->lda; a.C = d->C;
    a.B = d->B; a.H = d->Hin; a.W = d->Win; a.TW = (d->Win + 3) / 4;
    a.U = U; a.N = d->N; a.Mt = (long long)d->B * d->Hin * a.TW;
    a.bias = d->bias; a.scale = d->scale; a.shift = d->shift; a.residual = nullptr; a.ldr = 0;
    a.act = d->act; a.slope = d->slope; a.out = d->out; a.ldo = d->ldo; a.stats = nullptr; a.smode = 0;
    a.bn_raw = nullptr; a.bn_ldraw = 0; a.bn_y = nullptr; a.bn_ldy = 0;
    a.bn_psc = a.bn_psh = a.bn_mean = a.bn_invstd = nullptr; a.bn_act = 0; a.bn_slope = 0.f;
    a.nbx = (unsigned)(d->N / TN);
    const long long nby = (a.Mt + TM - 1) / TM;
    EFGH_CHECK_ARG(a.nbx * nby < 0x7fffffffLL);
    k_wino43<false, true><<<(unsigned)(a.nbx * nby), 256, 0, (hipStream_t)stream_>>>(a);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_wino_wgrad_supported(const efgh_gemm_desc *d) {
    return (supported(d) && d->C % 64 == 0) ? 1 : 0;
}

// k_wino_wgrad_rows: units = samples x strips of 16 tiles x row chunks; one 12-wave workgroup per CU, so the number of row chunks is
// the one (<= 8, >= 8 rows each) that fills whole rounds of 256 workgroups best
static long long wino_wgrad_row_units(const efgh_gemm_desc *d, int *strips_out, int *chunks_out, int *rh_out) {
    const int strips = ((d->Win + 3) / 4 + TT - 1) / TT, blocks = (d->C / 64) * (d->N / 64);
    int best = 1; double beste = -1.0;
    for (int c = 1; c <= 8; ++c) {
        const int rh = (d->Hin + c - 1) / c;
        if (c > 1 && rh < 8) break;
        const long long wg = (long long)d->B * strips * ((d->Hin + rh - 1) / rh) * blocks;
        const double eff = (double)wg / (double)(((wg + 255) / 256) * 256) * (double)rh / (double)(rh + 2);
        if (eff > beste + 1e-9) { beste = eff; best = c; }
    }
    const int rh = (d->Hin + best - 1) / best, chunks = (d->Hin + rh - 1) / rh;
    if (strips_out) *strips_out = strips;
    if (chunks_out) *chunks_out = chunks;
    if (rh_out) *rh_out = rh;
    return (long long)d->B * strips * chunks;
}

/* floats of scratch `S` efgh_wino_wgrad needs: one [6][N][3C] partial per tile range */
extern "C" int64_t efgh_wino_wgrad_workspace(const efgh_gemm_desc *d) {
    if (!supported(d) || d->C % 64 != 0 || d->B <= 0) return 0;
    return wino_wgrad_row_units(d, nullptr, nullptr, nullptr) * 3 * (int64_t)d->N * 3 * d->C;
}

extern "C" int efgh_wino_wgrad(const efgh_gemm_desc *d, const float *G, int64_t ldg, float *S, float *dWp,
                               const efgh_wgrad_out_desc *out, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(supported(d) && d->C % 64 == 0 && d->A && G && S && dWp && ldg % 4 == 0);
    EFGH_CHECK_ARG((((uintptr_t)d->A) & 15) == 0 && (((uintptr_t)G) & 15) == 0);
    EFGH_CHECK_ARG(d->B > 0 && d->M == (int64_t)d->B * d->Hin * d->Win);
    WinoWArgs a;
    a.A = d->A; a.lda = d->lda; a.C = d->C; a.G = G; a.ldg = ldg; a.N = d->N;
    a.B = d->B; a.H = d->Hin; a.W = d->Win; a.TW = (d->Win + 3) / 4;
    a.Mt = (long long)d->B * d->Hin * a.TW;
    a.S = S;
    const int kt = 3 * d->C / 64, nt = d->N / 64;
    int strips = 0, chunks = 0, rh = 0;
    const long long zs = wino_wgrad_row_units(d, &strips, &chunks, &rh);
    EFGH_CHECK_ARG(zs < 0x7fffffffLL && (long long)(d->C / 64) * nt < 65536);
    a.tchunk = 0; a.kt = (unsigned)kt; a.nt = (unsigned)nt;
    k_wino_wgrad_rows<<<dim3((unsigned)zs, (unsigned)((d->C / 64) * nt)), 64 * WR_WAVES, 0, st>>>(a, strips, chunks, rh, d->C / 64);
    EFGH_CHECK_LAUNCH();
    if (zs > 1) efgh_launch_fold_splits(S, (int)zs, 3LL * d->N * 3 * d->C, S, st);      // into the first partial, fixed order (not the packed gradient: no `out`)
    const long long total = (long long)d->N * 3 * d->C;
    long long g = (total + 255) / 256;
    const bool direct = efgh_wgrad_out_fits(out, d->N, 9, d->C);      // the finish kernel writes the caller's layout itself
    if (direct)
        k_wino_wgrad_finish3<<<(int)(g > 4096 ? 4096 : g), 256, 0, st>>>(S, out->W, d->N, d->C, out->sn, out->sc, out->st, out->N, out->C, out->accumulate);
    else
        k_wino_wgrad_finish3<<<(int)(g > 4096 ? 4096 : g), 256, 0, st>>>(S, dWp, d->N, d->C, 9LL * d->C, 1, d->C, d->N, d->C, 0);
    EFGH_CHECK_LAUNCH();
    return direct ? EFGH_WROTE_OUT : EFGH_OK;
}
