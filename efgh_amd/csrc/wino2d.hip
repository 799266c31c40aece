// Winograd F(4x4, 3x3) for the 3x3 / stride 1 / pad 1 convolutions with >= 128 channels (gfx950), fp32 throughout.
//
// The 1-D kernel of wino.hip (F(4,3) along the image row, 2x fewer MFMAs than the direct form) is held at 63 % of the fp32 matrix
// pipe by its staging traffic per MFMA (profiles/r02_wino_ablation.txt).  The 2-D form needs 36 products per 4x4 output tile
// and channel pair instead of 144 - a quarter of the direct MFMA work - and for C >= 128 the transforms are cheap enough to
// run as separate streaming passes around a plain batched GEMM:
//     V[tile][a][c] = (B^T d B)_a         k_w2_input     reads x once (tiles overlap: 2.25x from L2), writes 2.25 x |x|
//                                                         (tile-major: a tile's 36 planes are ONE contiguous block - with
//                                                         plane-major [a][tile][c] the 36 write streams ran at 1.9 TB/s)
//     M_a[tile][n] = sum_c V_a[tile][c] U_a[n][c]      36 GEMMs of depth C: k_gather_gemm, mode 0, batched (gemm.hip)
//     y = A^T M A (+ bias, BatchNorm statistics / affine, residual, activation)     k_w2_output
// with U_a = (G w G^T)_a packed once per weight version (k_w2_pack), a = 6 i + j.  Layers: every VGG layer of H / F from
// 128 channels up and the ResNet-18 layers 2-4 of G (nets/vgg.py:77, nets/resnet.py:22-30), their data gradients (same path on
// the transposed, tap-reversed weights) and their weight gradients:
//     dW = A3^T [ sum_tiles (G4 dy G4^T)_a (x) (B^T x B)_a ] A3     k_w2_dy, k_w2_input, batched k_gather_wgrad, k_w2_wfinish
// Rounding: with the interpolation points chosen below 7.5e-7 rms / 1.0e-6 max relative on a 256-channel layer (numpy fp32
// model of exactly these transforms) - about the level of the 1-D kernel, which still uses Lavin & Gray's points.
#include "common.h"

namespace {
constexpr int TPB = 256;

struct TileGeo { int B, H, W, TH, TW; long long T; };

// launches are 2-D: blockIdx.y walks the rows of tiles (b, ty) - one scalar division per workgroup - and blockIdx.x the
// (tile column, channel pair) pairs of a row, split with a shift (the pair count is a power of two): no per-thread division
// (the first version spent more instructions on 64-bit index divisions than on the transform and ran at 1 TB/s)
__device__ __forceinline__ int ilog2(int v) { return 31 - __builtin_clz(v); }

// the transform kernels work on channel PAIRS (8 bytes per lane): 36 accumulators of a 6x6 tile are 72 registers, which keeps
// four waves per SIMD resident; with 16-byte quads the kernels needed 256 registers (+ spills) and ran at 1.1 TB/s
typedef float2 vec_t;
constexpr int VW = 2;
struct vec1_t { float x; };          // one channel per lane (k_w2_bwd: twice the tensors of the plain transforms live per tile)
#define F4OP(dst, expr)                                                                              \
    { dst.x = expr(x); if constexpr (sizeof(dst) == 8) dst.y = expr(y); }

// Interpolation points 0, +-3/4, +-3/2, inf (Toom-Cook F(4,3)).  Lavin & Gray's 0, +-1, +-2, inf put factors of 4, 5 and 8 into
// the transforms; in fp32 the 2-D form then carries 1.5e-6 rms / 3.6e-6 max relative error on a 256-channel layer, this set
// 7.5e-7 / 1.0e-6 (best of the 1001 symmetric-or-not 5-subsets of {+-1/4 .. +-3} tried; numpy model of exactly these transforms).
// With a = 3/4, b = 3/2 (pairs +-a, +-b keep the sum / difference structure):
//   B^T d : v0 = a^2 b^2 d0 - (a^2+b^2) d2 + d4,   v1,2 = (d4 - b^2 d2) +- a (d3 - b^2 d1),   v3,4 = (d4 - a^2 d2) +- b (d3 - a^2 d1),
//           v5 = a^2 b^2 d1 - (a^2+b^2) d3 + d5
//   G     : rows [1, p, p^2] / f(p), f(0) = a^2 b^2, f(+-a) = 2 a^2 (a^2 - b^2), f(+-b) = 2 b^2 (b^2 - a^2); inf: [0, 0, 1]
//   A^T m : y_i = sum_j p_j^i m_j (+ m5 for i = 3)
#define W2_A 0.75f
#define W2_B 1.5f
#define W2_A2 0.5625f          /* a^2 */
#define W2_B2 2.25f            /* b^2 */
#define W2_A2B2 1.265625f      /* a^2 b^2 */
#define W2_SUM 2.8125f         /* a^2 + b^2 */
#define W2_A3 0.421875f        /* a^3 */
#define W2_B3 3.375f           /* b^3 */

// B^T (6 -> 6)
template <class vec_t>
__device__ __forceinline__ void bt6(const vec_t d[6], vec_t v[6]) {
#define E0(e) (W2_A2B2 * d[0].e - W2_SUM * d[2].e + d[4].e)
#define E1(e) ((d[4].e - W2_B2 * d[2].e) + W2_A * (d[3].e - W2_B2 * d[1].e))
#define E2(e) ((d[4].e - W2_B2 * d[2].e) - W2_A * (d[3].e - W2_B2 * d[1].e))
#define E3(e) ((d[4].e - W2_A2 * d[2].e) + W2_B * (d[3].e - W2_A2 * d[1].e))
#define E4(e) ((d[4].e - W2_A2 * d[2].e) - W2_B * (d[3].e - W2_A2 * d[1].e))
#define E5(e) (W2_A2B2 * d[1].e - W2_SUM * d[3].e + d[5].e)
    F4OP(v[0], E0) F4OP(v[1], E1) F4OP(v[2], E2) F4OP(v[3], E3) F4OP(v[4], E4) F4OP(v[5], E5)
#undef E0
#undef E1
#undef E2
#undef E3
#undef E4
#undef E5
}

// A^T (6 -> 4)
__device__ __forceinline__ void at4(const vec_t m[6], vec_t y[4]) {
#define E0(e) (m[0].e + (m[1].e + m[2].e) + (m[3].e + m[4].e))
// (the contraction order of the a*b + c*d forms is WRITTEN OUT: the pooled and the plain instantiation of k_w2_output, left to
// -ffp-contract, differed in the last bit of a third of their outputs)
#define E1(e) fmaf(W2_A, m[1].e - m[2].e, W2_B * (m[3].e - m[4].e))
#define E2(e) fmaf(W2_A2, m[1].e + m[2].e, W2_B2 * (m[3].e + m[4].e))
#define E3(e) (fmaf(W2_A3, m[1].e - m[2].e, W2_B3 * (m[3].e - m[4].e)) + m[5].e)
    F4OP(y[0], E0) F4OP(y[1], E1) F4OP(y[2], E2) F4OP(y[3], E3)
#undef E0
#undef E1
#undef E2
#undef E3
}

// G4 (4 -> 6): the gradient-side transform of F(3,4): rows [1, p, p^2, p^3] / f(p); inf: [0, 0, 0, 1]
#define W2_F0 (1.0f / 1.265625f)                  /* 1 / f(0)    = 64/81   */
#define W2_FA (-128.0f / 243.0f)                  /* 1 / f(+-a)             */
#define W2_FB (32.0f / 243.0f)                    /* 1 / f(+-b)             */
template <class vec_t>
__device__ __forceinline__ void g46(const vec_t g[4], vec_t u[6]) {
#define E0(e) (W2_F0 * g[0].e)
#define E1(e) (((g[0].e + W2_A2 * g[2].e) + (W2_A * g[1].e + W2_A3 * g[3].e)) * W2_FA)
#define E2(e) (((g[0].e + W2_A2 * g[2].e) - (W2_A * g[1].e + W2_A3 * g[3].e)) * W2_FA)
#define E3(e) (((g[0].e + W2_B2 * g[2].e) + (W2_B * g[1].e + W2_B3 * g[3].e)) * W2_FB)
#define E4(e) (((g[0].e + W2_B2 * g[2].e) - (W2_B * g[1].e + W2_B3 * g[3].e)) * W2_FB)
#define E5(e) (g[3].e)
    F4OP(u[0], E0) F4OP(u[1], E1) F4OP(u[2], E2) F4OP(u[3], E3) F4OP(u[4], E4) F4OP(u[5], E5)
#undef E0
#undef E1
#undef E2
#undef E3
#undef E4
#undef E5
}

// 8-byte streaming accesses of the 2.25x-sized Winograd-domain tensors (see ld_stream / st_stream in common.h)
template <bool NT>
__device__ __forceinline__ vec_t ld2(const float *p) {
    if (NT) {
        typedef float v2f __attribute__((ext_vector_type(2)));
        const v2f v = __builtin_nontemporal_load(reinterpret_cast<const v2f *>(__builtin_assume_aligned(p, 8)));
        return make_float2(v.x, v.y);
    }
    return *reinterpret_cast<const vec_t *>(__builtin_assume_aligned(p, 8));
}
template <bool NT>
__device__ __forceinline__ void st2(float *p, const vec_t &v) {
    if (NT) {
        typedef float v2f __attribute__((ext_vector_type(2)));
        v2f w; w.x = v.x; w.y = v.y;
        __builtin_nontemporal_store(w, reinterpret_cast<v2f *>(__builtin_assume_aligned(p, 8)));
    } else {
        *reinterpret_cast<vec_t *>(p) = v;
    }
}

__device__ __forceinline__ void axpy4(vec_t &a, float s, const vec_t &v) {
    a.x += s * v.x; a.y += s * v.y;
}

// ---- input transform: thread = (tile, channel pair); 36 x 8-byte loads (pixels outside the image are zero), 36 stores
// AFF (round 6): A is the RAW output of a train-mode BatchNorm layer whose normalise + activate pass was not run - the transform
// applies act(raw * scale[c] + shift[c]) on the way in (the producer's activation has this layer as its only consumer, so it is
// never written or re-read: one write + one read of the activation less per such pair of layers, nets/resnet.py:55-71 conv1 -> conv2,
// nets/vgg.py:69-83 un-pooled pairs, nets/net_utils.py:66-98 convT -> conv).  Same expression as k_scale_shift_act (one fma, act_f).
template <bool NT, bool AFF>
__global__ void __launch_bounds__(TPB)
k_w2_input(const float *__restrict__ A, long long lda, int C, const TileGeo g, float *__restrict__ V,
           const float *__restrict__ scale, const float *__restrict__ shift, int act, float slope) {
    const int q4n = C / VW, sh = ilog2(q4n);
    const int idx = blockIdx.x * TPB + threadIdx.x;
    const int tx = idx >> sh, q = idx & (q4n - 1);
    if (tx >= g.TW) return;
    const int rowt = blockIdx.y, ty = rowt % g.TH;
    const long long b = rowt / g.TH;
    const long long t = (long long)rowt * g.TW + tx;
    const int y0 = 4 * ty, x0 = 4 * tx;
    // all 36 loads first (one round trip: the kernel is bound by waves in flight x dependent memory round trips), then the two
    // 1-D transforms in place.  No branch around a load: out-of-image taps read a clamped (valid) address and are zeroed by a
    // select (36 exec-masked branches, one per load, kept the loads from being issued together)
    const vec_t zero = make_float2(0.f, 0.f);
    vec_t sc = make_float2(1.f, 1.f), sf = zero;
    if (AFF) {
        sc = *reinterpret_cast<const vec_t *>(scale + q * VW);
        sf = *reinterpret_cast<const vec_t *>(shift + q * VW);
    }
    vec_t d[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int y = y0 - 1 + r;
        const bool rok = (unsigned)y < (unsigned)g.H;
        const int yc = min(max(y, 0), g.H - 1);
        const float *row = A + ((b * g.H + yc) * g.W) * lda + q * VW;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int x = x0 - 1 + c;
            const bool ok = rok && (unsigned)x < (unsigned)g.W;
            const int xc = min(max(x, 0), g.W - 1);
            const vec_t v = *reinterpret_cast<const vec_t *>(__builtin_assume_aligned(row + (long long)xc * lda, 8));
            d[r][c] = (AFF || ok) ? v : zero;
        }
    }
    if (AFF) {
        // after ALL loads, and branch-free: a branch on `act` per element made the compiler wait for every load before issuing the
        // next one (36 round trips: 3.5 TB/s instead of 6.3).  act(v) = max(v, 0) + neg * min(v, 0), neg = 0 / slope / 1 - the same bits
        const float neg = act == 1 ? 0.f : (act == 2 ? slope : 1.f);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const bool rok = (unsigned)(y0 - 1 + r) < (unsigned)g.H;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const bool ok = rok && (unsigned)(x0 - 1 + c) < (unsigned)g.W;
                vec_t v = d[r][c];
                v.x = fmaf(v.x, sc.x, sf.x); v.y = fmaf(v.y, sc.y, sf.y);
                v.x = fmaxf(v.x, 0.f) + neg * fminf(v.x, 0.f); v.y = fmaxf(v.y, 0.f) + neg * fminf(v.y, 0.f);
                d[r][c] = ok ? v : zero;
            }
        }
    }
    vec_t acc[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {        // along x
        vec_t w[6];
        bt6(d[r], w);
#pragma unroll
        for (int j = 0; j < 6; ++j) d[r][j] = w[j];
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {        // along y
        vec_t col[6], w[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) col[r] = d[r][j];
        bt6(col, w);
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[i][j] = w[i];
    }
    float *vp = V + t * 36 * C + q * VW;           // V [T][36][C]: the 36 planes of a tile are one contiguous block
    const long long as = C;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) st2<NT>(vp + (long long)(6 * i + j) * as, acc[i][j]);
}

// ---- gradient-side transform of the weight gradient: Gy_a[tile][n] = (G4 dy G4^T)_a, dy outside the image is zero
template <bool NT>
__global__ void __launch_bounds__(TPB)
k_w2_dy(const float *__restrict__ G, long long ldg, int N, const TileGeo g, float *__restrict__ Gy) {
    const int q4n = N / VW, sh = ilog2(q4n);
    const int idx = blockIdx.x * TPB + threadIdx.x;
    const int tx = idx >> sh, q = idx & (q4n - 1);
    if (tx >= g.TW) return;
    const int rowt = blockIdx.y, ty = rowt % g.TH;
    const long long b = rowt / g.TH;
    const long long t = (long long)rowt * g.TW + tx;
    const int y0 = 4 * ty, x0 = 4 * tx;
    const vec_t zero = make_float2(0.f, 0.f);
    vec_t d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int y = y0 + r;
        const bool rok = y < g.H;
        const float *row = G + ((b * g.H + min(y, g.H - 1)) * g.W) * ldg + q * VW;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int x = x0 + c;
            const vec_t v = ld2<NT>(row + (long long)min(x, g.W - 1) * ldg);
            d[r][c] = (rok && x < g.W) ? v : zero;
        }
    }
    vec_t wx[4][6], acc[6][6];
#pragma unroll
    for (int r = 0; r < 4; ++r) g46(d[r], wx[r]);          // along x
#pragma unroll
    for (int j = 0; j < 6; ++j) {                          // along y
        vec_t col[4], w[6];
#pragma unroll
        for (int r = 0; r < 4; ++r) col[r] = wx[r][j];
        g46(col, w);
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[i][j] = w[i];
    }
    float *vp = Gy + t * 36 * N + q * VW;
    const long long as = N;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) st2<NT>(vp + (long long)(6 * i + j) * as, acc[i][j]);
}

// ---- BatchNorm backward "apply" inside the two gradient-side transforms (round 6).  A 2-D Winograd layer with a train-mode BatchNorm
// ran its backward as  reduce (dy, raw -> column sums) -> apply (dy, raw -> draw) -> k_w2_input(draw) for the data gradient +
// k_w2_dy(draw) for the weight gradient: draw was written once and read twice.  This kernel computes
//     dpre = dy * act'(y),   draw = coef * (dpre - m1 - xhat * m2),   xhat = (raw - mean) * invstd        (float64, as k_act_bn_bwd_apply)
// on the fly for the 6x6 window of its tile and leaves BOTH transforms - Vd = B^T draw B over the window, Gy = G4 draw G4^T over the
// window's inner 4x4 (the tile itself) - and, for a residual layer, dres = dpre of the tile: 2 reads + 4.5 (5.5) writes instead
// of 5 + 4.5 (6.5).  The activation mask comes from the sign bits the forward pass left (BITS: residual layers) or is re-derived
// from raw * pscale + pshift, exactly as in the two-pass form.
// both gradient-side transforms of one (tile, channel) from the 6x6 window of draw: Gy = G4 draw G4^T over the tile itself (rows /
// columns 1..4 of the window) and Vd = B^T draw B over the window, each stored column by column
template <bool NT>
__device__ __forceinline__ void st1_(float *p, float v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <bool NT>
__device__ __forceinline__ void w2_bwd_store(vec1_t d[6][6], float *gp, float *vp, int N) {
    {
        vec1_t wx[4][6];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const vec1_t in4[4] = {d[r + 1][1], d[r + 1][2], d[r + 1][3], d[r + 1][4]};
            g46(in4, wx[r]);
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            vec1_t col[4], w[6];
#pragma unroll
            for (int r = 0; r < 4; ++r) col[r] = wx[r][j];
            g46(col, w);
#pragma unroll
            for (int i = 0; i < 6; ++i) st1_<NT>(gp + (long long)(6 * i + j) * N, w[i].x);
        }
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        vec1_t w[6];
        bt6(d[r], w);
#pragma unroll
        for (int j = 0; j < 6; ++j) d[r][j] = w[j];
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        vec1_t col[6], w[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) col[r] = d[r][j];
        bt6(col, w);
#pragma unroll
        for (int i = 0; i < 6; ++i) st1_<NT>(vp + (long long)(6 * i + j) * N, w[i].x);
    }
}

struct W2BwdArgs {
    const float *dy; long long lddy; const float *raw; long long ldraw; const unsigned *ybits;
    const float *mean, *invstd, *coef, *pscale, *pshift; const double *m1, *m2;
    int N, act; float slope; TileGeo g;
    float *Vd, *Gy, *dres; long long lddres;
};

template <bool NT>
__device__ __forceinline__ float ld1(const float *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT>
__device__ __forceinline__ void st1(float *p, float v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// thread = (tile, ONE channel): a wave covers 64 consecutive channels of a pixel (256-byte rows).  With channel pairs the window
// of dy AND raw plus both transforms held 213-256 VGPRs (1-2 waves per SIMD; k_w2_output ran at a third of its rate like that)
template <bool NT, bool BITS>
__global__ void __launch_bounds__(TPB, 4)          // (<= 128 VGPRs: an HBM-bound pass lives on bytes in flight)
k_w2_bwd(const W2BwdArgs p) {
    const int sh = ilog2(p.N);
    const int idx = blockIdx.x * TPB + threadIdx.x;
    // N >= 64 (a power of two): the 64 lanes of a wave are 64 consecutive channels of ONE tile - the tile index is wave-uniform and
    // is told to the compiler as such: every pixel address is then a scalar base + the lane's channel (no 64-bit address VGPRs),
    // and the mask words come in through the scalar cache
    const int tx = __builtin_amdgcn_readfirstlane(idx >> sh), c0 = idx & (p.N - 1);
    if (tx >= p.g.TW) return;
    const int rowt = blockIdx.y, ty = rowt % p.g.TH;
    const long long b = rowt / p.g.TH;
    const long long t = (long long)rowt * p.g.TW + tx;
    const int y0 = 4 * ty, x0 = 4 * tx;
    const float mu = p.mean[c0], is = p.invstd[c0], cf = p.coef[c0];
    const double m1 = p.m1[c0], m2 = p.m2[c0];
    const float negd = p.act == 1 ? 0.f : (p.act == 2 ? p.slope : 1.f);        // act'(y) for y <= 0 (branch-free: 1 for y > 0)
    const double Ad = (double)cf * (double)is * m2, Bd = (double)cf * m1 - Ad * (double)mu;
    float psc = 0.f, psh = 0.f;
    if (!BITS) { psc = p.pscale[c0]; psh = p.pshift[c0]; }
    // addresses: one scalar base per window row and tensor + a 32-bit lane offset per window column (clamped; out-of-image taps
    // are zeroed after the arithmetic).  72 independent 64-bit addresses cost 144 SGPRs / VGPRs and spilled.
    int xcs[6];
    unsigned offd[6], offr[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        xcs[c] = min(max(x0 - 1 + c, 0), p.g.W - 1);
        offd[c] = 4u * (unsigned)(xcs[c] * (int)p.lddy + c0);          // BYTE offsets: base (SGPR pair) + zext(32-bit VGPR) is the
        offr[c] = 4u * (unsigned)(xcs[c] * (int)p.ldraw + c0);         // addressing mode; an element offset would be shifted in 64 bits
    }
    // BITS: the 36 mask bits first, packed into one 64-bit register.  The 64 mask bits of a wave's 64 channels of a pixel are one
    // aligned 8-byte word at a wave-uniform address: scalar loads, in two halves of 18 (36 SGPRs each)
    unsigned long long mk = 0ULL;
    if (BITS) {
        const int cb = __builtin_amdgcn_readfirstlane(c0) & ~63;
        const unsigned cl = (unsigned)(c0 & 63);
        const int wpp = p.N >> 5;                                            // mask words per pixel
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {
                const int r = 3 * half + rr;
                const int yc = min(max(y0 - 1 + r, 0), p.g.H - 1);
                const unsigned *bw = p.ybits + ((b * p.g.H + yc) * p.g.W) * wpp + (cb >> 5);
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const unsigned long long w2 = *reinterpret_cast<const unsigned long long *>(bw + xcs[c] * wpp);
                    mk |= ((w2 >> cl) & 1ULL) << (6 * r + c);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    vec1_t d[6][6];
    float rw[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int yc = min(max(y0 - 1 + r, 0), p.g.H - 1);
        const long long prow = (b * p.g.H + yc) * p.g.W;
        const float *dr = p.dy + prow * p.lddy, *rr = p.raw + prow * p.ldraw;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            // (plain loads: the windows of neighbouring tiles overlap - 2.25 reads per element, all but one served by the L2; a
            // non-temporal load does not leave the line there)
            d[r][c].x = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(dr) + offd[c]);
            rw[r][c] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(rr) + offr[c]);
        }
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int y = y0 - 1 + r;
        const bool rok = (unsigned)y < (unsigned)p.g.H;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int x = x0 - 1 + c;
            const bool ok = rok && (unsigned)x < (unsigned)p.g.W;
            const float yy = BITS ? (((mk >> (6 * r + c)) & 1ULL) ? 1.f : -1.f) : fmaf(rw[r][c], psc, psh);
            const float gv = d[r][c].x * (yy > 0.f ? 1.f : negd);
            if (p.dres && ok && r >= 1 && r <= 4 && c >= 1 && c <= 4)
                st1<NT>(p.dres + ((b * p.g.H + y) * p.g.W + x) * p.lddres + c0, gv);
            // coef*(gv - m1 - xhat*m2), xhat = (raw - mean)*invstd, with the per-channel products taken out of the loop: two float64
            // fused multiply-adds per element (A = coef*invstd*m2, Bc = coef*m1 - A*mean)
            const float o = (float)fma(-(double)rw[r][c], Ad, fma((double)cf, (double)gv, -Bd));
            d[r][c].x = ok ? o : 0.f;
        }
    }
    w2_bwd_store<NT>(d, p.Gy + t * 36 * p.N + c0, p.Vd + t * 36 * p.N + c0, p.N);
}

// ---- the same for a layer whose activation went straight into MaxPool2d(2,2) (vgg.py:69-83; k_maxpool2_affine forward): dy is the
// POOLED gradient [B][H/2][W/2][N]; dpre is dy * act' at the first maximum (scan order) of every 2x2 window of act(raw*pscale +
// pshift) and 0 elsewhere - exactly pool_cell_dpre() of backward.hip, which the two-pass form (k_pool_bn_bwd_apply -> draw ->
// k_w2_input + k_w2_dy) evaluates.  The 6x6 window of a tile (origin a multiple of 4) is covered by 4x4 whole pooling cells = the
// 8x8 raw region [y0-2, y0+5] x [x0-2, x0+5]: 64 raw + 16 pooled-gradient loads per (tile, channel).
template <bool NT>
__global__ void __launch_bounds__(TPB, 3)
k_w2_bwd_pool(const W2BwdArgs p) {
    const int sh = ilog2(p.N);
    const int idx = blockIdx.x * TPB + threadIdx.x;
    const int tx = __builtin_amdgcn_readfirstlane(idx >> sh), c0 = idx & (p.N - 1);
    if (tx >= p.g.TW) return;
    const int rowt = blockIdx.y, ty = rowt % p.g.TH;
    const long long b = rowt / p.g.TH;
    const long long t = (long long)rowt * p.g.TW + tx;
    const int y0 = 4 * ty, x0 = 4 * tx;
    const int Ho = p.g.H >> 1, Wo = p.g.W >> 1;
    const float mu = p.mean[c0], is = p.invstd[c0], cf = p.coef[c0];
    const double m1 = p.m1[c0], m2 = p.m2[c0];
    const float negd = p.act == 1 ? 0.f : (p.act == 2 ? p.slope : 1.f), nega = negd;      // act(t) = max(t,0) + nega*min(t,0); act'(t <= 0) = negd
    const double Ad = (double)cf * (double)is * m2, Bd = (double)cf * m1 - Ad * (double)mu;
    const float psc = p.pscale[c0], psh = p.pshift[c0];
    unsigned offr[8], offg[4];
#pragma unroll
    for (int c = 0; c < 8; ++c) offr[c] = 4u * (unsigned)(min(max(x0 - 2 + c, 0), p.g.W - 1) * (int)p.ldraw + c0);
#pragma unroll
    for (int j = 0; j < 4; ++j) offg[j] = 4u * (unsigned)(min(max((x0 >> 1) - 1 + j, 0), max(Wo - 1, 0)) * (int)p.lddy + c0);
    float rw[8][8], gp_[4][4];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int yc = min(max(y0 - 2 + r, 0), p.g.H - 1);
        const float *rr = p.raw + ((b * p.g.H + yc) * p.g.W) * p.ldraw;
#pragma unroll
        for (int c = 0; c < 8; ++c) rw[r][c] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(rr) + offr[c]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ic = min(max((y0 >> 1) - 1 + i, 0), max(Ho - 1, 0));
        const float *gr = p.dy + ((b * Ho + ic) * Wo) * p.lddy;
#pragma unroll
        for (int j = 0; j < 4; ++j) gp_[i][j] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(gr) + offg[j]);
    }
    vec1_t d[6][6];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ci = (y0 >> 1) - 1 + i;                       // pooling cell row (may be -1 or beyond the pooled map)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cj = (x0 >> 1) - 1 + j;
            const bool pooled = (unsigned)ci < (unsigned)Ho && (unsigned)cj < (unsigned)Wo;
            // the four activated values of the cell, first maximum in scan order (pool_cell_dpre)
            float e[4], tt[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                tt[q] = fmaf(rw[2 * i + (q >> 1)][2 * j + (q & 1)], psc, psh);
                e[q] = fmaxf(tt[q], 0.f) + nega * fminf(tt[q], 0.f);
            }
            int best = 0;
#pragma unroll
            for (int q = 1; q < 4; ++q) if (e[q] > e[best]) best = q;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 2 * i + (q >> 1) - 1, c = 2 * j + (q & 1) - 1;          // window coordinates of the element
                if (r < 0 || r > 5 || c < 0 || c > 5) continue;                           // (compile-time: the region's outer ring)
                const int y = y0 - 1 + r, x = x0 - 1 + c;
                const bool ok = (unsigned)y < (unsigned)p.g.H && (unsigned)x < (unsigned)p.g.W;
                const float gv = (pooled && q == best) ? gp_[i][j] * (tt[q] > 0.f ? 1.f : negd) : 0.f;
                const float o = (float)fma(-(double)rw[2 * i + (q >> 1)][2 * j + (q & 1)], Ad, fma((double)cf, (double)gv, -Bd));
                d[r][c].x = ok ? o : 0.f;
            }
        }
    }
    w2_bwd_store<NT>(d, p.Gy + t * 36 * p.N + c0, p.Vd + t * 36 * p.N + c0, p.N);
}

// ---- output transform + the k_gather_gemm epilogue.  Thread = (tile column, channel pair); a workgroup walks ROWS_PER_BLOCK
// rows of tiles and leaves one row of BatchNorm statistics (column sums of v and v^2).
struct W2OutArgs {
    const float *M; int N; TileGeo g;
    const float *bias, *scale, *shift, *residual; long long ldr;
    int act; float slope;
    float *out; long long ldo;
    float *stats;
    // BNM (efgh_gemm_desc.stats_mode 1): `stats` receives the BatchNorm-BACKWARD column sums of the written value instead of the
    // forward statistics - sum g and sum g * xhat, g = out * act'(raw*pscale + pshift), xhat = (raw - mean) * invstd - where raw is
    // the raw output of the layer whose activation this launch's `out` is the gradient of (that layer's reduction pass is then skipped)
    const float *bn_raw; long long bn_ldraw;
    const float *bn_psc, *bn_psh, *bn_mean, *bn_invstd; int bn_act; float bn_slope;
};

constexpr int ROWS_PER_BLOCK = 1;              // rows of tiles per workgroup of k_w2_output (one statistics row per workgroup)

// POOL: the layer is followed by MaxPool2d(2,2) (inference path of the VGG trunks, nets/vgg.py:69-83): a 4x4 output tile holds four
// whole pooling windows (tiles start at multiples of 4), so the epilogue writes max over each window of act(v*scale + shift) into
// the POOLED map [B][H/2][W/2][N] - the full-resolution activation is neither written nor read back by a pooling pass.
template <bool NT, bool POOL, bool BNM = false>
__global__ void __launch_bounds__(TPB, 4)          // (four workgroups per CU: <= 128 VGPRs - the pass is bound by bytes in flight)
k_w2_output(const W2OutArgs p) {
    __shared__ float red[2][TPB][VW];
    const int q4n = p.N / VW, sh = ilog2(q4n);
    const int idx = blockIdx.x * TPB + threadIdx.x;
    const int tx = idx >> sh, q = idx & (q4n - 1);
    const int col = q * VW;
    const vec_t one = make_float2(1.f, 1.f), zero = make_float2(0.f, 0.f);
    const vec_t bi = p.bias ? *reinterpret_cast<const vec_t *>(p.bias + col) : zero;
    const vec_t sc = p.scale ? *reinterpret_cast<const vec_t *>(p.scale + col) : one;
    const vec_t sf = p.shift ? *reinterpret_cast<const vec_t *>(p.shift + col) : zero;
    vec_t s1 = zero, s2 = zero;
    vec_t b_psc = zero, b_psh = zero, b_mu = zero, b_is = zero;
    float b_neg = 1.f;
    if (BNM) {
        b_psc = *reinterpret_cast<const vec_t *>(p.bn_psc + col); b_psh = *reinterpret_cast<const vec_t *>(p.bn_psh + col);
        b_mu = *reinterpret_cast<const vec_t *>(p.bn_mean + col); b_is = *reinterpret_cast<const vec_t *>(p.bn_invstd + col);
        b_neg = p.bn_act == 1 ? 0.f : (p.bn_act == 2 ? p.bn_slope : 1.f);
    }
    const long long as = p.N;                      // M [T][36][N]
    const int nrows = p.g.B * p.g.TH;
    for (int k = 0; k < ROWS_PER_BLOCK; ++k) {
        const int rowt = blockIdx.y * ROWS_PER_BLOCK + k;
        if (rowt >= nrows || tx >= p.g.TW) break;
        const int ty = rowt % p.g.TH;
        const long long b = rowt / p.g.TH;
        const long long t = (long long)rowt * p.g.TW + tx;
        const int y0 = 4 * ty, x0 = 4 * tx;
        const float *mp = p.M + t * 36 * p.N + col;
        // A^T M A streamed over the rows of M: the row pairs (1,2) and (3,4) belong to the point pairs +-a, +-b, so the column
        // transform only needs their sum and difference - two rows of M (24 registers) are live at a time instead of all 36
        // loads (72) + both intermediate arrays: 232 VGPRs / 2 waves per SIMD before, which held this HBM-bound pass at a third
        // of the rate of k_w2_input.  Same expressions, same order of additions as at4() applied along y.
        vec_t Y[4][4];
        {
            vec_t m0[6], w0[4];
#pragma unroll
            for (int j = 0; j < 6; ++j) m0[j] = ld2<NT>(mp + (long long)j * as);
            at4(m0, w0);
#pragma unroll
            for (int x = 0; x < 4; ++x) Y[0][x] = w0[x];
        }
        __builtin_amdgcn_sched_barrier(0);         // (keeps the scheduler from hoisting all 36 loads to the top again)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const float c1 = pr ? W2_B : W2_A, c2 = pr ? W2_B2 : W2_A2, c3 = pr ? W2_B3 : W2_A3;
            vec_t ma[6], mb[6], wa[4], wb[4];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                ma[j] = ld2<NT>(mp + (long long)(6 * (1 + 2 * pr) + j) * as);
                mb[j] = ld2<NT>(mp + (long long)(6 * (2 + 2 * pr) + j) * as);
            }
            at4(ma, wa);
            at4(mb, wb);
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                vec_t sm, df;
                sm.x = wa[x].x + wb[x].x; sm.y = wa[x].y + wb[x].y;
                df.x = wa[x].x - wb[x].x; df.y = wa[x].y - wb[x].y;
                Y[0][x].x += sm.x; Y[0][x].y += sm.y;
                if (pr == 0) {
                    Y[1][x].x = c1 * df.x; Y[1][x].y = c1 * df.y;
                    Y[2][x].x = c2 * sm.x; Y[2][x].y = c2 * sm.y;
                    Y[3][x].x = c3 * df.x; Y[3][x].y = c3 * df.y;
                } else {
                    Y[1][x].x = fmaf(c1, df.x, Y[1][x].x); Y[1][x].y = fmaf(c1, df.y, Y[1][x].y);
                    Y[2][x].x = fmaf(c2, sm.x, Y[2][x].x); Y[2][x].y = fmaf(c2, sm.y, Y[2][x].y);
                    Y[3][x].x = fmaf(c3, df.x, Y[3][x].x); Y[3][x].y = fmaf(c3, df.y, Y[3][x].y);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        {
            vec_t m5[6], w5[4];
#pragma unroll
            for (int j = 0; j < 6; ++j) m5[j] = ld2<NT>(mp + (long long)(30 + j) * as);
            at4(m5, w5);
#pragma unroll
            for (int x = 0; x < 4; ++x) { Y[3][x].x += w5[x].x; Y[3][x].y += w5[x].y; }
        }
        if (POOL) {
            const int Hp = p.g.H >> 1, Wp = p.g.W >> 1;
            float *ob = p.out + ((b * Hp + (y0 >> 1)) * Wp + (x0 >> 1)) * p.ldo + col;
            const int ldo_i = (int)p.ldo;
#pragma unroll
            for (int pa = 0; pa < 2; ++pa) {
                if ((y0 >> 1) + pa >= Hp) continue;            // (floor mode: an odd last row / column belongs to no window)
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    if ((x0 >> 1) + px >= Wp) continue;
                    vec_t best = zero;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        vec_t v = Y[2 * pa + (e >> 1)][2 * px + (e & 1)];
                        v.x = (v.x + bi.x) * sc.x + sf.x; v.y = (v.y + bi.y) * sc.y + sf.y;
                        if (p.act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); }
                        else if (p.act == 2) { v.x = v.x > 0.f ? v.x : v.x * p.slope; v.y = v.y > 0.f ? v.y : v.y * p.slope; }
                        if (e == 0) best = v;
                        else { best.x = fmaxf(best.x, v.x); best.y = fmaxf(best.y, v.y); }
                    }
                    st2<NT>(ob + (pa * Wp + px) * ldo_i, best);
                }
            }
            continue;
        }
        // one 64-bit base per operand, 32-bit element offsets inside the tile (4 rows x 4 pixels: far below 2^31 elements)
        const long long pix0 = (b * p.g.H + y0) * p.g.W + x0;
        float *ob = p.out + pix0 * p.ldo + col;
        const float *rb = p.residual ? p.residual + pix0 * p.ldr + col : nullptr;
        const float *bnr = BNM ? p.bn_raw + pix0 * p.bn_ldraw + col : nullptr;
        const int ldo_i = (int)p.ldo, ldr_i = (int)p.ldr, ldb_i = (int)p.bn_ldraw;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (y0 + a >= p.g.H) continue;
            vec_t rwv[4];
            if (BNM) {       // the row's four raw values first (clamped column: one round trip per row, not one per element)
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    rwv[x] = *reinterpret_cast<const vec_t *>(bnr + (a * p.g.W + min(x, p.g.W - 1 - x0)) * ldb_i);
            }
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                if (x0 + x >= p.g.W) continue;
                const int po = a * p.g.W + x;
                vec_t v = Y[a][x];
                v.x += bi.x; v.y += bi.y;
                if (!BNM) {
                    s1.x += v.x; s1.y += v.y;
                    s2.x = fmaf(v.x, v.x, s2.x); s2.y = fmaf(v.y, v.y, s2.y);
                }
                v.x = v.x * sc.x + sf.x; v.y = v.y * sc.y + sf.y;
                if (rb) {
                    const vec_t r = *reinterpret_cast<const vec_t *>(rb + po * ldr_i);
                    v.x += r.x; v.y += r.y;
                }
                if (p.act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); }
                else if (p.act == 2) { v.x = v.x > 0.f ? v.x : v.x * p.slope; v.y = v.y > 0.f ? v.y : v.y * p.slope; }
                st2<NT>(ob + po * ldo_i, v);
                if (BNM) {
                    const vec_t rw = rwv[x];
                    const float gx = v.x * (fmaf(rw.x, b_psc.x, b_psh.x) > 0.f ? 1.f : b_neg);
                    const float gy = v.y * (fmaf(rw.y, b_psc.y, b_psh.y) > 0.f ? 1.f : b_neg);
                    s1.x += gx; s1.y += gy;
                    s2.x = fmaf(gx, (rw.x - b_mu.x) * b_is.x, s2.x); s2.y = fmaf(gy, (rw.y - b_mu.y) * b_is.y, s2.y);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (p.stats) {
        red[0][threadIdx.x][0] = s1.x; red[0][threadIdx.x][1] = s1.y;
        red[1][threadIdx.x][0] = s2.x; red[1][threadIdx.x][1] = s2.y;
        __syncthreads();
        const int nslots = q4n >= TPB ? 1 : TPB / q4n;          // tile columns per workgroup
        if (threadIdx.x < q4n) {
            float a1[VW] = {0, 0}, a2[VW] = {0, 0};
            for (int s_ = 0; s_ < nslots; ++s_)
#pragma unroll
                for (int e = 0; e < VW; ++e) { a1[e] += red[0][s_ * q4n + threadIdx.x][e]; a2[e] += red[1][s_ * q4n + threadIdx.x][e]; }
            const long long brow = (long long)blockIdx.y * gridDim.x + blockIdx.x;
            float *o = p.stats + (brow * 2) * p.N + threadIdx.x * VW;
            *reinterpret_cast<vec_t *>(o) = make_float2(a1[0], a1[1]);
            *reinterpret_cast<vec_t *>(o + p.N) = make_float2(a2[0], a2[1]);
        }
    }
}

// U[6i + j][n][c] = sum_{kh,kw} G[i][kh] G[j][kw] Wp[n][kh*3 + kw][c]        (Wp: packed [N][9][C])
__global__ void k_w2_pack(const float *__restrict__ Wp, float *__restrict__ U, int N, int C) {
    const long long total = (long long)N * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
        w2_pack_item(Wp, U, N, C, i);
}

// every Winograd-domain weight of a model in ONE launch (after an optimizer step all of them are stale, like the packed layouts they
// are made from: 96 launches of 5-12 us per training step otherwise).  Job j owns the workgroups [first_block[j], first_block[j+1]):
// 256 work items each, no workgroup straddles two jobs.
__global__ void __launch_bounds__(256) k_wino_pack_batched(const efgh_wino_pack_job *__restrict__ jobs, int njobs) {
    __shared__ int job_s;
    if (threadIdx.x == 0) {
        int lo = 0, hi = njobs - 1;                      // largest j with first_block[j] <= blockIdx.x
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (jobs[mid].first_block <= (long long)blockIdx.x) lo = mid; else hi = mid - 1; }
        job_s = lo;
    }
    __syncthreads();
    const efgh_wino_pack_job j = jobs[job_s];
    const long long i = ((long long)blockIdx.x - j.first_block) * 256 + threadIdx.x;
    if (j.kind == 0) { if (i < 3LL * j.C * j.N) wino_pack_item(j.Wp, j.U, j.N, j.C, i); }
    else if (i < (long long)j.N * j.C) w2_pack_item(j.Wp, j.U, j.N, j.C, i);
}

// dWp[n][kh*3 + kw][c] = sum_{i,j} A3T[kh][i] A3T[kw][j] S[6i + j][n][c]
// out: the packed [N][9][C] gradient (on = 9 C, oc = 1, ot = C) or, with an armed unpack descriptor, the reference layout itself
// (rows < Nr, channels < Cr only)
__global__ void k_w2_wfinish(const float *__restrict__ S, float *__restrict__ dWp, int N, int C, long long on, long long oc, long long ot,
                             int Nr, int Cr, int accumulate) {
    const float A3[3][6] = {{1, 1, 1, 1, 1, 0}, {0, W2_A, -W2_A, W2_B, -W2_B, 0}, {0, W2_A2, W2_A2, W2_B2, W2_B2, 1}};
    const long long total = (long long)N * C, plane = total;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C); const long long n = i / C;
        float r[6][3];               // r[i][kw] = sum_j A3[kw][j] S[6i + j]
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            float s[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) s[j] = S[(long long)(6 * a + j) * plane + i];
            r[a][0] = s[0] + (s[1] + s[2]) + (s[3] + s[4]);
            r[a][1] = W2_A * (s[1] - s[2]) + W2_B * (s[3] - s[4]);
            r[a][2] = W2_A2 * (s[1] + s[2]) + W2_B2 * (s[3] + s[4]) + s[5];
        }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                float v = 0.f;
#pragma unroll
                for (int a = 0; a < 6; ++a) v += A3[kh][a] * r[a][kw];
                if (n < Nr && c < Cr) {
                    float *o = dWp + n * on + c * oc + (kh * 3 + kw) * ot;
                    *o = accumulate ? *o + v : v;
                }
            }
    }
}

bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

bool supported(const efgh_gemm_desc *d) {
    if (!d || d->mode != 1 || d->T != 9 || d->M_dev || d->nbatch > 1) return false;
    if (d->C % 4 || d->N % 4 || !pow2(d->N / VW) || d->N / VW > TPB || !pow2(d->C / VW) || d->C / VW > TPB) return false;
    if (d->sh != 1 || d->sw != 1 || d->osh != 1 || d->osw != 1 || d->oh0 || d->ow0) return false;
    if (d->Hv != d->Hin || d->Wv != d->Win || d->Ho != d->Hin || d->Wo != d->Win) return false;
    for (int t = 0; t < 9; ++t) if (d->dh[t] != t / 3 - 1 || d->dw[t] != t % 3 - 1) return false;
    return d->lda % 4 == 0;
}

TileGeo geo(int B, int H, int W) {
    TileGeo g;
    g.B = B; g.H = H; g.W = W; g.TH = (H + 3) / 4; g.TW = (W + 3) / 4;
    g.T = (long long)B * g.TH * g.TW;
    return g;
}
}  // namespace

extern "C" int efgh_wino2d_supported(const efgh_gemm_desc *d) { return supported(d) ? 1 : 0; }

extern "C" int64_t efgh_wino2d_tiles(int32_t B, int32_t H, int32_t W) { return geo(B, H, W).T; }

static dim3 row_grid(const TileGeo &g, int pairs, int rows_per_block) {
    return dim3((unsigned)(((long long)g.TW * pairs + TPB - 1) / TPB), (unsigned)((g.B * g.TH + rows_per_block - 1) / rows_per_block));
}

extern "C" int32_t efgh_wino2d_stats_rows(int32_t B, int32_t H, int32_t W, int32_t N) {
    const dim3 gr = row_grid(geo(B, H, W), N / VW, ROWS_PER_BLOCK);
    return (int32_t)(gr.x * gr.y);
}

extern "C" int efgh_wino2d_pack(const float *Wp, float *U, int32_t N, int32_t C, void *stream_) {
    EFGH_CHECK_ARG(Wp && U && N > 0 && C > 0);
    const long long total = (long long)N * C;
    long long g = (total + 255) / 256;
    k_w2_pack<<<(int)(g > 8192 ? 8192 : g), 256, 0, (hipStream_t)stream_>>>(Wp, U, N, C);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_wino_pack_batched(const efgh_wino_pack_job *jobs_dev, int32_t njobs, int64_t nblocks, void *stream_) {
    EFGH_CHECK_ARG(jobs_dev && njobs > 0 && nblocks > 0 && nblocks < 0x7fffffffLL);
    k_wino_pack_batched<<<(unsigned)nblocks, 256, 0, (hipStream_t)stream_>>>(jobs_dev, njobs);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_wino2d_input(const float *A, int64_t lda, int32_t C, int32_t B, int32_t H, int32_t W, float *V,
                                 void *stream_) {
    EFGH_CHECK_ARG(A && V && C > 0 && C % 4 == 0 && lda % 4 == 0 && B > 0 && H > 0 && W > 0);
    EFGH_CHECK_ARG((((uintptr_t)A) & 15) == 0 && (((uintptr_t)V) & 15) == 0);
    const TileGeo g = geo(B, H, W);
    EFGH_CHECK_ARG(pow2(C / VW) && (long long)B * g.TH < 65536);
    if (efgh_stream_nt(g.T * 36ll * C * 4)) k_w2_input<true, false><<<row_grid(g, C / VW, 1), TPB, 0, (hipStream_t)stream_>>>(A, lda, C, g, V, nullptr, nullptr, 0, 0.f);
    else k_w2_input<false, false><<<row_grid(g, C / VW, 1), TPB, 0, (hipStream_t)stream_>>>(A, lda, C, g, V, nullptr, nullptr, 0, 0.f);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

/* the same transform of act(A * scale[c] + shift[c]): A = the raw output of a train-mode BatchNorm layer (see k_w2_input<.., AFF>) */
extern "C" int efgh_wino2d_input_act(const float *A, int64_t lda, int32_t C, int32_t B, int32_t H, int32_t W, const float *scale,
                                     const float *shift, int32_t act, float slope, float *V, void *stream_) {
    EFGH_CHECK_ARG(A && V && scale && shift && C > 0 && C % 4 == 0 && lda % 4 == 0 && B > 0 && H > 0 && W > 0 && act >= 0 && act <= 2);
    EFGH_CHECK_ARG((((uintptr_t)A) & 15) == 0 && (((uintptr_t)V) & 15) == 0 && (((uintptr_t)scale) & 7) == 0 && (((uintptr_t)shift) & 7) == 0);
    const TileGeo g = geo(B, H, W);
    EFGH_CHECK_ARG(pow2(C / VW) && (long long)B * g.TH < 65536);
    if (efgh_stream_nt(g.T * 36ll * C * 4)) k_w2_input<true, true><<<row_grid(g, C / VW, 1), TPB, 0, (hipStream_t)stream_>>>(A, lda, C, g, V, scale, shift, act, slope);
    else k_w2_input<false, true><<<row_grid(g, C / VW, 1), TPB, 0, (hipStream_t)stream_>>>(A, lda, C, g, V, scale, shift, act, slope);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_wino2d_dy(const float *G, int64_t ldg, int32_t N, int32_t B, int32_t H, int32_t W, float *Gy,
                              void *stream_) {
    EFGH_CHECK_ARG(G && Gy && N > 0 && N % 4 == 0 && ldg % 4 == 0 && B > 0 && H > 0 && W > 0);
    EFGH_CHECK_ARG((((uintptr_t)G) & 15) == 0 && (((uintptr_t)Gy) & 15) == 0);
    const TileGeo g = geo(B, H, W);
    EFGH_CHECK_ARG(pow2(N / VW) && (long long)B * g.TH < 65536);
    if (efgh_stream_nt(g.T * 36ll * N * 4)) k_w2_dy<true><<<row_grid(g, N / VW, 1), TPB, 0, (hipStream_t)stream_>>>(G, ldg, N, g, Gy);
    else k_w2_dy<false><<<row_grid(g, N / VW, 1), TPB, 0, (hipStream_t)stream_>>>(G, ldg, N, g, Gy);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_wino2d_bwd_transforms(const float *dy, int64_t lddy, const float *raw, int64_t ldraw, const uint32_t *ybits,
                                          const float *pscale, const float *pshift, const float *mean, const float *invstd,
                                          const float *coef, const double *m1, const double *m2, int32_t N, int32_t B, int32_t H,
                                          int32_t W, int32_t act, float slope, float *Vd, float *Gy, float *dres, int64_t lddres,
                                          void *stream_) {
    EFGH_CHECK_ARG(dy && raw && mean && invstd && coef && m1 && m2 && Vd && Gy && N > 0 && N % 4 == 0 && B > 0 && H > 0 && W > 0);
    EFGH_CHECK_ARG((ybits != nullptr) != (pscale != nullptr && pshift != nullptr) && act >= 0 && act <= 2);
    EFGH_CHECK_ARG(lddy % 4 == 0 && ldraw % 4 == 0 && (!dres || lddres % 4 == 0) && (!ybits || N % 32 == 0));
    EFGH_CHECK_ARG(((((uintptr_t)dy) | ((uintptr_t)raw) | ((uintptr_t)Vd) | ((uintptr_t)Gy) | ((uintptr_t)dres)) & 15) == 0);
    W2BwdArgs a;
    a.dy = dy; a.lddy = lddy; a.raw = raw; a.ldraw = ldraw; a.ybits = ybits;
    a.mean = mean; a.invstd = invstd; a.coef = coef; a.pscale = pscale; a.pshift = pshift; a.m1 = m1; a.m2 = m2;
    a.N = N; a.act = act; a.slope = slope; a.g = geo(B, H, W);
    a.Vd = Vd; a.Gy = Gy; a.dres = dres; a.lddres = lddres;
    EFGH_CHECK_ARG(pow2(N) && N >= 64 && (long long)B * a.g.TH < 65536 && (((uintptr_t)ybits) & 7) == 0);
    EFGH_CHECK_ARG((long long)W * lddy < (1ll << 30) && (long long)W * ldraw < (1ll << 30));        // (32-bit lane offsets inside a row)
    const dim3 grid = row_grid(a.g, N, 1);             // (one channel per lane)
    const bool nt = efgh_stream_nt(a.g.T * 36ll * N * 4);
    if (ybits) { if (nt) k_w2_bwd<true, true><<<grid, TPB, 0, (hipStream_t)stream_>>>(a); else k_w2_bwd<false, true><<<grid, TPB, 0, (hipStream_t)stream_>>>(a); }
    else { if (nt) k_w2_bwd<true, false><<<grid, TPB, 0, (hipStream_t)stream_>>>(a); else k_w2_bwd<false, false><<<grid, TPB, 0, (hipStream_t)stream_>>>(a); }
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_wino2d_bwd_transforms_pooled(const float *dy_pool, int64_t lddy, const float *raw, int64_t ldraw, const float *pscale,
                                                 const float *pshift, const float *mean, const float *invstd, const float *coef,
                                                 const double *m1, const double *m2, int32_t N, int32_t B, int32_t H, int32_t W,
                                                 int32_t act, float slope, float *Vd, float *Gy, void *stream_) {
    EFGH_CHECK_ARG(dy_pool && raw && pscale && pshift && mean && invstd && coef && m1 && m2 && Vd && Gy && N > 0 && B > 0 && H >= 2 && W >= 2);
    EFGH_CHECK_ARG(lddy % 4 == 0 && ldraw % 4 == 0 && act >= 0 && act <= 2);
    EFGH_CHECK_ARG(((((uintptr_t)dy_pool) | ((uintptr_t)raw) | ((uintptr_t)Vd) | ((uintptr_t)Gy)) & 15) == 0);
    W2BwdArgs a;
    a.dy = dy_pool; a.lddy = lddy; a.raw = raw; a.ldraw = ldraw; a.ybits = nullptr;
    a.mean = mean; a.invstd = invstd; a.coef = coef; a.pscale = pscale; a.pshift = pshift; a.m1 = m1; a.m2 = m2;
    a.N = N; a.act = act; a.slope = slope; a.g = geo(B, H, W);
    a.Vd = Vd; a.Gy = Gy; a.dres = nullptr; a.lddres = 0;
    EFGH_CHECK_ARG(pow2(N) && N >= 64 && (long long)B * a.g.TH < 65536);
    EFGH_CHECK_ARG((long long)W * lddy < (1ll << 30) && (long long)W * ldraw < (1ll << 30));
    const dim3 grid = row_grid(a.g, N, 1);
    if (efgh_stream_nt(a.g.T * 36ll * N * 4)) k_w2_bwd_pool<true><<<grid, TPB, 0, (hipStream_t)stream_>>>(a);
    else k_w2_bwd_pool<false><<<grid, TPB, 0, (hipStream_t)stream_>>>(a);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_wino2d_output(const float *M, const efgh_gemm_desc *d, void *stream_) {
    EFGH_CHECK_ARG(M && supported(d) && d->out && d->B > 0);
    EFGH_CHECK_ARG((((uintptr_t)M) & 15) == 0 && (((uintptr_t)d->out) & 15) == 0 && d->ldo % 4 == 0);
    EFGH_CHECK_ARG(!d->residual || (d->ldr % 4 == 0 && (((uintptr_t)d->residual) & 15) == 0));
    W2OutArgs a;
    a.M = M; a.N = d->N; a.g = geo(d->B, d->Hin, d->Win);
    a.bias = d->bias; a.scale = d->scale; a.shift = d->shift; a.residual = d->residual; a.ldr = d->ldr;
    a.act = d->act; a.slope = d->slope; a.out = d->out; a.ldo = d->ldo; a.stats = d->stats;
    a.bn_raw = nullptr; a.bn_ldraw = 0; a.bn_psc = a.bn_psh = a.bn_mean = a.bn_invstd = nullptr; a.bn_act = 0; a.bn_slope = 0.f;
    const dim3 grid = row_grid(a.g, d->N / VW, ROWS_PER_BLOCK);
    const bool nt = efgh_stream_nt(a.g.T * 36ll * d->N * 4);
    if (d->stats && d->stats_mode == 1) {
        // the BatchNorm-backward sums of the layer this launch writes the gradient of (mask re-derived from raw: layers without a residual)
        EFGH_CHECK_ARG(d->bn_raw && d->bn_mean && d->bn_invstd && d->bn_pscale && d->bn_pshift && !d->bn_y && d->bn_ldraw % 4 == 0);
        EFGH_CHECK_ARG((long long)4 * d->Win * d->bn_ldraw < (1ll << 31) && (((uintptr_t)d->bn_raw) & 7) == 0);
        a.bn_raw = d->bn_raw; a.bn_ldraw = d->bn_ldraw; a.bn_psc = d->bn_pscale; a.bn_psh = d->bn_pshift;
        a.bn_mean = d->bn_mean; a.bn_invstd = d->bn_invstd; a.bn_act = d->bn_act; a.bn_slope = d->bn_slope;
        if (nt) k_w2_output<true, false, true><<<grid, TPB, 0, (hipStream_t)stream_>>>(a);
        else k_w2_output<false, false, true><<<grid, TPB, 0, (hipStream_t)stream_>>>(a);
    } else if (nt) k_w2_output<true, false><<<grid, TPB, 0, (hipStream_t)stream_>>>(a);
    else k_w2_output<false, false><<<grid, TPB, 0, (hipStream_t)stream_>>>(a);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_wino2d_output_pooled(const float *M, const efgh_gemm_desc *d, void *stream_) {
    EFGH_CHECK_ARG(M && supported(d) && d->out && d->B > 0 && d->Hin >= 2 && d->Win >= 2);
    EFGH_CHECK_ARG((((uintptr_t)M) & 15) == 0 && (((uintptr_t)d->out) & 15) == 0 && d->ldo % 4 == 0);
    EFGH_CHECK_ARG(!d->residual && !d->stats);          // (an inference epilogue: bias, folded BatchNorm affine, activation)
    W2OutArgs a;
    a.M = M; a.N = d->N; a.g = geo(d->B, d->Hin, d->Win);
    a.bias = d->bias; a.scale = d->scale; a.shift = d->shift; a.residual = nullptr; a.ldr = 0;
    a.act = d->act; a.slope = d->slope; a.out = d->out; a.ldo = d->ldo; a.stats = nullptr;
    a.bn_raw = nullptr; a.bn_ldraw = 0; a.bn_psc = a.bn_psh = a.bn_mean = a.bn_invstd = nullptr; a.bn_act = 0; a.bn_slope = 0.f;
    if (efgh_stream_nt(a.g.T * 36ll * d->N * 4)) k_w2_output<true, true><<<row_grid(a.g, d->N / VW, ROWS_PER_BLOCK), TPB, 0, (hipStream_t)stream_>>>(a);
    else k_w2_output<false, true><<<row_grid(a.g, d->N / VW, ROWS_PER_BLOCK), TPB, 0, (hipStream_t)stream_>>>(a);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_wino2d_wfinish(const float *S, float *dWp, int32_t N, int32_t C, const efgh_wgrad_out_desc *out, void *stream_) {
    EFGH_CHECK_ARG(S && dWp && N > 0 && C > 0);
    const long long total = (long long)N * C;
    long long g = (total + 255) / 256;
    const bool direct = efgh_wgrad_out_fits(out, N, 9, C);          // straight into the caller's layout (no packed plane, no unpack launch)
    if (direct)
        k_w2_wfinish<<<(int)(g > 4096 ? 4096 : g), 256, 0, (hipStream_t)stream_>>>(S, out->W, N, C, out->sn, out->sc, out->st, out->N, out->C, out->accumulate);
    else
        k_w2_wfinish<<<(int)(g > 4096 ? 4096 : g), 256, 0, (hipStream_t)stream_>>>(S, dWp, N, C, 9LL * C, 1, C, N, C, 0);
    EFGH_CHECK_LAUNCH();
    return direct ? EFGH_WROTE_OUT : EFGH_OK;
}
