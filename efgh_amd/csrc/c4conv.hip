// 3x3 convolutions with FOUR input channels per tap and 32..128 outputs on fp32 MFMA (gfx950): the RGB / range / depth
// input layers (nets/vgg.py:77 first layer, nets/gnet.py:21,80) and the data gradient of G's 1- and 2-channel transposed
// heads (a stride-2 3x3 convolution over the 4-channel-padded output gradient).  K = 36 only, so the layer is bound by
// writing (forward) or reading (weight gradient) the wide side; the generic implicit-GEMM kernel spends its time on the
// per-quad gather arithmetic instead (12-26 TFLOP/s, 1.9 TB/s).  Here one WAVE owns 32 consecutive output pixels of
// an image row (no workgroup barrier anywhere in the loop): the three input rows go to its LDS slice once (two planes of channel pairs, so that the 32 pixels of an MFMA operand
// are 256 contiguous bytes), the 36 x N weights live in registers for the whole (persistent) workgroup, and each wave
// runs 18 v_mfma_f32_32x32x2_f32 per 32 pixels x 32 outputs: MFMA step (tap t, e) contracts channels {e, 2 + e}.
// Epilogue = that of k_gather_gemm (bias, BN statistics of the pre-activation value, scale/shift, residual, activation).
//
// Weight gradient: dW[n][t][c] = sum_p G[p][n] * X[p + tap t][c] on v_mfma_f32_16x16x4_f32 (k = 4 output pixels per
// instruction; 36 columns = 2.25 -> 3 column tiles), G read once straight from global memory, per-workgroup partial sums
// combined with fp32 atomics.
#include "common.h"


namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TP = 32;                  // output pixels per unit: one wave, no workgroup barrier in the loop
constexpr int LWMAX = (TP - 1) * 2 + 3; // staged input pixels per row at stride 2
constexpr int WAVES = 4;

struct C4Args {
    const float *A; long long lda;
    int B, Hin, Win, Ho, Wo, s;
    const float *W; int N;
    const float *bias, *scale, *shift, *residual; long long ldr;
    int act; float slope;
    float *out; long long ldo;
    float *stats;
    long long units; int jblocks;
    const float *G; long long ldg; float *dW;       // wgrad
};

__device__ __forceinline__ void unit_coords(const C4Args &p, long long unit, long long &row, int &i, long long &b, int &j0) {
    const int jb = (int)(unit % p.jblocks);
    row = unit / p.jblocks;                          // b * Ho + i
    i = (int)(row % p.Ho); b = row / p.Ho;
    j0 = jb * TP;
}

// the (up to) four staged pixels of this lane for one unit: flat index idx = lane + 64 q over [3][LW]
__device__ __forceinline__ float4 load_px(const C4Args &p, int idx, int LW, int i, long long b, int j0) {
    const int kh = idx / LW, x = idx - kh * LW;
    const int yin = p.s * i - 1 + kh, xin = p.s * j0 - 1 + x;
    if (kh < 3 && (unsigned)yin < (unsigned)p.Hin && (unsigned)xin < (unsigned)p.Win)
        return *reinterpret_cast<const float4 *>(p.A + ((b * p.Hin + yin) * p.Win + xin) * p.lda);
    return make_float4(0.f, 0.f, 0.f, 0.f);
}

// LDS traffic of ONE wave: the hardware keeps a wave's LDS instructions in order, the fences keep the compiler from moving them
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int NT, bool RES>
__global__ void __launch_bounds__(64 * WAVES, 2) k_c4_conv(const C4Args p) {
    __shared__ float2 Pw[WAVES][3][2][LWMAX + 1];
    __shared__ float red[2][WAVES][NT * 32];
    constexpr int TPITCH = 40;                       // floats per row of the transpose tile (h = 0 / 1 rows land on disjoint banks)
    __shared__ __attribute__((aligned(16))) float Tw[WAVES][32 * TPITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int LW = (TP - 1) * p.s + 3;
    float2 (*P)[2][LWMAX + 1] = Pw[wave];
    float *T = Tw[wave];

    float2 bw[9][NT];
    float bi[NT], sc[NT], sf[NT], s1[NT], s2[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = 32 * j + r;
#pragma unroll
        for (int t = 0; t < 9; ++t) bw[t][j] = *reinterpret_cast<const float2 *>(p.W + ((long long)n * 9 + t) * 4 + 2 * h);
        bi[j] = p.bias ? p.bias[n] : 0.f;
        sc[j] = p.scale ? p.scale[n] : 1.f;
        sf[j] = p.shift ? p.shift[n] : 0.f;
        s1[j] = 0.f; s2[j] = 0.f;
    }
    const float neg = p.act == 1 ? 0.f : (p.act == 2 ? p.slope : 1.f);

    // the weight loads are complete before the loop (an empty asm that reads them): otherwise the loop body inherits their
    // outstanding-load state and waits on the NEXT unit's prefetch in the middle of the MFMAs
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int t = 0; t < 9; ++t) asm volatile("" ::"v"(bw[t][j].x), "v"(bw[t][j].y));

    const long long nwaves = (long long)gridDim.x * WAVES;
    long long unit = (long long)blockIdx.x * WAVES + wave;
    long long row = 0, b = 0; int i = 0, j0 = 0;     // coordinates of the unit whose pixels are in pf0..3
    long long crow = 0; int cj0 = 0;                 // ... of the unit staged in LDS
    float4 pf0, pf1, pf2, pf3;                       // (named registers: no scratch)
#define EFGH_FETCH(u)                                                                    \
    {                                                                                   \
        unit_coords(p, u, row, i, b, j0);                                               \
        pf0 = load_px(p, lane, LW, i, b, j0); pf1 = load_px(p, lane + 64, LW, i, b, j0); \
        pf2 = load_px(p, lane + 128, LW, i, b, j0); pf3 = load_px(p, lane + 192, LW, i, b, j0); \
    }
#define EFGH_PUT(q, v)                                                                   \
    {                                                                                   \
        const int idx = lane + 64 * q;                                                  \
        if (idx < 3 * LW) {                                                             \
            const int kh = idx / LW, x = idx - kh * LW;                                 \
            P[kh][0][x] = make_float2(v.x, v.y); P[kh][1][x] = make_float2(v.z, v.w);   \
        }                                                                               \
    }
#define EFGH_STAGE()                                                                     \
    {                                                                                   \
        crow = row; cj0 = j0;                                                           \
        EFGH_PUT(0, pf0) EFGH_PUT(1, pf1) EFGH_PUT(2, pf2) EFGH_PUT(3, pf3)             \
    }
    if (unit < p.units) {
        EFGH_FETCH(unit)
        EFGH_STAGE()
        if (unit + nwaves < p.units) EFGH_FETCH(unit + nwaves)
    }
    // per unit: operand reads + MFMAs, THEN the LDS refill with the next unit (its loads are one MFMA phase old, and so are
    // the previous unit's stores: the vmcnt(0) in front of the refill costs nothing), the loads of the unit after that, and
    // only then this unit's stores
    for (; unit < p.units; unit += nwaves) {
        const long long orow0 = crow; const int oj0 = cj0;
        wave_lds_sync();
        f32x16 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
        const int px = r * p.s;
        float2 af[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) af[t] = P[t / 3][h][px + t % 3];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t].x, bw[t][j].x, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t].y, bw[t][j].y, acc[j], 0, 0, 0);
            }
        }
        wave_lds_sync();
        if (unit + nwaves < p.units) {
            EFGH_STAGE()
            if (unit + 2 * nwaves < p.units) EFGH_FETCH(unit + 2 * nwaves)
        }
        // branch-free epilogue: act(v) = max(v, 0) + neg * min(v, 0) with neg = 0 (ReLU) / slope (leaky) / 1 (none).  The accumulator
        // layout has one output channel per lane (4-byte stores); every 32 x 32 block goes through a wave-private LDS tile
        // [pixel][channel] and leaves as 16-byte stores, eight lanes per pixel row (128 contiguous bytes)
        const bool full = oj0 + TP <= p.Wo;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            // the residual is read in the STORE layout (16 bytes per lane, issued before the tile is even written, so the loads
            // overlap the LDS round trip): the first version read it in the accumulator layout - sixteen dependent 4-byte loads
            // per lane and block in a wave that has nobody to hide their latency behind (3.6 ms instead of 0.6 on G's depth head)
            float4 rr[4];
            if (RES) {
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int pl = 8 * pass + (lane >> 3), c4 = (lane & 7) * 4;
                    rr[pass] = (full || oj0 + pl < p.Wo)
                                   ? *reinterpret_cast<const float4 *>(p.residual + (orow0 * p.Wo + oj0 + pl) * p.ldr + 32 * j + c4)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            wave_lds_sync();                         // the tile's previous readers are done
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int pl = (q & 3) + 8 * (q >> 2) + 4 * h;
                const bool ok = full || oj0 + pl < p.Wo;
                float v = acc[j][q] + bi[j];
                if (ok) { s1[j] += v; s2[j] = fmaf(v, v, s2[j]); }
                v = fmaf(v, sc[j], sf[j]);
                T[pl * TPITCH + r] = RES ? v : act_neg(v, neg);
            }
            wave_lds_sync();
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int pl = 8 * pass + (lane >> 3), c4 = (lane & 7) * 4;
                float4 v = *reinterpret_cast<const float4 *>(&T[pl * TPITCH + c4]);
                if (RES) {
                    v.x += rr[pass].x; v.y += rr[pass].y; v.z += rr[pass].z; v.w += rr[pass].w;
                    v.x = act_neg(v.x, neg); v.y = act_neg(v.y, neg);
                    v.z = act_neg(v.z, neg); v.w = act_neg(v.w, neg);
                }
                if (full || oj0 + pl < p.Wo)
                    *reinterpret_cast<float4 *>(p.out + (orow0 * p.Wo + oj0 + pl) * p.ldo + 32 * j + c4) = v;
            }
        }
    }
#undef EFGH_FETCH
#undef EFGH_PUT
#undef EFGH_STAGE
    if (p.stats) {                                   // one statistics row per workgroup
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            s1[j] += __shfl_xor(s1[j], 32); s2[j] += __shfl_xor(s2[j], 32);
            if (h == 0) { red[0][wave][32 * j + r] = s1[j]; red[1][wave][32 * j + r] = s2[j]; }
        }
        __syncthreads();
        for (int n = tid; n < NT * 32; n += 64 * WAVES) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) { a1 += red[0][w][n]; a2 += red[1][w][n]; }
            p.stats[((long long)blockIdx.x * 2 + 0) * p.N + n] = a1;
            p.stats[((long long)blockIdx.x * 2 + 1) * p.N + n] = a2;
        }
    }
}

// ---- the same layer followed by MaxPool2d(2,2) (inference path of the VGG trunks, nets/vgg.py:69-83) ----------------------------
// unit = 32 output pixels of a ROW PAIR (2 ip, 2 ip + 1): four input rows in the wave's LDS slice, both rows' accumulators in
// registers (2 x NT x 16), the window maximum of act((v + bias) * scale + shift) taken vertically in the accumulator layout and
// horizontally on the way out of the transpose tile - 16 pooled pixels x 32 channels leave per tile, the full-resolution map
// (4 GB at 15.7 M pixels x 64 channels) is neither written nor read back by a pooling pass.  Stride 1 only; same MFMA order as
// k_c4_conv, so the result is bit-identical to k_c4_conv followed by efgh_maxpool2.
constexpr int LW1 = TP + 2;
__device__ __forceinline__ float4 load_px4(const C4Args &p, int idx, int ip, long long b, int j0) {
    const int kh = idx / LW1, x = idx - kh * LW1;
    const int yin = 2 * ip - 1 + kh, xin = j0 - 1 + x;
    if (kh < 4 && (unsigned)yin < (unsigned)p.Hin && (unsigned)xin < (unsigned)p.Win)
        return *reinterpret_cast<const float4 *>(p.A + ((b * p.Hin + yin) * p.Win + xin) * p.lda);
    return make_float4(0.f, 0.f, 0.f, 0.f);
}

template <int NT>
__global__ void __launch_bounds__(64 * WAVES, 2) k_c4_conv_pool(const C4Args p) {
    __shared__ float2 Pw[WAVES][4][2][LW1 + 1];
    constexpr int TPITCH = 40;
    __shared__ __attribute__((aligned(16))) float Tw[WAVES][32 * TPITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    float2 (*P)[2][LW1 + 1] = Pw[wave];
    float *T = Tw[wave];
    const int Hp = p.Ho >> 1, Wp = p.Wo >> 1;

    float2 bw[9][NT];
    float bi[NT], sc[NT], sf[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = 32 * j + r;
#pragma unroll
        for (int t = 0; t < 9; ++t) bw[t][j] = *reinterpret_cast<const float2 *>(p.W + ((long long)n * 9 + t) * 4 + 2 * h);
        bi[j] = p.bias ? p.bias[n] : 0.f;
        sc[j] = p.scale ? p.scale[n] : 1.f;
        sf[j] = p.shift ? p.shift[n] : 0.f;
    }
    const float neg = p.act == 1 ? 0.f : (p.act == 2 ? p.slope : 1.f);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int t = 0; t < 9; ++t) asm volatile("" ::"v"(bw[t][j].x), "v"(bw[t][j].y));

    const long long nwaves = (long long)gridDim.x * WAVES;
    long long unit = (long long)blockIdx.x * WAVES + wave;
    long long rowp = 0, b = 0; int ip = 0, j0 = 0;   // coordinates of the unit whose pixels are in pf0..2 (rowp = b * Hp + ip)
    long long crow = 0; int cj0 = 0;                 // ... of the unit staged in LDS
    float4 pf0, pf1, pf2;                            // 4 x 34 staged pixels = 136 <= 192
#define EFGH_FETCH(u)                                                                    \
    {                                                                                   \
        const int jb = (int)((u) % p.jblocks);                                          \
        rowp = (u) / p.jblocks; ip = (int)(rowp % Hp); b = rowp / Hp; j0 = jb * TP;     \
        pf0 = load_px4(p, lane, ip, b, j0); pf1 = load_px4(p, lane + 64, ip, b, j0);    \
        pf2 = load_px4(p, lane + 128, ip, b, j0);                                       \
    }
#define EFGH_PUT(q, v)                                                                   \
    {                                                                                   \
        const int idx = lane + 64 * q;                                                  \
        if (idx < 4 * LW1) {                                                            \
            const int kh = idx / LW1, x = idx - kh * LW1;                               \
            P[kh][0][x] = make_float2(v.x, v.y); P[kh][1][x] = make_float2(v.z, v.w);   \
        }                                                                               \
    }
#define EFGH_STAGE()                                                                     \
    {                                                                                   \
        crow = rowp; cj0 = j0;                                                          \
        EFGH_PUT(0, pf0) EFGH_PUT(1, pf1) EFGH_PUT(2, pf2)                              \
    }
    if (unit < p.units) {
        EFGH_FETCH(unit)
        EFGH_STAGE()
        if (unit + nwaves < p.units) EFGH_FETCH(unit + nwaves)
    }
    for (; unit < p.units; unit += nwaves) {
        const long long orow0 = crow; const int oj0 = cj0;
        wave_lds_sync();
        f32x16 acc[2][NT];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[half][j][q] = 0.f;
            float2 af[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) af[t] = P[t / 3 + half][h][r + t % 3];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[half][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t].x, bw[t][j].x, acc[half][j], 0, 0, 0);
                    acc[half][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t].y, bw[t][j].y, acc[half][j], 0, 0, 0);
                }
            }
        }
        wave_lds_sync();
        if (unit + nwaves < p.units) {
            EFGH_STAGE()
            if (unit + 2 * nwaves < p.units) EFGH_FETCH(unit + 2 * nwaves)
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            wave_lds_sync();                         // the tile's previous readers are done
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int pl = (q & 3) + 8 * (q >> 2) + 4 * h;
                const float v0 = act_neg(fmaf(acc[0][j][q] + bi[j], sc[j], sf[j]), neg);
                const float v1 = act_neg(fmaf(acc[1][j][q] + bi[j], sc[j], sf[j]), neg);
                T[pl * TPITCH + r] = fmaxf(v0, v1);
            }
            wave_lds_sync();
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int item = 64 * pass + lane, pp = item >> 3, c4 = (item & 7) * 4;
                const float4 a = *reinterpret_cast<const float4 *>(&T[(2 * pp) * TPITCH + c4]);
                const float4 c = *reinterpret_cast<const float4 *>(&T[(2 * pp + 1) * TPITCH + c4]);
                float4 v;
                v.x = fmaxf(a.x, c.x); v.y = fmaxf(a.y, c.y); v.z = fmaxf(a.z, c.z); v.w = fmaxf(a.w, c.w);
                if ((oj0 >> 1) + pp < Wp)
                    *reinterpret_cast<float4 *>(p.out + (orow0 * Wp + (oj0 >> 1) + pp) * p.ldo + 32 * j + c4) = v;
            }
        }
    }
#undef EFGH_FETCH
#undef EFGH_PUT
#undef EFGH_STAGE
}

// ---- weight gradient ---------------------------------------------------------------------------------------------
// unit = 32 output (= gradient) pixels of one image row, taken by one wave in 8 groups of 4 (the k of
// v_mfma_f32_16x16x4_f32).  A operand: G[pixel k][n] (16 outputs of an n-tile), B operand: X[pixel k + tap][c] for column
// (t, c) = 16 qt + lane&15 (columns >= 36 are zero).
template <int NT16>
__global__ void __launch_bounds__(64 * WAVES) k_c4_wgrad(const C4Args p) {
    __shared__ float4 Xw[WAVES][3][LWMAX + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q16 = lane & 15, k4 = lane >> 4;
    const int LW = (TP - 1) * p.s + 3;
    float4 (*X)[LWMAX + 1] = Xw[wave];
    f32x4 acc[NT16][3];
#pragma unroll
    for (int a = 0; a < NT16; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[a][c][e] = 0.f;
    // LDS float offset of this lane's column in each column tile, relative to the group's first pixel
    int coff[3]; bool cok[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int col = 16 * c + q16, t = col >> 2, ch = col & 3;
        cok[c] = col < 36;
        coff[c] = cok[c] ? ((t / 3) * (LWMAX + 1) + t % 3) * 4 + ch : 0;
    }
    const float *Xf = reinterpret_cast<const float *>(&X[0][0]);

    const long long nwaves = (long long)gridDim.x * WAVES;
    long long unit = (long long)blockIdx.x * WAVES + wave;
    long long row = 0, b = 0; int i = 0, j0 = 0;
    float4 pf0, pf1, pf2, pf3;
    float ga[8][NT16], gn[8][NT16];                  // G of the current / next unit (all 8 groups in flight at once)
#define EFGH_FETCH(u)                                                                    \
    {                                                                                   \
        unit_coords(p, u, row, i, b, j0);                                               \
        pf0 = load_px(p, lane, LW, i, b, j0); pf1 = load_px(p, lane + 64, LW, i, b, j0); \
        pf2 = load_px(p, lane + 128, LW, i, b, j0); pf3 = load_px(p, lane + 192, LW, i, b, j0); \
        const float *grow = p.G + (row * p.Wo + j0) * p.ldg + q16;                       \
        _Pragma("unroll") for (int g = 0; g < 8; ++g) {                                 \
            const int pl = 4 * g + k4;                                                  \
            const bool ok = j0 + pl < p.Wo;                                             \
            _Pragma("unroll") for (int a = 0; a < NT16; ++a) gn[g][a] = ok ? grow[(long long)pl * p.ldg + 16 * a] : 0.f; \
        }                                                                               \
    }
#define EFGH_PUT(q, v)                                                                   \
    {                                                                                   \
        const int idx = lane + 64 * q;                                                  \
        if (idx < 3 * LW) X[idx / LW][idx % LW] = v;                                     \
    }
    if (unit < p.units) EFGH_FETCH(unit)
    for (; unit < p.units; unit += nwaves) {
        wave_lds_sync();                             // the previous unit's operand reads are done
        EFGH_PUT(0, pf0) EFGH_PUT(1, pf1) EFGH_PUT(2, pf2) EFGH_PUT(3, pf3)
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int a = 0; a < NT16; ++a) ga[g][a] = gn[g][a];
        wave_lds_sync();
        if (unit + nwaves < p.units) EFGH_FETCH(unit + nwaves)        // in flight during the MFMAs below
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int pl = 4 * g + k4;                               // this lane's pixel of the group
            float xb[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) xb[c] = cok[c] ? Xf[coff[c] + pl * p.s * 4] : 0.f;
#pragma unroll
            for (int a = 0; a < NT16; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[g][a], xb[c], acc[a][c], 0, 0, 0);
        }
    }
#undef EFGH_FETCH
#undef EFGH_PUT
    // acc[a][c][e]: n = 16a + 4*(lane>>4) + e, column = 16c + (lane&15).  Every wave leaves its own partial [N][36] plane (a
    // wave without units leaves zeros); efgh_c4_wgrad folds the planes in wave order: no atomics, bit-reproducible run to run
    float *plane = p.dW + ((long long)blockIdx.x * WAVES + wave) * (16 * NT16 * 36);
#pragma unroll
    for (int a = 0; a < NT16; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (!cok[c]) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                plane[(16 * a + 4 * k4 + e) * 36 + 16 * c + q16] = acc[a][c][e];
        }
}

bool c4_geometry_ok(const efgh_gemm_desc *d) {
    if (!d || d->mode != 1 || d->C != 4 || d->T != 9 || d->M_dev || d->nbatch > 1) return false;
    if (d->sh != d->sw || (d->sh != 1 && d->sh != 2) || d->osh != 1 || d->osw != 1 || d->oh0 || d->ow0) return false;
    if (d->Hv != d->Ho || d->Wv != d->Wo || d->B <= 0 || d->M != (int64_t)d->B * d->Ho * d->Wo) return false;
    for (int t = 0; t < 9; ++t) if (d->dh[t] != t / 3 - 1 || d->dw[t] != t % 3 - 1) return false;
    // every input pixel a valid output touches must exist or be padding: s*(Ho-1) - 1 + 2 <= Hin is NOT required (out-of-image
    // taps read zeros), but the LDS row must cover them: guaranteed by LW = (TP-1)*s + 3
    if (d->residual && (d->ldr % 4 != 0 || (((uintptr_t)d->residual) & 15) != 0)) return false;      // 16-byte residual reads
    return d->lda % 4 == 0 && (((uintptr_t)d->A) & 15) == 0;
}

void fill(C4Args &a, const efgh_gemm_desc *d) {
    a.A = d->A; a.lda = d->lda; a.B = d->B; a.Hin = d->Hin; a.Win = d->Win; a.Ho = d->Ho; a.Wo = d->Wo; a.s = d->sh;
    a.W = d->W; a.N = d->N; a.bias = d->bias; a.scale = d->scale; a.shift = d->shift; a.residual = d->residual; a.ldr = d->ldr; a.act = d->act; a.slope = d->slope;
    a.out = d->out; a.ldo = d->ldo; a.stats = d->stats;
    a.jblocks = (d->Wo + TP - 1) / TP;
    a.units = (long long)d->B * d->Ho * a.jblocks;
    a.G = nullptr; a.ldg = 0; a.dW = nullptr;
}

int grid_of(long long units) {                     // persistent: 3 workgroups of 4 waves per CU
    const long long g = (units + WAVES - 1) / WAVES;
    return (int)(g < 768 ? g : 768);
}

}  // namespace

extern "C" int efgh_c4_supported(const efgh_gemm_desc *d) {
    return (c4_geometry_ok(d) && (d->N == 32 || d->N == 64 || d->N == 128)) ? 1 : 0;
}

extern "C" int32_t efgh_c4_stats_rows(int32_t B, int32_t Ho, int32_t Wo) {
    return grid_of((long long)B * Ho * ((Wo + TP - 1) / TP));
}

extern "C" int efgh_c4_conv3x3(const efgh_gemm_desc *d, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(efgh_c4_supported(d) && d->W && d->out && (((uintptr_t)d->W) & 7) == 0);
    EFGH_CHECK_ARG(d->ldo % 4 == 0 && (((uintptr_t)d->out) & 15) == 0);           // 16-byte output stores
    C4Args a;
    fill(a, d);
    const int grid = grid_of(a.units);
    const bool res = d->residual != nullptr;
#define EFGH_GO(NT)                                                                  \
    {                                                                               \
        if (res) k_c4_conv<NT, true><<<grid, 64 * WAVES, 0, st>>>(a);               \
        else k_c4_conv<NT, false><<<grid, 64 * WAVES, 0, st>>>(a);                  \
    }
    if (d->N == 32) EFGH_GO(1) else if (d->N == 64) EFGH_GO(2) else EFGH_GO(4)
#undef EFGH_GO
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_c4_pooled_supported(const efgh_gemm_desc *d) {
    return (efgh_c4_supported(d) && d->sh == 1 && !d->residual && !d->stats && d->Ho >= 2 && d->Wo >= 2) ? 1 : 0;
}

extern "C" int efgh_c4_conv3x3_pooled(const efgh_gemm_desc *d, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(efgh_c4_pooled_supported(d) && d->W && d->out && (((uintptr_t)d->W) & 7) == 0);
    EFGH_CHECK_ARG(d->ldo % 4 == 0 && (((uintptr_t)d->out) & 15) == 0);
    C4Args a;
    fill(a, d);
    a.units = (long long)d->B * (d->Ho / 2) * a.jblocks;          // row pairs x 32-pixel blocks
    const int grid = grid_of(a.units);
    if (d->N == 32) k_c4_conv_pool<1><<<grid, 64 * WAVES, 0, st>>>(a);
    else if (d->N == 64) k_c4_conv_pool<2><<<grid, 64 * WAVES, 0, st>>>(a);
    else k_c4_conv_pool<4><<<grid, 64 * WAVES, 0, st>>>(a);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_c4_wgrad_supported(const efgh_gemm_desc *d) {
    return (c4_geometry_ok(d) && d->N % 16 == 0 && d->N >= 32 && d->N <= 128) ? 1 : 0;
}

/* floats of scratch efgh_c4_wgrad needs: one partial [N][36] plane per wave of the launch */
extern "C" int64_t efgh_c4_wgrad_workspace(const efgh_gemm_desc *d) {
    if (!efgh_c4_wgrad_supported(d)) return 0;
    C4Args a;
    fill(a, d);
    return (int64_t)grid_of(a.units) * WAVES * d->N * 36;
}

extern "C" int efgh_c4_wgrad(const efgh_gemm_desc *d, const float *G, int64_t ldg, float *dWp, float *workspace,
                             const efgh_wgrad_out_desc *out, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(efgh_c4_wgrad_supported(d) && G && dWp && ldg >= d->N);
    EFGH_CHECK_ARG(workspace && (((uintptr_t)workspace) & 15) == 0 && (((uintptr_t)dWp) & 15) == 0);
    C4Args a;
    fill(a, d);
    a.G = G; a.ldg = ldg; a.dW = workspace;
    const int grid = grid_of(a.units);
    switch (d->N / 16) {
    case 2: k_c4_wgrad<2><<<grid, 256, 0, st>>>(a); break;
    case 3: k_c4_wgrad<3><<<grid, 256, 0, st>>>(a); break;
    case 4: k_c4_wgrad<4><<<grid, 256, 0, st>>>(a); break;
    case 5: k_c4_wgrad<5><<<grid, 256, 0, st>>>(a); break;
    case 6: k_c4_wgrad<6><<<grid, 256, 0, st>>>(a); break;
    case 7: k_c4_wgrad<7><<<grid, 256, 0, st>>>(a); break;
    default: k_c4_wgrad<8><<<grid, 256, 0, st>>>(a); break;
    }
    const bool wrote = efgh_launch_fold_splits(workspace, grid * WAVES, (long long)d->N * 36, dWp, st, out);
    EFGH_CHECK_LAUNCH();
    return wrote ? EFGH_WROTE_OUT : EFGH_OK;
}
