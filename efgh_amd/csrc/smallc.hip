// 3x3 / stride-1 / pad-1 convolutions with 16 or 32 channels on BOTH sides (and two 1x1 shapes, below) on fp32 MFMA (gfx950): the up-sampling stages of F's two
// trunks (nets/fnet.py:22-31 through nets/net_utils.py:66-98: conv_bn_relu after every convt_bn_relu), forward, data gradient (the
// same kernel on the gradient with the transposed, tap-reversed weights) and weight gradient.
//
// These layers move 128-256 bytes per pixel for 2 x 9 x C x N <= 18 k FLOP: 17.7 GFLOP over 0.5 GB at 3.85 M pixels, i.e. 0.11 ms
// at the fp32-MFMA rate and 0.10 ms at the HBM rate.  The generic implicit-GEMM tile (128 x 32 x 32, K in blocks of 32 of which the
// last is mostly empty, one 4-byte store per lane and element) ran them at 12-33 TFLOP/s: 0.58 ms forward / data gradient and
// 1.41 ms for the weight gradient of the 16-channel layer.  Here, as in c4conv.hip, one WAVE owns 32 consecutive output pixels of an
// image row and there is no workgroup barrier in the loop:
//   * the three input rows (34 pixels each) go to the wave's LDS slice once, as one float4 per (row, channel quad, pixel) - the
//     16-byte global loads are contiguous along the row, the LDS image is quad-major so that the 16 pixels of an MFMA operand are
//     consecutive float4s (each bank is hit exactly twice by the 64 lanes: the optimum for a 4-byte read);
//   * v_mfma_f32_16x16x4_f32: 16 pixels x 16 outputs, k = the 4 channels of a quad; 9 C / 4 steps per tile, the weights of all
//     steps live in registers of the persistent wave;
//   * epilogue of k_gather_gemm (bias, BatchNorm statistics of the pre-activation value, scale / shift, residual, activation)
//     through a wave-private LDS tile, leaving as 16-byte stores.
// Weight gradient: dW[n][t][c] = sum_p G[p][n] X[p + tap t][c] with k = 4 pixels per MFMA, G and X staged the same way, one
// partial [N][9 C] plane per wave, planes added in a fixed order by k_fold_splits (no atomics: bit-reproducible).  The same
// kernel serves the 4-channel INPUT layers at stride 1 (RGB / range / depth -> 32 or 64 channels: 36 columns, the last tile
// masked) in place of k_c4_wgrad, which loads G four bytes at a time and combines its partial sums with fp32 atomics.
// TAPS == 1: the 1x1 / stride-1 layers with 32 or 64 channels (64 -> 32 at full resolution and its data gradient 32 -> 64): one
// staged row without a halo, 1.5 GB of operands for 16 GFLOP - HBM-bound at any MFMA rate, so what counts is that every byte moves
// once, as 16-byte accesses, with the next unit's loads in flight during the MFMAs.
#include "common.h"


namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TP = 32;                  // output pixels per unit (one wave)
constexpr int lw_of(int taps) { return taps == 9 ? TP + 2 : TP + 1; } // staged row pitch in pixels (3x3: one halo pixel either side; 1x1: one pad
                                                                      // pixel, which keeps the 16 channel quads of a pixel on different banks)
constexpr int rows_of(int taps) { return taps == 9 ? 3 : 1; }         // staged input rows
constexpr int WAVES = 4;

struct SCArgs {
    const float *A; long long lda;
    int B, H, W;
    const float *Wp; int N;             // packed [N][9][C]
    const float *bias, *scale, *shift, *residual; long long ldr;
    int act; float slope;
    float *out; long long ldo;
    float *stats;
    long long units; int jblocks;
    const float *G; long long ldg; float *part;      // wgrad
};

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void unit_coords(const SCArgs &p, long long unit, long long &row, int &i, long long &b, int &j0) {
    const int jb = (int)(unit % p.jblocks);
    row = unit / p.jblocks;                          // b * H + i
    i = (int)(row % p.H); b = row / p.H;
    j0 = jb * TP;
}

// staged float4 number idx of a unit: idx = (kh * LW + x) * NG + g  (g fastest: contiguous 16-byte chunks along the image row)
template <int NG, int TAPS>
__device__ __forceinline__ float4 load_quad(const SCArgs &p, int idx, int i, long long b, int j0) {
    constexpr int LW = lw_of(TAPS), HALO = TAPS == 9 ? 1 : 0;
    const int g = idx % NG, r = idx / NG;
    const int kh = r / LW, x = r - kh * LW;
    const int yin = i - HALO + kh, xin = j0 - HALO + x;
    if (kh < rows_of(TAPS) && (unsigned)yin < (unsigned)p.H && (unsigned)xin < (unsigned)p.W)
        return *reinterpret_cast<const float4 *>(p.A + ((b * p.H + yin) * p.W + xin) * p.lda + 4 * g);
    return make_float4(0.f, 0.f, 0.f, 0.f);
}

template <int C, int N, int TAPS, bool RES>
__global__ void __launch_bounds__(64 * WAVES, 2) k_sc_conv(const SCArgs p) {
    constexpr int LW = lw_of(TAPS), ROWS = rows_of(TAPS);
    constexpr int NG = C / 4, NT = N / 16, STEPS = TAPS * NG;
    constexpr int NQ = ROWS * LW * NG, NPF = (NQ + 63) / 64;       // float4s per unit, per lane
    constexpr int TPITCH = N + 4;                                   // floats per pixel row of the transpose tile
    __shared__ __attribute__((aligned(16))) float4 Qw[WAVES][ROWS * NG * LW];
    __shared__ __attribute__((aligned(16))) float Tw[WAVES][TP * TPITCH];
    __shared__ float red[2][WAVES][N];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q16 = lane & 15, kq = lane >> 4;
    float4 *Q = Qw[wave];
    const float *Qf = reinterpret_cast<const float *>(Q);
    float *T = Tw[wave];

    // B operand of step s = (tap t, quad g): W[16 nt + q16][t][4 g + kq]
    float bw[STEPS][NT];
    float bi[NT], sc[NT], sf[NT], s1[NT], s2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = 16 * nt + q16;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) bw[s][nt] = p.Wp[((long long)n * TAPS + s / NG) * C + 4 * (s % NG) + kq];
        bi[nt] = p.bias ? p.bias[n] : 0.f;
        sc[nt] = p.scale ? p.scale[n] : 1.f;
        sf[nt] = p.shift ? p.shift[n] : 0.f;
        s1[nt] = 0.f; s2[nt] = 0.f;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < STEPS; ++s) asm volatile("" ::"v"(bw[s][nt]));      // (weight loads complete before the loop, see c4conv.hip)
    const float neg = p.act == 1 ? 0.f : (p.act == 2 ? p.slope : 1.f);

    const long long nwaves = (long long)gridDim.x * WAVES;
    long long unit = (long long)blockIdx.x * WAVES + wave;
    long long row = 0, b = 0; int i = 0, j0 = 0;     // coordinates of the unit whose quads are in pf[]
    long long crow = 0; int cj0 = 0;                 // ... of the unit staged in LDS
    float4 pf[NPF];
    auto fetch = [&](long long u) {
        unit_coords(p, u, row, i, b, j0);
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int idx = lane + 64 * q;
            pf[q] = idx < NQ ? load_quad<NG, TAPS>(p, idx, i, b, j0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage = [&]() {
        crow = row; cj0 = j0;
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int idx = lane + 64 * q;
            if (idx < NQ) {
                const int g = idx % NG, r = idx / NG;          // r = kh * LW + x
                const int kh = r / LW, x = r - kh * LW;
                Q[(kh * NG + g) * LW + x] = pf[q];
            }
        }
    };
    // C = 16 and the 1x1 layers: the next unit's quads are prefetched into registers during the MFMAs (7-8 float4s); 3x3 with C = 32
    // has no registers left for that (144 of them hold the weights): its waves load, stage and compute in turn and overlap with each other
    constexpr bool PRE = C == 16 || TAPS == 1;
    if (PRE && unit < p.units) {
        fetch(unit);
        stage();
        if (unit + nwaves < p.units) fetch(unit + nwaves);
    }
    for (; unit < p.units; unit += nwaves) {
        if (!PRE) {
            wave_lds_sync();                         // (the previous unit's operand reads are done)
            unit_coords(p, unit, row, i, b, j0);
            crow = row; cj0 = j0;
#pragma unroll
            for (int q0 = 0; q0 < NPF; q0 += 4) {    // four quads at a time: the loads of a group in flight together, few registers
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = lane + 64 * (q0 + u);
                    v[u] = (q0 + u < NPF && idx < NQ) ? load_quad<NG, TAPS>(p, idx, i, b, j0) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = lane + 64 * (q0 + u);
                    if (q0 + u < NPF && idx < NQ) {
                        const int g = idx % NG, r = idx / NG;
                        const int kh = r / LW, x = r - kh * LW;
                        Q[(kh * NG + g) * LW + x] = v[u];
                    }
                }
            }
        }
        const long long orow0 = crow; const int oj0 = cj0;
        wave_lds_sync();
        f32x4 acc[2][NT];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[pt][nt][e] = 0.f;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int t = s / NG, g = s % NG, kh = TAPS == 9 ? t / 3 : 0, kw = TAPS == 9 ? t % 3 : 0;
            // A operand: pixel 16 pt + q16 (+ kw), channel 4 g + kq
            const float a0 = Qf[((kh * NG + g) * LW + q16 + kw) * 4 + kq];
            const float a1 = Qf[((kh * NG + g) * LW + 16 + q16 + kw) * 4 + kq];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bw[s][nt], acc[0][nt], 0, 0, 0);
                acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bw[s][nt], acc[1][nt], 0, 0, 0);
            }
        }
        wave_lds_sync();
        if (PRE && unit + nwaves < p.units) {
            stage();
            if (unit + 2 * nwaves < p.units) fetch(unit + 2 * nwaves);
        }
        // epilogue.  acc[pt][nt][e]: pixel 16 pt + 4 kq + e, output 16 nt + q16 -> wave-private tile [pixel][output] -> 16-byte stores
        const bool full = oj0 + TP <= p.W;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int pl = 16 * pt + 4 * kq + e;
                    float v = acc[pt][nt][e] + bi[nt];
                    if (full || oj0 + pl < p.W) { s1[nt] += v; s2[nt] = fmaf(v, v, s2[nt]); }
                    v = fmaf(v, sc[nt], sf[nt]);
                    T[pl * TPITCH + 16 * nt + q16] = RES ? v : act_neg(v, neg);
                }
        wave_lds_sync();
        constexpr int F4 = N / 4;                               // float4s per pixel row
#pragma unroll
        for (int pass = 0; pass < TP * F4 / 64; ++pass) {
            const int idx = lane + 64 * pass, pl = idx / F4, c4 = (idx % F4) * 4;
            if (!(full || oj0 + pl < p.W)) continue;
            float4 v = *reinterpret_cast<const float4 *>(&T[pl * TPITCH + c4]);
            if (RES) {
                const float4 rr = *reinterpret_cast<const float4 *>(p.residual + (orow0 * p.W + oj0 + pl) * p.ldr + c4);
                v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                v.x = act_neg(v.x, neg); v.y = act_neg(v.y, neg);
                v.z = act_neg(v.z, neg); v.w = act_neg(v.w, neg);
            }
            *reinterpret_cast<float4 *>(p.out + (orow0 * p.W + oj0 + pl) * p.ldo + c4) = v;
        }
    }
    if (p.stats) {                                   // one statistics row per workgroup
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            s1[nt] += __shfl_xor(s1[nt], 16); s2[nt] += __shfl_xor(s2[nt], 16);
            s1[nt] += __shfl_xor(s1[nt], 32); s2[nt] += __shfl_xor(s2[nt], 32);
            if (kq == 0) { red[0][wave][16 * nt + q16] = s1[nt]; red[1][wave][16 * nt + q16] = s2[nt]; }
        }
        __syncthreads();
        for (int n = tid; n < N; n += 64 * WAVES) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) { a1 += red[0][w][n]; a2 += red[1][w][n]; }
            p.stats[((long long)blockIdx.x * 2 + 0) * p.N + n] = a1;
            p.stats[((long long)blockIdx.x * 2 + 1) * p.N + n] = a2;
        }
    }
}

// ---- weight gradient: unit = 32 gradient pixels of one image row, taken by one wave in 8 groups of 4 pixels (the k of
// v_mfma_f32_16x16x4_f32).  A operand: G[pixel 4 grp + kq][16 nt + q16]; B operand: X[pixel + tap t][c] for column (t, c) = 16 ct + q16
// of the [9][C] plane.  Accumulators: NT x (9 C / 16) tiles, kept over all units of the (persistent) wave.
// CS column splits (blockIdx.y): with 32 input channels the 18 column tiles do not fit one wave's registers next to the operands;
// every split walks all units and keeps its half of the columns (the 125-MB operands of these layers are read twice)
template <int C, int N, int TAPS, int CS>
__global__ void __launch_bounds__(64 * WAVES, 2) k_sc_wgrad(const SCArgs p) {
    constexpr int LW = lw_of(TAPS), ROWS = rows_of(TAPS);
    constexpr int NG = C / 4, NT = N / 16, CT = (TAPS * C + 15) / 16 / CS;     // (C = 4: 36 columns = 2.25 -> 3 tiles, the rest masked)
    const int ct0 = blockIdx.y * CT;
    constexpr int NQ = ROWS * LW * NG, NPF = (NQ + 63) / 64;
    __shared__ __attribute__((aligned(16))) float4 Qw[WAVES][ROWS * NG * LW];
    __shared__ float Gw[WAVES][TP][N + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q16 = lane & 15, kq = lane >> 4;
    float4 *Q = Qw[wave];
    const float *Qf = reinterpret_cast<const float *>(Q);
    float (*Gs)[N + 1] = Gw[wave];
    f32x4 acc[NT][CT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[a][c][e] = 0.f;
    // LDS float offset of this lane's column (t, c) in each column tile, relative to the group's first pixel
    constexpr bool MASK = (TAPS * C) % 16 != 0;                           // (only C = 4 has a ragged last column tile)
    int coff[CT]; bool cok[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int col = 16 * (ct0 + c) + q16, t = col / C, ch = col % C;
        cok[c] = !MASK || col < TAPS * C;
        coff[c] = cok[c] ? (((TAPS == 9 ? t / 3 : 0) * NG + ch / 4) * LW + (TAPS == 9 ? t % 3 : 0)) * 4 + (ch & 3) : 0;
    }
    const long long nwaves = (long long)gridDim.x * WAVES;
    long long unit = (long long)blockIdx.x * WAVES + wave;
    long long row = 0, b = 0; int i = 0, j0 = 0;
    float4 pf[NPF];
    constexpr int NGF = TP * N / 4 / 64;                               // float4s of G per lane and unit
    float4 gf[NGF];
    auto fetch = [&](long long u) {
        unit_coords(p, u, row, i, b, j0);
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int idx = lane + 64 * q;
            pf[q] = idx < NQ ? load_quad<NG, TAPS>(p, idx, i, b, j0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < NGF; ++q) {
            const int idx = lane + 64 * q, pl = idx / (N / 4), c4 = (idx % (N / 4)) * 4;
            gf[q] = (j0 + pl < p.W) ? *reinterpret_cast<const float4 *>(p.G + (row * p.W + j0 + pl) * p.ldg + c4)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if (unit < p.units) fetch(unit);
    for (; unit < p.units; unit += nwaves) {
        wave_lds_sync();                             // the previous unit's operand reads are done
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int idx = lane + 64 * q;
            if (idx < NQ) {
                const int g = idx % NG, r = idx / NG;
                const int kh = r / LW, x = r - kh * LW;
                Q[(kh * NG + g) * LW + x] = pf[q];
            }
        }
#pragma unroll
        for (int q = 0; q < NGF; ++q) {
            const int idx = lane + 64 * q, pl = idx / (N / 4), c4 = (idx % (N / 4)) * 4;
            Gs[pl][c4] = gf[q].x; Gs[pl][c4 + 1] = gf[q].y; Gs[pl][c4 + 2] = gf[q].z; Gs[pl][c4 + 3] = gf[q].w;
        }
        wave_lds_sync();
        if (unit + nwaves < p.units) fetch(unit + nwaves);            // in flight during the MFMAs below
#pragma unroll
        for (int grp = 0; grp < TP / 4; ++grp) {
            const int pl = 4 * grp + kq;                             // this lane's pixel of the group
            float ga[NT];
#pragma unroll
            for (int a = 0; a < NT; ++a) ga[a] = Gs[pl][16 * a + q16];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float xb = (!MASK || cok[c]) ? Qf[coff[c] + pl * 4] : 0.f;
#pragma unroll
                for (int a = 0; a < NT; ++a) acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[a], xb, acc[a][c], 0, 0, 0);
            }
        }
    }
    // acc[a][c][e]: n = 16 a + 4 kq + e, column = 16 c + q16.  One partial plane per WAVE (plain stores), folded in a fixed order.
    float *plane = p.part + ((long long)blockIdx.x * WAVES + wave) * (N * TAPS * C);
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            if (MASK && !cok[c]) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) plane[(16 * a + 4 * kq + e) * (TAPS * C) + 16 * (ct0 + c) + q16] = acc[a][c][e];
        }
}

bool sc_geometry_ok(const efgh_gemm_desc *d, bool four_in = false) {
    if (!d || d->mode != 1 || (d->T != 9 && d->T != 1) || d->M_dev || d->nbatch > 1) return false;
    if (d->T == 1) { if (four_in || !((d->C == 64 && d->N == 32) || (d->C == 32 && d->N == 64))) return false; }
    else if (four_in) { if (!(d->C == 4 && (d->N == 32 || d->N == 64))) return false; }
    else if (!((d->C == 16 || d->C == 32) && (d->N == 16 || d->N == 32))) return false;
    if (d->sh != 1 || d->sw != 1 || d->osh != 1 || d->osw != 1 || d->oh0 || d->ow0) return false;
    if (d->Hv != d->Ho || d->Wv != d->Wo || d->Ho != d->Hin || d->Wo != d->Win || d->B <= 0) return false;
    if (d->M != (int64_t)d->B * d->Ho * d->Wo) return false;
    if (d->T == 1) { if (d->dh[0] || d->dw[0]) return false; }
    else for (int t = 0; t < 9; ++t) if (d->dh[t] != t / 3 - 1 || d->dw[t] != t % 3 - 1) return false;
    if (d->residual && (d->ldr % 4 != 0 || (((uintptr_t)d->residual) & 15) != 0)) return false;
    return d->lda % 4 == 0 && (((uintptr_t)d->A) & 15) == 0;
}

void fill(SCArgs &a, const efgh_gemm_desc *d) {
    a.A = d->A; a.lda = d->lda; a.B = d->B; a.H = d->Hin; a.W = d->Win;
    a.Wp = d->W; a.N = d->N; a.bias = d->bias; a.scale = d->scale; a.shift = d->shift; a.residual = d->residual; a.ldr = d->ldr;
    a.act = d->act; a.slope = d->slope; a.out = d->out; a.ldo = d->ldo; a.stats = d->stats;
    a.jblocks = (d->Win + TP - 1) / TP;
    a.units = (long long)d->B * d->Hin * a.jblocks;
    a.G = nullptr; a.ldg = 0; a.part = nullptr;
}

int grid_of(long long units) {                     // persistent: 2 workgroups of 4 waves per CU
    const long long g = (units + WAVES - 1) / WAVES;
    return (int)(g < 512 ? g : 512);
}

}  // namespace

extern "C" int efgh_sc_supported(const efgh_gemm_desc *d) { return sc_geometry_ok(d) ? 1 : 0; }

extern "C" int32_t efgh_sc_stats_rows(int32_t B, int32_t H, int32_t W) {
    return grid_of((long long)B * H * ((W + TP - 1) / TP));
}

extern "C" int efgh_sc_conv3x3(const efgh_gemm_desc *d, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(efgh_sc_supported(d) && d->W && d->out);
    EFGH_CHECK_ARG(d->ldo % 4 == 0 && (((uintptr_t)d->out) & 15) == 0);           // 16-byte output stores
    SCArgs a;
    fill(a, d);
    const int grid = grid_of(a.units);
    const bool res = d->residual != nullptr;
#define EFGH_GO(C_, N_, T_)                                                          \
    {                                                                               \
        if (res) k_sc_conv<C_, N_, T_, true><<<grid, 64 * WAVES, 0, st>>>(a);       \
        else k_sc_conv<C_, N_, T_, false><<<grid, 64 * WAVES, 0, st>>>(a);          \
    }
    if (d->T == 1 && d->C == 64) EFGH_GO(64, 32, 1)
    else if (d->T == 1) EFGH_GO(32, 64, 1)
    else if (d->C == 16 && d->N == 16) EFGH_GO(16, 16, 9)
    else if (d->C == 16 && d->N == 32) EFGH_GO(16, 32, 9)
    else if (d->C == 32 && d->N == 16) EFGH_GO(32, 16, 9)
    else EFGH_GO(32, 32, 9)
#undef EFGH_GO
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

/* the weight gradient also serves the 4-channel input layers at stride 1 (32 or 64 outputs): the same staging, 36 columns */
extern "C" int efgh_sc_wgrad_supported(const efgh_gemm_desc *d) { return (sc_geometry_ok(d) || sc_geometry_ok(d, true)) ? 1 : 0; }

/* floats of scratch efgh_sc_wgrad needs: one [N][T][C] partial per wave of the launch */
extern "C" int64_t efgh_sc_wgrad_workspace(const efgh_gemm_desc *d) {
    if (!efgh_sc_wgrad_supported(d)) return 0;
    SCArgs a;
    fill(a, d);
    return (int64_t)grid_of(a.units) * WAVES * d->N * d->T * d->C;
}

extern "C" int efgh_sc_wgrad(const efgh_gemm_desc *d, const float *G, int64_t ldg, float *dWp, float *workspace,
                             const efgh_wgrad_out_desc *out, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(efgh_sc_wgrad_supported(d) && G && dWp && workspace && ldg >= d->N && ldg % 4 == 0 && (((uintptr_t)G) & 15) == 0);
    SCArgs a;
    fill(a, d);
    a.G = G; a.ldg = ldg; a.part = workspace;
    const int grid = grid_of(a.units);
    if (d->T == 1 && d->C == 64) k_sc_wgrad<64, 32, 1, 1><<<grid, 64 * WAVES, 0, st>>>(a);
    else if (d->T == 1) k_sc_wgrad<32, 64, 1, 1><<<grid, 64 * WAVES, 0, st>>>(a);
    else if (d->C == 4 && d->N == 32) k_sc_wgrad<4, 32, 9, 1><<<grid, 64 * WAVES, 0, st>>>(a);
    else if (d->C == 4) k_sc_wgrad<4, 64, 9, 1><<<grid, 64 * WAVES, 0, st>>>(a);
    else if (d->C == 16 && d->N == 16) k_sc_wgrad<16, 16, 9, 1><<<grid, 64 * WAVES, 0, st>>>(a);
    else if (d->C == 16 && d->N == 32) k_sc_wgrad<16, 32, 9, 1><<<grid, 64 * WAVES, 0, st>>>(a);
    else if (d->C == 32 && d->N == 16) k_sc_wgrad<32, 16, 9, 2><<<dim3(grid, 2), 64 * WAVES, 0, st>>>(a);
    else k_sc_wgrad<32, 32, 9, 2><<<dim3(grid, 2), 64 * WAVES, 0, st>>>(a);
    // waves without a unit never ran: their planes are garbage - only the planes of waves that had work are folded
    const long long nw = (long long)grid * WAVES;
    const long long used = a.units < nw ? a.units : nw;
    const bool wrote = efgh_launch_fold_splits(workspace, (int)used, (long long)d->N * d->T * d->C, dWp, st, out);
    EFGH_CHECK_LAUNCH();
    return wrote ? EFGH_WROTE_OUT : EFGH_OK;
}
