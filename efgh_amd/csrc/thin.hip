// "Thin" layers: convolutions whose channel count on one side is <= 4 (the RGB / range / depth input
// convs, the 1- and 2-channel depth/mask heads of G and their transposed convs, and the matching
// dgrad / wgrad launches).  They are HBM-bound (a few FLOP per byte), so they run on the VALU with
// 16-B accesses instead of wasting 7/8 of an MFMA tile:
//   k_thin_c4  : C == 4 input channels per tap, any N   (thread = one output row x one quad of n)
//   k_thin_n4  : N <= 4 outputs, any C                  (8 lanes split K of one output row)
//   k_thin_c4_wgrad / k_thin_n4_wgrad : the corresponding weight gradients
// Same descriptor, gather modes (conv geometry incl. stride-2 / output-parity classes) and epilogue
// (bias, scale/shift, residual, activation) as k_gather_gemm; packed weights [N][T][C].
#include "common.h"


namespace {
constexpr int TPB = 256;

struct TArgs {
    const float *A; int64_t lda;
    int C, T, K;
    int Hin, Win, Hv, Wv, sh, sw;
    int dh[16], dw[16];
    int Ho, Wo, osh, osw, oh0, ow0;
    const float *W; int N;
    long long M;
    const float *bias, *scale, *shift, *residual; int64_t ldr;
    int act; float slope;
    float *out; int64_t ldo;
    const float *G; int64_t ldg; float *dW; int mchunk;     // wgrad
};

__device__ __forceinline__ float act_f(float v, int act, float slope) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v > 0.f ? v : v * slope;
    return v;
}

__device__ __forceinline__ void decode(const TArgs &p, long long m, int &i, int &j, long long &b) {
    j = (int)(m % p.Wv); long long r = m / p.Wv;
    i = (int)(r % p.Hv); b = r / p.Hv;
}
__device__ __forceinline__ long long out_row(const TArgs &p, int i, int j, long long b) {
    return (b * p.Ho + (i * p.osh + p.oh0)) * p.Wo + (j * p.osw + p.ow0);
}
__device__ __forceinline__ long long in_row(const TArgs &p, int i, int j, long long b, int t) {
    int ih = i * p.sh + p.dh[t], iw = j * p.sw + p.dw[t];
    if ((unsigned)ih >= (unsigned)p.Hin || (unsigned)iw >= (unsigned)p.Win) return -1;
    return (b * p.Hin + ih) * p.Win + iw;
}

__device__ __forceinline__ void epilogue_store(const TArgs &p, long long orow, int n, float4 v) {
    float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float x = o[q] + (p.bias ? p.bias[n + q] : 0.f);
        x = x * (p.scale ? p.scale[n + q] : 1.f) + (p.shift ? p.shift[n + q] : 0.f);
        if (p.residual) x += p.residual[orow * p.ldr + n + q];
        o[q] = act_f(x, p.act, p.slope);
    }
    *reinterpret_cast<float4 *>(p.out + orow * p.ldo + n) = make_float4(o[0], o[1], o[2], o[3]);
}

// ---- C == 4: one thread = one output row, ALL n (in passes of 64 channels); the <=9 input taps live
// in registers, weights are wave-uniform LDS broadcasts ([t][n][4]); each 32-channel slab of the result is
// transposed through a padded LDS stage so that global stores are full 128-B row segments.
template <int T>
__global__ void __launch_bounds__(TPB)
k_thin_c4(const TArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *wl = smem;                                    // T*N*4 floats
    float *stage = smem + (size_t)T * p.N * 4;           // 4 waves x 64 rows x 36 floats
    for (int i = threadIdx.x; i < T * p.N; i += TPB) {
        int t = i / p.N, n = i - t * p.N;
        reinterpret_cast<float4 *>(wl)[i] = *reinterpret_cast<const float4 *>(p.W + ((long long)n * T + t) * 4);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *st = stage + wave * 64 * 36;
    const long long nwaves = (long long)gridDim.x * (TPB / 64);
    for (long long w0 = ((long long)blockIdx.x * (TPB / 64) + wave) * 64; w0 < p.M; w0 += nwaves * 64) {
        const long long m = w0 + lane;
        const bool ok = m < p.M;
        int i = 0, j = 0; long long b = 0;
        if (ok) decode(p, m, i, j, b);
        float4 a[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            long long r = ok ? in_row(p, i, j, b, t) : -1;
            a[t] = r >= 0 ? *reinterpret_cast<const float4 *>(p.A + r * p.lda) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const long long orow = ok ? out_row(p, i, j, b) : 0;
        for (int nb = 0; nb < p.N; nb += 32) {
            const int nq = (p.N - nb) < 32 ? (p.N - nb) >> 2 : 8;      // n-quads in this pass
            for (int q = 0; q < nq; ++q) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float4 *w = reinterpret_cast<const float4 *>(wl) + t * p.N + nb + q * 4;
                    float4 w0_ = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
                    acc.x += a[t].x * w0_.x + a[t].y * w0_.y + a[t].z * w0_.z + a[t].w * w0_.w;
                    acc.y += a[t].x * w1.x + a[t].y * w1.y + a[t].z * w1.z + a[t].w * w1.w;
                    acc.z += a[t].x * w2.x + a[t].y * w2.y + a[t].z * w2.z + a[t].w * w2.w;
                    acc.w += a[t].x * w3.x + a[t].y * w3.y + a[t].z * w3.z + a[t].w * w3.w;
                }
                if (nq == 1) { if (ok) epilogue_store(p, orow, nb, acc); }
                else *reinterpret_cast<float4 *>(&st[lane * 36 + q * 4]) = acc;
            }
            if (nq > 1) {
                // cooperative, coalesced store: slot s -> (row = s / nq, quad = s % nq)
                const int slots = 64 * nq;
                for (int s = lane; s < slots; s += 64) {
                    int rr = s / nq, qq = s - rr * nq;
                    long long mm = w0 + rr;
                    long long orr = __shfl(orow, rr);          // all 64 lanes active here
                    if (mm >= p.M) continue;
                    float4 v = *reinterpret_cast<const float4 *>(&st[rr * 36 + qq * 4]);
                    epilogue_store(p, orr, nb + qq * 4, v);
                }
            }
        }
    }
}

// ---- N == 4: 8 lanes per output row split K; weights in LDS as [n][K] ------------------------------
__global__ void __launch_bounds__(TPB)
k_thin_n4(const TArgs p) {
    extern __shared__ __attribute__((aligned(16))) float wl[];          // 4*K floats
    for (int i = threadIdx.x; i < p.K; i += TPB)
        reinterpret_cast<float4 *>(wl)[i] = *reinterpret_cast<const float4 *>(p.W + (long long)i * 4);
    __syncthreads();
    const int l8 = threadIdx.x & 7;
    const int c4n = p.C >> 2;
    for (long long m = (long long)blockIdx.x * (TPB / 8) + (threadIdx.x >> 3); m < p.M;
         m += (long long)gridDim.x * (TPB / 8)) {
        int i, j; long long b;
        decode(p, m, i, j, b);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = 0; t < p.T; ++t) {
            long long r = in_row(p, i, j, b, t);
            if (r < 0) continue;
            const float4 *arow = reinterpret_cast<const float4 *>(p.A + r * p.lda);
            const float4 *w0 = reinterpret_cast<const float4 *>(wl + t * p.C);
            const float4 *w1 = reinterpret_cast<const float4 *>(wl + p.K + t * p.C);
            const float4 *w2 = reinterpret_cast<const float4 *>(wl + 2 * p.K + t * p.C);
            const float4 *w3 = reinterpret_cast<const float4 *>(wl + 3 * p.K + t * p.C);
            for (int c = l8; c < c4n; c += 8) {
                float4 a = arow[c], x0 = w0[c], x1 = w1[c], x2 = w2[c], x3 = w3[c];
                acc.x += a.x * x0.x + a.y * x0.y + a.z * x0.z + a.w * x0.w;
                acc.y += a.x * x1.x + a.y * x1.y + a.z * x1.z + a.w * x1.w;
                acc.z += a.x * x2.x + a.y * x2.y + a.z * x2.z + a.w * x2.w;
                acc.w += a.x * x3.x + a.y * x3.y + a.z * x3.z + a.w * x3.w;
            }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            acc.x += __shfl_xor(acc.x, o); acc.y += __shfl_xor(acc.y, o);
            acc.z += __shfl_xor(acc.z, o); acc.w += __shfl_xor(acc.w, o);
        }
        if (l8 == 0) epilogue_store(p, out_row(p, i, j, b), 0, acc);
    }
}

// ---- N == 4, C == 64, 3x3 / stride 1 / pad 1 on fp32 MFMA: the data gradient of the 4 -> 64 input layer of F's range trunk (15.7 M
// pixels at batch 8: 4 GB of gradient for 72 GFLOP).  k_thin_n4 fetches every 256-byte gradient row nine times through L1 (one per
// tap) and runs its 36 G FMAs on the VALU: 3.0 ms.  Here one WAVE owns a strip of 30 output columns and walks down RC image rows; for
// every input row it computes the 36 products P[pixel][(kh, kw, n)] = x[pixel][:] . W[n][kh][kw][:] of 32 pixels (30 + one halo pixel
// either side) with v_mfma_f32_16x16x4_f32 - columns (kw, n) = 12 of 16 per kernel row kh - and then adds them into the three output
// rows they belong to (y[i][j] = sum P_{i + kh - 1}[j + kw - 1][kh][kw]): the column shift goes through a wave-private LDS tile, the
// row shift is a rotation of two partial-row registers.  Every gradient row is read once (+ 7 % halo columns, + 2 / RC halo rows).
// The contraction axis is permuted so that a lane's MFMA operands are four contiguous float4s of its pixel (lane group kq owns
// channels 16 kq .. 16 kq + 15).
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int N4_TP = 30, N4_RC = 32, N4_WAVES = 4, N4_QP = 33;       // strip width, rows per unit, waves per block, LDS pitch
// (units are NOT walked by persistent waves: 8 k - 16 k units over 2048 wave slots leave a tail of a whole unit; one wave per unit,
// dispatched as slots free up, ends within a fraction of one)

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ void __launch_bounds__(64 * N4_WAVES, 3)
k_n4_conv3x3_c64(const TArgs p, int strips, int chunks, long long units) {
    constexpr int C = 64, NG = C / 4;
    // per wave: the staged row Q[channel quad][pixel] (float4) and, in the same bytes once its operands are in registers, the product
    // tile T[kh][column][pixel] (pitch N4_TPI: a lane's four consecutive pixels leave as one 16-byte write)
    constexpr int N4_TPI = 36;
    static_assert(3 * 16 * N4_TPI <= NG * N4_QP * 4, "the product tile must fit the staged row's bytes");
    __shared__ __attribute__((aligned(16))) float4 Qw[N4_WAVES][NG * N4_QP];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), q16 = lane & 15, kq = lane >> 4;
    float4 *Q = Qw[wave];
    float *T = reinterpret_cast<float *>(Qw[wave]);
    // B operand of (kh, step s): column q16 = kw * 4 + n (12 used), channel 16 kq + s
    float bw[3][16];
    {
        const int kw = q16 >> 2, n = q16 & 3;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kw < 3) w = *reinterpret_cast<const float4 *>(p.W + ((long long)n * 9 + kh * 3 + kw) * C + 16 * kq + 4 * s4);
                bw[kh][4 * s4] = w.x; bw[kh][4 * s4 + 1] = w.y; bw[kh][4 * s4 + 2] = w.z; bw[kh][4 * s4 + 3] = w.w;
            }
    }
    // this lane's two outputs of a row: pixel jj of the strip, channels n0, n0 + 1
    const int jj = lane & 31, n0 = (lane >> 5) * 2;
    float ebi[2], esc[2], esf[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        ebi[u] = p.bias ? p.bias[n0 + u] : 0.f;
        esc[u] = p.scale ? p.scale[n0 + u] : 1.f;
        esf[u] = p.shift ? p.shift[n0 + u] : 0.f;
    }
    const int H = p.Hin, W = p.Win;
    for (long long unit = (long long)blockIdx.x * N4_WAVES + wave; unit < units; unit += (long long)gridDim.x * N4_WAVES) {
        const int js = (int)(unit % strips);
        const long long r1 = unit / strips;
        const int rc = (int)(r1 % chunks);
        const long long b = r1 / chunks;
        const int j0 = js * N4_TP, i0 = rc * N4_RC;
        const int iend = i0 + N4_RC < H ? i0 + N4_RC : H;            // output rows [i0, iend)
        // one buffer resource per input row: pixels left / right of the image fall outside its num_records and rows above / below
        // it get an empty one - the hardware's range check returns zeros, so there is no select or branch around the loads
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 pf[8];
        const int voff = (((j0 - 1 + (lane >> 4)) * (int)p.lda) + 4 * (lane & 15)) * 4;      // bytes; lane's pixel of load 0, its quad
        const int vstep = 4 * (int)p.lda * 4;                                               // four pixels on per load
        const unsigned rowbytes = (unsigned)W * (unsigned)p.lda * 4u;
        auto fetch = [&](int r) {                                    // input row r, pixels j0 - 1 .. j0 + 30
            const bool rok = (unsigned)r < (unsigned)H;
            const float *rowp = p.A + (b * H + (rok ? r : 0)) * W * p.lda;
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(rowp), 0, rok ? rowbytes : 0u, 0x00020000);
#pragma unroll
            for (int q = 0; q < 8; ++q) pf[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + q * vstep, 0, 0);
        };
        float R0[2] = {0.f, 0.f}, R1[2] = {0.f, 0.f};               // partial sums of output rows r and r - 1
        fetch(i0 - 1);
        for (int r = i0 - 1; r <= iend; ++r) {                       // input rows; output row r - 1 completes with input row r
            const bool rok = (unsigned)r < (unsigned)H;              // (wave-uniform)
            float S[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
            if (rok) {
                wave_lds_sync();                                     // the previous row's operand / tile reads are done
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int idx = lane + 64 * q, g = idx & 15, px = idx >> 4;
                    Q[g * N4_QP + px] = make_float4(__uint_as_float(pf[q].x), __uint_as_float(pf[q].y), __uint_as_float(pf[q].z),
                                                    __uint_as_float(pf[q].w));
                }
                wave_lds_sync();
            }
            if (r < iend) fetch(r + 1);                              // in flight during the MFMAs
            if (rok) {
                f32x4 acc[2][3];
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[pt][kh][e] = 0.f;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const float4 a0 = Q[(4 * kq + m) * N4_QP + q16], a1 = Q[(4 * kq + m) * N4_QP + 16 + q16];
                    const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int kh = 0; kh < 3; ++kh) {
                            acc[0][kh] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[e], bw[kh][4 * m + e], acc[0][kh], 0, 0, 0);
                            acc[1][kh] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[e], bw[kh][4 * m + e], acc[1][kh], 0, 0, 0);
                        }
                }
                // acc[pt][kh][e]: pixel 16 pt + 4 kq + e, column q16
                wave_lds_sync();                                     // (every lane's operand reads of Q are complete)
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
                        *reinterpret_cast<float4 *>(&T[(kh * 16 + q16) * N4_TPI + 16 * pt + 4 * kq]) =
                            make_float4(acc[pt][kh][0], acc[pt][kh][1], acc[pt][kh][2], acc[pt][kh][3]);
                wave_lds_sync();
                if (jj < N4_TP) {
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                            for (int u = 0; u < 2; ++u) S[kh][u] += T[(kh * 16 + kw * 4 + n0 + u) * N4_TPI + jj + kw];
                }
            }
            const int io = r - 1;                                    // the output row that is complete now
            if (io >= i0 && jj < N4_TP && j0 + jj < W) {
                const long long orow = (b * H + io) * W + j0 + jj;
                float o[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float x = R1[u] + S[2][u] + ebi[u];
                    x = x * esc[u] + esf[u];
                    if (p.residual) x += p.residual[orow * p.ldr + n0 + u];
                    o[u] = act_f(x, p.act, p.slope);
                }
                *reinterpret_cast<float2 *>(p.out + orow * p.ldo + n0) = make_float2(o[0], o[1]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) { R1[u] = R0[u] + S[1][u]; R0[u] = S[0][u]; }
        }
    }
}

// ---- C == 4 -> N == 4, 3x3 / stride 1 / pad 1 ("same"): the 1- and 2-channel convolutions behind G's transposed depth / mask heads
// (gnet.py:56-68 through net_utils.py:66-98) at full raw resolution: 15.7 M pixels at batch 8, 16 bytes in and 16 bytes out per pixel
// - a pure stencil, 0.5 GB per launch.  k_thin_c4 / k_thin_c4_wgrad spend their time on addressing (two 64-bit divisions and nine
// bounds-checked 64-bit row addresses per pixel: 0.44 ms forward / data gradient, 0.82 ms weight gradient with its atomics).  Here the
// grid is the image (x: 256 columns, y: a band of rows, z: sample), a thread walks down one column of its band with the 3x3 window in
// registers, and every image row is a buffer resource whose range check supplies the zero padding left, right, above and below.
constexpr int CN_RP = 4;                // output rows per thread, forward / data gradient
constexpr int CN_RPW = 32;              // rows per thread, weight gradient
typedef unsigned cn_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 cn_load(const float *rowp, unsigned bytes, int voff) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(rowp), 0, bytes, 0x00020000);
    const cn_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

__global__ void __launch_bounds__(256, 4)
k_c4n4_conv3x3(const TArgs p) {
    __shared__ __attribute__((aligned(16))) float4 wl[36];             // [t][n] -> W[n][t][0..3]
    if (threadIdx.x < 36) {
        const int t = threadIdx.x >> 2, n = threadIdx.x & 3;
        wl[threadIdx.x] = *reinterpret_cast<const float4 *>(p.W + ((long long)n * 9 + t) * 4);
    }
    __syncthreads();
    const int H = p.Hin, W = p.Win, lda4 = (int)p.lda * 4;
    const int j = blockIdx.x * 256 + threadIdx.x, i0 = blockIdx.y * CN_RP;
    const long long b = blockIdx.z;
    const unsigned rowbytes = (unsigned)W * (unsigned)lda4;
    float4 a[CN_RP + 2][3];
#pragma unroll
    for (int rr = 0; rr < CN_RP + 2; ++rr) {
        const int r = i0 - 1 + rr;
        const bool rok = (unsigned)r < (unsigned)H;                    // (uniform)
        const float *rowp = p.A + (b * H + (rok ? r : 0)) * W * p.lda;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) a[rr][kw] = cn_load(rowp, rok ? rowbytes : 0u, (j + kw - 1) * lda4);
    }
    float acc[CN_RP][4];
#pragma unroll
    for (int rr = 0; rr < CN_RP; ++rr)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[rr][n] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const float4 w = wl[t * 4 + n];                            // (wave-uniform address: an LDS broadcast)
#pragma unroll
            for (int rr = 0; rr < CN_RP; ++rr) {
                const float4 x = a[rr + t / 3][t % 3];
                acc[rr][n] += x.x * w.x + x.y * w.y + x.z * w.z + x.w * w.w;
            }
        }
        // pin the sums here: otherwise the whole computation is sunk into the per-row store branches below, behind ALL 36 weight
        // reads (144 live registers and, under an occupancy bound, a spilled kernel)
#pragma unroll
        for (int rr = 0; rr < CN_RP; ++rr)
#pragma unroll
            for (int n = 0; n < 4; ++n) asm volatile("" : "+v"(acc[rr][n]));
    }
    if (j >= W) return;
#pragma unroll
    for (int rr = 0; rr < CN_RP; ++rr)
        if (i0 + rr < H)
            epilogue_store(p, (b * H + i0 + rr) * W + j, 0, make_float4(acc[rr][0], acc[rr][1], acc[rr][2], acc[rr][3]));
}

// weight gradient: dW[n][t][c] = sum_p G[p][n] x[p + t][c]; 144 accumulators per thread over its column of CN_RPW rows (the window
// rolls down: three new 16-byte loads of x and one of G per pixel), one partial [4][9][4] plane per workgroup, planes folded in a
// fixed order (no atomics: bit-reproducible)
__global__ void __launch_bounds__(256)
k_c4n4_wgrad3x3(const TArgs p, float *part) {
    __shared__ float red[4][144];
    const int H = p.Hin, W = p.Win, lda4 = (int)p.lda * 4, ldg4 = (int)p.ldg * 4;
    const int j = blockIdx.x * 256 + threadIdx.x, i0 = blockIdx.y * CN_RPW;
    const long long b = blockIdx.z;
    const unsigned rowbytes = (unsigned)W * (unsigned)lda4, growbytes = (unsigned)W * (unsigned)ldg4;
    const int iend = i0 + CN_RPW < H ? i0 + CN_RPW : H;
    float acc[4][9][4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[n][t][c] = 0.f;
    float4 win[3][3];                                                  // x rows i - 1, i, i + 1 at columns j - 1 .. j + 1
    auto load_row = [&](int r, float4 (&dst)[3]) {
        const bool rok = (unsigned)r < (unsigned)H;
        const float *rowp = p.A + (b * H + (rok ? r : 0)) * W * p.lda;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) dst[kw] = cn_load(rowp, rok ? rowbytes : 0u, (j + kw - 1) * lda4);
    };
    auto load_g = [&](int r) {                                         // (columns >= W and rows >= iend: zeros)
        return cn_load(p.G + (b * H + (r < iend ? r : 0)) * W * p.ldg, r < iend ? growbytes : 0u, j * ldg4);
    };
    load_row(i0 - 1, win[0]);
    load_row(i0, win[1]);
    load_row(i0 + 1, win[2]);
    float4 g = load_g(i0);
    for (int i = i0; i < iend; ++i) {
        float4 nxt[3];
        load_row(i + 2, nxt);                                          // one row ahead: in flight during this row's 144 FMAs
        const float4 gn = load_g(i + 1);
        const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4 x = win[t / 3][t % 3];
            const float xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[n][t][c] = fmaf(gv[n], xv[c], acc[n][t][c]);
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) { win[0][kw] = win[1][kw]; win[1][kw] = win[2][kw]; win[2][kw] = nxt[kw]; }
        g = gn;
    }
    // workgroup sum: lanes by xor-shuffles, the four waves through LDS; plane of this workgroup in a fixed order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float v = acc[n][t][c];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                if (lane == 0) red[wave][(n * 9 + t) * 4 + c] = v;
            }
    __syncthreads();
    const long long blk = ((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (threadIdx.x < 144)
        part[blk * 144 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---- wgrad, C == 4:  dW[n][t][0..3] = sum_m G[orow(m)][n] * A[row(m,t)][0..3] -----------------------
// thread = (row lane, n-quad), the n-quads (a power of two) fastest; 16*T accumulators.  No atomics: the row lanes of a wave are
// summed by a shuffle butterfly (fixed shape), the four waves in LDS in wave order, and every workgroup leaves its own partial
// [N][T][4] plane in `p.dW` (= the workspace); efgh_thin_wgrad folds the planes in workgroup order.
template <int T>
__global__ void __launch_bounds__(TPB)
k_thin_c4_wgrad(const TArgs p) {
    extern __shared__ float red[];                                      // [4 waves][N*T*4]
    const int nq = p.N >> 2;                                            // n-quads: 1, 2, 4, ... 64
    const int RL = TPB / nq;                                            // row lanes
    const int q = threadIdx.x % nq, rl = threadIdx.x / nq;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[4][T][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[a][t][c] = 0.f;
    const long long mbeg = (long long)blockIdx.x * p.mchunk;
    long long mend = mbeg + p.mchunk;
    if (mend > p.M) mend = p.M;
    for (long long m = mbeg + rl; m < mend; m += RL) {
        int i, j; long long b;
        decode(p, m, i, j, b);
        float4 g = *reinterpret_cast<const float4 *>(p.G + out_row(p, i, j, b) * p.ldg + q * 4);
        const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int t = 0; t < T; ++t) {
            long long r = in_row(p, i, j, b, t);
            if (r < 0) continue;
            float4 a = *reinterpret_cast<const float4 *>(p.A + r * p.lda);
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int nn = 0; nn < 4; ++nn)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[nn][t][c] += gv[nn] * av[c];
        }
    }
    const int plane = p.N * T * 4;
#pragma unroll
    for (int nn = 0; nn < 4; ++nn)
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float v = acc[nn][t][c];
                for (int o = 32; o >= nq; o >>= 1) v += __shfl_xor(v, o);          // lanes nq apart hold the same n-quad
                if (lane < nq) red[wave * plane + ((q * 4 + nn) * T + t) * 4 + c] = v;
            }
    __syncthreads();
    float *out = p.dW + (long long)blockIdx.x * plane;
    for (int i = threadIdx.x; i < plane; i += TPB)
        out[i] = (red[i] + red[plane + i]) + (red[2 * plane + i] + red[3 * plane + i]);
}

// ---- wgrad, N == 4:  dW[n][k] = sum_m G[orow(m)][n] * A[row(m,t)][c]; thread = (row lane, k-quad).  No atomics: the row lanes
// are summed through LDS in row-lane order, every workgroup leaves its partial [4][K] plane in `p.dW` (= the workspace)
__global__ void __launch_bounds__(TPB)
k_thin_n4_wgrad(const TArgs p) {
    __shared__ float red[TPB * 16];
    const int kq = p.K >> 2;                       // k-quads
    const int KL = kq < TPB ? kq : TPB;            // k lanes per block row-lane group
    const int RL = TPB / KL;
    const int kl = threadIdx.x % KL, rl = threadIdx.x / KL;
    const int c4n = p.C >> 2;
    const long long mbeg = (long long)blockIdx.x * p.mchunk;
    long long mend = mbeg + p.mchunk;
    if (mend > p.M) mend = p.M;
    float *out = p.dW + (long long)blockIdx.x * 4 * p.K;
    for (int k0 = 0; k0 < kq; k0 += KL) {           // usually one pass (K <= 1024)
        const int k4 = k0 + kl;
        float acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
        if (rl < RL && k4 < kq) {
            const int t = k4 / c4n, c = k4 - t * c4n;
            for (long long m = mbeg + rl; m < mend; m += RL) {
                int i, j; long long b;
                decode(p, m, i, j, b);
                long long r = in_row(p, i, j, b, t);
                if (r < 0) continue;
                float4 g = *reinterpret_cast<const float4 *>(p.G + out_row(p, i, j, b) * p.ldg);
                float4 a = reinterpret_cast<const float4 *>(p.A + r * p.lda)[c];
                const float gv[4] = {g.x, g.y, g.z, g.w}, av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int nn = 0; nn < 4; ++nn)
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) acc[nn][cc] += gv[nn] * av[cc];
            }
        }
        __syncthreads();                            // (the previous pass has read red)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) red[(nn * 4 + cc) * TPB + threadIdx.x] = acc[nn][cc];
        __syncthreads();
        if (rl == 0 && k4 < kq) {
#pragma unroll
            for (int nn = 0; nn < 4; ++nn)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    float v = 0.f;
                    for (int r2 = 0; r2 < RL; ++r2) v += red[(nn * 4 + cc) * TPB + r2 * KL + kl];
                    out[(long long)nn * p.K + k4 * 4 + cc] = v;
                }
        }
    }
}

int fill(TArgs &a, const efgh_gemm_desc *d) {
    a.A = d->A; a.lda = d->lda; a.C = d->C; a.T = d->T; a.K = d->T * d->C;
    a.Hin = d->Hin; a.Win = d->Win; a.Hv = d->Hv; a.Wv = d->Wv; a.sh = d->sh; a.sw = d->sw;
    for (int t = 0; t < 16; ++t) { a.dh[t] = t < d->T ? d->dh[t] : 0; a.dw[t] = t < d->T ? d->dw[t] : 0; }
    a.Ho = d->Ho; a.Wo = d->Wo; a.osh = d->osh; a.osw = d->osw; a.oh0 = d->oh0; a.ow0 = d->ow0;
    a.W = d->W; a.N = d->N; a.M = d->M;
    a.bias = d->bias; a.scale = d->scale; a.shift = d->shift; a.residual = d->residual; a.ldr = d->ldr;
    a.act = d->act; a.slope = d->slope; a.out = d->out; a.ldo = d->ldo;
    a.G = nullptr; a.ldg = 0; a.dW = nullptr; a.mchunk = 0;
    return 0;
}

// 3x3, stride 1, pad 1, "same" size, one launch, image rows addressable with 32-bit byte offsets
bool same3x3_ok(const efgh_gemm_desc *d) {
    if (!d || d->mode != 1 || d->T != 9 || d->M_dev || d->nbatch > 1) return false;
    if (d->sh != 1 || d->sw != 1 || d->osh != 1 || d->osw != 1 || d->oh0 || d->ow0) return false;
    if (d->Hv != d->Ho || d->Wv != d->Wo || d->Ho != d->Hin || d->Wo != d->Win || d->B <= 0) return false;
    for (int t = 0; t < 9; ++t) if (d->dh[t] != t / 3 - 1 || d->dw[t] != t % 3 - 1) return false;
    return (int64_t)(d->Win + 512) * d->lda * 4 < (1ll << 31) && d->lda % 4 == 0 && (((uintptr_t)d->A) & 15) == 0;
}

// k_c4n4_conv3x3 / k_c4n4_wgrad3x3
bool c4n4_ok(const efgh_gemm_desc *d) {
    return same3x3_ok(d) && d->C == 4 && d->N == 4 && d->B <= 65535 && (((uintptr_t)d->W) & 15) == 0 && d->ldo % 4 == 0 &&
           (((uintptr_t)d->out) & 15) == 0 && (!d->residual || d->ldr % 4 == 0);
}

// k_n4_conv3x3_c64: 64 -> 4 channels, 3x3, stride 1, pad 1, "same" size, 16-byte input rows, 8-byte output pairs
bool n4_mfma_ok(const efgh_gemm_desc *d) {
    if (d->N != 4 || d->C != 64 || d->T != 9 || d->M_dev || d->nbatch > 1) return false;
    if (d->sh != 1 || d->sw != 1 || d->osh != 1 || d->osw != 1 || d->oh0 || d->ow0) return false;
    if (d->Hv != d->Ho || d->Wv != d->Wo || d->Ho != d->Hin || d->Wo != d->Win || d->B <= 0) return false;
    for (int t = 0; t < 9; ++t) if (d->dh[t] != t / 3 - 1 || d->dw[t] != t % 3 - 1) return false;
    if (d->residual && d->ldr % 2 != 0) return false;
    if ((int64_t)(d->Win + 32) * d->lda * 4 >= (1ll << 31)) return false;                 // 32-bit byte offsets within an image row
    return d->lda % 4 == 0 && (((uintptr_t)d->A) & 15) == 0 && (((uintptr_t)d->W) & 15) == 0 && d->ldo % 2 == 0 &&
           (((uintptr_t)d->out) & 7) == 0;
}

int grid_for(long long total, int per) {
    long long g = (total + per - 1) / per;
    return (int)(g > 32768 ? 32768 : (g < 1 ? 1 : g));
}
}  // namespace

extern "C" int efgh_thin_supported(const efgh_gemm_desc *d) {
    if (!d || d->mode != 1 || d->N % 4 != 0 || d->C % 4 != 0 || d->stats) return 0;
    if (c4n4_ok(d)) return 4;
    if (d->C == 4 && d->N <= 256 && (d->T == 1 || d->T == 2 || d->T == 4 || d->T == 9) &&
        (int64_t)d->T * d->N * 16 + 4 * 64 * 36 * 4 <= 64 * 1024) return 1;
    if (d->N == 4 && (int64_t)d->T * d->C * 16 <= 60 * 1024) return n4_mfma_ok(d) ? 3 : 2;
    return 0;
}

extern "C" int efgh_thin_gemm(const efgh_gemm_desc *d, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    int kind = efgh_thin_supported(d);
    EFGH_CHECK_ARG(kind != 0);
    EFGH_CHECK_ARG(d->A && d->W && d->out && d->M == (int64_t)d->B * d->Hv * d->Wv && d->lda % 4 == 0 && d->ldo % 4 == 0);
    TArgs a;
    fill(a, d);
    if (kind == 4) {
        k_c4n4_conv3x3<<<dim3((a.Win + 255) / 256, (a.Hin + CN_RP - 1) / CN_RP, d->B), 256, 0, st>>>(a);
    } else if (kind == 1) {
        size_t lds = (size_t)a.T * a.N * 16 + 4 * 64 * 36 * 4;
        int grid = grid_for(a.M, TPB);
        switch (a.T) {
        case 1: k_thin_c4<1><<<grid, TPB, lds, st>>>(a); break;
        case 2: k_thin_c4<2><<<grid, TPB, lds, st>>>(a); break;
        case 4: k_thin_c4<4><<<grid, TPB, lds, st>>>(a); break;
        case 9: k_thin_c4<9><<<grid, TPB, lds, st>>>(a); break;
        default: efgh_set_error("thin c4: unsupported tap count %d", a.T); return EFGH_E_INVALID;
        }
    } else if (kind == 3) {
        const int strips = (a.Win + N4_TP - 1) / N4_TP, chunks = (a.Hin + N4_RC - 1) / N4_RC;
        const long long units = (long long)d->B * strips * chunks, g = (units + N4_WAVES - 1) / N4_WAVES;
        k_n4_conv3x3_c64<<<(int)g, 64 * N4_WAVES, 0, st>>>(a, strips, chunks, units);
    } else
        k_thin_n4<<<grid_for(a.M, TPB / 8), TPB, (size_t)a.K * 16, st>>>(a);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

namespace {
int thin_wgrad_grid(const efgh_gemm_desc *d, long long *chunk_out) {
    long long chunk = (d->M + 2047) / 2048;
    if (chunk < 512) chunk = 512;
    if (chunk_out) *chunk_out = chunk;
    return (int)((d->M + chunk - 1) / chunk);
}
bool thin_wgrad_c4(const efgh_gemm_desc *d) {
    const int nq = d->N >> 2;
    return d->C == 4 && d->N <= 256 && !(nq & (nq - 1)) && (d->T == 9 || d->T == 2 || d->T == 1 || d->T == 4) &&
           (size_t)d->N * d->T * 64 <= 60 * 1024;
}
}  // namespace

/* floats of scratch efgh_thin_wgrad needs: one partial [N][K] plane per workgroup, folded in workgroup order (no atomics) */
extern "C" int64_t efgh_thin_wgrad_workspace(const efgh_gemm_desc *d) {
    if (!d || d->mode != 1 || d->M < 1 || d->N < 4 || d->C < 4 || d->T < 1) return 0;
    return (int64_t)thin_wgrad_grid(d, nullptr) * d->N * d->T * d->C;
}

extern "C" int efgh_thin_wgrad(const efgh_gemm_desc *d, const float *G, int64_t ldg, float *dWp, float *workspace,
                               const efgh_wgrad_out_desc *out, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(d && G && dWp && d->mode == 1 && d->N % 4 == 0 && d->C % 4 == 0 && ldg % 4 == 0);
    EFGH_CHECK_ARG(d->M == (int64_t)d->B * d->Hv * d->Wv);
    EFGH_CHECK_ARG(workspace && (((uintptr_t)workspace) & 15) == 0 && (((uintptr_t)dWp) & 15) == 0);
    TArgs a;
    fill(a, d);
    a.G = G; a.ldg = ldg; a.dW = workspace;
    long long chunk = 0;
    const int grid = thin_wgrad_grid(d, &chunk);
    a.mchunk = (int)chunk;
    if (thin_wgrad_c4(d)) {
        const size_t lds = (size_t)a.N * a.T * 64;
        if (d->T == 9) k_thin_c4_wgrad<9><<<grid, TPB, lds, st>>>(a);
        else if (d->T == 4) k_thin_c4_wgrad<4><<<grid, TPB, lds, st>>>(a);
        else if (d->T == 2) k_thin_c4_wgrad<2><<<grid, TPB, lds, st>>>(a);
        else k_thin_c4_wgrad<1><<<grid, TPB, lds, st>>>(a);
    } else if (d->N == 4) {
        k_thin_n4_wgrad<<<grid, TPB, 0, st>>>(a);
    } else {
        efgh_set_error("thin wgrad: unsupported shape C=%d N=%d T=%d", d->C, d->N, d->T);
        return EFGH_E_INVALID;
    }
    const bool wrote = efgh_launch_fold_splits(workspace, grid, (long long)a.N * a.K, dWp, st, out);
    EFGH_CHECK_LAUNCH();
    return wrote ? EFGH_WROTE_OUT : EFGH_OK;
}

/* 4 -> 4 channels, 3x3, stride 1, pad 1: deterministic weight gradient (k_c4n4_wgrad3x3); workspace in floats */
extern "C" int efgh_c4n4_supported(const efgh_gemm_desc *d) { return c4n4_ok(d) ? 1 : 0; }

extern "C" int64_t efgh_c4n4_wgrad_workspace(const efgh_gemm_desc *d) {
    if (!c4n4_ok(d)) return 0;
    return (int64_t)((d->Win + 255) / 256) * ((d->Hin + CN_RPW - 1) / CN_RPW) * d->B * 144;
}

extern "C" int efgh_c4n4_wgrad(const efgh_gemm_desc *d, const float *G, int64_t ldg, float *dWp, float *workspace,
                               const efgh_wgrad_out_desc *out, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(c4n4_ok(d) && G && dWp && workspace && ldg % 4 == 0 && (((uintptr_t)G) & 15) == 0 &&
                   (int64_t)(d->Win + 512) * ldg * 4 < (1ll << 31));
    TArgs a;
    fill(a, d);
    a.G = G; a.ldg = ldg;
    const dim3 grid((a.Win + 255) / 256, (a.Hin + CN_RPW - 1) / CN_RPW, d->B);
    k_c4n4_wgrad3x3<<<grid, 256, 0, st>>>(a, workspace);
    const bool wrote = efgh_launch_fold_splits(workspace, (int)(grid.x * grid.y * grid.z), 144, dWp, st, out);
    EFGH_CHECK_LAUNCH();
    return wrote ? EFGH_WROTE_OUT : EFGH_OK;
}

// ---- stride-2 transposed conv with <= 2 output channels as GEMM + col2im ------------------------------
// (gnet.py:56-68 heads).  Y[pix_in][(kh*3+kw)*O + o] = x[pix_in][:] . W[:, o, kh, kw] comes from ONE
// gather-GEMM launch that reads the 128-channel input once; these kernels fold / unfold the 3x3 taps.
namespace {
// out[b][oh][ow][o] = act(scale*(sum_{taps} Y[b][ih][iw][tap*O+o]) + shift), oh = 2*ih - pad + kh
__global__ void __launch_bounds__(256)
k_convt_col2im(const float *__restrict__ Y, long long ldy, int B, int Hin, int Win, int Ho, int Wo, int O, int pad,
               const float *__restrict__ scale, const float *__restrict__ shift, int act, float slope,
               float *__restrict__ out, long long ldo) {
    long long total = (long long)B * Ho * Wo;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        int ow = (int)(g % Wo); long long r = g / Wo;
        int oh = (int)(r % Ho); long long b = r / Ho;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            int th = oh + pad - kh;
            if (th < 0 || (th & 1)) continue;
            int ih = th >> 1;
            if (ih >= Hin) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                int tw = ow + pad - kw;
                if (tw < 0 || (tw & 1)) continue;
                int iw = tw >> 1;
                if (iw >= Win) continue;
                const float *y = Y + ((b * Hin + ih) * Win + iw) * ldy + (kh * 3 + kw) * O;
                for (int o = 0; o < O; ++o) acc[o] += y[o];
            }
        }
        float4 v;
        float *vp = &v.x;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float x = acc[o];
            if (o < O) {
                x = x * (scale ? scale[o] : 1.f) + (shift ? shift[o] : 0.f);
                x = act_f(x, act, slope);
            } else x = 0.f;
            vp[o] = x;
        }
        *reinterpret_cast<float4 *>(out + g * ldo) = v;
    }
}

// dYcol[b][ih][iw][tap*O+o] = G[b][2ih-pad+kh][2iw-pad+kw][o]  (0 outside), columns >= 9*O zero
__global__ void __launch_bounds__(256)
k_convt_im2col(const float *__restrict__ G, long long ldg, int B, int Hin, int Win, int Ho, int Wo, int O, int pad,
               float *__restrict__ Ycol, long long ldy) {
    long long total = (long long)B * Hin * Win;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        int iw = (int)(g % Win); long long r = g / Win;
        int ih = (int)(r % Hin); long long b = r / Hin;
        float *y = Ycol + g * ldy;
        for (int t = 0; t < 9; ++t) {
            int oh = 2 * ih - pad + t / 3, ow = 2 * iw - pad + t % 3;
            bool ok = oh >= 0 && oh < Ho && ow >= 0 && ow < Wo;
            const float *src = G + ((b * Ho + (ok ? oh : 0)) * Wo + (ok ? ow : 0)) * ldg;
            for (int o = 0; o < O; ++o) y[t * O + o] = ok ? src[o] : 0.f;
        }
        for (int c = 9 * O; c < ldy; ++c) y[c] = 0.f;
    }
}
// the same two folds with the grid as the image (no divisions), the nine gradient taps as 16-byte buffer loads whose range check
// supplies the zeros outside the output, and the column row assembled in registers and written as 16-byte stores
template <int O>
__global__ void __launch_bounds__(256)
k_convt_im2col_v(const float *__restrict__ G, long long ldg, int Hin, int Win, int Ho, int Wo, int pad, float *__restrict__ Ycol,
                 int ldy4) {
    const int iw = blockIdx.x * 256 + threadIdx.x, ih = blockIdx.y;
    const long long b = blockIdx.z;
    const int ldg4 = (int)ldg * 4;
    const unsigned rowbytes = (unsigned)Wo * (unsigned)ldg4;
    float v[40];
#pragma unroll
    for (int q = 0; q < 40; ++q) v[q] = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int oh = 2 * ih - pad + kh;
        const bool rok = (unsigned)oh < (unsigned)Ho;                 // (uniform)
        const float *rowp = G + (b * Ho + (rok ? oh : 0)) * Wo * ldg;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const float4 g = cn_load(rowp, rok ? rowbytes : 0u, (2 * iw - pad + kw) * ldg4);
            const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int o = 0; o < O; ++o) v[(kh * 3 + kw) * O + o] = gv[o];
        }
    }
    if (iw >= Win) return;
    float4 *y = reinterpret_cast<float4 *>(Ycol + ((b * Hin + ih) * Win + iw) * (long long)(ldy4 * 4));
#pragma unroll
    for (int q = 0; q < 10; ++q)
        if (q < ldy4) y[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

template <int O>
__global__ void __launch_bounds__(256)
k_convt_col2im_v(const float *__restrict__ Y, long long ldy, int Hin, int Win, int Ho, int Wo, int pad,
                 const float *__restrict__ scale, const float *__restrict__ shift, int act, float slope, float *__restrict__ out,
                 long long ldo) {
    const int ow = blockIdx.x * 256 + threadIdx.x, oh = blockIdx.y;
    const long long b = blockIdx.z;
    if (ow >= Wo) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int th = oh + pad - kh;                                 // (uniform)
        if (th < 0 || (th & 1) || (th >> 1) >= Hin) continue;
        const float *yrow = Y + ((b * Hin + (th >> 1)) * Win) * ldy + kh * 3 * O;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int tw = ow + pad - kw;
            if (tw < 0 || (tw & 1) || (tw >> 1) >= Win) continue;
            const float *y = yrow + (long long)(tw >> 1) * ldy + kw * O;
#pragma unroll
            for (int o = 0; o < O; ++o) acc[o] += y[o];
        }
    }
    float4 v;
    float *vp = &v.x;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        float x = acc[o];
        if (o < O) {
            x = x * (scale ? scale[o] : 1.f) + (shift ? shift[o] : 0.f);
            x = act_f(x, act, slope);
        } else x = 0.f;
        vp[o] = x;
    }
    *reinterpret_cast<float4 *>(out + ((b * Ho + oh) * Wo + ow) * ldo) = v;
}
}  // namespace

extern "C" int efgh_convt_col2im(const float *Y, int64_t ldy, int32_t B, int32_t Hin, int32_t Win, int32_t Ho,
                                 int32_t Wo, int32_t O, int32_t pad, const float *scale, const float *shift,
                                 int32_t act, float slope, float *out, int64_t ldo, void *stream_) {
    EFGH_CHECK_ARG(Y && out && B > 0 && Hin > 0 && Win > 0 && Ho > 0 && Wo > 0 && O >= 1 && O <= 4);
    EFGH_CHECK_ARG(ldy >= 9 * O && ldo >= 4 && ldo % 4 == 0);
    hipStream_t st = (hipStream_t)stream_;
    if (B <= 65535 && Ho <= 65535) {
        const dim3 grid((Wo + 255) / 256, Ho, B);
        switch (O) {
        case 1: k_convt_col2im_v<1><<<grid, 256, 0, st>>>(Y, ldy, Hin, Win, Ho, Wo, pad, scale, shift, act, slope, out, ldo); break;
        case 2: k_convt_col2im_v<2><<<grid, 256, 0, st>>>(Y, ldy, Hin, Win, Ho, Wo, pad, scale, shift, act, slope, out, ldo); break;
        case 3: k_convt_col2im_v<3><<<grid, 256, 0, st>>>(Y, ldy, Hin, Win, Ho, Wo, pad, scale, shift, act, slope, out, ldo); break;
        default: k_convt_col2im_v<4><<<grid, 256, 0, st>>>(Y, ldy, Hin, Win, Ho, Wo, pad, scale, shift, act, slope, out, ldo); break;
        }
    } else
        k_convt_col2im<<<grid_for((long long)B * Ho * Wo, 256), 256, 0, st>>>(
            Y, ldy, B, Hin, Win, Ho, Wo, O, pad, scale, shift, act, slope, out, ldo);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_convt_im2col(const float *G, int64_t ldg, int32_t B, int32_t Hin, int32_t Win, int32_t Ho,
                                 int32_t Wo, int32_t O, int32_t pad, float *Ycol, int64_t ldy, void *stream_) {
    EFGH_CHECK_ARG(G && Ycol && B > 0 && Hin > 0 && Win > 0 && Ho > 0 && Wo > 0 && O >= 1 && O <= 4 && ldy >= 9 * O);
    hipStream_t st = (hipStream_t)stream_;
    const bool vec = B <= 65535 && Hin <= 65535 && ldy % 4 == 0 && ldy <= 40 && ldg % 4 == 0 && (((uintptr_t)G) & 15) == 0 &&
                     (((uintptr_t)Ycol) & 15) == 0 && (int64_t)(2 * Win + 8) * ldg * 4 < (1ll << 31);
    if (vec) {
        const dim3 grid((Win + 255) / 256, Hin, B);
        const int ldy4 = (int)(ldy / 4);
        switch (O) {
        case 1: k_convt_im2col_v<1><<<grid, 256, 0, st>>>(G, ldg, Hin, Win, Ho, Wo, pad, Ycol, ldy4); break;
        case 2: k_convt_im2col_v<2><<<grid, 256, 0, st>>>(G, ldg, Hin, Win, Ho, Wo, pad, Ycol, ldy4); break;
        case 3: k_convt_im2col_v<3><<<grid, 256, 0, st>>>(G, ldg, Hin, Win, Ho, Wo, pad, Ycol, ldy4); break;
        default: k_convt_im2col_v<4><<<grid, 256, 0, st>>>(G, ldg, Hin, Win, Ho, Wo, pad, Ycol, ldy4); break;
        }
    } else
        k_convt_im2col<<<grid_for((long long)B * Hin * Win, 256), 256, 0, st>>>(G, ldg, B, Hin, Win, Ho, Wo, O, pad, Ycol, ldy);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}
