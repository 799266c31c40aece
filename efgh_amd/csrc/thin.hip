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

// ---- wgrad, C == 4:  dW[n][t][0..3] += sum_m G[orow(m)][n] * A[row(m,t)][0..3] -----------------------
// thread = (row lane, n-quad); 16*T accumulators; block-level LDS reduction, then global atomics.
template <int T>
__global__ void __launch_bounds__(TPB)
k_thin_c4_wgrad(const TArgs p) {
    extern __shared__ float red[];                                      // N*T*4 floats
    const int nq = p.N >> 2;                                            // n-quads (<= 64)
    const int RL = TPB / nq;                                            // row lanes
    const int q = threadIdx.x % nq, rl = threadIdx.x / nq;
    for (int i = threadIdx.x; i < p.N * T * 4; i += TPB) red[i] = 0.f;
    __syncthreads();
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
    if (rl < RL)
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
    if (rl < RL) {
#pragma unroll
        for (int nn = 0; nn < 4; ++nn)
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int c = 0; c < 4; ++c) atomicAdd(&red[((q * 4 + nn) * T + t) * 4 + c], acc[nn][t][c]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < p.N * T * 4; i += TPB) atomicAdd(&p.dW[i], red[i]);
}

// ---- wgrad, N == 4:  dW[n][k] += sum_m G[orow(m)][n] * A[row(m,t)][c]; thread = (row lane, k-quad) ----
__global__ void __launch_bounds__(TPB)
k_thin_n4_wgrad(const TArgs p) {
    const int kq = p.K >> 2;                       // k-quads
    const int KL = kq < TPB ? kq : TPB;            // k lanes per block row-lane group
    const int RL = TPB / KL;
    const int kl = threadIdx.x % KL, rl = threadIdx.x / KL;
    const int c4n = p.C >> 2;
    const long long mbeg = (long long)blockIdx.x * p.mchunk;
    long long mend = mbeg + p.mchunk;
    if (mend > p.M) mend = p.M;
    for (int k4 = kl; k4 < kq; k4 += KL) {          // usually one pass (K <= 1024)
        const int t = k4 / c4n, c = k4 - t * c4n;
        float acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
        if (rl < RL)
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
        if (rl < RL) {
#pragma unroll
            for (int nn = 0; nn < 4; ++nn)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
                    atomicAdd(&p.dW[(long long)nn * p.K + k4 * 4 + cc], acc[nn][cc]);
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

int grid_for(long long total, int per) {
    long long g = (total + per - 1) / per;
    return (int)(g > 32768 ? 32768 : (g < 1 ? 1 : g));
}
}  // namespace

extern "C" int efgh_thin_supported(const efgh_gemm_desc *d) {
    if (!d || d->mode != 1 || d->N % 4 != 0 || d->C % 4 != 0 || d->stats) return 0;
    if (d->C == 4 && d->N <= 256 && (d->T == 1 || d->T == 2 || d->T == 4 || d->T == 9) &&
        (int64_t)d->T * d->N * 16 + 4 * 64 * 36 * 4 <= 64 * 1024) return 1;
    if (d->N == 4 && (int64_t)d->T * d->C * 16 <= 60 * 1024) return 2;
    return 0;
}

extern "C" int efgh_thin_gemm(const efgh_gemm_desc *d, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    int kind = efgh_thin_supported(d);
    EFGH_CHECK_ARG(kind != 0);
    EFGH_CHECK_ARG(d->A && d->W && d->out && d->M == (int64_t)d->B * d->Hv * d->Wv && d->lda % 4 == 0 && d->ldo % 4 == 0);
    TArgs a;
    fill(a, d);
    if (kind == 1) {
        size_t lds = (size_t)a.T * a.N * 16 + 4 * 64 * 36 * 4;
        int grid = grid_for(a.M, TPB);
        switch (a.T) {
        case 1: k_thin_c4<1><<<grid, TPB, lds, st>>>(a); break;
        case 2: k_thin_c4<2><<<grid, TPB, lds, st>>>(a); break;
        case 4: k_thin_c4<4><<<grid, TPB, lds, st>>>(a); break;
        case 9: k_thin_c4<9><<<grid, TPB, lds, st>>>(a); break;
        default: efgh_set_error("thin c4: unsupported tap count %d", a.T); return EFGH_E_INVALID;
        }
    } else
        k_thin_n4<<<grid_for(a.M, TPB / 8), TPB, (size_t)a.K * 16, st>>>(a);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_thin_wgrad(const efgh_gemm_desc *d, const float *G, int64_t ldg, float *dWp, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(d && G && dWp && d->mode == 1 && d->N % 4 == 0 && d->C % 4 == 0 && ldg % 4 == 0);
    EFGH_CHECK_ARG(d->M == (int64_t)d->B * d->Hv * d->Wv);
    TArgs a;
    fill(a, d);
    a.G = G; a.ldg = ldg; a.dW = dWp;
    if (hipMemsetAsync(dWp, 0, (size_t)a.N * a.K * 4, st) != hipSuccess) {
        efgh_set_error("thin wgrad: memset failed");
        return EFGH_E_LAUNCH;
    }
    long long chunk = (a.M + 2047) / 2048;
    if (chunk < 512) chunk = 512;
    a.mchunk = (int)chunk;
    int grid = (int)((a.M + chunk - 1) / chunk);
    if (d->C == 4 && d->N <= 256 && (d->T == 9 || d->T == 2 || d->T == 1 || d->T == 4)) {
        size_t lds = (size_t)a.N * a.T * 16;
        EFGH_CHECK_ARG(lds <= 60 * 1024);
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
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
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
}  // namespace

extern "C" int efgh_convt_col2im(const float *Y, int64_t ldy, int32_t B, int32_t Hin, int32_t Win, int32_t Ho,
                                 int32_t Wo, int32_t O, int32_t pad, const float *scale, const float *shift,
                                 int32_t act, float slope, float *out, int64_t ldo, void *stream_) {
    EFGH_CHECK_ARG(Y && out && B > 0 && Hin > 0 && Win > 0 && Ho > 0 && Wo > 0 && O >= 1 && O <= 4);
    EFGH_CHECK_ARG(ldy >= 9 * O && ldo >= 4 && ldo % 4 == 0);
    k_convt_col2im<<<grid_for((long long)B * Ho * Wo, 256), 256, 0, (hipStream_t)stream_>>>(
        Y, ldy, B, Hin, Win, Ho, Wo, O, pad, scale, shift, act, slope, out, ldo);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_convt_im2col(const float *G, int64_t ldg, int32_t B, int32_t Hin, int32_t Win, int32_t Ho,
                                 int32_t Wo, int32_t O, int32_t pad, float *Ycol, int64_t ldy, void *stream_) {
    EFGH_CHECK_ARG(G && Ycol && B > 0 && Hin > 0 && Win > 0 && Ho > 0 && Wo > 0 && O >= 1 && O <= 4 && ldy >= 9 * O);
    k_convt_im2col<<<grid_for((long long)B * Hin * Win, 256), 256, 0, (hipStream_t)stream_>>>(G, ldg, B, Hin, Win, Ho,
                                                                                             Wo, O, pad, Ycol, ldy);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}
