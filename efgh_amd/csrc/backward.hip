// Backward passes of the HBM-bound layers (autograd of BatchNorm + activation + residual,
// MaxPool2d, max/mean over positions, the 2-channel softmax and the F correlation head).
// Each kernel is one streaming pass with 16-B accesses along the channel axis.
#include "common.h"

namespace {
constexpr int TPB = 256;

// the sign bits efgh_scale_shift_act_bits left (bit e = element e of the [M][C] activation is > 0) as a stand-in for the activation:
// +1 / -1 for the four channels of a quad (C % 32 == 0: a quad never straddles a word)
__device__ __forceinline__ float4 sign4(const unsigned *__restrict__ bits, long long e) {
    const unsigned w = bits[e >> 5] >> ((unsigned)e & 31u);
    return make_float4((w & 1u) ? 1.f : -1.f, (w & 2u) ? 1.f : -1.f, (w & 4u) ? 1.f : -1.f, (w & 8u) ? 1.f : -1.f);
}

__device__ __forceinline__ float dact(float y, int act, float slope) {
    if (act == 1) return y > 0.f ? 1.f : 0.f;
    if (act == 2) return y > 0.f ? 1.f : slope;
    return 1.f;
}

// per-column partial sums of  dpre = dy*act'(y)  and  dpre*xhat,  xhat = (raw-mean)*invstd
// (mean == NULL: xhat := 0, only sum dpre is meaningful -> bias gradient).
// block = (CL float4 channel lanes) x (256/CL row lanes), 512 rows per block.
template <bool NT>
__global__ void __launch_bounds__(TPB)
k_act_bn_bwd_reduce(const float *__restrict__ dy, long long lddy, const float *__restrict__ y, long long ldy,
                    const float *__restrict__ raw, long long ldraw, const float *__restrict__ mean,
                    const float *__restrict__ invstd, const float *__restrict__ pscale,
                    const float *__restrict__ pshift, long long M, int C, int act, float slope, int rows_per_block,
                    int CL, double *__restrict__ part, const unsigned *__restrict__ ybits) {
    // The column sums are accumulated in float64 (as torch's CPU batch-norm backward does, acc_type<float> = double):
    // mean(dpre) and mean(dpre*xhat) are subtracted from EVERY row, so a 1e-7 relative error in them is a systematic
    // bias of draw that the following weight gradient multiplies by the channel mean of the layer input - with nearly
    // constant feature maps (the G translation head at config S) that was a 10-25 % error of the upstream gradients.
    __shared__ double s1s[TPB][4], s2s[TPB][4];
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL, RL = TPB / CL;
    const int c = (blockIdx.x * CL + cl) * 4;
    long long r0 = (long long)blockIdx.y * rows_per_block, r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    double s1[4] = {0., 0., 0., 0.}, s2[4] = {0., 0., 0., 0.};
    if (c < C) {
        float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = mu, psc = mu, psh = mu;
        if (mean) { mu = *reinterpret_cast<const float4 *>(mean + c); is = *reinterpret_cast<const float4 *>(invstd + c); }
        const bool from_raw = !y && !ybits;
        if (from_raw) { psc = *reinterpret_cast<const float4 *>(pscale + c); psh = *reinterpret_cast<const float4 *>(pshift + c); }
        for (long long r = r0 + rl; r < r1; r += RL) {
            float4 g = ld_stream<NT>(dy + r * lddy + c);
            float4 rw = make_float4(0.f, 0.f, 0.f, 0.f);
            if (mean || from_raw) rw = ld_stream<NT>(raw + r * ldraw + c);
            float4 yy;
            if (ybits) yy = sign4(ybits, r * C + c);       // (1 bit per element instead of a second read of the activation)
            else if (y) yy = *reinterpret_cast<const float4 *>(y + r * ldy + c);
            else yy = make_float4(rw.x * psc.x + psh.x, rw.y * psc.y + psh.y, rw.z * psc.z + psh.z, rw.w * psc.w + psh.w);
            g.x *= dact(yy.x, act, slope); g.y *= dact(yy.y, act, slope);
            g.z *= dact(yy.z, act, slope); g.w *= dact(yy.w, act, slope);
            s1[0] += (double)g.x; s1[1] += (double)g.y; s1[2] += (double)g.z; s1[3] += (double)g.w;
            if (mean) {
                s2[0] += (double)g.x * (double)((rw.x - mu.x) * is.x); s2[1] += (double)g.y * (double)((rw.y - mu.y) * is.y);
                s2[2] += (double)g.z * (double)((rw.z - mu.z) * is.z); s2[3] += (double)g.w * (double)((rw.w - mu.w) * is.w);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { s1s[threadIdx.x][q] = s1[q]; s2s[threadIdx.x][q] = s2[q]; }
    __syncthreads();
    if (rl != 0 || c >= C) return;
    for (int i = 1; i < RL; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) { s1[q] += s1s[i * CL + cl][q]; s2[q] += s2s[i * CL + cl][q]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        part[((long long)blockIdx.y * 2) * C + c + q] = s1[q];
        part[((long long)blockIdx.y * 2 + 1) * C + c + q] = s2[q];
    }
}

__global__ void __launch_bounds__(1024)
k_bwd_finalize(const double *__restrict__ part, int G, int C, double count,
               float *__restrict__ sum_dpre, float *__restrict__ sum_dpre_xhat,
               double *__restrict__ mean_dpre, double *__restrict__ mean_dpre_xhat) {
    __shared__ double sa[32][33], sb[32][33];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int c = blockIdx.x * 32 + tx;
    double a = 0.0, b = 0.0;
    if (c < C) {
        // four independent partial sums: the loads of a trip are in flight together (a single chain costs one memory latency per
        // group: 18 us for the 3 840 groups of a full-resolution layer)
        double a1 = 0.0, b1 = 0.0, a2 = 0.0, b2 = 0.0, a3 = 0.0, b3 = 0.0;
        int g = ty;
        for (; g + 96 < G; g += 128) {
            a += part[((long long)g * 2) * C + c]; b += part[((long long)g * 2 + 1) * C + c];
            a1 += part[((long long)(g + 32) * 2) * C + c]; b1 += part[((long long)(g + 32) * 2 + 1) * C + c];
            a2 += part[((long long)(g + 64) * 2) * C + c]; b2 += part[((long long)(g + 64) * 2 + 1) * C + c];
            a3 += part[((long long)(g + 96) * 2) * C + c]; b3 += part[((long long)(g + 96) * 2 + 1) * C + c];
        }
        for (; g < G; g += 32) { a += part[((long long)g * 2) * C + c]; b += part[((long long)g * 2 + 1) * C + c]; }
        a += (a1 + a2) + a3; b += (b1 + b2) + b3;
    }
    sa[ty][tx] = a; sb[ty][tx] = b;
    __syncthreads();
    if (ty != 0 || c >= C) return;
    for (int i = 1; i < 32; ++i) { a += sa[i][tx]; b += sb[i][tx]; }
    sum_dpre[c] = (float)a; sum_dpre_xhat[c] = (float)b;
    if (mean_dpre) { mean_dpre[c] = a / count; mean_dpre_xhat[c] = b / count; }
}

// draw = coef[c] * (dpre - m1[c] - xhat*m2[c])   (train BN; coef = gamma*invstd)
//      = coef[c] * dpre                          (eval BN / no BN: mean == NULL, coef optional)
// dres (optional) = dpre
template <bool NT>
__global__ void __launch_bounds__(TPB)
k_act_bn_bwd_apply(const float *__restrict__ dy, long long lddy, const float *__restrict__ y, long long ldy,
                   const float *__restrict__ raw, long long ldraw, const float *__restrict__ mean,
                   const float *__restrict__ invstd, const float *__restrict__ coef,
                   const double *__restrict__ m1, const double *__restrict__ m2, const float *__restrict__ pscale,
                   const float *__restrict__ pshift, long long M, int C, int act,
                   float slope, float *__restrict__ draw, long long lddraw, float *__restrict__ dres,
                   long long lddres, const unsigned *__restrict__ ybits) {
    const int c4n = C >> 2;
    long long total = M * c4n;
    const bool from_raw = !y && !ybits;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        long long r = i / c4n; int c = (int)(i - r * c4n) * 4;
        float4 g = ld_stream<NT>(dy + r * lddy + c);
        float4 rw = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mean || from_raw) rw = ld_stream<NT>(raw + r * ldraw + c);
        float4 yy;
        if (ybits) yy = sign4(ybits, r * C + c);
        else if (y) yy = *reinterpret_cast<const float4 *>(y + r * ldy + c);
        else yy = make_float4(rw.x * pscale[c] + pshift[c], rw.y * pscale[c + 1] + pshift[c + 1],
                              rw.z * pscale[c + 2] + pshift[c + 2], rw.w * pscale[c + 3] + pshift[c + 3]);
        float gv[4] = {g.x * dact(yy.x, act, slope), g.y * dact(yy.y, act, slope), g.z * dact(yy.z, act, slope),
                       g.w * dact(yy.w, act, slope)};
        if (dres) st_stream<NT>(dres + r * lddres + c, make_float4(gv[0], gv[1], gv[2], gv[3]));
        float o[4];
        if (mean) {
            float rv[4] = {rw.x, rw.y, rw.z, rw.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // float64 like torch's CPU kernel: the two means are common to all rows, their rounding must not bias draw
                const double xh = ((double)rv[q] - (double)mean[c + q]) * (double)invstd[c + q];
                o[q] = (float)((double)coef[c + q] * ((double)gv[q] - m1[c + q] - xh * m2[c + q]));
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = coef ? coef[c + q] * gv[q] : gv[q];
        }
        if (draw) st_stream<NT>(draw + r * lddraw + c, make_float4(o[0], o[1], o[2], o[3]));
    }
}

// ---- BatchNorm backward of a layer whose activation went straight into MaxPool2d(2,2) (k_maxpool2_affine) -----------------
// One thread = one 2x2 cell (b, i, j) x one channel quad; the cell's activated values are recomputed from raw, the pooled gradient
// goes to the first maximum, every other element has dpre = 0.  Cells of an odd last row / column have no pooled gradient at all.
// Replaces pool-backward + reduce (4 activation passes) by 1.25 and pool-backward + apply (3) by 2.25.
__device__ __forceinline__ void pool_cell_dpre(const float4 v[4], const bool ok[4], bool pooled, const float4 g, const float sc[4],
                                               const float sf[4], int act, float slope, float dpre[4][4]) {
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        float e[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float t = ((const float *)&v[q])[ch] * sc[ch] + sf[ch];
            e[q] = act == 1 ? (t > 0.f ? t : 0.f) : (act == 2 ? (t > 0.f ? t : t * slope) : t);
        }
        int best = 0;
#pragma unroll
        for (int q = 1; q < 4; ++q) if (e[q] > e[best]) best = q;
        const float gg = ((const float *)&g)[ch];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // d act / d pre at the winning element, from its pre-activation sign (as dact() does on y = act(pre))
            const float t = ((const float *)&v[q])[ch] * sc[ch] + sf[ch];
            const float da = act == 1 ? (t > 0.f ? 1.f : 0.f) : (act == 2 ? (t > 0.f ? 1.f : slope) : 1.f);
            dpre[q][ch] = (pooled && q == best && ok[q]) ? gg * da : 0.f;
        }
    }
}

template <bool NT>
__global__ void __launch_bounds__(TPB)
k_pool_bn_bwd_reduce(const float *__restrict__ dyp, const float *__restrict__ raw, const float *__restrict__ mean,
                     const float *__restrict__ invstd, const float *__restrict__ pscale, const float *__restrict__ pshift,
                     int B, int H, int W, int C, int act, float slope, int cells_per_block, int CL, double *__restrict__ part) {
    __shared__ double s1s[TPB][4], s2s[TPB][4];
    const int Hc = (H + 1) / 2, Wc = (W + 1) / 2, Ho = H / 2, Wo = W / 2;
    const long long ncell = (long long)B * Hc * Wc;
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL, RL = TPB / CL;
    const int c = (blockIdx.x * CL + cl) * 4;
    long long r0 = (long long)blockIdx.y * cells_per_block, r1 = r0 + cells_per_block;
    if (r1 > ncell) r1 = ncell;
    double s1[4] = {0., 0., 0., 0.}, s2[4] = {0., 0., 0., 0.};
    if (c < C) {
        const float4 mu = *reinterpret_cast<const float4 *>(mean + c), is = *reinterpret_cast<const float4 *>(invstd + c);
        const float4 psc = *reinterpret_cast<const float4 *>(pscale + c), psh = *reinterpret_cast<const float4 *>(pshift + c);
        const float sc[4] = {psc.x, psc.y, psc.z, psc.w}, sf[4] = {psh.x, psh.y, psh.z, psh.w};
        const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, isv[4] = {is.x, is.y, is.z, is.w};
        for (long long r = r0 + rl; r < r1; r += RL) {
            const int j = (int)(r % Wc); const long long t = r / Wc;
            const int i = (int)(t % Hc); const long long b = t / Hc;
            if (i >= Ho || j >= Wo) continue;                       // no pooled gradient: dpre = 0 everywhere in the cell
            const long long base = (((b * H + 2 * i) * W) + 2 * j) * (long long)C + c;
            const long long offs[4] = {0, C, (long long)W * C, (long long)W * C + C};
            float4 v[4];
            const bool ok[4] = {true, true, true, true};
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = ld_stream<NT>(raw + base + offs[q]);
            const float4 g = *reinterpret_cast<const float4 *>(dyp + (((b * Ho + i) * Wo) + j) * (long long)C + c);
            float dpre[4][4];
            pool_cell_dpre(v, ok, true, g, sc, sf, act, slope, dpre);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) {
                    const float d = dpre[q][ch];
                    s1[ch] += (double)d;
                    s2[ch] += (double)d * (double)((((const float *)&v[q])[ch] - muv[ch]) * isv[ch]);
                }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { s1s[threadIdx.x][q] = s1[q]; s2s[threadIdx.x][q] = s2[q]; }
    __syncthreads();
    if (rl != 0 || c >= C) return;
    for (int k = 1; k < RL; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) { s1[q] += s1s[k * CL + cl][q]; s2[q] += s2s[k * CL + cl][q]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        part[((long long)blockIdx.y * 2) * C + c + q] = s1[q];
        part[((long long)blockIdx.y * 2 + 1) * C + c + q] = s2[q];
    }
}

// ---- the same two sums from POOLED tensors only (round 6).  With a ReLU, dpre is non-zero only at the window's first maximum and only
// where the pooled activation y is positive, and there y = gamma*xhat + beta of that very element: xhat = (y - beta) / gamma with
// beta = mean*pscale + pshift, 1/gamma = invstd/pscale.  So  sum dpre = sum_{y>0} dy_pool  and  sum dpre*xhat = sum_{y>0} dy_pool *
// (y - beta)/gamma  need the pooled gradient and the pooled activation (which the next layer's backward keeps alive anyway) - a
// quarter of a read each - instead of dy_pool + the full-resolution raw map: 0.5 units of traffic against 1.25.  A channel with
// gamma == 0 (pscale == 0: y is constant, every window's first element wins) takes xhat from raw at that element, as before.
template <bool NT>
__global__ void __launch_bounds__(TPB)
k_pool_bn_bwd_reduce_y(const float *__restrict__ dyp, const float *__restrict__ y, const float *__restrict__ raw,
                       const float *__restrict__ mean, const float *__restrict__ invstd, const float *__restrict__ pscale,
                       const float *__restrict__ pshift, int H, int W, int C, long long Mp, int rows_per_block, int CL,
                       double *__restrict__ part) {
    __shared__ double s1s[TPB][4], s2s[TPB][4];
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL, RL = TPB / CL;
    const int c = (blockIdx.x * CL + cl) * 4;
    const int Ho = H >> 1, Wo = W >> 1;
    long long r0 = (long long)blockIdx.y * rows_per_block, r1 = r0 + rows_per_block;
    if (r1 > Mp) r1 = Mp;
    double s1[4] = {0., 0., 0., 0.}, s2[4] = {0., 0., 0., 0.};
    if (c < C) {
        const float4 mu = *reinterpret_cast<const float4 *>(mean + c), is = *reinterpret_cast<const float4 *>(invstd + c);
        const float4 psc = *reinterpret_cast<const float4 *>(pscale + c), psh = *reinterpret_cast<const float4 *>(pshift + c);
        const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, isv[4] = {is.x, is.y, is.z, is.w};
        const float scv[4] = {psc.x, psc.y, psc.z, psc.w}, shv[4] = {psh.x, psh.y, psh.z, psh.w};
        float bet[4], rg[4];
        bool flat = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bet[q] = fmaf(muv[q], scv[q], shv[q]);
            rg[q] = scv[q] != 0.f ? isv[q] / scv[q] : 0.f;
            flat |= scv[q] == 0.f;
        }
        for (long long r = r0 + rl; r < r1; r += RL) {
            const float4 g4 = ld_stream<NT>(dyp + r * C + c), y4 = ld_stream<NT>(y + r * C + c);
            const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, yv[4] = {y4.x, y4.y, y4.z, y4.w};
            float rw0[4] = {0.f, 0.f, 0.f, 0.f};
            if (flat) {                              // (a channel of this quad has gamma == 0: xhat of the window's first element from raw)
                const int j = (int)(r % Wo); const long long t = r / Wo;
                const int i = (int)(t % Ho); const long long b = t / Ho;
                const float4 w4 = *reinterpret_cast<const float4 *>(raw + (((b * H + 2 * i) * W) + 2 * j) * (long long)C + c);
                rw0[0] = w4.x; rw0[1] = w4.y; rw0[2] = w4.z; rw0[3] = w4.w;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float d = yv[q] > 0.f ? gv[q] : 0.f;
                const float xh = scv[q] != 0.f ? (yv[q] - bet[q]) * rg[q] : (rw0[q] - muv[q]) * isv[q];
                s1[q] += (double)d;
                s2[q] += (double)d * (double)xh;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { s1s[threadIdx.x][q] = s1[q]; s2s[threadIdx.x][q] = s2[q]; }
    __syncthreads();
    if (rl != 0 || c >= C) return;
    for (int k = 1; k < RL; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) { s1[q] += s1s[k * CL + cl][q]; s2[q] += s2s[k * CL + cl][q]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        part[((long long)blockIdx.y * 2) * C + c + q] = s1[q];
        part[((long long)blockIdx.y * 2 + 1) * C + c + q] = s2[q];
    }
}

template <bool NT>
__global__ void __launch_bounds__(TPB)
k_pool_bn_bwd_apply(const float *__restrict__ dyp, const float *__restrict__ raw, const float *__restrict__ mean,
                    const float *__restrict__ invstd, const float *__restrict__ coef, const double *__restrict__ m1,
                    const double *__restrict__ m2, const float *__restrict__ pscale, const float *__restrict__ pshift, int B,
                    int H, int W, int C, int act, float slope, float *__restrict__ draw) {
    const int Hc = (H + 1) / 2, Wc = (W + 1) / 2, Ho = H / 2, Wo = W / 2, c4n = C >> 2;
    const long long total = (long long)B * Hc * Wc * c4n;
    for (long long idx = (long long)blockIdx.x * TPB + threadIdx.x; idx < total; idx += (long long)gridDim.x * TPB) {
        const int c = (int)(idx % c4n) * 4; long long r = idx / c4n;
        const int j = (int)(r % Wc); r /= Wc;
        const int i = (int)(r % Hc); const long long b = r / Hc;
        const bool pooled = i < Ho && j < Wo;
        const long long base = (((b * H + 2 * i) * W) + 2 * j) * (long long)C + c;
        const long long offs[4] = {0, C, (long long)W * C, (long long)W * C + C};
        const bool ok[4] = {true, 2 * j + 1 < W, 2 * i + 1 < H, (2 * j + 1 < W) && (2 * i + 1 < H)};
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = ok[q] ? ld_stream<NT>(raw + base + offs[q]) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pooled) g = *reinterpret_cast<const float4 *>(dyp + (((b * Ho + i) * Wo) + j) * (long long)C + c);
        const float4 psc = *reinterpret_cast<const float4 *>(pscale + c), psh = *reinterpret_cast<const float4 *>(pshift + c);
        const float sc[4] = {psc.x, psc.y, psc.z, psc.w}, sf[4] = {psh.x, psh.y, psh.z, psh.w};
        float dpre[4][4];
        pool_cell_dpre(v, ok, pooled, g, sc, sf, act, slope, dpre);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!ok[q]) continue;
            float o[4];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                const double xh = ((double)((const float *)&v[q])[ch] - (double)mean[c + ch]) * (double)invstd[c + ch];
                o[ch] = (float)((double)coef[c + ch] * ((double)dpre[q][ch] - m1[c + ch] - xh * m2[c + ch]));
            }
            st_stream<NT>(draw + base + offs[q], make_float4(o[0], o[1], o[2], o[3]));
        }
    }
}

// backward of the fused BatchNorm + activation + max pool (elementwise.hip:k_maxpool2_affine): the four activated values of a
// window are recomputed from the raw conv output, the gradient goes to the first maximal one (scan order), the others get 0
__global__ void __launch_bounds__(TPB)
k_maxpool2_bwd_affine(const float *__restrict__ x, const float *__restrict__ scale, const float *__restrict__ shift, int act,
                      float slope, const float *__restrict__ dy, float *__restrict__ dx, int B, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2, c4 = C >> 2;
    long long total = (long long)B * Ho * Wo * c4;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        int c = (int)(i % c4) * 4; long long r = i / c4;
        int ow = (int)(r % Wo); r /= Wo;
        int oh = (int)(r % Ho); long long b = r / Ho;
        long long base = (((b * H + oh * 2) * W) + ow * 2) * (long long)C + c;
        const long long offs[4] = {0, C, (long long)W * C, (long long)W * C + C};
        const float4 sc = *reinterpret_cast<const float4 *>(scale + c), sf = *reinterpret_cast<const float4 *>(shift + c);
        const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, sfv[4] = {sf.x, sf.y, sf.z, sf.w};
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4 *>(x + base + offs[q]);
        float4 g = *reinterpret_cast<const float4 *>(dy + (((b * Ho + oh) * Wo) + ow) * (long long)C + c);
        float o[4][4];
        const float gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            float e[4] = {((float *)&v[0])[ch], ((float *)&v[1])[ch], ((float *)&v[2])[ch], ((float *)&v[3])[ch]};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float t = e[q] * scv[ch] + sfv[ch];
                e[q] = act == 1 ? (t > 0.f ? t : 0.f) : (act == 2 ? (t > 0.f ? t : t * slope) : t);
            }
            int best = 0;
#pragma unroll
            for (int q = 1; q < 4; ++q) if (e[q] > e[best]) best = q;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q][ch] = (q == best) ? gg[ch] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4 *>(dx + base + offs[q]) = make_float4(o[q][0], o[q][1], o[q][2], o[q][3]);
    }
}

// MaxPool2d(2,2) backward: gradient goes to the first maximal element in (h,w) scan order
__global__ void __launch_bounds__(TPB)
k_maxpool2_bwd(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ dx, int B, int H,
               int W, int C) {
    const int Ho = H / 2, Wo = W / 2, c4 = C >> 2;
    long long total = (long long)B * Ho * Wo * c4;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        int c = (int)(i % c4) * 4; long long r = i / c4;
        int ow = (int)(r % Wo); r /= Wo;
        int oh = (int)(r % Ho); long long b = r / Ho;
        long long base = (((b * H + oh * 2) * W) + ow * 2) * (long long)C + c;
        const long long offs[4] = {0, C, (long long)W * C, (long long)W * C + C};
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4 *>(x + base + offs[q]);
        float4 g = *reinterpret_cast<const float4 *>(dy + (((b * Ho + oh) * Wo) + ow) * (long long)C + c);
        float o[4][4];
        const float gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            float e[4] = {((float *)&v[0])[ch], ((float *)&v[1])[ch], ((float *)&v[2])[ch], ((float *)&v[3])[ch]};
            int best = 0;
#pragma unroll
            for (int q = 1; q < 4; ++q) if (e[q] > e[best]) best = q;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q][ch] = (q == best) ? gg[ch] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4 *>(dx + base + offs[q]) = make_float4(o[q][0], o[q][1], o[q][2], o[q][3]);
    }
    // odd trailing rows / columns (floor pooling) receive no gradient: caller zero-fills dx when H or W is odd
}

// dx[argrow[s][c]][c] = dy[s][c], dx pre-zeroed
__global__ void k_segment_colmax_bwd(const float *__restrict__ dy, const int *__restrict__ argrow, int nseg, int C,
                                     float *__restrict__ dx, long long ld) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nseg * C) return;
    int c = i % C;
    dx[(long long)argrow[i] * ld + c] = dy[i];
}

// dx[s*P + r][c] = dy[s][c] / P
__global__ void __launch_bounds__(TPB)
k_segment_colmean_bwd(const float *__restrict__ dy, int P, int nseg, int C, float *__restrict__ dx, long long ld) {
    long long total = (long long)nseg * P * C;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        int c = (int)(i % C); long long r = i / C;
        int s = (int)(r / P);
        dx[r * ld + c] = dy[(long long)s * C + c] / (float)P;
    }
}

// softmax over 2 channels: y planar (B,2,HW), dy planar -> dx [B*HW][ld] channels 0,1 (others 0)
__global__ void __launch_bounds__(TPB)
k_softmax2_bwd(const float *__restrict__ y, const float *__restrict__ dy, int B, long long HW,
               float *__restrict__ dx, long long ld) {
    long long total = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        long long b = i / HW, p = i - b * HW;
        float y0 = y[(b * 2) * HW + p], y1 = y[(b * 2 + 1) * HW + p];
        float g0 = dy[(b * 2) * HW + p], g1 = dy[(b * 2 + 1) * HW + p];
        float dot = g0 * y0 + g1 * y1;
        for (int c = 0; c < (int)ld; ++c) dx[i * ld + c] = 0.f;
        dx[i * ld] = y0 * (g0 - dot);
        dx[i * ld + 1] = y1 * (g1 - dot);
    }
}

// backward of k_heads_to_nchw: dx[pixel] = (d depth, y0 (g0 - dot), y1 (g1 - dot), 0) as one 16-byte store; either gradient may be absent
__global__ void __launch_bounds__(TPB)
k_heads_bwd(const float *__restrict__ y, const float *__restrict__ dmask, const float *__restrict__ ddepth, int B, long long HW,
            float4 *__restrict__ dx) {
    long long total = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        long long b = i / HW, p = i - b * HW;
        float4 o = make_float4(ddepth ? ddepth[i] : 0.f, 0.f, 0.f, 0.f);
        if (dmask) {
            float y0 = y[(b * 2) * HW + p], y1 = y[(b * 2 + 1) * HW + p];
            float g0 = dmask[(b * 2) * HW + p], g1 = dmask[(b * 2 + 1) * HW + p];
            float dot = g0 * y0 + g1 * y1;
            o.y = y0 * (g0 - dot);
            o.z = y1 * (g1 - dot);
        }
        dx[i] = o;
    }
}

// ---- correlation backward -------------------------------------------------------------------------
// dcam_n[b][y][x][c] = sum_j dl[b][j] * rp[b][y][j+x][c]
__global__ void __launch_bounds__(TPB)
k_corr_bwd_cam(const float *__restrict__ rp, const float *__restrict__ dl, int h, int wc, int wp, int nj,
               float *__restrict__ dcam) {
    const int b = blockIdx.z, y = blockIdx.y;
    int i = blockIdx.x * TPB + threadIdx.x;           // over wc*4 float4 slots of the row
    if (i >= wc * 4) return;
    const float4 *r = reinterpret_cast<const float4 *>(rp + (((long long)b * h + y) * wp) * 16) + i;
    const float *g = dl + (long long)b * nj;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < nj; ++j) {
        float w = g[j]; float4 v = r[(long long)j * 4];
        a.x += w * v.x; a.y += w * v.y; a.z += w * v.z; a.w += w * v.w;
    }
    reinterpret_cast<float4 *>(dcam + (((long long)b * h + y) * wc) * 16)[i] = a;
}

// drp[b][y][xp][c] = sum_x dl[b][xp-x] * cam_n[b][y][x][c]   (0 <= xp-x < nj)
// block = one (b, y): the normalised camera row and dl live in LDS; thread = one (xp, channel quad)
__global__ void __launch_bounds__(TPB)
k_corr_bwd_rng(const float *__restrict__ cam, const float *__restrict__ cam_mm, const float *__restrict__ dl,
               int h, int wc, int wp, int nj, float *__restrict__ drp) {
    extern __shared__ __attribute__((aligned(16))) float sm[];        // wc*16 floats + nj floats
    float4 *srow = reinterpret_cast<float4 *>(sm);
    float *sdl = sm + (size_t)wc * 16;
    const int b = blockIdx.z, y = blockIdx.y;
    const float d = cam_mm[b * 2 + 1] - cam_mm[b * 2];
    const float4 *cr = reinterpret_cast<const float4 *>(cam + (((long long)b * h + y) * wc) * 16);
    for (int i = threadIdx.x; i < wc * 4; i += TPB) {
        float4 v = cr[i];
        v.x /= d; v.y /= d; v.z /= d; v.w /= d;
        srow[i] = v;
    }
    for (int i = threadIdx.x; i < nj; i += TPB) sdl[i] = dl[(long long)b * nj + i];
    __syncthreads();
    for (int i = blockIdx.x * TPB + threadIdx.x; i < wp * 4; i += gridDim.x * TPB) {
        int xp = i >> 2, cq = i & 3;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        int x0 = xp - (nj - 1); if (x0 < 0) x0 = 0;
        int x1 = xp < wc - 1 ? xp : wc - 1;
        for (int x = x0; x <= x1; ++x) {
            float w = sdl[xp - x]; float4 v = srow[x * 4 + cq];
            a.x += w * v.x; a.y += w * v.y; a.z += w * v.z; a.w += w * v.w;
        }
        reinterpret_cast<float4 *>(drp + (((long long)b * h + y) * wp) * 16)[i] = a;
    }
}

// fold the padded gradient back: drng_n[b][y][xs][c] = drp[xs+off] (+ drp[w-1-xs] if w-1-xs < off) (+ drp[xs+off+w] if xs < off)
__global__ void __launch_bounds__(TPB)
k_corr_unpad(const float *__restrict__ drp, int B, int h, int w, int C, int off, float *__restrict__ dx) {
    const int wp = w + 2 * off, c4n = C >> 2;
    long long total = (long long)B * h * w * c4n;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        int cq = (int)(i % c4n); long long r = i / c4n;
        int xs = (int)(r % w); r /= w;                 // r = b*h + y
        const float4 *row = reinterpret_cast<const float4 *>(drp + (r * wp) * C) ;
        float4 a = row[(long long)(xs + off) * c4n + cq];
        int xm = w - 1 - xs;
        if (xm < off) { float4 v = row[(long long)xm * c4n + cq]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
        if (xs < off) { float4 v = row[(long long)(xs + off + w) * c4n + cq]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
        reinterpret_cast<float4 *>(dx)[i] = a;
    }
}

int grid_for(long long total) {
    long long g = (total + TPB - 1) / TPB;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}
}  // namespace

// rows per partial-sum block: M / 1024 (the finalize pass folds at most 1 024 partial rows), at least 16.  (Until round 5 at least
// 512: a batch-1 layer of 5 700 positions x 512 channels then ran as 24 workgroups whose threads walked 128 rows each - 50 us for
// 23 MB; the reference's own configuration is batch 1.)
#ifndef EFGH_BWD_ROW_GRAN
#define EFGH_BWD_ROW_GRAN 16
#endif
static long long bwd_rows_per_block(long long M) {
    // partial rows a reduction leaves for k_bwd_finalize (4096 until round 4: the fold of 4 096 rows cost 14 us per layer;
    // tools/bench_elementwise.py: 0.358 -> 0.326 ms at 1 GB)
    // (multiples of 16 rows, at least 16 - until the end of round 5 multiples of 64: the batch-1 layers of the reference's own
    // configuration, 1 400-22 600 positions, then ran as 22-350 workgroups walking 16 dependent trips each, 60-70 us for 11 MB)
    const long long groups = 1024;
    long long rows = (M + groups - 1) / groups;
    rows = (rows + EFGH_BWD_ROW_GRAN - 1) / EFGH_BWD_ROW_GRAN * EFGH_BWD_ROW_GRAN;
    return rows < EFGH_BWD_ROW_GRAN ? EFGH_BWD_ROW_GRAN : rows;
}

extern "C" int32_t efgh_bwd_groups(int64_t M) {
    const long long rows = bwd_rows_per_block(M);
    return (int32_t)((M + rows - 1) / rows);
}

extern "C" int efgh_act_bn_bwd_reduce(const float *dy, int64_t lddy, const float *y, int64_t ldy, const float *raw,
                                      int64_t ldraw, const float *mean, const float *invstd, const float *pscale,
                                      const float *pshift, int64_t M, int32_t C,
                                      int32_t act, float slope, double *part, float *sum_dpre, float *sum_dpre_xhat,
                                      double *mean_dpre, double *mean_dpre_xhat, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(dy && part && sum_dpre && sum_dpre_xhat && M > 0 && C > 0 && C % 4 == 0);
    EFGH_CHECK_ARG(y || (raw && pscale && pshift));
    EFGH_CHECK_ARG(!mean || (raw && invstd));
    EFGH_CHECK_ARG(lddy % 4 == 0 && ldy % 4 == 0 && (!mean || ldraw % 4 == 0));
    const unsigned *ybits = nullptr;
    if (y && ldy == 0) {                 // y = the sign bits of the activation (efgh_scale_shift_act_bits)
        EFGH_CHECK_ARG(C % 32 == 0);
        ybits = (const unsigned *)y; y = nullptr;
    }
    int G = efgh_bwd_groups(M);
    int CL = 1;
    while (CL < 64 && CL * 4 < C) CL <<= 1;
    if (efgh_stream_nt(M * C * 4ll))
        k_act_bn_bwd_reduce<true><<<dim3(cdiv(C / 4, CL), G), TPB, 0, st>>>(dy, lddy, y, ldy, raw, ldraw, mean, invstd, pscale, pshift, M,
                                                                           C, act, slope, (int)bwd_rows_per_block(M), CL, part, ybits);
    else
        k_act_bn_bwd_reduce<false><<<dim3(cdiv(C / 4, CL), G), TPB, 0, st>>>(dy, lddy, y, ldy, raw, ldraw, mean, invstd, pscale, pshift, M,
                                                                            C, act, slope, (int)bwd_rows_per_block(M), CL, part, ybits);
    k_bwd_finalize<<<cdiv(C, 32), dim3(32, 32), 0, st>>>(part, G, C, (double)M, sum_dpre, sum_dpre_xhat, mean_dpre,
                                                         mean_dpre_xhat);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

// the same fold over float rows [G][2][C] (the BatchNorm-backward sums an MFMA kernel left per row block, efgh_gemm_desc.stats_mode)
__global__ void __launch_bounds__(1024)
k_bwd_finalize_f32(const float *__restrict__ part, int G, int C, double count, float *__restrict__ sum_dpre,
                   float *__restrict__ sum_dpre_xhat, double *__restrict__ mean_dpre, double *__restrict__ mean_dpre_xhat) {
    __shared__ double sa[32][33], sb[32][33];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int c = blockIdx.x * 32 + tx;
    double a = 0.0, b = 0.0;
    if (c < C) {
        double a1 = 0.0, b1 = 0.0, a2 = 0.0, b2 = 0.0, a3 = 0.0, b3 = 0.0;
        int g = ty;
        for (; g + 96 < G; g += 128) {
            a += (double)part[((long long)g * 2) * C + c]; b += (double)part[((long long)g * 2 + 1) * C + c];
            a1 += (double)part[((long long)(g + 32) * 2) * C + c]; b1 += (double)part[((long long)(g + 32) * 2 + 1) * C + c];
            a2 += (double)part[((long long)(g + 64) * 2) * C + c]; b2 += (double)part[((long long)(g + 64) * 2 + 1) * C + c];
            a3 += (double)part[((long long)(g + 96) * 2) * C + c]; b3 += (double)part[((long long)(g + 96) * 2 + 1) * C + c];
        }
        for (; g < G; g += 32) { a += (double)part[((long long)g * 2) * C + c]; b += (double)part[((long long)g * 2 + 1) * C + c]; }
        a += (a1 + a2) + a3; b += (b1 + b2) + b3;
    }
    sa[ty][tx] = a; sb[ty][tx] = b;
    __syncthreads();
    if (ty != 0 || c >= C) return;
    for (int i = 1; i < 32; ++i) { a += sa[i][tx]; b += sb[i][tx]; }
    sum_dpre[c] = (float)a; sum_dpre_xhat[c] = (float)b;
    mean_dpre[c] = a / count; mean_dpre_xhat[c] = b / count;
}

extern "C" int efgh_bwd_finalize_f32(const float *stats, int32_t rows, int32_t C, double count, float *sum_dpre,
                                     float *sum_dpre_xhat, double *mean_dpre, double *mean_dpre_xhat, void *stream_) {
    EFGH_CHECK_ARG(stats && rows > 0 && C > 0 && count > 0 && sum_dpre && sum_dpre_xhat && mean_dpre && mean_dpre_xhat);
    k_bwd_finalize_f32<<<cdiv(C, 32), dim3(32, 32), 0, (hipStream_t)stream_>>>(stats, rows, C, count, sum_dpre, sum_dpre_xhat,
                                                                               mean_dpre, mean_dpre_xhat);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_act_bn_bwd_apply(const float *dy, int64_t lddy, const float *y, int64_t ldy, const float *raw,
                                     int64_t ldraw, const float *mean, const float *invstd, const float *coef,
                                     const double *m1, const double *m2, const float *pscale, const float *pshift,
                                     int64_t M, int32_t C, int32_t act, float slope,
                                     float *draw, int64_t lddraw, float *dres, int64_t lddres, void *stream_) {
    EFGH_CHECK_ARG(dy && (draw || dres) && M > 0 && C > 0 && C % 4 == 0);
    EFGH_CHECK_ARG(y || (raw && pscale && pshift && ldraw % 4 == 0));
    EFGH_CHECK_ARG(lddy % 4 == 0 && ldy % 4 == 0 && (!draw || lddraw % 4 == 0) && (!dres || lddres % 4 == 0));
    EFGH_CHECK_ARG(!mean || (raw && invstd && coef && m1 && m2 && ldraw % 4 == 0));
    const unsigned *ybits = nullptr;
    if (y && ldy == 0) {                 // y = the sign bits of the activation (efgh_scale_shift_act_bits)
        EFGH_CHECK_ARG(C % 32 == 0);
        ybits = (const unsigned *)y; y = nullptr;
    }
    if (efgh_stream_nt(M * C * 4ll))
        k_act_bn_bwd_apply<true><<<grid_for(M * (C / 4)), TPB, 0, (hipStream_t)stream_>>>(
            dy, lddy, y, ldy, raw, ldraw, mean, invstd, coef, m1, m2, pscale, pshift, M, C, act, slope, draw, lddraw, dres, lddres, ybits);
    else
        k_act_bn_bwd_apply<false><<<grid_for(M * (C / 4)), TPB, 0, (hipStream_t)stream_>>>(
            dy, lddy, y, ldy, raw, ldraw, mean, invstd, coef, m1, m2, pscale, pshift, M, C, act, slope, draw, lddraw, dres, lddres, ybits);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_maxpool2_bwd(const float *x, const float *dy, float *dx, int32_t B, int32_t H, int32_t W,
                                 int32_t C, void *stream_) {
    EFGH_CHECK_ARG(x && dy && dx && B > 0 && H >= 2 && W >= 2 && C % 4 == 0);
    k_maxpool2_bwd<<<grid_for((long long)B * (H / 2) * (W / 2) * (C / 4)), TPB, 0, (hipStream_t)stream_>>>(x, dy, dx, B, H,
                                                                                                       W, C);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_maxpool2_bwd_affine(const float *x, const float *scale, const float *shift, int32_t act, float slope,
                                        const float *dy, float *dx, int32_t B, int32_t H, int32_t W, int32_t C, void *stream_) {
    EFGH_CHECK_ARG(x && scale && shift && dy && dx && B > 0 && H >= 2 && W >= 2 && C % 4 == 0);
    k_maxpool2_bwd_affine<<<grid_for((long long)B * (H / 2) * (W / 2) * (C / 4)), TPB, 0, (hipStream_t)stream_>>>(
        x, scale, shift, act, slope, dy, dx, B, H, W, C);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_segment_colmax_bwd(const float *dy, const int32_t *argrow, int32_t nseg, int32_t C, float *dx,
                                       int64_t ld, void *stream_) {
    EFGH_CHECK_ARG(dy && argrow && dx && nseg > 0 && C > 0);
    k_segment_colmax_bwd<<<cdiv((long long)nseg * C, 256), 256, 0, (hipStream_t)stream_>>>(dy, argrow, nseg, C, dx, ld);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_segment_colmean_bwd(const float *dy, int32_t P, int32_t nseg, int32_t C, float *dx, int64_t ld,
                                        void *stream_) {
    EFGH_CHECK_ARG(dy && dx && P > 0 && nseg > 0 && C > 0);
    k_segment_colmean_bwd<<<grid_for((long long)nseg * P * C), TPB, 0, (hipStream_t)stream_>>>(dy, P, nseg, C, dx, ld);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_heads_bwd(const float *y, const float *dmask, const float *ddepth, int32_t B, int64_t HW, float *dx,
                              void *stream_) {
    EFGH_CHECK_ARG(y && dx && B > 0 && HW > 0 && (((uintptr_t)dx) & 15) == 0);
    k_heads_bwd<<<grid_for((long long)B * HW), TPB, 0, (hipStream_t)stream_>>>(y, dmask, ddepth, B, HW, (float4 *)dx);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_softmax2_bwd(const float *y, const float *dy, int32_t B, int64_t HW, float *dx, int64_t ld,
                                 void *stream_) {
    EFGH_CHECK_ARG(y && dy && dx && B > 0 && HW > 0 && ld >= 2);
    k_softmax2_bwd<<<grid_for((long long)B * HW), TPB, 0, (hipStream_t)stream_>>>(y, dy, B, HW, dx, ld);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_corr1d_bwd(const float *rp, const float *cam, const float *cam_mm, const float *dlogit, int32_t B,
                               int32_t h, int32_t wc, int32_t wp, float *dcam_n, float *drp, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(rp && cam && cam_mm && dlogit && dcam_n && drp && B > 0 && h > 0 && wc > 0 && wp >= wc);
    int nj = wp - wc + 1;
    k_corr_bwd_cam<<<dim3(cdiv(wc * 4, TPB), h, B), TPB, 0, st>>>(rp, dlogit, h, wc, wp, nj, dcam_n);
    EFGH_CHECK_ARG((size_t)wc * 64 + (size_t)nj * 4 <= 64 * 1024);
    k_corr_bwd_rng<<<dim3(cdiv(wp * 4, TPB * 4), h, B), TPB, (size_t)wc * 64 + (size_t)nj * 4, st>>>(cam, cam_mm, dlogit, h, wc, wp, nj,
                                                                                              drp);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_corr_unpad(const float *drp, int32_t B, int32_t h, int32_t w, int32_t C, int32_t off, float *dx,
                               void *stream_) {
    EFGH_CHECK_ARG(drp && dx && B > 0 && h > 0 && w > 0 && C % 4 == 0 && off >= 0 && off <= w);
    k_corr_unpad<<<grid_for((long long)B * h * w * (C / 4)), TPB, 0, (hipStream_t)stream_>>>(drp, B, h, w, C, off, dx);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int32_t efgh_pool_bwd_groups(int32_t B, int32_t H, int32_t W) {
    const long long ncell = (long long)B * ((H + 1) / 2) * ((W + 1) / 2);
    const long long rows = bwd_rows_per_block(ncell);
    return (int32_t)((ncell + rows - 1) / rows);
}

extern "C" int efgh_pool_bn_bwd_reduce(const float *dy_pool, const float *raw, const float *mean, const float *invstd,
                                       const float *pscale, const float *pshift, int32_t B, int32_t H, int32_t W, int32_t C,
                                       int32_t act, float slope, double *part, float *sum_dpre, float *sum_dpre_xhat,
                                       double *mean_dpre, double *mean_dpre_xhat, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(dy_pool && raw && mean && invstd && pscale && pshift && part && sum_dpre && sum_dpre_xhat && mean_dpre &&
                   mean_dpre_xhat);
    EFGH_CHECK_ARG(B > 0 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0);
    const long long ncell = (long long)B * ((H + 1) / 2) * ((W + 1) / 2);
    const int G = efgh_pool_bwd_groups(B, H, W);
    int CL = 1;
    while (CL < 64 && CL * 4 < C) CL <<= 1;
    if (efgh_stream_nt((long long)B * H * W * C * 4)) k_pool_bn_bwd_reduce<true><<<dim3(cdiv(C / 4, CL), G), TPB, 0, st>>>(dy_pool, raw, mean, invstd, pscale, pshift, B, H, W, C, act,
                                                                  slope, (int)bwd_rows_per_block(ncell), CL, part);
    else k_pool_bn_bwd_reduce<false><<<dim3(cdiv(C / 4, CL), G), TPB, 0, st>>>(dy_pool, raw, mean, invstd, pscale, pshift, B, H, W, C, act,
                                                                  slope, (int)bwd_rows_per_block(ncell), CL, part);
    k_bwd_finalize<<<cdiv(C, 32), dim3(32, 32), 0, st>>>(part, G, C, (double)B * H * W, sum_dpre, sum_dpre_xhat, mean_dpre,
                                                         mean_dpre_xhat);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

/* the same sums from the POOLED gradient and the POOLED activation y [B][H/2][W/2][C] (ReLU layers; see k_pool_bn_bwd_reduce_y);
 * `part` has efgh_bwd_groups(B * (H/2) * (W/2)) rows */
extern "C" int efgh_pool_bn_bwd_reduce_pooled(const float *dy_pool, const float *y_pool, const float *raw, const float *mean,
                                              const float *invstd, const float *pscale, const float *pshift, int32_t B, int32_t H,
                                              int32_t W, int32_t C, double *part, float *sum_dpre, float *sum_dpre_xhat,
                                              double *mean_dpre, double *mean_dpre_xhat, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(dy_pool && y_pool && raw && mean && invstd && pscale && pshift && part && sum_dpre && sum_dpre_xhat && mean_dpre &&
                   mean_dpre_xhat);
    EFGH_CHECK_ARG(B > 0 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0);
    const long long Mp = (long long)B * (H / 2) * (W / 2);
    const int G = efgh_bwd_groups(Mp);
    int CL = 1;
    while (CL < 64 && CL * 4 < C) CL <<= 1;
    if (efgh_stream_nt(Mp * C * 4)) k_pool_bn_bwd_reduce_y<true><<<dim3(cdiv(C / 4, CL), G), TPB, 0, st>>>(dy_pool, y_pool, raw, mean, invstd, pscale, pshift, H, W, C, Mp,
                                                                  (int)bwd_rows_per_block(Mp), CL, part);
    else k_pool_bn_bwd_reduce_y<false><<<dim3(cdiv(C / 4, CL), G), TPB, 0, st>>>(dy_pool, y_pool, raw, mean, invstd, pscale, pshift, H, W, C, Mp,
                                                                  (int)bwd_rows_per_block(Mp), CL, part);
    k_bwd_finalize<<<cdiv(C, 32), dim3(32, 32), 0, st>>>(part, G, C, (double)B * H * W, sum_dpre, sum_dpre_xhat, mean_dpre,
                                                         mean_dpre_xhat);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_pool_bn_bwd_apply(const float *dy_pool, const float *raw, const float *mean, const float *invstd,
                                      const float *coef, const double *m1, const double *m2, const float *pscale,
                                      const float *pshift, int32_t B, int32_t H, int32_t W, int32_t C, int32_t act,
                                      float slope, float *draw, void *stream_) {
    EFGH_CHECK_ARG(dy_pool && raw && mean && invstd && coef && m1 && m2 && pscale && pshift && draw);
    EFGH_CHECK_ARG(B > 0 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0);
    const long long total = (long long)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
    if (efgh_stream_nt((long long)B * H * W * C * 4)) k_pool_bn_bwd_apply<true><<<grid_for(total), TPB, 0, (hipStream_t)stream_>>>(dy_pool, raw, mean, invstd, coef, m1, m2, pscale,
                                                                          pshift, B, H, W, C, act, slope, draw);
    else k_pool_bn_bwd_apply<false><<<grid_for(total), TPB, 0, (hipStream_t)stream_>>>(dy_pool, raw, mean, invstd, coef, m1, m2, pscale,
                                                                          pshift, B, H, W, C, act, slope, draw);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}
