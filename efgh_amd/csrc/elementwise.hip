// HBM-bound helper kernels around the gather-GEMM: BatchNorm statistics / apply, activation,
// residual add, 2x2 max-pool, layout changes at the model boundary, column reductions.
// All image tensors are channels-last [rows][ld] with the channel axis contiguous; every kernel
// moves 16 B per lane where the channel count allows it.
#include "common.h"

namespace {
constexpr int TPB = 256;

__device__ __forceinline__ float act_f(float v, int act, float slope) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v > 0.f ? v : v * slope;
    return v;
}

// stats: [G][2][C] per-block partial (sum, sumsq) from the GEMM epilogue.
// -> scale/shift for y = x*scale + shift, running stats update (nn.BatchNorm, momentum form).
// stage 1 for many partial rows: slice s sums rows g = s, s+S, s+2S, ... into row s (in place: row s is
// read only by the block that owns slice s, before it writes)
__global__ void __launch_bounds__(1024)
k_stats_fold(float *__restrict__ stats, int G, int C, int S) {
    __shared__ double ss[32][33], sq[32][33];
    const int tx = threadIdx.x, ty = threadIdx.y, sl = blockIdx.y;
    const int c = blockIdx.x * 32 + tx;
    double s = 0.0, q = 0.0;
    if (c < C)
        for (int g = sl + S * ty; g < G; g += S * 32) { s += stats[((long long)g * 2) * C + c]; q += stats[((long long)g * 2 + 1) * C + c]; }
    ss[ty][tx] = s; sq[ty][tx] = q;
    __syncthreads();
    if (ty != 0 || c >= C) return;
    for (int i = 1; i < 32; ++i) { s += ss[i][tx]; q += sq[i][tx]; }
    stats[((long long)sl * 2) * C + c] = (float)s;
    stats[((long long)sl * 2 + 1) * C + c] = (float)q;
}

// block = (32 channels, 32 partial-lanes): lanes stride over the G partial rows, LDS tree over lanes.
__global__ void __launch_bounds__(1024)
k_bn_finalize(const float *__restrict__ stats, int G, int C, double count,
              const float *__restrict__ gamma, const float *__restrict__ beta,
              float *__restrict__ rmean, float *__restrict__ rvar, float momentum, float eps,
              float *__restrict__ scale, float *__restrict__ shift,
              float *__restrict__ save_mean, float *__restrict__ save_invstd) {
    __shared__ double ss[32][33], sq[32][33];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int c = blockIdx.x * 32 + tx;
    double s = 0.0, q = 0.0;
    if (c < C)
        for (int g = ty; g < G; g += 32) { s += stats[((long long)g * 2) * C + c]; q += stats[((long long)g * 2 + 1) * C + c]; }
    ss[ty][tx] = s; sq[ty][tx] = q;
    __syncthreads();
    if (ty != 0 || c >= C) return;
    for (int i = 1; i < 32; ++i) { s += ss[i][tx]; q += sq[i][tx]; }
    double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0) var = 0;
    float invstd = (float)(1.0 / sqrt(var + (double)eps));
    float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
    if (save_mean) { save_mean[c] = (float)mean; save_invstd[c] = invstd; }
    if (rmean) {
        double unb = count > 1 ? var * count / (count - 1) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
    }
}

// per-column (sum, sumsq) partials of a [M][ld] matrix (used when the producer is not the GEMM).
// block = (CL float4 channel lanes) x (256/CL row lanes); rows_per_block rows per block.
__global__ void __launch_bounds__(TPB)
k_col_stats(const float *__restrict__ x, long long M, int C, long long ld, int rows_per_block, int CL,
            float *__restrict__ stats) {
    __shared__ float4 s1s[TPB], s2s[TPB];
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL, RL = TPB / CL;
    const int c = (blockIdx.x * CL + cl) * 4;
    long long r0 = (long long)blockIdx.y * rows_per_block, r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
    if (c < C)
        for (long long r = r0 + rl; r < r1; r += RL) {
            float4 v = *reinterpret_cast<const float4 *>(x + r * ld + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
        }
    s1s[threadIdx.x] = s; s2s[threadIdx.x] = q;
    __syncthreads();
    if (rl != 0 || c >= C) return;
    for (int i = 1; i < RL; ++i) {
        float4 a = s1s[i * CL + cl], b = s2s[i * CL + cl];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w; q.x += b.x; q.y += b.y; q.z += b.z; q.w += b.w;
    }
    *reinterpret_cast<float4 *>(&stats[((long long)blockIdx.y * 2) * C + c]) = s;
    *reinterpret_cast<float4 *>(&stats[((long long)blockIdx.y * 2 + 1) * C + c]) = q;
}

// y[r][c] = act(x[r][c]*scale[c] + shift[c] + res[r][c]); vector of 4 channels per thread.
// BITS: also leaves bit e = (y > 0) of element e = r*C + c in `bits` (C % 32 == 0: eight consecutive lanes hold one 32-bit word) - the
// activation mask the BatchNorm backward of a RESIDUAL layer needs, at 1/32 of the bytes of reading the activation back twice
template <bool BITS, bool NT>
__global__ void __launch_bounds__(TPB)
k_scale_shift_act(const float *__restrict__ x, long long ldx, const float *__restrict__ scale,
                  const float *__restrict__ shift, const float *__restrict__ res, long long ldr,
                  float *__restrict__ y, long long ldy, long long M, int C, int act, float slope, unsigned *__restrict__ bits) {
    const int c4 = C >> 2;
    long long total = M * c4;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        long long r = i / c4; int c = (int)(i - r * c4) * 4;
        float4 v = ld_stream<NT>(x + r * ldx + c);
        float4 s = scale ? *reinterpret_cast<const float4 *>(scale + c) : make_float4(1.f, 1.f, 1.f, 1.f);
        float4 h = shift ? *reinterpret_cast<const float4 *>(shift + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        v.x = v.x * s.x + h.x; v.y = v.y * s.y + h.y; v.z = v.z * s.z + h.z; v.w = v.w * s.w + h.w;
        if (res) {
            float4 q = *reinterpret_cast<const float4 *>(res + r * ldr + c);
            v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        v.x = act_f(v.x, act, slope); v.y = act_f(v.y, act, slope);
        v.z = act_f(v.z, act, slope); v.w = act_f(v.w, act, slope);
        st_stream<NT>(y + r * ldy + c, v);
        if (BITS) {
            // (total % 8 == 0 and the loop advances all lanes together: the eight lanes of a word are active together)
            unsigned w = ((v.x > 0.f) ? 1u : 0u) | ((v.y > 0.f) ? 2u : 0u) | ((v.z > 0.f) ? 4u : 0u) | ((v.w > 0.f) ? 8u : 0u);
            w <<= 4 * (threadIdx.x & 7);
            w |= __shfl_xor(w, 1); w |= __shfl_xor(w, 2); w |= __shfl_xor(w, 4);
            if ((threadIdx.x & 7) == 0) bits[i >> 3] = w;
        }
    }
}

// scalar-channel fallback (C % 4 != 0)
__global__ void __launch_bounds__(TPB)
k_scale_shift_act1(const float *__restrict__ x, long long ldx, const float *__restrict__ scale,
                   const float *__restrict__ shift, const float *__restrict__ res, long long ldr,
                   float *__restrict__ y, long long ldy, long long M, int C, int act, float slope) {
    long long total = M * C;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        long long r = i / C; int c = (int)(i - r * C);
        float v = x[r * ldx + c] * (scale ? scale[c] : 1.f) + (shift ? shift[c] : 0.f);
        if (res) v += res[r * ldr + c];
        y[r * ldy + c] = act_f(v, act, slope);
    }
}

// 2x2/2 max pool, [B][H][W][C] -> [B][H/2][W/2][C]  (nn.MaxPool2d(2,2): floor)
__global__ void __launch_bounds__(TPB)
k_maxpool2(const float *__restrict__ x, float *__restrict__ y, int B, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2, c4 = C >> 2;
    long long total = (long long)B * Ho * Wo * c4;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        int c = (int)(i % c4) * 4; long long r = i / c4;
        int ow = (int)(r % Wo); r /= Wo;
        int oh = (int)(r % Ho); long long b = r / Ho;
        const float *p = x + (((b * H + oh * 2) * W) + ow * 2) * (long long)C + c;
        float4 a = *reinterpret_cast<const float4 *>(p);
        float4 b4 = *reinterpret_cast<const float4 *>(p + C);
        float4 c4v = *reinterpret_cast<const float4 *>(p + (long long)W * C);
        float4 d = *reinterpret_cast<const float4 *>(p + (long long)W * C + C);
        float4 m;
        m.x = fmaxf(fmaxf(a.x, b4.x), fmaxf(c4v.x, d.x)); m.y = fmaxf(fmaxf(a.y, b4.y), fmaxf(c4v.y, d.y));
        m.z = fmaxf(fmaxf(a.z, b4.z), fmaxf(c4v.z, d.z)); m.w = fmaxf(fmaxf(a.w, b4.w), fmaxf(c4v.w, d.w));
        *reinterpret_cast<float4 *>(y + (((b * Ho + oh) * Wo) + ow) * (long long)C + c) = m;
    }
}

// the vertical half of a 2x2/2 max pool: [B][H][W][C] -> [B][H/2][W][C] (rows 2i, 2i+1; the horizontal half was taken by the producer,
// k_wino43<.., HPOOL>)
__global__ void __launch_bounds__(TPB)
k_maxpool_v2(const float4 *__restrict__ x, float4 *__restrict__ y, int B, int H, long long rowq) {        // rowq = W * C / 4
    const int Ho = H / 2;
    const long long total = (long long)B * Ho * rowq;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const long long q = i % rowq, r = i / rowq;
        const int oh = (int)(r % Ho); const long long b = r / Ho;
        const float4 a = x[((b * H + 2 * oh) * rowq) + q], c = x[((b * H + 2 * oh + 1) * rowq) + q];
        y[i] = make_float4(fmaxf(a.x, c.x), fmaxf(a.y, c.y), fmaxf(a.z, c.z), fmaxf(a.w, c.w));
    }
}

// BatchNorm + activation + 2x2/2 max pool in one pass over the raw conv output (training path of the VGG trunks):
// y[b][oh][ow][c] = max over the window of act(raw*scale[c] + shift[c]); the full-resolution activation is never stored.
template <bool NT>
__global__ void __launch_bounds__(TPB)
k_maxpool2_affine(const float *__restrict__ x, const float *__restrict__ scale, const float *__restrict__ shift, int act,
                  float slope, float *__restrict__ y, int B, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2, c4 = C >> 2;
    long long total = (long long)B * Ho * Wo * c4;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        int c = (int)(i % c4) * 4; long long r = i / c4;
        int ow = (int)(r % Wo); r /= Wo;
        int oh = (int)(r % Ho); long long b = r / Ho;
        const float *p = x + (((b * H + oh * 2) * W) + ow * 2) * (long long)C + c;
        const float4 sc = *reinterpret_cast<const float4 *>(scale + c), sf = *reinterpret_cast<const float4 *>(shift + c);
        const long long offs[4] = {0, C, (long long)W * C, (long long)W * C + C};
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = ld_stream<NT>(p + offs[q]);
            m.x = fmaxf(m.x, act_f(v.x * sc.x + sf.x, act, slope)); m.y = fmaxf(m.y, act_f(v.y * sc.y + sf.y, act, slope));
            m.z = fmaxf(m.z, act_f(v.z * sc.z + sf.z, act, slope)); m.w = fmaxf(m.w, act_f(v.w * sc.w + sf.w, act, slope));
        }
        *reinterpret_cast<float4 *>(y + (((b * Ho + oh) * Wo) + ow) * (long long)C + c) = m;
    }
}

// (B,Cs,H,W) planar -> [B][H][W][Cd] channels-last, Cd >= Cs, extra channels zero
__global__ void __launch_bounds__(TPB)
k_nchw_to_nhwc(const float *__restrict__ x, float *__restrict__ y, int B, int Cs, long long HW, int Cd) {
    long long total = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        long long b = i / HW, p = i - b * HW;
        for (int c = 0; c < Cd; ++c)
            y[i * Cd + c] = c < Cs ? x[(b * Cs + c) * HW + p] : 0.f;
    }
}

// [B][H][W][ld] (first Cs channels) -> (B,Cs,H,W) planar
__global__ void __launch_bounds__(TPB)
k_nhwc_to_nchw(const float *__restrict__ x, long long ld, float *__restrict__ y, int B, int Cs, long long HW) {
    long long total = (long long)B * Cs * HW;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        long long p = i % HW; long long r = i / HW;
        int c = (int)(r % Cs); long long b = r / Cs;
        y[i] = x[(b * HW + p) * ld + c];
    }
}

// per-segment column max: x [M][ld], segment s = rows [seg[s], seg[s+1])  -> y[s][C]
__global__ void __launch_bounds__(TPB)
k_segment_colmax(const float *__restrict__ x, long long ld, int C, const int *__restrict__ seg,
                 float *__restrict__ y, int *__restrict__ argrow) {
    int s = blockIdx.y, c = blockIdx.x * TPB + threadIdx.x;
    if (c >= C) return;
    int r0 = seg[s], r1 = seg[s + 1];
    float m = -INFINITY; int am = r0;
    for (int r = r0; r < r1; ++r) { float v = x[(long long)r * ld + c]; if (v > m) { m = v; am = r; } }
    y[(long long)s * C + c] = m;
    if (argrow) argrow[(long long)s * C + c] = am;
}

// per-segment column mean: block = (32 columns) x (8 row lanes), tree over the row lanes
__global__ void __launch_bounds__(TPB)
k_segment_colmean(const float *__restrict__ x, long long ld, int C, int rows_per_seg, float *__restrict__ y) {
    __shared__ double sm[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int s = blockIdx.y, c = blockIdx.x * 32 + tx;
    double a = 0.0;
    if (c < C)
        for (int r = ty; r < rows_per_seg; r += 8) a += x[((long long)s * rows_per_seg + r) * ld + c];
    sm[ty][tx] = a;
    __syncthreads();
    if (ty != 0 || c >= C) return;
    for (int i = 1; i < 8; ++i) a += sm[i][tx];
    y[(long long)s * C + c] = (float)(a / rows_per_seg);
}

// ---- the same two reductions in two stages (round 5): the one-stage kernels above give a segment ONE workgroup per 256 (32)
// columns - 4-8 workgroups on 256 CUs for the network's shapes (E / H: 128 columns x 31 k vertices per sample; G's translation head:
// 3 columns x 7 680 positions), each thread walking its whole segment: 224 us for 64 MB.  Stage 1: RS row slices per segment, block =
// CL column lanes x 256 / CL row lanes, partial (max, first row) / double sums per slice; stage 2 folds the slices IN ORDER
// (first maximum wins, as torch.max; the sums in a fixed order).  Same results as the one-stage kernels (the mean may differ in
// its last bit: another summation order of the same doubles).
__global__ void __launch_bounds__(TPB)
k_segment_colmax_s1(const float *__restrict__ x, long long ld, int C, const int *__restrict__ seg, int RS, int CL,
                    float *__restrict__ pmax, int *__restrict__ prow) {
    __shared__ float sm[TPB]; __shared__ int sr[TPB];
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL, RL = TPB / CL;
    const int c = blockIdx.x * CL + cl, s = blockIdx.y, k = blockIdx.z;
    const int r0 = seg[s], len = seg[s + 1] - r0;
    const int a = r0 + (int)((long long)len * k / RS), b = r0 + (int)((long long)len * (k + 1) / RS);
    float m = -INFINITY; int am = a;
    if (c < C)
        for (int r = a + rl; r < b; r += RL) { const float v = x[(long long)r * ld + c]; if (v > m) { m = v; am = r; } }
    sm[threadIdx.x] = m; sr[threadIdx.x] = am;
    __syncthreads();
    if (rl != 0 || c >= C) return;
    for (int i = 1; i < RL; ++i) {                   // row lanes interleave the rows: a tie goes to the smaller row index
        const float v = sm[i * CL + cl]; const int r = sr[i * CL + cl];
        if (v > m || (v == m && r < am)) { m = v; am = r; }
    }
    if (b <= a) { m = -INFINITY; am = a; }
    pmax[((long long)s * RS + k) * C + c] = m; prow[((long long)s * RS + k) * C + c] = am;
}

__global__ void __launch_bounds__(TPB)
k_segment_colmax_s2(const float *__restrict__ pmax, const int *__restrict__ prow, int C, int RS, const int *__restrict__ seg,
                    float *__restrict__ y, int *__restrict__ argrow) {
    const int s = blockIdx.y, c = blockIdx.x * TPB + threadIdx.x;
    if (c >= C) return;
    float m = -INFINITY; int am = seg[s];
    for (int k = 0; k < RS; ++k) {
        const float v = pmax[((long long)s * RS + k) * C + c];
        if (v > m) { m = v; am = prow[((long long)s * RS + k) * C + c]; }
    }
    y[(long long)s * C + c] = m;
    if (argrow) argrow[(long long)s * C + c] = am;
}

__global__ void __launch_bounds__(TPB)
k_segment_colmean_s1(const float *__restrict__ x, long long ld, int C, int rows_per_seg, int RS, int CL, double *__restrict__ part) {
    __shared__ double sm[TPB];
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL, RL = TPB / CL;
    const int c = blockIdx.x * CL + cl, s = blockIdx.y, k = blockIdx.z;
    const int a = (int)((long long)rows_per_seg * k / RS), b = (int)((long long)rows_per_seg * (k + 1) / RS);
    double acc = 0.0;
    if (c < C)
        for (int r = a + rl; r < b; r += RL) acc += x[((long long)s * rows_per_seg + r) * ld + c];
    sm[threadIdx.x] = acc;
    __syncthreads();
    if (rl != 0 || c >= C) return;
    for (int i = 1; i < RL; ++i) acc += sm[i * CL + cl];
    part[((long long)s * RS + k) * C + c] = acc;
}

__global__ void __launch_bounds__(TPB)
k_segment_colmean_s2(const double *__restrict__ part, int C, int RS, int rows_per_seg, float *__restrict__ y) {
    const int s = blockIdx.y, c = blockIdx.x * TPB + threadIdx.x;
    if (c >= C) return;
    double a = 0.0;
    for (int k = 0; k < RS; ++k) a += part[((long long)s * RS + k) * C + c];
    y[(long long)s * C + c] = (float)(a / rows_per_seg);
}

// softmax over the first 2 channels of [rows][ld] -> planar (B,2,HW)  (g_mask, gnet.py:124)
__global__ void __launch_bounds__(TPB)
k_softmax2_to_nchw(const float *__restrict__ x, long long ld, float *__restrict__ y, int B, long long HW) {
    long long total = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        long long b = i / HW, p = i - b * HW;
        float a = x[i * ld], c = x[i * ld + 1];
        float m = fmaxf(a, c);
        float ea = expf(a - m), ec = expf(c - m), inv = 1.f / (ea + ec);
        y[(b * 2) * HW + p] = ea * inv;
        y[(b * 2 + 1) * HW + p] = ec * inv;
    }
}

// G's two heads as ONE 4-channel map (channel 0: depth, channels 1-2: mask logits; gnet.py:121-124): planar depth (B,1,HW) and the
// 2-way softmax (B,2,HW) from one 16-byte read per pixel
__global__ void __launch_bounds__(TPB)
k_heads_to_nchw(const float4 *__restrict__ x, float *__restrict__ depth, float *__restrict__ mask, int B, long long HW) {
    long long total = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        long long b = i / HW, p = i - b * HW;
        const float4 v = x[i];
        depth[i] = v.x;
        float m = fmaxf(v.y, v.z);
        float ea = expf(v.y - m), ec = expf(v.z - m), inv = 1.f / (ea + ec);
        mask[(b * 2) * HW + p] = ea * inv;
        mask[(b * 2 + 1) * HW + p] = ec * inv;
    }
}

int grid_for(long long total) {
    long long g = (total + TPB - 1) / TPB;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}
}  // namespace

extern "C" int efgh_bn_finalize(const float *stats, int32_t G, int32_t C, double count, const float *gamma,
                                const float *beta, float *rmean, float *rvar, float momentum, float eps,
                                float *scale, float *shift, float *save_mean, float *save_invstd,
                                void *stream) {
    EFGH_CHECK_ARG(stats && gamma && beta && scale && shift && C > 0 && G > 0 && count > 0);
    if (G > 256) {       // two-stage reduction of the per-block partials (`stats` is scratch and is overwritten)
        const int S = 16;
        k_stats_fold<<<dim3(cdiv(C, 32), S), dim3(32, 32), 0, (hipStream_t)stream>>>(const_cast<float *>(stats), G, C, S);
        G = S;
    }
    k_bn_finalize<<<cdiv(C, 32), dim3(32, 32), 0, (hipStream_t)stream>>>(stats, G, C, count, gamma, beta, rmean,
                                                                         rvar, momentum, eps, scale, shift,
                                                                         save_mean, save_invstd);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int32_t efgh_col_stats_groups(int64_t M) { return (int32_t)((M + 511) / 512); }

extern "C" int efgh_col_stats(const float *x, int64_t M, int32_t C, int64_t ld, float *stats, void *stream) {
    EFGH_CHECK_ARG(x && stats && M > 0 && C > 0 && C % 4 == 0 && ld % 4 == 0);
    int CL = 1;
    while (CL < 64 && CL * 4 < C) CL <<= 1;
    dim3 grid(cdiv(C / 4, CL), efgh_col_stats_groups(M));
    k_col_stats<<<grid, TPB, 0, (hipStream_t)stream>>>(x, M, C, ld, 512, CL, stats);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_scale_shift_act(const float *x, int64_t ldx, const float *scale, const float *shift,
                                    const float *res, int64_t ldr, float *y, int64_t ldy, int64_t M,
                                    int32_t C, int32_t act, float slope, void *stream) {
    EFGH_CHECK_ARG(x && y && M > 0 && C > 0);
    bool vec = (C % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && (!res || ldr % 4 == 0) &&
               ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)res) | ((uintptr_t)scale) | ((uintptr_t)shift)) & 15) == 0;
    if (vec && efgh_stream_nt(M * C * 4ll))
        k_scale_shift_act<false, true><<<grid_for(M * (C / 4)), TPB, 0, (hipStream_t)stream>>>(x, ldx, scale, shift, res, ldr,
                                                                                            y, ldy, M, C, act, slope, nullptr);
    else if (vec)
        k_scale_shift_act<false, false><<<grid_for(M * (C / 4)), TPB, 0, (hipStream_t)stream>>>(x, ldx, scale, shift, res, ldr,
                                                                                             y, ldy, M, C, act, slope, nullptr);
    else
        k_scale_shift_act1<<<grid_for(M * C), TPB, 0, (hipStream_t)stream>>>(x, ldx, scale, shift, res, ldr, y,
                                                                           ldy, M, C, act, slope);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

/* the same pass, also leaving the sign bits of y: bits[(r*C + c) / 32] bit (r*C + c) % 32 = (y[r][c] > 0); M*C/32 words.  C % 32 == 0 */
extern "C" int efgh_scale_shift_act_bits(const float *x, int64_t ldx, const float *scale, const float *shift,
                                         const float *res, int64_t ldr, float *y, int64_t ldy, uint32_t *bits, int64_t M,
                                         int32_t C, int32_t act, float slope, void *stream) {
    EFGH_CHECK_ARG(x && y && bits && M > 0 && C > 0 && C % 32 == 0);
    EFGH_CHECK_ARG((ldx % 4 == 0) && (ldy % 4 == 0) && (!res || ldr % 4 == 0) &&
                   ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)res) | ((uintptr_t)scale) | ((uintptr_t)shift)) & 15) == 0);
    if (efgh_stream_nt(M * C * 4ll))
        k_scale_shift_act<true, true><<<grid_for(M * (C / 4)), TPB, 0, (hipStream_t)stream>>>(x, ldx, scale, shift, res, ldr, y, ldy, M,
                                                                                           C, act, slope, bits);
    else
        k_scale_shift_act<true, false><<<grid_for(M * (C / 4)), TPB, 0, (hipStream_t)stream>>>(x, ldx, scale, shift, res, ldr, y, ldy,
                                                                                            M, C, act, slope, bits);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_maxpool2(const float *x, float *y, int32_t B, int32_t H, int32_t W, int32_t C, void *stream) {
    EFGH_CHECK_ARG(x && y && B > 0 && H >= 2 && W >= 2 && C % 4 == 0);
    k_maxpool2<<<grid_for((long long)B * (H / 2) * (W / 2) * (C / 4)), TPB, 0, (hipStream_t)stream>>>(x, y, B, H, W, C);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_maxpool_v2(const float *x, float *y, int32_t B, int32_t H, int32_t W, int32_t C, void *stream) {
    EFGH_CHECK_ARG(x && y && B > 0 && H >= 2 && W >= 1 && C % 4 == 0 && (((uintptr_t)x) & 15) == 0 && (((uintptr_t)y) & 15) == 0);
    const long long rowq = (long long)W * (C / 4);
    k_maxpool_v2<<<grid_for((long long)B * (H / 2) * rowq), TPB, 0, (hipStream_t)stream>>>((const float4 *)x, (float4 *)y, B, H, rowq);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_maxpool2_affine(const float *x, const float *scale, const float *shift, int32_t act, float slope,
                                    float *y, int32_t B, int32_t H, int32_t W, int32_t C, void *stream) {
    EFGH_CHECK_ARG(x && scale && shift && y && B > 0 && H >= 2 && W >= 2 && C % 4 == 0);
    if (efgh_stream_nt((long long)B * H * W * C * 4))
        k_maxpool2_affine<true><<<grid_for((long long)B * (H / 2) * (W / 2) * (C / 4)), TPB, 0, (hipStream_t)stream>>>(x, scale, shift, act,
                                                                                                          slope, y, B, H, W, C);
    else
        k_maxpool2_affine<false><<<grid_for((long long)B * (H / 2) * (W / 2) * (C / 4)), TPB, 0, (hipStream_t)stream>>>(x, scale, shift, act,
                                                                                                          slope, y, B, H, W, C);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_nchw_to_nhwc(const float *x, float *y, int32_t B, int32_t Cs, int64_t HW, int32_t Cd, void *stream) {
    EFGH_CHECK_ARG(x && y && B > 0 && Cs > 0 && Cd >= Cs && HW > 0);
    k_nchw_to_nhwc<<<grid_for((long long)B * HW), TPB, 0, (hipStream_t)stream>>>(x, y, B, Cs, HW, Cd);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_nhwc_to_nchw(const float *x, int64_t ld, float *y, int32_t B, int32_t Cs, int64_t HW, void *stream) {
    EFGH_CHECK_ARG(x && y && B > 0 && Cs > 0 && ld >= Cs && HW > 0);
    k_nhwc_to_nchw<<<grid_for((long long)B * Cs * HW), TPB, 0, (hipStream_t)stream>>>(x, ld, y, B, Cs, HW);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_segment_colmax(const float *x, int64_t ld, int32_t C, const int32_t *seg, int32_t nseg,
                                   float *y, int32_t *argrow, void *stream) {
    EFGH_CHECK_ARG(x && seg && y && C > 0 && nseg > 0);
    dim3 grid(cdiv(C, TPB), nseg);
    k_segment_colmax<<<grid, TPB, 0, (hipStream_t)stream>>>(x, ld, C, seg, y, argrow);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_segment_colmean(const float *x, int64_t ld, int32_t C, int32_t rows_per_seg, int32_t nseg,
                                    float *y, void *stream) {
    EFGH_CHECK_ARG(x && y && C > 0 && nseg > 0 && rows_per_seg > 0);
    dim3 grid(cdiv(C, 32), nseg);
    k_segment_colmean<<<grid, TPB, 0, (hipStream_t)stream>>>(x, ld, C, rows_per_seg, y);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

// slices per segment of the two-stage forms: enough workgroups to fill the part (~1024), at least ~64 rows per slice
static int seg_slices(long long rows_per_seg, int nseg, int cblocks) {
    long long rs = 1024 / ((long long)nseg * cblocks);
    if (rs > rows_per_seg / 64) rs = rows_per_seg / 64;
    if (rs > 256) rs = 256;
    return (int)(rs < 1 ? 1 : rs);
}
static int seg_col_lanes(int C) { int cl = 1; while (cl < 32 && cl < C) cl <<= 1; return cl; }

extern "C" int64_t efgh_segment_workspace(int32_t C, int32_t nseg) {      // bytes (either reduction): 256 slices x (8 + 4) bytes at most
    return (int64_t)nseg * 256 * C * 12;
}

extern "C" int efgh_segment_colmax_ws(const float *x, int64_t ld, int32_t C, const int32_t *seg, int32_t nseg, int64_t rows_hint,
                                      float *y, int32_t *argrow, void *workspace, void *stream) {
    EFGH_CHECK_ARG(x && seg && y && workspace && C > 0 && nseg > 0 && rows_hint > 0);
    const int CL = seg_col_lanes(C), cb = cdiv(C, CL);
    const int RS = seg_slices(rows_hint / nseg, nseg, cb);
    float *pmax = (float *)workspace;
    int *prow = (int *)(pmax + (long long)nseg * RS * C);
    k_segment_colmax_s1<<<dim3(cb, nseg, RS), TPB, 0, (hipStream_t)stream>>>(x, ld, C, seg, RS, CL, pmax, prow);
    k_segment_colmax_s2<<<dim3(cdiv(C, TPB), nseg), TPB, 0, (hipStream_t)stream>>>(pmax, prow, C, RS, seg, y, argrow);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_segment_colmean_ws(const float *x, int64_t ld, int32_t C, int32_t rows_per_seg, int32_t nseg, float *y,
                                       void *workspace, void *stream) {
    EFGH_CHECK_ARG(x && y && workspace && C > 0 && nseg > 0 && rows_per_seg > 0 && (((uintptr_t)workspace) & 7) == 0);
    const int CL = seg_col_lanes(C), cb = cdiv(C, CL);
    const int RS = seg_slices(rows_per_seg, nseg, cb);
    double *part = (double *)workspace;
    k_segment_colmean_s1<<<dim3(cb, nseg, RS), TPB, 0, (hipStream_t)stream>>>(x, ld, C, rows_per_seg, RS, CL, part);
    k_segment_colmean_s2<<<dim3(cdiv(C, TPB), nseg), TPB, 0, (hipStream_t)stream>>>(part, C, RS, rows_per_seg, y);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_softmax2_to_nchw(const float *x, int64_t ld, float *y, int32_t B, int64_t HW, void *stream) {
    EFGH_CHECK_ARG(x && y && B > 0 && HW > 0 && ld >= 2);
    k_softmax2_to_nchw<<<grid_for((long long)B * HW), TPB, 0, (hipStream_t)stream>>>(x, ld, y, B, HW);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_heads_to_nchw(const float *x, float *depth, float *mask, int32_t B, int64_t HW, void *stream) {
    EFGH_CHECK_ARG(x && depth && mask && B > 0 && HW > 0 && (((uintptr_t)x) & 15) == 0);
    k_heads_to_nchw<<<grid_for((long long)B * HW), TPB, 0, (hipStream_t)stream>>>((const float4 *)x, depth, mask, B, HW);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

// ---- fused Adam over one flat parameter buffer (torch.optim.Adam semantics, main.py:181-183) ----
namespace {
__global__ void __launch_bounds__(256)
k_adam(float *__restrict__ w, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
       long long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
    long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 ww = reinterpret_cast<float4 *>(w)[i], gg = reinterpret_cast<const float4 *>(g)[i];
        float4 mm = reinterpret_cast<float4 *>(m)[i], vv = reinterpret_cast<float4 *>(v)[i];
        float *wp = &ww.x, *gp = &gg.x, *mp = &mm.x, *vp = &vv.x;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float gr = gp[q] * gscale + wd * wp[q];
            mp[q] = b1 * mp[q] + (1.f - b1) * gr;
            vp[q] = b2 * vp[q] + (1.f - b2) * gr * gr;
            float denom = sqrtf(vp[q]) / bc2_sqrt + eps;
            wp[q] -= (lr / bc1) * (mp[q] / denom);
        }
        reinterpret_cast<float4 *>(w)[i] = ww;
        reinterpret_cast<float4 *>(m)[i] = mm;
        reinterpret_cast<float4 *>(v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        long long i = (n4 << 2) + threadIdx.x;
        float gr = g[i] * gscale + wd * w[i];
        m[i] = b1 * m[i] + (1.f - b1) * gr;
        v[i] = b2 * v[i] + (1.f - b2) * gr * gr;
        w[i] -= (lr / bc1) * (m[i] / (sqrtf(v[i]) / bc2_sqrt + eps));
    }
}
}  // namespace

extern "C" int efgh_adam_step(float *w, const float *g, float *m, float *v, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int32_t step, float grad_scale,
                              void *stream) {
    EFGH_CHECK_ARG(w && g && m && v && n > 0 && step >= 1);
    EFGH_CHECK_ARG(((((uintptr_t)w) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0);
    float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    k_adam<<<grid_for(n / 4 + 1), 256, 0, (hipStream_t)stream>>>(w, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
                                                                 bc1, sqrtf(bc2), grad_scale);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}
