// F-net cross-modal correlation head (K11 of SURVEY.md §2a), nets/fnet.py:57,64,78-81:
//   cam_n = cam / (max(cam)-min(cam));  rng_n = rng / (max(rng)-min(rng))
//   rng_p = [mirror(last W/8 cols) | rng_n | first W/8 cols]          (torch_utils.py:271-284)
//   score[j] = sigmoid( (1/C) * sum_{c,y,x} rng_p[y][j+x][c] * cam_n[y][x][c] )
// Feature maps are channels-last [B][h][w][C] with C = 16.
#include "common.h"

namespace {
constexpr int TPB = 256;

// ---- global min / max per sample: x [B][n] -> mm[B][2] ----------------------------------------
__global__ void __launch_bounds__(TPB)
k_minmax_part(const float *__restrict__ x, long long n, int G, float *__restrict__ part) {
    int b = blockIdx.y, g = blockIdx.x;
    const float *p = x + (long long)b * n;
    float mn = INFINITY, mx = -INFINITY;
    if ((n & 3) == 0 && (((uintptr_t)p) & 15) == 0) {          // 16-byte loads (min / max do not care about the order)
        const float4 *p4 = reinterpret_cast<const float4 *>(p);
        const long long n4 = n >> 2;
        for (long long i = (long long)g * TPB + threadIdx.x; i < n4; i += (long long)G * TPB) {
            const float4 v = p4[i];
            mn = fminf(fminf(mn, v.x), fminf(v.y, fminf(v.z, v.w))); mx = fmaxf(fmaxf(mx, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
        }
    } else {
        for (long long i = (long long)g * TPB + threadIdx.x; i < n; i += (long long)G * TPB) {
            float v = p[i]; mn = fminf(mn, v); mx = fmaxf(mx, v);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
    __shared__ float smn[TPB / 64], smx[TPB / 64];
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < TPB / 64; ++i) { mn = fminf(mn, smn[i]); mx = fmaxf(mx, smx[i]); }
        part[((long long)b * G + g) * 2] = mn; part[((long long)b * G + g) * 2 + 1] = mx;
    }
}
__global__ void k_minmax_final(const float *__restrict__ part, int G, float *__restrict__ mm) {
    int b = blockIdx.x;
    float mn = INFINITY, mx = -INFINITY;
    for (int g = threadIdx.x; g < G; g += 64) { mn = fminf(mn, part[((long long)b * G + g) * 2]); mx = fmaxf(mx, part[((long long)b * G + g) * 2 + 1]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
    if (threadIdx.x == 0) { mm[b * 2] = mn; mm[b * 2 + 1] = mx; }
}

// ---- normalise + pad: rng [B][h][w][C] -> rp [B][h][w+2*off][C] -----------------------------------
// rows of y have `wpitch` >= w + 2*off pixels; the pixels past the padded width are zero
// grid (pixels of a padded row x channel quads, h, B): no 64-bit index divisions per element (the flat version spent 75 us on 20 MB)
__global__ void __launch_bounds__(TPB)
k_norm_pad(const float *__restrict__ x, const float *__restrict__ mm, int B, int h, int w, int C, int off, int wpitch,
           float *__restrict__ y) {
    const int wp = w + 2 * off, c4n = C >> 2;
    const int idx = blockIdx.x * TPB + threadIdx.x;
    if (idx >= wpitch * c4n) return;
    const int xp = idx / c4n, cq = idx - xp * c4n, yy = blockIdx.y, b = blockIdx.z;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (xp < wp) {
        const int xs = xp < off ? (w - 1 - xp) : (xp < off + w ? xp - off : xp - off - w);
        const float d = mm[b * 2 + 1] - mm[b * 2];
        v = *reinterpret_cast<const float4 *>(x + (((long long)b * h + yy) * w + xs) * C + cq * 4);
        v.x /= d; v.y /= d; v.z /= d; v.w /= d;
    }
    reinterpret_cast<float4 *>(y)[((long long)b * h + yy) * wpitch * c4n + idx] = v;
}

// ---- correlation partials: part[b][y][j] = sum_{x,c} rp[b][y][j+x][c] * cam[b][y][x][c]/(max-min) ----
// one block = one (b, y, tile of 256*JT shifts); the normalised camera row lives in LDS; every thread
// owns JT consecutive shifts and slides a register window over the range row, so each 64-B range
// pixel fetched feeds JT*16 FMAs.
constexpr int JT = 4;
__global__ void __launch_bounds__(TPB)
k_corr_rows(const float *__restrict__ rp, const float *__restrict__ cam, const float *__restrict__ cam_mm,
            int h, int wc, int wp, int nj, float *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float srow[];      // wc*16 floats
    const int b = blockIdx.z, y = blockIdx.y;
    const int j0 = (blockIdx.x * TPB + threadIdx.x) * JT;
    const float d = cam_mm[b * 2 + 1] - cam_mm[b * 2];
    const float4 *crow = reinterpret_cast<const float4 *>(cam + (((long long)b * h + y) * wc) * 16);
    for (int i = threadIdx.x; i < wc * 4; i += TPB) {
        float4 v = crow[i];
        v.x /= d; v.y /= d; v.z /= d; v.w /= d;
        reinterpret_cast<float4 *>(srow)[i] = v;
    }
    __syncthreads();
    if (j0 >= nj) return;
    const float4 *r = reinterpret_cast<const float4 *>(rp + (((long long)b * h + y) * wp) * 16);
    const float4 *s = reinterpret_cast<const float4 *>(srow);
    float acc[JT];
    float4 win[JT][4];                     // range pixels j0+x .. j0+x+JT-1 (4 float4 = 16 channels each)
#pragma unroll
    for (int q = 0; q < JT; ++q) {
        acc[q] = 0.f;
        int px = j0 + q; if (px > wp - 1) px = wp - 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) win[q][k] = r[(long long)px * 4 + k];
    }
    for (int x = 0; x < wc; ++x) {
        float4 s0 = s[x * 4], s1 = s[x * 4 + 1], s2 = s[x * 4 + 2], s3 = s[x * 4 + 3];
#pragma unroll
        for (int q = 0; q < JT; ++q) {
            float4 a0 = win[q][0], a1 = win[q][1], a2 = win[q][2], a3 = win[q][3];
            acc[q] += (a0.x * s0.x + a0.y * s0.y + a0.z * s0.z + a0.w * s0.w) +
                      (a1.x * s1.x + a1.y * s1.y + a1.z * s1.z + a1.w * s1.w) +
                      (a2.x * s2.x + a2.y * s2.y + a2.z * s2.z + a2.w * s2.w) +
                      (a3.x * s3.x + a3.y * s3.y + a3.z * s3.z + a3.w * s3.w);
        }
        // slide: drop pixel j0+x, fetch pixel j0+x+JT
#pragma unroll
        for (int q = 0; q < JT - 1; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) win[q][k] = win[q + 1][k];
        int px = j0 + x + JT; if (px > wp - 1) px = wp - 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) win[JT - 1][k] = r[(long long)px * 4 + k];
    }
#pragma unroll
    for (int q = 0; q < JT; ++q)
        if (j0 + q < nj) part[((long long)b * h + y) * nj + j0 + q] = acc[q];
}

// ---- reduce over rows, scale by 1/C, sigmoid -------------------------------------------------------
__global__ void __launch_bounds__(TPB)
k_corr_finish(const float *__restrict__ part, int B, int h, int nj, float invC, float *__restrict__ logit,
              float *__restrict__ score) {
    long long total = (long long)B * nj;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        int b = (int)(i / nj), j = (int)(i - (long long)b * nj);
        float a = 0.f;
        for (int y = 0; y < h; ++y) a += part[((long long)b * h + y) * nj + j];
        a *= invC;
        if (logit) logit[i] = a;
        score[i] = 1.0f / (1.0f + expf(-a));
    }
}

// ---- MFMA formulation of the correlation (fnet.py:79): the camera row is cut into N segments of `segw`
// pixels; P[m][s] = sum_{y, x<segw, c} rp[y][m+x][c] * cam_n[y][s*segw+x][c] is ONE gather-GEMM (mode 3:
// tap = image row, C = segw*16 contiguous floats), and score[j] = sigmoid((1/16) sum_s P[j + s*segw][s]).
// Wc[b][ks][s][yy][x*16+c] = cam[b][ks*T+yy][s*segw+x][c] / (max-min)   (0 beyond the camera width);
// the image rows are cut into nsplit groups of T = h/nsplit (split-K: one GEMM problem per group)
__global__ void __launch_bounds__(TPB)
k_corr_pack_cam(const float *__restrict__ cam, const float *__restrict__ cam_mm, int B, int h, int wc, int segw,
                int nseg, int nsplit, float *__restrict__ Wc) {
    const int cseg = segw * 4;                        // float4 per (segment, row)
    const int T = h / nsplit;
    long long total = (long long)B * nseg * h * cseg;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        int q = (int)(i % cseg); long long r = i / cseg;
        int yy = (int)(r % T); r /= T;
        int s = (int)(r % nseg); r /= nseg;
        int ks = (int)(r % nsplit); int b = (int)(r / nsplit);
        int y = ks * T + yy;
        int x = s * segw + (q >> 2);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (x < wc) {
            const float d = cam_mm[b * 2 + 1] - cam_mm[b * 2];
            v = reinterpret_cast<const float4 *>(cam + (((long long)b * h + y) * wc + x) * 16)[q & 3];
            v.x /= d; v.y /= d; v.z /= d; v.w /= d;
        }
        reinterpret_cast<float4 *>(Wc)[i] = v;
    }
}

// one output (b, j) per 32 lanes: lane s sums its camera segment over the nsplit row groups (nsplit dependent loads instead of
// nsplit * nseg = 512 per thread - the first version gave every output ONE thread: 40 workgroups, 128 us for 26 MB), then a fixed
// xor tree over the lanes (deterministic; nseg <= 32)
__global__ void __launch_bounds__(TPB)
k_corr_fold(const float *__restrict__ P, int B, int nsplit, long long Mv, int ldp, int nseg, int segw, int nj,
            float invC, float *__restrict__ logit, float *__restrict__ score) {
    const long long total = (long long)B * nj;
    const int s = threadIdx.x & 31;
    for (long long i = (long long)blockIdx.x * (TPB / 32) + (threadIdx.x >> 5); i < total; i += (long long)gridDim.x * (TPB / 32)) {
        const int b = (int)(i / nj), j = (int)(i - (long long)b * nj);
        float a = 0.f;
        if (s < nseg)
            for (int ks = 0; ks < nsplit; ++ks)
                a += P[(((long long)b * nsplit + ks) * Mv + j + (long long)s * segw) * ldp + s];
        a += __shfl_xor(a, 16); a += __shfl_xor(a, 8); a += __shfl_xor(a, 4); a += __shfl_xor(a, 2); a += __shfl_xor(a, 1);
        if (s == 0) {
            a *= invC;
            if (logit) logit[i] = a;
            score[i] = 1.0f / (1.0f + expf(-a));
        }
    }
}

int grid_for(long long total) {
    long long g = (total + TPB - 1) / TPB;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

// ---- MFMA formulation of the correlation's backward -----------------------------------------------------
// dcam_n[b][y][x][c] = sum_j dl[b][j] rp[b][y][j+x][c]  and  drp[b][y][m][c] = sum_x dl[b][m-x] cam_n[b][y][x][c]
// are products with the Toeplitz matrix of dl once the operands are laid out as planes [(y,c)][position]:
//   dcamT[(y,c)][x] = sum_m rpT[(y,c)][m] * T[x][m],   T[x][m]  = dl[m - x]
//   drpT [(y,c)][m] = sum_x camT[(y,c)][x] * TT[m][x],  TT[m][x] = dl[m - x]
// i.e. two batched efgh_gather_gemm launches (mode 0, M = 16h, depth wp resp. wc) instead of 2*nj FMAs per output element on
// the VALU.  The kernels below only re-lay out the operands (HBM-bound, a few hundred MB per step).
__global__ void k_corr_planes(const float *__restrict__ x, const float *__restrict__ mm, int h, int w, int w_in_pitch,
                              int wP, float *__restrict__ out) {
    // block = 64 positions x 16 channels of one (b, y); LDS transpose so that both sides are coalesced
    __shared__ float tile[64][17];
    const int b = blockIdx.z, y = blockIdx.y, m0 = blockIdx.x * 64;
    const float inv = mm ? 1.0f / (mm[b * 2 + 1] - mm[b * 2]) : 1.0f;
    const float *src = x + (((long long)b * h + y) * w_in_pitch) * 16;
    for (int i = threadIdx.x; i < 64 * 16; i += TPB) {
        const int m = i >> 4, c = i & 15;
        tile[m][c] = (m0 + m < w) ? src[(long long)(m0 + m) * 16 + c] : 0.f;
    }
    __syncthreads();
    float *dst = out + (((long long)b * h + y) * 16) * wP;
    for (int i = threadIdx.x; i < 64 * 16; i += TPB) {
        const int c = i >> 6, m = i & 63;
        if (m0 + m < wP) dst[(long long)c * wP + m0 + m] = mm ? tile[m][c] * inv : tile[m][c];
    }
}

__global__ void k_corr_unplanes(const float *__restrict__ in, int h, int w, int wP, float *__restrict__ out) {
    __shared__ float tile[16][65];
    const int b = blockIdx.z, y = blockIdx.y, m0 = blockIdx.x * 64;
    const float *src = in + (((long long)b * h + y) * 16) * wP;
    for (int i = threadIdx.x; i < 64 * 16; i += TPB) {
        const int c = i >> 6, m = i & 63;
        tile[c][m] = (m0 + m < w) ? src[(long long)c * wP + m0 + m] : 0.f;
    }
    __syncthreads();
    float *dst = out + (((long long)b * h + y) * w) * 16;
    for (int i = threadIdx.x; i < 64 * 16; i += TPB) {
        const int m = i >> 4, c = i & 15;
        if (m0 + m < w) dst[(long long)(m0 + m) * 16 + c] = tile[c][m];
    }
}

// T[b][r][c] = dl[b][c - r] (transpose == 0) or dl[b][r - c] (transpose == 1); zero outside [0, nj) and in the row padding
__global__ void k_corr_toeplitz(const float *__restrict__ dl, int nj, int rows, int cols, int colsP, int transpose,
                                float *__restrict__ T) {
    const int b = blockIdx.y;
    const long long total = (long long)rows * colsP;
    const float *g = dl + (long long)b * nj;
    float *dst = T + (long long)b * total;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const int c = (int)(i % colsP), r = (int)(i / colsP);
        const int j = transpose ? r - c : c - r;
        dst[i] = (c < cols && j >= 0 && j < nj) ? g[j] : 0.f;
    }
}


// ---- backward of x_n = x / (max(x) - min(x)) per sample (fnet.py:57,64) -------------------------------------------
// dx = dxn/d - [x == max] * S/(d^2 k_max) + [x == min] * S/(d^2 k_min),  S = sum(dxn * x), d = max - min,
// k_* = number of elements attaining the extremum (torch's max()/min() backward splits the gradient evenly among ties)
__global__ void __launch_bounds__(TPB)
k_norm_bwd_part(const float *__restrict__ x, const float *__restrict__ dxn, const float *__restrict__ mm, long long n,
                int G, float *__restrict__ part) {
    const int b = blockIdx.y, g = blockIdx.x;
    const float *px = x + (long long)b * n, *pd = dxn + (long long)b * n;
    const float mn = mm[b * 2], mx = mm[b * 2 + 1];
    float s = 0.f, kmx = 0.f, kmn = 0.f;
    for (long long i = (long long)g * TPB + threadIdx.x; i < n; i += (long long)G * TPB) {
        const float v = px[i];
        s += pd[i] * v;
        kmx += v == mx ? 1.f : 0.f;
        kmn += v == mn ? 1.f : 0.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); kmx += __shfl_xor(kmx, o); kmn += __shfl_xor(kmn, o); }
    __shared__ float sh[3][TPB / 64];
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = kmx; sh[2][threadIdx.x >> 6] = kmn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < TPB / 64; ++i) { s += sh[0][i]; kmx += sh[1][i]; kmn += sh[2][i]; }
        float *o = part + ((long long)b * G + g) * 3;
        o[0] = s; o[1] = kmx; o[2] = kmn;
    }
}

// one block per sample: part[b][0][0..2] = sum over the G partial rows (in double)
__global__ void k_norm_bwd_fold(float *__restrict__ part, int G) {
    const int b = blockIdx.x;
    __shared__ double sh[3][64];
    double a[3] = {0.0, 0.0, 0.0};
    for (int g = threadIdx.x; g < G; g += 64)
        for (int q = 0; q < 3; ++q) a[q] += part[((long long)b * G + g) * 3 + q];
    for (int q = 0; q < 3; ++q) sh[q][threadIdx.x] = a[q];
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 0; q < 3; ++q) { double t = 0.0; for (int i = 0; i < 64; ++i) t += sh[q][i]; part[(long long)b * G * 3 + q] = (float)t; }
    }
}

__global__ void __launch_bounds__(TPB)
k_norm_bwd_apply(const float *__restrict__ x, const float *__restrict__ dxn, const float *__restrict__ mm,
                 const float *__restrict__ part, long long n, int G, float *__restrict__ dx) {
    const int b = blockIdx.y;
    const float *tot = part + (long long)b * G * 3;            // folded by k_norm_bwd_fold into row 0 of the sample
    const float mn = mm[b * 2], mx = mm[b * 2 + 1], d = mx - mn;
    const float cmx = tot[0] / (d * d) / tot[1], cmn = tot[0] / (d * d) / tot[2];
    const float *px = x + (long long)b * n, *pd = dxn + (long long)b * n;
    float *po = dx + (long long)b * n;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) {
        const float v = px[i];
        float r = pd[i] / d;
        if (v == mx) r -= cmx;
        if (v == mn) r += cmn;
        po[i] = r;
    }
}

}  // namespace

extern "C" int32_t efgh_minmax_groups(int64_t n) {
    long long g = (n + TPB * 8 - 1) / (TPB * 8);
    return (int32_t)(g > 1024 ? 1024 : (g < 1 ? 1 : g));
}

extern "C" int efgh_minmax(const float *x, int32_t B, int64_t n, float *part, float *mm, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(x && part && mm && B > 0 && n > 0);
    int G = efgh_minmax_groups(n);
    k_minmax_part<<<dim3(G, B), TPB, 0, st>>>(x, n, G, part);
    k_minmax_final<<<B, 64, 0, st>>>(part, G, mm);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_corr_pad(const float *rng, const float *rng_mm, int32_t B, int32_t h, int32_t w, int32_t C,
                             int32_t off, int32_t wpitch, float *rp, void *stream_) {
    EFGH_CHECK_ARG(rng && rng_mm && rp && B > 0 && h > 0 && w > 0 && C % 4 == 0 && off >= 0 && off <= w);
    EFGH_CHECK_ARG(wpitch >= w + 2 * off);
    EFGH_CHECK_ARG(h <= 65535 && B <= 65535 && (long long)wpitch * (C / 4) < 0x7fffffffLL);
    k_norm_pad<<<dim3((unsigned)(((long long)wpitch * (C / 4) + TPB - 1) / TPB), h, B), TPB, 0, (hipStream_t)stream_>>>(
        rng, rng_mm, B, h, w, C, off, wpitch, rp);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_corr1d(const float *rp, const float *cam, const float *cam_mm, int32_t B, int32_t h,
                           int32_t wc, int32_t wp, float *part, float *logit, float *score, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(rp && cam && cam_mm && part && score && B > 0 && h > 0 && wc > 0 && wp >= wc);
    EFGH_CHECK_ARG(wc * 16 * 4 <= 64 * 1024);
    int nj = wp - wc + 1;
    dim3 grid(cdiv(nj, TPB * JT), h, B);
    k_corr_rows<<<grid, TPB, (size_t)wc * 16 * 4, st>>>(rp, cam, cam_mm, h, wc, wp, nj, part);
    k_corr_finish<<<grid_for((long long)B * nj), TPB, 0, st>>>(part, B, h, nj, 1.0f / 16.0f, logit, score);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}


extern "C" int efgh_corr_pack_cam(const float *cam, const float *cam_mm, int32_t B, int32_t h, int32_t wc, int32_t segw,
                                  int32_t nseg, int32_t nsplit, float *Wc, void *stream_) {
    EFGH_CHECK_ARG(cam && cam_mm && Wc && B > 0 && h > 0 && wc > 0 && segw > 0 && nseg * segw >= wc);
    EFGH_CHECK_ARG(nsplit >= 1 && h % nsplit == 0);
    k_corr_pack_cam<<<grid_for((long long)B * nseg * h * segw * 4), TPB, 0, (hipStream_t)stream_>>>(cam, cam_mm, B, h, wc,
                                                                                                  segw, nseg, nsplit, Wc);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_corr_fold(const float *P, int32_t B, int32_t nsplit, int64_t Mv, int32_t ldp, int32_t nseg,
                              int32_t segw, int32_t nj, float *logit, float *score, void *stream_) {
    EFGH_CHECK_ARG(P && score && B > 0 && nsplit >= 1 && Mv > 0 && nseg > 0 && nseg <= ldp && nj > 0);
    EFGH_CHECK_ARG((int64_t)(nj - 1) + (int64_t)(nseg - 1) * segw < Mv);
    EFGH_CHECK_ARG(nseg <= 32);
    k_corr_fold<<<grid_for((long long)B * nj * 32), TPB, 0, (hipStream_t)stream_>>>(P, B, nsplit, Mv, ldp, nseg, segw, nj,
                                                                             1.0f / 16.0f, logit, score);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_corr_planes(const float *x, const float *mm, int32_t B, int32_t h, int32_t w, int32_t w_in_pitch,
                                int32_t wP, float *out, void *stream_) {
    EFGH_CHECK_ARG(x && out && B > 0 && h > 0 && w > 0 && w_in_pitch >= w && wP >= w && wP % 4 == 0);
    k_corr_planes<<<dim3(cdiv(wP, 64), h, B), TPB, 0, (hipStream_t)stream_>>>(x, mm, h, w, w_in_pitch, wP, out);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_corr_unplanes(const float *in, int32_t B, int32_t h, int32_t w, int32_t wP, float *out, void *stream_) {
    EFGH_CHECK_ARG(in && out && B > 0 && h > 0 && w > 0 && wP >= w);
    k_corr_unplanes<<<dim3(cdiv(w, 64), h, B), TPB, 0, (hipStream_t)stream_>>>(in, h, w, wP, out);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_corr_toeplitz(const float *dl, int32_t B, int32_t nj, int32_t rows, int32_t cols, int32_t colsP,
                                  int32_t transpose, float *T, void *stream_) {
    EFGH_CHECK_ARG(dl && T && B > 0 && nj > 0 && rows > 0 && cols > 0 && colsP >= cols && colsP % 4 == 0);
    const long long total = (long long)rows * colsP;
    long long g = (total + TPB - 1) / TPB;
    k_corr_toeplitz<<<dim3((unsigned)(g > 4096 ? 4096 : g), B), TPB, 0, (hipStream_t)stream_>>>(dl, nj, rows, cols, colsP,
                                                                                               transpose, T);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_norm_bwd(const float *x, const float *dxn, const float *mm, int32_t B, int64_t n, float *part,
                             float *dx, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(x && dxn && mm && part && dx && B > 0 && n > 0);
    const int G = efgh_minmax_groups(n);
    k_norm_bwd_part<<<dim3(G, B), TPB, 0, st>>>(x, dxn, mm, n, G, part);
    k_norm_bwd_fold<<<B, 64, 0, st>>>(part, G);
    long long g = (n + TPB - 1) / TPB;
    k_norm_bwd_apply<<<dim3((unsigned)(g > 2048 ? 2048 : g), B), TPB, 0, st>>>(x, dxn, mm, part, n, G, dx);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}
