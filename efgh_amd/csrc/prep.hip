// GPU-side sample preparation: the per-sample CPU transforms of the reference's loaders
// (`data_loader/loader_utils.py:104-202`: Pillow rotate(expand)/crop/bicubic resize/pad/valid-mask of the camera image,
// radius filter + sub-sampling + random mis-calibration of the sweep) as HBM-bound byte/integer kernels.
// The O(1) geometry (rotation matrix -> 16.16 fixed-point affine coefficients, resample coefficient tables, crop
// offsets) is computed on the host exactly as Pillow / numpy_utils.py do (efgh_amd/data/prepare.py) and handed in.
#include "common.h"

namespace {
constexpr int TPB = 256;

int grid_for(long long total) {
    long long g = (total + TPB - 1) / TPB;
    return (int)(g > 65535 ? 65535 : (g < 1 ? 1 : g));
}

// Image.rotate / ImagingTransformAffine nearest, Geometry.c:affine_fixed:
//   xin = (a2 + a0*x + a1*y) >> 16, yin = (a5 + a3*x + a4*y) >> 16, zero fill outside the source
__global__ void k_affine_nearest_u8(const uint8_t *__restrict__ in, int h, int w, long long a0, long long a1,
                                    long long a2, long long a3, long long a4, long long a5,
                                    uint8_t *__restrict__ out, int nh, int nw) {
    const long long total = (long long)nh * nw;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const int x = (int)(i % nw), y = (int)(i / nw);
        const long long xin = (a2 + a0 * x + a1 * y) >> 16, yin = (a5 + a3 * x + a4 * y) >> 16;
        uint8_t r = 0, g = 0, b = 0;
        if (xin >= 0 && xin < w && yin >= 0 && yin < h) {
            const uint8_t *s = in + (yin * w + xin) * 3;
            r = s[0]; g = s[1]; b = s[2];
        }
        uint8_t *d = out + i * 3;
        d[0] = r; d[1] = g; d[2] = b;
    }
}

// zero_pad_image + crop_image (numpy_utils.py:447-503) as one gather: out[y][x] = in[y+oy][x+ox] or 0
__global__ void k_crop_pad_u8(const uint8_t *__restrict__ in, int h, int w, int oy, int ox,
                              uint8_t *__restrict__ out, int th, int tw) {
    const long long total = (long long)th * tw;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const int x = (int)(i % tw) + ox, y = (int)(i / tw) + oy;
        uint8_t r = 0, g = 0, b = 0;
        if (x >= 0 && x < w && y >= 0 && y < h) {
            const uint8_t *s = in + ((long long)y * w + x) * 3;
            r = s[0]; g = s[1]; b = s[2];
        }
        uint8_t *d = out + i * 3;
        d[0] = r; d[1] = g; d[2] = b;
    }
}

// One pass of Pillow's 8-bit resampling (Resample.c:ImagingResampleHorizontal_8bpc / Vertical_8bpc):
//   out = clip8((2^21 + sum_k in[first + k] * coeff[k]) >> 22) along `axis` (1 = x, 0 = y)
__global__ void k_resample_u8(const uint8_t *__restrict__ in, int h, int w, int axis, int out_size,
                              const int *__restrict__ bounds, const int *__restrict__ coeffs, int ksize,
                              uint8_t *__restrict__ out) {
    const int oh = axis ? h : out_size, ow = axis ? out_size : w;
    const long long total = (long long)oh * ow;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const int x = (int)(i % ow), y = (int)(i / ow);
        const int o = axis ? x : y;
        const int first = bounds[2 * o], n = bounds[2 * o + 1];
        const int *k = coeffs + (long long)o * ksize;
        int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
        const long long step = axis ? 3 : (long long)w * 3;
        const uint8_t *p = in + (axis ? ((long long)y * w + first) * 3 : ((long long)first * w + x) * 3);
        for (int q = 0; q < n; ++q, p += step) {
            const int c = k[q];
            s0 += (int)p[0] * c; s1 += (int)p[1] * c; s2 += (int)p[2] * c;
        }
        s0 >>= 22; s1 >>= 22; s2 >>= 22;
        uint8_t *d = out + i * 3;
        d[0] = (uint8_t)(s0 < 0 ? 0 : (s0 > 255 ? 255 : s0));
        d[1] = (uint8_t)(s1 < 0 ? 0 : (s1 > 255 ? 255 : s1));
        d[2] = (uint8_t)(s2 < 0 ? 0 : (s2 > 255 ? 255 : s2));
    }
}

// (H,W,3) u8 -> (3,H,W) u8 and the valid mask (1,H,W): 0 where all three channels are 0 (numpy_utils.py:505-517)
__global__ void k_hwc_to_chw_u8(const uint8_t *__restrict__ in, long long hw, uint8_t *__restrict__ chw,
                                uint8_t *__restrict__ mask) {
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < hw; i += (long long)gridDim.x * TPB) {
        const uint8_t r = in[i * 3], g = in[i * 3 + 1], b = in[i * 3 + 2];
        if (chw) { chw[i] = r; chw[hw + i] = g; chw[2 * hw + i] = b; }
        if (mask) mask[i] = (r | g | b) ? 1 : 0;
    }
}

// network input: zero_pad_image to (th,tw) + uint8 -> float32, CHW (loader_utils.py:110-114)
__global__ void k_u8_to_f32_chw_pad(const uint8_t *__restrict__ in, int h, int w, int oy, int ox,
                                    float *__restrict__ out, int th, int tw) {
    const long long total = (long long)th * tw;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const int x = (int)(i % tw) - ox, y = (int)(i / tw) - oy;
        float r = 0.f, g = 0.f, b = 0.f;
        if (x >= 0 && x < w && y >= 0 && y < h) {
            const uint8_t *s = in + ((long long)y * w + x) * 3;
            r = (float)s[0]; g = (float)s[1]; b = (float)s[2];
        }
        out[i] = r; out[total + i] = g; out[2 * total + i] = b;
    }
}

// ---- sweep: radius filter (stable compaction), sub-sample gather, mis-calibration transform ---------------
// flag[i] = -radius <= x < radius && -radius <= y < radius of the (optionally pre-gathered, sign-flipped) point
__device__ __forceinline__ bool keep_point(const float *p, float sx, float sy, float radius) {
    const float x = p[0] * sx, y = p[1] * sy;
    return x >= -radius && x < radius && y >= -radius && y < radius;
}

__global__ void k_flag_count(const float *__restrict__ pcd, const int *__restrict__ pre, int n, float sx, float sy,
                             float radius, int *__restrict__ block_count) {
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const int i = blockIdx.x * TPB + threadIdx.x;
    bool k = false;
    if (i < n) k = keep_point(pcd + 4LL * (pre ? pre[i] : i), sx, sy, radius);
    const unsigned long long m = __ballot(k);
    if ((threadIdx.x & 63) == 0) atomicAdd(&cnt, __popcll(m));
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = cnt;
}

__global__ void k_scan_blocks(int *__restrict__ block_count, int nblocks, int *__restrict__ total) {
    // single block: exclusive scan of the per-block counts (nblocks is a few thousand at most)
    __shared__ int carry;
    __shared__ int buf[TPB];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += TPB) {
        const int i = base + threadIdx.x;
        const int v = i < nblocks ? block_count[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < TPB; o <<= 1) {
            int t = threadIdx.x >= o ? buf[threadIdx.x - o] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nblocks) block_count[i] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == TPB - 1) carry += buf[TPB - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ void k_compact(const float *__restrict__ pcd, const int *__restrict__ pre, int n, float sx, float sy,
                          float radius, const int *__restrict__ block_off, int *__restrict__ keep_idx) {
    __shared__ int wave_cnt[TPB / 64];
    const int i = blockIdx.x * TPB + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int src = i < n ? (pre ? pre[i] : i) : 0;
    const bool k = i < n && keep_point(pcd + 4LL * src, sx, sy, radius);
    const unsigned long long m = __ballot(k);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = block_off[blockIdx.x];
    for (int q = 0; q < wave; ++q) off += wave_cnt[q];
    if (k) keep_idx[off + __popcll(m & ((1ull << lane) - 1ull))] = src;
}

// out[r][j] = (float)(T[r][0]*x + T[r][1]*y + T[r][2]*z + T[r][3]) in float64, for the j-th selected point;
// columns past the population are the transform of (0,0,0,1) (loader_utils.py:190-199)
__global__ void k_gather_transform(const float *__restrict__ pcd, const int *__restrict__ keep_idx,
                                   const int *__restrict__ sel, int n_sel, float sx, float sy, const double *__restrict__ T,
                                   int num_points, float *__restrict__ out32, double *__restrict__ out64) {
    const int j = blockIdx.x * TPB + threadIdx.x;
    if (j >= num_points) return;
    double x = 0., y = 0., z = 0.;
    if (j < n_sel) {
        const int k = sel ? sel[j] : j;
        const float *p = pcd + 4LL * keep_idx[k];
        x = (double)(p[0] * sx); y = (double)(p[1] * sy); z = (double)p[2];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        // numpy's (4x4)@(4xN) float64 product: a plain left-to-right sum of the four terms
        const double v = ((T[r * 4] * x + T[r * 4 + 1] * y) + T[r * 4 + 2] * z) + T[r * 4 + 3] * 1.0;
        if (out32) out32[(long long)r * num_points + j] = (float)v;
        if (out64) out64[(long long)r * num_points + j] = v;
    }
}

}  // namespace

extern "C" int efgh_prep_affine_nearest_u8(const uint8_t *in, int32_t h, int32_t w, const int64_t *coef6,
                                           uint8_t *out, int32_t nh, int32_t nw, void *stream) {
    EFGH_CHECK_ARG(in && out && coef6 && h > 0 && w > 0 && nh > 0 && nw > 0 && h < 32768 && w < 32768);
    k_affine_nearest_u8<<<grid_for((long long)nh * nw), TPB, 0, (hipStream_t)stream>>>(
        in, h, w, coef6[0], coef6[1], coef6[2], coef6[3], coef6[4], coef6[5], out, nh, nw);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_prep_crop_pad_u8(const uint8_t *in, int32_t h, int32_t w, int32_t oy, int32_t ox, uint8_t *out,
                                     int32_t th, int32_t tw, void *stream) {
    EFGH_CHECK_ARG(in && out && h > 0 && w > 0 && th > 0 && tw > 0);
    k_crop_pad_u8<<<grid_for((long long)th * tw), TPB, 0, (hipStream_t)stream>>>(in, h, w, oy, ox, out, th, tw);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_prep_resample_u8(const uint8_t *in, int32_t h, int32_t w, int32_t axis, int32_t out_size,
                                     const int32_t *bounds, const int32_t *coeffs, int32_t ksize, uint8_t *out,
                                     void *stream) {
    EFGH_CHECK_ARG(in && out && bounds && coeffs && h > 0 && w > 0 && out_size > 0 && ksize > 0 && (axis == 0 || axis == 1));
    const long long total = axis ? (long long)h * out_size : (long long)out_size * w;
    k_resample_u8<<<grid_for(total), TPB, 0, (hipStream_t)stream>>>(in, h, w, axis, out_size, bounds, coeffs, ksize, out);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_prep_hwc_to_chw_u8(const uint8_t *in, int32_t h, int32_t w, uint8_t *chw, uint8_t *mask,
                                       void *stream) {
    EFGH_CHECK_ARG(in && (chw || mask) && h > 0 && w > 0);
    k_hwc_to_chw_u8<<<grid_for((long long)h * w), TPB, 0, (hipStream_t)stream>>>(in, (long long)h * w, chw, mask);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_prep_u8_to_f32_chw_pad(const uint8_t *in, int32_t h, int32_t w, int32_t oy, int32_t ox, float *out,
                                           int32_t th, int32_t tw, void *stream) {
    EFGH_CHECK_ARG(in && out && h > 0 && w > 0 && th > 0 && tw > 0);
    k_u8_to_f32_chw_pad<<<grid_for((long long)th * tw), TPB, 0, (hipStream_t)stream>>>(in, h, w, oy, ox, out, th, tw);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int32_t efgh_prep_filter_blocks(int32_t n) { return (n + TPB - 1) / TPB; }

extern "C" int efgh_prep_radius_filter(const float *pcd, const int32_t *pre_idx, int32_t n, int32_t flip_xy,
                                       float radius, int32_t *block_scratch, int32_t *keep_idx, int32_t *count,
                                       void *stream) {
    hipStream_t st = (hipStream_t)stream;
    EFGH_CHECK_ARG(pcd && block_scratch && keep_idx && count && n > 0 && radius > 0.f);
    const float s = flip_xy ? -1.f : 1.f;
    const int nb = (n + TPB - 1) / TPB;
    k_flag_count<<<nb, TPB, 0, st>>>(pcd, pre_idx, n, s, s, radius, block_scratch);
    k_scan_blocks<<<1, TPB, 0, st>>>(block_scratch, nb, count);
    k_compact<<<nb, TPB, 0, st>>>(pcd, pre_idx, n, s, s, radius, block_scratch, keep_idx);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_prep_gather_transform(const float *pcd, const int32_t *keep_idx, const int32_t *sel, int32_t n_sel,
                                          int32_t flip_xy, const double *T34, int32_t num_points, float *out32,
                                          double *out64, void *stream) {
    EFGH_CHECK_ARG(pcd && keep_idx && T34 && (out32 || out64) && num_points > 0 && n_sel >= 0 && n_sel <= num_points);
    const float s = flip_xy ? -1.f : 1.f;
    k_gather_transform<<<cdiv(num_points, TPB), TPB, 0, (hipStream_t)stream>>>(pcd, keep_idx, sel, n_sel, s, s, T34,
                                                                             num_points, out32, out64);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}
