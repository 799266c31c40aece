// TensorBoard / evaluation overlay images on the GPU (second half): the float64 rasterisers of
// common/numpy_utils.py:299-358 ("the last point of the sweep that lands on a pixel wins"), the raster-order colouring
// of :377-400 and matplotlib's look-up-table colour map.  HBM-bound byte / index work, all in the reference's float64.
//
// The colouring (minmax_color_img_from_img_numpy) is sequential in the reference: in raster order every pixel with a positive value
// paints its (2px+1)^2 window if nothing in the window is already >= its value.  Two pixels interact only if their windows
// intersect, i.e. within 2px rows and columns of each other, and the earlier one in raster order goes first.  With
// c = 2px + 1 the schedule t(y, x) = x + c*y is therefore exact: every pixel a given pixel depends on has a smaller t, and the
// pixels of one t (one per row, c columns apart) have disjoint windows.  One workgroup walks t = 0 .. W-1 + c*(H-1) with a
// barrier per step, up to ceil(W/c) pixels in parallel; independent images go to different workgroups of the same launch.
#include "common.h"

namespace {
constexpr int TPB = 256;

__device__ __forceinline__ bool project_depth(const float *pc, long long ps, int i, const double *T, int H, int W, int &pix,
                                              double &w) {
    const double px = (double)pc[i], py = (double)pc[ps + i], pz = (double)pc[2 * ps + i];
    // numpy's float32 @ float64 product: the 3x4 matrix widened, one FMA chain over k per element (k = 0 first)
    const double x = fma(T[3], 1.0, fma(T[2], pz, fma(T[1], py, T[0] * px)));
    const double y = fma(T[7], 1.0, fma(T[6], pz, fma(T[5], py, T[4] * px)));
    w = fma(T[11], 1.0, fma(T[10], pz, fma(T[9], py, T[8] * px)));
    if (!(w > 0.0 && 0.0 <= x && x < w * (double)W && 0.0 <= y && y < w * (double)H)) return false;
    const int r = (int)(y / w), c = (int)(x / w);
    if (r >= H || c >= W) return false;                  // (cannot happen for finite values; guards the store)
    pix = r * W + c;
    return true;
}

__global__ void __launch_bounds__(TPB)
k_depth_last1(const float *__restrict__ pc, long long ps, int N, const double *__restrict__ T, int H, int W, int *__restrict__ idx) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= N) return;
    int pix; double w;
    if (project_depth(pc, ps, i, T, H, W, pix, w)) atomicMax(&idx[pix], i);
}

__global__ void __launch_bounds__(TPB)
k_depth_last2(const float *__restrict__ pc, long long ps, const double *__restrict__ T, int H, int W, const int *__restrict__ idx,
              uint8_t *__restrict__ out) {
    const int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= H * W) return;
    const int i = idx[p];
    uint8_t v = 0;
    if (i >= 0) {
        int pix; double w;
        project_depth(pc, ps, i, T, H, W, pix, w);
        v = (uint8_t)(long long)w;                       // np.zeros(float64)[...] = w, then .astype('uint8')
    }
    out[p] = v;
}

__device__ __forceinline__ bool project_range(const float *pc, long long ps, int i, const double *T, int H, int W, double up,
                                              double down, int &pix, double &r) {
    const double px = (double)pc[i], py = (double)pc[ps + i], pz = (double)pc[2 * ps + i];
    const double x = fma(T[3], 1.0, fma(T[2], pz, fma(T[1], py, T[0] * px)));
    const double y = fma(T[7], 1.0, fma(T[6], pz, fma(T[5], py, T[4] * px)));
    const double z = fma(T[11], 1.0, fma(T[10], pz, fma(T[9], py, T[8] * px)));
    r = sqrt(x * x + y * y + z * z + 1.0);               // the homogeneous 1 is part of the reference's norm
    const double pitch = asin(z / r), yaw = atan2(y, x);
    if (!(pitch < up && pitch > down)) return false;
    const double PI = 3.141592653589793;
    const double u = ((up - pitch) / (up - down)) * (double)(H - 1);
    const double v = ((-yaw + PI) / (2.0 * PI)) * (double)(W - 1);
    const int ui = (int)u, vi = (int)v;
    if ((unsigned)ui >= (unsigned)H || (unsigned)vi >= (unsigned)W) return false;
    pix = ui * W + vi;
    return true;
}

__global__ void __launch_bounds__(TPB)
k_range_last1(const float *__restrict__ pc, long long ps, int N, const double *__restrict__ T, int H, int W, double up, double down,
              int *__restrict__ idx) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= N) return;
    int pix; double r;
    if (project_range(pc, ps, i, T, H, W, up, down, pix, r)) atomicMax(&idx[pix], i);
}

__global__ void __launch_bounds__(TPB)
k_range_last2(const float *__restrict__ pc, long long ps, const double *__restrict__ T, int H, int W, double up, double down,
              const int *__restrict__ idx, double *__restrict__ out) {
    const int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= H * W) return;
    const int i = idx[p];
    double v = 0.0;
    if (i >= 0) { int pix; project_range(pc, ps, i, T, H, W, up, down, pix, v); }
    out[p] = v;
}

struct PaintJob { const double *in; double *out; int H, W, px; };

__global__ void __launch_bounds__(1024) k_paint(const PaintJob *__restrict__ jobs) {
    const PaintJob j = jobs[blockIdx.x];
    const int H = j.H, W = j.W, px = j.px, c = 2 * px + 1;
    const int steps = W + c * (H - 1);
    for (int t = 0; t < steps; ++t) {
        // rows y with 0 <= t - c*y < W
        const int ylo = t - (W - 1) > 0 ? (t - (W - 1) + c - 1) / c : 0;
        int yhi = t / c; if (yhi > H - 1) yhi = H - 1;
        for (int y = ylo + (int)threadIdx.x; y <= yhi; y += blockDim.x) {
            const int x = t - c * y;
            const double v = j.in[(long long)y * W + x];
            if (v > 0.0) {
                const int y0 = y - px > 0 ? y - px : 0, y1 = y + px + 1 < H - 1 ? y + px + 1 : H - 1;
                const int x0 = x - px > 0 ? x - px : 0, x1 = x + px + 1 < W - 1 ? x + px + 1 : W - 1;
                bool paint = true;
                for (int yy = y0; yy < y1 && paint; ++yy)
                    for (int xx = x0; xx < x1; ++xx)
                        if (!(j.out[(long long)yy * W + xx] < v)) { paint = false; break; }
                if (paint)
                    for (int yy = y0; yy < y1; ++yy)
                        for (int xx = x0; xx < x1; ++xx) j.out[(long long)yy * W + xx] = v;
            }
        }
        __threadfence_block();
        __syncthreads();
    }
}

__global__ void __launch_bounds__(TPB)
k_colorize(const double *__restrict__ mm, long long n, const uint8_t *__restrict__ lut, uint8_t *__restrict__ rgb,
           uint8_t *__restrict__ mask) {
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const double v = mm[i], s = v * 256.0;
    int k = s == 256.0 ? 255 : (int)s;                   // matplotlib: xa *= N; xa[xa == N] = N - 1; astype(int)
    k = k < 0 ? 0 : (k > 255 ? 255 : k);
    rgb[3 * i + 0] = lut[3 * k + 0]; rgb[3 * i + 1] = lut[3 * k + 1]; rgb[3 * i + 2] = lut[3 * k + 2];
    if (mask) mask[i] = v != 0.0 ? 1 : 0;
}

}  // namespace

extern "C" int efgh_sum_depth_last(const float *pc, int64_t pc_cstride, int32_t N, const double *T34, int32_t H, int32_t W,
                                   int32_t *idx_ws, uint8_t *out, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(pc && T34 && idx_ws && out && N > 0 && H > 0 && W > 0 && (int64_t)H * W < 0x7fffffff);
    if (hipMemsetAsync(idx_ws, 0xff, (size_t)H * W * 4, st) != hipSuccess) { efgh_set_error("sum depth: memset failed"); return EFGH_E_LAUNCH; }
    k_depth_last1<<<cdiv(N, TPB), TPB, 0, st>>>(pc, pc_cstride, N, T34, H, W, idx_ws);
    k_depth_last2<<<cdiv((int64_t)H * W, TPB), TPB, 0, st>>>(pc, pc_cstride, T34, H, W, idx_ws, out);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_sum_range_last(const float *pc, int64_t pc_cstride, int32_t N, const double *T34, int32_t H, int32_t W,
                                   double fov_up, double fov_down, int32_t *idx_ws, double *out, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(pc && T34 && idx_ws && out && N > 0 && H > 1 && W > 1 && (int64_t)H * W < 0x7fffffff);
    if (hipMemsetAsync(idx_ws, 0xff, (size_t)H * W * 4, st) != hipSuccess) { efgh_set_error("sum range: memset failed"); return EFGH_E_LAUNCH; }
    k_range_last1<<<cdiv(N, TPB), TPB, 0, st>>>(pc, pc_cstride, N, T34, H, W, fov_up, fov_down, idx_ws);
    k_range_last2<<<cdiv((int64_t)H * W, TPB), TPB, 0, st>>>(pc, pc_cstride, T34, H, W, fov_up, fov_down, idx_ws, out);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_sum_paint(const void *jobs_dev, int32_t njobs, void *stream_) {
    EFGH_CHECK_ARG(jobs_dev && njobs > 0);
    k_paint<<<njobs, 1024, 0, (hipStream_t)stream_>>>((const PaintJob *)jobs_dev);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_sum_colorize(const double *minmax, int64_t n, const uint8_t *lut, uint8_t *rgb, uint8_t *mask, void *stream_) {
    EFGH_CHECK_ARG(minmax && lut && rgb && n > 0);
    k_colorize<<<cdiv(n, TPB), TPB, 0, (hipStream_t)stream_>>>(minmax, n, lut, rgb, mask);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}
