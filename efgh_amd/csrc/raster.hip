// Point-cloud rasterisers and the PIL-exact image rotate (K9, K10, K12 of SURVEY.md §2a).
//   range image   common/torch_utils.py:11-59   (x,y,z,sqrt(x^2+y^2+z^2+w^2)), last point wins
//   depth image   common/torch_utils.py:61-103  (px,py,pz,w), strict bounds, last point wins
//   rotate        common/torch_utils.py:235-254 -> PIL.Image.rotate NEAREST, 16.16 fixed point
// "Last point wins" (largest point index, the reference's single-thread index_put order) is
// realised with atomicMax on the point index followed by a gather, so the result is deterministic.
#include "common.h"

namespace {
constexpr int TPB = 256;

// pass 1 (range): per point -> pixel index (or -1) + its 4 values; winner = max point index
__global__ void __launch_bounds__(TPB)
k_range_pass1(const float *__restrict__ pc, const float *__restrict__ T, int B, int N, int H, int W,
              float fov_up, float fov_down, float inv_span_den, int *__restrict__ pix,
              float4 *__restrict__ vals, int *__restrict__ winner) {
    long long total = (long long)B * N;
    for (long long g = (long long)blockIdx.x * TPB + threadIdx.x; g < total; g += (long long)gridDim.x * TPB) {
        int b = (int)(g / N), i = (int)(g - (long long)b * N);
        const float *p = pc + (long long)b * 3 * N;
        const float *m = T + b * 16;
        float x0 = p[i], y0 = p[N + i], z0 = p[2 * N + i];
        // e_pc = e_l . [p;1]  (fnet.py:43-44)
        float x = m[0] * x0 + m[1] * y0 + m[2] * z0 + m[3];
        float y = m[4] * x0 + m[5] * y0 + m[6] * z0 + m[7];
        float z = m[8] * x0 + m[9] * y0 + m[10] * z0 + m[11];
        float w = m[12] * x0 + m[13] * y0 + m[14] * z0 + m[15];
        float r = sqrtf(x * x + y * y + z * z + w * w);            // :29 (w row included)
        float pitch = asinf(z / r), yaw = atan2f(y, x);            // :30-31
        int px = -1;
        if (pitch < fov_up && pitch > fov_down) {                  // :38
            float u = ((fov_up - pitch) / inv_span_den) * (float)(H - 1);            // :48
            float v = ((-yaw + 3.14159274101257324f) / 6.28318548202514648f) * (float)(W - 1);  // :49
            int ui = (int)u, vi = (int)v;
            if (ui >= 0 && ui < H && vi >= 0 && vi < W) {
                px = ui * W + vi;
                atomicMax(&winner[(long long)b * H * W + px], i);
            }
        }
        pix[g] = px;
        vals[g] = make_float4(x, y, z, r);
    }
}

// pass 1 (depth)
__global__ void __launch_bounds__(TPB)
k_depth_pass1(const float *__restrict__ pc, const float *__restrict__ P, int B, int N, int H, int W,
              int *__restrict__ pix, float4 *__restrict__ vals, int *__restrict__ winner) {
    long long total = (long long)B * N;
    for (long long g = (long long)blockIdx.x * TPB + threadIdx.x; g < total; g += (long long)gridDim.x * TPB) {
        int b = (int)(g / N), i = (int)(g - (long long)b * N);
        const float *p = pc + (long long)b * 3 * N;
        const float *m = P + b * 12;
        float x0 = p[i], y0 = p[N + i], z0 = p[2 * N + i];
        float xx = m[0] * x0 + m[1] * y0 + m[2] * z0 + m[3];       // :74
        float yy = m[4] * x0 + m[5] * y0 + m[6] * z0 + m[7];
        float w = m[8] * x0 + m[9] * y0 + m[10] * z0 + m[11];
        float x = xx / w, y = yy / w;                               // :77-79
        int px = -1;
        if (x < (float)W && x > 0.f && y < (float)H && y > 0.f && w > 0.f) {   // :81
            int xi = (int)x, yi = (int)y;
            px = yi * W + xi;
            atomicMax(&winner[(long long)b * H * W + px], i);
        }
        pix[g] = px;
        vals[g] = make_float4(x0, y0, z0, w);
    }
}

// pass 2: img[b][pix][0..3] = vals[b][winner] or 0
__global__ void __launch_bounds__(TPB)
k_raster_pass2(const int *__restrict__ winner, const float4 *__restrict__ vals, int B, long long HW, int N,
               float4 *__restrict__ img) {
    long long total = (long long)B * HW;
    for (long long g = (long long)blockIdx.x * TPB + threadIdx.x; g < total; g += (long long)gridDim.x * TPB) {
        int wv = winner[g];
        long long b = g / HW;
        img[g] = wv >= 0 ? vals[b * N + wv] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// backward of both rasters w.r.t. the per-point values: index_put's autograd formula gives
// grad_img[u_i,v_i] to EVERY rasterised point i, overwritten ones included (oracle note).
__global__ void __launch_bounds__(TPB)
k_raster_bwd(const int *__restrict__ pix, const float4 *__restrict__ gimg, int B, int N, long long HW,
             float4 *__restrict__ gvals) {
    long long total = (long long)B * N;
    for (long long g = (long long)blockIdx.x * TPB + threadIdx.x; g < total; g += (long long)gridDim.x * TPB) {
        int px = pix[g];
        long long b = g / N;
        gvals[g] = px >= 0 ? gimg[b * HW + px] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// backward of both rasters w.r.t. the POSE that transformed the points (the training path of the pose heads): the gradient image
// is gathered at every rasterised point (as k_raster_bwd) and contracted with the point on the fly, without the [B][N][4]
// intermediate:   range (mode 0): q = E [p;1], r = |q|;  gq = (g.xyz, 0) + g.w q / r;  gE[i][j] = sum_n gq_i [p;1]_j   (4x4)
//                 depth (mode 1): only the depth channel depends on the projection's third row:  gP[2][j] = sum_n g.w [p;1]_j
// float64 accumulation, fixed summation order (partials per workgroup, summed in order by k_pose_grad_finish).
#define POSE_GRAD_GROUPS 64
__global__ void __launch_bounds__(TPB)
k_raster_pose_bwd(const int *__restrict__ pix, const float4 *__restrict__ gimg, const float *__restrict__ pc,
                  const float *__restrict__ E, int N, long long HW, int mode, double *__restrict__ part) {
    const int b = blockIdx.y, tid = threadIdx.x;
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0;
    float e[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) e[i] = (mode == 0) ? E[(long long)b * 16 + i] : 0.f;
    const float *px_ = pc + (long long)b * 3 * N;
    for (int n = blockIdx.x * TPB + tid; n < N; n += gridDim.x * TPB) {
        const int px = pix[(long long)b * N + n];
        if (px < 0) continue;
        const float4 g = gimg[(long long)b * HW + px];
        const float p[4] = {px_[n], px_[N + n], px_[2 * N + n], 1.f};
        if (mode == 0) {
            float q[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) q[i] = ((e[i * 4] * p[0] + e[i * 4 + 1] * p[1]) + e[i * 4 + 2] * p[2]) + e[i * 4 + 3];
            const float r = sqrtf(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
            const float gq[4] = {g.x + g.w * (q[0] / r), g.y + g.w * (q[1] / r), g.z + g.w * (q[2] / r), g.w * (q[3] / r)};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i * 4 + j] += (double)(gq[i] * p[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[8 + j] += (double)(g.w * p[j]);
        }
    }
    __shared__ double red[TPB / 64][16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        double v = acc[i];
        for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
        if ((tid & 63) == 0) red[tid >> 6][i] = v;
    }
    __syncthreads();
    if (tid < 16) {
        double v = 0.0;
        for (int w = 0; w < TPB / 64; ++w) v += red[w][tid];
        part[((long long)b * gridDim.x + blockIdx.x) * 16 + tid] = v;
    }
}

__global__ void k_pose_grad_finish(const double *__restrict__ part, int G, int nout, float *__restrict__ out) {
    const int b = blockIdx.x, i = threadIdx.x;
    if (i >= nout) return;
    double v = 0.0;
    for (int g = 0; g < G; ++g) v += part[((long long)b * G + g) * 16 + i];
    out[(long long)b * nout + i] = (float)v;
}

// ---- PIL rotate -------------------------------------------------------------------------------
__device__ __forceinline__ double round15(double v) {      // python round(v, 15) for |v| <= 1
    return nearbyint(v * 1e15) / 1e15;
}
__device__ __forceinline__ long long fix16(double v) { return (long long)floor(v * 65536.0 + 0.5); }

// img (B,3,H,W) float holding uint8 values; out_nchw (B,3,H,W) and/or out_nhwc4 [B][H][W][4]
__global__ void __launch_bounds__(TPB)
k_rotate(const float *__restrict__ img, const float *__restrict__ rot_deg, int B, int H, int W,
         float *__restrict__ out_nchw, float *__restrict__ out_nhwc4) {
    const int b = blockIdx.y;
    __shared__ long long coef[6];
    __shared__ int identity;
    if (threadIdx.x == 0) {
        // PIL: angle = angle % 360.0 evaluated on the float32 scalar (python-style remainder)
        float a32 = rot_deg[b];
        float m = fmodf(a32, 360.0f);
        if (m != 0.0f && m < 0.0f) m = m + 360.0f;
        identity = (m == 0.0f);
        double ang = -((double)m) * (3.14159265358979323846 / 180.0);   // -math.radians(angle)
        double c = cos(ang), s = sin(ang);
        double m0 = round15(c), m1 = round15(s), m3 = round15(-s), m4 = round15(c);
        double cx = W / 2.0, cy = H / 2.0;
        double m2 = m0 * (-cx) + m1 * (-cy) + cx;
        double m5 = m3 * (-cx) + m4 * (-cy) + cy;
        coef[0] = fix16(m0); coef[1] = fix16(m1); coef[3] = fix16(m3); coef[4] = fix16(m4);
        coef[2] = fix16(m2 + m0 * 0.5 + m1 * 0.5);
        coef[5] = fix16(m5 + m3 * 0.5 + m4 * 0.5);
    }
    __syncthreads();
    const long long a0 = coef[0], a1 = coef[1], a2 = coef[2], a3 = coef[3], a4 = coef[4], a5 = coef[5];
    const long long HW = (long long)H * W;
    for (long long g = (long long)blockIdx.x * TPB + threadIdx.x; g < HW; g += (long long)gridDim.x * TPB) {
        int y = (int)(g / W), x = (int)(g - (long long)y * W);
        long long xin = x, yin = y;
        if (!identity) { xin = (a2 + a0 * x + a1 * y) >> 16; yin = (a5 + a3 * x + a4 * y) >> 16; }
        float v[3] = {0.f, 0.f, 0.f};
        if (xin >= 0 && xin < W && yin >= 0 && yin < H) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                v[c] = (float)(unsigned char)(int)img[((long long)b * 3 + c) * HW + yin * W + xin];
        }
        if (out_nchw) {
#pragma unroll
            for (int c = 0; c < 3; ++c) out_nchw[((long long)b * 3 + c) * HW + g] = v[c];
        }
        if (out_nhwc4)
            reinterpret_cast<float4 *>(out_nhwc4)[(long long)b * HW + g] = make_float4(v[0], v[1], v[2], 0.f);
    }
}

int grid_for(long long total) {
    long long g = (total + TPB - 1) / TPB;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}
}  // namespace

extern "C" int efgh_range_image(const float *pc, const float *e_l, int32_t B, int32_t N, int32_t H, int32_t W,
                                double fov_up, double fov_down, int32_t *pix, float *vals, int32_t *winner,
                                float *img, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(pc && e_l && pix && vals && winner && img && B > 0 && N > 0 && H > 1 && W > 1);
    long long HW = (long long)H * W;
    if (hipMemsetAsync(winner, 0xFF, (size_t)B * HW * 4, st) != hipSuccess) {
        efgh_set_error("range_image: memset failed");
        return EFGH_E_LAUNCH;
    }
    k_range_pass1<<<grid_for((long long)B * N), TPB, 0, st>>>(pc, e_l, B, N, H, W, (float)fov_up, (float)fov_down,
                                                            (float)(fov_up - fov_down), pix, (float4 *)vals, winner);
    k_raster_pass2<<<grid_for(B * HW), TPB, 0, st>>>(winner, (const float4 *)vals, B, HW, N, (float4 *)img);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_depth_image(const float *pc, const float *cam_T_velo, int32_t B, int32_t N, int32_t H,
                                int32_t W, int32_t *pix, float *vals, int32_t *winner, float *img, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(pc && cam_T_velo && pix && vals && winner && img && B > 0 && N > 0 && H > 0 && W > 0);
    long long HW = (long long)H * W;
    if (hipMemsetAsync(winner, 0xFF, (size_t)B * HW * 4, st) != hipSuccess) {
        efgh_set_error("depth_image: memset failed");
        return EFGH_E_LAUNCH;
    }
    k_depth_pass1<<<grid_for((long long)B * N), TPB, 0, st>>>(pc, cam_T_velo, B, N, H, W, pix, (float4 *)vals, winner);
    k_raster_pass2<<<grid_for(B * HW), TPB, 0, st>>>(winner, (const float4 *)vals, B, HW, N, (float4 *)img);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_raster_bwd(const int32_t *pix, const float *gimg, int32_t B, int32_t N, int64_t HW,
                               float *gvals, void *stream_) {
    EFGH_CHECK_ARG(pix && gimg && gvals && B > 0 && N > 0 && HW > 0);
    k_raster_bwd<<<grid_for((long long)B * N), TPB, 0, (hipStream_t)stream_>>>(pix, (const float4 *)gimg, B, N, HW,
                                                                             (float4 *)gvals);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_raster_pose_bwd(const int32_t *pix, const float *gimg, const float *pc, const float *e_l, int32_t B, int32_t N,
                                    int64_t HW, int32_t mode, double *partials, float *g_pose, void *stream_) {
    EFGH_CHECK_ARG(pix && gimg && pc && partials && g_pose && B > 0 && N > 0 && HW > 0 && (mode == 1 || (mode == 0 && e_l)));
    hipStream_t st = (hipStream_t)stream_;
    const int G = (int)(cdiv(N, TPB) < POSE_GRAD_GROUPS ? cdiv(N, TPB) : POSE_GRAD_GROUPS);
    k_raster_pose_bwd<<<dim3(G, B), TPB, 0, st>>>(pix, (const float4 *)gimg, pc, e_l, N, HW, mode, partials);
    k_pose_grad_finish<<<B, 64, 0, st>>>(partials, G, mode == 0 ? 16 : 12, g_pose);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_rotate_nearest_u8(const float *img, const float *rot_deg, int32_t B, int32_t H, int32_t W,
                                      float *out_nchw, float *out_nhwc4, void *stream_) {
    EFGH_CHECK_ARG(img && rot_deg && (out_nchw || out_nhwc4) && B > 0 && H > 0 && W > 0);
    dim3 grid(grid_for((long long)H * W), B);
    k_rotate<<<grid, TPB, 0, (hipStream_t)stream_>>>(img, rot_deg, B, H, W, out_nchw, out_nhwc4);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}
