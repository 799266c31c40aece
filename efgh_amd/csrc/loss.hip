// Image-sized terms of Gloss (losses/loss_utils.py:146-207) as two streaming kernels:
//   valid = (gt_depth > 0) & (img_mask > 0);  l_depth = sum(valid * (gt_depth - pred_depth)^2) / sum(valid)        (:186-196)
//   gt_mask = gt_depth > 0;                   l_mask  = mean BCE(pred_mask[:,0], gt_mask)                          (:198-199)
// The forward pass also materialises gt['g_depth'] / gt['g_mask'] (the reference returns them in the gt dict) from the
// rasteriser's [B][H][W][4] output (depth = channel 3); the backward pass writes both prediction gradients in one sweep.
#include "common.h"

namespace {
constexpr int TPB = 256;

__device__ __forceinline__ float bce_term(float p, float y) {
    // torch.nn.functional.binary_cross_entropy: log terms clamped at -100
    const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(logf(1.f - p), -100.f);
    return -(y * lp + (1.f - y) * lq);
}

__global__ void __launch_bounds__(TPB)
k_gimg_loss_fwd(const float *__restrict__ pred_depth, const float *__restrict__ pred_mask, long long mask_bstride,
                const float *__restrict__ gdep4, const uint8_t *__restrict__ img_mask, int B, long long HW,
                float *__restrict__ gt_depth, float *__restrict__ gt_mask, double *__restrict__ part) {
    const long long total = (long long)B * HW;
    double s_sq = 0.0, s_valid = 0.0, s_bce = 0.0;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const long long b = i / HW, r = i - b * HW;
        const float gd = gdep4[i * 4 + 3];
        const float pd = pred_depth[i];
        const float pm = pred_mask[b * mask_bstride + r];
        const float gm = gd > 0.f ? 1.f : 0.f;
        const bool valid = gd > 0.f && img_mask[i] > 0;
        gt_depth[i] = gd;
        gt_mask[i] = gm;
        if (valid) { const float d = gd - pd; s_sq += (double)(d * d); s_valid += 1.0; }
        s_bce += (double)bce_term(pm, gm);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s_sq += __shfl_xor(s_sq, o); s_valid += __shfl_xor(s_valid, o); s_bce += __shfl_xor(s_bce, o); }
    __shared__ double sh[3][TPB / 64];
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s_sq; sh[1][threadIdx.x >> 6] = s_valid; sh[2][threadIdx.x >> 6] = s_bce; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double a = 0.0;
        for (int w = 0; w < TPB / 64; ++w) a += sh[threadIdx.x][w];
        part[(long long)blockIdx.x * 3 + threadIdx.x] = a;
    }
}

// out[0] = l_depth, out[1] = l_mask (mean), out[2] = sum(valid)
__global__ void k_gimg_loss_final(const double *__restrict__ part, int G, double total, float *__restrict__ out) {
    __shared__ double sh[3][64];
    double a[3] = {0.0, 0.0, 0.0};
    for (int g = threadIdx.x; g < G; g += 64)
        for (int q = 0; q < 3; ++q) a[q] += part[(long long)g * 3 + q];
    for (int q = 0; q < 3; ++q) sh[q][threadIdx.x] = a[q];
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 0; q < 3; ++q) { a[q] = 0.0; for (int t = 0; t < 64; ++t) a[q] += sh[q][t]; }
        out[0] = (float)(a[0] / a[1]);
        out[1] = (float)(a[2] / total);
        out[2] = (float)a[1];
    }
}

// d l_depth / d pred_depth = -2 (gt - pred) valid / sum(valid);  d l_mask / d p = (p - y) / max(p (1 - p), 1e-12) / (B*HW)
__global__ void __launch_bounds__(TPB)
k_gimg_loss_bwd(const float *__restrict__ pred_depth, const float *__restrict__ pred_mask, long long mask_bstride,
                const float *__restrict__ gt_depth, const uint8_t *__restrict__ img_mask, int B, long long HW,
                const float *__restrict__ sums, const float *__restrict__ g_depth, const float *__restrict__ g_mask,
                float *__restrict__ d_pred_depth, float *__restrict__ d_pred_mask, long long dmask_bstride) {
    const long long total = (long long)B * HW;
    const float kd = g_depth[0] * (-2.f) / sums[2], km = g_mask[0] / (float)total;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const long long b = i / HW, r = i - b * HW;
        const float gd = gt_depth[i], pd = pred_depth[i], p = pred_mask[b * mask_bstride + r];
        const bool valid = gd > 0.f && img_mask[i] > 0;
        d_pred_depth[i] = valid ? kd * (gd - pd) : 0.f;
        const float y = gd > 0.f ? 1.f : 0.f;
        d_pred_mask[b * dmask_bstride + r] = km * (p - y) / fmaxf(p * (1.f - p), 1e-12f);
    }
}

}  // namespace

extern "C" int32_t efgh_gimg_loss_groups(int64_t n) {
    long long g = (n + TPB * 8 - 1) / (TPB * 8);
    return (int32_t)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

extern "C" int efgh_gimg_loss_fwd(const float *pred_depth, const float *pred_mask, int64_t mask_bstride, const float *gdep4,
                                  const uint8_t *img_mask, int32_t B, int64_t HW, float *gt_depth, float *gt_mask,
                                  double *part, float *out3, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(pred_depth && pred_mask && gdep4 && img_mask && gt_depth && gt_mask && part && out3 && B > 0 && HW > 0);
    const int G = efgh_gimg_loss_groups((int64_t)B * HW);
    k_gimg_loss_fwd<<<G, TPB, 0, st>>>(pred_depth, pred_mask, mask_bstride, gdep4, img_mask, B, HW, gt_depth, gt_mask, part);
    k_gimg_loss_final<<<1, 64, 0, st>>>(part, G, (double)B * (double)HW, out3);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_gimg_loss_bwd(const float *pred_depth, const float *pred_mask, int64_t mask_bstride, const float *gt_depth,
                                  const uint8_t *img_mask, int32_t B, int64_t HW, const float *sums3, const float *g_depth,
                                  const float *g_mask, float *d_pred_depth, float *d_pred_mask, int64_t dmask_bstride,
                                  void *stream_) {
    EFGH_CHECK_ARG(pred_depth && pred_mask && gt_depth && img_mask && sums3 && g_depth && g_mask && d_pred_depth && d_pred_mask);
    EFGH_CHECK_ARG(B > 0 && HW > 0);
    const int G = efgh_gimg_loss_groups((int64_t)B * HW) * 4;
    k_gimg_loss_bwd<<<G, TPB, 0, (hipStream_t)stream_>>>(pred_depth, pred_mask, mask_bstride, gt_depth, img_mask, B, HW, sums3,
                                                         g_depth, g_mask, d_pred_depth, d_pred_mask, dmask_bstride);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}
