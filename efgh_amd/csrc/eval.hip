// Pose-error metrics of the evaluation loop on the device (metrics part):
// `Err.calc_error_odom_np` / `calc_error_raw_np` (common/helper.py:163-207), which the reference evaluates on the host
// after a `.cpu()` of both 4x4 poses every step (helper.py:143-144).  One thread per pose pair; the history of
// per-step errors stays in HBM and is only read back when the running mean/std are printed.
#include "common.h"

namespace {

__global__ void k_pose_errors(const float *__restrict__ gt, const float *__restrict__ pred, int B, int mode,
                              float *__restrict__ rot_err, float *__restrict__ trs_err) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float *g = gt + 16 * b, *p = pred + 16 * b;
    const float dt0 = p[3] - g[3], dt1 = p[7] - g[7], dt2 = p[11] - g[11];
    if (mode == 0) {
        // helper.py:198-207 in the float32 the reference's numpy arrays carry: tr(pred_R^T gt_R) = sum_ij pred_ij*gt_ij
        float tr = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float d = 0.f;                      // diagonal element i of pred_R^T . gt_R
#pragma unroll
            for (int k = 0; k < 3; ++k) d += p[4 * k + i] * g[4 * k + i];
            tr += d;
        }
        float t = (tr - 1.f) / 2.f;
        t = fminf(fmaxf(t, -1.f), 1.f);
        rot_err[b] = 180.f * acosf(t) / 3.14159265358979323846f;
        trs_err[b] = sqrtf(dt0 * dt0 + dt1 * dt1 + dt2 * dt2);
    } else {
        // helper.py:165-196: angle of q_gt * q_pred^-1 = rotation angle of gt_R . pred_R^T, as 2*atan2(|v|, |w|)
        double R[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double s = 0.;
#pragma unroll
                for (int k = 0; k < 3; ++k) s += (double)g[4 * i + k] * (double)p[4 * j + k];
                R[i][j] = s;
            }
        const double vx = R[2][1] - R[1][2], vy = R[0][2] - R[2][0], vz = R[1][0] - R[0][1];
        const double s = 0.5 * sqrt(vx * vx + vy * vy + vz * vz), c = 0.5 * (R[0][0] + R[1][1] + R[2][2] - 1.0);
        rot_err[b] = (float)(atan2(s, c) * (180.0 / 3.14159265358979323846));
        trs_err[b] = (fabsf(dt0) + fabsf(dt1) + fabsf(dt2)) / 3.f;
    }
}

}  // namespace

extern "C" int efgh_pose_errors(const float *gt, const float *pred, int32_t B, int32_t mode, float *rot_err,
                                float *trs_err, void *stream) {
    EFGH_CHECK_ARG(gt && pred && rot_err && trs_err && B > 0 && (mode == 0 || mode == 1));
    k_pose_errors<<<cdiv(B, 64), 64, 0, (hipStream_t)stream>>>(gt, pred, B, mode, rot_err, trs_err);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}
