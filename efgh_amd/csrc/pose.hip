// Pose heads of E / H / F and the calibration chain, one launch each (inference path): what the reference does per sample with
// python loops and .item() (common/torch_utils.py:105-146 sign decode, :170-200 rotation between two vectors, :256-269
// A^-1 c_T A calib l_T, fnet.py:87-91 yaw from the correlation peak) and what the training path keeps as ~50 tiny tensor
// expressions per head because autograd has to see them.  One thread per sample; float32 arithmetic in the order of those
// expressions (this file is compiled without FMA contraction), float64 only where python's math.cos / math.sin are.
#include "common.h"

namespace {

__device__ __forceinline__ void rotation_between(const float v1[3], const float v2[3], float R[16]) {
    // torch_utils.py:170-200: v = v1 x v2, c = v1.v2, s = |v|; R = I + K + K^2 (1 - c)/s^2; identity when 1 - c == 0;
    // when 1 + c == 0 a -I whose [0][0] (or [2][2]) is flipped back if both x (or z) components vanish, and R[3][3] = -1
    const float v[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]};
    const float c = (v1[0] * v2[0] + v1[1] * v2[1]) + v1[2] * v2[2];
    const float s2 = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    const float s = sqrtf(s2);
    const float coef = (1.f - c) / (s * s);
    const float K[3][3] = {{0.f, -v[2], v[1]}, {v[2], 0.f, -v[0]}, {-v[1], v[0], 0.f}};
    float r3[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float kk = (K[i][0] * K[0][j] + K[i][1] * K[1][j]) + K[i][2] * K[2][j];
            r3[i][j] = ((i == j ? 1.f : 0.f) + K[i][j]) + kk * coef;
        }
    const bool same = (1.f - c) == 0.f, opp = (1.f + c) == 0.f;
    if (opp) {
        const bool fix0 = v1[0] == 0.f && v2[0] == 0.f;
        const bool fix2 = v1[2] == 0.f && v2[2] == 0.f && !fix0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) r3[i][j] = i == j ? -1.f : 0.f;
        if (fix0) r3[0][0] = 1.f;
        if (fix2) r3[2][2] = 1.f;
    }
    if (same) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) r3[i][j] = i == j ? 1.f : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) R[i * 4 + j] = (i < 3 && j < 3) ? r3[i][j] : 0.f;
    R[15] = (opp && !same) ? -1.f : 1.f;
}

// softmax over nd logits, L2-normalised (enet.py:161-164, hnet.py:59-63); sign class = FIRST argmax of the 2^nd sign logits,
// bits MSB-first, 0 -> -1 (torch_utils.py:105-146); normal = abs * sign (a zero z for nd == 2); rotation onto `dest`
__global__ void k_head_normal(const float *__restrict__ abs_logits, long long lda, const float *__restrict__ sgn_logits,
                              long long lds, int B, int nd, float dx, float dy, float dz, float *__restrict__ abs_out,
                              float *__restrict__ normal, float *__restrict__ R) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float a[3] = {0.f, 0.f, 0.f};
    float mx = abs_logits[b * lda];
    for (int i = 1; i < nd; ++i) mx = fmaxf(mx, abs_logits[b * lda + i]);
    float sum = 0.f;
    for (int i = 0; i < nd; ++i) { a[i] = expf(abs_logits[b * lda + i] - mx); sum += a[i]; }
    float n2 = 0.f;
    for (int i = 0; i < nd; ++i) { a[i] = a[i] / sum; n2 += a[i] * a[i]; }
    const float nrm = sqrtf(n2);
    int cls = 0;
    float best = sgn_logits[b * lds];
    for (int i = 1; i < (1 << nd); ++i) { const float v = sgn_logits[b * lds + i]; if (v > best) { best = v; cls = i; } }
    float nv[3] = {0.f, 0.f, 0.f};
    for (int i = 0; i < nd; ++i) {
        a[i] = a[i] / nrm;
        abs_out[b * nd + i] = a[i];
        nv[i] = a[i] * (((cls >> (nd - 1 - i)) & 1) ? 1.f : -1.f);
        normal[b * nd + i] = nv[i];
    }
    const float d[3] = {dx, dy, dz};
    rotation_between(nv, d, R + (long long)b * 16);
}

// fnet.py:87-91: peak of the correlation -> yaw -> (cos, sin, 0) -> rotation onto e1
__global__ void k_head_yaw(const float *__restrict__ score, long long lds, int B, int n, float *__restrict__ R) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int idx = 0;
    float best = score[b * lds];
    for (int i = 1; i < n; ++i) { const float v = score[b * lds + i]; if (v > best) { best = v; idx = i; } }
    const float f_rad = -((float)idx / (float)(n - 1)) * 2.f * 3.14159274101257324f + 3.14159274101257324f;   // float32, as the tensor expression
    const double rad = (double)f_rad;
    const float v1[3] = {(float)cos(rad), (float)sin(rad), 0.f};
    const float e1[3] = {1.f, 0.f, 0.f};
    rotation_between(v1, e1, R + (long long)b * 16);
}

// torch_utils.py:256-269: A^-1 (c_T (A (calib l_T)))
__global__ void k_cam_T_velo(const float *__restrict__ cT, long long ldc, const float *__restrict__ lT,
                             const float *__restrict__ calib, const float *__restrict__ A, int B, float *__restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float *c = cT + b * ldc, *l = lT + (long long)b * 16, *k = calib + (long long)b * 12, *a = A + (long long)b * 9;
    float m0[3][4], m1[3][4], m2[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            m0[i][j] = ((k[i * 4 + 0] * l[0 * 4 + j] + k[i * 4 + 1] * l[1 * 4 + j]) + k[i * 4 + 2] * l[2 * 4 + j]) + k[i * 4 + 3] * l[3 * 4 + j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) m1[i][j] = (a[i * 3 + 0] * m0[0][j] + a[i * 3 + 1] * m0[1][j]) + a[i * 3 + 2] * m0[2][j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) m2[i][j] = (c[i * 3 + 0] * m1[0][j] + c[i * 3 + 1] * m1[1][j]) + c[i * 3 + 2] * m1[2][j];
    // inverse of the 3x3 A by cofactors (exact for the pixel-centre shift [[1,0,-W/2],[0,1,-H/2],[0,0,1]] the loaders produce)
    const float c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
    const float det = (a[0] * c00 + a[1] * c01) + a[2] * c02;
    const float inv[3][3] = {{c00 / det, (a[2] * a[7] - a[1] * a[8]) / det, (a[1] * a[5] - a[2] * a[4]) / det},
                             {c01 / det, (a[0] * a[8] - a[2] * a[6]) / det, (a[2] * a[3] - a[0] * a[5]) / det},
                             {c02 / det, (a[1] * a[6] - a[0] * a[7]) / det, (a[0] * a[4] - a[1] * a[3]) / det}};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            out[(long long)b * 12 + i * 4 + j] = (inv[i][0] * m2[0][j] + inv[i][1] * m2[1][j]) + inv[i][2] * m2[2][j];
}

}  // namespace

extern "C" int efgh_pose_head_normal(const float *abs_logits, int64_t lda, const float *sgn_logits, int64_t lds, int32_t B,
                                     int32_t nd, float dx, float dy, float dz, float *abs_out, float *normal, float *R44,
                                     void *stream_) {
    EFGH_CHECK_ARG(abs_logits && sgn_logits && abs_out && normal && R44 && B > 0 && (nd == 2 || nd == 3));
    k_head_normal<<<cdiv(B, 64), 64, 0, (hipStream_t)stream_>>>(abs_logits, lda, sgn_logits, lds, B, nd, dx, dy, dz, abs_out, normal, R44);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_pose_head_yaw(const float *score, int64_t lds, int32_t B, int32_t n, float *R44, void *stream_) {
    EFGH_CHECK_ARG(score && R44 && B > 0 && n > 1);
    k_head_yaw<<<cdiv(B, 64), 64, 0, (hipStream_t)stream_>>>(score, lds, B, n, R44);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_pose_cam_T_velo(const float *c_T, int64_t ldc, const float *l_T, const float *calib, const float *A, int32_t B,
                                    float *out34, void *stream_) {
    EFGH_CHECK_ARG(c_T && l_T && calib && A && out34 && B > 0 && ldc >= 9);
    k_cam_T_velo<<<cdiv(B, 64), 64, 0, (hipStream_t)stream_>>>(c_T, ldc, l_T, calib, A, B, out34);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}
