// Pose heads of E / H / F and the calibration chain, one launch each (inference path): what the reference does per sample with
// python loops and .item() (common/torch_utils.py:105-146 sign decode, :170-200 rotation between two vectors, :256-269
// A^-1 c_T A calib l_T, fnet.py:87-91 yaw from the correlation peak) and what the training path keeps as ~50 tiny tensor
// expressions per head because autograd has to see them.  One thread per sample; float32 arithmetic in the order of those
// expressions (this file is compiled without FMA contraction), float64 only where python's math.cos / math.sin are.
#include "common.h"

namespace {

// forward-mode dual numbers: the hand-written backward of the pose heads and of the pose loss terms evaluates the SAME templated
// expressions once per input with that input's derivative seeded to 1 and contracts the resulting output derivatives with the
// incoming gradient (the functions have <= 32 inputs per sample; this is exact and needs no second derivation by hand)
struct Dual { float v, d; };
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) { const float q = a.v / b.v; return {q, (a.d - q * b.d) / b.v}; }
__device__ __forceinline__ Dual operator+(Dual a, float b) { return {a.v + b, a.d}; }
__device__ __forceinline__ Dual operator+(float a, Dual b) { return {a + b.v, b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, float b) { return {a.v - b, a.d}; }
__device__ __forceinline__ Dual operator-(float a, Dual b) { return {a - b.v, -b.d}; }
__device__ __forceinline__ Dual operator*(Dual a, float b) { return {a.v * b, a.d * b}; }
__device__ __forceinline__ Dual operator*(float a, Dual b) { return {a * b.v, a * b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, float b) { return {a.v / b, a.d / b}; }
__device__ __forceinline__ Dual operator-(Dual a) { return {-a.v, -a.d}; }
__device__ __forceinline__ float val(float x) { return x; }
__device__ __forceinline__ float val(Dual x) { return x.v; }
__device__ __forceinline__ float tan_of(float) { return 0.f; }
__device__ __forceinline__ float tan_of(Dual x) { return x.d; }
__device__ __forceinline__ float t_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ Dual t_sqrt(Dual x) { const float r = sqrtf(x.v); return {r, x.d / (2.f * r)}; }
__device__ __forceinline__ float t_exp(float x) { return expf(x); }
__device__ __forceinline__ Dual t_exp(Dual x) { const float e = expf(x.v); return {e, e * x.d}; }
__device__ __forceinline__ float t_log(float x) { return logf(x); }
__device__ __forceinline__ Dual t_log(Dual x) { return {logf(x.v), x.d / x.v}; }
__device__ __forceinline__ float t_abs(float x) { return fabsf(x); }
__device__ __forceinline__ Dual t_abs(Dual x) { return {fabsf(x.v), x.v < 0.f ? -x.d : (x.v > 0.f ? x.d : 0.f)}; }
template <typename T> __device__ __forceinline__ T lift(float x);
template <> __device__ __forceinline__ float lift<float>(float x) { return x; }
template <> __device__ __forceinline__ Dual lift<Dual>(float x) { return {x, 0.f}; }

template <typename T>
__device__ __forceinline__ void rotation_between(const T v1[3], const float v2[3], T R[16]) {
    // torch_utils.py:170-200: v = v1 x v2, c = v1.v2, s = |v|; R = I + K + K^2 (1 - c)/s^2; identity when 1 - c == 0;
    // when 1 + c == 0 a -I whose [0][0] (or [2][2]) is flipped back if both x (or z) components vanish, and R[3][3] = -1.
    // The skew matrix K is built from DETACHED values (:184,194): only (1 - c)/s^2 carries a derivative.
    const T v[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]};
    const T c = (v1[0] * v2[0] + v1[1] * v2[1]) + v1[2] * v2[2];
    const T s2 = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    const T s = t_sqrt(s2);
    const T coef = (1.f - c) / (s * s);
    const float K[3][3] = {{0.f, -val(v[2]), val(v[1])}, {val(v[2]), 0.f, -val(v[0])}, {-val(v[1]), val(v[0]), 0.f}};
    T r3[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float kk = (K[i][0] * K[0][j] + K[i][1] * K[1][j]) + K[i][2] * K[2][j];
            r3[i][j] = ((i == j ? 1.f : 0.f) + K[i][j]) + kk * coef;
        }
    const bool same = (1.f - val(c)) == 0.f, opp = (1.f + val(c)) == 0.f;
    if (opp) {
        const bool fix0 = val(v1[0]) == 0.f && v2[0] == 0.f;
        const bool fix2 = val(v1[2]) == 0.f && v2[2] == 0.f && !fix0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) r3[i][j] = lift<T>(i == j ? -1.f : 0.f);
        if (fix0) r3[0][0] = lift<T>(1.f);
        if (fix2) r3[2][2] = lift<T>(1.f);
    }
    if (same) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) r3[i][j] = lift<T>(i == j ? 1.f : 0.f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) R[i * 4 + j] = (i < 3 && j < 3) ? r3[i][j] : lift<T>(0.f);
    R[15] = lift<T>((opp && !same) ? -1.f : 1.f);
}

// softmax over nd logits, L2-normalised (enet.py:161-164, hnet.py:59-63); sign class `cls` = FIRST argmax of the 2^nd sign
// logits, bits MSB-first, 0 -> -1 (torch_utils.py:105-146); normal = abs * sign (a zero z for nd == 2); rotation onto `dest`
template <typename T>
__device__ __forceinline__ void head_core(const T lg[3], int nd, int cls, const float dest[3], T a[3], T nv[3], T R[16]) {
    float mx = val(lg[0]);
    for (int i = 1; i < nd; ++i) mx = fmaxf(mx, val(lg[i]));
    T sum = lift<T>(0.f);
    for (int i = 0; i < 3; ++i) a[i] = nv[i] = lift<T>(0.f);
    for (int i = 0; i < nd; ++i) { a[i] = t_exp(lg[i] - mx); sum = sum + a[i]; }
    T n2 = lift<T>(0.f);
    for (int i = 0; i < nd; ++i) { a[i] = a[i] / sum; n2 = n2 + a[i] * a[i]; }
    const T nrm = t_sqrt(n2);
    for (int i = 0; i < nd; ++i) {
        a[i] = a[i] / nrm;
        nv[i] = a[i] * (((cls >> (nd - 1 - i)) & 1) ? 1.f : -1.f);
    }
    rotation_between<T>(nv, dest, R);
}

__device__ __forceinline__ int first_argmax(const float *__restrict__ x, int n) {
    int cls = 0;
    float best = x[0];
    for (int i = 1; i < n; ++i) { const float v = x[i]; if (v > best) { best = v; cls = i; } }
    return cls;
}

__global__ void k_head_normal(const float *__restrict__ abs_logits, long long lda, const float *__restrict__ sgn_logits,
                              long long lds, int B, int nd, float dx, float dy, float dz, float *__restrict__ abs_out,
                              float *__restrict__ normal, float *__restrict__ R) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float lg[3] = {0.f, 0.f, 0.f}, a[3], nv[3];
    for (int i = 0; i < nd; ++i) lg[i] = abs_logits[b * lda + i];
    const int cls = first_argmax(sgn_logits + b * lds, 1 << nd);
    const float d[3] = {dx, dy, dz};
    head_core<float>(lg, nd, cls, d, a, nv, R + (long long)b * 16);
    for (int i = 0; i < nd; ++i) { abs_out[b * nd + i] = a[i]; normal[b * nd + i] = nv[i]; }
}

// d loss / d abs_logits[b][i] = <g_abs, d abs/d l_i> + <g_normal, d normal/d l_i> + <g_R, d R/d l_i>   (any g may be NULL)
__global__ void k_head_normal_bwd(const float *__restrict__ abs_logits, long long lda, const float *__restrict__ sgn_logits,
                                  long long lds, int B, int nd, float dx, float dy, float dz, const float *__restrict__ g_abs,
                                  const float *__restrict__ g_normal, const float *__restrict__ g_R,
                                  float *__restrict__ g_logits) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int cls = first_argmax(sgn_logits + b * lds, 1 << nd);
    const float d[3] = {dx, dy, dz};
    for (int seed = 0; seed < nd; ++seed) {
        Dual lg[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}}, a[3], nv[3], R[16];
        for (int i = 0; i < nd; ++i) lg[i] = {abs_logits[b * lda + i], i == seed ? 1.f : 0.f};
        head_core<Dual>(lg, nd, cls, d, a, nv, R);
        float g = 0.f;
        for (int i = 0; i < nd; ++i) {
            if (g_abs) g += g_abs[b * nd + i] * a[i].d;
            if (g_normal) g += g_normal[b * nd + i] * nv[i].d;
        }
        if (g_R)
            for (int i = 0; i < 16; ++i) g += g_R[(long long)b * 16 + i] * R[i].d;
        g_logits[b * nd + seed] = g;
    }
}

// rotation of unit vectors onto a constant (ground-truth poses of the loss, loss_utils.py:25-58)
__global__ void k_rotation_between(const float *__restrict__ src, int B, float dx, float dy, float dz, float *__restrict__ R) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float v1[3] = {src[b * 3], src[b * 3 + 1], src[b * 3 + 2]}, d[3] = {dx, dy, dz};
    rotation_between<float>(v1, d, R + (long long)b * 16);
}

// fnet.py:87-91: peak of the correlation (FIRST maximum) -> yaw -> (cos, sin, 0) -> rotation onto e1; one wave per sample
__global__ void __launch_bounds__(64)
k_head_yaw(const float *__restrict__ score, long long lds, int B, int n, float *__restrict__ R) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int idx = n;
    float best = -INFINITY;
    for (int i = lane; i < n; i += 64) { const float v = score[b * lds + i]; if (v > best) { best = v; idx = i; } }
    for (int o = 32; o; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(idx, o);
        if (ob > best || (ob == best && oi < idx)) { best = ob; idx = oi; }
    }
    if (lane) return;
    if (idx >= n) idx = 0;                               // all NaN: argmax convention of the scalar loop
    const float f_rad = -((float)idx / (float)(n - 1)) * 2.f * 3.14159274101257324f + 3.14159274101257324f;   // float32, as the tensor expression
    const double rad = (double)f_rad;
    const float v1[3] = {(float)cos(rad), (float)sin(rad), 0.f};
    const float e1[3] = {1.f, 0.f, 0.f};
    rotation_between<float>(v1, e1, R + (long long)b * 16);
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

// out = op(a) op(b) for [B][4][4] matrices (pose accumulation sensor2_T_sensor1 <- f_l sensor2_T_sensor1, fnet.py:101, gnet.py:180)
__global__ void k_mat44_mul(const float *__restrict__ a, const float *__restrict__ b, int B, int ta, int tb, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 16) return;
    const int s = i >> 4, r = (i >> 2) & 3, c = i & 3;
    const float *A = a + s * 16, *Bm = b + s * 16;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float x = ta ? A[k * 4 + r] : A[r * 4 + k], y = tb ? Bm[c * 4 + k] : Bm[k * 4 + c];
        acc = k == 0 ? x * y : acc + x * y;
    }
    out[i] = acc;
}

// backward of k_cam_T_velo: out = A^-1 (c_T (P l_T)), P = A calib  ->  g_cT = (A^-T g) (P l_T)^T,  g_lT = P^T c_T^T (A^-T g)
__global__ void k_cam_T_velo_bwd(const float *__restrict__ cT, long long ldc, const float *__restrict__ lT,
                                 const float *__restrict__ calib, const float *__restrict__ A, const float *__restrict__ g_out,
                                 int B, float *__restrict__ g_cT, float *__restrict__ g_lT) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float *c = cT + b * ldc, *l = lT + (long long)b * 16, *k = calib + (long long)b * 12, *a = A + (long long)b * 9;
    const float *g = g_out + (long long)b * 12;
    float P[3][4], m1[3][4], gm2[3][4], gm1[3][4];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) P[i][j] = (a[i * 3 + 0] * k[0 * 4 + j] + a[i * 3 + 1] * k[1 * 4 + j]) + a[i * 3 + 2] * k[2 * 4 + j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j)
            m1[i][j] = ((P[i][0] * l[0 * 4 + j] + P[i][1] * l[1 * 4 + j]) + P[i][2] * l[2 * 4 + j]) + P[i][3] * l[3 * 4 + j];
    const float c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
    const float det = (a[0] * c00 + a[1] * c01) + a[2] * c02;
    const float inv[3][3] = {{c00 / det, (a[2] * a[7] - a[1] * a[8]) / det, (a[1] * a[5] - a[2] * a[4]) / det},
                             {c01 / det, (a[0] * a[8] - a[2] * a[6]) / det, (a[2] * a[3] - a[0] * a[5]) / det},
                             {c02 / det, (a[1] * a[6] - a[0] * a[7]) / det, (a[0] * a[4] - a[1] * a[3]) / det}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) gm2[i][j] = (inv[0][i] * g[0 * 4 + j] + inv[1][i] * g[1 * 4 + j]) + inv[2][i] * g[2 * 4 + j];
    if (g_cT)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                g_cT[(long long)b * 9 + i * 3 + j] = ((gm2[i][0] * m1[j][0] + gm2[i][1] * m1[j][1]) + gm2[i][2] * m1[j][2]) + gm2[i][3] * m1[j][3];
    if (g_lT) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) gm1[i][j] = (c[0 * 3 + i] * gm2[0][j] + c[1 * 3 + i] * gm2[1][j]) + c[2 * 3 + i] * gm2[2][j];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                g_lT[(long long)b * 16 + i * 4 + j] = (P[0][i] * gm1[0][j] + P[1][i] * gm1[1][j]) + P[2][i] * gm1[2][j];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Pose terms of the loss (losses/loss_utils.py:25-58 E, :227-262 H, :77-144 F, :165-185 G translation; efghloss.py:19-38):
// ground-truth construction + cosine / cross-entropy / hard-negative-mined BCE / smooth-L1 terms, forward and backward.
// One workgroup (one wave) per sample; lane 0 does the per-sample pose algebra, all lanes share the correlation row.
struct PoseLossArgs {
    const float *e_abs, *e_sgn, *h_abs, *h_sgn, *f_score, *g_trs, *e_l, *f_l;     // predictions
    long long ld_esgn, ld_hsgn, ld_fs;
    const float *rand_l, *rand_c, *T4;                                              // ground truth: rotations (3x3 or 4x4), 4x4
    int rl_b, rl_r, rc_b, rc_r;                                                     // sample / row pitch of rand_l, rand_c
    int B, W, pos_num;
    float neg_ratio;
};
enum { GT_E_GN = 0, GT_E_L = 3, GT_H_HRZN = 19, GT_H_C = 22, GT_F_L = 31, GT_G_TRS = 47, GT_G_L = 50, GT_E_ABS = 66,
       GT_H_ABS = 69, GT_LD = 72 };
enum { PT_COS_E = 0, PT_CE_E, PT_COS_H, PT_CE_H, PT_SL1, PT_FOV_SUM, PT_FOV_CNT, PT_LD };

template <typename T> __device__ __forceinline__ T cosine3(const T x[3], const float y[3], int n) {
    // F.cosine_similarity: sum(x / max(|x|, eps) * y / max(|y|, eps)), eps = 1e-8
    T nx = lift<T>(0.f);
    float ny = 0.f;
    for (int i = 0; i < n; ++i) { nx = nx + x[i] * x[i]; ny += y[i] * y[i]; }
    nx = t_sqrt(nx);
    ny = sqrtf(ny);
    if (val(nx) < 1e-8f) nx = lift<T>(1e-8f);
    if (ny < 1e-8f) ny = 1e-8f;
    T acc = lift<T>(0.f);
    for (int i = 0; i < n; ++i) acc = acc + (x[i] / nx) * (y[i] / ny);
    return acc;
}

template <typename T> __device__ __forceinline__ T cross_entropy(const T *x, int n, int cls) {
    float mx = val(x[0]);
    for (int i = 1; i < n; ++i) mx = fmaxf(mx, val(x[i]));
    T sum = lift<T>(0.f);
    for (int i = 0; i < n; ++i) sum = sum + t_exp(x[i] - mx);
    return (t_log(sum) + mx) - x[cls];
}

// x = A^-1 e_col for a 4x4 (Gaussian elimination, partial pivoting on the values)
template <typename T> __device__ __forceinline__ void solve4(const T A_[16], int col, T x[4]) {
    T M[4][5];
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 4; ++j) M[i][j] = A_[i * 4 + j];
        M[i][4] = lift<T>(i == col ? 1.f : 0.f);
    }
    for (int k = 0; k < 4; ++k) {
        int p = k;
        for (int i = k + 1; i < 4; ++i) if (fabsf(val(M[i][k])) > fabsf(val(M[p][k]))) p = i;
        if (p != k) for (int j = 0; j < 5; ++j) { const T t = M[k][j]; M[k][j] = M[p][j]; M[p][j] = t; }
        for (int i = k + 1; i < 4; ++i) {
            const T f = M[i][k] / M[k][k];
            for (int j = k; j < 5; ++j) M[i][j] = M[i][j] - f * M[k][j];
        }
    }
    for (int i = 3; i >= 0; --i) {
        T acc = M[i][4];
        for (int j = i + 1; j < 4; ++j) acc = acc - M[i][j] * x[j];
        x[i] = acc / M[i][i];
    }
}

__device__ __forceinline__ void inv3(const float *m, long long ld, float o[3][3]) {
    const float a = m[0], b = m[1], c = m[2], d = m[ld], e = m[ld + 1], f = m[ld + 2], g = m[2 * ld], h = m[2 * ld + 1], i = m[2 * ld + 2];
    const float c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
    const float det = (a * c00 + b * c01) + c * c02;
    o[0][0] = c00 / det; o[0][1] = (c * h - b * i) / det; o[0][2] = (b * f - c * e) / det;
    o[1][0] = c01 / det; o[1][1] = (a * i - c * g) / det; o[1][2] = (c * d - a * f) / det;
    o[2][0] = c02 / det; o[2][1] = (b * g - a * h) / det; o[2][2] = (a * e - b * d) / det;
}

// ground truth of one sample (no derivative): everything the loss dictionary `gt` gains, plus the sign classes
struct PoseGt { float e_gn[3], e_l[16], h_hrzn[3], h_c16[16], f_l[16], g_l[16], e_absv[3], h_absv[2]; int cls_e, cls_h, xmin; };

__device__ void pose_gt(const PoseLossArgs &a, int b, PoseGt &o) {
    const float *L = a.rand_l + (long long)b * a.rl_b, *C = a.rand_c + (long long)b * a.rc_b, *T4 = a.T4 + (long long)b * 16;
    {   // E: third column of R_l, normalised; rotation onto e3
        float g[3] = {L[2], L[a.rl_r + 2], L[2 * a.rl_r + 2]};
        const float n = sqrtf((g[0] * g[0] + g[1] * g[1]) + g[2] * g[2]);
        for (int i = 0; i < 3; ++i) { g[i] = g[i] / n; o.e_gn[i] = g[i]; o.e_absv[i] = fabsf(g[i]); }
        const float e3[3] = {0.f, 0.f, 1.f};
        rotation_between<float>(g, e3, o.e_l);
        o.cls_e = (g[0] > 0.f ? 4 : 0) + (g[1] > 0.f ? 2 : 0) + (g[2] > 0.f ? 1 : 0);
    }
    {   // H: second column of R_c
        float g[3] = {C[1], C[a.rc_r + 1], C[2 * a.rc_r + 1]};
        const float n = sqrtf((g[0] * g[0] + g[1] * g[1]) + g[2] * g[2]);
        for (int i = 0; i < 3; ++i) { g[i] = g[i] / n; o.h_hrzn[i] = g[i]; }
        o.h_absv[0] = fabsf(g[0]); o.h_absv[1] = fabsf(g[1]);
        const float e2[3] = {0.f, 1.f, 0.f};
        rotation_between<float>(g, e2, o.h_c16);
        o.cls_h = (g[0] > 0.f ? 2 : 0) + (g[1] > 0.f ? 1 : 0);
    }
    float Tinv[3][3];
    inv3(T4, 4, Tinv);
    {   // F: yaw of (pred e_l * T^-1) e1 -> first positive column; f_l = (gt e_l * T^-1)^-1
        const float *pe = a.e_l + (long long)b * 16;
        float ax[2];
        for (int i = 0; i < 2; ++i) ax[i] = (pe[i * 4 + 0] * Tinv[0][0] + pe[i * 4 + 1] * Tinv[1][0]) + pe[i * 4 + 2] * Tinv[2][0];
        const float yaw = atan2f(ax[1], ax[0]);
        const float f_idx = ((-yaw + 3.14159274101257324f) / (2.f * 3.14159274101257324f)) * (float)a.W;
        o.xmin = (int)f_idx - a.pos_num / 2;
        float m[9], fi[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                m[i * 3 + j] = (o.e_l[i * 4 + 0] * Tinv[0][j] + o.e_l[i * 4 + 1] * Tinv[1][j]) + o.e_l[i * 4 + 2] * Tinv[2][j];
        inv3(m, 3, fi);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) o.f_l[i * 4 + j] = (i < 3 && j < 3) ? fi[i][j] : (i == j ? 1.f : 0.f);
    }
    {   // G: translation matrix of T4 (gt f_l gt e_l)^-1 origin
        float gef[16], x[4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                gef[i * 4 + j] = ((o.f_l[i * 4 + 0] * o.e_l[0 * 4 + j] + o.f_l[i * 4 + 1] * o.e_l[1 * 4 + j]) + o.f_l[i * 4 + 2] * o.e_l[2 * 4 + j]) + o.f_l[i * 4 + 3] * o.e_l[3 * 4 + j];
        solve4<float>(gef, 3, x);
        for (int i = 0; i < 16; ++i) o.g_l[i] = (i % 5 == 0) ? 1.f : 0.f;
        for (int i = 0; i < 3; ++i)
            o.g_l[i * 4 + 3] = ((T4[i * 4 + 0] * x[0] + T4[i * 4 + 1] * x[1]) + T4[i * 4 + 2] * x[2]) + T4[i * 4 + 3] * x[3];
    }
}

// the differentiable per-sample terms: 1 - cos (E, H), cross-entropy (E, H), sum of smooth-L1 over the 3 translation components
template <typename T>
__device__ void pose_terms(const PoseLossArgs &a, int b, const PoseGt &gt, const T e_abs[3], const T e_sgn[8], const T h_abs[3],
                           const T h_sgn[4], const T g_trs[3], const T e_l[16], T out[5], float gtrs_gt[3]) {
    out[PT_COS_E] = 1.f - cosine3<T>(e_abs, gt.e_absv, 3);
    out[PT_CE_E] = cross_entropy<T>(e_sgn, 8, gt.cls_e);
    out[PT_COS_H] = 1.f - cosine3<T>(h_abs, gt.h_absv, 2);
    out[PT_CE_H] = cross_entropy<T>(h_sgn, 4, gt.cls_h);
    // gt g_trs = (T4 (pred f_l pred e_l)^-1 origin)[:3]: built from UN-detached predictions (loss_utils.py:170-175)
    const float *fl = a.f_l + (long long)b * 16, *T4 = a.T4 + (long long)b * 16;
    T pef[16], x[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            pef[i * 4 + j] = ((fl[i * 4 + 0] * e_l[0 * 4 + j] + fl[i * 4 + 1] * e_l[1 * 4 + j]) + fl[i * 4 + 2] * e_l[2 * 4 + j]) + fl[i * 4 + 3] * e_l[3 * 4 + j];
    solve4<T>(pef, 3, x);
    T acc = lift<T>(0.f);
    for (int i = 0; i < 3; ++i) {
        const T t = ((T4[i * 4 + 0] * x[0] + T4[i * 4 + 1] * x[1]) + T4[i * 4 + 2] * x[2]) + T4[i * 4 + 3] * x[3];
        gtrs_gt[i] = val(t);
        const T d = t - g_trs[i];
        const T ad = t_abs(d);
        acc = acc + (val(ad) < 1.f ? 0.5f * d * d : ad - 0.5f);
    }
    out[PT_SL1] = acc;
}

template <typename T> __device__ __forceinline__ void load_pred(const PoseLossArgs &a, int b, T e_abs[3], T e_sgn[8], T h_abs[3],
                                                                T h_sgn[4], T g_trs[3], T e_l[16]) {
    for (int i = 0; i < 3; ++i) { e_abs[i] = lift<T>(a.e_abs[b * 3 + i]); g_trs[i] = lift<T>(a.g_trs[b * 3 + i]); }
    for (int i = 0; i < 8; ++i) e_sgn[i] = lift<T>(a.e_sgn[b * a.ld_esgn + i]);
    h_abs[0] = lift<T>(a.h_abs[b * 2]); h_abs[1] = lift<T>(a.h_abs[b * 2 + 1]); h_abs[2] = lift<T>(0.f);
    for (int i = 0; i < 4; ++i) h_sgn[i] = lift<T>(a.h_sgn[b * a.ld_hsgn + i]);
    for (int i = 0; i < 16; ++i) e_l[i] = lift<T>(a.e_l[(long long)b * 16 + i]);
}

__device__ __forceinline__ float bce(float p, float t) {
    return -(t * fmaxf(logf(p), -100.f) + (1.f - t) * fmaxf(logf(1.f - p), -100.f));
}

// forward: gt buffers, the selection mask of the mined BCE, per-sample partial terms
#define PL_TPB 256
__device__ __forceinline__ float block_sum(float v, float *red) {      // PL_TPB threads; every thread gets the total
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < PL_TPB / 64; ++w) t += red[w];
    return t;
}

// grid (ceil(W / PL_TPB), B): every workgroup ranks PL_TPB columns of one sample's score row against the whole row (staged in LDS)
__global__ void __launch_bounds__(PL_TPB)
k_pose_loss_fwd(PoseLossArgs a, float *__restrict__ gtbuf, long long *__restrict__ gtcls, float *__restrict__ gt_fscore,
                float *__restrict__ wsel, float *__restrict__ part, float *__restrict__ fpart) {
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    __shared__ PoseGt gt;
    __shared__ float red[PL_TPB / 64];
    extern __shared__ float lc[];                      // [W] BCE of the non-positive columns against 0, positives as 0
    if (tid == 0) {
        pose_gt(a, b, gt);
        if (chunk == 0) {
            float e_abs[3], e_sgn[8], h_abs[3], h_sgn[4], g_trs[3], e_l[16], out[5], gtrs[3];
            load_pred<float>(a, b, e_abs, e_sgn, h_abs, h_sgn, g_trs, e_l);
            pose_terms<float>(a, b, gt, e_abs, e_sgn, h_abs, h_sgn, g_trs, e_l, out, gtrs);
            float *g = gtbuf + (long long)b * GT_LD;
            for (int i = 0; i < 3; ++i) { g[GT_E_GN + i] = gt.e_gn[i]; g[GT_H_HRZN + i] = gt.h_hrzn[i]; g[GT_G_TRS + i] = gtrs[i]; g[GT_E_ABS + i] = gt.e_absv[i]; }
            g[GT_H_ABS] = gt.h_absv[0]; g[GT_H_ABS + 1] = gt.h_absv[1]; g[GT_H_ABS + 2] = 0.f;
            for (int i = 0; i < 16; ++i) { g[GT_E_L + i] = gt.e_l[i]; g[GT_F_L + i] = gt.f_l[i]; g[GT_G_L + i] = gt.g_l[i]; }
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) g[GT_H_C + i * 3 + j] = gt.h_c16[i * 4 + j];
            gtcls[b * 2] = gt.cls_e; gtcls[b * 2 + 1] = gt.cls_h;
            for (int i = 0; i < 5; ++i) part[b * PT_LD + i] = out[i];
        }
    }
    __syncthreads();
    // F: positives = pos_num columns from xmin (wrapping); negatives = the neg_ratio * #pos largest BCE values among the rest
    const int W = a.W, xmin = gt.xmin;
    const float *p = a.f_score + b * a.ld_fs;
    float npos_f = 0.f;
    for (int j = tid; j < W; j += PL_TPB) {
        int r = (j - xmin) % W; if (r < 0) r += W;
        const bool pos = r < a.pos_num;
        lc[j] = pos ? 0.f : bce(p[j], 0.f);
        npos_f += pos ? 1.f : 0.f;
    }
    npos_f = block_sum(npos_f, red);                   // (the barriers inside also publish lc)
    const float num_neg = fminf(a.neg_ratio * npos_f, (float)(W - 1));
    float lsum = 0.f, cnt = 0.f;
    const int j = chunk * PL_TPB + tid;
    if (j < W) {
        int r = (j - xmin) % W; if (r < 0) r += W;
        const bool pos = r < a.pos_num;
        const float lj = lc[j];
        int rank = 0;                                   // position in the descending sort (ties: lower column first)
        for (int q = 0; q < W; ++q) { const float lq = lc[q]; rank += (lq > lj) || (lq == lj && q < j); }
        const bool sel = pos || (float)rank < num_neg;
        gt_fscore[(long long)b * W + j] = pos ? 1.f : 0.f;
        wsel[(long long)b * W + j] = sel ? 1.f : 0.f;
        if (sel) { lsum = bce(p[j], pos ? 1.f : 0.f); cnt = 1.f; }
    }
    lsum = block_sum(lsum, red);
    cnt = block_sum(cnt, red);
    if (tid == 0) { fpart[((long long)b * gridDim.x + chunk) * 2] = lsum; fpart[((long long)b * gridDim.x + chunk) * 2 + 1] = cnt; }
}

// L[11] in efghloss.py:13-17 order: total, e_gn, e_gn_sgn, e_gn_abs, h_hrzn, h_hrzn_abs, h_hrzn_sgn, fov, g_trs, g_depth, g_mask
struct PoseLossLambda { float e_gn, h_hrzn, fov, g_trs, g_depth, g_mask; };
enum { L_TOTAL = 0, L_E_GN, L_E_SGN, L_E_ABS, L_H, L_H_ABS, L_H_SGN, L_FOV, L_G_TRS, L_G_DEPTH, L_G_MASK, L_N };

__global__ void k_pose_loss_finish(const float *__restrict__ part, const float *__restrict__ fpart, int nchunk, int B,
                                   PoseLossLambda lam, const float *__restrict__ l_dep, const float *__restrict__ l_msk,
                                   float *__restrict__ L, float *__restrict__ nsel) {
    if (threadIdx.x || blockIdx.x) return;
    float s[PT_LD] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < B; ++b) {
        for (int i = 0; i < 5; ++i) s[i] += part[b * PT_LD + i];
        for (int c = 0; c < nchunk; ++c) {              // fixed order
            s[PT_FOV_SUM] += fpart[((long long)b * nchunk + c) * 2];
            s[PT_FOV_CNT] += fpart[((long long)b * nchunk + c) * 2 + 1];
        }
    }
    const float la_e = s[PT_COS_E] / (float)B * 10.f, ls_e = s[PT_CE_E] / (float)B;
    const float la_h = s[PT_COS_H] / (float)B * 10.f, ls_h = s[PT_CE_H] / (float)B;
    L[L_E_GN] = (la_e + ls_e) * lam.e_gn; L[L_E_ABS] = la_e * lam.e_gn; L[L_E_SGN] = ls_e * lam.e_gn;
    L[L_H] = (la_h + ls_h) * lam.h_hrzn; L[L_H_ABS] = la_h * lam.h_hrzn; L[L_H_SGN] = ls_h * lam.h_hrzn;
    L[L_FOV] = s[PT_FOV_SUM] / s[PT_FOV_CNT] * lam.fov;
    L[L_G_TRS] = s[PT_SL1] / (float)(3 * B) * lam.g_trs;
    L[L_G_DEPTH] = l_dep[0] * lam.g_depth;
    L[L_G_MASK] = (l_msk[0] * lam.g_mask) * lam.g_depth;            // scaled twice (loss_utils.py:199,204)
    float total = 0.f;                                              // efghloss.py:33-36: every entry, in insertion order
    const int order[10] = {L_E_GN, L_E_ABS, L_E_SGN, L_H, L_H_ABS, L_H_SGN, L_FOV, L_G_TRS, L_G_DEPTH, L_G_MASK};
    for (int i = 0; i < 10; ++i) total = total + L[order[i]];
    L[L_TOTAL] = total;
    nsel[0] = s[PT_FOV_CNT];
}

// backward: lanes 0..35 each seed one of the 36 differentiable per-sample inputs; all lanes share the correlation row
__global__ void __launch_bounds__(64)
k_pose_loss_bwd(PoseLossArgs a, PoseLossLambda lam, const float *__restrict__ gL, const float *__restrict__ wsel,
                const float *__restrict__ nsel, float *__restrict__ g_e_abs, float *__restrict__ g_e_sgn,
                float *__restrict__ g_h_abs, float *__restrict__ g_h_sgn, float *__restrict__ g_fscore,
                float *__restrict__ g_gtrs, float *__restrict__ g_e_l, float *__restrict__ g_ldep_lmsk) {
    const int b = blockIdx.x, lane = threadIdx.x;
    __shared__ PoseGt gt;
    if (lane == 0) pose_gt(a, b, gt);
    __syncthreads();
    const float gt_ = gL[L_TOTAL];
    const float w_la_e = lam.e_gn * ((gL[L_E_GN] + gL[L_E_ABS]) + 2.f * gt_) * 10.f / (float)a.B;
    const float w_ls_e = lam.e_gn * ((gL[L_E_GN] + gL[L_E_SGN]) + 2.f * gt_) / (float)a.B;
    const float w_la_h = lam.h_hrzn * ((gL[L_H] + gL[L_H_ABS]) + 2.f * gt_) * 10.f / (float)a.B;
    const float w_ls_h = lam.h_hrzn * ((gL[L_H] + gL[L_H_SGN]) + 2.f * gt_) / (float)a.B;
    const float w_trs = lam.g_trs * (gL[L_G_TRS] + gt_) / (float)(3 * a.B);
    const float w_fov = lam.fov * (gL[L_FOV] + gt_) / nsel[0];
    if (b == 0 && lane == 0) {
        g_ldep_lmsk[0] = lam.g_depth * (gL[L_G_DEPTH] + gt_);
        g_ldep_lmsk[1] = lam.g_mask * lam.g_depth * (gL[L_G_MASK] + gt_);
    }
    if (lane < 36) {
        Dual e_abs[3], e_sgn[8], h_abs[3], h_sgn[4], g_trs[3], e_l[16], out[5];
        float gtrs[3];
        load_pred<Dual>(a, b, e_abs, e_sgn, h_abs, h_sgn, g_trs, e_l);
        float *dst;
        if (lane < 3) { e_abs[lane].d = 1.f; dst = g_e_abs + b * 3 + lane; }
        else if (lane < 11) { e_sgn[lane - 3].d = 1.f; dst = g_e_sgn + b * 8 + (lane - 3); }
        else if (lane < 13) { h_abs[lane - 11].d = 1.f; dst = g_h_abs + b * 2 + (lane - 11); }
        else if (lane < 17) { h_sgn[lane - 13].d = 1.f; dst = g_h_sgn + b * 4 + (lane - 13); }
        else if (lane < 20) { g_trs[lane - 17].d = 1.f; dst = g_gtrs + b * 3 + (lane - 17); }
        else { e_l[lane - 20].d = 1.f; dst = g_e_l + (long long)b * 16 + (lane - 20); }
        pose_terms<Dual>(a, b, gt, e_abs, e_sgn, h_abs, h_sgn, g_trs, e_l, out, gtrs);
        *dst = (((w_la_e * out[PT_COS_E].d + w_ls_e * out[PT_CE_E].d) + w_la_h * out[PT_COS_H].d) + w_ls_h * out[PT_CE_H].d) +
               w_trs * out[PT_SL1].d;
    }
    // d BCE / d p = (p - t) / max((1 - p) p, 1e-12)   (aten binary_cross_entropy_backward)
    const float *p = a.f_score + b * a.ld_fs;
    for (int j = lane; j < a.W; j += 64) {
        int r = (j - gt.xmin) % a.W; if (r < 0) r += a.W;
        const float t = r < a.pos_num ? 1.f : 0.f, pj = p[j];
        g_fscore[(long long)b * a.W + j] = wsel[(long long)b * a.W + j] * w_fov * (pj - t) / fmaxf((1.f - pj) * pj, 1e-12f);
    }
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
    k_head_yaw<<<B, 64, 0, (hipStream_t)stream_>>>(score, lds, B, n, R44);
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

extern "C" int efgh_pose_head_normal_bwd(const float *abs_logits, int64_t lda, const float *sgn_logits, int64_t lds, int32_t B,
                                         int32_t nd, float dx, float dy, float dz, const float *g_abs, const float *g_normal,
                                         const float *g_R44, float *g_abs_logits, void *stream_) {
    EFGH_CHECK_ARG(abs_logits && sgn_logits && g_abs_logits && B > 0 && (nd == 2 || nd == 3));
    k_head_normal_bwd<<<cdiv(B, 64), 64, 0, (hipStream_t)stream_>>>(abs_logits, lda, sgn_logits, lds, B, nd, dx, dy, dz, g_abs,
                                                                   g_normal, g_R44, g_abs_logits);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_pose_rotation_between(const float *src3, int32_t B, float dx, float dy, float dz, float *R44, void *stream_) {
    EFGH_CHECK_ARG(src3 && R44 && B > 0);
    k_rotation_between<<<cdiv(B, 64), 64, 0, (hipStream_t)stream_>>>(src3, B, dx, dy, dz, R44);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_pose_cam_T_velo_bwd(const float *c_T, int64_t ldc, const float *l_T, const float *calib, const float *A,
                                        const float *g_out34, int32_t B, float *g_cT33, float *g_lT44, void *stream_) {
    EFGH_CHECK_ARG(c_T && l_T && calib && A && g_out34 && (g_cT33 || g_lT44) && B > 0 && ldc >= 9);
    k_cam_T_velo_bwd<<<cdiv(B, 64), 64, 0, (hipStream_t)stream_>>>(c_T, ldc, l_T, calib, A, g_out34, B, g_cT33, g_lT44);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

static PoseLossArgs pose_args(const efgh_pose_loss_desc *d) {
    PoseLossArgs a;
    a.e_abs = d->e_gn_abs; a.e_sgn = d->e_gn_sgn; a.h_abs = d->h_hrzn_abs; a.h_sgn = d->h_hrzn_sgn; a.f_score = d->f_score;
    a.g_trs = d->g_trs; a.e_l = d->e_l; a.f_l = d->f_l;
    a.ld_esgn = d->ld_e_gn_sgn; a.ld_hsgn = d->ld_h_hrzn_sgn; a.ld_fs = d->ld_f_score;
    a.rand_l = d->rand_init_l; a.rand_c = d->rand_init_c; a.T4 = d->sensor2_T_sensor1;
    a.rl_r = d->rand_init_l_dim; a.rl_b = a.rl_r * a.rl_r; a.rc_r = d->rand_init_c_dim; a.rc_b = a.rc_r * a.rc_r;
    a.B = d->B; a.W = d->W; a.pos_num = d->fov_pos_num; a.neg_ratio = d->fov_neg_ratio;
    return a;
}

static bool pose_desc_ok(const efgh_pose_loss_desc *d) {
    return d && d->e_gn_abs && d->e_gn_sgn && d->h_hrzn_abs && d->h_hrzn_sgn && d->f_score && d->g_trs && d->e_l && d->f_l &&
           d->rand_init_l && d->rand_init_c && d->sensor2_T_sensor1 && d->W <= 8192 && (d->rand_init_l_dim == 3 || d->rand_init_l_dim == 4) &&
           (d->rand_init_c_dim == 3 || d->rand_init_c_dim == 4) && d->B > 0 && d->W > 1 && d->fov_pos_num > 0 &&
           d->ld_e_gn_sgn >= 8 && d->ld_h_hrzn_sgn >= 4 && d->ld_f_score >= d->W;
}

extern "C" int efgh_pose_loss_fwd(const efgh_pose_loss_desc *d, const float *l_depth, const float *l_mask, float *gt72,
                                  int64_t *gt_cls2, float *gt_f_score, float *selected, float *partials, float *L11,
                                  float *n_selected, void *stream_) {
    EFGH_CHECK_ARG(pose_desc_ok(d) && l_depth && l_mask && gt72 && gt_cls2 && gt_f_score && selected && partials && L11 && n_selected);
    hipStream_t st = (hipStream_t)stream_;
    const int nchunk = cdiv(d->W, PL_TPB);
    float *fpart = partials + (size_t)d->B * PT_LD;   // [B][nchunk][2] behind the per-sample pose terms
    k_pose_loss_fwd<<<dim3(nchunk, d->B), PL_TPB, (size_t)d->W * sizeof(float), st>>>(pose_args(d), gt72, (long long *)gt_cls2,
                                                                                       gt_f_score, selected, partials, fpart);
    const PoseLossLambda lam = {d->lambda_e_gn, d->lambda_h_hrzn, d->lambda_fov, d->lambda_g_trs, d->lambda_g_depth, d->lambda_g_mask};
    k_pose_loss_finish<<<1, 64, 0, st>>>(partials, fpart, nchunk, d->B, lam, l_depth, l_mask, L11, n_selected);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_pose_loss_bwd(const efgh_pose_loss_desc *d, const float *g_L11, const float *selected, const float *n_selected,
                                  float *g_e_gn_abs, float *g_e_gn_sgn, float *g_h_hrzn_abs, float *g_h_hrzn_sgn,
                                  float *g_f_score, float *g_g_trs, float *g_e_l, float *g_ldepth_lmask, void *stream_) {
    EFGH_CHECK_ARG(pose_desc_ok(d) && g_L11 && selected && n_selected && g_e_gn_abs && g_e_gn_sgn && g_h_hrzn_abs && g_h_hrzn_sgn &&
                   g_f_score && g_g_trs && g_e_l && g_ldepth_lmask);
    const PoseLossLambda lam = {d->lambda_e_gn, d->lambda_h_hrzn, d->lambda_fov, d->lambda_g_trs, d->lambda_g_depth, d->lambda_g_mask};
    k_pose_loss_bwd<<<d->B, 64, 0, (hipStream_t)stream_>>>(pose_args(d), lam, g_L11, selected, n_selected, g_e_gn_abs, g_e_gn_sgn,
                                                          g_h_hrzn_abs, g_h_hrzn_sgn, g_f_score, g_g_trs, g_e_l, g_ldepth_lmask);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_pose_mat44_mul(const float *a, const float *b, int32_t B, int32_t transpose_a, int32_t transpose_b, float *out,
                                   void *stream_) {
    EFGH_CHECK_ARG(a && b && out && B > 0);
    k_mat44_mul<<<cdiv((int64_t)B * 16, 64), 64, 0, (hipStream_t)stream_>>>(a, b, B, transpose_a, transpose_b, out);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}
