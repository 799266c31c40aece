// Probe (not part of the product): can fp32 contractions run on the bf16 matrix cores of gfx950 with fp32-level accuracy?
//   x = x0 + x1 + x2 (three bf16 pieces, 24 mantissa bits), a.b ~= a0b0 + a0b1 + a1b0 + a0b2 + a1b1 + a2b0 (6 bf16 MFMAs, fp32 accumulate).
// Part 1: error of fp32 MFMA / x3 / x6 / x9 against a float64 host result (one wave, 32x32 output, K = 2048).
// Part 2: bare MFMA issue rate, fp32 32x32x2 vs bf16 32x32x16, all CUs.
// build: hipcc --offload-arch=gfx950 -O3 -o probe_bf16_split probe_bf16_split.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void split3(float x, __bf16 &p0, __bf16 &p1, __bf16 &p2) {
    p0 = (__bf16)x; float r = x - (float)p0;
    p1 = (__bf16)r; r -= (float)p1;
    p2 = (__bf16)r;
}

// A [32][K] row-major, B [K][32] row-major, C [32][32]; mode 0: fp32 MFMA, 3/6/9: number of bf16 products
__global__ void k_acc(const float *A, const float *B, float *C, int K, int mode) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    f16v acc = {0};
    if (mode == 0) {
        for (int k = 0; k < K; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], B[(k + h) * 32 + r], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            bf8 a[3], b[3];
            for (int j = 0; j < 8; ++j) {
                __bf16 p0, p1, p2;
                split3(A[r * K + k + 8 * h + j], p0, p1, p2); a[0][j] = p0; a[1][j] = p1; a[2][j] = p2;
                split3(B[(k + 8 * h + j) * 32 + r], p0, p1, p2); b[0][j] = p0; b[1][j] = p1; b[2][j] = p2;
            }
            // smallest terms first
            if (mode >= 9) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[1], acc, 0, 0, 0);
            }
            if (mode >= 6) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
        }
    }
    for (int q = 0; q < 16; ++q) C[((q & 3) + 8 * (q >> 2) + 4 * h) * 32 + r] = acc[q];
}

// mode 16: as mode 6 but the small products go to a second accumulator that is added at the end
__global__ void k_acc2(const float *A, const float *B, float *C, int K) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    f16v acc = {0}, lo = {0};
    for (int k = 0; k < K; k += 16) {
        bf8 a[3], b[3];
        for (int j = 0; j < 8; ++j) {
            __bf16 p0, p1, p2;
            split3(A[r * K + k + 8 * h + j], p0, p1, p2); a[0][j] = p0; a[1][j] = p1; a[2][j] = p2;
            split3(B[(k + 8 * h + j) * 32 + r], p0, p1, p2); b[0][j] = p0; b[1][j] = p1; b[2][j] = p2;
        }
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], lo, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    }
    for (int q = 0; q < 16; ++q) C[((q & 3) + 8 * (q >> 2) + 4 * h) * 32 + r] = acc[q] + lo[q];
}

template <int BF>
__global__ void __launch_bounds__(256) k_rate(float *out, int iters) {
    f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const float s = (float)threadIdx.x * 1e-3f;
    if (BF) {
        bf8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(s + j); b[j] = (__bf16)(s - j); }
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
        }
    } else {
        float a = s, b = 1.f - s;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
        }
    }
    float t = 0.f;
    for (int q = 0; q < 16; ++q) t += c0[q] + c1[q] + c2[q] + c3[q];
    if (t == 123.456f) out[0] = t;
}

static double urand(unsigned long long &s) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return ((s >> 11) * (1.0 / 9007199254740992.0)) * 2.0 - 1.0; }

int main() {
    const int K = 2048;
    for (int dist = 0; dist < 3; ++dist) {
        std::vector<float> A(32 * K), B(K * 32), C(32 * 32);
        unsigned long long s = 1234 + dist;
        for (auto &v : A) { double u = urand(s); v = (float)(dist == 0 ? u : dist == 1 ? u * std::exp(6.0 * urand(s)) : 1.0 + 0.01 * u); }
        for (auto &v : B) { double u = urand(s); v = (float)(dist == 0 ? u : dist == 1 ? u * std::exp(6.0 * urand(s)) : 1.0 + 0.01 * u); }
        std::vector<double> ref(32 * 32, 0.0), mag(32 * 32, 0.0);
        for (int i = 0; i < 32; ++i) for (int k = 0; k < K; ++k) for (int j = 0; j < 32; ++j) {
            ref[i * 32 + j] += (double)A[i * K + k] * B[k * 32 + j]; mag[i * 32 + j] += std::fabs((double)A[i * K + k] * B[k * 32 + j]); }
        float *dA, *dB, *dC;
        CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, C.size() * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        const int modes[5] = {0, 3, 6, 9, 16};
        for (int m : modes) {
            if (m == 16) k_acc2<<<1, 64>>>(dA, dB, dC, K); else k_acc<<<1, 64>>>(dA, dB, dC, K, m);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
            double e2 = 0, r2 = 0, emax = 0;     // error relative to sum |a.b| (the scale rounding errors live on)
            for (int i = 0; i < 1024; ++i) { double e = C[i] - ref[i]; e2 += e * e; r2 += ref[i] * ref[i]; emax = std::fmax(emax, std::fabs(e) / mag[i]); }
            printf("dist %d mode %2d: rel-L2 %.3e   max |err|/sum|ab| %.3e\n", dist, m, std::sqrt(e2 / r2), emax);
        }
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
    }
    float *out; CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4096, blocks = 256 * 8;
    for (int bf = 0; bf < 2; ++bf) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (bf) k_rate<1><<<blocks, 256>>>(out, iters); else k_rate<0><<<blocks, 256>>>(out, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double fl = (double)blocks * 4 * iters * 4 * 2.0 * 32 * 32 * (bf ? 16 : 2);
            if (rep) printf("%s MFMA rate: %.1f TFLOP/s (%.3f ms)\n", bf ? "bf16 32x32x16" : "fp32 32x32x2 ", fl / ms * 1e-9, ms);
        }
    }
    return 0;
}
