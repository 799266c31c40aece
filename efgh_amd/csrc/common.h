// Shared host-side helpers for libefgh_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>

#include "../../include/efgh_hip.h"

void efgh_set_error(const char *fmt, ...);

#define EFGH_CHECK_ARG(cond)                                                         \
    do {                                                                             \
        if (!(cond)) {                                                               \
            efgh_set_error("%s:%d: invalid argument: %s", __FILE__, __LINE__, #cond); \
            return EFGH_E_INVALID;                                                   \
        }                                                                            \
    } while (0)

#define EFGH_CHECK_LAUNCH()                                                          \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            efgh_set_error("%s:%d: launch failed: %s", __FILE__, __LINE__,           \
                           hipGetErrorString(e_));                                   \
            return EFGH_E_LAUNCH;                                                    \
        }                                                                            \
    } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// activation of the fused epilogues: v for v > 0, neg * v otherwise (neg = 1: none, 0: ReLU, slope: LeakyReLU) as one compare, one
// fma and one select - no branch.  A NaN stays a NaN under every activation (the max / min form max(v,0) + neg * min(v,0) returns
// the other operand and turns a diverged pre-activation into a finite 0).  For every non-NaN input the result has the bits of
// that form (neg * v + 0: a clamped value is +0, never -0).
__device__ __forceinline__ float act_neg(float v, float neg) { return v > 0.f ? v : fmaf(neg, v, 0.f); }

// streaming accesses of the HBM-bound passes (an activation that is read or written exactly once by the kernel).  NT = true: with
// the non-temporal hint (the line is not kept in L2 / MALL) - measured +5-10 % on the BatchNorm passes over tensors that do not
// fit the 256-MB MALL anyway (tools/bench_elementwise.py: reduce 5.25 -> 5.64 TB/s, normalise 5.02 -> 5.26 at 1 GB), and a LOSS
// on tensors that do (0.13 GB: 6.9 -> 6.0 TB/s) - so the host side picks it by size (efgh_stream_nt).
template <bool NT>
__device__ __forceinline__ float4 ld_stream(const float *p) {
    if (NT) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *reinterpret_cast<const float4 *>(p);
}
template <bool NT>
__device__ __forceinline__ void st_stream(float *p, const float4 &v) {
    if (NT) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
        __builtin_nontemporal_store(w, reinterpret_cast<v4f *>(p));
    } else {
        *reinterpret_cast<float4 *>(p) = v;
    }
}
// tensors of at least this many bytes stream past the caches (384 MB: beyond the 256-MB MALL with room for the other operand)
static inline bool efgh_stream_nt(long long bytes) { return bytes >= (384ll << 20); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting: one flag bit per device ordinal, set after the first
// successful call on that device (two host threads racing here both make the same idempotent call)
static inline bool efgh_raise_lds_once(std::atomic<unsigned long long> &done, const void *fn, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    done.fetch_or(bit, std::memory_order_release);
    return true;
}

// output stores of the MFMA kernels' epilogues (4-byte, one 128-B row segment per half-wave).  -DEFGH_MFMA_NT_OUT=1 gives them the
// non-temporal hint (an experiment switch: see DESIGN 7 for the measurement)
#ifndef EFGH_MFMA_NT_OUT
#define EFGH_MFMA_NT_OUT 0
#endif
__device__ __forceinline__ void st_out(float *p, float v) {
#if EFGH_MFMA_NT_OUT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// ---- LDS-DMA (global -> LDS without a VGPR round trip), used by planes.hip and the DMA-staged instances of gemm.hip ----
// one 1-KiB piece: lane l -> LDS base + 16 l  (base wave-uniform, passed in M0).  Inline assembly on purpose: hipcc orders every
// ds_read behind a pending `__builtin_amdgcn_global_load_lds` with `s_waitcnt vmcnt(0)` (it cannot tell the ring slots of one
// __shared__ object apart), which drains the prefetch at the top of every step - seen in the ISA of the first version of this
// file.  The compiler does not see these loads; the counted `s_waitcnt vmcnt` + barrier below are the only ordering, and the
// only other vector-memory operations of the kernels are the epilogue's stores, issued after the last wait.  M0 is saved and
// restored around the statement (the compiler owns it).
__device__ __forceinline__ void dma16(const float *src, unsigned lds_byte_addr) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_byte_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const float *p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)p;
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }


// ---- row chunks of a weight-gradient launch (the contraction runs over the rows m and is cut into `zs` chunks, one partial plane
// each, folded in chunk order).  `wg_per_cu` workgroups are resident per CU, 256 * wg_per_cu at a time; a grid that is a round and
// an eighth runs the eighth alone on its CUs (measured on the 512 x 512-channel planes: 97 TFLOP/s against 125 on grids that fill
// their rounds).  zs is chosen by a small cost model: rounds of resident workgroups (a partly filled last round costs between half
// and a whole one: its workgroups have their CU to themselves) x rows per chunk, plus the fold's traffic (2 x zs planes at ~5 TB/s)
// in row-equivalents.  `blocks`: workgroups per chunk, `plane_elems`: floats of one partial plane set, `unit`: row granularity.
static inline long long efgh_round_chunks(long long M, long long blocks, int wg_per_cu, int unit, long long min_chunk,
                                          double plane_elems, long long *chunk_out) {
    const long long slots = 256LL * wg_per_cu;
    const double fold_rows = plane_elems * 1.31e-5 * 2.0 / wg_per_cu;
    long long best_zs = 1, best_chunk = (M + unit - 1) / unit * unit;
    double best = -1.0;
    auto consider = [&](long long zs) {
        if (zs < 1) return;
        long long chunk = (M + zs - 1) / zs;
        chunk = (chunk + unit - 1) / unit * unit;
        if (zs > 1 && chunk < min_chunk) return;
        const long long z = (M + chunk - 1) / chunk;              // chunks that actually hold rows
        const long long wg = z * blocks, full = wg / slots, rem = wg % slots;
        const double rounds = (double)full + (rem ? 0.5 + 0.5 * (double)rem / (double)slots : 0.0);
        // (+ a workgroup's fixed part - row iterators, the 64-KB partial tile it stores - priced at 128 rows: without it the model
        // liked eight rounds of 25-step workgroups as much as one round of 200-step ones, and lost 30 % on k_gather_wgrad)
        const double cost = rounds * ((double)chunk + 128.0) + (z > 1 ? (double)z * fold_rows : 0.0);
        if (best < 0.0 || cost < best * 0.995) { best = cost; best_zs = z; best_chunk = chunk; }
    };
    // candidates: one chunk, and the chunk counts that just fill 1 .. 4 rounds (and their neighbours: the rounding of the chunk to
    // `unit` rows can drop one) - a dozen evaluations per launch, this runs on the host for every weight-gradient launch
    const long long zmax = M / min_chunk > 1 ? M / min_chunk : 1;        // the most chunks the minimum chunk size allows
    auto clamp = [&](long long z) { return z < 1 ? 1 : (z > zmax ? zmax : z); };
    consider(1);
    for (int r = 1; r <= 4; ++r) {
        const long long zs = r * slots / blocks;
        consider(clamp(zs)); consider(clamp(zs - 1)); consider(clamp(zs + 1));
    }
    consider(zmax); consider(clamp(zmax - 1)); consider(clamp(zmax / 2));
    if (chunk_out) *chunk_out = best_chunk;
    return best_zs;
}

// resident workgroups per CU of a kernel, asked of the runtime once per instance (register and LDS allocation decide it; reading
// it off the code object's register counts by hand was wrong once: round 5)
static inline int efgh_wg_per_cu(const void *kernel, int threads, size_t dyn_lds) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, dyn_lds) != hipSuccess || n < 1) n = 1;
    return n;
}

// ---- fold / finish + unpack in one launch: every weight-gradient entry point takes an optional `const efgh_wgrad_out_desc *out`
// (include/efgh_hip.h; ABI 3 - rounds 5's thread-local arm / disarm hand-off is gone: no state outlives a call) ----
// the fold of `zs` partial planes (wgrad.hip); with `out` describing exactly the folded [>= N rows][T][Cp] plane it writes the
// caller's layout itself and returns true, else the packed plane goes to dst and it returns false
bool efgh_launch_fold_splits(const float *part, int zs, long long total, float *dst, hipStream_t st, const efgh_wgrad_out_desc *out = nullptr);
// does `out` describe a [rows][T][Cp] packed gradient with identity taps (what a FINISH kernel can write through plain strides)?
static inline bool efgh_wgrad_out_fits(const efgh_wgrad_out_desc *out, int rows, int T, int Cp) {
    if (!out || !out->W || out->T != T || out->Cp != Cp || rows < out->N || rows >= out->N + 4) return false;
    for (int t = 0; t < T; ++t) if (out->taps[t] != t) return false;
    return true;
}

// ---- Winograd weight transforms, one work item per call (k_wino_pack / k_w2_pack and the batched form k_wino_pack_batched) ----
// float64 with the contraction order WRITTEN OUT (explicit fma): every kernel that inlines these produces the same bits - left to
// -ffp-contract the two instantiations differed in the last bit of ~0.4 % of the entries, enough to flip near-tie max-pool windows
// downstream (tests/test_gpu_fullsize.py: H's gradient error against the fp32 oracle moved between 1.6e-3 and 1.0e-2)
// 1-D F(4,3), item i in [0, 3*C*N): U[(cc*3 + kh)*6 + a][n][ci] = sum_kw G[a][kw] * Wp[n][kh*3 + kw][cc*16 + ci]
__device__ __forceinline__ void wino_pack_item(const float *__restrict__ Wp, float *__restrict__ U, int N, int C, long long i) {
    const double G[6][3] = {{0.25, 0., 0.}, {-1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6},
                            {1. / 24, 1. / 12, 1. / 6}, {1. / 24, -1. / 12, 1. / 6}, {0., 0., 1.}};
    const int ci = (int)(i % 16); long long r = i / 16;
    const int n = (int)(r % N); const int ch = (int)(r / N);
    const int cc = ch / 3, kh = ch - cc * 3, c = cc * 16 + ci;
    const float *w = Wp + ((long long)n * 9 + kh * 3) * C + c;
    const double w0 = w[0], w1 = w[C], w2 = w[2 * (long long)C];
#pragma unroll
    for (int a = 0; a < 6; ++a)
        U[(((long long)ch * 6 + a) * N + n) * 16 + ci] = (float)fma(G[a][2], w2, fma(G[a][1], w1, G[a][0] * w0));
}

// 2-D F(4x4,3x3), points 0, +-3/4, +-3/2, inf (wino2d.hip), item i in [0, N*C): U[6i + j][n][c] = sum_{kh,kw} G[i][kh] G[j][kw] Wp[n][kh*3 + kw][c]
__device__ __forceinline__ void w2_pack_item(const float *__restrict__ Wp, float *__restrict__ U, int N, int C, long long i) {
    const double a = 0.75, b = 1.5, f0 = a * a * b * b, fa = 2 * a * a * (a * a - b * b), fb = 2 * b * b * (b * b - a * a);
    const double G[6][3] = {{1 / f0, 0., 0.}, {1 / fa, a / fa, a * a / fa}, {1 / fa, -a / fa, a * a / fa},
                            {1 / fb, b / fb, b * b / fb}, {1 / fb, -b / fb, b * b / fb}, {0., 0., 1.}};
    const int c = (int)(i % C); const long long n = i / C;
    double w[3][3];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t / 3][t % 3] = Wp[(n * 9 + t) * C + c];
    double gw[6][3];
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) gw[p][kw] = fma(G[p][2], w[2][kw], fma(G[p][1], w[1][kw], G[p][0] * w[0][kw]));
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int q = 0; q < 6; ++q)
            U[((long long)(6 * p + q) * N + n) * C + c] = (float)fma(gw[p][2], G[q][2], fma(gw[p][1], G[q][1], gw[p][0] * G[q][0]));
}
