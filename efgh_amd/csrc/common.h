// Shared host-side helpers for libefgh_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

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
