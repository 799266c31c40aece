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
// select and one multiply - no branch.  A NaN stays a NaN under every activation (max / min would return their other operand
// and turn a diverged pre-activation into a finite 0); ReLU(-inf) comes out as NaN as well, which is what a diverged network
// should look like.
__device__ __forceinline__ float act_neg(float v, float neg) { return v * (v > 0.f ? 1.f : neg); }
