// Shared host/device helpers for libntmtrack_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/ntmtrack.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void ntk_set_error(const char* fmt, ...);

#define NTK_REQUIRE(cond, code, ...)                 \
    do {                                             \
        if (!(cond)) {                               \
            ntk_set_error(__VA_ARGS__);              \
            return (code);                           \
        }                                            \
    } while (0)

// launch check: hipGetLastError only (never synchronises the stream)
#define NTK_CHECK_LAUNCH(name)                                                 \
    do {                                                                       \
        hipError_t e_ = hipGetLastError();                                     \
        if (e_ != hipSuccess) {                                                \
            ntk_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return NTK_ERR_HIP;                                                \
        }                                                                      \
    } while (0)

static inline bool ntk_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
