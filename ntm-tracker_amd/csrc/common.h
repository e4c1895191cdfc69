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

// wave64 reductions on the DPP data path (no LDS crossbar traffic): butterfly inside each row of 16 lanes with
// quad_perm / row_half_mirror / row_mirror, then the four row totals are read with v_readlane.  The result is
// wave-uniform.  (__shfl_xor lowers to ds_bpermute_b32 on gfx950: ~5x the latency per step.)
template <int CTRL>
__device__ __forceinline__ float ntk_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += ntk_dpp<0xB1>(v);     // quad_perm [1,0,3,2]
    v += ntk_dpp<0x4E>(v);     // quad_perm [2,3,0,1]
    v += ntk_dpp<0x141>(v);    // row_half_mirror
    v += ntk_dpp<0x140>(v);    // row_mirror
    const int b = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_readlane(b, 0)) + __int_as_float(__builtin_amdgcn_readlane(b, 16)) +
           __int_as_float(__builtin_amdgcn_readlane(b, 32)) + __int_as_float(__builtin_amdgcn_readlane(b, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, ntk_dpp<0xB1>(v));
    v = fmaxf(v, ntk_dpp<0x4E>(v));
    v = fmaxf(v, ntk_dpp<0x141>(v));
    v = fmaxf(v, ntk_dpp<0x140>(v));
    const int b = __float_as_int(v);
    return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(b, 0)), __int_as_float(__builtin_amdgcn_readlane(b, 16))),
                 fmaxf(__int_as_float(__builtin_amdgcn_readlane(b, 32)), __int_as_float(__builtin_amdgcn_readlane(b, 48))));
}
