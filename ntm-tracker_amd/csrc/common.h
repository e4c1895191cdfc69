// Shared host/device helpers for libntmtrack_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/ntmtrack.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void ntk_set_error(const char* fmt, ...);
int ntk_device_cu_count();      // api.cpp: compute units of the current device (cached per device)

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

// Kernels that use more than 64 KiB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised once PER
// DEVICE.  The only global state of the library is this read-only-after-first-use per-device cache.
struct NtkLdsAttrCache {
    unsigned long long done = 0;     // bit d: attribute set on device d (benign race: setting it twice is harmless)
};
static inline int ntk_raise_lds_limit(NtkLdsAttrCache& cache, const void* const* kernels, int n, const char* who) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess && dev >= 0 && dev < 64 && ((__atomic_load_n(&cache.done, __ATOMIC_RELAXED) >> dev) & 1ull)) return NTK_OK;
    for (int i = 0; i < n && e == hipSuccess; ++i)
        e = hipFuncSetAttribute(kernels[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
        ntk_set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e));
        return NTK_ERR_HIP;
    }
    if (dev >= 0 && dev < 64) __atomic_fetch_or(&cache.done, 1ull << dev, __ATOMIC_RELAXED);
    return NTK_OK;
}

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

// ntk_stream_matvec (below) with the prefetch cut at the slice's end: rows at or past r1 are not loaded (the loads sit under lane
// predicates; same products in the same order).  The plain form fetches 2 PF rows past a slice -- 15 rows for nothing per 25-row
// slice of the NTM BPTT's read columns.  Measured (B32 x S1300, alone): BPTT 20.1 -> 19.4 ms with this form in its three
// products; the forward's unpack product is 0.9 ms SLOWER with it (its extra rows are L1 hits, the predicates are not free), so
// only the BPTT kernel uses it.
template <int PF>
__device__ __forceinline__ f32x4 ntk_stream_matvec_exact(const f32x4* __restrict__ w4, int stride4, const float* __restrict__ x, int r0, int r1,
                                                         f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}) {        // acc: the sum so far (rows before r0)
    f32x4 wa[PF], wb[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) { wa[q] = f32x4{0.f, 0.f, 0.f, 0.f}; if (r0 + q < r1) wa[q] = w4[(size_t)(r0 + q) * stride4]; }
#pragma unroll
    for (int q = 0; q < PF; ++q) { wb[q] = f32x4{0.f, 0.f, 0.f, 0.f}; if (r0 + PF + q < r1) wb[q] = w4[(size_t)(r0 + PF + q) * stride4]; }
    for (int r = r0; r < r1; r += 2 * PF) {
#pragma unroll
        for (int q = 0; q < PF; ++q) acc += ((r + q < r1) ? x[r + q] : 0.f) * wa[q];
#pragma unroll
        for (int q = 0; q < PF; ++q) if (r + 2 * PF + q < r1) wa[q] = w4[(size_t)(r + 2 * PF + q) * stride4];
#pragma unroll
        for (int q = 0; q < PF; ++q) acc += ((r + PF + q < r1) ? x[r + PF + q] : 0.f) * wb[q];
#pragma unroll
        for (int q = 0; q < PF; ++q) if (r + 3 * PF + q < r1) wb[q] = w4[(size_t)(r + 3 * PF + q) * stride4];
    }
    return acc;
}

// Streaming mat-vec slice: sum_{r = r0}^{r1 - 1} x[r] * w4[r * stride4] with two register batches of PF rows in flight.
// Written out explicitly because the compiler, left alone, sinks every load to just before its use (one 16-byte
// load in flight per thread): fine while the weights hit L2, 2.5-2.8x slower when a concurrent kernel evicts them.
// rlast = last row index that may be read (rows past r1 are only prefetched, never used).
template <int PF>
__device__ __forceinline__ f32x4 ntk_stream_matvec(const f32x4* __restrict__ w4, int stride4, const float* __restrict__ x,
                                                   int r0, int r1, int rlast) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 wa[PF], wb[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) wa[q] = w4[(size_t)min(r0 + q, rlast) * stride4];
#pragma unroll
    for (int q = 0; q < PF; ++q) wb[q] = w4[(size_t)min(r0 + PF + q, rlast) * stride4];
    for (int r = r0; r < r1; r += 2 * PF) {
#pragma unroll
        for (int q = 0; q < PF; ++q) acc += ((r + q < r1) ? x[r + q] : 0.f) * wa[q];
#pragma unroll
        for (int q = 0; q < PF; ++q) wa[q] = w4[(size_t)min(r + 2 * PF + q, rlast) * stride4];
#pragma unroll
        for (int q = 0; q < PF; ++q) acc += ((r + PF + q < r1) ? x[r + PF + q] : 0.f) * wb[q];
#pragma unroll
        for (int q = 0; q < PF; ++q) wb[q] = w4[(size_t)min(r + 3 * PF + q, rlast) * stride4];
    }
    return acc;
}
