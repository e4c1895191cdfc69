// Probe: how much non-MFMA work fits beside v_mfma_f32_32x32x2_f32 on a gfx950 SIMD before the MFMA rate drops?
// Each wave runs ITER iterations of [NM independent MFMAs interleaved with NV "other" instructions of one kind].
// Reported: SIMD cycles per iteration (s_memtime based wall clock of the whole grid / iterations) for 1 and 2 waves
// per SIMD.  NM MFMAs alone cost NM*64 cycles (16 passes x 4 cycles); anything above that is work that did not overlap.
//   build: hipcc -O3 --offload-arch=gfx950 -o mfma_coissue_probe mfma_coissue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum Kind { K_NONE = 0, K_FMA = 1, K_PKFMA = 2, K_IADD = 3, K_DSREAD = 4, K_MOV = 5, K_CVT = 6 };

// the same experiment on v_mfma_f32_16x16x4_f32 (8 passes, 4 accumulator registers: half the result-write traffic per
// cycle of 32x32x2): 16 MFMAs = the same 512 cycles per iteration
template <int KIND, int NV>
__global__ __launch_bounds__(512) void probe16(float* out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = seed * i;
    __syncthreads();
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = seed + threadIdx.x, b = seed * 2.f;
    float v[8];
    f32x2 p[4];
    int iv[8];
    f32x4 dl[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = seed * (i + 1); iv[i] = threadIdx.x + i; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { p[i] = f32x2{seed, seed * i}; dl[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const float* lp = lds + (threadIdx.x & 63) * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a), "v"(b));
#pragma unroll
            for (int q = 0; q < (NV + 8 * (m & 1)) / 16; ++q) {
                const int r = (m * (NV / 16) + q) & 7;
                if constexpr (KIND == K_FMA) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[r]) : "v"(a), "v"(b));
                if constexpr (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p[r & 3]) : "v"(p[(r + 1) & 3]));
                if constexpr (KIND == K_IADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(iv[r]) : "v"(iv[(r + 1) & 7]));
                if constexpr (KIND == K_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(v[r]) : "v"(a));
                if constexpr (KIND == K_DSREAD) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dl[r & 3]) : "v"((unsigned)(size_t)lp), "n"(0));
            }
        }
        if constexpr (KIND == K_DSREAD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i] + (float)iv[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += p[i][0] + p[i][1] + dl[i][0] + dl[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int NV>
__global__ __launch_bounds__(512) void probe(float* out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = seed * i;
    __syncthreads();
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    float a = seed + threadIdx.x, b = seed * 2.f;
    float v[8];
    f32x2 p[4];
    int iv[8];
    f32x4 dl[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = seed * (i + 1); iv[i] = threadIdx.x + i; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { p[i] = f32x2{seed, seed * i}; dl[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const float* lp = lds + (threadIdx.x & 63) * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            // one MFMA, then NV/8 other instructions
            // inline asm so the compiler keeps the interleaving (three other MFMAs = 192 cycles separate two uses of one
            // accumulator: no software hazard)
            if ((m & 3) == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
            if ((m & 3) == 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
            if ((m & 3) == 2) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc2) : "v"(a), "v"(b));
            if ((m & 3) == 3) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc3) : "v"(a), "v"(b));
#pragma unroll
            for (int q = 0; q < NV / 8; ++q) {
                const int r = (m * (NV / 8) + q) & 7;
                if constexpr (KIND == K_FMA) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[r]) : "v"(a), "v"(b));
                if constexpr (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p[r & 3]) : "v"(p[(r + 1) & 3]));
                if constexpr (KIND == K_IADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(iv[r]) : "v"(iv[(r + 1) & 7]));
                if constexpr (KIND == K_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(v[r]) : "v"(a));
                if constexpr (KIND == K_CVT) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[r]) : "v"(a), "v"(b));
                if constexpr (KIND == K_DSREAD) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dl[r & 3]) : "v"((unsigned)(size_t)lp), "n"(0));
            }
        }
        if constexpr (KIND == K_DSREAD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i] + (float)iv[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += p[i][0] + p[i][1] + dl[i][0] + dl[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int NV, bool M16 = false>
static void run(const char* name, float* out, int threads) {
    const int iters = 4000, grid = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    if (M16) probe16<KIND, NV><<<grid, threads>>>(out, 100, 1e-9f); else probe<KIND, NV><<<grid, threads>>>(out, 100, 1e-9f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    if (M16) probe16<KIND, NV><<<grid, threads>>>(out, iters, 1e-9f); else probe<KIND, NV><<<grid, threads>>>(out, iters, 1e-9f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    // one workgroup per CU (grid 256): waves per SIMD = threads / 256
    const double us_per_iter = ms * 1e3 / iters;
    printf("%-8s %s NV=%3d threads=%3d waves/SIMD=%d  %.4f us/iter  (MFMAs alone = 512 cyc/wave)\n", name,
           M16 ? "16x16x4" : "32x32x2", NV, threads, threads / 256, us_per_iter);
}

#define RUNK(KIND, name)                                                                                   \
    for (int th : {256, 512}) {                                                                            \
        run<KIND, 8>(name, out, th); run<KIND, 16>(name, out, th); run<KIND, 32>(name, out, th);            \
        run<KIND, 64>(name, out, th); run<KIND, 128>(name, out, th);                                        \
    }

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("clock %d kHz\n", clk);
    for (int th : {256, 512}) run<K_NONE, 8>("none", out, th);
    RUNK(K_FMA, "fma")
    RUNK(K_PKFMA, "pkfma")
    RUNK(K_IADD, "iadd")
    RUNK(K_MOV, "mov")
    RUNK(K_DSREAD, "dsread")
    for (int th : {256, 512}) run<K_NONE, 16, true>("none", out, th);
#define RUNK16(KIND, name)                                                                                       \
    for (int th : {256, 512}) {                                                                                  \
        run<KIND, 16, true>(name, out, th); run<KIND, 32, true>(name, out, th); run<KIND, 64, true>(name, out, th); \
        run<KIND, 128, true>(name, out, th); run<KIND, 256, true>(name, out, th);                                  \
    }
    RUNK16(K_FMA, "fma")
    RUNK16(K_PKFMA, "pkfma")
    RUNK16(K_IADD, "iadd")
    RUNK16(K_DSREAD, "dsread")
    hipFree(out);
    return 0;
}
