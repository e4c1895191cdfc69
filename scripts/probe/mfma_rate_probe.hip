// Probe: sustained rate of v_mfma_f32_32x32x2_f32 vs v_mfma_f32_16x16x4_f32 (same nominal 256 FLOP/clk/CU) with NACC
// independent accumulators per wave, 1 and 2 waves per SIMD, one workgroup per CU.
//   build: hipcc -O3 --offload-arch=gfx950 -o mfma_rate_probe mfma_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512) void k32(float* out, int iters, float seed) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = seed + threadIdx.x, b = seed * 2.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[m % NACC]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(512) void k16(float* out, int iters, float seed) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = seed + threadIdx.x, b = seed * 2.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m % NACC]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static void timeit(const char* name, int nacc, int threads, F launch) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(200);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    launch(iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * (threads / 64) * iters * 8.0 * 4096.0;     // 8 x 32x32x2 (or 16 x 16x16x4) per wave-iteration
    printf("%-8s nacc=%d waves/SIMD=%d  %.4f us/iter  %.1f TFLOP/s\n", name, nacc, threads / 256, ms * 1e3 / iters, flops / (ms * 1e-3) / 1e12);
}

int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    for (int th : {256, 512}) {
        timeit("32x32x2", 2, th, [&](int it) { k32<2><<<256, th>>>(out, it, 1e-9f); });
        timeit("32x32x2", 4, th, [&](int it) { k32<4><<<256, th>>>(out, it, 1e-9f); });
        timeit("32x32x2", 8, th, [&](int it) { k32<8><<<256, th>>>(out, it, 1e-9f); });
        timeit("16x16x4", 2, th, [&](int it) { k16<2><<<256, th>>>(out, it, 1e-9f); });
        timeit("16x16x4", 4, th, [&](int it) { k16<4><<<256, th>>>(out, it, 1e-9f); });
        timeit("16x16x4", 8, th, [&](int it) { k16<8><<<256, th>>>(out, it, 1e-9f); });
        timeit("16x16x4", 16, th, [&](int it) { k16<16><<<256, th>>>(out, it, 1e-9f); });
    }
    // long runs: does the rate hold when the chip is warm (power management)?
    for (int rep = 0; rep < 3; ++rep) {
        timeit("32x32x2", 8, 512, [&](int it) { k32<8><<<256, 512>>>(out, it * 10, 1e-9f); });
        timeit("16x16x4", 16, 512, [&](int it) { k16<16><<<256, 512>>>(out, it * 10, 1e-9f); });
    }
    (void)hipFree(out);
    return 0;
}
