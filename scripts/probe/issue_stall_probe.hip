// Probe: what does it cost a wave to ISSUE a burst of small global loads whose addresses lie in many different arrays
// (the per-step BPTT records of the DNC cluster kernel: ~24 loads per thread from ~18 arrays of several GB in all),
// compared with the same bytes taken from ONE packed row per step?  One 512-thread workgroup per CU walks `steps`
// steps backwards (as BPTT does); per step every wave issues NL dword loads (lane-contiguous, 256 B per wave-load) and
// the cycles between the first issue and the last ISSUE (not the data) are accumulated with s_memtime; the data are
// consumed one step later.
//   mode 0: load j reads array j (NL arrays, each rows * rowbytes, `gap` bytes apart), row (wg, step)
//   mode 1: all NL loads read consecutive 2-KB pieces of ONE packed row per (wg, step)
//   build: hipcc -O3 --offload-arch=gfx950 -o issue_stall_probe issue_stall_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int NL = 24;

template <int MODE>
__global__ __launch_bounds__(512) void probe(const float* base, size_t array_floats, int steps, unsigned long long* cyc, float* sink) {
    const int wg = blockIdx.x, tid = threadIdx.x;
    float v[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) v[j] = 0.f;
    float acc = 0.f;
    unsigned long long issue = 0, total0 = __builtin_amdgcn_s_memtime();
    for (int t = steps - 1; t >= 0; --t) {
#pragma unroll
        for (int j = 0; j < NL; ++j) acc += v[j];          // consume the previous step's loads (waits for them)
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        const size_t row = (size_t)wg * steps + t;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const float* p = MODE == 0 ? base + (size_t)j * array_floats + row * 512 + tid
                                       : base + row * (512 * NL) + j * 512 + tid;
            v[j] = __builtin_nontemporal_load(p);
        }
        asm volatile("" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        issue += t1 - t0;
        // ~4 us of other work per step (LDS-free ALU), as the kernel has between the request and the use
        float x = acc;
        for (int i = 0; i < 1500; ++i) x = x * 1.0000001f + 0.5f;
        acc = x;
    }
    const unsigned long long total = __builtin_amdgcn_s_memtime() - total0;
    if (tid == 0) { cyc[wg * 2] = issue; cyc[wg * 2 + 1] = total; }
    sink[wg * 512 + tid] = acc;
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 1300, wgs = 256;
    const size_t rows = (size_t)wgs * steps;
    const size_t array_floats = rows * 512;                 // 2 KB per (wg, step) per array: 680 MB per array at 1300 steps
    float* base; unsigned long long* cyc; float* sink;
    if (hipMalloc(&base, array_floats * NL * sizeof(float)) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(base, 0, array_floats * NL * sizeof(float));
    hipMalloc(&cyc, wgs * 2 * 8); hipMalloc(&sink, wgs * 512 * 4);
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode == 0) probe<0><<<wgs, 512>>>(base, array_floats, steps, cyc, sink);
            else probe<1><<<wgs, 512>>>(base, array_floats, steps, cyc, sink);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[512];
            hipMemcpy(h, cyc, wgs * 2 * 8, hipMemcpyDeviceToHost);
            double si = 0, st = 0;
            for (int i = 0; i < wgs; ++i) { si += h[2 * i]; st += h[2 * i + 1]; }
            printf("mode %d (%s): %.2f us/step wall; wave 0 of a workgroup: issue of %d loads %.0f shader cycles per step, whole step %.0f cycles\n",
                   mode, mode == 0 ? "24 arrays" : "one packed row", ms * 1e3 / steps, NL, si / wgs / steps, st / wgs / steps);
        }
    return 0;
}
