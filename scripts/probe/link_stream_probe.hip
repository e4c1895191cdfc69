// Dev probe (not part of the library): what does a per-step pass over a sequence's temporal link cost when the link
// lives in HBM and every CU streams its own row slice?  Shapes of BASELINE configs[4] (DNC 512 x 128, batch 64):
// 256 workgroups (64 sequences x 4 row slices), each owning 128 rows x 512 columns = 256 KB of every sequence-step's
// 1 MB link.  Per step a workgroup reads its slice of L_{t-1}, applies the link update and writes its slice of L_t:
//   MODE 0  ping-pong between two 64 MB buffers (inference: the state is overwritten)
//   MODE 1  record stream: L_{t-1} = record t-1, L_t = record t (training: every step is kept for BPTT)
//   MODE 2  BPTT shape: reads records L_t and L_{t-1}, reads and rewrites the carried d(link) slice in place
// PF = link rows a wave keeps in flight (2 float4 per lane and row).  NT = non-temporal loads / stores.
// No cross-workgroup hand-offs here: this is the streaming floor the cluster kernels' link phase sits on.
//   hipcc --offload-arch=gfx950 -O3 -o link_stream_probe link_stream_probe.hip && ./link_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool NT>
__device__ __forceinline__ f32x4 ld(const f32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT>
__device__ __forceinline__ void st(f32x4* p, f32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <int MODE, int PF, bool NT, bool NTS = NT>
__global__ __launch_bounds__(512) void link_stream(float* rec, float* gl, int steps, int N, int NR, int k, float* sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / k, g = blockIdx.x % k, B = gridDim.x / k;
    const size_t NN = (size_t)N * N;
    const int N4 = N >> 2;
    f32x4 colacc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    float rowacc = 0.f;
    for (int t = 1; t < steps; ++t) {
        const size_t src_t = (MODE == 0) ? (size_t)((t - 1) & 1) : (size_t)(t - 1);
        const size_t dst_t = (MODE == 0) ? (size_t)(t & 1) : (size_t)t;
        const f32x4* src = reinterpret_cast<const f32x4*>(rec + (src_t * B + b) * NN + (size_t)g * NR * N);
        f32x4* dst = reinterpret_cast<f32x4*>(rec + (dst_t * B + b) * NN + (size_t)g * NR * N);
        f32x4* gp = reinterpret_cast<f32x4*>(gl + (size_t)b * NN + (size_t)g * NR * N);
        const float wa = 1e-3f * (float)(t & 7);
        for (int r0 = wave * PF; r0 < NR; r0 += 8 * PF) {
            f32x4 v[PF][2], w[PF][2], gq[PF][2];
#pragma unroll
            for (int u = 0; u < PF; ++u)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    v[u][h] = ld<NT>(src + (size_t)(r0 + u) * N4 + h * 64 + lane);
                    if (MODE == 2) {
                        w[u][h] = ld<NT>(reinterpret_cast<const f32x4*>(dst) + (size_t)(r0 + u) * N4 + h * 64 + lane);
                        gq[u][h] = gp[(size_t)(r0 + u) * N4 + h * 64 + lane];
                    }
                }
#pragma unroll
            for (int u = 0; u < PF; ++u)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (MODE == 2) {
                        f32x4 q = gq[u][h] + wa * w[u][h];
                        colacc[h] -= q * v[u][h];
                        rowacc += q[0] * w[u][h][0] + q[1] * w[u][h][1] + q[2] * w[u][h][2] + q[3] * w[u][h][3];
                        gp[(size_t)(r0 + u) * N4 + h * 64 + lane] = (1.0f - wa) * q;
                    } else {
                        f32x4 x = (1.0f - wa - 1e-3f) * v[u][h] + wa;
                        colacc[h] += wa * x;
                        rowacc += x[0] + x[1] + x[2] + x[3];
                        st<NTS>(dst + (size_t)(r0 + u) * N4 + h * 64 + lane, x);
                    }
                }
        }
        __syncthreads();     // one barrier per step, as a real step has several
    }
    if (rowacc + colacc[0][0] + colacc[1][3] == 12345.678f) sink[0] = rowacc;
}

template <int MODE, int PF, bool NT, bool NTS = NT>
static void run(float* rec, float* gl, float* sink, int B, int k, int steps, const char* name) {
    const int N = 512, NR = N / k;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    link_stream<MODE, PF, NT, NTS><<<B * k, 512>>>(rec, gl, 8, N, NR, k, sink);      // warm
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    link_stream<MODE, PF, NT, NTS><<<B * k, 512>>>(rec, gl, steps, N, NR, k, sink);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double per_step_bytes = (double)B * N * N * 4 * (MODE == 2 ? 4.0 : 2.0);
    const double us = ms * 1e3 / (steps - 1);
    printf("%-34s B=%d k=%d PF=%d NTload=%d NTstore=%d: %8.2f us/step  %7.1f GB/s (%.0f MB per step)\n", name, B, k, PF, (int)NT, (int)NTS, us,
           per_step_bytes / (us * 1e-6) / 1e9, per_step_bytes / 1e6);
    fflush(stdout);
}

int main() {
    const int B = 64, steps = 200;
    const size_t NN = 512 * 512;
    float *rec, *gl, *sink;
    CK(hipMalloc(&rec, (size_t)steps * B * NN * 4));        // 200 x 64 MB = 12.8 GB
    CK(hipMalloc(&gl, (size_t)B * NN * 4));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(rec, 0, (size_t)steps * B * NN * 4));
    CK(hipMemset(gl, 0, (size_t)B * NN * 4));
    run<0, 2, false>(rec, gl, sink, B, 4, steps, "ping-pong");
    run<0, 4, false>(rec, gl, sink, B, 4, steps, "ping-pong");
    run<0, 4, true>(rec, gl, sink, B, 4, steps, "ping-pong");
    run<0, 8, false>(rec, gl, sink, B, 4, steps, "ping-pong");
    run<1, 2, false>(rec, gl, sink, B, 4, steps, "record stream");
    run<1, 4, false>(rec, gl, sink, B, 4, steps, "record stream");
    run<1, 4, true>(rec, gl, sink, B, 4, steps, "record stream");
    run<1, 8, false>(rec, gl, sink, B, 4, steps, "record stream");
    run<1, 8, true>(rec, gl, sink, B, 4, steps, "record stream");
    run<1, 4, true, false>(rec, gl, sink, B, 4, steps, "record stream");
    run<1, 4, false, true>(rec, gl, sink, B, 4, steps, "record stream");
    run<0, 4, true, false>(rec, gl, sink, B, 4, steps, "ping-pong");
    run<0, 4, false, true>(rec, gl, sink, B, 4, steps, "ping-pong");
    run<2, 2, false>(rec, gl, sink, B, 4, steps, "bptt (L_t, L_t-1, dL r/w)");
    run<2, 4, false>(rec, gl, sink, B, 4, steps, "bptt (L_t, L_t-1, dL r/w)");
    run<2, 4, true>(rec, gl, sink, B, 4, steps, "bptt (L_t, L_t-1, dL r/w)");
    // 32 sequences x 8 slices (two rounds per batch of 64) and 32 x 4 (half the CUs left to the trunk)
    run<1, 4, false>(rec, gl, sink, 32, 8, steps, "record stream");
    run<1, 4, false>(rec, gl, sink, 32, 4, steps, "record stream");
    run<2, 4, false>(rec, gl, sink, 32, 8, steps, "bptt (L_t, L_t-1, dL r/w)");
    run<2, 4, false>(rec, gl, sink, 32, 4, steps, "bptt (L_t, L_t-1, dL r/w)");
    return 0;
}
