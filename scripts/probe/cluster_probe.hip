// Dev probe (not part of the library): (1) operand / result lane map of v_mfma_f32_4x4x1_16b_f32 with exact integer
// data; (2) the in-launch hand-off used by the DNC cluster kernels (sc1 payload stores, drained, one sc1 flag
// store; sc1 flag poll by one wave, workgroup barrier, sc1 payload loads) under a 20000-step ping-pong of k
// workgroups per cluster, every word checked, with a bounded spin.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void mfma_probe(const float* A, const float* B, float* D) {   // A[16][4], B[16][4] -> D[16][4][4]
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l], B[l], acc, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[l * 4 + v] = acc[v];
}

#define RLX __ATOMIC_RELAXED
#define AGENT __HIP_MEMORY_SCOPE_AGENT
// MODE 0: cluster = consecutive block ids (spread over the XCDs by the round-robin dispatch), sc1 (write-through) stores
// MODE 1: cluster = blocks b with equal b % 8 (observed: one XCD), sc1 stores
// MODE 2: as 1, PLAIN payload and flag stores (stay in that XCD's L2; only valid when the cluster shares an XCD: xcc[] says)
template <int MODE>
__global__ __launch_bounds__(512) void xchg_probe(unsigned* flags, unsigned long long* mbox, unsigned* err, int k, int steps, int words,
                                                  unsigned* bad, unsigned* xcc) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int cl, g;
    if (MODE == 0) { cl = blockIdx.x / k; g = blockIdx.x % k; }
    else { const int j = blockIdx.x >> 3; cl = (j / k) * 8 + (blockIdx.x & 7); g = j % k; }
    if (tid == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[cl * k + g] = id & 15u;
    }
    __shared__ int s_abort;
    if (tid == 0) s_abort = 0;
    unsigned* fl = flags + (size_t)cl * k;
    unsigned nbad = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < steps; ++t) {
        const unsigned epoch = t + 1;
        unsigned long long* slot = mbox + (((size_t)cl * 2 + (t & 1)) * k + g) * words;
        for (int i = tid; i < words; i += 512) {
            const unsigned long long x = ((unsigned long long)epoch << 32) | (unsigned)(g * 100000 + i);
            if (MODE == 2) slot[i] = x;
            else __hip_atomic_store(slot + i, x, RLX, AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (MODE == 2) { fl[g] = epoch; asm volatile("" ::: "memory"); }
            else __hip_atomic_store(fl + g, epoch, RLX, AGENT);
        }
        if (wave == 0) {
            unsigned spins = 0;
            for (;;) {
                unsigned v = (lane < k) ? __hip_atomic_load(fl + lane, RLX, AGENT) : epoch;
                if (__all((int)(v - epoch) >= 0)) break;
                if ((++spins & 255u) == 0) {
                    if (__hip_atomic_load(err, RLX, AGENT) != 0 || __builtin_amdgcn_s_memrealtime() - t0 > 300000000ull) {
                        if (lane == 0) { __hip_atomic_store(err, 1u, RLX, AGENT); s_abort = 1; }
                        break;
                    }
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __syncthreads();
        if (s_abort) return;
        for (int gg = 0; gg < k; ++gg) {
            const unsigned long long* src = mbox + (((size_t)cl * 2 + (t & 1)) * k + gg) * words;
            for (int i = tid; i < words; i += 512) {
                const unsigned long long x = __hip_atomic_load(src + i, RLX, AGENT);
                if ((unsigned)(x >> 32) != epoch || (unsigned)x != (unsigned)(gg * 100000 + i)) ++nbad;
            }
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main(int argc, char** argv) {
    {
        std::vector<float> A(64), B(64), D(256);
        for (int l = 0; l < 64; ++l) { A[l] = (float)(1 + l); B[l] = (float)(1000 + 7 * l); }
        float *dA, *dB, *dD;
        hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 1024);
        hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice);
        mfma_probe<<<1, 64>>>(dA, dB, dD);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        // hypothesis: lane l = 4b + x; A holds A[b][i = x], B holds B[b][j = x]; D reg v of lane 4b + j = A[b][v] * B[b][j]
        int ok = 1;
        for (int l = 0; l < 64; ++l)
            for (int v = 0; v < 4; ++v) {
                const int b = l >> 2, j = l & 3;
                const float want = A[4 * b + v] * B[4 * b + j];
                if (D[l * 4 + v] != want) { ok = 0; if (l < 8) printf("lane %d reg %d: got %g want %g\n", l, v, D[l * 4 + v], want); }
            }
        printf("mfma_4x4x1 layout hypothesis (D[v] of lane 4b+j = A[lane 4b+v] * B[lane 4b+j]): %s\n", ok ? "CONFIRMED" : "WRONG");
        if (!ok) for (int l = 0; l < 8; ++l) printf("  lane %d: %g %g %g %g\n", l, D[l * 4], D[l * 4 + 1], D[l * 4 + 2], D[l * 4 + 3]);
    }
    const int k = argc > 1 ? atoi(argv[1]) : 8, clusters = argc > 2 ? atoi(argv[2]) : 32, steps = argc > 3 ? atoi(argv[3]) : 20000;
    const int words = argc > 4 ? atoi(argv[4]) : 1024;
    unsigned *flags, *err, *bad, *xcc; unsigned long long* mbox;
    hipMalloc(&flags, clusters * k * 4 + 64); hipMalloc(&err, 64); hipMalloc(&bad, 64); hipMalloc(&xcc, clusters * k * 4);
    hipMalloc(&mbox, (size_t)clusters * 2 * k * words * 8);
    for (int rep = 0; rep < 6; ++rep) {
        const int mode = rep >> 1;
        if (mode && (clusters % 8)) continue;
        hipMemset(flags, 0, clusters * k * 4 + 64); hipMemset(err, 0, 64); hipMemset(bad, 0, 64);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (mode == 0) xchg_probe<0><<<clusters * k, 512>>>(flags, mbox, err, k, steps, words, bad, xcc);
        if (mode == 1) xchg_probe<1><<<clusters * k, 512>>>(flags, mbox, err, k, steps, words, bad, xcc);
        if (mode == 2) xchg_probe<2><<<clusters * k, 512>>>(flags, mbox, err, k, steps, words, bad, xcc);
        hipEventRecord(e1);
        hipError_t e = hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        unsigned herr = 0, hbad = 0;
        hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost); hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
        std::vector<unsigned> hx(clusters * k);
        hipMemcpy(hx.data(), xcc, clusters * k * 4, hipMemcpyDeviceToHost);
        int split = 0;
        for (int c = 0; c < clusters; ++c) for (int g = 1; g < k; ++g) if (hx[c * k + g] != hx[c * k]) { ++split; break; }
        printf("mode %d (%s): clusters spanning more than one XCD: %d of %d\n", mode,
               mode == 0 ? "consecutive blocks, sc1 stores" : mode == 1 ? "b%%8 clusters, sc1 stores" : "b%%8 clusters, plain stores", split, clusters);
        printf("exchange probe k=%d clusters=%d steps=%d payload %d B/WG: %s, timeout=%u, bad words=%u, %.3f us/step (all-gather of %d KB per WG)\n",
               k, clusters, steps, words * 8, hipGetErrorString(e), herr, hbad, ms * 1e3 / steps, k * words * 8 / 1024);
    }
    return 0;
}
