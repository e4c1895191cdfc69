// Stand-alone DNC addressing modules (dnc/addressing.py) as individual kernels -- the pieces the reference's
// own module tests call directly (dnc/addressing_test.py, access_test.py).  The fused sequence kernels
// (dnc_seq_fwd.hip / dnc_seq_bwd.hip) are the hot path; these exist for the module-level API
// (CosineWeights, TemporalLinkage, Freeness) and reuse the same arithmetic.  One workgroup per batch element.
#include "dnc_common.h"

#pragma clang fp contract(off)    // op-by-op rounding as in the TF graph (the allocation sorts: ties matter)

namespace {

// CosineWeights._build (addressing.py:83-105): out[b,h,:] = softmax_n( softplus(strength) * cos(key_h, mem_n) )
__global__ __launch_bounds__(256) void cosine_weights_kernel(const float* __restrict__ mem, const float* __restrict__ keys,
                                                              const float* __restrict__ strengths, float* __restrict__ out,
                                                              int N, int W, int H) {
    extern __shared__ float sm[];              // [H*N] scores
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float EPS = 1e-6f;
    const float* M = mem + (size_t)b * N * W;
    for (int idx = tid; idx < H * N; idx += blockDim.x) {
        const int h = idx / N, n = idx - h * N;
        const float* k = keys + ((size_t)b * H + h) * W;
        float dot = 0.f, ks = 0.f, ms = 0.f;
        for (int w = 0; w < W; ++w) { const float kv = k[w], mv = M[(size_t)n * W + w]; dot += kv * mv; ks += kv * kv; ms += mv * mv; }
        const float sim = dot / (sqrtf(ks + EPS) * sqrtf(ms + EPS) + EPS);
        sm[idx] = sim * dnc_softplus(strengths[(size_t)b * H + h]);
    }
    __syncthreads();
    for (int h = wave; h < H; h += blockDim.x >> 6) {
        float* r = sm + h * N;
        float mx = -INFINITY;
        for (int n = lane; n < N; n += 64) mx = fmaxf(mx, r[n]);
        mx = wave_max(mx);
        float s = 0.f;
        for (int n = lane; n < N; n += 64) { const float e = expf(r[n] - mx); r[n] = e; s += e; }
        s = wave_sum(s);
        for (int n = lane; n < N; n += 64) out[((size_t)b * H + h) * N + n] = r[n] / s;
    }
}

// TemporalLinkage._build (addressing.py:133-153, :183-240): link and precedence update for every write head
__global__ void linkage_kernel(const float* __restrict__ prev_link, const float* __restrict__ prev_prec,
                               const float* __restrict__ ww, float* __restrict__ link, float* __restrict__ prec, int N, int Wn) {
    const int bj = blockIdx.x;                  // b*Wn + j
    const float* L0 = prev_link + (size_t)bj * N * N;
    float* L1 = link + (size_t)bj * N * N;
    const float* w = ww + (size_t)bj * N;
    const float* p = prev_prec + (size_t)bj * N;
    __shared__ float red[16];
    float s = 0.f;
    for (int n = threadIdx.x; n < N; n += blockDim.x) s += w[n];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    float tot = 0.f;
    for (int q = 0; q < (int)(blockDim.x >> 6); ++q) tot += red[q];
    for (int idx = threadIdx.x; idx < N * N; idx += blockDim.x) {
        const int a = idx / N, c = idx - a * N;
        const float v = (1.0f - w[a] - w[c]) * L0[idx] + w[a] * p[c];
        L1[idx] = (a == c) ? 0.f : v;
    }
    for (int n = threadIdx.x; n < N; n += blockDim.x) prec[(size_t)bj * N + n] = (1.0f - tot) * p[n] + w[n];
}

// TemporalLinkage.directional_read_weights (addressing.py:155-181): out[b,r,j,:] = rw[b,r,:] @ L[b,j]^T (forward) or @ L[b,j]
__global__ void directional_kernel(const float* __restrict__ link, const float* __restrict__ prw, float* __restrict__ out,
                                   int N, int Wn, int R, int forward) {
    const int brj = blockIdx.x;                 // (b*R + r)*Wn + j
    const int j = brj % Wn, br = brj / Wn, b = br / R;
    const float* L = link + ((size_t)b * Wn + j) * N * N;
    const float* rw = prw + (size_t)br * N;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float s = 0.f;
        if (forward) for (int c = 0; c < N; ++c) s += rw[c] * L[(size_t)n * N + c];
        else for (int a = 0; a < N; ++a) s += rw[a] * L[(size_t)a * N + n];
        out[(size_t)brj * N + n] = s;
    }
}

// Freeness._build (addressing.py:279-305): usage after the previous step's writes and this step's frees
__global__ void freeness_kernel(const float* __restrict__ ww, const float* __restrict__ free_gate, const float* __restrict__ rw,
                                const float* __restrict__ prev_usage, float* __restrict__ usage, int N, int Wn, int R) {
    const int b = blockIdx.x;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float pw = 1.f;
        for (int j = 0; j < Wn; ++j) pw *= (1.0f - ww[((size_t)b * Wn + j) * N + n]);
        float u = prev_usage[(size_t)b * N + n];
        u = u + (1.0f - u) * (1.0f - pw);
        float phi = 1.f;
        for (int i = 0; i < R; ++i) phi *= (1.0f - free_gate[(size_t)b * R + i] * rw[((size_t)b * R + i) * N + n]);
        usage[(size_t)b * N + n] = u * phi;
    }
}

// Freeness.write_allocation_weights (addressing.py:307-340, :376-405), rank form (see dnc_seq_fwd.hip)
__global__ __launch_bounds__(1024) void allocation_kernel(const float* __restrict__ usage, const float* __restrict__ write_gates,
                                                           float* __restrict__ out, int N, int Wn) {
    extern __shared__ float su[];               // [N] simulated usage, [N] allocation
    float* sa = su + N;
    const int b = blockIdx.x;
    const float EPS = 1e-6f;
    for (int n = threadIdx.x; n < N; n += blockDim.x) su[n] = usage[(size_t)b * N + n];
    __syncthreads();
    for (int j = 0; j < Wn; ++j) {
        for (int n = threadIdx.x; n < N; n += blockDim.x) {
            const float nun = 1.0f - (EPS + (1.0f - EPS) * su[n]);
            float prod = 1.f;
            for (int m = 0; m < N; ++m) {
                const float num = 1.0f - (EPS + (1.0f - EPS) * su[m]);
                const bool before = (num > nun) || (num == nun && m < n);
                prod *= before ? (1.0f - num) : 1.0f;
            }
            sa[n] = nun * prod;
        }
        __syncthreads();
        const float g = write_gates ? write_gates[(size_t)b * Wn + j] : 0.f;
        for (int n = threadIdx.x; n < N; n += blockDim.x) {
            out[((size_t)b * Wn + j) * N + n] = sa[n];
            su[n] = su[n] + (1.0f - su[n]) * g * sa[n];
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int ntk_dnc_cosine_weights(const float* memory, const float* keys, const float* strengths, float* out,
                                      int B, int N, int W, int H, void* stream) {
    NTK_REQUIRE(memory && keys && strengths && out, NTK_ERR_BAD_PTR, "ntk_dnc_cosine_weights: null pointer");
    NTK_REQUIRE(B > 0 && N > 0 && W > 0 && H > 0 && (size_t)H * N * 4 <= 64 * 1024, NTK_ERR_BAD_SHAPE,
                "ntk_dnc_cosine_weights: B=%d N=%d W=%d H=%d (H*N*4 bytes must fit 64 KiB)", B, N, W, H);
    cosine_weights_kernel<<<B, 256, (size_t)H * N * sizeof(float), (hipStream_t)stream>>>(memory, keys, strengths, out, N, W, H);
    NTK_CHECK_LAUNCH("ntk_dnc_cosine_weights");
    return NTK_OK;
}

extern "C" int ntk_dnc_linkage(const float* prev_link, const float* prev_prec, const float* write_weights, float* link,
                               float* prec, int B, int N, int Wn, void* stream) {
    NTK_REQUIRE(prev_link && prev_prec && write_weights && link && prec, NTK_ERR_BAD_PTR, "ntk_dnc_linkage: null pointer");
    NTK_REQUIRE(B > 0 && N > 0 && Wn > 0, NTK_ERR_BAD_SHAPE, "ntk_dnc_linkage: B=%d N=%d Wn=%d", B, N, Wn);
    linkage_kernel<<<B * Wn, 256, 0, (hipStream_t)stream>>>(prev_link, prev_prec, write_weights, link, prec, N, Wn);
    NTK_CHECK_LAUNCH("ntk_dnc_linkage");
    return NTK_OK;
}

extern "C" int ntk_dnc_directional_read_weights(const float* link, const float* prev_read_weights, float* out, int B, int N,
                                                int Wn, int R, int forward, void* stream) {
    NTK_REQUIRE(link && prev_read_weights && out, NTK_ERR_BAD_PTR, "ntk_dnc_directional_read_weights: null pointer");
    NTK_REQUIRE(B > 0 && N > 0 && Wn > 0 && R > 0, NTK_ERR_BAD_SHAPE, "ntk_dnc_directional_read_weights: B=%d N=%d Wn=%d R=%d", B, N, Wn, R);
    directional_kernel<<<B * R * Wn, 256, 0, (hipStream_t)stream>>>(link, prev_read_weights, out, N, Wn, R, forward);
    NTK_CHECK_LAUNCH("ntk_dnc_directional_read_weights");
    return NTK_OK;
}

extern "C" int ntk_dnc_freeness(const float* write_weights, const float* free_gate, const float* read_weights,
                                const float* prev_usage, float* usage, int B, int N, int Wn, int R, void* stream) {
    NTK_REQUIRE(write_weights && free_gate && read_weights && prev_usage && usage, NTK_ERR_BAD_PTR, "ntk_dnc_freeness: null pointer");
    NTK_REQUIRE(B > 0 && N > 0 && Wn > 0 && R > 0, NTK_ERR_BAD_SHAPE, "ntk_dnc_freeness: B=%d N=%d Wn=%d R=%d", B, N, Wn, R);
    freeness_kernel<<<B, 256, 0, (hipStream_t)stream>>>(write_weights, free_gate, read_weights, prev_usage, usage, N, Wn, R);
    NTK_CHECK_LAUNCH("ntk_dnc_freeness");
    return NTK_OK;
}

extern "C" int ntk_dnc_write_allocation_weights(const float* usage, const float* write_gates, float* out, int B, int N, int Wn,
                                                void* stream) {
    NTK_REQUIRE(usage && out, NTK_ERR_BAD_PTR, "ntk_dnc_write_allocation_weights: null pointer");
    NTK_REQUIRE(B > 0 && N > 0 && Wn > 0 && (size_t)N * 8 <= 64 * 1024, NTK_ERR_BAD_SHAPE,
                "ntk_dnc_write_allocation_weights: B=%d N=%d Wn=%d", B, N, Wn);
    NTK_REQUIRE(Wn == 1 || write_gates, NTK_ERR_BAD_PTR, "ntk_dnc_write_allocation_weights: write_gates needed for several heads");
    allocation_kernel<<<B, 1024, (size_t)2 * N * sizeof(float), (hipStream_t)stream>>>(usage, write_gates, out, N, Wn);
    NTK_CHECK_LAUNCH("ntk_dnc_write_allocation_weights");
    return NTK_OK;
}
