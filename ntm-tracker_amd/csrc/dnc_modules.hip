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


// MemoryAccess._read_inputs activations (access.py:160-218) on the raw outputs of the ten linears laid out in the
// packed interface order (DncDims); fields are written de-interleaved, each as a contiguous [B, width] array at
// act + B * offset(field), so the module kernels take plain pointers.
__global__ void interface_act_kernel(const float* __restrict__ raw, int ldr, float* __restrict__ act, DncDims d) {
    const int b = blockIdx.x, B = gridDim.x;
    const float* x = raw + (size_t)b * ldr;
    const int M = 1 + 2 * d.Wn;
    for (int c = threadIdx.x; c < d.I; c += blockDim.x) {
        float v = x[c];
        int off, width;
        if (c < d.oE) { off = d.oV; width = d.oE - d.oV; }
        else if (c < d.oF) { off = d.oE; width = d.oF - d.oE; v = dnc_sigmoid(v); }
        else if (c < d.oAg) { off = d.oF; width = d.oAg - d.oF; v = dnc_sigmoid(v); }
        else if (c < d.oWg) { off = d.oAg; width = d.oWg - d.oAg; v = dnc_sigmoid(v); }
        else if (c < d.oRm) { off = d.oWg; width = d.oRm - d.oWg; v = dnc_sigmoid(v); }
        else if (c < d.oKw) {
            off = d.oRm; width = d.oKw - d.oRm;
            const int i = (c - d.oRm) / M;
            const float* r = x + d.oRm + i * M;
            float mx = r[0];
            for (int m = 1; m < M; ++m) mx = fmaxf(mx, r[m]);
            float sum = 0.f;
            for (int m = 0; m < M; ++m) sum += expf(r[m] - mx);
            v = expf(v - mx) / sum;
        }
        else if (c < d.oBw) { off = d.oKw; width = d.oBw - d.oKw; }
        else if (c < d.oKr) { off = d.oBw; width = d.oKr - d.oBw; }
        else if (c < d.oBr) { off = d.oKr; width = d.oBr - d.oKr; }
        else { off = d.oBr; width = d.I - d.oBr; }
        act[(size_t)B * off + (size_t)b * width + (c - off)] = v;
    }
}

// _write_weights mix (access.py:247-257): ww = write_gate * (alloc_gate * allocation + (1 - alloc_gate) * content);
// also the gate product handed to write_allocation_weights (:238-241)
__global__ void gate_product_kernel(const float* __restrict__ ag, const float* __restrict__ wg, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = ag[i] * wg[i];
}
__global__ void write_mix_kernel(const float* __restrict__ alloc, const float* __restrict__ content, const float* __restrict__ ag,
                                 const float* __restrict__ wg, float* __restrict__ out, int N) {
    const int bj = blockIdx.x;
    const float a = ag[bj], g = wg[bj];
    for (int n = threadIdx.x; n < N; n += blockDim.x)
        out[(size_t)bj * N + n] = g * (a * alloc[(size_t)bj * N + n] + (1.0f - a) * content[(size_t)bj * N + n]);
}

// _erase_and_write (access.py:32-63): M' = M * prod_j (1 - ww_j (x) e_j) + sum_j ww_j (x) v_j
__global__ void erase_write_kernel(const float* __restrict__ mem, const float* __restrict__ ww, const float* __restrict__ er,
                                   const float* __restrict__ val, float* __restrict__ out, int N, int W, int Wn) {
    const int b = blockIdx.x;
    for (int idx = threadIdx.x; idx < N * W; idx += blockDim.x) {
        const int n = idx / W, w = idx - n * W;
        float keep = 1.f, add = 0.f;
        for (int j = 0; j < Wn; ++j) {
            const float a = ww[((size_t)b * Wn + j) * N + n];
            keep *= (1.0f - a * er[((size_t)b * Wn + j) * W + w]);
            add += a * val[((size_t)b * Wn + j) * W + w];
        }
        out[(size_t)b * N * W + idx] = mem[(size_t)b * N * W + idx] * keep + add;
    }
}

// _read_weights mix (access.py:283-303): read_mode [B,R,1+2Wn] = [backward x Wn | forward x Wn | content]
__global__ void read_mix_kernel(const float* __restrict__ content, const float* __restrict__ fwd, const float* __restrict__ bwd,
                                const float* __restrict__ mode, float* __restrict__ out, int N, int Wn) {
    const int bi = blockIdx.x;
    const float* m = mode + (size_t)bi * (1 + 2 * Wn);
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float v = m[2 * Wn] * content[(size_t)bi * N + n];
        float f = 0.f, bsum = 0.f;
        for (int j = 0; j < Wn; ++j) {
            f += m[Wn + j] * fwd[((size_t)bi * Wn + j) * N + n];
            bsum += m[j] * bwd[((size_t)bi * Wn + j) * N + n];
        }
        out[(size_t)bi * N + n] = v + f + bsum;
    }
}

// read_words = read_weights @ memory (access.py:151)
__global__ void read_words_kernel(const float* __restrict__ rw, const float* __restrict__ mem, float* __restrict__ out, int N,
                                  int W, int R) {
    const int bi = blockIdx.x, b = bi / R;
    for (int w = threadIdx.x; w < W; w += blockDim.x) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += rw[(size_t)bi * N + n] * mem[((size_t)b * N + n) * W + w];
        out[(size_t)bi * W + w] = s;
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

extern "C" int ntk_dnc_interface_activations(const float* raw, int ldr, float* act, int B, int N, int W, int R, int Wn,
                                             void* stream) {
    NTK_REQUIRE(raw && act, NTK_ERR_BAD_PTR, "ntk_dnc_interface_activations: null pointer");
    DncDims d;
    dnc_fill_dims(d, B, 1, N, W, R, Wn, 4, 1, 0.f);
    NTK_REQUIRE(B > 0 && ldr >= d.I, NTK_ERR_BAD_SHAPE, "ntk_dnc_interface_activations: B=%d ldr=%d (interface %d)", B, ldr, d.I);
    interface_act_kernel<<<B, 256, 0, (hipStream_t)stream>>>(raw, ldr, act, d);
    NTK_CHECK_LAUNCH("ntk_dnc_interface_activations");
    return NTK_OK;
}

extern "C" int ntk_dnc_write_weights(const float* memory, const float* usage, const float* write_keys, const float* write_strengths,
                                     const float* allocation_gate, const float* write_gate, float* write_weights,
                                     float* workspace, int B, int N, int W, int Wn, void* stream) {
    NTK_REQUIRE(memory && usage && write_keys && write_strengths && allocation_gate && write_gate && write_weights && workspace,
                NTK_ERR_BAD_PTR, "ntk_dnc_write_weights: null pointer");
    float* content = workspace;                       // [B,Wn,N]
    float* alloc = content + (size_t)B * Wn * N;      // [B,Wn,N]
    float* gates = alloc + (size_t)B * Wn * N;        // [B,Wn]
    int rc = ntk_dnc_cosine_weights(memory, write_keys, write_strengths, content, B, N, W, Wn, stream);
    if (rc) return rc;
    gate_product_kernel<<<(B * Wn + 255) / 256, 256, 0, (hipStream_t)stream>>>(allocation_gate, write_gate, gates, B * Wn);
    rc = ntk_dnc_write_allocation_weights(usage, gates, alloc, B, N, Wn, stream);
    if (rc) return rc;
    write_mix_kernel<<<B * Wn, 256, 0, (hipStream_t)stream>>>(alloc, content, allocation_gate, write_gate, write_weights, N);
    NTK_CHECK_LAUNCH("ntk_dnc_write_weights");
    return NTK_OK;
}

extern "C" int ntk_dnc_erase_and_write(const float* memory, const float* address, const float* reset_weights, const float* values,
                                       float* out, int B, int N, int W, int Wn, void* stream) {
    NTK_REQUIRE(memory && address && reset_weights && values && out, NTK_ERR_BAD_PTR, "ntk_dnc_erase_and_write: null pointer");
    NTK_REQUIRE(B > 0 && N > 0 && W > 0 && Wn > 0, NTK_ERR_BAD_SHAPE, "ntk_dnc_erase_and_write: B=%d N=%d W=%d Wn=%d", B, N, W, Wn);
    erase_write_kernel<<<B, 1024, 0, (hipStream_t)stream>>>(memory, address, reset_weights, values, out, N, W, Wn);
    NTK_CHECK_LAUNCH("ntk_dnc_erase_and_write");
    return NTK_OK;
}

extern "C" int ntk_dnc_read_weights(const float* memory, const float* prev_read_weights, const float* link, const float* read_keys,
                                    const float* read_strengths, const float* read_mode, float* read_weights, float* workspace,
                                    int B, int N, int W, int R, int Wn, void* stream) {
    NTK_REQUIRE(memory && prev_read_weights && link && read_keys && read_strengths && read_mode && read_weights && workspace,
                NTK_ERR_BAD_PTR, "ntk_dnc_read_weights: null pointer");
    float* content = workspace;                        // [B,R,N]
    float* fwd = content + (size_t)B * R * N;          // [B,R,Wn,N]
    float* bwd = fwd + (size_t)B * R * Wn * N;         // [B,R,Wn,N]
    int rc = ntk_dnc_cosine_weights(memory, read_keys, read_strengths, content, B, N, W, R, stream);
    if (rc) return rc;
    rc = ntk_dnc_directional_read_weights(link, prev_read_weights, fwd, B, N, Wn, R, 1, stream);
    if (rc) return rc;
    rc = ntk_dnc_directional_read_weights(link, prev_read_weights, bwd, B, N, Wn, R, 0, stream);
    if (rc) return rc;
    read_mix_kernel<<<B * R, 256, 0, (hipStream_t)stream>>>(content, fwd, bwd, read_mode, read_weights, N, Wn);
    NTK_CHECK_LAUNCH("ntk_dnc_read_weights");
    return NTK_OK;
}

extern "C" int ntk_dnc_read_words(const float* read_weights, const float* memory, float* out, int B, int N, int W, int R,
                                  void* stream) {
    NTK_REQUIRE(read_weights && memory && out, NTK_ERR_BAD_PTR, "ntk_dnc_read_words: null pointer");
    NTK_REQUIRE(B > 0 && N > 0 && W > 0 && R > 0, NTK_ERR_BAD_SHAPE, "ntk_dnc_read_words: B=%d N=%d W=%d R=%d", B, N, W, R);
    read_words_kernel<<<B * R, 128, 0, (hipStream_t)stream>>>(read_weights, memory, out, N, W, R);
    NTK_CHECK_LAUNCH("ntk_dnc_read_words");
    return NTK_OK;
}

// workspace of ntk_dnc_access_step_fwd: activated interface + the larger of the write / read scratch areas
extern "C" size_t ntk_dnc_access_step_workspace_bytes(int B, int N, int W, int R, int Wn) {
    DncDims d;
    dnc_fill_dims(d, B, 1, N, W, R, Wn, 4, 1, 0.f);
    const size_t wr = (size_t)2 * B * Wn * N + (size_t)B * Wn, rd = (size_t)B * R * N * (1 + 2 * Wn);
    return ((size_t)B * d.IP + (wr > rd ? wr : rd)) * sizeof(float);
}

// MemoryAccess._build (access.py:113-158) from the RAW outputs of the ten interface linears (packed order, row
// stride ldr): usage -> write weights -> erase/write -> linkage -> read weights -> read words.  Every *_out buffer
// is distinct from its input (the step reads the previous state while it writes the next).
extern "C" int ntk_dnc_access_step_fwd(const float* iface_raw, int ldr, const float* memory, const float* read_weights,
                                       const float* write_weights, const float* link, const float* precedence,
                                       const float* usage, float* memory_out, float* read_weights_out, float* write_weights_out,
                                       float* link_out, float* precedence_out, float* usage_out, float* read_words,
                                       float* workspace, int B, int N, int W, int R, int Wn, void* stream) {
    NTK_REQUIRE(iface_raw && memory && read_weights && write_weights && link && precedence && usage && memory_out &&
                    read_weights_out && write_weights_out && link_out && precedence_out && usage_out && read_words && workspace,
                NTK_ERR_BAD_PTR, "ntk_dnc_access_step_fwd: null pointer");
    NTK_REQUIRE(memory != memory_out && link != link_out && read_weights != read_weights_out && usage != usage_out,
                NTK_ERR_BAD_PTR, "ntk_dnc_access_step_fwd: outputs must not alias the previous state");
    DncDims d;
    dnc_fill_dims(d, B, 1, N, W, R, Wn, 4, 1, 0.f);
    float* act = workspace;
    float* scratch = workspace + (size_t)B * d.IP;
    auto f = [&](int off) { return act + (size_t)B * off; };
    int rc = ntk_dnc_interface_activations(iface_raw, ldr, act, B, N, W, R, Wn, stream);
    if (rc) return rc;
    if ((rc = ntk_dnc_freeness(write_weights, f(d.oF), read_weights, usage, usage_out, B, N, Wn, R, stream))) return rc;
    if ((rc = ntk_dnc_write_weights(memory, usage_out, f(d.oKw), f(d.oBw), f(d.oAg), f(d.oWg), write_weights_out, scratch, B, N, W,
                                    Wn, stream))) return rc;
    if ((rc = ntk_dnc_erase_and_write(memory, write_weights_out, f(d.oE), f(d.oV), memory_out, B, N, W, Wn, stream))) return rc;
    if ((rc = ntk_dnc_linkage(link, precedence, write_weights_out, link_out, precedence_out, B, N, Wn, stream))) return rc;
    if ((rc = ntk_dnc_read_weights(memory_out, read_weights, link_out, f(d.oKr), f(d.oBr), f(d.oRm), read_weights_out, scratch, B,
                                   N, W, R, Wn, stream))) return rc;
    return ntk_dnc_read_words(read_weights_out, memory_out, read_words, B, N, W, R, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of ONE MemoryAccess step at module granularity (what tf.gradients gives dnc/access_test.py:145-159: the
// gradient of a function of the step's outputs w.r.t. the raw interface and the previous memory / read weights / link /
// precedence / usage).  The step is recomputed from the previous state with the module kernels above into the record
// layout of the sequence BPTT kernels, which then run for S = 1 with a dummy 4-unit controller whose weights are zero:
// upstream gradients enter through the carried-gradient buffers (carry_in), previous-state gradients leave through them.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

struct AccessBwdWs {
    size_t act, rows, usage, ww, wws, M, L, prec, rws, rw, WrT, WiT, Wy, gates, c, hc0, ypre, dout, dgates, dypre, gcarry, total;
    int ldkT, ncar;
};

void access_bwd_ws(const DncDims& d, AccessBwdWs& w) {
    const size_t B = d.B, N = d.N, W = d.W, R = d.R, Wn = d.Wn, hid = d.hid;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 3) & ~(size_t)3; return r; };
    w.ldkT = (d.K + 3) & ~3;
    w.ncar = (int)(Wn * N + N + R * N + w.ldkT + hid);
    w.act = take(B * d.IP); w.rows = take(B * d.IP); w.usage = take(B * N); w.ww = take(B * Wn * N);
    w.wws = take(2 * B * Wn * N + B * Wn); w.M = take(B * N * W); w.L = take(B * Wn * N * N); w.prec = take(B * Wn * N);
    w.rws = take(B * R * N * (1 + 2 * Wn)); w.rw = take(B * R * N);
    w.WrT = take(4 * hid * w.ldkT); w.WiT = take((size_t)d.IP * 4); w.Wy = take((size_t)d.ldy * d.OP);
    w.gates = take(B * 4 * hid); w.c = take(B * hid); w.hc0 = take(B * 2 * hid); w.ypre = take(B * d.O); w.dout = take(B * d.O);
    w.dgates = take(B * 4 * hid); w.dypre = take(B * d.OP); w.gcarry = take(B * w.ncar);
    w.total = o;
}

// field-major activations (field f of every batch element contiguous) -> one interface row per batch element, in the
// sequence kernels' record convention
__global__ void ifc_rows_kernel(const float* __restrict__ act, float* __restrict__ rows, DncDims d) {
    const int offs[11] = {d.oV, d.oE, d.oF, d.oAg, d.oWg, d.oRm, d.oKw, d.oBw, d.oKr, d.oBr, d.I};
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < d.B * d.IP; idx += gridDim.x * blockDim.x) {
        const int b = idx / d.IP, c = idx - b * d.IP;
        float v = 0.f;
        if (c < d.I) {
            int f = 0;
            while (c >= offs[f + 1]) ++f;
            const int width = offs[f + 1] - offs[f];
            v = act[(size_t)d.B * offs[f] + (size_t)b * width + (c - offs[f])];
            // the module activations leave the strengths raw (CosineWeights applies softplus itself, addressing.py:96-101);
            // the sequence kernels record them activated
            if (f == 7 || f == 9) v = dnc_softplus(v);
        }
        rows[idx] = v;
    }
}

// carried-gradient rows [precedence (Wn N) | usage (N) | read weights (R N) | d[reads ; h] (ldkT) | cell (hid)]
__global__ void access_carry_kernel(float* __restrict__ gcarry, float* __restrict__ g_prec, float* __restrict__ g_usage,
                                    float* __restrict__ g_rw, const float* __restrict__ d_reads, int B, int N, int R, int Wn,
                                    int RW, int ncar, int scatter) {
    const int HN = Wn * N, RN = R * N;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < B * ncar; idx += gridDim.x * blockDim.x) {
        const int b = idx / ncar, c = idx - b * ncar;
        if (scatter) {
            if (c < HN) g_prec[(size_t)b * HN + c] = gcarry[idx];
            else if (c < HN + N) g_usage[(size_t)b * N + (c - HN)] = gcarry[idx];
            else if (c < HN + N + RN) g_rw[(size_t)b * RN + (c - HN - N)] = gcarry[idx];
        } else {
            float v = 0.f;
            if (c < HN) v = g_prec[(size_t)b * HN + c];
            else if (c < HN + N) v = g_usage[(size_t)b * N + (c - HN)];
            else if (c < HN + N + RN) v = g_rw[(size_t)b * RN + (c - HN - N)];
            else if (c < HN + N + RN + RW) v = d_reads[(size_t)b * RW + (c - HN - N - RN)];
            gcarry[idx] = v;
        }
    }
}

}  // namespace

extern "C" size_t ntk_dnc_access_step_bwd_workspace_bytes(int B, int N, int W, int R, int Wn) {
    DncDims d;
    dnc_fill_dims(d, B, 1, N, W, R, Wn, 4, 1, 0.f);
    AccessBwdWs w;
    access_bwd_ws(d, w);
    return w.total * sizeof(float);
}

extern "C" int ntk_dnc_access_step_bwd(const float* iface_raw, int ldr, const float* memory, const float* read_weights,
                                       const float* write_weights, const float* link, const float* precedence,
                                       const float* usage, const float* d_read_words, float* g_memory, float* g_read_weights,
                                       float* g_link, float* g_precedence, float* g_usage, float* d_iface_raw,
                                       float* workspace, int B, int N, int W, int R, int Wn, void* stream) {
    NTK_REQUIRE(iface_raw && memory && read_weights && write_weights && link && precedence && usage && d_read_words && g_memory &&
                    g_read_weights && g_link && g_precedence && g_usage && d_iface_raw && workspace,
                NTK_ERR_BAD_PTR, "ntk_dnc_access_step_bwd: null pointer");
    NTK_REQUIRE(B > 0 && N >= 4 && (N % 4) == 0 && W >= 4 && (W % 4) == 0 && R >= 1 && R <= 4 && Wn >= 1 && Wn <= 4,
                NTK_ERR_UNSUPPORTED, "ntk_dnc_access_step_bwd: N=%d W=%d (multiples of 4) R=%d (1..4) Wn=%d (1..4)", N, W, R, Wn);
    DncDims d;
    dnc_fill_dims(d, B, 1, N, W, R, Wn, 4, 1, 0.f);
    AccessBwdWs w;
    access_bwd_ws(d, w);
    hipStream_t st = (hipStream_t)stream;
    float* ws = workspace;
    auto f = [&](int off) { return ws + w.act + (size_t)B * off; };
    if (hipMemsetAsync(ws + w.WrT, 0, (w.gcarry - w.WrT) * sizeof(float), st) != hipSuccess) return NTK_ERR_HIP;   // dummies, zero dout
    int rc;
    // the step, recomputed into the record layout
    if ((rc = ntk_dnc_interface_activations(iface_raw, ldr, ws + w.act, B, N, W, R, Wn, stream))) return rc;
    ifc_rows_kernel<<<(B * d.IP + 255) / 256, 256, 0, st>>>(ws + w.act, ws + w.rows, d);
    if ((rc = ntk_dnc_freeness(write_weights, f(d.oF), read_weights, usage, ws + w.usage, B, N, Wn, R, stream))) return rc;
    if ((rc = ntk_dnc_write_weights(memory, ws + w.usage, f(d.oKw), f(d.oBw), f(d.oAg), f(d.oWg), ws + w.ww, ws + w.wws, B, N, W, Wn,
                                    stream))) return rc;
    if ((rc = ntk_dnc_erase_and_write(memory, ws + w.ww, f(d.oE), f(d.oV), ws + w.M, B, N, W, Wn, stream))) return rc;
    if ((rc = ntk_dnc_linkage(link, precedence, ws + w.ww, ws + w.L, ws + w.prec, B, N, Wn, stream))) return rc;
    if ((rc = ntk_dnc_read_weights(ws + w.M, read_weights, ws + w.L, f(d.oKr), f(d.oBr), f(d.oRm), ws + w.rw, ws + w.rws, B, N, W, R, Wn,
                                   stream))) return rc;
    // upstream gradients -> carried-gradient rows
    access_carry_kernel<<<(B * w.ncar + 255) / 256, 256, 0, st>>>(ws + w.gcarry, g_precedence, g_usage, g_read_weights, d_read_words, B, N,
                                                                 R, Wn, R * W, w.ncar, 0);
    NTK_CHECK_LAUNCH("ntk_dnc_access_step_bwd(setup)");
    const float* cw = ws + w.wws;
    const float* al = cw + (size_t)B * Wn * N;
    const float* cr = ws + w.rws;
    const float* fw = cr + (size_t)B * R * N;
    const float* bw = fw + (size_t)B * R * Wn * N;
    rc = ntk_dnc_seq_bwd(B, 1, N, W, R, Wn, 4, 1, 0.f, ws + w.WrT, w.ldkT, ws + w.WiT, 4, ws + w.Wy, memory, link, usage, read_weights,
                         write_weights, precedence, ws + w.hc0, ws + w.gates, ws + w.c, ws + w.rows, ws + w.usage, ws + w.ww, ws + w.rw, cw, cr,
                         al, ws + w.prec, fw, bw, ws + w.M, ws + w.L, ws + w.ypre, ws + w.dout, g_memory, g_link, ws + w.dgates, d_iface_raw,
                         ws + w.dypre, ws + w.gcarry, 1, stream);
    if (rc) return rc;
    access_carry_kernel<<<(B * w.ncar + 255) / 256, 256, 0, st>>>(ws + w.gcarry, g_precedence, g_usage, g_read_weights, d_read_words, B, N,
                                                                 R, Wn, R * W, w.ncar, 1);
    NTK_CHECK_LAUNCH("ntk_dnc_access_step_bwd");
    return NTK_OK;
}
