// Stand-alone NTM addressing ops (ops.py) -- the module-level functions the reference's ops_test.py calls.
// The sequence kernels fuse the same arithmetic; these take plain [B,H,*] tensors.
#include "common.h"

namespace {

// batched_smooth_cosine_similarity (ops.py:135-158).  memory [B,N,Md], keys [B,H,Md] -> out [B,H,N].
// mode 0 ("as coded", quirk Q1): memory is transposed to [B,Md,N] and l2-normalised along the SLOT axis
//   (each feature column divided by sqrt(max(sum_n M[n,m]^2, 1e-12))), keys normalised per head over Md;
// mode 1 ("smooth_cosine", what ops_test.py:20-34 expects): dot / (|m_n| |k| + 1e-3) per memory row.
__global__ __launch_bounds__(256) void cosine_similarity_kernel(const float* __restrict__ mem, const float* __restrict__ keys,
                                                                 float* __restrict__ out, int N, int Md, int H, int mode) {
    extern __shared__ float sm[];          // mode 0: [Md] column scales, [H*Md] normalised keys; mode 1: [H] key norms
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* M = mem + (size_t)b * N * Md;
    const float* K = keys + (size_t)b * H * Md;
    if (mode == 0) {
        float* cs = sm;
        float* kh = sm + Md;
        for (int m = tid; m < Md; m += blockDim.x) {
            float s = 0.f;
            for (int n = 0; n < N; ++n) { const float v = M[(size_t)n * Md + m]; s += v * v; }
            cs[m] = rsqrtf(fmaxf(s, 1e-12f));
        }
        for (int h = tid; h < H; h += blockDim.x) {
            float s = 0.f;
            for (int m = 0; m < Md; ++m) s += K[h * Md + m] * K[h * Md + m];
            const float r = rsqrtf(fmaxf(s, 1e-12f));
            for (int m = 0; m < Md; ++m) kh[h * Md + m] = K[h * Md + m] * r;
        }
        __syncthreads();
        for (int idx = tid; idx < H * N; idx += blockDim.x) {
            const int h = idx / N, n = idx - h * N;
            float s = 0.f;
            for (int m = 0; m < Md; ++m) s += kh[h * Md + m] * (M[(size_t)n * Md + m] * cs[m]);
            out[((size_t)b * H + h) * N + n] = s;
        }
    } else {
        float* kn = sm;
        for (int h = tid; h < H; h += blockDim.x) {
            float s = 0.f;
            for (int m = 0; m < Md; ++m) s += K[h * Md + m] * K[h * Md + m];
            kn[h] = sqrtf(s);
        }
        __syncthreads();
        for (int idx = tid; idx < H * N; idx += blockDim.x) {
            const int h = idx / N, n = idx - h * N;
            float dot = 0.f, ms = 0.f;
            for (int m = 0; m < Md; ++m) { const float v = M[(size_t)n * Md + m]; dot += K[h * Md + m] * v; ms += v * v; }
            out[((size_t)b * H + h) * N + n] = dot / (sqrtf(ms) * kn[h] + 1e-3f);
        }
    }
}

// batched_circular_convolution (ops.py:180-214) with the taps the Python-2 source evaluates (quirk Q2):
// start = floor(-SS/2), out[i] = sum_j kernel[j] * w[(i + start + j) mod N]   (SS = 3 -> taps -2,-1,0)
__global__ void circular_convolution_kernel(const float* __restrict__ w, const float* __restrict__ kern, float* __restrict__ out,
                                            int N, int SS) {
    const int bh = blockIdx.x;
    const int start = -((SS + 1) / 2);              // floor(-SS / 2) for odd and even SS
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        float s = 0.f;
        for (int j = 0; j < SS; ++j) {
            int src = (i + start + j) % N;
            if (src < 0) src += N;
            s += kern[(size_t)bh * SS + j] * w[(size_t)bh * N + src];
        }
        out[(size_t)bh * N + i] = s;
    }
}

// The intermediate tensors of one NTMCell step that the reference returns in `debug` but the fused step keeps in
// registers (ntm_cell.py:153-206): w_gated = g wc + (1 - g) w_prev (:153-156), powed_w_conv = w_conv ^ gamma (:173),
// M_erase = prod_heads (1 - w_write (x) erase) (:197-199), M_write = sum_heads w_write (x) add (:201-203), and the shift weights
// sw = softmax over the shift space (:161).
// Inputs are what the step records: g / gamma / erase / add as activated ([B,H], [B,H], [B,Wh,Md], [B,Wh,Md] inside the
// control vector u with row stride ldu), wc / wv / w [B,H,N], w_prev [B,H,N].
__global__ __launch_bounds__(256) void step_debug_kernel(const float* __restrict__ u, int ldu, int oG, int oY, int oE, int oA,
                                                          const float* __restrict__ wc, const float* __restrict__ wv,
                                                          const float* __restrict__ w, const float* __restrict__ w_prev,
                                                          float* __restrict__ w_gated, float* __restrict__ powed,
                                                          float* __restrict__ M_write, float* __restrict__ M_erase,
                                                          float* __restrict__ sw, int oS, int SS,
                                                          int N, int Md, int R, int Wh) {
    const int b = blockIdx.x, H = R + Wh;
    const float* ub = u + (size_t)b * ldu;
    for (int h = threadIdx.x; h < H; h += blockDim.x) {          // shift weights: softmax over the shift space (:161; u holds the raw values)
        float mx = -INFINITY, s = 0.f;
        for (int j = 0; j < SS; ++j) mx = fmaxf(mx, ub[oS + h * SS + j]);
        for (int j = 0; j < SS; ++j) s += expf(ub[oS + h * SS + j] - mx);
        for (int j = 0; j < SS; ++j) sw[((size_t)b * H + h) * SS + j] = expf(ub[oS + h * SS + j] - mx) / s;
    }
    for (int idx = threadIdx.x; idx < H * N; idx += blockDim.x) {
        const int h = idx / N;
        const size_t o = (size_t)b * H * N + idx;
        const float g = ub[oG + h], gamma = ub[oY + h];
        w_gated[o] = wc[o] * g + w_prev[o] * (1.0f - g);
        powed[o] = powf(wv[o], gamma);
    }
    for (int idx = threadIdx.x; idx < N * Md; idx += blockDim.x) {
        const int n = idx / Md, m = idx - n * Md;
        float er = 1.f, wr = 0.f;
        for (int j = 0; j < Wh; ++j) {
            const float ww = w[((size_t)b * H + R + j) * N + n];
            er *= 1.0f - ww * ub[oE + j * Md + m];
            wr += ww * ub[oA + j * Md + m];
        }
        M_erase[(size_t)b * N * Md + idx] = er;
        M_write[(size_t)b * N * Md + idx] = wr;
    }
}

}  // namespace

extern "C" int ntk_ntm_step_debug(const float* u, int ldu, int oG, int oS, int shift_space, int oY, int oE, int oA, const float* wc, const float* wv,
                                  const float* w, const float* w_prev, float* sw, float* w_gated, float* w_conv_powed, float* M_write,
                                  float* M_erase, int B, int N, int Md, int R, int Wh, void* stream) {
    NTK_REQUIRE(u && wc && wv && w && w_prev && sw && w_gated && w_conv_powed && M_write && M_erase, NTK_ERR_BAD_PTR, "ntk_ntm_step_debug: null pointer");
    NTK_REQUIRE(oS >= 0 && shift_space > 0 && oS + (R + Wh) * shift_space <= ldu, NTK_ERR_BAD_SHAPE, "ntk_ntm_step_debug: shift block");
    NTK_REQUIRE(B > 0 && N > 0 && Md > 0 && R >= 0 && Wh > 0 && ldu > 0 && oG >= 0 && oY >= 0 && oE >= 0 && oA + Wh * Md <= ldu, NTK_ERR_BAD_SHAPE,
                "ntk_ntm_step_debug: B=%d N=%d Md=%d R=%d Wh=%d ldu=%d", B, N, Md, R, Wh, ldu);
    step_debug_kernel<<<B, 256, 0, (hipStream_t)stream>>>(u, ldu, oG, oY, oE, oA, wc, wv, w, w_prev, w_gated, w_conv_powed, M_write, M_erase,
                                                         sw, oS, shift_space, N, Md, R, Wh);
    NTK_CHECK_LAUNCH("ntk_ntm_step_debug");
    return NTK_OK;
}

extern "C" int ntk_ntm_cosine_similarity(const float* memory, const float* keys, float* out, int B, int N, int Md, int H,
                                         int mode, void* stream) {
    NTK_REQUIRE(memory && keys && out, NTK_ERR_BAD_PTR, "ntk_ntm_cosine_similarity: null pointer");
    NTK_REQUIRE(B > 0 && N > 0 && Md > 0 && H > 0 && (mode == 0 || mode == 1), NTK_ERR_BAD_SHAPE,
                "ntk_ntm_cosine_similarity: B=%d N=%d Md=%d H=%d mode=%d", B, N, Md, H, mode);
    const size_t lds = (mode == 0 ? (size_t)(Md + H * Md) : (size_t)H) * sizeof(float);
    NTK_REQUIRE(lds <= 64 * 1024, NTK_ERR_UNSUPPORTED, "ntk_ntm_cosine_similarity: H*Md too large");
    cosine_similarity_kernel<<<B, 256, lds, (hipStream_t)stream>>>(memory, keys, out, N, Md, H, mode);
    NTK_CHECK_LAUNCH("ntk_ntm_cosine_similarity");
    return NTK_OK;
}

extern "C" int ntk_ntm_circular_convolution(const float* w, const float* kernel, float* out, int B, int H, int N, int shift_space,
                                            void* stream) {
    NTK_REQUIRE(w && kernel && out && w != out, NTK_ERR_BAD_PTR, "ntk_ntm_circular_convolution: null or aliased pointer");
    NTK_REQUIRE(B > 0 && H > 0 && N > 0 && shift_space > 0 && shift_space <= N, NTK_ERR_BAD_SHAPE,
                "ntk_ntm_circular_convolution: B=%d H=%d N=%d shift_space=%d", B, H, N, shift_space);
    circular_convolution_kernel<<<B * H, 256, 0, (hipStream_t)stream>>>(w, kernel, out, N, shift_space);
    NTK_CHECK_LAUNCH("ntk_ntm_circular_convolution");
    return NTK_OK;
}
