// Step-granular entry points of the C ABI (SURVEY 8b minimum export set): the single-step forms of the sequence
// kernels, the BasicLSTMCell pointwise step, a stand-alone 2x2 max pool and the split loss entry points.
// The training/inference hot path uses the persistent sequence kernels and the pool fused into the conv epilogue;
// these exist so a caller that drives the cell one step() at a time (test_tracker.py:340-341) binds the same library.
#include "common.h"

// ---------------------------------------------------------------------------------------------------------
// one NTM cell step = the persistent kernel with S = 1 (ntm_cell.py:53-253)
// ---------------------------------------------------------------------------------------------------------
extern "C" int ntk_ntm_step_fwd(int B, int N, int Md, int R, int Wh, int hid, int shift_range, int O, int write_first,
                                const float* xproj, const float* Wr, const float* Wa,
                                const float* M_prev, const float* w_prev, const float* read_prev, const float* cs_prev,
                                float* logits, float* outputs, float* M, float* w, float* read, float* cs,
                                float* st_z, float* st_gates, float* st_c, float* st_h, float* st_u,
                                float* st_wc, float* st_wv, float* st_w, float* st_M, float* st_read, void* stream) {
    return ntk_ntm_seq_fwd(B, 1, N, Md, R, Wh, hid, shift_range, O, write_first, xproj, Wr, Wa, M_prev, w_prev, read_prev, cs_prev,
                           logits, outputs, M, w, read, cs, st_z, st_gates, st_c, st_h, st_u, st_wc, st_wv, st_w, st_M, st_read,
                           stream);
}

extern "C" int ntk_ntm_step_bwd(int B, int N, int Md, int R, int Wh, int hid, int shift_range, int O, int write_first,
                                const float* WrT, int ldkT, const float* WaT, int ldhT,
                                const float* M_prev, const float* w_prev, const float* cs_prev,
                                const float* st_gates, const float* st_c, const float* st_u,
                                const float* st_wc, const float* st_wv, const float* st_w, const float* st_M,
                                const float* dlogits,
                                const float* dM, const float* dw, const float* dread, const float* dcs,
                                float* dgates, float* du, float* dM_prev, float* dw_prev, float* dread_prev, float* dcs_prev,
                                void* stream) {
    return ntk_ntm_seq_bwd(B, 1, N, Md, R, Wh, hid, shift_range, O, write_first, WrT, ldkT, WaT, ldhT, M_prev, w_prev, cs_prev,
                           st_gates, st_c, st_u, st_wc, st_wv, st_w, st_M, dlogits, dM, dw, dread, dcs, dgates, du,
                           dM_prev, dw_prev, dread_prev, dcs_prev, stream);
}

// ---------------------------------------------------------------------------------------------------------
// tf.contrib.rnn.BasicLSTMCell pointwise step (ntm_cell.py:45-50): pre [B,4*hid] = [x,h] W + b in TF's block
// order i | j | f | o;  c' = c * sigmoid(f + forget_bias) + sigmoid(i) * tanh(j);  h' = tanh(c') * sigmoid(o)
// ---------------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void lstm_step_fwd_kernel(const float* __restrict__ pre, const float* __restrict__ c_prev, float fb,
                                     float* __restrict__ c, float* __restrict__ h, float* __restrict__ act, int B, int hid) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * hid) return;
    const int b = idx / hid, u = idx - b * hid;
    const float* p = pre + (size_t)b * 4 * hid;
    const float gi = sigm(p[u]), gj = tanhf(p[hid + u]), gf = sigm(p[2 * hid + u] + fb), go = sigm(p[3 * hid + u]);
    const float c2 = c_prev[idx] * gf + gi * gj;
    c[idx] = c2;
    h[idx] = tanhf(c2) * go;
    if (act) {
        float* a = act + (size_t)b * 4 * hid;
        a[u] = gi; a[hid + u] = gj; a[2 * hid + u] = gf; a[3 * hid + u] = go;
    }
}

__global__ void lstm_step_bwd_kernel(const float* __restrict__ act, const float* __restrict__ c_prev, const float* __restrict__ c,
                                     const float* __restrict__ dh, const float* __restrict__ dc, float* __restrict__ dpre,
                                     float* __restrict__ dc_prev, int B, int hid) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * hid) return;
    const int b = idx / hid, u = idx - b * hid;
    const float* a = act + (size_t)b * 4 * hid;
    const float gi = a[u], gj = a[hid + u], gf = a[2 * hid + u], go = a[3 * hid + u];
    const float tc = tanhf(c[idx]);
    const float g_h = dh ? dh[idx] : 0.f;
    const float dc2 = (dc ? dc[idx] : 0.f) + g_h * go * (1.0f - tc * tc);
    float* d = dpre + (size_t)b * 4 * hid;
    d[u] = dc2 * gj * gi * (1.0f - gi);
    d[hid + u] = dc2 * gi * (1.0f - gj * gj);
    d[2 * hid + u] = dc2 * c_prev[idx] * gf * (1.0f - gf);
    d[3 * hid + u] = g_h * tc * go * (1.0f - go);
    dc_prev[idx] = dc2 * gf;
}

// slim.max_pool2d [2,2] stride 2 VALID on NHWC fp32 (vgg.py:155-161); 4 channels per thread
__global__ void maxpool2x2_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int H, int W, int C) {
    const int Ho = H >> 1, Wo = W >> 1, C4 = C >> 2;
    const size_t tot = (size_t)n * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        size_t r = i / C4;
        const int x = (int)(r % Wo); r /= Wo;
        const int y = (int)(r % Ho);
        const int f = (int)(r / Ho);
        const f32x4* p = reinterpret_cast<const f32x4*>(in + (((size_t)f * H + 2 * y) * W + 2 * x) * C) + c4;
        const f32x4 a = p[0], b = p[C4], c = p[(size_t)W * C4], d = p[(size_t)W * C4 + C4];
        f32x4 m;
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(a[e], b[e]), fmaxf(c[e], d[e]));
        reinterpret_cast<f32x4*>(out)[i] = m;
    }
}
}  // namespace

extern "C" int ntk_lstm_step_fwd(const float* pre, const float* c_prev, float forget_bias, float* c, float* h, float* act,
                                 int B, int hid, void* stream) {
    NTK_REQUIRE(pre && c_prev && c && h, NTK_ERR_BAD_PTR, "ntk_lstm_step_fwd: null pointer");
    NTK_REQUIRE(B > 0 && hid > 0, NTK_ERR_BAD_SHAPE, "ntk_lstm_step_fwd: B=%d hid=%d", B, hid);
    lstm_step_fwd_kernel<<<(B * hid + 255) / 256, 256, 0, (hipStream_t)stream>>>(pre, c_prev, forget_bias, c, h, act, B, hid);
    NTK_CHECK_LAUNCH("ntk_lstm_step_fwd");
    return NTK_OK;
}

extern "C" int ntk_lstm_step_bwd(const float* act, const float* c_prev, const float* c, const float* dh, const float* dc,
                                 float* dpre, float* dc_prev, int B, int hid, void* stream) {
    NTK_REQUIRE(act && c_prev && c && dpre && dc_prev && (dh || dc), NTK_ERR_BAD_PTR, "ntk_lstm_step_bwd: null pointer");
    NTK_REQUIRE(B > 0 && hid > 0, NTK_ERR_BAD_SHAPE, "ntk_lstm_step_bwd: B=%d hid=%d", B, hid);
    lstm_step_bwd_kernel<<<(B * hid + 255) / 256, 256, 0, (hipStream_t)stream>>>(act, c_prev, c, dh, dc, dpre, dc_prev, B, hid);
    NTK_CHECK_LAUNCH("ntk_lstm_step_bwd");
    return NTK_OK;
}

extern "C" int ntk_maxpool2x2(const float* in, float* out, int n, int H, int W, int C, void* stream) {
    NTK_REQUIRE(in && out, NTK_ERR_BAD_PTR, "ntk_maxpool2x2: null pointer");
    NTK_REQUIRE(n > 0 && H >= 2 && W >= 2 && (H % 2) == 0 && (W % 2) == 0 && C > 0 && (C % 4) == 0, NTK_ERR_BAD_SHAPE,
                "ntk_maxpool2x2: n=%d H=%d W=%d C=%d (H, W even; C multiple of 4)", n, H, W, C);
    NTK_REQUIRE(ntk_aligned16(in) && ntk_aligned16(out), NTK_ERR_BAD_PTR, "ntk_maxpool2x2: 16-byte alignment");
    const size_t tot = (size_t)n * (H / 2) * (W / 2) * (C / 4);
    const int blocks = (int)((tot + 255) / 256 < 16384 ? (tot + 255) / 256 : 16384);
    maxpool2x2_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(in, out, n, H, W, C);
    NTK_CHECK_LAUNCH("ntk_maxpool2x2");
    return NTK_OK;
}

// loss split in two calls (direct_offset_output.py:581-606): forward = pred + loss, backward = d loss / d logits
extern "C" int ntk_offset_loss_fwd(const float* logits, const float* offsets, float* pred, float* loss, int B, int T, int NF,
                                   int O, void* stream) {
    NTK_REQUIRE(loss, NTK_ERR_BAD_PTR, "ntk_offset_loss_fwd: null pointer");
    return ntk_offset_loss(logits, offsets, pred, loss, nullptr, B, T, NF, O, stream);
}
extern "C" int ntk_offset_loss_bwd(const float* logits, const float* offsets, float* dlogits, int B, int T, int NF, int O,
                                   void* stream) {
    NTK_REQUIRE(dlogits, NTK_ERR_BAD_PTR, "ntk_offset_loss_bwd: null pointer");
    return ntk_offset_loss(logits, offsets, nullptr, nullptr, dlogits, B, T, NF, O, stream);
}
