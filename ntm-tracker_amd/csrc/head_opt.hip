// Tracking head around the NTM cell, initial state, and the optimiser:
//   * 64-point gather from conv4_3 + input serialiser   (direct_offset_output.py:392-399, :439-500)
//   * output gather + tanh + l2 loss, and its gradient   (:581-606)
//   * trainable initial state tanh/sigmoid + its gradient (ntm_cell.py:284-315)
//   * clip_by_global_norm + TF RMSProp                    (direct_offset_output.py:620-626)
// All HBM-bound elementwise / gather work: coalesced 16-byte accesses, no LDS.
#include "common.h"

namespace {

// one 128-thread workgroup per serialised row (b, t, i), i in [0, NF]; row i == NF is the delimiter
__global__ void gather_serialize_kernel(const float* __restrict__ fmap, const float* __restrict__ gts0,
                                        float* __restrict__ X, int T, int Hf, int Wf, int C, int ldx,
                                        int g0, int gstep, int gn, int delim_first) {
    const int NF = gn * gn;
    const int row = blockIdx.x;               // (b*T + t)*(NF+1) + position
    const int pos = row % (NF + 1);
    // training order: 64 feature rows then the delimiter (direct_offset_output.py:480-481);
    // inference order (quirk Q8): the delimiter row FIRST (test_tracker.py:400-404)
    const int i = delim_first ? (pos == 0 ? NF : pos - 1) : pos;
    const int ft = row / (NF + 1);            // b*T + t
    const int t = ft % T, b = ft / T;
    float* xr = X + (size_t)row * ldx;
    const int C4 = C >> 2;
    if (i < NF) {
        const int y = g0 + (i / gn) * gstep, x = g0 + (i % gn) * gstep;
        const f32x4* src = reinterpret_cast<const f32x4*>(fmap + (((size_t)ft * Hf + y) * Wf + x) * C);
        f32x4* dst = reinterpret_cast<f32x4*>(xr);
        for (int c = threadIdx.x; c < C4; c += blockDim.x) dst[c] = src[c];
    } else {
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        f32x4* dst = reinterpret_cast<f32x4*>(xr);
        for (int c = threadIdx.x; c < C4; c += blockDim.x) dst[c] = z;
    }
    for (int c = C + threadIdx.x; c < ldx; c += blockDim.x) {
        float v = 0.f;
        if (c == C) v = (i == NF) ? 1.f : 0.f;                                    // frame delimiter bit
        else if (c == C + 1) v = (t == 0 && i < NF && gts0) ? gts0[(size_t)b * NF + i] : 0.f;  // target, frame 0 only
        xr[c] = v;
    }
}

// single workgroup: pred = tanh(logit at the delimiter step of frames 1..T-1), loss = 0.5*sum (pred-off)^2,
// dlogits = (pred-off)*(1-pred^2) at those steps and 0 elsewhere
__global__ __launch_bounds__(1024) void offset_loss_kernel(const float* __restrict__ logits,
                                                            const float* __restrict__ offsets,
                                                            float* __restrict__ pred, float* __restrict__ loss,
                                                            float* __restrict__ dlogits, int B, int T, int NF, int O) {
    __shared__ float red[16];
    const int S = T * (NF + 1);
    const int tid = threadIdx.x;
    if (dlogits) {
        const size_t tot = (size_t)B * S * O;
        for (size_t i = tid; i < tot; i += blockDim.x) dlogits[i] = 0.f;
    }
    __syncthreads();
    float acc = 0.f;
    const int cnt = B * (T - 1) * O;
    for (int i = tid; i < cnt; i += blockDim.x) {
        const int o = i % O;
        const int bt = i / O;
        const int t = bt % (T - 1) + 1, b = bt / (T - 1);
        const size_t li = ((size_t)b * S + (size_t)t * (NF + 1) + NF) * O + o;
        const float p = tanhf(logits[li]);
        const float dlt = p - offsets[((size_t)b * T + t) * O + o];
        acc += dlt * dlt;
        if (pred) pred[i] = p;
        if (dlogits) dlogits[li] = dlt * (1.0f - p * p);
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
        if (loss) *loss = 0.5f * s;
    }
}

// Sequential presentation of main.py's heat-map trackers (ntm_sevenbyseven, main.py:1701-1775; the same layout in
// :979-1291): every position of the feature map is a feature (F = Hf * Wf), rows are [feat(C), feature delimiter,
// frame delimiter, target]:
//   frame 0      : F rows [feat_i, 0, 0, gt0_i]
//   frame t >= 1 : one frame-delimiter row [0.., 0, 1, 0], then per feature [feat_i, 0, 0, 0] and [0.., 1, 0, 0]
// -> S = F + (T - 1) (2 F + 1) rows.  One 128-thread workgroup per row.
__global__ void serialize_sequential_kernel(const float* __restrict__ fmap, const float* __restrict__ gts0,
                                            float* __restrict__ X, int T, int F, int C, int ldx) {
    const int S = F + (T - 1) * (2 * F + 1);
    const int row = blockIdx.x, b = row / S, s = row - b * S;
    int t = 0, i = s, kind = 0;                 // kind 0: feature row, 1: feature delimiter, 2: frame delimiter
    if (s >= F) {
        const int r = s - F;
        t = 1 + r / (2 * F + 1);
        const int p = r % (2 * F + 1);
        if (p == 0) { kind = 2; i = 0; }
        else { i = (p - 1) >> 1; kind = ((p - 1) & 1) ? 1 : 0; }
    }
    float* xr = X + (size_t)row * ldx;
    const int C4 = C >> 2;
    f32x4* dst = reinterpret_cast<f32x4*>(xr);
    if (kind == 0) {
        const f32x4* src = reinterpret_cast<const f32x4*>(fmap + (((size_t)b * T + t) * F + i) * C);
        for (int c = threadIdx.x; c < C4; c += blockDim.x) dst[c] = src[c];
    } else {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int c = threadIdx.x; c < C4; c += blockDim.x) dst[c] = z;
    }
    for (int c = C + threadIdx.x; c < ldx; c += blockDim.x) {
        float v = 0.f;
        if (c == C) v = kind == 1 ? 1.f : 0.f;
        else if (c == C + 1) v = kind == 2 ? 1.f : 0.f;
        else if (c == C + 2) v = (t == 0 && gts0) ? gts0[(size_t)b * F + i] : 0.f;
        xr[c] = v;
    }
}

// Heat-map head of the sequential trackers (main.py:1880-1922): the logit (output_dim 1) at each FEATURE-DELIMITER step of
// frames 1..T-1 is one entry of that frame's F-way score vector; loss = sum_{b,t} softmax_cross_entropy(scores, gt[b,t]) /
// (T - 1).  One wave per (b, t >= 1): softmax over F, cross entropy against the (soft) labels, and the gradient
// (softmax * sum(labels) - labels) / (T - 1) scattered to the gathered steps (zero elsewhere; dlogits zeroed first).
__global__ __launch_bounds__(1024) void heatmap_ce_loss_kernel(const float* __restrict__ logits, const float* __restrict__ gt,
                                                                float* __restrict__ probs, float* __restrict__ loss,
                                                                float* __restrict__ dlogits, int B, int T, int F) {
    __shared__ float red[16];
    const int S = F + (T - 1) * (2 * F + 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    if (dlogits) {
        for (size_t i = tid; i < (size_t)B * S; i += blockDim.x) dlogits[i] = 0.f;
    }
    __syncthreads();
    float acc = 0.f;
    const float inv = 1.0f / (float)(T - 1);
    for (int bt = wave; bt < B * (T - 1); bt += nw) {
        const int b = bt / (T - 1), t = bt % (T - 1) + 1;
        const size_t base = (size_t)b * S + F + (size_t)(t - 1) * (2 * F + 1) + 1;      // first feature row of frame t
        const float* lab = gt + ((size_t)b * (T - 1) + (t - 1)) * F;
        float mx = -INFINITY;
        for (int i = lane; i < F; i += 64) mx = fmaxf(mx, logits[base + 2 * i + 1]);
        mx = wave_max(mx);
        float se = 0.f, sl = 0.f, dotl = 0.f;
        for (int i = lane; i < F; i += 64) {
            const float z = logits[base + 2 * i + 1] - mx, y = lab[i];
            se += expf(z); sl += y; dotl += y * z;
        }
        se = wave_sum(se); sl = wave_sum(sl); dotl = wave_sum(dotl);
        const float lse = logf(se);
        if (lane == 0) acc += sl * lse - dotl;                   // -sum y (z - lse)
        for (int i = lane; i < F; i += 64) {
            const float p = expf(logits[base + 2 * i + 1] - mx - lse);
            if (probs) probs[((size_t)b * (T - 1) + (t - 1)) * F + i] = p;
            if (dlogits) dlogits[base + 2 * i + 1] = (p * sl - lab[i]) * inv;
        }
    }
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int w = 0; w < nw; ++w) s += red[w];
        if (loss) *loss = s * inv;
    }
}

// Two-step presentation of main.py's ntm_two_step (:862-977) / ntm_tracker_new.NTMTracker(two_step=True) (:112-195): a whole
// frame is ONE step.  Rows [switch, feat(D), target(F)]: step 0 = [0, feat_0, target]; frame t >= 1 = [0, feat_t, 0] then the
// query step [1, 0, 0]  ->  S = 2 T - 1 steps.  feat [B, T, D] (D a multiple of 4 is NOT required), target [B, F].
__global__ void serialize_two_step_kernel(const float* __restrict__ feat, const float* __restrict__ target, float* __restrict__ X,
                                          int B, int T, int D, int F, int ldx) {
    const int S = 2 * T - 1;
    const size_t total = (size_t)B * S * ldx;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % ldx);
        const size_t r = idx / ldx;
        const int s = (int)(r % S), b = (int)(r / S);
        const bool query = s > 0 && (s & 1) == 0;               // steps 2, 4, ...: "ask for output"
        const int t = (s + 1) >> 1;                              // frame shown at steps 0, 1, 3, 5, ...
        float v = 0.f;
        if (c == 0) v = query ? 1.f : 0.f;
        else if (c <= D) v = query ? 0.f : feat[((size_t)b * T + t) * D + (c - 1)];
        else if (c <= D + F) v = (s == 0 && target) ? target[(size_t)b * F + (c - 1 - D)] : 0.f;
        X[idx] = v;
    }
}

// Loss of ntm_two_step (main.py:903-951): labels [B, 2T-1, F+1] = background row [0..0,1] at step 0 and at every
// presentation step, [gt_t, 0] at the query step of frame t >= 1; the labels pass through tf.nn.softmax before the cross
// entropy (as coded); loss = sum_rows CE(logits_row, softmax(label_row)) / ((2T-1) B).  One wave per row.
__global__ __launch_bounds__(1024) void two_step_ce_loss_kernel(const float* __restrict__ logits, const float* __restrict__ gt,
                                                                 float* __restrict__ probs, float* __restrict__ loss,
                                                                 float* __restrict__ dlogits, int B, int T, int F) {
    __shared__ float red[16];
    const int S = 2 * T - 1, K = F + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const float inv = 1.0f / ((float)S * (float)B);
    float acc = 0.f;
    for (int r = wave; r < B * S; r += nw) {
        const int b = r / S, s = r - b * S;
        const bool query = s > 0 && (s & 1) == 0;
        const float* g = query ? gt + ((size_t)b * T + (s >> 1)) * F : nullptr;
        const float* z = logits + (size_t)r * K;
        // softmax of the label row (values 0/1, so no max subtraction is needed; done anyway for arbitrary heat-maps)
        float lmx = query ? 0.f : 1.f;
        if (query) { for (int i = lane; i < F; i += 64) lmx = fmaxf(lmx, g[i]); lmx = wave_max(lmx); }
        float lse_l = 0.f, mx = -INFINITY;
        for (int i = lane; i < K; i += 64) {
            const float y = (i < F) ? (query ? g[i] : 0.f) : (query ? 0.f : 1.f);
            lse_l += expf(y - lmx);
            mx = fmaxf(mx, z[i]);
        }
        lse_l = wave_sum(lse_l); mx = wave_max(mx);
        float se = 0.f;
        for (int i = lane; i < K; i += 64) se += expf(z[i] - mx);
        se = wave_sum(se);
        const float lse = logf(se);
        float ce = 0.f;
        for (int i = lane; i < K; i += 64) {
            const float y = (i < F) ? (query ? g[i] : 0.f) : (query ? 0.f : 1.f);
            const float q = expf(y - lmx) / lse_l;               // softmax(labels)
            const float lp = z[i] - mx - lse;
            ce -= q * lp;
            const float p = expf(lp);
            if (probs) probs[(size_t)r * K + i] = p;
            if (dlogits) dlogits[(size_t)r * K + i] = (p - q) * inv;
        }
        ce = wave_sum(ce);
        if (lane == 0) acc += ce;
    }
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid == 0) {
        float sacc = 0.f;
        for (int w = 0; w < nw; ++w) sacc += red[w];
        if (loss) *loss = sacc * inv;
    }
}

// copy task head (main.py:1603-1610): p = sigmoid(logit); loss = mean(-(y log(p+eps) + (1-y) log(1-p+eps))), eps = 1e-7
// (tf.losses.log_loss defaults); dlogits = d loss / d logit.  Single workgroup, fixed-order reduction.
__global__ __launch_bounds__(1024) void log_loss_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                                         float* __restrict__ loss, float* __restrict__ dlogits, int n) {
    __shared__ float red[16];
    const float eps = 1e-7f, inv = 1.0f / (float)n;
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float p = 1.0f / (1.0f + expf(-logits[i]));
        const float y = labels[i];
        acc += -(y * logf(p + eps) + (1.0f - y) * logf(1.0f - p + eps));
        if (dlogits) dlogits[i] = inv * (-(y / (p + eps)) + (1.0f - y) / (1.0f - p + eps)) * p * (1.0f - p);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
        *loss = s * inv;
    }
}

// act: 0 = tanh, 1 = sigmoid.  out[b][i] = act(v[i])
__global__ void init_state_kernel(const float* __restrict__ v, float* __restrict__ out, int n, int B, int act) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = v[i];
    const float y = act ? 1.0f / (1.0f + expf(-x)) : tanhf(x);
    for (int b = 0; b < B; ++b) out[(size_t)b * n + i] = y;
}

// dv[i] (+)= act'(v[i]) * sum_b dout[b][i]
__global__ void init_state_bwd_kernel(const float* __restrict__ v, const float* __restrict__ dout,
                                      float* __restrict__ dv, int n, int B, int act, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dout[(size_t)b * n + i];
    const float x = v[i];
    float dy;
    if (act) { const float y = 1.0f / (1.0f + expf(-x)); dy = y * (1.0f - y); }
    else { const float y = tanhf(x); dy = 1.0f - y * y; }
    const float g = s * dy;
    dv[i] = accumulate ? dv[i] + g : g;
}

constexpr int SUMSQ_BLOCK = 256;
constexpr int SUMSQ_PER_BLOCK = 4096;

__global__ void sumsq_partial_kernel(const float* __restrict__ g, float* __restrict__ partial, size_t n) {
    __shared__ float red[SUMSQ_BLOCK / 64];
    const size_t base = (size_t)blockIdx.x * SUMSQ_PER_BLOCK;
    float acc = 0.f;
    for (int i = threadIdx.x; i < SUMSQ_PER_BLOCK; i += SUMSQ_BLOCK) {
        const size_t idx = base + i;
        if (idx < n) { const float v = g[idx]; acc += v * v; }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int w = 0; w < SUMSQ_BLOCK / 64; ++w) s += red[w];
        partial[blockIdx.x] = s;
    }
}

__global__ void sumsq_final_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ gnorm) {
    __shared__ float red[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) acc += partial[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
        *gnorm = sqrtf(s);
    }
}

// tf.clip_by_global_norm then tf.train.RMSPropOptimizer (ms slot starts at ONE, eps inside the sqrt)
__global__ void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ ms,
                               float* __restrict__ mom, size_t n, float lr, float decay, float momentum,
                               float eps, float clip, const float* __restrict__ gnorm) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float scale = 1.0f;
    if (clip > 0.f) scale = clip / fmaxf(*gnorm, clip);
    const float gi = g[i] * scale;
    const float m2 = decay * ms[i] + (1.0f - decay) * gi * gi;
    const float mo = momentum * mom[i] + lr * gi / sqrtf(m2 + eps);
    ms[i] = m2;
    mom[i] = mo;
    p[i] -= mo;
}

// The same update behind a finiteness check of the global norm (every thread reads the same word, so the whole grid takes
// the same branch): a NaN / Inf norm -- an aborted cluster launch poisons the gradient, csrc/dnc_cluster_fwd.hip, or a
// genuine overflow -- leaves parameters and slots untouched, turns *loss into NaN and counts the skipped step.
__global__ void rmsprop_checked_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ ms,
                                       float* __restrict__ mom, size_t n, float lr, float decay, float momentum,
                                       float eps, float clip, const float* __restrict__ gnorm, float* __restrict__ loss,
                                       unsigned* __restrict__ skipped) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float gn = *gnorm;
    if (!(fabsf(gn) <= 3.402823466e38f)) {                    // NaN or Inf
        if (i == 0) {
            if (loss) loss[0] = __int_as_float(0x7fc00000);
            if (skipped) *skipped += 1u;
        }
        return;
    }
    if (i >= n) return;
    float scale = 1.0f;
    if (clip > 0.f) scale = clip / fmaxf(gn, clip);
    const float gi = g[i] * scale;
    const float m2 = decay * ms[i] + (1.0f - decay) * gi * gi;
    const float mo = momentum * mom[i] + lr * gi / sqrtf(m2 + eps);
    ms[i] = m2;
    mom[i] = mo;
    p[i] -= mo;
}

}  // namespace

static int gather_serialize_impl(const float* fmap, const float* gts0, float* X, int B, int T,
                                 int Hf, int Wf, int C, int ldx, int grid_start, int grid_step,
                                 int grid_n, int delim_first, void* stream) {
    NTK_REQUIRE(fmap && X, NTK_ERR_BAD_PTR, "ntk_gather_serialize: null pointer");
    NTK_REQUIRE(ntk_aligned16(fmap) && ntk_aligned16(X), NTK_ERR_BAD_PTR, "ntk_gather_serialize: 16-byte alignment");
    NTK_REQUIRE(B > 0 && T > 0 && C > 0 && (C % 4) == 0 && ldx >= C + 2 && (ldx % 4) == 0 && grid_n > 0 &&
                    grid_start >= 0 && grid_step > 0 && grid_start + (grid_n - 1) * grid_step < Hf &&
                    grid_start + (grid_n - 1) * grid_step < Wf,
                NTK_ERR_BAD_SHAPE, "ntk_gather_serialize: B=%d T=%d C=%d ldx=%d grid=(%d,%d,%d) map=%dx%d", B, T, C,
                ldx, grid_start, grid_step, grid_n, Hf, Wf);
    const long rows = (long)B * T * (grid_n * grid_n + 1);
    NTK_REQUIRE(rows < 2147483647L, NTK_ERR_BAD_SHAPE, "ntk_gather_serialize: too many rows");
    gather_serialize_kernel<<<(unsigned)rows, 128, 0, (hipStream_t)stream>>>(fmap, gts0, X, T, Hf, Wf, C, ldx,
                                                                            grid_start, grid_step, grid_n, delim_first);
    NTK_CHECK_LAUNCH("ntk_gather_serialize");
    return NTK_OK;
}

extern "C" int ntk_gather_serialize(const float* fmap, const float* gts0, float* X, int B, int T,
                                    int Hf, int Wf, int C, int ldx, int grid_start, int grid_step,
                                    int grid_n, void* stream) {
    return gather_serialize_impl(fmap, gts0, X, B, T, Hf, Wf, C, ldx, grid_start, grid_step, grid_n, 0, stream);
}

extern "C" int ntk_gather_serialize_online(const float* fmap, const float* gts0, float* X, int B, int T,
                                           int Hf, int Wf, int C, int ldx, int grid_start, int grid_step,
                                           int grid_n, void* stream) {
    return gather_serialize_impl(fmap, gts0, X, B, T, Hf, Wf, C, ldx, grid_start, grid_step, grid_n, 1, stream);
}

// tf.image.crop_and_resize (bilinear, one box) of (image - mean): out[y][x][c], extrapolation outside the image
__global__ void crop_resize_kernel(const float* __restrict__ img, int H, int W, int C, const float* __restrict__ mean,
                                   float y1, float x1, float y2, float x2, float* __restrict__ out, int ch, int cw,
                                   float extrapolation) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ch * cw * C) return;
    const int c = idx % C, x = (idx / C) % cw, y = idx / (C * cw);
    const float hs = (ch > 1) ? (y2 - y1) * (float)(H - 1) / (float)(ch - 1) : 0.f;
    const float wsc = (cw > 1) ? (x2 - x1) * (float)(W - 1) / (float)(cw - 1) : 0.f;
    const float in_y = (ch > 1) ? y1 * (float)(H - 1) + (float)y * hs : 0.5f * (y1 + y2) * (float)(H - 1);
    const float in_x = (cw > 1) ? x1 * (float)(W - 1) + (float)x * wsc : 0.5f * (x1 + x2) * (float)(W - 1);
    float v = extrapolation;
    if (in_y >= 0.f && in_y <= (float)(H - 1) && in_x >= 0.f && in_x <= (float)(W - 1)) {
        const int ty = (int)floorf(in_y), by = (int)ceilf(in_y);
        const int lx = (int)floorf(in_x), rx = (int)ceilf(in_x);
        const float yl = in_y - (float)ty, xl = in_x - (float)lx;
        const float m = mean ? mean[c] : 0.f;
        const float tl = img[((size_t)ty * W + lx) * C + c] - m, tr = img[((size_t)ty * W + rx) * C + c] - m;
        const float bl = img[((size_t)by * W + lx) * C + c] - m, br = img[((size_t)by * W + rx) * C + c] - m;
        const float top = tl + (tr - tl) * xl, bot = bl + (br - bl) * xl;
        v = top + (bot - top) * yl;
    }
    out[idx] = v;
}

// tf.image.resize_images(..., BILINEAR) with TF-1 defaults (align_corners=False, no half-pixel centres):
// src = dst * (in / out); neighbours floor(src) and min(floor(src)+1, in-1)
__global__ void resize_bilinear_kernel(const float* __restrict__ img, int H, int W, int C, float* __restrict__ out,
                                       int OH, int OW) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)OH * OW * C) return;
    const int c = (int)(idx % C), x = (int)((idx / C) % OW), y = (int)(idx / ((size_t)C * OW));
    const float sy = (float)y * ((float)H / (float)OH), sx = (float)x * ((float)W / (float)OW);
    const int y0 = (int)floorf(sy), x0 = (int)floorf(sx);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float fy = sy - (float)y0, fx = sx - (float)x0;
    const float tl = img[((size_t)y0 * W + x0) * C + c], tr = img[((size_t)y0 * W + x1) * C + c];
    const float bl = img[((size_t)y1 * W + x0) * C + c], br = img[((size_t)y1 * W + x1) * C + c];
    const float top = tl + (tr - tl) * fx, bot = bl + (br - bl) * fx;
    out[idx] = top + (bot - top) * fy;
}

extern "C" int ntk_resize_bilinear(const float* image, int H, int W, int C, float* out, int out_h, int out_w, void* stream) {
    NTK_REQUIRE(image && out, NTK_ERR_BAD_PTR, "ntk_resize_bilinear: null pointer");
    NTK_REQUIRE(H > 0 && W > 0 && C > 0 && out_h > 0 && out_w > 0, NTK_ERR_BAD_SHAPE, "ntk_resize_bilinear: %dx%dx%d -> %dx%d", H, W, C, out_h, out_w);
    const size_t total = (size_t)out_h * out_w * C;
    resize_bilinear_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(image, H, W, C, out, out_h, out_w);
    NTK_CHECK_LAUNCH("ntk_resize_bilinear");
    return NTK_OK;
}

extern "C" int ntk_crop_and_resize(const float* image, int H, int W, int C, const float* mean, float y1, float x1,
                                   float y2, float x2, float* out, int crop_h, int crop_w, float extrapolation,
                                   void* stream) {
    NTK_REQUIRE(image && out, NTK_ERR_BAD_PTR, "ntk_crop_and_resize: null pointer");
    NTK_REQUIRE(H > 0 && W > 0 && C > 0 && crop_h > 0 && crop_w > 0, NTK_ERR_BAD_SHAPE,
                "ntk_crop_and_resize: H=%d W=%d C=%d crop=%dx%d", H, W, C, crop_h, crop_w);
    const int total = crop_h * crop_w * C;
    crop_resize_kernel<<<(total + 255) / 256, 256, 0, (hipStream_t)stream>>>(image, H, W, C, mean, y1, x1, y2, x2, out,
                                                                             crop_h, crop_w, extrapolation);
    NTK_CHECK_LAUNCH("ntk_crop_and_resize");
    return NTK_OK;
}

extern "C" int ntk_offset_loss(const float* logits, const float* offsets, float* pred, float* loss,
                               float* dlogits, int B, int T, int NF, int O, void* stream) {
    NTK_REQUIRE(logits && offsets && (loss || dlogits), NTK_ERR_BAD_PTR, "ntk_offset_loss: null pointer");
    NTK_REQUIRE(B > 0 && T >= 2 && NF > 0 && O > 0, NTK_ERR_BAD_SHAPE, "ntk_offset_loss: B=%d T=%d NF=%d O=%d (T >= 2)", B, T, NF, O);
    offset_loss_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(logits, offsets, pred, loss, dlogits, B, T, NF, O);
    NTK_CHECK_LAUNCH("ntk_offset_loss");
    return NTK_OK;
}

extern "C" int ntk_log_loss(const float* logits, const float* labels, float* loss, float* dlogits, int n, void* stream) {
    NTK_REQUIRE(logits && labels && loss, NTK_ERR_BAD_PTR, "ntk_log_loss: null pointer");
    NTK_REQUIRE(n > 0, NTK_ERR_BAD_SHAPE, "ntk_log_loss: n=%d", n);
    log_loss_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(logits, labels, loss, dlogits, n);
    NTK_CHECK_LAUNCH("ntk_log_loss");
    return NTK_OK;
}

extern "C" int ntk_serialize_sequential(const float* fmap, const float* gts0, float* X, int B, int T, int F, int C, int ldx,
                                        void* stream) {
    NTK_REQUIRE(fmap && X, NTK_ERR_BAD_PTR, "ntk_serialize_sequential: null pointer");
    NTK_REQUIRE(B > 0 && T > 0 && F > 0 && C > 0 && (C % 4) == 0 && ldx >= C + 3 && (ldx % 4) == 0, NTK_ERR_BAD_SHAPE,
                "ntk_serialize_sequential: B=%d T=%d F=%d C=%d (multiple of 4) ldx=%d (>= C + 3, multiple of 4)", B, T, F, C, ldx);
    NTK_REQUIRE(ntk_aligned16(fmap) && ntk_aligned16(X), NTK_ERR_BAD_PTR, "ntk_serialize_sequential: 16-byte alignment");
    const long rows = (long)B * (F + (long)(T - 1) * (2 * F + 1));
    NTK_REQUIRE(rows < 2147483647L, NTK_ERR_BAD_SHAPE, "ntk_serialize_sequential: too many rows");
    serialize_sequential_kernel<<<(unsigned)rows, 128, 0, (hipStream_t)stream>>>(fmap, gts0, X, T, F, C, ldx);
    NTK_CHECK_LAUNCH("ntk_serialize_sequential");
    return NTK_OK;
}

extern "C" int ntk_heatmap_ce_loss(const float* logits, const float* gt, float* probs, float* loss, float* dlogits,
                                   int B, int T, int F, void* stream) {
    NTK_REQUIRE(logits && gt && (loss || dlogits || probs), NTK_ERR_BAD_PTR, "ntk_heatmap_ce_loss: null pointer");
    NTK_REQUIRE(B > 0 && T >= 2 && F > 0, NTK_ERR_BAD_SHAPE, "ntk_heatmap_ce_loss: B=%d T=%d (>= 2) F=%d", B, T, F);
    heatmap_ce_loss_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(logits, gt, probs, loss, dlogits, B, T, F);
    NTK_CHECK_LAUNCH("ntk_heatmap_ce_loss");
    return NTK_OK;
}

extern "C" int ntk_serialize_two_step(const float* feat, const float* target, float* X, int B, int T, int D, int F, int ldx,
                                      void* stream) {
    NTK_REQUIRE(feat && X, NTK_ERR_BAD_PTR, "ntk_serialize_two_step: null pointer");
    NTK_REQUIRE(B > 0 && T >= 1 && D > 0 && F > 0 && ldx >= 1 + D + F, NTK_ERR_BAD_SHAPE,
                "ntk_serialize_two_step: B=%d T=%d D=%d F=%d ldx=%d (>= 1 + D + F)", B, T, D, F, ldx);
    const size_t total = (size_t)B * (2 * T - 1) * ldx;
    const unsigned nb = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    serialize_two_step_kernel<<<nb, 256, 0, (hipStream_t)stream>>>(feat, target, X, B, T, D, F, ldx);
    NTK_CHECK_LAUNCH("ntk_serialize_two_step");
    return NTK_OK;
}

extern "C" int ntk_two_step_ce_loss(const float* logits, const float* gt, float* probs, float* loss, float* dlogits,
                                    int B, int T, int F, void* stream) {
    NTK_REQUIRE(logits && gt && (loss || dlogits || probs), NTK_ERR_BAD_PTR, "ntk_two_step_ce_loss: null pointer");
    NTK_REQUIRE(B > 0 && T >= 1 && F > 0, NTK_ERR_BAD_SHAPE, "ntk_two_step_ce_loss: B=%d T=%d F=%d", B, T, F);
    two_step_ce_loss_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(logits, gt, probs, loss, dlogits, B, T, F);
    NTK_CHECK_LAUNCH("ntk_two_step_ce_loss");
    return NTK_OK;
}

extern "C" int ntk_ntm_init_state(const float* v, float* out, int n, int B, int act, void* stream) {
    NTK_REQUIRE(v && out, NTK_ERR_BAD_PTR, "ntk_ntm_init_state: null pointer");
    NTK_REQUIRE(n > 0 && B > 0 && (act == 0 || act == 1), NTK_ERR_BAD_SHAPE, "ntk_ntm_init_state: n=%d B=%d act=%d", n, B, act);
    init_state_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(v, out, n, B, act);
    NTK_CHECK_LAUNCH("ntk_ntm_init_state");
    return NTK_OK;
}

extern "C" int ntk_ntm_init_state_bwd(const float* v, const float* dout, float* dv, int n, int B, int act,
                                      int accumulate, void* stream) {
    NTK_REQUIRE(v && dout && dv, NTK_ERR_BAD_PTR, "ntk_ntm_init_state_bwd: null pointer");
    NTK_REQUIRE(n > 0 && B > 0 && (act == 0 || act == 1), NTK_ERR_BAD_SHAPE, "ntk_ntm_init_state_bwd: n=%d B=%d act=%d", n, B, act);
    init_state_bwd_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(v, dout, dv, n, B, act, accumulate);
    NTK_CHECK_LAUNCH("ntk_ntm_init_state_bwd");
    return NTK_OK;
}

extern "C" size_t ntk_global_norm_workspace_bytes(size_t n) {
    return ((n + SUMSQ_PER_BLOCK - 1) / SUMSQ_PER_BLOCK) * sizeof(float);
}

extern "C" int ntk_global_norm(const float* grads, size_t n, float* workspace, float* gnorm, void* stream) {
    NTK_REQUIRE(grads && workspace && gnorm, NTK_ERR_BAD_PTR, "ntk_global_norm: null pointer");
    NTK_REQUIRE(n > 0, NTK_ERR_BAD_SHAPE, "ntk_global_norm: n=0");
    const int nb = (int)((n + SUMSQ_PER_BLOCK - 1) / SUMSQ_PER_BLOCK);
    sumsq_partial_kernel<<<nb, SUMSQ_BLOCK, 0, (hipStream_t)stream>>>(grads, workspace, n);
    NTK_CHECK_LAUNCH("ntk_global_norm(partial)");
    sumsq_final_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(workspace, nb, gnorm);
    NTK_CHECK_LAUNCH("ntk_global_norm(final)");
    return NTK_OK;
}

extern "C" int ntk_rmsprop_clip_step(float* params, const float* grads, float* ms, float* mom, size_t n,
                                     float lr, float decay, float momentum, float eps, float clip_norm,
                                     const float* gnorm, void* stream) {
    NTK_REQUIRE(params && grads && ms && mom, NTK_ERR_BAD_PTR, "ntk_rmsprop_clip_step: null pointer");
    NTK_REQUIRE(n > 0, NTK_ERR_BAD_SHAPE, "ntk_rmsprop_clip_step: n=0");
    NTK_REQUIRE(clip_norm <= 0.f || gnorm, NTK_ERR_BAD_PTR, "ntk_rmsprop_clip_step: clip_norm > 0 needs gnorm");
    const unsigned nb = (unsigned)((n + 255) / 256);
    rmsprop_kernel<<<nb, 256, 0, (hipStream_t)stream>>>(params, grads, ms, mom, n, lr, decay, momentum, eps, clip_norm, gnorm);
    NTK_CHECK_LAUNCH("ntk_rmsprop_clip_step");
    return NTK_OK;
}

extern "C" int ntk_rmsprop_clip_step_checked(float* params, const float* grads, float* ms, float* mom, size_t n,
                                             float lr, float decay, float momentum, float eps, float clip_norm,
                                             const float* gnorm, float* loss, unsigned* skipped, void* stream) {
    NTK_REQUIRE(params && grads && ms && mom && gnorm, NTK_ERR_BAD_PTR, "ntk_rmsprop_clip_step_checked: null pointer");
    NTK_REQUIRE(n > 0, NTK_ERR_BAD_SHAPE, "ntk_rmsprop_clip_step_checked: n=0");
    const unsigned nb = (unsigned)((n + 255) / 256);
    rmsprop_checked_kernel<<<nb, 256, 0, (hipStream_t)stream>>>(params, grads, ms, mom, n, lr, decay, momentum, eps, clip_norm,
                                                                gnorm, loss, skipped);
    NTK_CHECK_LAUNCH("ntk_rmsprop_clip_step_checked");
    return NTK_OK;
}
