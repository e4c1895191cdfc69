// Shared definitions of the DNC sequence kernels (forward and BPTT).
// Packed parameter layouts: see include/ntmtrack.h (ntk_dnc_seq_fwd).
#pragma once
#include "common.h"

struct DncDims {
    int B, S, N, W, R, Wn, hid, O;
    int I, IP;       // interface width, padded
    int K, ldz;      // R*W + hid, padded K+1
    int ldh;         // padded hid+1
    int Ky, ldy, OP; // hid + R*W, padded Ky+1, padded O
    float clip;
    // interface offsets
    int oV, oE, oF, oAg, oWg, oRm, oKw, oBw, oKr, oBr;
};

static void dnc_fill_dims(DncDims& d, int B, int S, int N, int W, int R, int Wn, int hid, int O, float clip) {
    d.B = B; d.S = S; d.N = N; d.W = W; d.R = R; d.Wn = Wn; d.hid = hid; d.O = O; d.clip = clip;
    d.oV = 0;
    d.oE = d.oV + Wn * W;
    d.oF = d.oE + Wn * W;
    d.oAg = d.oF + R;
    d.oWg = d.oAg + Wn;
    d.oRm = d.oWg + Wn;
    d.oKw = d.oRm + R * (1 + 2 * Wn);
    d.oBw = d.oKw + Wn * W;
    d.oKr = d.oBw + Wn;
    d.oBr = d.oKr + R * W;
    d.I = d.oBr + R;
    d.IP = (d.I + 3) & ~3;
    d.K = R * W + hid;
    d.ldz = (d.K + 1 + 3) & ~3;
    d.ldh = (hid + 1 + 3) & ~3;
    d.Ky = hid + R * W;
    d.ldy = (d.Ky + 1 + 3) & ~3;
    d.OP = (O + 3) & ~3;
}

constexpr int DT = 1024;      // threads per workgroup
constexpr int DW = DT / 64;   // waves

__device__ __forceinline__ float dnc_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float dnc_softplus(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float dnc_clip(float x, float c) { return c > 0.f ? fminf(fmaxf(x, -c), c) : x; }

// in-place softmax of H rows of length N held in LDS; wave w owns rows w, w+DW, ... (no block barrier inside)
__device__ __forceinline__ void lds_softmax_rows(float* v, int H, int N, int wave, int lane) {
    for (int h = wave; h < H; h += DW) {
        float* r = v + h * N;
        float mx = -INFINITY;
        for (int n = lane; n < N; n += 64) mx = fmaxf(mx, r[n]);
        mx = wave_max(mx);
        float s = 0.f;
        for (int n = lane; n < N; n += 64) { const float e = expf(r[n] - mx); r[n] = e; s += e; }
        s = wave_sum(s);
        for (int n = lane; n < N; n += 64) r[n] = r[n] / s;
    }
}

