// Shared definitions of the NTM sequence kernels (forward and BPTT).
//
// Packed ("kernel") parameter layouts -- the master copy of the trainable
// parameters lives in these layouts; gradients are produced in the same
// layouts, so the optimiser is a flat elementwise pass:
//   WxT  [4*hid][ldx]      input part of the BasicLSTMCell kernel, transposed
//                          (row n' = unit*4 + gate, gate order i,j,f,o;
//                          column = input feature, zero padded to ldx)
//   Wr   [ldz][4*hid]      recurrent part: rows 0..R*Md-1 multiply read_prev,
//                          rows R*Md..K-1 multiply h_prev, row K = LSTM bias,
//                          rows K+1..ldz-1 zero; columns n' = unit*4 + gate
//   Wa   [ldh][PP]         [addressing/weights | output/weights] (columns
//                          0..P-1 controls, P..P+O-1 logits, zero pad to PP),
//                          row hid = the two bias vectors, rows > hid zero
// Reference: ntm_cell.py:45-50,101-105 (controller), :113-130 (unpack),
// :220 (output linear).
#pragma once
#include "common.h"

// shift taps the sequence kernels hold in registers: shift_range <= 4 (the reference's default and every BASELINE config use 1)
constexpr int NTM_MAX_SHIFT_TAPS = 9;

struct NtmDims {
    int B, S;          // sequences, steps
    int N, Md;         // memory slots, word size          (mem_size, mem_dim)
    int R, Wh, H;      // read heads, write heads, R+Wh
    int hid;           // controller_hidden_size (single layer)
    int SS;            // shift space 2*shift_range+1
    int O;             // output_dim
    int P, PP;         // control width, padded P+O (multiple of 4)
    int K, ldz;        // R*Md+hid, padded K+1 (multiple of 4)
    int ldh;           // padded hid+1 (multiple of 4)
    int write_first;
    // control offsets inside the unpacked vector (ntm_cell.py:128-130)
    int oK, oB, oG, oS, oY, oE, oA;   // k, beta, g, shift, gamma, erase, add
};

static inline void ntm_fill_dims(NtmDims& d, int B, int S, int N, int Md, int R, int Wh, int hid,
                                 int shift_range, int O, int write_first) {
    d.B = B; d.S = S; d.N = N; d.Md = Md; d.R = R; d.Wh = Wh; d.H = R + Wh; d.hid = hid;
    d.SS = 2 * shift_range + 1; d.O = O;
    d.oK = 0;
    d.oB = d.oK + d.H * Md;
    d.oG = d.oB + d.H;
    d.oS = d.oG + d.H;
    d.oY = d.oS + d.H * d.SS;
    d.oE = d.oY + d.H;
    d.oA = d.oE + Wh * Md;
    d.P = d.oA + Wh * Md;
    d.PP = ((d.P + O + 3) / 4) * 4;
    d.K = R * Md + hid;
    d.ldz = ((d.K + 1 + 3) / 4) * 4;
    d.ldh = ((hid + 1 + 3) / 4) * 4;
    d.write_first = write_first;
}

// LDS carve-up shared by forward and backward (offsets in floats)
struct NtmLds {
    int part, M, W, Wg, Z, C, U, Ks, Cn, Pw, total;
};

static inline __host__ __device__ int ntm_imax(int a, int b) { return a > b ? a : b; }
static inline __host__ __device__ int ntm_imin(int a, int b) { return a < b ? a : b; }
static inline __host__ __device__ int ntm_align4(int x) { return (x + 3) & ~3; }

// Elementwise math of the recurrence.  NTM_FAST_MATH (default) maps exp/log/pow/division onto the hardware
// transcendental units (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1-2 ulp): ~10x fewer instructions on the serial
// critical path of a step; tests/test_ntm_gpu.py::test_full_length_sequence_drift bounds the accumulated
// effect over 1300 steps against the float64 oracle (north_star tolerance 1e-4).
#ifndef NTM_FAST_MATH
#define NTM_FAST_MATH 1
#endif
#if NTM_FAST_MATH
__device__ __forceinline__ float ntm_exp(float x) { return __expf(x); }
__device__ __forceinline__ float ntm_sigmoid(float x) { return __fdividef(1.0f, 1.0f + __expf(-x)); }
__device__ __forceinline__ float ntm_tanh(float x) { return 1.0f - __fdividef(2.0f, 1.0f + __expf(2.0f * x)); }
__device__ __forceinline__ float ntm_softplus(float x) { return fmaxf(x, 0.f) + __logf(1.0f + __expf(-fabsf(x))); }
__device__ __forceinline__ float ntm_pow(float x, float y) { return x > 0.f ? __expf(y * __logf(x)) : 0.f; }
#else
__device__ __forceinline__ float ntm_exp(float x) { return expf(x); }
__device__ __forceinline__ float ntm_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float ntm_tanh(float x) { return tanhf(x); }
__device__ __forceinline__ float ntm_softplus(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float ntm_pow(float x, float y) { return powf(x, y); }
#endif
