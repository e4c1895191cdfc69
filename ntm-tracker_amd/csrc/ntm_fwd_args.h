// Arguments and LDS carve-up of the NTM forward sequence kernels (ntm_seq_fwd.hip, ntm_seq_fwd_ws.hip).
#pragma once
#include "ntm_common.h"

struct NtmFwdArgs {
    NtmDims d;
    // inputs
    const float* xproj;    // [B,S,4*hid]  X * Wx (columns n' = unit*4+gate), no bias
    const float* Wr;       // [ldz][4*hid]
    const float* Wa;       // [ldh][PP]
    const float* M0;       // [B,N,Md]
    const float* w0;       // [B,H,N]
    const float* read0;    // [B,R,Md]
    const float* cs0;      // [B,2*hid]  (c then h)
    // outputs
    float* logits;         // [B,S,O]
    float* outputs;        // [B,S,O] softmax(logits) or null
    float* M_out;          // [B,N,Md]
    float* w_out;          // [B,H,N]
    float* read_out;       // [B,R,Md]
    float* cs_out;         // [B,2*hid]
    // per-step records (all nullable): what LoopNTMTracker writes to its TensorArrays plus the BPTT stash
    float* st_z;           // [B,S,ldz]   step input [read_prev;h_prev;1;0..]
    float* st_gates;       // [B,S,4*hid] activated gates (i,j,f,o per unit)
    float* st_c;           // [B,S,hid]
    float* st_h;           // [B,S,ldh]   [h';1;0..]
    float* st_u;           // [B,S,PP]    activated controls, raw shift logits, raw output logits
    float* st_wc;          // [B,S,H,N]   content-focused weights
    float* st_wv;          // [B,S,H,N]   shifted weights (before sharpening)
    float* st_w;           // [B,S,H,N]
    float* st_M;           // [B,S,N,Md]
    float* st_read;        // [B,S,R,Md]
};

static inline void ntm_fwd_lds(const NtmDims& d, int T, NtmLds& L) {
    const int MP = d.Md | 1;
    const int nsl = ntm_imax(1, T / d.hid);
    const int ncg = d.PP / 4;
    const int nslB = ntm_imin(ntm_imax(1, T / ncg), d.hid);
    const int RM = d.R * d.Md;
    const int nslR = ntm_imin(ntm_imax(1, T / RM), d.N);
    int o = 0;
    L.part = o; o += ntm_align4(ntm_imax(ntm_imax(nsl * 4 * d.hid, nslB * d.PP), nslR * RM));
    L.M = o; o += ntm_align4(d.N * MP);
    L.W = o; o += ntm_align4(d.H * d.N);
    L.Wg = o; o += ntm_align4(d.H * d.N);
    L.Z = o; o += ntm_align4(d.K);
    L.C = o; o += ntm_align4(d.hid);
    L.U = o; o += ntm_align4(d.PP);
    L.Ks = o; o += ntm_align4(d.H * d.Md);
    L.Cn = o; o += ntm_align4(d.Md);
    L.Pw = o; o += ntm_align4(d.H * d.N);
    L.total = o;
}

