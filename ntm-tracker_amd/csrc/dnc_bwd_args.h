// Argument block shared by the DNC BPTT kernels of one workgroup per sequence (dnc_seq_bwd.hip: one write head, the
// benchmark path; dnc_seq_bwd_mw.hip: 1..4 write heads).
#pragma once
#include "dnc_common.h"

struct DncBwdArgs {
    DncDims d;
    const float* WrT; int ldkT;      // [4*hid][ldkT]
    const float* WiT; int ldhT;      // [IP][ldhT]
    const float* Wy;                 // [ldy][OP]
    const float* mem0; const float* link0; const float* usage0; const float* rw0; const float* ww0;
    const float* prec0; const float* hc0;
    const float* rec_gates; const float* rec_c; const float* rec_ifc; const float* rec_u; const float* rec_ww;
    const float* rec_rw; const float* rec_cw; const float* rec_cr; const float* rec_al; const float* rec_p;
    const float* rec_fwd; const float* rec_bwd; const float* rec_M; const float* rec_L; const float* rec_ypre;
    const float* dout;               // [B,S,O]
    float* gM; float* gL;            // [B,N,W], [B,Wn,N,N] zero-initialised scratch (carried gradients)
    float* dgates; float* dxi; float* dypre;
    float* gcarry; int carry_in;     // [B, (Wn+1)*N + R*N + ldkT + hid] gradients carried into state t=-1 (segmented BPTT)
};


// dnc_seq_bwd_mw.hip: the same BPTT for any number of write heads <= 4 (arguments already validated)
int dnc_seq_bwd_mw_launch(const DncBwdArgs& a, hipStream_t stream);
