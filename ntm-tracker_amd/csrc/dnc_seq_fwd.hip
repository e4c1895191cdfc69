// DNC core sequence forward (dnc/dnc.py:84-127 unrolled by tf.nn.dynamic_rnn,
// direct_offset_output_with_dnc.py:66-88): one persistent 1024-thread workgroup per
// sequence walks all S steps.  usage / read & write weights / precedence / reads /
// LSTM state live in LDS; the N x W memory and the Wn x N x N temporal link matrix stay
// in global memory (per sequence 64 KB + 256 KB at N=256,W=64; 256 KB + 1 MB at
// N=512,W=128 -- L2 / Infinity-Cache resident) and are updated in place.
//
// Per step:
//   P1  LSTM gates: [reads_prev ; h_prev] x Wr (+ hoisted x_t Wx), snt.LSTM forget_bias 1.0, clip  (dnc.py:105-113)
//   P2  interface  : h x Wi -> the ten linears of MemoryAccess._read_inputs                         (access.py:160-218)
//   P3  usage      : Freeness._usage_after_write/_after_read                                       (addressing.py:342-374)
//   P4  write content weights on M_{t-1}: one pass over M                                          (addressing.py:83-105, access.py:236-238)
//   P5  allocation : rank-based restatement of sort + exclusive cumprod + unsort                   (addressing.py:376-405, :307-340)
//   P6  erase + write pass over M fused with the read-key dot products on M_t                      (access.py:32-63, :283-284)
//   P7  link       : ONE pass over L: update, zero diagonal, store, forward / backward reads        (addressing.py:155-218)
//   P8  read weights, precedence, reads = rw x M_t, output linear + clip                          (access.py:285-303, :151; dnc.py:118-122)
//
// Allocation (P5): the reference sorts usage ascending (top_k of 1-u, ties to the lower index),
// takes an exclusive cumprod and un-sorts.  Equivalently a[n] = (1-u[n]) * prod_{m before n} u[m]
// with "m before n" <=> u[m] < u[n] or (u[m] == u[n] and m < n): an O(N^2 / threads) product with
// no sort and no permutation (the products are taken in index order, not sorted order: same
// value up to fp32 rounding of a <=512-term product of numbers in [0,1]).
#include "dnc_common.h"

// The reference evaluates usage / allocation as separate TF ops, each rounded to fp32; the allocation
// SORTS usage, so which of two almost-equal slots wins can hinge on the last bit.  Keep the same
// op-by-op rounding here (no fused multiply-add contraction) so ties and near-ties resolve as they do
// in an op-by-op fp32 evaluation.
#pragma clang fp contract(off)

struct DncFwdArgs {
    DncDims d;
    const float* xproj;   // [B,S,4*hid] (n' = unit*4+gate), no bias
    const float* Wr;      // [ldz][4*hid], row K = bias
    const float* Wi;      // [ldh][IP],   row hid = bias
    const float* Wy;      // [ldy][OP],   row Ky = bias
    // state, updated in place
    float* mem;           // [B,N,W]
    float* link;          // [B,Wn,N,N]
    float* usage;         // [B,N]
    float* rw;            // [B,R,N]
    float* ww;            // [B,Wn,N]
    float* prec;          // [B,Wn,N]
    float* reads;         // [B,R,W]    (access_output)
    float* hc;            // [B,2*hid]  (hidden then cell)
    float* out;           // [B,S,O]
    // per-step records for BPTT (all nullable, all-or-none)
    float* rec_z;         // [B,S,ldz]  [reads_prev ; h_prev ; 1 ; 0..]
    float* rec_gates;     // [B,S,4*hid] activated gates (i, j, sigmoid(f+1), o per unit)
    float* rec_c;         // [B,S,hid]  cell before clipping
    float* rec_hc;        // [B,S,ldh]  [clipped h ; 1 ; 0..]
    float* rec_yin;       // [B,S,ldy]  [clipped h ; reads_t ; 1 ; 0..]
    float* rec_ifc;       // [B,S,IP]   activated interface
    float* rec_u;         // [B,S,N]
    float* rec_ww;        // [B,S,Wn,N]
    float* rec_rw;        // [B,S,R,N]
    float* rec_cw;        // [B,S,Wn,N]
    float* rec_cr;        // [B,S,R,N]
    float* rec_al;        // [B,S,Wn,N] allocation weights
    float* rec_p;         // [B,S,Wn,N] precedence after the step
    float* rec_fwd;       // [B,S,R,Wn,N]
    float* rec_bwd;       // [B,S,R,Wn,N]
    float* rec_M;         // [B,S,N,W]
    float* rec_L;         // [B,S,Wn,N,N]
    float* rec_ypre;      // [B,S,O]    output before clipping
};

constexpr int DNC_PFM = 4;    // rows of M a lane group requests per batch
constexpr int DNC_PFL = 1;    // rows of the link a wave requests per batch (more spill at 128 VGPRs)

struct DncLds {
    int part, Z, C, I, U, RW, WW, P, CW, CR, AL, FWD, BWD, SC, total;
};

static void dnc_fwd_lds(const DncDims& d, DncLds& L) {
    const int nslG = DT / d.hid > 0 ? DT / d.hid : 1;
    const int nslI = DT / (d.IP / 4) > 0 ? DT / (d.IP / 4) : 1;
    int part = nslG * 4 * d.hid;
    if (nslI * d.IP > part) part = nslI * d.IP;
    const int nslA = DT / d.N > 0 ? DT / d.N : 1;
    if (nslA * d.N > part) part = nslA * d.N;
    if (DW * d.R * 256 > part) part = DW * d.R * 256;                 // link pass: per-wave backward partials of one 256-column block
    const int RWd = d.R * d.W;
    int nslR4 = DT / (RWd / 4) > 0 ? DT / (RWd / 4) : 1;              // reads: (head, float4 of the word) x row slices
    if (nslR4 > d.N) nslR4 = d.N;
    if (nslR4 * RWd > part) part = nslR4 * RWd;
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    L.part = take(part);
    L.Z = take(d.K); L.C = take(d.hid); L.I = take(d.IP);
    L.U = take(d.N); L.RW = take(d.R * d.N); L.WW = take(d.Wn * d.N); L.P = take(d.Wn * d.N);
    L.CW = take(d.Wn * d.N); L.CR = take(d.R * d.N); L.AL = take(d.N);
    L.FWD = take(d.R * d.Wn * d.N); L.BWD = take(d.R * d.Wn * d.N);
    L.SC = take(64);
    L.total = o;
}

__global__ __launch_bounds__(DT) void dnc_seq_fwd_kernel(DncFwdArgs a, DncLds L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DncDims& d = a.d;
    const int b = blockIdx.x, tid0 = threadIdx.x;
    const int N = d.N, W = d.W, R = d.R, Wn = d.Wn, hid = d.hid, S = d.S, K = d.K, IP = d.IP, RWd = R * W;
    const float clipv = d.clip;
    const float EPS = 1e-6f;

    float* sPart = smem + L.part;
    float* sZ = smem + L.Z;      // [reads_prev ; h_prev]
    float* sC = smem + L.C;      // cell
    float* sI = smem + L.I;      // interface (activated)
    float* sU = smem + L.U;      // usage
    float* sRW = smem + L.RW;    // read weights
    float* sWW = smem + L.WW;    // write weights
    float* sP = smem + L.P;      // precedence
    float* sCW = smem + L.CW;    // write content weights
    float* sCR = smem + L.CR;    // read content weights
    float* sAL = smem + L.AL;    // allocation scratch / simulated usage
    float* sFWD = smem + L.FWD;  // [R][Wn][N]
    float* sBWD = smem + L.BWD;  // [R][Wn][N]
    float* sSC = smem + L.SC;    // scalars
    f32x4* sPart4 = reinterpret_cast<f32x4*>(sPart);

    float* gM = a.mem + (size_t)b * N * W;
    float* gL = a.link + (size_t)b * Wn * N * N;

    // rows of M are handled by groups of LPR lanes (one float4 of the word per lane)
    int LPR = 1;
    while (LPR * 4 < W) LPR <<= 1;             // W <= 256 -> LPR <= 64
    const int W4 = W >> 2;

    // work decomposition
    const int nslG = max(1, DT / hid), kperG = (K + nslG - 1) / nslG;
    const int icg = IP >> 2, nslI = max(1, DT / icg), kperI = (hid + nslI - 1) / nslI;
    const int nslA = max(1, DT / N), mperA = (N + nslA - 1) / nslA;
    const int RW4 = RWd >> 2, nslR4 = min(max(1, DT / RW4), N), nperR4 = (N + nslR4 - 1) / nslR4;

    // ---- load state
    for (int i = tid0; i < N; i += DT) sU[i] = a.usage[(size_t)b * N + i];
    for (int i = tid0; i < R * N; i += DT) sRW[i] = a.rw[(size_t)b * R * N + i];
    for (int i = tid0; i < Wn * N; i += DT) { sWW[i] = a.ww[(size_t)b * Wn * N + i]; sP[i] = a.prec[(size_t)b * Wn * N + i]; }
    for (int i = tid0; i < RWd; i += DT) sZ[i] = a.reads[(size_t)b * RWd + i];
    for (int i = tid0; i < hid; i += DT) {
        sZ[RWd + i] = a.hc[(size_t)b * 2 * hid + i];
        sC[i] = a.hc[(size_t)b * 2 * hid + hid + i];
    }
    __syncthreads();

    const f32x4* Wr4 = reinterpret_cast<const f32x4*>(a.Wr);
    const f32x4* Wi4 = reinterpret_cast<const f32x4*>(a.Wi);

    for (int t = 0; t < S; ++t) {
        int tid_op = tid0;
        asm volatile("" : "+v"(tid_op));       // keep per-thread index math inside the step (no hoist + spill)
        const int tid = tid_op, lane = tid & 63, wave = tid >> 6;
        const size_t bt = (size_t)b * S + t;

        // ------------------------------------------------------------ P1: LSTM
        f32x4 xg = {0.f, 0.f, 0.f, 0.f};
        if (tid < hid) xg = reinterpret_cast<const f32x4*>(a.xproj)[bt * hid + tid] + Wr4[(size_t)K * hid + tid];
        const bool rec = a.rec_z != nullptr;
        if (rec) for (int i = tid; i < d.ldz; i += DT) a.rec_z[bt * d.ldz + i] = (i < K) ? sZ[i] : (i == K ? 1.f : 0.f);
        if (tid < nslG * hid) {
            const int j = tid % hid, ks = tid / hid;
            const int k0 = ks * kperG, k1 = min(K, k0 + kperG);
            // explicit two-batch stream: the rolled loop keeps ONE load in flight (common.h)
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (k0 < k1) acc = ntk_stream_matvec<4>(Wr4 + j, hid, sZ, k0, k1, K);
            sPart4[ks * hid + j] = acc;
        }
        __syncthreads();
        if (tid < hid) {
            f32x4 g = xg;
            for (int ks = 0; ks < nslG; ++ks) g += sPart4[ks * hid + tid];
            const float gi = dnc_sigmoid(g[0]), gj = tanhf(g[1]);
            const float gf = dnc_sigmoid(g[2] + 1.0f);              // snt.LSTM forget_bias = 1.0
            const float go = dnc_sigmoid(g[3]);
            const float c2 = gf * sC[tid] + gi * gj;
            const float h2 = tanhf(c2) * go;
            sC[tid] = dnc_clip(c2, clipv);                          // dnc.py:112-113
            sZ[RWd + tid] = dnc_clip(h2, clipv);
            if (rec) {
                f32x4 ga = {gi, gj, gf, go};
                reinterpret_cast<f32x4*>(a.rec_gates)[bt * hid + tid] = ga;
                a.rec_c[bt * hid + tid] = c2;
                a.rec_hc[bt * d.ldh + tid] = dnc_clip(h2, clipv);
                a.rec_yin[bt * d.ldy + tid] = dnc_clip(h2, clipv);
            }
        } else if (rec && tid - hid < d.ldh - hid) {
            a.rec_hc[bt * d.ldh + tid] = (tid == hid) ? 1.f : 0.f;
        }
        __syncthreads();
        // ------------------------------------------------------------ P2: interface
        if (tid < nslI * icg) {
            const int cg = tid % icg, ks = tid / icg;
            const int k0 = ks * kperI, k1 = min(hid, k0 + kperI);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (k0 < k1) acc = ntk_stream_matvec<4>(Wi4 + cg, icg, sZ + RWd, k0, k1, hid);
            sPart4[ks * icg + cg] = acc;
        }
        __syncthreads();
        for (int c = tid; c < IP; c += DT) {
            float v = a.Wi[(size_t)hid * IP + c];
            for (int ks = 0; ks < nslI; ++ks) v += sPart[ks * IP + c];
            float r = v;
            if (c >= d.oE && c < d.oRm) r = dnc_sigmoid(v);                      // erase, free, alloc, write gates
            else if ((c >= d.oBw && c < d.oKr) || (c >= d.oBr && c < d.I)) r = dnc_softplus(v);   // strengths (weighted_softmax)
            sI[c] = r;
        }
        __syncthreads();
        if (tid < R) {                                                           // read_mode softmax (access.py:186-187)
            const int nm = 1 + 2 * Wn;
            float* rm = sI + d.oRm + tid * nm;
            float mx = -INFINITY;
            for (int q = 0; q < nm; ++q) mx = fmaxf(mx, rm[q]);
            float s = 0.f;
            for (int q = 0; q < nm; ++q) s += expf(rm[q] - mx);
            for (int q = 0; q < nm; ++q) rm[q] = expf(rm[q] - mx) / s;
        }
        // ------------------------------------------------------------ P3: usage (addressing.py:342-374)
        for (int n = tid; n < N; n += DT) {
            float pw = 1.f;
            for (int j = 0; j < Wn; ++j) pw *= (1.0f - sWW[j * N + n]);
            float u = sU[n];
            u = u + (1.0f - u) * (1.0f - pw);
            float phi = 1.f;
            for (int i = 0; i < R; ++i) phi *= (1.0f - sI[d.oF + i] * sRW[i * N + n]);
            u *= phi;
            sU[n] = u;
            sAL[n] = u;          // simulated usage for the allocation of successive write heads
        }
        // ------------------------------------------------------------ P4: write content weights on M_{t-1}
        {
            const int grp = tid / LPR, gl = tid % LPR, ngrp = DT / LPR;
            // key norms (tiny, recomputed by every group leader; keys are in LDS)
            // the rows of a group are requested DNC_PFM at a time (one L2 / Infinity-Cache round trip per batch, not per row)
            for (int n0 = grp; n0 < N; n0 += ngrp * DNC_PFM) {
                f32x4 mb[DNC_PFM];
#pragma unroll
                for (int u = 0; u < DNC_PFM; ++u) {
                    const int n = n0 + u * ngrp;
                    mb[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (n < N && gl < W4) mb[u] = reinterpret_cast<const f32x4*>(gM + (size_t)n * W)[gl];
                }
#pragma unroll
                for (int u = 0; u < DNC_PFM; ++u) {
                    const int n = n0 + u * ngrp;
                    if (n >= N) break;
                    const f32x4 m = mb[u];
                    float nsq = m[0] * m[0] + m[1] * m[1] + m[2] * m[2] + m[3] * m[3];
                    for (int o = LPR >> 1; o > 0; o >>= 1) nsq += __shfl_xor(nsq, o, 64);
                    for (int j = 0; j < Wn; ++j) {
                        float dot = 0.f, ksq = 0.f;
                        if (gl < W4) {
                            const float* kp = sI + d.oKw + j * W + gl * 4;
                            dot = kp[0] * m[0] + kp[1] * m[1] + kp[2] * m[2] + kp[3] * m[3];
                            ksq = kp[0] * kp[0] + kp[1] * kp[1] + kp[2] * kp[2] + kp[3] * kp[3];
                        }
                        for (int o = LPR >> 1; o > 0; o >>= 1) { dot += __shfl_xor(dot, o, 64); ksq += __shfl_xor(ksq, o, 64); }
                        if (gl == 0) {
                            const float sim = dot / (sqrtf(ksq + EPS) * sqrtf(nsq + EPS) + EPS);
                            sCW[j * N + n] = sim * sI[d.oBw + j];
                        }
                    }
                }
            }
        }
        __syncthreads();
        lds_softmax_rows(sCW, Wn, N, wave, lane);
        __syncthreads();
        // ------------------------------------------------------------ P5: allocation + write weights (access.py:220-257)
        if (rec) {
            for (int c = tid; c < IP; c += DT) a.rec_ifc[bt * IP + c] = sI[c];
            for (int n = tid; n < N; n += DT) a.rec_u[bt * N + n] = sU[n];
            for (int i = tid; i < Wn * N; i += DT) a.rec_cw[bt * Wn * N + i] = sCW[i];
        }
        for (int j = 0; j < Wn; ++j) {
            if (tid < nslA * N) {
                const int n = tid % N, sl = tid / N;
                // as in the reference: nonusage = 1 - (eps + (1-eps) u); ordering on nonusage (descending, ties to
                // the lower index = tf.nn.top_k); the cumprod runs over sorted_usage = 1 - sorted_nonusage
                const float nun = 1.0f - (EPS + (1.0f - EPS) * sAL[n]);
                const int m0 = sl * mperA, m1 = min(N, m0 + mperA);
                float prod = 1.f;
                for (int m = m0; m < m1; ++m) {
                    const float num = 1.0f - (EPS + (1.0f - EPS) * sAL[m]);
                    const bool before = (num > nun) || (num == nun && m < n);
                    prod *= before ? (1.0f - num) : 1.0f;
                }
                sPart[sl * N + n] = prod;
            }
            __syncthreads();
            const float ag = sI[d.oAg + j], wg = sI[d.oWg + j];
            for (int n = tid; n < N; n += DT) {
                const float nun = 1.0f - (EPS + (1.0f - EPS) * sAL[n]);
                float prod = 1.f;
                for (int sl = 0; sl < nslA; ++sl) prod *= sPart[sl * N + n];
                const float al = nun * prod;
                if (rec) a.rec_al[(bt * Wn + j) * N + n] = al;
                sWW[j * N + n] = wg * (ag * al + (1.0f - ag) * sCW[j * N + n]);
                sAL[n] = sAL[n] + (1.0f - sAL[n]) * (ag * wg) * al;            // addressing.py:336-337
            }
            __syncthreads();
        }
        // ------------------------------------------------------------ P6: erase + write on M, read-key dots on M_t
        {
            const int grp = tid / LPR, gl = tid % LPR, ngrp = DT / LPR;
            for (int n0 = grp; n0 < N; n0 += ngrp * DNC_PFM) {
              f32x4 mb[DNC_PFM];
#pragma unroll
              for (int u = 0; u < DNC_PFM; ++u) {
                  const int n = n0 + u * ngrp;
                  mb[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                  if (n < N && gl < W4) mb[u] = reinterpret_cast<const f32x4*>(gM + (size_t)n * W)[gl];
              }
#pragma unroll
              for (int u = 0; u < DNC_PFM; ++u) {
                const int n = n0 + u * ngrp;
                if (n >= N) break;
                f32x4 m = mb[u];
                if (gl < W4) {
                    f32x4 E = {1.f, 1.f, 1.f, 1.f}, A = {0.f, 0.f, 0.f, 0.f};
                    for (int j = 0; j < Wn; ++j) {
                        const float wwn = sWW[j * N + n];
                        const float* ep = sI + d.oE + j * W + gl * 4;
                        const float* vp = sI + d.oV + j * W + gl * 4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { E[e] *= (1.0f - wwn * ep[e]); A[e] += wwn * vp[e]; }
                    }
                    m = m * E + A;
                    reinterpret_cast<f32x4*>(gM + (size_t)n * W)[gl] = m;
                    if (rec) reinterpret_cast<f32x4*>(a.rec_M + (bt * N + n) * W)[gl] = m;
                }
                float nsq = m[0] * m[0] + m[1] * m[1] + m[2] * m[2] + m[3] * m[3];
                for (int o = LPR >> 1; o > 0; o >>= 1) nsq += __shfl_xor(nsq, o, 64);
                for (int i = 0; i < R; ++i) {
                    float dot = 0.f, ksq = 0.f;
                    if (gl < W4) {
                        const float* kp = sI + d.oKr + i * W + gl * 4;
                        dot = kp[0] * m[0] + kp[1] * m[1] + kp[2] * m[2] + kp[3] * m[3];
                        ksq = kp[0] * kp[0] + kp[1] * kp[1] + kp[2] * kp[2] + kp[3] * kp[3];
                    }
                    for (int o = LPR >> 1; o > 0; o >>= 1) { dot += __shfl_xor(dot, o, 64); ksq += __shfl_xor(ksq, o, 64); }
                    if (gl == 0) {
                        const float sim = dot / (sqrtf(ksq + EPS) * sqrtf(nsq + EPS) + EPS);
                        sCR[i * N + n] = sim * sI[d.oBr + i];
                    }
                }
              }
            }
        }
        // ------------------------------------------------------------ P7: link pass (one read + one write of L)
        for (int i = tid; i < R * Wn * N; i += DT) sFWD[i] = 0.f;
        __syncthreads();          // sWW final, sFWD zeroed, M_t stored (needed by P8 after the later barriers)
        for (int j = 0; j < Wn; ++j) {
            float* Lj = gL + (size_t)j * N * N;
            for (int cb = 0; cb < N; cb += 256) {             // 256-column blocks: lane owns 4 columns
                const int b0 = cb + lane * 4;
                const bool colok = b0 < N;
                f32x4 wwb = {0.f, 0.f, 0.f, 0.f}, pb = {0.f, 0.f, 0.f, 0.f};
                f32x4 rwb[4];
                f32x4 accB[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { rwb[i] = wwb; accB[i] = wwb; }
                if (colok) {
                    wwb = *reinterpret_cast<const f32x4*>(sWW + j * N + b0);
                    pb = *reinterpret_cast<const f32x4*>(sP + j * N + b0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (i < R) rwb[i] = *reinterpret_cast<const f32x4*>(sRW + i * N + b0);
                }
                // a wave's rows in batches of DNC_PFL (1: at the 128 registers a 1024-thread workgroup leaves a thread, a batch of
                // four rows spills 100+ registers -- measured; the rows are private to the wave)
                for (int r0 = wave; r0 < N; r0 += DW * DNC_PFL) {
                  f32x4 lcur[DNC_PFL];
#pragma unroll
                  for (int u = 0; u < DNC_PFL; ++u) {
                      const int r = r0 + u * DW;
                      lcur[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                      if (colok && r < N) lcur[u] = *reinterpret_cast<const f32x4*>(Lj + (size_t)r * N + b0);
                  }
#pragma unroll
                  for (int u = 0; u < DNC_PFL; ++u) {
                    const int r = r0 + u * DW;
                    if (r < N) {                                 // wave-uniform
                    const float wwa = sWW[j * N + r];
                    f32x4 l = lcur[u];
                    if (colok) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = (1.0f - wwa - wwb[e]) * l[e] + wwa * pb[e];
                            if (b0 + e == r) v = 0.f;                            // matrix_set_diag(link, 0)
                            l[e] = v;
                        }
                        *reinterpret_cast<f32x4*>(Lj + (size_t)r * N + b0) = l;
                        if (rec) *reinterpret_cast<f32x4*>(a.rec_L + ((bt * Wn + j) * N + r) * N + b0) = l;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (i < R) {
                            // forward: fwd[i][r] = sum_b rw_prev[i][b] * L[r][b]; backward: bwd[i][b] += rw_prev[i][r] * L[r][b]
                            float f = rwb[i][0] * l[0] + rwb[i][1] * l[1] + rwb[i][2] * l[2] + rwb[i][3] * l[3];
                            f = wave_sum(f);
                            if (lane == 0) sFWD[(i * Wn + j) * N + r] += f;      // only this wave touches row r
                            accB[i] += sRW[i * N + r] * l;
                        }
                    }
                    }
                    __builtin_amdgcn_sched_barrier(0);      // one row at a time: the scheduler otherwise hoists every row's LDS operands
                  }
                }
                // fixed-order reduction of the per-wave backward partials of this column block
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < R) *reinterpret_cast<f32x4*>(sPart + ((wave * R + i) * 256 + lane * 4)) = accB[i];
                __syncthreads();
                for (int idx = tid; idx < R * 256; idx += DT) {
                    const int i = idx >> 8, c = idx & 255;
                    if (cb + c < N) {
                        float s = 0.f;
                        for (int w = 0; w < DW; ++w) s += sPart[(w * R + i) * 256 + c];
                        sBWD[(i * Wn + j) * N + cb + c] = s;
                    }
                }
                __syncthreads();
            }
        }
        // ------------------------------------------------------------ P8: read weights, precedence, reads, output
        lds_softmax_rows(sCR, R, N, wave, lane);
        for (int j = wave; j < Wn; j += DW) {                 // sum of write weights per head (waves not used above)
            float s = 0.f;
            for (int n = lane; n < N; n += 64) s += sWW[j * N + n];
            s = wave_sum(s);
            if (lane == 0) sSC[j] = s;
        }
        __syncthreads();
        {
            const int nm = 1 + 2 * Wn;
            for (int idx = tid; idx < R * N; idx += DT) {
                const int i = idx / N, n = idx - i * N;
                const float* rm = sI + d.oRm + i * nm;
                float v = rm[2 * Wn] * sCR[idx];
                for (int j = 0; j < Wn; ++j)
                    v += rm[Wn + j] * sFWD[(i * Wn + j) * N + n] + rm[j] * sBWD[(i * Wn + j) * N + n];
                sRW[idx] = v;
                if (rec) a.rec_rw[bt * R * N + idx] = v;
            }
            for (int idx = tid; idx < Wn * N; idx += DT) {
                const int j = idx / N;
                const float pn = (1.0f - sSC[j]) * sP[idx] + sWW[idx];           // addressing.py:238-240
                sP[idx] = pn;
                if (rec) { a.rec_p[bt * Wn * N + idx] = pn; a.rec_ww[bt * Wn * N + idx] = sWW[idx]; }
            }
            if (rec) {
                for (int idx = tid; idx < R * N; idx += DT) a.rec_cr[bt * R * N + idx] = sCR[idx];
                for (int idx = tid; idx < R * Wn * N; idx += DT) {
                    a.rec_fwd[bt * R * Wn * N + idx] = sFWD[idx];
                    a.rec_bwd[bt * R * Wn * N + idx] = sBWD[idx];
                }
            }
        }
        __syncthreads();
        if (tid < nslR4 * RW4) {                              // reads = rw x M_t: a thread = (head, float4 of the word) x row slice
            const int o4 = tid % RW4, sl = tid / RW4;
            const int i = o4 / W4, w4 = o4 - i * W4;
            const int n0 = sl * nperR4, n1 = min(N, n0 + nperR4);
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            const f32x4* mp = reinterpret_cast<const f32x4*>(gM) + w4;
            for (int n = n0; n < n1; n += 4) {
                f32x4 mv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) mv[u] = mp[(size_t)min(n + u, N - 1) * W4];
#pragma unroll
                for (int u = 0; u < 4; ++u) s += ((n + u < n1) ? sRW[i * N + n + u] : 0.f) * mv[u];
            }
            reinterpret_cast<f32x4*>(sPart)[sl * RW4 + o4] = s;
        }
        __syncthreads();
        if (tid < RWd) {
            float s = 0.f;
            for (int sl = 0; sl < nslR4; ++sl) s += sPart[sl * RWd + tid];
            sZ[tid] = s;
            if (rec) a.rec_yin[bt * d.ldy + hid + tid] = s;
        } else if (rec && tid >= RWd && tid < RWd + (d.ldy - d.Ky)) {
            a.rec_yin[bt * d.ldy + d.Ky + (tid - RWd)] = (tid == RWd) ? 1.f : 0.f;
        }
        __syncthreads();
        for (int o = wave; o < d.O; o += DW) {                // y = clip([h ; reads] Wy + by)   (dnc.py:118-122)
            float s = 0.f;
            for (int k = lane; k < d.Ky; k += 64) {
                const float zv = (k < hid) ? sZ[RWd + k] : sZ[k - hid];
                s += zv * a.Wy[(size_t)k * d.OP + o];
            }
            s = wave_sum(s);
            if (lane == 0) {
                const float pre = s + a.Wy[(size_t)d.Ky * d.OP + o];
                a.out[bt * d.O + o] = dnc_clip(pre, clipv);
                if (rec) a.rec_ypre[bt * d.O + o] = pre;
            }
        }
        __syncthreads();
    }

    // ---- store state
    for (int i = tid0; i < N; i += DT) a.usage[(size_t)b * N + i] = sU[i];
    for (int i = tid0; i < R * N; i += DT) a.rw[(size_t)b * R * N + i] = sRW[i];
    for (int i = tid0; i < Wn * N; i += DT) { a.ww[(size_t)b * Wn * N + i] = sWW[i]; a.prec[(size_t)b * Wn * N + i] = sP[i]; }
    for (int i = tid0; i < RWd; i += DT) a.reads[(size_t)b * RWd + i] = sZ[i];
    for (int i = tid0; i < hid; i += DT) {
        a.hc[(size_t)b * 2 * hid + i] = sZ[RWd + i];
        a.hc[(size_t)b * 2 * hid + hid + i] = sC[i];
    }
}

extern "C" int ntk_dnc_padded_dims(int N, int W, int R, int Wn, int hid, int O,
                                   int* I, int* IP, int* K, int* ldz, int* ldh, int* Ky, int* ldy, int* OP) {
    DncDims d;
    dnc_fill_dims(d, 1, 1, N, W, R, Wn, hid, O, 0.f);
    if (I) *I = d.I;
    if (IP) *IP = d.IP;
    if (K) *K = d.K;
    if (ldz) *ldz = d.ldz;
    if (ldh) *ldh = d.ldh;
    if (Ky) *Ky = d.Ky;
    if (ldy) *ldy = d.ldy;
    if (OP) *OP = d.OP;
    return NTK_OK;
}

extern "C" int ntk_dnc_seq_fwd(int B, int S, int N, int W, int R, int Wn, int hid, int O, float clip_value,
                               const float* xproj, const float* Wr, const float* Wi, const float* Wy,
                               float* mem, float* link, float* usage, float* rw, float* ww, float* prec,
                               float* reads, float* hc, float* out,
                               float* rec_z, float* rec_gates, float* rec_c, float* rec_hc, float* rec_yin,
                               float* rec_ifc, float* rec_u, float* rec_ww, float* rec_rw, float* rec_cw,
                               float* rec_cr, float* rec_al, float* rec_p, float* rec_fwd, float* rec_bwd,
                               float* rec_M, float* rec_L, float* rec_ypre, void* stream) {
    DncFwdArgs a;
    dnc_fill_dims(a.d, B, S, N, W, R, Wn, hid, O, clip_value);
    NTK_REQUIRE(B > 0 && S > 0, NTK_ERR_BAD_SHAPE, "ntk_dnc_seq_fwd: B=%d S=%d", B, S);
    NTK_REQUIRE(N >= 4 && (N % 4) == 0 && N <= DT, NTK_ERR_UNSUPPORTED, "ntk_dnc_seq_fwd: memory_size=%d must be a multiple of 4 <= 1024", N);
    NTK_REQUIRE(W >= 4 && (W % 4) == 0 && W <= 256, NTK_ERR_UNSUPPORTED, "ntk_dnc_seq_fwd: word_size=%d must be a multiple of 4 <= 256", W);
    NTK_REQUIRE(R >= 1 && R <= 4 && Wn >= 1 && Wn <= 8, NTK_ERR_UNSUPPORTED, "ntk_dnc_seq_fwd: num_reads=%d (1..4) num_writes=%d (1..8)", R, Wn);
    NTK_REQUIRE(hid >= 1 && hid <= DT && R * W <= DT && O >= 1 && O <= DW, NTK_ERR_UNSUPPORTED,
                "ntk_dnc_seq_fwd: hidden=%d reads*word=%d output=%d exceed one workgroup", hid, R * W, O);
    NTK_REQUIRE(xproj && Wr && Wi && Wy && mem && link && usage && rw && ww && prec && reads && hc && out, NTK_ERR_BAD_PTR,
                "ntk_dnc_seq_fwd: null pointer");
    NTK_REQUIRE(ntk_aligned16(xproj) && ntk_aligned16(Wr) && ntk_aligned16(Wi) && ntk_aligned16(mem) && ntk_aligned16(link),
                NTK_ERR_BAD_PTR, "ntk_dnc_seq_fwd: xproj/Wr/Wi/mem/link must be 16-byte aligned");
    a.xproj = xproj; a.Wr = Wr; a.Wi = Wi; a.Wy = Wy; a.mem = mem; a.link = link; a.usage = usage; a.rw = rw; a.ww = ww;
    a.prec = prec; a.reads = reads; a.hc = hc; a.out = out;
    {
        float* recs[] = {rec_z, rec_gates, rec_c, rec_hc, rec_yin, rec_ifc, rec_u, rec_ww, rec_rw, rec_cw, rec_cr, rec_al,
                         rec_p, rec_fwd, rec_bwd, rec_M, rec_L, rec_ypre};
        int nn = 0;
        for (float* r : recs) nn += (r != nullptr);
        NTK_REQUIRE(nn == 0 || nn == 18, NTK_ERR_BAD_PTR, "ntk_dnc_seq_fwd: record pointers are all-or-none (%d of 18 given)", nn);
        NTK_REQUIRE(nn == 0 || (ntk_aligned16(rec_gates) && ntk_aligned16(rec_M) && ntk_aligned16(rec_L)), NTK_ERR_BAD_PTR,
                    "ntk_dnc_seq_fwd: rec_gates/rec_M/rec_L must be 16-byte aligned");
    }
    a.rec_z = rec_z; a.rec_gates = rec_gates; a.rec_c = rec_c; a.rec_hc = rec_hc; a.rec_yin = rec_yin; a.rec_ifc = rec_ifc;
    a.rec_u = rec_u; a.rec_ww = rec_ww; a.rec_rw = rec_rw; a.rec_cw = rec_cw; a.rec_cr = rec_cr; a.rec_al = rec_al;
    a.rec_p = rec_p; a.rec_fwd = rec_fwd; a.rec_bwd = rec_bwd; a.rec_M = rec_M; a.rec_L = rec_L; a.rec_ypre = rec_ypre;
    DncLds L;
    dnc_fwd_lds(a.d, L);
    const size_t lds_bytes = (size_t)L.total * sizeof(float);
    NTK_REQUIRE(lds_bytes <= 160 * 1024, NTK_ERR_UNSUPPORTED, "ntk_dnc_seq_fwd: needs %zu B of LDS (> 160 KiB)", lds_bytes);
    {
        static NtkLdsAttrCache lds_cache;
        const void* const ks[] = {(const void*)dnc_seq_fwd_kernel};
        const int rc_lds = ntk_raise_lds_limit(lds_cache, ks, 1, "ntk_dnc_seq_fwd");
        if (rc_lds != NTK_OK) return rc_lds;
    }
    dnc_seq_fwd_kernel<<<B, DT, lds_bytes, (hipStream_t)stream>>>(a, L);
    NTK_CHECK_LAUNCH("ntk_dnc_seq_fwd");
    return NTK_OK;
}
