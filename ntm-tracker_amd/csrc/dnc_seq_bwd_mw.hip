// DNC core sequence backward for SEVERAL write heads (num_writes 1..4): the same BPTT as dnc_seq_bwd.hip (what
// tf.gradients computes through tf.nn.dynamic_rnn over dnc.DNC, direct_offset_output_with_dnc.py:615-620) with every
// per-write-head quantity carried per head.  One persistent 1024-thread workgroup per sequence.  This is the general
// path (the reference's own DNC tests use 3 write heads, dnc/access_test.py:28-34, dnc_test-style shapes); every
// benchmark configuration has ONE write head and runs dnc_seq_bwd.hip / dnc_cluster_bwd.hip instead, so this file is
// written for clarity, not tuned.
//
// What several write heads add to the single-head derivation (dnc_seq_bwd.hip header):
//   * memory:     M_t = M_{t-1} . prod_h (1 - ww_h (x) e_h) + sum_h ww_h (x) v_h      (access.py:32-63): the erase
//                 gradient of head h sees the product over the OTHER heads;
//   * link:       one N x N link, precedence vector and gradient per head (addressing.py:183-240); the read heads mix
//                 forward/backward weights of every link: read_mode = [backward x Wn, forward x Wn, content]
//                 (access.py:283-289);
//   * allocation: head h allocates on a SIMULATED usage uc_h, uc_0 = usage_t, uc_{h+1} = uc_h + (1 - uc_h) g_h a_h with
//                 g_h = allocation_gate_h * write_gate_h (addressing.py:307-340, no stop_gradient): gradient flows from
//                 the allocation of head h+1 into a_h, g_h and uc_h, heads walked last to first;
//   * usage:      u_t = (u' + (1 - u')(1 - prod_h (1 - ww'_h))) prod_i (1 - f_i rw'_i), ww' under stop_gradient
//                 (addressing.py:302).
// The simulated usages are re-derived from the recorded usage and allocation weights with the forward kernel's
// expression, op by op (fp contract off), so the usage ORDER every head sorted on is reproduced exactly.
#include "dnc_bwd_args.h"

#pragma clang fp contract(off)

constexpr int MW = 4;         // write heads this kernel is compiled for (loops are guarded by h < Wn)

struct DncMwLds {
    int part, I, DX, WW, WWp, U, Up, Pp, CW, AL, UC, NM, NMw, SIMw, DWW, DCW, DA, gP, DPp, gU, dUC, DAT,
        RWp, CR, gRW, DRW, DF, DB, DSIM, SIMr, DRWp, GZ, DR, DKR, DKW, DE, DV, DHC, gC, DG, SC, total;
};

static void dnc_mw_lds(const DncDims& d, int ldkT, int ldhT, DncMwLds& L) {
    const int N = d.N, RN = d.R * d.N, RWd = d.R * d.W, HN = d.Wn * d.N, HW = d.Wn * d.W;
    const int nslA = DT / N > 0 ? DT / N : 1;
    int part = 2 * nslA * N;
    const int nslZ = DT / (ldkT / 4) > 0 ? DT / (ldkT / 4) : 1;
    const int nslH = DT / (ldhT / 4) > 0 ? DT / (ldhT / 4) : 1;
    if (nslZ * ldkT > part) part = nslZ * ldkT;
    if (nslH * ldhT > part) part = nslH * ldhT;
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    L.part = take(part);
    L.I = take(d.IP); L.DX = take(d.IP);
    L.WW = take(HN); L.WWp = take(HN); L.U = take(N); L.Up = take(N); L.Pp = take(HN); L.CW = take(HN); L.AL = take(HN);
    L.UC = take(HN); L.NM = take(N); L.NMw = take(N); L.SIMw = take(HN); L.DWW = take(HN); L.DCW = take(HN); L.DA = take(HN);
    L.gP = take(HN); L.DPp = take(HN); L.gU = take(N); L.dUC = take(N); L.DAT = take(N);
    L.RWp = take(RN); L.CR = take(RN); L.gRW = take(RN);
    L.DRW = take(RN); L.DF = take(d.Wn * RN); L.DB = take(d.Wn * RN); L.DSIM = take(RN); L.SIMr = take(RN); L.DRWp = take(RN);
    L.GZ = take(ldkT); L.DR = take(RWd); L.DKR = take(RWd); L.DKW = take(HW); L.DE = take(HW); L.DV = take(HW);
    L.DHC = take(d.hid); L.gC = take(d.hid); L.DG = take(4 * d.hid); L.SC = take(64);
    L.total = o;
}

// sSC slots
enum { SC_NKR = 0, SC_NKW = 4, SC_DNKR = 8, SC_DNKW = 12, SC_SW = 16, SC_GPP = 20, SC_S1 = 24, SC_DG = 28, SC_DY = 32,
       SC_DGA = 48, SC_DGW = 52 };

__global__ __launch_bounds__(DT) void dnc_seq_bwd_mw_kernel(DncBwdArgs a, DncMwLds L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DncDims& d = a.d;
    const int b = blockIdx.x, tid0 = threadIdx.x;
    const int N = d.N, W = d.W, R = d.R, Wn = d.Wn, hid = d.hid, S = d.S, K = d.K, IP = d.IP;
    const int RWd = R * W, RN = R * N, HN = Wn * N, HW = Wn * W, NM = 1 + 2 * Wn;
    const float clipv = d.clip, EPS = 1e-6f;

    float* sPart = smem + L.part;
    float* sI = smem + L.I;     float* sDX = smem + L.DX;
    float* sWW = smem + L.WW;   float* sWWp = smem + L.WWp; float* sU = smem + L.U;   float* sUp = smem + L.Up;
    float* sPp = smem + L.Pp;   float* sCW = smem + L.CW;   float* sAL = smem + L.AL; float* sUC = smem + L.UC;
    float* sNM = smem + L.NM;   float* sNMw = smem + L.NMw; float* sSIMw = smem + L.SIMw; float* sDWW = smem + L.DWW;
    float* sDCW = smem + L.DCW; float* sDA = smem + L.DA;   float* sgP = smem + L.gP;   float* sDPp = smem + L.DPp;
    float* sgU = smem + L.gU;   float* sdUC = smem + L.dUC; float* sDAT = smem + L.DAT;
    float* sRWp = smem + L.RWp; float* sCR = smem + L.CR; float* sgRW = smem + L.gRW; float* sDRW = smem + L.DRW;
    float* sDF = smem + L.DF;   float* sDB = smem + L.DB;   float* sDSIM = smem + L.DSIM; float* sSIMr = smem + L.SIMr;
    float* sDRWp = smem + L.DRWp;
    float* sGZ = smem + L.GZ;   float* sDR = smem + L.DR;   float* sDKR = smem + L.DKR; float* sDKW = smem + L.DKW;
    float* sDE = smem + L.DE;   float* sDV = smem + L.DV;   float* sDHC = smem + L.DHC; float* sgC = smem + L.gC;
    float* sDG = smem + L.DG;   float* sSC = smem + L.SC;
    f32x4* sPart4 = reinterpret_cast<f32x4*>(sPart);

    float* gM = a.gM + (size_t)b * N * W;
    float* gLb = a.gL + (size_t)b * Wn * N * N;

    int LPR = 1;
    while (LPR * 4 < W) LPR <<= 1;
    const int W4 = W >> 2;
    const int nslA = max(1, DT / N), mperA = (N + nslA - 1) / nslA;
    const int kg4 = a.ldkT >> 2, hg4 = a.ldhT >> 2;
    const int nslZ = max(1, DT / kg4), nperZ = (4 * hid + nslZ - 1) / nslZ;
    const int nslH = max(1, DT / hg4), nperH = (IP + nslH - 1) / nslH;

    // carried gradients: zero (the loss depends on the outputs only) or what the following segment left behind
    const int ncar = HN + N + RN + a.ldkT + hid;
    float* cy = a.gcarry ? a.gcarry + (size_t)b * ncar : nullptr;
    const bool cin = cy && a.carry_in;
    for (int i = tid0; i < HN; i += DT) sgP[i] = cin ? cy[i] : 0.f;
    for (int i = tid0; i < N; i += DT) sgU[i] = cin ? cy[HN + i] : 0.f;
    for (int i = tid0; i < RN; i += DT) sgRW[i] = cin ? cy[HN + N + i] : 0.f;
    for (int i = tid0; i < a.ldkT; i += DT) sGZ[i] = (cin && i < K) ? cy[HN + N + RN + i] : 0.f;
    for (int i = tid0; i < hid; i += DT) sgC[i] = cin ? cy[HN + N + RN + a.ldkT + i] : 0.f;
    __syncthreads();

    for (int t = S - 1; t >= 0; --t) {
        int tid_op = tid0;
        asm volatile("" : "+v"(tid_op));
        const int tid = tid_op, lane = tid & 63, wave = tid >> 6;
        const int grp = tid / LPR, gl = tid % LPR, ngrp = DT / LPR;
        const size_t bt = (size_t)b * S + t;
        const float* Mt = a.rec_M + bt * N * W;
        const float* Mp = (t > 0) ? a.rec_M + (bt - 1) * N * W : a.mem0 + (size_t)b * N * W;
        const float* gRW = a.rec_rw + bt * RN;
        const float* gFV = a.rec_fwd + bt * Wn * RN;          // [R][Wn][N]
        const float* gBV = a.rec_bwd + bt * Wn * RN;

        // ------------------------------------------------------------ load this step's records
        for (int c = tid; c < IP; c += DT) { sI[c] = a.rec_ifc[bt * IP + c]; sDX[c] = 0.f; }
        for (int n = tid; n < N; n += DT) {
            sU[n] = a.rec_u[bt * N + n];
            sUp[n] = (t > 0) ? a.rec_u[(bt - 1) * N + n] : a.usage0[(size_t)b * N + n];
        }
        for (int i = tid; i < HN; i += DT) {
            sWW[i] = a.rec_ww[bt * HN + i];
            sCW[i] = a.rec_cw[bt * HN + i];
            sAL[i] = a.rec_al[bt * HN + i];
            sWWp[i] = (t > 0) ? a.rec_ww[(bt - 1) * HN + i] : a.ww0[(size_t)b * HN + i];
            sPp[i] = (t > 0) ? a.rec_p[(bt - 1) * HN + i] : a.prec0[(size_t)b * HN + i];
            sDWW[i] = 0.f; sDPp[i] = 0.f;
        }
        for (int i = tid; i < RN; i += DT) {
            sCR[i] = a.rec_cr[bt * RN + i];
            sRWp[i] = (t > 0) ? a.rec_rw[(bt - 1) * RN + i] : a.rw0[(size_t)b * RN + i];
            sDRWp[i] = 0.f;
        }
        for (int i = tid; i < RWd; i += DT) sDKR[i] = 0.f;
        for (int i = tid; i < HW; i += DT) { sDKW[i] = 0.f; sDE[i] = 0.f; sDV[i] = 0.f; }
        if (tid < 64) sSC[tid] = 0.f;
        __syncthreads();
        if (tid < d.OP) {                         // B1: output clip + linear
            float g = 0.f;
            if (tid < d.O) {
                const float pre = a.rec_ypre[bt * d.O + tid];
                g = (clipv <= 0.f || fabsf(pre) < clipv) ? a.dout[bt * d.O + tid] : 0.f;
            }
            sSC[SC_DY + tid] = g;
            a.dypre[bt * d.OP + tid] = g;
        }
        // key norms: read keys, write keys
        if (tid >= 64 && tid < 64 + R + Wn) {
            const int q = tid - 64;
            const float* kp = (q < R) ? sI + d.oKr + q * W : sI + d.oKw + (q - R) * W;
            float ss = 0.f;
            for (int w = 0; w < W; ++w) ss += kp[w] * kp[w];
            sSC[(q < R) ? SC_NKR + q : SC_NKW + (q - R)] = sqrtf(ss + EPS);
        }
        __syncthreads();
        for (int k = tid; k < d.Ky; k += DT) {
            float s = 0.f;
            for (int o = 0; o < d.O; ++o) s += a.Wy[(size_t)k * d.OP + o] * sSC[SC_DY + o];
            if (k < hid) sDHC[k] = sGZ[RWd + k] + s;          // carried d(clipped h) + this step's output path
            else sDR[k - hid] = sGZ[k - hid] + s;             // carried d(reads) + output path
        }
        __syncthreads();

        // ------------------------------------------------------------ B2: pass 1 over M_t: d(rw) from reads, read-key similarities
        for (int n = grp; n < N; n += ngrp) {
            f32x4 m = {0.f, 0.f, 0.f, 0.f};
            if (gl < W4) m = reinterpret_cast<const f32x4*>(Mt + (size_t)n * W)[gl];
            float nsq = m[0] * m[0] + m[1] * m[1] + m[2] * m[2] + m[3] * m[3];
            for (int o = LPR >> 1; o > 0; o >>= 1) nsq += __shfl_xor(nsq, o, 64);
            const float nm = sqrtf(nsq + EPS);
            if (gl == 0) sNM[n] = nm;
            for (int i = 0; i < R; ++i) {
                float t1 = 0.f, dot = 0.f;
                if (gl < W4) {
                    const float* dr = sDR + i * W + gl * 4;
                    const float* kp = sI + d.oKr + i * W + gl * 4;
                    t1 = dr[0] * m[0] + dr[1] * m[1] + dr[2] * m[2] + dr[3] * m[3];
                    dot = kp[0] * m[0] + kp[1] * m[1] + kp[2] * m[2] + kp[3] * m[3];
                }
                for (int o = LPR >> 1; o > 0; o >>= 1) { t1 += __shfl_xor(t1, o, 64); dot += __shfl_xor(dot, o, 64); }
                if (gl == 0) {
                    sDRW[i * N + n] = sgRW[i * N + n] + t1;
                    sSIMr[i * N + n] = dot / (sSC[SC_NKR + i] * nm + EPS);
                }
            }
        }
        __syncthreads();
        // ------------------------------------------------------------ B3: read-weight mix over 1 + 2 Wn modes, read-content softmax (wave i = head i)
        if (wave < R) {
            const int i = wave;
            const float* rm = sI + d.oRm + i * NM;             // [backward x Wn, forward x Wn, content] (access.py:283-289)
            float pb[MW], pf[MW], pc = 0.f, s1 = 0.f;
#pragma unroll
            for (int h = 0; h < MW; ++h) { pb[h] = 0.f; pf[h] = 0.f; }
            const float rc = rm[2 * Wn];
            for (int n = lane; n < N; n += 64) {
                const float g = sDRW[i * N + n];
#pragma unroll
                for (int h = 0; h < MW; ++h)
                    if (h < Wn) { pb[h] += g * gBV[(i * Wn + h) * N + n]; pf[h] += g * gFV[(i * Wn + h) * N + n]; }
                pc += g * sCR[i * N + n];
                s1 += sCR[i * N + n] * (rc * g);
            }
            pc = wave_sum(pc); s1 = wave_sum(s1);
            float dotp = rc * pc;
#pragma unroll
            for (int h = 0; h < MW; ++h)
                if (h < Wn) { pb[h] = wave_sum(pb[h]); pf[h] = wave_sum(pf[h]); dotp += rm[h] * pb[h] + rm[Wn + h] * pf[h]; }
            const float br = sI[d.oBr + i];
            float dbeta = 0.f;
            for (int n = lane; n < N; n += 64) {
                const float g = sDRW[i * N + n];
                const float dscore = sCR[i * N + n] * (rc * g - s1);
                dbeta += dscore * sSIMr[i * N + n];
                sDSIM[i * N + n] = dscore * br;
#pragma unroll
                for (int h = 0; h < MW; ++h)
                    if (h < Wn) { sDF[(i * Wn + h) * N + n] = rm[Wn + h] * g; sDB[(i * Wn + h) * N + n] = rm[h] * g; }
            }
            dbeta = wave_sum(dbeta);
            if (lane == 0) {
#pragma unroll
                for (int h = 0; h < MW; ++h)
                    if (h < Wn) {
                        sDX[d.oRm + i * NM + h] = rm[h] * (pb[h] - dotp);
                        sDX[d.oRm + i * NM + Wn + h] = rm[Wn + h] * (pf[h] - dotp);
                    }
                sDX[d.oRm + i * NM + 2 * Wn] = rc * (pc - dotp);
                sDX[d.oBr + i] = dbeta * (1.0f - expf(-br));   // strengths pass through softplus
            }
        }
        __syncthreads();
        // ------------------------------------------------------------ B4: pass 2 over M_t: d(M_t) and d(read keys)
        {
            f32x4 accK[4];
            float accNk[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { accK[i] = f32x4{0.f, 0.f, 0.f, 0.f}; accNk[i] = 0.f; }
            for (int n = grp; n < N; n += ngrp) {
                if (gl < W4) {
                    const f32x4 m = reinterpret_cast<const f32x4*>(Mt + (size_t)n * W)[gl];
                    f32x4 g = reinterpret_cast<f32x4*>(gM + (size_t)n * W)[gl];
                    const float nm = sNM[n];
                    float dnm = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (i < R) {
                            const float dsim = sDSIM[i * N + n];
                            const float D = sSC[SC_NKR + i] * nm + EPS;
                            const float dot = sSIMr[i * N + n] * D;
                            const float ddot = dsim / D;
                            const float dD = -dsim * dot / (D * D);
                            dnm += dD * sSC[SC_NKR + i];
                            if (gl == 0) accNk[i] += dD * nm;
                            const float* kp = sI + d.oKr + i * W + gl * 4;
                            const float* dr = sDR + i * W + gl * 4;
                            const float rwn = gRW[i * N + n];
#pragma unroll
                            for (int e = 0; e < 4; ++e) g[e] += rwn * dr[e] + ddot * kp[e];
                            accK[i] += ddot * m;
                        }
                    }
                    g += (dnm / nm) * m;
                    reinterpret_cast<f32x4*>(gM + (size_t)n * W)[gl] = g;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < R && gl < W4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(&sDKR[i * W + gl * 4 + e], accK[i][e]);
                    if (gl == 0) atomicAdd(&sSC[SC_DNKR + i], accNk[i]);      // d|kr_i|
                }
            }
        }
        __syncthreads();
        for (int idx = tid; idx < RWd; idx += DT) {
            const int i = idx / W;
            sDX[d.oKr + idx] = sDKR[idx] + sSC[SC_DNKR + i] * sI[d.oKr + idx] / sSC[SC_NKR + i];
        }
        // ------------------------------------------------------------ B5: link pass per write head (gL_h, L_t,h, L_{t-1},h)
        for (int h = 0; h < Wn; ++h) {
            float* gL = gLb + (size_t)h * N * N;
            const float* Lt = a.rec_L + (bt * Wn + h) * N * N;
            const float* Lp = (t > 0) ? a.rec_L + ((bt - 1) * Wn + h) * N * N : a.link0 + ((size_t)b * Wn + h) * N * N;
            const float* ww = sWW + h * N;
            const float* pp = sPp + h * N;
            for (int cb = 0; cb < N; cb += 256) {
                const int b0 = cb + lane * 4;
                const bool colok = b0 < N;
                f32x4 wwb = {0.f, 0.f, 0.f, 0.f}, ppb = wwb, rwpb[4], dbb[4], colRW[4], colWW = wwb, colP = wwb;
#pragma unroll
                for (int i = 0; i < 4; ++i) { rwpb[i] = wwb; dbb[i] = wwb; colRW[i] = wwb; }
                if (colok) {
                    wwb = *reinterpret_cast<const f32x4*>(ww + b0);
                    ppb = *reinterpret_cast<const f32x4*>(pp + b0);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (i < R) {
                            rwpb[i] = *reinterpret_cast<const f32x4*>(sRWp + i * N + b0);
                            dbb[i] = *reinterpret_cast<const f32x4*>(sDB + (i * Wn + h) * N + b0);
                        }
                }
                for (int r = wave; r < N; r += DW) {
                    const float wwa = ww[r];
                    float rowRW[4] = {0.f, 0.f, 0.f, 0.f}, rowWW = 0.f;
                    if (colok) {
                        f32x4 g = *reinterpret_cast<const f32x4*>(gL + (size_t)r * N + b0);
                        const f32x4 lt = *reinterpret_cast<const f32x4*>(Lt + (size_t)r * N + b0);
                        const f32x4 lp = *reinterpret_cast<const f32x4*>(Lp + (size_t)r * N + b0);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (i < R) {
                                const float dfa = sDF[(i * Wn + h) * N + r], rwpa = sRWp[i * N + r];
                                g += dfa * rwpb[i] + rwpa * dbb[i];
                                rowRW[i] = dbb[i][0] * lt[0] + dbb[i][1] * lt[1] + dbb[i][2] * lt[2] + dbb[i][3] * lt[3];
                                colRW[i] += dfa * lt;
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (b0 + e == r) g[e] = 0.f;      // the diagonal of L_t is forced to 0
#pragma unroll
                        for (int e = 0; e < 4; ++e) rowWW += g[e] * (ppb[e] - lp[e]);
                        colWW -= g * lp;
                        colP += wwa * g;
                        f32x4 gn;
#pragma unroll
                        for (int e = 0; e < 4; ++e) gn[e] = (1.0f - wwa - wwb[e]) * g[e];
                        *reinterpret_cast<f32x4*>(gL + (size_t)r * N + b0) = gn;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (i < R) {
                            const float s = wave_sum(rowRW[i]);
                            if (lane == 0) atomicAdd(&sDRWp[i * N + r], s);
                        }
                    }
                    const float s = wave_sum(rowWW);
                    if (lane == 0) atomicAdd(&sDWW[h * N + r], s);
                }
                if (colok) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) if (i < R) atomicAdd(&sDRWp[i * N + b0 + e], colRW[i][e]);
                        atomicAdd(&sDWW[h * N + b0 + e], colWW[e]);
                        atomicAdd(&sDPp[h * N + b0 + e], colP[e]);
                    }
                }
            }
        }
        __syncthreads();
        // ------------------------------------------------------------ B6: precedence per head (wave h computes its two scalars)
        if (wave < Wn) {
            const int h = wave;
            float sw = 0.f, t1 = 0.f;
            for (int n = lane; n < N; n += 64) { sw += sWW[h * N + n]; t1 += sgP[h * N + n] * sPp[h * N + n]; }
            sw = wave_sum(sw); t1 = wave_sum(t1);
            if (lane == 0) { sSC[SC_SW + h] = sw; sSC[SC_GPP + h] = t1; }
        }
        __syncthreads();
        for (int i = tid; i < HN; i += DT) {
            const int h = i / N;
            sDPp[i] += (1.0f - sSC[SC_SW + h]) * sgP[i];
            sDWW[i] += sgP[i] - sSC[SC_GPP + h];
        }
        __syncthreads();
        // ------------------------------------------------------------ B7: write backward over (gM, M_{t-1}); write-key similarities
        {
            f32x4 accE[MW], accV[MW];
#pragma unroll
            for (int h = 0; h < MW; ++h) { accE[h] = f32x4{0.f, 0.f, 0.f, 0.f}; accV[h] = accE[h]; }
            for (int n = grp; n < N; n += ngrp) {
                f32x4 mp = {0.f, 0.f, 0.f, 0.f};
                float t1[MW], dot[MW];
#pragma unroll
                for (int h = 0; h < MW; ++h) { t1[h] = 0.f; dot[h] = 0.f; }
                if (gl < W4) {
                    mp = reinterpret_cast<const f32x4*>(Mp + (size_t)n * W)[gl];
                    const f32x4 g = reinterpret_cast<f32x4*>(gM + (size_t)n * W)[gl];
                    f32x4 fac[MW], Eall = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
                    for (int h = 0; h < MW; ++h) {
                        fac[h] = f32x4{1.f, 1.f, 1.f, 1.f};
                        if (h < Wn) {
                            const float wwn = sWW[h * N + n];
                            const float* ep = sI + d.oE + h * W + gl * 4;
#pragma unroll
                            for (int e = 0; e < 4; ++e) fac[h][e] = 1.0f - wwn * ep[e];
                            Eall = Eall * fac[h];
                        }
                    }
#pragma unroll
                    for (int h = 0; h < MW; ++h) {
                        if (h < Wn) {
                            f32x4 oth = {1.f, 1.f, 1.f, 1.f};                     // product over the other heads
#pragma unroll
                            for (int h2 = 0; h2 < MW; ++h2) if (h2 != h && h2 < Wn) oth = oth * fac[h2];
                            const float wwn = sWW[h * N + n];
                            const float* ep = sI + d.oE + h * W + gl * 4;
                            const float* vp = sI + d.oV + h * W + gl * 4;
                            const float* kp = sI + d.oKw + h * W + gl * 4;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                t1[h] += g[e] * (vp[e] - mp[e] * ep[e] * oth[e]);
                                accE[h][e] -= g[e] * mp[e] * wwn * oth[e];
                                accV[h][e] += g[e] * wwn;
                                dot[h] += kp[e] * mp[e];
                            }
                        }
                    }
                    reinterpret_cast<f32x4*>(gM + (size_t)n * W)[gl] = g * Eall;      // now d(M_{t-1}) (content part added in B10)
                }
                float nsq = mp[0] * mp[0] + mp[1] * mp[1] + mp[2] * mp[2] + mp[3] * mp[3];
                for (int o = LPR >> 1; o > 0; o >>= 1) nsq += __shfl_xor(nsq, o, 64);
#pragma unroll
                for (int h = 0; h < MW; ++h)
                    if (h < Wn)
                        for (int o = LPR >> 1; o > 0; o >>= 1) { t1[h] += __shfl_xor(t1[h], o, 64); dot[h] += __shfl_xor(dot[h], o, 64); }
                if (gl == 0) {
                    const float nm = sqrtf(nsq + EPS);
                    sNMw[n] = nm;
#pragma unroll
                    for (int h = 0; h < MW; ++h)
                        if (h < Wn) {
                            sDWW[h * N + n] += t1[h];
                            sSIMw[h * N + n] = dot[h] / (sSC[SC_NKW + h] * nm + EPS);
                        }
                }
            }
            if (gl < W4) {
#pragma unroll
                for (int h = 0; h < MW; ++h)
                    if (h < Wn) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            atomicAdd(&sDE[h * W + gl * 4 + e], accE[h][e]);
                            atomicAdd(&sDV[h * W + gl * 4 + e], accV[h][e]);
                        }
                    }
            }
        }
        __syncthreads();
        // ------------------------------------------------------------ B8: write-weight mix per head (access.py:252-257); simulated usages
        for (int i = tid; i < HN; i += DT) {
            const int h = i / N;
            const float ga = sI[d.oAg + h], gw = sI[d.oWg + h];
            const float dww = sDWW[i];
            sDA[i] = gw * ga * dww;
            sDCW[i] = gw * (1.0f - ga) * dww;
        }
        if (wave < Wn) {
            const int h = wave;
            const float ga = sI[d.oAg + h], gw = sI[d.oWg + h];
            float dgw = 0.f, dga = 0.f, s1 = 0.f;
            for (int n = lane; n < N; n += 64) {
                const float dww = sDWW[h * N + n];
                dgw += dww * (ga * sAL[h * N + n] + (1.0f - ga) * sCW[h * N + n]);
                dga += gw * dww * (sAL[h * N + n] - sCW[h * N + n]);
                s1 += sCW[h * N + n] * (gw * (1.0f - ga) * dww);
            }
            dgw = wave_sum(dgw); dga = wave_sum(dga); s1 = wave_sum(s1);
            if (lane == 0) { sSC[SC_DGW + h] = dgw; sSC[SC_DGA + h] = dga; sSC[SC_S1 + h] = s1; }
        }
        // uc_0 = usage_t, uc_{h+1} = uc_h + (1 - uc_h) (ga_h gw_h) a_h: the forward kernel's expression, op by op
        for (int n = tid; n < N; n += DT) {
            float uc = sU[n];
            sUC[n] = uc;
            for (int h = 0; h + 1 < Wn; ++h) {
                const float ag = sI[d.oAg + h], wg = sI[d.oWg + h];
                uc = uc + (1.0f - uc) * (ag * wg) * sAL[h * N + n];
                sUC[(h + 1) * N + n] = uc;
            }
            sdUC[n] = 0.f;                                     // d(uc_Wn): the last simulated usage is never used
        }
        __syncthreads();
        // ------------------------------------------------------------ B9: allocation chain, last head first (addressing.py:307-340, :376-405)
        for (int h = Wn - 1; h >= 0; --h) {
            const float gh = sI[d.oAg + h] * sI[d.oWg + h];
            const float* uc = sUC + h * N;
            const float* al = sAL + h * N;
            // gradient reaching a_h: from the write weights and from the next head's simulated usage
            float dgp = 0.f;
            for (int n = tid; n < N; n += DT) {
                const float du = sdUC[n];
                sDAT[n] = sDA[h * N + n] + du * (1.0f - uc[n]) * gh;
                dgp += du * (1.0f - uc[n]) * al[n];
            }
            dgp = wave_sum(dgp);
            if (lane == 0 && dgp != 0.f) atomicAdd(&sSC[SC_DG + h], dgp);
            __syncthreads();
            if (tid < nslA * N) {
                const int n = tid % N, sl = tid / N;
                const float nun = 1.0f - (EPS + (1.0f - EPS) * uc[n]);
                const int m0 = sl * mperA, m1 = min(N, m0 + mperA);
                float prod = 1.f, gsum = 0.f;
                for (int m = m0; m < m1; ++m) {
                    const float num = 1.0f - (EPS + (1.0f - EPS) * uc[m]);
                    const bool before = (num > nun) || (num == nun && m < n);      // m precedes n in the usage sort
                    const bool after = (nun > num) || (nun == num && n < m);
                    prod *= before ? (1.0f - num) : 1.0f;
                    gsum += after ? sDAT[m] * al[m] : 0.f;
                }
                sPart[sl * N + n] = prod;
                sPart[(nslA + sl) * N + n] = gsum;
            }
            __syncthreads();
            for (int n = tid; n < N; n += DT) {
                float prod = 1.f, gsum = 0.f;
                for (int sl = 0; sl < nslA; ++sl) { prod *= sPart[sl * N + n]; gsum += sPart[(nslA + sl) * N + n]; }
                const float ut = 1.0f - (1.0f - (EPS + (1.0f - EPS) * uc[n]));     // sorted_usage = 1 - sorted_nonusage
                const float dut = -sDAT[n] * prod + gsum / ut;
                sdUC[n] = sdUC[n] * (1.0f - gh * al[n]) + (1.0f - EPS) * dut;       // d(uc_h)
            }
            __syncthreads();
        }
        // sdUC is now d(usage_t) through the allocation; write-content softmax backward per head
        for (int i = tid; i < HN; i += DT) {
            const int h = i / N;
            sDCW[i] = sCW[i] * (sDCW[i] - sSC[SC_S1 + h]);                          // d(score)
        }
        __syncthreads();
        if (wave < Wn) {
            const int h = wave;
            float dbeta = 0.f;
            for (int n = lane; n < N; n += 64) dbeta += sDCW[h * N + n] * sSIMw[h * N + n];
            dbeta = wave_sum(dbeta);
            const float bw = sI[d.oBw + h];
            if (lane == 0) {
                sDX[d.oBw + h] = dbeta * (1.0f - expf(-bw));
                const float ga = sI[d.oAg + h], gw = sI[d.oWg + h], dg = sSC[SC_DG + h];
                sDX[d.oWg + h] = (sSC[SC_DGW + h] + dg * ga) * gw * (1.0f - gw);
                sDX[d.oAg + h] = (sSC[SC_DGA + h] + dg * gw) * ga * (1.0f - ga);
            }
        }
        // ------------------------------------------------------------ B10b: pass over M_{t-1}: content part of d(M_{t-1}), d(write keys)
        {
            f32x4 accK[MW];
            float accNk[MW];
#pragma unroll
            for (int h = 0; h < MW; ++h) { accK[h] = f32x4{0.f, 0.f, 0.f, 0.f}; accNk[h] = 0.f; }
            for (int n = grp; n < N; n += ngrp) {
                if (gl < W4) {
                    const f32x4 mp = reinterpret_cast<const f32x4*>(Mp + (size_t)n * W)[gl];
                    f32x4 g = reinterpret_cast<f32x4*>(gM + (size_t)n * W)[gl];
                    const float nm = sNMw[n];
#pragma unroll
                    for (int h = 0; h < MW; ++h) {
                        if (h < Wn) {
                            const float bw = sI[d.oBw + h], nk = sSC[SC_NKW + h];
                            const float dsim = sDCW[h * N + n] * bw;
                            const float D = nk * nm + EPS;
                            const float dot = sSIMw[h * N + n] * D;
                            const float ddot = dsim / D;
                            const float dD = -dsim * dot / (D * D);
                            if (gl == 0) accNk[h] += dD * nm;
                            const float* kp = sI + d.oKw + h * W + gl * 4;
#pragma unroll
                            for (int e = 0; e < 4; ++e) g[e] += ddot * kp[e] + (dD * nk / nm) * mp[e];
                            accK[h] += ddot * mp;
                        }
                    }
                    reinterpret_cast<f32x4*>(gM + (size_t)n * W)[gl] = g;
                }
            }
            if (gl < W4) {
#pragma unroll
                for (int h = 0; h < MW; ++h)
                    if (h < Wn) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) atomicAdd(&sDKW[h * W + gl * 4 + e], accK[h][e]);
                        if (gl == 0) atomicAdd(&sSC[SC_DNKW + h], accNk[h]);
                    }
            }
        }
        // ------------------------------------------------------------ B11: usage backward (addressing.py:279-305, :342-374)
        for (int n = tid; n < N; n += DT) {
            const float g = sgU[n] + sdUC[n];                                       // total d(usage_t)
            float pw = 1.f;
            for (int h = 0; h < Wn; ++h) pw *= (1.0f - sWWp[h * N + n]);
            const float u1 = sUp[n] + (1.0f - sUp[n]) * (1.0f - pw);                // write weights: stop_gradient
            float phi = 1.f;
            for (int i = 0; i < R; ++i) phi *= (1.0f - sI[d.oF + i] * sRWp[i * N + n]);
            const float dphi = g * u1;
            sgU[n] = g * phi * pw;                                                  // carried d(usage_{t-1})
            for (int i = 0; i < R; ++i) {
                float oth = 1.f;
                for (int i2 = 0; i2 < R; ++i2) if (i2 != i) oth *= (1.0f - sI[d.oF + i2] * sRWp[i2 * N + n]);
                sDRWp[i * N + n] += dphi * (-sI[d.oF + i]) * oth;
                sDSIM[i * N + n] = dphi * (-sRWp[i * N + n]) * oth;                 // reuse: per-slot term of d(free_gate_i)
            }
        }
        __syncthreads();
        if (wave < R) {
            const int i = wave;
            float s = 0.f;
            for (int n = lane; n < N; n += 64) s += sDSIM[i * N + n];
            s = wave_sum(s);
            const float fg = sI[d.oF + i];
            if (lane == 0) sDX[d.oF + i] = s * fg * (1.0f - fg);
        }
        for (int idx = tid; idx < HW; idx += DT) {                                  // remaining interface gradients
            const int h = idx / W;
            const float e = sI[d.oE + idx];
            sDX[d.oE + idx] = sDE[idx] * e * (1.0f - e);
            sDX[d.oV + idx] = sDV[idx];
            sDX[d.oKw + idx] = sDKW[idx] + sSC[SC_DNKW + h] * sI[d.oKw + idx] / sSC[SC_NKW + h];
        }
        for (int i = tid; i < RN; i += DT) sgRW[i] = sDRWp[i];                      // carried d(read weights_{t-1})
        for (int i = tid; i < HN; i += DT) sgP[i] = sDPp[i];                        // carried d(precedence_{t-1})
        __syncthreads();
        for (int c = tid; c < IP; c += DT) a.dxi[bt * IP + c] = sDX[c];

        // ------------------------------------------------------------ B14: d(clipped h) += d(interface) . Wi^T
        if (tid < nslH * hg4) {
            const int cg = tid % hg4, sl = tid / hg4;
            const int c0 = sl * nperH, c1 = min(IP, c0 + nperH);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const f32x4* wp4 = reinterpret_cast<const f32x4*>(a.WiT) + (size_t)c0 * hg4 + cg;
#pragma unroll 4
            for (int c = c0; c < c1; ++c, wp4 += hg4) acc += sDX[c] * (*wp4);
            sPart4[sl * hg4 + cg] = acc;
        }
        __syncthreads();
        // ------------------------------------------------------------ B15: clip + snt.LSTM backward
        if (tid < hid) {
            float dh = sDHC[tid];
            for (int sl = 0; sl < nslH; ++sl) dh += sPart[sl * a.ldhT + tid];
            const f32x4 g = reinterpret_cast<const f32x4*>(a.rec_gates)[bt * hid + tid];
            const float gi = g[0], gj = g[1], gf = g[2], go = g[3];
            const float c2 = a.rec_c[bt * hid + tid];
            const float cprev = (t > 0) ? dnc_clip(a.rec_c[(bt - 1) * hid + tid], clipv) : a.hc0[(size_t)b * 2 * hid + hid + tid];
            const float tc = tanhf(c2);
            const float h2 = tc * go;
            const float dh2 = (clipv <= 0.f || fabsf(h2) < clipv) ? dh : 0.f;
            const float dcc = (clipv <= 0.f || fabsf(c2) < clipv) ? sgC[tid] : 0.f;
            const float dc2 = dcc + dh2 * go * (1.0f - tc * tc);
            f32x4 dg;
            dg[0] = dc2 * gj * gi * (1.0f - gi);
            dg[1] = dc2 * gi * (1.0f - gj * gj);
            dg[2] = dc2 * cprev * gf * (1.0f - gf);
            dg[3] = dh2 * tc * go * (1.0f - go);
            sgC[tid] = dc2 * gf;
            reinterpret_cast<f32x4*>(sDG)[tid] = dg;
            reinterpret_cast<f32x4*>(a.dgates)[bt * hid + tid] = dg;
        }
        __syncthreads();
        // ------------------------------------------------------------ B16: d[reads_prev ; h_prev] = dgates . Wr^T
        if (tid < nslZ * kg4) {
            const int cg = tid % kg4, sl = tid / kg4;
            const int r0 = sl * nperZ, r1 = min(4 * hid, r0 + nperZ);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const f32x4* wp4 = reinterpret_cast<const f32x4*>(a.WrT) + (size_t)r0 * kg4 + cg;
#pragma unroll 8
            for (int r = r0; r < r1; ++r, wp4 += kg4) acc += sDG[r] * (*wp4);
            sPart4[sl * kg4 + cg] = acc;
        }
        __syncthreads();
        if (tid < K) {
            float s = 0.f;
            for (int sl = 0; sl < nslZ; ++sl) s += sPart[sl * a.ldkT + tid];
            sGZ[tid] = s;
        }
        __syncthreads();
    }
    if (cy) {
        for (int i = tid0; i < HN; i += DT) cy[i] = sgP[i];
        for (int i = tid0; i < N; i += DT) cy[HN + i] = sgU[i];
        for (int i = tid0; i < RN; i += DT) cy[HN + N + i] = sgRW[i];
        for (int i = tid0; i < a.ldkT; i += DT) cy[HN + N + RN + i] = sGZ[i];
        for (int i = tid0; i < hid; i += DT) cy[HN + N + RN + a.ldkT + i] = sgC[i];
    }
}

int dnc_seq_bwd_mw_launch(const DncBwdArgs& a, hipStream_t stream) {
    NTK_REQUIRE(a.d.Wn >= 1 && a.d.Wn <= MW, NTK_ERR_UNSUPPORTED,
                "ntk_dnc_seq_bwd: num_writes=%d (the BPTT kernels implement 1..%d write heads)", a.d.Wn, MW);
    DncMwLds L;
    dnc_mw_lds(a.d, a.ldkT, a.ldhT, L);
    const size_t lds_bytes = (size_t)L.total * sizeof(float);
    NTK_REQUIRE(lds_bytes <= 160 * 1024, NTK_ERR_UNSUPPORTED, "ntk_dnc_seq_bwd: num_writes=%d needs %zu B of LDS (> 160 KiB)",
                a.d.Wn, lds_bytes);
    {
        static NtkLdsAttrCache lds_cache;
        const void* const ks[] = {(const void*)dnc_seq_bwd_mw_kernel};
        const int rc_lds = ntk_raise_lds_limit(lds_cache, ks, 1, "ntk_dnc_seq_bwd(multi-write)");
        if (rc_lds != NTK_OK) return rc_lds;
    }
    dnc_seq_bwd_mw_kernel<<<a.d.B, DT, lds_bytes, stream>>>(a, L);
    NTK_CHECK_LAUNCH("ntk_dnc_seq_bwd(multi-write)");
    return NTK_OK;
}
