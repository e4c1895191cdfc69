// DNC core sequence backward: full BPTT through the steps recorded by dnc_seq_fwd.hip
// (what tf.gradients computes through tf.nn.dynamic_rnn over dnc.DNC,
// direct_offset_output_with_dnc.py:615-620).  One persistent 1024-thread workgroup per sequence walks the
// steps in reverse.  Carried gradients: memory (N x W) and link (N x N) in global scratch, everything else
// (precedence, usage, read weights, reads, LSTM h/c) in LDS.  Single write head (num_writes == 1, the
// reference default and every benchmark config); up to 4 read heads.  ntk_dnc_seq_bwd hands 2..4 write heads to
// the general kernel in dnc_seq_bwd_mw.hip.
//
// Non-differentiable edges of the reference: write weights enter the usage update under
// tf.stop_gradient (addressing.py:302); the allocation sort passes gradient to the sorted VALUES only
// (top_k, :396-397) -- here the rank-based form: a[n] = (1-u[n]) P[n], P[n] = prod_{m before n} u[m], so
// du[n] = -da[n] P[n] + (sum_{m after n} da[m] a[m]) / u[n]; clip_by_value passes gradient inside the
// interval only (dnc.py:78-82).
//
// Outputs: raw gate gradients [B,S,4*hid], raw interface gradients [B,S,IP], clipped-output gradients
// [B,S,OP]; the weight gradients are k-major GEMMs over those rows (ntk_gemm_tn_f32).
// Column sums of the link pass use LDS float atomics (summation order across waves is not fixed:
// gradients are reproducible to rounding, not bitwise).
#include "dnc_bwd_args.h"

// The allocation gradient re-derives the usage ORDER of the forward pass from the recorded usages.  The forward
// kernel evaluates nonusage = 1 - (eps + (1 - eps) u) op by op (no fused multiply-add); with contraction on, this
// file's fma rounds differently in the last bit, two almost-tied slots can swap order between forward and backward,
// and the usage / free-gate gradients come out wrong by orders of magnitude (found by the per-step gradient probe
// scripts/dev_dnc_stepgrad.py: free_gate 1e-7 -> 8.7e-15 absolute error once the rounding matches).
#pragma clang fp contract(off)

struct DncBwdLds {
    int part, I, DX, WW, WWp, U, Up, Pp, CW, AL, NM, NMw, SIMw, DWW, DCW, DA, gP, DPp, gU, gUn,
        RWp, CR, gRW, DRW, DF, DB, DSIM, SIMr, DRWp, GZ, DR, DKR, DKW, DE, DV, DHC, gC, DG, SC, total;
};

static void dnc_bwd_lds(const DncDims& d, int ldkT, int ldhT, DncBwdLds& L) {
    const int N = d.N, RN = d.R * d.N, RWd = d.R * d.W;
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
    L.WW = take(N); L.WWp = take(N); L.U = take(N); L.Up = take(N); L.Pp = take(N); L.CW = take(N); L.AL = take(N);
    L.NM = take(N); L.NMw = take(N); L.SIMw = take(N); L.DWW = take(N); L.DCW = take(N); L.DA = take(N);
    L.gP = take(N); L.DPp = take(N); L.gU = take(N); L.gUn = take(N);
    L.RWp = take(RN); L.CR = take(RN); L.gRW = take(RN);
    L.DRW = take(RN); L.DF = take(RN); L.DB = take(RN); L.DSIM = take(RN); L.SIMr = take(RN); L.DRWp = take(RN);
    L.GZ = take(ldkT); L.DR = take(RWd); L.DKR = take(RWd); L.DKW = take(d.W); L.DE = take(d.W); L.DV = take(d.W);
    L.DHC = take(d.hid); L.gC = take(d.hid); L.DG = take(4 * d.hid); L.SC = take(64);
    L.total = o;
}

__global__ __launch_bounds__(DT) void dnc_seq_bwd_kernel(DncBwdArgs a, DncBwdLds L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DncDims& d = a.d;
    const int b = blockIdx.x, tid0 = threadIdx.x;
    const int N = d.N, W = d.W, R = d.R, hid = d.hid, S = d.S, K = d.K, IP = d.IP, RWd = R * W, RN = R * N;
    const float clipv = d.clip, EPS = 1e-6f;

    float* sPart = smem + L.part;
    float* sI = smem + L.I;     float* sDX = smem + L.DX;
    float* sWW = smem + L.WW;   float* sWWp = smem + L.WWp; float* sU = smem + L.U;   float* sUp = smem + L.Up;
    float* sPp = smem + L.Pp;   float* sCW = smem + L.CW;   float* sAL = smem + L.AL; float* sNM = smem + L.NM;
    float* sNMw = smem + L.NMw; float* sSIMw = smem + L.SIMw; float* sDWW = smem + L.DWW; float* sDCW = smem + L.DCW;
    float* sDA = smem + L.DA;   float* sgP = smem + L.gP;   float* sDPp = smem + L.DPp; float* sgU = smem + L.gU;
    float* sgUn = smem + L.gUn;
    float* sRWp = smem + L.RWp; float* sCR = smem + L.CR; float* sgRW = smem + L.gRW; float* sDRW = smem + L.DRW; float* sDF = smem + L.DF;
    float* sDB = smem + L.DB;   float* sDSIM = smem + L.DSIM; float* sSIMr = smem + L.SIMr; float* sDRWp = smem + L.DRWp;
    float* sGZ = smem + L.GZ;   float* sDR = smem + L.DR;   float* sDKR = smem + L.DKR; float* sDKW = smem + L.DKW;
    float* sDE = smem + L.DE;   float* sDV = smem + L.DV;   float* sDHC = smem + L.DHC; float* sgC = smem + L.gC;
    float* sDG = smem + L.DG;   float* sSC = smem + L.SC;
    f32x4* sPart4 = reinterpret_cast<f32x4*>(sPart);

    float* gM = a.gM + (size_t)b * N * W;
    float* gL = a.gL + (size_t)b * N * N;

    int LPR = 1;
    while (LPR * 4 < W) LPR <<= 1;
    const int W4 = W >> 2;
    const int nslA = max(1, DT / N), mperA = (N + nslA - 1) / nslA;
    const int kg4 = a.ldkT >> 2, hg4 = a.ldhT >> 2;
    const int nslZ = max(1, DT / kg4), nperZ = (4 * hid + nslZ - 1) / nslZ;
    const int nslH = max(1, DT / hg4), nperH = (IP + nslH - 1) / nslH;

    // carried gradients start at zero (the loss depends on the outputs only) or, for a segment of a longer
    // sequence, at the values the following segment left behind
    float* cy = a.gcarry ? a.gcarry + (size_t)b * (2 * N + RN + a.ldkT + hid) : nullptr;
    const bool cin = cy && a.carry_in;
    for (int i = tid0; i < N; i += DT) { sgP[i] = cin ? cy[i] : 0.f; sgU[i] = cin ? cy[N + i] : 0.f; }
    for (int i = tid0; i < RN; i += DT) sgRW[i] = cin ? cy[2 * N + i] : 0.f;
    for (int i = tid0; i < a.ldkT; i += DT) sGZ[i] = (cin && i < K) ? cy[2 * N + RN + i] : 0.f;
    for (int i = tid0; i < hid; i += DT) sgC[i] = cin ? cy[2 * N + RN + a.ldkT + i] : 0.f;
    __syncthreads();

    for (int t = S - 1; t >= 0; --t) {
        int tid_op = tid0;
        asm volatile("" : "+v"(tid_op));
        const int tid = tid_op, lane = tid & 63, wave = tid >> 6;
        const int grp = tid / LPR, gl = tid % LPR, ngrp = DT / LPR;
        const size_t bt = (size_t)b * S + t;
        const float* Mt = a.rec_M + bt * N * W;
        const float* Mp = (t > 0) ? a.rec_M + (bt - 1) * N * W : a.mem0 + (size_t)b * N * W;
        const float* Lt = a.rec_L + bt * N * N;
        const float* Lp = (t > 0) ? a.rec_L + (bt - 1) * N * N : a.link0 + (size_t)b * N * N;
        // single-use records are read where they are consumed (L2 hits), not staged: keeps config 5 inside 160 KiB
        const float* gRW = a.rec_rw + bt * RN;
        const float* gFV = a.rec_fwd + bt * RN;
        const float* gBV = a.rec_bwd + bt * RN;

        // ------------------------------------------------------------ load this step's records
        for (int c = tid; c < IP; c += DT) { sI[c] = a.rec_ifc[bt * IP + c]; sDX[c] = 0.f; }
        for (int n = tid; n < N; n += DT) {
            sWW[n] = a.rec_ww[bt * N + n];
            sU[n] = a.rec_u[bt * N + n];
            sCW[n] = a.rec_cw[bt * N + n];
            sAL[n] = a.rec_al[bt * N + n];
            sWWp[n] = (t > 0) ? a.rec_ww[(bt - 1) * N + n] : a.ww0[(size_t)b * N + n];
            sUp[n] = (t > 0) ? a.rec_u[(bt - 1) * N + n] : a.usage0[(size_t)b * N + n];
            sPp[n] = (t > 0) ? a.rec_p[(bt - 1) * N + n] : a.prec0[(size_t)b * N + n];
            sDWW[n] = 0.f; sDPp[n] = 0.f;
        }
        for (int i = tid; i < RN; i += DT) {
            sCR[i] = a.rec_cr[bt * RN + i];
            sRWp[i] = (t > 0) ? a.rec_rw[(bt - 1) * RN + i] : a.rw0[(size_t)b * RN + i];
            sDRWp[i] = 0.f;
        }
        for (int i = tid; i < RWd; i += DT) sDKR[i] = 0.f;
        for (int i = tid; i < W; i += DT) { sDKW[i] = 0.f; sDE[i] = 0.f; sDV[i] = 0.f; }
        if (tid < 32) sSC[tid] = 0.f;
        if (tid < d.OP) {                         // B1: output clip + linear
            float g = 0.f;
            if (tid < d.O) {
                const float pre = a.rec_ypre[bt * d.O + tid];
                g = (clipv <= 0.f || fabsf(pre) < clipv) ? a.dout[bt * d.O + tid] : 0.f;
            }
            sSC[32 + tid] = g;
            a.dypre[bt * d.OP + tid] = g;
        }
        __syncthreads();
        for (int k = tid; k < d.Ky; k += DT) {
            float s = 0.f;
            for (int o = 0; o < d.O; ++o) s += a.Wy[(size_t)k * d.OP + o] * sSC[32 + o];
            if (k < hid) sDHC[k] = sGZ[RWd + k] + s;          // carried d(clipped h) + this step's output path
            else sDR[k - hid] = sGZ[k - hid] + s;             // carried d(reads) + output path
        }
        // per-head key norms of the read keys and the write key
        if (tid <= R) {
            const float* kp = (tid < R) ? sI + d.oKr + tid * W : sI + d.oKw;
            float ss = 0.f;
            for (int w = 0; w < W; ++w) ss += kp[w] * kp[w];
            sSC[tid] = sqrtf(ss + EPS);                        // sSC[0..R-1] = |kr_i|, sSC[R] = |kw|
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
                    sSIMr[i * N + n] = dot / (sSC[i] * nm + EPS);
                }
            }
        }
        __syncthreads();
        // ------------------------------------------------------------ B3: read-weight mix, read-content softmax (wave i = head i)
        if (wave < R) {
            const int i = wave;
            const float* rm = sI + d.oRm + i * 3;              // [backward, forward, content] (access.py:283-289)
            float p0 = 0.f, p1 = 0.f, p2 = 0.f, s1 = 0.f;
            for (int n = lane; n < N; n += 64) {
                const float g = sDRW[i * N + n];
                p0 += g * gBV[i * N + n]; p1 += g * gFV[i * N + n]; p2 += g * sCR[i * N + n];
                s1 += sCR[i * N + n] * (rm[2] * g);
            }
            p0 = wave_sum(p0); p1 = wave_sum(p1); p2 = wave_sum(p2); s1 = wave_sum(s1);
            const float br = sI[d.oBr + i];
            float dbeta = 0.f;
            for (int n = lane; n < N; n += 64) {
                const float g = sDRW[i * N + n];
                const float dscore = sCR[i * N + n] * (rm[2] * g - s1);
                dbeta += dscore * sSIMr[i * N + n];
                sDSIM[i * N + n] = dscore * br;
                sDF[i * N + n] = rm[1] * g;
                sDB[i * N + n] = rm[0] * g;
            }
            dbeta = wave_sum(dbeta);
            if (lane == 0) {
                const float dotp = rm[0] * p0 + rm[1] * p1 + rm[2] * p2;
                sDX[d.oRm + i * 3 + 0] = rm[0] * (p0 - dotp);
                sDX[d.oRm + i * 3 + 1] = rm[1] * (p1 - dotp);
                sDX[d.oRm + i * 3 + 2] = rm[2] * (p2 - dotp);
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
                            const float D = sSC[i] * nm + EPS;
                            const float dot = sSIMr[i * N + n] * D;
                            const float ddot = dsim / D;
                            const float dD = -dsim * dot / (D * D);
                            dnm += dD * sSC[i];
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
                    if (gl == 0) atomicAdd(&sSC[8 + i], accNk[i]);      // d|kr_i|
                }
            }
        }
        __syncthreads();
        for (int idx = tid; idx < RWd; idx += DT) {
            const int i = idx / W;
            sDX[d.oKr + idx] = sDKR[idx] + sSC[8 + i] * sI[d.oKr + idx] / sSC[i];
        }
        // ------------------------------------------------------------ B5: link pass (gL, L_t, L_{t-1})
        for (int cb = 0; cb < N; cb += 256) {
            const int b0 = cb + lane * 4;
            const bool colok = b0 < N;
            f32x4 wwb = {0.f, 0.f, 0.f, 0.f}, ppb = wwb, rwpb[4], dbb[4], colRW[4], colWW = wwb, colP = wwb;
#pragma unroll
            for (int i = 0; i < 4; ++i) { rwpb[i] = wwb; dbb[i] = wwb; colRW[i] = wwb; }
            if (colok) {
                wwb = *reinterpret_cast<const f32x4*>(sWW + b0);
                ppb = *reinterpret_cast<const f32x4*>(sPp + b0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < R) { rwpb[i] = *reinterpret_cast<const f32x4*>(sRWp + i * N + b0); dbb[i] = *reinterpret_cast<const f32x4*>(sDB + i * N + b0); }
            }
            for (int r = wave; r < N; r += DW) {
                const float wwa = sWW[r];
                float rowRW[4] = {0.f, 0.f, 0.f, 0.f}, rowWW = 0.f;
                if (colok) {
                    f32x4 g = *reinterpret_cast<const f32x4*>(gL + (size_t)r * N + b0);
                    const f32x4 lt = *reinterpret_cast<const f32x4*>(Lt + (size_t)r * N + b0);
                    const f32x4 lp = *reinterpret_cast<const f32x4*>(Lp + (size_t)r * N + b0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (i < R) {
                            const float dfa = sDF[i * N + r], rwpa = sRWp[i * N + r];
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
                if (lane == 0) atomicAdd(&sDWW[r], s);
            }
            if (colok) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (i < R) atomicAdd(&sDRWp[i * N + b0 + e], colRW[i][e]);
                    atomicAdd(&sDWW[b0 + e], colWW[e]);
                    atomicAdd(&sDPp[b0 + e], colP[e]);
                }
            }
        }
        __syncthreads();
        // ------------------------------------------------------------ B6: precedence (wave 0 computes the two scalars)
        if (wave == 0) {
            float sw = 0.f, t1 = 0.f;
            for (int n = lane; n < N; n += 64) { sw += sWW[n]; t1 += sgP[n] * sPp[n]; }
            sw = wave_sum(sw); t1 = wave_sum(t1);
            if (lane == 0) { sSC[16] = sw; sSC[17] = t1; }
        }
        __syncthreads();
        for (int n = tid; n < N; n += DT) {
            sDPp[n] += (1.0f - sSC[16]) * sgP[n];
            sDWW[n] += sgP[n] - sSC[17];
        }
        __syncthreads();
        // ------------------------------------------------------------ B7: write backward over (gM, M_{t-1}); write-key similarities
        {
            f32x4 accE = {0.f, 0.f, 0.f, 0.f}, accV = accE;
            for (int n = grp; n < N; n += ngrp) {
                f32x4 mp = {0.f, 0.f, 0.f, 0.f};
                float t1 = 0.f, dot = 0.f;
                const float wwn = sWW[n];
                if (gl < W4) {
                    mp = reinterpret_cast<const f32x4*>(Mp + (size_t)n * W)[gl];
                    f32x4 g = reinterpret_cast<f32x4*>(gM + (size_t)n * W)[gl];
                    const float* ep = sI + d.oE + gl * 4;
                    const float* vp = sI + d.oV + gl * 4;
                    const float* kp = sI + d.oKw + gl * 4;
                    f32x4 gn;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        t1 += g[e] * (vp[e] - mp[e] * ep[e]);
                        accE[e] -= g[e] * mp[e] * wwn;
                        accV[e] += g[e] * wwn;
                        gn[e] = g[e] * (1.0f - wwn * ep[e]);
                        dot += kp[e] * mp[e];
                    }
                    reinterpret_cast<f32x4*>(gM + (size_t)n * W)[gl] = gn;        // now d(M_{t-1}) (content part added in B10)
                }
                float nsq = mp[0] * mp[0] + mp[1] * mp[1] + mp[2] * mp[2] + mp[3] * mp[3];
                for (int o = LPR >> 1; o > 0; o >>= 1) {
                    t1 += __shfl_xor(t1, o, 64); dot += __shfl_xor(dot, o, 64); nsq += __shfl_xor(nsq, o, 64);
                }
                if (gl == 0) {
                    const float nm = sqrtf(nsq + EPS);
                    sDWW[n] += t1;
                    sNMw[n] = nm;
                    sSIMw[n] = dot / (sSC[R] * nm + EPS);
                }
            }
            if (gl < W4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { atomicAdd(&sDE[gl * 4 + e], accE[e]); atomicAdd(&sDV[gl * 4 + e], accV[e]); }
            }
        }
        __syncthreads();
        // ------------------------------------------------------------ B8: write-weight mix (access.py:252-257)
        {
            const float ga = sI[d.oAg], gw = sI[d.oWg];
            for (int n = tid; n < N; n += DT) {
                const float dww = sDWW[n];
                sDA[n] = gw * ga * dww;
                sDCW[n] = gw * (1.0f - ga) * dww;
            }
            if (wave == 0) {
                float dgw = 0.f, dga = 0.f;
                for (int n = lane; n < N; n += 64) {
                    const float dww = sDWW[n];
                    dgw += dww * (ga * sAL[n] + (1.0f - ga) * sCW[n]);
                    dga += gw * dww * (sAL[n] - sCW[n]);
                }
                dgw = wave_sum(dgw); dga = wave_sum(dga);
                if (lane == 0) { sDX[d.oWg] = dgw * gw * (1.0f - gw); sDX[d.oAg] = dga * ga * (1.0f - ga); }
            }
        }
        __syncthreads();
        // ------------------------------------------------------------ B9: allocation backward (rank form) + B10a: write-content softmax
        if (tid < nslA * N) {
            const int n = tid % N, sl = tid / N;
            const float nun = 1.0f - (EPS + (1.0f - EPS) * sU[n]);
            const int m0 = sl * mperA, m1 = min(N, m0 + mperA);
            float prod = 1.f, gsum = 0.f;
            for (int m = m0; m < m1; ++m) {
                const float num = 1.0f - (EPS + (1.0f - EPS) * sU[m]);
                const bool before = (num > nun) || (num == nun && m < n);      // m precedes n in the usage sort
                const bool after = (nun > num) || (nun == num && n < m);
                prod *= before ? (1.0f - num) : 1.0f;
                gsum += after ? sDA[m] * sAL[m] : 0.f;
            }
            sPart[sl * N + n] = prod;
            sPart[(nslA + sl) * N + n] = gsum;
        }
        if (wave == DW - 1) {
            float s1 = 0.f;
            for (int n = lane; n < N; n += 64) s1 += sCW[n] * sDCW[n];
            s1 = wave_sum(s1);
            if (lane == 0) sSC[18] = s1;
        }
        __syncthreads();
        for (int n = tid; n < N; n += DT) {
            float prod = 1.f, gsum = 0.f;
            for (int sl = 0; sl < nslA; ++sl) { prod *= sPart[sl * N + n]; gsum += sPart[(nslA + sl) * N + n]; }
            const float ut = 1.0f - (1.0f - (EPS + (1.0f - EPS) * sU[n]));       // sorted_usage = 1 - sorted_nonusage
            const float dut = -sDA[n] * prod + gsum / ut;
            sgUn[n] = sgU[n] + (1.0f - EPS) * dut;                              // total d(usage_t)
            sDCW[n] = sCW[n] * (sDCW[n] - sSC[18]);                             // d(score) of the write-content softmax
        }
        __syncthreads();
        if (wave == 0) {
            float dbeta = 0.f;
            for (int n = lane; n < N; n += 64) dbeta += sDCW[n] * sSIMw[n];
            dbeta = wave_sum(dbeta);
            const float bw = sI[d.oBw];
            if (lane == 0) sDX[d.oBw] = dbeta * (1.0f - expf(-bw));
        }
        // ------------------------------------------------------------ B10b: pass over M_{t-1}: content part of d(M_{t-1}), d(write key)
        {
            f32x4 accK = {0.f, 0.f, 0.f, 0.f};
            float accNk = 0.f;
            const float bw = sI[d.oBw], nk = sSC[R];
            for (int n = grp; n < N; n += ngrp) {
                if (gl < W4) {
                    const f32x4 mp = reinterpret_cast<const f32x4*>(Mp + (size_t)n * W)[gl];
                    f32x4 g = reinterpret_cast<f32x4*>(gM + (size_t)n * W)[gl];
                    const float nm = sNMw[n];
                    const float dsim = sDCW[n] * bw;
                    const float D = nk * nm + EPS;
                    const float dot = sSIMw[n] * D;
                    const float ddot = dsim / D;
                    const float dD = -dsim * dot / (D * D);
                    if (gl == 0) accNk += dD * nm;
                    const float* kp = sI + d.oKw + gl * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] += ddot * kp[e] + (dD * nk / nm) * mp[e];
                    accK += ddot * mp;
                    reinterpret_cast<f32x4*>(gM + (size_t)n * W)[gl] = g;
                }
            }
            if (gl < W4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(&sDKW[gl * 4 + e], accK[e]);
                if (gl == 0) atomicAdd(&sSC[19], accNk);
            }
        }
        // ------------------------------------------------------------ B11: usage backward (addressing.py:342-374)
        for (int n = tid; n < N; n += DT) {
            const float g = sgUn[n];
            const float u1 = sUp[n] + (1.0f - sUp[n]) * sWWp[n];                // write weights: stop_gradient
            float phi = 1.f;
            for (int i = 0; i < R; ++i) phi *= (1.0f - sI[d.oF + i] * sRWp[i * N + n]);
            const float dphi = g * u1;
            sgU[n] = g * phi * (1.0f - sWWp[n]);                                // carried d(usage_{t-1})
            for (int i = 0; i < R; ++i) {
                float oth = 1.f;
                for (int i2 = 0; i2 < R; ++i2) if (i2 != i) oth *= (1.0f - sI[d.oF + i2] * sRWp[i2 * N + n]);
                sDRWp[i * N + n] += dphi * (-sI[d.oF + i]) * oth;
                sDSIM[i * N + n] = dphi * (-sRWp[i * N + n]) * oth;             // reuse: per-slot term of d(free_gate_i)
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
        for (int w = tid; w < W; w += DT) {                                     // remaining interface gradients
            const float e = sI[d.oE + w];
            sDX[d.oE + w] = sDE[w] * e * (1.0f - e);
            sDX[d.oV + w] = sDV[w];
            sDX[d.oKw + w] = sDKW[w] + sSC[19] * sI[d.oKw + w] / sSC[R];
        }
        for (int i = tid; i < RN; i += DT) sgRW[i] = sDRWp[i];                  // carried d(read weights_{t-1})
        for (int n = tid; n < N; n += DT) sgP[n] = sDPp[n];                     // carried d(precedence_{t-1})
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
        for (int i = tid0; i < N; i += DT) { cy[i] = sgP[i]; cy[N + i] = sgU[i]; }
        for (int i = tid0; i < RN; i += DT) cy[2 * N + i] = sgRW[i];
        for (int i = tid0; i < a.ldkT; i += DT) cy[2 * N + RN + i] = sGZ[i];
        for (int i = tid0; i < hid; i += DT) cy[2 * N + RN + a.ldkT + i] = sgC[i];
    }
}

extern "C" int ntk_dnc_seq_bwd(int B, int S, int N, int W, int R, int Wn, int hid, int O, float clip_value,
                               const float* WrT, int ldkT, const float* WiT, int ldhT, const float* Wy,
                               const float* mem0, const float* link0, const float* usage0, const float* rw0,
                               const float* ww0, const float* prec0, const float* hc0,
                               const float* rec_gates, const float* rec_c, const float* rec_ifc, const float* rec_u,
                               const float* rec_ww, const float* rec_rw, const float* rec_cw, const float* rec_cr,
                               const float* rec_al, const float* rec_p, const float* rec_fwd, const float* rec_bwd,
                               const float* rec_M, const float* rec_L, const float* rec_ypre,
                               const float* dout, float* gM, float* gL, float* dgates, float* dxi, float* dypre,
                               float* gcarry, int carry_in, void* stream) {
    DncBwdArgs a;
    dnc_fill_dims(a.d, B, S, N, W, R, Wn, hid, O, clip_value);
    NTK_REQUIRE(B > 0 && S > 0, NTK_ERR_BAD_SHAPE, "ntk_dnc_seq_bwd: B=%d S=%d", B, S);
    NTK_REQUIRE(Wn >= 1 && Wn <= 4, NTK_ERR_UNSUPPORTED, "ntk_dnc_seq_bwd: num_writes=%d (the BPTT kernels implement 1..4 write heads)", Wn);
    NTK_REQUIRE(N >= 4 && (N % 4) == 0 && N <= DT && W >= 4 && (W % 4) == 0 && W <= 256 && R >= 1 && R <= 4,
                NTK_ERR_UNSUPPORTED, "ntk_dnc_seq_bwd: N=%d W=%d R=%d unsupported", N, W, R);
    NTK_REQUIRE(hid >= 4 && (hid % 4) == 0 && hid <= DT && R * W <= DT && O >= 1 && O <= 16, NTK_ERR_UNSUPPORTED,
                "ntk_dnc_seq_bwd: hidden=%d (multiple of 4) output=%d", hid, O);
    NTK_REQUIRE(ldkT >= a.d.K && (ldkT % 4) == 0 && ldhT >= hid && (ldhT % 4) == 0, NTK_ERR_BAD_SHAPE,
                "ntk_dnc_seq_bwd: ldkT=%d ldhT=%d", ldkT, ldhT);
    NTK_REQUIRE(WrT && WiT && Wy && mem0 && link0 && usage0 && rw0 && ww0 && prec0 && hc0 && rec_gates && rec_c && rec_ifc &&
                    rec_u && rec_ww && rec_rw && rec_cw && rec_cr && rec_al && rec_p && rec_fwd && rec_bwd && rec_M && rec_L &&
                    rec_ypre && dout && gM && gL && dgates && dxi && dypre,
                NTK_ERR_BAD_PTR, "ntk_dnc_seq_bwd: null pointer");
    NTK_REQUIRE(ntk_aligned16(WrT) && ntk_aligned16(WiT) && ntk_aligned16(rec_gates) && ntk_aligned16(rec_M) &&
                    ntk_aligned16(rec_L) && ntk_aligned16(gM) && ntk_aligned16(gL) && ntk_aligned16(dgates) &&
                    ntk_aligned16(mem0) && ntk_aligned16(link0),
                NTK_ERR_BAD_PTR, "ntk_dnc_seq_bwd: 16-byte alignment");
    a.WrT = WrT; a.ldkT = ldkT; a.WiT = WiT; a.ldhT = ldhT; a.Wy = Wy;
    a.mem0 = mem0; a.link0 = link0; a.usage0 = usage0; a.rw0 = rw0; a.ww0 = ww0; a.prec0 = prec0; a.hc0 = hc0;
    a.rec_gates = rec_gates; a.rec_c = rec_c; a.rec_ifc = rec_ifc; a.rec_u = rec_u; a.rec_ww = rec_ww; a.rec_rw = rec_rw;
    a.rec_cw = rec_cw; a.rec_cr = rec_cr; a.rec_al = rec_al; a.rec_p = rec_p; a.rec_fwd = rec_fwd; a.rec_bwd = rec_bwd;
    a.rec_M = rec_M; a.rec_L = rec_L; a.rec_ypre = rec_ypre; a.dout = dout; a.gM = gM; a.gL = gL;
    a.dgates = dgates; a.dxi = dxi; a.dypre = dypre; a.gcarry = gcarry; a.carry_in = carry_in;
    if (Wn > 1) return dnc_seq_bwd_mw_launch(a, (hipStream_t)stream);      // general kernel (dnc_seq_bwd_mw.hip)
    DncBwdLds L;
    dnc_bwd_lds(a.d, ldkT, ldhT, L);
    const size_t lds_bytes = (size_t)L.total * sizeof(float);
    NTK_REQUIRE(lds_bytes <= 160 * 1024, NTK_ERR_UNSUPPORTED, "ntk_dnc_seq_bwd: needs %zu B of LDS (> 160 KiB)", lds_bytes);
    {
        static NtkLdsAttrCache lds_cache;
        const void* const ks[] = {(const void*)dnc_seq_bwd_kernel};
        const int rc_lds = ntk_raise_lds_limit(lds_cache, ks, 1, "ntk_dnc_seq_bwd");
        if (rc_lds != NTK_OK) return rc_lds;
    }
    dnc_seq_bwd_kernel<<<B, DT, lds_bytes, (hipStream_t)stream>>>(a, L);
    NTK_CHECK_LAUNCH("ntk_dnc_seq_bwd");
    return NTK_OK;
}
