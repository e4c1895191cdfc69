// DNC core sequence forward, CLUSTER form: k workgroups (512 threads, one per CU) per sequence.
// Same arithmetic as dnc_seq_fwd.hip (dnc/dnc.py:84-127, dnc/access.py:113-303, dnc/addressing.py) -- see that file
// for the step's phases P1..P8 -- re-partitioned so that a sequence's strictly serial step runs on k CUs:
//
//   partitioned  LSTM gate product (hidden units split k ways: each workgroup streams 1/k of Wr from its XCD's L2),
//                interface product (by the same hidden-unit split: partial sums over the own units),
//                temporal link (N/k rows per workgroup, LDS RESIDENT for the whole sequence: the N x N matrix never
//                touches L2 / HBM in inference; in training its rows are written once per step as BPTT records),
//                the BPTT records (each workgroup writes its own rows / slots / units);
//   replicated   usage, allocation, write / read weights, precedence and the N x W memory (LDS resident, every
//                workgroup updates its private copy from bit-identical inputs);
//   exchanged    twice per step through the mailbox (dnc_cluster.h): (0) h slice + interface partials,
//                (1) forward directional reads of the own link rows + the partial backward directional reads
//                (column sums over the own rows), summed in a fixed workgroup order -> bitwise reproducible.
//
// The two directional reads (addressing.py:155-181: tf.matmul of the read weights with the link and its transpose)
// run on v_mfma_f32_4x4x1_16b_f32: sixteen independent 4 x 4 outer-product accumulators per instruction, so the
// R <= 4 read heads need no padding and the column / row sums need no cross-lane reductions:
//   backward  D[head][col]  += rw_prev[head][row] * L[row][col]      block = 4 columns, one link row per instruction
//   forward   D[row][head]  += L[row][col] * rw_prev[head][col]      block = 4 rows,    one link column per block
// (exact fp32 FMA chains, k-ordered: same numerics as a scalar loop).
#include "dnc_cluster.h"

namespace {

struct DncClFwdArgs {
    DncDims d;
    DncClusterGeom g;
    const float* xproj; const float* Wr; const float* Wi; const float* Wy;
    float* mem; float* link; float* usage; float* rw; float* ww; float* prec; float* reads; float* hc; float* out;
    float* rec_z; float* rec_gates; float* rec_c; float* rec_hc; float* rec_yin; float* rec_ifc; float* rec_u;
    float* rec_ww; float* rec_rw; float* rec_cw; float* rec_cr; float* rec_al; float* rec_p; float* rec_fwd;
    float* rec_bwd; float* rec_M; float* rec_L; float* rec_ypre;
    float* mbox; unsigned* flags; unsigned* err;
};

struct DncClFwdLds {
    int part, M, L, Z, C, I, U, NU, RW, WW, P, CW, CR, SC, total;
};

int cl_imax(int a, int b) { return a > b ? a : b; }

void dnc_cl_fwd_lds(const DncDims& d, const DncClusterGeom& g, DncClFwdLds& L) {
    const int N = d.N, RWd = d.R * d.W, W4 = d.W / 4;
    const int ksl = cl_imax(1, CT / cl_imax(1, g.upk));
    const int icg = d.IP / 4, nslI = cl_imax(1, CT / icg);
    const int nslA = cl_imax(1, CT / N);
    const int strips = N / 64, NRp = ((g.NR + 31) / 32) * 32;
    const int nslR = cl_imax(1, CT / (d.R * W4));
    int part = ksl * g.upk * 4;
    part = cl_imax(part, nslI * d.IP);
    part = cl_imax(part, nslA * N);
    part = cl_imax(part, strips * 2 * NRp * 4);
    part = cl_imax(part, nslR * RWd);
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    L.part = take(part);
    L.M = take(N * d.W);
    L.L = take(g.NR * N);
    L.Z = take(d.K); L.C = take(g.upk); L.I = take(d.IP);
    L.U = take(N); L.NU = take(N); L.RW = take(d.R * N); L.WW = take(N); L.P = take(N); L.CW = take(N); L.CR = take(d.R * N);
    L.SC = take(64);
    L.total = o;
}

__device__ __forceinline__ void cl_softmax_row(float* r, int N, int lane) {      // one wave, in place
    float mx = -INFINITY;
    for (int n = lane; n < N; n += 64) mx = fmaxf(mx, r[n]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int n = lane; n < N; n += 64) { const float e = expf(r[n] - mx); r[n] = e; s += e; }
    s = wave_sum(s);
    for (int n = lane; n < N; n += 64) r[n] = r[n] / s;
}

__global__ __launch_bounds__(CT) void dnc_cluster_fwd_kernel(DncClFwdArgs a, DncClFwdLds L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DncDims& d = a.d;
    const int tid0 = threadIdx.x;
    const int k = a.g.k, NR = a.g.NR, upk = a.g.upk;
    int b, g;
    if (a.g.xcd_local) {                    // the k members of a sequence share blockIdx % 8 (speed only, never correctness)
        const int x = blockIdx.x & 7, s = blockIdx.x >> 3;
        b = x + 8 * (s / k);
        g = s % k;
    } else {
        b = blockIdx.x / k;
        g = blockIdx.x % k;
    }
    const int N = d.N, W = d.W, R = d.R, hid = d.hid, S = d.S, K = d.K, IP = d.IP, RWd = R * W, N4 = N >> 2, W4 = W >> 2;
    const float clipv = d.clip;
    const float EPS = 1e-6f;
    const int row0 = g * NR;                                  // first link row / memory row / slot owned by this workgroup
    const int u0 = min(hid, g * upk), u1 = min(hid, u0 + upk), nU = u1 - u0;
    const int upkp = dnc_cluster_align4(upk);

    float* sPart = smem + L.part;
    float* sM = smem + L.M;
    float* sL = smem + L.L;
    float* sZ = smem + L.Z;      // [reads_prev ; h_prev]
    float* sC = smem + L.C;      // cell of the own units
    float* sI = smem + L.I;
    float* sU = smem + L.U;
    float* sNU = smem + L.NU;    // nonusage 1 - (eps + (1 - eps) u): the allocation order is decided on these stored values
    float* sRW = smem + L.RW;
    float* sWW = smem + L.WW;
    float* sP = smem + L.P;
    float* sCW = smem + L.CW;
    float* sCR = smem + L.CR;
    float* sSC = smem + L.SC;
    int* sAbort = reinterpret_cast<int*>(sSC + 32);
    f32x4* sPart4 = reinterpret_cast<f32x4*>(sPart);
    f32x4* sM4 = reinterpret_cast<f32x4*>(sM);
    f32x4* sL4 = reinterpret_cast<f32x4*>(sL);

    // mailbox of this sequence
    const int slot0 = a.g.slot0, slot1 = a.g.slot1;
    float* mb0 = a.mbox + (size_t)b * 2 * k * ((size_t)slot0 + slot1);        // [parity][g][slot0]
    float* mb1 = mb0 + (size_t)2 * k * slot0;                                 // [parity][g][slot1]
    unsigned* fl0 = a.flags + (size_t)b * 2 * k;
    unsigned* fl1 = fl0 + k;
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();

    int LPR = 1;
    while (LPR * 4 < W) LPR <<= 1;
    const int ngrp = CT / LPR;

    // ---- load state (memory replicated, link rows of this workgroup, per-slot vectors replicated)
    if (tid0 == 0) *sAbort = 0;
    {
        const f32x4* gM4 = reinterpret_cast<const f32x4*>(a.mem + (size_t)b * N * W);
        for (int i = tid0; i < N * W4; i += CT) sM4[i] = gM4[i];
        const f32x4* gL4 = reinterpret_cast<const f32x4*>(a.link + ((size_t)b * N + row0) * N);
        for (int i = tid0; i < NR * N4; i += CT) {
            const int r = i / N4, q = i - r * N4;
            sL4[r * N4 + (q ^ (r & 7))] = gL4[i];
        }
    }
    for (int i = tid0; i < N; i += CT) {
        sU[i] = a.usage[(size_t)b * N + i];
        sWW[i] = a.ww[(size_t)b * N + i];
        sP[i] = a.prec[(size_t)b * N + i];
    }
    for (int i = tid0; i < R * N; i += CT) sRW[i] = a.rw[(size_t)b * R * N + i];
    for (int i = tid0; i < RWd; i += CT) sZ[i] = a.reads[(size_t)b * RWd + i];
    for (int i = tid0; i < hid; i += CT) sZ[RWd + i] = a.hc[(size_t)b * 2 * hid + i];
    for (int i = tid0; i < nU; i += CT) sC[i] = a.hc[(size_t)b * 2 * hid + hid + u0 + i];
    __syncthreads();

    const f32x4* Wr4 = reinterpret_cast<const f32x4*>(a.Wr);
    const f32x4* Wi4 = reinterpret_cast<const f32x4*>(a.Wi);
    const bool rec = a.rec_z != nullptr;

    // work decomposition (constant over the sequence)
    const int ksl = max(1, min(K, CT / max(1, nU))), kperG = (K + ksl - 1) / ksl;
    const int icg = IP >> 2, nslI = max(1, CT / icg), uperI = (max(nU, 1) + nslI - 1) / nslI;
    const int nslA = max(1, CT / N), mperA = (N + nslA - 1) / nslA;
    const int strips = N >> 6, NRp = ((NR + 31) >> 5) << 5;
    const int nRW4 = R * W4, nslR = max(1, CT / nRW4), nperR = (N + nslR - 1) / nslR;

    for (int t = 0; t < S; ++t) {
        int tid_op = tid0;
        asm volatile("" : "+v"(tid_op));       // keep per-thread index math inside the step (no hoist + spill)
        const int tid = tid_op, lane = tid & 63, wave = tid >> 6;
        const size_t bt = (size_t)b * S + t;
        const unsigned epoch = (unsigned)t + 1u;
        const int par = t & 1;

        // ------------------------------------------------------------ P1: LSTM gates of the own hidden units
        f32x4 xg = {0.f, 0.f, 0.f, 0.f};
        if (tid < nU) xg = reinterpret_cast<const f32x4*>(a.xproj)[bt * hid + u0 + tid] + Wr4[(size_t)K * hid + u0 + tid];
        if (rec && g == 0) for (int i = tid; i < d.ldz; i += CT) a.rec_z[bt * d.ldz + i] = (i < K) ? sZ[i] : (i == K ? 1.f : 0.f);
        if (tid < ksl * nU) {
            const int j = tid % nU, ks = tid / nU;
            const int k0 = ks * kperG, k1 = min(K, k0 + kperG);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (k0 < k1) acc = ntk_stream_matvec<4>(Wr4 + u0 + j, hid, sZ, k0, k1, K - 1);
            sPart4[ks * nU + j] = acc;
        }
        __syncthreads();
        if (tid < nU) {
            f32x4 gsum = xg;
            for (int ks = 0; ks < ksl; ++ks) gsum += sPart4[ks * nU + tid];
            const float gi = dnc_sigmoid(gsum[0]), gj = tanhf(gsum[1]);
            const float gf = dnc_sigmoid(gsum[2] + 1.0f);            // snt.LSTM forget_bias = 1.0
            const float go = dnc_sigmoid(gsum[3]);
            const float c2 = gf * sC[tid] + gi * gj;
            const float h2 = tanhf(c2) * go;
            sC[tid] = dnc_clip(c2, clipv);                           // dnc.py:112-113
            sZ[RWd + u0 + tid] = dnc_clip(h2, clipv);
            if (rec) {
                f32x4 ga = {gi, gj, gf, go};
                reinterpret_cast<f32x4*>(a.rec_gates)[bt * hid + u0 + tid] = ga;
                a.rec_c[bt * hid + u0 + tid] = c2;
            }
        }
        __syncthreads();
        // ------------------------------------------------------------ P2: interface partial sums over the own units
        if (tid < nslI * icg) {
            const int cg = tid % icg, us = tid / icg;
            const int ua = u0 + us * uperI, ub = min(u1, ua + uperI);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const f32x4* wp = Wi4 + (size_t)ua * icg + cg;
#pragma unroll 4
            for (int u = ua; u < ub; ++u, wp += icg) acc += sZ[RWd + u] * (*wp);
            sPart4[us * icg + cg] = acc;
        }
        __syncthreads();
        {   // publish exchange 0: [h of the own units | interface partial]
            float* slot = mb0 + ((size_t)par * k + g) * slot0;
            if (tid < nU) cl_store(slot + tid, sZ[RWd + u0 + tid]);
            for (int c = tid; c < IP; c += CT) {
                float v = 0.f;
                for (int us = 0; us < nslI; ++us) v += sPart[us * IP + c];
                cl_store(slot + upkp + c, v);
            }
            cl_publish(fl0 + g, epoch, tid);
        }
        if (!cl_wait(fl0, epoch, k, a.err, sAbort, t_start, tid)) return;
        {   // consume exchange 0: full h, activated interface
            const float* base = mb0 + (size_t)par * k * slot0;
            for (int u = tid; u < hid; u += CT) {
                const int gg = u / upk;
                sZ[RWd + u] = cl_load(base + (size_t)gg * slot0 + (u - gg * upk));
            }
            for (int c = tid; c < IP; c += CT) {
                float pv[8];
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) pv[gg] = (gg < k) ? cl_load(base + (size_t)gg * slot0 + upkp + c) : 0.f;
                float v = a.Wi[(size_t)hid * IP + c];
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) if (gg < k) v += pv[gg];
                float r = v;
                if (c >= d.oE && c < d.oRm) r = dnc_sigmoid(v);                      // erase, free, alloc, write gates
                else if ((c >= d.oBw && c < d.oKr) || (c >= d.oBr && c < d.I)) r = dnc_softplus(v);   // strengths
                sI[c] = r;
            }
        }
        __syncthreads();
        if (rec && g == 0) {
            for (int i = tid; i < d.ldh; i += CT) {
                const float v = (i < hid) ? sZ[RWd + i] : (i == hid ? 1.f : 0.f);
                a.rec_hc[bt * d.ldh + i] = v;
                if (i < hid) a.rec_yin[bt * d.ldy + i] = v;
            }
        }
        if (tid < R) {                                                               // read_mode softmax (access.py:186-187)
            float* rm = sI + d.oRm + tid * 3;
            const float mx = fmaxf(rm[0], fmaxf(rm[1], rm[2]));
            const float e0 = expf(rm[0] - mx), e1 = expf(rm[1] - mx), e2 = expf(rm[2] - mx);
            const float s = e0 + e1 + e2;
            rm[0] = e0 / s; rm[1] = e1 / s; rm[2] = e2 / s;
        }
        // ------------------------------------------------------------ P3: usage (addressing.py:342-374), op by op
        {
#pragma clang fp contract(off)
            for (int n = tid; n < N; n += CT) {
                float pw = 1.f;
                pw *= (1.0f - sWW[n]);
                float u = sU[n];
                u = u + (1.0f - u) * (1.0f - pw);
                float phi = 1.f;
                for (int i = 0; i < R; ++i) phi *= (1.0f - sI[d.oF + i] * sRW[i * N + n]);
                u *= phi;
                sU[n] = u;
                sNU[n] = 1.0f - (EPS + (1.0f - EPS) * u);
            }
        }
        // ------------------------------------------------------------ P4: write content weights on M_{t-1}
        {
            const int grp = tid / LPR, gl = tid % LPR;
            f32x4 kw = {0.f, 0.f, 0.f, 0.f};
            if (gl < W4) { const float* kp = sI + d.oKw + gl * 4; kw = f32x4{kp[0], kp[1], kp[2], kp[3]}; }
            const float ksq = group_sum_rt(kw[0] * kw[0] + kw[1] * kw[1] + kw[2] * kw[2] + kw[3] * kw[3], LPR);
            const float kn = sqrtf(ksq + EPS), bw = sI[d.oBw];
            for (int n = grp; n < N; n += ngrp) {
                f32x4 m = {0.f, 0.f, 0.f, 0.f};
                if (gl < W4) m = sM4[n * W4 + gl];
                float nsq = m[0] * m[0] + m[1] * m[1] + m[2] * m[2] + m[3] * m[3];
                float dot = kw[0] * m[0] + kw[1] * m[1] + kw[2] * m[2] + kw[3] * m[3];
                nsq = group_sum_rt(nsq, LPR);
                dot = group_sum_rt(dot, LPR);
                if (gl == 0) sCW[n] = (dot / (kn * sqrtf(nsq + EPS) + EPS)) * bw;
            }
        }
        __syncthreads();
        if (wave == 0) cl_softmax_row(sCW, N, lane);
        if (rec) {                                           // the own slots of the per-slot records (sU final since the barrier)
            for (int n = tid; n < NR; n += CT) a.rec_u[bt * N + row0 + n] = sU[row0 + n];
            if (g == 0) for (int c = tid; c < IP; c += CT) a.rec_ifc[bt * IP + c] = sI[c];
        }
        // ------------------------------------------------------------ P5: allocation (rank form) + write weights
        if (tid < nslA * N) {
            const int n = tid % N, sl = tid / N;
            const float nun = sNU[n];
            const int m0 = sl * mperA, m1 = min(N, m0 + mperA);
            float prod = 1.f;
            for (int m = m0; m < m1; ++m) {
                const float num = sNU[m];
                const bool before = (num > nun) || (num == nun && m < n);
                prod *= before ? (1.0f - num) : 1.0f;
            }
            sPart[sl * N + n] = prod;
        }
        __syncthreads();
        {
#pragma clang fp contract(off)
            const float ag = sI[d.oAg], wg = sI[d.oWg];
            for (int n = tid; n < N; n += CT) {
                float prod = 1.f;
                for (int sl = 0; sl < nslA; ++sl) prod *= sPart[sl * N + n];
                const float al = sNU[n] * prod;
                const float cw = sCW[n];
                sWW[n] = wg * (ag * al + (1.0f - ag) * cw);
                if (rec && n >= row0 && n < row0 + NR) { a.rec_al[bt * N + n] = al; a.rec_cw[bt * N + n] = cw; }
            }
        }
        __syncthreads();
        // ------------------------------------------------------------ P6: erase + write on M (every row), read-key scores on M_t
        {
            const int grp = tid / LPR, gl = tid % LPR;
            f32x4 ev = {0.f, 0.f, 0.f, 0.f}, vv = ev, kr[4];
            float krn[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) kr[i] = ev;
            if (gl < W4) {
                ev = *reinterpret_cast<const f32x4*>(sI + d.oE + gl * 4);
                vv = *reinterpret_cast<const f32x4*>(sI + d.oV + gl * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < R) { const float* kp = sI + d.oKr + i * W + gl * 4; kr[i] = f32x4{kp[0], kp[1], kp[2], kp[3]}; }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                krn[i] = sqrtf(group_sum_rt(kr[i][0] * kr[i][0] + kr[i][1] * kr[i][1] + kr[i][2] * kr[i][2] + kr[i][3] * kr[i][3], LPR) + EPS);
            for (int n = grp; n < N; n += ngrp) {
                f32x4 m = {0.f, 0.f, 0.f, 0.f};
                if (gl < W4) {
                    m = sM4[n * W4 + gl];
                    const float wwn = sWW[n];
#pragma unroll
                    for (int e = 0; e < 4; ++e) m[e] = m[e] * (1.0f - wwn * ev[e]) + wwn * vv[e];
                    sM4[n * W4 + gl] = m;
                    if (rec && n >= row0 && n < row0 + NR) reinterpret_cast<f32x4*>(a.rec_M + (bt * N + n) * W)[gl] = m;
                }
                float nsq = group_sum_rt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2] + m[3] * m[3], LPR);
                const float nm = sqrtf(nsq + EPS);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < R) {
                        float dot = kr[i][0] * m[0] + kr[i][1] * m[1] + kr[i][2] * m[2] + kr[i][3] * m[3];
                        dot = group_sum_rt(dot, LPR);
                        if (gl == 0) sCR[i * N + n] = (dot / (krn[i] * nm + EPS)) * sI[d.oBr + i];
                    }
                }
            }
        }
        // ------------------------------------------------------------ P7a: link update of the own rows (LDS in place)
        for (int gi = tid; gi < NR * N4; gi += CT) {
            const int r = gi / N4, q = gi - r * N4;
            const int ph = r * N4 + (q ^ (r & 7));
            f32x4 l = sL4[ph];
            const float wwa = sWW[row0 + r];
            const f32x4 wwb = *reinterpret_cast<const f32x4*>(sWW + 4 * q);
            const f32x4 pb = *reinterpret_cast<const f32x4*>(sP + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = (1.0f - wwa - wwb[e]) * l[e] + wwa * pb[e];
                if (4 * q + e == row0 + r) v = 0.f;                               // matrix_set_diag(link, 0)
                l[e] = v;
            }
            sL4[ph] = l;
            if (rec) reinterpret_cast<f32x4*>(a.rec_L + (bt * N + row0 + r) * N)[q] = l;
        }
        __syncthreads();
        // ------------------------------------------------------------ P7b: directional reads on the 4x4x1 MFMA
        float* slot1p = mb1 + ((size_t)par * k + g) * slot1;                       // [fwd R x NR | bwd partial R x N]
        for (int job = wave; job < 2 * strips; job += CW) {
            const int hsel = lane & 3;
            if (job < strips) {                              // backward: column sums over the own rows
                const int c = 64 * job + lane;
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
                const float* rwp = sRW + hsel * N + row0;
                const bool hok = hsel < R;
                for (int r = 0; r < NR; r += 2) {
                    const float a0 = hok ? rwp[r] : 0.f, a1 = hok ? rwp[r + 1] : 0.f;
                    const float b0 = sL[cl_lidx(r, c, N)], b1 = sL[cl_lidx(r + 1, c, N)];
                    acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, b0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, b1, acc1, 0, 0, 0);
                }
                acc0 += acc1;
#pragma unroll
                for (int v = 0; v < 4; ++v) if (v < R) cl_store(slot1p + R * NR + v * N + c, acc0[v]);
            } else {                                         // forward: row sums of the own rows over a 64-column range
                const int rg = job - strips;
                const int par2 = lane >> 5;
                const bool hok = hsel < R;
                const float* rwp = sRW + hsel * N + 64 * rg + par2;
                for (int rb = 0; rb < NRp; rb += 32) {
                    const int row = rb + (lane & 31);
                    const bool rok = row < NR;
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
                    for (int s = 0; s < 32; s += 2) {
                        const int c0 = 64 * rg + 2 * s + par2, c1 = c0 + 2;
                        const float a0 = rok ? sL[cl_lidx(row, c0, N)] : 0.f, a1 = rok ? sL[cl_lidx(row, c1, N)] : 0.f;
                        const float b0 = hok ? rwp[2 * s] : 0.f, b1 = hok ? rwp[2 * s + 2] : 0.f;
                        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, b0, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, b1, acc1, 0, 0, 0);
                    }
                    acc0 += acc1;
                    // D[v] of this lane = (row rb + 4 * ((lane >> 2) & 7) + v, head lane & 3), column parity lane >> 5
                    const int rbase = rb + 4 * ((lane >> 2) & 7);
#pragma unroll
                    for (int v = 0; v < 4; ++v) sPart[((rg * 2 + par2) * NRp + rbase + v) * 4 + hsel] = acc0[v];
                }
            }
        }
        __syncthreads();
        for (int idx = tid; idx < R * NR; idx += CT) {       // forward reads of the own rows: fixed-order sum of the partials
            const int i = idx / NR, r = idx - i * NR;
            float f = 0.f;
            for (int q = 0; q < 2 * strips; ++q) f += sPart[(q * NRp + r) * 4 + i];
            cl_store(slot1p + idx, f);
        }
        if (wave >= CW - R) cl_softmax_row(sCR + (wave - (CW - R)) * N, N, lane);     // read content weights
        if (wave == CW - R - 1) {                                                     // sum of the write weights (precedence)
            float s = 0.f;
            for (int n = lane; n < N; n += 64) s += sWW[n];
            s = wave_sum(s);
            if (lane == 0) sSC[0] = s;
        }
        cl_publish(fl1 + g, epoch, tid);
        if (!cl_wait(fl1, epoch, k, a.err, sAbort, t_start, tid)) return;
        // ------------------------------------------------------------ P8: read weights, precedence, reads, output
        {
            const float* base = mb1 + (size_t)par * k * slot1;
            for (int idx = tid; idx < R * N; idx += CT) {
                const int i = idx / N, n = idx - i * N;
                const int og = n / NR;
                float pv[8];
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) pv[gg] = (gg < k) ? cl_load(base + (size_t)gg * slot1 + R * NR + idx) : 0.f;
                const float fwd = cl_load(base + (size_t)og * slot1 + i * NR + (n - og * NR));
                float bwd = 0.f;
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) if (gg < k) bwd += pv[gg];
                const float* rm = sI + d.oRm + i * 3;
                const float cr = sCR[idx];
                const float v = rm[2] * cr + rm[1] * fwd + rm[0] * bwd;                // access.py:283-303 (num_writes = 1)
                sRW[idx] = v;
                if (rec && n >= row0 && n < row0 + NR) {
                    a.rec_rw[bt * R * N + idx] = v;
                    a.rec_cr[bt * R * N + idx] = cr;
                    a.rec_fwd[bt * R * N + idx] = fwd;
                    a.rec_bwd[bt * R * N + idx] = bwd;
                }
            }
            for (int n = tid; n < N; n += CT) {
                const float pn = (1.0f - sSC[0]) * sP[n] + sWW[n];                      // addressing.py:238-240
                sP[n] = pn;
                if (rec && n >= row0 && n < row0 + NR) { a.rec_p[bt * N + n] = pn; a.rec_ww[bt * N + n] = sWW[n]; }
            }
        }
        __syncthreads();
        if (tid < nslR * nRW4) {                              // reads = rw x M_t
            const int o = tid % nRW4, sl = tid / nRW4;
            const int i = o / W4, w4 = o - i * W4;
            const int n0 = sl * nperR, n1 = min(N, n0 + nperR);
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int n = n0; n < n1; ++n) s += sRW[i * N + n] * sM4[n * W4 + w4];
            sPart4[sl * nRW4 + o] = s;
        }
        __syncthreads();
        if (tid < RWd) {
            float s = 0.f;
            for (int sl = 0; sl < nslR; ++sl) s += sPart[sl * RWd + tid];
            sZ[tid] = s;
            if (rec && g == 0) a.rec_yin[bt * d.ldy + hid + tid] = s;
        } else if (rec && g == 0 && tid < RWd + (d.ldy - d.Ky)) {
            a.rec_yin[bt * d.ldy + d.Ky + (tid - RWd)] = (tid == RWd) ? 1.f : 0.f;
        }
        __syncthreads();
        if (g == 0) {
            for (int o = wave; o < d.O; o += CW) {            // y = clip([h ; reads] Wy + by)   (dnc.py:118-122)
                float s = 0.f;
                for (int kk = lane; kk < d.Ky; kk += 64) {
                    const float zv = (kk < hid) ? sZ[RWd + kk] : sZ[kk - hid];
                    s += zv * a.Wy[(size_t)kk * d.OP + o];
                }
                s = wave_sum(s);
                if (lane == 0) {
                    const float pre = s + a.Wy[(size_t)d.Ky * d.OP + o];
                    a.out[bt * d.O + o] = dnc_clip(pre, clipv);
                    if (rec) a.rec_ypre[bt * d.O + o] = pre;
                }
            }
        }
    }
    __syncthreads();

    // ---- store state: own rows of memory and link, own units of the cell; the replicated vectors by workgroup 0
    {
        f32x4* gM4 = reinterpret_cast<f32x4*>(a.mem + ((size_t)b * N + row0) * W);
        for (int i = tid0; i < NR * W4; i += CT) gM4[i] = sM4[row0 * W4 + i];
        f32x4* gL4 = reinterpret_cast<f32x4*>(a.link + ((size_t)b * N + row0) * N);
        for (int i = tid0; i < NR * N4; i += CT) {
            const int r = i / N4, q = i - r * N4;
            gL4[i] = sL4[r * N4 + (q ^ (r & 7))];
        }
    }
    for (int i = tid0; i < nU; i += CT) a.hc[(size_t)b * 2 * hid + hid + u0 + i] = sC[i];
    if (g == 0) {
        for (int i = tid0; i < N; i += CT) {
            a.usage[(size_t)b * N + i] = sU[i];
            a.ww[(size_t)b * N + i] = sWW[i];
            a.prec[(size_t)b * N + i] = sP[i];
        }
        for (int i = tid0; i < R * N; i += CT) a.rw[(size_t)b * R * N + i] = sRW[i];
        for (int i = tid0; i < RWd; i += CT) a.reads[(size_t)b * RWd + i] = sZ[i];
        for (int i = tid0; i < hid; i += CT) a.hc[(size_t)b * 2 * hid + i] = sZ[RWd + i];
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
static int dnc_cluster_geom(int B, int N, int W, int R, int Wn, int hid, int O, int k_req, DncDims& d, DncClusterGeom& g,
                            size_t* lds_bytes) {
    dnc_fill_dims(d, B, 1, N, W, R, Wn, hid, O, 0.f);
    if (Wn != 1 || R < 1 || R > 4 || N < 64 || (N % 64) != 0 || N > 1024 || W < 4 || (W % 4) != 0 || W > 256 || hid < 4 ||
        hid > 1024 || O < 1 || O > CW || B < 1 || (d.IP / 4) > CT || R * (W / 4) > CT)
        return 0;
    for (int k = 8; k >= 1; k >>= 1) {
        if (k_req > 0 && k != k_req) continue;
        if ((long)B * k > 256) continue;                       // one workgroup per CU, all co-resident
        const int NR = N / k;
        if (NR * k != N || NR < 8 || (NR % 4) != 0) continue;
        g.k = k; g.NR = NR; g.upk = (hid + k - 1) / k;
        g.slot0 = dnc_cluster_align4(dnc_cluster_align4(g.upk) + d.IP);
        g.slot1 = dnc_cluster_align4(R * NR + R * N);
        g.xcd_local = 0;
        DncClFwdLds L;
        dnc_cl_fwd_lds(d, g, L);
        const size_t bytes = (size_t)L.total * sizeof(float);
        if (bytes > 160 * 1024) continue;
        if (lds_bytes) *lds_bytes = bytes;
        return k;
    }
    return 0;
}

// control block (flags + error word) first, padded to 256 bytes, then the mailbox
static size_t dnc_cluster_ctrl_bytes(int B, int k) { return (((size_t)B * 2 * k + 1) * sizeof(unsigned) + 255) & ~(size_t)255; }

extern "C" int ntk_dnc_cluster_plan(int B, int N, int W, int R, int Wn, int hid, int O, int k_request, int* k,
                                    size_t* workspace_bytes) {
    DncDims d;
    DncClusterGeom g;
    const int kk = dnc_cluster_geom(B, N, W, R, Wn, hid, O, k_request, d, g, nullptr);
    if (k) *k = kk;
    if (workspace_bytes) *workspace_bytes = 0;
    if (kk <= 0) {
        ntk_set_error("ntk_dnc_cluster_plan: B=%d N=%d W=%d R=%d Wn=%d hid=%d is outside the cluster kernels' range "
                      "(num_writes 1, memory_size a multiple of 64, link rows and memory LDS resident, B * k <= 256)",
                      B, N, W, R, Wn, hid);
        return NTK_ERR_UNSUPPORTED;
    }
    if (workspace_bytes) *workspace_bytes = dnc_cluster_ctrl_bytes(B, kk) + dnc_cluster_mbox_floats(B, kk, g.slot0, g.slot1) * sizeof(float);
    return NTK_OK;
}

extern "C" int ntk_dnc_cluster_status(const void* workspace, int B, int k, void* stream) {
    NTK_REQUIRE(workspace && B > 0 && k > 0, NTK_ERR_BAD_PTR, "ntk_dnc_cluster_status: bad arguments");
    unsigned e = 0;
    const unsigned* errw = reinterpret_cast<const unsigned*>(workspace) + (size_t)B * 2 * k;
    hipError_t rc = hipMemcpyAsync(&e, errw, sizeof(e), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (rc == hipSuccess) rc = hipStreamSynchronize((hipStream_t)stream);
    NTK_REQUIRE(rc == hipSuccess, NTK_ERR_HIP, "ntk_dnc_cluster_status: %s", hipGetErrorString(rc));
    NTK_REQUIRE(e == 0, NTK_ERR_HIP, "ntk_dnc_cluster_status: a cluster hand-off timed out (the launch was aborted; its outputs are invalid)");
    return NTK_OK;
}

extern "C" int ntk_dnc_cluster_fwd(int B, int S, int N, int W, int R, int Wn, int hid, int O, float clip_value, int k,
                                   const float* xproj, const float* Wr, const float* Wi, const float* Wy,
                                   float* mem, float* link, float* usage, float* rw, float* ww, float* prec,
                                   float* reads, float* hc, float* out,
                                   float* rec_z, float* rec_gates, float* rec_c, float* rec_hc, float* rec_yin,
                                   float* rec_ifc, float* rec_u, float* rec_ww, float* rec_rw, float* rec_cw,
                                   float* rec_cr, float* rec_al, float* rec_p, float* rec_fwd, float* rec_bwd,
                                   float* rec_M, float* rec_L, float* rec_ypre, void* workspace, void* stream) {
    DncClFwdArgs a;
    size_t lds_bytes = 0;
    NTK_REQUIRE(B > 0 && S > 0 && k > 0, NTK_ERR_BAD_SHAPE, "ntk_dnc_cluster_fwd: B=%d S=%d k=%d", B, S, k);
    const int kk = dnc_cluster_geom(B, N, W, R, Wn, hid, O, k, a.d, a.g, &lds_bytes);
    NTK_REQUIRE(kk == k, NTK_ERR_UNSUPPORTED, "ntk_dnc_cluster_fwd: k=%d is not a valid cluster size for B=%d N=%d W=%d R=%d Wn=%d hid=%d "
                "(ask ntk_dnc_cluster_plan)", k, B, N, W, R, Wn, hid);
    dnc_fill_dims(a.d, B, S, N, W, R, Wn, hid, O, clip_value);
    NTK_REQUIRE(xproj && Wr && Wi && Wy && mem && link && usage && rw && ww && prec && reads && hc && out && workspace, NTK_ERR_BAD_PTR,
                "ntk_dnc_cluster_fwd: null pointer");
    NTK_REQUIRE(ntk_aligned16(xproj) && ntk_aligned16(Wr) && ntk_aligned16(Wi) && ntk_aligned16(mem) && ntk_aligned16(link) &&
                    ntk_aligned16(workspace), NTK_ERR_BAD_PTR, "ntk_dnc_cluster_fwd: xproj/Wr/Wi/mem/link/workspace must be 16-byte aligned");
    {
        float* recs[] = {rec_z, rec_gates, rec_c, rec_hc, rec_yin, rec_ifc, rec_u, rec_ww, rec_rw, rec_cw, rec_cr, rec_al,
                         rec_p, rec_fwd, rec_bwd, rec_M, rec_L, rec_ypre};
        int nn = 0;
        for (float* r : recs) nn += (r != nullptr);
        NTK_REQUIRE(nn == 0 || nn == 18, NTK_ERR_BAD_PTR, "ntk_dnc_cluster_fwd: record pointers are all-or-none (%d of 18 given)", nn);
        NTK_REQUIRE(nn == 0 || (ntk_aligned16(rec_gates) && ntk_aligned16(rec_M) && ntk_aligned16(rec_L)), NTK_ERR_BAD_PTR,
                    "ntk_dnc_cluster_fwd: rec_gates/rec_M/rec_L must be 16-byte aligned");
    }
    a.g.xcd_local = (B % 8) == 0 ? 1 : 0;
    a.xproj = xproj; a.Wr = Wr; a.Wi = Wi; a.Wy = Wy; a.mem = mem; a.link = link; a.usage = usage; a.rw = rw; a.ww = ww;
    a.prec = prec; a.reads = reads; a.hc = hc; a.out = out;
    a.rec_z = rec_z; a.rec_gates = rec_gates; a.rec_c = rec_c; a.rec_hc = rec_hc; a.rec_yin = rec_yin; a.rec_ifc = rec_ifc;
    a.rec_u = rec_u; a.rec_ww = rec_ww; a.rec_rw = rec_rw; a.rec_cw = rec_cw; a.rec_cr = rec_cr; a.rec_al = rec_al;
    a.rec_p = rec_p; a.rec_fwd = rec_fwd; a.rec_bwd = rec_bwd; a.rec_M = rec_M; a.rec_L = rec_L; a.rec_ypre = rec_ypre;
    const size_t ctrl = dnc_cluster_ctrl_bytes(B, k);
    a.flags = reinterpret_cast<unsigned*>(workspace);
    a.err = a.flags + (size_t)B * 2 * k;
    a.mbox = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + ctrl);
    DncClFwdLds L;
    dnc_cl_fwd_lds(a.d, a.g, L);
    {
        static NtkLdsAttrCache lds_cache;
        const void* const ks[] = {(const void*)dnc_cluster_fwd_kernel};
        const int rc_lds = ntk_raise_lds_limit(lds_cache, ks, 1, "ntk_dnc_cluster_fwd");
        if (rc_lds != NTK_OK) return rc_lds;
    }
    hipError_t e = hipMemsetAsync(workspace, 0, ctrl, (hipStream_t)stream);     // flags + error word: zero before EVERY launch
    NTK_REQUIRE(e == hipSuccess, NTK_ERR_HIP, "ntk_dnc_cluster_fwd: hipMemsetAsync: %s", hipGetErrorString(e));
    dnc_cluster_fwd_kernel<<<B * k, CT, lds_bytes, (hipStream_t)stream>>>(a, L);
    NTK_CHECK_LAUNCH("ntk_dnc_cluster_fwd");
    return NTK_OK;
}
