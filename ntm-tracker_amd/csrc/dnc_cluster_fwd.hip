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
#include <vector>

// Diagnostic build only (-DNTK_CL_PROF): workgroup 0 accumulates s_memtime deltas per phase (its wave 0, lane 0) into
// g_cl_prof; ntk_dnc_cluster_prof() copies them out.  The stamps serialise the phases: read SHARES, not totals.
#ifdef NTK_CL_PROF
__device__ unsigned long long g_cl_prof[32];
#define CL_STAMP(i)                                                                   \
    do {                                                                              \
        if (blockIdx.x == 0 && tid == 0) {                                            \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();             \
            prof_acc[i] += now_ - prof_last;                                          \
            prof_last = now_;                                                         \
        }                                                                             \
    } while (0)
#else
#define CL_STAMP(i) do { } while (0)
#endif

namespace {

struct DncClFwdLds {
    int part, M, L, Z, C, HP, I, K, U, NU, KEY, RANK, RW, RWT, WW, P, CW, CR, SC, total;
};

// LDS carve-up (offsets in floats) of a shape: constexpr, so the FIX kernel sees immediates
constexpr __host__ __device__ DncClFwdLds dnc_cl_fwd_lds(const DncClusterCfg& c) {
    DncClFwdLds L = {};
    const int N = c.N, RWd = c.R * c.W;
    int part = c.ksl * c.upk * 4;
    part = dnc_cluster_max(part, c.nslI * c.IP);
    part = dnc_cluster_max(part, c.nslA * N + N);          // rank partials + the rank-ordered usage vector
    part = dnc_cluster_max(part, c.strips * 2 * c.NRp * 4);
    part = dnc_cluster_max(part, CW * RWd);
    int o = 0;
    L.part = o; o += dnc_cluster_align4(part);
    L.M = o; o += dnc_cluster_align4(N * (c.W + 4));       // rows padded by one float4: a thread pair walks ONE row conflict-free
    L.L = o; o += dnc_cluster_align4(c.NR * N);
    L.Z = o; o += dnc_cluster_align4(c.K);
    L.C = o; o += dnc_cluster_align4(c.upk);
    L.HP = o; o += dnc_cluster_align4(c.upk);             // h_{t-1} of the own units (the deferred output product reads it)
    L.I = o; o += dnc_cluster_align4(c.IP);
    L.K = o; o += dnc_cluster_align4((1 + c.R) * c.W);
    L.U = o; o += dnc_cluster_align4(N);
    L.NU = o; o += dnc_cluster_align4(N);
    L.KEY = o; o += dnc_cluster_align4(2 * N);
    L.RANK = o; o += dnc_cluster_align4(N);
    L.RW = o; o += dnc_cluster_align4(c.R * N);
    L.RWT = o; o += dnc_cluster_align4(4 * N);
    L.WW = o; o += dnc_cluster_align4(N);
    L.P = o; o += dnc_cluster_align4(N);
    L.CW = o; o += dnc_cluster_align4(N);
    L.CR = o; o += dnc_cluster_align4(c.R * N);
    L.SC = o; o += 64;
    L.total = o;
    return L;
}
constexpr DncClFwdLds kDncClFixFwdLds = dnc_cl_fwd_lds(kDncClusterFixCfg);

struct DncClFwdArgs {
    int B, S, xcd_local;
    float clip;
    DncClusterCfg c;
    DncClFwdLds lds;
    const float* xproj; const float* Wr; const float* Wi; const float* Wy;
    float* mem; float* link; float* usage; float* rw; float* ww; float* prec; float* reads; float* hc; float* out;
    float* rec_z; float* rec_gates; float* rec_c; float* rec_hc; float* rec_yin; float* rec_ifc; float* rec_u;
    float* rec_ww; float* rec_rw; float* rec_cw; float* rec_cr; float* rec_al; float* rec_p; float* rec_fwd;
    float* rec_bwd; float* rec_M; float* rec_L; float* rec_ypre;
    float* mbox; unsigned* flags; unsigned* err;
    unsigned* xcc;         // [B][k] handshake words of cl_same_xcd (control block)
};

__device__ __forceinline__ void cl_softmax_row(float* r, int N, int lane) {      // one wave, in place
    float mx = -INFINITY;
    for (int n = lane; n < N; n += 64) mx = fmaxf(mx, r[n]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int n = lane; n < N; n += 64) { const float e = expf(r[n] - mx); r[n] = e; s += e; }
    s = wave_sum(s);
    for (int n = lane; n < N; n += 64) r[n] = r[n] / s;
}

__device__ __forceinline__ float cl_dot4(const f32x4& x, const f32x4& y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3]; }
__device__ __forceinline__ float cl_pair_sum(float v) { return v + ntk_dpp<0xB1>(v); }      // lanes 2p, 2p+1

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

// LDS views and launch constants derived from the shape configuration C and the LDS layout L in scope
#define CL_FWD_VIEWS()                                                                                                        \
    const int k = C.k, NR = C.NR, upk = C.upk, upkp = C.upkp;                                                                 \
    const int N = C.N, W = C.W, R = C.R;                                                                                      \
    const int hid = C.hid, K = C.K, IP = C.IP, RWd = R * W, N4 = N >> 2, W4 = W >> 2, WS = W + 4, WS4 = W4 + 1;               \
    const int row0 = g * NR, u0 = min(hid, g * upk), u1 = min(hid, u0 + upk), nU = u1 - u0;                                   \
    float* sPart = smem + L.part; float* sM = smem + L.M; float* sL = smem + L.L;                                             \
    float* sZ = smem + L.Z; float* sC = smem + L.C; float* sHP = smem + L.HP; float* sI = smem + L.I; float* sK = smem + L.K;                            \
    float* sU = smem + L.U; float* sNU = smem + L.NU;                                                                         \
    unsigned long long* sKEY = reinterpret_cast<unsigned long long*>(smem + L.KEY);                                          \
    int* sRank = reinterpret_cast<int*>(smem + L.RANK);                                                                       \
    float* sRW = smem + L.RW; float* sRWT = smem + L.RWT; float* sWW = smem + L.WW;                                           \
    float* sP = smem + L.P; float* sCW = smem + L.CW; float* sCR = smem + L.CR;                                               \
    float* sSC = smem + L.SC; int* sAbort = reinterpret_cast<int*>(sSC + 32);                                                 \
    f32x4* sPart4 = reinterpret_cast<f32x4*>(sPart); f32x4* sM4 = reinterpret_cast<f32x4*>(sM);                              \
    f32x4* sL4 = reinterpret_cast<f32x4*>(sL); const f32x4* sK4 = reinterpret_cast<const f32x4*>(sK);                        \
    (void)upkp; (void)K; (void)IP; (void)RWd; (void)N4; (void)WS; (void)WS4; (void)u1; (void)nU; (void)sPart4; (void)sK4;    \
    (void)sKEY; (void)sRank; (void)sNU; (void)sCW; (void)sCR; (void)sK; (void)sAbort; (void)sL; (void)sM; (void)sRWT; (void)sI; (void)sC; (void)sHP; (void)k

// FIX: the benchmark shape (kDncClusterFixCfg) with every dimension, offset and LDS address a compile-time constant:
// the short per-row and per-slot loops unroll, their LDS loads are issued in batches instead of one dependent load
// per iteration (the first version of this kernel spent 60 % of a step waiting for such loads), and the uniform
// state fits the SGPR file.  The generic instantiation reads the same values from the kernarg segment INSIDE the
// time loop (through a pointer made opaque once per step), so they are not hoisted and spilled.
template <bool FIX>
__global__ __launch_bounds__(CT) void dnc_cluster_fwd_kernel(DncClFwdArgs a0) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef const __attribute__((address_space(4))) DncClFwdArgs* ArgsK;
    const ArgsK ak0 = (ArgsK)__builtin_amdgcn_kernarg_segment_ptr();
    const int tid0 = threadIdx.x;
    const int kk0 = FIX ? kDncClusterFixCfg.k : a0.c.k;
    int b, g;
    if (a0.xcd_local) {                     // the k members of a sequence share blockIdx % 8 (speed only, never correctness)
        const int x = blockIdx.x & 7, s = blockIdx.x >> 3;
        b = x + 8 * (s / kk0);
        g = s % kk0;
    } else {
        b = blockIdx.x / kk0;
        g = blockIdx.x % kk0;
    }
    const float EPS = 1e-6f;
    const int S = a0.S;
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();

    // ---- load state (memory replicated, link rows of this workgroup, per-slot vectors replicated)
    {
        const DncClFwdArgs& a = a0;
        const DncClusterCfg C = FIX ? kDncClusterFixCfg : a.c;
        const DncClFwdLds L = FIX ? kDncClFixFwdLds : a.lds;
        CL_FWD_VIEWS();
        if (tid0 == 0) *sAbort = 0;
        const f32x4* gM4 = reinterpret_cast<const f32x4*>(a.mem + (size_t)b * N * W);
        for (int i = tid0; i < N * W4; i += CT) { const int n = i / W4, j = i - n * W4; sM4[n * WS4 + j] = gM4[i]; }
        const f32x4* gL4 = reinterpret_cast<const f32x4*>(a.link + ((size_t)b * N + row0) * N);
        for (int i = tid0; i < NR * N4; i += CT) {
            const int r = i / N4, q = i - r * N4;
            sL4[r * N4 + (q ^ (r & 7))] = gL4[i];
        }
        for (int i = tid0; i < N; i += CT) {
            sU[i] = a.usage[(size_t)b * N + i];
            sWW[i] = a.ww[(size_t)b * N + i];
            sP[i] = a.prec[(size_t)b * N + i];
        }
        for (int i = tid0; i < 4 * N; i += CT) sRWT[i] = 0.f;
        for (int i = tid0; i < R * N; i += CT) sRW[i] = a.rw[(size_t)b * R * N + i];
        for (int i = tid0; i < RWd; i += CT) sZ[i] = a.reads[(size_t)b * RWd + i];
        for (int i = tid0; i < hid; i += CT) sZ[RWd + i] = a.hc[(size_t)b * 2 * hid + i];
        for (int i = tid0; i < nU; i += CT) sC[i] = a.hc[(size_t)b * 2 * hid + hid + u0 + i];
        for (int i = tid0; i < (1 + R) * W; i += CT) sK[i] = 0.f;
    }
    __syncthreads();
    // same-XCD fast form of the hand-offs (dnc_cluster.h): decided per cluster by a handshake, never assumed
    bool plain = false;
    if (a0.xcd_local) {
        int* const sw = reinterpret_cast<int*>(smem + (FIX ? kDncClFixFwdLds.SC : a0.lds.SC)) + 33;
        const int same = cl_same_xcd(a0.xcc + (size_t)b * kk0, g, kk0, a0.err, sw - 1, sw, t_start, tid0);
        if (same < 0) return;
        plain = __builtin_amdgcn_readfirstlane(same) != 0;
    }

    // FIX: the slice of Wr a thread multiplies in P1 (own unit j x K-slice ks: kperG rows of one float4 gate column) never
    // changes -- it lives in registers for the whole sequence (23 x 4 of the 256 a thread has at two waves per SIMD)
    // instead of being streamed from L2 every step (182 KB per workgroup and step); the step's input projection row is
    // requested one step ahead.  Same products in the same order as ntk_stream_matvec: results are unchanged.
    constexpr int KPG = kDncClusterFixCfg.kperG;
    f32x4 wres[FIX ? KPG : 1];
    f32x4 xg_bias = {0.f, 0.f, 0.f, 0.f}, xp_next = {0.f, 0.f, 0.f, 0.f};
    if constexpr (FIX) {
        constexpr DncClusterCfg C = kDncClusterFixCfg;
        const f32x4* Wr4 = reinterpret_cast<const f32x4*>(a0.Wr);
        const int u0 = min(C.hid, g * C.upk), nU = min(C.hid, u0 + C.upk) - u0;
        const int ks = cl_div(tid0, C.mg_upk), j = tid0 - ks * C.upk, k0 = ks * KPG;
        const bool act = tid0 < C.ksl * C.upk && j < nU;
#pragma unroll
        for (int q = 0; q < KPG; ++q)
            wres[q] = (act && k0 + q < C.K) ? Wr4[(size_t)(k0 + q) * C.hid + u0 + j] : f32x4{0.f, 0.f, 0.f, 0.f};
        if (tid0 < nU) {
            xg_bias = Wr4[(size_t)C.K * C.hid + u0 + tid0];
            xp_next = reinterpret_cast<const f32x4*>(a0.xproj)[(size_t)b * S * C.hid + u0 + tid0];
        }
    }
#ifdef NTK_CL_PROF
    unsigned long long prof_acc[16] = {0}, prof_last = __builtin_amdgcn_s_memtime();
#endif
    for (int t = 0; t < S; ++t) {
        ArgsK ak = ak0;
        asm volatile("" : "+s"(ak));
        const auto& a = *ak;
        DncClusterCfg C = kDncClusterFixCfg;
        DncClFwdLds L = kDncClFixFwdLds;
        if constexpr (!FIX) {                  // generic shape: (dead fields of) the two blocks come from the kernarg segment
            __builtin_memcpy(&C, (const void*)&a.c, sizeof(C));
            __builtin_memcpy(&L, (const void*)&a.lds, sizeof(L));
        }
        CL_FWD_VIEWS();
        const float clipv = a.clip;
        const int ksl = C.ksl, kperG = C.kperG, icg = C.icg, nslI = C.nslI, uperI = C.uperI;
        const int nslA = C.nslA, mperA = C.mperA, strips = C.strips, NRp = C.NRp, HW4 = C.HW4, nperW = C.nperW;
        const int slot0 = C.slot0, slot1 = C.slot1;
        float* mb0 = a.mbox + (size_t)b * 2 * k * ((size_t)slot0 + slot1);        // [parity][g][slot0]
        float* mb1 = mb0 + (size_t)2 * k * slot0;                                 // [parity][g][slot1]
        unsigned* fl0 = a.flags + (size_t)b * 2 * k;
        unsigned* fl1 = fl0 + k;
        const f32x4* Wr4 = reinterpret_cast<const f32x4*>(a.Wr);
        const f32x4* Wi4 = reinterpret_cast<const f32x4*>(a.Wi);
        const bool rec = a.rec_z != nullptr;
        int tid_op = tid0;
        asm volatile("" : "+v"(tid_op));       // keep per-thread index math inside the step (no hoist + spill)
        const int tid = tid_op, lane = tid & 63, wave = tid >> 6;
        const size_t bt = (size_t)b * S + t;
        const unsigned epoch = (unsigned)t + 1u;
        const int par = t & 1;
        CL_STAMP(15);

        // ------------------------------------------------------------ P1: LSTM gates of the own hidden units
        f32x4 xg = {0.f, 0.f, 0.f, 0.f};
        if constexpr (FIX) xg = xp_next + xg_bias;
        else if (tid < nU) xg = reinterpret_cast<const f32x4*>(a.xproj)[bt * hid + u0 + tid] + Wr4[(size_t)K * hid + u0 + tid];
        if (rec && g == 0) for (int i = tid; i < C.ldz; i += CT) a.rec_z[bt * C.ldz + i] = (i < K) ? sZ[i] : (i == K ? 1.f : 0.f);
        if (tid < ksl * upk) {
            const int ks = cl_div(tid, C.mg_upk), j = tid - ks * upk;
            const int k0 = ks * kperG, k1 = min(K, k0 + kperG);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if constexpr (FIX) {
#pragma unroll
                for (int q = 0; q < KPG; ++q) acc += ((k0 + q < k1) ? sZ[k0 + q] : 0.f) * wres[q];
            } else {
                if (j < nU && k0 < k1) acc = ntk_stream_matvec<4>(Wr4 + u0 + j, hid, sZ, k0, k1, K - 1);
            }
            sPart4[ks * upk + j] = acc;
        }
        __syncthreads();
        if (tid < nU) {
            f32x4 gsum = xg;
            for (int ks = 0; ks < ksl; ++ks) gsum += sPart4[ks * upk + tid];
            const float gi = cl_sigmoid(gsum[0]), gj = cl_tanh(gsum[1]);
            const float gf = cl_sigmoid(gsum[2] + 1.0f);             // snt.LSTM forget_bias = 1.0
            const float go = cl_sigmoid(gsum[3]);
            const float c2 = gf * sC[tid] + gi * gj;
            const float h2 = cl_tanh(c2) * go;
            sC[tid] = dnc_clip(c2, clipv);                           // dnc.py:112-113
            sHP[tid] = sZ[RWd + u0 + tid];                           // h_{t-1}: still needed by the deferred output of step t-1
            sZ[RWd + u0 + tid] = dnc_clip(h2, clipv);
            if (rec) {
                f32x4 ga = {gi, gj, gf, go};
                reinterpret_cast<f32x4*>(a.rec_gates)[bt * hid + u0 + tid] = ga;
                a.rec_c[bt * hid + u0 + tid] = c2;
            }
        }
        __syncthreads();
        CL_STAMP(0);
        // ------------------------------------------------------------ P2: interface partial sums over the own units
        if (tid < nslI * icg) {
            const int us = cl_div(tid, C.mg_icg), cg = tid - us * icg;
            const int ua = u0 + us * uperI, ub = min(u1, ua + uperI);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const f32x4* wp = Wi4 + (size_t)ua * icg + cg;
#pragma unroll 8
            for (int u = ua; u < ub; ++u, wp += icg) acc += sZ[RWd + u] * (*wp);
            sPart4[us * icg + cg] = acc;
        }
        __syncthreads();
        CL_STAMP(1);
        {   // publish exchange 0: [h of the own units | interface partial]
            float* slot = mb0 + ((size_t)par * k + g) * slot0;
            if (tid < nU) cl_store(slot + tid, sZ[RWd + u0 + tid], plain);
            for (int c = tid; c < IP; c += CT) {
                float v = 0.f;
                for (int us = 0; us < nslI; ++us) v += sPart[us * IP + c];
                cl_store(slot + upkp + c, v, plain);
            }
            cl_publish(fl0 + g, epoch, tid, plain);
        }
        // y_{t-1} = clip([h_{t-1} ; reads_{t-1}] Wy + by) (dnc.py:118-122) does not feed the recurrence: workgroup 0 computes it
        // HERE, in the shadow of the hand-off (its wave 0 polls, waves 1.. have nothing else to do), not on the step's
        // critical path.  h_{t-1}: sHP for the own units (already overwritten in sZ), sZ for the others (overwritten after the wait).
        if (g == 0 && t > 0 && wave >= 1 && wave <= C.O) {
            const int o = wave - 1;
            float s = 0.f;
            for (int kk = lane; kk < C.Ky; kk += 64) {
                const float zv = (kk < hid) ? ((kk >= u0 && kk < u1) ? sHP[kk - u0] : sZ[RWd + kk]) : sZ[kk - hid];
                s += zv * a.Wy[(size_t)kk * C.OP + o];
            }
            s = wave_sum(s);
            if (lane == 0) {
                const float pre = s + a.Wy[(size_t)C.Ky * C.OP + o];
                a.out[(bt - 1) * C.O + o] = dnc_clip(pre, clipv);
                if (rec) a.rec_ypre[(bt - 1) * C.O + o] = pre;
            }
        }
        CL_STAMP(2);
        if (!cl_wait(fl0, epoch, k, a.err, sAbort, t_start, tid)) return;
        CL_STAMP(3);
        {   // consume exchange 0: full h, activated interface (+ aligned copies of the keys)
            const float* base = mb0 + (size_t)par * k * slot0;
            for (int u = tid; u < hid; u += CT) {
                const int gg = cl_div(u, C.mg_upk);
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
                if (c >= C.oE && c < C.oRm) r = dnc_sigmoid(v);                      // erase, free, alloc, write gates
                else if ((c >= C.oBw && c < C.oKr) || (c >= C.oBr && c < C.I)) r = dnc_softplus(v);   // strengths
                sI[c] = r;
                if (c >= C.oKw && c < C.oBw) sK[c - C.oKw] = r;
                else if (c >= C.oKr && c < C.oBr) sK[W + (c - C.oKr)] = r;
            }
        }
        if constexpr (FIX) {   // the next step's input projection row (HBM): requested here, used at the top of the next step
            if (tid < nU && t + 1 < S) xp_next = reinterpret_cast<const f32x4*>(a.xproj)[(bt + 1) * hid + u0 + tid];
        }
        __syncthreads();
        CL_STAMP(4);
        if (rec && g == 0) {
            for (int i = tid; i < C.ldh; i += CT) {
                const float v = (i < hid) ? sZ[RWd + i] : (i == hid ? 1.f : 0.f);
                a.rec_hc[bt * C.ldh + i] = v;
                if (i < hid) a.rec_yin[bt * C.ldy + i] = v;
            }
            for (int c = tid; c < IP; c += CT) {
                float v = sI[c];
                if (c >= C.oRm && c < C.oKw) {             // the read modes are recorded after their softmax (computed below)
                    const float* rm = sI + C.oRm + ((c - C.oRm) / 3) * 3;
                    const float mx = fmaxf(rm[0], fmaxf(rm[1], rm[2]));
                    const float e0 = expf(rm[0] - mx), e1 = expf(rm[1] - mx), e2 = expf(rm[2] - mx);
                    v = expf(v - mx) / (e0 + e1 + e2);
                }
                a.rec_ifc[bt * IP + c] = v;
            }
        }
        // key norms: wave i < 1 + R  ->  sSC[8 + i] = sqrt(|key_i|^2 + eps)
        if (wave < 1 + R) {
            float ss = 0.f;
            for (int w = lane; w < W; w += 64) { const float kv = sK[wave * W + w]; ss += kv * kv; }
            ss = wave_sum(ss);
            if (lane == 0) sSC[8 + wave] = sqrtf(ss + EPS);
        }
        // ------------------------------------------------------------ P3: usage (addressing.py:342-374), op by op
        {
#pragma clang fp contract(off)
            float fg[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fg[i] = (i < R) ? sI[C.oF + i] : 0.f;
            for (int n = tid; n < N; n += CT) {
                float pw = 1.f;
                pw *= (1.0f - sWW[n]);
                float u = sU[n];
                u = u + (1.0f - u) * (1.0f - pw);
                float phi = 1.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) if (i < R) phi *= (1.0f - fg[i] * sRW[i * N + n]);
                u *= phi;
                sU[n] = u;
                const float nu = 1.0f - (EPS + (1.0f - EPS) * u);
                sNU[n] = nu;
                // sort key of the allocation: larger nonusage first, ties to the lower slot (tf.nn.top_k); nonusage >= +0,
                // so its bit pattern orders like its value
                sKEY[n] = ((unsigned long long)__float_as_uint(nu) << 32) | (unsigned)(0xFFFF - n);
                if (rec && n >= row0 && n < row0 + NR) a.rec_u[bt * N + n] = u;
            }
        }
        __syncthreads();
        if (tid < R) {                                                               // read_mode softmax (access.py:186-187)
            float* rm = sI + C.oRm + tid * 3;
            const float mx = fmaxf(rm[0], fmaxf(rm[1], rm[2]));
            const float e0 = expf(rm[0] - mx), e1 = expf(rm[1] - mx), e2 = expf(rm[2] - mx);
            const float s = e0 + e1 + e2;
            rm[0] = e0 / s; rm[1] = e1 / s; rm[2] = e2 / s;
        }
        CL_STAMP(5);
        // ------------------------------------------------------------ P4: write content scores on M_{t-1}: a thread PAIR per row
        {
            const int h = tid & 1, j0 = h * HW4, j1 = min(W4, j0 + HW4);
            const float kn = sSC[8], bw = sI[C.oBw];
            for (int n = tid >> 1; n < N; n += CT / 2) {
                float nsq = 0.f, dot = 0.f;
                const f32x4* mr = sM4 + n * WS4;
#pragma unroll 4
                for (int j = j0; j < j1; ++j) { const f32x4 m = mr[j]; nsq += cl_dot4(m, m); dot += cl_dot4(sK4[j], m); }
                nsq = cl_pair_sum(nsq);
                dot = cl_pair_sum(dot);
                if (h == 0) sCW[n] = (dot * cl_rcp(kn * cl_sqrt(nsq + EPS) + EPS)) * bw;
            }
        }
        // ------------------------------------------------------------ P5a: rank of every slot in the usage order
        if (tid < nslA * N) {
            const int sl = (FIX ? tid / N : cl_div(tid, C.mg_N)), n = tid - sl * N;
            const unsigned long long mine = sKEY[n];
            const u64x2* kp = reinterpret_cast<const u64x2*>(sKEY + sl * mperA);
            int cnt = 0;
            for (int m = 0; m < mperA; m += 8) {
                const u64x2 k0 = kp[(m >> 1)], k1 = kp[(m >> 1) + 1], k2 = kp[(m >> 1) + 2], k3 = kp[(m >> 1) + 3];
                cnt += (k0[0] > mine) + (k0[1] > mine) + (k1[0] > mine) + (k1[1] > mine) + (k2[0] > mine) + (k2[1] > mine) +
                       (k3[0] > mine) + (k3[1] > mine);
            }
            reinterpret_cast<int*>(sPart)[sl * N + n] = cnt;
        }
        __syncthreads();
        CL_STAMP(6);
        // ------------------------------------------------------------ P5b: usages scattered into rank order; write-content softmax
        {
            float* sT = sPart + nslA * N;
            for (int n = tid; n < N; n += CT) {
                int rk = 0;
                for (int sl = 0; sl < nslA; ++sl) rk += reinterpret_cast<const int*>(sPart)[sl * N + n];
                sRank[n] = rk;
                sT[rk] = 1.0f - sNU[n];                    // sorted_usage = 1 - sorted_nonusage (addressing.py:398)
            }
            if (wave == CW - 1) cl_softmax_row(sCW, N, lane);
            __syncthreads();
            // P5c: exclusive cumulative product in rank order (tf.cumprod(exclusive=True), addressing.py:399) by wave 0
            if (wave == 0) {
                const int PER = N >> 6, base = lane * PER;
                float ex[8], run = 1.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) if (j < PER) { ex[j] = run; run *= sT[base + j]; }
                float inc = run;
#pragma unroll
                for (int dd = 1; dd < 64; dd <<= 1) { const float o = __shfl_up(inc, dd, 64); if (lane >= dd) inc *= o; }
                float excl = __shfl_up(inc, 1, 64);
                if (lane == 0) excl = 1.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) if (j < PER) sT[base + j] = excl * ex[j];
            }
            __syncthreads();
            // P5d: allocation and write weights (access.py:220-257), op by op
            {
#pragma clang fp contract(off)
                const float ag = sI[C.oAg], wg = sI[C.oWg];
                for (int n = tid; n < N; n += CT) {
                    const float al = sNU[n] * sT[sRank[n]];
                    const float cw = sCW[n];
                    sWW[n] = wg * (ag * al + (1.0f - ag) * cw);
                    if (rec && n >= row0 && n < row0 + NR) { a.rec_al[bt * N + n] = al; a.rec_cw[bt * N + n] = cw; }
                }
            }
        }
        __syncthreads();
        CL_STAMP(7);
        // ------------------------------------------------------------ P6: erase + write on M (every row), read-key scores on M_t
        {
            const int h = tid & 1, j0 = h * HW4, j1 = min(W4, j0 + HW4);
            const f32x4* sE4 = reinterpret_cast<const f32x4*>(sI + C.oE);
            const f32x4* sV4 = reinterpret_cast<const f32x4*>(sI + C.oV);
            float krn[4], br[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { krn[i] = (i < R) ? sSC[9 + i] : 1.f; br[i] = (i < R) ? sI[C.oBr + i] : 0.f; }
            for (int n = tid >> 1; n < N; n += CT / 2) {
                const float wwn = sWW[n];
                const bool own = rec && n >= row0 && n < row0 + NR;
                f32x4* mr = sM4 + n * WS4;
                float nsq = 0.f, dot[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
                for (int j = j0; j < j1; ++j) {
                    f32x4 m = mr[j];
                    const f32x4 ev = sE4[j], vv = sV4[j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) m[e] = m[e] * (1.0f - wwn * ev[e]) + wwn * vv[e];
                    mr[j] = m;
                    if (own) reinterpret_cast<f32x4*>(a.rec_M + (bt * N + n) * W)[j] = m;
                    nsq += cl_dot4(m, m);
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (i < R) dot[i] += cl_dot4(sK4[(1 + i) * W4 + j], m);
                }
                nsq = cl_pair_sum(nsq);
#pragma unroll
                for (int i = 0; i < 4; ++i) if (i < R) dot[i] = cl_pair_sum(dot[i]);
                const float nm = cl_sqrt(nsq + EPS);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < R && (i & 1) == h) sCR[i * N + n] = (dot[i] * cl_rcp(krn[i] * nm + EPS)) * br[i];
            }
        }
        CL_STAMP(8);
        // ------------------------------------------------------------ P7a: link update of the own rows (LDS in place)
        for (int base = tid; base < NR * N4; base += 4 * CT) {
            f32x4 l[4];
            int ph[4], rr[4], qq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int gi = base + u * CT;
                rr[u] = (FIX ? gi / N4 : cl_div(gi, C.mg_N4)); qq[u] = gi - rr[u] * N4;
                ph[u] = rr[u] * N4 + (qq[u] ^ (rr[u] & 7));
                if (gi < NR * N4) l[u] = sL4[ph[u]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (base + u * CT < NR * N4) {
                    const int r = rr[u], q = qq[u];
                    const float wwa = sWW[row0 + r];
                    const f32x4 wwb = *reinterpret_cast<const f32x4*>(sWW + 4 * q);
                    const f32x4 pb = *reinterpret_cast<const f32x4*>(sP + 4 * q);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = (1.0f - wwa - wwb[e]) * l[u][e] + wwa * pb[e];
                        if (4 * q + e == row0 + r) v[e] = 0.f;                    // matrix_set_diag(link, 0)
                    }
                    sL4[ph[u]] = v;
                    if (rec) reinterpret_cast<f32x4*>(a.rec_L + (bt * N + row0 + r) * N)[q] = v;
                }
            }
        }
        __syncthreads();
        CL_STAMP(9);
        // ------------------------------------------------------------ P7b: directional reads on the 4x4x1 MFMA
        float* slot1p = mb1 + ((size_t)par * k + g) * slot1;                       // [fwd R x NR | bwd partial R x N]
        for (int job = wave; job < 2 * strips; job += CW) {
            const int hsel = lane & 3;
            const bool hok = hsel < R;
            if (job < strips) {                              // backward: column sums over the own rows
                const int c = 64 * job + lane;
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
                const float* rwp = sRW + hsel * N + row0;
                for (int r = 0; r < NR; r += 8) {            // NR a multiple of 8
                    float av[8], bv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) { av[q] = rwp[r + q]; bv[q] = sL[cl_lidx(r + q, c, N)]; }
#pragma unroll
                    for (int q = 0; q < 8; q += 2) {
                        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(hok ? av[q] : 0.f, bv[q], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(hok ? av[q + 1] : 0.f, bv[q + 1], acc1, 0, 0, 0);
                    }
                }
                acc0 += acc1;
#pragma unroll
                for (int v = 0; v < 4; ++v) if (v < R) cl_store(slot1p + R * NR + v * N + c, acc0[v], plain);
            } else {                                         // forward: row sums of the own rows over a 64-column range
                const int rg = job - strips;
                const int par2 = lane >> 5;
                const float* rwp = sRW + hsel * N + 64 * rg + par2;
                for (int rb = 0; rb < NRp; rb += 32) {
                    const int row = rb + (lane & 31);
                    const bool rok = row < NR;
                    const int rowc = rok ? row : 0;
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
                    for (int s = 0; s < 32; s += 8) {
                        float av[8], bv[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) { av[q] = sL[cl_lidx(rowc, 64 * rg + 2 * (s + q) + par2, N)]; bv[q] = rwp[2 * (s + q)]; }
#pragma unroll
                        for (int q = 0; q < 8; q += 2) {
                            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(rok ? av[q] : 0.f, hok ? bv[q] : 0.f, acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(rok ? av[q + 1] : 0.f, hok ? bv[q + 1] : 0.f, acc1, 0, 0, 0);
                        }
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
        CL_STAMP(10);
        for (int idx = tid; idx < R * NR; idx += CT) {       // forward reads of the own rows: fixed-order sum of the partials
            const int i = cl_div(idx, C.mg_NR), r = idx - i * NR;
            float f = 0.f;
            for (int q = 0; q < 2 * strips; ++q) f += sPart[(q * NRp + r) * 4 + i];
            cl_store(slot1p + idx, f, plain);
        }
        if (wave >= CW - R) cl_softmax_row(sCR + (wave - (CW - R)) * N, N, lane);     // read content weights
        if (wave == CW - R - 1) {                                                     // sum of the write weights (precedence)
            float s = 0.f;
            for (int n = lane; n < N; n += 64) s += sWW[n];
            s = wave_sum(s);
            if (lane == 0) sSC[0] = s;
        }
        cl_publish(fl1 + g, epoch, tid, plain);
        CL_STAMP(11);
        if (!cl_wait(fl1, epoch, k, a.err, sAbort, t_start, tid)) return;
        CL_STAMP(12);
        // ------------------------------------------------------------ P8: read weights, precedence, reads, output
        {
            const float* base = mb1 + (size_t)par * k * slot1;
            for (int idx = tid; idx < R * N; idx += CT) {
                const int i = (FIX ? idx / N : cl_div(idx, C.mg_N)), n = idx - i * N;
                const int og = cl_div(n, C.mg_NR);
                float pv[8];
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) pv[gg] = (gg < k) ? cl_load(base + (size_t)gg * slot1 + R * NR + idx) : 0.f;
                const float fwd = cl_load(base + (size_t)og * slot1 + i * NR + (n - og * NR));
                float bwd = 0.f;
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) if (gg < k) bwd += pv[gg];
                const float* rm = sI + C.oRm + i * 3;
                const float cr = sCR[idx];
                const float v = rm[2] * cr + rm[1] * fwd + rm[0] * bwd;                // access.py:283-303 (num_writes = 1)
                sRW[idx] = v;
                sRWT[n * 4 + i] = v;
                if (rec && n >= row0 && n < row0 + NR) {
                    a.rec_rw[bt * R * N + idx] = v;
                    a.rec_cr[bt * R * N + idx] = cr;
                    a.rec_fwd[bt * R * N + idx] = fwd;
                    a.rec_bwd[bt * R * N + idx] = bwd;
                }
            }
            const float sww = sSC[0];
            for (int n = tid; n < N; n += CT) {
                const float pn = (1.0f - sww) * sP[n] + sWW[n];                         // addressing.py:238-240
                sP[n] = pn;
                if (rec && n >= row0 && n < row0 + NR) { a.rec_p[bt * N + n] = pn; a.rec_ww[bt * N + n] = sWW[n]; }
            }
        }
        __syncthreads();
        CL_STAMP(13);
        // reads = rw x M_t on the 4x4x1 MFMA (the "backward" form: block = 4 columns of the word, one memory row per
        // instruction); wave w sums its N / 8 rows, the eight partial vectors are added in a fixed order
        for (int cs = 0; cs < W; cs += 64) {
            const int c = cs + lane;
            const bool cok = c < W;
            const int cc = cok ? c : 0;
            const int hsel = lane & 3;
            const float* mp = sM + cc;
            const float* rwt = sRWT + hsel;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
            const int nA = wave * nperW;
            for (int n = nA; n < nA + nperW; n += 8) {       // N a multiple of 64: nperW a multiple of 8
                float av[8], bv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { av[q] = rwt[(n + q) * 4]; bv[q] = mp[(n + q) * WS]; }
#pragma unroll
                for (int q = 0; q < 8; q += 2) {
                    acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[q], cok ? bv[q] : 0.f, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[q + 1], cok ? bv[q + 1] : 0.f, acc1, 0, 0, 0);
                }
            }
            acc0 += acc1;
            if (cok) {
#pragma unroll
                for (int v = 0; v < 4; ++v) if (v < R) sPart[(wave * R + v) * W + c] = acc0[v];
            }
        }
        __syncthreads();
        if (tid < RWd) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < CW; ++w) s += sPart[w * RWd + tid];
            sZ[tid] = s;
            if (rec && g == 0) a.rec_yin[bt * C.ldy + hid + tid] = s;
        } else if (rec && g == 0 && tid < RWd + (C.ldy - C.Ky)) {
            a.rec_yin[bt * C.ldy + C.Ky + (tid - RWd)] = (tid == RWd) ? 1.f : 0.f;
        }
        __syncthreads();
        CL_STAMP(14);
    }
#ifdef NTK_CL_PROF
    if (blockIdx.x == 0 && tid0 == 0) for (int i = 0; i < 16; ++i) g_cl_prof[i] = prof_acc[i];
#endif
    __syncthreads();

    // ---- store state: own rows of memory and link, own units of the cell; the replicated vectors by workgroup 0
    {
        const DncClFwdArgs& a = a0;
        const DncClusterCfg C = FIX ? kDncClusterFixCfg : a.c;
        const DncClFwdLds L = FIX ? kDncClFixFwdLds : a.lds;
        CL_FWD_VIEWS();
        if (g == 0) {                                            // output of the last step (the loop defers each step's by one)
            const int wave = tid0 >> 6, lane = tid0 & 63;
            const size_t bt = (size_t)b * S + (S - 1);
            for (int o = wave; o < C.O; o += CW) {
                float s = 0.f;
                for (int kk = lane; kk < C.Ky; kk += 64) {
                    const float zv = (kk < hid) ? sZ[RWd + kk] : sZ[kk - hid];
                    s += zv * a.Wy[(size_t)kk * C.OP + o];
                }
                s = wave_sum(s);
                if (lane == 0) {
                    const float pre = s + a.Wy[(size_t)C.Ky * C.OP + o];
                    a.out[bt * C.O + o] = dnc_clip(pre, a.clip);
                    if (a.rec_z != nullptr) a.rec_ypre[bt * C.O + o] = pre;
                }
            }
        }
        f32x4* gM4 = reinterpret_cast<f32x4*>(a.mem + ((size_t)b * N + row0) * W);
        for (int i = tid0; i < NR * W4; i += CT) { const int n = i / W4, j = i - n * W4; gM4[i] = sM4[(row0 + n) * WS4 + j]; }
        f32x4* gL4 = reinterpret_cast<f32x4*>(a.link + ((size_t)b * N + row0) * N);
        for (int i = tid0; i < NR * N4; i += CT) {
            const int r = i / N4, q = i - r * N4;
            gL4[i] = sL4[r * N4 + (q ^ (r & 7))];
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
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
#ifdef NTK_CL_PROF
extern "C" int ntk_dnc_cluster_prof(unsigned long long* out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_cl_prof), 16 * sizeof(unsigned long long)) == hipSuccess ? NTK_OK : NTK_ERR_HIP;
}
#endif

// one thread: OR the launch's error word into the workspace's sticky word (dnc_cluster.h)
__global__ void dnc_cluster_latch_kernel(const unsigned* err, unsigned* sticky) { if (*err != 0u) *sticky = 1u; }
void dnc_cluster_latch(const unsigned* err, unsigned* sticky, void* stream) {
    dnc_cluster_latch_kernel<<<1, 1, 0, (hipStream_t)stream>>>(err, sticky);
}

// device-side propagation of an abort (no host synchronisation): when the sticky word of a workspace is set, the loss
// and the WHOLE gradient become NaN.  NaN, not zero: under data parallelism the gradient is SUM all-reduced next, and a
// rank that contributed zeros would let every other rank step on a partial gradient with no sign of failure; a NaN
// reaches every rank through the sum, makes the global norm NaN everywhere, and ntk_rmsprop_clip_step_checked then
// skips the update on all of them.
__global__ void dnc_cluster_guard_kernel(const unsigned* sticky, float* loss, float* grad, size_t n) {
    if (*sticky == 0u) return;
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    const float qnan = __int_as_float(0x7fc00000);
    if (i0 == 0 && loss) loss[0] = qnan;
    if (grad) for (size_t i = i0; i < n; i += step) grad[i] = qnan;
}

__global__ void dnc_cluster_inject_kernel(unsigned* sticky) { *sticky = 1u; }

extern "C" int ntk_dnc_cluster_guard(const void* workspace, size_t workspace_bytes, int mp_form, int B, int k, float* loss, float* grad,
                                     size_t n, void* stream) {
    NTK_REQUIRE(workspace && B > 0 && k > 0, NTK_ERR_BAD_PTR, "ntk_dnc_cluster_guard: bad arguments");
    const char* w = reinterpret_cast<const char*>(workspace);
    const unsigned* stk = reinterpret_cast<const unsigned*>(mp_form ? w + workspace_bytes - 256 : w + dnc_cluster_ctrl_bytes(B, k) - 256);
    NTK_REQUIRE(!mp_form || workspace_bytes >= 512, NTK_ERR_BAD_PTR, "ntk_dnc_cluster_guard: workspace_bytes");
    dnc_cluster_guard_kernel<<<64, 256, 0, (hipStream_t)stream>>>(stk, loss, grad, grad ? n : 0);
    NTK_CHECK_LAUNCH("ntk_dnc_cluster_guard");
    return NTK_OK;
}

// fault injection (tests): sets the sticky error word exactly as a timed-out hand-off does
extern "C" int ntk_dnc_cluster_inject_abort(void* workspace, size_t workspace_bytes, int mp_form, int B, int k, void* stream) {
    NTK_REQUIRE(workspace && B > 0 && k > 0, NTK_ERR_BAD_PTR, "ntk_dnc_cluster_inject_abort: bad arguments");
    NTK_REQUIRE(!mp_form || workspace_bytes >= 512, NTK_ERR_BAD_PTR, "ntk_dnc_cluster_inject_abort: workspace_bytes");
    char* w = reinterpret_cast<char*>(workspace);
    unsigned* stk = reinterpret_cast<unsigned*>(mp_form ? w + workspace_bytes - 256 : w + dnc_cluster_ctrl_bytes(B, k) - 256);
    dnc_cluster_inject_kernel<<<1, 1, 0, (hipStream_t)stream>>>(stk);
    NTK_CHECK_LAUNCH("ntk_dnc_cluster_inject_abort");
    return NTK_OK;
}

// the cluster size (0 = none) and configuration of a shape; k_req 0 = the largest k that fits
static int dnc_cluster_pick(int B, int N, int W, int R, int Wn, int hid, int O, int k_req, DncClusterCfg& c, size_t* lds_bytes) {
    if (Wn != 1 || R < 1 || R > 4 || N < 64 || (N % 64) != 0 || N > CT || W < 4 || (W % 4) != 0 || W > 256 || hid < 4 ||
        hid > 1024 || O < 1 || O > CW - 1 || B < 1)
        return 0;
    for (int k = 8; k >= 1; k >>= 1) {
        if (k_req > 0 && k != k_req) continue;
        if ((long)B * k > ntk_device_cu_count()) continue;     // one workgroup per CU, all co-resident (api.cpp)
        const int NR = N / k;
        if (NR * k != N || NR < 8 || (NR % 8) != 0) continue;
        c = dnc_cluster_cfg(N, W, R, hid, O, k);
        if (c.icg > CT || R * (W / 4) > CT) continue;
        const DncClFwdLds L = dnc_cl_fwd_lds(c);
        const size_t bytes = (size_t)L.total * sizeof(float);
        if (bytes > 160 * 1024) continue;
        if (lds_bytes) *lds_bytes = bytes;
        return k;
    }
    return 0;
}


extern "C" int ntk_dnc_cluster_plan(int B, int N, int W, int R, int Wn, int hid, int O, int k_request, int* k,
                                    size_t* workspace_bytes) {
    DncClusterCfg c;
    const int kk = dnc_cluster_pick(B, N, W, R, Wn, hid, O, k_request, c, nullptr);
    if (k) *k = kk;
    if (workspace_bytes) *workspace_bytes = 0;
    if (kk <= 0) {
        ntk_set_error("ntk_dnc_cluster_plan: B=%d N=%d W=%d R=%d Wn=%d hid=%d is outside the cluster kernels' range "
                      "(num_writes 1, memory_size a multiple of 64 up to 512, link rows and memory LDS resident, B * k <= 256)",
                      B, N, W, R, Wn, hid);
        return NTK_ERR_UNSUPPORTED;
    }
    if (workspace_bytes) *workspace_bytes = dnc_cluster_ctrl_bytes(B, kk) + dnc_cluster_mbox_floats(B, kk, c.slot0, c.slot1) * sizeof(float);
    return NTK_OK;
}

extern "C" int ntk_dnc_cluster_status(const void* workspace, int B, int k, void* stream) {
    NTK_REQUIRE(workspace && B > 0 && k > 0, NTK_ERR_BAD_PTR, "ntk_dnc_cluster_status: bad arguments");
    unsigned e[2] = {0, 0};
    const unsigned* errw = reinterpret_cast<const unsigned*>(workspace) + (size_t)B * 2 * k;
    unsigned* stk = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(const_cast<void*>(workspace)) + dnc_cluster_ctrl_bytes(B, k) - 256);
    hipError_t rc = hipMemcpyAsync(&e[0], errw, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (rc == hipSuccess) rc = hipMemcpyAsync(&e[1], stk, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (rc == hipSuccess) rc = hipMemsetAsync(stk, 0, sizeof(unsigned), (hipStream_t)stream);      // read and clear
    if (rc == hipSuccess) rc = hipStreamSynchronize((hipStream_t)stream);
    NTK_REQUIRE(rc == hipSuccess, NTK_ERR_HIP, "ntk_dnc_cluster_status: %s", hipGetErrorString(rc));
    NTK_REQUIRE(e[0] == 0 && e[1] == 0, NTK_ERR_HIP, "ntk_dnc_cluster_status: a cluster hand-off timed out (%s was aborted; its outputs are invalid)",
                e[0] ? "the last launch" : "an earlier launch on this workspace");
    return NTK_OK;
}

extern "C" int ntk_dnc_cluster_placement(const void* workspace, int B, int k, int* same_xcd_clusters, void* stream) {
    NTK_REQUIRE(workspace && same_xcd_clusters && B > 0 && k > 0 && k <= 64, NTK_ERR_BAD_PTR, "ntk_dnc_cluster_placement: bad arguments");
    std::vector<unsigned> w((size_t)B * k);
    const unsigned* xcc = reinterpret_cast<const unsigned*>(workspace) + (size_t)B * 2 * k + 1;
    hipError_t rc = hipMemcpyAsync(w.data(), xcc, w.size() * sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (rc == hipSuccess) rc = hipStreamSynchronize((hipStream_t)stream);
    NTK_REQUIRE(rc == hipSuccess, NTK_ERR_HIP, "ntk_dnc_cluster_placement: %s", hipGetErrorString(rc));
    int n = 0;
    for (int b = 0; b < B; ++b) {
        bool same = w[(size_t)b * k] != 0;          // 0 = no handshake ran (batch not a multiple of 8)
        for (int g = 1; g < k; ++g) same = same && w[(size_t)b * k + g] == w[(size_t)b * k];
        n += same ? 1 : 0;
    }
    *same_xcd_clusters = n;
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
    const int kk = dnc_cluster_pick(B, N, W, R, Wn, hid, O, k, a.c, &lds_bytes);
    NTK_REQUIRE(kk == k, NTK_ERR_UNSUPPORTED, "ntk_dnc_cluster_fwd: k=%d is not a valid cluster size for B=%d N=%d W=%d R=%d Wn=%d hid=%d "
                "(ask ntk_dnc_cluster_plan)", k, B, N, W, R, Wn, hid);
    a.lds = dnc_cl_fwd_lds(a.c);
    a.B = B; a.S = S; a.clip = clip_value;
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
    a.xcd_local = (B % 8) == 0 ? 1 : 0;
    a.xproj = xproj; a.Wr = Wr; a.Wi = Wi; a.Wy = Wy; a.mem = mem; a.link = link; a.usage = usage; a.rw = rw; a.ww = ww;
    a.prec = prec; a.reads = reads; a.hc = hc; a.out = out;
    a.rec_z = rec_z; a.rec_gates = rec_gates; a.rec_c = rec_c; a.rec_hc = rec_hc; a.rec_yin = rec_yin; a.rec_ifc = rec_ifc;
    a.rec_u = rec_u; a.rec_ww = rec_ww; a.rec_rw = rec_rw; a.rec_cw = rec_cw; a.rec_cr = rec_cr; a.rec_al = rec_al;
    a.rec_p = rec_p; a.rec_fwd = rec_fwd; a.rec_bwd = rec_bwd; a.rec_M = rec_M; a.rec_L = rec_L; a.rec_ypre = rec_ypre;
    const size_t ctrl = dnc_cluster_ctrl_bytes(B, k);
    a.flags = reinterpret_cast<unsigned*>(workspace);
    a.err = a.flags + (size_t)B * 2 * k;
    a.xcc = a.err + 1;
    a.mbox = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + ctrl);
    {
        static NtkLdsAttrCache lds_cache;
        const void* const ks[] = {(const void*)dnc_cluster_fwd_kernel<false>, (const void*)dnc_cluster_fwd_kernel<true>};
        const int rc_lds = ntk_raise_lds_limit(lds_cache, ks, 2, "ntk_dnc_cluster_fwd");
        if (rc_lds != NTK_OK) return rc_lds;
    }
    hipError_t e = hipMemsetAsync(workspace, 0, dnc_cluster_ctrl_zero_bytes(B, k), (hipStream_t)stream);     // flags + error word: zero before EVERY launch
    NTK_REQUIRE(e == hipSuccess, NTK_ERR_HIP, "ntk_dnc_cluster_fwd: hipMemsetAsync: %s", hipGetErrorString(e));
    if (dnc_cluster_is_fix(a.c)) dnc_cluster_fwd_kernel<true><<<B * k, CT, lds_bytes, (hipStream_t)stream>>>(a);
    else dnc_cluster_fwd_kernel<false><<<B * k, CT, lds_bytes, (hipStream_t)stream>>>(a);
    NTK_CHECK_LAUNCH("ntk_dnc_cluster_fwd");
    dnc_cluster_latch(a.err, reinterpret_cast<unsigned*>(reinterpret_cast<char*>(workspace) + ctrl - 256), stream);
    NTK_CHECK_LAUNCH("ntk_dnc_cluster_fwd (latch)");
    return NTK_OK;
}
