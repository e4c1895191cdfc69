// DNC core sequence backward (full BPTT), CLUSTER form: k workgroups (512 threads, one per CU) per sequence walk the
// steps recorded by the forward pass in reverse.  Same arithmetic as dnc_seq_bwd.hip (what tf.gradients computes through
// tf.nn.dynamic_rnn over dnc.DNC, direct_offset_output_with_dnc.py:615-620; non-differentiable edges: SURVEY A.4);
// see dnc_cluster.h for the hand-off protocol and dnc_cluster_fwd.hip for the forward partition.
//
//   partitioned  d(link): N/k rows per workgroup, LDS resident for the whole sequence; the pass over (dL, L_t, L_{t-1})
//                reads its rows of the recorded links once per step; row sums are complete, column sums are partial
//                and all-reduced through the mailbox;
//                the controller: dh of the own hidden units (rows of Wi), snt.LSTM backward of the own units, and
//                d[reads ; h]_{t-1} as a partial product over the own gate columns (rows of Wr^T), all-reduced;
//   replicated   everything per slot, and d(memory): the N x W gradient lives in REGISTERS for the whole sequence
//                (a 16-lane group owns one float4 column of 8 rows), next to the memory rows of the current step,
//                which are carried over from the previous iteration (M_{t-1} of step t is M_t of step t-1): one
//                64 KB record read per step instead of four passes over L2;
//   exchanged    twice per step: (0) link row / column sums, (1) the partial d[reads ; h]_{t-1}.
//
// The column sums of the memory passes (d read keys, d erase, d write vector, d write key) are accumulated in registers
// over a thread's rows, folded across the four 16-lane groups of a wave and across the waves in a FIXED order; the
// link pass uses the 4x4x1 MFMA for its two read-weight products and fixed-order sums for the rest: gradients are
// bitwise reproducible (the one-workgroup kernel used LDS float atomics).
// The allocation gradient uses the usage ORDER of the forward pass: it is re-derived from the recorded usages with
// the forward kernel's op-by-op rounding (fp contract off) and compared on stored values, never recomputed inline.
#include "dnc_cluster.h"
#include <stdlib.h>

// Diagnostic build only (-DNTK_CL_PROF): see dnc_cluster_fwd.hip
#ifdef NTK_CL_PROF
__device__ unsigned long long g_clb_prof[32];
// accumulators in global memory (no-return atomic adds by one thread): kept in registers (40 VGPRs) they made the record
// prefetch of this 246-VGPR kernel spill, and the stamped build then measured its own spill traffic; LDS is full
#define CLB_STAMP(i)                                                                  \
    do {                                                                              \
        if (blockIdx.x == 0 && tid == 0) {                                            \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();             \
            __hip_atomic_fetch_add(&g_clb_prof[i], now_ - prof_last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
            prof_last = now_;                                                         \
        }                                                                             \
    } while (0)
#else
#define CLB_STAMP(i) do { } while (0)
#endif

namespace {

struct DncClBwdLds {
    int part, CS, GL, LT, I, DX, WW, WWp, U, Up, Pp, CW, AL, SIMw, DWW, DCW, DA, gP, DPp, gU, gUn, NU, KEY, RANK,
        RWp, RWt, gRW, G, DRWp, DSIM, SIMr, GZ, DR, DHC, DG, gC, SC, total;
};

struct DncClBwdGeo {
    int ldkT, ldhT;       // leading dimensions of Wr^T [4*hid][ldkT]; Wi is used un-transposed ([ldh][IP])
    int NQ;               // memory rows per 16-lane group = N / 32
    int kg4;              // ldkT / 4
    int nslZ, nperZ;      // d[reads ; h] partial: slices of the own gate rows, rows per slice
    int nslH, cperH;      // dh of the own units: slices of the interface columns (float4), float4s per slice
    unsigned mg_kg4;
    int slot0, slot1;     // mailbox slots: link sums, partial d[reads ; h]
};

constexpr __host__ __device__ DncClBwdGeo dnc_cl_bwd_geo(const DncClusterCfg& c) {
    DncClBwdGeo q = {};
    q.ldkT = (c.K + 3) & ~3;
    q.ldhT = (c.hid + 3) & ~3;
    q.NQ = c.N / 32;
    q.kg4 = q.ldkT / 4;
    q.nslZ = dnc_cluster_max(1, CLT / q.kg4);
    q.nperZ = (4 * c.upk + q.nslZ - 1) / q.nslZ;
    q.nslH = dnc_cluster_max(1, CLT / dnc_cluster_max(1, c.upk));
    if (q.nslH > c.icg) q.nslH = c.icg;
    q.cperH = (c.icg + q.nslH - 1) / q.nslH;
    q.mg_kg4 = dnc_cluster_magic(q.kg4);
    q.slot0 = dnc_cluster_align4((c.R + 1) * c.NR + (c.R + 2) * c.N);
    q.slot1 = dnc_cluster_align4(q.ldkT);
    return q;
}

constexpr __host__ __device__ DncClBwdLds dnc_cl_bwd_lds(const DncClusterCfg& c, const DncClBwdGeo& q) {
    DncClBwdLds L = {};
    const int N = c.N, RN = c.R * c.N;
    int part = c.nslA * N + 2 * N;                                     // rank partials + the two rank-ordered vectors
    part = dnc_cluster_max(part, (CLT / 64) * 2 * N);                  // link pass: per-wave column partials (d ww, d precedence)
    part = dnc_cluster_max(part, c.strips * 2 * c.NRp * 4);            // link pass: forward-form MFMA partials
    part = dnc_cluster_max(part, q.nslZ * q.ldkT);
    part = dnc_cluster_max(part, q.nslH * c.upk);
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    L.part = take(part);
    L.CS = take((CLT / 64) * 7 * 64);                                  // column sums of the memory passes: [wave][7 float4 x 16 lanes]
    L.GL = take(c.NR * N); L.LT = take(c.NR * N);
    L.I = take(c.IP); L.DX = take(c.IP);
    L.WW = take(N); L.WWp = take(N); L.U = take(N); L.Up = take(N); L.Pp = take(N); L.CW = take(N); L.AL = take(N);
    L.SIMw = take(N); L.DWW = take(N); L.DCW = take(N); L.DA = take(N); L.gP = take(N); L.DPp = take(N); L.gU = take(N);
    L.gUn = take(N); L.NU = take(N); L.KEY = take(2 * N); L.RANK = take(N);
    L.RWp = take(RN); L.RWt = take(RN); L.gRW = take(RN); L.G = take(RN); L.DRWp = take(RN); L.DSIM = take(RN); L.SIMr = take(RN);
    L.GZ = take(q.ldkT); L.DR = take(c.R * c.W); L.DHC = take(c.hid); L.DG = take(4 * c.upk); L.gC = take(c.upk);
    L.SC = take(128);
    L.total = o;
    return L;
}
constexpr DncClBwdGeo kDncClFixBwdGeo = dnc_cl_bwd_geo(kDncClusterFixCfg);
constexpr DncClBwdLds kDncClFixBwdLds = dnc_cl_bwd_lds(kDncClusterFixCfg, kDncClFixBwdGeo);

struct DncClBwdArgs {
    int B, S, xcd_local, carry_in;
    float clip;
    DncClusterCfg c;
    DncClBwdGeo q;
    DncClBwdLds lds;
    const float* WrT; const float* Wi; const float* Wy;
    const float* mem0; const float* link0; const float* usage0; const float* rw0; const float* ww0; const float* prec0;
    const float* hc0;
    const float* rec_gates; const float* rec_c; const float* rec_ifc; const float* rec_u; const float* rec_ww;
    const float* rec_rw; const float* rec_cw; const float* rec_cr; const float* rec_al; const float* rec_p;
    const float* rec_fwd; const float* rec_bwd; const float* rec_M; const float* rec_L; const float* rec_ypre;
    const float* dout;
    float* gM; float* gL; float* dgates; float* dxi; float* dypre; float* gcarry;
    float* mbox; unsigned* flags; unsigned* err;
    unsigned* xcc;         // [B][k] handshake words of cl_same_xcd (control block)
};

__device__ __forceinline__ float cl_dot4(const f32x4& x, const f32x4& y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3]; }
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

constexpr int MAXQ = 8;       // memory rows per 16-lane group (N <= 256)

#define CL_BWD_VIEWS()                                                                                                        \
    const int k = C.k, NR = C.NR, upk = C.upk;                                                                                \
    const int N = C.N, W = C.W, R = C.R, RN = R * N;                                                                          \
    const int hid = C.hid, K = C.K, IP = C.IP, RWd = R * W, N4 = N >> 2, W4 = W >> 2;                                         \
    const int row0 = g * NR, u0 = min(hid, g * upk), u1 = min(hid, u0 + upk), nU = u1 - u0;                                   \
    float* sPart = smem + L.part; float* sCS = smem + L.CS; float* sGL = smem + L.GL; float* sLt = smem + L.LT;               \
    float* sI = smem + L.I; float* sDX = smem + L.DX;                                                                         \
    float* sWW = smem + L.WW; float* sWWp = smem + L.WWp; float* sU = smem + L.U; float* sUp = smem + L.Up;                   \
    float* sPp = smem + L.Pp; float* sCW = smem + L.CW; float* sAL = smem + L.AL; float* sSIMw = smem + L.SIMw;               \
    float* sDWW = smem + L.DWW; float* sDCW = smem + L.DCW; float* sDA = smem + L.DA; float* sgP = smem + L.gP;               \
    float* sDPp = smem + L.DPp; float* sgU = smem + L.gU; float* sgUn = smem + L.gUn; float* sNU = smem + L.NU;               \
    unsigned long long* sKEY = reinterpret_cast<unsigned long long*>(smem + L.KEY);                                          \
    int* sRank = reinterpret_cast<int*>(smem + L.RANK);                                                                       \
    float* sRWp = smem + L.RWp; float* sRWt = smem + L.RWt; float* sgRW = smem + L.gRW; float* sG = smem + L.G; float* sDRWp = smem + L.DRWp;             \
    float* sDSIM = smem + L.DSIM; float* sSIMr = smem + L.SIMr;                                                               \
    float* sGZ = smem + L.GZ; float* sDR = smem + L.DR; float* sDHC = smem + L.DHC; float* sDG = smem + L.DG;                 \
    float* sgC = smem + L.gC; float* sSC = smem + L.SC; int* sAbort = reinterpret_cast<int*>(sSC + 120);                       \
    f32x4* sGL4 = reinterpret_cast<f32x4*>(sGL); f32x4* sLt4 = reinterpret_cast<f32x4*>(sLt);                                 \
    (void)k; (void)K; (void)IP; (void)RWd; (void)N4; (void)W4; (void)u1; (void)nU; (void)RN; (void)sPart; (void)sCS; (void)sGL;\
    (void)sLt; (void)sI; (void)sDX; (void)sWW; (void)sWWp; (void)sU; (void)sUp; (void)sPp; (void)sCW; (void)sAL; (void)sSIMw; \
    (void)sDWW; (void)sDCW; (void)sDA; (void)sgP; (void)sDPp; (void)sgU; (void)sgUn; (void)sNU; (void)sKEY; (void)sRank;      \
    (void)sRWp; (void)sRWt; (void)sgRW; (void)sG; (void)sDRWp; (void)sDSIM; (void)sSIMr; (void)sGZ; (void)sDR; (void)sDHC; (void)sDG;     \
    (void)sgC; (void)sAbort; (void)sGL4; (void)sLt4; (void)row0; (void)u0

template <bool FIX>
__global__ __launch_bounds__(CT) void dnc_cluster_bwd_kernel(DncClBwdArgs a0) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef const __attribute__((address_space(4))) DncClBwdArgs* ArgsK;
    const ArgsK ak0 = (ArgsK)__builtin_amdgcn_kernarg_segment_ptr();
    const int tid0 = threadIdx.x;
    const int kk0 = FIX ? kDncClusterFixCfg.k : a0.c.k;
    int b, g;
    if (a0.xcd_local) {
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

    // register-resident d(memory) and memory rows of the current step: lane gl of group grp holds float4 column gl of
    // rows grp + 32 q
    const int gl0 = tid0 & 15, grp0 = tid0 >> 4;
    f32x4 gMr[MAXQ], Mt[MAXQ];     // Mt: M_t during B2..B4, then reloaded with M_{t-1} (B7, B10b) = the next iteration's M_t

    // ---- carried gradients: zero (the loss depends on the outputs only) or what the following segment left behind
    {
        const DncClBwdArgs& a = a0;
        const DncClusterCfg C = FIX ? kDncClusterFixCfg : a.c;
        const DncClBwdGeo Q = FIX ? kDncClFixBwdGeo : a.q;
        const DncClBwdLds L = FIX ? kDncClFixBwdLds : a.lds;
        CL_BWD_VIEWS();
        if (tid0 == 0) *sAbort = 0;
        float* cy = a.gcarry ? a.gcarry + (size_t)b * (2 * N + RN + Q.ldkT + hid) : nullptr;
        const bool cin = cy && a.carry_in;
        for (int i = tid0; i < N; i += CT) { sgP[i] = cin ? cy[i] : 0.f; sgU[i] = cin ? cy[N + i] : 0.f; }
        for (int i = tid0; i < RN; i += CT) sgRW[i] = cin ? cy[2 * N + i] : 0.f;
        for (int i = tid0; i < Q.ldkT; i += CT) sGZ[i] = (cin && i < K) ? cy[2 * N + RN + i] : 0.f;
        for (int i = tid0; i < nU; i += CT) sgC[i] = cin ? cy[2 * N + RN + Q.ldkT + u0 + i] : 0.f;
        const f32x4* gG4 = reinterpret_cast<const f32x4*>(a.gL + ((size_t)b * N + row0) * N);
        for (int i = tid0; i < NR * N4; i += CT) {
            const int r = i / N4, qq = i - r * N4;
            sGL4[r * N4 + (qq ^ (r & 7))] = gG4[i];
        }
        const size_t btl = (size_t)b * S + (S - 1);
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            gMr[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            Mt[q] = gMr[q];
            const int n = grp0 + 32 * q;
            if (q < Q.NQ && gl0 < W4) {
                gMr[q] = reinterpret_cast<const f32x4*>(a.gM + ((size_t)b * N + n) * W)[gl0];
                Mt[q] = reinterpret_cast<const f32x4*>(a.rec_M + (btl * N + n) * W)[gl0];
            }
        }
    }
    __syncthreads();
    // same-XCD fast form of the hand-offs (dnc_cluster.h): decided per cluster by a handshake, never assumed
    bool plain = false;
    if (a0.xcd_local) {
        int* const sw = reinterpret_cast<int*>(smem + (FIX ? kDncClFixBwdLds.SC : a0.lds.SC)) + 121;
        const int same = cl_same_xcd(a0.xcc + (size_t)b * kk0, g, kk0, a0.err, sw - 1, sw, t_start, tid0);
        if (same < 0) return;
        plain = __builtin_amdgcn_readfirstlane(same) != 0;
    }
#ifdef NTK_CL_PROF
    if (blockIdx.x == 0 && tid0 == 0) for (int i = 0; i < 20; ++i) g_clb_prof[i] = 0;
    unsigned long long prof_last = __builtin_amdgcn_s_memtime();
#endif
    // the small per-step records are requested one step ahead (they stream from HBM: a step's own loads would sit on
    // its critical path): pf_* hold step t's values when step t starts
    float pf_ifc = 0.f, pf_rwp[2] = {0.f, 0.f}, pf_rwt[2] = {0.f, 0.f}, pf_ww = 0.f, pf_u = 0.f, pf_cw = 0.f, pf_al = 0.f,
          pf_wwp = 0.f, pf_up = 0.f, pf_pp = 0.f;
    float pf_ypre = 0.f, pf_dout = 0.f;                       // B1: threads tid < O
    float pf_cr[4] = {0.f, 0.f, 0.f, 0.f}, pf_fv[4] = {0.f, 0.f, 0.f, 0.f}, pf_bv[4] = {0.f, 0.f, 0.f, 0.f};   // B3: wave i < R, slots lane + 64 j
    f32x4 pf_gates = {0.f, 0.f, 0.f, 0.f};                    // B15: threads tid < nU
    float pf_c = 0.f, pf_cprev = 0.f;
    auto prefetch_records = [&](auto ap, int t, int N, int RN, int IP) {      // ap: by-value block or kernarg (address space 4) pointer
        const size_t bt = (size_t)b * S + t;
        const float* p_wwp = (t > 0) ? ap->rec_ww + (bt - 1) * N : ap->ww0 + (size_t)b * N;
        const float* p_up = (t > 0) ? ap->rec_u + (bt - 1) * N : ap->usage0 + (size_t)b * N;
        const float* p_pp = (t > 0) ? ap->rec_p + (bt - 1) * N : ap->prec0 + (size_t)b * N;
        const float* p_rwp = (t > 0) ? ap->rec_rw + (bt - 1) * RN : ap->rw0 + (size_t)b * RN;
        if (tid0 < IP) pf_ifc = ap->rec_ifc[bt * IP + tid0];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid0 + u * CT;
            if (i < RN) { pf_rwp[u] = p_rwp[i]; pf_rwt[u] = ap->rec_rw[bt * RN + i]; }
        }
        if (tid0 < N) {
            pf_ww = ap->rec_ww[bt * N + tid0]; pf_u = ap->rec_u[bt * N + tid0]; pf_cw = ap->rec_cw[bt * N + tid0];
            pf_al = ap->rec_al[bt * N + tid0]; pf_wwp = p_wwp[tid0]; pf_up = p_up[tid0]; pf_pp = p_pp[tid0];
        }
        const int O_ = FIX ? kDncClusterFixCfg.O : ap->c.O, R_ = RN / N;
        if (tid0 < O_) { pf_ypre = ap->rec_ypre[bt * O_ + tid0]; pf_dout = ap->dout[bt * O_ + tid0]; }
        if ((tid0 >> 6) < R_) {
            const int i = tid0 >> 6, ln = tid0 & 63;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = ln + 64 * j;
                if (n < N) { pf_cr[j] = ap->rec_cr[bt * RN + i * N + n]; pf_fv[j] = ap->rec_fwd[bt * RN + i * N + n]; pf_bv[j] = ap->rec_bwd[bt * RN + i * N + n]; }
            }
        }
    };
    auto prefetch_gates = [&](auto ap, int t) {               // B15's records of step t (own hidden units)
        const size_t bt = (size_t)b * S + t;
        const int hid_ = FIX ? kDncClusterFixCfg.hid : ap->c.hid, upk_ = FIX ? kDncClusterFixCfg.upk : ap->c.upk;
        const int u0_ = min(hid_, g * upk_), nU_ = min(hid_, u0_ + upk_) - u0_;
        if (tid0 < nU_) {
            const int u = u0_ + tid0;
            const float* p_cprev = (t > 0) ? ap->rec_c + (bt - 1) * hid_ : ap->hc0 + (size_t)b * 2 * hid_ + hid_;
            pf_gates = reinterpret_cast<const f32x4*>(ap->rec_gates)[bt * hid_ + u];
            pf_c = ap->rec_c[bt * hid_ + u];
            pf_cprev = p_cprev[u];
        }
    };
    {
        const DncClusterCfg C0 = FIX ? kDncClusterFixCfg : a0.c;
        prefetch_records(&a0, S - 1, C0.N, C0.R * C0.N, C0.IP);
    }

    for (int t = S - 1; t >= 0; --t) {
        ArgsK ak = ak0;
        asm volatile("" : "+s"(ak));
        const auto& a = *ak;
        DncClusterCfg C = kDncClusterFixCfg;
        DncClBwdGeo Q = kDncClFixBwdGeo;
        DncClBwdLds L = kDncClFixBwdLds;
        if constexpr (!FIX) {
            __builtin_memcpy(&C, (const void*)&a.c, sizeof(C));
            __builtin_memcpy(&Q, (const void*)&a.q, sizeof(Q));
            __builtin_memcpy(&L, (const void*)&a.lds, sizeof(L));
        }
        CL_BWD_VIEWS();
        const float clipv = a.clip;
        int NQ = Q.NQ;
        if constexpr (FIX) asm volatile("" : "+s"(NQ));     // opaque: the per-row guards stay branches, so the scheduler cannot hoist all rows' loads at once
        const int slot0 = Q.slot0, slot1 = Q.slot1;
        float* mb0 = a.mbox + (size_t)b * 2 * k * ((size_t)slot0 + slot1);        // [parity][g][slot0]
        float* mb1 = mb0 + (size_t)2 * k * slot0;
        unsigned* fl0 = a.flags + (size_t)b * 2 * k;
        unsigned* fl1 = fl0 + k;
        int tid_op = tid0;
        asm volatile("" : "+v"(tid_op));
        const int tid = tid_op, lane = tid & 63, wave = tid >> 6;
        const int gl = tid & 15, grp = tid >> 4;
        const size_t bt = (size_t)b * S + t;
        const unsigned epoch = (unsigned)(S - t);
        const int par = t & 1;
        const float* Ltg = a.rec_L + (bt * N + row0) * N;
        const float* Lpg = (t > 0) ? a.rec_L + ((bt - 1) * N + row0) * N : a.link0 + ((size_t)b * N + row0) * N;

        CLB_STAMP(0);
        // ------------------------------------------------------------ this step's records: registers -> LDS; next step's requested
        {
            if (tid < IP) { sI[tid] = pf_ifc; sDX[tid] = 0.f; }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = tid + u * CT;
                if (i < RN) { sRWp[i] = pf_rwp[u]; sRWt[i] = pf_rwt[u]; }
            }
            {
#pragma clang fp contract(off)
                if (tid < N) {
                    const float u = pf_u;
                    sWW[tid] = pf_ww;
                    sU[tid] = u;
                    sCW[tid] = pf_cw;
                    sAL[tid] = pf_al;
                    sWWp[tid] = pf_wwp;
                    sUp[tid] = pf_up;
                    sPp[tid] = pf_pp;
                    const float nu = 1.0f - (EPS + (1.0f - EPS) * u);             // exactly the forward kernel's expression
                    sNU[tid] = nu;
                    sKEY[tid] = ((unsigned long long)__float_as_uint(nu) << 32) | (unsigned)(0xFFFF - tid);
                }
            }
        }
        if (tid < 64) sSC[tid] = 0.f;
        if (tid < C.OP) {                         // B1: output clip + linear
            float gy = 0.f;
            if (tid < C.O) gy = (clipv <= 0.f || fabsf(pf_ypre) < clipv) ? pf_dout : 0.f;
            sSC[32 + tid] = gy;
            if (g == 0) a.dypre[bt * C.OP + tid] = gy;
        }
        __syncthreads();
        CLB_STAMP(1);
        for (int kk = tid; kk < C.Ky; kk += CT) {
            float s = 0.f;
            for (int o = 0; o < C.O; ++o) s += a.Wy[(size_t)kk * C.OP + o] * sSC[32 + o];
            if (kk < hid) sDHC[kk] = sGZ[RWd + kk] + s;       // carried d(clipped h) + this step's output path
            else sDR[kk - hid] = sGZ[kk - hid] + s;           // carried d(reads) + output path
        }
        if (wave <= R) {                                      // key norms: sSC[0..R-1] = |kr_i|, sSC[R] = |kw|
            const float* kp = (wave < R) ? sI + C.oKr + wave * W : sI + C.oKw;
            float ss = 0.f;
            for (int w = lane; w < W; w += 64) ss += kp[w] * kp[w];
            ss = wave_sum(ss);
            if (lane == 0) sSC[wave] = sqrtf(ss + EPS);
        }
        // rank of every slot in the usage order (independent of the gradients: done here, used in B9)
        if (tid < C.nslA * N) {
            const int sl = FIX ? tid / N : cl_div(tid, C.mg_N), n = tid - sl * N;
            const unsigned long long mine = sKEY[n];
            const u64x2* kp = reinterpret_cast<const u64x2*>(sKEY + sl * C.mperA);
            int cnt = 0;
#pragma unroll 2
            for (int m = 0; m < C.mperA; m += 8) {
                const u64x2 k0 = kp[(m >> 1)], k1 = kp[(m >> 1) + 1], k2 = kp[(m >> 1) + 2], k3 = kp[(m >> 1) + 3];
                cnt += (k0[0] > mine) + (k0[1] > mine) + (k1[0] > mine) + (k1[1] > mine) + (k2[0] > mine) + (k2[1] > mine) +
                       (k3[0] > mine) + (k3[1] > mine);
            }
            reinterpret_cast<int*>(sPart)[sl * N + n] = cnt;
        }
        __syncthreads();
        for (int n = tid; n < N; n += CT) {
            int rk = 0;
            for (int sl = 0; sl < C.nslA; ++sl) rk += reinterpret_cast<const int*>(sPart)[sl * N + n];
            sRank[n] = rk;
        }

        CLB_STAMP(2);
        // ------------------------------------------------------------ B2: pass 1 over M_t (registers): d(rw) from the reads, read-key scores
        // B15's records (gates, cells of the own units) of THIS step are requested here: the vector-memory counter is in
        // order, so any later load waits for earlier ones too -- B2..B4 touch only registers and LDS while these land
        prefetch_gates(ak, t);
        float nmr[MAXQ];                                       // |M_t[n]| of the rows of this group
        {
            f32x4 dr[4], kr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dr[i] = f32x4{0.f, 0.f, 0.f, 0.f}; kr[i] = dr[i];
                if (i < R && gl < W4) {
                    dr[i] = *reinterpret_cast<const f32x4*>(sDR + i * W + gl * 4);
                    const float* kp = sI + C.oKr + i * W + gl * 4;
                    kr[i] = f32x4{kp[0], kp[1], kp[2], kp[3]};
                }
            }
#pragma unroll
            for (int q = 0; q < MAXQ; ++q) {
                nmr[q] = 1.f;
                if (q < NQ) {
                    const int n = grp + 32 * q;
                    const f32x4 m = Mt[q];
                    const float nm = cl_sqrt(group_sum<16>(cl_dot4(m, m)) + EPS);
                    nmr[q] = nm;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (i < R) {
                            const float t1 = group_sum<16>(cl_dot4(dr[i], m));
                            const float dot = group_sum<16>(cl_dot4(kr[i], m));
                            if (gl == 0) {
                                sG[i * N + n] = sgRW[i * N + n] + t1;
                                sSIMr[i * N + n] = dot * cl_rcp(sSC[i] * nm + EPS);
                            }
                        }
                    }
                }
                if (FIX || q == 3) __builtin_amdgcn_sched_barrier(0);      // two halves: bounds the live set of the unrolled loop
            }
        }
        __syncthreads();
        CLB_STAMP(3);
        // ------------------------------------------------------------ B3: read-weight mix, read-content softmax (wave i = head i)
        if (wave < R) {
            const int i = wave;
            const float* rm = sI + C.oRm + i * 3;              // [backward, forward, content] (access.py:283-289)
            float p0 = 0.f, p1 = 0.f, p2 = 0.f, s1 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = lane + 64 * j;
                if (n < N) {
                    const float gg = sG[i * N + n], cr = pf_cr[j];
                    p0 += gg * pf_bv[j]; p1 += gg * pf_fv[j]; p2 += gg * cr;
                    s1 += cr * (rm[2] * gg);
                }
            }
            p0 = wave_sum(p0); p1 = wave_sum(p1); p2 = wave_sum(p2); s1 = wave_sum(s1);
            const float br = sI[C.oBr + i];
            float dbeta = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = lane + 64 * j;
                if (n < N) {
                    const float gg = sG[i * N + n];
                    const float dscore = pf_cr[j] * (rm[2] * gg - s1);
                    dbeta += dscore * sSIMr[i * N + n];
                    sDSIM[i * N + n] = dscore * br;
                }
            }
            dbeta = wave_sum(dbeta);
            if (lane == 0) {
                const float dotp = rm[0] * p0 + rm[1] * p1 + rm[2] * p2;
                sDX[C.oRm + i * 3 + 0] = rm[0] * (p0 - dotp);
                sDX[C.oRm + i * 3 + 1] = rm[1] * (p1 - dotp);
                sDX[C.oRm + i * 3 + 2] = rm[2] * (p2 - dotp);
                sDX[C.oBr + i] = dbeta * (1.0f - expf(-br));   // strengths pass through softplus
            }
        }
        __syncthreads();
        CLB_STAMP(4);
        // ------------------------------------------------------------ B4: pass 2 over M_t: d(M_t) (registers) and d(read keys)
        f32x4 accK[4];
        float accNk[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { accK[i] = f32x4{0.f, 0.f, 0.f, 0.f}; accNk[i] = 0.f; }
        {
            f32x4 dr[4], kr[4];
            float krn[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dr[i] = f32x4{0.f, 0.f, 0.f, 0.f}; kr[i] = dr[i]; krn[i] = (i < R) ? sSC[i] : 1.f;
                if (i < R && gl < W4) {
                    dr[i] = *reinterpret_cast<const f32x4*>(sDR + i * W + gl * 4);
                    const float* kp = sI + C.oKr + i * W + gl * 4;
                    kr[i] = f32x4{kp[0], kp[1], kp[2], kp[3]};
                }
            }
#pragma unroll
            for (int q = 0; q < MAXQ; ++q) {
                if (q < NQ && gl < W4) {
                    const int n = grp + 32 * q;
                    const f32x4 m = Mt[q];
                    const float nm = nmr[q];
                    f32x4 gq = gMr[q];
                    float dnm = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (i < R) {
                            const float dsim = sDSIM[i * N + n];
                            const float D = krn[i] * nm + EPS;
                            const float dot = sSIMr[i * N + n] * D;
                            const float rD = cl_rcp(D);
                            const float ddot = dsim * rD;
                            const float dD = -dsim * dot * rD * rD;
                            dnm += dD * krn[i];
                            if (gl == 0) accNk[i] += dD * nm;
                            const float rwn = sRWt[i * N + n];
                            gq += rwn * dr[i] + ddot * kr[i];
                            accK[i] += ddot * m;
                        }
                    }
                    gq += (dnm * cl_rcp(nm)) * m;
                    gMr[q] = gq;
                }
                if (FIX || q == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // column sums of B4 (d read keys): fold the four 16-lane groups of the wave, park per wave (summed at the end)
        {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { float v = accK[i][e]; v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); accK[i][e] = v; }
                float v = accNk[i]; v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); accNk[i] = v;
            }
            if (lane < 16) {
#pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(sCS + (wave * 7 + i) * 64 + lane * 4) = accK[i];
            }
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) sSC[72 + wave * 4 + i] = accNk[i];        // sSC[72 .. 72 + 8 * 4): d|kr_i| per wave
            }
        }
        CLB_STAMP(5);
        // M_t is dead from here: its registers take the memory rows of step t-1 (requested now, first used in B7)
        {
            const float* Mpg = (t > 0) ? a.rec_M + (bt - 1) * N * W : a.mem0 + (size_t)b * N * W;
#pragma unroll
            for (int q = 0; q < MAXQ; ++q)
                if (q < NQ && gl < W4) Mt[q] = reinterpret_cast<const f32x4*>(Mpg + (size_t)(grp + 32 * q) * W)[gl];
        }
        // ------------------------------------------------------------ B5: link pass over the own rows (dL in LDS, L_t and L_{t-1} records)
        {
            // (a) the recorded L_t rows go to LDS (operands of the two MFMA products) -- from HBM only on the first step of a
            //     launch or when a wave owns more than four rows: otherwise the previous iteration left its L_{t-1} rows there
            const bool tile_kept = (NR <= 4 * CW) && (t != S - 1);
            if (!tile_kept)
#pragma unroll 2
            for (int base = tid; base < NR * N4; base += CT) {
                const int r = FIX ? base / N4 : cl_div(base, C.mg_N4), qq = base - r * N4;
                sLt4[r * N4 + (qq ^ (r & 7))] = reinterpret_cast<const f32x4*>(Ltg)[base];
            }
            // (b) elementwise part: wave per row, lane per float4 of columns (N <= 256: one float4 per lane)
            float rm0[4], rm1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { rm0[i] = (i < R) ? sI[C.oRm + i * 3 + 0] : 0.f; rm1[i] = (i < R) ? sI[C.oRm + i * 3 + 1] : 0.f; }
            const int b0 = lane * 4;
            const bool colok = b0 < N;
            f32x4 wwb = {0.f, 0.f, 0.f, 0.f}, ppb = wwb, rwpb[4], dbb[4], colWW = wwb, colP = wwb;
#pragma unroll
            for (int i = 0; i < 4; ++i) { rwpb[i] = wwb; dbb[i] = wwb; }
            if (colok) {
                wwb = *reinterpret_cast<const f32x4*>(sWW + b0);
                ppb = *reinterpret_cast<const f32x4*>(sPp + b0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < R) { rwpb[i] = *reinterpret_cast<const f32x4*>(sRWp + i * N + b0); dbb[i] = rm0[i] * *reinterpret_cast<const f32x4*>(sG + i * N + b0); }
            }
            float* slot = mb0 + ((size_t)par * k + g) * slot0;            // [rowRW R x NR | rowWW NR | colRW R x N | colWW N | colP N]
            f32x4 lpv[4];
            for (int rb = wave; rb < NR; rb += 4 * CW) {          // four rows of this wave per batch: their L_{t-1} loads go first
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = rb + u * CW;
                    lpv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (colok && r < NR) lpv[u] = reinterpret_cast<const f32x4*>(Lpg + (size_t)r * N)[lane];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = rb + u * CW;
                    if (r < NR) {
                        const int ra = row0 + r;
                        const float wwa = sWW[ra];
                        float rowWW = 0.f;
                        if (colok) {
                            const int ph = r * N4 + (lane ^ (r & 7));
                            f32x4 gq = sGL4[ph];
                            const f32x4 lp = lpv[u];
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (i < R) gq += (rm1[i] * sG[i * N + ra]) * rwpb[i] + sRWp[i * N + ra] * dbb[i];
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (b0 + e == ra) gq[e] = 0.f;   // the diagonal of L_t is forced to 0
#pragma unroll
                            for (int e = 0; e < 4; ++e) rowWW += gq[e] * (ppb[e] - lp[e]);
                            colWW -= gq * lp;
                            colP += wwa * gq;
                            f32x4 gn;
#pragma unroll
                            for (int e = 0; e < 4; ++e) gn[e] = (1.0f - wwa - wwb[e]) * gq[e];
                            sGL4[ph] = gn;
                        }
                        rowWW = wave_sum(rowWW);
                        if (lane == 0) cl_store(slot + R * NR + r, rowWW, plain);
                    }
                }
            }
            if (colok) {
                *reinterpret_cast<f32x4*>(sPart + (wave * 2 + 0) * N + b0) = colWW;
                *reinterpret_cast<f32x4*>(sPart + (wave * 2 + 1) * N + b0) = colP;
            }
            __syncthreads();
            for (int idx = tid; idx < 2 * N; idx += CT) {         // fixed-order sum of the per-wave column partials
                const int which = idx >= N ? 1 : 0, c = idx - which * N;
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < CW; ++w) s += sPart[(w * 2 + which) * N + c];
                cl_store(slot + (R + 1) * NR + R * N + idx, s, plain);
            }
            __syncthreads();
        CLB_STAMP(6);
            // (c) the two read-weight products on the 4x4x1 MFMA:
            //     column sums  colRW[i][b] = sum_a dF_i[a] L_t[a][b]   (over the own rows: partial)
            //     row sums     rowRW[i][a] = sum_b dB_i[b] L_t[a][b]   (complete)
            const int strips = C.strips, NRp = C.NRp;
#pragma unroll 1
            for (int job = wave; job < 2 * strips; job += CW) {
                const int hsel = lane & 3;
                const bool hok = hsel < R;
                const float sc1 = hok ? sI[C.oRm + hsel * 3 + 1] : 0.f, sc0 = hok ? sI[C.oRm + hsel * 3 + 0] : 0.f;
                if (job < strips) {
                    const int c = 64 * job + lane;
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
                    const float* gp = sG + (hok ? hsel : 0) * N + row0;
#pragma unroll 1
                    for (int r = 0; r < NR; r += 8) {
                        float av[8], bv[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) { av[q] = sc1 * gp[r + q]; bv[q] = sLt[cl_lidx(r + q, c, N)]; }
#pragma unroll
                        for (int q = 0; q < 8; q += 2) {
                            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[q], bv[q], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[q + 1], bv[q + 1], acc1, 0, 0, 0);
                        }
                    }
                    acc0 += acc1;
#pragma unroll
                    for (int v = 0; v < 4; ++v) if (v < R) cl_store(slot + (R + 1) * NR + v * N + c, acc0[v], plain);
                } else {
                    const int rg = job - strips;
                    const int par2 = lane >> 5;
                    const float* gp = sG + (hok ? hsel : 0) * N + 64 * rg + par2;
                    for (int rb = 0; rb < NRp; rb += 32) {
                        const int row = rb + (lane & 31);
                        const bool rok = row < NR;
                        const int rowc = rok ? row : 0;
                        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll 1
                        for (int s = 0; s < 32; s += 8) {
                            float av[8], bv[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) { av[q] = sLt[cl_lidx(rowc, 64 * rg + 2 * (s + q) + par2, N)]; bv[q] = sc0 * gp[2 * (s + q)]; }
#pragma unroll
                            for (int q = 0; q < 8; q += 2) {
                                acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(rok ? av[q] : 0.f, bv[q], acc0, 0, 0, 0);
                                acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(rok ? av[q + 1] : 0.f, bv[q + 1], acc1, 0, 0, 0);
                            }
                        }
                        acc0 += acc1;
                        const int rbase = rb + 4 * ((lane >> 2) & 7);
#pragma unroll
                        for (int v = 0; v < 4; ++v) sPart[((rg * 2 + par2) * NRp + rbase + v) * 4 + hsel] = acc0[v];
                    }
                }
            }
            __syncthreads();
            for (int idx = tid; idx < R * NR; idx += CT) {
                const int i = cl_div(idx, C.mg_NR), r = idx - i * NR;
                float f = 0.f;
                for (int qq = 0; qq < 2 * strips; ++qq) f += sPart[(qq * NRp + r) * 4 + i];
                cl_store(slot + idx, f, plain);
            }
            if (NR <= 4 * CW && colok) {          // L_{t-1} rows (still in registers) become the next iteration's L_t tile
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = wave + u * CW;
                    if (r < NR) sLt4[r * N4 + (lane ^ (r & 7))] = lpv[u];
                }
            }
            cl_publish(fl0 + g, epoch, tid, plain);
        }
        CLB_STAMP(7);
        if (!cl_wait(fl0, epoch, k, a.err, sAbort, t_start, tid)) return;
        CLB_STAMP(8);
        {   // consume exchange 0: d(rw_{t-1}) (link part), d(ww_t) (link part), d(precedence_{t-1}) (link part)
            const float* base = mb0 + (size_t)par * k * slot0;
            for (int idx = tid; idx < RN; idx += CT) {
                const int i = FIX ? idx / N : cl_div(idx, C.mg_N), n = idx - i * N;
                const int og = cl_div(n, C.mg_NR);
                float pv[8];
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) pv[gg] = (gg < k) ? cl_load(base + (size_t)gg * slot0 + (R + 1) * NR + idx) : 0.f;
                float s = cl_load(base + (size_t)og * slot0 + i * NR + (n - og * NR));
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) if (gg < k) s += pv[gg];
                sDRWp[idx] = s;
            }
            for (int n = tid; n < N; n += CT) {
                const int og = cl_div(n, C.mg_NR);
                float pw[8], pp[8];
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) {
                    pw[gg] = (gg < k) ? cl_load(base + (size_t)gg * slot0 + (R + 1) * NR + RN + n) : 0.f;
                    pp[gg] = (gg < k) ? cl_load(base + (size_t)gg * slot0 + (R + 1) * NR + RN + N + n) : 0.f;
                }
                float sw = cl_load(base + (size_t)og * slot0 + R * NR + (n - og * NR)), sp = 0.f;
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) if (gg < k) { sw += pw[gg]; sp += pp[gg]; }
                sDWW[n] = sw;
                sDPp[n] = sp;
            }
        }
        CLB_STAMP(9);
        // the records that the FIRST half of the next iteration reads (top of the step .. B3) are requested now: they land
        // while B6..B11 run from registers and LDS, and the drain of the second publish finds them complete
        if (t > 0) prefetch_records(ak, t - 1, N, RN, IP);
        // ------------------------------------------------------------ B6: precedence (wave 0 computes the two scalars)
        if (wave == 0) {
            float sw = 0.f, t1 = 0.f;
            for (int n = lane; n < N; n += 64) { sw += sWW[n]; t1 += sgP[n] * sPp[n]; }
            sw = wave_sum(sw); t1 = wave_sum(t1);
            if (lane == 0) { sSC[16] = sw; sSC[17] = t1; }
        }
        __syncthreads();
        for (int n = tid; n < N; n += CT) {
            sDPp[n] += (1.0f - sSC[16]) * sgP[n];
            sDWW[n] += sgP[n] - sSC[17];
        }
        __syncthreads();
        CLB_STAMP(10);
        // ------------------------------------------------------------ B7: write backward over (dM, M_{t-1}); write-key scores
        f32x4 accE = {0.f, 0.f, 0.f, 0.f}, accV = accE;
        float nmw[MAXQ];
        {
            f32x4 ep = {0.f, 0.f, 0.f, 0.f}, vp = ep, kp = ep;
            if (gl < W4) {
                ep = *reinterpret_cast<const f32x4*>(sI + C.oE + gl * 4);
                vp = *reinterpret_cast<const f32x4*>(sI + C.oV + gl * 4);
                const float* kq = sI + C.oKw + gl * 4;
                kp = f32x4{kq[0], kq[1], kq[2], kq[3]};
            }
            const float nkw = sSC[R];
#pragma unroll
            for (int q = 0; q < MAXQ; ++q) {
                nmw[q] = 1.f;
                if (q < NQ) {
                    const int n = grp + 32 * q;
                    const f32x4 mp = Mt[q];
                    const float wwn = sWW[n];
                    f32x4 gq = gMr[q];
                    float t1 = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        t1 += gq[e] * (vp[e] - mp[e] * ep[e]);
                        accE[e] -= gq[e] * mp[e] * wwn;
                        accV[e] += gq[e] * wwn;
                        gq[e] = gq[e] * (1.0f - wwn * ep[e]);
                    }
                    gMr[q] = gq;                                   // now d(M_{t-1}) (content part added in B10)
                    t1 = group_sum<16>(t1);
                    const float dot = group_sum<16>(cl_dot4(kp, mp));
                    const float nm = cl_sqrt(group_sum<16>(cl_dot4(mp, mp)) + EPS);
                    nmw[q] = nm;
                    if (gl == 0) {
                        sDWW[n] += t1;
                        sSIMw[n] = dot * cl_rcp(nkw * nm + EPS);
                    }
                }
                if (FIX || q == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        CLB_STAMP(11);
        // ------------------------------------------------------------ B8: write-weight mix (access.py:252-257)
        {
            const float ga = sI[C.oAg], gw = sI[C.oWg];
            float* sT = sPart + C.nslA * N;                        // rank-ordered usages
            float* sS = sT + N;                                    // rank-ordered dA * a
            for (int n = tid; n < N; n += CT) {
                const float dww = sDWW[n];
                const float dA = gw * ga * dww;
                sDA[n] = dA;
                sDCW[n] = gw * (1.0f - ga) * dww;
                const int rk = sRank[n];
                sT[rk] = 1.0f - sNU[n];
                sS[rk] = dA * sAL[n];
            }
            if (wave == CW - 1) {
                float dgw = 0.f, dga = 0.f, s18 = 0.f;
                for (int n = lane; n < N; n += 64) {
                    const float dww = sDWW[n];
                    dgw += dww * (ga * sAL[n] + (1.0f - ga) * sCW[n]);
                    dga += gw * dww * (sAL[n] - sCW[n]);
                    s18 += sCW[n] * (gw * (1.0f - ga) * dww);
                }
                dgw = wave_sum(dgw); dga = wave_sum(dga); s18 = wave_sum(s18);
                if (lane == 0) { sDX[C.oWg] = dgw * gw * (1.0f - gw); sDX[C.oAg] = dga * ga * (1.0f - ga); sSC[18] = s18; }
            }
            __syncthreads();
            // ------------------------------------------------------------ B9: allocation backward in rank order
            //   a[n] = nonusage[n] * prod_{before n} usage  ->  d usage[n] = -dA[n] * prod[n] + (sum_{after n} dA a) / usage[n]
            if (wave == 0) {                                       // exclusive prefix product (as the forward pass)
                const int PER = N >> 6, bs = lane * PER;
                float ex[8], run = 1.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) if (j < PER) { ex[j] = run; run *= sT[bs + j]; }
                float inc = run;
#pragma unroll
                for (int dd = 1; dd < 64; dd <<= 1) { const float o = __shfl_up(inc, dd, 64); if (lane >= dd) inc *= o; }
                float excl = __shfl_up(inc, 1, 64);
                if (lane == 0) excl = 1.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) if (j < PER) sT[bs + j] = excl * ex[j];
            } else if (wave == 1) {                                // exclusive SUFFIX sum: S[r] = sum_{r' > r} dA a
                const int PER = N >> 6, bs = lane * PER;
                float ex[8], run = 0.f;
#pragma unroll
                for (int j = 7; j >= 0; --j) if (j < PER) { ex[j] = run; run += sS[bs + j]; }
                float inc = run;
#pragma unroll
                for (int dd = 1; dd < 64; dd <<= 1) { const float o = __shfl_down(inc, dd, 64); if (lane + dd < 64) inc += o; }
                float excl = __shfl_down(inc, 1, 64);
                if (lane == 63) excl = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) if (j < PER) sS[bs + j] = excl + ex[j];
            }
            __syncthreads();
            for (int n = tid; n < N; n += CT) {
                const int rk = sRank[n];
                const float ut = 1.0f - sNU[n];                    // sorted_usage = 1 - sorted_nonusage
                const float dut = -sDA[n] * sT[rk] + sS[rk] / ut;
                sgUn[n] = sgU[n] + (1.0f - EPS) * dut;             // total d(usage_t)
                sDCW[n] = sCW[n] * (sDCW[n] - sSC[18]);            // d(score) of the write-content softmax
            }
        }
        __syncthreads();
        if (wave == 0) {
            float dbeta = 0.f;
            for (int n = lane; n < N; n += 64) dbeta += sDCW[n] * sSIMw[n];
            dbeta = wave_sum(dbeta);
            const float bw = sI[C.oBw];
            if (lane == 0) sDX[C.oBw] = dbeta * (1.0f - expf(-bw));
        }
        CLB_STAMP(12);
        // ------------------------------------------------------------ B10b: content part of d(M_{t-1}) (registers), d(write key)
        f32x4 accKw = {0.f, 0.f, 0.f, 0.f};
        float accNkw = 0.f;
        {
            const float bw = sI[C.oBw], nk = sSC[R];
            f32x4 kp = {0.f, 0.f, 0.f, 0.f};
            if (gl < W4) { const float* kq = sI + C.oKw + gl * 4; kp = f32x4{kq[0], kq[1], kq[2], kq[3]}; }
#pragma unroll
            for (int q = 0; q < MAXQ; ++q) {
                if (q < NQ && gl < W4) {
                    const int n = grp + 32 * q;
                    const f32x4 mp = Mt[q];
                    const float nm = nmw[q];
                    const float dsim = sDCW[n] * bw;
                    const float D = nk * nm + EPS;
                    const float dot = sSIMw[n] * D;
                    const float rD = cl_rcp(D);
                    const float ddot = dsim * rD;
                    const float dD = -dsim * dot * rD * rD;
                    if (gl == 0) accNkw += dD * nm;
                    gMr[q] += ddot * kp + (dD * nk * cl_rcp(nm)) * mp;
                    accKw += ddot * mp;
                }
                if (FIX || q == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ------------------------------------------------------------ B11: usage backward (addressing.py:342-374)
        {
            float fgv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fgv[i] = (i < R) ? sI[C.oF + i] : 0.f;
            for (int n = tid; n < N; n += CT) {
                const float gq = sgUn[n];
                const float wwp = sWWp[n];
                const float u1v = sUp[n] + (1.0f - sUp[n]) * wwp;                // write weights: stop_gradient
                float rwp[4], phi = 1.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) { rwp[i] = (i < R) ? sRWp[i * N + n] : 0.f; phi *= (1.0f - fgv[i] * rwp[i]); }
                const float dphi = gq * u1v;
                sgU[n] = gq * phi * (1.0f - wwp);                                // carried d(usage_{t-1})
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < R) {
                        float oth = 1.f;
#pragma unroll
                        for (int i2 = 0; i2 < 4; ++i2) if (i2 != i) oth *= (1.0f - fgv[i2] * rwp[i2]);
                        sDRWp[i * N + n] += dphi * (-fgv[i]) * oth;
                        sDSIM[i * N + n] = dphi * (-rwp[i]) * oth;               // reuse: per-slot term of d(free_gate_i)
                    }
                }
            }
        }
        // column sums of B7 / B10b: fold the wave's groups, park per wave
        {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = accE[e]; v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); accE[e] = v;
                v = accV[e]; v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); accV[e] = v;
                v = accKw[e]; v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); accKw[e] = v;
            }
            float v = accNkw; v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            if (lane < 16) {
                *reinterpret_cast<f32x4*>(sCS + (wave * 7 + 4) * 64 + lane * 4) = accE;
                *reinterpret_cast<f32x4*>(sCS + (wave * 7 + 5) * 64 + lane * 4) = accV;
                *reinterpret_cast<f32x4*>(sCS + (wave * 7 + 6) * 64 + lane * 4) = accKw;
            }
            if (lane == 0) sSC[20 + wave] = v;                                   // sSC[20 .. 28): d|kw| per wave
        }
        __syncthreads();
        if (wave < R) {
            const int i = wave;
            float s = 0.f;
            for (int n = lane; n < N; n += 64) s += sDSIM[i * N + n];
            s = wave_sum(s);
            const float fg = sI[C.oF + i];
            if (lane == 0) sDX[C.oF + i] = s * fg * (1.0f - fg);
        }
        for (int idx = tid; idx < 7 * W; idx += CT) {                            // remaining interface gradients
            const int which = idx / W, w = idx - which * W;
            float s = 0.f;
#pragma unroll
            for (int wv = 0; wv < CW; ++wv) s += sCS[(wv * 7 + which) * 64 + w];
            if (which < 4) {
                if (which < R) {
                    float dn = 0.f;
#pragma unroll
                    for (int wv = 0; wv < CW; ++wv) dn += sSC[72 + wv * 4 + which];
                    sDX[C.oKr + which * W + w] = s + dn * sI[C.oKr + which * W + w] / sSC[which];
                }
            } else if (which == 4) {
                const float e = sI[C.oE + w];
                sDX[C.oE + w] = s * e * (1.0f - e);
            } else if (which == 5) {
                sDX[C.oV + w] = s;
            } else {
                float dn = 0.f;
#pragma unroll
                for (int wv = 0; wv < CW; ++wv) dn += sSC[20 + wv];
                sDX[C.oKw + w] = s + dn * sI[C.oKw + w] / sSC[R];
            }
        }
        for (int i = tid; i < RN; i += CT) sgRW[i] = sDRWp[i];                  // carried d(read weights_{t-1})
        for (int n = tid; n < N; n += CT) sgP[n] = sDPp[n];                     // carried d(precedence_{t-1})
        __syncthreads();
        if (g == 0) for (int c = tid; c < IP; c += CT) a.dxi[bt * IP + c] = sDX[c];

        CLB_STAMP(13);
        // ------------------------------------------------------------ B14: d(clipped h) of the own units += d(interface) . Wi^T
        //   Wi is [unit][IP]: the own units' rows are contiguous -- thread = (own unit, slice of the float4 columns)
        if (tid < Q.nslH * upk) {
            const int sl = cl_div(tid, C.mg_upk), j = tid - sl * upk;
            float acc = 0.f;
            if (j < nU) {
                const int c0 = sl * Q.cperH, c1 = min(C.icg, c0 + Q.cperH);
                const f32x4* wp = reinterpret_cast<const f32x4*>(a.Wi) + (size_t)(u0 + j) * C.icg;
                const f32x4* dx4 = reinterpret_cast<const f32x4*>(sDX);
#pragma unroll 4
                for (int c = c0; c < c1; ++c) acc += cl_dot4(dx4[c], wp[c]);
            }
            sPart[sl * upk + j] = acc;
        }
        __syncthreads();
        // ------------------------------------------------------------ B15: clip + snt.LSTM backward of the own units
        if (tid < nU) {
            const int u = u0 + tid;
            float dh = sDHC[u];
            for (int sl = 0; sl < Q.nslH; ++sl) dh += sPart[sl * upk + tid];
            const f32x4 gg = pf_gates;
            const float gi = gg[0], gj = gg[1], gf = gg[2], go = gg[3];
            const float c2 = pf_c;
            const float cprev = dnc_clip(pf_cprev, clipv);        // the recorded cell is pre-clip; the carried state was clipped
            const float tc = cl_tanh(c2);
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
            reinterpret_cast<f32x4*>(a.dgates)[bt * hid + u] = dg;
        }
        __syncthreads();
        CLB_STAMP(14);
        // ------------------------------------------------------------ B16: partial d[reads_prev ; h_prev] over the own gate columns
        {
            const int kg4 = Q.kg4, nrow = 4 * nU;
            if (tid < Q.nslZ * kg4) {
                const int sl = cl_div(tid, Q.mg_kg4), cg = tid - sl * kg4;
                const int r0 = sl * Q.nperZ, r1 = min(nrow, r0 + Q.nperZ);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                if (r0 < r1) acc = ntk_stream_matvec<4>(reinterpret_cast<const f32x4*>(a.WrT) + (size_t)(4 * u0) * kg4 + cg, kg4, sDG, r0, r1, nrow - 1);
                *reinterpret_cast<f32x4*>(sPart + sl * Q.ldkT + cg * 4) = acc;
            }
            __syncthreads();
            float* slot = mb1 + ((size_t)par * k + g) * slot1;
            for (int kk = tid; kk < Q.ldkT; kk += CT) {
                float s = 0.f;
                for (int sl = 0; sl < Q.nslZ; ++sl) s += sPart[sl * Q.ldkT + kk];
                cl_store(slot + kk, s, plain);
            }
            cl_publish(fl1 + g, epoch, tid, plain);
        }
        CLB_STAMP(15);
        if (!cl_wait(fl1, epoch, k, a.err, sAbort, t_start, tid)) return;
        CLB_STAMP(16);
        {
            const float* base = mb1 + (size_t)par * k * slot1;
            for (int kk = tid; kk < K; kk += CT) {
                float pv[8];
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) pv[gg] = (gg < k) ? cl_load(base + (size_t)gg * slot1 + kk) : 0.f;
                float s = 0.f;
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) if (gg < k) s += pv[gg];
                sGZ[kk] = s;
            }
        }
        __syncthreads();
        CLB_STAMP(17);
    }
#ifdef NTK_CL_PROF
    (void)prof_last;
#endif

    // ---- carried gradients out (segmented BPTT); d(memory) / d(link) scratch updated in place
    {
        const DncClBwdArgs& a = a0;
        const DncClusterCfg C = FIX ? kDncClusterFixCfg : a.c;
        const DncClBwdGeo Q = FIX ? kDncClFixBwdGeo : a.q;
        const DncClBwdLds L = FIX ? kDncClFixBwdLds : a.lds;
        CL_BWD_VIEWS();
        float* cy = a.gcarry ? a.gcarry + (size_t)b * (2 * N + RN + Q.ldkT + hid) : nullptr;
        if (cy) {
            if (g == 0) {
                for (int i = tid0; i < N; i += CT) { cy[i] = sgP[i]; cy[N + i] = sgU[i]; }
                for (int i = tid0; i < RN; i += CT) cy[2 * N + i] = sgRW[i];
                for (int i = tid0; i < Q.ldkT; i += CT) cy[2 * N + RN + i] = sGZ[i];
            }
            for (int i = tid0; i < nU; i += CT) cy[2 * N + RN + Q.ldkT + u0 + i] = sgC[i];
        }
        f32x4* gG4 = reinterpret_cast<f32x4*>(a.gL + ((size_t)b * N + row0) * N);
        for (int i = tid0; i < NR * N4; i += CT) {
            const int r = i / N4, qq = i - r * N4;
            gG4[i] = sGL4[r * N4 + (qq ^ (r & 7))];
        }
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int n = grp0 + 32 * q;
            if (q < Q.NQ && gl0 < W4 && n >= row0 && n < row0 + NR) reinterpret_cast<f32x4*>(a.gM + ((size_t)b * N + n) * W)[gl0] = gMr[q];
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
#ifdef NTK_CL_PROF
extern "C" int ntk_dnc_cluster_bwd_prof(unsigned long long* out20) {
    return hipMemcpyFromSymbol(out20, HIP_SYMBOL(g_clb_prof), 20 * sizeof(unsigned long long)) == hipSuccess ? NTK_OK : NTK_ERR_HIP;
}
#endif

// cluster size for the backward pass (0 = outside its range): the forward constraints plus N <= 256 (d(memory) is
// register resident: one row of 8 per 16-lane group), W <= 64, hid % 4 == 0
static int dnc_cluster_bwd_pick(int B, int N, int W, int R, int Wn, int hid, int O, int k_req, DncClusterCfg& c, DncClBwdGeo& q,
                                size_t* lds_bytes) {
    if (Wn != 1 || R < 1 || R > 4 || N < 64 || (N % 64) != 0 || N > 256 || W < 4 || (W % 4) != 0 || W > 64 || hid < 4 ||
        (hid % 4) != 0 || hid > 1024 || O < 1 || O > 16 || B < 1)
        return 0;
    for (int k = 8; k >= 1; k >>= 1) {
        if (k_req > 0 && k != k_req) continue;
        if ((long)B * k > ntk_device_cu_count()) continue;     // one workgroup per CU, all co-resident (api.cpp)
        const int NR = N / k;
        if (NR * k != N || NR < 8 || (NR % 8) != 0) continue;
        c = dnc_cluster_cfg(N, W, R, hid, O, k);
        q = dnc_cl_bwd_geo(c);
        if (c.IP > CT || c.R * c.N > 2 * CT || q.kg4 > CT || c.upk > CT) continue;
        const DncClBwdLds L = dnc_cl_bwd_lds(c, q);
        const size_t bytes = (size_t)L.total * sizeof(float);
        if (bytes > 160 * 1024) continue;
        if (lds_bytes) *lds_bytes = bytes;
        return k;
    }
    return 0;
}

extern "C" int ntk_dnc_cluster_bwd_plan(int B, int N, int W, int R, int Wn, int hid, int O, int k_request, int* k,
                                        size_t* workspace_bytes) {
    DncClusterCfg c;
    DncClBwdGeo q;
    const int kk = dnc_cluster_bwd_pick(B, N, W, R, Wn, hid, O, k_request, c, q, nullptr);
    if (k) *k = kk;
    if (workspace_bytes) *workspace_bytes = 0;
    if (kk <= 0) {
        ntk_set_error("ntk_dnc_cluster_bwd_plan: B=%d N=%d W=%d R=%d Wn=%d hid=%d is outside the cluster BPTT kernel's range "
                      "(num_writes 1, memory_size a multiple of 64 up to 256, word_size <= 64, hidden %% 4 == 0, B * k <= 256)",
                      B, N, W, R, Wn, hid);
        return NTK_ERR_UNSUPPORTED;
    }
    if (workspace_bytes) *workspace_bytes = dnc_cluster_ctrl_bytes(B, kk) + dnc_cluster_mbox_floats(B, kk, q.slot0, q.slot1) * sizeof(float);
    return NTK_OK;
}

extern "C" int ntk_dnc_cluster_bwd(int B, int S, int N, int W, int R, int Wn, int hid, int O, float clip_value, int k,
                                   const float* WrT, int ldkT, const float* Wi, const float* Wy,
                                   const float* mem0, const float* link0, const float* usage0, const float* rw0,
                                   const float* ww0, const float* prec0, const float* hc0,
                                   const float* rec_gates, const float* rec_c, const float* rec_ifc, const float* rec_u,
                                   const float* rec_ww, const float* rec_rw, const float* rec_cw, const float* rec_cr,
                                   const float* rec_al, const float* rec_p, const float* rec_fwd, const float* rec_bwd,
                                   const float* rec_M, const float* rec_L, const float* rec_ypre,
                                   const float* dout, float* gM, float* gL, float* dgates, float* dxi, float* dypre,
                                   float* gcarry, int carry_in, void* workspace, void* stream) {
    DncClBwdArgs a;
    size_t lds_bytes = 0;
    NTK_REQUIRE(B > 0 && S > 0 && k > 0, NTK_ERR_BAD_SHAPE, "ntk_dnc_cluster_bwd: B=%d S=%d k=%d", B, S, k);
    const int kk = dnc_cluster_bwd_pick(B, N, W, R, Wn, hid, O, k, a.c, a.q, &lds_bytes);
    NTK_REQUIRE(kk == k, NTK_ERR_UNSUPPORTED, "ntk_dnc_cluster_bwd: k=%d is not a valid cluster size for B=%d N=%d W=%d R=%d Wn=%d hid=%d "
                "(ask ntk_dnc_cluster_bwd_plan)", k, B, N, W, R, Wn, hid);
    NTK_REQUIRE(ldkT == a.q.ldkT, NTK_ERR_BAD_SHAPE, "ntk_dnc_cluster_bwd: ldkT=%d (expected %d = K rounded up to 4)", ldkT, a.q.ldkT);
    a.lds = dnc_cl_bwd_lds(a.c, a.q);
    a.B = B; a.S = S; a.clip = clip_value; a.carry_in = carry_in;
    NTK_REQUIRE(WrT && Wi && Wy && mem0 && link0 && usage0 && rw0 && ww0 && prec0 && hc0 && rec_gates && rec_c && rec_ifc &&
                    rec_u && rec_ww && rec_rw && rec_cw && rec_cr && rec_al && rec_p && rec_fwd && rec_bwd && rec_M && rec_L &&
                    rec_ypre && dout && gM && gL && dgates && dxi && dypre && workspace,
                NTK_ERR_BAD_PTR, "ntk_dnc_cluster_bwd: null pointer");
    NTK_REQUIRE(ntk_aligned16(WrT) && ntk_aligned16(Wi) && ntk_aligned16(rec_gates) && ntk_aligned16(rec_M) && ntk_aligned16(rec_L) &&
                    ntk_aligned16(gM) && ntk_aligned16(gL) && ntk_aligned16(dgates) && ntk_aligned16(mem0) && ntk_aligned16(link0) &&
                    ntk_aligned16(workspace),
                NTK_ERR_BAD_PTR, "ntk_dnc_cluster_bwd: 16-byte alignment");
    a.xcd_local = (B % 8) == 0 ? 1 : 0;
    a.WrT = WrT; a.Wi = Wi; a.Wy = Wy;
    a.mem0 = mem0; a.link0 = link0; a.usage0 = usage0; a.rw0 = rw0; a.ww0 = ww0; a.prec0 = prec0; a.hc0 = hc0;
    a.rec_gates = rec_gates; a.rec_c = rec_c; a.rec_ifc = rec_ifc; a.rec_u = rec_u; a.rec_ww = rec_ww; a.rec_rw = rec_rw;
    a.rec_cw = rec_cw; a.rec_cr = rec_cr; a.rec_al = rec_al; a.rec_p = rec_p; a.rec_fwd = rec_fwd; a.rec_bwd = rec_bwd;
    a.rec_M = rec_M; a.rec_L = rec_L; a.rec_ypre = rec_ypre; a.dout = dout; a.gM = gM; a.gL = gL;
    a.dgates = dgates; a.dxi = dxi; a.dypre = dypre; a.gcarry = gcarry;
    const size_t ctrl = dnc_cluster_ctrl_bytes(B, k);
    a.flags = reinterpret_cast<unsigned*>(workspace);
    a.err = a.flags + (size_t)B * 2 * k;
    a.xcc = a.err + 1;
    a.mbox = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + ctrl);
    {
        static NtkLdsAttrCache lds_cache;
        const void* const ks[] = {(const void*)dnc_cluster_bwd_kernel<false>, (const void*)dnc_cluster_bwd_kernel<true>};
        const int rc_lds = ntk_raise_lds_limit(lds_cache, ks, 2, "ntk_dnc_cluster_bwd");
        if (rc_lds != NTK_OK) return rc_lds;
    }
    hipError_t e = hipMemsetAsync(workspace, 0, dnc_cluster_ctrl_zero_bytes(B, k), (hipStream_t)stream);
    NTK_REQUIRE(e == hipSuccess, NTK_ERR_HIP, "ntk_dnc_cluster_bwd: hipMemsetAsync: %s", hipGetErrorString(e));
#ifdef NTK_DNC_BWD_GENERIC                                                  // dev build: the benchmark shape through the generic instantiation
    const bool use_fix = false;
#else
    const bool use_fix = dnc_cluster_is_fix(a.c);
#endif
    if (use_fix) dnc_cluster_bwd_kernel<true><<<B * k, CT, lds_bytes, (hipStream_t)stream>>>(a);
    else dnc_cluster_bwd_kernel<false><<<B * k, CT, lds_bytes, (hipStream_t)stream>>>(a);
    NTK_CHECK_LAUNCH("ntk_dnc_cluster_bwd");
    dnc_cluster_latch(a.err, reinterpret_cast<unsigned*>(reinterpret_cast<char*>(workspace) + ctrl - 256), stream);
    NTK_CHECK_LAUNCH("ntk_dnc_cluster_bwd (latch)");
    return NTK_OK;
}
