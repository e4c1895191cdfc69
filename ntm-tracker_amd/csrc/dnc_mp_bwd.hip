// DNC core sequence backward (full BPTT), MEMORY-PARTITIONED cluster form: k workgroups (512 threads, one per CU) per
// sequence walk the steps recorded by the forward pass in reverse.  Same arithmetic as dnc_seq_bwd.hip / dnc_cluster_bwd.hip
// (what tf.gradients computes through tf.nn.dynamic_rnn over dnc.DNC, direct_offset_output_with_dnc.py:615-620;
// non-differentiable edges: SURVEY A.4); partition: dnc_mp.h.
//
//   d(link)     N/k rows per workgroup, STREAMED: one pass per step reads the own rows of d(link) (carried, updated in place
//               in HBM), of the recorded L_t and of L_{t-1}; row sums are complete, column sums are partial and summed over
//               the workgroups by the consumers of hand-off 2;
//   d(memory)   own rows in REGISTERS for the whole launch (a group of LPR lanes owns float4 column gl of up to 8 rows),
//               next to the own memory rows of the current step, carried over from the previous iteration;
//   per slot    replicated; the controller is split by hidden units as in the forward pass.
//
// Hand-offs per step: (1) d(read weights) of the own rows through the reads, read-key scores of the own rows, rank partial
// counts; (2) link row / column sums, write-path terms of the own rows and the column sums of the memory passes; (3) the
// partial d(write key); (4) the partial d[reads ; h]_{t-1}.  Every sum over workgroups, waves and lanes has a fixed order:
// gradients are bitwise reproducible.
#include "dnc_mp.h"
#include <stdlib.h>

// Diagnostic build only (-DNTK_CL_PROF): workgroup 0's thread 0 adds s_memtime deltas per phase to g_mpb_prof (global atomics: no
// registers held across the step); ntk_dnc_mp_bwd_prof copies them out.  The stamps serialise the phases: read SHARES.
#ifdef NTK_CL_PROF
__device__ unsigned long long g_mpb_prof[32];
#define MP_STAMP(i)                                                                   \
    do {                                                                              \
        if (blockIdx.x == 0 && tid == 0) {                                            \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();             \
            __hip_atomic_fetch_add(&g_mpb_prof[i], now_ - prof_last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
            prof_last = now_;                                                         \
        }                                                                             \
    } while (0)
#else
#define MP_STAMP(i) do { } while (0)
#endif

namespace {

constexpr int MPQ = 8;        // memory rows per lane group (own rows <= 8 * 512 / LPR)

struct DncMpBwdGeo {
    int ldkT, ldhT, kg4;
    int nslZ, nperZ;          // d[reads ; h] partial: slices of the own gate rows, rows per slice
    int nslH, cperH;          // dh of the own units: slices of the interface columns (float4), float4s per slice
    int LPR, NG, NQ;          // memory rows in registers: lanes per row, lane groups, rows per group
    int RG, NP;               // link pass: row groups (waves / column halves), columns covered by the waves = 256 * NH
    int slot[MPX];
    int oRowWW, oT1, oSimw, oColRW, oColWW, oColP, oK, oNk, oE, oV;      // offsets inside slot 2 (row sums of the read weights at 0)
    unsigned mg_kg4;
};

constexpr __host__ __device__ DncMpBwdGeo dnc_mp_bwd_geo(const DncMpCfg& c) {
    DncMpBwdGeo q = {};
    q.ldkT = (c.K + 3) & ~3;
    q.ldhT = (c.hid + 3) & ~3;
    q.kg4 = q.ldkT / 4;
    q.nslZ = dnc_cluster_max(1, CLT / q.kg4);
    q.nperZ = (4 * c.upk + q.nslZ - 1) / q.nslZ;
    q.nslH = dnc_cluster_max(1, CLT / dnc_cluster_max(1, c.upk));
    if (q.nslH > c.icg) q.nslH = c.icg;
    q.cperH = (c.icg + q.nslH - 1) / q.nslH;
    int lpr = 1;
    while (lpr < c.W4) lpr <<= 1;
    q.LPR = lpr;
    q.NG = CLT / lpr;
    q.NQ = (c.NR + q.NG - 1) / q.NG;
    q.RG = (CLT / 64) / c.NH;
    q.NP = 256 * c.NH;
    q.mg_kg4 = dnc_cluster_magic(q.kg4);
    const int R = c.R, NR = c.NR, N = c.N, W = c.W;
    q.slot[0] = dnc_cluster_align4(2 * R * NR + N);
    q.oRowWW = R * NR; q.oT1 = q.oRowWW + NR; q.oSimw = q.oT1 + NR; q.oColRW = q.oSimw + NR; q.oColWW = q.oColRW + R * N;
    q.oColP = q.oColWW + N; q.oK = q.oColP + N; q.oNk = q.oK + R * W; q.oE = q.oNk + 4; q.oV = q.oE + W;
    q.slot[1] = dnc_cluster_align4(q.oV + W);
    q.slot[2] = dnc_cluster_align4(W + 4);
    q.slot[3] = dnc_cluster_align4(q.ldkT);
    return q;
}

struct DncMpBwdLds {
    int part, RP, NMr, NMw, I, DX, WW, WWp, U, Up, Pp, CW, AL, SIMw, DWW, DCW, DA, gP, DPp, gU, gUn, NU, KEY, RANK,
        RWp, RWt, gRW, G, DRWp, DSIM, SIMr, GZ, DR, DHC, DG, gC, SC, total;
};

constexpr __host__ __device__ DncMpBwdLds dnc_mp_bwd_lds(const DncMpCfg& c, const DncMpBwdGeo& q) {
    DncMpBwdLds L = {};
    const int N = c.N, RN = c.R * c.N;
    int part = 2 * N;                                                  // the two rank-ordered vectors
    part = dnc_cluster_max(part, q.RG * dnc_cluster_max(c.R, 2) * q.NP);            // link pass: per-row-group column partials (R read-weight sums, then 2)
    part = dnc_cluster_max(part, (CLT / 64) * dnc_cluster_max(c.R, 2) * c.W);       // memory passes: per-wave column sums (R key sums, then erase + write vector)
    part = dnc_cluster_max(part, q.nslZ * q.ldkT);
    part = dnc_cluster_max(part, q.nslH * c.upk);
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    L.part = take(part);
    L.RP = take(c.NH * (c.R + 1) * c.NR);                              // link pass: per-half row sums
    L.NMr = take(c.NR); L.NMw = take(c.NR);                            // |M_t[n]|, |M_{t-1}[n]| of the own rows (B2 -> B4, B7 -> B10b)
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

struct DncMpBwdArgs {
    int B, S, xcd_local, carry_in;
    float clip;
    DncMpCfg c;
    DncMpBwdGeo q;
    DncMpBwdLds lds;
    const float* WrT; const float* Wi; const float* Wy;
    const float* mem0; const float* link0; const float* usage0; const float* rw0; const float* ww0; const float* prec0;
    const float* hc0;
    const float* rec_gates; const float* rec_c; const float* rec_ifc; const float* rec_u; const float* rec_ww;
    const float* rec_rw; const float* rec_cw; const float* rec_cr; const float* rec_al; const float* rec_p;
    const float* rec_fwd; const float* rec_bwd; const float* rec_M; const float* rec_L; const float* rec_ypre;
    const float* dout;
    float* gM; float* gL; float* dgates; float* dxi; float* dypre; float* gcarry;
    float* mbox; unsigned* flags; unsigned* err; unsigned* xcc; unsigned* sticky;
};

__device__ __forceinline__ float mpb_dot4(const f32x4& x, const f32x4& y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3]; }
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

// sum over the lane groups of a wave (groups of LPR consecutive lanes; lanes with equal index inside their group are added):
// afterwards lanes < LPR hold the wave's totals
__device__ __forceinline__ float mpb_fold(float v, int LPR) {
    for (int o = LPR; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ f32x4 mpb_fold4(f32x4 v, int LPR) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = mpb_fold(v[e], LPR);
    return v;
}

#define MP_BWD_VIEWS()                                                                                                        \
    const int k = C.k, NR = C.NR, upk = C.upk;                                                                                \
    const int N = C.N, W = C.W, R = C.R, RN = R * N;                                                                          \
    const int hid = C.hid, K = C.K, IP = C.IP, RWd = R * W, N4 = C.N4, W4 = C.W4;                                             \
    const int row0 = g * NR, u0 = min(hid, g * upk), u1 = min(hid, u0 + upk), nU = u1 - u0;                                   \
    float* sPart = smem + L.part; float* sRP = smem + L.RP; float* sNMr = smem + L.NMr; float* sNMw = smem + L.NMw;           \
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
    (void)k; (void)K; (void)IP; (void)RWd; (void)N4; (void)W4; (void)u1; (void)nU; (void)RN; (void)sPart; (void)sRP; (void)sNMr; (void)sNMw; \
    (void)sI; (void)sDX; (void)sWW; (void)sWWp; (void)sU; (void)sUp; (void)sPp; (void)sCW; (void)sAL; (void)sSIMw;            \
    (void)sDWW; (void)sDCW; (void)sDA; (void)sgP; (void)sDPp; (void)sgU; (void)sgUn; (void)sNU; (void)sKEY; (void)sRank;      \
    (void)sRWp; (void)sRWt; (void)sgRW; (void)sG; (void)sDRWp; (void)sDSIM; (void)sSIMr; (void)sGZ; (void)sDR; (void)sDHC; (void)sDG;     \
    (void)sgC; (void)sAbort; (void)row0; (void)u0

template <int SH>
__global__ __launch_bounds__(CT) void dnc_mp_bwd_kernel(DncMpBwdArgs a0) {
    constexpr bool FIX = SH != 0;
    constexpr DncMpCfg kDncMpFixCfg = dnc_mp_shape_cfg(SH);
    constexpr DncMpBwdGeo kDncMpFixBwdGeo = dnc_mp_bwd_geo(kDncMpFixCfg);
    constexpr DncMpBwdLds kDncMpFixBwdLds = dnc_mp_bwd_lds(kDncMpFixCfg, kDncMpFixBwdGeo);
    // link rows a wave keeps in flight (three streams per row): 2 at 512 columns (bandwidth bound), 4 where the pass is a few
    // rows per wave and latency bound (256 columns)
    constexpr int MPB_PFL = (FIX && kDncMpFixCfg.NH == 1) ? 4 : 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef const __attribute__((address_space(4))) DncMpBwdArgs* ArgsK;
    const ArgsK ak0 = (ArgsK)__builtin_amdgcn_kernarg_segment_ptr();
    const int tid0 = threadIdx.x;
    const int kk0 = FIX ? kDncMpFixCfg.k : a0.c.k;
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

    // register-resident d(memory): lane gl of group grp holds float4 column gl of the own rows grp + NG q.  The memory rows
    // (M_t for B2 / B4, then M_{t-1} for B7 / B10b) are requested from the records where a step needs them, all rows of a
    // thread in one batch: carried over from the previous iteration they sat in registers through the link pass, the compiler
    // spilled them there and reloaded them one dependent scratch load at a time (6 us of a step in B10b alone).
    f32x4 gMr[MPQ], Mt[MPQ], Mp[MPQ];

    // ---- carried gradients: zero (the loss depends on the outputs only) or what the following segment left behind
    {
        const DncMpBwdArgs& a = a0;
        const DncMpCfg C = FIX ? kDncMpFixCfg : a.c;
        const DncMpBwdGeo Q = FIX ? kDncMpFixBwdGeo : a.q;
        const DncMpBwdLds L = FIX ? kDncMpFixBwdLds : a.lds;
        MP_BWD_VIEWS();
        if (tid0 == 0) *sAbort = 0;
        float* cy = a.gcarry ? a.gcarry + (size_t)b * (2 * N + RN + Q.ldkT + hid) : nullptr;
        const bool cin = cy && a.carry_in;
        for (int i = tid0; i < N; i += CT) { sgP[i] = cin ? cy[i] : 0.f; sgU[i] = cin ? cy[N + i] : 0.f; }
        for (int i = tid0; i < RN; i += CT) sgRW[i] = cin ? cy[2 * N + i] : 0.f;
        for (int i = tid0; i < Q.ldkT; i += CT) sGZ[i] = (cin && i < K) ? cy[2 * N + RN + i] : 0.f;
        for (int i = tid0; i < nU; i += CT) sgC[i] = cin ? cy[2 * N + RN + Q.ldkT + u0 + i] : 0.f;
        const int gl0 = tid0 & (Q.LPR - 1), grp0 = tid0 / Q.LPR;
#pragma unroll
        for (int q = 0; q < MPQ; ++q) {
            gMr[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            Mt[q] = gMr[q];
            const int nl = grp0 + Q.NG * q;
            if (q < Q.NQ && nl < NR && gl0 < W4) gMr[q] = reinterpret_cast<const f32x4*>(a.gM + ((size_t)b * N + row0 + nl) * W)[gl0];
        }
    }
    __syncthreads();
    bool plain = false;
    if (a0.xcd_local) {
        int* const sw = reinterpret_cast<int*>(smem + (FIX ? kDncMpFixBwdLds.SC : a0.lds.SC)) + 121;
        const int same = cl_same_xcd(a0.xcc + (size_t)b * kk0, g, kk0, a0.err, sw - 1, sw, t_start, tid0);
        if (same < 0) { if (tid0 == 0) __hip_atomic_store(a0.sticky, 1u, NTK_RLX, NTK_AGENT); return; }
        plain = __builtin_amdgcn_readfirstlane(same) != 0;
    }

#ifdef NTK_CL_PROF
    unsigned long long prof_last = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && tid0 == 0) for (int i = 0; i < 32; ++i) g_mpb_prof[i] = 0;
#endif
    for (int t = S - 1; t >= 0; --t) {
        ArgsK ak = ak0;
        asm volatile("" : "+s"(ak));
        const auto& a = *ak;
        DncMpCfg C = kDncMpFixCfg;
        DncMpBwdGeo Q = kDncMpFixBwdGeo;
        DncMpBwdLds L = kDncMpFixBwdLds;
        if constexpr (!FIX) {
            __builtin_memcpy(&C, (const void*)&a.c, sizeof(C));
            __builtin_memcpy(&Q, (const void*)&a.q, sizeof(Q));
            __builtin_memcpy(&L, (const void*)&a.lds, sizeof(L));
        }
        MP_BWD_VIEWS();
        const float clipv = a.clip;
        const int LPR = Q.LPR, NG = Q.NG;
        int NQ = Q.NQ;
        if constexpr (FIX) asm volatile("" : "+s"(NQ));     // opaque: the per-row guards stay branches (bounded live ranges)
        const int sl1 = Q.slot[0], sl2 = Q.slot[1], sl3 = Q.slot[2], sl4 = Q.slot[3];
        float* mb1 = a.mbox + (size_t)b * 2 * k * ((size_t)sl1 + sl2 + sl3 + sl4);     // [parity][g][slot] per hand-off
        float* mb2 = mb1 + (size_t)2 * k * sl1;
        float* mb3 = mb2 + (size_t)2 * k * sl2;
        float* mb4 = mb3 + (size_t)2 * k * sl3;
        unsigned* fl1 = a.flags + (size_t)b * MPX * k;
        unsigned* fl2 = fl1 + k; unsigned* fl3 = fl2 + k; unsigned* fl4 = fl3 + k;
        int tid_op = tid0;
        asm volatile("" : "+v"(tid_op));
        const int tid = tid_op, lane = tid & 63, wave = tid >> 6;
        const int gl = tid & (LPR - 1), grp = tid / LPR;
        const size_t bt = (size_t)b * S + t;
        const unsigned epoch = (unsigned)(S - t);
        const int par = t & 1;
        const float* Ltg = a.rec_L + (bt * N + row0) * N;
        const float* Lpg = (t > 0) ? a.rec_L + ((bt - 1) * N + row0) * N : a.link0 + ((size_t)b * N + row0) * N;
        float* gLg = a.gL + ((size_t)b * N + row0) * N;
        float* slot1 = mb1 + ((size_t)par * k + g) * sl1;      // [G own R x NR | read-key scores own R x NR | rank partial N]
        float* slot2 = mb2 + ((size_t)par * k + g) * sl2;
        float* slot3 = mb3 + ((size_t)par * k + g) * sl3;
        float* slot4 = mb4 + ((size_t)par * k + g) * sl4;

        MP_STAMP(0);       // loop top
        // Memory rows in registers: Mt = own rows of M_t (B2, B4), Mp = own rows of M_{t-1} (B7; requested again behind hand-off 2
        // for B10b, then carried over as the next step's Mt).  The next step's Mp is requested right after B10b, so a step
        // starts with both in place; only the first step of a launch loads them here.
        const float* Mpg = (t > 0) ? a.rec_M + ((bt - 1) * N + row0) * W : a.mem0 + ((size_t)b * N + row0) * W;
        if (t == S - 1) {
            const float* Mtg = a.rec_M + (bt * N + row0) * W;
#pragma unroll
            for (int q = 0; q < MPQ; ++q) {
                const int nl = grp + NG * q;
                Mt[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                Mp[q] = Mt[q];
                if (q < NQ && nl < NR && gl < W4) {
                    Mt[q] = reinterpret_cast<const f32x4*>(Mtg + (size_t)nl * W)[gl];
                    Mp[q] = reinterpret_cast<const f32x4*>(Mpg + (size_t)nl * W)[gl];
                }
            }
        }
        // ------------------------------------------------------------ this step's records -> LDS
        float pf_cr[8], pf_fv[8], pf_bv[8];                   // B3: wave i < R, slots lane + 64 j
        {
            const float* p_wwp = (t > 0) ? a.rec_ww + (bt - 1) * N : a.ww0 + (size_t)b * N;
            const float* p_up = (t > 0) ? a.rec_u + (bt - 1) * N : a.usage0 + (size_t)b * N;
            const float* p_pp = (t > 0) ? a.rec_p + (bt - 1) * N : a.prec0 + (size_t)b * N;
            const float* p_rwp = (t > 0) ? a.rec_rw + (bt - 1) * RN : a.rw0 + (size_t)b * RN;
            for (int c = tid; c < IP; c += CT) { sI[c] = a.rec_ifc[bt * IP + c]; sDX[c] = 0.f; }
            for (int i = tid; i < RN; i += CT) { sRWp[i] = p_rwp[i]; sRWt[i] = a.rec_rw[bt * RN + i]; }
            {
#pragma clang fp contract(off)
                for (int n = tid; n < N; n += CT) {
                    const float u = a.rec_u[bt * N + n];
                    sWW[n] = a.rec_ww[bt * N + n];
                    sU[n] = u;
                    sCW[n] = a.rec_cw[bt * N + n];
                    sAL[n] = a.rec_al[bt * N + n];
                    sWWp[n] = p_wwp[n];
                    sUp[n] = p_up[n];
                    sPp[n] = p_pp[n];
                    const float nu = 1.0f - (EPS + (1.0f - EPS) * u);             // exactly the forward kernel's expression
                    sNU[n] = nu;
                    sKEY[n] = ((unsigned long long)__float_as_uint(nu) << 32) | (unsigned)(0xFFFF - n);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                pf_cr[j] = 0.f; pf_fv[j] = 0.f; pf_bv[j] = 0.f;
                const int n = lane + 64 * j;
                if (wave < R && n < N) {
                    pf_cr[j] = a.rec_cr[bt * RN + wave * N + n];
                    pf_fv[j] = a.rec_fwd[bt * RN + wave * N + n];
                    pf_bv[j] = a.rec_bwd[bt * RN + wave * N + n];
                }
            }
        }
        if (tid < 64) sSC[tid] = 0.f;
        if (tid < C.OP) {                         // B1: output clip + linear
            float gy = 0.f;
            if (tid < C.O) {
                const float ypre = a.rec_ypre[bt * C.O + tid];
                gy = (clipv <= 0.f || fabsf(ypre) < clipv) ? a.dout[bt * C.O + tid] : 0.f;
            }
            sSC[32 + tid] = gy;
            if (g == 0) a.dypre[bt * C.OP + tid] = gy;
        }
        __syncthreads();
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
        // rank of every slot among the own N/k keys (partial count; summed after hand-off 1)
        for (int n = tid; n < N; n += CT) {
            const unsigned long long mine = sKEY[n];
            const u64x2* kp = reinterpret_cast<const u64x2*>(sKEY + g * C.mperA);
            int cnt = 0;
            for (int m = 0; m < C.mperA; m += 8) {
                const u64x2 k0 = kp[(m >> 1)], k1 = kp[(m >> 1) + 1], k2 = kp[(m >> 1) + 2], k3 = kp[(m >> 1) + 3];
                cnt += (k0[0] > mine) + (k0[1] > mine) + (k1[0] > mine) + (k1[1] > mine) + (k2[0] > mine) + (k2[1] > mine) +
                       (k3[0] > mine) + (k3[1] > mine);
            }
            cl_store(slot1 + 2 * R * NR + n, __int_as_float(cnt), plain);
        }
        __syncthreads();
        MP_STAMP(1);       // records -> LDS, B1, key norms, rank partial
        // ------------------------------------------------------------ B2: pass 1 over M_t (registers): d(rw) through the reads, read-key scores
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
            for (int q = 0; q < MPQ; ++q) {
                if (q < NQ) {
                    const int nl = grp + NG * q;
                    const bool rok = nl < NR;
                    const f32x4 m = Mt[q];
                    const float nm = cl_sqrt(group_sum_rt(mpb_dot4(m, m), LPR) + EPS);
                    if (gl == 0 && rok) sNMr[nl] = nm;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (i < R) {
                            const float t1 = group_sum_rt(mpb_dot4(dr[i], m), LPR);
                            const float dot = group_sum_rt(mpb_dot4(kr[i], m), LPR);
                            if (gl == 0 && rok) {
                                cl_store(slot1 + i * NR + nl, sgRW[i * N + row0 + nl] + t1, plain);
                                cl_store(slot1 + (R + i) * NR + nl, dot * cl_rcp(sSC[i] * nm + EPS), plain);
                            }
                        }
                    }
                }
            }
        }
        MP_STAMP(2);       // B2
        cl_publish(fl1 + g, epoch, tid, plain);
        // B15's records (gates, cells of the own units) of this step: requested here, used after the link pass
        f32x4 pf_gates = {0.f, 0.f, 0.f, 0.f};
        float pf_c = 0.f, pf_cprev = 0.f;
        if (tid < nU) {
            const int u = u0 + tid;
            const float* p_cprev = (t > 0) ? a.rec_c + (bt - 1) * hid : a.hc0 + (size_t)b * 2 * hid + hid;
            pf_gates = reinterpret_cast<const f32x4*>(a.rec_gates)[bt * hid + u];
            pf_c = a.rec_c[bt * hid + u];
            pf_cprev = p_cprev[u];
        }
        MP_STAMP(3);       // publish 1 + gate record requests
        if (!mp_wait(fl1, epoch, k, a.err, a.sticky, sAbort, tid)) return;
        MP_STAMP(4);       // wait 1
        {   // consume hand-off 1: G and the read-key scores of every slot; ranks (16-byte loads: four consecutive slots per thread)
            const __amdgpu_buffer_rsrc_t rs = mp_rsrc(mb1 + (size_t)par * k * sl1, (size_t)k * sl1);
            for (int i4 = tid; i4 < (RN >> 2); i4 += CT) {
                const int idx = 4 * i4;
                const int i = cl_div(idx, C.mg_N), n = idx - i * N;
                const int og = cl_div(n, C.mg_NR), rl = n - og * NR;
                *reinterpret_cast<f32x4*>(sG + idx) = mp_load4(rs, og * sl1 + i * NR + rl);
                *reinterpret_cast<f32x4*>(sSIMr + idx) = mp_load4(rs, og * sl1 + (R + i) * NR + rl);
            }
            for (int n4 = tid; n4 < N4; n4 += CT) {
                i32x4 rk = {0, 0, 0, 0};
                i32x4 pv[8];
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) pv[gg] = (gg < k) ? mp_load4i(rs, gg * sl1 + 2 * R * NR + 4 * n4) : rk;
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) if (gg < k) rk += pv[gg];
                *reinterpret_cast<i32x4*>(sRank + 4 * n4) = rk;
            }
        }
        __syncthreads();
        MP_STAMP(5);       // consume 1
        // ------------------------------------------------------------ B3: read-weight mix, read-content softmax (wave i = head i)
        if (wave < R) {
            const int i = wave;
            const float* rm = sI + C.oRm + i * 3;              // [backward, forward, content] (access.py:283-289)
            float p0 = 0.f, p1 = 0.f, p2 = 0.f, s1 = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
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
            for (int j = 0; j < 8; ++j) {
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
        MP_STAMP(6);       // B3
        // ------------------------------------------------------------ B4: pass 2 over M_t: d(M_t) (registers) and d(read keys) of the own rows
        {
            f32x4 accK[4];
            float accNk[4];
            f32x4 dr[4], kr[4];
            float krn[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                accK[i] = f32x4{0.f, 0.f, 0.f, 0.f}; accNk[i] = 0.f;
                dr[i] = accK[i]; kr[i] = accK[i]; krn[i] = (i < R) ? sSC[i] : 1.f;
                if (i < R && gl < W4) {
                    dr[i] = *reinterpret_cast<const f32x4*>(sDR + i * W + gl * 4);
                    const float* kp = sI + C.oKr + i * W + gl * 4;
                    kr[i] = f32x4{kp[0], kp[1], kp[2], kp[3]};
                }
            }
#pragma unroll
            for (int q = 0; q < MPQ; ++q) {
                const int nl = grp + NG * q;
                if (q < NQ && nl < NR && gl < W4) {
                    const int n = row0 + nl;
                    const f32x4 m = Mt[q];
                    const float nm = sNMr[nl];
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
            }
            // column sums (d read keys) over the own rows: lane groups of the wave, then the waves in a fixed order
#pragma unroll
            for (int i = 0; i < 4; ++i) { accK[i] = mpb_fold4(accK[i], LPR); accNk[i] = mpb_fold(accNk[i], LPR); }
            if (lane < LPR && gl < W4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (i < R) *reinterpret_cast<f32x4*>(sPart + (wave * R + i) * W + gl * 4) = accK[i];
            }
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) sSC[72 + wave * 4 + i] = accNk[i];
            }
            __syncthreads();
            for (int idx = tid; idx < RWd; idx += CT) {
                float s = 0.f;
#pragma unroll
                for (int wv = 0; wv < CW; ++wv) s += sPart[wv * RWd + idx];
                cl_store(slot2 + Q.oK + idx, s, plain);
            }
            if (tid < 4) {
                float s = 0.f;
#pragma unroll
                for (int wv = 0; wv < CW; ++wv) s += sSC[72 + wv * 4 + tid];
                cl_store(slot2 + Q.oNk + tid, s, plain);
            }
        }
        MP_STAMP(7);       // B4 + its column sums
        __syncthreads();                                      // sPart free again
        // ------------------------------------------------------------ B7: write backward over (dM, M_{t-1}) of the own rows; write-key scores
        {
            f32x4 accE = {0.f, 0.f, 0.f, 0.f}, accV = accE;
            f32x4 ep = {0.f, 0.f, 0.f, 0.f}, vp = ep, kp = ep;
            if (gl < W4) {
                ep = *reinterpret_cast<const f32x4*>(sI + C.oE + gl * 4);
                vp = *reinterpret_cast<const f32x4*>(sI + C.oV + gl * 4);
                const float* kq = sI + C.oKw + gl * 4;
                kp = f32x4{kq[0], kq[1], kq[2], kq[3]};
            }
            const float nkw = sSC[R];
#pragma unroll
            for (int q = 0; q < MPQ; ++q) {
                if (q < NQ) {
                    const int nl = grp + NG * q;
                    const bool rok = nl < NR;
                    const f32x4 mp = Mp[q];
                    const float wwn = rok ? sWW[row0 + nl] : 0.f;
                    f32x4 gq = gMr[q];
                    float t1 = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        t1 += gq[e] * (vp[e] - mp[e] * ep[e]);
                        accE[e] -= gq[e] * mp[e] * wwn;
                        accV[e] += gq[e] * wwn;
                        gq[e] = gq[e] * (1.0f - wwn * ep[e]);
                    }
                    gMr[q] = gq;                                   // now d(M_{t-1}) (content part added in B10b)
                    t1 = group_sum_rt(t1, LPR);
                    const float dot = group_sum_rt(mpb_dot4(kp, mp), LPR);
                    const float nm = cl_sqrt(group_sum_rt(mpb_dot4(mp, mp), LPR) + EPS);
                    if (gl == 0 && rok) {
                        sNMw[nl] = nm;
                        cl_store(slot2 + Q.oT1 + nl, t1, plain);
                        cl_store(slot2 + Q.oSimw + nl, dot * cl_rcp(nkw * nm + EPS), plain);
                    }
                }
            }
            accE = mpb_fold4(accE, LPR);
            accV = mpb_fold4(accV, LPR);
            if (lane < LPR && gl < W4) {
                *reinterpret_cast<f32x4*>(sPart + (wave * 2 + 0) * W + gl * 4) = accE;
                *reinterpret_cast<f32x4*>(sPart + (wave * 2 + 1) * W + gl * 4) = accV;
            }
            __syncthreads();
            for (int idx = tid; idx < 2 * W; idx += CT) {
                const int which = idx / W, w = idx - which * W;
                float s = 0.f;
#pragma unroll
                for (int wv = 0; wv < CW; ++wv) s += sPart[(wv * 2 + which) * W + w];
                cl_store(slot2 + (which ? Q.oV : Q.oE) + w, s, plain);
            }
        }
        __syncthreads();                                      // sPart free again
        MP_STAMP(8);       // B7 (before the link pass: the memory rows need not live through it)
        // ------------------------------------------------------------ B5: link pass over the own rows (d(link), L_t, L_{t-1}: HBM streams)
        {
            const int NH = C.NH, RG = Q.RG, NP = Q.NP;
            const int hh = (NH == 2) ? (wave & 1) : 0, rg = (NH == 2) ? (wave >> 1) : wave;
            const int c4 = lane + 64 * hh;
            const bool colok = c4 < N4;
            const int b0 = 4 * c4;
            float rm0[4], rm1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { rm0[i] = (i < R) ? sI[C.oRm + i * 3 + 0] : 0.f; rm1[i] = (i < R) ? sI[C.oRm + i * 3 + 1] : 0.f; }
            f32x4 wwb = {0.f, 0.f, 0.f, 0.f}, ppb = wwb, rwpb[4], dbb[4], colRW[4], colWW = wwb, colP = wwb;
#pragma unroll
            for (int i = 0; i < 4; ++i) { rwpb[i] = wwb; dbb[i] = wwb; colRW[i] = wwb; }
            if (colok) {
                wwb = *reinterpret_cast<const f32x4*>(sWW + b0);
                ppb = *reinterpret_cast<const f32x4*>(sPp + b0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < R) { rwpb[i] = *reinterpret_cast<const f32x4*>(sRWp + i * N + b0); dbb[i] = rm0[i] * *reinterpret_cast<const f32x4*>(sG + i * N + b0); }
            }
            for (int r0 = rg; r0 < NR; r0 += RG * MPB_PFL) {
                f32x4 gv[MPB_PFL], ltv[MPB_PFL], lpv[MPB_PFL];
#pragma unroll
                for (int u = 0; u < MPB_PFL; ++u) {
                    const int r = r0 + u * RG;
                    gv[u] = f32x4{0.f, 0.f, 0.f, 0.f}; ltv[u] = gv[u]; lpv[u] = gv[u];
                    if (r < NR && colok) {
                        // non-temporal LOADS (read once: they should not push the weights out of L2; measured free in the probe,
                        // non-temporal STORES cost 40 %: profiles/r03_link_stream_probe.txt)
                        gv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gLg + (size_t)r * N) + c4);
                        ltv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Ltg + (size_t)r * N) + c4);
                        lpv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Lpg + (size_t)r * N) + c4);
                    }
                }
#pragma unroll
                for (int u = 0; u < MPB_PFL; ++u) {
                    const int r = r0 + u * RG;
                    if (r < NR) {                                  // wave-uniform
                        const int ra = row0 + r;
                        const float wwa = sWW[ra];
                        float rowRW[4] = {0.f, 0.f, 0.f, 0.f}, rowWW = 0.f;
                        if (colok) {
                            f32x4 gq = gv[u];
                            const f32x4 lt = ltv[u], lp = lpv[u];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                if (i < R) {
                                    const float dfa = rm1[i] * sG[i * N + ra], rwpa = sRWp[i * N + ra];
                                    gq += dfa * rwpb[i] + rwpa * dbb[i];
                                    rowRW[i] = mpb_dot4(dbb[i], lt);
                                    colRW[i] += dfa * lt;
                                }
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (b0 + e == ra) gq[e] = 0.f;   // the diagonal of L_t is forced to 0
#pragma unroll
                            for (int e = 0; e < 4; ++e) rowWW += gq[e] * (ppb[e] - lp[e]);
                            colWW -= gq * lp;
                            colP += wwa * gq;
                            f32x4 gn;
#pragma unroll
                            for (int e = 0; e < 4; ++e) gn[e] = (1.0f - wwa - wwb[e]) * gq[e];
                            reinterpret_cast<f32x4*>(gLg + (size_t)r * N)[c4] = gn;
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (i < R) {
                                const float s = wave_sum(rowRW[i]);
                                if (lane == 0) sRP[(hh * (R + 1) + i) * NR + r] = s;
                            }
                        }
                        const float s = wave_sum(rowWW);
                        if (lane == 0) sRP[(hh * (R + 1) + R) * NR + r] = s;
                    }
                }
            }
            MP_STAMP(9);       // B5 link rows
            // column partials: fixed-order sums over the row groups.  round 1: the read-weight products
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < R) *reinterpret_cast<f32x4*>(sPart + (size_t)(rg * R + i) * NP + hh * 256 + lane * 4) = colRW[i];
            __syncthreads();
            for (int idx = tid; idx < R * NP; idx += CT) {
                const int i = idx / NP, c = idx - i * NP;
                if (c < N) {
                    float s = 0.f;
                    for (int q = 0; q < RG; ++q) s += sPart[(size_t)(q * R + i) * NP + c];
                    cl_store(slot2 + Q.oColRW + i * N + c, s, plain);
                }
            }
            for (int idx = tid; idx < (R + 1) * NR; idx += CT) {    // row sums: the two column halves
                float s = sRP[idx];
                if (NH == 2) s += sRP[(R + 1) * NR + idx];
                cl_store(slot2 + idx, s, plain);                    // [rowRW R x NR | rowWW NR] = offsets 0, oRowWW
            }
            __syncthreads();
            // round 2: d(ww) and d(precedence) column partials
            *reinterpret_cast<f32x4*>(sPart + (size_t)(rg * 2 + 0) * NP + hh * 256 + lane * 4) = colWW;
            *reinterpret_cast<f32x4*>(sPart + (size_t)(rg * 2 + 1) * NP + hh * 256 + lane * 4) = colP;
            __syncthreads();
            for (int idx = tid; idx < 2 * NP; idx += CT) {
                const int which = idx / NP, c = idx - which * NP;
                if (c < N) {
                    float s = 0.f;
                    for (int q = 0; q < RG; ++q) s += sPart[(size_t)(q * 2 + which) * NP + c];
                    cl_store(slot2 + (which ? Q.oColP : Q.oColWW) + c, s, plain);
                }
            }
            __syncthreads();
        }
        MP_STAMP(10);      // B5 column reductions
        MP_STAMP(11);      // B7
        cl_publish(fl2 + g, epoch, tid, plain);
        // M_{t-1} rows again, for B10b: requested here (L2 hits, behind the wait), not kept in registers through the link pass
#pragma unroll
        for (int q = 0; q < MPQ; ++q) {
            const int nl = grp + NG * q;
            Mp[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (q < NQ && nl < NR && gl < W4) Mp[q] = reinterpret_cast<const f32x4*>(Mpg + (size_t)nl * W)[gl];
        }
        if (!mp_wait(fl2, epoch, k, a.err, a.sticky, sAbort, tid)) return;
        MP_STAMP(12);      // publish 2 + wait 2
        {   // consume hand-off 2 (16-byte loads; the order of every sum is fixed: owner's row sum, then workgroups 0..k-1)
            const __amdgpu_buffer_rsrc_t rs = mp_rsrc(mb2 + (size_t)par * k * sl2, (size_t)k * sl2);
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            for (int i4 = tid; i4 < (RN >> 2); i4 += CT) {
                const int idx = 4 * i4;
                const int i = cl_div(idx, C.mg_N), n = idx - i * N;
                const int og = cl_div(n, C.mg_NR), rl = n - og * NR;
                f32x4 pv[8];
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) pv[gg] = (gg < k) ? mp_load4(rs, gg * sl2 + Q.oColRW + idx) : z4;
                f32x4 s = mp_load4(rs, og * sl2 + i * NR + rl);
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) if (gg < k) s += pv[gg];
                *reinterpret_cast<f32x4*>(sDRWp + idx) = s;
            }
            for (int n4 = tid; n4 < N4; n4 += CT) {
                const int n = 4 * n4;
                const int og = cl_div(n, C.mg_NR), rl = n - og * NR;
                f32x4 pw[8], pp[8];
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) {
                    pw[gg] = (gg < k) ? mp_load4(rs, gg * sl2 + Q.oColWW + n) : z4;
                    pp[gg] = (gg < k) ? mp_load4(rs, gg * sl2 + Q.oColP + n) : z4;
                }
                f32x4 sw = mp_load4(rs, og * sl2 + Q.oRowWW + rl), sp = z4;
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) if (gg < k) { sw += pw[gg]; sp += pp[gg]; }
                *reinterpret_cast<f32x4*>(sDWW + n) = sw + mp_load4(rs, og * sl2 + Q.oT1 + rl);
                *reinterpret_cast<f32x4*>(sDPp + n) = sp;
                *reinterpret_cast<f32x4*>(sSIMw + n) = mp_load4(rs, og * sl2 + Q.oSimw + rl);
            }
            // column sums of the memory passes -> interface gradients (d read keys, d erase, d write vector)
            if (tid < 4) {
                float s = 0.f;
                const float* base = mb2 + (size_t)par * k * sl2;
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) if (gg < k) s += cl_load(base + (size_t)gg * sl2 + Q.oNk + tid);
                sSC[40 + tid] = s;                                 // d|kr_i|
            }
            __syncthreads();
            for (int i4 = tid; i4 < ((RWd + 2 * W) >> 2); i4 += CT) {
                const int idx = 4 * i4;
                const int off = (idx < RWd) ? Q.oK + idx : ((idx < RWd + W) ? Q.oE + (idx - RWd) : Q.oV + (idx - RWd - W));
                f32x4 s = z4;
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) if (gg < k) s += mp_load4(rs, gg * sl2 + off);
                if (idx < RWd) {
                    const int i = cl_div(i4, C.mg_W4);
                    const f32x4 kv = *reinterpret_cast<const f32x4*>(sI + C.oKr + idx);
                    const float dn = sSC[40 + i], rn = sSC[i];
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = s[e] + dn * kv[e] / rn;
#pragma unroll
                    for (int e = 0; e < 4; ++e) sDX[C.oKr + idx + e] = o[e];
                } else if (idx < RWd + W) {
                    const f32x4 ev = *reinterpret_cast<const f32x4*>(sI + C.oE + (idx - RWd));
#pragma unroll
                    for (int e = 0; e < 4; ++e) sDX[C.oE + (idx - RWd) + e] = s[e] * ev[e] * (1.0f - ev[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) sDX[C.oV + (idx - RWd - W) + e] = s[e];
                }
            }
        }
        MP_STAMP(13);      // consume 2
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
        // ------------------------------------------------------------ B8: write-weight mix (access.py:252-257)
        {
            const float ga = sI[C.oAg], gw = sI[C.oWg];
            float* sT = sPart;                                     // rank-ordered usages
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
        MP_STAMP(14);      // B6, B8, B9
        // ------------------------------------------------------------ B10b: content part of d(M_{t-1}) (registers), d(write key) of the own rows
        {
            f32x4 accKw = {0.f, 0.f, 0.f, 0.f};
            float accNkw = 0.f;
            const float bw = sI[C.oBw], nk = sSC[R];
            f32x4 kp = {0.f, 0.f, 0.f, 0.f};
            if (gl < W4) { const float* kq = sI + C.oKw + gl * 4; kp = f32x4{kq[0], kq[1], kq[2], kq[3]}; }
#pragma unroll
            for (int q = 0; q < MPQ; ++q) {
                const int nl = grp + NG * q;
                if (q < NQ && nl < NR && gl < W4) {
                    const int n = row0 + nl;
                    const f32x4 mp = Mp[q];
                    const float nm = sNMw[nl];
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
            }
            MP_STAMP(21);      // B10b rows
            {   // hand the rows over to the next step: its M_t is this step's M_{t-1}; its M_{t-1} is requested now
                const float* Mqg = (t > 1) ? a.rec_M + ((bt - 2) * N + row0) * W : a.mem0 + ((size_t)b * N + row0) * W;
#pragma unroll
                for (int q = 0; q < MPQ; ++q) {
                    const int nl = grp + NG * q;
                    Mt[q] = Mp[q];
                    if (t > 0 && q < NQ && nl < NR && gl < W4) Mp[q] = reinterpret_cast<const f32x4*>(Mqg + (size_t)nl * W)[gl];
                }
            }
            accKw = mpb_fold4(accKw, LPR);
            accNkw = mpb_fold(accNkw, LPR);
            MP_STAMP(22);      // B10b fold
            __syncthreads();                                       // sT / sS (sPart) are dead
            if (lane < LPR && gl < W4) *reinterpret_cast<f32x4*>(sPart + wave * W + gl * 4) = accKw;
            if (lane == 0) sSC[20 + wave] = accNkw;
        }
        MP_STAMP(23);      // B10b park
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
        __syncthreads();
        MP_STAMP(15);      // B10b, B11
        {   // publish hand-off 3: partial d(write key) column sums + partial d|kw|
            for (int w = tid; w < W; w += CT) {
                float s = 0.f;
#pragma unroll
                for (int wv = 0; wv < CW; ++wv) s += sPart[wv * W + w];
                cl_store(slot3 + w, s, plain);
            }
            if (tid == 0) {
                float s = 0.f;
#pragma unroll
                for (int wv = 0; wv < CW; ++wv) s += sSC[20 + wv];
                cl_store(slot3 + W, s, plain);
            }
            cl_publish(fl3 + g, epoch, tid, plain);
        }
        if (wave < R) {                                                          // d(free gates)
            const int i = wave;
            float s = 0.f;
            for (int n = lane; n < N; n += 64) s += sDSIM[i * N + n];
            s = wave_sum(s);
            const float fg = sI[C.oF + i];
            if (lane == 0) sDX[C.oF + i] = s * fg * (1.0f - fg);
        }
        for (int i = tid; i < RN; i += CT) sgRW[i] = sDRWp[i];                  // carried d(read weights_{t-1})
        for (int n = tid; n < N; n += CT) sgP[n] = sDPp[n];                     // carried d(precedence_{t-1})
        __syncthreads();
        MP_STAMP(16);      // publish 3, free gates, carried vectors
        // ------------------------------------------------------------ B14: d(clipped h) of the own units += d(interface) . Wi^T
        //   Wi is [unit][IP]: a row is contiguous -- EIGHT LANES per own unit walk its row (128-byte segments; one unit per thread
        //   read 64 different rows per load instruction).  Every interface gradient except the write key's is final here, so
        //   that part of the product runs IN THE SHADOW of hand-off 3; only the write key's columns wait for it.
        float dh_acc[2] = {0.f, 0.f};                                           // own units tid / 8 and 64 + tid / 8
        const int l8 = tid & 7;
        {
            const f32x4* dx4 = reinterpret_cast<const f32x4*>(sDX);
            const int kw0 = C.oKw >> 2, kw1 = (C.oKw + W + 3) >> 2;            // float4 columns that hold a write-key entry
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                const int ju = (tid >> 3) + 64 * p2;
                if (ju < nU) {
                    const f32x4* wp = reinterpret_cast<const f32x4*>(a.Wi) + (size_t)(u0 + ju) * C.icg;
                    const int c1 = kw1 + ((l8 - kw1) & 7);             // first column >= kw1 of this lane's residue class
#pragma unroll 8
                    for (int c = l8; c < kw0; c += 8) dh_acc[p2] += mpb_dot4(dx4[c], wp[c]);
#pragma unroll 8
                    for (int c = c1; c < C.icg; c += 8) dh_acc[p2] += mpb_dot4(dx4[c], wp[c]);
                }
            }
        }
        if (!mp_wait(fl3, epoch, k, a.err, a.sticky, sAbort, tid)) return;
        {
            const float* base = mb3 + (size_t)par * k * sl3;
            float dn = 0.f;
#pragma unroll
            for (int gg = 0; gg < 8; ++gg) if (gg < k) dn += cl_load(base + (size_t)gg * sl3 + W);
            for (int w = tid; w < W; w += CT) {
                float s = 0.f;
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) if (gg < k) s += cl_load(base + (size_t)gg * sl3 + w);
                sDX[C.oKw + w] = s + dn * sI[C.oKw + w] / sSC[R];
            }
        }
        __syncthreads();
        if (g == 0) for (int c = tid; c < IP; c += CT) a.dxi[bt * IP + c] = sDX[c];
        MP_STAMP(17);      // wait 3 + consume + dxi out
        {
            const f32x4* dx4 = reinterpret_cast<const f32x4*>(sDX);
            const int kw0 = C.oKw >> 2, kw1 = (C.oKw + W + 3) >> 2;
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                const int ju = (tid >> 3) + 64 * p2;
                if (ju < nU) {
                    const f32x4* wp = reinterpret_cast<const f32x4*>(a.Wi) + (size_t)(u0 + ju) * C.icg;
                    for (int c = kw0 + l8; c < kw1; c += 8) dh_acc[p2] += mpb_dot4(dx4[c], wp[c]);
                }
                const float tot = group_sum<8>(dh_acc[p2]);
                if (ju < nU && l8 == 0) sPart[ju] = tot;
            }
        }
        __syncthreads();
        // ------------------------------------------------------------ B15: clip + snt.LSTM backward of the own units
        if (tid < nU) {
            const int u = u0 + tid;
            const float dh = sDHC[u] + sPart[tid];
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
        MP_STAMP(18);      // B14, B15
        // ------------------------------------------------------------ B16: partial d[reads_prev ; h_prev] over the own gate columns
        {
            const int kg4 = Q.kg4, nrow = 4 * nU;
            if (tid < Q.nslZ * kg4) {
                const int sl = cl_div(tid, Q.mg_kg4), cg = tid - sl * kg4;
                const int r0 = sl * Q.nperZ, r1 = min(nrow, r0 + Q.nperZ);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                if (r0 < r1) acc = ntk_stream_matvec<FIX ? 8 : 4>(reinterpret_cast<const f32x4*>(a.WrT) + (size_t)(4 * u0) * kg4 + cg, kg4, sDG, r0, r1, nrow - 1);
                *reinterpret_cast<f32x4*>(sPart + sl * Q.ldkT + cg * 4) = acc;
            }
            __syncthreads();
            for (int kk = tid; kk < Q.ldkT; kk += CT) {
                float s = 0.f;
                for (int sl = 0; sl < Q.nslZ; ++sl) s += sPart[sl * Q.ldkT + kk];
                cl_store(slot4 + kk, s, plain);
            }
            cl_publish(fl4 + g, epoch, tid, plain);
        }
        MP_STAMP(19);      // B16 + publish 4
        if (!mp_wait(fl4, epoch, k, a.err, a.sticky, sAbort, tid)) return;
        {
            const float* base = mb4 + (size_t)par * k * sl4;
            for (int kk = tid; kk < K; kk += CT) {
                float pv[8];
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) pv[gg] = (gg < k) ? cl_load(base + (size_t)gg * sl4 + kk) : 0.f;
                float s = 0.f;
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) if (gg < k) s += pv[gg];
                sGZ[kk] = s;
            }
        }
        MP_STAMP(20);      // wait 4 + consume
        __syncthreads();
    }
#ifdef NTK_CL_PROF
    (void)prof_last;
#endif

    // ---- carried gradients out (segmented BPTT); d(memory) scratch updated in place (d(link) already is)
    {
        const DncMpBwdArgs& a = a0;
        const DncMpCfg C = FIX ? kDncMpFixCfg : a.c;
        const DncMpBwdGeo Q = FIX ? kDncMpFixBwdGeo : a.q;
        const DncMpBwdLds L = FIX ? kDncMpFixBwdLds : a.lds;
        MP_BWD_VIEWS();
        float* cy = a.gcarry ? a.gcarry + (size_t)b * (2 * N + RN + Q.ldkT + hid) : nullptr;
        if (cy) {
            if (g == 0) {
                for (int i = tid0; i < N; i += CT) { cy[i] = sgP[i]; cy[N + i] = sgU[i]; }
                for (int i = tid0; i < RN; i += CT) cy[2 * N + i] = sgRW[i];
                for (int i = tid0; i < Q.ldkT; i += CT) cy[2 * N + RN + i] = sGZ[i];
            }
            for (int i = tid0; i < nU; i += CT) cy[2 * N + RN + Q.ldkT + u0 + i] = sgC[i];
        }
        const int gl0 = tid0 & (Q.LPR - 1), grp0 = tid0 / Q.LPR;
#pragma unroll
        for (int q = 0; q < MPQ; ++q) {
            const int nl = grp0 + Q.NG * q;
            if (q < Q.NQ && nl < NR && gl0 < W4) reinterpret_cast<f32x4*>(a.gM + ((size_t)b * N + row0 + nl) * W)[gl0] = gMr[q];
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
#ifdef NTK_CL_PROF
extern "C" int ntk_dnc_mp_bwd_prof(unsigned long long* out32) {
    return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_mpb_prof), 32 * sizeof(unsigned long long)) == hipSuccess ? NTK_OK : NTK_ERR_HIP;
}
#endif

static int dnc_mp_bwd_pick(int B, int N, int W, int R, int Wn, int hid, int O, int k_req, DncMpCfg& c, DncMpBwdGeo& q, size_t* lds_bytes) {
    if (Wn != 1 || R < 1 || R > 4 || N < 64 || (N % 64) != 0 || N > CT || W < 4 || (W % 4) != 0 || W > 256 || hid < 4 ||
        (hid % 4) != 0 || hid > 1024 || O < 1 || O > 16 || B < 1)
        return 0;
    const int cus = ntk_device_cu_count();
    // k_req 0: first the cluster sizes whose shape has a compile-time instantiation (several times faster), then any
    for (int pass = 0; pass < 2; ++pass)
    for (int k = 2; k <= 8; k <<= 1) {
        if (k_req > 0 && k != k_req) continue;
        if (pass == 0 && (k_req > 0 ? false : dnc_mp_shape_of(dnc_mp_cfg(N, W, R, hid, O, k)) == 0)) continue;
        if ((long)B * k > cus) continue;                       // one workgroup per CU, all co-resident
        const int NR = N / k;
        if (NR * k != N || NR < 8 || (NR % 8) != 0) continue;
        c = dnc_mp_cfg(N, W, R, hid, O, k);
        q = dnc_mp_bwd_geo(c);
        if (c.upk > 128 || q.NQ > MPQ || q.kg4 > CT || c.icg > 4 * CT) continue;      // B14: two passes of 64 own units
        const DncMpBwdLds L = dnc_mp_bwd_lds(c, q);
        const size_t bytes = (size_t)L.total * sizeof(float);
        if (bytes > 160 * 1024) continue;
        if (lds_bytes) *lds_bytes = bytes;
        return k;
    }
    return 0;
}

extern "C" int ntk_dnc_mp_bwd_plan(int B, int N, int W, int R, int Wn, int hid, int O, int k_request, int* k, size_t* workspace_bytes) {
    DncMpCfg c;
    DncMpBwdGeo q;
    const int kk = dnc_mp_bwd_pick(B, N, W, R, Wn, hid, O, k_request, c, q, nullptr);
    if (k) *k = kk;
    if (workspace_bytes) *workspace_bytes = 0;
    if (kk <= 0) {
        ntk_set_error("ntk_dnc_mp_bwd_plan: B=%d N=%d W=%d R=%d Wn=%d hid=%d is outside the memory-partitioned cluster BPTT kernel's range "
                      "(num_writes 1, memory_size a multiple of 64 up to 512, hidden %% 4 == 0, B * k <= the device's CUs)",
                      B, N, W, R, Wn, hid);
        return NTK_ERR_UNSUPPORTED;
    }
    if (workspace_bytes) *workspace_bytes = dnc_mp_workspace_bytes(B, kk, q.slot);
    return NTK_OK;
}

extern "C" int ntk_dnc_mp_bwd(int B, int S, int N, int W, int R, int Wn, int hid, int O, float clip_value, int k,
                              const float* WrT, int ldkT, const float* Wi, const float* Wy,
                              const float* mem0, const float* link0, const float* usage0, const float* rw0,
                              const float* ww0, const float* prec0, const float* hc0,
                              const float* rec_gates, const float* rec_c, const float* rec_ifc, const float* rec_u,
                              const float* rec_ww, const float* rec_rw, const float* rec_cw, const float* rec_cr,
                              const float* rec_al, const float* rec_p, const float* rec_fwd, const float* rec_bwd,
                              const float* rec_M, const float* rec_L, const float* rec_ypre,
                              const float* dout, float* gM, float* gL, float* dgates, float* dxi, float* dypre,
                              float* gcarry, int carry_in, void* workspace, void* stream) {
    DncMpBwdArgs a;
    size_t lds_bytes = 0;
    NTK_REQUIRE(B > 0 && S > 0 && k > 0, NTK_ERR_BAD_SHAPE, "ntk_dnc_mp_bwd: B=%d S=%d k=%d", B, S, k);
    const int kk = dnc_mp_bwd_pick(B, N, W, R, Wn, hid, O, k, a.c, a.q, &lds_bytes);
    NTK_REQUIRE(kk == k, NTK_ERR_UNSUPPORTED, "ntk_dnc_mp_bwd: k=%d is not a valid cluster size for B=%d N=%d W=%d R=%d Wn=%d hid=%d "
                "(ask ntk_dnc_mp_bwd_plan)", k, B, N, W, R, Wn, hid);
    NTK_REQUIRE(ldkT == a.q.ldkT, NTK_ERR_BAD_SHAPE, "ntk_dnc_mp_bwd: ldkT=%d (expected %d = K rounded up to 4)", ldkT, a.q.ldkT);
    a.lds = dnc_mp_bwd_lds(a.c, a.q);
    a.B = B; a.S = S; a.clip = clip_value; a.carry_in = carry_in;
    NTK_REQUIRE(WrT && Wi && Wy && mem0 && link0 && usage0 && rw0 && ww0 && prec0 && hc0 && rec_gates && rec_c && rec_ifc &&
                    rec_u && rec_ww && rec_rw && rec_cw && rec_cr && rec_al && rec_p && rec_fwd && rec_bwd && rec_M && rec_L &&
                    rec_ypre && dout && gM && gL && dgates && dxi && dypre && workspace,
                NTK_ERR_BAD_PTR, "ntk_dnc_mp_bwd: null pointer");
    NTK_REQUIRE(ntk_aligned16(WrT) && ntk_aligned16(Wi) && ntk_aligned16(rec_gates) && ntk_aligned16(rec_M) && ntk_aligned16(rec_L) &&
                    ntk_aligned16(gM) && ntk_aligned16(gL) && ntk_aligned16(dgates) && ntk_aligned16(mem0) && ntk_aligned16(link0) &&
                    ntk_aligned16(workspace),
                NTK_ERR_BAD_PTR, "ntk_dnc_mp_bwd: 16-byte alignment");
    a.xcd_local = (B % 8) == 0 ? 1 : 0;
    a.WrT = WrT; a.Wi = Wi; a.Wy = Wy;
    a.mem0 = mem0; a.link0 = link0; a.usage0 = usage0; a.rw0 = rw0; a.ww0 = ww0; a.prec0 = prec0; a.hc0 = hc0;
    a.rec_gates = rec_gates; a.rec_c = rec_c; a.rec_ifc = rec_ifc; a.rec_u = rec_u; a.rec_ww = rec_ww; a.rec_rw = rec_rw;
    a.rec_cw = rec_cw; a.rec_cr = rec_cr; a.rec_al = rec_al; a.rec_p = rec_p; a.rec_fwd = rec_fwd; a.rec_bwd = rec_bwd;
    a.rec_M = rec_M; a.rec_L = rec_L; a.rec_ypre = rec_ypre; a.dout = dout; a.gM = gM; a.gL = gL;
    a.dgates = dgates; a.dxi = dxi; a.dypre = dypre; a.gcarry = gcarry;
    const size_t ctrl = dnc_mp_ctrl_bytes(B, k);
    const size_t wsb = dnc_mp_workspace_bytes(B, k, a.q.slot);
    a.flags = reinterpret_cast<unsigned*>(workspace);
    a.err = a.flags + (size_t)B * MPX * k;
    a.xcc = a.err + 1;
    a.mbox = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + ctrl);
    a.sticky = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(workspace) + wsb - 256);
    {
        static NtkLdsAttrCache lds_cache;
        const void* const ks[] = {(const void*)dnc_mp_bwd_kernel<0>, (const void*)dnc_mp_bwd_kernel<1>, (const void*)dnc_mp_bwd_kernel<2>,
                                  (const void*)dnc_mp_bwd_kernel<3>};
        const int rc_lds = ntk_raise_lds_limit(lds_cache, ks, 4, "ntk_dnc_mp_bwd");
        if (rc_lds != NTK_OK) return rc_lds;
    }
    hipError_t e = hipMemsetAsync(workspace, 0, ctrl, (hipStream_t)stream);
    NTK_REQUIRE(e == hipSuccess, NTK_ERR_HIP, "ntk_dnc_mp_bwd: hipMemsetAsync: %s", hipGetErrorString(e));
#ifdef NTK_DNC_MP_GENERIC
    const int shape = 0;
#else
    const int shape = dnc_mp_shape_of(a.c);
#endif
    const dim3 grid(B * k);
    switch (shape) {
        case 1: dnc_mp_bwd_kernel<1><<<grid, CT, lds_bytes, (hipStream_t)stream>>>(a); break;
        case 2: dnc_mp_bwd_kernel<2><<<grid, CT, lds_bytes, (hipStream_t)stream>>>(a); break;
        case 3: dnc_mp_bwd_kernel<3><<<grid, CT, lds_bytes, (hipStream_t)stream>>>(a); break;
        default: dnc_mp_bwd_kernel<0><<<grid, CT, lds_bytes, (hipStream_t)stream>>>(a); break;
    }
    NTK_CHECK_LAUNCH("ntk_dnc_mp_bwd");
    return NTK_OK;
}
