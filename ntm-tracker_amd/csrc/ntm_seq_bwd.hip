// NTM sequence backward (full BPTT, no truncation -- what tf.gradients through
// the tf.while_loop of LoopNTMTracker computes, direct_offset_output.py:611-621).
// One persistent workgroup per sequence walks the steps in reverse with the
// carried gradients (dM, dw, dread, dh, dc) resident in LDS.  The forward pass
// recorded every tensor the step derivative needs (ntm_seq_fwd.hip), so nothing
// is recomputed except cheap elementwise terms; the next step's records are
// prefetched into registers while the current step computes.
//
// Outputs: per-step raw (pre-activation) gradients of the LSTM gates and of the
// unpack/output linear -- the weight gradients are then three k-major GEMMs over
// all B*S rows (ntk_gemm_tn_f32) -- and the gradient of the initial state.
//
// Gradient semantics: pow: d/dx = y*x^(y-1), d/dy = x^y*log(x)
// with log(x) -> 0 for x <= 0; l2_normalize differentiates through
// rsqrt(max(sum x^2, 1e-12)) (zero through the norm when clamped).
#include "ntm_common.h"
#include <stdlib.h>
#include <type_traits>

// Diagnostic build only (-DNTK_CL_PROF): s_memtime shares between consecutive workgroup barriers of a step (LDS accumulators)
#ifdef NTK_CL_PROF
__device__ unsigned long long g_ntm_bwd_prof[16];
#define NTMB_STAMP(i)                                                       \
    do {                                                                    \
        if (blockIdx.x == 0 && tid == 0) {                                  \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
            s_prof[i] += now_ - s_prof[15];                                 \
            s_prof[15] = now_;                                              \
        }                                                                   \
    } while (0)
extern "C" int ntk_ntm_bwd_prof(unsigned long long* out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_ntm_bwd_prof), 16 * sizeof(unsigned long long)) == hipSuccess ? NTK_OK : NTK_ERR_HIP;
}
#else
#define NTMB_STAMP(i) do { } while (0)
#endif

struct NtmBwdArgs {
    NtmDims d;
    const float* WrT;      // [4*hid][ldkT]  transposed recurrent weights (rows n' = unit*4+gate)
    const float* WaT;      // [PP][ldhT]     transposed unpack/output weights
    int ldkT, ldhT;
    const float* M0; const float* w0; const float* cs0;
    const float* st_gates; const float* st_c; const float* st_u;
    const float* st_wc; const float* st_wv; const float* st_w; const float* st_M;
    const float* dlogits;  // [B,S,O]
    const float* dM_fin; const float* dw_fin; const float* dread_fin; const float* dcs_fin;  // nullable
    float* dgates;         // [B,S,4*hid]
    float* du;             // [B,S,PP]
    float* dM0; float* dw0; float* dread0; float* dcs0;
};

struct NtmBwdLds {
    int part, dM, G, Mp, Mt, dW, Wp, Wt, Wc, Wv, Wg, Dwv, Dsim, U, DU, DG, dZ, dC, Gt, Ct, Cp,
        Khat, Ks, Kinv, Kss, Cinv, Css, C2, Dkhat, Sw, Red, Dmh, total;
};

constexpr int NQ = 1 + NTM_MAX_SHIFT_TAPS;   // max simultaneous per-head reductions in one stage (d gamma + one per shift tap)
constexpr int QR1 = 0, QR2 = 2, QR3 = QR2 + NQ, QR4 = QR3 + 2;
constexpr int NQT = QR4 + 1; // reduction slots per head; every stage owns its own slots (no read/write reuse inside a step)
constexpr int MAXM = 8;      // max memory elements prefetched per thread
constexpr int NTMB_RES_WA = 7;   // benchmark shape: rows of Wa^T per thread (of its 15) kept in the LDS the state leaves free

static void ntm_bwd_lds(const NtmDims& d, int T, int ldkT, int ldhT, NtmBwdLds& L) {
    const int MP = d.Md | 1, NM = d.N * MP, HN = d.H * d.N;
    const int nout = d.H * d.Md + 2 * d.Wh * d.Md;
    const int nslP = ntm_imin(ntm_imax(1, T / nout), d.N);
    const int nslZ = ntm_imax(1, T / (ldkT / 4));
    const int nslH = ntm_imax(1, T / (ldhT / 4));
    const int nslC = ntm_imax(1, T / d.Md);
    int part = ntm_imax(nslP * nout, nslZ * ldkT);
    part = ntm_imax(part, nslH * ldhT);
    part = ntm_imax(part, nslC * d.Md);
    part = ntm_imax(part, 32 * d.R * d.Md);                  // the wave-specialised form's read columns of B11: 32 row slices
    int o = 0;
    auto take = [&](int n) { int r = o; o += ntm_align4(n); return r; };
    L.part = take(part);
    L.dM = take(NM); L.G = take(NM); L.Mp = take(NM); L.Mt = take(d.write_first ? NM : 4);
    L.dW = take(HN); L.Wp = take(HN); L.Wt = take(HN); L.Wc = take(HN); L.Wv = take(HN); L.Wg = take(HN);
    L.Dwv = take(HN); L.Dsim = take(HN);
    L.U = take(d.PP); L.DU = take(d.PP); L.DG = take(4 * d.hid); L.dZ = take(ldkT); L.dC = take(d.hid);
    L.Gt = take(4 * d.hid); L.Ct = take(d.hid); L.Cp = take(d.hid);
    L.Khat = take(d.H * d.Md); L.Ks = take(d.H * d.Md); L.Kinv = take(d.H); L.Kss = take(d.H);
    L.Cinv = take(d.Md); L.Css = take(d.Md); L.C2 = take(d.Md); L.Dkhat = take(d.H * d.Md);
    L.Sw = take(d.H * d.SS);
    L.Red = take(d.H * NQT * (d.N / 64));
    L.Dmh = take(d.N * (d.Md | 1));
    L.total = o;
}

// WS (benchmark shape only, 768 threads): the h columns of B11 -- d h_{t-1} = dgates_t . Wr^T[:, 80:280], 71 % of the kernel's
// largest weight stream (42 % of a BPTT step) -- are NOT needed until the LSTM cell backward of the NEXT iteration, ten phases
// later; only the 80 read columns feed the top of the next iteration.  Waves 10, 11 ("stream", 100 of 128 lanes = one float4
// column group x one half of the 800 rows) walk those columns in a circle, one lap per step, through a ring of row registers
// that is never drained, beside the phases of the 640 compute threads, and hand the product over (2 x 200 floats in LDS)
// before the barrier in front of B10.  Both kinds of wave run the same twelve workgroup barriers per step; the stream waves
// consume a fixed number of 4-row groups between consecutive barriers (kBwdWsGroups).  Same idea as ntm_seq_fwd_ws.hip.
constexpr int BWS_RING = 20, BWS_LAP = 400, BWS_NG = BWS_LAP / 4;
// groups consumed before barriers 1 .. 9 of an iteration (X1, X2, R2, R3, R4, the three parts of B7 / B8, B9), then after
// barriers 10 and 11 (beside the read columns of B11 and the carry / commit): 100 in all
constexpr int kBwdWsGroups[11] = {14, 16, 7, 4, 10, 14, 7, 9, 6, 6, 7};
constexpr int bws_sum(int n) { int s = 0; for (int i = 0; i < n; ++i) s += kBwdWsGroups[i]; return s; }
static_assert(bws_sum(11) == BWS_NG && BWS_LAP % BWS_RING == 0, "one lap per step; the ring's phase is static");
template <int N> using bwsic = std::integral_constant<int, N>;
template <int I0, int I1, class F>
__device__ __forceinline__ void bws_for(F&& f) {
    if constexpr (I0 < I1) { f(bwsic<I0>{}); bws_for<I0 + 1, I1>(f); }
}

template <int MAXT, bool FIX, bool WS = false>
__global__ __launch_bounds__(MAXT) void ntm_seq_bwd_kernel(NtmBwdArgs a, NtmBwdLds L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    static_assert(!WS || FIX, "the wave-specialised form exists for the benchmark shape only");
    const int b = blockIdx.x, tid0 = threadIdx.x, T = FIX ? 640 : blockDim.x;
    const int N = FIX ? 128 : a.d.N, Md = FIX ? 20 : a.d.Md, MP = Md | 1, R = FIX ? 4 : a.d.R, Wh = FIX ? 1 : a.d.Wh;
    const int H = R + Wh, hid = FIX ? 200 : a.d.hid, SS = FIX ? 3 : a.d.SS;
    const int S = a.d.S, RM = R * Md, K = RM + hid, NW = N >> 6, NMd = N * Md, HN = H * N;
    struct {
        int O, oK, oB, oG, oS, oY, oE, oA, P, PP;
    } d;
    d.O = FIX ? 2 : a.d.O;
    d.oK = 0; d.oB = H * Md; d.oG = d.oB + H; d.oS = d.oG + H; d.oY = d.oS + H * SS; d.oE = d.oY + H;
    d.oA = d.oE + Wh * Md; d.P = d.oA + Wh * Md;
    d.PP = (d.P + d.O + 3) & ~3;
    const int PP = d.PP;
    const bool wf = FIX ? false : (a.d.write_first != 0);
    const int ldkT = FIX ? 280 : a.ldkT, ldhT = FIX ? 200 : a.ldhT;
    int tid = tid0, lane = tid0 & 63;

    float* sPart = smem + L.part;
    float* sdM = smem + L.dM;  float* sG = smem + L.G;  float* sMp = smem + L.Mp;  float* sMt = smem + L.Mt;
    float* sdW = smem + L.dW;  float* sWp = smem + L.Wp; float* sWt = smem + L.Wt; float* sWc = smem + L.Wc;
    float* sWv = smem + L.Wv;  float* sWg = smem + L.Wg; float* sDwv = smem + L.Dwv; float* sDsim = smem + L.Dsim;
    float* sU = smem + L.U;    float* sDU = smem + L.DU; float* sDG = smem + L.DG; float* sdZ = smem + L.dZ;
    float* sdC = smem + L.dC;  float* sGt = smem + L.Gt; float* sCt = smem + L.Ct; float* sCp = smem + L.Cp;
    float* sKhat = smem + L.Khat; float* sKs = smem + L.Ks; float* sKinv = smem + L.Kinv; float* sKss = smem + L.Kss;
    float* sCinv = smem + L.Cinv; float* sCss = smem + L.Css; float* sC2 = smem + L.C2; float* sDkhat = smem + L.Dkhat;
    float* sSw = smem + L.Sw;  float* sRed = smem + L.Red;  float* sDmh = smem + L.Dmh;
    f32x4* sPart4 = reinterpret_cast<f32x4*>(sPart);
    // WS: d h_{t-1} partials of the stream waves, [2 row halves][200] floats, behind the resident rows of Wa^T
    float* sPartH = smem + L.total + 32 + NTMB_RES_WA * 600 * 4;

    if constexpr (WS) {
        if (tid0 >= 640) {
            // =========================================================== stream waves (see the note above the kernel)
            const int sidx = min(tid0 - 640, 99);
            const int cgs = sidx % 50, sls = sidx / 50;              // float4 column group of the h columns, half of the rows
            const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.WrT + 80), 0,
                                                                                  (800 * 280 - 80) * (int)sizeof(float), 0x00020000);
            const unsigned voff = (unsigned)cgs * 16u + (unsigned)sls * (unsigned)(BWS_LAP * 280 * sizeof(float));
            auto wrow = [&](int r) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, voff, r * (280 * (int)sizeof(float)), 0)); };
            const float* dgp = sDG + BWS_LAP * sls;                  // this half's dgates (written by B10, read after barrier 10)
            constexpr int G0 = kBwdWsGroups[9] + kBwdWsGroups[10];   // groups of a lap consumed in the iteration that starts it
            f32x4 ring[BWS_RING];
#pragma unroll
            for (int q = 0; q < BWS_RING; ++q) ring[(4 * G0 + q) % BWS_RING] = wrow((4 * G0 + q) % BWS_LAP);
            // the first iteration's "product of the step after the last" is the gradient of the final state: dgates are zero
            // there (the compute waves clear them), so the groups it still runs add nothing to this initial value
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (sls == 0 && a.dcs_fin) acc = *reinterpret_cast<const f32x4*>(a.dcs_fin + (size_t)b * 2 * hid + hid + 4 * cgs);
            auto group = [&](auto gc) {                              // rows 4 g .. 4 g + 3 of this lane's half
                constexpr int r = 4 * decltype(gc)::value;
                const f32x4 hv = *reinterpret_cast<const f32x4*>(dgp + r);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc += hv[q] * ring[(r + q) % BWS_RING];
                    ring[(r + q) % BWS_RING] = wrow((r + q + BWS_RING) % BWS_LAP);
                }
                __builtin_amdgcn_sched_barrier(0);                   // keep the groups in program order (ntm_seq_fwd_ws.hip)
            };
            auto finish_lap = [&]() {                                // groups G0 .. 99 with the barriers 1 .. 8 between them, then the hand-over
                constexpr int c0 = G0, c1 = c0 + kBwdWsGroups[0], c2 = c1 + kBwdWsGroups[1], c3 = c2 + kBwdWsGroups[2],
                              c4 = c3 + kBwdWsGroups[3], c5 = c4 + kBwdWsGroups[4], c6 = c5 + kBwdWsGroups[5],
                              c7 = c6 + kBwdWsGroups[6], c8 = c7 + kBwdWsGroups[7];
                static_assert(c8 + kBwdWsGroups[8] == BWS_NG, "lap");
                bws_for<c0, c1>(group); __syncthreads();             // 1
                bws_for<c1, c2>(group); __syncthreads();             // 2
                bws_for<c2, c3>(group); __syncthreads();             // 3
                bws_for<c3, c4>(group); __syncthreads();             // 4
                bws_for<c4, c5>(group); __syncthreads();             // 5
                bws_for<c5, c6>(group); __syncthreads();             // 6
                bws_for<c6, c7>(group); __syncthreads();             // 7
                bws_for<c7, c8>(group); __syncthreads();             // 8
                bws_for<c8, BWS_NG>(group);
                // every lane stores (lanes 100 .. 127 shadow lane 99: same address, same value; see ntm_seq_fwd_ws.hip)
                reinterpret_cast<f32x4*>(sPartH)[sls * 50 + cgs] = acc;
                acc = f32x4{0.f, 0.f, 0.f, 0.f};
            };
            __syncthreads();                                         // initial state, records of the last step committed
            __syncthreads();                                         // resident rows of Wa^T
            for (int t = S - 1; t >= 0; --t) {
                finish_lap();
                __syncthreads();                                     // 9: the partials of d h_t are in LDS
                __syncthreads();                                     // 10: dgates_t are in LDS
                bws_for<0, kBwdWsGroups[9]>(group);
                __syncthreads();                                     // 11
                bws_for<kBwdWsGroups[9], G0>(group);
                __syncthreads();                                     // 12
            }
            finish_lap();                                            // d h_{-1}: the gradient of the initial controller state
            __syncthreads();                                         // (its barriers 1 .. 8 are matched by the compute waves' epilogue)
            return;
        }
    }

    // thread roles
    int hh = tid / N, nn = tid - hh * N;                // (head, slot) owner; active iff hh < H
    bool hn = hh < H;
    int wi = nn >> 6;
    const int nout = H * Md + 2 * Wh * Md;
    const int nslP = min(max(1, T / nout), N);
    const int nperP = (N + nslP - 1) / nslP;
    const int kg4 = ldkT >> 2, hg4 = ldhT >> 2;
    const int nslZ = max(1, T / kg4), nperZ = (4 * hid + nslZ - 1) / nslZ;
    const int nslH = max(1, T / hg4), nperH = (PP + nslH - 1) / nslH;
    const int nslC = max(1, T / Md), nperC = (N + nslC - 1) / nslC;

    // ---- prefetch registers for one step's records
    float pM[MAXM], pMt[MAXM], pWp = 0.f, pWt = 0.f, pWc = 0.f, pWv = 0.f, pU = 0.f, pCt = 0.f, pCp = 0.f, pDl = 0.f;
    f32x4 pG = {0.f, 0.f, 0.f, 0.f};
    auto prefetch = [&](int t) {
        const size_t bt = (size_t)b * S + t;
        const float* Mp = (t > 0) ? a.st_M + (bt - 1) * NMd : a.M0 + (size_t)b * NMd;
#pragma unroll
        for (int q = 0; q < MAXM; ++q) {
            const int idx = tid + q * T;
            pM[q] = (idx < NMd) ? Mp[idx] : 0.f;
            pMt[q] = (wf && idx < NMd) ? a.st_M[bt * NMd + idx] : 0.f;
        }
        if (hn) {
            pWp = (t > 0) ? a.st_w[(bt - 1) * HN + tid] : a.w0[(size_t)b * HN + tid];
            pWt = a.st_w[bt * HN + tid];
            pWc = a.st_wc[bt * HN + tid];
            pWv = a.st_wv[bt * HN + tid];
        }
        if (tid < PP) {
            pU = a.st_u[bt * PP + tid];
            pDl = (tid >= d.P && tid < d.P + d.O) ? a.dlogits[bt * d.O + (tid - d.P)] : 0.f;
        }
        if (tid < hid) {
            pG = reinterpret_cast<const f32x4*>(a.st_gates)[bt * hid + tid];
            pCt = a.st_c[bt * hid + tid];
            pCp = (t > 0) ? a.st_c[(bt - 1) * hid + tid] : a.cs0[(size_t)b * 2 * hid + tid];
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int q = 0; q < MAXM; ++q) {
            const int idx = tid + q * T;
            if (idx < NMd) {
                const int n = idx / Md, m = idx - n * Md;
                sMp[n * MP + m] = pM[q];
                if (wf) sMt[n * MP + m] = pMt[q];
            }
        }
        if (hn) { sWp[tid] = pWp; sWt[tid] = pWt; sWc[tid] = pWc; sWv[tid] = pWv; }
        if (tid < PP) { sU[tid] = pU; sDU[tid] = pDl; }
        if (tid < hid) { reinterpret_cast<f32x4*>(sGt)[tid] = pG; sCt[tid] = pCt; sCp[tid] = pCp; }
    };

    // ---- carried gradients start from the (optional) gradient of the final state
    for (int i = tid; i < NMd; i += T)
        sdM[(i / Md) * MP + (i % Md)] = a.dM_fin ? a.dM_fin[(size_t)b * NMd + i] : 0.f;
    for (int i = tid; i < HN; i += T) sdW[i] = a.dw_fin ? a.dw_fin[(size_t)b * HN + i] : 0.f;
    for (int i = tid; i < ldkT; i += T) {
        float v = 0.f;
        if (i < RM) v = a.dread_fin ? a.dread_fin[(size_t)b * RM + i] : 0.f;
        else if (i < K) v = a.dcs_fin ? a.dcs_fin[(size_t)b * 2 * hid + hid + (i - RM)] : 0.f;
        sdZ[i] = v;
    }
    for (int i = tid; i < hid; i += T) sdC[i] = a.dcs_fin ? a.dcs_fin[(size_t)b * 2 * hid + i] : 0.f;
    if constexpr (WS) {
        for (int i = tid; i < 4 * hid; i += T) sDG[i] = 0.f;          // the stream waves' first (partial) lap multiplies these
    }
    prefetch(S - 1);
    commit();
    __syncthreads();

    // per-head block reduction of nq values held by the (h, n) owner threads
    auto red_write = [&](const float (&v)[NQ], int nq, int base) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q < nq) {
                const float s = wave_sum(hn ? v[q] : 0.f);
                if (hn && lane == 0) sRed[(hh * NQT + base + q) * NW + wi] = s;
            }
        }
    };
    auto red_read = [&](int h, int q) -> float {
        float s = 0.f;
        for (int w = 0; w < NW; ++w) s += sRed[(h * NQT + q) * NW + w];
        return s;
    };

    // benchmark shape: 7 of the 15 rows of Wa^T a thread multiplies in B9 stay in LDS for the whole sequence (67 KB of the 74 KB
    // the state leaves free); B9's stream runs at ~67 GB/s, the worst of the kernel's three weight streams
    const f32x4* sWaRes4 = reinterpret_cast<const f32x4*>(smem + L.total + 32);
    if constexpr (FIX) {
        if (tid0 < nslH * hg4) {
            const int cg = tid0 % hg4, sl = tid0 / hg4;
            const int c0 = sl * nperH, c1 = min(PP, c0 + nperH);
            f32x4* wl = reinterpret_cast<f32x4*>(smem + L.total + 32);
            for (int q = 0; q < NTMB_RES_WA; ++q)
                wl[q * (nslH * hg4) + tid0] = (c0 + q < c1) ? reinterpret_cast<const f32x4*>(a.WaT)[(size_t)(c0 + q) * hg4 + cg] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
    }
    // WS: the first NTMB_RES_B11 rows of every compute thread's 25-row slice of B11's read columns stay in registers for the whole
    // launch (the kernel has 146 of the 168 registers twelve waves allow): 51 of the 256 KB that product draws from L2 per step
    constexpr int NTMB_RES_B11 = 5;
    f32x4 wres11[NTMB_RES_B11];
    if constexpr (WS) {
        const int cg = tid0 % 20, sl = tid0 / 20;
#pragma unroll
        for (int q = 0; q < NTMB_RES_B11; ++q) wres11[q] = reinterpret_cast<const f32x4*>(a.WrT)[(size_t)(sl * 25 + q) * kg4 + cg];
    }
#ifdef NTK_CL_PROF
    unsigned long long* s_prof = reinterpret_cast<unsigned long long*>(smem + L.total);      // 128 B behind the state (the launch adds them)
    if (threadIdx.x == 0) { for (int i = 0; i < 15; ++i) s_prof[i] = 0; s_prof[15] = __builtin_amdgcn_s_memtime(); }
#endif
    for (int t = S - 1; t >= 0; --t) {
        // opaque thread id: keeps loop-invariant index/address expressions from being hoisted out of the
        // t-loop (they would be spilled to scratch and reloaded every step)
        {
            int tid_op = tid0;
            asm volatile("" : "+v"(tid_op));
            tid = tid_op; lane = tid & 63;
            hh = tid / N; nn = tid - hh * N; hn = hh < H; wi = nn >> 6;
        }
        const size_t bt = (size_t)b * S + t;
        if (t > 0) prefetch(t - 1);

        // ------------------------------------------------ X1: memory-shaped elementwise + column norms + small vectors
        if constexpr (FIX) {
            // benchmark shape (write_first off): a thread = (slot n, four adjacent columns): the five head weights of the slot
            // are read once for four elements and d(read) comes in 16-byte reads: 29 LDS operations per thread instead of 52
            const int n = tid / 5, m0 = (tid - n * 5) * 4;
            float wt[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) wt[i] = sWt[i * N + n];
            f32x4 dmr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) dmr += wt[i] * *reinterpret_cast<const f32x4*>(sdZ + i * Md + m0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ai = n * MP + m0 + e;
                const float dMt = sdM[ai];
                const float E = 1.0f - wt[4] * sU[d.oE + m0 + e];
                sG[ai] = dMt;
                sdM[ai] = dMt * E + dmr[e];
            }
        } else
        for (int idx = tid; idx < NMd; idx += T) {
            const int n = idx / Md, m = idx - n * Md, ai = n * MP + m;
            float dMt = sdM[ai];
            float dMr = 0.f;
            for (int i = 0; i < R; ++i) dMr += sWt[i * N + n] * sdZ[i * Md + m];
            if (wf) dMt += dMr;
            float E = 1.f;
            for (int j = 0; j < Wh; ++j) E *= (1.0f - sWt[(R + j) * N + n] * sU[d.oE + j * Md + m]);
            sG[ai] = dMt;
            sdM[ai] = dMt * E + (wf ? 0.f : dMr);
        }
        if (tid < nslC * Md) {     // column sum of squares of M_prev (quirk Q1 normaliser)
            const int m = tid % Md, sl = tid / Md;
            const int n0 = sl * nperC, n1 = min(N, n0 + nperC);
            float s = 0.f;
            for (int n = n0; n < n1; ++n) { const float v = sMp[n * MP + m]; s += v * v; }
            sPart[sl * Md + m] = s;
        }
        if (tid < H) {             // key norms and shift softmax
            const int h = tid;
            float ss = 0.f;
            for (int m = 0; m < Md; ++m) { const float kv = sU[d.oK + h * Md + m]; ss += kv * kv; }
            sKss[h] = ss;
            sKinv[h] = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
            float mx = -INFINITY;
            for (int j = 0; j < SS; ++j) mx = fmaxf(mx, sU[d.oS + h * SS + j]);
            float sum = 0.f;
            for (int j = 0; j < SS; ++j) sum += expf(sU[d.oS + h * SS + j] - mx);
            for (int j = 0; j < SS; ++j) sSw[h * SS + j] = expf(sU[d.oS + h * SS + j] - mx) / sum;
        }
        __syncthreads();
        NTMB_STAMP(0);

        // ------------------------------------------------ X2: d(w_t) for every head; R1 sums
        float dwt = 0.f, pw = 0.f, wv = 0.f, wt = 0.f, wc = 0.f, wp = 0.f, gam = 1.f, gate = 0.f;
        float rv[NQ];
        if (tid < Md) {
            float s = 0.f;
            for (int sl = 0; sl < nslC; ++sl) s += sPart[sl * Md + tid];
            sCss[tid] = s;
            sCinv[tid] = 1.0f / sqrtf(fmaxf(s, 1e-12f));
        }
        if (hn) {
            const int h = hh, n = nn;
            float acc = sdW[tid];
            if (h < R) {
                const float* Mr = wf ? sMt : sMp;
                for (int m = 0; m < Md; ++m) acc += sdZ[h * Md + m] * Mr[n * MP + m];
            } else {
                const int j = h - R;
                for (int m = 0; m < Md; ++m) {
                    float oth = 1.f;
                    for (int j2 = 0; j2 < Wh; ++j2)
                        if (j2 != j) oth *= (1.0f - sWt[(R + j2) * N + n] * sU[d.oE + j2 * Md + m]);
                    const float g = sG[n * MP + m];
                    const float Tj = g * sMp[n * MP + m] * oth;
                    acc += -sU[d.oE + j * Md + m] * Tj + sU[d.oA + j * Md + m] * g;
                }
            }
            dwt = acc;
            wv = sWv[tid]; wt = sWt[tid]; wc = sWc[tid]; wp = sWp[tid];
            gam = sU[d.oY + h]; gate = sU[d.oG + h];
            pw = powf(wv, gam);
            sWg[tid] = gate * wc + (1.0f - gate) * wp;
            rv[0] = pw; rv[1] = dwt * wt;
        }
        red_write(rv, 2, QR1);
        __syncthreads();
        NTMB_STAMP(1);

        // ------------------------------------------------ R2: sharpen backward, shift-weight sums
        float dpw = 0.f, dwv = 0.f;
        if (tid < H * Md) {        // normalised keys (needed from R4 on)
            const int h = tid / Md, m = tid - h * Md;
            const float kh = sU[d.oK + tid] * sKinv[h];
            sKhat[tid] = kh;
            sKs[tid] = kh * sCinv[m];
        }
        if (hn) {
            const float den = red_read(hh, QR1) + 1e-3f;
            const float s2 = red_read(hh, QR1 + 1);
            dpw = (dwt - s2) / den;
            dwv = (wv > 0.f) ? dpw * gam * pw / wv : 0.f;
            sDwv[tid] = dwv;
            rv[0] = (wv > 0.f) ? dpw * pw * logf(wv) : 0.f;        // d gamma
            const int start = -((SS + 1) >> 1);
#pragma unroll
            for (int j = 0; j < NQ - 1; ++j) {
                if (j < SS) {
                    int src = nn + start + j; src = (src % N + N) % N;
                    rv[1 + j] = dwv * sWg[hh * N + src];            // d shift_j
                }
            }
        }
        red_write(rv, 1 + SS, QR2);
        __syncthreads();
        NTMB_STAMP(2);

        // ------------------------------------------------ R3: shift + gate backward
        float dwg = 0.f, dwc = 0.f;
        float Sgam = 0.f, Ssw[NQ - 1];
        if (hn) {
            Sgam = red_read(hh, QR2);
#pragma unroll
            for (int j = 0; j < NQ - 1; ++j) Ssw[j] = (j < SS) ? red_read(hh, QR2 + 1 + j) : 0.f;
            const int start = -((SS + 1) >> 1);
            for (int j = 0; j < SS; ++j) {
                int src = nn - (start + j); src = (src % N + N) % N;
                dwg += sSw[hh * SS + j] * sDwv[hh * N + src];
            }
            sdW[tid] = (1.0f - gate) * dwg;                         // carried d(w_{t-1})
            dwc = gate * dwg;
            rv[0] = dwg * (wc - wp);                                // d g
            rv[1] = wc * dwc;                                       // softmax backward inner product
        }
        red_write(rv, 2, QR3);
        __syncthreads();
        NTMB_STAMP(3);

        // ------------------------------------------------ R4: content softmax backward
        float Sg = 0.f, dv = 0.f;
        if (hn) {
            Sg = red_read(hh, QR3);
            const float Bs = red_read(hh, QR3 + 1);
            dv = wc * (dwc - Bs);
            float sim = 0.f;
            for (int m = 0; m < Md; ++m) sim += sKs[hh * Md + m] * sMp[nn * MP + m];
            rv[0] = dv * sim;                                       // d beta
            sDsim[tid] = dv * sU[d.oB + hh];
        }
        red_write(rv, 1, QR4);
        __syncthreads();
        NTMB_STAMP(4);
        if (hn && nn == 0) {       // per-head scalar controls -> raw gradients
            const int h = hh;
            const float beta = sU[d.oB + h];
            sDU[d.oB + h] = red_read(h, QR4) * (1.0f - expf(-beta));                 // softplus' = 1 - exp(-softplus)
            sDU[d.oG + h] = Sg * gate * (1.0f - gate);
            sDU[d.oY + h] = Sgam * (1.0f - expf(-(gam - 1.0f)));
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < NQ - 1; ++j) if (j < SS) dot += sSw[h * SS + j] * Ssw[j];
#pragma unroll
            for (int j = 0; j < NQ - 1; ++j) if (j < SS) sDU[d.oS + h * SS + j] = sSw[h * SS + j] * (Ssw[j] - dot);
        }

        NTMB_STAMP(12);
        // ------------------------------------------------ B7: dMhat[n][m] = sum_h dsim[h][n] khat[h][m], computed ONCE
        //                                                  (the column-norm sum and the d(M_prev) update both use it);
        //                                                  reductions over slots (keys, erase, add)
        for (int idx = tid; idx < NMd; idx += T) {
            const int n = idx / Md, m = idx - n * Md;
            float dmh = 0.f;
            for (int h = 0; h < H; ++h) dmh += sDsim[h * N + n] * sKhat[h * Md + m];
            sDmh[n * MP + m] = dmh;
        }
        NTMB_STAMP(13);
        if constexpr (FIX) {
            // benchmark shape (5 heads, 1 write head): a thread = (memory column m = tid >> 4, row class sl = tid & 15) reads
            // M_prev[n][m], G[n][m], ww[n] and dsim[0..4][n] ONCE per row n = sl, sl + 16, ... and feeds seven sums; the sixteen
            // row classes of a column are sixteen adjacent lanes, reduced on the DPP path: 20 K LDS reads per step instead of 41 K,
            // no slot partials (the one-output-per-thread form below was 14 % of a BPTT step)
            if (tid < 16 * Md) {
                const int m = tid >> 4, sl = tid & 15;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, ae = 0.f, aa = 0.f;
#pragma unroll 4
                for (int n = sl; n < N; n += 16) {
                    const float mp = sMp[n * MP + m], g = sG[n * MP + m], ww = sWt[R * N + n];
                    a0 += sDsim[0 * N + n] * mp; a1 += sDsim[1 * N + n] * mp; a2 += sDsim[2 * N + n] * mp;
                    a3 += sDsim[3 * N + n] * mp; a4 += sDsim[4 * N + n] * mp;
                    const float wg = ww * g;
                    ae -= wg * mp;
                    aa += wg;
                }
                auto r16 = [](float v) { v += ntk_dpp<0xB1>(v); v += ntk_dpp<0x4E>(v); v += ntk_dpp<0x141>(v); v += ntk_dpp<0x140>(v); return v; };
                a0 = r16(a0); a1 = r16(a1); a2 = r16(a2); a3 = r16(a3); a4 = r16(a4); ae = r16(ae); aa = r16(aa);
                if (sl == 0) {
                    sPart[0 * Md + m] = a0; sPart[1 * Md + m] = a1; sPart[2 * Md + m] = a2; sPart[3 * Md + m] = a3; sPart[4 * Md + m] = a4;
                    sPart[H * Md + m] = ae; sPart[H * Md + Wh * Md + m] = aa;
                }
            }
        } else if (tid < nslP * nout) {
            const int o = tid % nout, sl = tid / nout;
            const int n0 = sl * nperP, n1 = min(N, n0 + nperP);
            float s = 0.f;
            if (o < H * Md) {                                  // sum_n dsim[h][n] * M_prev[n][m]
                const int h = o / Md, m = o - h * Md;
                // four independent chains, four rows per trip: the rolled single-chain loop paid an LDS round trip + the add
                // latency per row (this phase was 18 % of a BPTT step)
                float s1 = 0.f, s2 = 0.f, s3 = 0.f;
                int n = n0;
                for (; n + 3 < n1; n += 4) {
                    s += sDsim[h * N + n] * sMp[n * MP + m];
                    s1 += sDsim[h * N + n + 1] * sMp[(n + 1) * MP + m];
                    s2 += sDsim[h * N + n + 2] * sMp[(n + 2) * MP + m];
                    s3 += sDsim[h * N + n + 3] * sMp[(n + 3) * MP + m];
                }
                for (; n < n1; ++n) s += sDsim[h * N + n] * sMp[n * MP + m];
                s = (s + s1) + (s2 + s3);
            } else {
                const int o2 = o - H * Md;
                const int which = o2 / (Wh * Md);              // 0: erase, 1: add
                const int jm = o2 - which * Wh * Md;
                const int j = jm / Md, m = jm - j * Md;
                float sa = 0.f, sb = 0.f;                       // two chains (even / odd rows)
                for (int n = n0; n < n1; ++n) {
                    const float ww = sWt[(R + j) * N + n];
                    const float g = sG[n * MP + m];
                    float term;
                    if (which == 0) {
                        float oth = 1.f;
                        for (int j2 = 0; j2 < Wh; ++j2)
                            if (j2 != j) oth *= (1.0f - sWt[(R + j2) * N + n] * sU[d.oE + j2 * Md + m]);
                        term = -ww * g * sMp[n * MP + m] * oth;
                    } else {
                        term = ww * g;
                    }
                    if ((n - n0) & 1) sb += term; else sa += term;
                }
                s = sa + sb;
            }
            sPart[sl * nout + o] = s;
        }
        __syncthreads();
        NTMB_STAMP(5);
        // column-norm term: s_m = sum_n dMhat[n][m] * M_prev[n][m], one wave_sum per column (waves stride over m)
        for (int m = (tid >> 6); m < Md; m += (T >> 6)) {
            float s = 0.f;
            for (int n = lane; n < N; n += 64) s += sDmh[n * MP + m] * sMp[n * MP + m];
            s = wave_sum(s);
            if (lane == 0) {
                const float ci = sCinv[m];
                sC2[m] = (sCss[m] > 1e-12f) ? -ci * ci * ci * s : 0.f;   // dM += M * C2 (2 * d css)
            }
        }
        if (tid < nout) {
            float s = 0.f;
            for (int sl = 0; sl < (FIX ? 1 : nslP); ++sl) s += sPart[sl * nout + tid];
            if (tid < H * Md) {
                sDkhat[tid] = s * sCinv[tid % Md];
            } else {
                const int o2 = tid - H * Md;
                const int which = o2 / (Wh * Md);
                const int jm = o2 - which * Wh * Md;
                if (which == 0) { const float e = sU[d.oE + jm]; sDU[d.oE + jm] = s * e * (1.0f - e); }
                else { const float av = sU[d.oA + jm]; sDU[d.oA + jm] = s * (1.0f - av * av); }
            }
        }
        __syncthreads();
        NTMB_STAMP(6);
        if (tid < H * Md) {
            const int h = tid / Md;
            float dot = 0.f;
            for (int m = 0; m < Md; ++m) dot += sDkhat[h * Md + m] * sU[d.oK + h * Md + m];
            const float ki = sKinv[h];
            const float ck = (sKss[h] > 1e-12f) ? -ki * ki * ki * dot : 0.f;
            const float kv = sU[d.oK + tid];
            const float dk = ki * sDkhat[tid] + kv * ck;
            sDU[d.oK + tid] = dk * (1.0f - kv * kv);
        }
        for (int idx = tid; idx < NMd; idx += T) {
            const int n = idx / Md, m = idx - n * Md, ai = n * MP + m;
            sdM[ai] += sCinv[m] * sDmh[ai] + sMp[ai] * sC2[m];
        }
        __syncthreads();
        NTMB_STAMP(7);
        if (tid < PP) a.du[bt * PP + tid] = sDU[tid];

        // ------------------------------------------------ B9: dh' = carried dh + dU . Wa^T
        if (tid < nslH * hg4) {
            const int cg = tid % hg4, sl = tid / hg4;
            const int c0 = sl * nperH, c1 = min(PP, c0 + nperH);
            if constexpr (FIX) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const int cs = min(c1, c0 + NTMB_RES_WA);
                const f32x4 str = ntk_stream_matvec_exact<4>(reinterpret_cast<const f32x4*>(a.WaT) + cg, hg4, sDU, cs, c1);
#pragma unroll
                for (int q = 0; q < NTMB_RES_WA; ++q) acc += ((c0 + q < c1) ? sDU[c0 + q] : 0.f) * sWaRes4[q * (nslH * hg4) + tid];
                sPart4[sl * hg4 + cg] = acc + str;
            } else {
                sPart4[sl * hg4 + cg] = ntk_stream_matvec<(MAXT > 768 ? 2 : 4)>(reinterpret_cast<const f32x4*>(a.WaT) + cg, hg4, sDU, c0, c1, PP - 1);
            }
        }
        __syncthreads();
        NTMB_STAMP(8);
        // ------------------------------------------------ B10: LSTM cell backward
        if (tid < hid) {
            float dh = WS ? sPartH[tid] + sPartH[hid + tid] : sdZ[RM + tid];
            for (int sl = 0; sl < nslH; ++sl) dh += sPart[sl * ldhT + tid];
            const f32x4 g = reinterpret_cast<const f32x4*>(sGt)[tid];
            const float gi = g[0], gj = g[1], gf = g[2], go = g[3];
            const float tc = tanhf(sCt[tid]);
            const float dct = sdC[tid] + dh * go * (1.0f - tc * tc);
            f32x4 dg;
            dg[0] = dct * gj * gi * (1.0f - gi);
            dg[1] = dct * gi * (1.0f - gj * gj);
            dg[2] = dct * sCp[tid] * gf * (1.0f - gf);
            dg[3] = dh * tc * go * (1.0f - go);
            sdC[tid] = dct * gf;
            reinterpret_cast<f32x4*>(sDG)[tid] = dg;
            reinterpret_cast<f32x4*>(a.dgates)[bt * hid + tid] = dg;
        }
        __syncthreads();
        NTMB_STAMP(9);
        // ------------------------------------------------ B11: d[read_prev; h_prev] = dgates . Wr^T
        if constexpr (WS) {
            // the 80 read columns only (20 float4 column groups x 32 row slices of 25): the h columns are the stream waves'
            const int cg = tid % 20, sl = tid / 20;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < NTMB_RES_B11; ++q) acc += sDG[sl * 25 + q] * wres11[q];       // (same order of the 25 rows as before)
            sPart4[sl * 20 + cg] = ntk_stream_matvec_exact<4>(reinterpret_cast<const f32x4*>(a.WrT) + cg, kg4, sDG, sl * 25 + NTMB_RES_B11, sl * 25 + 25, acc);
        } else if (tid < nslZ * kg4) {
            const int cg = tid % kg4, sl = tid / kg4;
            const int r0 = sl * nperZ, r1 = min(4 * hid, r0 + nperZ);
            sPart4[sl * kg4 + cg] = ntk_stream_matvec<(MAXT > 768 ? 2 : 8)>(reinterpret_cast<const f32x4*>(a.WrT) + cg, kg4, sDG, r0, r1, 4 * hid - 1);
        }
        __syncthreads();
        NTMB_STAMP(10);
        if constexpr (WS) {
            if (tid < RM) {
                float s = 0.f;
#pragma unroll 8
                for (int sl = 0; sl < 32; ++sl) s += sPart[sl * RM + tid];
                sdZ[tid] = s;
            }
        } else if (tid < K) {
            float s = 0.f;
            for (int sl = 0; sl < nslZ; ++sl) s += sPart[sl * ldkT + tid];
            sdZ[tid] = s;
        }
        if (t > 0) commit();       // next (earlier) step's records: every reader of the old ones has passed a barrier
        __syncthreads();
        NTMB_STAMP(11);
    }

#ifdef NTK_CL_PROF
    if (blockIdx.x == 0 && threadIdx.x == 0) for (int i = 0; i < 16; ++i) g_ntm_bwd_prof[i] = s_prof[i];
#endif
    // ---- gradient of the initial state
    for (int i = tid; i < NMd; i += T) a.dM0[(size_t)b * NMd + i] = sdM[(i / Md) * MP + (i % Md)];
    for (int i = tid; i < HN; i += T) a.dw0[(size_t)b * HN + i] = sdW[i];
    for (int i = tid; i < RM; i += T) a.dread0[(size_t)b * RM + i] = sdZ[i];
    if constexpr (WS) {
        // the stream waves finish the last lap (d h_{-1}) behind eight more barriers and one for the hand-over
#pragma unroll
        for (int i = 0; i < 9; ++i) __syncthreads();
    }
    for (int i = tid; i < hid; i += T) {
        a.dcs0[(size_t)b * 2 * hid + i] = sdC[i];
        a.dcs0[(size_t)b * 2 * hid + hid + i] = WS ? sPartH[i] + sPartH[hid + i] : sdZ[RM + i];
    }
}

int ntm_validate_dims(const NtmDims& d, const char* who);

// [rows][cols] -> [cols][ldo] (zero padded), used for WrT / WaT once per optimiser step
__global__ void transpose_pad_kernel(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo,
                                     int rows, int cols) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int r = r0 + i, c = c0 + threadIdx.x;
        tile[i][threadIdx.x] = (r < rows && c < cols) ? in[(size_t)r * ldi + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int c = c0 + i, r = r0 + threadIdx.x;
        if (c < cols && r < ldo) out[(size_t)c * ldo + r] = (r < rows) ? tile[threadIdx.x][i] : 0.f;
    }
}

extern "C" int ntk_transpose_pad(const float* in, int ldi, float* out, int ldo, int rows, int cols, void* stream) {
    NTK_REQUIRE(in && out, NTK_ERR_BAD_PTR, "ntk_transpose_pad: null pointer");
    NTK_REQUIRE(rows > 0 && cols > 0 && ldi >= cols && ldo >= rows, NTK_ERR_BAD_SHAPE,
                "ntk_transpose_pad: rows=%d cols=%d ldi=%d ldo=%d", rows, cols, ldi, ldo);
    dim3 grid((cols + 31) / 32, (ldo + 31) / 32), block(32, 8);
    transpose_pad_kernel<<<grid, block, 0, (hipStream_t)stream>>>(in, ldi, out, ldo, rows, cols);
    NTK_CHECK_LAUNCH("ntk_transpose_pad");
    return NTK_OK;
}

extern "C" int ntk_ntm_seq_bwd(int B, int S, int N, int Md, int R, int Wh, int hid, int shift_range, int O,
                               int write_first,
                               const float* WrT, int ldkT, const float* WaT, int ldhT,
                               const float* M0, const float* w0, const float* cs0,
                               const float* st_gates, const float* st_c, const float* st_u,
                               const float* st_wc, const float* st_wv, const float* st_w, const float* st_M,
                               const float* dlogits,
                               const float* dM_fin, const float* dw_fin, const float* dread_fin, const float* dcs_fin,
                               float* dgates, float* du, float* dM0, float* dw0, float* dread0, float* dcs0,
                               void* stream) {
    NtmBwdArgs a;
    ntm_fill_dims(a.d, B, S, N, Md, R, Wh, hid, shift_range, O, write_first);
    int rc = ntm_validate_dims(a.d, "ntk_ntm_seq_bwd");
    if (rc != NTK_OK) return rc;
    NTK_REQUIRE(WrT && WaT && M0 && w0 && cs0 && st_gates && st_c && st_u && st_wc && st_wv && st_w && st_M &&
                    dlogits && dgates && du && dM0 && dw0 && dread0 && dcs0,
                NTK_ERR_BAD_PTR, "ntk_ntm_seq_bwd: null pointer");
    NTK_REQUIRE(ntk_aligned16(WrT) && ntk_aligned16(WaT) && ntk_aligned16(st_gates) && ntk_aligned16(dgates),
                NTK_ERR_BAD_PTR, "ntk_ntm_seq_bwd: WrT/WaT/st_gates/dgates must be 16-byte aligned");
    NTK_REQUIRE((hid % 4) == 0, NTK_ERR_UNSUPPORTED, "ntk_ntm_seq_bwd: hidden=%d must be a multiple of 4", hid);
    NTK_REQUIRE(ldkT >= a.d.K && (ldkT % 4) == 0 && ldhT >= hid && (ldhT % 4) == 0, NTK_ERR_BAD_SHAPE,
                "ntk_ntm_seq_bwd: ldkT=%d (K=%d) ldhT=%d (hid=%d)", ldkT, a.d.K, ldhT, hid);
    NTK_REQUIRE(a.d.SS + 1 <= NQ, NTK_ERR_UNSUPPORTED, "ntk_ntm_seq_bwd: shift_range=%d too wide", shift_range);
    int T = a.d.H * a.d.N;
    T = ntm_imax(T, 3 * hid);
    T = ntm_imax(T, a.d.PP);
    T = ntm_imax(T, a.d.K);
    T = ((T + 63) / 64) * 64;
    T = ntm_imax(T, a.d.H * a.d.Md + a.d.Md + 2 * a.d.Wh * a.d.Md);
    T = ((T + 63) / 64) * 64;
    NTK_REQUIRE(T <= 1024 && a.d.H * a.d.N <= 1024 && a.d.N * a.d.Md <= MAXM * T, NTK_ERR_UNSUPPORTED,
                "ntk_ntm_seq_bwd: heads*mem_size=%d (max 1024) / mem_size*mem_dim=%d exceed one workgroup",
                a.d.H * a.d.N, a.d.N * a.d.Md);
    a.WrT = WrT; a.WaT = WaT; a.ldkT = ldkT; a.ldhT = ldhT; a.M0 = M0; a.w0 = w0; a.cs0 = cs0;
    a.st_gates = st_gates; a.st_c = st_c; a.st_u = st_u; a.st_wc = st_wc; a.st_wv = st_wv; a.st_w = st_w; a.st_M = st_M;
    a.dlogits = dlogits; a.dM_fin = dM_fin; a.dw_fin = dw_fin; a.dread_fin = dread_fin; a.dcs_fin = dcs_fin;
    a.dgates = dgates; a.du = du; a.dM0 = dM0; a.dw0 = dw0; a.dread0 = dread0; a.dcs0 = dcs0;
    NtmBwdLds L;
    ntm_bwd_lds(a.d, T, ldkT, ldhT, L);
    const bool fix = (N == 128 && Md == 20 && R == 4 && Wh == 1 && hid == 200 && shift_range == 1 && O == 2 &&
                      T == 640 && !write_first && ldkT == 280 && ldhT == 200);
    size_t lds_bytes = (size_t)L.total * sizeof(float) + 128;                // + the diagnostic build's stamp words
    if (fix) lds_bytes += (size_t)NTMB_RES_WA * (T / (ldhT / 4)) * (ldhT / 4) * sizeof(f32x4);     // resident rows of Wa^T
    // benchmark shape: the form whose h columns of Wr^T stream beside the step (NTK_NTM_BWD_FORM=res: round 2's kernel, for comparison)
    const char* form_env = getenv("NTK_NTM_BWD_FORM");
    const bool ws = fix && !(form_env && form_env[0] == 'r');
    if (ws) lds_bytes += (size_t)2 * hid * sizeof(float);
    NTK_REQUIRE(lds_bytes <= 160 * 1024, NTK_ERR_UNSUPPORTED, "ntk_ntm_seq_bwd: needs %zu B of LDS (> 160 KiB)", lds_bytes);
    {
        static NtkLdsAttrCache lds_cache;
        const void* const ks[] = {(const void*)ntm_seq_bwd_kernel<768, false>, (const void*)ntm_seq_bwd_kernel<1024, false>, (const void*)ntm_seq_bwd_kernel<768, true>,
                                  (const void*)ntm_seq_bwd_kernel<768, true, true>};
        const int rc_lds = ntk_raise_lds_limit(lds_cache, ks, 4, "ntk_ntm_seq_bwd");
        if (rc_lds != NTK_OK) return rc_lds;
    }
    if (ws) ntm_seq_bwd_kernel<768, true, true><<<B, 768, lds_bytes, (hipStream_t)stream>>>(a, L);
    else if (fix) ntm_seq_bwd_kernel<768, true><<<B, T, lds_bytes, (hipStream_t)stream>>>(a, L);
    else if (T <= 768) ntm_seq_bwd_kernel<768, false><<<B, T, lds_bytes, (hipStream_t)stream>>>(a, L);
    else ntm_seq_bwd_kernel<1024, false><<<B, T, lds_bytes, (hipStream_t)stream>>>(a, L);
    NTK_CHECK_LAUNCH("ntk_ntm_seq_bwd");
    return NTK_OK;
}
