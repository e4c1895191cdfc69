// DNC core sequence forward, MEMORY-PARTITIONED cluster form: k workgroups (512 threads, one per CU) per sequence, the
// N x N temporal link streamed through HBM (N/k rows per workgroup), the N x W memory partitioned by rows and LDS resident.
// Same arithmetic as dnc_seq_fwd.hip (dnc/dnc.py:84-127, dnc/access.py:113-303, dnc/addressing.py; see that file for the
// step's phases P1..P8); partition and hand-offs: dnc_mp.h.  This is the form that serves BASELINE configs[4]'s core
// (512 x 128: 1 MiB of link and 256 KiB of memory per sequence-step), which the LDS-resident form (dnc_cluster_fwd.hip)
// cannot hold, and any shape whose B * k workgroups fit the chip.
//
// Link pass (dnc/addressing.py:183-218 + :155-181 in ONE sweep over the own rows): a wave owns whole rows (a lane: one
// float4 of each 256-column half), four rows in flight per wave;
//   L_t[r][c] = (1 - ww[r] - ww[c]) L_{t-1}[r][c] + ww[r] p_{t-1}[c], diagonal 0  -> stored (record t, or in place)
//   fwd[i][r] = sum_c rw_{t-1}[i][c] L_t[r][c]                (complete for the own rows: wave reduction)
//   bwd[i][c] += rw_{t-1}[i][r] L_t[r][c]                     (partial over the own rows: per-lane accumulators, reduced over
//                                                             the waves in a fixed order, summed over the workgroups by
//                                                             the consumers in a fixed order: bitwise reproducible)
#include "dnc_mp.h"
#include <type_traits>
#include <vector>

// Diagnostic build only (-DNTK_CL_PROF): workgroup 0's thread 0 adds s_memtime deltas per phase to g_mpf_prof (global atomics: no
// registers held across the step); ntk_dnc_mp_fwd_prof copies them out.  The stamps serialise the phases: read SHARES.
#ifdef NTK_CL_PROF
__device__ unsigned long long g_mpf_prof[24];
#define MP_STAMP(i)                                                                   \
    do {                                                                              \
        if (blockIdx.x == 0 && tid == 0) {                                            \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();             \
            __hip_atomic_fetch_add(&g_mpf_prof[i], now_ - prof_last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
            prof_last = now_;                                                         \
        }                                                                             \
    } while (0)
#else
#define MP_STAMP(i) do { } while (0)
#endif

namespace {

struct DncMpFwdLds {
    int part, M, Z, C, HP, I, K, U, NU, KEY, RANK, RW, WW, P, CW, CR, SC, total;
};

constexpr __host__ __device__ DncMpFwdLds dnc_mp_fwd_lds(const DncMpCfg& c) {
    DncMpFwdLds L = {};
    const int N = c.N, RWd = c.R * c.W;
    int part = c.ksl * c.upk * 4;
    part = dnc_cluster_max(part, c.nslI * c.IP);
    part = dnc_cluster_max(part, N);                       // rank-ordered usage vector
    part = dnc_cluster_max(part, CW * c.R * 256);          // link pass: per-wave backward-read partials of one 256-column half
    part = dnc_cluster_max(part, c.nslR * RWd);
    int o = 0;
    L.part = o; o += dnc_cluster_align4(part);
    L.M = o; o += c.NR * c.WS4 * 4;
    L.Z = o; o += dnc_cluster_align4(c.K);
    L.C = o; o += dnc_cluster_align4(c.upk);
    L.HP = o; o += dnc_cluster_align4(c.upk);
    L.I = o; o += dnc_cluster_align4(c.IP);
    L.K = o; o += dnc_cluster_align4((1 + c.R) * c.W);
    L.U = o; o += dnc_cluster_align4(N);
    L.NU = o; o += dnc_cluster_align4(N);
    L.KEY = o; o += dnc_cluster_align4(2 * N);
    L.RANK = o; o += dnc_cluster_align4(N);
    L.RW = o; o += dnc_cluster_align4(c.R * N);
    L.WW = o; o += dnc_cluster_align4(N);
    L.P = o; o += dnc_cluster_align4(N);
    L.CW = o; o += dnc_cluster_align4(N);
    L.CR = o; o += dnc_cluster_align4(c.R * N);
    L.SC = o; o += 64;
    L.total = o;
    return L;
}

struct DncMpFwdArgs {
    int B, S, xcd_local;
    float clip;
    DncMpCfg c;
    DncMpFwdLds lds;
    const float* xproj; const float* Wr; const float* Wi; const float* Wy;
    float* mem; float* link; float* usage; float* rw; float* ww; float* prec; float* reads; float* hc; float* out;
    float* rec_z; float* rec_gates; float* rec_c; float* rec_hc; float* rec_yin; float* rec_ifc; float* rec_u;
    float* rec_ww; float* rec_rw; float* rec_cw; float* rec_cr; float* rec_al; float* rec_p; float* rec_fwd;
    float* rec_bwd; float* rec_M; float* rec_L; float* rec_ypre;
    float* mbox; unsigned* flags; unsigned* err; unsigned* xcc; unsigned* sticky;
};

__device__ __forceinline__ void mp_softmax_row(float* r, int N, int lane) {      // one wave, in place
    float mx = -INFINITY;
    for (int n = lane; n < N; n += 64) mx = fmaxf(mx, r[n]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int n = lane; n < N; n += 64) { const float e = expf(r[n] - mx); r[n] = e; s += e; }
    s = wave_sum(s);
    for (int n = lane; n < N; n += 64) r[n] = r[n] / s;
}

__device__ __forceinline__ float mp_dot4(const f32x4& x, const f32x4& y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3]; }

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));


#define MP_FWD_VIEWS()                                                                                                        \
    const int k = C.k, NR = C.NR, upk = C.upk, upkp = C.upkp;                                                                 \
    const int N = C.N, W = C.W, R = C.R;                                                                                      \
    const int hid = C.hid, K = C.K, IP = C.IP, RWd = R * W, N4 = C.N4, W4 = C.W4, WS4 = C.WS4;                                \
    const int row0 = g * NR, u0 = min(hid, g * upk), u1 = min(hid, u0 + upk), nU = u1 - u0;                                   \
    float* sPart = smem + L.part; float* sM = smem + L.M;                                                                     \
    float* sZ = smem + L.Z; float* sC = smem + L.C; float* sHP = smem + L.HP; float* sI = smem + L.I; float* sK = smem + L.K; \
    float* sU = smem + L.U; float* sNU = smem + L.NU;                                                                         \
    unsigned long long* sKEY = reinterpret_cast<unsigned long long*>(smem + L.KEY);                                          \
    int* sRank = reinterpret_cast<int*>(smem + L.RANK);                                                                       \
    float* sRW = smem + L.RW; float* sWW = smem + L.WW;                                                                       \
    float* sP = smem + L.P; float* sCW = smem + L.CW; float* sCR = smem + L.CR;                                               \
    float* sSC = smem + L.SC; int* sAbort = reinterpret_cast<int*>(sSC + 32);                                                 \
    f32x4* sPart4 = reinterpret_cast<f32x4*>(sPart); f32x4* sM4 = reinterpret_cast<f32x4*>(sM);                              \
    const f32x4* sK4 = reinterpret_cast<const f32x4*>(sK);                                                                    \
    (void)upkp; (void)K; (void)IP; (void)RWd; (void)N4; (void)W4; (void)WS4; (void)u1; (void)nU; (void)sPart4; (void)sK4;    \
    (void)sKEY; (void)sRank; (void)sNU; (void)sCW; (void)sCR; (void)sK; (void)sAbort; (void)sM; (void)sI; (void)sC; (void)sHP; (void)k; \
    (void)sM4; (void)sU; (void)sP; (void)sWW; (void)sRW; (void)row0

template <int SH>
__global__ __launch_bounds__(CT) void dnc_mp_fwd_kernel(DncMpFwdArgs a0) {
    constexpr bool FIX = SH != 0;
    constexpr DncMpCfg kDncMpFixCfg = dnc_mp_shape_cfg(SH);
    constexpr DncMpFwdLds kDncMpFixFwdLds = dnc_mp_fwd_lds(kDncMpFixCfg);
    // link rows a wave keeps in flight: 4 rows of two float4 per lane (512 columns: the pass is bandwidth bound), 8 rows of one
    // float4 where a row fits one 256-column half (256 x 64 at k = 4: a wave owns 8 rows, two batches were two HBM round trips)
    constexpr int MP_PFL = (FIX && kDncMpFixCfg.NH == 1) ? 8 : 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef const __attribute__((address_space(4))) DncMpFwdArgs* ArgsK;
    const ArgsK ak0 = (ArgsK)__builtin_amdgcn_kernarg_segment_ptr();
    const int tid0 = threadIdx.x;
    const int kk0 = FIX ? kDncMpFixCfg.k : a0.c.k;
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

    // ---- load state (own memory rows; per-slot vectors replicated)
    {
        const DncMpFwdArgs& a = a0;
        const DncMpCfg C = FIX ? kDncMpFixCfg : a.c;
        const DncMpFwdLds L = FIX ? kDncMpFixFwdLds : a.lds;
        MP_FWD_VIEWS();
        if (tid0 == 0) *sAbort = 0;
        const f32x4* gM4 = reinterpret_cast<const f32x4*>(a.mem + ((size_t)b * N + row0) * W);
        for (int i = tid0; i < NR * W4; i += CT) { const int n = i / W4, j = i - n * W4; sM4[n * WS4 + j] = gM4[i]; }
        for (int i = tid0; i < N; i += CT) {
            sU[i] = a.usage[(size_t)b * N + i];
            sWW[i] = a.ww[(size_t)b * N + i];
            sP[i] = a.prec[(size_t)b * N + i];
        }
        for (int i = tid0; i < R * N; i += CT) sRW[i] = a.rw[(size_t)b * R * N + i];
        for (int i = tid0; i < RWd; i += CT) sZ[i] = a.reads[(size_t)b * RWd + i];
        for (int i = tid0; i < hid; i += CT) sZ[RWd + i] = a.hc[(size_t)b * 2 * hid + i];
        for (int i = tid0; i < nU; i += CT) sC[i] = a.hc[(size_t)b * 2 * hid + hid + u0 + i];
        for (int i = tid0; i < (1 + R) * W; i += CT) sK[i] = 0.f;
    }
    __syncthreads();
    bool plain = false;
    if (a0.xcd_local) {
        int* const sw = reinterpret_cast<int*>(smem + (FIX ? kDncMpFixFwdLds.SC : a0.lds.SC)) + 33;
        const int same = cl_same_xcd(a0.xcc + (size_t)b * kk0, g, kk0, a0.err, sw - 1, sw, t_start, tid0);
        if (same < 0) { if (tid0 == 0) __hip_atomic_store(a0.sticky, 1u, NTK_RLX, NTK_AGENT); return; }
        plain = __builtin_amdgcn_readfirstlane(same) != 0;
    }

#ifdef NTK_CL_PROF
    unsigned long long prof_last = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && tid0 == 0) for (int i = 0; i < 24; ++i) g_mpf_prof[i] = 0;
#endif
    for (int t = 0; t < S; ++t) {
        ArgsK ak = ak0;
        asm volatile("" : "+s"(ak));
        const auto& a = *ak;
        DncMpCfg C = kDncMpFixCfg;
        DncMpFwdLds L = kDncMpFixFwdLds;
        if constexpr (!FIX) {                  // generic shape: the two blocks come from the kernarg segment, inside the step
            __builtin_memcpy(&C, (const void*)&a.c, sizeof(C));
            __builtin_memcpy(&L, (const void*)&a.lds, sizeof(L));
        }
        MP_FWD_VIEWS();
        const float clipv = a.clip;
        const int ksl = C.ksl, kperG = C.kperG, icg = C.icg, nslI = C.nslI, uperI = C.uperI;
        const int TPR = C.TPR, FPT = C.FPT, RPP = C.RPP, NH = C.NH;
        const int sl0 = C.slot[0], slA = C.slot[1], slB = C.slot[2], slC = C.slot[3];
        float* mb0 = a.mbox + (size_t)b * 2 * k * ((size_t)sl0 + slA + slB + slC);     // [parity][g][slot] per hand-off
        float* mbA = mb0 + (size_t)2 * k * sl0;
        float* mbB = mbA + (size_t)2 * k * slA;
        float* mbC = mbB + (size_t)2 * k * slB;
        unsigned* fl0 = a.flags + (size_t)b * MPX * k;
        unsigned* flA = fl0 + k; unsigned* flB = flA + k; unsigned* flC = flB + k;
        const f32x4* Wr4 = reinterpret_cast<const f32x4*>(a.Wr);
        const f32x4* Wi4 = reinterpret_cast<const f32x4*>(a.Wi);
        const bool rec = a.rec_z != nullptr;
        int tid_op = tid0;
        asm volatile("" : "+v"(tid_op));       // keep per-thread index math inside the step (no hoist + spill)
        const int tid = tid_op, lane = tid & 63, wave = tid >> 6;
        const size_t bt = (size_t)b * S + t;
        const unsigned epoch = (unsigned)t + 1u;
        const int par = t & 1;

        MP_STAMP(0);       // loop top / previous step's tail
        // ------------------------------------------------------------ P1: LSTM gates of the own hidden units
        f32x4 xg = {0.f, 0.f, 0.f, 0.f};
        if (tid < nU) xg = reinterpret_cast<const f32x4*>(a.xproj)[bt * hid + u0 + tid] + Wr4[(size_t)K * hid + u0 + tid];
        if (rec && g == 0) for (int i = tid; i < C.ldz; i += CT) a.rec_z[bt * C.ldz + i] = (i < K) ? sZ[i] : (i == K ? 1.f : 0.f);
        if (tid < ksl * upk) {
            const int ks = cl_div(tid, C.mg_upk), j = tid - ks * upk;
            const int k0 = ks * kperG, k1 = min(K, k0 + kperG);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (j < nU && k0 < k1) acc = ntk_stream_matvec<4>(Wr4 + u0 + j, hid, sZ, k0, k1, K - 1);
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
        MP_STAMP(1);       // P1 gates + LSTM
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
        MP_STAMP(2);       // P2 interface partial
        {   // publish hand-off 0: [h of the own units | interface partial]
            float* slot = mb0 + ((size_t)par * k + g) * sl0;
            if (tid < nU) cl_store(slot + tid, sZ[RWd + u0 + tid], plain);
            for (int c = tid; c < IP; c += CT) {
                float v = 0.f;
                for (int us = 0; us < nslI; ++us) v += sPart[us * IP + c];
                cl_store(slot + upkp + c, v, plain);
            }
            cl_publish(fl0 + g, epoch, tid, plain);
        }
        // y_{t-1} = clip([h_{t-1} ; reads_{t-1}] Wy + by) (dnc.py:118-122) does not feed the recurrence: workgroup 0 computes it
        // in the shadow of the hand-off.  h_{t-1}: sHP for the own units (already overwritten in sZ), sZ for the others.
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
        MP_STAMP(3);       // publish 0 + deferred output
        if (!mp_wait(fl0, epoch, k, a.err, a.sticky, sAbort, tid)) return;
        MP_STAMP(4);       // wait 0
        {   // consume hand-off 0: full h, activated interface (+ aligned copies of the keys)
            const float* base = mb0 + (size_t)par * k * sl0;
            for (int u = tid; u < hid; u += CT) {
                const int gg = cl_div(u, C.mg_upk);
                sZ[RWd + u] = cl_load(base + (size_t)gg * sl0 + (u - gg * upk));
            }
            const __amdgpu_buffer_rsrc_t rs = mp_rsrc(base, (size_t)k * sl0);
            for (int c4 = tid; c4 < (IP >> 2); c4 += CT) {          // 16-byte loads: four interface columns per thread
                const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                f32x4 pv[8];
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) pv[gg] = (gg < k) ? mp_load4(rs, gg * sl0 + upkp + 4 * c4) : z4;
                f32x4 v4 = reinterpret_cast<const f32x4*>(a.Wi + (size_t)hid * IP)[c4];
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) if (gg < k) v4 += pv[gg];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 4 * c4 + e;
                    const float v = v4[e];
                    float r = v;
                    if (c >= C.oE && c < C.oRm) r = dnc_sigmoid(v);                      // erase, free, alloc, write gates
                    else if ((c >= C.oBw && c < C.oKr) || (c >= C.oBr && c < C.I)) r = dnc_softplus(v);   // strengths
                    sI[c] = r;
                    if (c >= C.oKw && c < C.oBw) sK[c - C.oKw] = r;
                    else if (c >= C.oKr && c < C.oBr) sK[W + (c - C.oKr)] = r;
                }
            }
        }
        __syncthreads();
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
        MP_STAMP(5);       // consume 0 + records
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
        MP_STAMP(6);       // key norms, P3 usage
        float* slotA = mbA + ((size_t)par * k + g) * slA;                          // [scores NR | rank partial counts N]
        // ------------------------------------------------------------ P4: write content scores of the own rows on M_{t-1}
        {
            const int h = tid & (TPR - 1), rr = tid / TPR;
            const float kn = sSC[8], bw = sI[C.oBw];
            for (int r = rr; r < NR; r += RPP) {
                float nsq = 0.f, dot = 0.f;
                const f32x4* mr = sM4 + r * WS4;
                for (int jj = 0; jj < FPT; ++jj) {
                    const int j = h + jj * TPR;
                    if (j < W4) { const f32x4 m = mr[j]; nsq += mp_dot4(m, m); dot += mp_dot4(sK4[j], m); }
                }
                nsq = group_sum_rt(nsq, TPR);
                dot = group_sum_rt(dot, TPR);
                if (h == 0) cl_store(slotA + r, (dot * cl_rcp(kn * cl_sqrt(nsq + EPS) + EPS)) * bw, plain);
            }
        }
        // ------------------------------------------------------------ P5a: rank of every slot among the own N/k keys (partial count)
        for (int n = tid; n < N; n += CT) {
            const unsigned long long mine = sKEY[n];
            const u64x2* kp = reinterpret_cast<const u64x2*>(sKEY + g * C.mperA);
            int cnt = 0;
            for (int m = 0; m < C.mperA; m += 8) {
                const u64x2 k0 = kp[(m >> 1)], k1 = kp[(m >> 1) + 1], k2 = kp[(m >> 1) + 2], k3 = kp[(m >> 1) + 3];
                cnt += (k0[0] > mine) + (k0[1] > mine) + (k1[0] > mine) + (k1[1] > mine) + (k2[0] > mine) + (k2[1] > mine) +
                       (k3[0] > mine) + (k3[1] > mine);
            }
            cl_store(slotA + NR + n, __int_as_float(cnt), plain);
        }
        MP_STAMP(7);       // P4 write scores + P5a rank partial
        cl_publish(flA + g, epoch, tid, plain);
        if (!mp_wait(flA, epoch, k, a.err, a.sticky, sAbort, tid)) return;
        MP_STAMP(8);       // publish A + wait A
        // ------------------------------------------------------------ P5b: ranks; usages scattered into rank order; write-content scores
        {
            const float* base = mbA + (size_t)par * k * slA;
            float* sT = sPart;
            const __amdgpu_buffer_rsrc_t rs = mp_rsrc(base, (size_t)k * slA);
            for (int n4 = tid; n4 < N4; n4 += CT) {
                const int n = 4 * n4;
                const int og = cl_div(n, C.mg_NR);
                i32x4 rk = {0, 0, 0, 0};
                i32x4 pv[8];
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) pv[gg] = (gg < k) ? mp_load4i(rs, gg * slA + NR + n) : rk;
                *reinterpret_cast<f32x4*>(sCW + n) = mp_load4(rs, og * slA + (n - og * NR));
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) if (gg < k) rk += pv[gg];
                *reinterpret_cast<i32x4*>(sRank + n) = rk;
#pragma unroll
                for (int e = 0; e < 4; ++e) sT[rk[e]] = 1.0f - sNU[n + e];      // sorted_usage = 1 - sorted_nonusage (addressing.py:398)
            }
            __syncthreads();
            if (wave == CW - 1) mp_softmax_row(sCW, N, lane);
            // P5c: exclusive cumulative product in rank order (tf.cumprod(exclusive=True), addressing.py:399) by wave 0
            if (wave == 0) {
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
        MP_STAMP(9);       // P5b-d allocation, write weights
        float* slotB = mbB + ((size_t)par * k + g) * slB;              // [fwd R x NR | read scores R x NR | bwd partial R x N]
        if (wave == CW - 1) {                                          // sum of the write weights (precedence update)
            float s = 0.f;
            for (int n = lane; n < N; n += 64) s += sWW[n];
            s = wave_sum(s);
            if (lane == 0) sSC[0] = s;
        }
        // ------------------------------------------------------------ P6: erase + write on the own rows of M, read-key scores on M_t
        {
            const int h = tid & (TPR - 1), rr = tid / TPR;
            const f32x4* sE4 = reinterpret_cast<const f32x4*>(sI + C.oE);
            const f32x4* sV4 = reinterpret_cast<const f32x4*>(sI + C.oV);
            float krn[4], br[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { krn[i] = (i < R) ? sSC[9 + i] : 1.f; br[i] = (i < R) ? sI[C.oBr + i] : 0.f; }
            for (int r = rr; r < NR; r += RPP) {
                const float wwn = sWW[row0 + r];
                f32x4* mr = sM4 + r * WS4;
                float nsq = 0.f, dot[4] = {0.f, 0.f, 0.f, 0.f};
                for (int jj = 0; jj < FPT; ++jj) {
                    const int j = h + jj * TPR;
                    if (j < W4) {
                        f32x4 m = mr[j];
                        const f32x4 ev = sE4[j], vv = sV4[j];
#pragma unroll
                        for (int e = 0; e < 4; ++e) m[e] = m[e] * (1.0f - wwn * ev[e]) + wwn * vv[e];
                        mr[j] = m;
                        if (rec) reinterpret_cast<f32x4*>(a.rec_M + (bt * N + row0 + r) * W)[j] = m;
                        nsq += mp_dot4(m, m);
#pragma unroll
                        for (int i = 0; i < 4; ++i) if (i < R) dot[i] += mp_dot4(sK4[(1 + i) * W4 + j], m);
                    }
                }
                nsq = group_sum_rt(nsq, TPR);
#pragma unroll
                for (int i = 0; i < 4; ++i) if (i < R) dot[i] = group_sum_rt(dot[i], TPR);
                const float nm = cl_sqrt(nsq + EPS);
                if (h == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (i < R) cl_store(slotB + (R + i) * NR + r, (dot[i] * cl_rcp(krn[i] * nm + EPS)) * br[i], plain);
                }
            }
        }
        MP_STAMP(10);      // P6 memory write + read scores
        // ------------------------------------------------------------ P7: link pass over the own rows (HBM stream)
        {
            const float* Lsrc = (rec && t > 0) ? a.rec_L + ((bt - 1) * N + row0) * N : a.link + ((size_t)b * N + row0) * N;
            float* Ldst = rec ? a.rec_L + (bt * N + row0) * N : a.link + ((size_t)b * N + row0) * N;
            float* Lfin = (rec && t == S - 1) ? a.link + ((size_t)b * N + row0) * N : nullptr;      // the state the caller gets back
            f32x4 wwb[2], pb[2], rwb[4][2], accB[4][2];
            bool colok[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c4 = lane + 64 * h;
                colok[h] = h < NH && c4 < N4;
                wwb[h] = f32x4{0.f, 0.f, 0.f, 0.f}; pb[h] = wwb[h];
                if (colok[h]) { wwb[h] = *reinterpret_cast<const f32x4*>(sWW + 4 * c4); pb[h] = *reinterpret_cast<const f32x4*>(sP + 4 * c4); }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    accB[i][h] = f32x4{0.f, 0.f, 0.f, 0.f};
                    rwb[i][h] = accB[i][h];
                    if (i < R && colok[h]) rwb[i][h] = *reinterpret_cast<const f32x4*>(sRW + i * N + 4 * c4);
                }
            }
            auto link_rows = [&](auto nt) {          // nt: std::true_type = non-temporal row loads (two straight-line copies: a
                                                     // per-load select broke the four-rows-in-flight batches)
            for (int r0 = wave; r0 < NR; r0 += CW * MP_PFL) {
                f32x4 lv[MP_PFL][2];
#pragma unroll
                for (int u = 0; u < MP_PFL; ++u) {
                    const int r = r0 + u * CW;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        lv[u][h] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (r < NR && colok[h]) {
                            // recording: the rows of record t-1 are read once -> non-temporal LOADS (they should not push the
                            // weights out of L2; free in the probe, unlike non-temporal stores).  In place (inference) the same
                            // lines are stored right back: plain loads (measured: 45.8 vs 47.0 us per step)
                            const f32x4* lp = reinterpret_cast<const f32x4*>(Lsrc + (size_t)r * N) + lane + 64 * h;
                            lv[u][h] = decltype(nt)::value ? __builtin_nontemporal_load(lp) : *lp;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < MP_PFL; ++u) {
                    const int r = r0 + u * CW;
                    if (r < NR) {                                  // wave-uniform
                        const int ra = row0 + r;
                        const float wwa = sWW[ra];
                        float f[4] = {0.f, 0.f, 0.f, 0.f};
                        float rwa[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) rwa[i] = (i < R) ? sRW[i * N + ra] : 0.f;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            if (colok[h]) {
                                const int c0 = 4 * (lane + 64 * h);
                                f32x4 v;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    float x = (1.0f - wwa - wwb[h][e]) * lv[u][h][e] + wwa * pb[h][e];
                                    if (c0 + e == ra) x = 0.f;                        // matrix_set_diag(link, 0)
                                    v[e] = x;
                                }
                                reinterpret_cast<f32x4*>(Ldst + (size_t)r * N)[lane + 64 * h] = v;
                                if (Lfin) reinterpret_cast<f32x4*>(Lfin + (size_t)r * N)[lane + 64 * h] = v;
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    if (i < R) {
                                        f[i] += mp_dot4(rwb[i][h], v);
                                        accB[i][h] += rwa[i] * v;
                                    }
                                }
                            }
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (i < R) {
                                const float s = wave_sum(f[i]);
                                if (lane == 0) cl_store(slotB + i * NR + r, s, plain);
                            }
                        }
                    }
                }
            }
            };
            if (rec) link_rows(std::true_type{}); else link_rows(std::false_type{});
            MP_STAMP(11);      // P7 link rows
            // backward-read partials: fixed-order reduction over the waves, one 256-column half at a time
            for (int h = 0; h < NH; ++h) {
                if (h > 0) __syncthreads();
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < R) *reinterpret_cast<f32x4*>(sPart + ((wave * R + i) * 256 + lane * 4)) = (h == 0) ? accB[i][0] : accB[i][1];
                __syncthreads();
                for (int idx = tid; idx < R * 256; idx += CT) {
                    const int i = idx >> 8, c = idx & 255, col = 256 * h + c;
                    if (col < N) {
                        float s = 0.f;
#pragma unroll
                        for (int w = 0; w < CW; ++w) s += sPart[(w * R + i) * 256 + c];
                        cl_store(slotB + 2 * R * NR + i * N + col, s, plain);
                    }
                }
            }
        }
        MP_STAMP(12);      // P7 column reduction
        cl_publish(flB + g, epoch, tid, plain);
        if (!mp_wait(flB, epoch, k, a.err, a.sticky, sAbort, tid)) return;
        MP_STAMP(13);      // publish B + wait B
        // ------------------------------------------------------------ P8: read weights, precedence, reads
        {
            const float* base = mbB + (size_t)par * k * slB;
            // 16-byte loads: a thread = four consecutive slots of one head (R * N / 4 quads over the 512 threads)
            const __amdgpu_buffer_rsrc_t rs = mp_rsrc(base, (size_t)k * slB);
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            f32x4 fw_ = z4, bw_ = z4;
            const int q4 = tid;                                        // R * N / 4 <= 512 (R <= 4, N <= 512)
            const bool qok = q4 < ((R * N) >> 2);
            int qi = 0, qn = 0;
            if (qok) {
                const int idx = 4 * q4;
                qi = cl_div(idx, C.mg_N); qn = idx - qi * N;
                const int og = cl_div(qn, C.mg_NR), rl = qn - og * NR;
                f32x4 pv[8];
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) pv[gg] = (gg < k) ? mp_load4(rs, gg * slB + 2 * R * NR + idx) : z4;
                fw_ = mp_load4(rs, og * slB + qi * NR + rl);
                *reinterpret_cast<f32x4*>(sCR + idx) = mp_load4(rs, og * slB + (R + qi) * NR + rl);
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) if (gg < k) bw_ += pv[gg];
            }
            __syncthreads();
            if (wave < R) mp_softmax_row(sCR + wave * N, N, lane);                      // read content weights
            __syncthreads();
            if (qok) {
                const int idx = 4 * q4;
                const float* rm = sI + C.oRm + qi * 3;
                const f32x4 cr = *reinterpret_cast<const f32x4*>(sCR + idx);
                const f32x4 v = rm[2] * cr + rm[1] * fw_ + rm[0] * bw_;                // access.py:283-303 (num_writes = 1)
                *reinterpret_cast<f32x4*>(sRW + idx) = v;
                if (rec && qn >= row0 && qn < row0 + NR) {
                    *reinterpret_cast<f32x4*>(a.rec_rw + bt * R * N + idx) = v;
                    *reinterpret_cast<f32x4*>(a.rec_cr + bt * R * N + idx) = cr;
                    *reinterpret_cast<f32x4*>(a.rec_fwd + bt * R * N + idx) = fw_;
                    *reinterpret_cast<f32x4*>(a.rec_bwd + bt * R * N + idx) = bw_;
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
        MP_STAMP(14);      // P8 read weights
        // reads = rw x M_t over the own rows: thread = (head, float4 of the word) x row slice, slices summed in a fixed order
        if (tid < C.nslR * C.RW4) {
            const int sl = cl_div(tid, C.mg_RW4), o4 = tid - sl * C.RW4;
            const int i = cl_div(o4, C.mg_W4), w4 = o4 - i * W4;
            const int n0 = sl * C.nperR, n1 = min(NR, n0 + C.nperR);
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            const float* rwp = sRW + i * N + row0;
#pragma unroll 4
            for (int n = n0; n < n1; ++n) s += rwp[n] * sM4[n * WS4 + w4];
            sPart4[sl * C.RW4 + o4] = s;
        }
        __syncthreads();
        {
            float* slotC = mbC + ((size_t)par * k + g) * slC;
            for (int c = tid; c < RWd; c += CT) {
                float s = 0.f;
                for (int sl = 0; sl < C.nslR; ++sl) s += sPart[sl * RWd + c];
                cl_store(slotC + c, s, plain);
            }
            cl_publish(flC + g, epoch, tid, plain);
        }
        MP_STAMP(15);      // reads partial + publish C
        if (!mp_wait(flC, epoch, k, a.err, a.sticky, sAbort, tid)) return;
        {
            const float* base = mbC + (size_t)par * k * slC;
            const __amdgpu_buffer_rsrc_t rs = mp_rsrc(base, (size_t)k * slC);
            for (int c4 = tid; c4 < (RWd >> 2); c4 += CT) {
                const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                f32x4 pv[8];
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) pv[gg] = (gg < k) ? mp_load4(rs, gg * slC + 4 * c4) : z4;
                f32x4 s = z4;
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) if (gg < k) s += pv[gg];
                *reinterpret_cast<f32x4*>(sZ + 4 * c4) = s;
                if (rec && g == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) a.rec_yin[bt * C.ldy + hid + 4 * c4 + e] = s[e];
                }
            }
            if (rec && g == 0 && tid < C.ldy - C.Ky) a.rec_yin[bt * C.ldy + C.Ky + tid] = (tid == 0) ? 1.f : 0.f;
        }
        MP_STAMP(16);      // wait C
        __syncthreads();
    }
#ifdef NTK_CL_PROF
    (void)prof_last;
#endif

    // ---- store state: own rows of the memory, own units of the cell; the replicated vectors by workgroup 0
    {
        const DncMpFwdArgs& a = a0;
        const DncMpCfg C = FIX ? kDncMpFixCfg : a.c;
        const DncMpFwdLds L = FIX ? kDncMpFixFwdLds : a.lds;
        MP_FWD_VIEWS();
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
        for (int i = tid0; i < NR * W4; i += CT) { const int n = i / W4, j = i - n * W4; gM4[i] = sM4[n * WS4 + j]; }
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
// the cluster size (0 = none) and configuration of a shape; k_req 0 = the SMALLEST k that fits (least replication of the
// per-slot work, most sequences per launch); B * k workgroups must be co-resident: one per CU
static int dnc_mp_pick(int B, int N, int W, int R, int Wn, int hid, int O, int k_req, DncMpCfg& c, size_t* lds_bytes) {
    if (Wn != 1 || R < 1 || R > 4 || N < 64 || (N % 64) != 0 || N > CT || W < 4 || (W % 4) != 0 || W > 256 || hid < 4 ||
        hid > 1024 || O < 1 || O > CW - 1 || B < 1)
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
        if (c.icg > CT || c.RW4 > CT || c.RNP > 4 || c.upk > CT) continue;
        const DncMpFwdLds L = dnc_mp_fwd_lds(c);
        const size_t bytes = (size_t)L.total * sizeof(float);
        if (bytes > 160 * 1024) continue;
        if (lds_bytes) *lds_bytes = bytes;
        return k;
    }
    return 0;
}

#ifdef NTK_CL_PROF
extern "C" int ntk_dnc_mp_fwd_prof(unsigned long long* out24) {
    return hipMemcpyFromSymbol(out24, HIP_SYMBOL(g_mpf_prof), 24 * sizeof(unsigned long long)) == hipSuccess ? NTK_OK : NTK_ERR_HIP;
}
#endif

extern "C" int ntk_dnc_mp_plan(int B, int N, int W, int R, int Wn, int hid, int O, int k_request, int* k, size_t* workspace_bytes) {
    DncMpCfg c;
    const int kk = dnc_mp_pick(B, N, W, R, Wn, hid, O, k_request, c, nullptr);
    if (k) *k = kk;
    if (workspace_bytes) *workspace_bytes = 0;
    if (kk <= 0) {
        ntk_set_error("ntk_dnc_mp_plan: B=%d N=%d W=%d R=%d Wn=%d hid=%d is outside the memory-partitioned cluster kernels' range "
                      "(num_writes 1, memory_size a multiple of 64 up to 512, N / k memory rows LDS resident, B * k <= the device's CUs)",
                      B, N, W, R, Wn, hid);
        return NTK_ERR_UNSUPPORTED;
    }
    if (workspace_bytes) *workspace_bytes = dnc_mp_workspace_bytes(B, kk, c.slot);
    return NTK_OK;
}

// > 0 when the shape has a compile-time instantiation of the mp kernels (csrc/dnc_mp.h: the generic instantiation is several
// times slower), 0 otherwise
extern "C" int ntk_dnc_mp_compiled_shape(int N, int W, int R, int Wn, int hid, int O, int k) {
    if (Wn != 1 || k < 1 || N < 1 || (N % k) != 0) return 0;
    const DncMpCfg c = dnc_mp_cfg(N, W, R, hid, O, k);
    return dnc_mp_shape_of(c);
}

extern "C" int ntk_dnc_mp_status(const void* workspace, size_t workspace_bytes, int B, int k, int clear_sticky, void* stream) {
    NTK_REQUIRE(workspace && B > 0 && k > 0 && workspace_bytes >= dnc_mp_ctrl_bytes(B, k) + 256, NTK_ERR_BAD_PTR, "ntk_dnc_mp_status: bad arguments");
    unsigned e[2] = {0, 0};
    const unsigned* errw = reinterpret_cast<const unsigned*>(workspace) + (size_t)B * MPX * k;
    const unsigned* stk = reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(workspace) + workspace_bytes - 256);
    hipError_t rc = hipMemcpyAsync(&e[0], errw, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (rc == hipSuccess) rc = hipMemcpyAsync(&e[1], stk, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (rc == hipSuccess && clear_sticky) rc = hipMemsetAsync(const_cast<unsigned*>(stk), 0, sizeof(unsigned), (hipStream_t)stream);
    if (rc == hipSuccess) rc = hipStreamSynchronize((hipStream_t)stream);
    NTK_REQUIRE(rc == hipSuccess, NTK_ERR_HIP, "ntk_dnc_mp_status: %s", hipGetErrorString(rc));
    NTK_REQUIRE(e[0] == 0 && e[1] == 0, NTK_ERR_HIP, "ntk_dnc_mp_status: a cluster hand-off timed out (%s; outputs of that launch are invalid)",
                e[0] ? "last launch" : "an earlier launch on this workspace");
    return NTK_OK;
}

extern "C" int ntk_dnc_mp_placement(const void* workspace, int B, int k, int* same_xcd_clusters, void* stream) {
    NTK_REQUIRE(workspace && same_xcd_clusters && B > 0 && k > 0 && k <= 64, NTK_ERR_BAD_PTR, "ntk_dnc_mp_placement: bad arguments");
    std::vector<unsigned> w((size_t)B * k);
    const unsigned* xcc = reinterpret_cast<const unsigned*>(workspace) + (size_t)B * MPX * k + 1;
    hipError_t rc = hipMemcpyAsync(w.data(), xcc, w.size() * sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (rc == hipSuccess) rc = hipStreamSynchronize((hipStream_t)stream);
    NTK_REQUIRE(rc == hipSuccess, NTK_ERR_HIP, "ntk_dnc_mp_placement: %s", hipGetErrorString(rc));
    int n = 0;
    for (int b = 0; b < B; ++b) {
        bool same = w[(size_t)b * k] != 0;          // 0 = no handshake ran (batch not a multiple of 8)
        for (int g = 1; g < k; ++g) same = same && w[(size_t)b * k + g] == w[(size_t)b * k];
        n += same ? 1 : 0;
    }
    *same_xcd_clusters = n;
    return NTK_OK;
}

extern "C" int ntk_dnc_mp_fwd(int B, int S, int N, int W, int R, int Wn, int hid, int O, float clip_value, int k,
                              const float* xproj, const float* Wr, const float* Wi, const float* Wy,
                              float* mem, float* link, float* usage, float* rw, float* ww, float* prec,
                              float* reads, float* hc, float* out,
                              float* rec_z, float* rec_gates, float* rec_c, float* rec_hc, float* rec_yin,
                              float* rec_ifc, float* rec_u, float* rec_ww, float* rec_rw, float* rec_cw,
                              float* rec_cr, float* rec_al, float* rec_p, float* rec_fwd, float* rec_bwd,
                              float* rec_M, float* rec_L, float* rec_ypre, void* workspace, void* stream) {
    DncMpFwdArgs a;
    size_t lds_bytes = 0;
    NTK_REQUIRE(B > 0 && S > 0 && k > 0, NTK_ERR_BAD_SHAPE, "ntk_dnc_mp_fwd: B=%d S=%d k=%d", B, S, k);
    const int kk = dnc_mp_pick(B, N, W, R, Wn, hid, O, k, a.c, &lds_bytes);
    NTK_REQUIRE(kk == k, NTK_ERR_UNSUPPORTED, "ntk_dnc_mp_fwd: k=%d is not a valid cluster size for B=%d N=%d W=%d R=%d Wn=%d hid=%d "
                "(ask ntk_dnc_mp_plan)", k, B, N, W, R, Wn, hid);
    a.lds = dnc_mp_fwd_lds(a.c);
    a.B = B; a.S = S; a.clip = clip_value;
    NTK_REQUIRE(xproj && Wr && Wi && Wy && mem && link && usage && rw && ww && prec && reads && hc && out && workspace, NTK_ERR_BAD_PTR,
                "ntk_dnc_mp_fwd: null pointer");
    NTK_REQUIRE(ntk_aligned16(xproj) && ntk_aligned16(Wr) && ntk_aligned16(Wi) && ntk_aligned16(mem) && ntk_aligned16(link) &&
                    ntk_aligned16(workspace), NTK_ERR_BAD_PTR, "ntk_dnc_mp_fwd: xproj/Wr/Wi/mem/link/workspace must be 16-byte aligned");
    {
        float* recs[] = {rec_z, rec_gates, rec_c, rec_hc, rec_yin, rec_ifc, rec_u, rec_ww, rec_rw, rec_cw, rec_cr, rec_al,
                         rec_p, rec_fwd, rec_bwd, rec_M, rec_L, rec_ypre};
        int nn = 0;
        for (float* r : recs) nn += (r != nullptr);
        NTK_REQUIRE(nn == 0 || nn == 18, NTK_ERR_BAD_PTR, "ntk_dnc_mp_fwd: record pointers are all-or-none (%d of 18 given)", nn);
        NTK_REQUIRE(nn == 0 || (ntk_aligned16(rec_gates) && ntk_aligned16(rec_M) && ntk_aligned16(rec_L)), NTK_ERR_BAD_PTR,
                    "ntk_dnc_mp_fwd: rec_gates/rec_M/rec_L must be 16-byte aligned");
    }
    a.xcd_local = (B % 8) == 0 ? 1 : 0;
    a.xproj = xproj; a.Wr = Wr; a.Wi = Wi; a.Wy = Wy; a.mem = mem; a.link = link; a.usage = usage; a.rw = rw; a.ww = ww;
    a.prec = prec; a.reads = reads; a.hc = hc; a.out = out;
    a.rec_z = rec_z; a.rec_gates = rec_gates; a.rec_c = rec_c; a.rec_hc = rec_hc; a.rec_yin = rec_yin; a.rec_ifc = rec_ifc;
    a.rec_u = rec_u; a.rec_ww = rec_ww; a.rec_rw = rec_rw; a.rec_cw = rec_cw; a.rec_cr = rec_cr; a.rec_al = rec_al;
    a.rec_p = rec_p; a.rec_fwd = rec_fwd; a.rec_bwd = rec_bwd; a.rec_M = rec_M; a.rec_L = rec_L; a.rec_ypre = rec_ypre;
    const size_t ctrl = dnc_mp_ctrl_bytes(B, k);
    const size_t wsb = dnc_mp_workspace_bytes(B, k, a.c.slot);
    a.flags = reinterpret_cast<unsigned*>(workspace);
    a.err = a.flags + (size_t)B * MPX * k;
    a.xcc = a.err + 1;
    a.mbox = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + ctrl);
    a.sticky = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(workspace) + wsb - 256);
    {
        static NtkLdsAttrCache lds_cache;
        const void* const ks[] = {(const void*)dnc_mp_fwd_kernel<0>, (const void*)dnc_mp_fwd_kernel<1>, (const void*)dnc_mp_fwd_kernel<2>,
                                  (const void*)dnc_mp_fwd_kernel<3>};
        const int rc_lds = ntk_raise_lds_limit(lds_cache, ks, 4, "ntk_dnc_mp_fwd");
        if (rc_lds != NTK_OK) return rc_lds;
    }
    hipError_t e = hipMemsetAsync(workspace, 0, ctrl, (hipStream_t)stream);     // flags + error word: zero before EVERY launch
    NTK_REQUIRE(e == hipSuccess, NTK_ERR_HIP, "ntk_dnc_mp_fwd: hipMemsetAsync: %s", hipGetErrorString(e));
#ifdef NTK_DNC_MP_GENERIC
    const int shape = 0;
#else
    const int shape = dnc_mp_shape_of(a.c);
#endif
    const dim3 grid(B * k);
    switch (shape) {
        case 1: dnc_mp_fwd_kernel<1><<<grid, CT, lds_bytes, (hipStream_t)stream>>>(a); break;
        case 2: dnc_mp_fwd_kernel<2><<<grid, CT, lds_bytes, (hipStream_t)stream>>>(a); break;
        case 3: dnc_mp_fwd_kernel<3><<<grid, CT, lds_bytes, (hipStream_t)stream>>>(a); break;
        default: dnc_mp_fwd_kernel<0><<<grid, CT, lds_bytes, (hipStream_t)stream>>>(a); break;
    }
    NTK_CHECK_LAUNCH("ntk_dnc_mp_fwd");
    return NTK_OK;
}
