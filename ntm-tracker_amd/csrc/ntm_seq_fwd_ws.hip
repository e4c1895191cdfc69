// NTM sequence forward for the benchmark shape (direct_offset_output.py:21-27: memory 128 x 20, hidden 200, 4 read + 1 write
// heads, shift_range 1, output_dim 2) with the recurrent gate product taken OFF the step's critical path.
//
// A step of ntm_seq_fwd_kernel (ntm_seq_fwd.hip) starts with gates = [read_{t-1}; h_{t-1}] . Wr: 280 rows x 800 columns, 51 % of
// the step, bound by the ~100 GB/s one CU draws from L2 (Wr is 909 KB; 281 KB of it fit the CU's registers and LDS).  But only
// the 80 read rows depend on the END of the previous step; the 200 rows that multiply h_{t-1} -- 71 % of the bytes -- have
// their operand as soon as the previous step's LSTM cell is done, seven phases earlier.  So:
//
//   waves 0..7  ("compute", 512 threads): the step as before, except that their gate product covers the READ rows only, all
//               80 of them resident (20 rows per thread in registers, 20 in the 128 KB of LDS the state leaves free): no weight
//               stream between the end of one step and the LSTM cell of the next;
//   waves 8..11 ("stream", 200 of 256 lanes = one LSTM unit each, four gate columns): walk the 200 h rows of Wr in a circle,
//               one lap per step, through a ring of 20 / 25 row registers that is never drained (the weights are the
//               same every step, so the lap's last prefetches are the next lap's first rows); the product for step t + 1
//               starts right after step t's LSTM cell and is handed over (800 floats through LDS) before step t + 1's.
//
// Both kinds of wave run the SAME seven workgroup barriers per step (s_barrier counts every wave of the workgroup); the stream
// waves consume a fixed number of batches between consecutive barriers (template parameters, sized to the compute phases they
// run beside), and their loads stay in flight across the barriers (__syncthreads() waits for LDS traffic only on gfx950).
// One workgroup per sequence, one CU per workgroup, no communication between workgroups: the summation ORDER of a gate
// differs from ntm_seq_fwd_kernel's (read rows in two slices + h rows in one chain), the arithmetic does not.
// Reference: ntm_cell.py:45-50, :101-105 (controller), ops.py / ntm_cell.py lines as cited in ntm_seq_fwd.hip.
#include "ntm_fwd_args.h"
#include <stdlib.h>
#include <type_traits>

// Diagnostic build only (-DNTK_CL_PROF): s_memtime shares of workgroup 0, one set for compute thread 0 and one for the first stream
// lane: entry 2 i = work before barrier B(i+1), entry 2 i + 1 = the wait at that barrier
#ifdef NTK_CL_PROF
__device__ unsigned long long g_ntm_ws_prof[2][16];
#define WS_STAMP(role, i)                                                   \
    do {                                                                    \
        if (prof_on) {                                                      \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
            s_prof[(role) * 16 + (i)] += now_ - s_prof[(role) * 16 + 15];   \
            s_prof[(role) * 16 + 15] = now_;                                \
        }                                                                   \
    } while (0)
extern "C" int ntk_ntm_ws_prof(unsigned long long* out32) {
    return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_ntm_ws_prof), 32 * sizeof(unsigned long long)) == hipSuccess ? NTK_OK : NTK_ERR_HIP;
}
#else
#define WS_STAMP(role, i) do { } while (0)
#endif
#define WS_BARRIER(role, i) do { WS_STAMP(role, 2 * (i)); __syncthreads(); WS_STAMP(role, 2 * (i) + 1); } while (0)

namespace {

constexpr int WN = 128, WMd = 20, WMP = 21, WR = 4, WH = 5, Whid = 200, WSS = 3, WRM = 80, WK = 280, WO = 2;
constexpr int WoB = 100, WoG = 105, WoS = 110, WoY = 125, WoE = 130, WoA = 150, WP = 170, WPP = 172, Wldz = 284, Wldh = 204;
constexpr int TC = 512, TA = 768;          // compute threads, all threads
constexpr int RESQ = 20, RESL = 20;        // read rows per compute thread resident in registers / in LDS (2 slices x 40 rows)
constexpr int NG = Whid / 4;               // groups of four h rows per lap (50)

template <int N> using wsic = std::integral_constant<int, N>;
template <int I0, int I1, class F>
__device__ __forceinline__ void ws_for(F&& f) {
    if constexpr (I0 < I1) { f(wsic<I0>{}); ws_for<I0 + 1, I1>(f); }
}

// P3 .. P7: groups (4 rows each) the stream waves consume after barriers B2 .. B6 of a step; the other 50 - sum groups are
// consumed at the top of the next step, before its barrier B1 (that is where the product is handed over).  RING: rows in flight.
template <int RING, int P3, int P4, int P5, int P6, int P7>
__global__ __launch_bounds__(TA) void ntm_seq_fwd_ws_kernel(NtmFwdArgs a, NtmLds L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PFIN = P3 + P4 + P5 + P6 + P7;
    static_assert(PFIN >= 0 && PFIN <= NG && Whid % RING == 0, "groups per lap; the ring's phase is static when it divides the lap");
    const int b = blockIdx.x, tid0 = threadIdx.x;
    const int S = a.d.S;
    constexpr int N = WN, Md = WMd, MP = WMP, R = WR, H = WH, hid = Whid, SS = WSS, RM = WRM, K = WK, PP = WPP;

    float* sPart = smem + L.part;
    float* sM = smem + L.M;
    float* sW = smem + L.W;
    float* sWg = smem + L.Wg;
    float* sZ = smem + L.Z;
    float* sC = smem + L.C;
    float* sU = smem + L.U;
    float* sKs = smem + L.Ks;
    float* sCn = smem + L.Cn;
    float* sPw = smem + L.Pw;
    f32x4* sWres4 = reinterpret_cast<f32x4*>(smem + L.total + 32);                           // [RESL][400]
    f32x4* sPartH4 = reinterpret_cast<f32x4*>(smem + L.total + 32 + RESL * 400 * 4);        // [200]: h rows . Wr for the coming step
    f32x4* sPart4 = reinterpret_cast<f32x4*>(sPart);
    const f32x4* Wr4 = reinterpret_cast<const f32x4*>(a.Wr);
    const f32x4* Wa4 = reinterpret_cast<const f32x4*>(a.Wa);

    // ---- initial state (all twelve waves)
    for (int i = tid0; i < N * Md; i += TA) sM[(i / Md) * MP + (i % Md)] = a.M0[(size_t)b * N * Md + i];
    for (int i = tid0; i < H * N; i += TA) sW[i] = a.w0[(size_t)b * H * N + i];
    for (int i = tid0; i < RM; i += TA) sZ[i] = a.read0[(size_t)b * RM + i];
    for (int i = tid0; i < hid; i += TA) {
        sC[i] = a.cs0[(size_t)b * 2 * hid + i];
        sZ[RM + i] = a.cs0[(size_t)b * 2 * hid + hid + i];
    }
#ifdef NTK_CL_PROF
    unsigned long long* s_prof = reinterpret_cast<unsigned long long*>(smem + L.total + 32 + RESL * 400 * 4 + 4 * hid);
    const bool prof_on = blockIdx.x == 0 && (tid0 == 0 || tid0 == TC);
    if (prof_on) { const int ro = tid0 == 0 ? 0 : 16; for (int i = 0; i < 15; ++i) s_prof[ro + i] = 0; }
#endif
    __syncthreads();
#ifdef NTK_CL_PROF
    if (prof_on) s_prof[(tid0 == 0 ? 0 : 16) + 15] = __builtin_amdgcn_s_memtime();
#endif

    if (tid0 >= TC) {
        // =========================================================== stream waves
        const int j = min(tid0 - TC, hid - 1);                       // LSTM unit (lanes 200..255 of the group shadow unit 199)
        // row r of the h part through a buffer resource: lane offset 16 j in a VGPR, the row's offset 3 200 r in the scalar
        // operand -- 200 per-row 64-bit addresses would otherwise be hoisted out of the t-loop and spilled
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Wr + (size_t)RM * 4 * hid), 0,
                                                                              hid * 4 * hid * (int)sizeof(float), 0x00020000);
        const unsigned voff = (unsigned)j * 16u;
        auto wrow = [&](int r) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, voff, r * (4 * hid * (int)sizeof(float)), 0)); };
        const float* sHp = sZ + RM;                                  // h_{t-1} (wave-uniform reads)
        // RING rows are in flight at all times: row r of the lap lives in ring[r % RING] and is replaced, as soon as it has been
        // multiplied, by row r + RING (mod the lap: the weights are the same every step)
        f32x4 ring[RING];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // the product for step 0 (h_{-1} = the initial controller state): every group but those the loop's first trip adds, in a
        // plain rolled loop (once per launch), then the ring as the loop expects it: rows 4 PFIN .. 4 PFIN + RING - 1 in flight
#pragma unroll 1
        for (int r0 = 0; r0 < 4 * PFIN; r0 += 4) {
            f32x4 w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = wrow(r0 + q);
            const f32x4 hv = *reinterpret_cast<const f32x4*>(sHp + r0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += hv[q] * w[q];
        }
#pragma unroll
        for (int q = 0; q < RING; ++q) ring[(4 * PFIN + q) % RING] = wrow((4 * PFIN + q) % hid);
        auto group = [&](auto gc) {                                  // rows 4 g .. 4 g + 3
            constexpr int r = 4 * decltype(gc)::value;
            const f32x4 hv = *reinterpret_cast<const f32x4*>(sHp + r);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc += hv[q] * ring[(r + q) % RING];
                ring[(r + q) % RING] = wrow((r + q + RING) % hid);
            }
            // keep the groups in program order: left alone the scheduler clusters the loads of many groups, their destination
            // registers overlap live ring rows, and the spills that follow (scratch traffic shares vmcnt with the ring: every
            // reload drains it) cost more than the whole stream
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int t = 0; t < S; ++t) {
            ws_for<PFIN, NG>(group);
            // every lane stores (lanes 200..255 shadow unit 199: same address, same value): under a lane condition the compiler
            // sinks the whole product of this segment into the branch and parks the ring's refills in scratch, one drained load each
            sPartH4[j] = acc;
            acc = f32x4{0.f, 0.f, 0.f, 0.f};
            WS_BARRIER(1, 0);                                        // B1: the compute waves' read-row partials and this are complete
            WS_BARRIER(1, 1);                                        // B2: h_t is in LDS
            ws_for<0, P3>(group);
            WS_BARRIER(1, 2);                                        // B3
            ws_for<P3, P3 + P4>(group);
            WS_BARRIER(1, 3);                                        // B4
            ws_for<P3 + P4, P3 + P4 + P5>(group);
            WS_BARRIER(1, 4);                                        // B5
            ws_for<P3 + P4 + P5, P3 + P4 + P5 + P6>(group);
            WS_BARRIER(1, 5);                                        // B6
            ws_for<P3 + P4 + P5 + P6, PFIN>(group);
            WS_BARRIER(1, 6);                                        // B7
        }
#ifdef NTK_CL_PROF
        if (prof_on) for (int i = 0; i < 16; ++i) g_ntm_ws_prof[1][i] = s_prof[16 + i];
#endif
        return;
    }

    // =============================================================== compute waves (512 threads)
    constexpr int T = TC;
    constexpr int ncg = PP >> 2;                                     // 43 float4 column groups of the unpack product
    constexpr int nslB = 11, kperB = (hid + nslB - 1) / nslB;        // T / ncg slices of the hidden units
    constexpr int nslR = 6, nperR = (N + nslR - 1) / nslR;           // T / RM slices of the slots in the read product
    static_assert(nslB * ncg <= T && nslR * RM <= T, "decomposition");
    f32x4 wres[RESQ];
    if (tid0 < 2 * hid) {
        const int j = tid0 % hid, k0 = (tid0 / hid) * (RESQ + RESL);
#pragma unroll
        for (int q = 0; q < RESQ; ++q) wres[q] = Wr4[(size_t)(k0 + q) * hid + j];
        for (int q = 0; q < RESL; ++q) sWres4[q * (2 * hid) + tid0] = Wr4[(size_t)(k0 + RESQ + q) * hid + j];
    }
    for (int t = 0; t < S; ++t) {
        // an opaque copy of the thread id keeps the compiler from hoisting dozens of loop-invariant addresses out of the
        // t-loop and spilling them (ntm_seq_fwd.hip)
        int tid_op = tid0;
        asm volatile("" : "+v"(tid_op));
        const int tid = tid_op;
        const int lane = tid & 63, wave = tid >> 6;
        const size_t bt = (size_t)b * S + t;
        // ------------------------------------------------------------ P1: the READ rows of the gate product, all resident
        f32x4 xg = {0.f, 0.f, 0.f, 0.f};
        if (tid < hid) {   // this step's input projection + LSTM bias (row K of Wr)
            xg = reinterpret_cast<const f32x4*>(a.xproj)[bt * hid + tid];
            xg += Wr4[(size_t)K * hid + tid];
        }
        if (a.st_z) {
            for (int i = tid; i < Wldz; i += T) a.st_z[bt * Wldz + i] = (i < K) ? sZ[i] : (i == K ? 1.f : 0.f);
        }
        if (tid < 2 * hid) {
            const int k0 = (tid / hid) * (RESQ + RESL);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < RESQ; ++q) acc += sZ[k0 + q] * wres[q];
#pragma unroll 4
            for (int q = 0; q < RESL; ++q) acc += sZ[k0 + RESQ + q] * sWres4[q * (2 * hid) + tid];
            sPart4[tid] = acc;
        }
        WS_BARRIER(0, 0);                                            // B1
        // ------------------------------------------------------------ P2: LSTM cell  ||  column norms of M (Q1)
        if (tid < hid) {
            const f32x4 g = ((xg + sPart4[tid]) + sPart4[hid + tid]) + sPartH4[tid];
            const float gi = ntm_sigmoid(g[0]);
            const float gj = ntm_tanh(g[1]);
            const float gf = ntm_sigmoid(g[2]);      // forget_bias = 0.0 (ntm_cell.py:47)
            const float go = ntm_sigmoid(g[3]);
            const float c2 = sC[tid] * gf + gi * gj;
            const float h2 = ntm_tanh(c2) * go;
            sC[tid] = c2;
            sZ[RM + tid] = h2;
            if (a.st_gates) {
                f32x4 ga = {gi, gj, gf, go};
                reinterpret_cast<f32x4*>(a.st_gates)[bt * hid + tid] = ga;
                a.st_c[bt * hid + tid] = c2;
            }
            if (a.st_h) a.st_h[bt * Wldh + tid] = h2;
        } else if (a.st_h && tid < Wldh) {
            a.st_h[bt * Wldh + tid] = (tid == hid) ? 1.f : 0.f;
        }
        {   // waves 4..7 normalise the feature columns over the slot axis (tf.nn.l2_normalize, ops.py:150)
            constexpr int w0 = (hid + 63) >> 6, nwaves = T >> 6;
            if (wave >= w0) {
                for (int m = wave - w0; m < Md; m += nwaves - w0) {
                    float s = 0.f;
                    for (int n = lane; n < N; n += 64) { const float v = sM[n * MP + m]; s += v * v; }
                    s = wave_sum(s);
                    if (lane == 0) sCn[m] = 1.0f / sqrtf(fmaxf(s, 1e-12f));
                }
            }
        }
        WS_BARRIER(0, 1);                                            // B2
        // ------------------------------------------------------------ P3: unpack / output partials (Wa streams: 137 KB)
        if (tid < nslB * ncg) {
            const int cg = tid % ncg, ks = tid / ncg;
            const int k0 = ks * kperB, k1 = min(hid, k0 + kperB);
            sPart4[ks * ncg + cg] = ntk_stream_matvec<4>(Wa4 + cg, ncg, sZ + RM, k0, k1, hid);
        }
        WS_BARRIER(0, 2);                                            // B3
        // ------------------------------------------------------------ P4: control activations
        if (tid < PP) {
            float v = a.Wa[(size_t)hid * PP + tid];
            for (int ks = 0; ks < nslB; ++ks) v += sPart[ks * PP + tid];
            float r = v;
            if (tid < WoB) r = ntm_tanh(v);                       // k      :133
            else if (tid < WoG) r = ntm_softplus(v);              // beta   :140
            else if (tid < WoS) r = ntm_sigmoid(v);               // g      :151
            else if (tid < WoY) r = v;                            // shift logits (softmax per head below)
            else if (tid < WoE) r = ntm_softplus(v) + 1.0f;       // gamma  :169-170
            else if (tid < WoA) r = ntm_sigmoid(v);               // erase  :193
            else if (tid < WP) r = ntm_tanh(v);                   // add    :195
            sU[tid] = r;
            if (a.st_u) a.st_u[bt * PP + tid] = r;
            if (tid >= WP && tid < WP + WO) a.logits[bt * WO + (tid - WP)] = v;
        }
        WS_BARRIER(0, 3);                                            // B4
        // ------------------------------------------------------------ P5-P7: one WAVE per head, no workgroup barrier inside
        if (wave < H) {
            const int h = wave;
            float kss = 0.f;
            for (int m = 0; m < Md; ++m) { const float kv = sU[h * Md + m]; kss += kv * kv; }
            const float kinv = 1.0f / sqrtf(fmaxf(kss, 1e-12f));
            if (lane < Md) sKs[h * Md + lane] = sU[h * Md + lane] * kinv * sCn[lane];
            const float beta = sU[WoB + h], g = sU[WoG + h], gamma = sU[WoY + h];
            float swv[SS];                                         // softmax of the shift logits (ntm_cell.py:161)
            {
                float mx = -INFINITY;
#pragma unroll
                for (int jj = 0; jj < SS; ++jj) mx = fmaxf(mx, sU[WoS + h * SS + jj]);
                float sum = 0.f;
#pragma unroll
                for (int jj = 0; jj < SS; ++jj) { swv[jj] = ntm_exp(sU[WoS + h * SS + jj] - mx); sum += swv[jj]; }
#pragma unroll
                for (int jj = 0; jj < SS; ++jj) swv[jj] = swv[jj] / sum;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float mxv = -INFINITY;
            for (int n = lane; n < N; n += 64) {
                float sim = 0.f;
                for (int m = 0; m < Md; ++m) sim += sKs[h * Md + m] * sM[n * MP + m];
                const float v = sim * beta;
                sWg[h * N + n] = v;
                mxv = fmaxf(mxv, v);
            }
            mxv = wave_max(mxv);
            float sum = 0.f;
            for (int n = lane; n < N; n += 64) { const float e = ntm_exp(sWg[h * N + n] - mxv); sWg[h * N + n] = e; sum += e; }
            sum = wave_sum(sum);
            for (int n = lane; n < N; n += 64) {
                const float wc = sWg[h * N + n] / sum;
                if (a.st_wc) a.st_wc[(bt * H + h) * N + n] = wc;
                sWg[h * N + n] = wc * g + sW[h * N + n] * (1.0f - g);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            constexpr int start = -((SS + 1) >> 1);                // Py2 floor of -SS/2 (Q2): 3 -> -2
            float psum = 0.f;
            for (int n = lane; n < N; n += 64) {
                float wv = 0.f;
#pragma unroll
                for (int jj = 0; jj < SS; ++jj) wv += swv[jj] * sWg[h * N + ((n + start + jj + N) & (N - 1))];
                if (a.st_wv) a.st_wv[(bt * H + h) * N + n] = wv;
                const float pw = ntm_pow(wv, gamma);
                sPw[h * N + n] = pw;
                psum += pw;
            }
            psum = wave_sum(psum);
            for (int n = lane; n < N; n += 64) {
                const float w = sPw[h * N + n] / (psum + 1e-3f);
                sW[h * N + n] = w;
                if (a.st_w) a.st_w[(bt * H + h) * N + n] = w;
            }
        } else if (wave == H && lane == 0 && a.outputs) {
            const float l0 = sU[WP], l1 = sU[WP + 1], mx = fmaxf(l0, l1);
            const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
            a.outputs[bt * WO] = e0 / (e0 + e1);
            a.outputs[bt * WO + 1] = e1 / (e0 + e1);
        }
        WS_BARRIER(0, 4);                                            // B5
        // ------------------------------------------------------------ P8: read (of the PRE-write memory, Q6), then write
        if (tid < nslR * RM) {
            const int o = tid % RM, sl = tid / RM;
            const int i = o / Md, m = o - i * Md;
            const int n0 = sl * nperR, n1 = min(N, n0 + nperR);
            float s0 = 0.f, s1 = 0.f;                 // two chains: the loop is bound by the add latency, not by LDS
            int n = n0;
            for (; n + 1 < n1; n += 2) {
                s0 += sW[i * N + n] * sM[n * MP + m];
                s1 += sW[i * N + n + 1] * sM[(n + 1) * MP + m];
            }
            if (n < n1) s0 += sW[i * N + n] * sM[n * MP + m];
            sPart[sl * RM + o] = s0 + s1;
        }
        WS_BARRIER(0, 5);                                            // B6
        for (int idx = tid; idx < N * Md; idx += T) {
            const int n = idx / Md, m = idx - n * Md;
            const float ww = sW[R * N + n];
            const float nm = sM[n * MP + m] * (1.0f - ww * sU[WoE + m]) + ww * sU[WoA + m];
            sM[n * MP + m] = nm;
            if (a.st_M) a.st_M[bt * N * Md + idx] = nm;
        }
        if (tid < RM) {
            float s = 0.f;
#pragma unroll
            for (int sl = 0; sl < nslR; ++sl) s += sPart[sl * RM + tid];
            sZ[tid] = s;
            if (a.st_read) a.st_read[bt * RM + tid] = s;
        }
        WS_BARRIER(0, 6);                                            // B7
    }

#ifdef NTK_CL_PROF
    if (prof_on) for (int i = 0; i < 16; ++i) g_ntm_ws_prof[0][i] = s_prof[i];
#endif
    // ---- final state
    for (int i = tid0; i < N * Md; i += T) a.M_out[(size_t)b * N * Md + i] = sM[(i / Md) * MP + (i % Md)];
    for (int i = tid0; i < H * N; i += T) a.w_out[(size_t)b * H * N + i] = sW[i];
    for (int i = tid0; i < RM; i += T) a.read_out[(size_t)b * RM + i] = sZ[i];
    for (int i = tid0; i < hid; i += T) {
        a.cs_out[(size_t)b * 2 * hid + i] = sC[i];
        a.cs_out[(size_t)b * 2 * hid + hid + i] = sZ[RM + i];
    }
}

}  // namespace

bool ntm_seq_fwd_ws_takes(const NtmDims& d) {
    return d.N == WN && d.Md == WMd && d.R == WR && d.Wh == 1 && d.hid == Whid && d.SS == WSS && d.O == WO && !d.write_first;
}

// the launch of the benchmark-shape kernel (ntm_seq_fwd.hip validates the arguments and dispatches here)
int ntm_seq_fwd_ws_launch(const NtmFwdArgs& a, void* stream) {
    NtmLds L;
    ntm_fwd_lds(a.d, TC, L);
    size_t lds_bytes = (size_t)L.total * sizeof(float) + 128 + (size_t)RESL * 400 * sizeof(f32x4) + (size_t)Whid * sizeof(f32x4);
#ifdef NTK_CL_PROF
    lds_bytes += 256;
#endif
    NTK_REQUIRE(lds_bytes <= 160 * 1024, NTK_ERR_UNSUPPORTED, "ntk_ntm_seq_fwd: state needs %zu B of LDS (> 160 KiB)", lds_bytes);
    // stream batches beside: the unpack product (its own 137 KB stream has the pipe), the activations, the addressing, the read,
    // the write; the other five at the top of the next step (beside the read rows of its gate product)
    // (NTK_NTM_WS_SPLIT=n: development variants of the split and of the ring depth, same results)
    const char* split_env = getenv("NTK_NTM_WS_SPLIT");              // read per launch: a development sweep switches it in one process
    const int split = split_env ? atoi(split_env) : 0;
    typedef void (*kern_t)(NtmFwdArgs, NtmLds);
    static const kern_t ks[] = {ntm_seq_fwd_ws_kernel<20, 5, 5, 23, 7, 7>, ntm_seq_fwd_ws_kernel<20, 5, 5, 21, 6, 7>,
                                ntm_seq_fwd_ws_kernel<20, 4, 5, 20, 6, 6>, ntm_seq_fwd_ws_kernel<25, 5, 5, 23, 7, 7>,
                                ntm_seq_fwd_ws_kernel<25, 5, 5, 21, 6, 7>, ntm_seq_fwd_ws_kernel<25, 4, 5, 20, 6, 6>,
                                ntm_seq_fwd_ws_kernel<10, 5, 5, 23, 7, 7>, ntm_seq_fwd_ws_kernel<20, 6, 6, 24, 7, 7>,
                                ntm_seq_fwd_ws_kernel<25, 6, 6, 24, 7, 7>, ntm_seq_fwd_ws_kernel<20, 3, 4, 18, 5, 5>};
    constexpr int NK = (int)(sizeof(ks) / sizeof(ks[0]));
    static NtkLdsAttrCache lds_cache;
    const int rc = ntk_raise_lds_limit(lds_cache, reinterpret_cast<const void* const*>(ks), NK, "ntk_ntm_seq_fwd");
    if (rc != NTK_OK) return rc;
    ks[(split >= 0 && split < NK) ? split : 0]<<<a.d.B, TA, lds_bytes, (hipStream_t)stream>>>(a, L);
    NTK_CHECK_LAUNCH("ntk_ntm_seq_fwd(ws)");
    return NTK_OK;
}
