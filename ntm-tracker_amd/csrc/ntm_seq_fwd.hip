// NTM sequence forward: one persistent workgroup per sequence walks all S
// steps of NTMCell.__call__ (ntm_cell.py:53-253) with the whole recurrent state
// (memory N x Md, head weights H x N, reads, LSTM c/h) resident in LDS; only
// the weights are streamed (from L2) and the per-step tensors the BPTT pass
// needs are written once, coalesced, to HBM.  Replaces the tf.while_loop of
// LoopNTMTracker (ntm_tracker_new.py:13-64), S sequential TF graph iterations.
//
// Per step (all phases separated by workgroup barriers):
//   P1 gate partials      z=[read_prev;h_prev] (K) x Wr[K][4*hid]   (K-sliced over thread groups)
//   P2 LSTM cell          BasicLSTMCell, gate order i,j,f,o, forget_bias 0 (ntm_cell.py:45-50)
//                         || the other waves l2-normalise the feature columns of M over the slots (quirk Q1)
//   P3 unpack partials    h' x Wa[hid][PP]                           (ntm_cell.py:124-126, :220)
//   P4 control activations tanh/softplus/sigmoid/1+softplus           (:133,140,151,169,193,195)
//   P5-P7 ONE WAVE PER HEAD, no workgroup barrier: key scaling, shift softmax (ops.py:150-152, ntm_cell.py:161),
//      similarity (Q1: feature columns normalised over slots), beta, softmax over N, gate (:136-156),
//      circular shift with taps -(r+1)..r-1 (Q2), sharpen with +1e-3 (Q4)   (ops.py:204-213, ntm_cell.py:173-176)
//   P8 erase/add write and read (reads see the pre-write memory unless write_first, Q6) (:202-215)
#include "ntm_common.h"

// Diagnostic build only (-DNTK_CL_PROF): s_memtime shares per phase, accumulated in LDS by thread 0 of workgroup 0
#ifdef NTK_CL_PROF
__device__ unsigned long long g_ntm_fwd_prof[16];
#define NTM_STAMP(i)                                                        \
    do {                                                                    \
        if (blockIdx.x == 0 && tid == 0) {                                  \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
            s_prof[i] += now_ - s_prof[15];                                 \
            s_prof[15] = now_;                                              \
        }                                                                   \
    } while (0)
extern "C" int ntk_ntm_fwd_prof(unsigned long long* out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_ntm_fwd_prof), 16 * sizeof(unsigned long long)) == hipSuccess ? NTK_OK : NTK_ERR_HIP;
}
#else
#define NTM_STAMP(i) do { } while (0)
#endif

#include "ntm_fwd_args.h"
#include <stdlib.h>

// FIX = true specialises every dimension to the reference defaults the benchmark configs run
// (direct_offset_output.py:21-27: mem 128x20, hidden 200, 4 read + 1 write heads, shift_range 1,
// output_dim 2, 640 threads): index arithmetic constant-folds and the small loops unroll.
// FIXT = 512 with NTM_RES_REG / NTM_RES_LDS > 0 is the same specialisation with RESIDENT gate weights: 512 threads leave a
// thread 256 registers, and 44 of the 140 rows of Wr a thread multiplies per step (one float4 gate column of its unit x half
// of K) stay on the CU for the whole sequence -- 24 rows in registers, 20 in the 128 KB of LDS the state leaves free --
// instead of streaming from L2 every step.  The stream of the remaining rows is requested first and lands while the
// resident rows are multiplied; the products are summed in the same order as before.  Measured (scripts/dev_ntm_prof.py,
// dev_ntm_timing.py): the gate stream is 51 % of a forward step and runs at the ~100 GB/s one CU draws from L2 however
// deep the prefetch (16 or 24 rows per thread in flight: the same) and however lean the loop (a guard-free form with
// scalar row bases: the same) -- only fewer bytes help: 912 -> 614 KB per step, forward 21.6 -> 20.3 ms at B32 x S1300.
// LDS rows alone at 640 threads gave nothing (the other phases lose at 512 threads what 14 % fewer bytes win).
template <int MAXT, int FIXT, int NTM_RES_REG = 0, int NTM_RES_LDS = 0>
__global__ __launch_bounds__(MAXT) void ntm_seq_fwd_kernel(NtmFwdArgs a, NtmLds L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr bool FIX = FIXT != 0, RES = NTM_RES_REG + NTM_RES_LDS > 0;
    const int b = blockIdx.x, tid = threadIdx.x, T = FIX ? FIXT : blockDim.x;
    const int N = FIX ? 128 : a.d.N, Md = FIX ? 20 : a.d.Md, MP = Md | 1, R = FIX ? 4 : a.d.R, Wh = FIX ? 1 : a.d.Wh;
    const int H = R + Wh, hid = FIX ? 200 : a.d.hid, SS = FIX ? 3 : a.d.SS;
    const int S = a.d.S, RM = R * Md, K = RM + hid;
    struct {
        int O, oK, oB, oG, oS, oY, oE, oA, P, PP, ldz, ldh, write_first;
    } d;
    d.O = FIX ? 2 : a.d.O;
    d.oK = 0; d.oB = H * Md; d.oG = d.oB + H; d.oS = d.oG + H; d.oY = d.oS + H * SS; d.oE = d.oY + H;
    d.oA = d.oE + Wh * Md; d.P = d.oA + Wh * Md;
    d.PP = (d.P + d.O + 3) & ~3; d.ldz = (K + 1 + 3) & ~3; d.ldh = (hid + 1 + 3) & ~3;
    d.write_first = a.d.write_first;
    const int PP = d.PP;

    float* sPart = smem + L.part;
    float* sM = smem + L.M;
    float* sW = smem + L.W;
    float* sWg = smem + L.Wg;
    float* sZ = smem + L.Z;
    float* sC = smem + L.C;
    float* sU = smem + L.U;
    float* sKs = smem + L.Ks;
    float* sCn = smem + L.Cn;
    float* sPw = smem + L.Pw;              // [H][N] sharpened weights before normalisation

    // work decomposition (uniform per kernel)
    const int nsl = max(1, T / hid);                 // K-slices of the gate product
    const int kper = (K + nsl - 1) / nsl;
    const int ncg = PP >> 2;                          // float4 column groups of the unpack product
    const int nslB = min(max(1, T / ncg), hid);
    const int kperB = (hid + nslB - 1) / nslB;
    const int nslR = min(max(1, T / RM), N);          // N-slices of the read product
    const int nperR = (N + nslR - 1) / nslR;

    // ---- load the initial state
    for (int i = tid; i < N * Md; i += T) sM[(i / Md) * MP + (i % Md)] = a.M0[(size_t)b * N * Md + i];
    for (int i = tid; i < H * N; i += T) sW[i] = a.w0[(size_t)b * H * N + i];
    for (int i = tid; i < RM; i += T) sZ[i] = a.read0[(size_t)b * RM + i];
    for (int i = tid; i < hid; i += T) {
        sC[i] = a.cs0[(size_t)b * 2 * hid + i];
        sZ[RM + i] = a.cs0[(size_t)b * 2 * hid + hid + i];
    }
    __syncthreads();

    const f32x4* Wr4 = reinterpret_cast<const f32x4*>(a.Wr);
    const f32x4* Wa4 = reinterpret_cast<const f32x4*>(a.Wa);
    f32x4* sPart4 = reinterpret_cast<f32x4*>(sPart);

    const int tid0 = tid;
    f32x4 wres[NTM_RES_REG > 0 ? NTM_RES_REG : 1];
    const f32x4* sWres4 = reinterpret_cast<const f32x4*>(smem + L.total + 32);     // [NTM_RES_LDS][nsl * hid] behind the state (+ the diagnostic words)
    if constexpr (RES) {
        if (tid < nsl * hid) {
            const int j = tid % hid, ks = tid / hid, k0 = ks * kper;
#pragma unroll
            for (int q = 0; q < NTM_RES_REG; ++q) wres[q] = Wr4[(size_t)min(k0 + q, a.d.ldz - 1) * hid + j];
            f32x4* wl = reinterpret_cast<f32x4*>(smem + L.total + 32);
            for (int q = 0; q < NTM_RES_LDS; ++q) wl[q * (nsl * hid) + tid] = Wr4[(size_t)min(k0 + NTM_RES_REG + q, a.d.ldz - 1) * hid + j];
        }
        __syncthreads();
    }
#ifdef NTK_CL_PROF
    unsigned long long* s_prof = reinterpret_cast<unsigned long long*>(smem + L.total);      // 128 B behind the state (the launch adds them)
    if (tid == 0) { for (int i = 0; i < 15; ++i) s_prof[i] = 0; s_prof[15] = __builtin_amdgcn_s_memtime(); }
#endif
    for (int t = 0; t < S; ++t) {
        // re-derive every thread-index expression inside the step: an opaque copy of the thread id keeps
        // the compiler from hoisting dozens of loop-invariant addresses out of the t-loop and spilling them
        int tid_op = tid0;
        asm volatile("" : "+v"(tid_op));
        const int tid = tid_op;
        const int lane = tid & 63;
        const size_t bt = (size_t)b * S + t;
        // ------------------------------------------------------------ P1
        f32x4 xg = {0.f, 0.f, 0.f, 0.f};
        if (tid < hid) {   // prefetch this step's input projection + LSTM bias (row K of Wr)
            xg = reinterpret_cast<const f32x4*>(a.xproj)[bt * hid + tid];
            const f32x4 bb = Wr4[(size_t)K * hid + tid];
            xg += bb;
        }
        if (a.st_z) {
            for (int i = tid; i < d.ldz; i += T)
                a.st_z[bt * d.ldz + i] = (i < K) ? sZ[i] : (i == K ? 1.f : 0.f);
        }
        if (tid < nsl * hid) {
            const int j = tid % hid, ks = tid / hid;
            const int k0 = ks * kper, k1 = min(K, k0 + kper);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const f32x4* wp = Wr4 + j;
            // rolling prefetch: two register batches of 8 rows keep >= 8 x 16 B per thread in flight from L2
            constexpr int PF = 8;
            f32x4 wa[PF], wb[PF];
            const int klast = a.d.ldz - 1;          // last row of the Wr allocation (a zero pad row)
            const int ks0 = RES ? min(k1, k0 + NTM_RES_REG + NTM_RES_LDS) : k0;      // first streamed row
#pragma unroll
            for (int q = 0; q < PF; ++q) wa[q] = wp[(size_t)min(ks0 + q, klast) * hid];
#pragma unroll
            for (int q = 0; q < PF; ++q) wb[q] = wp[(size_t)min(ks0 + PF + q, klast) * hid];
            if constexpr (RES) {                    // the resident rows, while the first streamed batches are in flight
#pragma unroll
                for (int q = 0; q < NTM_RES_REG; ++q) acc += ((k0 + q < k1) ? sZ[k0 + q] : 0.f) * wres[q];
#pragma unroll 4
                for (int q = 0; q < NTM_RES_LDS; ++q)
                    acc += ((k0 + NTM_RES_REG + q < k1) ? sZ[k0 + NTM_RES_REG + q] : 0.f) * sWres4[q * (nsl * hid) + tid];
            }
#pragma nounroll
            for (int k = ks0; k < k1; k += 2 * PF) {
#pragma unroll
                for (int q = 0; q < PF; ++q) {
                    const float zk = (k + q < k1) ? sZ[k + q] : 0.f;
                    acc += zk * wa[q];
                }
#pragma unroll
                for (int q = 0; q < PF; ++q) wa[q] = wp[(size_t)min(k + 2 * PF + q, klast) * hid];
#pragma unroll
                for (int q = 0; q < PF; ++q) {
                    const float zk = (k + PF + q < k1) ? sZ[k + PF + q] : 0.f;
                    acc += zk * wb[q];
                }
#pragma unroll
                for (int q = 0; q < PF; ++q) wb[q] = wp[(size_t)min(k + 3 * PF + q, klast) * hid];
            }
            sPart4[ks * hid + j] = acc;
        }
        __syncthreads();
        NTM_STAMP(0);
        // ------------------------------------------------------------ P2: LSTM cell  ||  column norms of M (Q1)
        const int wave = tid >> 6, nwaves = T >> 6;
        if (tid < hid) {
            f32x4 g = xg;
            for (int ks = 0; ks < nsl; ++ks) g += sPart4[ks * hid + tid];
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
            if (a.st_h) a.st_h[bt * d.ldh + tid] = h2;
        } else if (a.st_h && tid < d.ldh) {
            a.st_h[bt * d.ldh + tid] = (tid == hid) ? 1.f : 0.f;
        }
        {   // waves not running the LSTM normalise the feature columns over the slot axis (tf.nn.l2_normalize, ops.py:150)
            const int w0 = (hid + 63) >> 6;
            if (wave >= w0) {
                for (int m = wave - w0; m < Md; m += nwaves - w0) {
                    float s = 0.f;
                    for (int n = lane; n < N; n += 64) { const float v = sM[n * MP + m]; s += v * v; }
                    s = wave_sum(s);
                    if (lane == 0) sCn[m] = 1.0f / sqrtf(fmaxf(s, 1e-12f));
                }
            }
        }
        __syncthreads();
        NTM_STAMP(1);
        // ------------------------------------------------------------ P3: unpack / output partials
        if (tid < nslB * ncg) {
            const int cg = tid % ncg, ks = tid / ncg;
            const int k0 = ks * kperB, k1 = min(hid, k0 + kperB);
            // explicit two-batch stream (the compiler otherwise keeps ONE load in flight, see common.h)
            sPart4[ks * ncg + cg] = ntk_stream_matvec<(MAXT > 768 ? 2 : 4)>(Wa4 + cg, ncg, sZ + RM, k0, k1, hid);
        }
        __syncthreads();
        NTM_STAMP(2);
        // ------------------------------------------------------------ P4: control activations
        if (tid < PP) {
            float v = a.Wa[(size_t)hid * PP + tid];
            for (int ks = 0; ks < nslB; ++ks) v += sPart[ks * PP + tid];
            float r = v;
            if (tid < d.oB) r = ntm_tanh(v);                       // k      :133
            else if (tid < d.oG) r = ntm_softplus(v);              // beta   :140
            else if (tid < d.oS) r = ntm_sigmoid(v);               // g      :151
            else if (tid < d.oY) r = v;                            // shift logits (softmax per head below)
            else if (tid < d.oE) r = ntm_softplus(v) + 1.0f;       // gamma  :169-170
            else if (tid < d.oA) r = ntm_sigmoid(v);               // erase  :193
            else if (tid < d.P) r = ntm_tanh(v);                   // add    :195
            sU[tid] = r;
            if (a.st_u) a.st_u[bt * PP + tid] = r;
            if (tid >= d.P && tid < d.P + d.O) a.logits[bt * d.O + (tid - d.P)] = v;
        }
        __syncthreads();
        NTM_STAMP(3);
        // ------------------------------------------------------------ P5-P7: one WAVE per head, no workgroup barrier inside:
        // key scaling, similarity (Q1), beta, softmax over N, gate, circular shift (Q2), sharpen (Q4)
        if (wave < H) {
            const int h = wave;
            float kss = 0.f;
            for (int m = 0; m < Md; ++m) { const float kv = sU[d.oK + h * Md + m]; kss += kv * kv; }
            const float kinv = 1.0f / sqrtf(fmaxf(kss, 1e-12f));
            if (lane < Md) sKs[h * Md + lane] = sU[d.oK + h * Md + lane] * kinv * sCn[lane];
            for (int m = lane + 64; m < Md; m += 64) sKs[h * Md + m] = sU[d.oK + h * Md + m] * kinv * sCn[m];
            const float beta = sU[d.oB + h], g = sU[d.oG + h], gamma = sU[d.oY + h];
            constexpr int MAXSS = FIX ? 3 : NTM_MAX_SHIFT_TAPS;     // (a compile-time 3 taps at the benchmark shape)
            float swv[MAXSS];                                      // softmax of the shift logits (ntm_cell.py:161)
            {
                float mx = -INFINITY;
#pragma unroll
                for (int j = 0; j < MAXSS; ++j) if (j < SS) mx = fmaxf(mx, sU[d.oS + h * SS + j]);
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < MAXSS; ++j) { swv[j] = (j < SS) ? ntm_exp(sU[d.oS + h * SS + j] - mx) : 0.f; sum += swv[j]; }
#pragma unroll
                for (int j = 0; j < MAXSS; ++j) swv[j] = swv[j] / sum;
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
            const int start = -((SS + 1) >> 1);                    // Py2 floor of -SS/2 (Q2): 3 -> -2
            float psum = 0.f;
            for (int n = lane; n < N; n += 64) {
                float wv = 0.f;
#pragma unroll
                for (int j = 0; j < MAXSS; ++j) {
                    if (j < SS) {
                        int src = n + start + j;
                        src = (src % N + N) % N;
                        wv += swv[j] * sWg[h * N + src];
                    }
                }
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
            float mx = -INFINITY;
            for (int j = 0; j < d.O; ++j) mx = fmaxf(mx, sU[d.P + j]);
            float sum = 0.f;
            for (int j = 0; j < d.O; ++j) sum += expf(sU[d.P + j] - mx);
            for (int j = 0; j < d.O; ++j) a.outputs[bt * d.O + j] = expf(sU[d.P + j] - mx) / sum;
        }
        __syncthreads();
        NTM_STAMP(4);
        // ------------------------------------------------------------ P8: write + read
        auto update_M = [&]() {
            for (int idx = tid; idx < N * Md; idx += T) {
                const int n = idx / Md, m = idx - n * Md;
                float E = 1.f, A = 0.f;
                for (int j = 0; j < Wh; ++j) {
                    const float ww = sW[(R + j) * N + n];
                    E *= (1.0f - ww * sU[d.oE + j * Md + m]);
                    A += ww * sU[d.oA + j * Md + m];
                }
                const float nm = sM[n * MP + m] * E + A;
                sM[n * MP + m] = nm;
                if (a.st_M) a.st_M[bt * N * Md + idx] = nm;
            }
        };
        if (d.write_first) { update_M(); __syncthreads(); }
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
        __syncthreads();
        NTM_STAMP(5);
        if (!d.write_first) update_M();
        if (tid < RM) {
            float s = 0.f;
            for (int sl = 0; sl < nslR; ++sl) s += sPart[sl * RM + tid];
            sZ[tid] = s;
            if (a.st_read) a.st_read[bt * RM + tid] = s;
        }
        __syncthreads();
        NTM_STAMP(6);
    }
#ifdef NTK_CL_PROF
    if (blockIdx.x == 0 && tid == 0) for (int i = 0; i < 16; ++i) g_ntm_fwd_prof[i] = s_prof[i];
#endif

    // ---- final state
    for (int i = tid; i < N * Md; i += T) a.M_out[(size_t)b * N * Md + i] = sM[(i / Md) * MP + (i % Md)];
    for (int i = tid; i < H * N; i += T) a.w_out[(size_t)b * H * N + i] = sW[i];
    for (int i = tid; i < RM; i += T) a.read_out[(size_t)b * RM + i] = sZ[i];
    for (int i = tid; i < hid; i += T) {
        a.cs_out[(size_t)b * 2 * hid + i] = sC[i];
        a.cs_out[(size_t)b * 2 * hid + hid + i] = sZ[RM + i];
    }
}

// pick the workgroup size: whole waves, enough threads for N slots x >=1 head, hid units + Md columns
static int ntm_pick_threads(const NtmDims& d) {
    int want = ntm_imax(d.H * d.N, 3 * d.hid);
    want = ntm_imax(want, d.hid + ntm_imax(d.Md, 4));
    want = ntm_imax(want, d.PP + d.Md);
    want = ntm_imax(want, d.H * d.Md + d.H + 1);
    want = ntm_imax(want, ((d.hid + 63) / 64 + 1) * 64);
    want = ntm_imax(want, (d.H + 1) * 64);
    want = ((want + 63) / 64) * 64;
    if (want > 1024) want = 1024;
    return want;
}

bool ntm_seq_fwd_ws_takes(const NtmDims& d);                        // ntm_seq_fwd_ws.hip
int ntm_seq_fwd_ws_launch(const NtmFwdArgs& a, void* stream);

int ntm_validate_dims(const NtmDims& d, const char* who) {
    NTK_REQUIRE(d.B > 0 && d.S > 0, NTK_ERR_BAD_SHAPE, "%s: B=%d S=%d", who, d.B, d.S);
    NTK_REQUIRE(d.N >= 64 && (d.N % 64) == 0 && d.N <= 1024, NTK_ERR_UNSUPPORTED,
                "%s: mem_size=%d must be a multiple of 64 in [64,1024]", who, d.N);
    NTK_REQUIRE(d.Md >= 1 && d.Md <= 256, NTK_ERR_UNSUPPORTED, "%s: mem_dim=%d out of range", who, d.Md);
    NTK_REQUIRE(d.R >= 1 && d.Wh >= 1, NTK_ERR_BAD_SHAPE, "%s: need >=1 read and write head (R=%d W=%d)", who, d.R, d.Wh);
    NTK_REQUIRE(d.hid >= 1 && d.hid + d.Md <= 1024 && d.PP + d.Md <= 1024 && d.H * d.Md + d.H + 1 <= 1024 &&
                    d.R * d.Md <= 1024,
                NTK_ERR_UNSUPPORTED, "%s: hidden=%d heads=%d mem_dim=%d exceed one workgroup", who, d.hid, d.H, d.Md);
    NTK_REQUIRE(d.SS >= 1 && d.SS < d.N && d.O >= 1, NTK_ERR_BAD_SHAPE, "%s: shift space %d / output_dim %d", who, d.SS, d.O);
    // limits of the kernels' fixed decomposition: shift taps live in a register array of NTM_MAX_SHIFT_TAPS (9: shift_range <= 4),
    // addressing runs one WAVE per head, and the column norms of M need one wave beyond those of the hidden units
    NTK_REQUIRE(d.SS <= NTM_MAX_SHIFT_TAPS, NTK_ERR_UNSUPPORTED, "%s: shift_range=%d (shift space %d > %d taps)", who, (d.SS - 1) / 2, d.SS,
                NTM_MAX_SHIFT_TAPS);
    NTK_REQUIRE((d.H + 1) * 64 <= 1024, NTK_ERR_UNSUPPORTED, "%s: %d heads (one wave per head: at most 15)", who, d.H);
    NTK_REQUIRE(((d.hid + 63) / 64 + 1) * 64 <= 1024, NTK_ERR_UNSUPPORTED, "%s: hidden=%d (at most 960)", who, d.hid);
    return NTK_OK;
}

extern "C" int ntk_ntm_padded_dims(int N, int Md, int R, int Wh, int hid, int shift_range, int O,
                                   int* P, int* PP, int* K, int* ldz, int* ldh) {
    NtmDims d;
    ntm_fill_dims(d, 1, 1, N, Md, R, Wh, hid, shift_range, O, 0);
    if (P) *P = d.P;
    if (PP) *PP = d.PP;
    if (K) *K = d.K;
    if (ldz) *ldz = d.ldz;
    if (ldh) *ldh = d.ldh;
    return NTK_OK;
}

extern "C" int ntk_ntm_seq_fwd(int B, int S, int N, int Md, int R, int Wh, int hid, int shift_range, int O,
                               int write_first,
                               const float* xproj, const float* Wr, const float* Wa,
                               const float* M0, const float* w0, const float* read0, const float* cs0,
                               float* logits, float* outputs,
                               float* M_out, float* w_out, float* read_out, float* cs_out,
                               float* st_z, float* st_gates, float* st_c, float* st_h, float* st_u,
                               float* st_wc, float* st_wv, float* st_w, float* st_M, float* st_read,
                               void* stream) {
    NtmFwdArgs a;
    ntm_fill_dims(a.d, B, S, N, Md, R, Wh, hid, shift_range, O, write_first);
    int rc = ntm_validate_dims(a.d, "ntk_ntm_seq_fwd");
    if (rc != NTK_OK) return rc;
    NTK_REQUIRE(xproj && Wr && Wa && M0 && w0 && read0 && cs0 && logits && M_out && w_out && read_out && cs_out,
                NTK_ERR_BAD_PTR, "ntk_ntm_seq_fwd: null pointer");
    NTK_REQUIRE(ntk_aligned16(xproj) && ntk_aligned16(Wr) && ntk_aligned16(Wa) &&
                    (!st_gates || ntk_aligned16(st_gates)),
                NTK_ERR_BAD_PTR, "ntk_ntm_seq_fwd: xproj/Wr/Wa/st_gates must be 16-byte aligned");
    const bool any = st_z || st_gates || st_c || st_h || st_u || st_wc || st_wv || st_w || st_M || st_read;
    NTK_REQUIRE(!st_gates == !st_c, NTK_ERR_BAD_PTR, "ntk_ntm_seq_fwd: st_gates and st_c go together");
    (void)any;
    a.xproj = xproj; a.Wr = Wr; a.Wa = Wa; a.M0 = M0; a.w0 = w0; a.read0 = read0; a.cs0 = cs0;
    a.logits = logits; a.outputs = outputs; a.M_out = M_out; a.w_out = w_out; a.read_out = read_out; a.cs_out = cs_out;
    a.st_z = st_z; a.st_gates = st_gates; a.st_c = st_c; a.st_h = st_h; a.st_u = st_u;
    a.st_wc = st_wc; a.st_wv = st_wv; a.st_w = st_w; a.st_M = st_M; a.st_read = st_read;
    // benchmark shape: the kernel whose recurrent weight stream runs beside the step instead of in front of it (ntm_seq_fwd_ws.hip;
    // NTK_NTM_FWD_FORM=res selects round 2's resident-rows kernel below, for comparison)
    const char* form_env = getenv("NTK_NTM_FWD_FORM");               // read per launch (development switch)
    const bool ws_off = form_env && form_env[0] == 'r';
    if (!ws_off && ntm_seq_fwd_ws_takes(a.d)) return ntm_seq_fwd_ws_launch(a, stream);
    int T = ntm_pick_threads(a.d);
    NTK_REQUIRE(T >= a.d.N, NTK_ERR_UNSUPPORTED, "ntk_ntm_seq_fwd: mem_size %d exceeds the workgroup", a.d.N);
    NtmLds L;
    ntm_fwd_lds(a.d, T, L);
    const bool fixdims = (N == 128 && Md == 20 && R == 4 && Wh == 1 && hid == 200 && shift_range == 1 && O == 2);
#ifdef NTK_NTM_FWD_STREAM_ONLY                                              // dev build: the all-streaming 640-thread specialisation of round 1
    const int variant = 0;
#else
    const int variant = fixdims ? 1 : 0;
#endif
    if (variant == 1) { T = 512; ntm_fwd_lds(a.d, T, L); }
    size_t lds_bytes = (size_t)L.total * sizeof(float) + 128;                    // + the diagnostic build's stamp words
    if (variant == 1) lds_bytes += (size_t)20 * 2 * hid * sizeof(f32x4);
    NTK_REQUIRE(lds_bytes <= 160 * 1024, NTK_ERR_UNSUPPORTED,
                "ntk_ntm_seq_fwd: state needs %zu B of LDS (> 160 KiB)", lds_bytes);
    {
        static NtkLdsAttrCache lds_cache;
        const void* const ks[] = {(const void*)ntm_seq_fwd_kernel<768, 0>, (const void*)ntm_seq_fwd_kernel<1024, 0>, (const void*)ntm_seq_fwd_kernel<768, 640>,
                                  (const void*)ntm_seq_fwd_kernel<512, 512, 24, 20>};
        const int rc_lds = ntk_raise_lds_limit(lds_cache, ks, 4, "ntk_ntm_seq_fwd");
        if (rc_lds != NTK_OK) return rc_lds;
    }
    if (variant == 1) ntm_seq_fwd_kernel<512, 512, 24, 20><<<B, T, lds_bytes, (hipStream_t)stream>>>(a, L);
    else if (fixdims && T == 640) ntm_seq_fwd_kernel<768, 640><<<B, T, lds_bytes, (hipStream_t)stream>>>(a, L);
    else if (T <= 768) ntm_seq_fwd_kernel<768, 0><<<B, T, lds_bytes, (hipStream_t)stream>>>(a, L);
    else ntm_seq_fwd_kernel<1024, 0><<<B, T, lds_bytes, (hipStream_t)stream>>>(a, L);
    NTK_CHECK_LAUNCH("ntk_ntm_seq_fwd");
    return NTK_OK;
}
