// Shared pieces of the DNC CLUSTER kernels (dnc_cluster_fwd.hip / dnc_cluster_bwd.hip): k workgroups, one per CU,
// cooperate on ONE sequence.  Each workgroup owns N/k rows of the N x N temporal link (LDS resident) and 1/k of the
// controller's hidden units; everything per-slot (usage, weights, the N x W memory) is replicated and computed
// redundantly from bit-identical inputs, so the replicas never diverge.  What must cross workgroups does so through
// a per-sequence mailbox in global memory, twice per step in each direction.
//
// Hand-off protocol (MI355X_MICROARCH.md "Valid forms", first row; cdna_hip_programming.md Guideline 16, R1):
//   producer  every payload store is an agent-scope relaxed atomic store (global_store ... sc1: write-through),
//             every storing wave drains with s_waitcnt vmcnt(0), workgroup barrier, ONE lane stores the flag
//             (sc1) = epoch;
//   consumer  ONE wave polls the k flags of its cluster with sc1 loads (relaxed, s_sleep between polls), then a
//             workgroup barrier, then EVERY load of the payload is an sc1 load to registers (no acquire fence
//             needed in that form; per-CU L1 is bypassed, placement on XCDs is irrelevant for correctness).
// Same-XCD fast form: when the k workgroups of a cluster find, by an agent-scope handshake at launch, that they all run
// on ONE XCD (cl_same_xcd: each publishes its HW_REG_XCC_ID; observed for blocks with equal blockIdx % 8, never
// assumed), payload and flag stores are PLAIN stores: they stay in that XCD's L2, which every CU of the XCD reads
// through, and the consumer's sc1 loads (L1 bypass) are served from it.  Measured (scripts/probe/cluster_probe.hip, k = 8,
// 1 KB payloads): 4.8 us per exchange across XCDs, 3.5 us with write-through stores on one XCD, 1.9 us with plain stores.
// A cluster that spans XCDs (or a launch whose batch is not a multiple of 8) keeps the write-through form.
// Epochs are step + 1, flags are zeroed by a memset node ahead of every launch, payload slots are double-buffered by
// step parity.  Every spin is bounded (s_memrealtime, ~3 s from the start of THAT wait): on timeout the waiter raises the launch's error word,
// which every other spin also watches, and all workgroups leave the kernel (outputs are then garbage and the host
// reports NTK_ERR_HIP from ntk_dnc_cluster_status).  All k * B workgroups must be co-resident: the host caps the
// grid at one workgroup per CU.
#pragma once
#include "dnc_common.h"

void dnc_cluster_latch(const unsigned* err, unsigned* sticky, void* stream);      // dnc_cluster_fwd.hip

constexpr int CT = 512;       // threads per cluster workgroup
constexpr int CW = CT / 64;   // waves

#define NTK_RLX __ATOMIC_RELAXED
#define NTK_AGENT __HIP_MEMORY_SCOPE_AGENT

// plain: wave-uniform, true only after cl_same_xcd() said the cluster shares an XCD
__device__ __forceinline__ void cl_store(float* p, float v, bool plain) {
    if (plain) *p = v;
    else __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), NTK_RLX, NTK_AGENT);
}
__device__ __forceinline__ float cl_load(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), NTK_RLX, NTK_AGENT));
}
__device__ __forceinline__ void cl_load2(const float* p, float& a, float& b) {    // p 8-byte aligned
    const unsigned long long x = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), NTK_RLX, NTK_AGENT);
    a = __uint_as_float((unsigned)x);
    b = __uint_as_float((unsigned)(x >> 32));
}

// publish: call from ALL threads of the workgroup after the payload stores
__device__ __forceinline__ void cl_publish(unsigned* my_flag, unsigned epoch, int tid, bool plain) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        if (plain) { *my_flag = epoch; asm volatile("" ::: "memory"); }
        else __hip_atomic_store(my_flag, epoch, NTK_RLX, NTK_AGENT);
    }
}

// Launch-time handshake: does every workgroup of this cluster run on the same XCD?  xw: the cluster's k words of the
// control block (zeroed with the flags).  Every workgroup publishes 1 + its XCC id with a write-through store and polls
// all k words (bounded like cl_wait); all of them see the same k values, so the answer is uniform over the cluster.
// Returns 1 / 0, or -1 when the launch was aborted.  s_word: one int in LDS.
__device__ __forceinline__ int cl_same_xcd(unsigned* xw, int g, int k, unsigned* err, int* s_abort, int* s_word,
                                           unsigned long long t_start, int tid) {
    if (tid < 64) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        id = (id & 15u) + 1u;
        if (tid == 0) __hip_atomic_store(xw + g, id, NTK_RLX, NTK_AGENT);
        unsigned spins = 0;
        int same = 0;
        for (;;) {
            const unsigned v = (tid < k) ? __hip_atomic_load(xw + tid, NTK_RLX, NTK_AGENT) : id;
            if (__all(v != 0u)) { same = __all(v == id) ? 1 : 0; break; }
            if ((++spins & 127u) == 0) {
                const bool dead = __hip_atomic_load(err, NTK_RLX, NTK_AGENT) != 0 ||
                                  (__builtin_amdgcn_s_memrealtime() - t_start) > 300000000ull;
                if (dead) {
                    if (tid == 0) { __hip_atomic_store(err, 1u, NTK_RLX, NTK_AGENT); *s_abort = 1; }
                    same = -1;
                    break;
                }
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (tid == 0) *s_word = same;
    }
    __syncthreads();
    return *s_word;
}

// wait until all k flags of the cluster have reached `epoch`; returns false (uniformly over the workgroup) when the
// launch was aborted.  s_abort: one int in LDS, zero-initialised before the first call.
__device__ __forceinline__ bool cl_wait(const unsigned* flags, unsigned epoch, int k, unsigned* err, int* s_abort,
                                        unsigned long long t_start, int tid) {
    (void)t_start;      // the bound is on ONE stalled exchange, not on the kernel's run time: the time base is taken inside the wait
    if (tid < 64) {
        unsigned spins = 0;
        unsigned long long t0 = 0;
        for (;;) {
            const unsigned v = (tid < k) ? __hip_atomic_load(flags + tid, NTK_RLX, NTK_AGENT) : epoch;
            if (__all((int)(v - epoch) >= 0)) break;
            if ((++spins & 127u) == 0) {
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                if (t0 == 0) t0 = now;
                const bool dead = __hip_atomic_load(err, NTK_RLX, NTK_AGENT) != 0 || (now - t0) > 300000000ull;      // 3 s at 100 MHz
                if (dead) {
                    if (tid == 0) { __hip_atomic_store(err, 1u, NTK_RLX, NTK_AGENT); *s_abort = 1; }
                    break;
                }
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // compiler-only: payload loads stay below the poll
    __syncthreads();
    return *s_abort == 0;
}

// sum over groups of G consecutive lanes (G a power of two <= 64), result in every lane of the group.
// G <= 16 stays on the DPP data path; 32 / 64 add ds_swizzle / readlane steps.
template <int G>
__device__ __forceinline__ float group_sum(float v) {
    if constexpr (G >= 2) v += ntk_dpp<0xB1>(v);       // quad_perm [1,0,3,2]
    if constexpr (G >= 4) v += ntk_dpp<0x4E>(v);       // quad_perm [2,3,0,1]
    if constexpr (G >= 8) v += ntk_dpp<0x141>(v);      // row_half_mirror
    if constexpr (G >= 16) v += ntk_dpp<0x140>(v);     // row_mirror
    if constexpr (G >= 32) v += __shfl_xor(v, 16, 64);
    if constexpr (G >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float group_sum_rt(float v, int G) {
    switch (G) {
        case 1: return v;
        case 2: return group_sum<2>(v);
        case 4: return group_sum<4>(v);
        case 8: return group_sum<8>(v);
        case 16: return group_sum<16>(v);
        case 32: return group_sum<32>(v);
        default: return group_sum<64>(v);
    }
}

// Reciprocal / square root on the hardware transcendental units (v_rcp_f32, v_sqrt_f32: ~1 ulp).  The per-row score and
// norm terms of the memory passes are computed by every lane of a row's group: with IEEE division (a ~10-instruction
// Newton sequence each) they were a third of the backward memory passes' instructions.
__device__ __forceinline__ float cl_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float cl_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// Pointwise functions of the controller / interface on the hardware exp / log / rcp units (v_exp_f32, v_log_f32, ~1-2 ulp),
// as the NTM sequence kernels do (ntm_common.h): the snt.LSTM pointwise step of 25 units sits on the step's critical path.
__device__ __forceinline__ float cl_sigmoid(float x) { return cl_rcp(1.0f + __expf(-x)); }
__device__ __forceinline__ float cl_tanh(float x) { return 1.0f - 2.0f * cl_rcp(1.0f + __expf(2.0f * x)); }
__device__ __forceinline__ float cl_softplus(float x) { return fmaxf(x, 0.f) + __logf(1.0f + __expf(-fabsf(x))); }

// LDS image of the link slice: row r (local), column c -> float index.  float4 groups are XOR-swizzled with the row
// so that the row-major float4 / scalar accesses of the update and of the backward-read MFMA operand are
// conflict-free and the column-major scalar reads of the forward-read operand are 4-way at worst.
__device__ __forceinline__ int cl_lidx(int r, int c, int N) { return r * N + ((((c >> 2) ^ (r & 7))) << 2) + (c & 3); }

// Everything about a launch that depends only on the SHAPE (memory N x W, R read heads, hidden size, outputs, cluster
// size k): interface offsets, padded leading dimensions, the work decomposition of a step and the magic reciprocals
// that replace integer divisions in the kernels.  One constexpr function fills it, on the host for every launch and at
// COMPILE time for the benchmark shape (the FIX instantiations of the kernels fold all of it into immediates, which
// is what keeps their uniform state inside the 102 SGPRs a wave has).
struct DncClusterCfg {
    int N, W, R, hid, O;
    int I, IP, K, ldz, ldh, Ky, ldy, OP;
    int oV, oE, oF, oAg, oWg, oRm, oKw, oBw, oKr, oBr;
    int k;            // workgroups per sequence
    int NR;           // link / record rows owned by one workgroup = N / k
    int upk, upkp;    // hidden units per workgroup = ceil(hid / k), rounded up to 4
    int slot0;        // floats per mailbox slot of exchange 0 (controller: h slice + interface partial)
    int slot1;        // floats per mailbox slot of exchange 1 (link: forward rows + backward partial)
    int ksl, kperG;   // gate product: K-slices, rows per slice (a thread = one own unit x one slice)
    int icg, nslI, uperI;   // interface partial: float4 column groups, unit slices, units per slice
    int nslA, mperA;  // rank computation: slices of the slot range, slots per slice
    int strips, NRp;  // N / 64, NR rounded up to 32
    int HW4;          // float4s of a memory row per thread of the row pair
    int nperW;        // memory rows per wave in the read-vector product
    // magic reciprocals: x / dv == __umulhi(x, mg) for x * dv < 2^32 (mg = ceil(2^32 / dv), dv >= 2; 0 means dv == 1)
    unsigned mg_upk, mg_icg, mg_NR, mg_N, mg_N4, mg_W4;
};

constexpr int CLT = 512;      // threads per cluster workgroup (= CT below)

static constexpr __host__ __device__ int dnc_cluster_align4(int x) { return (x + 3) & ~3; }
static constexpr __host__ __device__ int dnc_cluster_max(int a, int b) { return a > b ? a : b; }
static constexpr __host__ __device__ unsigned dnc_cluster_magic(int dv) {
    return dv <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)dv - 1) / (unsigned)dv);
}
__device__ __forceinline__ int cl_div(int x, unsigned mg) { return mg ? (int)__umulhi((unsigned)x, mg) : x; }

static constexpr __host__ __device__ DncClusterCfg dnc_cluster_cfg(int N, int W, int R, int hid, int O, int k) {
    DncClusterCfg c = {};
    c.N = N; c.W = W; c.R = R; c.hid = hid; c.O = O; c.k = k;
    // interface layout and padded widths (num_writes = 1): as dnc_fill_dims (dnc_common.h)
    c.oV = 0; c.oE = W; c.oF = 2 * W; c.oAg = c.oF + R; c.oWg = c.oAg + 1; c.oRm = c.oWg + 1; c.oKw = c.oRm + 3 * R;
    c.oBw = c.oKw + W; c.oKr = c.oBw + 1; c.oBr = c.oKr + R * W; c.I = c.oBr + R; c.IP = (c.I + 3) & ~3;
    c.K = R * W + hid; c.ldz = (c.K + 1 + 3) & ~3; c.ldh = (hid + 1 + 3) & ~3;
    c.Ky = hid + R * W; c.ldy = (c.Ky + 1 + 3) & ~3; c.OP = (O + 3) & ~3;
    c.NR = N / k;
    c.upk = (hid + k - 1) / k;
    c.upkp = dnc_cluster_align4(c.upk);
    c.slot0 = dnc_cluster_align4(c.upkp + c.IP);
    c.slot1 = dnc_cluster_align4(R * c.NR + R * N);
    c.ksl = dnc_cluster_max(1, CLT / dnc_cluster_max(1, c.upk));
    if (c.ksl > c.K) c.ksl = c.K;
    c.kperG = (c.K + c.ksl - 1) / c.ksl;
    c.icg = c.IP / 4;
    c.nslI = dnc_cluster_max(1, CLT / c.icg);
    c.uperI = (c.upk + c.nslI - 1) / c.nslI;
    c.nslA = dnc_cluster_max(1, CLT / N);
    c.mperA = N / c.nslA;
    c.strips = N / 64;
    c.NRp = ((c.NR + 31) / 32) * 32;
    c.HW4 = (W / 4 + 1) / 2;
    c.nperW = N / (CLT / 64);
    c.mg_upk = dnc_cluster_magic(c.upk); c.mg_icg = dnc_cluster_magic(c.icg); c.mg_NR = dnc_cluster_magic(c.NR);
    c.mg_N = dnc_cluster_magic(N); c.mg_N4 = dnc_cluster_magic(N / 4); c.mg_W4 = dnc_cluster_magic(W / 4);
    return c;
}

// the benchmark shape (BASELINE configs[2]: DNC 256 x 64, 4 read heads, hidden 200, 2 outputs) at 8 workgroups per sequence
constexpr DncClusterCfg kDncClusterFixCfg = dnc_cluster_cfg(256, 64, 4, 200, 2, 8);
static inline bool dnc_cluster_is_fix(const DncClusterCfg& c) {
    return c.N == 256 && c.W == 64 && c.R == 4 && c.hid == 200 && c.O == 2 && c.k == 8;
}

// control block of a launch: flags [B][2][k], the error word, the XCC words of the handshake [B][k]; padded to 256 bytes and
// zeroed before EVERY launch (dnc_cluster_ctrl_zero_bytes); then one more 256-byte line whose first word is the STICKY
// error word: a one-thread latch kernel enqueued behind every launch ORs the launch's error word into it, no launch clears
// it (the owner of the workspace zeroes it once), so an abort in an EARLIER launch of a multi-launch pass (segmented
// forward, re-recording, BPTT) is still there when the caller looks.  The mailbox follows.
static inline size_t dnc_cluster_ctrl_zero_bytes(int B, int k) { return (((size_t)B * 3 * k + 1) * sizeof(unsigned) + 255) & ~(size_t)255; }
static inline size_t dnc_cluster_ctrl_bytes(int B, int k) { return dnc_cluster_ctrl_zero_bytes(B, k) + 256; }

// mailbox layout (floats): per sequence [exchange][parity][g][slot]; flags (unsigned): per sequence [exchange][g]
static inline size_t dnc_cluster_mbox_floats(int B, int k, int slot0, int slot1) {
    return (size_t)B * 2 * k * ((size_t)slot0 + slot1);
}
