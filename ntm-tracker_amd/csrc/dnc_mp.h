// Shared pieces of the memory-partitioned DNC cluster kernels (dnc_mp_fwd.hip / dnc_mp_bwd.hip): k workgroups, one per CU,
// cooperate on ONE sequence, as in dnc_cluster.h, but NOTHING of size N x N or N x W is replicated or kept per workgroup
// beyond its own rows, so the form has no upper bound on the memory other than HBM:
//
//   temporal link   N/k rows per workgroup, STREAMED from and to HBM once per step (the recorded forward pass reads the
//                   rows of record t-1 and writes record t: the BPTT record IS the state; inference updates the state in
//                   place).  dnc/addressing.py:183-218 makes three passes over the link per step (update, two batched
//                   matmuls for the directional reads); here it is one: update, diagonal, store, both directional reads.
//                   Measured floor of that pass alone (scripts/probe/link_stream_probe.hip, profiles/r03_link_stream_probe.txt):
//                   64 sequences x 512 x 512 at 7.0 TB/s = 19 us per step forward, 36 us per step for the BPTT pass.
//   memory          N/k rows per workgroup, LDS resident (forward) / in registers (BPTT) for the whole launch;
//   per slot        usage, allocation, write / read weights, precedence: replicated and computed redundantly from
//                   bit-identical inputs (the rank of a slot in the usage order is counted over N/k keys per workgroup and
//                   the integer partial counts are exchanged: exact);
//   controller      hidden units split k ways (gate product and interface product by own units).
//
// Four hand-offs per step and direction (protocol and the same-XCD fast form: dnc_cluster.h), forward:
//   (0) h slice + interface partials            (A) write-content scores of the own rows + rank partial counts
//   (B) forward directional reads of the own rows, backward directional read partials, read-content scores of the own rows
//   (C) partial read vectors (own rows of rw x M)
#pragma once
#include "dnc_cluster.h"

constexpr int MPX = 4;        // hand-offs per step

struct DncMpCfg {
    int N, W, R, hid, O;
    int I, IP, K, ldz, ldh, Ky, ldy, OP;
    int oV, oE, oF, oAg, oWg, oRm, oKw, oBw, oKr, oBr;
    int k, NR, upk, upkp;
    int slot[MPX];              // floats per mailbox slot of each hand-off (forward)
    int ksl, kperG;             // gate product: K slices, rows per slice (thread = own unit x slice)
    int icg, nslI, uperI;       // interface partial: float4 column groups, unit slices, units per slice
    int mperA;                  // keys of the rank partial count = N / k
    int N4, W4, NH;             // float4s per link row / word; 64-lane float4 groups per link row (1 or 2)
    int TPR, FPT, WS4, RPP;     // memory passes: threads per row, float4s per thread, padded row stride (float4), rows per pass
    int RW4, nslR, nperR;       // read vectors: (head, float4 of the word) pairs, row slices, rows per slice
    int RNP;                    // ceil(R * N / CLT): (head, slot) pairs per thread in the read-weight mix
    unsigned mg_upk, mg_icg, mg_NR, mg_N, mg_W4, mg_RW4;
};

static constexpr __host__ __device__ int dnc_mp_pow2floor(int x) { int p = 1; while (2 * p <= x) p *= 2; return p; }

static constexpr __host__ __device__ DncMpCfg dnc_mp_cfg(int N, int W, int R, int hid, int O, int k) {
    DncMpCfg c = {};
    c.N = N; c.W = W; c.R = R; c.hid = hid; c.O = O; c.k = k;
    c.oV = 0; c.oE = W; c.oF = 2 * W; c.oAg = c.oF + R; c.oWg = c.oAg + 1; c.oRm = c.oWg + 1; c.oKw = c.oRm + 3 * R;
    c.oBw = c.oKw + W; c.oKr = c.oBw + 1; c.oBr = c.oKr + R * W; c.I = c.oBr + R; c.IP = (c.I + 3) & ~3;
    c.K = R * W + hid; c.ldz = (c.K + 1 + 3) & ~3; c.ldh = (hid + 1 + 3) & ~3;
    c.Ky = hid + R * W; c.ldy = (c.Ky + 1 + 3) & ~3; c.OP = (O + 3) & ~3;
    c.NR = N / k;
    c.upk = (hid + k - 1) / k;
    c.upkp = dnc_cluster_align4(c.upk);
    c.slot[0] = dnc_cluster_align4(c.upkp + c.IP);
    c.slot[1] = dnc_cluster_align4(c.NR + N);
    c.slot[2] = dnc_cluster_align4(2 * R * c.NR + R * N);
    c.slot[3] = dnc_cluster_align4(R * W);
    c.ksl = dnc_cluster_max(1, CLT / dnc_cluster_max(1, c.upk));
    if (c.ksl > c.K) c.ksl = c.K;
    c.kperG = (c.K + c.ksl - 1) / c.ksl;
    c.icg = c.IP / 4;
    c.nslI = dnc_cluster_max(1, CLT / c.icg);
    c.uperI = (c.upk + c.nslI - 1) / c.nslI;
    c.mperA = N / k;
    c.N4 = N / 4; c.W4 = W / 4;
    c.NH = (c.N4 + 63) / 64;
    c.TPR = dnc_mp_pow2floor(dnc_cluster_max(1, CLT / c.NR));
    if (c.TPR > dnc_mp_pow2floor(c.W4)) c.TPR = dnc_mp_pow2floor(c.W4);
    if (c.TPR > 64) c.TPR = 64;
    c.FPT = (c.W4 + c.TPR - 1) / c.TPR;
    c.WS4 = c.W4 + c.TPR;
    c.RPP = CLT / c.TPR;
    c.RW4 = R * c.W4;
    c.nslR = dnc_cluster_max(1, CLT / c.RW4);
    if (c.nslR > c.NR) c.nslR = c.NR;
    c.nperR = (c.NR + c.nslR - 1) / c.nslR;
    c.RNP = (R * N + CLT - 1) / CLT;
    c.mg_upk = dnc_cluster_magic(c.upk); c.mg_icg = dnc_cluster_magic(c.icg); c.mg_NR = dnc_cluster_magic(c.NR);
    c.mg_N = dnc_cluster_magic(N); c.mg_W4 = dnc_cluster_magic(c.W4); c.mg_RW4 = dnc_cluster_magic(c.RW4);
    return c;
}

// Shapes with a compile-time instantiation of the kernels (every dimension, offset and LDS address an immediate; the generic
// instantiation reads them from the kernarg segment and its unrolled eight-row loops with run-time guards spill heavily):
//   1  BASELINE configs[4]'s core: DNC 512 x 128, 4 read heads, hidden 200, 2 outputs, 4 workgroups per sequence
//      (64 sequences = 256 workgroups = every CU) -- the shape this form exists for;
//   2  BASELINE configs[2]'s core (256 x 64) at 4 workgroups per sequence: 32 sequences on HALF the chip, so that the trunk pass
//      of the next batch runs on the other half at the same time (the LDS-resident form needs k = 8 = every CU);
//   3  the same core at 8 workgroups per sequence.
constexpr int MP_SHAPES = 3;
static constexpr __host__ __device__ DncMpCfg dnc_mp_shape_cfg(int shape) {
    return shape == 2 ? dnc_mp_cfg(256, 64, 4, 200, 2, 4) : (shape == 3 ? dnc_mp_cfg(256, 64, 4, 200, 2, 8) : dnc_mp_cfg(512, 128, 4, 200, 2, 4));
}
static inline int dnc_mp_shape_of(const DncMpCfg& c) {
    for (int sh = 1; sh <= MP_SHAPES; ++sh) {
        const DncMpCfg f = dnc_mp_shape_cfg(sh);
        if (c.N == f.N && c.W == f.W && c.R == f.R && c.hid == f.hid && c.O == f.O && c.k == f.k) return sh;
    }
    return 0;
}

// control block of a launch: flags [B][MPX][k], the error word, the XCC words of the handshake [B][k]; padded to 256 bytes;
// zeroed before EVERY launch.  The mailbox follows, then one 256-byte line whose first word is the STICKY error word: set
// together with the launch's error word, never cleared by a launch (the owner of the workspace zeroes it once), so that a
// hand-off that timed out in an earlier launch of a multi-launch pass is still seen when the caller next looks.
static inline size_t dnc_mp_ctrl_bytes(int B, int k) { return (((size_t)B * (MPX + 1) * k + 1) * sizeof(unsigned) + 255) & ~(size_t)255; }
static inline size_t dnc_mp_mbox_floats(int B, int k, const int* slot) {
    size_t per = 0;
    for (int e = 0; e < MPX; ++e) per += (size_t)2 * k * slot[e];
    return (size_t)B * per;
}
static inline size_t dnc_mp_workspace_bytes(int B, int k, const int* slot) {
    return dnc_mp_ctrl_bytes(B, k) + ((dnc_mp_mbox_floats(B, k, slot) * sizeof(float) + 255) & ~(size_t)255) + 256;
}

// 16-byte payload loads of a hand-off: raw buffer loads with sc1 (aux 16: the per-CU L1 is bypassed, as the 4- and 8-byte agent-scope
// loads of cl_load do; MI355X_MICROARCH.md "valid forms": buffer_load_dwordx4 sc1 is one of them).  The consumers of hand-offs B / 1 / 2
// read 32-80 KB per step: with 4-byte loads that was 5-8 us of a step.  rs: resource over one sequence's [parity][g][slot] region
// of a hand-off (wave-uniform); offsets in floats, multiples of 4.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mp_rsrc(const float* base, size_t floats) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(floats * sizeof(float)), 0x00020000);
}
__device__ __forceinline__ f32x4 mp_load4(__amdgpu_buffer_rsrc_t rs, int float_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, float_off * 4, 0, 16));
}
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 mp_load4i(__amdgpu_buffer_rsrc_t rs, int float_off) {
    return __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, float_off * 4, 0, 16));
}

// wait with its OWN time base (a bound on one stalled exchange, not on the kernel's run time) and a sticky error word
__device__ __forceinline__ bool mp_wait(const unsigned* flags, unsigned epoch, int k, unsigned* err, unsigned* sticky, int* s_abort, int tid) {
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
                    if (tid == 0) {
                        __hip_atomic_store(err, 1u, NTK_RLX, NTK_AGENT);
                        __hip_atomic_store(sticky, 1u, NTK_RLX, NTK_AGENT);
                        *s_abort = 1;
                    }
                    break;
                }
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __syncthreads();
    return *s_abort == 0;
}
