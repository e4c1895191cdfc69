// VGG 3x3 convolution as fused Winograd F(4x4,3x3) on the fp32 MFMA pipe (gfx950).
//
// out = relu(conv3x3_same(in, w) + b) [+ 2x2/2 max pool] -- the same operator as mfma_f32.hip / conv_wino.hip
// (vgg.py:155-161 via slim.conv2d).  Every 4x4 output tile is
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A          (Lavin & Gray, interpolation points 0, +-1, +-2, inf),
// summed over input channels: 36 independent GEMMs  M_p[tile, cout] = sum_c V_p[tile, c] U_p[c, cout], p = 6 i + j --
// 4x fewer multiplies than the direct form (F(2x2,3x3): 2.25x).  fp32 rounding error is ~16x that of F(2x2,3x3)
// (4e-6 .. 9e-6 of the activation scale per layer, scripts/dev_wino43_numerics.py), inside the 1e-4 bound of the path.
//
// Why this shape of kernel.  Measured on gfx950 (scripts/probe/mfma_coissue_probe.hip, SQ_VALU_MFMA_COEXEC_CYCLES = 0):
// v_mfma_f32_32x32x2_f32 and ordinary VALU / LDS / VMEM instruction issue exclude each other on a SIMD, so a SIMD's time is
// the SUM of its MFMA cycles and everything else it issues.  F(4x4) cuts the MFMA cycles per output by 1.78x at about
// the same transform arithmetic per output.  36 planes x (32 tiles x 64 channels) of accumulators are 288 KB: exactly ONE
// such block fits a CU's register file, so a workgroup is 4 waves, ONE PER SIMD, each with up to 512 registers
// (9 planes x 2 column blocks x 16 = 288 accumulators), one workgroup per CU.
//
//   wave w multiplies planes 9w .. 9w+8.  A operand (V_p) from LDS, B operand (U_p) from global memory, lane-major
//     packed, requested one whole K step (8 input channels) ahead.
//   input transform: per K step four tasks, one per wave: plane rows (1,2) and (3,4) (pairs share their row pass),
//     row 0, row 5 (the K loop is specialised per task: four straight-line bodies); lane = (tile, channel quad); window reads are conflict-free ds_read_b128 (the patch image
//     keeps a spare pixel after every four columns: a tile's 4-pixel stride becomes 5 pixel slots = 10 bank slots).
//   patch and V are double-buffered: one workgroup barrier per K step.
//   epilogue: accumulators -> LDS (one 32-channel half at a time), 256 threads = (tile, four adjacent channels) run
//     A^T M A, bias, ReLU (+ 2x2 pool inside the 4x4 tile) and store one float4 per pixel, NHWC.
#include "common.h"
#include <type_traits>

// Diagnostic build only (make prof, -DNTK_CL_PROF): lane 0 of every wave of ONE workgroup accumulates s_memtime deltas per
// section of the K loop into g_w43_prof[wave][section]; ntk_vgg_wino43_prof() copies them out (read SHARES, not totals).
#ifdef NTK_CL_PROF
__device__ unsigned long long g_w43_prof[4][24];
#define W43_STAMP(i)                                                                  \
    do {                                                                              \
        if (prof_on) {                                                                \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();             \
            prof_acc[i] += now_ - prof_last;                                          \
            prof_last = now_;                                                         \
        }                                                                             \
    } while (0)
#else
#define W43_STAMP(i) do { } while (0)
#endif

// Ablation switches for scripts/dev_wino_variant.sh (results are WRONG with any of them set; never defined in the product
// build): bit 0 no U loads in the K loop, bit 1 no input transform, bit 2 no patch staging, bit 3 no workgroup barrier in
// the K loop, bit 4 no A-operand reads, bit 5 (eight-wave kernel) contiguous staging addresses.
#ifndef W43_ABL
#define W43_ABL 0
#endif
// output stores are non-temporal: a layer's 0.5 - 8 GB of activations are re-read by the next launch from HBM whatever the
// policy, and keeping them out of L2 leaves it to the weights and the patch halos (measured: nine layers -0.45 %)
#ifdef W43_PLAIN_STORES
#define W43_STORE(p, v) (*(p) = (v))
#else
#define W43_STORE(p, v) __builtin_nontemporal_store(v, p)
#endif
#ifndef W43_STAGE_AUX
#define W43_STAGE_AUX 0      // cache-policy bits of the patch loads (2 = non-temporal: measured)
#endif

namespace {

constexpr int W4T = 256;

struct Wino43Args {
    const float* in; const float* U; const float* bias; float* out;
    int frames, H, W, Cin, Cout;
    int nCB;            // Cout / 64
    int bxN, byN;       // sub-blocks per frame (of the computed window): W/(4 TW), H/(4 TH) for the whole frame
    int bx0, by0;       // first sub-block column / row of the window (0: whole frame)
    int NS;             // workgroups per column block = ceil(NQ / NSUB)
    int NQ;             // sub-blocks = frames * byN * bxN
};

// weights: HWIO [3][3][Cin][Cout] -> U_p = G g G^T (float64 arithmetic, stored fp32), packed for the B operand:
// index = ((((cb * n8 + c8) * 4 + w) * 9 + j) * 2 + nb) * 256 + lane * 4 + q
//   with plane p = 9 w + j = 6 pi + pj,  c = 8 c8 + 4 (lane >> 5) + q,  cout = 64 cb + 32 nb + (lane & 31)
__global__ void wino43_pack_kernel(const float* __restrict__ w, float* __restrict__ U, int Cin, int Cout) {
    const size_t total = (size_t)36 * Cin * Cout;
    const int n8 = Cin / 8;
    const double G[6][3] = {{0.25, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int q = idx & 3, lane = (idx >> 2) & 63;
        size_t r = idx >> 8;
        const int nb = r & 1; r >>= 1;
        const int j = (int)(r % 9); r /= 9;
        const int wv = r & 3; r >>= 2;
        const int c8 = (int)(r % n8);
        const int cb = (int)(r / n8);
        const int p = 9 * wv + j, pi = p / 6, pj = p % 6;
        const int c = 8 * c8 + 4 * (lane >> 5) + q;
        const int o = 64 * cb + 32 * nb + (lane & 31);
        double s = 0.0;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) s += G[pi][ky] * G[pj][kx] * (double)w[((size_t)(ky * 3 + kx) * Cin + c) * Cout + o];
        U[idx] = (float)s;
    }
}

template <int N> using ic = std::integral_constant<int, N>;

// B^T of F(4x4,3x3) applied to six values (rows or columns): the pieces the tasks are built from
__device__ __forceinline__ f32x4 w43_r0(f32x4 d0, f32x4 d2, f32x4 d4) { return 4.f * d0 + (d4 - 5.f * d2); }
__device__ __forceinline__ f32x4 w43_r5(f32x4 d1, f32x4 d3, f32x4 d5) { return 4.f * d1 + (d5 - 5.f * d3); }

// MFMA with the accumulator in ARCHITECTURAL registers.  The compiler keeps every builtin MFMA's accumulator in the 256
// accumulation registers; a wave here owns 288, so the ninth plane's two accumulators live in VGPRs and are multiplied
// through this form (without it the compiler shuttles 32 registers through v_accvgpr_read/write around every use).
// Hazards: operands come from LDS / global loads (the compiler's waitcnt insertion sees the asm operands); an accumulator
// is only ever read back by the epilogue, hundreds of instructions after its last MFMA.
__device__ __forceinline__ void w43_mfma_v(f32x16& c, float av, float bv) {
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
}

template <bool POOL, int TW, int TH, int NSUB, int PWS, int SPXS>
__global__ __launch_bounds__(W4T) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_wino43_kernel(Wino43Args a) {
    constexpr int STILE = TW * TH, NTILE = NSUB * STILE, PW = 4 * TW + 2, PH = 4 * TH + 2, SPX = PW * PH, NPX = NSUB * SPX;
    constexpr int NPXS = NSUB * SPXS;
    constexpr int NST = (NPX * 2 + W4T - 1) / W4T;                   // float4 staging slots per thread (2 per pixel)
    constexpr int RAWF = NPXS * 8 + 64;                              // floats per patch buffer (+ scratch for unused slots)
    constexpr int VF = 36 * 32 * 8;                                  // floats per V buffer
    constexpr int ZF = 36 * 32 * 32;                                 // epilogue image of one 32-channel half
    constexpr int LDSF = (2 * RAWF + 2 * VF) > ZF ? (2 * RAWF + 2 * VF) : ZF;
    static_assert(NTILE == 32, "tile block = 32 MFMA rows");
    static_assert(LDSF * 4 + 512 <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) float s_mem[LDSF];
    __shared__ int s_sbf[NSUB], s_sby[NSUB], s_sbx[NSUB];
    float* s_raw = s_mem;                    // [2][RAWF]
    float* s_V = s_mem + 2 * RAWF;           // [2][VF]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    int cb, sp;
    if (a.nCB >= 8) {
        const int kN = a.nCB >> 3;
        cb = (slot % kN) * 8 + xcd;
        sp = slot / kN;
    } else {
        const int per = 8 / a.nCB;
        cb = xcd % a.nCB;
        sp = slot * per + xcd / a.nCB;
    }
    if (sp >= a.NS) return;
    const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
    if (tid < NSUB) {
        const int sq = sp * NSUB + tid;
        if (sq < a.NQ) {
            const int bx = sq % a.bxN;
            const int t1 = sq / a.bxN;
            s_sbf[tid] = t1 / a.byN; s_sby[tid] = 4 * TH * (a.by0 + t1 % a.byN); s_sbx[tid] = 4 * TW * (a.bx0 + bx);
        } else {
            s_sbf[tid] = -1; s_sby[tid] = 0; s_sbx[tid] = 0;
        }
    }
    __syncthreads();

    // ---- patch staging: NPX pixels x 2 float4 slots per K step through raw buffer loads: ONE resource per K step (its base
    // advanced by the scalar unit) + a 32-bit byte offset per slot: no vector address arithmetic in the loop.  Padding
    // pixels carry an out-of-range offset: the buffer unit returns zeros for them.  Slots beyond the patch store into a
    // scratch area behind the image.
    const int sq0 = sp * NSUB;
    const int f0 = (sq0 / a.bxN) / a.byN;                              // first frame this workgroup touches (uniform)
    const float* pin = a.in + (size_t)f0 * H * W * Cin;
    const size_t in_left = ((size_t)(a.frames - f0) * H * W * Cin) * sizeof(float);
    const unsigned in_bytes = (unsigned)(in_left < 0x40000000ull ? in_left : 0x40000000ull);
    constexpr unsigned W43_OOB = 0x7ffffff0u;
    unsigned soff[NST];
    int dst[NST];                                                      // float4 units inside a patch buffer
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int s = tid + k * W4T;
        const int px = s >> 1, c4 = s & 1;
        dst[k] = (NPXS * 8) / 4 + (tid & 15);                           // scratch behind the image
        soff[k] = W43_OOB;
        if (px < NPX) {
            const int q = px / SPX, lp = px - q * SPX;
            // lp -> (row, column) of the patch: the aligned 4-pixel segments first, then the 2-pixel row tails, two rows per
            // group of four, so that the eight lanes of a ds_write_b128 group (= 4 pixels) store to 32 different banks: a
            // segment is 128 contiguous bytes; two tails pair up when their rows are 4 bank slots apart mod 8 (rows r, r + 2
            // for an odd row stride, r, r + 1 for an even one)
            constexpr int NFULL = PH * TW * 4;
            int pr, pc;
            if (lp < NFULL) {
                const int seg = lp >> 2;
                pr = seg / TW;
                pc = 4 * (seg - pr * TW) + (lp & 3);
            } else {
                const int h = lp - NFULL, ri = h >> 1;
                pr = ri;
                if (PWS & 1) {
                    const int w = ri & 3;
                    const int r = (ri & ~3) + (w == 1 ? 2 : w == 2 ? 1 : w);
                    if (r < PH) pr = r;
                }
                pc = 4 * TW + (h & 1);
            }
            const int fq = s_sbf[q];
            const int y = s_sby[q] - 1 + pr, x = s_sbx[q] - 1 + pc;
            dst[k] = (q * SPXS + pr * PWS + 5 * (pc >> 2) + (pc & 3)) * 2 + c4;
            if (fq >= 0 && y >= 0 && y < H && x >= 0 && x < W)
                soff[k] = (unsigned)(((((size_t)(fq - f0) * H + y) * W + x) * Cin + c4 * 4) * sizeof(float));
        }
    }
    f32x4 stage[NST];
    auto stage_load = [&](int cs) {                                    // K step cs: channels 8 cs .. 8 cs + 7
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pin + cs * 8), 0, (int)(in_bytes - cs * 32), 0x00020000);
#pragma unroll
        for (int k = 0; k < NST; ++k) stage[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, soff[k], 0, W43_STAGE_AUX));
    };
    auto stage_store = [&](int buf) {
        f32x4* rb = reinterpret_cast<f32x4*>(s_raw + buf * RAWF);
#pragma unroll
        for (int k = 0; k < NST; ++k) rb[dst[k]] = stage[k];
    };

    // ---- transform lane role: tile = lane >> 1, channel quad = lane & 1
    // ds_read_b128 serves a wave in four NON-contiguous groups of sixteen lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, the same
    // + 32: MI355X_MICROARCH.md, LDS): in terms of the tile a lane pair works on, a group holds the tiles whose bits 1..3 have
    // even (odd) parity.  With 8 sub-blocks of 2x2 tiles (sub-block stride = 4 bank slots mod 16) that needs sub-block bit 2,
    // whose stride is 0 mod 16, on tile bit 3: lanes walk the tiles with bits 3 and 4 swapped (the others are conflict-free as
    // numbered).  Measured on conv3_2: conflict cycles of the window reads 169 M -> 0 per launch.
    const int pt_u = lane >> 1, pt_c4 = lane & 1;
    const int pt_tile = (TW == 2 && TH == 2 && NSUB == 8) ? ((pt_u & 7) | ((pt_u & 8) << 1) | ((pt_u & 16) >> 1)) : pt_u;
    const int pt_q = pt_tile / STILE, pt_tl = pt_tile - pt_q * STILE;
    const int pt_tr = pt_tl / TW, pt_tc = pt_tl - pt_tr * TW;
    const int wbase = (pt_q * SPXS + 4 * pt_tr * PWS + 5 * pt_tc) * 8 + pt_c4 * 4;                  // floats
    const int vwbase = pt_tile * 8 + ((pt_c4 ^ ((pt_tile >> 3) & 1)) * 4);                          // floats, + plane * 256
    // ---- MFMA lane role
    const int mrow = lane & 31, kh = lane >> 5;
    const int vabase = wave * 9 * 256 + mrow * 8 + ((kh ^ ((mrow >> 3) & 1)) * 4);                   // floats, + j * 256
    const int n8 = Cin / 8;

    // planes 0..7 (accumulation registers), plane 8 (VGPRs, w43_mfma_v).  Work balance: the two waves with the heavy transform
    // tasks (rows (1,2), (3,4): +90 VALU, +12 LDS instructions per K step) hand the second column half of their last plane
    // (4 MFMAs = 256 cycles per K step) to the wave two above them, which accumulates it in accx.
    f32x16 acc[8][2], accv[2], accx;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][nb][r] = 0.f;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) accv[nb][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) accx[r] = 0.f;

    // ---- K loop, specialised by the wave's transform task (ROLE = wave: 0 plane rows (1,2), 1 rows (3,4), 2 row 0, 3 row 5;
    // the pairs share their row pass).  The dispatch is hoisted out of the loop: four straight-line loop bodies.
    const int n8m1 = n8 - 1;
#ifdef NTK_CL_PROF
    const bool prof_on = blockIdx.x == 1000 && lane == 0;
    unsigned long long prof_acc[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, prof_last = __builtin_amdgcn_s_memtime();
#endif
    auto k_loop = [&](auto role_c) {
        constexpr int ROLE = decltype(role_c)::value;
        constexpr int NR = ROLE <= 1 ? 4 : 3;                       // window rows the task reads
        f32x4 D[4], R1[6], R2[6];
        // window element (aa, b) of this lane's tile: patch row 4 tr + aa, column slot 5 tc + (0,1,2,3,5,6)[b]
        auto tr_reads = [&](const float* rp, int b) {
#pragma unroll
            for (int s = 0; s < NR; ++s) {
                const int aa = ROLE <= 1 ? 1 + s : (ROLE == 2 ? 2 * s : 1 + 2 * s);
                D[s] = *reinterpret_cast<const f32x4*>(rp + wbase + (aa * PWS + 5 * (b >> 2) + (b & 3)) * 8);
            }
        };
        auto tr_rows = [&](int b) {
            if constexpr (ROLE == 0) {                          // rows 1, 2:  (d4 - 4 d2) +- (d3 - 4 d1)
                const f32x4 t1 = D[3] - 4.f * D[1], t2 = D[2] - 4.f * D[0];
                R1[b] = t1 + t2; R2[b] = t1 - t2;
            } else if constexpr (ROLE == 1) {                   // rows 3, 4:  (d4 - d2) +- 2 (d3 - d1)
                const f32x4 t3 = D[3] - D[1], u = D[2] - D[0];
                R1[b] = t3 + 2.f * u; R2[b] = t3 - 2.f * u;
            } else if constexpr (ROLE == 2) {
                R1[b] = w43_r0(D[0], D[1], D[2]);
            } else {
                R1[b] = w43_r5(D[0], D[1], D[2]);
            }
        };
        // column pass of one row-transformed row + store of its six planes (plane row pi)
        auto tr_cols = [&](float* vb, int pi, const f32x4 (&R)[6]) {
            float* o = vb + pi * 6 * 256 + vwbase;
            *reinterpret_cast<f32x4*>(o) = w43_r0(R[0], R[2], R[4]);
            const f32x4 t1 = R[4] - 4.f * R[2], t2 = R[3] - 4.f * R[1];
            *reinterpret_cast<f32x4*>(o + 256) = t1 + t2;
            *reinterpret_cast<f32x4*>(o + 2 * 256) = t1 - t2;
            const f32x4 t3 = R[4] - R[2], u = R[3] - R[1];
            *reinterpret_cast<f32x4*>(o + 3 * 256) = t3 + 2.f * u;
            *reinterpret_cast<f32x4*>(o + 4 * 256) = t3 - 2.f * u;
            *reinterpret_cast<f32x4*>(o + 5 * 256) = w43_r5(R[1], R[3], R[5]);
        };
        auto tr_finish = [&](float* vb) {
            if constexpr (ROLE == 0) { tr_cols(vb, 1, R1); tr_cols(vb, 2, R2); }
            else if constexpr (ROLE == 1) { tr_cols(vb, 3, R1); tr_cols(vb, 4, R2); }
            else if constexpr (ROLE == 2) tr_cols(vb, 0, R1);
            else tr_cols(vb, 5, R1);
        };

        // prologue: patches 0 and 1 -> raw[0], raw[1]; patch 0 transformed -> V[0]; U of the first plane group requested
        W43_STAMP(16);
        {
            // both patches are requested before either is stored: one HBM round trip, not two
            f32x4 stage1[NST];
            const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pin + (n8 > 1 ? 8 : 0)), 0,
                                                                                   (int)(in_bytes - (n8 > 1 ? 32 : 0)), 0x00020000);
            stage_load(0);
#pragma unroll
            for (int k = 0; k < NST; ++k) stage1[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, soff[k], 0, W43_STAGE_AUX));
            stage_store(0);
            f32x4* rb1 = reinterpret_cast<f32x4*>(s_raw + RAWF);
#pragma unroll
            for (int k = 0; k < NST; ++k) rb1[dst[k]] = stage1[k];
        }
        __syncthreads();
        W43_STAMP(17);
#pragma unroll
        for (int b = 0; b < 6; ++b) { tr_reads(s_raw, b); tr_rows(b); }
        tr_finish(s_V);
        // B operand: two register sets alternating by plane-group parity (the K loop is unrolled by two so that the parity is
        // static); group G + 1 is requested when group G starts.  vmcnt retires in order, so the request order matters: the
        // U request precedes the K step's patch requests (HBM latency), which then only have to be back two groups later,
        // when they are stored.  A operand: ONE set, re-read for the next group right after a group's last MFMA has issued.
        constexpr bool DONOR = ROLE <= 1;                               // gives (plane 8, half 1) to wave ROLE + 2
        // The A sets alternate by K-step parity: the LAST twelve MFMAs of a K step (group 2, K pairs 2-3) are issued at the top
        // of the NEXT one, right after the barrier and after that step's first A operands have been requested -- their LDS
        // latency hides behind MFMAs whose operands are already in registers.
        f32x4 Bq[2][3][2], As[2][3], Bx[2], Axs[2];
        Bx[0] = Bx[1] = Axs[0] = Axs[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        // MFMAs of plane group g, K pairs 2 half, 2 half + 1, operands from A set `as` and B set `bs`
        auto mfma_half = [&](auto g_c, auto half_c, auto as_c, auto bs_c) {
            constexpr int g = decltype(g_c)::value, half = decltype(half_c)::value, as = decltype(as_c)::value, bs = decltype(bs_c)::value;
#pragma unroll
            for (int q = 2 * half; q < 2 * half + 2; ++q)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    if (3 * g + pl < 8) {
                        acc[3 * g + pl][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(As[as][pl][q], Bq[bs][pl][0][q], acc[3 * g + pl][0], 0, 0, 0);
                        acc[3 * g + pl][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(As[as][pl][q], Bq[bs][pl][1][q], acc[3 * g + pl][1], 0, 0, 0);
                    } else {
                        w43_mfma_v(accv[0], As[as][pl][q], Bq[bs][pl][0][q]);
                        if constexpr (!DONOR) {
                            w43_mfma_v(accv[1], As[as][pl][q], Bq[bs][pl][1][q]);
                            w43_mfma_v(accx, Axs[as][q], Bx[bs][q]);
                        }
                    }
                }
        };
        const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.U + (size_t)cb * n8 * 18432), 0, (int)((size_t)n8 * 18432 * sizeof(float)), 0x00020000);
        const unsigned ulane = (unsigned)lane * 16u + (unsigned)ROLE * 18432u;          // bytes: + this wave's 4608 floats
        const unsigned uxlane = (unsigned)lane * 16u + (unsigned)(ROLE & 1) * 18432u + 17u * 1024u;   // the donated unit
        auto load_B = [&](int c, int g, int set) {                        // K step c, plane group g
            const int sbase = c * 73728 + g * 6144;                       // bytes: 18432 floats per K step, 3 planes x 2 x 256 per group
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                Bq[set][pl][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, ulane + (pl * 2) * 1024, sbase, 0));
                if (!(DONOR && g == 2 && pl == 2))
                    Bq[set][pl][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, ulane + (pl * 2 + 1) * 1024, sbase, 0));
            }
            if (!DONOR && g == 2) Bx[set] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, uxlane, c * 73728, 0));
        };
        const int vxbase = (9 * (ROLE & 1) + 8) * 256 + mrow * 8 + ((kh ^ ((mrow >> 3) & 1)) * 4);      // the donated plane's A operand
        load_B(0, 0, 0);
        W43_STAMP(18);

        auto k_step = [&](int c8, auto par_c) {
            constexpr int PAR = decltype(par_c)::value;            // c8 & 1
            const int cn = (c8 + 1 < n8) ? c8 + 1 : 0;             // next K step (wraps harmlessly on the last iteration)
            const int cs = (c8 + 2 < n8) ? c8 + 2 : n8m1;          // K step staged now
            W43_STAMP(15);
            if constexpr (!(W43_ABL & 8)) __syncthreads();         // V[PAR] and raw[PAR ^ 1] are complete
            W43_STAMP(0);
            const float* vcur = s_V + PAR * VF + vabase;
            const float* rnext = s_raw + (PAR ^ 1) * RAWF;
            float* vnext = s_V + (PAR ^ 1) * VF;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) if constexpr (!(W43_ABL & 16)) As[PAR][pl] = *reinterpret_cast<const f32x4*>(vcur + pl * 256);
            if constexpr (!(W43_ABL & 2)) tr_reads(rnext, 0);
            // the previous K step's last twelve MFMAs (its group 2 sits in A set PAR ^ 1 and B set PAR ^ 1)
            if (PAR == 1 || c8 > 0) {
                __builtin_amdgcn_sched_barrier(0);
                mfma_half(ic<2>{}, ic<1>{}, ic<PAR ^ 1>{}, ic<PAR ^ 1>{});
                __builtin_amdgcn_sched_barrier(0);
            }
            // A set PAR ^ 1 is free now: group 1's operands are requested a whole group ahead (the groups alternate between
            // the two sets: 0 -> PAR, 1 -> PAR ^ 1, 2 -> PAR, where the next K step's deferred MFMAs expect them)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) if constexpr (!(W43_ABL & 16)) As[PAR ^ 1][pl] = *reinterpret_cast<const f32x4*>(vcur + (3 + pl) * 256);
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const int bs = (PAR + g) & 1;
                if constexpr (!(W43_ABL & 1)) { if (g < 2) load_B(c8, g + 1, bs ^ 1); else load_B(cn, 0, bs ^ 1); }
                if constexpr (!(W43_ABL & 4)) if (g == 0) stage_load(cs);
                W43_STAMP(1);
                // a plane group's 24 MFMAs in two halves (K pairs 0-1, 2-3) with one of the gap's two transform columns
                // after each: the window reads of a column get 12 MFMAs (768 cycles) of cover -- with one wave per SIMD a
                // wait is an idle SIMD
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(g == 2 && half == 1)) {                   // (2, 1) is issued at the top of the next K step
                        if (g == 0 && half == 0) mfma_half(ic<0>{}, ic<0>{}, ic<PAR>{}, ic<PAR>{});
                        if (g == 0 && half == 1) mfma_half(ic<0>{}, ic<1>{}, ic<PAR>{}, ic<PAR>{});
                        if (g == 1 && half == 0) mfma_half(ic<1>{}, ic<0>{}, ic<PAR ^ 1>{}, ic<PAR ^ 1>{});
                        if (g == 1 && half == 1) mfma_half(ic<1>{}, ic<1>{}, ic<PAR ^ 1>{}, ic<PAR ^ 1>{});
                        if (g == 2 && half == 0) mfma_half(ic<2>{}, ic<0>{}, ic<PAR>{}, ic<PAR>{});
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    W43_STAMP(2 + 4 * g + 2 * half);
                    if (half == 0) {
                        if constexpr (!(W43_ABL & 2)) {
                            tr_rows(2 * g);
                            tr_reads(rnext, 2 * g + 1);
                        }
                        W43_STAMP(3 + 4 * g);
                    } else {
                        if (g == 0) {                              // group 0 has issued: its set takes group 2's operands
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl) if constexpr (!(W43_ABL & 16)) As[PAR][pl] = *reinterpret_cast<const f32x4*>(vcur + (6 + pl) * 256);
                            if constexpr (!DONOR && !(W43_ABL & 16)) Axs[PAR] = *reinterpret_cast<const f32x4*>(s_V + PAR * VF + vxbase);
                        } else if (g == 2) {
                            if constexpr (!(W43_ABL & 4)) stage_store(PAR);
                        }
                        if constexpr (!(W43_ABL & 2)) {
                            tr_rows(2 * g + 1);
                            if (g < 2) tr_reads(rnext, 2 * g + 2);
                            else tr_finish(vnext);
                        }
                        W43_STAMP(5 + 4 * g);
                    }
                }
            }
        };
        for (int c8 = 0; c8 < n8; c8 += 2) {
            k_step(c8, ic<0>{});
            k_step(c8 + 1, ic<1>{});
        }
        mfma_half(ic<2>{}, ic<1>{}, ic<1>{}, ic<1>{});             // the last K step's (odd parity: n8 is even) deferred MFMAs
    };
    if (wave == 0) k_loop(ic<0>{});
    else if (wave == 1) k_loop(ic<1>{});
    else if (wave == 2) k_loop(ic<2>{});
    else k_loop(ic<3>{});

#ifdef NTK_CL_PROF
    W43_STAMP(15);
#endif
    // ---- epilogue: Y = A^T M A, A^T = [[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]]
    float* sZ = s_mem;                                   // [36][32 tiles][32 channels]
    const int col = lane & 31;
    // the asm-form MFMAs (the last instructions of the K loop) have written their results before anything reads them: a
    // 16-pass MFMA needs 18 passes of distance to a dependent VALU / LDS instruction
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {                      // unrolled: the accumulator selection below is static
        __syncthreads();
        W43_STAMP(19);
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            if (j == 8 && nb == 1 && wave < 2) continue;            // that unit was accumulated by wave + 2 (accx)
            const f32x16 v = j < 8 ? (nb == 0 ? acc[j][0] : acc[j][1]) : (nb == 0 ? accv[0] : accv[1]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 4 * kh + (r & 3) + 8 * (r >> 2);
                sZ[((9 * wave + j) * 32 + m) * 32 + col] = v[r];
            }
        }
        if (nb == 1 && wave >= 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 4 * kh + (r & 3) + 8 * (r >> 2);
                sZ[((9 * (wave - 2) + 8) * 32 + m) * 32 + col] = accx[r];
            }
        }
        W43_STAMP(20);
        __syncthreads();
        W43_STAMP(21);
        // thread = (tile m, four adjacent output channels): 36 conflict-free ds_read_b128, one float4 store per pixel
        const int m = tid >> 3, cq = tid & 7;
        const int n = 64 * cb + 32 * nb + 4 * cq;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + n);
        const int mq = m / STILE, ml = m - mq * STILE;
        const int f = s_sbf[mq];
        if (f >= 0) {
            const f32x4* zp = reinterpret_cast<const f32x4*>(sZ + m * 32 + 4 * cq);      // + plane * 256 float4
            // column part: z[k][j] = sum_i AT[k][i] M[i][j]
            f32x4 z[4][6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                f32x4 mm[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) mm[i] = zp[(i * 6 + j) * 256];
                const f32x4 s12 = mm[1] + mm[2], d12 = mm[1] - mm[2], s34 = mm[3] + mm[4], d34 = mm[3] - mm[4];
                z[0][j] = mm[0] + s12 + s34;
                z[1][j] = d12 + 2.f * d34;
                z[2][j] = s12 + 4.f * s34;
                z[3][j] = d12 + 8.f * d34 + mm[5];
            }
            f32x4 y[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 s12 = z[k][1] + z[k][2], d12 = z[k][1] - z[k][2], s34 = z[k][3] + z[k][4], d34 = z[k][3] - z[k][4];
                y[k][0] = z[k][0] + s12 + s34;
                y[k][1] = d12 + 2.f * d34;
                y[k][2] = s12 + 4.f * s34;
                y[k][3] = d12 + 8.f * d34 + z[k][5];
            }
            const int tr = ml / TW, tc = ml - tr * TW;
            const int oy = s_sby[mq] + 4 * tr, ox = s_sbx[mq] + 4 * tc;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            if constexpr (POOL) {
                float* op = a.out + (((size_t)f * (H >> 1) + (oy >> 1)) * (W >> 1) + (ox >> 1)) * Cout + n;
#pragma unroll
                for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            v[e] = fmaxf(fmaxf(y[2 * aa][2 * b][e], y[2 * aa][2 * b + 1][e]), fmaxf(y[2 * aa + 1][2 * b][e], y[2 * aa + 1][2 * b + 1][e]));
                        v = v + bv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], zero[e]);
                        W43_STORE(reinterpret_cast<f32x4*>(op + ((size_t)aa * (W >> 1) + b) * Cout), v);
                    }
            } else {
                float* op = a.out + (((size_t)f * H + oy) * W + ox) * Cout + n;
#pragma unroll
                for (int aa = 0; aa < 4; ++aa)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        f32x4 v = y[aa][b] + bv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], zero[e]);
                        W43_STORE(reinterpret_cast<f32x4*>(op + ((size_t)aa * W + b) * Cout), v);
                    }
            }
        }
        W43_STAMP(22);
    }
#ifdef NTK_CL_PROF
    if (prof_on)
        for (int i = 0; i < 24; ++i) g_w43_prof[wave][i] = prof_acc[i];
#endif
}


// ---------------------------------------------------------------------------------------------------------------------------
// The same block (32 tiles x 64 output channels x 36 planes) on EIGHT waves, two per SIMD, 256 registers each.
//
// With one wave per SIMD every wait of that wave (LDS latency of the A / window reads, the U loads, the barrier) is an idle
// matrix pipe.  Here a SIMD's nine planes x two 32-channel halves = 18 accumulator units are split UNEVENLY between its two
// waves, and the rest of the work the other way round:
//   waves 0..3 ("T"): 5 or 7 units (20 / 28 MFMAs per K step); the patch staging (global -> registers -> LDS) and the input
//                     transform of the next K step (task = wave: plane rows (1,2), (3,4) -- 5 units --, 0, 5 -- 7 units),
//                     done FIRST in a K step;
//   waves 4..7 ("S"): 12 units (48 MFMAs per K step) and nothing else: the matrix pipe has their work while the T wave of the
//                     same SIMD waits for its window reads and runs its row / column passes.
// (An even 9 / 9 split was measured first: the T wave's transform then takes longer than the S wave's 36 MFMAs, the S waves
// spend a third of the K step at the barrier and the T wave's MFMAs run alone: no faster than four waves.)
// LDS, the packed weights, the arithmetic and its order are those of conv3x3_wino43_kernel: results are bit-identical.

// Packed fp32 forms of the transform's arithmetic with the negations folded into the instruction.  Left to itself the compiler
// splits a vector subtraction into scalar v_sub_f32 and materialises negations with v_xor_b32: 164 VALU instructions per K
// step of a pair task where 96 packed ones do.  Beside another wave's fp32 MFMA stream every VALU instruction waits for the
// MFMA in flight, so the COUNT is what the transform costs (the results are the same bits: the same fused operations).
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define W43_PK(name, text)                                                                    \
    __device__ __forceinline__ f32x4 name(f32x4 a, f32x4 b) {                                  \
        f32x2 lo, hi;                                                                         \
        asm(text : "=v"(lo) : "v"(a.xy), "v"(b.xy));                                           \
        asm(text : "=v"(hi) : "v"(a.zw), "v"(b.zw));                                           \
        return f32x4{lo.x, lo.y, hi.x, hi.y};                                                 \
    }
W43_PK(w43_add, "v_pk_add_f32 %0, %1, %2")                                                     // a + b
W43_PK(w43_sub, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]")                           // a - b
W43_PK(w43_fma4, "v_pk_fma_f32 %0, %1, 4.0, %2 op_sel_hi:[1,0,1]")                             // 4 a + b
W43_PK(w43_fnma4, "v_pk_fma_f32 %0, %1, 4.0, %2 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]")   // b - 4 a
W43_PK(w43_fma2, "v_pk_fma_f32 %0, %1, 2.0, %2 op_sel_hi:[1,0,1]")                             // 2 a + b
W43_PK(w43_fnma2, "v_pk_fma_f32 %0, %1, 2.0, %2 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]")   // b - 2 a
#undef W43_PK
__device__ __forceinline__ f32x4 w43_fnma5(f32x4 a, f32x4 b) {                                 // b - 5 a
    f32x2 lo, hi;
    const f32x2 k5 = {5.f, 5.f};
    asm("v_pk_fma_f32 %0, %1, %3, %2 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(lo) : "v"(a.xy), "v"(b.xy), "s"(k5));
    asm("v_pk_fma_f32 %0, %1, %3, %2 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(hi) : "v"(a.zw), "v"(b.zw), "s"(k5));
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ f32x4 w43p_r0(f32x4 d0, f32x4 d2, f32x4 d4) { return w43_fma4(d0, w43_fnma5(d2, d4)); }     // 4 d0 + (d4 - 5 d2)

constexpr int W4D = 512;
#ifdef NTK_CL_PROF
__device__ unsigned long long g_w43d_prof[8][12];
#endif

// INB / OUTB: the activation layout on the input / output side.  false = NHWC; true = CHANNEL-BLOCKED per image row,
// [H][C/8][W][8] per frame: the eight channels of a K step are one contiguous 32-byte piece per pixel and a patch row is one
// contiguous run, so a wave's staging load covers 1 KB of whole cache lines instead of thirty-two 32-byte pieces 4 * Cin bytes
// apart (the staging was 9 % of the kernel, ablation bit 5 its upper bound).  Same arithmetic, same bits; only addresses change.
// The layers of a trunk hand blocked maps to each other (ntk_vgg_conv3x3_relu_wino43_layout_f32); the first reads conv1_1's NHWC
// map, the last one writes NHWC for gather_serialize.  (Measured on 640 frames, nine layers: NHWC 45.98 ms; whole planes per
// channel block, [C/8][H][W][8]: 45.27 of 47.27 on a slower box, -4.2 %; blocks interleaved per image row, this form: 43.56, -5.3 %.
// conv1_1 itself stays NHWC: its store-bound row kernel is 0.45 - 0.7 ms slower writing eight 1 KB runs per wave.)
template <bool POOL, int TW, int TH, int NSUB, int PWS, int SPXS, bool INB = false, bool OUTB = false>
__global__ __launch_bounds__(W4D) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_wino43d_kernel(Wino43Args a) {
    constexpr int STILE = TW * TH, NTILE = NSUB * STILE, PW = 4 * TW + 2, PH = 4 * TH + 2, SPX = PW * PH, NPX = NSUB * SPX;
    constexpr int NPXS = NSUB * SPXS;
    constexpr int NST = (NPX * 2 + 255) / 256;                       // float4 staging slots per thread of the four S waves
    // many slots per thread (the 1x1x32 shape: 9): a slot's source offset and LDS destination share ONE register -- destination
    // (float4 units, < 4096) in the low 12 bits, source offset / 16 above it (the host checks that a block's input span is below 16 MB)
    constexpr bool PACK = NST > 7;
    constexpr int RAWF = NPXS * 8 + 64;
    constexpr int VF = 36 * 32 * 8;
    constexpr int ZF = 36 * 32 * 32;
    constexpr int LDSF = (2 * RAWF + 2 * VF) > ZF ? (2 * RAWF + 2 * VF) : ZF;
    static_assert(NTILE == 32, "tile block = 32 MFMA rows");
    static_assert(LDSF * 4 + 512 <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) float s_mem[LDSF];
    __shared__ int s_sbf[NSUB], s_sby[NSUB], s_sbx[NSUB];
    float* s_raw = s_mem;                    // [2][RAWF]
    float* s_V = s_mem + 2 * RAWF;           // [2][VF]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave & 3;                                           // plane group (= SIMD)
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    int cb, sp;
    if (a.nCB >= 8) {
        const int kN = a.nCB >> 3;
        cb = (slot % kN) * 8 + xcd;
        sp = slot / kN;
    } else {
        const int per = 8 / a.nCB;
        cb = xcd % a.nCB;
        sp = slot * per + xcd / a.nCB;
    }
    if (sp >= a.NS) return;
    const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
    if (tid < NSUB) {
        const int sq = sp * NSUB + tid;
        if (sq < a.NQ) {
            const int bx = sq % a.bxN;
            const int t1 = sq / a.bxN;
            s_sbf[tid] = t1 / a.byN; s_sby[tid] = 4 * TH * (a.by0 + t1 % a.byN); s_sbx[tid] = 4 * TW * (a.bx0 + bx);
        } else {
            s_sbf[tid] = -1; s_sby[tid] = 0; s_sbx[tid] = 0;
        }
    }
    __syncthreads();

    const int sq0 = sp * NSUB;
    const int f0 = (sq0 / a.bxN) / a.byN;
    const float* pin = a.in + (size_t)f0 * H * W * Cin;
    const size_t in_left = ((size_t)(a.frames - f0) * H * W * Cin) * sizeof(float);
    const unsigned in_bytes = (unsigned)(in_left < 0x40000000ull ? in_left : 0x40000000ull);
    constexpr unsigned W43_OOB = 0x7ffffff0u;
    const int n8 = Cin / 8, n8m1 = n8 - 1;
#ifdef NTK_CL_PROF
    const bool prof_on = blockIdx.x == 1000 && lane == 0;
    unsigned long long prof_acc[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, prof_last = __builtin_amdgcn_s_memtime();
#endif

    // slot table of a staging thread (stid = tid & 255): as in conv3x3_wino43_kernel
    auto slot_table = [&](unsigned (&soff)[NST], int (&dst)[NST]) {
        const int stid = tid & 255;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int s = stid + k * 256;
            const int px = s >> 1, c4 = s & 1;
            dst[k] = (NPXS * 8) / 4 + (stid & 15);
            soff[k] = W43_OOB;
            if (px < NPX) {
                const int q = px / SPX, lp = px - q * SPX;
                constexpr int NFULL = PH * TW * 4;
                int pr, pc;
                if (lp < NFULL) {
                    const int seg = lp >> 2;
                    pr = seg / TW;
                    pc = 4 * (seg - pr * TW) + (lp & 3);
                } else {
                    const int h = lp - NFULL, ri = h >> 1;
                    pr = ri;
                    if (PWS & 1) {
                        const int w = ri & 3;
                        const int r = (ri & ~3) + (w == 1 ? 2 : w == 2 ? 1 : w);
                        if (r < PH) pr = r;
                    }
                    pc = 4 * TW + (h & 1);
                }
                const int fq = s_sbf[q];
                const int y = s_sby[q] - 1 + pr, x = s_sbx[q] - 1 + pc;
                dst[k] = (q * SPXS + pr * PWS + 5 * (pc >> 2) + (pc & 3)) * 2 + c4;
                if (fq >= 0 && y >= 0 && y < H && x >= 0 && x < W)
                    soff[k] = INB ? (unsigned)((((((size_t)(fq - f0) * H + y) * (Cin >> 3)) * W + x) * 8 + c4 * 4) * sizeof(float))
                                  : (unsigned)(((((size_t)(fq - f0) * H + y) * W + x) * Cin + c4 * 4) * sizeof(float));
            }
            // ablation bit 5 (timing only, results wrong): the slots of a patch read CONTIGUOUS 16-byte pieces -- what a channel-blocked
            // activation layout [C/8][H][W][8] would give the staging loads (an upper bound: also drops the halo overlap between patches)
            if constexpr ((W43_ABL & 32) != 0) soff[k] = (unsigned)(stid + k * 256) * 16u;
            if constexpr (PACK) soff[k] = (soff[k] == W43_OOB ? 0xfffff000u : (soff[k] >> 4) << 12) | (unsigned)dst[k];
        }
    };
    auto slot_src = [&](unsigned w) { return PACK ? ((w >> 8) & ~15u) : w; };
    const unsigned in_window = PACK ? (in_bytes < 0xfffff0u ? in_bytes : 0xfffff0u) : in_bytes;     // the out-of-range marker must stay out of range

    // ---- MFMA lane role (both kinds of wave).  unit = (plane j of this SIMD's nine, 32-channel half nb):
    //   T waves own units (0..4, 0), waves 2, 3 also (5, 0) and unit (5, 0) of waves 0, 1: 5 / 7 x 16 accumulators;
    //   S waves own units (0..8, 1) and (6..8, 0): 12 x 16 accumulators, 48 MFMAs per K step.
    const int mrow = lane & 31, kh = lane >> 5;
    const int vabase = pg * 9 * 256 + mrow * 8 + ((kh ^ ((mrow >> 3) & 1)) * 4);                    // floats, + j * 256
    f32x16 acc[12];
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.U + (size_t)cb * n8 * 18432), 0, (int)((size_t)n8 * 18432 * sizeof(float)), 0x00020000);
    const unsigned ubase = (unsigned)lane * 16u + (unsigned)pg * 18432u;                            // bytes, + (2 j + nb) * 1024
    auto load_unit = [&](int c, int j, int nb) {
        if constexpr ((W43_ABL & 1) != 0) return f32x4{1.f, 1.f, 1.f, 1.f};
        else return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, ubase + (unsigned)(2 * j + nb) * 1024u, c * 73728, 0));
    };
    auto load_plane = [&](const float* vcur, int j) {
        if constexpr ((W43_ABL & 16) != 0) return f32x4{1.f, 1.f, 1.f, 1.f};
        else return *reinterpret_cast<const f32x4*>(vcur + j * 256);
    };
    // ---- epilogue pieces (called from the role branches: a branch then carries only its own accumulators through them).  One
    // 32-channel half at a time through LDS: half 0 sits in the T waves (planes 0..4 / 0..5 and an adopted one) and the S waves
    // (planes 6..8) and is transformed by the T waves, whose accumulators are dead by then (the S waves still hold half 1);
    // half 1 by the S waves.  thread = (tile, four adjacent channels)
    float* sZ = s_mem;
    const int col = lane & 31;
    auto z_store = [&](const f32x16& v, int plane) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = 4 * kh + (r & 3) + 8 * (r >> 2);
            sZ[((9 * pg + plane) * 32 + m) * 32 + col] = v[r];
        }
    };
    auto out_transform = [&](auto nb_c) {
        constexpr int nb = decltype(nb_c)::value;
        const int et = tid & 255;
        const int m = et >> 3, cq = et & 7;
        const int n = 64 * cb + 32 * nb + 4 * cq;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + n);
        const int mq = m / STILE, ml = m - mq * STILE;
        const int f = s_sbf[mq];
        if (f >= 0) {
            const f32x4* zp = reinterpret_cast<const f32x4*>(sZ + m * 32 + 4 * cq);
            f32x4 z[4][6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                f32x4 mm[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) mm[i] = zp[(i * 6 + j) * 256];
                const f32x4 s12 = mm[1] + mm[2], d12 = mm[1] - mm[2], s34 = mm[3] + mm[4], d34 = mm[3] - mm[4];
                z[0][j] = mm[0] + s12 + s34;
                z[1][j] = d12 + 2.f * d34;
                z[2][j] = s12 + 4.f * s34;
                z[3][j] = d12 + 8.f * d34 + mm[5];
            }
            f32x4 y[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 s12 = z[k][1] + z[k][2], d12 = z[k][1] - z[k][2], s34 = z[k][3] + z[k][4], d34 = z[k][3] - z[k][4];
                y[k][0] = z[k][0] + s12 + s34;
                y[k][1] = d12 + 2.f * d34;
                y[k][2] = s12 + 4.f * s34;
                y[k][3] = d12 + 8.f * d34 + z[k][5];
            }
            const int tr = ml / TW, tc = ml - tr * TW;
            const int oy = s_sby[mq] + 4 * tr, ox = s_sbx[mq] + 4 * tc;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            // output addressing: NHWC [f][y][x][Cout], or channel-blocked [f][Cout / 8][y][x][8] (n is a multiple of 4)
            const int Ho = POOL ? (H >> 1) : H, Wo = POOL ? (W >> 1) : W;
            const size_t xstep = OUTB ? 8 : (size_t)Cout, ystep = (size_t)Cout * Wo;
            float* const obase = OUTB ? a.out + (size_t)f * Cout * Ho * Wo + (size_t)(n >> 3) * Wo * 8 + (n & 7)
                                      : a.out + (size_t)f * Cout * Ho * Wo + n;
            if constexpr (POOL) {
                float* op = obase + (size_t)(oy >> 1) * ystep + (size_t)(ox >> 1) * xstep;
#pragma unroll
                for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            v[e] = fmaxf(fmaxf(y[2 * aa][2 * b][e], y[2 * aa][2 * b + 1][e]), fmaxf(y[2 * aa + 1][2 * b][e], y[2 * aa + 1][2 * b + 1][e]));
                        v = v + bv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], zero[e]);
                        // blocked output: a lane pair writes 32 bytes of a 128-byte line that the next three stores complete --
                        // plain stores (L2 merges the line); non-temporal ones would go out as four partial writes
                        if constexpr (OUTB) *reinterpret_cast<f32x4*>(op + (size_t)aa * ystep + (size_t)b * xstep) = v;
                        else W43_STORE(reinterpret_cast<f32x4*>(op + (size_t)aa * ystep + (size_t)b * xstep), v);
                    }
            } else {
                float* op = obase + (size_t)oy * ystep + (size_t)ox * xstep;
#pragma unroll
                for (int aa = 0; aa < 4; ++aa)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        f32x4 v = y[aa][b] + bv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], zero[e]);
                        if constexpr (OUTB) *reinterpret_cast<f32x4*>(op + (size_t)aa * ystep + (size_t)b * xstep) = v;
                        else W43_STORE(reinterpret_cast<f32x4*>(op + (size_t)aa * ystep + (size_t)b * xstep), v);
                    }
            }
        }
    };
#define W43D_MFMA(C, A, B) C = __builtin_amdgcn_mfma_f32_32x32x2f32(A, B, C, 0, 0, 0)

    if (wave < 4) {
        // ================== T waves: staging, the transform of the next K step (first), then 20 / 28 MFMAs ==================
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const int pt_u = lane >> 1, pt_c4 = lane & 1;
        const int pt_tile = (TW == 2 && TH == 2 && NSUB == 8) ? ((pt_u & 7) | ((pt_u & 8) << 1) | ((pt_u & 16) >> 1)) : pt_u;
        const int pt_q = pt_tile / STILE, pt_tl = pt_tile - pt_q * STILE;
        const int pt_tr = pt_tl / TW, pt_tc = pt_tl - pt_tr * TW;
        const int wbase = (pt_q * SPXS + 4 * pt_tr * PWS + 5 * pt_tc) * 8 + pt_c4 * 4;
        const int vwbase = pt_tile * 8 + ((pt_c4 ^ ((pt_tile >> 3) & 1)) * 4);
        unsigned soff[NST]; int dst[NST];
        slot_table(soff, dst);
        f32x4 stage[NST];
        const int kstep_floats = INB ? W * 8 : 8;                    // one K step further: the next channel block of the image row / the next 8 channels of a pixel
        auto stage_load = [&](int cs) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pin + (size_t)cs * kstep_floats), 0,
                                                                                  (int)(in_window - (unsigned)cs * (unsigned)kstep_floats * 4u), 0x00020000);
#pragma unroll
            for (int k = 0; k < NST; ++k) stage[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, slot_src(soff[k]), 0, W43_STAGE_AUX));
        };
        auto stage_store = [&](int buf) {
            f32x4* rb = reinterpret_cast<f32x4*>(s_raw + buf * RAWF);
#pragma unroll
            for (int k = 0; k < NST; ++k) rb[PACK ? (int)(soff[k] & 4095u) : dst[k]] = stage[k];
        };
        {   // prologue: both patches are requested before either is stored: one HBM round trip
            f32x4 stage1[NST];
            const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pin + (n8 > 1 ? kstep_floats : 0)), 0,
                                                                                   (int)(in_window - (n8 > 1 ? (unsigned)kstep_floats * 4u : 0u)), 0x00020000);
            stage_load(0);
#pragma unroll
            for (int k = 0; k < NST; ++k) stage1[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, slot_src(soff[k]), 0, W43_STAGE_AUX));
            stage_store(0);
            f32x4* rb1 = reinterpret_cast<f32x4*>(s_raw + RAWF);
#pragma unroll
            for (int k = 0; k < NST; ++k) rb1[PACK ? (int)(soff[k] & 4095u) : dst[k]] = stage1[k];
        }
        auto t_loop = [&](auto role_c) {
            constexpr int ROLE = decltype(role_c)::value;
            constexpr int NR = ROLE <= 1 ? 4 : 3;
            f32x4 R1[6], R2[6];
            auto tr_reads = [&](f32x4 (&D)[4], const float* rp, int b) {
#pragma unroll
                for (int s = 0; s < NR; ++s) {
                    const int aa = ROLE <= 1 ? 1 + s : (ROLE == 2 ? 2 * s : 1 + 2 * s);
                    D[s] = *reinterpret_cast<const f32x4*>(rp + wbase + (aa * PWS + 5 * (b >> 2) + (b & 3)) * 8);
                }
            };
            auto tr_rows = [&](const f32x4 (&D)[4], int b) {
                if constexpr (ROLE == 0) {                          // rows 1, 2:  (d4 - 4 d2) +- (d3 - 4 d1)
                    const f32x4 t1 = w43_fnma4(D[1], D[3]), t2 = w43_fnma4(D[0], D[2]);
                    R1[b] = w43_add(t1, t2); R2[b] = w43_sub(t1, t2);
                } else if constexpr (ROLE == 1) {                   // rows 3, 4:  (d4 - d2) +- 2 (d3 - d1)
                    const f32x4 t3 = w43_sub(D[3], D[1]), u = w43_sub(D[2], D[0]);
                    R1[b] = w43_fma2(u, t3); R2[b] = w43_fnma2(u, t3);
                } else {                                            // row 0 (5): 4 d0 + (d4 - 5 d2)  (4 d1 + (d5 - 5 d3))
                    R1[b] = w43p_r0(D[0], D[1], D[2]);
                }
            };
            auto tr_cols = [&](float* vb, int pi, const f32x4 (&R)[6]) {
                float* o = vb + pi * 6 * 256 + vwbase;
                *reinterpret_cast<f32x4*>(o) = w43p_r0(R[0], R[2], R[4]);
                const f32x4 t1 = w43_fnma4(R[2], R[4]), t2 = w43_fnma4(R[1], R[3]);
                *reinterpret_cast<f32x4*>(o + 256) = w43_add(t1, t2);
                *reinterpret_cast<f32x4*>(o + 2 * 256) = w43_sub(t1, t2);
                const f32x4 t3 = w43_sub(R[4], R[2]), u = w43_sub(R[3], R[1]);
                *reinterpret_cast<f32x4*>(o + 3 * 256) = w43_fma2(u, t3);
                *reinterpret_cast<f32x4*>(o + 4 * 256) = w43_fnma2(u, t3);
                *reinterpret_cast<f32x4*>(o + 5 * 256) = w43p_r0(R[1], R[3], R[5]);
            };
            // the window reads of column b + 1 are in flight while column b's row pass runs (two D sets)
            f32x4 D[2][4];
            auto tr_col_pair = [&](const float* rp, int b0, bool more) {      // row pass of columns b0, b0 + 1; requests column b0 + 2
                tr_reads(D[1], rp, b0 + 1);
                tr_rows(D[0], b0);
                if (more) tr_reads(D[0], rp, b0 + 2);
                tr_rows(D[1], b0 + 1);
            };
            auto tr_finish = [&](float* vb) {
                if constexpr (ROLE == 0) { tr_cols(vb, 1, R1); tr_cols(vb, 2, R2); }
                else if constexpr (ROLE == 1) { tr_cols(vb, 3, R1); tr_cols(vb, 4, R2); }
                else if constexpr (ROLE == 2) tr_cols(vb, 0, R1);
                else tr_cols(vb, 5, R1);
            };
            __syncthreads();                                         // patches 0 and 1 are in raw[0], raw[1]
            tr_reads(D[0], s_raw, 0);
            tr_col_pair(s_raw, 0, true);
            tr_col_pair(s_raw, 2, true);
            tr_col_pair(s_raw, 4, false);
            tr_finish(s_V);
            // B operand: two sets of three units, set g for plane group g; group 0 of the NEXT K step is requested during group 1.
            // A operand: ONE set; a plane's register is re-read for the next group right after its last MFMA has issued.
            // Order of a K step: the previous step's group 1 (operands already in registers), the transform, group 0.  While the S
            // wave has MFMAs to issue the SIMD alternates strictly -- ONE VALU instruction of this wave per MFMA of the other (s_setprio
            // changes nothing) -- so the transform's first 48 VALU instructions take the S wave's 3 072 MFMA cycles whatever they are,
            // the rest run at full rate, and what decides the K step is how much of this wave's own work is left AFTER that:
            // interleaving its MFMAs with the transform's pieces was measured 1.5 % slower than transform-first.
            // Units: the pair tasks (ROLE 0, 1) issue twice the VALU instructions of the single-row ones, so their waves hand
            // unit (5, 0) to the wave two above (same arithmetic, another owner): 5 units = 20 MFMAs per K step beside 96 VALU
            // instructions, against 7 units = 28 MFMAs beside 48.
            constexpr bool HEAVY = ROLE <= 1;
            constexpr int NG1 = HEAVY ? 2 : 4;                       // units of group 1: planes 3, 4 | planes 3, 4, 5 + the adopted one
            f32x4 Bq0[3], Bq1[NG1], As[NG1 > 3 ? NG1 : 3];
            const int vxbase = (ROLE & 1) * 9 * 256 + 5 * 256 + mrow * 8 + ((kh ^ ((mrow >> 3) & 1)) * 4);    // plane 5 of wave ROLE - 2
            const unsigned uxbase = (unsigned)lane * 16u + (unsigned)(ROLE & 1) * 18432u + 10u * 1024u;        // unit (5, 0) of wave ROLE - 2
            auto load_g1 = [&](int c, const float* vplanes) {          // B of group 1 (vplanes == nullptr) or its A planes
#pragma unroll
                for (int u = 0; u < NG1; ++u) {
                    if (u < 3) {
                        if (vplanes) As[u] = load_plane(vplanes + vabase, 3 + u); else Bq1[u] = load_unit(c, 3 + u, 0);
                    } else {
                        if (vplanes) {
                            if constexpr ((W43_ABL & 16) != 0) As[u] = f32x4{1.f, 1.f, 1.f, 1.f};
                            else As[u] = *reinterpret_cast<const f32x4*>(vplanes + vxbase);
                        } else {
                            if constexpr ((W43_ABL & 1) != 0) Bq1[u] = f32x4{1.f, 1.f, 1.f, 1.f};
                            else Bq1[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, uxbase, c * 73728, 0));
                        }
                    }
                }
            };
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) Bq0[pl] = load_unit(0, pl, 0);
            auto t_step = [&](int c8, auto par_c) {
                constexpr int PAR = decltype(par_c)::value;
                const int cn = (c8 + 1 < n8) ? c8 + 1 : 0;
                const int cs = (c8 + 2 < n8) ? c8 + 2 : n8m1;
                const float* rnext = s_raw + (PAR ^ 1) * RAWF;
                W43_STAMP(0);
                __builtin_amdgcn_sched_barrier(0);                   // (the scheduler moves MFMAs across s_barrier otherwise)
                if constexpr (!(W43_ABL & 8)) __syncthreads();       // V[PAR] and raw[PAR ^ 1] are complete
                __builtin_amdgcn_sched_barrier(0);
                W43_STAMP(1);
                const float* vcur = s_V + PAR * VF + vabase;
                if constexpr (!(W43_ABL & 2)) tr_reads(D[0], rnext, 0);
                // the previous K step's group 1 (its A planes and B units are in registers): the matrix pipe has work while this
                // wave's first window reads and the S wave's first A reads are in flight
                if (PAR == 1 || c8 > 0) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int u = 0; u < NG1; ++u) W43D_MFMA(acc[3 + u], As[u][q], Bq1[u][q]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) As[pl] = load_plane(vcur, pl);
                if constexpr (!(W43_ABL & 4)) stage_load(cs);
                if constexpr (!(W43_ABL & 2)) {
                    tr_col_pair(rnext, 0, true);
                    tr_col_pair(rnext, 2, true);
                    tr_col_pair(rnext, 4, false);
                    tr_finish(s_V + (PAR ^ 1) * VF);
                }
                __builtin_amdgcn_sched_barrier(0);
                W43_STAMP(2);
                load_g1(c8, nullptr);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) W43D_MFMA(acc[pl], As[pl][q], Bq0[pl][q]);
                if constexpr (!(W43_ABL & 4)) stage_store(PAR);      // raw[PAR] was consumed in the previous K step
                load_g1(0, s_V + PAR * VF);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) Bq0[pl] = load_unit(cn, pl, 0);
                W43_STAMP(8);
            };
            W43_STAMP(6);
            for (int c8 = 0; c8 < n8; c8 += 2) {
                t_step(c8, ic<0>{});
                t_step(c8 + 1, ic<1>{});
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int u = 0; u < NG1; ++u) W43D_MFMA(acc[3 + u], As[u][q], Bq1[u][q]);
            // ---- epilogue, T side (four barriers, as on the S side)
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                                         // (E0) every wave has left the K loop: the Z image may overwrite raw / V
            W43_STAMP(4);
#pragma unroll
            for (int j = 0; j < 5; ++j) z_store(acc[j], j);
            if constexpr (!HEAVY) {
                z_store(acc[5], 5);
                z_store(acc[6], 5 - 18);                             // plane 5 of wave pg - 2
            }
            __syncthreads();                                         // (E1) half 0 is in LDS
            out_transform(ic<0>{});
            __syncthreads();                                         // (E2) half 0 has been read
            __syncthreads();                                         // (E3) half 1 is in LDS (S waves)
        };
        if (wave == 0) t_loop(ic<0>{});
        else if (wave == 1) t_loop(ic<1>{});
        else if (wave == 2) t_loop(ic<2>{});
        else t_loop(ic<3>{});
    } else {
        // ================== S waves: 48 MFMAs per K step and nothing else ==================
        // group g = units (3g, 1), (3g + 1, 1), (3g + 2, 1), (6 + g, 0) -> acc[4g .. 4g + 3]
#pragma unroll
        for (int j = 0; j < 12; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        f32x4 Bq[2][4], As[4];
        auto load_Bg = [&](f32x4 (&B)[4], int c, int g) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) B[pl] = load_unit(c, 3 * g + pl, 1);
            B[3] = load_unit(c, 6 + g, 0);
        };
        __syncthreads();                                             // patches 0 and 1 are in raw[0], raw[1]
        load_Bg(Bq[0], 0, 0);
        auto s_step = [&](int c8, auto par_c) {
            constexpr int PAR = decltype(par_c)::value;
            const int cn = (c8 + 1 < n8) ? c8 + 1 : 0;
            W43_STAMP(0);
            __builtin_amdgcn_sched_barrier(0);                       // (the scheduler moves MFMAs across s_barrier otherwise)
            if constexpr (!(W43_ABL & 8)) __syncthreads();           // V[PAR] and raw[PAR ^ 1] are complete
            __builtin_amdgcn_sched_barrier(0);
            W43_STAMP(1);
            const float* vcur = s_V + PAR * VF + vabase;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) As[pl] = load_plane(vcur, pl);
            As[3] = load_plane(vcur, 6);
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                constexpr int dummy = 0; (void)dummy;
                const int bs = (PAR + g) & 1;
                if (g < 2) load_Bg(Bq[bs ^ 1], c8, g + 1); else load_Bg(Bq[bs ^ 1], cn, 0);
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int u = 0; u < 4; ++u) W43D_MFMA(acc[4 * g + u], As[u][q], Bq[bs][u][q]);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    W43D_MFMA(acc[4 * g + u], As[u][3], Bq[bs][u][3]);
                    if (g < 2) As[u] = load_plane(vcur, u < 3 ? 3 * (g + 1) + u : 7 + g);
                }
            }
            W43_STAMP(2);
            W43_STAMP(3);
        };
        W43_STAMP(6);
        for (int c8 = 0; c8 < n8; c8 += 2) {
            s_step(c8, ic<0>{});
            s_step(c8 + 1, ic<1>{});
        }
        // ---- epilogue, S side
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                             // (E0)
        W43_STAMP(4);
#pragma unroll
        for (int g = 0; g < 3; ++g) z_store(acc[4 * g + 3], 6 + g);
        __syncthreads();                                             // (E1)
        __syncthreads();                                             // (E2) half 0 has been read by the T waves
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) z_store(acc[4 * g + pl], 3 * g + pl);
        __syncthreads();                                             // (E3)
        out_transform(ic<1>{});
    }
#undef W43D_MFMA

#ifdef NTK_CL_PROF
    W43_STAMP(5);
    if (prof_on)
        for (int i = 0; i < 12; ++i) g_w43d_prof[wave][i] = prof_acc[i];
#endif
}

}  // namespace

#ifdef NTK_CL_PROF
extern "C" int ntk_vgg_wino43_prof(unsigned long long* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_w43_prof), 96 * sizeof(unsigned long long)) == hipSuccess ? NTK_OK : NTK_ERR_HIP;
}
#endif

#ifdef NTK_CL_PROF
extern "C" int ntk_vgg_wino43d_prof(unsigned long long* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_w43d_prof), 96 * sizeof(unsigned long long)) == hipSuccess ? NTK_OK : NTK_ERR_HIP;
}
#endif

extern "C" size_t ntk_vgg_wino43_packed_floats(int cin, int cout) { return (size_t)36 * cin * cout; }

extern "C" int ntk_vgg_pack_weights_wino43(const float* w_hwio, float* u_packed, int cin, int cout, void* stream) {
    NTK_REQUIRE(w_hwio && u_packed, NTK_ERR_BAD_PTR, "ntk_vgg_pack_weights_wino43: null pointer");
    NTK_REQUIRE(cin >= 16 && (cin % 16) == 0 && cout >= 64 && (cout % 64) == 0, NTK_ERR_UNSUPPORTED,
                "ntk_vgg_pack_weights_wino43: cin=%d (multiple of 16) cout=%d (multiple of 64)", cin, cout);
    wino43_pack_kernel<<<2048, 256, 0, (hipStream_t)stream>>>(w_hwio, u_packed, cin, cout);
    NTK_CHECK_LAUNCH("ntk_vgg_pack_weights_wino43");
    return NTK_OK;
}

static int wino43_launch(const float* in, const float* u_packed, const float* bias, float* out,
                         int frames, int H, int W, int cin, int cout, int fuse_pool, int y0, int x0, int y1, int x1, int waves, void* stream,
                         int in_blocked = 0, int out_blocked = 0) {
    NTK_REQUIRE(waves == 4 || waves == 8, NTK_ERR_UNSUPPORTED, "ntk_vgg_conv3x3_relu_wino43: waves=%d (4 or 8 per workgroup)", waves);
    NTK_REQUIRE(in && u_packed && bias && out, NTK_ERR_BAD_PTR, "ntk_vgg_conv3x3_relu_wino43_f32: null pointer");
    NTK_REQUIRE(ntk_aligned16(in) && ntk_aligned16(u_packed) && ntk_aligned16(out), NTK_ERR_BAD_PTR,
                "ntk_vgg_conv3x3_relu_wino43_f32: 16-byte alignment");
    NTK_REQUIRE(frames > 0 && H >= 4 && (H % 4) == 0 && W >= 4 && (W % 4) == 0, NTK_ERR_UNSUPPORTED,
                "ntk_vgg_conv3x3_relu_wino43_f32: frames=%d H=%d W=%d (H, W multiples of 4)", frames, H, W);
    NTK_REQUIRE(cin >= 16 && (cin % 16) == 0 && cout >= 64 && (cout % 64) == 0, NTK_ERR_UNSUPPORTED,
                "ntk_vgg_conv3x3_relu_wino43_f32: cin=%d (multiple of 16: the K loop takes two 8-channel steps per trip) "
                "cout=%d (multiple of 64)", cin, cout);
    NTK_REQUIRE((unsigned long long)2 * H * W * cin * sizeof(float) <= 0x40000000ull, NTK_ERR_UNSUPPORTED,
                "ntk_vgg_conv3x3_relu_wino43_f32: two frames of %d x %d x %d floats exceed the 1 GiB window of the staging buffer resource",
                H, W, cin);
    Wino43Args a;
    a.in = in; a.U = u_packed; a.bias = bias; a.out = out;
    a.frames = frames; a.H = H; a.W = W; a.Cin = cin; a.Cout = cout;
    // tile-block shape: 0 = 8x4x1 (tile grid multiple of 8 x 4), 1 = 4x4x2, 2 = 2x2x8, 3 = 1x1x32 (any grid)
    // the computed window [y0, y1) x [x0, x1) in output pixels before the pool (whole frame: 0, 0, H, W), multiples of 4
    NTK_REQUIRE(y0 >= 0 && x0 >= 0 && y1 <= H && x1 <= W && y0 < y1 && x0 < x1 && ((y0 | x0 | y1 | x1) & 3) == 0, NTK_ERR_BAD_SHAPE,
                "ntk_vgg_conv3x3_relu_wino43: window [%d,%d) x [%d,%d) of a %d x %d frame (multiples of 4 inside the frame)", y0, y1, x0, x1, H, W);
    const int gw = (x1 - x0) / 4, gh = (y1 - y0) / 4, gx0 = x0 / 4, gy0 = y0 / 4;
    auto fits = [&](int tw, int th) { return (gw % tw) == 0 && (gh % th) == 0 && (gx0 % tw) == 0 && (gy0 % th) == 0; };
    const int shape = fits(8, 4) ? 0 : (fits(4, 4) ? 1 : (fits(2, 2) ? 2 : 3));
    static const int TWs[4] = {8, 4, 2, 1}, THs[4] = {4, 4, 2, 1}, NSUBs[4] = {1, 2, 8, 32};
    a.nCB = cout / 64;
    a.bxN = gw / TWs[shape];
    a.byN = gh / THs[shape];
    a.bx0 = gx0 / TWs[shape];
    a.by0 = gy0 / THs[shape];
    const long long NQ = (long long)frames * a.byN * a.bxN;
    const long long NS = (NQ + NSUBs[shape] - 1) / NSUBs[shape];
    NTK_REQUIRE(NS < (1ll << 30) && (a.nCB <= 8 ? (8 % a.nCB) == 0 : (a.nCB % 8) == 0), NTK_ERR_UNSUPPORTED,
                "ntk_vgg_conv3x3_relu_wino43_f32: cout/64=%d must divide or be a multiple of 8", a.nCB);
    a.NS = (int)NS;
    a.NQ = (int)NQ;
    long long slots;
    if (a.nCB >= 8) slots = NS * (a.nCB / 8);
    else { const int per = 8 / a.nCB; slots = (NS + per - 1) / per; }
    const long long grid = slots * 8;
    NTK_REQUIRE(grid < (1ll << 31), NTK_ERR_UNSUPPORTED, "ntk_vgg_conv3x3_relu_wino43_f32: grid too large");
    // the eight-wave kernel packs a staging slot into one register on the 1x1x32 shape: a block's input span (the frames its 32
    // tiles touch) must stay below 16 MB there
    const long long tpf = (long long)a.bxN * a.byN;
    const long long span = ((32 + tpf - 1) / tpf + 1) * (long long)H * W * cin * (long long)sizeof(float);
    const bool dual = waves == 8 && (shape != 3 || span <= 0xfffff0ll);
    NTK_REQUIRE(!(in_blocked || out_blocked) || dual, NTK_ERR_UNSUPPORTED,
                "ntk_vgg_conv3x3_relu_wino43_layout_f32: channel-blocked maps need the eight-wave kernel "
                "(in_blocked=%d out_blocked=%d waves=%d, input span of a block %lld B)", in_blocked, out_blocked, waves, span);
#define W43_LAUNCH(POOL_, TW_, TH_, NSUB_, PWS_, SPXS_)                                                                       \
    do {                                                                                                                      \
        if (dual && in_blocked && out_blocked)                                                                                \
            conv3x3_wino43d_kernel<POOL_, TW_, TH_, NSUB_, PWS_, SPXS_, true, true><<<(unsigned)grid, W4D, 0, (hipStream_t)stream>>>(a); \
        else if (dual && in_blocked)                                                                                          \
            conv3x3_wino43d_kernel<POOL_, TW_, TH_, NSUB_, PWS_, SPXS_, true, false><<<(unsigned)grid, W4D, 0, (hipStream_t)stream>>>(a); \
        else if (dual && out_blocked)                                                                                         \
            conv3x3_wino43d_kernel<POOL_, TW_, TH_, NSUB_, PWS_, SPXS_, false, true><<<(unsigned)grid, W4D, 0, (hipStream_t)stream>>>(a); \
        else if (dual) conv3x3_wino43d_kernel<POOL_, TW_, TH_, NSUB_, PWS_, SPXS_><<<(unsigned)grid, W4D, 0, (hipStream_t)stream>>>(a); \
        else conv3x3_wino43_kernel<POOL_, TW_, TH_, NSUB_, PWS_, SPXS_><<<(unsigned)grid, W4T, 0, (hipStream_t)stream>>>(a);  \
    } while (0)
    // PWS / SPXS: pixel-slot strides of a patch row / a sub-block, chosen so that the sixteen lanes of every window
    // ds_read_b128 fall on sixteen different 16-byte bank slots (tile stride 10 slots; row / sub-block strides = 8 / 6 mod 16)
    if (shape == 0) { if (fuse_pool) W43_LAUNCH(true, 8, 4, 1, 42, 18 * 42); else W43_LAUNCH(false, 8, 4, 1, 42, 18 * 42); }
    else if (shape == 1) { if (fuse_pool) W43_LAUNCH(true, 4, 4, 2, 23, 18 * 23); else W43_LAUNCH(false, 4, 4, 2, 23, 18 * 23); }
    else if (shape == 2) { if (fuse_pool) W43_LAUNCH(true, 2, 2, 8, 13, 10 * 13); else W43_LAUNCH(false, 2, 2, 8, 13, 10 * 13); }
    else { if (fuse_pool) W43_LAUNCH(true, 1, 1, 32, 7, 43); else W43_LAUNCH(false, 1, 1, 32, 7, 43); }
#undef W43_LAUNCH
    NTK_CHECK_LAUNCH("ntk_vgg_conv3x3_relu_wino43_f32");
    return NTK_OK;
}

extern "C" int ntk_vgg_conv3x3_relu_wino43_f32(const float* in, const float* u_packed, const float* bias, float* out,
                                               int frames, int H, int W, int cin, int cout, int fuse_pool, void* stream) {
    return wino43_launch(in, u_packed, bias, out, frames, H, W, cin, cout, fuse_pool, 0, 0, H, W, 8, stream);
}

// The same layer computed only inside the window [y0, y1) x [x0, x1) of the (un-pooled) output, multiples of 4: every 4x4 output
// tile of the window is written exactly as the whole-frame call writes it, nothing outside the window is touched.  For the last
// layer of a trunk whose consumer reads a fixed set of positions (extract_features' 64 points of conv4_3 lie in rows / columns
// 4..23 of 28: 25 of the 49 tiles).
extern "C" int ntk_vgg_conv3x3_relu_wino43_window_f32(const float* in, const float* u_packed, const float* bias, float* out,
                                                      int frames, int H, int W, int cin, int cout, int fuse_pool,
                                                      int y0, int x0, int y1, int x1, void* stream) {
    return wino43_launch(in, u_packed, bias, out, frames, H, W, cin, cout, fuse_pool, y0, x0, y1, x1, 8, stream);
}

// The general form: window + the kernel form.  waves = 8 (what the two entries above use): conv3x3_wino43d_kernel, two waves per
// SIMD; waves = 4: conv3x3_wino43_kernel, one wave per SIMD (round 2's kernel; also what a 1x1-tile-block layer whose blocks span
// more than 16 MB of input falls back to).  The two forms give the same bits.
extern "C" int ntk_vgg_conv3x3_relu_wino43_form_f32(const float* in, const float* u_packed, const float* bias, float* out,
                                                    int frames, int H, int W, int cin, int cout, int fuse_pool,
                                                    int y0, int x0, int y1, int x1, int waves, void* stream) {
    return wino43_launch(in, u_packed, bias, out, frames, H, W, cin, cout, fuse_pool, y0, x0, y1, x1, waves, stream);
}

// The eight-wave kernel with CHANNEL-BLOCKED activation maps [frames][H][C / 8][W][8] on either side (what the layers of a trunk
// hand to each other): in_blocked / out_blocked say which side is blocked (0 = NHWC: the first Winograd layer reads conv1_1's NHWC
// output, the last one writes NHWC for gather_serialize).  Whole frames; same arithmetic and bits as the NHWC entries.
extern "C" int ntk_vgg_conv3x3_relu_wino43_layout_f32(const float* in, const float* u_packed, const float* bias, float* out,
                                                      int frames, int H, int W, int cin, int cout, int fuse_pool,
                                                      int in_blocked, int out_blocked, void* stream) {
    return wino43_launch(in, u_packed, bias, out, frames, H, W, cin, cout, fuse_pool, 0, 0, H, W, 8, stream, in_blocked, out_blocked);
}
