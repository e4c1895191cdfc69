// conv3x3 SAME + bias + ReLU (+ fused 2x2/2 max-pool) on the 16-bit matrix pipe, PATCH form (round 4).  Two operators share this kernel:
//   * the bf16 trunk of BASELINE config 5 (bf16 operands, fp32 accumulate): ntk_vgg_conv3x3_relu_bf16p;
//   * the SPLIT form of the fp32 trunk (template flag X3, further down): every fp32 value as two fp16 numbers, every product as
//     three fp16 MFMA products with fp32 accumulators -- the default trunk of BASELINE configs 1 and 3's NTM tracker:
//     ntk_vgg_conv3x3_relu_split3.
// What follows describes the kernel on the bf16 operator; the split form differs in what a "step" is, in the epilogue's split and
// in the 28-wide maps (see the comment in front of the kernel).
//
// conv_bf16.hip's kernel re-stages a 128-pixel x 64-channel A tile from L2 for every one of the nine taps and a 128-column
// B tile beside it: 512 B of global -> LDS traffic per v_mfma_f32_32x32x16_bf16, at four workgroups per CU exactly the
// 64 B/clk the fill path has -- the kernel sat at 31 % of the matrix peak with the pipe half idle (three
// experiments on the tile shape, all bound the same way).  This kernel removes the bytes instead:
//   * the INPUT of a block of 512 output pixels is staged ONCE per 32-channel chunk as a patch with its one-pixel halo
//     (sub-blocks of 32x16 / 16x16 / 8x8 / 4x4 pixels, as the Winograd kernels cut a frame) and serves all nine taps: the
//     operand of tap (dy, dx) is the same LDS image read at a uniform pixel shift;
//   * the WEIGHTS of a stage (one kernel row = three taps x 32 channels x 128 columns, 24 KB) go through LDS once per
//     workgroup and are shared by the sixteen 32-pixel tiles;
//   so a workgroup moves 113 KB of global -> LDS bytes per 1 152 MFMAs of a chunk (98 B per MFMA, a fifth of before), all by
//   LDS-DMA (buffer_load ... lds: no staging registers, padding pixels are out-of-range lanes that write zeros), double
//   buffered, one workgroup barrier per stage of 48 MFMAs per wave;
//   * 8 waves = 4 (pixel tiles) x 2 (column halves), two per SIMD, each 128 pixels x 64 columns (8 accumulator tiles): per
//     MFMA 768 B of fragment reads, 96 B/clk per CU against the LDS array's 256; both fragment images are XOR-swizzled at
//     16-byte granularity on the SOURCE side of the DMA (the LDS side of a DMA is lane-linear), ds_read_b128 conflict-free on
//     runs of sixteen pixels;
//   * the product is computed TRANSPOSED, D[channel][pixel] (weights as the A operand): a lane then holds four adjacent
//     channels of one pixel per register quad and stores 8 bytes (bf16) / 16 bytes (fp32) at a time; vertically adjacent
//     rows of a sub-block live in consecutive tiles of one wave, so the 2x2 pool is a register max + one DPP exchange.
// Same operator, operand rounding and accumulation type as conv_bf16.hip (bf16 operands, fp32 accumulate, one bf16 rounding
// on store; vgg.py:155-161); the summation ORDER over (chunk, tap, channel) differs, so results agree to fp32 rounding.
#include "common.h"
#include <type_traits>

// Ablation switches (timing only, results wrong; never defined in the product build): bit 0 no DMA after the prologue, bit 1 no
// workgroup barrier / DMA wait in the K loop, bit 2 fragments not re-read (one set for the whole kernel), bit 3 no MFMAs, bit 4 no epilogue, bit 5 the un-pooled epilogue without its global stores
#ifndef BF16P_ABL
#define BF16P_ABL 0
#endif

namespace {

template <int I0, int I1, class F>
__device__ __forceinline__ void bp_for(F&& f) {
    if constexpr (I0 < I1) { f(std::integral_constant<int, I0>{}); bp_for<I0 + 1, I1>(f); }
}

typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pbf16x4 __attribute__((ext_vector_type(4)));


struct Bf16pArgs {
    const __bf16* in; const __bf16* wq; const float* bias; void* out;
    int frames, H, W, Cin, Cout;
    int bxN, byN;       // sub-blocks per frame
    int NQ, NS;         // sub-blocks in all, workgroups per column block
    int nCB;            // Cout / BN
};

// One LDS-DMA piece (1 KB per wave): buffer_load_dwordx4 ... lds through inline assembly.  The builtin
// (__builtin_amdgcn_raw_ptr_buffer_load_lds) is correct but the compiler guards it: it cannot tell the buffer a DMA fills from
// the one the fragment reads use (both are runtime halves of one array) and puts s_waitcnt vmcnt(0) in front of later DMAs and
// LDS reads -- every DMA then waits for the one before it to land.  Here the ordering is the kernel's own: one
// s_waitcnt vmcnt(0) + workgroup barrier per stage.  M0 = LDS byte address of the piece (saved and restored).
typedef int pv4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void p_dma16(pv4i rs, unsigned lds_addr, int voff, int soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ pv4i p_rsrc(const void* base, unsigned bytes) {
    const unsigned long long p = (unsigned long long)base;
    return pv4i{(int)(unsigned)p, (int)(unsigned)(p >> 32), (int)bytes, 0x00020000};
}

__device__ __forceinline__ pbf16x8 p_as_bf16x8(const f32x4& v) {
    union { f32x4 f; pbf16x8 b; } u;
    u.f = v;
    return u.b;
}

typedef _Float16 ph16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ph16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 ph16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ph16x8 p_as_f16x8(const f32x4& v) {
    union { f32x4 f; ph16x8 h; } u;
    u.f = v;
    return u.h;
}
// The split form's two parts of four fp32 values: hi = fp16(v) rounded TOWARD ZERO (v_cvt_pkrtz_f16_f32: a value beyond the fp16
// range saturates at 65504 instead of becoming infinite, and lo then carries the rest up to 131008), lo = fp16(v - hi) rounded to
// nearest: v = hi + lo to 2^-22 |v| (eleven + eleven bits) for 6e-5 < |v| < 65504; below, to fp16's absolute 6e-8.
// PRECONDITION: |v| <= S3_MAX = 131008 (the callers clamp: the epilogues with the same v_med3_f32 that is their ReLU) -- so that
// lo cannot overflow either.
__device__ __forceinline__ void s3_split4(const f32x4& v, ph16x4& hi, ph16x4& lo) {
    const auto h01 = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]), h23 = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
    hi = ph16x4{(_Float16)h01[0], (_Float16)h01[1], (_Float16)h23[0], (_Float16)h23[1]};
#pragma unroll
    for (int e = 0; e < 4; ++e) lo[e] = (_Float16)(v[e] - (float)hi[e]);
}
constexpr float S3_MAX = 131008.f;

// 16-byte piece `piece` of LDS pixel / column `idx` lives in slot piece ^ sw(idx): PIECES = 4 (64-byte rows): bits 2..3 of the
// index; PIECES = 2 (32-byte rows): bit 3 -- sixteen consecutive indices then cover all 64 banks once per ds_read_b128 group
template <int PIECES>
__device__ __forceinline__ int p_sw(int idx) { return PIECES == 4 ? ((idx >> 2) & 3) : ((idx >> 3) & 1); }

// HWIO fp32 [3][3][Cin][Cout] -> bf16, the LDS image of every stage in stage order:
// [cb][chunk][dy][dx][n (BN)][slot (KC/8)][8], slot = piece ^ sw(n)
template <int BN, int KC>
__global__ void bf16p_pack_kernel(const float* __restrict__ w, __bf16* __restrict__ wq, int Cin, int Cout) {
    constexpr int PIECES = KC / 8;
    const size_t total = (size_t)9 * Cin * Cout;
    const int NC = Cin / KC;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = idx & 7;
        size_t r = idx >> 3;
        const int slot = (int)(r % PIECES); r /= PIECES;
        const int n = (int)(r % BN); r /= BN;
        const int tap = (int)(r % 9); r /= 9;
        const int chunk = (int)(r % NC);
        const int cb = (int)(r / NC);
        const int piece = slot ^ p_sw<PIECES>(n);
        const int c = chunk * KC + piece * 8 + e;
        wq[idx] = (__bf16)w[((size_t)tap * Cin + c) * Cout + (size_t)cb * BN + n];
    }
}

// The SPLIT form's weights (X3 below): a 32-"channel" chunk = 16 real input channels, pieces 0, 1 = the fp16 HIGH parts of channels
// 0..7 / 8..15, pieces 2, 3 = the fp16 LOW parts (w - high, rounded to fp16) of the same channels.  Same stage order as above.
// The weights are scaled by a power of two first (exact) so that the largest is in [2^13, 2^14): the low parts are then normal fp16
// numbers down to weights 2^-17 of the largest.  The inverse scale sits behind the packed image (tail[0]); the kernel's epilogue
// multiplies by it (exact).  Two launches: the maximum (as its bit pattern: positive floats order as integers), then the packing.
__global__ void split3_absmax_kernel(const float* __restrict__ w, size_t n, unsigned* __restrict__ tail) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = fabsf(w[i]);
        if (v < 3.0e38f) m = fmaxf(m, v);                                       // not-a-number and infinite weights do not set the scale
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(tail + 1, __float_as_uint(m));
}
__global__ void split3_scale_kernel(unsigned* __restrict__ tail) {
    const float m = __uint_as_float(tail[1]);
    int e = 0;
    if (m > 0.f) frexpf(m, &e);                                                 // m = f * 2^e, f in [0.5, 1)
    int k = 14 - e;                                                             // m * 2^k in [2^13, 2^14)
    k = k > 100 ? 100 : (k < -100 ? -100 : k);
    reinterpret_cast<float*>(tail)[0] = ldexpf(1.f, -k);                        // what the convolution multiplies its sums by
    reinterpret_cast<float*>(tail)[2] = ldexpf(1.f, k);                         // what the packing multiplies the weights by
}
template <int BN>
__global__ void split3_pack_kernel(const float* __restrict__ w, _Float16* __restrict__ wq, int Cin, int Cout) {
    const size_t total = (size_t)18 * Cin * Cout;
    const float wscale = reinterpret_cast<const float*>(wq + total)[2];
    const int NC = Cin / 16;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = idx & 7;
        size_t r = idx >> 3;
        const int slot = (int)(r & 3); r >>= 2;
        const int n = (int)(r % BN); r /= BN;
        const int tap = (int)(r % 9); r /= 9;
        const int chunk = (int)(r % NC);
        const int cb = (int)(r / NC);
        const int piece = slot ^ p_sw<4>(n);
        const int c = chunk * 16 + (piece & 1) * 8 + e;
        const float v = w[((size_t)tap * Cin + c) * Cout + (size_t)cb * BN + n] * wscale;
        const _Float16 hi = (_Float16)v;
        wq[idx] = (piece & 2) ? (_Float16)(v - (float)hi) : hi;
    }
}

// NW: waves per workgroup.  8: 512 output pixels, one workgroup per CU.  4 (64 columns only): 256 pixels and under 80 KB of LDS,
// TWO workgroups per CU -- for the layers with few stages (64 input channels: six), where a workgroup's prologue (the first patch
// and weights have to land before the first MFMA) and epilogue are a third of its life and nothing else overlaps them.
//
// X3 -- the SPLIT form: an fp32 convolution on the 16-bit matrix pipe.  Every fp32 value v is carried as two fp16 numbers, hi = fp16(v)
// and lo = fp16(v - hi) (v = hi + lo to 2^-22 |v|: s3_split4), and a product x w is accumulated as xh wh + xh wl + xl wh in fp32 (the
// dropped xl wl is 2^-20 relative): three v_mfma_f32_32x32x16_f16 where the fp32 pipe needs sixteen times the cycles of one.  Maps are
// [pixel][C / 16][hi x16 | lo x16] fp16 (the bytes of an fp32 map); to this kernel that is an NHWC 16-bit map of 2 C channels in
// 32-channel chunks, so the staging is the bf16 form's, byte for byte.  (First built with bf16 parts: 2^-17 per product, 1.5e-5 of the
// activation scale through the trunk -- four times the F(4x4) Winograd kernel's error; fp16 parts cost the same and give 2^-20.  fp16's
// narrow exponent is handled on both sides: weights are scaled by a power of two when packed, activations saturate instead of
// overflowing.)  A "step" becomes a tap: 24 MFMAs per wave on four fragment
// sets (xh, xl, wh, wl), the three products of a tap software-pipelined against each other's fragment reads.
// ROWS (TW not a power of two: the 28 x 28 maps of conv4_x, TW = W = 28): no sub-blocks -- a workgroup takes 512 CONSECUTIVE pixels
// of the batch in (frame, row, column) order, about 18.3 image rows, whatever frame they belong to (no padding tiles; first version:
// 28 x 4 sub-blocks, four per workgroup = 448 of 512 columns used).  Its patch is the run's rows plus one above and one below, with ONE
// all-zero row between two frames (the row below a frame's last row and above the next frame's first): image row (f, y) sits in patch
// row (f - f0)(H + 1) + y + 1 - y0, so a tap is the same uniform shift as everywhere else.  H >= 20: at most one frame boundary per run.
// INF32 (split form, four-wave workgroups): the input is an fp32 NHWC map and the STAGING splits it -- 16-byte pieces through registers
// (two or three per lane and stage, requested at the top of a stage, split and written to the patch image at its end) instead of DMA.
// That is how the trunk enters the split form: conv1_1 keeps its fp32 kernel and conv1_2 reads its map as it is.
template <int BN, int TW, int TH, int NSUB, int KC, bool POOL, bool OUTF32, int NW = 8, bool X3 = false, bool INF32 = false>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_relu_bf16p_kernel(Bf16pArgs a) {
    static_assert(!INF32 || (X3 && NW == 4), "fp32 input: split form on four waves (the eight-wave form has no registers for the staging)");
    constexpr int PNT = 2 * NW;                                                 // 32-pixel tiles per workgroup
    constexpr bool ROWS = (TW & (TW - 1)) != 0;
    static_assert(!X3 || KC == 32, "split form: 16 real channels = one 32-channel chunk");
    static_assert(!ROWS || (!POOL && NSUB == 1 && (TW - 1 + 32 * PNT - 1) / TW + 1 <= TH - 1 && NW == 8), "runs of rows: no row pairs; TH = the patch's image rows");
    static_assert(NW == 8 || (NW == 4 && BN == 64), "four-wave workgroups take 64 columns");
    // patch row pitch: the sub-block's width + the halo; ROWS: 32 -- the run's pixels then keep their residue sequence mod 16 across
    // a row end (28 image pixels + 4 = 32 patch pixels further: a_sw below), which the fragment reads need to stay conflict-free
    constexpr int PW = ROWS ? 32 : TW + 2, PH = TH + 2, SPX = PW * PH, NPX = NSUB * SPX;
    static_assert(!ROWS || TW + 2 <= 32, "runs of rows: the patch row pitch is 32");
    constexpr int PIXB = KC * 2, PIECES = KC / 8, PPP = 1024 / PIXB;            // bytes per pixel row, 16-B pieces per row, rows per DMA
    constexpr int NPA = (NPX + PPP - 1) / PPP;                                  // DMA pieces of the patch
    constexpr int ABYTES = NPA * 1024;
    // a stage = the taps whose weights share one LDS buffer and one barrier: a kernel row (3 taps) with 32-channel chunks, the whole
    // chunk (9 taps) with 16-channel chunks (its weights are half the size, and its steps half as many per tap)
    constexpr int STAPS = KC == 16 ? 9 : 3, SPC = 9 / STAPS;                    // taps per stage, stages per chunk
    constexpr int SBYTES = STAPS * BN * PIXB, NPB = SBYTES / 1024;              // one stage of weights
    constexpr int NPAW = (NPA + NW - 1) / NW;                                   // patch pieces per wave
    constexpr int TM = BN == 128 ? 4 : 2;                                       // pixel tiles per wave (two column tiles either way)
    constexpr int TPS = (TW * TH) / 32;                                         // tiles per sub-block (0: two sub-blocks per tile)
    constexpr int K16 = KC / 16;
    static_assert(ROWS || NSUB * TW * TH == 32 * PNT, "512 (256) output pixels per workgroup");
    constexpr int TRB = POOL ? 0 : NW * TM * 32 * 144;                           // the un-pooled epilogue's transpose image (below)
    constexpr int LDSB = (2 * ABYTES + 2 * SBYTES) > TRB ? (2 * ABYTES + 2 * SBYTES) : TRB;
    static_assert(LDSB + 1024 <= (NW == 8 ? 160 : 80) * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) unsigned char s_mem[LDSB];
    // s_mem: two patch buffers of ABYTES, then two weight-stage buffers of SBYTES
    __shared__ int s_sbf[NSUB], s_sby[NSUB], s_sbx[NSUB];
    // the column block's biases (and the split form's output scale) are fetched with the tables: read from global memory in the
    // epilogue they were eight dependent round trips at the very end of a workgroup that is alone on its CU
    __shared__ __attribute__((aligned(16))) float s_bias[BN + 4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int id = blockIdx.x, xcd = id & 7, slotid = id >> 3;
    int cb, sp;
    if (a.nCB >= 8) {
        const int kN = a.nCB >> 3;
        cb = (slotid % kN) * 8 + xcd;
        sp = slotid / kN;
    } else {
        const int per = 8 / a.nCB;
        cb = xcd % a.nCB;
        sp = slotid * per + xcd / a.nCB;
    }
    if (sp >= a.NS) return;
    const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
    // ROWS: first pixel of the run, its global row g0 = (frame rf0, row ry0)
    const int rP0 = sp * (32 * PNT), rg0 = ROWS ? rP0 / TW : 0, rf0 = ROWS ? rg0 / H : 0, ry0 = ROWS ? rg0 - rf0 * H : 0;
    if (tid < BN) s_bias[tid] = a.bias[cb * BN + tid];
    if (X3 && tid == BN) s_bias[BN] = *reinterpret_cast<const float*>(a.wq + (size_t)9 * a.Cin * a.Cout);
    if (!ROWS && tid < NSUB) {
        const int sq = sp * NSUB + tid;
        if (sq < a.NQ) {
            const int bx = sq % a.bxN;
            const int t1 = sq / a.bxN;
            s_sbf[tid] = t1 / a.byN; s_sby[tid] = TH * (t1 % a.byN); s_sbx[tid] = TW * bx;
        } else {
            s_sbf[tid] = -1; s_sby[tid] = 0; s_sbx[tid] = 0;
        }
    }
    __syncthreads();

    // the patch image's swizzle: slot of piece 0 of patch pixel p.  ROWS: by the pixel's position in the RUN (28 per row), not in the
    // patch (32 per row): consecutive run pixels then have consecutive residues mod 16 across row ends, and since 28 = 0 mod 4 the
    // bank quarter (p mod 4) follows the same sequence
    auto a_sw = [](int p) { return ROWS ? ((((p & 31) + 28 * (p >> 5)) >> 2) & 3) : p_sw<PIECES>(p); };
    // ---- patch DMA: this wave's pieces pa = wave + 8 i; a lane = one 16-byte slot of one LDS pixel row
    const int f0 = ROWS ? rf0 : ((sp * NSUB) / a.bxN) / a.byN;
    const size_t in_left = (size_t)(a.frames - f0) * H * W * Cin * sizeof(__bf16);
    const unsigned in_bytes = (unsigned)(in_left < 0x7ffffff0ull ? in_left : 0x7ffffff0ull);
    const __bf16* pin = a.in + (size_t)f0 * H * W * Cin;
    unsigned aoff[NPAW];
#pragma unroll
    for (int i = 0; i < NPAW; ++i) {
        const int pa = wave + NW * i;
        const int lp = pa * PPP + lane / PIECES, slot = lane % PIECES;
        aoff[i] = 0x80000000u;                                                  // out of range: the DMA writes zeros
        if (pa < NPA && lp < NPX) {
            const int q = lp / SPX, rem = lp - q * SPX;
            const int py = rem / PW, px = rem - py * PW;
            int fq, y;
            const int x = (ROWS ? 0 : s_sbx[q]) - 1 + px;
            if constexpr (ROWS) {                                               // patch row py <-> virtual row U = ry0 + py, H + 1 per frame
                const int U = ry0 + py, df = U / (H + 1);
                fq = rf0 + df; y = U - df * (H + 1) - 1;                        // y = -1: the zero row above a frame
                if (fq >= a.frames) fq = -1;
            } else {
                fq = s_sbf[q]; y = s_sby[q] - 1 + py;
            }
            if (fq >= 0 && y >= 0 && y < H && x >= 0 && x < W)
                aoff[i] = (unsigned)(((((size_t)(fq - f0) * H + y) * W + x) * Cin + (slot ^ a_sw(lp)) * 8) * sizeof(__bf16));
        }
    }
    // INF32: item it = tid + 64 NW i = (patch pixel it / 4, channel quad it % 4): 16 bytes of the fp32 map -> 8 bytes of high parts in
    // piece (q >> 1) of the pixel's row and 8 bytes of low parts in piece 2 + (q >> 1) (= the same byte address with bit 5 flipped)
    constexpr int NIT = INF32 ? (NPX * 4 + 64 * NW - 1) / (64 * NW) : 1;       // items per lane and chunk
    constexpr int NITS = (NIT + SPC - 1) / SPC;                                  // ... and stage
    unsigned foff[NIT];
    int fdst[NIT];
    if constexpr (INF32) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int it = tid + 64 * NW * i, lp = it >> 2, qd = it & 3;
            foff[i] = 0x80000000u; fdst[i] = -1;
            if (lp < NPX) {
                fdst[i] = lp * PIXB + (((qd >> 1) ^ a_sw(lp)) << 4) + (qd & 1) * 8;
                const int q = lp / SPX, rem = lp - q * SPX;
                const int py = rem / PW, px = rem - py * PW;
                const int fq = s_sbf[q];
                const int y = s_sby[q] - 1 + py, x = s_sbx[q] - 1 + px;
                if (fq >= 0 && y >= 0 && y < H && x >= 0 && x < W)
                    foff[i] = (unsigned)(((((size_t)(fq - f0) * H + y) * W + x) * Cin) * sizeof(__bf16) + qd * 16);   // Cin bf16 = Cin / 2 floats
            }
        }
    }
    auto f32_rsrc = [&](int chunk) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(pin + (size_t)chunk * KC), 0, (int)(in_bytes - (unsigned)chunk * PIXB), 0x00020000);
    };
    auto f32_put = [&](const f32x4& v, int dst) {                               // split one item into the patch image at byte dst of s_mem
        ph16x4 h, l;
        f32x4 c;
#pragma unroll
        for (int e = 0; e < 4; ++e) c[e] = __builtin_amdgcn_fmed3f(v[e], -S3_MAX, S3_MAX);
        s3_split4(c, h, l);
        *reinterpret_cast<ph16x4*>(s_mem + dst) = h;
        *reinterpret_cast<ph16x4*>(s_mem + (dst ^ 32)) = l;
    };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)s_mem;       // LDS byte address of the block
    auto dma_patch = [&](int chunk, int buf, int i0, int i1) {                 // pieces i0 .. i1 - 1 of this wave
        const pv4i rs = p_rsrc(pin + (size_t)chunk * KC, in_bytes - (unsigned)chunk * PIXB);
#pragma unroll
        for (int i = 0; i < NPAW; ++i) {
            if (i >= i0 && i < i1 && wave + NW * i < NPA)
                p_dma16(rs, lds0 + buf * ABYTES + (wave + NW * i) * 1024, (int)aoff[i], 0);
        }
    };
    // one patch piece (index i of this wave's NPAW) of `chunk` into buffer `buf`
    auto dma_patch1 = [&](int chunk, int buf, auto ic) {
        constexpr int i = decltype(ic)::value;
        const pv4i rs = p_rsrc(pin + (size_t)chunk * KC, in_bytes - (unsigned)chunk * PIXB);
        if (wave + NW * i < NPA) p_dma16(rs, lds0 + buf * ABYTES + (wave + NW * i) * 1024, (int)aoff[i], 0);
    };
    const int NC = Cin / KC, NSTG = SPC * NC;
    const pv4i wrs = p_rsrc(a.wq + (size_t)cb * NSTG * (SBYTES / 2), (unsigned)NSTG * SBYTES);
    auto dma_weights1 = [&](int stage, int buf, int i) {                      // weight piece wave + 8 i of `stage`
        const int pb = wave + NW * i;
        if (pb < NPB) p_dma16(wrs, lds0 + 2 * ABYTES + buf * SBYTES + pb * 1024, lane * 16, stage * SBYTES + pb * 1024);
    };
    auto dma_weights = [&](int stage, int buf) {
#pragma unroll
        for (int i = 0; i < (NPB + NW - 1) / NW; ++i) {
            const int pb = wave + NW * i;
            if (pb < NPB) p_dma16(wrs, lds0 + 2 * ABYTES + buf * SBYTES + pb * 1024, lane * 16, stage * SBYTES + pb * 1024);
        }
    };

    // ---- wave roles
    const int wm = BN == 128 ? (wave >> 1) : wave, wn = BN == 128 ? (wave & 1) : 0;
    const int mcol = lane & 31, kh = lane >> 5;
    // patch pixel (tap 0,0 = the output pixel itself, halo offset included) of this lane's column in each of its tiles
    int pix0[TM];
    int opix[TM];                                                               // output pixel (element offset / Cout) or -1
    // column m of tile t -> its patch pixel pc (the tap (0, 0) operand, halo offset included) and its output pixel op (pooled: of
    // the 2x2 window's result), -1 where the column is beyond the batch
    auto locate = [&](int t, int m, int& pc, int& op) {
        if constexpr (ROWS) {
            const int P = rP0 + t * 32 + m;
            const int g = P / TW, x = P - g * TW, f = g / H, y = g - f * H;
            const bool real = f < a.frames;
            pc = real ? ((f - rf0) * (H + 1) + y + 1 - ry0) * PW + x + 1 : PW + 1;
            op = real ? P : -1;
        } else {
            int q, y, x;
            if constexpr (TPS == 0) {                                           // 4x4 sub-blocks: two per tile
                q = 2 * t + (m >> 4); y = (m >> 2) & 3; x = m & 3;
            } else {
                constexpr int RPT = 32 / TW;                                    // rows of a sub-block per tile
                const int tl = t % TPS, j = m / TW;
                q = t / TPS;
                y = 2 * ((tl >> 1) * RPT + j) + (tl & 1);                       // rows y, y + 1 sit in consecutive tiles (the pool's pairs)
                // Which column of its row a lane takes is free, and it decides the bank conflicts of the fragment reads: a
                // ds_read_b128 is served in groups of sixteen lanes {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}, conflict-free when the
                // group's patch pixels differ mod 16 (p_sw).  One 32-pixel row: they do.  Two rows of 16 (36 pixels apart) / four rows
                // of 8 (20 apart) taken in lane order: 40 % of the LDS cycles were conflicts (PMC) -> the second row rotated by 12
                // columns / rows 0 and 3 with their halves swapped: every group again covers sixteen residues.
                x = m % TW;
                if constexpr (TW == 16) x = (x + 12 * j) & 15;
                if constexpr (TW == 8) x ^= ((0x9 >> j) & 1) << 2;
            }
            pc = q * SPX + (y + 1) * PW + (x + 1);
            const int fq = s_sbf[q];
            const int Y = s_sby[q] + y, X = s_sbx[q] + x;
            if constexpr (POOL) op = fq < 0 ? -1 : ((fq * (H >> 1) + (Y >> 1)) * (W >> 1) + (X >> 1));
            else op = fq < 0 ? -1 : ((fq * H + Y) * W + X);
        }
    };
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) locate(wm * TM + tm, mcol, pix0[tm], opix[tm]);
    // LDS byte offsets (inside patch buffer 0) of this lane's A-operand fragments, per tap and tile, for the first 16-channel half of a
    // chunk (the second half is the same address with bit 5 flipped: the other slot pair of the pixel's 64-byte row).  They are the same
    // in every chunk, so they are computed ONCE and toggled between the two patch buffers per chunk: recomputed per step (pixel shift,
    // swizzle, scale: ~5 vector instructions per fragment) they were most of what the fragment reads cost beside the MFMAs.
    int xa[9][TM];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int p = pix0[tm] + (tap / 3 - 1) * PW + (tap % 3 - 1);
            xa[tap][tm] = p * PIXB + ((kh ^ a_sw(p)) << 4);
        }
    f32x16 acc[TM][2];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][nt][r] = 0.f;
    // weight fragment: row n = wn * 64 + nt * 32 + mcol of tap dx, piece 2 k16 + kh
    const int wa0 = 2 * ABYTES + (wn * 64 + mcol) * PIXB + ((kh ^ p_sw<PIECES>(mcol)) << 4);        // byte offset in s_mem, weight buffer 0

    // ---- prologue: patch of chunk 0 and the first stage of weights
    if constexpr (INF32) {
        const __amdgpu_buffer_rsrc_t rs = f32_rsrc(0);
        f32x4 v[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)foff[i], 0, 0));
#pragma unroll
        for (int i = 0; i < NIT; ++i)
            if (fdst[i] >= 0) f32_put(v[i], fdst[i]);
    } else {
        dma_patch(0, 0, 0, NPAW);
    }
    dma_weights(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // one stage = one kernel row of one chunk; the three rows are three copies of the body (the patch pieces a stage's DMAs
    // take are then compile-time indices into aoff[]: indexed at run time the array went to scratch, and every scratch read
    // waited for all DMAs in flight)
    auto stage_body = [&](int chunk, auto part_c) {
        constexpr int part = decltype(part_c)::value;
        const int s = SPC * chunk + part;
        const int abuf = chunk & 1, bbuf = s & 1;
        // The next stage's weights and a third of the next chunk's patch land while this stage multiplies.  Their DMA
        // instructions are issued ONE PER STEP, behind the step's MFMAs: an LDS-DMA instruction holds its wave's issue for
        // 60 - 185 cycles, and all eight waves leave the stage's barrier together -- issued in a block at the top of the stage
        // they stop both waves of every SIMD at once (ablation: 24 % of the kernel); behind eight queued MFMAs the other wave
        // of the SIMD has the pipe meanwhile.
        constexpr int NDW = (NPB + NW - 1) / NW;                                 // weight pieces per wave and stage
        constexpr int NDP = INF32 ? 0 : (NPAW + SPC - 1) / SPC;                  // patch pieces per wave and stage
        const bool more_w = !(BF16P_ABL & 1) && s + 1 < NSTG, more_p = !(BF16P_ABL & 1) && chunk + 1 < NC;
        auto dma_slot = [&](auto kc) {                                           // k-th DMA of this stage
            constexpr int k = decltype(kc)::value;
            if constexpr (k < NDW) { if (more_w) dma_weights1(s + 1, bbuf ^ 1, k); }
            else if constexpr (k < NDW + NDP && part * NDP + (k - NDW) < NPAW) {
                if (more_p) dma_patch1(chunk + 1, abuf ^ 1, std::integral_constant<int, part * NDP + (k - NDW)>{});
            }
        };
        const int wa = wa0 + bbuf * SBYTES;
        f32x4 fst[NITS];                                                         // INF32: this stage's share of the next chunk's patch
        if constexpr (INF32) {
            const __amdgpu_buffer_rsrc_t rs = f32_rsrc(more_p ? chunk + 1 : chunk);
#pragma unroll
            for (int j = 0; j < NITS; ++j)
                if (part * NITS + j < NIT) fst[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)foff[part * NITS + j], 0, 0));
        }
        // the stage's STAPS * K16 steps (tap, 16-channel half), software-pipelined: the fragments of step i + 1 are requested
        // before the MFMAs of step i (left to itself the compiler reads every fragment right in front of its first MFMA and
        // waits for it there: four exposed LDS round trips per eight MFMAs)
        if constexpr (X3) {
            // tap by tap: A = wh xh, B = wl xh, C = wh xl, eight MFMAs each.  wl is requested in front of A's MFMAs, xl in front of
            // B's, the next tap's wh and xh (xh's registers are free once B has issued) in front of C's: every fragment read has at
            // least eight MFMAs (256 cycles) of the wave's own work in front of its first use.
            f32x4 wh[2][2], wl[2], xh[TM], xl[TM];
            auto ld_w = [&](auto dc, auto kc, f32x4 (&w)[2]) {
                constexpr int dxi = decltype(dc)::value, k16 = decltype(kc)::value;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    w[nt] = *reinterpret_cast<const f32x4*>(s_mem + ((wa ^ (k16 << 5)) + (dxi * BN + nt * 32) * PIXB));
            };
            auto ld_x = [&](auto dc, auto kc, f32x4 (&x)[TM]) {
                constexpr int tap = part * STAPS + decltype(dc)::value, k16 = decltype(kc)::value;
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) x[tm] = *reinterpret_cast<const f32x4*>(s_mem + (xa[tap][tm] ^ (k16 << 5)));
            };
            auto mm = [&](const f32x4 (&w)[2], const f32x4 (&x)[TM]) {
                if constexpr ((BF16P_ABL & 8) != 0) return;
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[tm][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p_as_f16x8(w[nt]), p_as_f16x8(x[tm]), acc[tm][nt], 0, 0, 0);
            };
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            constexpr int NSS = STAPS * 3;                                       // sub-steps of eight MFMAs
            static_assert(NDW + NDP <= NSS, "one DMA per sub-step");
            auto tapstep = [&](auto dc) {
                constexpr int dxi = decltype(dc)::value;
                if constexpr (!(BF16P_ABL & 4)) ld_w(dc, I1{}, wl);
                __builtin_amdgcn_sched_barrier(0);
                mm(wh[dxi & 1], xh);
                dma_slot(std::integral_constant<int, 3 * dxi>{});
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(BF16P_ABL & 4)) ld_x(dc, I1{}, xl);
                __builtin_amdgcn_sched_barrier(0);
                mm(wl, xh);
                dma_slot(std::integral_constant<int, 3 * dxi + 1>{});
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (dxi + 1 < STAPS && !(BF16P_ABL & 4)) {
                    ld_w(std::integral_constant<int, dxi + 1>{}, I0{}, wh[(dxi + 1) & 1]);
                    ld_x(std::integral_constant<int, dxi + 1>{}, I0{}, xh);
                }
                __builtin_amdgcn_sched_barrier(0);
                mm(wh[dxi & 1], xl);
                dma_slot(std::integral_constant<int, 3 * dxi + 2>{});
                __builtin_amdgcn_sched_barrier(0);
            };
            ld_w(I0{}, I0{}, wh[0]);
            ld_x(I0{}, I0{}, xh);
            if constexpr ((BF16P_ABL & 4) != 0) {                                // timing only: one fragment set per stage
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) { wh[1][nt] = wh[0][nt]; wl[nt] = wh[0][nt]; }
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) xl[tm] = xh[tm];
            }
            bp_for<0, STAPS>(tapstep);
            if constexpr (INF32) {
                if (more_p) {
#pragma unroll
                    for (int j = 0; j < NITS; ++j)
                        if (part * NITS + j < NIT && fdst[part * NITS + j] >= 0) f32_put(fst[j], (abuf ^ 1) * ABYTES + fdst[part * NITS + j]);
                }
            }
            if (!(BF16P_ABL & 2)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            return;
        }
        constexpr int NSTEP = STAPS * K16;
        f32x4 wf[2][2], xf[2][TM];
        auto load_frags = [&](auto stc, f32x4 (&wfs)[2], f32x4 (&xfs)[TM]) {
            constexpr int st = decltype(stc)::value, dxi = st / K16, k16 = st % K16;      // dxi: tap inside the stage
            constexpr int tap = part * STAPS + dxi;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                wfs[nt] = *reinterpret_cast<const f32x4*>(s_mem + ((wa ^ (k16 << 5)) + (dxi * BN + nt * 32) * PIXB));
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) xfs[tm] = *reinterpret_cast<const f32x4*>(s_mem + (xa[tap][tm] ^ (k16 << 5)));
        };
        auto step = [&](auto stc) {
            constexpr int st = decltype(stc)::value;
            if constexpr (st + 1 < NSTEP && !(BF16P_ABL & 4)) load_frags(std::integral_constant<int, st + 1>{}, wf[(st + 1) & 1], xf[(st + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((BF16P_ABL & 8) != 0) return;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[tm][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p_as_bf16x8(wf[(BF16P_ABL & 4) ? 0 : (st & 1)][nt]),
                                                                          p_as_bf16x8(xf[(BF16P_ABL & 4) ? 0 : (st & 1)][tm]), acc[tm][nt], 0, 0, 0);
            // this step's share of the stage's DMAs (all of them fit the first steps: NDW + NDP <= 2 NSTEP)
            constexpr int PER = (NDW + NDP + NSTEP - 1) / NSTEP;
            dma_slot(std::integral_constant<int, st * PER>{});
            if constexpr (PER > 1) dma_slot(std::integral_constant<int, st * PER + 1>{});
            static_assert(PER <= 2, "DMAs per step");
            __builtin_amdgcn_sched_barrier(0);
        };
        if (!(BF16P_ABL & 4) || s == 0) load_frags(std::integral_constant<int, 0>{}, wf[0], xf[0]);
        bp_for<0, NSTEP>(step);
        if (!(BF16P_ABL & 2)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    };
    for (int chunk = 0; chunk < NC; ++chunk) {
        stage_body(chunk, std::integral_constant<int, 0>{});
        if constexpr (SPC > 1) {
            stage_body(chunk, std::integral_constant<int, 1>{});
            stage_body(chunk, std::integral_constant<int, 2>{});
        }
        const int flip = (chunk & 1) ? -ABYTES : ABYTES;                        // the next chunk's patch is in the other buffer
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) xa[tap][tm] += flip;
    }

    if constexpr ((BF16P_ABL & 16) != 0) {                                       // ablation: no epilogue (one store keeps the accumulators alive)
        float sacc = 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[tm][nt][r];
        if (sacc == 12345.f) reinterpret_cast<float*>(a.out)[tid] = sacc;
        return;
    }
    // ---- epilogue: register r of acc[tm][nt] = channel nt * 32 + 8 (r >> 2) + 4 kh + (r & 3) of this lane's pixel
    const int nbase = cb * BN + wn * 64 + 4 * kh;
    const float oscale = X3 ? s_bias[BN] : 1.f;                                 // split form: the inverse of the weights' power-of-two scale
    if constexpr (POOL) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nbase + nt * 32 + 8 * g;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(s_bias + n - cb * BN);
#pragma unroll
                for (int tm = 0; tm < TM; tm += 2) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float m2 = fmaxf(acc[tm][nt][4 * g + e], acc[tm + 1][nt][4 * g + e]);        // rows y, y + 1
                        const float m4 = fmaxf(m2, ntk_dpp<0xB1>(m2));                                    // columns x, x ^ 1
                        if constexpr (X3 && !OUTF32) v[e] = __builtin_amdgcn_fmed3f(m4 * oscale + bv[e], 0.f, S3_MAX);   // ReLU and the split's range in one
                        else v[e] = fmaxf((X3 ? m4 * oscale : m4) + bv[e], 0.f);
                    }
                    if (opix[tm] >= 0 && (mcol & 1) == 0) {
                        if constexpr (OUTF32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + (size_t)opix[tm] * Cout + n) = v;
                        else if constexpr (X3) {                                // split map: [n / 16][hi x16 | lo x16]
                            ph16x4 oh, ol;
                            s3_split4(v, oh, ol);
                            _Float16* op = reinterpret_cast<_Float16*>(a.out) + (size_t)opix[tm] * (2 * Cout) + (n >> 4) * 32 + (n & 15);
                            *reinterpret_cast<ph16x4*>(op) = oh;
                            *reinterpret_cast<ph16x4*>(op + 16) = ol;
                        } else {
                            pbf16x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
                            *reinterpret_cast<pbf16x4*>(reinterpret_cast<__bf16*>(a.out) + (size_t)opix[tm] * Cout + n) = o;
                        }
                    }
                }
            }
    } else {
        // Un-pooled: a lane's four channels are 8 (16) bytes of a pixel whose neighbours in the wave are Cout * 2 (4) bytes away --
        // stored as they are, every store instruction touches 32 cache lines for 16 (32) bytes each (first version: the six
        // un-pooled layers ran 5 - 20 % SLOWER than the tile kernel, the three pooled ones 15 - 20 % faster).  So the wave's
        // 128 pixels x 64 columns go through ITS OWN slice of the (now idle) staging buffers once and come back row by row:
        // eight lanes = one pixel's 128 contiguous bytes, 16 bytes per lane per store.  Rows are 128 + 16 bytes apart.
        constexpr int RB = 144;
        constexpr int WREG = TM * 32 * RB;                                      // bytes per wave
        static_assert(NW * WREG <= LDSB, "the transpose image fits");
        unsigned char* tr = s_mem + wave * WREG;
        constexpr bool WIDE = OUTF32 || X3;                                     // four bytes per channel: fp32, or a split map's hi + lo
        constexpr int NPASS = WIDE ? 2 : 1;                                     // ... then one 32-column half at a time
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                if (WIDE && nt != ps) continue;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(s_bias + nbase - cb * BN + nt * 32 + 8 * g);
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if constexpr (X3 && !OUTF32) v[e] = __builtin_amdgcn_fmed3f(acc[tm][nt][4 * g + e] * oscale + bv[e], 0.f, S3_MAX);
                            else v[e] = fmaxf((X3 ? acc[tm][nt][4 * g + e] * oscale : acc[tm][nt][4 * g + e]) + bv[e], 0.f);
                        }
                        unsigned char* row = tr + (tm * 32 + mcol) * RB;
                        if constexpr (OUTF32) *reinterpret_cast<f32x4*>(row + (8 * g + 4 * kh) * 4) = v;
                        else if constexpr (X3) {                                // the half's two 16-channel groups: [hi x16 | lo x16] x 2
                            ph16x4 oh, ol;
                            s3_split4(v, oh, ol);
                            unsigned char* pr = row + (g >> 1) * 64 + ((g & 1) * 8 + 4 * kh) * 2;
                            *reinterpret_cast<ph16x4*>(pr) = oh;
                            *reinterpret_cast<ph16x4*>(pr + 32) = ol;
                        } else {
                            pbf16x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
                            *reinterpret_cast<pbf16x4*>(row + (nt * 32 + 8 * g + 4 * kh) * 2) = o;
                        }
                    }
                }
            }
            // read back: piece pi = lane + 64 j of the wave's TM * 32 rows x 8 pieces; row = pi >> 3 -> tile j >> 2, column 8 (j & 3) + (lane >> 3)
#pragma unroll
            for (int j = 0; j < TM * 4; ++j) {
                const int tmr = j >> 2, mr = 8 * (j & 3) + (lane >> 3), piece = lane & 7;
                const f32x4 v = *reinterpret_cast<const f32x4*>(tr + (tmr * 32 + mr) * RB + piece * 16);
                // the output pixel of row (tile tmr, column mr) is what lane mr holds in opix[tmr]: one cross-lane read instead of
                // redoing the tile -> pixel map (index arithmetic, table reads and a 64-bit multiply per row: that was 40 % of the epilogue's
                // instructions, and the epilogue is vector-ALU bound)
                const int opx = __builtin_amdgcn_ds_bpermute(mr << 2, opix[tmr]);
                if ((BF16P_ABL & 32) ? (opx >= 0 && v[0] == 12345.f) : (opx >= 0)) {       // (ablation bit 5: everything but the global stores)
                    const size_t op = (size_t)opx * Cout + cb * BN + wn * 64;
                    if constexpr (OUTF32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + op + ps * 32 + piece * 4) = v;
                    else if constexpr (X3) *reinterpret_cast<f32x4*>(reinterpret_cast<__bf16*>(a.out) + 2 * (op + ps * 32) + piece * 8) = v;
                    else *reinterpret_cast<f32x4*>(reinterpret_cast<__bf16*>(a.out) + op + piece * 8) = v;
                }
            }
        }
    }
}

template <int BN, int KC, int NW, bool X3 = false, bool INF32 = false>
int bf16p_launch(const Bf16pArgs& a0, int H, int W, int pool, int out_f32, hipStream_t st) {
    Bf16pArgs a = a0;
    if constexpr (X3 && !INF32) {
        if (W == 28 && !(H % 8 == 0)) {                                         // runs of 512 consecutive pixels over 28-wide rows
            if (pool || NW != 8 || H < 20) return NTK_ERR_UNSUPPORTED;
            a.bxN = 1; a.byN = 1;
            const long long NQ = (long long)a.frames * H * W, NS = (NQ + 511) / 512;
            a.NQ = 0; a.NS = (int)NS; a.nCB = a.Cout / BN;
            long long slots;
            if (a.nCB >= 8) slots = NS * (a.nCB / 8);
            else { const int per = 8 / a.nCB; slots = (NS + per - 1) / per; }
            const unsigned grid = (unsigned)(slots * 8);
            if constexpr (NW == 8) {
                if (out_f32) conv3x3_relu_bf16p_kernel<BN, 28, 21, 1, 32, false, true, 8, true><<<grid, 512, 0, st>>>(a);
                else conv3x3_relu_bf16p_kernel<BN, 28, 21, 1, 32, false, false, 8, true><<<grid, 512, 0, st>>>(a);
            }
            return NTK_OK;
        }
    }
    // sub-block shape, eight waves (512 pixels): 0 = 32x16 x1, 1 = 16x16 x2, 2 = 8x8 x8, 3 = 4x4 x32 (KC 16 only: its patches are 2.25x the
    // pixels); four waves (256 pixels): 0 = 32x8 x1, 1 = 16x16 x1, 2 = 8x8 x4
    int shape;
    if (NW == 8) shape = (W % 32 == 0 && H % 16 == 0) ? 0 : ((W % 16 == 0 && H % 16 == 0) ? 1 : ((W % 8 == 0 && H % 8 == 0) ? 2 : 3));
    else shape = (W % 32 == 0 && H % 8 == 0) ? 0 : ((W % 16 == 0 && H % 16 == 0) ? 1 : ((W % 8 == 0 && H % 8 == 0) ? 2 : 3));
    static const int TWs[4] = {32, 16, 8, 4}, THs8[4] = {16, 16, 8, 4}, THs4[4] = {8, 16, 8, 4}, NSUB8[4] = {1, 2, 8, 32}, NSUB4[4] = {1, 1, 4, 16};
    const int TWv = TWs[shape], THv = NW == 8 ? THs8[shape] : THs4[shape], NSUBv = NW == 8 ? NSUB8[shape] : NSUB4[shape];
    a.bxN = W / TWv;
    a.byN = H / THv;
    const long long NQ = (long long)a.frames * a.bxN * a.byN;
    const long long NS = (NQ + NSUBv - 1) / NSUBv;
    a.NQ = (int)NQ; a.NS = (int)NS; a.nCB = a.Cout / BN;
    long long slots;
    if (a.nCB >= 8) slots = NS * (a.nCB / 8);
    else { const int per = 8 / a.nCB; slots = (NS + per - 1) / per; }
    const unsigned grid = (unsigned)(slots * 8);
#define BF16P_GO(TW_, TH_, NSUB_)                                                                                             \
    do {                                                                                                                      \
        if (pool) {                                                                                                           \
            if (out_f32) conv3x3_relu_bf16p_kernel<BN, TW_, TH_, NSUB_, KC, true, true, NW, X3, INF32><<<grid, 64 * NW, 0, st>>>(a);     \
            else conv3x3_relu_bf16p_kernel<BN, TW_, TH_, NSUB_, KC, true, false, NW, X3, INF32><<<grid, 64 * NW, 0, st>>>(a);            \
        } else {                                                                                                              \
            if (out_f32) conv3x3_relu_bf16p_kernel<BN, TW_, TH_, NSUB_, KC, false, true, NW, X3, INF32><<<grid, 64 * NW, 0, st>>>(a);    \
            else conv3x3_relu_bf16p_kernel<BN, TW_, TH_, NSUB_, KC, false, false, NW, X3, INF32><<<grid, 64 * NW, 0, st>>>(a);           \
        }                                                                                                                     \
    } while (0)
    if constexpr (KC == 32 && NW == 8) {
        if (shape == 0) BF16P_GO(32, 16, 1);
        else if (shape == 1) BF16P_GO(16, 16, 2);
        else if (shape == 2) BF16P_GO(8, 8, 8);
        else return NTK_ERR_UNSUPPORTED;
    } else if constexpr (KC == 32) {
        if (shape == 0) BF16P_GO(32, 8, 1);
        else if (shape == 1) BF16P_GO(16, 16, 1);
        else if (shape == 2) BF16P_GO(8, 8, 4);
        else return NTK_ERR_UNSUPPORTED;
    } else {
        if (shape == 3 && !pool && NW == 8) {
            if (out_f32) conv3x3_relu_bf16p_kernel<BN, 4, 4, 32, KC, false, true><<<grid, 512, 0, st>>>(a);
            else conv3x3_relu_bf16p_kernel<BN, 4, 4, 32, KC, false, false><<<grid, 512, 0, st>>>(a);
        } else return NTK_ERR_UNSUPPORTED;
    }
#undef BF16P_GO
    return NTK_OK;
}

}  // namespace

// Which (BN, KC, NW) form a layer shape takes: columns in blocks of 128 where Cout allows (64 otherwise); 32-channel chunks on the
// rectangular sub-block shapes, 16-channel chunks on 4x4 sub-blocks (frames whose sides are multiples of 4 but not of 8); a layer of
// 64 -> 64 channels (conv1_2: two chunks) on four-wave workgroups, two per CU
static int bf16p_form(int H, int W, int cin, int cout, int pool, int* bn, int* kc, int* nw) {
    if ((H % 4) || (W % 4) || cin % 32 || cout % 64) return 0;
    const bool rect = (W % 8 == 0 && H % 8 == 0);
    if (!rect && pool) return 0;
    *nw = (rect && cin <= 64 && cout == 64) ? 4 : 8;       // (measured: conv2_1, 64 -> 128 columns, LOSES 4 % as two 64-column halves that both stage the patch)
    *bn = (*nw == 8 && cout % 128 == 0) ? 128 : 64;
    *kc = rect ? 32 : 16;
    const int nCB = cout / *bn;
    if (!(nCB <= 8 ? (8 % nCB) == 0 : (nCB % 8) == 0)) return 0;
    return 1;
}

extern "C" size_t ntk_vgg_bf16p_packed_elems(int cin, int cout) { return (size_t)9 * cin * cout; }

// weights for ntk_vgg_conv3x3_relu_bf16p: the packing depends on the layer's FRAME shape only through the chunk size (H, W
// multiples of 8: 32-channel chunks; else 16)
extern "C" int ntk_vgg_pack_weights_bf16p(const float* w_hwio, void* w_packed_bf16, int cin, int cout, int H, int W, void* stream) {
    NTK_REQUIRE(w_hwio && w_packed_bf16, NTK_ERR_BAD_PTR, "ntk_vgg_pack_weights_bf16p: null pointer");
    int bn = 0, kc = 0, nw = 0;
    NTK_REQUIRE(bf16p_form(H, W, cin, cout, 0, &bn, &kc, &nw), NTK_ERR_UNSUPPORTED,
                "ntk_vgg_pack_weights_bf16p: cin=%d (multiple of 32) cout=%d (multiple of 64) H=%d W=%d (multiples of 4)", cin, cout, H, W);
    __bf16* wq = reinterpret_cast<__bf16*>(w_packed_bf16);
    hipStream_t st = (hipStream_t)stream;
    if (bn == 128 && kc == 32) bf16p_pack_kernel<128, 32><<<1024, 256, 0, st>>>(w_hwio, wq, cin, cout);
    else if (bn == 64 && kc == 32) bf16p_pack_kernel<64, 32><<<1024, 256, 0, st>>>(w_hwio, wq, cin, cout);
    else if (bn == 128) bf16p_pack_kernel<128, 16><<<1024, 256, 0, st>>>(w_hwio, wq, cin, cout);
    else bf16p_pack_kernel<64, 16><<<1024, 256, 0, st>>>(w_hwio, wq, cin, cout);
    NTK_CHECK_LAUNCH("ntk_vgg_pack_weights_bf16p");
    return NTK_OK;
}

// 1 when ntk_vgg_conv3x3_relu_bf16p takes the layer shape (else ntk_vgg_conv3x3_relu_bf16 runs it)
extern "C" int ntk_vgg_bf16p_supported(int H, int W, int cin, int cout, int fuse_pool) {
    int bn = 0, kc = 0, nw = 0;
    return bf16p_form(H, W, cin, cout, fuse_pool, &bn, &kc, &nw);
}

extern "C" int ntk_vgg_conv3x3_relu_bf16p(const void* in_bf16, const void* w_packed_bf16p, const float* bias, void* out,
                                          int frames, int H, int W, int cin, int cout, int fuse_pool, int out_f32, void* stream) {
    NTK_REQUIRE(in_bf16 && w_packed_bf16p && bias && out, NTK_ERR_BAD_PTR, "ntk_vgg_conv3x3_relu_bf16p: null pointer");
    NTK_REQUIRE(ntk_aligned16(in_bf16) && ntk_aligned16(w_packed_bf16p) && ntk_aligned16(out) && ntk_aligned16(bias), NTK_ERR_BAD_PTR,
                "ntk_vgg_conv3x3_relu_bf16p: pointers must be 16-byte aligned");
    int bn = 0, kc = 0, nw = 0;
    NTK_REQUIRE(frames > 0 && H > 0 && W > 0 && bf16p_form(H, W, cin, cout, fuse_pool, &bn, &kc, &nw), NTK_ERR_UNSUPPORTED,
                "ntk_vgg_conv3x3_relu_bf16p: frames=%d H=%d W=%d cin=%d cout=%d pool=%d (H, W multiples of 4 -- of 8 with the pool; "
                "cin a multiple of 32, cout of 64)", frames, H, W, cin, cout, fuse_pool);
    NTK_REQUIRE((unsigned long long)2 * H * W * cin * sizeof(__bf16) <= 0x7ffffff0ull && (long long)frames * H * W < (1ll << 31), NTK_ERR_UNSUPPORTED,
                "ntk_vgg_conv3x3_relu_bf16p: frame too large for 32-bit offsets");
    Bf16pArgs a;
    a.in = reinterpret_cast<const __bf16*>(in_bf16); a.wq = reinterpret_cast<const __bf16*>(w_packed_bf16p); a.bias = bias; a.out = out;
    a.frames = frames; a.H = H; a.W = W; a.Cin = cin; a.Cout = cout;
    a.bxN = a.byN = a.NQ = a.NS = a.nCB = 0;
    int rc;
    hipStream_t st = (hipStream_t)stream;
    if (nw == 4) rc = bf16p_launch<64, 32, 4>(a, H, W, fuse_pool, out_f32, st);
    else if (bn == 128 && kc == 32) rc = bf16p_launch<128, 32, 8>(a, H, W, fuse_pool, out_f32, st);
    else if (bn == 64 && kc == 32) rc = bf16p_launch<64, 32, 8>(a, H, W, fuse_pool, out_f32, st);
    else if (bn == 128) rc = bf16p_launch<128, 16, 8>(a, H, W, fuse_pool, out_f32, st);
    else rc = bf16p_launch<64, 16, 8>(a, H, W, fuse_pool, out_f32, st);
    NTK_REQUIRE(rc == NTK_OK, rc, "ntk_vgg_conv3x3_relu_bf16p: no instantiation for this shape");
    NTK_CHECK_LAUNCH("ntk_vgg_conv3x3_relu_bf16p");
    return NTK_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The SPLIT form (X3): the fp32 trunk on the bf16 matrix pipe.  Maps are "split" maps: [frames][H][W][C / 16][hi x16 | lo x16] bf16.
// ---------------------------------------------------------------------------------------------------------------------------------
static int split3_form(int H, int W, int cin, int cout, int pool, int* bn, int* nw) {
    if (H <= 0 || W <= 0 || cin % 16 || cout % 64) return 0;
    const bool rect = (W % 8 == 0 && H % 8 == 0);
    if (!rect && (pool || W != 28 || H < 20)) return 0;                          // 28-wide maps: runs of rows (conv4_x), un-pooled
    *nw = (rect && cin <= 64 && cout == 64) ? 4 : 8;
    *bn = (*nw == 8 && cout % 128 == 0) ? 128 : 64;
    const int nCB = cout / *bn;
    if (!(nCB <= 8 ? (8 % nCB) == 0 : (nCB % 8) == 0)) return 0;
    return 1;
}

extern "C" size_t ntk_vgg_split3_packed_elems(int cin, int cout) { return (size_t)18 * cin * cout + 8; }    // + 16 bytes: the scales

extern "C" int ntk_vgg_split3_supported(int H, int W, int cin, int cout, int fuse_pool) {
    int bn = 0, nw = 0;
    return split3_form(H, W, cin, cout, fuse_pool, &bn, &nw);
}

// HWIO fp32 weights -> the stage images of ntk_vgg_conv3x3_relu_split3 (bf16 high and low parts; 18 cin cout elements)
extern "C" int ntk_vgg_pack_weights_split3(const float* w_hwio, void* w_packed, int cin, int cout, int H, int W, void* stream) {
    NTK_REQUIRE(w_hwio && w_packed, NTK_ERR_BAD_PTR, "ntk_vgg_pack_weights_split3: null pointer");
    int bn = 0, nw = 0;
    NTK_REQUIRE(split3_form(H, W, cin, cout, 0, &bn, &nw), NTK_ERR_UNSUPPORTED,
                "ntk_vgg_pack_weights_split3: cin=%d (multiple of 16) cout=%d (multiple of 64) H=%d W=%d", cin, cout, H, W);
    _Float16* wq = reinterpret_cast<_Float16*>(w_packed);
    hipStream_t st = (hipStream_t)stream;
    unsigned* tail = reinterpret_cast<unsigned*>(wq + (size_t)18 * cin * cout);
    if (hipMemsetAsync(tail, 0, 16, st) != hipSuccess) return NTK_ERR_HIP;
    split3_absmax_kernel<<<256, 256, 0, st>>>(w_hwio, (size_t)9 * cin * cout, tail);
    split3_scale_kernel<<<1, 1, 0, st>>>(tail);
    if (bn == 128) split3_pack_kernel<128><<<1024, 256, 0, st>>>(w_hwio, wq, cin, cout);
    else split3_pack_kernel<64><<<1024, 256, 0, st>>>(w_hwio, wq, cin, cout);
    NTK_CHECK_LAUNCH("ntk_vgg_pack_weights_split3");
    return NTK_OK;
}

// conv3x3 SAME + bias + ReLU (+ 2x2 max-pool) of a split map (in_f32 = 1: of an fp32 NHWC map, split by the staging; the four-wave
// form only); out: a split map (out_f32 = 0) or fp32 NHWC (out_f32 = 1)
extern "C" int ntk_vgg_conv3x3_relu_split3(const void* in_split, const void* w_packed, const float* bias, void* out,
                                           int frames, int H, int W, int cin, int cout, int fuse_pool, int in_f32, int out_f32, void* stream) {
    NTK_REQUIRE(in_split && w_packed && bias && out, NTK_ERR_BAD_PTR, "ntk_vgg_conv3x3_relu_split3: null pointer");
    NTK_REQUIRE(ntk_aligned16(in_split) && ntk_aligned16(w_packed) && ntk_aligned16(out) && ntk_aligned16(bias), NTK_ERR_BAD_PTR,
                "ntk_vgg_conv3x3_relu_split3: pointers must be 16-byte aligned");
    int bn = 0, nw = 0;
    NTK_REQUIRE(frames > 0 && H > 0 && W > 0 && split3_form(H, W, cin, cout, fuse_pool, &bn, &nw), NTK_ERR_UNSUPPORTED,
                "ntk_vgg_conv3x3_relu_split3: frames=%d H=%d W=%d cin=%d cout=%d pool=%d (H, W multiples of 8, or W = 28 and H >= 20 "
                "without the pool; cin a multiple of 16, cout of 64)", frames, H, W, cin, cout, fuse_pool);
    NTK_REQUIRE((unsigned long long)2 * H * W * cin * 4 <= 0x7ffffff0ull && (long long)frames * H * W < (1ll << 31), NTK_ERR_UNSUPPORTED,
                "ntk_vgg_conv3x3_relu_split3: frame too large for 32-bit offsets");
    Bf16pArgs a;
    a.in = reinterpret_cast<const __bf16*>(in_split); a.wq = reinterpret_cast<const __bf16*>(w_packed); a.bias = bias; a.out = out;
    a.frames = frames; a.H = H; a.W = W; a.Cin = 2 * cin; a.Cout = cout;       // to the kernel: a bf16 map of 2 cin channels
    a.bxN = a.byN = a.NQ = a.NS = a.nCB = 0;
    int rc;
    hipStream_t st = (hipStream_t)stream;
    NTK_REQUIRE(!in_f32 || nw == 4, NTK_ERR_UNSUPPORTED,
                "ntk_vgg_conv3x3_relu_split3: an fp32 input map is read by the four-wave form only (cin <= 64, cout = 64, H and W multiples of 8)");
    if (nw == 4 && in_f32) rc = bf16p_launch<64, 32, 4, true, true>(a, H, W, fuse_pool, out_f32, st);
    else if (nw == 4) rc = bf16p_launch<64, 32, 4, true>(a, H, W, fuse_pool, out_f32, st);
    else if (bn == 128) rc = bf16p_launch<128, 32, 8, true>(a, H, W, fuse_pool, out_f32, st);
    else rc = bf16p_launch<64, 32, 8, true>(a, H, W, fuse_pool, out_f32, st);
    NTK_REQUIRE(rc == NTK_OK, rc, "ntk_vgg_conv3x3_relu_split3: no instantiation for this shape");
    NTK_CHECK_LAUNCH("ntk_vgg_conv3x3_relu_split3");
    return NTK_OK;
}
