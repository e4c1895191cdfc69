// fp32 MFMA tile kernels for gfx950 (v_mfma_f32_32x32x2_f32):
//   * conv3x3 SAME + bias + ReLU (+ fused 2x2/2 max-pool) as an implicit GEMM
//     over NHWC activations            -- VGG conv1_1..conv4_3 (vgg.py:155-161)
//   * C = A * B^T (+bias)              -- hoisted LSTM input projection
//   * C (+)= A^T * B (k-major, split-K) -- BPTT weight-gradient contractions
//
// Tile: 128 x BN x 32 per 256-thread workgroup (4 waves as 2x2, each wave a
// 64 x BN/2 block of 32x32 MFMA tiles).  Operands are staged global -> VGPR ->
// LDS; the next K-tile's global loads are in flight while the current tile's
// 64 MFMAs per wave issue.  fp32 MFMA runs at 64 cycles per instruction per
// SIMD, so one K-tile is ~4096 MFMA cycles per wave against 8 x 16-byte loads
// per thread: the loop is matrix-pipe bound by construction.  Measured on
// MI355X (profiles/): co-resident waves of one SIMD run in lockstep
// and stall at their barriers together, so occupancy decides the pipe's duty
// cycle -- the VGPR-staged kernel (variant 2) keeps ONE LDS buffer (two barriers per
// K-tile, 38 KB) so that 3 workgroups fit a CU: 126-128 TFLOP/s per layer (two LDS buffers at 2 workgroups/CU,
// static wave priority, 4 workgroups/CU with spills were measured slower and removed).
// Variant 4 (default of the direct path): LDS-DMA staging (global_load_lds_dwordx4) frees the 32 staging VGPRs ->
// 128 VGPRs, 32 KB LDS, FOUR workgroups per CU: 132-133 TFLOP/s.  The default trunk is the Winograd kernel
// (conv_wino.hip); this file serves algo="direct", tiny-Cin / odd shapes and the plain GEMMs around the recurrence.
#include "common.h"
#include "conv_common.h"

namespace {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDT = BK + 4;   // LDS row stride (floats): conflict-free ds_read_b128 (36*i mod 64 distinct slots)

// ---------------------------------------------------------------------------
// one K-tile of MFMAs from LDS images As[BM][LDT], Bs[BN][LDT] (K contiguous)
// lane l supplies A[row l&31][k] and B[k][col l&31] with k picked by l>>5.
// ---------------------------------------------------------------------------
template <int BN>
__device__ __forceinline__ void mma_ktile(const float* __restrict__ As, const float* __restrict__ Bs,
                                          f32x16 (&acc)[2][BN / 64], int wm, int wn, int lane) {
    constexpr int TN = BN / 64;
    const int i = lane & 31, kh = lane >> 5;
    const float* ap = As + (wm * 64 + i) * LDT + kh * 4;
    const float* bp = Bs + (wn * (BN / 2) + i) * LDT + kh * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 a[2], b[TN];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) a[tm] = *reinterpret_cast<const f32x4*>(ap + tm * 32 * LDT + q * 8);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) b[tn] = *reinterpret_cast<const f32x4*>(bp + tn * 32 * LDT + q * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][e], b[tn][e], acc[tm][tn], 0, 0, 0);
    }
}

// generic double-buffered pipeline.  LA/LB: functors  f32x4 operator()(int slot, int kt)
// slot = which of the thread's rows (A: 4 rows, B: BN/32 rows); LDS row = (tid>>3) + 32*slot, k-group = tid&7
template <int BN, class LA, class LB>
__device__ __forceinline__ void gemm_pipeline_sb(LA& la, LB& lb, int nk, float* lds, f32x16 (&acc)[2][BN / 64]) {
    // single LDS buffer: registers hold tile kt+1 while tile kt is consumed; two barriers per K-tile,
    // half the LDS footprint -> one more workgroup per CU
    constexpr int NB = BN / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = tid >> 3, kg = tid & 7;
    float* As = lds;
    float* Bs = lds + BM * LDT;
    f32x4 ra[4], rb[NB];
#pragma unroll
    for (int s = 0; s < 4; ++s) ra[s] = la(s, 0);
#pragma unroll
    for (int s = 0; s < NB; ++s) rb[s] = lb(s, 0);
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int s = 0; s < 4; ++s) *reinterpret_cast<f32x4*>(As + (lrow + 32 * s) * LDT + kg * 4) = ra[s];
#pragma unroll
        for (int s = 0; s < NB; ++s) *reinterpret_cast<f32x4*>(Bs + (lrow + 32 * s) * LDT + kg * 4) = rb[s];
        __syncthreads();
        if (kt + 1 < nk) {
#pragma unroll
            for (int s = 0; s < 4; ++s) ra[s] = la(s, kt + 1);
#pragma unroll
            for (int s = 0; s < NB; ++s) rb[s] = lb(s, kt + 1);
        }
        mma_ktile<BN>(As, Bs, acc, wm, wn, lane);
        __syncthreads();
    }
}

// two LDS buffers (used by the plain GEMM kernels below: their tiles are small and they run at 2 workgroups/CU)
template <int BN, class LA, class LB>
__device__ __forceinline__ void gemm_pipeline(LA& la, LB& lb, int nk, float* lds, f32x16 (&acc)[2][BN / 64]) {
    constexpr int NB = BN / 32;
    constexpr int TILE = (BM + BN) * LDT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = tid >> 3, kg = tid & 7;
    f32x4 ra[4], rb[NB];
#pragma unroll
    for (int s = 0; s < 4; ++s) ra[s] = la(s, 0);
#pragma unroll
    for (int s = 0; s < NB; ++s) rb[s] = lb(s, 0);
    {
        float* As = lds;
        float* Bs = lds + BM * LDT;
#pragma unroll
        for (int s = 0; s < 4; ++s) *reinterpret_cast<f32x4*>(As + (lrow + 32 * s) * LDT + kg * 4) = ra[s];
#pragma unroll
        for (int s = 0; s < NB; ++s) *reinterpret_cast<f32x4*>(Bs + (lrow + 32 * s) * LDT + kg * 4) = rb[s];
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1 < nk);
        if (more) {
#pragma unroll
            for (int s = 0; s < 4; ++s) ra[s] = la(s, kt + 1);
#pragma unroll
            for (int s = 0; s < NB; ++s) rb[s] = lb(s, kt + 1);
        }
        const float* As = lds + (kt & 1) * TILE;
        mma_ktile<BN>(As, As + BM * LDT, acc, wm, wn, lane);
        if (more) {
            float* An = lds + ((kt + 1) & 1) * TILE;
            float* Bn = An + BM * LDT;
#pragma unroll
            for (int s = 0; s < 4; ++s) *reinterpret_cast<f32x4*>(An + (lrow + 32 * s) * LDT + kg * 4) = ra[s];
#pragma unroll
            for (int s = 0; s < NB; ++s) *reinterpret_cast<f32x4*>(Bn + (lrow + 32 * s) * LDT + kg * 4) = rb[s];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// conv3x3 implicit GEMM.  GEMM row m <-> output pixel, ordered so that the 4
// pixels of every 2x2 pooling window are 4 consecutive rows (= the 4
// consecutive accumulator registers r&3 of one lane): rows are grouped in 4x4
// pixel patches, patch-major over (frame, py, px); inside a patch
// q = m & 15 -> window (q>>3, (q>>2)&1), pixel-in-window ((q>>1)&1, q&1).
// ---------------------------------------------------------------------------
template <int BN, bool SMALLC, bool POOL, bool OUTBF16 = false>
__global__ __launch_bounds__(256, 3) void conv3x3_relu_kernel(
    const float* __restrict__ in, const float* __restrict__ wp, const float* __restrict__ bias,
    float* __restrict__ out, int npatch, int H, int W, int Cin, int Cout, int Kp) {
    constexpr int TN = BN / 64;
    __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * LDT];
    __shared__ int s_pix[BM], s_yx[BM], s_ppix[BM];

    const int tid = threadIdx.x;
    // XCD-aware tile order (1-D grid): workgroups b, b+8, b+16, ... share an XCD (round-robin dispatch, speed
    // only).  Inside an XCD consecutive workgroups take the column tiles of ONE row tile, so the A rows that
    // every column tile re-reads are served by that XCD's L2 instead of being fetched once per column tile.
    const int ctiles = Cout / BN;
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
    const int rt = (li / ctiles) * 8 + xcd;
    const int m0 = rt * BM;
    const int n0 = (li % ctiles) * BN;
    if (m0 >= npatch * 16) return;
    if (tid < BM) {
        ConvRowInfo ri = conv_row_info(m0 + tid, npatch, H, W);
        s_pix[tid] = ri.pix; s_yx[tid] = ri.yx; s_ppix[tid] = ri.ppix;
    }
    __syncthreads();

    const int lrow = tid >> 3, kg = tid & 7;
    int rpix[4], ry[4], rx[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        rpix[s] = s_pix[lrow + 32 * s];
        const int yx = s_yx[lrow + 32 * s];
        ry[s] = yx >> 16; rx[s] = yx & 0xffff;
    }

    auto la = [&](int s, int kt) -> f32x4 {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if constexpr (!SMALLC) {
            // K order: 32-channel chunk outer, the 9 taps inner -- the nine shifted reads of one channel chunk are
            // consecutive K-tiles, so their overlapping 128-B lines are re-read while still L2 (mostly L1) resident
            const int chunk = kt / 9, tap = kt - chunk * 9;
            const int c0 = chunk * BK;
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            const int yy = ry[s] + dy, xx = rx[s] + dx;
            if (rpix[s] >= 0 && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
                const size_t off = (size_t)(rpix[s] + dy * W + dx) * Cin + c0 + kg * 4;
                v = *reinterpret_cast<const f32x4*>(in + off);
            }
        } else {
            // tiny Cin (conv1_1, Cin=3): whole K = 9*Cin fits one K-tile; scalar gather
            if (rpix[s] >= 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = kg * 4 + e;
                    if (k < 9 * Cin) {
                        const int tap = k / Cin, c = k - tap * Cin;
                        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
                        const int yy = ry[s] + dy, xx = rx[s] + dx;
                        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                            v[e] = in[(size_t)(rpix[s] + dy * W + dx) * Cin + c];
                    }
                }
            }
        }
        return v;
    };
    auto lb = [&](int s, int kt) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(wp + (size_t)(n0 + lrow + 32 * s) * Kp + kt * BK + kg * 4);
    };

    f32x16 acc[2][TN];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    gemm_pipeline_sb<BN>(la, lb, Kp / BK, lds, acc);

    // epilogue: bias + ReLU (+ 2x2 max over the 4 consecutive rows of a window)
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int kh = lane >> 5, col = lane & 31;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + wn * (BN / 2) + tn * 32 + col;
        const float bv = bias[n];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            const int mbase = wm * 64 + tm * 32 + 4 * kh;
            if constexpr (!POOL) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = mbase + (r & 3) + 8 * (r >> 2);
                    const int pix = s_pix[ml];
                    if (pix >= 0) {
                        const float v = fmaxf(acc[tm][tn][r] + bv, 0.f);
                        if constexpr (OUTBF16) reinterpret_cast<__bf16*>(out)[(size_t)pix * Cout + n] = (__bf16)v;
                        else out[(size_t)pix * Cout + n] = v;
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ml = mbase + 8 * g;
                    const int pp = s_ppix[ml];
                    float v = fmaxf(fmaxf(acc[tm][tn][4 * g], acc[tm][tn][4 * g + 1]),
                                    fmaxf(acc[tm][tn][4 * g + 2], acc[tm][tn][4 * g + 3]));
                    if (pp >= 0) {
                        const float o = fmaxf(v + bv, 0.f);
                        if constexpr (OUTBF16) reinterpret_cast<__bf16*>(out)[(size_t)pp * Cout + n] = (__bf16)o;
                        else out[(size_t)pp * Cout + n] = o;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Variant 4: operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR staging, no ds_write),
// one un-padded XOR-swizzled LDS buffer, 4 workgroups per CU.  LDS image: row r at byte r*128, its 16-byte
// chunk c stored in slot c ^ ((r >> 1) & 7): every ds_read_b128 lane group of the fragment reads then touches 16
// distinct 4-bank slots (rows of one parity x 8 slots).  One DMA wave-instruction fills 8 rows (1 KB,
// lane-linear), so the swizzle is applied to the per-lane SOURCE address.  Zero padding (image border, rows past
// the end) is fetched from a zero page.
// ---------------------------------------------------------------------------
__device__ __attribute__((aligned(128))) float g_zero_page[32];

template <int BN>
__device__ __forceinline__ void mma_ktile_swz(const float* __restrict__ As, const float* __restrict__ Bs,
                                              f32x16 (&acc)[2][BN / 64], int wm, int wn, int lane) {
    constexpr int TN = BN / 64;
    const int i = lane & 31, kh = lane >> 5, f = (i >> 1) & 7;
    const float* ap = As + (wm * 64 + i) * BK;
    const float* bp = Bs + (wn * (BN / 2) + i) * BK;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int slot = ((2 * q + kh) ^ f) * 4;
        f32x4 a[2], b[TN];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) a[tm] = *reinterpret_cast<const f32x4*>(ap + tm * 32 * BK + slot);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) b[tn] = *reinterpret_cast<const f32x4*>(bp + tn * 32 * BK + slot);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][e], b[tn][e], acc[tm][tn], 0, 0, 0);
    }
}

__device__ __forceinline__ void lds_dma16(const float* src, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int BN, bool POOL>
__global__ __launch_bounds__(256, 4) void conv3x3_relu_dma_kernel(
    const float* __restrict__ in, const float* __restrict__ wp, const float* __restrict__ bias,
    float* __restrict__ out, int npatch, int H, int W, int Cin, int Cout, int Kp) {
    constexpr int TN = BN / 64, NBI = BN / 32;      // B DMA instructions per wave
    constexpr int TILE = (BM + BN) * BK;
    __shared__ __attribute__((aligned(1024))) float lds[TILE];
    __shared__ int s_pix[BM], s_yx[BM], s_ppix[BM];

    const int tid = threadIdx.x;
    const int ctiles = Cout / BN;
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
    const int rt = (li / ctiles) * 8 + xcd;
    const int m0 = rt * BM;
    const int n0 = (li % ctiles) * BN;
    if (m0 >= npatch * 16) return;
    if (tid < BM) {
        ConvRowInfo ri = conv_row_info(m0 + tid, npatch, H, W);
        s_pix[tid] = ri.pix; s_yx[tid] = ri.yx; s_ppix[tid] = ri.ppix;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // DMA roles: instruction jj of this wave fills tile rows (wave*4 + jj)*8 .. +7; lane -> (row = lane>>3, slot = lane&7)
    const int lr = lane >> 3, slot = lane & 7;
    int rpix[4], ry[4], rx[4], achunk[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int r = (wave * 4 + jj) * 8 + lr;
        rpix[jj] = s_pix[r];
        const int yx = s_yx[r];
        ry[jj] = yx >> 16; rx[jj] = yx & 0xffff;
        achunk[jj] = (slot ^ ((r >> 1) & 7)) * 4;
    }
    const float* bsrc[NBI];
#pragma unroll
    for (int jj = 0; jj < NBI; ++jj) {
        const int r = (wave * NBI + jj) * 8 + lr;
        bsrc[jj] = wp + (size_t)(n0 + r) * Kp + (slot ^ ((r >> 1) & 7)) * 4;
    }
    float* As = lds;
    float* Bs = lds + BM * BK;

    f32x16 acc[2][TN];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    const int nk = Kp / BK;
    auto issue = [&](int kt, float* Ad, float* Bd) {
        const int chunk = kt / 9, tap = kt - chunk * 9;
        const int c0 = chunk * BK;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int yy = ry[jj] + dy, xx = rx[jj] + dx;
            const bool ok = rpix[jj] >= 0 && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            const float* src = ok ? in + (size_t)(rpix[jj] + dy * W + dx) * Cin + c0 + achunk[jj] : g_zero_page + (lane & 7) * 4;
            lds_dma16(src, Ad + (wave * 4 + jj) * 8 * BK);
        }
#pragma unroll
        for (int jj = 0; jj < NBI; ++jj) lds_dma16(bsrc[jj] + kt * BK, Bd + (wave * NBI + jj) * 8 * BK);
    };
    for (int kt = 0; kt < nk; ++kt) {
        if (kt > 0) __syncthreads();                   // every wave has finished reading the previous tile
        issue(kt, As, Bs);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // all four waves' DMA has landed
        mma_ktile_swz<BN>(As, Bs, acc, wm, wn, lane);
    }

    const int kh = lane >> 5, col = lane & 31;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + wn * (BN / 2) + tn * 32 + col;
        const float bv = bias[n];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            const int mbase = wm * 64 + tm * 32 + 4 * kh;
            if constexpr (!POOL) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pix = s_pix[mbase + (r & 3) + 8 * (r >> 2)];
                    if (pix >= 0) out[(size_t)pix * Cout + n] = fmaxf(acc[tm][tn][r] + bv, 0.f);
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int pp = s_ppix[mbase + 8 * g];
                    const float v = fmaxf(fmaxf(acc[tm][tn][4 * g], acc[tm][tn][4 * g + 1]),
                                          fmaxf(acc[tm][tn][4 * g + 2], acc[tm][tn][4 * g + 3]));
                    if (pp >= 0) out[(size_t)pp * Cout + n] = fmaxf(v + bv, 0.f);
                }
            }
        }
    }
}

// HWIO [3,3,Cin,Cout] -> [Cout][Kp].  Cin % 32 == 0: k = (c/32)*288 + tap*32 + c%32 (chunk outer, tap inner);
// tiny Cin (conv1_1): k = tap*Cin + c, zero padded to Kp.
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int Kp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cout * Kp) return;
    const int n = idx / Kp, k = idx - n * Kp;
    float v = 0.f;
    if (k < 9 * Cin) {
        int tap, c;
        if ((Cin % BK) == 0) { const int chunk = k / (9 * BK), r = k - chunk * 9 * BK; tap = r / BK; c = chunk * BK + (r - tap * BK); }
        else { tap = k / Cin; c = k - tap * Cin; }
        v = w[(size_t)(tap * Cin + c) * Cout + n];   // HWIO flat index = (tap*Cin + c)*Cout + n
    }
    wp[idx] = v;
}

// ---------------------------------------------------------------------------
// C[M,N] = A[M,K] * B[N,K]^T + bias
// ---------------------------------------------------------------------------
template <int BN>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
    const float* __restrict__ bias, float* __restrict__ C, int ldc, int M, int N, int K) {
    constexpr int TN = BN / 64;
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDT];
    const int tid = threadIdx.x;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int lrow = tid >> 3, kg = tid & 7;
    auto la = [&](int s, int kt) -> f32x4 {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int m = m0 + lrow + 32 * s, k = kt * BK + kg * 4;
        if (m < M && k < K) v = *reinterpret_cast<const f32x4*>(A + (size_t)m * lda + k);
        return v;
    };
    auto lb = [&](int s, int kt) -> f32x4 {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int n = n0 + lrow + 32 * s, k = kt * BK + kg * 4;
        if (n < N && k < K) v = *reinterpret_cast<const f32x4*>(B + (size_t)n * ldb + k);
        return v;
    };
    f32x16 acc[2][TN];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
    gemm_pipeline<BN>(la, lb, (K + BK - 1) / BK, lds, acc);

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int kh = lane >> 5, col = lane & 31;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + wn * (BN / 2) + tn * 32 + col;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + tm * 32 + 4 * kh + (r & 3) + 8 * (r >> 2);
                if (m < M) C[(size_t)m * ldc + n] = acc[tm][tn][r] + bv;
            }
    }
}

// ---------------------------------------------------------------------------
// slab[z][M,N] = sum_{k in split z} A[k,M]^T B[k,N]   (operands k-major)
// LDS images [BK][128+4]: row index contiguous, fragments by ds_read_b32
// (32 consecutive lanes -> 32 consecutive banks).
// ---------------------------------------------------------------------------
constexpr int LDM = BM + 4;

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
    float* __restrict__ slab, int M, int N, int K, int kchunk) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * BK * LDM];
    constexpr int TILE = 2 * BK * LDM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BM;
    const int kbeg = blockIdx.z * kchunk;
    const int kend = min(K, kbeg + kchunk);
    const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
    const int lk = tid >> 5, cg = tid & 31;   // k row (0..7)+8*s, column group of 4

    auto ld = [&](const float* P, int ldp, int c0, int lim, int s, int kt) -> f32x4 {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int k = kbeg + kt * BK + lk + 8 * s, c = c0 + cg * 4;
        if (k < kend && c < lim) v = *reinterpret_cast<const f32x4*>(P + (size_t)k * ldp + c);
        return v;
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    f32x4 ra[4], rb[4];
    auto fetch = [&](int kt) {
#pragma unroll
        for (int s = 0; s < 4; ++s) { ra[s] = ld(A, lda, m0, M, s, kt); rb[s] = ld(B, ldb, n0, N, s, kt); }
    };
    auto stash = [&](int buf) {
        float* As = lds + buf * TILE;
        float* Bs = As + BK * LDM;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            *reinterpret_cast<f32x4*>(As + (lk + 8 * s) * LDM + cg * 4) = ra[s];
            *reinterpret_cast<f32x4*>(Bs + (lk + 8 * s) * LDM + cg * 4) = rb[s];
        }
    };
    if (nk > 0) { fetch(0); stash(0); }
    __syncthreads();
    const int i = lane & 31, kh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) fetch(kt + 1);
        const float* As = lds + (kt & 1) * TILE;
        const float* Bs = As + BK * LDM;
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            float a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = As[(2 * s + kh) * LDM + wm * 64 + t * 32 + i];
                b[t] = Bs[(2 * s + kh) * LDM + wn * 64 + t * 32 + i];
            }
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
        }
        if (more) stash((kt + 1) & 1);
        __syncthreads();
    }
    float* out = slab + (size_t)blockIdx.z * M * N;
    const int col = lane & 31;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int n = n0 + wn * 64 + tn * 32 + col;
        if (n >= N) continue;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + tm * 32 + 4 * kh + (r & 3) + 8 * (r >> 2);
                if (m < M) out[(size_t)m * N + n] = acc[tm][tn][r];
            }
    }
}

__global__ void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ C, int ldc,
                                   int M, int N, int splits, int accumulate) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * N) return;
    const int m = idx / N, n = idx - m * N;
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += slab[(size_t)z * M * N + idx];
    float* c = C + (size_t)m * ldc + n;
    *c = accumulate ? (*c + s) : s;
}

}  // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" int ntk_vgg_packed_k(int cin) { return ((9 * cin + BK - 1) / BK) * BK; }

extern "C" int ntk_vgg_pack_weights(const float* w_hwio, float* w_packed, int cin, int cout, void* stream) {
    NTK_REQUIRE(w_hwio && w_packed, NTK_ERR_BAD_PTR, "ntk_vgg_pack_weights: null pointer");
    NTK_REQUIRE(cin > 0 && cout > 0, NTK_ERR_BAD_SHAPE, "ntk_vgg_pack_weights: cin=%d cout=%d", cin, cout);
    const int Kp = ntk_vgg_packed_k(cin);
    const int total = cout * Kp;
    pack_weights_kernel<<<(total + 255) / 256, 256, 0, (hipStream_t)stream>>>(w_hwio, w_packed, cin, cout, Kp);
    NTK_CHECK_LAUNCH("ntk_vgg_pack_weights");
    return NTK_OK;
}

template <int BN, bool SMALLC>
static void launch_conv(const float* in, const float* wp, const float* bias, float* out, int npatch,
                        int H, int W, int cin, int cout, int Kp, int pool, hipStream_t st) {
    const long rtiles = ((long)npatch * 16 + BM - 1) / BM;
    dim3 grid((unsigned)(((rtiles + 7) / 8) * 8 * (cout / BN)));
    if constexpr (!SMALLC) {       // LDS-DMA staging, swizzled un-padded LDS image, 4 workgroups per CU
        if (pool) conv3x3_relu_dma_kernel<BN, true><<<grid, 256, 0, st>>>(in, wp, bias, out, npatch, H, W, cin, cout, Kp);
        else conv3x3_relu_dma_kernel<BN, false><<<grid, 256, 0, st>>>(in, wp, bias, out, npatch, H, W, cin, cout, Kp);
    } else {                       // tiny Cin: VGPR-staged kernel with the scalar-gather loader
        if (pool) conv3x3_relu_kernel<BN, SMALLC, true><<<grid, 256, 0, st>>>(in, wp, bias, out, npatch, H, W, cin, cout, Kp);
        else conv3x3_relu_kernel<BN, SMALLC, false><<<grid, 256, 0, st>>>(in, wp, bias, out, npatch, H, W, cin, cout, Kp);
    }
}

namespace {
// ---------------------------------------------------------------------------
// conv1_1 (Cin = 3 -> Cout = 64, no pool): dedicated persistent kernel.  The layer is bound by its 12.8 MB/frame
// of output stores, not by its 27-deep contraction, so the generic tile kernel's per-workgroup set-up (250 000
// workgroups for a 640-frame batch) is what it pays for.  Here a wave owns 32 consecutive pixels of one image row x
// all 64 channels: K = 27 (+1 zero) = 14 steps of v_mfma_f32_32x32x2_f32 x 2 column blocks, the A operand gathered
// straight from the 3-channel frame (L1 hits: every input value is used by 9 taps of 3 rows), the 28 x 64 weights
// resident in registers for the whole kernel, and waves stride over the row segments of the batch.
// Same arithmetic as the tile kernel: a k-ordered fp32 MFMA chain over k = tap * 3 + c.
// ---------------------------------------------------------------------------
template <bool OUTBF16>        // OUTBF16: conv1_1 of the bf16 trunk (config 5): the same fp32 arithmetic, rounded once on store
__global__ __launch_bounds__(256, 4) void conv_c3_rows_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              int nseg, int H, int W) {
    const int lane = threadIdx.x & 63, m = lane & 31, kh = lane >> 5;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    float b0[14], b1[14];
    int off[14], dyv[14], dxv[14];
#pragma unroll
    for (int s = 0; s < 14; ++s) {
        const int k = 2 * s + kh;                       // < 28; k = 27 is the zero pad column of the packed weights
        b0[s] = wp[m * 32 + k];
        b1[s] = wp[(32 + m) * 32 + k];
        const int tap = k / 3, c = k - tap * 3;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        dyv[s] = (k < 27) ? dy : 4;                     // 4: never inside the image
        dxv[s] = dx;
        off[s] = (dy * W + dx) * 3 + c;
    }
    const float bv0 = bias[m], bv1 = bias[32 + m];
    const int spr = W >> 5;                             // 32-pixel segments per row
    for (int seg = gw; seg < nseg; seg += nw) {
        const int xs = seg % spr;
        const int t = seg / spr;
        const int y = t % H;                            // t = f * H + y
        const int x = xs * 32 + m;
        const float* base = in + ((size_t)t * W + x) * 3;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        float av[14];
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            const int yy = y + dyv[s], xx = x + dxv[s];
            av[s] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? base[off[s]] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], b0[s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], b1[s], acc1, 0, 0, 0);
        }
        const size_t o0 = ((size_t)t * W + xs * 32) * 64;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mm = 4 * kh + (r & 3) + 8 * (r >> 2);
            const float v0 = fmaxf(acc0[r] + bv0, 0.f), v1 = fmaxf(acc1[r] + bv1, 0.f);
            if constexpr (OUTBF16) {
                __bf16* orow = reinterpret_cast<__bf16*>(out) + o0;
                orow[(size_t)mm * 64 + m] = (__bf16)v0;
                orow[(size_t)mm * 64 + 32 + m] = (__bf16)v1;
            } else {
                float* orow = out + o0;
                // non-temporal: 8.2 GB of activations per 640-frame pass that the next launch re-reads from HBM anyway (2.53 -> 2.46 ms)
                __builtin_nontemporal_store(v0, orow + (size_t)mm * 64 + m);
                __builtin_nontemporal_store(v1, orow + (size_t)mm * 64 + 32 + m);
            }
        }
    }
}

}  // namespace

extern "C" int ntk_vgg_conv3x3_relu_f32(const float* in, const float* w_packed, const float* bias,
                                        float* out, int frames, int H, int W, int cin, int cout,
                                        int fuse_pool, void* stream) {
    NTK_REQUIRE(in && w_packed && bias && out, NTK_ERR_BAD_PTR, "ntk_vgg_conv3x3_relu_f32: null pointer");
    NTK_REQUIRE(ntk_aligned16(in) && ntk_aligned16(w_packed) && ntk_aligned16(out), NTK_ERR_BAD_PTR,
                "ntk_vgg_conv3x3_relu_f32: pointers must be 16-byte aligned");
    NTK_REQUIRE(frames > 0 && H > 0 && W > 0 && (H % 4) == 0 && (W % 4) == 0 && H < 32768 && W < 32768,
                NTK_ERR_BAD_SHAPE, "ntk_vgg_conv3x3_relu_f32: frames=%d H=%d W=%d (H, W must be multiples of 4)",
                frames, H, W);
    NTK_REQUIRE(cout > 0 && (cout % 64) == 0, NTK_ERR_BAD_SHAPE,
                "ntk_vgg_conv3x3_relu_f32: cout=%d must be a multiple of 64", cout);
    const bool smallc = (cin % 32) != 0;
    NTK_REQUIRE(cin > 0 && (!smallc || 9 * cin <= BK), NTK_ERR_BAD_SHAPE,
                "ntk_vgg_conv3x3_relu_f32: cin=%d must be <= 3 or a multiple of 32", cin);
    const long npatch_l = (long)frames * (H / 4) * (W / 4);
    NTK_REQUIRE(npatch_l * 16 < 2147483647L - BM, NTK_ERR_BAD_SHAPE,
                "ntk_vgg_conv3x3_relu_f32: %ld pixels exceed the 2^31 row index range", npatch_l * 16);
    const int npatch = (int)npatch_l;
    const int Kp = ntk_vgg_packed_k(cin);
    hipStream_t st = (hipStream_t)stream;
    const bool bn128 = (cout % 128) == 0;
    if (cin == 3 && cout == 64 && !fuse_pool && (W % 32) == 0) {
        const long nseg = (long)frames * H * (W / 32);
        NTK_REQUIRE(nseg < 2147483647L, NTK_ERR_BAD_SHAPE, "ntk_vgg_conv3x3_relu_f32: %ld row segments", nseg);
        // 256 CUs x 4 workgroups, waves stride over segments (4 096 / 16 384 shorter-lived workgroups measured the same, alone and in the step)
        const int wgs = (int)((nseg + 3) / 4 < 1024 ? (nseg + 3) / 4 : 1024);
        conv_c3_rows_kernel<false><<<wgs, 256, 0, st>>>(in, w_packed, bias, out, (int)nseg, H, W);
    } else if (smallc) {
        if (bn128) launch_conv<128, true>(in, w_packed, bias, out, npatch, H, W, cin, cout, Kp, fuse_pool, st);
        else launch_conv<64, true>(in, w_packed, bias, out, npatch, H, W, cin, cout, Kp, fuse_pool, st);
    } else {
        if (bn128) launch_conv<128, false>(in, w_packed, bias, out, npatch, H, W, cin, cout, Kp, fuse_pool, st);
        else launch_conv<64, false>(in, w_packed, bias, out, npatch, H, W, cin, cout, Kp, fuse_pool, st);
    }
    NTK_CHECK_LAUNCH("ntk_vgg_conv3x3_relu_f32");
    return NTK_OK;
}

// conv1_1 of the bf16 trunk: fp32 frames in, bf16 activations out (same arithmetic, rounded once on store)
extern "C" int ntk_vgg_conv3x3_relu_f32_to_bf16(const float* in, const float* w_packed, const float* bias, void* out_bf16,
                                                int frames, int H, int W, int cin, int cout, void* stream) {
    NTK_REQUIRE(in && w_packed && bias && out_bf16, NTK_ERR_BAD_PTR, "ntk_vgg_conv3x3_relu_f32_to_bf16: null pointer");
    NTK_REQUIRE(frames > 0 && H > 0 && W > 0 && (H % 4) == 0 && (W % 4) == 0 && H < 32768 && W < 32768 && cin > 0 &&
                    9 * cin <= BK && cout == 64,
                NTK_ERR_BAD_SHAPE, "ntk_vgg_conv3x3_relu_f32_to_bf16: frames=%d H=%d W=%d cin=%d cout=%d (cin <= 3, cout == 64)",
                frames, H, W, cin, cout);
    const long npatch_l = (long)frames * (H / 4) * (W / 4);
    NTK_REQUIRE(npatch_l * 16 < 2147483647L - BM, NTK_ERR_BAD_SHAPE, "ntk_vgg_conv3x3_relu_f32_to_bf16: too many pixels");
    const int npatch = (int)npatch_l;
    if (cin == 3 && (W % 32) == 0) {            // the row kernel (4.06 -> store-bound at half the bytes: bit-identical results)
        const long nseg = (long)frames * H * (W / 32);
        NTK_REQUIRE(nseg < 2147483647L, NTK_ERR_BAD_SHAPE, "ntk_vgg_conv3x3_relu_f32_to_bf16: %ld row segments", nseg);
        const int wgs = (int)((nseg + 3) / 4 < 1024 ? (nseg + 3) / 4 : 1024);
        conv_c3_rows_kernel<true><<<wgs, 256, 0, (hipStream_t)stream>>>(in, w_packed, bias, reinterpret_cast<float*>(out_bf16), (int)nseg, H, W);
        NTK_CHECK_LAUNCH("ntk_vgg_conv3x3_relu_f32_to_bf16");
        return NTK_OK;
    }
    const long rtiles = (npatch_l * 16 + BM - 1) / BM;
    dim3 grid((unsigned)(((rtiles + 7) / 8) * 8));
    conv3x3_relu_kernel<64, true, false, true><<<grid, 256, 0, (hipStream_t)stream>>>(
        in, w_packed, bias, reinterpret_cast<float*>(out_bf16), npatch, H, W, cin, cout, ntk_vgg_packed_k(cin));
    NTK_CHECK_LAUNCH("ntk_vgg_conv3x3_relu_f32_to_bf16");
    return NTK_OK;
}

extern "C" int ntk_gemm_nt_f32(const float* A, int lda, const float* B, int ldb, const float* bias,
                               float* C, int ldc, int M, int N, int K, void* stream) {
    NTK_REQUIRE(A && B && C, NTK_ERR_BAD_PTR, "ntk_gemm_nt_f32: null pointer");
    NTK_REQUIRE(ntk_aligned16(A) && ntk_aligned16(B), NTK_ERR_BAD_PTR, "ntk_gemm_nt_f32: A, B must be 16-byte aligned");
    NTK_REQUIRE(M > 0 && N > 0 && K > 0 && (K % 4) == 0 && (lda % 4) == 0 && (ldb % 4) == 0 && lda >= K &&
                    ldb >= K && ldc >= N,
                NTK_ERR_BAD_SHAPE, "ntk_gemm_nt_f32: M=%d N=%d K=%d lda=%d ldb=%d ldc=%d (K, lda, ldb multiples of 4)",
                M, N, K, lda, ldb, ldc);
    hipStream_t st = (hipStream_t)stream;
    if (N > 64) {
        dim3 grid((M + BM - 1) / BM, (N + 127) / 128);
        gemm_nt_kernel<128><<<grid, 256, 0, st>>>(A, lda, B, ldb, bias, C, ldc, M, N, K);
    } else {
        dim3 grid((M + BM - 1) / BM, 1);
        gemm_nt_kernel<64><<<grid, 256, 0, st>>>(A, lda, B, ldb, bias, C, ldc, M, N, K);
    }
    NTK_CHECK_LAUNCH("ntk_gemm_nt_f32");
    return NTK_OK;
}

extern "C" size_t ntk_gemm_tn_workspace_bytes(int M, int N, int splits) {
    if (M <= 0 || N <= 0 || splits <= 0) return 0;
    return (size_t)M * N * splits * sizeof(float);
}

extern "C" int ntk_gemm_tn_f32(const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                               int M, int N, int K, int splits, int accumulate, float* workspace,
                               void* stream) {
    NTK_REQUIRE(A && B && C && workspace, NTK_ERR_BAD_PTR, "ntk_gemm_tn_f32: null pointer");
    NTK_REQUIRE(ntk_aligned16(A) && ntk_aligned16(B), NTK_ERR_BAD_PTR, "ntk_gemm_tn_f32: A, B must be 16-byte aligned");
    NTK_REQUIRE(M > 0 && N > 0 && K > 0 && splits > 0 && splits <= 65535 && (M % 4) == 0 && (N % 4) == 0 &&
                    (lda % 4) == 0 && (ldb % 4) == 0 && lda >= M && ldb >= N && ldc >= N,
                NTK_ERR_BAD_SHAPE,
                "ntk_gemm_tn_f32: M=%d N=%d K=%d splits=%d lda=%d ldb=%d ldc=%d (M, N, lda, ldb multiples of 4)",
                M, N, K, splits, lda, ldb, ldc);
    hipStream_t st = (hipStream_t)stream;
    int kchunk = (K + splits - 1) / splits;
    kchunk = ((kchunk + BK - 1) / BK) * BK;
    dim3 grid((M + BM - 1) / BM, (N + BM - 1) / BM, splits);
    gemm_tn_kernel<<<grid, 256, 0, st>>>(A, lda, B, ldb, workspace, M, N, K, kchunk);
    NTK_CHECK_LAUNCH("ntk_gemm_tn_f32");
    const int total = M * N;
    slab_reduce_kernel<<<(total + 255) / 256, 256, 0, st>>>(workspace, C, ldc, M, N, splits, accumulate);
    NTK_CHECK_LAUNCH("ntk_gemm_tn_f32(reduce)");
    return NTK_OK;
}
