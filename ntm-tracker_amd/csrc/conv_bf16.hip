// bf16 MFMA implicit-GEMM conv3x3 SAME + bias + ReLU (+ fused 2x2/2 max-pool) for BASELINE config 5
// ("bf16 MFMA conv + fp32 memory"): bf16 NHWC activations and weights, fp32 accumulation in
// v_mfma_f32_32x32x16_bf16, bf16 (or fp32 for the last layer) output rounded once on store.
// Same decomposition as the fp32 kernel (mfma_f32.hip): 128 x {128,64} tile per 256-thread workgroup,
// 4x4-pixel patch row order (pool in the epilogue), K order = 64-channel chunk outer / tap inner, LDS-DMA
// staging at 4 workgroups per CU, XCD-aware tile order.  A K-tile is 64 bf16 = 128 B per row: the LDS image
// and the ds_read_b128 fragment addresses are byte-identical to the fp32 kernel's, one MFMA now consumes
// what four fp32 MFMAs did.
#include "common.h"
#include "conv_common.h"

// Ablation switches for scripts/dev_wino_variant.sh (timing only; never defined in the product build): bit 0 no global -> LDS
// staging after the first K-tile, bit 1 no MFMAs, bit 2 no barriers in the K loop, bit 3 no output stores.
#ifndef BF16_ABL
#define BF16_ABL 0
#endif

namespace {

constexpr int BM = 128;
constexpr int BKB = 64;            // bf16 elements per K-tile (128 bytes)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 as_bf16x8(const f32x4& v) {
    union { f32x4 f; bf16x8 b; } u;
    u.f = v;
    return u.b;
}

// LDS-DMA staging (as the fp32 direct kernel): rows are 128 B (64 bf16), chunk c of row r in slot
// c ^ ((r >> 1) & 7), one DMA wave-instruction fills 8 rows; no VGPR staging -> 4 workgroups per CU.
__device__ __attribute__((aligned(128))) float g_zero_page_bf16[32];

template <int BN>
__device__ __forceinline__ void mma_ktile_bf16_swz(const float* __restrict__ As, const float* __restrict__ Bs,
                                                   f32x16 (&acc)[2][BN / 64], int wm, int wn, int lane) {
    constexpr int TN = BN / 64;
    const int i = lane & 31, kh = lane >> 5, f = (i >> 1) & 7;
    const float* ap = As + (wm * 64 + i) * 32;
    const float* bp = Bs + (wn * (BN / 2) + i) * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int slot = ((2 * q + kh) ^ f) * 4;
        f32x4 a[2], b[TN];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) a[tm] = *reinterpret_cast<const f32x4*>(ap + tm * 32 * 32 + slot);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) b[tn] = *reinterpret_cast<const f32x4*>(bp + tn * 32 * 32 + slot);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a[tm]), as_bf16x8(b[tn]), acc[tm][tn], 0, 0, 0);
    }
}

__device__ __forceinline__ void lds_dma16b(const void* src, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int BN, bool POOL, bool OUTF32>
__global__ __launch_bounds__(256, 4) void conv3x3_relu_bf16_dma_kernel(
    const __bf16* __restrict__ in, const __bf16* __restrict__ wp, const float* __restrict__ bias,
    void* __restrict__ outv, int npatch, int H, int W, int Cin, int Cout, int Kp) {
    constexpr int TN = BN / 64, NBI = BN / 32;
    __shared__ __attribute__((aligned(1024))) float lds[(BM + BN) * 32];
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
    const int lr = lane >> 3, slot = lane & 7;
    int rpix[4], ry[4], rx[4], achunk[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int r = (wave * 4 + jj) * 8 + lr;
        rpix[jj] = s_pix[r];
        const int yx = s_yx[r];
        ry[jj] = yx >> 16; rx[jj] = yx & 0xffff;
        achunk[jj] = (slot ^ ((r >> 1) & 7)) * 8;          // bf16 elements
    }
    const __bf16* bsrc[NBI];
#pragma unroll
    for (int jj = 0; jj < NBI; ++jj) {
        const int r = (wave * NBI + jj) * 8 + lr;
        bsrc[jj] = wp + (size_t)(n0 + r) * Kp + (slot ^ ((r >> 1) & 7)) * 8;
    }
    float* As = lds;
    float* Bs = lds + BM * 32;
    f32x16 acc[2][TN];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    const int nk = Kp / BKB;
    for (int kt = 0; kt < nk; ++kt) {
        const int chunk = kt / 9, tap = kt - chunk * 9;
        const int c0 = chunk * BKB;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        if ((BF16_ABL & 4) == 0 && kt > 0) __syncthreads();
        if ((BF16_ABL & 1) == 0 || kt == 0) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int yy = ry[jj] + dy, xx = rx[jj] + dx;
            const bool ok = rpix[jj] >= 0 && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            const void* src = ok ? (const void*)(in + (size_t)(rpix[jj] + dy * W + dx) * Cin + c0 + achunk[jj])
                                 : (const void*)(g_zero_page_bf16 + (lane & 7) * 4);
            lds_dma16b(src, As + (wave * 4 + jj) * 8 * 32);
        }
#pragma unroll
        for (int jj = 0; jj < NBI; ++jj) lds_dma16b(bsrc[jj] + kt * BKB, Bs + (wave * NBI + jj) * 8 * 32);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((BF16_ABL & 4) == 0) __syncthreads();
        if ((BF16_ABL & 2) == 0) mma_ktile_bf16_swz<BN>(As, Bs, acc, wm, wn, lane);
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
                    if (pix >= 0 && ((BF16_ABL & 8) == 0 || acc[tm][tn][r] == 12345.f)) {
                        const float v = fmaxf(acc[tm][tn][r] + bv, 0.f);
                        if constexpr (OUTF32) reinterpret_cast<float*>(outv)[(size_t)pix * Cout + n] = v;
                        else reinterpret_cast<__bf16*>(outv)[(size_t)pix * Cout + n] = (__bf16)v;
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int pp = s_ppix[mbase + 8 * g];
                    const float v = fmaxf(fmaxf(acc[tm][tn][4 * g], acc[tm][tn][4 * g + 1]),
                                          fmaxf(acc[tm][tn][4 * g + 2], acc[tm][tn][4 * g + 3]));
                    if (pp >= 0 && ((BF16_ABL & 8) == 0 || v == 12345.f)) {
                        const float o = fmaxf(v + bv, 0.f);
                        if constexpr (OUTF32) reinterpret_cast<float*>(outv)[(size_t)pp * Cout + n] = o;
                        else reinterpret_cast<__bf16*>(outv)[(size_t)pp * Cout + n] = (__bf16)o;
                    }
                }
            }
        }
    }
}

// HWIO fp32 [3,3,Cin,Cout] -> bf16 [Cout][Kp], k = (c/64)*576 + tap*64 + c%64 (Cin % 64 == 0)
__global__ void pack_weights_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, int Cin, int Cout, int Kp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cout * Kp) return;
    const int n = idx / Kp, k = idx - n * Kp;
    const int chunk = k / (9 * BKB), r = k - chunk * 9 * BKB;
    const int tap = r / BKB, c = chunk * BKB + (r - tap * BKB);
    wp[idx] = (__bf16)w[(size_t)(tap * Cin + c) * Cout + n];
}

template <int BN>
void launch_bf16(const __bf16* in, const __bf16* wp, const float* bias, void* out, int npatch, int H, int W, int cin,
                 int cout, int pool, int out_f32, hipStream_t st) {
    const long rtiles = ((long)npatch * 16 + BM - 1) / BM;
    dim3 grid((unsigned)(((rtiles + 7) / 8) * 8 * (cout / BN)));
    const int Kp = 9 * cin;
    if (pool) {
        if (out_f32) conv3x3_relu_bf16_dma_kernel<BN, true, true><<<grid, 256, 0, st>>>(in, wp, bias, out, npatch, H, W, cin, cout, Kp);
        else conv3x3_relu_bf16_dma_kernel<BN, true, false><<<grid, 256, 0, st>>>(in, wp, bias, out, npatch, H, W, cin, cout, Kp);
    } else {
        if (out_f32) conv3x3_relu_bf16_dma_kernel<BN, false, true><<<grid, 256, 0, st>>>(in, wp, bias, out, npatch, H, W, cin, cout, Kp);
        else conv3x3_relu_bf16_dma_kernel<BN, false, false><<<grid, 256, 0, st>>>(in, wp, bias, out, npatch, H, W, cin, cout, Kp);
    }
}

}  // namespace

extern "C" int ntk_vgg_pack_weights_bf16(const float* w_hwio, void* w_packed_bf16, int cin, int cout, void* stream) {
    NTK_REQUIRE(w_hwio && w_packed_bf16, NTK_ERR_BAD_PTR, "ntk_vgg_pack_weights_bf16: null pointer");
    NTK_REQUIRE(cin > 0 && (cin % BKB) == 0 && cout > 0, NTK_ERR_BAD_SHAPE, "ntk_vgg_pack_weights_bf16: cin=%d (multiple of 64) cout=%d", cin, cout);
    const int Kp = 9 * cin, total = cout * Kp;
    pack_weights_bf16_kernel<<<(total + 255) / 256, 256, 0, (hipStream_t)stream>>>(w_hwio, reinterpret_cast<__bf16*>(w_packed_bf16), cin, cout, Kp);
    NTK_CHECK_LAUNCH("ntk_vgg_pack_weights_bf16");
    return NTK_OK;
}

extern "C" int ntk_vgg_conv3x3_relu_bf16(const void* in_bf16, const void* w_packed_bf16, const float* bias, void* out,
                                         int frames, int H, int W, int cin, int cout, int fuse_pool, int out_f32,
                                         void* stream) {
    NTK_REQUIRE(in_bf16 && w_packed_bf16 && bias && out, NTK_ERR_BAD_PTR, "ntk_vgg_conv3x3_relu_bf16: null pointer");
    NTK_REQUIRE(ntk_aligned16(in_bf16) && ntk_aligned16(w_packed_bf16) && ntk_aligned16(out), NTK_ERR_BAD_PTR,
                "ntk_vgg_conv3x3_relu_bf16: pointers must be 16-byte aligned");
    NTK_REQUIRE(frames > 0 && H > 0 && W > 0 && (H % 4) == 0 && (W % 4) == 0 && H < 32768 && W < 32768, NTK_ERR_BAD_SHAPE,
                "ntk_vgg_conv3x3_relu_bf16: frames=%d H=%d W=%d (H, W multiples of 4)", frames, H, W);
    NTK_REQUIRE(cin > 0 && (cin % BKB) == 0 && cout > 0 && (cout % 64) == 0, NTK_ERR_BAD_SHAPE,
                "ntk_vgg_conv3x3_relu_bf16: cin=%d must be a multiple of 64, cout=%d a multiple of 64", cin, cout);
    const long npatch_l = (long)frames * (H / 4) * (W / 4);
    NTK_REQUIRE(npatch_l * 16 < 2147483647L - BM, NTK_ERR_BAD_SHAPE, "ntk_vgg_conv3x3_relu_bf16: too many pixels");
    const __bf16* in = reinterpret_cast<const __bf16*>(in_bf16);
    const __bf16* wp = reinterpret_cast<const __bf16*>(w_packed_bf16);
    if ((cout % 128) == 0) launch_bf16<128>(in, wp, bias, out, (int)npatch_l, H, W, cin, cout, fuse_pool, out_f32, (hipStream_t)stream);
    else launch_bf16<64>(in, wp, bias, out, (int)npatch_l, H, W, cin, cout, fuse_pool, out_f32, (hipStream_t)stream);
    NTK_CHECK_LAUNCH("ntk_vgg_conv3x3_relu_bf16");
    return NTK_OK;
}
