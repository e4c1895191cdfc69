// VGG 3x3 convolution as fused Winograd F(2x2,3x3) on the fp32 MFMA pipe (gfx950).
//
// out = relu(conv3x3_same(in, w) + b) [+ 2x2/2 max pool] -- the same operator as mfma_f32.hip
// (vgg.py:155-161 via slim.conv2d), with 2.25x fewer multiplies: every 2x2 output tile is
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A          (Lavin & Gray), summed over input channels,
// i.e. 16 independent GEMMs  M_p[tile, cout] = sum_c V_p[tile, c] * U_p[c, cout],  p = 4i + j.
// Everything is fused in one kernel: the input transform B^T d B is computed from an NHWC patch staged in
// LDS, the 16 products run on v_mfma_f32_32x32x2_f32, the output transform A^T M A, bias, ReLU and the pool
// run in the epilogue.  Nothing but the NHWC activations and the pre-transformed weights touches HBM.
//
// Work decomposition
//   workgroup (256 threads = 4 waves): 32 tiles (= 32 MFMA rows; tile-block shapes below) x 64 output channels.
//   wave w owns the four planes p = 4w .. 4w+3 (i = w, j = 0..3): 4 planes x 2 column blocks x 16 = 128
//     accumulator registers.  A operand (V_p) from LDS, B operand (U_p) straight from global memory in a
//     lane-major packed layout (each wave reads only its own planes: no reuse inside the workgroup to stage for).
//   K loop: 8 input channels per iteration, software-pipelined (see the kernel): stage the patch (halo included),
//     transform, 32 MFMAs per wave.
//   Cross-wave part of A^T M A goes through LDS once per workgroup.
//   XCD-aware 1-D grid: an XCD keeps one column block, so its U slice stays in that XCD's L2.
// Results differ from the direct kernel by rounding only (different summation order, ~1e-6 relative per layer).
#include "common.h"

#ifndef WINO_UD
#define WINO_UD 3      // U prefetch distance in planes (1..3)
// Ablation switches for scripts/dev_wino_ablate.sh (results are WRONG with any of them set; never defined in the
// product build): bit 0 no U loads in the K loop, bit 1 no input transform, bit 2 no patch staging, bit 3 no
// workgroup barrier in the K loop, bit 4 no A-operand reads.
#ifndef WINO_ABL
#define WINO_ABL 0
#endif
#endif

namespace {

constexpr int WT = 256;            // threads
constexpr int KC = 16;             // channel granularity of the packed weights / entry-point checks
// tiles per workgroup = NSUB sub-blocks of TW x TH tiles, each sub-block a rectangle with its own halo patch:
//   8 x 4 x 1  all 32 MFMA rows, 180-pixel patch: tile grids that are multiples of 8 x 4 (conv1_2, conv2_x)
//   4 x 4 x 2  all 32 rows, two 100-pixel patches (consecutive sub-blocks, possibly of different frames): tile grids
//              that are multiples of 4 x 4 (conv3_x: 28 x 28 tiles)
//   2 x 2 x 8  all 32 rows, eight 36-pixel patches: any even tile grid (conv4_x: 14 x 14 tiles); 1.6x the staging
constexpr int BNW = 64;            // output channels per workgroup


struct WinoArgs {
    const float* in; const float* U; const float* bias; float* out;
    int frames, H, W, Cin, Cout;
    int nCB;            // Cout / 64
    int bxN, byN;       // workgroup blocks per frame: W/(2 TW), H/(2 TH)
    int NS;             // workgroups per column block = ceil(NQ / NSUB)
    int NQ;             // sub-blocks = frames * byN * bxN
};

// weights: HWIO [3][3][Cin][Cout] -> U_p = G g G^T, packed for the B operand so that ONE wave's fragments of one
// 8-channel K step are 8 KB contiguous (one scalar base + immediate offsets address all of them):
// index = ((((cb * n8 + c8) * 4 + w) * 4 + j) * 2 + nblk) * 256 + lane * 4 + q
//   with  p = 4 w + j,   c = 8 c8 + 4 (lane >> 5) + q,   cout = 64 cb + 32 nblk + (lane & 31)
__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ U, int Cin, int Cout) {
    const size_t total = (size_t)16 * Cin * Cout;
    const int n8 = Cin / 8;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int q = idx & 3, lane = (idx >> 2) & 63;
        size_t r = idx >> 8;
        const int nblk = r & 1; r >>= 1;
        const int p = r & 15; r >>= 4;                  // (w, j) = (p >> 2, p & 3): 4 w + j
        const int c8 = (int)(r % n8);
        const int cb = (int)(r / n8);
        const int c = 8 * c8 + 4 * (lane >> 5) + q;
        const int o = BNW * cb + 32 * nblk + (lane & 31);
        const int i = p >> 2, j = p & 3;
        // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
        const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
        float s = 0.f;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) s += G[i][ky] * G[j][kx] * w[((size_t)(ky * 3 + kx) * Cin + c) * Cout + o];
        U[idx] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Software-pipelined K loop.  8 input channels per iteration; wave w transforms exactly the planes it
// multiplies (i = w), into its half of a DOUBLE-buffered V, while it multiplies the previous chunk from the other
// half: the transform's LDS/VALU work sits in the shadow of the wave's own MFMAs instead of in a separate phase.
// The patch is double-buffered too: ONE workgroup barrier per iteration.  LDS images (bank rules:
// MI355X_MICROARCH.md): patch = 12 floats per pixel (8 channels + pad); V row = 8 floats with the two 16-byte
// chunks of tile row m swapped when (m >> 3) & 1 -- conflict-free ds_read_b128 for both the transform's window
// reads and the MFMA A fragments, conflict-free ds_write_b128 for the transform's stores.
// On this pipe every VALU / address instruction in the loop costs MFMA issue time (fp32 MFMA and VALU do not
// co-issue), so the loop is branch-free, copy-free (ping-pong operand registers) and addresses U through one
// scalar base per iteration + immediates.
// ---------------------------------------------------------------------------------------------------------
constexpr int KC2 = 8, RSR2 = 12, RSV2 = 8;
constexpr int WINO_MAX_CIN = 1024;
__device__ __attribute__((aligned(16))) float g_wino_zero[WINO_MAX_CIN] = {0.f};   // padding pixels read zeros here; pointers advance with K, so Cin <= WINO_MAX_CIN (checked by the entry point)

template <bool POOL, int TW, int TH, int NSUB>
__global__ __launch_bounds__(WT, 2) void conv3x3_wino_kernel(WinoArgs a) {
    constexpr int STILE = TW * TH, NTILE = NSUB * STILE, PW = 2 * TW + 2, PH = 2 * TH + 2, SPX = PW * PH, NPX = NSUB * SPX;
    // LDS strides of the patch image, in pixels: a patch row is padded from PW to PWS and a sub-block from PWS * PH to
    // SPXS so that the sixteen lanes of every ds_read_b128 lane group of the transform's window reads fall on sixteen
    // different 16-byte bank slots (scripts/dev_wino_lds_model.py: 256 -> 128 LDS cycles per K step; un-padded, every
    // group was 2-way conflicted: the 26-32 % SQ_LDS_BANK_CONFLICT share of profiles/r01_vgg_trunk_winograd_issue_lds_pmc.csv)
    constexpr int PWS = TW == 8 ? 24 : (TW == 4 ? 12 : PW), SPXS = TW == 2 ? 40 : PWS * PH, NPXS = NSUB * SPXS;
    constexpr int NST = (NPX * 2 + WT - 1) / WT;                      // float4 staging slots per thread (2 per pixel)
    constexpr int RAWSZ = NPXS * RSR2, VSZ = 16 * 32 * RSV2;           // (+16 floats of scratch per patch buffer)
    static_assert(NTILE <= 32 && NPX <= 288 && NPXS <= 320, "tile block");
    __shared__ __attribute__((aligned(16))) float s_raw[2 * (RAWSZ + 16)];  // 2 x 8.6-13.8 KB
    __shared__ __attribute__((aligned(16))) float s_V[2 * VSZ];             // 2 x 16 KB (32 KB: also the epilogue's Z)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    __shared__ int s_sbf[NSUB], s_sby[NSUB], s_sbx[NSUB];
    if (tid < NSUB) {
        const int sq = sp * NSUB + tid;
        if (sq < a.NQ) {
            const int bx = sq % a.bxN;
            const int t1 = sq / a.bxN;
            s_sbf[tid] = t1 / a.byN; s_sby[tid] = 2 * TH * (t1 % a.byN); s_sbx[tid] = 2 * TW * bx;
        } else {
            s_sbf[tid] = -1; s_sby[tid] = 0; s_sbx[tid] = 0;
        }
    }
    __syncthreads();

    // ---- patch staging: NPX pixels x 2 float4 slots
    // branch-free: padding pixels and slots beyond the patch read a zero page with stride 0 and, for the latter,
    // store into a scratch slot behind the patch image
    const float* src[NST];                             // running pointers: + 8 floats per K step
    int dst[NST];                                      // in float4 units (so the stores are ds_write_b128)
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int s = tid + k * WT;
        const int px = s >> 1, c4 = s & 1;
        dst[k] = (NPXS * RSR2) / 4 + (tid & 3);         // scratch (4 float4 behind each patch buffer)
        src[k] = g_wino_zero;
        if (px < NPX) {
            const int q = px / SPX, lp = px - q * SPX;
            const int pr = lp / PW, pc = lp - pr * PW;
            const int fq = s_sbf[q];
            const int y = s_sby[q] - 1 + pr, x = s_sbx[q] - 1 + pc;
            dst[k] = (q * SPXS + pr * PWS + pc) * (RSR2 / 4) + c4;
            if (fq >= 0 && y >= 0 && y < H && x >= 0 && x < W) src[k] = a.in + (((size_t)fq * H + y) * W + x) * Cin + c4 * 4;
        }
    }
    f32x4 stage[NST];
    auto load_patch = [&](int step) {                  // request the patch the pointers are at, then advance them
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            stage[k] = *reinterpret_cast<const f32x4*>(src[k]);
            src[k] += step;
        }
    };
    auto store_patch = [&](int buf) {
        f32x4* rb = reinterpret_cast<f32x4*>(s_raw) + buf * ((RAWSZ + 16) / 4);
#pragma unroll
        for (int k = 0; k < NST; ++k) rb[dst[k]] = stage[k];
    };

    // ---- transform role (plane row i = wave): lane -> tile lane >> 1, channel quad lane & 1
    const int pt_tile = lane >> 1, pt_c4 = lane & 1;
    const bool pt_on = pt_tile < NTILE;
    const int pt_q = pt_tile / STILE, pt_tl = pt_tile - pt_q * STILE;
    const int pt_tr = pt_tl / TW, pt_tc = pt_tl - pt_tr * TW;
    const int pt_rA = (wave == 0) ? 0 : (wave == 2 ? 2 : 1), pt_rB = (wave == 0) ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float pt_sg = (wave == 1) ? 1.f : -1.f;        // t = d[rA] + sg d[rB]: i0 r0-r2, i1 r1+r2, i2 r2-r1, i3 r1-r3
    // lanes of tile rows >= NTILE transform window 0 into their own (unused) V rows: no divergence
    const int pt_ra = pt_on ? (pt_q * SPXS + (2 * pt_tr + pt_rA) * PWS + 2 * pt_tc) * RSR2 + pt_c4 * 4 : pt_c4 * 4;
    const int pt_rb = pt_on ? (pt_q * SPXS + (2 * pt_tr + pt_rB) * PWS + 2 * pt_tc) * RSR2 + pt_c4 * 4 : pt_c4 * 4;
    const int pt_v = ((4 * wave) * 32 + pt_tile) * RSV2 + ((pt_c4 ^ ((pt_tile >> 3) & 1)) * 4);

    // ---- MFMA role
    const int mrow = lane & 31, kh = lane >> 5;
    const int va = ((4 * wave) * 32 + mrow) * RSV2 + ((kh ^ ((mrow >> 3) & 1)) * 4);
    const int n8 = Cin / KC2;

    f32x16 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][nb][r] = 0.f;

    // the transform in two halves: the window reads are requested two planes of MFMAs (~2000 cycles) before the
    // arithmetic and the stores, so the in-order wave never waits for LDS on its MFMA stream
    f32x4 tda[4], tdb[4];
    auto tr_load = [&](int rbuf) {
        const float* rp = s_raw + rbuf * (RAWSZ + 16);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            tda[c] = *reinterpret_cast<const f32x4*>(rp + pt_ra + c * RSR2);
            tdb[c] = *reinterpret_cast<const f32x4*>(rp + pt_rb + c * RSR2);
        }
    };
    auto tr_store = [&](int vbuf) {
        f32x4 t[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) t[c] = tda[c] + pt_sg * tdb[c];
        float* o = s_V + vbuf * VSZ + pt_v;
        *reinterpret_cast<f32x4*>(o) = t[0] - t[2];
        *reinterpret_cast<f32x4*>(o + 32 * RSV2) = t[1] + t[2];
        *reinterpret_cast<f32x4*>(o + 2 * 32 * RSV2) = t[2] - t[1];
        *reinterpret_cast<f32x4*>(o + 3 * 32 * RSV2) = t[1] - t[3];
    };
    auto transform = [&](int rbuf, int vbuf) { tr_load(rbuf); tr_store(vbuf); };

    // ---- prologue: patch 0 -> raw[0], transform -> V[0]; patch 1 in flight
    load_patch(n8 > 1 ? KC2 : 0);                      // patch 0; pointers now at patch min(1, n8 - 1)
    store_patch(0);
    __syncthreads();
    load_patch(n8 > 2 ? KC2 : 0);                      // patch min(1, n8 - 1); pointers at min(2, n8 - 1)
    transform(0, 0);
    // operand registers ping-pong by plane parity (4 planes per chunk: the parity carries over chunk boundaries),
    // so a prefetch lands in the registers its MFMAs read -- no copies.  The loop body is branch-free: the last
    // iteration harmlessly re-requests chunk 0's U, re-stages the last patch and re-transforms it.
    // U (B operand) ring of four register sets, one per plane index: plane j + UD is requested while plane j multiplies
    // (UD planes = UD x 512 MFMA cycles of cover for an L2 round trip; with UD = 1 the wave parks on every plane).
    constexpr int UD = WINO_UD;
    f32x4 Bq[4][2], Aq[2];                      // [plane][column block], [parity]
    const unsigned uw = __builtin_amdgcn_readfirstlane((unsigned)wave);
    // wave-uniform base of this wave's 8 KB of fragments for K step c8: ubase + c8 * 8192 floats
    const float* ubase = a.U + ((size_t)cb * n8 * 4 + uw) * 2048;
    const unsigned ulane = (unsigned)lane * 4u;
#pragma unroll
    for (int j = 0; j < UD; ++j) {
        Bq[j][0] = *reinterpret_cast<const f32x4*>(ubase + j * 512 + ulane);
        Bq[j][1] = *reinterpret_cast<const f32x4*>(ubase + j * 512 + 256 + ulane);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

    for (int c8 = 0; c8 < n8; ++c8) {
        const int cn = (c8 + 1 < n8) ? c8 + 1 : 0;                 // next chunk (wraps on the last iteration)
        if constexpr (!(WINO_ABL & 4)) store_patch((c8 + 1) & 1);
        if constexpr (!(WINO_ABL & 8)) __syncthreads();
        if constexpr (!(WINO_ABL & 4)) load_patch(c8 + 3 < n8 ? KC2 : 0);   // requests patch min(c8 + 2, n8 - 1)
        const float* uc = ubase + (size_t)c8 * 8192;
        const float* un = ubase + (size_t)cn * 8192;
        const float* vcur = s_V + (c8 & 1) * VSZ + va;
        if constexpr (!(WINO_ABL & 16)) Aq[0] = *reinterpret_cast<const f32x4*>(vcur);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int jn = j + UD;                                  // plane requested now
            const float* up = (jn < 4) ? uc + jn * 512 : un + (jn - 4) * 512;
            if constexpr (!(WINO_ABL & 1)) {
                Bq[jn & 3][0] = *reinterpret_cast<const f32x4*>(up + ulane);
                Bq[jn & 3][1] = *reinterpret_cast<const f32x4*>(up + 256 + ulane);
            }
            if constexpr (!(WINO_ABL & 16))
                if (j < 3) Aq[(j + 1) & 1] = *reinterpret_cast<const f32x4*>(vcur + (j + 1) * 32 * RSV2);
            if constexpr (!(WINO_ABL & 2)) {
                if (j == 0) tr_load((c8 + 1) & 1);          // next chunk's window reads ...
                if (j == 2) tr_store((c8 + 1) & 1);         // ... become its V two planes of MFMAs later
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Aq[j & 1][q], Bq[j][0][q], acc[j][0], 0, 0, 0);
                acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(Aq[j & 1][q], Bq[j][1][q], acc[j][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }

    // ---- epilogue: Y = A^T M A, A^T = [[1,1,1,0],[0,1,-1,-1]].  Row part (over j) inside the wave, column part (over
    //      i = wave) through LDS, one column block at a time: Z[i][b][r / 4][lane][r % 4] (32 KB = s_V), so both the
    //      stores and the loads are ds_*_b128 (wave w finishes the accumulator rows r = 4w .. 4w+3).
    f32x4* sZ4 = reinterpret_cast<f32x4*>(s_V);
    const int col = lane & 31;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        __syncthreads();
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            f32x4 z0, z1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * rq + e;
                z0[e] = acc[0][nb][r] + acc[1][nb][r] + acc[2][nb][r];
                z1[e] = acc[1][nb][r] - acc[2][nb][r] - acc[3][nb][r];
            }
            sZ4[((wave * 2 + 0) * 4 + rq) * 64 + lane] = z0;
            sZ4[((wave * 2 + 1) * 4 + rq) * 64 + lane] = z1;
        }
        __syncthreads();
        const int n = BNW * cb + 32 * nb + col;
        const float bv = a.bias[n];
        f32x4 zz[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b) zz[i][b] = sZ4[((i * 2 + b) * 4 + wave) * 64 + lane];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = 4 * wave + rr;
            const int m = 4 * kh + (r & 3) + 8 * (r >> 2);
            float y[2][2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                y[0][b] = zz[0][b][rr] + zz[1][b][rr] + zz[2][b][rr];
                y[1][b] = zz[1][b][rr] - zz[2][b][rr] - zz[3][b][rr];
            }
            const int mq = m / STILE, ml = m - mq * STILE;
            const int f = (m < NTILE) ? s_sbf[mq] : -1;
            if (f >= 0) {
                const int tr = ml / TW, tc = ml - tr * TW;
                const int oy = s_sby[mq] + 2 * tr, ox = s_sbx[mq] + 2 * tc;
                if constexpr (POOL) {
                    const float v = fmaxf(fmaxf(y[0][0], y[0][1]), fmaxf(y[1][0], y[1][1]));
                    a.out[(((size_t)f * (H >> 1) + (oy >> 1)) * (W >> 1) + (ox >> 1)) * Cout + n] = fmaxf(v + bv, 0.f);
                } else {
#pragma unroll
                    for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            a.out[(((size_t)f * H + oy + aa) * W + ox + b) * Cout + n] = fmaxf(y[aa][b] + bv, 0.f);
                }
            }
        }
    }
}

}  // namespace

extern "C" size_t ntk_vgg_wino_packed_floats(int cin, int cout) { return (size_t)16 * cin * cout; }

extern "C" int ntk_vgg_pack_weights_wino(const float* w_hwio, float* u_packed, int cin, int cout, void* stream) {
    NTK_REQUIRE(w_hwio && u_packed, NTK_ERR_BAD_PTR, "ntk_vgg_pack_weights_wino: null pointer");
    NTK_REQUIRE(cin >= KC && (cin % KC) == 0 && cout >= BNW && (cout % BNW) == 0, NTK_ERR_UNSUPPORTED,
                "ntk_vgg_pack_weights_wino: cin=%d (multiple of 16) cout=%d (multiple of 64)", cin, cout);
    wino_pack_kernel<<<2048, 256, 0, (hipStream_t)stream>>>(w_hwio, u_packed, cin, cout);
    NTK_CHECK_LAUNCH("ntk_vgg_pack_weights_wino");
    return NTK_OK;
}

extern "C" int ntk_vgg_conv3x3_relu_wino_f32(const float* in, const float* u_packed, const float* bias, float* out,
                                             int frames, int H, int W, int cin, int cout, int fuse_pool, void* stream) {
    NTK_REQUIRE(in && u_packed && bias && out, NTK_ERR_BAD_PTR, "ntk_vgg_conv3x3_relu_wino_f32: null pointer");
    NTK_REQUIRE(ntk_aligned16(in) && ntk_aligned16(u_packed) && ntk_aligned16(out), NTK_ERR_BAD_PTR,
                "ntk_vgg_conv3x3_relu_wino_f32: 16-byte alignment");
    NTK_REQUIRE(frames > 0 && H >= 4 && (H % 4) == 0 && W >= 4 && (W % 4) == 0, NTK_ERR_UNSUPPORTED,
                "ntk_vgg_conv3x3_relu_wino_f32: frames=%d H=%d W=%d (H, W multiples of 4)", frames, H, W);
    NTK_REQUIRE(cin >= KC && (cin % KC) == 0 && cin <= WINO_MAX_CIN && cout >= BNW && (cout % BNW) == 0, NTK_ERR_UNSUPPORTED,
                "ntk_vgg_conv3x3_relu_wino_f32: cin=%d (multiple of 16, at most %d: the zero page padding pixels read from) "
                "cout=%d (multiple of 64)", cin, WINO_MAX_CIN, cout);
    NTK_REQUIRE((unsigned long long)frames * H * W * cin < 0xffffffffull, NTK_ERR_UNSUPPORTED,
                "ntk_vgg_conv3x3_relu_wino_f32: input of %d x %d x %d x %d floats exceeds the 32-bit offset range", frames, H, W, cin);
    WinoArgs a;
    a.in = in; a.U = u_packed; a.bias = bias; a.out = out;
    a.frames = frames; a.H = H; a.W = W; a.Cin = cin; a.Cout = cout;
    // tile-block shape (see the table at the top): 0 = 8x4x1, 1 = 4x4x2, 3 = 2x2x8
    int shape = ((W % 16) == 0 && (H % 8) == 0) ? 0 : (((W % 8) == 0 && (H % 8) == 0) ? 1 : 3);
    a.nCB = cout / BNW;
    a.bxN = shape == 0 ? W / 16 : (shape == 1 ? W / 8 : W / 4);
    a.byN = shape == 3 ? H / 4 : H / 8;
    const long long NQ = (long long)frames * a.byN * a.bxN;
    const long long NS = shape == 1 ? (NQ + 1) / 2 : (shape == 3 ? (NQ + 7) / 8 : NQ);
    NTK_REQUIRE(NS < (1ll << 30) && (a.nCB <= 8 ? (8 % a.nCB) == 0 : (a.nCB % 8) == 0), NTK_ERR_UNSUPPORTED,
                "ntk_vgg_conv3x3_relu_wino_f32: cout/64=%d must divide or be a multiple of 8", a.nCB);
    a.NS = (int)NS;
    a.NQ = (int)NQ;
    long long slots;                                  // workgroup ids = slots * 8
    if (a.nCB >= 8) slots = NS * (a.nCB / 8);
    else { const int per = 8 / a.nCB; slots = (NS + per - 1) / per; }
    const long long grid = slots * 8;
    NTK_REQUIRE(grid < (1ll << 31), NTK_ERR_UNSUPPORTED, "ntk_vgg_conv3x3_relu_wino_f32: grid too large");
#define WINO_LAUNCH(POOL_, TW_, TH_, NSUB_) conv3x3_wino_kernel<POOL_, TW_, TH_, NSUB_><<<(unsigned)grid, WT, 0, (hipStream_t)stream>>>(a)
    if (shape == 0) {
        if (fuse_pool) WINO_LAUNCH(true, 8, 4, 1); else WINO_LAUNCH(false, 8, 4, 1);
    } else if (shape == 1) {
        if (fuse_pool) WINO_LAUNCH(true, 4, 4, 2); else WINO_LAUNCH(false, 4, 4, 2);
    } else {
        if (fuse_pool) WINO_LAUNCH(true, 2, 2, 8); else WINO_LAUNCH(false, 2, 2, 8);
    }
#undef WINO_LAUNCH
    NTK_CHECK_LAUNCH("ntk_vgg_conv3x3_relu_wino_f32");
    return NTK_OK;
}
