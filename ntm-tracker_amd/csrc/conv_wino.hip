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
//   workgroup (256 threads = 4 waves): 28 tiles (14 wide x 2 tall = 28 x 4 output pixels) x 64 output channels;
//     28 of the 32 MFMA rows are used (W/2 is a multiple of 14 for 224/112/56/28).
//   wave w owns the four planes p = 4w .. 4w+3 (i = w, j = 0..3): 4 planes x 2 column blocks x 16 = 128
//     accumulator registers.  A operand (V_p) from LDS, B operand (U_p) straight from global memory in a
//     lane-major packed layout (each wave reads only its own planes: no reuse inside the workgroup to stage for).
//   K loop: 16 input channels per iteration: stage the 6 x 30 pixel patch (halo included), transform, 64 MFMAs/wave.
//   Cross-wave part of A^T M A goes through LDS once per workgroup.
// Results differ from the direct kernel by rounding only (different summation order, ~1e-6 relative per layer).
#include "common.h"

namespace {

constexpr int WT = 256;            // threads
constexpr int KC = 16;             // input channels per K iteration
// tiles per workgroup: 14 wide x 2 tall (28 of the 32 MFMA rows; fits every VGG layer) or 8 x 4 (all 32 rows; layers
// whose tile grid is a multiple of 8 x 4: conv1_2, conv2_x).  Both patches are 180 pixels (6 x 30 or 10 x 18).
constexpr int NPX = 180;
constexpr int RS = 20;             // LDS row stride (floats) of a 16-channel row: conflict-free ds_read_b128
constexpr int BNW = 64;            // output channels per workgroup


struct WinoArgs {
    const float* in; const float* U; const float* bias; float* out;
    int frames, H, W, Cin, Cout;
    int nCB;            // Cout / 64
    int bxN, byN;       // workgroup blocks per frame: W/(2 TW), H/(2 TH)
    int NS;             // spatial blocks = frames * byN * bxN
};

// weights: HWIO [3][3][Cin][Cout] -> U_p = G g G^T, packed for the B operand:
// index = ((((cb * nChunk + cc) * 16 + p) * 2 + blk) * 2 + nblk) * 256 + lane * 4 + q
//   with  c = 16 cc + 8 blk + 4 (lane >> 5) + q,   cout = 64 cb + 32 nblk + (lane & 31)
__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ U, int Cin, int Cout) {
    const size_t total = (size_t)16 * Cin * Cout;
    const int nChunk = Cin / KC;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int q = idx & 3, lane = (idx >> 2) & 63;
        size_t r = idx >> 8;
        const int nblk = r & 1; r >>= 1;
        const int blk = r & 1; r >>= 1;
        const int p = r & 15; r >>= 4;
        const int cc = (int)(r % nChunk);
        const int cb = (int)(r / nChunk);
        const int c = KC * cc + 8 * blk + 4 * (lane >> 5) + q;
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

template <bool POOL, int TW, int TH>
__global__ __launch_bounds__(WT, 2) void conv3x3_wino_kernel(WinoArgs a) {
    constexpr int NTILE = TW * TH, PW = 2 * TW + 2, PH = 2 * TH + 2;
    static_assert(PW * PH == NPX && NTILE <= 32, "patch must be 180 pixels");
    __shared__ __attribute__((aligned(16))) float s_raw[NPX * RS];          // 14.4 KB  [pixel][16 ch]
    __shared__ __attribute__((aligned(16))) float s_V[16 * 32 * RS];        // 40 KB    [plane][tile][16 ch]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- XCD-aware decode: consecutive workgroup ids rotate over the 8 XCDs; an XCD keeps one column block
    //      (its U slice stays in that XCD's L2) and a contiguous run of spatial blocks
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    int cb, sp;
    if (a.nCB >= 8) {
        // column blocks 8k + xcd: slot enumerates (spatial, k)
        const int kN = a.nCB >> 3;
        cb = (slot % kN) * 8 + xcd;
        sp = slot / kN;
    } else {
        const int per = 8 / a.nCB;                 // XCDs per column block
        cb = xcd % a.nCB;
        sp = slot * per + xcd / a.nCB;
    }
    if (sp >= a.NS) return;
    const int bx = sp % a.bxN;
    const int t1 = sp / a.bxN;
    const int by = t1 % a.byN;
    const int f = t1 / a.byN;
    const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
    const int y0 = 2 * TH * by - 1, x0 = 2 * TW * bx - 1;        // top-left of the input patch

    // ---- patch staging: 180 pixels x 4 float4 = 720 float4 slots, 3 per thread (the last one partial)
    const float* src[3];
    int dst[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int s = tid + k * WT;
        const int px = s >> 2, c4 = s & 3;
        dst[k] = -1;
        src[k] = nullptr;
        if (px < NPX) {
            const int pr = px / PW, pc = px - pr * PW;
            const int y = y0 + pr, x = x0 + pc;
            dst[k] = px * RS + c4 * 4;
            if (y >= 0 && y < H && x >= 0 && x < W) src[k] = a.in + (((size_t)f * H + y) * W + x) * Cin + c4 * 4;
            else src[k] = nullptr;                     // zero padding (SAME)
        }
    }
    f32x4 stage[3];
    auto load_patch = [&](int cc) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (dst[k] >= 0 && src[k]) stage[k] = *reinterpret_cast<const f32x4*>(src[k] + cc * KC);
            else stage[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    // ---- input-transform role: item = (tile, channel quad, half of the plane rows)
    const int it_tile = tid >> 3, it_c4 = (tid >> 1) & 3, it_h = tid & 1;
    const bool it_on = it_tile < NTILE;
    const int it_tr = it_tile / TW, it_tc = it_tile - it_tr * TW;
    const float* rawp = s_raw + ((2 * it_tr + it_h) * PW + 2 * it_tc) * RS + it_c4 * 4;   // rows h .. h+2 of the 4x4 window
    float* vp = s_V + (8 * it_h * 32 + it_tile) * RS + it_c4 * 4;

    // ---- MFMA role
    const int mrow = lane & 31, kh = lane >> 5;
    const float* va = s_V + ((4 * wave) * 32 + mrow) * RS + 4 * kh;
    const int nChunk = Cin / KC;
    const float* ub = a.U + ((size_t)cb * nChunk * 16 + 4 * wave) * 1024 + lane * 4;      // + cc*16*1024 + j*1024 + blk*512 + nblk*256

    f32x16 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][nb][r] = 0.f;

    load_patch(0);
    for (int cc = 0; cc < nChunk; ++cc) {
        // (1) patch -> LDS
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (dst[k] >= 0) *reinterpret_cast<f32x4*>(s_raw + dst[k]) = stage[k];
        __syncthreads();                       // also: every wave is done reading V of the previous chunk
        if (cc + 1 < nChunk) load_patch(cc + 1);
        // B operand of plane j = 0 (prefetch before the transform)
        const float* uc = ub + (size_t)cc * 16 * 1024;
        f32x4 bq[2][2];                        // [blk][nblk]
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) bq[blk][nb] = *reinterpret_cast<const f32x4*>(uc + blk * 512 + nb * 256);

        // (2) input transform: V = B^T d B for plane rows i = 2h, 2h+1 (B^T rows: d0-d2, d1+d2, d2-d1, d1-d3)
        if (it_on) {
            f32x4 d[3][4];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) d[r][c] = *reinterpret_cast<const f32x4*>(rawp + (r * PW + c) * RS);
            // half 0 holds window rows 0,1,2 -> i=0: r0-r2, i=1: r1+r2 ; half 1 holds rows 1,2,3 -> i=2: r2-r1, i=3: r1-r3
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                f32x4 t[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (it_h == 0) t[c] = (ii == 0) ? (d[0][c] - d[2][c]) : (d[1][c] + d[2][c]);
                    else t[c] = (ii == 0) ? (d[1][c] - d[0][c]) : (d[0][c] - d[2][c]);
                }
                float* o = vp + (4 * ii) * 32 * RS;
                *reinterpret_cast<f32x4*>(o) = t[0] - t[2];
                *reinterpret_cast<f32x4*>(o + 32 * RS) = t[1] + t[2];
                *reinterpret_cast<f32x4*>(o + 2 * 32 * RS) = t[2] - t[1];
                *reinterpret_cast<f32x4*>(o + 3 * 32 * RS) = t[1] - t[3];
            }
        }
        __syncthreads();

        // (3) 4 planes x 2 column blocks x 8 k-steps of v_mfma_f32_32x32x2_f32
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 bn[2][2];
            if (j < 3) {
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        bn[blk][nb] = *reinterpret_cast<const f32x4*>(uc + (j + 1) * 1024 + blk * 512 + nb * 256);
            }
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(va + j * 32 * RS);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(va + j * 32 * RS + 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], bq[0][0][q], acc[j][0], 0, 0, 0);
                acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], bq[0][1][q], acc[j][1], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], bq[1][0][q], acc[j][0], 0, 0, 0);
                acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], bq[1][1][q], acc[j][1], 0, 0, 0);
            }
            if (j < 3) {
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) bq[blk][nb] = bn[blk][nb];
            }
        }
    }

    // ---- epilogue: Y = A^T M A, A^T = [[1,1,1,0],[0,1,-1,-1]].  Row part (over j) inside the wave, column part
    //      (over i = wave) through LDS: Z[i][b][r][lane], one column block at a time (32 KB, reuses s_V).
    float* sZ = s_V;
    const int col = lane & 31;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float z0 = acc[0][nb][r] + acc[1][nb][r] + acc[2][nb][r];
            const float z1 = acc[1][nb][r] - acc[2][nb][r] - acc[3][nb][r];
            sZ[((wave * 2 + 0) * 16 + r) * 64 + lane] = z0;
            sZ[((wave * 2 + 1) * 16 + r) * 64 + lane] = z1;
        }
        __syncthreads();
        const int n = BNW * cb + 32 * nb + col;
        const float bv = a.bias[n];
        // wave w finishes accumulator rows r = 4w .. 4w+3 for all four pixels of the tile
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = 4 * wave + rr;
            const int m = 4 * kh + (r & 3) + 8 * (r >> 2);         // tile index of this accumulator row
            float z[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int b = 0; b < 2; ++b) z[i][b] = sZ[((i * 2 + b) * 16 + r) * 64 + lane];
            float y[2][2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                y[0][b] = z[0][b] + z[1][b] + z[2][b];
                y[1][b] = z[1][b] - z[2][b] - z[3][b];
            }
            if (m < NTILE) {
                const int tr = m / TW, tc = m - tr * TW;
                const int oy = 2 * TH * by + 2 * tr, ox = 2 * TW * bx + 2 * tc;
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
    NTK_REQUIRE(frames > 0 && ((H >= 4 && (H % 4) == 0 && W >= 28 && (W % 28) == 0) || (H >= 8 && (H % 8) == 0 && (W % 16) == 0)),
                NTK_ERR_UNSUPPORTED,
                "ntk_vgg_conv3x3_relu_wino_f32: frames=%d H=%d W=%d (H %% 4 == 0 and W %% 28 == 0, or H %% 8 == 0 and W %% 16 == 0)",
                frames, H, W);
    NTK_REQUIRE(cin >= KC && (cin % KC) == 0 && cout >= BNW && (cout % BNW) == 0, NTK_ERR_UNSUPPORTED,
                "ntk_vgg_conv3x3_relu_wino_f32: cin=%d (multiple of 16) cout=%d (multiple of 64)", cin, cout);
    WinoArgs a;
    a.in = in; a.U = u_packed; a.bias = bias; a.out = out;
    a.frames = frames; a.H = H; a.W = W; a.Cin = cin; a.Cout = cout;
    const bool wide = (W % 16) == 0 && (H % 8) == 0;          // 8 x 4 tile blocks: every MFMA row used
    a.nCB = cout / BNW;
    a.bxN = wide ? W / 16 : W / 28;
    a.byN = wide ? H / 8 : H / 4;
    const long long NS = (long long)frames * a.byN * a.bxN;
    NTK_REQUIRE(NS < (1ll << 30) && (a.nCB <= 8 ? (8 % a.nCB) == 0 : (a.nCB % 8) == 0), NTK_ERR_UNSUPPORTED,
                "ntk_vgg_conv3x3_relu_wino_f32: cout/64=%d must divide or be a multiple of 8", a.nCB);
    a.NS = (int)NS;
    long long slots;                                  // workgroup ids = slots * 8
    if (a.nCB >= 8) slots = NS * (a.nCB / 8);
    else { const int per = 8 / a.nCB; slots = (NS + per - 1) / per; }
    const long long grid = slots * 8;
    NTK_REQUIRE(grid < (1ll << 31), NTK_ERR_UNSUPPORTED, "ntk_vgg_conv3x3_relu_wino_f32: grid too large");
    if (wide) {
        if (fuse_pool) conv3x3_wino_kernel<true, 8, 4><<<(unsigned)grid, WT, 0, (hipStream_t)stream>>>(a);
        else conv3x3_wino_kernel<false, 8, 4><<<(unsigned)grid, WT, 0, (hipStream_t)stream>>>(a);
    } else {
        if (fuse_pool) conv3x3_wino_kernel<true, 14, 2><<<(unsigned)grid, WT, 0, (hipStream_t)stream>>>(a);
        else conv3x3_wino_kernel<false, 14, 2><<<(unsigned)grid, WT, 0, (hipStream_t)stream>>>(a);
    }
    NTK_CHECK_LAUNCH("ntk_vgg_conv3x3_relu_wino_f32");
    return NTK_OK;
}
