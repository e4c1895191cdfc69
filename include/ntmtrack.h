/*
 * ntmtrack.h -- C ABI of libntmtrack_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the per-frame tracking hot path of
 * JeffOwOSun/ntm-tracker.  The reference is pure Python/TensorFlow-1 and has
 * no FFI of its own; each entry point below names the reference op group
 * (file:line in the reference tree) it replaces, and INTEGRATION.md shows the
 * ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is caller-owned DEVICE memory (fp32 unless noted);
 *     nothing is allocated, freed or retained by the library;
 *   - `stream` is a hipStream_t passed as void*; all work is asynchronous and
 *     ordered on that stream; no host synchronisation happens inside;
 *   - return value: NTK_OK (0) or a negative NTK_ERR_* code;
 *     ntk_last_error() returns a thread-local description of the last failure;
 *   - shapes are validated on the host before any launch: a bad shape is
 *     refused (NTK_ERR_BAD_SHAPE), never launched.
 */
#ifndef NTMTRACK_H_
#define NTMTRACK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NTK_OK               0
#define NTK_ERR_BAD_SHAPE   -1
#define NTK_ERR_BAD_PTR     -2
#define NTK_ERR_UNSUPPORTED -3
#define NTK_ERR_HIP         -4

/* library version (major*10000 + minor*100 + patch) and last error text */
int ntk_version(void);
const char* ntk_last_error(void);

/* ------------------------------------------------------------------------
 * VGG-16 feature extractor, conv1_1 .. conv4_3
 * replaces: frozen GraphDef import, direct_offset_output.py:417-422
 *           (layer spec vgg.py:155-161, arg scope vgg.py:49-63)
 * --------------------------------------------------------------------- */

/* Repack TF HWIO weights [3,3,Cin,Cout] into the kernel's [Cout][Kp] layout,
 * k = (ky*3+kx)*Cin + c, Kp = ntk_vgg_packed_k(Cin) (zero padded). */
int ntk_vgg_packed_k(int cin);
int ntk_vgg_pack_weights(const float* w_hwio, float* w_packed, int cin, int cout, void* stream);

/* out = relu(conv3x3_same(in, w) + bias), NHWC fp32; optional fused 2x2/2
 * max-pool (vgg.py:156,158,160).  in [frames,H,W,Cin]; out [frames,H,W,Cout]
 * or [frames,H/2,W/2,Cout] when fuse_pool.  H, W multiples of 4; Cin == 3 or
 * a multiple of 32; Cout a multiple of 64.  MFMA v_mfma_f32_32x32x2_f32. */
int ntk_vgg_conv3x3_relu_f32(const float* in, const float* w_packed, const float* bias,
                             float* out, int frames, int H, int W, int cin, int cout,
                             int fuse_pool, void* stream);

/* ------------------------------------------------------------------------
 * plain fp32 GEMMs used around the NTM recurrence (hoisted LSTM input
 * projection and the BPTT weight-gradient contractions)
 * replaces: tf.matmul inside BasicLSTMCell / _linear, ntm_cell.py:103-105,
 *           :124-126, :220 (and their tf.gradients)
 * --------------------------------------------------------------------- */

/* C[M,N] = A[M,K] * B[N,K]^T (+ bias[N]);  lda, ldb, K multiples of 4. */
int ntk_gemm_nt_f32(const float* A, int lda, const float* B, int ldb, const float* bias,
                    float* C, int ldc, int M, int N, int K, void* stream);

/* C[M,N] (+)= sum_k A[k,M] * B[k,N]  (both operands k-major), split over
 * `splits` K-ranges through `workspace` (splits*M*N floats) and reduced in a
 * fixed order (bitwise reproducible).  lda, ldb, M, N multiples of 4.
 * accumulate != 0 adds into C. */
size_t ntk_gemm_tn_workspace_bytes(int M, int N, int splits);
int ntk_gemm_tn_f32(const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                    int M, int N, int K, int splits, int accumulate, float* workspace,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NTMTRACK_H_ */
