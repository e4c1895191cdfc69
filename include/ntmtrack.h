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
 *     refused (NTK_ERR_BAD_SHAPE / NTK_ERR_UNSUPPORTED), never launched;
 *   - stateless and re-entrant: no mutable process-global switches; the only
 *     global state is a per-device cache of one-time kernel attributes.
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
 * Kp = ntk_vgg_packed_k(Cin) (zero padded); for Cin % 32 == 0 the K order is
 * 32-channel chunk outer / tap inner: k = (c/32)*288 + (ky*3+kx)*32 + c%32;
 * for Cin = 3: k = (ky*3+kx)*Cin + c. */
int ntk_vgg_packed_k(int cin);
int ntk_vgg_pack_weights(const float* w_hwio, float* w_packed, int cin, int cout, void* stream);

/* out = relu(conv3x3_same(in, w) + bias), NHWC fp32; optional fused 2x2/2
 * max-pool (vgg.py:156,158,160).  in [frames,H,W,Cin]; out [frames,H,W,Cout]
 * or [frames,H/2,W/2,Cout] when fuse_pool.  H, W multiples of 4; Cin == 3 or
 * a multiple of 32; Cout a multiple of 64.  MFMA v_mfma_f32_32x32x2_f32. */
int ntk_vgg_conv3x3_relu_f32(const float* in, const float* w_packed, const float* bias,
                             float* out, int frames, int H, int W, int cin, int cout,
                             int fuse_pool, void* stream);

/* bf16 trunk (BASELINE config 5: "bf16 MFMA conv + fp32 memory"): bf16 NHWC activations and weights, fp32
 * accumulation (v_mfma_f32_32x32x16_bf16), output rounded once to bf16 (or kept fp32 when out_f32, for the
 * last layer feeding the memory cell).  Cin a multiple of 64.  Packed weights: bf16 [Cout][9*Cin],
 * k = (c/64)*576 + (ky*3+kx)*64 + c%64.  conv1_1 (Cin = 3) runs the fp32 kernel on the fp32 frames and
 * stores bf16 (ntk_vgg_conv3x3_relu_f32_to_bf16, fp32 packed weights from ntk_vgg_pack_weights). */
int ntk_vgg_pack_weights_bf16(const float* w_hwio, void* w_packed_bf16, int cin, int cout, void* stream);
int ntk_vgg_conv3x3_relu_bf16(const void* in_bf16, const void* w_packed_bf16, const float* bias, void* out,
                              int frames, int H, int W, int cin, int cout, int fuse_pool, int out_f32,
                              void* stream);
/* The same operator in PATCH form (csrc/conv_bf16p.hip, round 4): the input of a block of 512 output pixels is staged once per
 * channel chunk as a patch with its halo and serves all nine taps; weights go through LDS once per workgroup; everything moves by
 * LDS-DMA.  Takes frames whose sides are multiples of 4 (of 8 with the fused pool), cin a multiple of 32, cout of 64
 * (ntk_vgg_bf16p_supported says so for a shape; ntk_vgg_conv3x3_relu_bf16 runs the others).  The weights are packed per layer
 * AND frame shape (ntk_vgg_pack_weights_bf16p: the chunk size depends on H, W): 9 * cin * cout bf16 elements.  Same operand
 * rounding, accumulation type and output rounding as ntk_vgg_conv3x3_relu_bf16; the summation order differs. */
size_t ntk_vgg_bf16p_packed_elems(int cin, int cout);
int ntk_vgg_bf16p_supported(int H, int W, int cin, int cout, int fuse_pool);
int ntk_vgg_pack_weights_bf16p(const float* w_hwio, void* w_packed_bf16, int cin, int cout, int H, int W, void* stream);
int ntk_vgg_conv3x3_relu_bf16p(const void* in_bf16, const void* w_packed_bf16p, const float* bias, void* out,
                               int frames, int H, int W, int cin, int cout, int fuse_pool, int out_f32, void* stream);
int ntk_vgg_conv3x3_relu_f32_to_bf16(const float* in, const float* w_packed, const float* bias, void* out_bf16,
                                     int frames, int H, int W, int cin, int cout, void* stream);

/* The SPLIT form of the fp32 trunk (csrc/conv_bf16p.hip, template flag X3; round 4): the same fp32 operator (vgg.py:155-161) on the
 * bf16 matrix pipe.  Every fp32 value v travels as hi = bf16(v), lo = bf16(v - hi); a product x w is accumulated in fp32 as
 * xh wh + xh wl + xl wh (relative error of the split 2^-17: per layer 4e-6 .. 5e-6 of the activation scale, what the F(4x4) Winograd
 * kernel has).  A split map is [frames][H][W][C / 16][2][16] bf16 (hi x16 | lo x16 per group of 16 channels: the bytes of the fp32 map).
 * Shapes: H, W multiples of 8, or W = 28 with H >= 20 (not a multiple of 8) and no pool; cin a multiple of 16, cout of 64.  Packed weights:
 * 18 * cin * cout bf16 elements, packed per layer and frame shape.  out_f32 = 1 writes fp32 NHWC (where the trunk leaves the split
 * form); in_f32 = 1 reads an fp32 NHWC map and splits it while staging (where the trunk enters it: cin <= 64 and cout = 64 only). */
size_t ntk_vgg_split3_packed_elems(int cin, int cout);
int ntk_vgg_split3_supported(int H, int W, int cin, int cout, int fuse_pool);
int ntk_vgg_pack_weights_split3(const float* w_hwio, void* w_packed, int cin, int cout, int H, int W, void* stream);
int ntk_vgg_conv3x3_relu_split3(const void* in_split, const void* w_packed, const float* bias, void* out,
                                int frames, int H, int W, int cin, int cout, int fuse_pool, int in_f32, int out_f32, void* stream);

/* The same operator by fused Winograd F(2x2,3x3) on the fp32 MFMA pipe (2.25x fewer multiplies; results equal to
 * ntk_vgg_conv3x3_relu_f32 up to rounding, ~1e-6 relative per layer).  Weights: U = G g G^T for the 16 transform
 * planes, packed lane-major for the MFMA B operand (ntk_vgg_wino_packed_floats(cin, cout) = 16*cin*cout floats).
 * cin a multiple of 16 and at most 1024, cout a multiple of 64 (64, 128, 256 or a multiple of 512); H and W multiples
 * of 4. */
size_t ntk_vgg_wino_packed_floats(int cin, int cout);
int ntk_vgg_pack_weights_wino(const float* w_hwio, float* u_packed, int cin, int cout, void* stream);
int ntk_vgg_conv3x3_relu_wino_f32(const float* in, const float* u_packed, const float* bias, float* out,
                                  int frames, int H, int W, int cin, int cout, int fuse_pool, void* stream);

/* The same operator by fused Winograd F(4x4,3x3) (csrc/conv_wino43.hip): 36 transform planes, 4x fewer multiplies
 * than the direct form; fp32 rounding error ~16x that of F(2x2,3x3) (4e-6 .. 9e-6 of the activation scale per layer),
 * inside the 1e-4 bound of the path.  cin multiple of 16, cout multiple of 64 (cout/64 dividing or a multiple of 8),
 * H and W multiples of 4.  ntk_vgg_wino43_packed_floats(cin, cout) = 36*cin*cout floats. */
size_t ntk_vgg_wino43_packed_floats(int cin, int cout);
int ntk_vgg_pack_weights_wino43(const float* w_hwio, float* u_packed, int cin, int cout, void* stream);
int ntk_vgg_conv3x3_relu_wino43_f32(const float* in, const float* u_packed, const float* bias, float* out,
                                    int frames, int H, int W, int cin, int cout, int fuse_pool, void* stream);
/* The same layer computed only inside the window [y0, y1) x [x0, x1) of its un-pooled output (multiples of 4 inside the
 * frame): the tiles of the window are written exactly as the whole-frame call writes them, nothing else is touched.  For
 * the last layer of a trunk whose consumer reads fixed positions only (direct_offset_output.py:392-399 gathers 64 points
 * of conv4_3, receptive_field_sizes.py:135-143: rows / columns 6, 8, ..., 20 of 28). */
int ntk_vgg_conv3x3_relu_wino43_window_f32(const float* in, const float* u_packed, const float* bias, float* out,
                                           int frames, int H, int W, int cin, int cout, int fuse_pool,
                                           int y0, int x0, int y1, int x1, void* stream);
/* The general form: window + kernel form.  waves = 8 (what the two entries above launch): eight waves per workgroup, two per
 * SIMD -- one runs the patch staging and the input transform beside a third of the MFMAs, the other two thirds of the MFMAs and
 * nothing else; waves = 4: round 2's one-wave-per-SIMD kernel.  Same bits either way (tests/test_vgg_gpu.py). */
int ntk_vgg_conv3x3_relu_wino43_form_f32(const float* in, const float* u_packed, const float* bias, float* out,
                                         int frames, int H, int W, int cin, int cout, int fuse_pool,
                                         int y0, int x0, int y1, int x1, int waves, void* stream);
/* The eight-wave kernel with CHANNEL-BLOCKED activation maps on either side, [frames][H][C / 8][W][8] (channel blocks of 8
 * interleaved per image row) -- the layout the layers of a trunk hand to each other: the eight channels of a K step are one
 * contiguous 32-byte piece per pixel and a patch row is one contiguous run, so the patch staging reads whole cache lines (NHWC:
 * 32-byte pieces 4 * cin bytes apart).  in_blocked / out_blocked: which side is blocked (0 = NHWC).  Whole frames, any shape the
 * eight-wave kernel takes.  Same arithmetic and bits as ntk_vgg_conv3x3_relu_wino43_f32.
 * (vgg.py:155-161: the layers themselves; the layout between them is this library's own business.) */
int ntk_vgg_conv3x3_relu_wino43_layout_f32(const float* in, const float* u_packed, const float* bias, float* out,
                                           int frames, int H, int W, int cin, int cout, int fuse_pool,
                                           int in_blocked, int out_blocked, void* stream);

/* slim.max_pool2d [2,2] stride 2 on NHWC fp32 (vgg.py:155-161) as its own launch (ntk_maxpool2x2).
 * The trunk fuses the pool into the epilogue of conv1_2 / conv2_2 / conv3_3 (fuse_pool); this entry point is the
 * un-fused form with identical results.  H, W even; C a multiple of 4. */
int ntk_maxpool2x2(const float* in, float* out, int frames, int H, int W, int C, void* stream);

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

/* out[cols][ldo] = in[rows][ldi]^T, zero padded to ldo (>= rows) */
int ntk_transpose_pad(const float* in, int ldi, float* out, int ldo, int rows, int cols, void* stream);

/* ------------------------------------------------------------------------
 * NTM cell sequence kernels
 * replaces: NTMCell.__call__ (ntm_cell.py:53-253) unrolled by LoopNTMTracker's
 *           tf.while_loop (ntm_tracker_new.py:13-64), ops.py:135-158 (cosine
 *           similarity, quirk Q1) and ops.py:180-242 (circular shift, quirk Q2)
 *
 * Packed parameter layouts (see ntm-tracker_amd/csrc/ntm_common.h):
 *   WxT [4*hid][ldx]  Wr [ldz][4*hid] (row K = LSTM bias)  Wa [ldh][PP] (row hid = biases)
 * with gate columns interleaved per unit (n' = unit*4 + gate, gate order i,j,f,o).
 * --------------------------------------------------------------------- */

/* control width P, padded widths and leading dimensions for a configuration */
int ntk_ntm_padded_dims(int N, int Md, int R, int Wh, int hid, int shift_range, int O,
                        int* P, int* PP, int* K, int* ldz, int* ldh);

/* S steps of the cell for B sequences, one persistent workgroup per sequence.
 * xproj [B,S,4*hid] = X * WxT^T (no bias).  State in: M0 [B,N,Md], w0 [B,H,N],
 * read0 [B,R,Md], cs0 [B,2*hid] (c then h).  Out: logits [B,S,O], outputs
 * (softmax, nullable), final state.  st_* (all nullable): per-step records
 * = what LoopNTMTracker writes to its TensorArrays plus the BPTT stash. */
int ntk_ntm_seq_fwd(int B, int S, int N, int Md, int R, int Wh, int hid, int shift_range, int O,
                    int write_first,
                    const float* xproj, const float* Wr, const float* Wa,
                    const float* M0, const float* w0, const float* read0, const float* cs0,
                    float* logits, float* outputs,
                    float* M_out, float* w_out, float* read_out, float* cs_out,
                    float* st_z, float* st_gates, float* st_c, float* st_h, float* st_u,
                    float* st_wc, float* st_wv, float* st_w, float* st_M, float* st_read,
                    void* stream);

/* Full BPTT through a recorded sequence (tf.gradients through the while_loop,
 * direct_offset_output.py:611-621).  In: transposed weights WrT [4*hid][ldkT],
 * WaT [PP][ldhT], the records, dlogits [B,S,O], optional gradient of the final
 * state.  Out: raw gate gradients dgates [B,S,4*hid], raw control/logit
 * gradients du [B,S,PP], gradient of the initial state. */
int ntk_ntm_seq_bwd(int B, int S, int N, int Md, int R, int Wh, int hid, int shift_range, int O,
                    int write_first,
                    const float* WrT, int ldkT, const float* WaT, int ldhT,
                    const float* M0, const float* w0, const float* cs0,
                    const float* st_gates, const float* st_c, const float* st_u,
                    const float* st_wc, const float* st_wv, const float* st_w, const float* st_M,
                    const float* dlogits,
                    const float* dM_fin, const float* dw_fin, const float* dread_fin, const float* dcs_fin,
                    float* dgates, float* du, float* dM0, float* dw0, float* dread0, float* dcs0,
                    void* stream);

/* Stand-alone addressing ops (ops.py, called by ops_test.py): batched_smooth_cosine_similarity (:135-158) --
 * mode 0 = as coded (quirk Q1: feature columns normalised over the slot axis; what NTMCell computes),
 * mode 1 = true smooth cosine dot/(|m||k| + 1e-3) (what ops_test.py:20-34 pins); memory [B,N,Md], keys [B,H,Md],
 * out [B,H,N].  batched_circular_convolution (:180-214) with the Python-2 taps (quirk Q2): w [B,H,N],
 * kernel [B,H,shift_space]. */
int ntk_ntm_cosine_similarity(const float* memory, const float* keys, float* out, int B, int N, int Md, int H,
                              int mode, void* stream);
int ntk_ntm_circular_convolution(const float* w, const float* kernel, float* out, int B, int H, int N, int shift_space,
                                 void* stream);
/* The intermediate tensors of ONE cell step that NTMCell.__call__ returns in `debug` (ntm_cell.py:230-250) and the fused
 * step keeps in registers: sw (:161, softmax of the raw shift block at oS), w_gated (:153-156), powed_w_conv (:173), M_write /
 * M_erase (:197-203), from what the step records
 * (u [B,ldu] = the activated control vector with g at oG, gamma at oY, erase at oE, add at oA; wc, wv, w, w_prev [B,H,N]). */
int ntk_ntm_step_debug(const float* u, int ldu, int oG, int oS, int shift_space, int oY, int oE, int oA, const float* wc, const float* wv,
                       const float* w, const float* w_prev, float* sw, float* w_gated, float* w_conv_powed, float* M_write,
                       float* M_erase, int B, int N, int Md, int R, int Wh, void* stream);

/* One step of the cell (ntk_ntm_step_fwd/bwd) = the sequence kernels with S = 1; argument meaning as
 * ntk_ntm_seq_fwd / ntk_ntm_seq_bwd (the state AFTER the step takes the place of the final state). */
int ntk_ntm_step_fwd(int B, int N, int Md, int R, int Wh, int hid, int shift_range, int O, int write_first,
                     const float* xproj, const float* Wr, const float* Wa,
                     const float* M_prev, const float* w_prev, const float* read_prev, const float* cs_prev,
                     float* logits, float* outputs, float* M, float* w, float* read, float* cs,
                     float* st_z, float* st_gates, float* st_c, float* st_h, float* st_u,
                     float* st_wc, float* st_wv, float* st_w, float* st_M, float* st_read, void* stream);
int ntk_ntm_step_bwd(int B, int N, int Md, int R, int Wh, int hid, int shift_range, int O, int write_first,
                     const float* WrT, int ldkT, const float* WaT, int ldhT,
                     const float* M_prev, const float* w_prev, const float* cs_prev,
                     const float* st_gates, const float* st_c, const float* st_u,
                     const float* st_wc, const float* st_wv, const float* st_w, const float* st_M,
                     const float* dlogits,
                     const float* dM, const float* dw, const float* dread, const float* dcs,
                     float* dgates, float* du, float* dM_prev, float* dw_prev, float* dread_prev, float* dcs_prev,
                     void* stream);

/* tf.contrib.rnn.BasicLSTMCell pointwise step (ntm_cell.py:45-50; gate pre-activations pre [B,4*hid] = [x,h] W + b from
 * ntk_gemm_nt_f32, TF block order i | j | f | o): c = c_prev*sigmoid(f + forget_bias) + sigmoid(i)*tanh(j),
 * h = tanh(c)*sigmoid(o); act [B,4*hid] (nullable) keeps the activated gates for the backward, which returns the
 * gradient of the pre-activations and of c_prev (dh or dc may be null = zero). */
int ntk_lstm_step_fwd(const float* pre, const float* c_prev, float forget_bias, float* c, float* h, float* act,
                      int B, int hid, void* stream);
int ntk_lstm_step_bwd(const float* act, const float* c_prev, const float* c, const float* dh, const float* dc,
                      float* dpre, float* dc_prev, int B, int hid, void* stream);

/* trainable initial state (ntm_cell.py:284-315): out[b][i] = act(v[i]),
 * act 0 = tanh, 1 = sigmoid; and its gradient summed over the batch */
int ntk_ntm_init_state(const float* v, float* out, int n, int B, int act, void* stream);
int ntk_ntm_init_state_bwd(const float* v, const float* dout, float* dv, int n, int B, int act,
                           int accumulate, void* stream);

/* ------------------------------------------------------------------------
 * DNC core sequence kernel (forward)
 * replaces: dnc.DNC._build (dnc/dnc.py:84-127) unrolled by tf.nn.dynamic_rnn
 *           (direct_offset_output_with_dnc.py:66-88): snt.LSTM controller,
 *           MemoryAccess (dnc/access.py:113-303), CosineWeights / TemporalLinkage /
 *           Freeness (dnc/addressing.py), output linear, clip_value.
 * Packed parameters: Wr [ldz][4*hid] (rows [reads ; h], row K = b_gates, columns
 * unit*4+gate), Wi [ldh][IP] = the ten interface linears side by side in the order
 * write_vectors, erase_vectors, free_gate, allocation_gate, write_gate, read_mode,
 * write_keys, write_strengths, read_keys, read_strengths (row hid = biases),
 * Wy [ldy][OP] (rows [h ; reads], row Ky = bias).  State tensors are updated IN PLACE.
 * --------------------------------------------------------------------- */
int ntk_dnc_padded_dims(int N, int W, int R, int Wn, int hid, int O,
                        int* I, int* IP, int* K, int* ldz, int* ldh, int* Ky, int* ldy, int* OP);
int ntk_dnc_seq_fwd(int B, int S, int N, int W, int R, int Wn, int hid, int O, float clip_value,
                    const float* xproj, const float* Wr, const float* Wi, const float* Wy,
                    float* mem, float* link, float* usage, float* rw, float* ww, float* prec,
                    float* reads, float* hc, float* out,
                    /* per-step records for BPTT, all-or-none (null = inference): */
                    float* rec_z, float* rec_gates, float* rec_c, float* rec_hc, float* rec_yin,
                    float* rec_ifc, float* rec_u, float* rec_ww, float* rec_rw, float* rec_cw,
                    float* rec_cr, float* rec_al, float* rec_p, float* rec_fwd, float* rec_bwd,
                    float* rec_M, float* rec_L, float* rec_ypre, void* stream);

/* The same sequence kernel in CLUSTER form: k workgroups (one per CU) cooperate on each sequence -- the link rows and
 * the controller's hidden units are split k ways, the per-slot state and the N x W memory are replicated in LDS, and
 * two small exchanges per step go through a mailbox in `workspace` (csrc/dnc_cluster.h).  Same arguments, same state
 * in-place semantics and same records as ntk_dnc_seq_fwd; results differ from it by summation order only.
 * ntk_dnc_cluster_plan: the cluster size for a shape (k_request 0 = the largest that fits: B * k <= 256 CUs, link
 * rows + memory LDS resident; NTK_ERR_UNSUPPORTED and *k = 0 when the shape is outside the cluster kernels' range:
 * num_writes != 1, memory_size not a multiple of 64, ...) and the workspace size in bytes.  The workspace is
 * caller-owned device memory, 16-byte aligned, ZEROED ONCE by its owner, private to one launch at a time; its control
 * words are re-zeroed by every launch, its sticky error word (csrc/dnc_cluster.h) by ntk_dnc_cluster_status only.  ntk_dnc_cluster_status synchronises `stream` and reports whether a hand-off of the last launch on
 * that workspace timed out (every in-kernel spin is bounded; a launch that could not make progress aborts itself).
 * ntk_dnc_cluster_placement synchronises `stream` and reports how many of the B clusters of the last launch on that
 * workspace found all their k workgroups on one XCD and therefore ran the same-XCD form of the hand-offs (plain stores
 * kept in that XCD's L2; the others ran the write-through form: a speed difference only, csrc/dnc_cluster.h). */
int ntk_dnc_cluster_plan(int B, int N, int W, int R, int Wn, int hid, int O, int k_request, int* k, size_t* workspace_bytes);
int ntk_dnc_cluster_status(const void* workspace, int B, int k, void* stream);
int ntk_dnc_cluster_placement(const void* workspace, int B, int k, int* same_xcd_clusters, void* stream);
/* Device-side propagation of an aborted cluster launch, without a host synchronisation: when the sticky error word of
 * `workspace` (either form: mp_form 0 = ntk_dnc_cluster_*, 1 = ntk_dnc_mp_*, with its workspace_bytes) is set, loss[0]
 * and grad[0..n) become NaN: the NaN survives the data-parallel SUM all-reduce, so EVERY rank sees a NaN global norm and
 * ntk_rmsprop_clip_step_checked skips the update everywhere (zeros would have let the other ranks step on a partial
 * gradient).  loss / grad may be null.  ntk_dnc_cluster_inject_abort sets that sticky word the way a timed-out hand-off
 * does (fault injection for tests). */
int ntk_dnc_cluster_guard(const void* workspace, size_t workspace_bytes, int mp_form, int B, int k, float* loss, float* grad,
                          size_t n, void* stream);
int ntk_dnc_cluster_inject_abort(void* workspace, size_t workspace_bytes, int mp_form, int B, int k, void* stream);
int ntk_dnc_cluster_fwd(int B, int S, int N, int W, int R, int Wn, int hid, int O, float clip_value, int k,
                        const float* xproj, const float* Wr, const float* Wi, const float* Wy,
                        float* mem, float* link, float* usage, float* rw, float* ww, float* prec,
                        float* reads, float* hc, float* out,
                        float* rec_z, float* rec_gates, float* rec_c, float* rec_hc, float* rec_yin,
                        float* rec_ifc, float* rec_u, float* rec_ww, float* rec_rw, float* rec_cw,
                        float* rec_cr, float* rec_al, float* rec_p, float* rec_fwd, float* rec_bwd,
                        float* rec_M, float* rec_L, float* rec_ypre, void* workspace, void* stream);

/* Full BPTT in cluster form (arguments as ntk_dnc_seq_bwd; Wi is used un-transposed, WrT [4*hid][ldkT] as there):
 * d(link) rows LDS resident and split k ways, d(memory) register resident, two exchanges per step; bitwise
 * reproducible gradients (no float atomics).  Range: num_writes 1, memory_size a multiple of 64 up to 256,
 * word_size <= 64, hidden a multiple of 4 (ntk_dnc_cluster_bwd_plan says so for a shape; its own workspace). */
int ntk_dnc_cluster_bwd_plan(int B, int N, int W, int R, int Wn, int hid, int O, int k_request, int* k, size_t* workspace_bytes);
int ntk_dnc_cluster_bwd(int B, int S, int N, int W, int R, int Wn, int hid, int O, float clip_value, int k,
                        const float* WrT, int ldkT, const float* Wi, const float* Wy,
                        const float* mem0, const float* link0, const float* usage0, const float* rw0,
                        const float* ww0, const float* prec0, const float* hc0,
                        const float* rec_gates, const float* rec_c, const float* rec_ifc, const float* rec_u,
                        const float* rec_ww, const float* rec_rw, const float* rec_cw, const float* rec_cr,
                        const float* rec_al, const float* rec_p, const float* rec_fwd, const float* rec_bwd,
                        const float* rec_M, const float* rec_L, const float* rec_ypre,
                        const float* dout, float* gM, float* gL, float* dgates, float* dxi, float* dypre,
                        float* gcarry, int carry_in, void* workspace, void* stream);

/* The sequence kernels in MEMORY-PARTITIONED cluster form (csrc/dnc_mp.h): k workgroups per sequence, the N x N link
 * streamed through HBM once per step (N/k rows per workgroup; the BPTT record of step t-1 is the state step t reads),
 * the N x W memory partitioned by rows (LDS resident forward, register resident in BPTT), per-slot state replicated, four
 * hand-offs per step.  Nothing of size N x N or N x W is replicated, so this is the form for BASELINE configs[4]'s
 * core (memory 512 x 128: dnc/addressing.py:183-240 on 1 MiB of link per sequence-step), which the LDS-resident form
 * above cannot hold.  Same arguments, in-place state semantics and records as ntk_dnc_seq_fwd / ntk_dnc_seq_bwd; results
 * differ by summation order only and are bitwise reproducible run to run.
 * ntk_dnc_mp_plan / ntk_dnc_mp_bwd_plan: cluster size (k_request 0 = the smallest that fits; B * k <= the device's compute
 * units, queried per device) and the workspace size.  The workspace is caller-owned, 16-byte aligned, ZEROED ONCE by its
 * owner: besides the per-launch control words (re-zeroed by every launch) its last 256-byte line holds a STICKY error
 * word that a timed-out hand-off sets and no launch clears.  ntk_dnc_mp_status synchronises `stream` and fails when the
 * last launch or (sticky word) any launch since the word was last cleared aborted; clear_sticky != 0 clears it. */
/* Compute units the cooperative kernels may count on for the current device: hipDeviceAttributeMultiprocessorCount (assumes
 * the process has the device to itself -- a CU mask or another process's kernels are invisible to it), capped by the
 * environment variable NTK_DNC_CU_BUDGET when set; 0 when the device cannot be queried (the cluster forms are then refused
 * by every planner and the one-workgroup-per-sequence kernels run). */
int ntk_cu_count(void);
int ntk_dnc_mp_plan(int B, int N, int W, int R, int Wn, int hid, int O, int k_request, int* k, size_t* workspace_bytes);
/* > 0 when (shape, k) has a compile-time instantiation of the mp kernels (the generic one is functional but several times slower) */
int ntk_dnc_mp_compiled_shape(int N, int W, int R, int Wn, int hid, int O, int k);
int ntk_dnc_mp_status(const void* workspace, size_t workspace_bytes, int B, int k, int clear_sticky, void* stream);
int ntk_dnc_mp_placement(const void* workspace, int B, int k, int* same_xcd_clusters, void* stream);
int ntk_dnc_mp_fwd(int B, int S, int N, int W, int R, int Wn, int hid, int O, float clip_value, int k,
                   const float* xproj, const float* Wr, const float* Wi, const float* Wy,
                   float* mem, float* link, float* usage, float* rw, float* ww, float* prec,
                   float* reads, float* hc, float* out,
                   float* rec_z, float* rec_gates, float* rec_c, float* rec_hc, float* rec_yin,
                   float* rec_ifc, float* rec_u, float* rec_ww, float* rec_rw, float* rec_cw,
                   float* rec_cr, float* rec_al, float* rec_p, float* rec_fwd, float* rec_bwd,
                   float* rec_M, float* rec_L, float* rec_ypre, void* workspace, void* stream);

int ntk_dnc_mp_bwd_plan(int B, int N, int W, int R, int Wn, int hid, int O, int k_request, int* k, size_t* workspace_bytes);
int ntk_dnc_mp_bwd(int B, int S, int N, int W, int R, int Wn, int hid, int O, float clip_value, int k,
                   const float* WrT, int ldkT, const float* Wi, const float* Wy,
                   const float* mem0, const float* link0, const float* usage0, const float* rw0,
                   const float* ww0, const float* prec0, const float* hc0,
                   const float* rec_gates, const float* rec_c, const float* rec_ifc, const float* rec_u,
                   const float* rec_ww, const float* rec_rw, const float* rec_cw, const float* rec_cr,
                   const float* rec_al, const float* rec_p, const float* rec_fwd, const float* rec_bwd,
                   const float* rec_M, const float* rec_L, const float* rec_ypre,
                   const float* dout, float* gM, float* gL, float* dgates, float* dxi, float* dypre,
                   float* gcarry, int carry_in, void* workspace, void* stream);

/* Stand-alone DNC addressing modules (dnc/addressing.py), the module-level API the reference's own tests call:
 * CosineWeights._build (:83-105), TemporalLinkage._build (:133-153) and directional_read_weights (:155-181),
 * Freeness._build (:279-305) and write_allocation_weights (:307-340; one head: _allocation :376-405). */
int ntk_dnc_cosine_weights(const float* memory, const float* keys, const float* strengths, float* out,
                           int B, int N, int W, int H, void* stream);
int ntk_dnc_linkage(const float* prev_link, const float* prev_prec, const float* write_weights, float* link,
                    float* prec, int B, int N, int Wn, void* stream);
int ntk_dnc_directional_read_weights(const float* link, const float* prev_read_weights, float* out, int B, int N,
                                     int Wn, int R, int forward, void* stream);
int ntk_dnc_freeness(const float* write_weights, const float* free_gate, const float* read_weights,
                     const float* prev_usage, float* usage, int B, int N, int Wn, int R, void* stream);
int ntk_dnc_write_allocation_weights(const float* usage, const float* write_gates, float* out, int B, int N, int Wn,
                                     void* stream);

/* MemoryAccess pieces (dnc/access.py): the activations of _read_inputs (:160-218) on the RAW outputs of the ten
 * interface linears in the packed order [write_vectors | erase_vectors | free_gate | allocation_gate | write_gate |
 * read_mode | write_keys | write_strengths | read_keys | read_strengths] (row stride ldr); `act` receives each field
 * as a contiguous [B,width] array at act + B*offset(field).  _write_weights (:220-257), _erase_and_write (:32-63),
 * _read_weights (:259-303), read words (:151), and the whole step MemoryAccess._build (:113-158) = SURVEY's
 * ntk_dnc_access_step_fwd (its backward: ntk_dnc_access_step_bwd, below).
 * Workspaces (floats): write_weights 2*B*Wn*N + B*Wn; read_weights B*R*N*(1+2*Wn). */
int ntk_dnc_interface_activations(const float* raw, int ldr, float* act, int B, int N, int W, int R, int Wn, void* stream);
int ntk_dnc_write_weights(const float* memory, const float* usage, const float* write_keys, const float* write_strengths,
                          const float* allocation_gate, const float* write_gate, float* write_weights, float* workspace,
                          int B, int N, int W, int Wn, void* stream);
int ntk_dnc_erase_and_write(const float* memory, const float* address, const float* reset_weights, const float* values,
                            float* out, int B, int N, int W, int Wn, void* stream);
int ntk_dnc_read_weights(const float* memory, const float* prev_read_weights, const float* link, const float* read_keys,
                         const float* read_strengths, const float* read_mode, float* read_weights, float* workspace,
                         int B, int N, int W, int R, int Wn, void* stream);
int ntk_dnc_read_words(const float* read_weights, const float* memory, float* out, int B, int N, int W, int R, void* stream);
size_t ntk_dnc_access_step_workspace_bytes(int B, int N, int W, int R, int Wn);
int ntk_dnc_access_step_fwd(const float* iface_raw, int ldr, const float* memory, const float* read_weights,
                            const float* write_weights, const float* link, const float* precedence, const float* usage,
                            float* memory_out, float* read_weights_out, float* write_weights_out, float* link_out,
                            float* precedence_out, float* usage_out, float* read_words, float* workspace,
                            int B, int N, int W, int R, int Wn, void* stream);

/* Backward of one MemoryAccess step at module granularity (what tf.gradients gives dnc/access_test.py:145-159).  The step
 * is recomputed from the PREVIOUS state and the raw interface; d_read_words [B,R,W] is the gradient w.r.t. the step's read
 * words; g_memory [B,N,W], g_read_weights [B,R,N], g_link [B,Wn,N,N], g_precedence [B,Wn,N], g_usage [B,N] are IN/OUT: in =
 * gradient w.r.t. the NEW state's field (zeros when the loss does not see it), out = gradient w.r.t. the previous state's
 * (write weights get none: they reach the next usage under stop_gradient only, addressing.py:302).  d_iface_raw [B,IP]
 * (IP from ntk_dnc_padded_dims): gradient w.r.t. the raw interface; the ten linears' gradients are GEMMs over it.
 * N and W multiples of 4, R <= 4, Wn <= 4.  Workspace: ntk_dnc_access_step_bwd_workspace_bytes. */
size_t ntk_dnc_access_step_bwd_workspace_bytes(int B, int N, int W, int R, int Wn);
int ntk_dnc_access_step_bwd(const float* iface_raw, int ldr, const float* memory, const float* read_weights,
                            const float* write_weights, const float* link, const float* precedence, const float* usage,
                            const float* d_read_words, float* g_memory, float* g_read_weights, float* g_link,
                            float* g_precedence, float* g_usage, float* d_iface_raw, float* workspace,
                            int B, int N, int W, int R, int Wn, void* stream);

/* Full BPTT through a recorded DNC sequence (num_writes 1..4: one write head runs the tuned kernel, 2..4 the
 * general one).  WrT [4*hid][ldkT], WiT [IP][ldhT]
 * are transposed copies of Wr / Wi; *0 pointers are the state BEFORE step 0; gM [B,N,W] and
 * gL [B,Wn,N,N] are zero-initialised scratch.  Out: raw gate gradients dgates [B,S,4*hid], raw
 * interface gradients dxi [B,S,IP], gradient of the pre-clip output dypre [B,S,OP]; weight
 * gradients follow as ntk_gemm_tn_f32 over the recorded rows.
 * Segmented BPTT (long sequences, config 5): run the segments last to first, re-recording each from its
 * checkpointed state; gM / gL are NOT re-zeroed between segments and gcarry [B, (Wn+1)*N + R*N + ldkT + hid]
 * (optional, may be null) carries the remaining state gradients: read when carry_in != 0, always written. */
int ntk_dnc_seq_bwd(int B, int S, int N, int W, int R, int Wn, int hid, int O, float clip_value,
                    const float* WrT, int ldkT, const float* WiT, int ldhT, const float* Wy,
                    const float* mem0, const float* link0, const float* usage0, const float* rw0,
                    const float* ww0, const float* prec0, const float* hc0,
                    const float* rec_gates, const float* rec_c, const float* rec_ifc, const float* rec_u,
                    const float* rec_ww, const float* rec_rw, const float* rec_cw, const float* rec_cr,
                    const float* rec_al, const float* rec_p, const float* rec_fwd, const float* rec_bwd,
                    const float* rec_M, const float* rec_L, const float* rec_ypre,
                    const float* dout, float* gM, float* gL, float* dgates, float* dxi, float* dypre,
                    float* gcarry, int carry_in, void* stream);

/* ------------------------------------------------------------------------
 * tracking head (direct_offset_output.py)
 * --------------------------------------------------------------------- */

/* 64-point gather from conv4_3 (:392-399, receptive_field_sizes.py:135-143)
 * + input serialiser (:439-500): fmap [B*T,Hf,Wf,C] -> X [B, T*(n*n+1), ldx]
 * rows [feat(C), delimiter, target, 0 pad]; gts0 [B, n*n] = frame-0 heat-map
 * (nullable -> zeros). */
int ntk_gather_serialize(const float* fmap, const float* gts0, float* X, int B, int T,
                         int Hf, int Wf, int C, int ldx, int grid_start, int grid_step,
                         int grid_n, void* stream);

/* inference serialisation (quirk Q8, test_tracker.py:400-404): the delimiter row comes FIRST */
int ntk_gather_serialize_online(const float* fmap, const float* gts0, float* X, int B, int T,
                                int Hf, int Wf, int C, int ldx, int grid_start, int grid_step,
                                int grid_n, void* stream);

/* tf.image.crop_and_resize (bilinear, one normalised box y1,x1,y2,x2, extrapolation value) of (image - mean):
 * image [H,W,C] fp32, mean [C] (nullable) -> out [crop_h,crop_w,C]
 * replaces: test_tracker.py:344-352 (online) / direct_offset_output.py:207-211 (training input pipeline) */
int ntk_crop_and_resize(const float* image, int H, int W, int C, const float* mean, float y1, float x1,
                        float y2, float x2, float* out, int crop_h, int crop_w, float extrapolation,
                        void* stream);

/* tf.image.resize_images(img, [out_h, out_w]) (bilinear, TF-1 defaults) -- direct_offset_output.py:193 */
int ntk_resize_bilinear(const float* image, int H, int W, int C, float* out, int out_h, int out_w, void* stream);

/* output gather + tanh + l2 loss (:581-606) and its gradient:
 * pred [B,T-1,O] (nullable), loss [1], dlogits [B,S,O] (nullable). */
int ntk_offset_loss(const float* logits, const float* offsets, float* pred, float* loss,
                    float* dlogits, int B, int T, int NF, int O, void* stream);

/* the same in two calls (ntk_offset_loss_fwd/bwd) */
int ntk_offset_loss_fwd(const float* logits, const float* offsets, float* pred, float* loss, int B, int T, int NF, int O,
                        void* stream);
int ntk_offset_loss_bwd(const float* logits, const float* offsets, float* dlogits, int B, int T, int NF, int O, void* stream);

/* Sequential presentation + heat-map head of main.py's earlier trackers (ntm_sevenbyseven,
 * main.py:1646-1969; the same serialisation in :979-1291).  Every position of the feature map is a feature
 * (F = Hf * Wf), rows are [feat(C), feature delimiter, frame delimiter, target, 0 pad]:
 *   frame 0: F rows [feat_i, 0, 0, gt0_i];  frame t >= 1: one frame-delimiter row, then per feature the rows
 *   [feat_i, 0, 0, 0] and [0.., 1, 0, 0]  ->  S = F + (T - 1)(2 F + 1) steps (main.py:1701-1775).
 * fmap [B*T, F, C] (C a multiple of 4), gts0 [B, F] (nullable), X [B, S, ldx] (ldx >= C + 3, multiple of 4).
 * Loss (main.py:1880-1922): the cell has output_dim 1; the logits at the FEATURE-DELIMITER steps of frames 1..T-1 form
 * an F-way score vector per frame; loss = sum softmax_cross_entropy_with_logits(scores, gt[b, t]) / (T - 1).
 * logits [B, S] (= [B,S,1]), gt [B, T-1, F] soft labels; probs [B, T-1, F] (nullable), loss [1], dlogits [B, S] (nullable). */
int ntk_serialize_sequential(const float* fmap, const float* gts0, float* X, int B, int T, int F, int C, int ldx, void* stream);
int ntk_heatmap_ce_loss(const float* logits, const float* gt, float* probs, float* loss, float* dlogits,
                        int B, int T, int F, void* stream);

/* Two-step presentation + (F+1)-way head of main.py's ntm_two_step (:862-977; ntm_tracker_new.py:112-195 with
 * two_step=True): a whole frame is ONE step, rows [switch, feat(D), target(F)]: step 0 = [0, feat_0, target], frame
 * t >= 1 = the presentation step [0, feat_t, 0] then the query step [1, 0, 0]  ->  S = 2T - 1 steps.
 * feat [B, T, D] (the flattened, optionally compressed feature map), target [B, F] (nullable), X [B, S, ldx], ldx >= 1 + D + F.
 * Loss (:903-951): labels are the background row [0..0, 1] at step 0 and at every presentation step and [gt_t, 0] at the
 * query step of frame t; they pass through a softmax before the cross entropy (as coded);
 * loss = sum_rows CE(logits_row, softmax(label_row)) / ((2T - 1) B).  logits [B, S, F+1], gt [B, T, F] (row 0 unused),
 * probs [B, S, F+1] (nullable), loss [1], dlogits [B, S, F+1] (nullable). */
int ntk_serialize_two_step(const float* feat, const float* target, float* X, int B, int T, int D, int F, int ldx, void* stream);
int ntk_two_step_ce_loss(const float* logits, const float* gt, float* probs, float* loss, float* dlogits,
                         int B, int T, int F, void* stream);

/* copy-task head (main.py:1603-1610, BASELINE configs[0]): loss = tf.losses.log_loss(labels,
 * sigmoid(logits)) (mean over all n elements, epsilon 1e-7) and d loss / d logits (nullable). */
int ntk_log_loss(const float* logits, const float* labels, float* loss, float* dlogits, int n, void* stream);

/* ------------------------------------------------------------------------
 * optimiser (direct_offset_output.py:620-626): tf.clip_by_global_norm +
 * tf.train.RMSPropOptimizer on one flat buffer
 * --------------------------------------------------------------------- */
size_t ntk_global_norm_workspace_bytes(size_t n);
int ntk_global_norm(const float* grads, size_t n, float* workspace, float* gnorm, void* stream);
int ntk_rmsprop_clip_step(float* params, const float* grads, float* ms, float* mom, size_t n,
                          float lr, float decay, float momentum, float eps, float clip_norm,
                          const float* gnorm, void* stream);
/* The same update behind a finiteness check of *gnorm (required): when the global norm is NaN or Inf -- a poisoned
 * gradient (ntk_dnc_cluster_guard), on ANY data-parallel rank once the SUM all-reduce has run -- parameters and slots
 * stay untouched, loss[0] (nullable) becomes NaN and *skipped (nullable, device counter) is incremented; otherwise it is
 * ntk_rmsprop_clip_step bit for bit.  What the trackers' training steps run (the reference has no failure path here:
 * tf.clip_by_global_norm would write NaN into every variable). */
int ntk_rmsprop_clip_step_checked(float* params, const float* grads, float* ms, float* mom, size_t n,
                                  float lr, float decay, float momentum, float eps, float clip_norm,
                                  const float* gnorm, float* loss, unsigned* skipped, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NTMTRACK_H_ */
