/*
 * voicemap_hip.h -- C ABI of libvoicemap_hip.so (gfx950 / MI355X).
 *
 * The reference (oscarknagg/voicemap) has no FFI of its own: its hot path is Keras-2.2.2 layer calls
 * that TensorFlow lowers to cuDNN/Eigen kernels.  Each entry point below replaces one such implicit
 * device op (or a fusion of several); the reference line that *creates* the op is cited.  Host code
 * (voicemap_amd/ *.py, the mirror of voicemap/{models,utils,librispeech}.py) binds these with ctypes;
 * INTEGRATION.md shows the stub.
 *
 * Conventions (every function):
 *   - plain pointers are DEVICE pointers unless named host_*; sizes are explicit; no torch types.
 *   - `dtype` selects the storage type of activations / GEMM operands: VM_F32, VM_F32S, VM_BF16 or VM_F16 (the enum below).
 *     Accumulation, batch statistics, the tail (global max -> dense -> head -> loss) and the optimizer are always fp32.
 *   - enqueue-only on `stream` (a hipStream_t passed as void*); no host sync, no allocation; re-entrant per
 *     stream.  Process-global mutable state: (a) the kernel-selection table behind vm_set_tuning, (b) a pool of zero-initialised
 *     ticket words in device memory that the fused two-stage reductions draw from (round robin per launch; every launch leaves its
 *     words zero again) -- it makes those entry points, like the table, not thread-safe across host threads.  The table
 *     (below): every selectable kernel computes the same result (each is parity-tested against the oracle), so
 *     the table changes speed, never values; it is read at launch time and is not thread-safe -- set it before
 *     work is enqueued from other threads, or leave the defaults (what the drop-in surface does).
 *   - returns 0 on success, <0 on error (VM_ERR_*); vm_last_error() gives a thread-local message.
 *   - activation layout is Keras' channels-last.  "padded" tensors are (N, L+2, C) with one zero halo
 *     row before and after each window so that a k=3 SAME convolution reads rows t..t+2 with no branch;
 *     the halo rows are zeroed once by the owner (vm_fill_zero) and never written.
 *   - "towers": BatchNorm statistics are taken per encoder call (voicemap/models.py:52-53 calls the
 *     shared encoder twice), so windows [0,wpt) are tower 0, [wpt,2*wpt) tower 1, ... with
 *     wpt = windows_per_tower.
 */
#ifndef VOICEMAP_HIP_H
#define VOICEMAP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* VM_F32S: fp32 storage like VM_F32; the k=3 convolution GEMMs (vm_conv_fwd / vm_conv_dgrad / vm_conv_wgrad) form every product
 * from bf16 hi/lo halves of the fp32 operands (a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, fp32 accumulation: ~2^-17 relative per product
 * instead of exact) on the bf16 matrix pipe, which is 16x faster than the fp32 one.  Every other entry point treats it as VM_F32. */
/* VM_F16: IEEE half storage (11 significand bits against bf16's 8: ~8x less rounding per stored value at the same bytes and the same
 * matrix-pipe rate, v_mfma_f32_32x32x16_f16).  Every entry point that takes VM_BF16 takes it.  Its exponent range is narrow (6e-8 ..
 * 65504), so the caller scales the loss gradient (vm_siamese_head_loss / vm_softmax_cce `grad_scale`) and un-scales in
 * vm_adam_clip_step (`grad_prescale`); everything between them is linear in the gradient. */
enum { VM_F32 = 0, VM_BF16 = 1, VM_F32S = 2, VM_F16 = 3 };
enum { VM_OK = 0, VM_ERR_ARG = -1, VM_ERR_LAUNCH = -2, VM_ERR_UNSUPPORTED = -3 };
enum { VM_LOSS_CONTRASTIVE = 0, VM_LOSS_BCE = 1 };
enum { VM_HEAD_UNIFORM_EUCLIDEAN = 0, VM_HEAD_WEIGHTED_L1 = 1 };
enum { VM_DIST_EUCLIDEAN = 0, VM_DIST_COSINE = 1, VM_DIST_DOT = 2 };

const char* vm_last_error(void);
/* 11.  History: 11 = vm_program_run / vm_program_table_hash, the native runner of a recorded step (round 6); 10 = config 4's log-mel image at twice the storage significand (vm_stft_logmel_f16s_split, vm_conv2d_first_fwd_split, vm_conv2d_first_bn_pool_stack, vm_bn_pool2d_stack_fwd_split) (round 6); 9 = the last block in pair form (vm_bn_drop_pool_gmax_partials_e, vm_bn_bwd_gmax_finalize_e, vm_bn_pool_bwd_apply_pairs_gmax) (round 6); 8 = vm_mfma_rate_probe[_flops], vm_bn_bwd_gmax_finalize; vm_pairdist_workspace_bytes grew by the scalar-path copy of the queries (round 6); 7 = the fused tail (vm_tail_fwd_bwd, vm_tail_param_grads, vm_bn_drop_pool_gmax_partials), vm_event_* / vm_stream_wait_event, centred tiles (`ctr_out` of vm_fold_bn_weights, `e_center` of vm_conv_fwd_fold / vm_bn_pool_bwd_apply_pairs, `tile_center` of vm_bn_finalize) (round 5); 6 = packed weights (vm_pack_nt_weights[_batch]; the `*_packed` argument of vm_conv_fwd_fold / vm_conv_fwd_pool /
 * vm_conv_dgrad_bnred; `bias`, `wf_packed` and the fourth hb row of vm_fold_bn_weights), the centred block-1 extreme (`center_bias` /
 * `shift_adj` / `mean_adj` of vm_bn_finalize) (round 4).  Earlier: 1 = round 1; 2 = vm_bn_finalize gained the zero-debias arguments (round 2); 3 = VM_F16, `dtype` in vm_conv1_fused_*,
 * `grad_scale` in the loss entry points, `skip_nonfinite` in vm_adam_clip_step, vm_embed_* / vm_pairdist_* (round 3); 4 = the folded-BatchNorm training forward
 * (vm_fold_bn_weights, vm_conv_fwd_fold, vm_conv_wgrad_fold, vm_du_tower_sums, vm_bn_pool_bwd_apply_pairs; vm_conv1_fused_fwd mode 2;
 * `wt` in vm_prep_conv_weights_batch; `sqnorm_parts` in vm_adam_clip_step; vm_siamese_head_reduce; vm_conv_fwd_flat, vm_conv2d_first_*,
 * `src_padded` in vm_fold_windows) (round 3); 5 = vm_bn_pool2d_stack_fwd, vm_fold_pool_windows_bwd, vm_colsum_strided, the operand-order basis of
 * vm_stft_split_basis (round 3). */
int vm_abi_version(void);
/* device smoke: hipGetDeviceProperties gcnArchName must start with "gfx950". */
int vm_check_device(void);

int vm_fill_zero(void* ptr, int64_t bytes, void* stream);
/* Stream-ordering primitives (hipEventCreateWithFlags(DisableTiming) / hipEventRecord / hipStreamWaitEvent) for a host that REPLAYS a
 * recorded sequence of the calls below with the ordering between its streams (voicemap_amd/engine.py records one training step per
 * configuration -- the reference's train_on_batch, experiments/train_siamese.py:65-94 runs it 500 times per epoch with identical
 * shapes -- and replays it without re-deriving pointers and dispatch decisions: at the reference's batch sizes the step is bound by
 * the host).  event_out receives an opaque handle. */
int vm_event_create(void** event_out);
int vm_event_destroy(void* event);
int vm_event_record(void* event, void* stream);
int vm_stream_wait_event(void* stream, void* event);
/* A recorded training step made from C (round 6).  The Python side records the C-ABI calls of one step -- entry point, argument list,
 * event records / waits between its streams -- and replays that list while the configuration stays the same (voicemap_amd/engine.py
 * _Program; the reference runs the same train_on_batch 500 times an epoch, experiments/train_siamese.py:65-94).  At the reference's own
 * batch sizes the step is bound by the host making ~70 calls, and ~1.5 us of each is the binding's argument marshalling: vm_program_run
 * takes the list as int64 words  [function id, argc, argument words ...]*  (pointers and integers as they are, floats / doubles as their
 * bits) and makes the calls itself, in order, stopping at the first non-zero return code (its word index in *fail_at).  Function ids are
 * positions in the name-sorted table of int-returning entry points whose arguments are pointers / int / int64_t / float / double
 * (csrc/program_run.hip, generated by tools/gen_program_run.py from the binding table); vm_program_table_hash() identifies that table so
 * that a binding generated from another one refuses to run. */
int vm_program_run(const int64_t* words, int64_t n_words, int64_t* fail_at);
int64_t vm_program_table_hash(void);
/* Measurement aid (round 6; not part of the drop-in surface, bench.py only): every SIMD of the device issuing dense
 * v_mfma_f32_32x32x16 (VM_BF16 / VM_F16 operands, non-zero values) back to back from registers for `iters` x 8 instructions per
 * wave -- the rate the part SUSTAINS under its power limit, which the step's GEMM launches (they run on that limit) are priced
 * against beside the nominal 2.5 PFLOP/s.  sink: 512 * 256 floats (never written).  vm_mfma_rate_probe_flops: the FLOPs of one launch. */
int vm_mfma_rate_probe(int dtype, int iters, float* sink, void* stream);
int64_t vm_mfma_rate_probe_flops(int iters);
/* Kernel-selection table for the tests that pin a fallback kernel and for A/B measurements (process-global, see the conventions
 * above; not part of the drop-in surface: voicemap_amd never calls it outside bench.py --tune).  Unknown keys / values out of
 * range return VM_ERR_ARG.  Keys:
 *   "nt_n2" 0..3      forward / dgrad, 16-bit storage: conv_nt2r_kernel (254 x 128 tiles, two workgroups per CU, input-resident A
 *                     by LDS-DMA; bit 0 forward, bit 1 dgrad; default 3); where it is off or does not serve the shape:
 *   "nt3" 0..3, "nt3_lean" 0..3   conv_nt3_kernel (weights L2 -> registers from the packed copy) for forward (bit 0) / dgrad (bit 1) where
 *                     the packed weights are given, and its lean prologue (defaults 3, 3); off: conv_nt2r_kernel
 *   "nt_glds" 0|1     the 128 x 128 LDS-DMA kernel where K * sizeof(T) % 64 == 0 (default 1), else the register-staged 128 x 128 one
 *   "tn_x" 0|1        wgrad, 16-bit storage: the input-resident (3 taps x 128 ci) x 128 co LDS-DMA tile (default 1), else
 *   "tn9" 0|1         ... with the free-running K loop (conv_tn9_kernel, default 1) or the READ / MFMA slots (conv_tn8x_kernel)
 *   "tn9_stages" 0|1  conv_tn9_kernel's split-K ranges are 64-position stages of a tower's window stream (default 1: the launch is
 *                     balanced to one stage) or whole windows.  Changes vm_conv_wgrad_splits / vm_conv_wgrad_workspace_bytes: set it
 *                     before plans are sized
 *   "tn_tile" 128|256 the tile of the register-transposing wgrad kernels (default 256 where the layer is wide enough)
 *   "fuse_finalize" mask 0..31   which two-stage column reductions finish inside their stage-1 launch (the last-arriving workgroup of
 *                     a channel block runs the finalize; bit-identical to the two launches): bit 0 vm_bn_finalize, bit 1
 *                     vm_bn_bwd_finalize / vm_bn_bwd_from_sums_finalize, bit 2 vm_colsum*, bit 3 vm_du_tower_sums, bit 4 = only where
 *                     the reduction is narrow and short (C <= 128, <= 4096 rows per tower).  Default 17: the statistics of small
 *                     layers -- a saved launch pays only there (measurements in bnpool.hip)
 *   "f1_blocks", "f1_fwd_blocks"   target workgroup counts of the fused block-1 kernels (launch geometry; the fp32 partial sums of a
 *                     window are grouped differently, i.e. results change in the last bits; defaults 1024, 1024).
 *   "f1_products" 1|2|3   (round 6; VM_F16 storage only -- VM_BF16 always takes 3) the 16-bit products that carry the block-1
 *                     convolution: 3 = waveform and filters split hi + lo in bf16 (xh*wh + xl*wh + xh*wl, ~16 significand bits of
 *                     both); 2 (default) = the waveform rounded to half, the filters split hi + lo in halves (xh*wh + xh*wl); 1 =
 *                     xh*wh.  NOT result-preserving: it is the operand precision of block 1 (embeddings against the CPU oracle at the
 *                     bench batch 6.98e-4 / 7.37e-4 / 7.99e-4, profiles/r06_block1_products.txt).
 *   "apply_order" 0|1|2   walk of vm_bn_pool_bwd_apply* over the windows: 0 = the segments of all windows together, 1 / 2 =
 *                     window-major ascending / descending (default 2: it starts on the windows the producer of dp wrote last, which the
 *                     memory-side cache still holds).  Bit-identical. */
int vm_set_tuning(const char* key, int value);

/* ---- a6: preprocess_instances / whiten  (voicemap/utils.py:22-34, 88-101) --------------------------
 * raw: (n_windows, raw_len) fp32 (or int16 if raw_is_i16, scaled by 1/32768).  Takes every
 * `downsampling`-th sample (utils.py:29), subtracts the per-window mean (utils.py:94-95) and multiplies
 * by ONE scalar per tower rms/sqrt(mean(batch^2)) of the un-centred decimated batch (utils.py:98).
 * out: (n_windows, L0 + 31) fp32 = the conv-1 input with its SAME halo (15 zeros left, 16 right).
 * ws: >= vm_decimate_whiten_workspace_bytes(n_windows).  whitening=0 skips utils.py:30-31. */
int64_t vm_decimate_whiten_workspace_bytes(int64_t n_windows);
int vm_decimate_whiten(const void* raw, int raw_is_i16, int64_t n_windows, int64_t raw_len, int downsampling,
                       int whitening, float rms, int64_t windows_per_tower, float* out, void* ws, void* stream);
/* The same with the crop of voicemap/librispeech.py:103-137 done on the device: `audio` is a resident buffer of decoded
 * recordings (int16 or fp32, all files back to back -- voicemap_amd/shards.py), window n is the raw_len samples starting at
 * audio[offsets[n]] (offsets: n_windows int64 on the device).  The host only chooses the offsets. */
int vm_crop_decimate_whiten(const void* audio, int raw_is_i16, const int64_t* offsets, int64_t n_windows, int64_t raw_len,
                            int downsampling, int whitening, float rms, int64_t windows_per_tower, float* out, void* ws,
                            void* stream);

/* ---- a1 block 1: Conv1D(filters, 32, padding='same', activation='relu')  (voicemap/models.py:13-16) --
 * x: (n_windows, L + 31) fp32 from vm_decimate_whiten; w: (32, 1, F) fp32 Keras layout; bias (F).
 * z: (n_windows, L, F) `dtype`, = relu(conv + bias).  If stat_sum != NULL also writes per-(window, tile)
 * partial sums of z and z^2 (fp32) for the BatchNorm that follows: stat_sum/stat_sq are
 * (n_windows * vm_conv1_stat_rows(L), F). */
int64_t vm_conv1_stat_rows(int64_t L);
int vm_conv1_fwd(const float* x, const float* w, const float* bias, int64_t n_windows, int64_t L, int F,
                 int dtype, void* z, float* stat_sum, float* stat_sq, void* stream);
/* wgrad of block 1: dW[k][c] = sum_{n,t} x[n][t+k] * du[n][t][c] over all windows.
 * du: (n_windows, L+2, F) padded `dtype`.  ws: vm_conv1_wgrad_workspace_bytes() of scratch for the per-window partials;
 * grad_w (32,1,F) is overwritten with the fixed-order sum (deterministic). */
int64_t vm_conv1_wgrad_workspace_bytes(int64_t n_windows, int F);
int vm_conv1_wgrad(const float* x, const void* du, int64_t n_windows, int64_t L, int F, int dtype,
                   float* ws, float* grad_w, void* stream);

/* Fused block 1 for 16-bit storage (dtype VM_BF16 or VM_F16; voicemap/models.py:13-19: Conv1D(F,32) -> BatchNormalization -> SpatialDropout1D ->
 * MaxPool1D(pool)): the full-resolution relu(conv) tensor is never written.
 *   training (inference = 0): out = e (n_windows, L/pool, F) bf16 = per-pool-window max (gamma >= 0) or min (gamma < 0) of
 *     relu(conv+b), rounded to bf16 -- BN is a monotone per-channel affine, so pooling commutes with it; the affine +
 *     dropout are then applied to e by vm_bn_drop_pool_fwd(z = e, L = L/pool, pool = 1).  gamma_or_scale = gamma (F);
 *     shift ignored; stat_sum / stat_sq as for vm_conv1_fwd (over all L positions), taken from the fp32 accumulator:
 *     relu(conv+b) itself is never stored, so it has no storage rounding -- the statistics, the pool arg-max and the
 *     backward recompute all see the same fp32 values.
 *   inference (inference = 1): out = padded act (n_windows, L/pool + 2, F) bf16 = bf16(extreme * scale + shift);
 *     gamma_or_scale = scale, shift from vm_bn_infer_affine.
 *   training, padded extreme (inference = 2): as 0, but e is written as a padded activation tensor (n_windows, L/pool + 2, F), rows
 *     1 .. L/pool (halo rows untouched: the caller zeroes them once) -- the input layout of vm_conv_fwd_fold.
 * pool: 2 or 4. */
int vm_conv1_fused_fwd(const float* x, const float* w, const float* bias, const float* gamma_or_scale, const float* shift,
                       int64_t n_windows, int64_t L, int F, int pool, int inference, int dtype, void* out, float* stat_sum,
                       float* stat_sq, void* stream);
/* Backward of the same block from dp (n_windows, L/pool, F) `dtype`: recomputes the conv tile on the matrix cores,
 * evaluates the pool/dropout/BN/ReLU backward in registers (c1, c2 from vm_bn_pool_bwd_reduce(z = e, pool = 1) +
 * vm_bn_bwd_finalize(count = wpt*L)) and accumulates grad_w (32,1,F) and grad_b (F) (overwritten; fixed order). */
int64_t vm_conv1_fused_bwd_workspace_bytes(int64_t n_windows, int64_t L, int F);
int vm_conv1_fused_bwd(const float* x, const float* w, const float* bias, const void* dp, const float* scale,
                       const float* mean, const float* invstd, const float* drop, const float* c1, const float* c2,
                       int64_t n_windows, int64_t windows_per_tower, int64_t L, int F, int pool, int dtype, void* ws,
                       float* grad_w, float* grad_b, void* stream);

/* ---- a1 blocks 2-4: Conv1D(c_out, 3, padding='same', activation='relu')  (voicemap/models.py:22,27,32)
 * implicit GEMM on MFMA.  in: padded (n_windows, L+2, c_in) `dtype`; wf: (c_out, 3*c_in) `dtype` from
 * vm_prep_conv_weights; z: (n_windows, L, c_out) `dtype`.  stat_* as for vm_conv1_fwd with
 * vm_conv_stat_rows(L) rows per window (NULL in inference). */
int64_t vm_conv_stat_rows(int64_t L);
int vm_conv_fwd(const void* in, const void* wf, const float* bias, int64_t n_windows, int64_t L, int c_in,
                int c_out, int dtype, void* z, float* stat_sum, float* stat_sq, void* stream);
/* vm_conv_fwd for windows too short to fill a tile (the 2-D variant: 149 .. 37 positions): every window of `in` carries its own zero
 * halo rows, so the concatenation of all windows is one valid k = 3 sequence; it is run as ONE window on the 128-row kernels and the
 * epilogue drops the halo positions (their results are junk) and writes the rest to the same un-padded z (n_windows, L, c_out) --
 * bit-identical to vm_conv_fwd.  stat_sum / stat_sq: vm_conv_flat_stat_rows(n_windows, L) rows in all (one per 128 rows of the
 * concatenation) instead of vm_conv_stat_rows(L) per window: a caller with BatchNorm statistics per tower launches once per tower. */
int64_t vm_conv_flat_stat_rows(int64_t n_windows, int64_t L);
int vm_conv_fwd_flat(const void* in, const void* wf, const float* bias, int64_t n_windows, int64_t L, int c_in, int c_out, int dtype,
                     void* z, float* stat_sum, float* stat_sq, void* stream);
/* vm_conv_fwd (training form: statistics required) that also writes the pool-window extreme of z for MaxPool1D(2):
 * e[n][q][c] = max(z[n][2q][c], z[n][2q+1][c]) where gamma[c] >= 0, the min where gamma[c] < 0 -- the element the max-pool of the
 * BatchNorm output will select, known before the statistics are (sign(scale) = sign(gamma)).  e: unpadded (n_windows, L/2, c_out).
 * vm_bn_drop_pool_fwd(e, ..., L/2, pool = 1) then gives the bit-identical pooled output from a pooled-size tensor, and
 * vm_conv_dgrad_bnred can take its sums against e directly (red_a_padded = 0).  Same kernel restriction as vm_conv_fwd_pool. */
int vm_conv_fwd_e_supported(int64_t n_windows, int64_t L, int c_in, int c_out, int dtype);
int vm_conv_fwd_e(const void* in, const void* wf, const float* bias, const float* gamma, int64_t n_windows, int64_t L,
                  int c_in, int c_out, int dtype, void* z, float* stat_sum, float* stat_sq, void* e, void* stream);
/* Training forward WITHOUT the BatchNorm / pool pass between two blocks (dropout rate 0).  The max-pool of a BatchNorm output picks
 * the pool-window extreme e of the conv output (vm_conv_fwd_e, vm_conv1_fused_fwd), and BatchNorm is a per-channel affine
 * y = scale[c] * e + shift[c] -- so the next Conv1D can read e itself with the affine folded into its weights:
 *     conv(y)[t][co] = sum_k sum_ci (W[k][ci][co] * scale[ci]) * e[t + k - 1][ci]  +  bias[co] + sum_{k inside} hb[k][co],
 *     hb[k][co] = sum_ci W[k][ci][co] * shift[ci];   tap 0 is outside the window at t = 0, tap 2 at t = L - 1 (SAME pads y with 0).
 * BatchNorm statistics are per encoder call ("tower": windows [t * windows_per_tower, (t + 1) * windows_per_tower)), so there is one
 * set of folded weights per tower.  vm_fold_bn_weights: wt = the fp32 kernel in wf's layout (c_out, 3 * c_in) (the `wt` output of
 * vm_prep_conv_weights_batch) + scale / shift (towers, c_in) + the layer's bias (c_out) -> wf_folded (towers, c_out, 3 * c_in) `dtype`
 * and / or wf_packed (the same values in vm_pack_nt_weights' fragment order; either may be NULL) and hb (towers, 4, c_out) fp32: rows
 * 0..2 the per-tap constants, row 3 = bias + their sum (what the accumulators of vm_conv_fwd_fold start from).  vm_conv_fwd_fold: hb
 * as written by vm_fold_bn_weights (its `bias` argument is not read: row 3 of hb carries it); wf_packed (optional): the fragment-
 * order copy, used where the shape has a conv_nt3_kernel (128 / 256 / 384 / 512 input channels);  in_e = padded extreme (n_windows, L + 2, c_in) of the layer below (zero halo
 * rows), z / stat_* as vm_conv_fwd; e (optional) = this layer's own extreme for MaxPool1D(2), PADDED (n_windows, L/2 + 2, c_out), the
 * maximum where gamma >= 0 else the minimum.  16-bit storage, conv_nt2r_kernel shapes only (vm_conv_fwd_fold_supported).  The
 * pooled BatchNorm output is never materialised: -1 read and -1 write of it per block and no pass over z in the forward.
 * o (optional, with e): the OTHER element of every position pair, unpadded (n_windows, L/2, c_out), with its sign bit set where the
 * extreme is the pair's second element (z >= 0 after the ReLU: the bit is free; ties: the first element is the extreme).  (e, o)
 * together are z, so with o given z is NOT written and may be NULL -- the epilogue stores as many bytes as a plain forward --
 * and the backward takes the pair form (vm_bn_pool_bwd_apply_pairs).
 * vm_conv_wgrad_fold is the matching weight gradient, vm_conv_dgrad[_bnred] is unchanged (it takes the un-folded wd). */
int vm_fold_bn_weights(const float* wt, const float* scale, const float* shift, const float* bias, int towers, int c_in, int c_out,
                       int dtype, void* wf_folded, void* wf_packed, float* hb, float* ctr_out, void* stream);
/* ctr_out (towers, c_out) or NULL (round 5): the centre of each output channel's tile for vm_conv_fwd_fold's `e_center` --
 * ctr = max(row 3 of hb, 0) rounded to the storage type (the pedestal the accumulators start from: conv bias + the contribution of
 * the input's BatchNorm shifts); row 3 of hb is then written with ctr already taken off. */
int vm_conv_fwd_fold_supported(int64_t n_windows, int64_t L, int c_in, int c_out, int dtype, int with_e);
int vm_conv_fwd_fold(const void* in_e, const void* wf_folded, const float* bias, const float* hb, const float* gamma,
                     int64_t n_windows, int64_t windows_per_tower, int64_t L, int c_in, int c_out, int dtype, void* z, float* stat_sum,
                     float* stat_sq, void* e, void* o, const void* wf_packed, const float* e_center, void* stream);
/* e_center (towers, c_out) or NULL (round 5; VM_F16 with the (e, o) output only): vm_fold_bn_weights' ctr_out.  The tile is computed
 * and rounded CENTRED: t = relu(z) - ctr, one rounding of a value whose size is the distance from the channel's pedestal instead of
 * the pedestal (a half spends its 11 bits on the latter otherwise: 1.06e-3 -> 0.66e-3 on the embeddings of the trained-like state of
 * tests/test_gpu_fullsize_oracle.py).  e receives the centred extreme, o the other element UN-centred (so that its sign bit stays
 * free for the position flag), stat_sum / stat_sq the sums of t and t^2: vm_bn_finalize's tile_center, vm_bn_pool_bwd_apply_pairs'
 * e_center and the *_adj constants take it from there. */
/* The k = 3 GEMM weights in the order the matrix cores consume them (round 4).  bt: `towers` matrices (n_rows, 3 * a_c) `dtype` back
 * to back -- wf (n_rows = c_out, a_c = c_in), wd (n_rows = c_in, a_c = c_out) of vm_prep_conv_weights, wf_folded of
 * vm_fold_bn_weights; packed: the same elements as [tower][n_rows / 64][K tile = (channel chunk of 32, tap)][32-row half][16-channel
 * half][64 lanes][8 values], i.e. every 1 KB piece is one v_mfma_f32_32x32x16 operand fragment of a wave and a wave's stream through a
 * K loop is contiguous.  The entry points that take a `*_packed` argument (vm_conv_fwd_fold, vm_conv_fwd_pool, vm_conv_dgrad_bnred)
 * then load the weights from L2 straight into registers (conv_nt3_kernel) instead of staging them in LDS; NULL there = the staged
 * kernel.  Same results bit for bit (the same products in the same order).  16-bit storage, n_rows % 128 == 0, a_c % 32 == 0. */
int vm_pack_nt_weights_supported(int n_rows, int a_c, int dtype);
int vm_pack_nt_weights(const void* bt, int towers, int n_rows, int a_c, int dtype, void* packed, void* stream);
/* the same for n (<= 8) matrices in ONE launch: host arrays, one entry per matrix. */
int vm_pack_nt_weights_batch(int n, const void* const* bt, const int* towers, const int* n_rows, const int* a_c, int dtype,
                             void* const* packed, void* stream);
/* inference-mode forward of a whole block in one launch: Conv1D + bias + ReLU, the BatchNorm affine (scale / shift per channel from
 * vm_bn_infer_affine: (c_out) floats each) and MaxPool1D(2), models.py:22-35 with learning_phase 0.  act: padded pooled output
 * (n_windows, L/2 + 2, c_out), halo rows untouched; the conv output z is never written.  Bit-identical to vm_conv_fwd followed by
 * vm_bn_drop_pool_fwd(pool = 2, drop = NULL).  Served by the 256 x 128 input-resident kernel only (bf16, even L):
 * vm_conv_fwd_pool_supported() says whether a shape is, VM_ERR_UNSUPPORTED otherwise. */
int vm_conv_fwd_pool_supported(int64_t n_windows, int64_t L, int c_in, int c_out, int dtype);
int vm_conv_fwd_pool(const void* in, const void* wf, const float* bias, const float* scale, const float* shift,
                     int64_t n_windows, int64_t L, int c_in, int c_out, int dtype, void* act, const void* wf_packed, void* stream);
/* dgrad: dx[n][t][ci] = sum_{k,co} du[n][t+1-k][co] * W[k][ci][co].  du padded (n_windows, L+2, c_out);
 * wd: (c_in, 3*c_out) `dtype` tap-flipped copy from vm_prep_conv_weights; dx: (n_windows, L, c_in). */
int vm_conv_dgrad(const void* du, const void* wd, int64_t n_windows, int64_t L, int c_in, int c_out, int dtype,
                  void* dx, void* stream);
/* vm_conv_dgrad that also forms, from the output tile it holds on chip, the two sums the BatchNorm backward of the layer BELOW
 * needs (that layer's output gradient is exactly this dx; keras BatchNormalization backward, voicemap/models.py:23,28,33):
 *   red_s0[row][ci] = sum_t dx[n][t][ci],   red_s1[row][ci] = sum_t dx[n][t][ci] * red_a[n][t][ci]
 * as vm_conv_dgrad_bnred_rows(L) partial rows per window (fp32, (n_windows * rows, c_in), every row written).  red_a: a
 * (n_windows, L, c_in) tensor of `dtype` -- the pooled output of the layer below (red_a_padded = 1: stored (L + 2) rows with a
 * halo row either side, as vm_bn_drop_pool_fwd writes it) or its pool-window extreme (red_a_padded = 0).  dx is written as by
 * vm_conv_dgrad (the sums use dx as rounded to `dtype`).  Served by the 256 x 128 input-resident kernel only:
 * vm_conv_dgrad_bnred_supported() says whether the shape / dtype / current tuning is, VM_ERR_UNSUPPORTED otherwise.
 * vm_bn_bwd_from_sums turns the rows into the partials vm_bn_bwd_finalize takes, replacing vm_bn_pool_bwd_reduce[_pooled]. */
int64_t vm_conv_dgrad_bnred_rows(int64_t L);
int vm_conv_dgrad_bnred_supported(int64_t n_windows, int64_t L, int c_in, int c_out, int dtype);
int vm_conv_dgrad_bnred(const void* du, const void* wd, int64_t n_windows, int64_t L, int c_in, int c_out, int dtype,
                        void* dx, const void* red_a, int red_a_padded, float* red_s0, float* red_s1, const void* wd_packed,
                        void* stream);
/* wgrad: dW[k][ci][co] = sum_{n,t} in[n][t+k][ci] * du[n][t+1][co] (both padded).  Split over windows into
 * vm_conv_wgrad_splits() slabs in ws (fp32), then summed in fixed order into grad_w (3, c_in, c_out). */
int vm_conv_wgrad_splits(int64_t n_windows, int64_t L, int c_in, int c_out);
int64_t vm_conv_wgrad_workspace_bytes(int64_t n_windows, int64_t L, int c_in, int c_out);
int vm_conv_wgrad(const void* in, const void* du, int64_t n_windows, int64_t L, int c_in, int c_out, int dtype,
                  void* ws, float* grad_w, void* stream);
/* wgrad of a layer that ran vm_conv_fwd_fold: the layer's true input is y = scale_t[ci] * e + shift_t[ci] inside the window (t = the
 * tower of the window: n_windows / windows_per_tower towers, scale / shift (towers, c_in)), 0 in the padding, hence
 *     dW[k][ci][co] = sum_t scale_t[ci] * (sum_{n in t, pos} e[n][pos+k][ci] * du[n][pos+1][co]) + shift_t[ci] * dsum[t][k][co]
 * -- the same split GEMM on e with every slab inside one tower, then one fixed-order pass that sums the slabs per tower and applies
 * the two factors.  dsum (towers, 3, c_out) from vm_du_tower_sums. */
int64_t vm_conv_wgrad_fold_workspace_bytes(int64_t n_windows, int64_t windows_per_tower, int64_t L, int c_in, int c_out);
int vm_conv_wgrad_fold(const void* in_e, const void* du, int64_t n_windows, int64_t windows_per_tower, int64_t L, int c_in, int c_out,
                       int dtype, const float* scale, const float* shift, const float* dsum, void* ws, float* grad_w, void* stream);
/* dsum == NULL above leaves the slabs in ws (the GEMM needs nothing but e and du, so it can start before vm_du_tower_sums has run);
 * this is the second half: the per-tower slab sums with the two factors applied -> grad_w. */
int vm_conv_wgrad_fold_finish(const void* ws, int64_t n_windows, int64_t windows_per_tower, int64_t L, int c_in, int c_out,
                              const float* scale, const float* shift, const float* dsum, float* grad_w, void* stream);
/* fp32 Keras kernel (3, c_in, c_out) -> wf (c_out, 3*c_in) and wd (c_in, 3*c_out) in `dtype`. */
int vm_prep_conv_weights(const float* w, int c_in, int c_out, int dtype, void* wf, void* wd, void* stream);
/* the same for n_layers (<= 8) layers in ONE launch: host arrays of device pointers / channel counts, one entry per layer.  wt
 * (optional array, entries may be NULL): also the fp32 kernel itself in wf's layout (c_out, 3*c_in) -- what vm_fold_bn_weights reads. */
int vm_prep_conv_weights_batch(int n_layers, const float* const* w, const int* c_in, const int* c_out, int dtype,
                               void* const* wf, void* const* wd, float* const* wt, void* stream);

/* ---- a1-BN: BatchNormalization()  (voicemap/models.py:17,23,28,33; Keras defaults eps 1e-3, momentum .99)
 * Reduces the conv partials per tower in a fixed order (fp64), producing per-tower
 *   mean, invstd = rsqrt(var_biased + eps), scale = gamma*invstd, shift = beta - mean*scale     (each (n_towers, C))
 * and applies the moving-average updates tower by tower:  moving -= (moving - batch) * (1 - momentum), with the
 * batch variance multiplied by n/(n-(1+eps)) first when unbiased_moving_var != 0 (Keras 2.2.x). */
/* ws (all three reducers below): >= vm_colreduce_workspace_bytes(n_segments, C) bytes of scratch for the first of the
 * two deterministic reduction stages (n_segments = n_towers; 1 for vm_colsum).  * zd_biased (n_towers, 2, C) fp32 or NULL: Keras 2.2.2's moving_average_update is TF 1.10's assign_moving_average with
 * zero_debias=True -- per encoder call a zero-initialised biased accumulator b -= (b - value)(1 - momentum) and
 * moving = b * zd_correction with zd_correction = 1 / (1 - momentum^t), t = number of training steps so far (host side); the
 * towers are applied in order.  NULL = the plain exponential average moving -= (moving - value)(1 - momentum). */
int64_t vm_colreduce_workspace_bytes(int n_segments, int C);
int vm_bn_finalize(const float* stat_sum, const float* stat_sq, int64_t rows_per_tower, int n_towers, int C,
                   double count_per_tower, const float* gamma, const float* beta, float eps, float momentum,
                   int unbiased_moving_var, float* moving_mean, float* moving_var, float* mean, float* invstd,
                   float* scale, float* shift, void* ws, float* zd_biased, float zd_correction, const float* center_bias,
                   float* shift_adj, float* mean_adj, const float* tile_center, void* stream);
/* tile_center (n_towers, C) or NULL, exclusive with center_bias (round 5): the layer ran as vm_conv_fwd_fold with `e_center` -- its
 * statistics partials are sums over t = z - ctr and its pool extreme is stored as e - ctr, ctr = tile_center[tower][c].  The sums are
 * taken back to z here (sum z = sum t + n ctr, sum z^2 = sum t^2 + 2 ctr sum t + n ctr^2, fp64) and shift_adj / mean_adj carry the
 * offset for the consumers of the stored extreme exactly as with center_bias. */
/* center_bias (C) or NULL, with shift_adj / mean_adj (n_towers, C): block 1's pool extreme is stored CENTRED in the folded training
 * path (vm_conv1_fused_fwd mode 2 writes e - ctr, ctr = max(conv bias, 0): the whitened waveform makes conv-1 outputs small next to a
 * bias, and a 16-bit value would spend its significand on that pedestal).  Its consumers are linear in e, so the offset moves into
 * their constants: shift_adj = shift + scale * ctr is the shift over the STORED value (vm_fold_bn_weights / vm_conv_wgrad_fold_finish of
 * block 2), mean_adj = mean - ctr the mean against which the sums over the stored value are taken (vm_bn_bwd_from_sums[_finalize]). */
/* inference affine from moving statistics (one "tower"). */
int vm_bn_infer_affine(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                       float eps, int C, float* scale, float* shift, void* stream);

/* BN apply + SpatialDropout1D + MaxPool1D(pool)  (voicemap/models.py:17-19 etc.)
 * y = (z*scale[tower] + shift[tower]) * drop[n][c];  out[n][1+q][c] = max_{j<pool} y[n][q*pool+j][c], q < L/pool.
 * z: (n_windows, L, C); drop: (n_windows, C) fp32 keep-mask/(1-rate) or NULL; out: padded (n_windows, L/pool + 2, C). */
int vm_bn_drop_pool_fwd(const void* z, const float* scale, const float* shift, const float* drop,
                        int64_t n_windows, int64_t windows_per_tower, int64_t L, int C, int pool, int dtype,
                        void* out, void* stream);
/* backward, pass 1: partials of sum(dy) and sum(dy*zhat) over the pooled positions, where dy is dp routed to
 * the first maximum of each pool window.  dp: (n_windows, L/pool, C).
 * part_*: (n_windows * vm_bn_part_rows(), C) fp32: every pass kernel fills all vm_bn_part_rows() rows of a window -- one per
 * workgroup of the window for long windows; for short ones (pooled rows x channel vectors < 2 048: the 2-D variant) one workgroup
 * takes the whole window, writes row 0 and zero-fills the others -- so a consumer simply adds all rows. */
int vm_bn_part_rows(void);
int vm_bn_pool_bwd_reduce(const void* z, const void* dp, const float* scale, const float* shift, const float* mean,
                          const float* invstd, const float* drop, int64_t n_windows, int64_t windows_per_tower,
                          int64_t L, int C, int pool, int dtype, float* part_dy, float* part_dyz, void* stream);
/* reduce those partials per tower: c1 = sum(dy)/count, c2 = sum(dy*zhat)/count (each (n_towers, C)); and add the
 * parameter gradients over all towers: grad_gamma = sum(dy*zhat), grad_beta = sum(dy)  (overwritten). */
/* vm_bn_bwd_from_sums followed by vm_bn_bwd_finalize in two launches instead of three and without the (n_windows * part_rows, C) partial
 * tensors in between: the per-window map is applied to the dgrad epilogue's partial rows on their way into the column sums (fp64).
 * Same arguments as the two calls; ws >= vm_colreduce_workspace_bytes(n_towers, C). */
int vm_bn_bwd_from_sums_finalize(const float* s0, const float* sa, int64_t rows_per_window, const void* z, const void* dp,
                                 const float* scale, const float* shift, const float* mean, const float* invstd, const float* drop,
                                 int64_t n_windows, int64_t windows_per_tower, int64_t L, int C, int pool, int dtype, int a_is_act,
                                 double count_per_tower, float* c1, float* c2, float* grad_gamma, float* grad_beta, void* ws,
                                 void* stream);
int vm_bn_bwd_finalize(const float* part_dy, const float* part_dyz, int64_t n_windows, int64_t windows_per_tower,
                       int C, double count_per_tower, float* c1, float* c2, float* grad_gamma, float* grad_beta,
                       void* ws, void* stream);
/* vm_bn_pool_bwd_reduce with the pool-window extreme of z recovered from the POOLED forward output `act` (padded
 * (n_windows, L/pool + 2, C), as written by vm_bn_drop_pool_fwd with the same scale / shift / drop) instead of re-derived from
 * z: ext = act / (scale*drop) - shift/scale.  Reads two pooled-size tensors instead of z + dp; differs from the z form by the
 * storage rounding of act (bit-identical sums are NOT guaranteed; channels with scale == 0 use z).  Throughput-mode option. */
int vm_bn_pool_bwd_reduce_pooled(const void* z, const void* act, const void* dp, const float* scale, const float* shift,
                                 const float* mean, const float* invstd, const float* drop, int64_t n_windows,
                                 int64_t windows_per_tower, int64_t L, int C, int pool, int dtype, float* part_dy,
                                 float* part_dyz, void* stream);
/* the same partials from the sums vm_conv_dgrad_bnred left behind (s0, sa: (n_windows * rows_per_window, C)).  a_is_act = 1:
 * sa was taken against the pooled output, the extreme is recovered as in vm_bn_pool_bwd_reduce_pooled (z is read only for
 * channels with scale == 0); a_is_act = 0: sa was taken against the extreme itself (z may be NULL).  L, pool: this block's
 * pre-pool length and pool size (dp is (n_windows, L / pool, C)). */
int vm_bn_bwd_from_sums(const float* s0, const float* sa, int64_t rows_per_window, const void* z, const void* dp,
                        const float* scale, const float* shift, const float* mean, const float* invstd, const float* drop,
                        int64_t n_windows, int64_t windows_per_tower, int64_t L, int C, int pool, int dtype, int a_is_act,
                        float* part_dy, float* part_dyz, void* stream);
/* backward, pass 2: du[n][1+t][c] = [z>0] * scale * (dy - c1 - zhat*c2)  (padded (n_windows, L+2, C) out), plus
 * partial column sums of du -> part_du (n_windows * vm_bn_part_rows(), C) for the conv bias gradient. */
int vm_bn_pool_bwd_apply(const void* z, const void* dp, const float* scale, const float* shift, const float* mean,
                         const float* invstd, const float* drop, const float* c1, const float* c2,
                         int64_t n_windows, int64_t windows_per_tower, int64_t L, int C, int pool, int dtype,
                         void* du, float* part_du, void* stream);
/* The same two passes for the LAST block, with dp given in the sparse form GlobalMaxPool1D's backward produces:
 * dp[n][q][c] = dg[n][c] if q == gidx[n][c] else 0 (dg, gidx as in vm_global_maxpool_fwd/bwd).  Avoids writing and
 * re-reading the dense (n_windows, L/pool, C) tensor (voicemap/models.py:35-37). */
int vm_bn_pool_bwd_reduce_gmax(const void* z, const float* dg, const int32_t* gidx, const float* scale, const float* shift,
                               const float* mean, const float* invstd, const float* drop, int64_t n_windows,
                               int64_t windows_per_tower, int64_t L, int C, int pool, int dtype, float* part_dy,
                               float* part_dyz, void* stream);
/* vm_bn_pool_bwd_reduce_gmax + vm_bn_bwd_finalize in one launch (round 6; the same sums in fp64, per (tower, channel) by one 32-lane
 * group): c1 / c2 (towers, C), grad_gamma / grad_beta (C) as vm_bn_bwd_finalize writes them (voicemap/models.py:32-37 backward). */
int vm_bn_bwd_gmax_finalize(const void* z, const float* dg, const int32_t* gidx, const float* scale, const float* shift,
                            const float* mean, const float* invstd, const float* drop, int64_t n_windows,
                            int64_t windows_per_tower, int64_t L, int C, int pool, int dtype, double count_per_tower, float* c1,
                            float* c2, float* grad_gamma, float* grad_beta, void* stream);
int vm_bn_pool_bwd_apply_gmax(const void* z, const float* dg, const int32_t* gidx, const float* scale, const float* shift,
                              const float* mean, const float* invstd, const float* drop, const float* c1, const float* c2,
                              int64_t n_windows, int64_t windows_per_tower, int64_t L, int C, int pool, int dtype, void* du,
                              float* part_du, void* stream);
/* vm_bn_pool_bwd_apply for pool = 2 with z given as the pair tensors of vm_conv_fwd_fold: e PADDED (n_windows, L/2 + 2, C), o
 * (n_windows, L/2, C) with the position flag in its sign bit.  Same arithmetic and outputs.  16-bit storage, L even. */
int vm_bn_pool_bwd_apply_pairs(const void* e, const void* o, const void* dp, const float* scale, const float* shift, const float* mean,
                               const float* invstd, const float* drop, const float* c1, const float* c2, int64_t n_windows,
                               int64_t windows_per_tower, int64_t L, int C, int dtype, void* du, float* part_du, const float* e_center,
                               void* stream);
/* e_center (towers, C) or NULL: e was stored centred (vm_conv_fwd_fold e_center); the kernel adds ctr back (fp32) before it applies
 * ReLU's mask and the BatchNorm-backward constants, which stay those of z (mean, not mean_adj). */
/* out[c] = sum_r part[r][c] in fixed order (bias gradients). */
int vm_colsum(const float* part, int64_t rows, int C, float* out, void* ws, void* stream);
/* The part_* tensors have vm_bn_part_rows() rows per window; a pass over short windows (the 2-D variant) fills only the first
 * vm_bn_part_rows_used(L, C, pool, dtype) of them and zeroes the rest.  vm_colsum_strided sums rows 0, row_step, 2 row_step, ...
 * (`rows` of them) -- with row_step = vm_bn_part_rows() the one live row per window instead of eight. */
int vm_bn_part_rows_used(int64_t L, int C, int pool, int dtype);
int vm_colsum_strided(const float* part, int64_t rows, int row_step, int C, float* out, void* ws, void* stream);
/* vm_colsum of the apply pass's part_du (n_windows * vm_bn_part_rows(), C) per tower, plus what vm_conv_wgrad_fold needs:
 * dsum[t][k][c] = sum over the windows of tower t and the positions whose tap k lies inside the window of du[n][pos][c] (k = 1: all
 * positions; k = 0: all but position 0; k = 2: all but position L - 1; du: padded (n_windows, L + 2, C) `dtype`).  grad_b (optional,
 * C) = the sum over all towers (the conv bias gradient).  ws: vm_colreduce_workspace_bytes(towers, C). */
int vm_du_tower_sums(const float* part_du, const void* du, int64_t n_windows, int64_t windows_per_tower, int64_t L, int C, int dtype,
                     float* grad_b, float* dsum, void* ws, void* stream);

/* LAST block only: vm_bn_drop_pool_fwd fused with GlobalMaxPool1D (voicemap/models.py:31-37).  The pooled tensor of the
 * last block is never materialised: gmax / gidx are exactly what vm_global_maxpool_fwd would return for it (values rounded
 * to `dtype`, gidx = first maximum).  ws: vm_bn_drop_pool_gmax_workspace_bytes(n_windows, C) bytes. */
int64_t vm_bn_drop_pool_gmax_workspace_bytes(int64_t n_windows, int C);
int vm_bn_drop_pool_gmax_fwd(const void* z, const float* scale, const float* shift, const float* drop, int64_t n_windows,
                             int64_t windows_per_tower, int64_t L, int C, int pool, int dtype, float* gmax, int32_t* gidx,
                             void* ws, void* stream);
/* The first launch of vm_bn_drop_pool_gmax_fwd alone: part_v / part_i receive the vm_bn_part_rows() partial (value, position) rows
 * of every window ((n_windows * rows, C) each; two arrays so that the two towers' launches can fill the halves of one pair of
 * arrays); vm_tail_fwd_bwd (below) finishes them inside its own launch. */
int vm_bn_drop_pool_gmax_partials(const void* z, const float* scale, const float* shift, const float* drop, int64_t n_windows,
                                  int64_t windows_per_tower, int64_t L, int C, int pool, int dtype, float* part_v, int32_t* part_i,
                                  void* stream);
/* The LAST block in pair form (round 6; 16-bit storage): vm_conv_fwd_fold leaves (e, o) for it like for the blocks below -- e PADDED
 * (n_windows, Lq + 2, C), Lq = L / 2 -- and the GlobalMaxPool1D pass reads e alone: BatchNorm is monotone per channel, so
 * max_j fma(z_j, s, h) == fma(ext_j z, s, h), the values and first-maximum positions are those of vm_bn_drop_pool_gmax_partials on z,
 * from half the bytes.  voicemap/models.py:31-37.  The backward's sparse sums (vm_bn_bwd_gmax_finalize_e: vm_bn_bwd_gmax_finalize with
 * e in z's place) and apply pass (vm_bn_pool_bwd_apply_pairs_gmax: vm_bn_pool_bwd_apply_gmax on the (e, o) pair form; du padded
 * (n_windows, L + 2, C) as there) take the same tensors. */
int vm_bn_drop_pool_gmax_partials_e(const void* e, const float* scale, const float* shift, const float* drop, int64_t n_windows,
                                    int64_t windows_per_tower, int64_t Lq, int C, int dtype, float* part_v, int32_t* part_i,
                                    void* stream);
int vm_bn_bwd_gmax_finalize_e(const void* e, const float* dg, const int32_t* gidx, const float* scale, const float* shift,
                              const float* mean, const float* invstd, const float* drop, int64_t n_windows, int64_t windows_per_tower,
                              int64_t Lq, int C, int dtype, double count_per_tower, float* c1, float* c2, float* grad_gamma,
                              float* grad_beta, void* stream);
int vm_bn_pool_bwd_apply_pairs_gmax(const void* e, const void* o, const float* dg, const int32_t* gidx, const float* scale,
                                    const float* shift, const float* mean, const float* invstd, const float* drop, const float* c1,
                                    const float* c2, int64_t n_windows, int64_t windows_per_tower, int64_t L, int C, int dtype, void* du,
                                    float* part_du, void* stream);

/* ---- a1 tail: GlobalMaxPool1D + Dense(E)  (voicemap/models.py:37-39) ------------------------------------
 * act: padded (n_windows, L+2, C) `dtype`; gmax (n_windows, C) fp32; gidx (n_windows, C) int32 = first argmax. */
int vm_global_maxpool_fwd(const void* act, int64_t n_windows, int64_t L, int C, int dtype, float* gmax,
                          int32_t* gidx, void* stream);
/* dp: (n_windows, L, C) `dtype`, zero except dp[n][gidx[n][c]][c] = dg[n][c]. */
int vm_global_maxpool_bwd(const float* dg, const int32_t* gidx, int64_t n_windows, int64_t L, int C, int dtype,
                          void* dp, void* stream);
/* out[r][o] = sum_i in[r][i]*w[i][o] + b[o]   (w: (n_in, n_out) Keras Dense kernel; all fp32). */
int vm_dense_fwd(const float* in, const float* w, const float* b, int64_t rows, int n_in, int n_out, float* out,
                 void* stream);
/* grad_w[i][o] = sum_r in[r][i]*dout[r][o]; grad_b[o] = sum_r dout[r][o]; din[r][i] = sum_o dout[r][o]*w[i][o]
 * (din may be NULL; grad_w and grad_b may both be NULL: then only din -- the two halves are independent launches and a caller may
 * put the parameter half on another stream).  Fixed summation order. */
int vm_dense_bwd(const float* in, const float* w, const float* dout, int64_t rows, int n_in, int n_out,
                 float* grad_w, float* grad_b, float* din, void* stream);

/* ---- a2 + a3/a4: siamese head + loss, forward and backward in one launch ----------------------------------
 * (voicemap/models.py:55-69; voicemap/utils.py:77-85; 'binary_crossentropy' experiments/train_siamese.py:57)
 * emb: (2*pairs, E) fp32, rows [0,pairs) = tower 0, [pairs, 2*pairs) = tower 1.
 * head_kind UNIFORM_EUCLIDEAN: d = sqrt(sum (e1-e2)^2), p = sigmoid(w[0]*d + b);  WEIGHTED_L1: p = sigmoid(sum w[j]|e1-e2|_j + b).
 * y: (pairs) fp32 labels, 0 = same speaker (voicemap/librispeech.py:194).  y may be NULL for predict-only
 * (then loss/backward outputs are not touched).
 * pred (pairs); loss_acc[0] = loss, [1] = binary accuracy; demb (2*pairs, E); grad_hw (1 or E); grad_hb (1).
 * ws: 4*pairs floats of scratch (per-pair terms, summed in fixed order by a second small launch); unused when y is NULL.
 * grad_scale: every gradient output (demb, grad_hw, grad_hb) is multiplied by it -- 1 for fp32 / bf16 storage; the loss scale of
 * VM_F16 storage, whose activation gradients would otherwise fall under half's 6e-8; loss_acc and pred are not scaled.
 * loss_acc may be NULL with y given: then only the per-pair pass runs (pred, demb, ws) and the caller enqueues vm_siamese_head_reduce
 * (the fixed-order sums: loss_acc, grad_hw, grad_hb -- nobody's input before the optimizer) where it likes, e.g. on another stream. */
int vm_siamese_head_loss(const float* emb, const float* head_w, const float* head_b, const float* y, int64_t pairs,
                         int E, int head_kind, int loss_kind, float grad_scale, float* pred, float* loss_acc, float* demb,
                         float* grad_hw, float* grad_hb, float* ws, void* stream);
int vm_siamese_head_reduce(const float* emb, const float* ws, int64_t pairs, int E, int head_kind, float* loss_acc, float* grad_hw,
                           float* grad_hb, void* stream);

/* ---- the whole tail of a siamese training step, forward and backward (SURVEY 8(b) `vm_tail_fwd_bwd`) -------------------------
 * GlobalMaxPool1D -> Dense(E) (voicemap/models.py:37-39) -> twin distance -> Dense(1, sigmoid) (models.py:55-69) -> loss
 * (voicemap/utils.py:77-85 / 'binary_crossentropy' experiments/train_siamese.py:57) -> d loss / d emb -> d loss / d gmax.
 * All of it is local to a pair: ONE launch, one workgroup per pair (windows b and pairs + b), replaces gmax_segments + vm_dense_fwd +
 * the per-pair pass of vm_siamese_head_loss + the input half of vm_dense_bwd.  What crosses pairs and is nobody's input before the
 * optimizer -- loss_acc, the head's and the dense layer's parameter gradients -- is vm_tail_param_grads, a second launch the
 * caller may put on another stream (it replaces vm_siamese_head_reduce + the parameter half of vm_dense_bwd).  Bit-identical to the
 * six launches they replace (same summation orders).
 *   gmax_part_v / gmax_part_i: the partial rows of vm_bn_drop_pool_gmax_partials ((2*pairs * seg_rows, C), seg_rows =
 *               vm_bn_part_rows()), or both NULL when gmax / gidx (2*pairs, C) are already final (vm_global_maxpool_fwd); with
 *               parts, gmax and gidx are WRITTEN here.
 *   dense_w (C, E), dense_b (E) or NULL; head_w / head_b / y / head_kind / loss_kind / grad_scale as in vm_siamese_head_loss.
 *   outputs: emb (2*pairs, E), pred (pairs), demb (2*pairs, E), dgmax (2*pairs, C), ws (4*pairs: the per-pair terms).
 * vm_tail_fwd_bwd_supported: C <= 1024 and E <= 256 (the pair's rows live in LDS). */
int vm_tail_fwd_bwd_supported(int C, int E);
int vm_tail_fwd_bwd(const float* gmax_part_v, const int32_t* gmax_part_i, int seg_rows, float* gmax, int32_t* gidx, const float* dense_w, const float* dense_b,
                    const float* head_w, const float* head_b, const float* y, int64_t pairs, int C, int E, int head_kind,
                    int loss_kind, float grad_scale, float* emb, float* pred, float* demb, float* dgmax, float* ws, void* stream);
int vm_tail_param_grads(const float* gmax, const float* demb, const float* emb, const float* ws, int64_t pairs, int C, int E,
                        int head_kind, float* loss_acc, float* grad_dense_w, float* grad_dense_b, float* grad_hw, float* grad_hb,
                        void* stream);

/* ---- a9: classifier head Dense(num_classes, softmax) + categorical CE  (experiments/train_classifier.py:112,115)
 * logits (rows, n_classes) fp32 -> prob; labels int32 (rows); loss_acc[0] = mean CE (Keras clip 1e-7), [1] = accuracy;
 * dlogits = grad_scale * d loss / d logits (may be NULL).  labels may be NULL for predict-only.  ws: 2*rows floats. */
int vm_softmax_cce(const float* logits, const int32_t* labels, int64_t rows, int n_classes, float grad_scale, float* prob,
                   float* loss_acc, float* dlogits, float* ws, void* stream);

/* ---- a5: Adam(clipnorm=1.)  (experiments/train_siamese.py:56; Keras 2.2.2 optimizers.py) -------------------
 * sqnorm: 1 fp32 on device <- sum(g^2) over the flat gradient buffer (fixed order; ws >= vm_sqnorm_workspace_bytes).  sqnorm may be
 * NULL: then only the partial sums are left in ws, for vm_adam_clip_step's sqnorm_parts (one launch less). */
int64_t vm_sqnorm_workspace_bytes(int64_t n);
int vm_grad_sqnorm(const float* g, int64_t n, void* ws, float* sqnorm, void* stream);
/* g *= grad_prescale (e.g. 1/world after an all-reduce sum); norm = grad_prescale*sqrt(*sqnorm);
 * if clipnorm > 0 and norm >= clipnorm: g *= clipnorm/norm;  m,v,p updated with lr_t = lr*sqrt(1-b2^t)/(1-b1^t)
 * (computed on the host, passed in), p -= lr_t*m/(sqrt(v)+eps).
 * skip_nonfinite != 0 (needs sqnorm): a step whose gradient norm is inf / NaN leaves p, m, v untouched (loss-scaled VM_F16
 * training: an overflowed activation gradient costs one step instead of the model) and increments the caller's running count *skipped
 * (1 int32 on the device, zeroed by the caller, may be NULL); with 0 the update is Keras' own arithmetic, NaNs included.
 * sqnorm_parts (optional): the ws vm_grad_sqnorm(…, sqnorm = NULL) filled -- every workgroup then adds the partials itself (same
 * order, same value) and *sqnorm (if not NULL) receives the sum; without it *sqnorm is read. */
int vm_adam_clip_step(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
                      float eps, float clipnorm, float grad_prescale, float* sqnorm, const void* sqnorm_parts, int skip_nonfinite,
                      int32_t* skipped, void* stream);

/* ---- a8: n-shot evaluation distances  (voicemap/utils.py:159-206) ------------------------------------------
 * For each task: query embedding (E) vs k class prototypes built from n support embeddings each
 * (support: (tasks, k*n, E), query: (tasks, E), fp32); writes pred (tasks, k) as float64-accumulated distances
 * and argmin (tasks) int32.  dist_kind per VM_DIST_*. */
int vm_nshot_distances(const float* query, const float* support, int64_t tasks, int k, int n, int E, int dist_kind,
                       float* pred, int32_t* argmin, void* stream);

/* ---- f2 / BASELINE.json config 5: evaluation over a CACHED embedding matrix -----------------------------------------
 * The reference embeds the k*n + 1 windows of every task anew (voicemap/utils.py:121-212; the sweep of
 * experiments/k_way_accuracy.py:52-69 does so 38 000 times).  Its evaluation datasets are built with stochastic=False
 * (experiments/train_siamese.py:44), i.e. every file has ONE window; embed the corpus once into emb (n_rows, E) fp32 and a task
 * is k*n + 1 row indices.
 * vm_nshot_indexed: vm_nshot_distances with the rows gathered by index: query_idx (tasks), support_idx (tasks, k*n) laid out
 * [class_1]*n + ... + [class_k]*n like voicemap/librispeech.py:224-237; float64 arithmetic of voicemap/utils.py:159-206;
 * pred (tasks, k) may be NULL; argmin (tasks) int32, first minimum, a NaN distance first like numpy.argmin.  E <= 256. */
int vm_nshot_indexed(const float* emb, int64_t n_rows, const int32_t* query_idx, const int32_t* support_idx, int64_t tasks, int k,
                     int n, int E, int dist_kind, float* pred, int32_t* argmin, void* stream);
/* vm_pairdist_argmin: the pairwise-distance matrix between q (M, E) and ref (N, E), fp32, and per query row the nearest reference
 * row (first minimum).  dist_kind: EUCLIDEAN sqrt(sum (a-b)^2) (direct form, no norm-expansion cancellation), COSINE
 * 1 - a.b / (|a| |b|), DOT -a.b.  q_row0 >= 0: query row m IS reference row q_row0 + m (q is a row shard of ref -- the
 * data-parallel form: all-gather the embeddings, every rank takes its rows) and is excluded from its own argmin; -1: no exclusion.
 * dist (M, N) may be NULL (argmin only: nothing but best_val / best_idx (M) leaves the chip; best_idx = -1 if every distance of the
 * row is NaN).  ws >= vm_pairdist_workspace_bytes(M, N).  E <= 256, N < 2^31. */
int64_t vm_pairdist_workspace_bytes(int64_t M, int64_t N);
int vm_pairdist_argmin(const float* q, const float* ref, int64_t M, int64_t N, int E, int dist_kind, int64_t q_row0, float* dist,
                       float* best_val, int32_t* best_idx, void* ws, void* stream);

/* ---- a10 / f4: log-mel front-end and the 2-D CNN encoder variant (BASELINE.json config 4) -------------------
 * NOT in the reference (nothing to cite under /root/reference); the specification is DESIGN.md section 9 and the
 * CPU oracle is oracle/voicemap_oracle.py (logmel_features, encoder2d_forward): parity unpinned by construction.
 *
 * vm_stft_logmel: raw (n_clips, raw_len) fp32 or int16 16 kHz windows -> frames of win_length samples every `hop` samples
 * (T = vm_stft_frames: no centre padding), 512-point DFT of the windowed frame, power of bins 0..255, mel projection, log:
 *   out[(b * n_mels + m)][1 + t] = log(sum_k melw[k][m] * |X_t[k]|^2 + log_floor)
 * written in `dtype` as the block-1 input of the 2-D encoder: n_clips * n_mels windows of T + 2 rows (halo rows are not
 * written: the caller zeroes the buffer once), 1 channel.  basis: (win_length, 512) fp32 = the analysis window times
 * [cos(2 pi k n / 512) for k < 256 | -sin(2 pi k n / 512) for k < 256]; melw: (256, n_mels) fp32; both built by the host side
 * (voicemap_amd/spectro.py).  n_mels in {32, 64, 96, 128}. */
int64_t vm_stft_frames(int64_t raw_len, int win_length, int hop);
int vm_stft_logmel(const void* raw, int is_int16, int64_t n_clips, int64_t raw_len, int win_length, int hop, const float* basis,
                   const float* melw, int n_mels, float log_floor, int dtype, void* out, void* stream);
/* vm_stft_logmel with the DFT on the f16 matrix pipe: the windowed basis and the (x 256) samples are split into two f16 halves each
 * (22 significand bits) and the three significant products accumulated in fp32 -- ~2^-21 relative per product against fp32's 2^-24, at
 * 1/5 of the matrix time (an fp32 MFMA moves K = 2 per 64 clocks).  Meant for the 16-bit storage modes, whose log-mel image is rounded
 * to 11 / 8 significand bits anyway; the fp32-storage mode keeps vm_stft_logmel.  basis16: vm_stft_split_basis_bytes(win_length) bytes
 * filled ONCE by vm_stft_split_basis from the fp32 basis; everything else as vm_stft_logmel. */
int64_t vm_stft_split_basis_bytes(int win_length);
int vm_stft_split_basis(const float* basis, int win_length, void* basis16, void* stream);
int vm_stft_logmel_f16s(const void* raw, int is_int16, int64_t n_clips, int64_t raw_len, int win_length, int hop, const void* basis16,
                        const float* melw, int n_mels, float log_floor, int dtype, void* out, void* stream);
/* ... which also leaves what the 16-bit storage type dropped as a second plane of the same layout: out_lo = image - out, rounded to
 * `dtype` (halo rows not written).  The log-mel image is the one tensor of the variant with a pedestal (mean -1.7, range -10 .. 5 on
 * speech-like clips) and ONE channel: its 11-bit rounding alone moved config 4's embeddings by 7.7e-4 of the 1e-3 budget
 * (tools/probe/config4_storage_sites.py); two planes cost 2 bytes per pixel more.  16-bit storage only. */
int vm_stft_logmel_f16s_split(const void* raw, int is_int16, int64_t n_clips, int64_t raw_len, int win_length, int hop, const void* basis16,
                              const float* melw, int n_mels, float log_floor, int dtype, void* out, void* out_lo, void* stream);
/* First Conv2D(3 x 3) of the variant (one input channel) with its own kernels instead of as a band-stacked GEMM with K = 24, N = 32
 * (16-bit storage and C % 32 == 0: one v_mfma_f32_32x32x16 per 32 positions x 32 channels, the nine taps as K, and
 * v_mfma_f32_16x16x32 tiles with the positions as K for the gradient; otherwise on the vector ALUs):
 * in: block input (n_clips * M, L + 2, 1) `dtype` with zero halo rows; w: the fp32 kernel (3, Cs, C) of the flat store (Cs >= 3, the
 * entries km >= 3 are padding), rounded to `dtype` inside like the GEMM path's copy; z (n_clips * M, L, C) = relu(conv + bias);
 * stat_sum / stat_sq (optional): the BatchNorm partial sums of the stored z, vm_conv_stat_rows(L) rows per window.  C % 8 == 0,
 * C <= 128.  vm_conv2d_first_wgrad: the matching kernel gradient (3, Cs, C) from du (n_clips * M, L + 2, C) (padding entries 0);
 * the layer has no input gradient. */
int vm_conv2d_first_supported(int C, int dtype);
int vm_conv2d_first_fwd(const void* in, const float* w, const float* bias, int64_t n_clips, int M, int64_t L, int Cs, int C, int dtype,
                        void* z, float* stat_sum, float* stat_sq, void* stream);
/* vm_conv2d_first_fwd on the two-plane image (in + in_lo of vm_stft_logmel_f16s_split) with the filters split the same way inside
 * (w = w_hi + w_lo, both `dtype`): the three significant products in w_hi + in_lo w_hi + in w_lo are 27 of the 32 K slots of TWO
 * matrix instructions per 32 positions x 32 channels, fp32 accumulation -- image and filters enter at ~2 x the storage significand
 * (the layer has 9 taps and one input channel: the second instruction is the whole price).  Same z layout, statistics and rounding
 * of the stored z as vm_conv2d_first_fwd; the weight gradient keeps vm_conv2d_first_wgrad on the high plane.  z_lo (optional): what
 * the storage type dropped of relu(conv + bias) as a second plane of z's layout.  The statistics are those of the two-plane value
 * (z + z_lo, stored or not): what vm_conv2d_first_bn_pool_stack recomputes and vm_bn_pool2d_stack_fwd_split reads back (the backward
 * keeps reading z alone).  16-bit storage and C % 32 == 0 only. */
int vm_conv2d_first_fwd_split(const void* in, const void* in_lo, const float* w, const float* bias, int64_t n_clips, int M, int64_t L, int Cs,
                              int C, int dtype, void* z, void* z_lo, float* stat_sum, float* stat_sq, void* stream);
int64_t vm_conv2d_first_wgrad_workspace_bytes(int64_t n_clips, int M, int C);
int vm_conv2d_first_wgrad(const void* in, const void* du, int64_t n_clips, int M, int64_t L, int Cs, int C, int dtype, void* ws,
                          float* grad_w, void* stream);
/* Conv2D(3 x 3, SAME) over (T, M) = the k = 3 convolution along T (vm_conv_fwd / _dgrad / _wgrad) of the band-stacked tensor:
 * x: (n_clips * M windows, rows, C) with rows = T + 2 (halo rows included: they are copied, so they stay zero);
 * out: (n_clips * M, rows, Cs), out[(b, m)][row][dm * C + c] = x[(b, m + dm - 1)][row][c] (0 outside the clip's M bands),
 * channels [3 C, Cs) = 0 (Cs >= 3 C, a multiple of 8 for the convolution kernels).  W2d[kt][km][ci][co] = W1d[kt][km * C + ci][co]. */
int vm_stack_windows(const void* x, int64_t n_clips, int M, int64_t rows, int C, int Cs, int dtype, void* out, void* stream);
/* adjoint of vm_stack_windows on un-padded rows: dxs (n_clips * M, L, Cs) -> dx (n_clips * M, L, C),
 * dx[(b, m)][t][c] = sum_dm dxs[(b, m - dm + 1)][t][dm * C + c].  src_padded != 0: dxs is (n_clips * M, L + 2, Cs) and rows 1 .. L of a
 * window are read -- the layout a dgrad over a clip's concatenated windows leaves (its outputs at the halo positions are not used). */
int vm_fold_windows(const void* dxs, int64_t n_clips, int M, int64_t L, int C, int Cs, int src_padded, int dtype, void* dx, void* stream);
/* mel half of MaxPool2D(2, 2): out[(b, m')][row][c] = max(q[(b, 2m')], q[(b, 2m'+1)]) over (n_clips * M, rows, C) -> (n_clips *
 * (M / 2), rows, C) (floor: an odd last band is dropped); backward: q as in the forward (rows = L + 2 with halo), dout (n_clips *
 * (M / 2), L, C), dq (n_clips * M, L, C) = dout routed to the first maximum of each pair. */
int vm_pool_windows_fwd(const void* q, int64_t n_clips, int M, int64_t rows, int C, int dtype, void* out, void* stream);
int vm_pool_windows_bwd(const void* q, const void* dout, int64_t n_clips, int M, int64_t L, int C, int dtype, void* dq, void* stream);
/* One block boundary of the 2-D variant in one pass (round 3; bit-identical to the three / two passes they replace):
 * vm_bn_pool2d_stack_fwd = vm_bn_drop_pool_fwd(pool 2) -> vm_pool_windows_fwd -> vm_stack_windows.  z (n_clips * M, L, C), scale / shift
 * (towers, C) with `clips_per_tower` clips a tower, drop (n_clips * M, C) or NULL; writes q (n_clips * M, L / 2 + 2, C) rows 1 .. L / 2
 * (kept for the backward) and the stacked block input xs (n_clips * (M / 2), L / 2 + 2, Cs).  NEITHER tensor's halo rows, out-of-clip band
 * slots or channels [3 C, Cs) are written: zero both once after allocation.  C, Cs multiples of the 16-byte vector.
 * vm_fold_pool_windows_bwd = vm_fold_windows -> vm_pool_windows_bwd: dxs (n_clips * (M / 2), L (+ 2 if src_padded), Cs) and q as above
 * (L = its un-padded rows) -> dq (n_clips * M, L, C).  s0 / sa (optional, both or neither; (n_clips * M * vm_fold_pool_windows_rows(...), C)
 * fp32): the sums of dq and of dq * q per window and workgroup row -- what vm_bn_bwd_from_sums_finalize(..., rows_per_window =
 * vm_fold_pool_windows_rows, a_is_act = 1) turns into the BatchNorm-backward constants of the block below without a pass over (z, dq). */
int vm_bn_pool2d_stack_fwd(const void* z, const float* scale, const float* shift, const float* drop, int64_t n_clips, int M,
                           int64_t clips_per_tower, int64_t L, int C, int Cs, int dtype, void* q, void* xs, void* stream);
/* Block 1's boundary without the round trip of z: given the BatchNorm affine (the statistics come from vm_conv2d_first_fwd_split, which
 * also leaves z for the backward), RECOMPUTE relu(conv + bias) on the two-plane image -- nine taps of a cache-resident one-channel
 * image: cheaper than reading z back -- apply scale / shift (+ drop) to the fp32 accumulator, pool 2 x 2 and write what
 * vm_bn_pool2d_stack_fwd writes (q per band, the band-stacked xs from the STORED q values; same layouts, same never-written halo /
 * out-of-clip / padding entries).  The rounding of block 1's conv output is gone from the forward (5.0e-4 of config 4's f16 embedding
 * error) and so are its bytes.  in / in_lo / w / bias / Cs as vm_conv2d_first_fwd_split; scale / shift (towers, C); drop (n_clips * M, C)
 * or NULL; Cs2 = channel pitch of xs (>= 3 C).  16-bit storage, C in {32, 64, 96, 128}. */
int vm_conv2d_first_bn_pool_stack(const void* in, const void* in_lo, const float* w, const float* bias, const float* scale, const float* shift,
                                  const float* drop, int64_t n_clips, int M, int64_t clips_per_tower, int64_t L, int Cs, int C, int Cs2,
                                  int dtype, void* q, void* xs, void* stream);
/* vm_bn_pool2d_stack_fwd on a z that came as two planes (vm_conv2d_first_fwd_split): the affine is taken of z + z_lo in fp32. */
int vm_bn_pool2d_stack_fwd_split(const void* z, const void* z_lo, const float* scale, const float* shift, const float* drop, int64_t n_clips,
                                 int M, int64_t clips_per_tower, int64_t L, int C, int Cs, int dtype, void* q, void* xs, void* stream);
int64_t vm_fold_pool_windows_rows(int64_t L, int C, int Cs, int dtype);
int vm_fold_pool_windows_bwd(const void* dxs, const void* q, int64_t n_clips, int M, int64_t L, int C, int Cs, int src_padded, int dtype,
                             void* dq, float* s0, float* sa, void* stream);
/* mel half of GlobalMaxPool2D: out[b][c] = max over m < M_valid of gmax[(b, m)][c] (fp32; first maximum -> widx[b][c]);
 * backward: dg[(b, m)][c] = dout[b][c] if m == widx[b][c] else 0. */
int vm_clip_max_fwd(const float* gmax, int64_t n_clips, int M, int M_valid, int C, float* out, int32_t* widx, void* stream);
int vm_clip_max_bwd(const float* dout, const int32_t* widx, int64_t n_clips, int M, int C, float* dg, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VOICEMAP_HIP_H */
