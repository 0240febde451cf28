/*
 * aitk_mi355.h — C ABI of the MI355X-native diffusion LoRA train-step hot path.
 *
 * Drop-in boundary: the reference (ostris/ai-toolkit) is pure Python; its boundary for this
 * path is a monkey-patched nn.Linear.forward (toolkit/lora_special.py:132-135, toolkit/network_mixins.py:274-348)
 * plus the model forward called from StableDiffusion.predict_noise (toolkit/stable_diffusion_model.py:2154-2222)
 * and the clip/AdamW/EMA calls in SDTrainer.hook_train_loop (extensions_built_in/sd_trainer/SDTrainer.py:2273-2293).
 * These entry points are what a ctypes binding on the reference side would call instead.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (torch allocates; no ownership transfer);
 *  - bf16 tensors are raw uint16 bit patterns; "f32" tensors are IEEE float;
 *  - `stream` is a hipStream_t passed as void*; kernels are enqueued, never synchronised;
 *  - return 0 on success, AITK_ERR_* (<0) for invalid shape/alignment/argument, >0 = hipError_t passed through;
 *  - single training thread per process; one process per GPU.
 */
#ifndef AITK_MI355_H
#define AITK_MI355_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* aitk_stream_t;
typedef uint16_t aitk_bf16;

#define AITK_ABI_VERSION 12 /* 12: aitk_lora_wgrad_main + aitk_lora_wgrad_finish_multi (the finish pass of up to 8 weight-gradient launches as one launch, deferred until something reads the gradient); 11: aitk_lora_down_ksplit (aitk_lora_down for a few rows over a long contraction: K slices across workgroups + the aitk_lora_t_finish pass, one call); 10: AITK_EPI_EMIT_T + AitkGemmArgs.t_* (a GELU launch emits the column-tile partials of the NEXT layer's lora_down product), aitk_lora_t_finish, aitk_lora_down_raw; 9: aitk_ema_update (the EMA of toolkit/ema.py over the flat arenas as its own launch: trainers that call optimizer.step() and ema.update() separately), AitkAttnArgs.dS (the dK/dV pass can emit dS for a GEMM-form dQ); 8: AitkMseArgs.max_loss / guard and AitkAdamWArgs.guard / n_micro (device-side failure handling of the train loop: non-finite loss, max_loss clamp, skipped optimizer step with device-resident step count); aitk_adamw_workspace_bytes grew by the 32-byte control block; 7: aitk_probe_gemm8_trace (no struct or semantic change: the fast epilogue forms of the persistent GEMM, the single-pass LN-modulate backward and the wave-per-token QK-norm + RoPE kernels keep their entry points' contracts); 6: aitk_grad_compress_bf16 / aitk_grad_expand_bf16 (bf16 transport of the DP all-reduce), aitk_lora_wgrad2 (lora_down gradient from a two-part operand [g | gelu(pre-activation)]), AitkShadowDesc.aux = row stride of the kind-1 data-gradient block (same-input groups share one [in, 3R] matrix); 5: aitk_slab_rescale, AitkAttnArgs.hstride (heads read in their native [tokens, H*d] layout), AITK_EPI_SPLIT_SLAB for N = 2 rp <= 128 with the stacked rows in 16-rank blocks (shadow kind 4 writes that order), aitk_lora_down / aitk_lora_wgrad accept split_rp > R (64-rank chunks of one slab); 4: AitkMseArgs.loss_type / huber_c (mae, pseudo_huber), AitkAdamWArgs.ema_feedback / param_multiplier, AitkGemmArgs.a_scale + b_scale_mode 3 (W8A8 on the MX-scaled fp8 MFMA), aitk_quant_rows_fp8, aitk_image_resize_to_nhwc8; 3: 3: conv_t3d (3-D convolution), AITK_EPI_SPLIT_SLAB, K-slab in conv mode, shadow kind 4, aitk_rmsnorm_rows, aitk_latent_sample_affine, aitk_pad_nhwc */

/* ---- GEMM epilogue flags ---- */
#define AITK_EPI_BIAS 1      /* + bias[n]                                                        */
#define AITK_EPI_ACCUM 2     /* + C_old[m][n]  (sum of several dgrads into one dX)               */
#define AITK_EPI_GELU 4      /* aux_out = u (pre-activation, bf16); C = gelu_tanh(u)             */
#define AITK_EPI_DGELU 8     /* C = val * gelu_tanh'(aux_in)                                     */
#define AITK_EPI_GATE_RES 16 /* aux_out = y; C = aux_in(residual) + gate[m / gate_rows][n] * y   */
#define AITK_EPI_BIAS_ROW 32 /* + bias[m]  (transposed products, e.g. V^T = W_v x^T)             */
#define AITK_EPI_ADD_AUX 64  /* + aux_in[m][n]  (residual add of ResnetBlock2D / VAE attention)  */
#define AITK_EPI_SPLIT_SLAB 256 /* B rows = 16-rank blocks [P_hi(16) ; P_lo(16)] (bf16 hi / lo halves of a rank-rp fp32 projection, N = 2 rp <= 128):
                                  the fp32 sum of columns n and n + 16 of every 32-column block is written as the K-slab triple
                                  [hi(rp) | lo(rp) | hi(rp)] (C is [M, >= 3 rp]) — lora_down of a 3x3-conv adapter
                                  (toolkit/lora_special.py:95-104) on the implicit-GEMM kernel; only COL_SCALE may accompany it */
#define AITK_EPI_COL_SCALE 128 /* product * col_scale[n] before the bias: DoRA magnitude / ||W + dW||_row (toolkit/models/DoRA.py:126-148) */
#define AITK_EPI_EMIT_T 512 /* with BIAS | GELU only (rank 16 or 32): the launch also leaves, per 256-column tile, the fp32 partial product of ITS OUTPUT with the
                              lora_down matrix of the layer that consumes it — t_partial[t_tile0 + n / 256][m][0..15] = sum over the tile's columns of
                              gelu(u)[m][n] * (t_p + t_p_lo)[r][n] on the bf16 value it stores — so that the consumer's T = x A^T (toolkit/network_mixins.py:309-321:
                              lora_down on the layer input) needs no second pass over the 792-MB GELU output: aitk_lora_t_finish sums the tiles.  Persistent
                              8-phase kernel only: N % 256 == 0 (any M), no row maps; anything else is AITK_ERR_SHAPE (the caller keeps aitk_lora_down). */

/*
 * C[M,N] = epi( A[M,K] B[N,K]^T + A2[M,K2] B2[N,K2]^T + bias )       (bf16 in/out, fp32 accumulate)
 * Replaces: org Linear forward + LoRAModule._call_forward (toolkit/network_mixins.py:197-239, 304-342) in one
 * tile pass, and autograd's dX for the same pair in backward.  Row m of A lives at
 *   A + (m / a_seg_rows) * a_seg_stride + (m % a_seg_rows) * lda     (a_seg_rows == 0: plain m * lda)
 * so the image / text halves of the joint [B, S_txt+S_img, C] attention buffers are addressable without copies;
 * C rows use the same scheme with c_seg_*.
 */
typedef struct AitkGemmArgs {
  const aitk_bf16* A; int64_t lda; int32_t a_seg_rows; int32_t _pad0; int64_t a_seg_stride;
  const aitk_bf16* B; int64_t ldb;
  const aitk_bf16* A2; int64_t lda2;
  const aitk_bf16* B2; int64_t ldb2;
  aitk_bf16* C; int64_t ldc; int32_t c_seg_rows; int32_t _pad1; int64_t c_seg_stride;
  const aitk_bf16* bias;
  aitk_bf16* aux_out; int64_t ld_aux_out;
  const aitk_bf16* aux_in; int64_t ld_aux_in;
  const aitk_bf16* gate; int64_t ld_gate; int32_t gate_rows;
  int32_t M, N, K, K2;
  int32_t flags;
  int32_t stage_mode; /* 0 = VGPR-staged 2-barrier kernel, 1 = auto: persistent 8-phase LDS-DMA kernel for big problems, else
                         2-barrier LDS-DMA kernel, 4 = force the 8-phase kernel, 5 = force the 2-barrier LDS-DMA kernels */
  int32_t tile_mode;  /* 0 = auto, 1 = 128x128 (4 waves), 2 = 256x256 (8 waves) */
  /* implicit-GEMM 3x3 convolution (conv_mode = 1): A = NHWC input [B, conv_H, conv_W, conv_Cin], M = B*Ho*Wo,
   * K = 9*conv_Cin with k = (ky*3+kx)*Cin + cin, B = weight [Cout, K]; zero_page = >=16 B of zeros on the device.
   * conv_t3d != 0 selects the 3-D form used by the Wan2.1 video VAE (toolkit/models/wan21/wan21.py:659): conv_t3d = kt | tstride << 8 |
   * ks << 16 (kt = 1..3 temporal taps, tstride = 1..2, ks = 3 or 1 spatial kernel size); the batch index is then a frame, K =
   * kt*ks*ks*conv_Cin with k = ((dt*ks + ky)*ks + kx)*Cin + cin, and tap dt of output frame t reads input frame t*tstride + dt of A
   * (A starts with the zero frames of a causal convolution: there is no bounds check in time). */
  int32_t conv_mode;
  int32_t conv_H, conv_W, conv_Cin, conv_Wo, conv_HoWo, conv_stride, conv_pad_t, conv_pad_l, conv_t3d;
  const aitk_bf16* zero_page;
  /* weight-only fp8 base operand (b_scale_mode != 0): B points to OCP e4m3 bytes [N, K] (ldb in bytes), dequantised to
   * bf16(fp8 * scale) on the way into LDS.  1: scale[n] per B row (forward); 2: scale[k] per contraction index (dgrad on W^T). */
  const float* b_scale; int32_t b_scale_mode; int32_t _pad4;
  const float* col_scale; /* fp32 [N], AITK_EPI_COL_SCALE */
  /* b_scale_mode 3 — W8A8 on the MX-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, unit block scales; BASELINE config 5's "CDNA4 fp8
   * MFMA base"): A AND B point to OCP e4m3 bytes ([M, K] / [N, K], lda / ldb / a_seg_stride in bytes, multiples of 16),
   *   C = epi( a_scale[m] * b_scale[n] * (A B^T)  +  A2 B2^T  + bias ),   a_scale fp32 [M] / b_scale fp32 [N] (NULL = 1),
   * i.e. per-token dynamic activation quantisation (aitk_quant_rows_fp8) against per-output-channel weight scales; the rank-r LoRA slab
   * A2 B2^T stays bf16 (split hi + lo) and is added un-scaled.  Persistent 8-phase kernel only (any M, N % 8 == 0, K % 16 == 0). */
  const float* a_scale;
  /* AITK_EPI_EMIT_T: t_partial fp32 [tiles][M][t_rank]; t_p / t_p_lo = bf16 hi / lo shadows of the consumer's lora_down rows, indexed by THIS launch's output
   * column (row stride t_ldp elements, 16-byte aligned; a column window of a wider matrix is a pointer offset); t_tile0 = first tile slot of this launch */
  float* t_partial; const aitk_bf16* t_p; const aitk_bf16* t_p_lo; int64_t t_ldp; int32_t t_tile0; int32_t t_rank; /* t_rank: 16 (or 0) / 32 = rows of t_p and ranks per slab row */
} AitkGemmArgs;

/* Per-token (per-row) dynamic fp8 quantisation of a GEMM A operand for b_scale_mode 3:
 *   Q[m][k] = e4m3(X[m][k] * col_mul[k] / row_scale[m]),  row_scale[m] = max_k |X[m][k] * col_mul[k]| / 448.
 * X rows follow the (seg_rows, seg_stride) row map of AitkGemmArgs.A.  col_mul (optional, fp32 [K]) folds a scale that runs along the
 * contraction axis into the operand: the data gradient dX = dY W contracts over the OUTPUT channels, whose per-channel weight scale
 * (toolkit/util/quantize.py:43-75 quantises per output channel) therefore multiplies dY before it is quantised. */
typedef struct AitkQuantRowsArgs {
  const aitk_bf16* X; int64_t ldx; int32_t seg_rows; int32_t _pad0; int64_t seg_stride;
  const float* col_mul;
  uint8_t* Q; int64_t ldq;
  float* row_scale;
  int32_t M, K;
} AitkQuantRowsArgs;
int aitk_quant_rows_fp8(const AitkQuantRowsArgs* args, aitk_stream_t stream);

int aitk_abi_version(void);
/* diagnostics: s_memtime stamps around the barriers of the wave-specialised dK/dV kernel's TRACE build (AITK_ATTN_DKDV_WS=2): 64 values */
int aitk_probe_attn_ws_trace(uint64_t* out64);
/* diagnostics: s_memtime stamps the persistent GEMM's TRACE build (AITK_GEMM8_TRACE=1) left at its tile-switch points: 60 values = [workgroup 0 / 100][wave 0 / 4][tile 1-3]
 * [top of the tile, after the top barrier, end of the steady K loop, end of the K loop, end of the epilogue] (tools/gpu_gemm8_ev.py trace) */
int aitk_probe_gemm8_trace(uint32_t* out60);
/* diagnostics: the work list of the persistent GEMM's opt-in stream-K tail (AITK_GEMM8_SK, csrc/gemm8.hip) evaluated on the host: item i of workgroup w of G
 * (G % 8 == 0) over ntiles output tiles of nsteps K-tiles -> out3 = {virtual tile, first K-tile, end K-tile}; 1 past the workgroup's last item.
 * aitk_probe_gemm8_sk_predecessor: the workgroup whose workspace slot a chunk of w that starts past K-tile 0 adds before it goes on. */
int aitk_probe_gemm8_sk_item(int32_t G, int32_t w, int32_t ntiles, int32_t nsteps, int32_t i, int32_t* out3);
int aitk_probe_gemm8_sk_predecessor(int32_t G, int32_t w, int32_t ntiles, int32_t nsteps);
int aitk_sizeof(int32_t which); /* 0: AitkGemmArgs, 1: AitkLoraDownArgs, 2: AitkLoraWgradArgs, ... — struct-size handshake for FFI mirrors */
int aitk_gemm_nt(const AitkGemmArgs* args, aitk_stream_t stream);
/* Two independent problems in one call (e.g. the image- and text-stream projections of a FLUX double block:
 * transformer_blocks.N.attn.to_q / add_q_proj — different weights, adapters and row counts).  With equal N, K, K2 and flags, and big
 * enough together, they run as ONE persistent launch whose tile list is the concatenation of both (the small problem fills the big
 * one's partly empty last tile round); otherwise they are launched back to back.  Results are bitwise those of two aitk_gemm_nt calls. */
int aitk_gemm_nt_grouped(const AitkGemmArgs* a, const AitkGemmArgs* b, aitk_stream_t stream);


/*
 * T[M,R] = scale * mult[m / rows_per_batch] * (X[M,K] (P + P_lo)[R,K]^T)     R <= 64, R % 4 == 0, K % 16 == 0
 * forward : P = lora_down.weight            -> T   (reference: toolkit/network_mixins.py:197-239, 309-321)
 * backward: X = dY, P = lora_up.weight^T    -> dT  (autograd of the same lines)
 * mult may be NULL (multiplier 1); X rows may be segmented like AitkGemmArgs.A.
 * Split precision — the reference keeps the adapter in fp32 (network_mixins.py:309, BaseSDTrainProcess.py:1982-1983):
 *   P_lo (may be NULL): bf16(w - bf16(w)) of the fp32 matrix whose bf16 rounding is P (same ldp); both are contracted into
 *     one fp32 accumulator, so X (P + P_lo)^T carries 16 mantissa bits of the fp32 weight.
 *   split_rp == 0: T is [M, R] bf16 (one rounding).
 *   split_rp  > 0 (R % split_rp == 0, split_rp % 4 == 0): T is [M, 3R]; rank block b (split_rp ranks) is written as
 *     [hi | lo | hi] at columns 3*b*split_rp, hi = bf16(t), lo = bf16(t - hi) — the K-slab A2 of aitk_gemm_nt, to be paired
 *     with B2 = [B_hi | B_hi | B_lo] (aitk_lora_refresh_shadows), and the S operand of aitk_lora_wgrad(split_rp).
 */
typedef struct AitkLoraDownArgs {
  const aitk_bf16* X; int64_t ldx; int32_t x_seg_rows; int32_t _pad0; int64_t x_seg_stride;
  const aitk_bf16* P; int64_t ldp;
  aitk_bf16* T; int64_t ldt;
  const float* mult; float scale; int32_t rows_per_batch;
  int32_t M, K, R, split_rp;
  const aitk_bf16* P_lo;
  /* dropout on the rank-space activation (toolkit/network_mixins.py:212-228): fp32 [rows, R] multipliers (0 or 1 / keep-probability)
   * applied to T before rounding; row = m / tmask_rows_per_batch (rank_dropout: one mask row per sample) or m (neuron dropout, 0).
   * The backward call (X = dY, P = lora_up^T) takes the SAME mask: d(T * mask)/dT = mask.  May be NULL. */
  const float* tmask; int32_t tmask_rows_per_batch; int32_t _pad1;
} AitkLoraDownArgs;
int aitk_lora_down(const AitkLoraDownArgs* args, aitk_stream_t stream);
/* The two halves of aitk_lora_down around a partial-sum slab [tiles][M][R] fp32 (AITK_EPI_EMIT_T fills tiles from inside the producing GEMM):
 *   aitk_lora_down_raw: raw[m][r] = sum_k X[m][k] (P + P_lo)[r][k]  — one more tile, un-scaled, nothing written to T (R = 16 or 32, K % 32 == 0): the part of
 *     a consumer's input that did NOT come out of an emitting launch (the attention half of the single blocks' [attn | gelu(mlp)] operand);
 *   aitk_lora_t_finish: T[m] = what aitk_lora_down writes (scale, mult, tmask, plain or [hi | lo | hi] slab) from the sum of `ntiles` tiles, fixed order.
 * X / P / ldx / K of the finish call are ignored. */
int aitk_lora_down_raw(const AitkLoraDownArgs* args, float* raw, aitk_stream_t stream);
int aitk_lora_t_finish(const AitkLoraDownArgs* args, const float* partial, int32_t ntiles, aitk_stream_t stream);
/* aitk_lora_down when M is a few rows and K is long (the adaLN adapters' backward, toolkit/network_mixins.py:309-321 under autograd: dT [B, r] = dmod [B, 6 d] lora_up):
 * one workgroup per 32 rows would pull the whole projection through one CU.  The contraction is cut into `nsplit` slices (1 <= nsplit <= K / 32), one workgroup
 * each; their raw fp32 tiles land in `partial` (aitk_lora_down_ksplit_workspace_bytes(M, R, nsplit) bytes, 16-byte aligned) and are summed in slice order
 * (deterministic) by a finish pass that applies scale / mult / tmask and writes T exactly like aitk_lora_down (same argument block, R == 16).  Equal to
 * aitk_lora_down up to the fp32 summation order. */
int64_t aitk_lora_down_ksplit_workspace_bytes(int32_t M, int32_t R, int32_t nsplit);
int aitk_lora_down_ksplit(const AitkLoraDownArgs* args, float* partial, int32_t nsplit, aitk_stream_t stream);

/*
 * out[r * out_stride_r + l * out_stride_l] (+)= sum_m S[m][r] * G[m][l]        fp32 out, R in {16,32,48,64}, L % 8 == 0
 *   lora_down.weight.grad [R,K]: S = dT, G = X,  strides (K, 1)
 *   lora_up.weight.grad   [N,R]: S = T,  G = dY, strides (1, R)
 * `partial` is caller-provided scratch of aitk_lora_wgrad_workspace_bytes(M,R,L) bytes.
 * split_rp > 0 (split_rp % 8 == 0): S is the [M, 3R] slab layout aitk_lora_down(split_rp) writes; rank r is read as
 *   hi + lo from columns (r / rp) * 3 rp + r % rp (+ rp), both contracted into the same fp32 accumulator.
 */
typedef struct AitkLoraWgradArgs {
  const aitk_bf16* S; int64_t lds;
  const aitk_bf16* G; int64_t ldg; int32_t g_seg_rows; int32_t split_rp; int64_t g_seg_stride;
  float* partial;
  float* out; int64_t out_stride_r; int64_t out_stride_l;
  int32_t accumulate;
  int32_t M, R, L;
} AitkLoraWgradArgs;
int64_t aitk_lora_wgrad_workspace_bytes(int32_t M, int32_t R, int32_t L);
int aitk_lora_wgrad(const AitkLoraWgradArgs* args, aitk_stream_t stream);
/* aitk_lora_wgrad whose G operand has a second part: columns l >= split_col (a multiple of 128, may be 0) are act(G2[m][l - split_col]),
 * act 0 = identity, 1 = tanh-GELU of a saved pre-activation (bit for bit the GEMM's AITK_EPI_GELU output).  G2 rows are plain (ldg2), the row map
 * (g_seg_*) applies to G only.  Lets a trainer drop the GELU outputs after the forward pass: lora_down.weight.grad of ff.net.2 / proj_out
 * (autograd of toolkit/network_mixins.py:309-321 on those layers) is formed from the pre-activation the GELU-backward epilogue keeps anyway. */
typedef struct AitkWgradSrc2 {
  const aitk_bf16* G2; int64_t ldg2;
  int32_t split_col; int32_t act;
} AitkWgradSrc2;
int aitk_lora_wgrad2(const AitkLoraWgradArgs* args, const AitkWgradSrc2* second, aitk_stream_t stream);
/* The two halves of aitk_lora_wgrad / aitk_lora_wgrad2 as calls of their own.  aitk_lora_wgrad_main (src2 NULL or the second operand part of aitk_lora_wgrad2) leaves
 * the chunk partials in args->partial and does not touch args->out; aitk_lora_wgrad_finish_multi takes 1..8 such argument blocks (each with its OWN partial buffer,
 * pairwise different `out`) and adds / stores their sums in one launch — bit for bit what the finish pass inside aitk_lora_wgrad does.  Nothing in a backward pass reads a
 * weight gradient (autograd of toolkit/network_mixins.py:309-321 ends in the optimizer), so a trainer can batch the 380 finish launches of a FLUX step eight at a time. */
int aitk_lora_wgrad_main(const AitkLoraWgradArgs* args, const AitkWgradSrc2* src2, aitk_stream_t stream);
int aitk_lora_wgrad_finish_multi(const AitkLoraWgradArgs* jobs, int32_t njobs, aitk_stream_t stream);
/* The two adapter-side products of a layer's backward that stream dY, from ONE read of it (autograd of toolkit/network_mixins.py:309-321:
 * lora_up.weight.grad = dY^T T and the gradient of the rank-r activation dT = c (dY B)):
 *   wgrad : aitk_lora_wgrad's arguments with S = T (slab), G = dY, out = lora_up.weight.grad (strides (1, R))
 *   down  : aitk_lora_down's arguments with X = dY (the same pointer / pitch / row map as wgrad->G), P / P_lo = lora_up^T shadows, T = dT
 *   dt_partial : aitk_lora_bwd_fused_workspace_bytes(M, R, L) bytes of scratch ([L / 128][M][R] fp32 column-tile partials of dT, summed in
 *   a fixed order).  R = 16 or 32.  Results equal the two separate calls up to fp32 summation order of dT. */
int64_t aitk_lora_bwd_fused_workspace_bytes(int32_t M, int32_t R, int32_t L);
int aitk_lora_bwd_fused(const AitkLoraWgradArgs* wgrad, const AitkLoraDownArgs* down, float* dt_partial, aitk_stream_t stream);
/* In place on a [hi(rp) | lo(rp) | hi(rp)] slab T [M, >= 3 rp]: (hi + lo)[m][r] * rowf[m / rows_per_batch] * tmask[m / tmask_rows_per_batch][r],
 * split again (rowf / tmask may be NULL, not both; tmask fp32 [rows, rp], tmask_rows_per_batch 0 = one mask row per slab row).  The per-sample
 * multiplier and the dropout / rank_dropout masks of a 3x3-conv adapter's rank-space activation, whose lora_down leaves the implicit-GEMM
 * epilogue with a uniform scale (toolkit/network_mixins.py:211-229, 235-239 on toolkit/lora_special.py:95-104 modules). */
int aitk_slab_rescale(aitk_bf16* T, int64_t ldt, int32_t M, int32_t rp, const float* rowf, int32_t rows_per_batch, const float* tmask,
                      int32_t tmask_rows_per_batch, aitk_stream_t stream);


/* ---- adaLN LayerNorm + modulate: out = LN(x; eps, no affine) * (1 + scale[b]) + shift[b],  b = m / rows_per_batch.
 * Replaces diffusers AdaLayerNormZero/ZeroSingle/Continuous bodies reached from
 * toolkit/stable_diffusion_model.py:2192-2205; mean/rstd (fp32 [M], may be NULL) are saved for backward. */
typedef struct AitkLnModArgs {
  const aitk_bf16* x; int64_t ldx;
  const aitk_bf16* shift; const aitk_bf16* scale; int64_t ld_mod;
  aitk_bf16* out; int64_t ld_out;
  float* mean; float* rstd;
  float eps; int32_t rows_per_batch; int32_t M, C;
} AitkLnModArgs;
int aitk_ln_mod_fwd(const AitkLnModArgs* args, aitk_stream_t stream);

/* backward of the above for B batches of S rows: dx = LN'(dxn*(1+scale)) + dres (dres may be NULL or alias dx);
 * partial (may be NULL) receives per-row-block column sums [B][nchunk][2][C] fp32 of (dshift, dscale),
 * nchunk = ceil(S / aitk_rows_per_block()); aitk_colsum_finish reduces them. */
typedef struct AitkLnModBwdArgs {
  const aitk_bf16* dxn; int64_t ld_dxn;
  const aitk_bf16* x; int64_t ldx;
  const float* mean; const float* rstd;
  const aitk_bf16* scale; int64_t ld_mod;
  const aitk_bf16* dres; int64_t ld_dres;
  aitk_bf16* dx; int64_t ld_dx;
  float* partial;
  int32_t S, B, C, _pad;
} AitkLnModBwdArgs;
int aitk_ln_mod_bwd(const AitkLnModBwdArgs* args, aitk_stream_t stream);
int32_t aitk_rows_per_block(void);

/* x_new = res + gate[b]*y  (forward is the GEMM epilogue AITK_EPI_GATE_RES):  dy = gate*dx ;
 * partial [B][nchunk][C] fp32 = per-row-block sums of dx*y (-> dgate). */
typedef struct AitkGateBwdArgs {
  const aitk_bf16* dx; int64_t ld_dx;
  const aitk_bf16* y; int64_t ld_y;
  const aitk_bf16* gate; int64_t ld_gate;
  aitk_bf16* dy; int64_t ld_dy;
  float* partial;
  int32_t S, B, C, _pad;
} AitkGateBwdArgs;
int aitk_gate_bwd(const AitkGateBwdArgs* args, aitk_stream_t stream);

/* partial [B][nchunk][V][C] fp32 -> out_v[b*ld_out + c] bf16 (V <= 2) */
typedef struct AitkColsumFinishArgs {
  const float* partial;
  aitk_bf16* out0; aitk_bf16* out1; int64_t ld_out;
  int32_t B, nchunk, V, C;
} AitkColsumFinishArgs;
int aitk_colsum_finish(const AitkColsumFinishArgs* args, aitk_stream_t stream);

/* ---- attention pre-processing: per-head RMSNorm(eps, weight[128]) + rotary embedding for q,k; plain copy for v
 * (weight == NULL).  Reads src rows [B*S_src, ld_src], writes rows (b*S_dst + s_off + s) of the joint buffer.
 * Order restated from toolkit/models/flux_sage_attn.py:36-74.  Backward uses the same struct: joint-side grads are
 * read from `dst`, raw-side grads written to `src`, `raw` = the forward input. */
typedef struct AitkQkvJob {
  const aitk_bf16* src; int64_t ld_src;
  aitk_bf16* dst; int64_t ld_dst;
  const aitk_bf16* weight;
  const aitk_bf16* raw; int64_t ld_raw;
} AitkQkvJob;
typedef struct AitkQkvPostArgs {
  AitkQkvJob job[3];
  const float* cos; const float* sin; /* [S_dst, 128] fp32 */
  float eps; int32_t njobs;
  int32_t B, H, D, S_src, S_dst, s_off;
} AitkQkvPostArgs;
int aitk_qkv_post_fwd(const AitkQkvPostArgs* args, aitk_stream_t stream);
int aitk_qkv_post_bwd(const AitkQkvPostArgs* args, aitk_stream_t stream);

/* small element-wise ops on [rows, C] bf16: op 0 y=silu(x); 1 y=x; 2 y=a+x; 3 y=alpha*x */
typedef struct AitkEwArgs {
  const aitk_bf16* x; int64_t ldx;
  const aitk_bf16* a; int64_t lda;
  aitk_bf16* y; int64_t ldy;
  int32_t rows, C, op; float alpha;
  int32_t a_rows_per_batch; int32_t _pad; /* op 2: > 0 broadcasts row (m / a_rows_per_batch) of `a` (ResnetBlock2D time-embedding add) */
} AitkEwArgs;
int aitk_ew(const AitkEwArgs* args, aitk_stream_t stream);
/* out[b] = [cos(t*tscale*f_i) | sin(...)], f_i = 10000^(-i/(dim/2))   (diffusers Timesteps, flip_sin_to_cos) */
int aitk_timestep_embed(const float* t, aitk_bf16* out, int32_t B, int32_t dim, float tscale, aitk_stream_t stream);
int aitk_copy2d(void* dst, int64_t dst_pitch_bytes, const void* src, int64_t src_pitch_bytes, int64_t width_bytes,
                int64_t rows, aitk_stream_t stream);


/* ---- attention (non-causal, unmasked, head_dim 128): O = softmax(Q K^T * scale) V over [B, S, H, 128] views
 * (row stride ld* elements, head h at column h*128).  Replaces F.scaled_dot_product_attention in the diffusers Flux
 * attention processor (order restated in toolkit/models/flux_sage_attn.py:76-93) and its autograd backward.
 * LSE [B,H,S] fp32 is in the scaled log2 domain: max2 + log2(sum exp2(s2 - max2)), s2 = q.k*scale*log2(e).
 * aitk_attn_bwd needs O, LSE from forward plus dO; writes dQ,dK,dV (same layout family) and delta [B,H,S] scratch. */
typedef struct AitkAttnArgs {
  const aitk_bf16* Q; const aitk_bf16* K; const aitk_bf16* V; int64_t ldq, ldk, ldv;
  aitk_bf16* O; int64_t ldo;
  float* LSE;
  const aitk_bf16* dO; int64_t lddo;
  aitk_bf16* dQ; aitk_bf16* dK; aitk_bf16* dV; int64_t lddq, lddk, lddv;
  float* delta;
  float scale; int32_t B, H, S, D, Skv; /* Skv: key/value rows per batch (0 = S); cross-attention has Skv != S */
  int32_t Dv, hstride; /* Dv: valid head width inside the 128-column layout (0 = 128): UNet heads of 40 / 64 / 80 are stored zero-padded; the
                      all-zero contraction steps and output blocks are skipped, the padded output columns are written as zeros.
                      hstride (0 = 128): elements between consecutive heads.  hstride == Dv in {64, 96} reads / writes the heads where
                      the projections put them ([B, S, H*Dv] — SDXL's 64-wide heads), no padded copies; other widths must be padded */
  /* ABI 9, probe of the 5-matmul backward (dQ = dS K as a product of its own instead of recomputing S and dP in a second pass): with
   * ds_mode != 0 the wave-specialised dK/dV kernel (head_dim 128) also writes the bf16 dS it forms for its own dK product to `dS`:
   *   1 = accumulator-native blocks: [b][h][kv block of 32][q block of 32] x 2 KiB, each block = the producer wave's two packed operand
   *       vectors as they sit in registers ([vector 2][lane 64][8 bf16]: lane = kv row, slot e of vector v = q row 16 v + 8 (e >> 2) +
   *       4 (lane >> 5) + (e & 3)), written with two coalesced 16-byte non-temporal stores per lane (S % 64 == 0, Skv % 128 == 0).
   * dQ is still produced by the recomputing kernel: the mode exists to MEASURE what emitting dS costs the dK/dV pass. */
  aitk_bf16* dS; int32_t ds_mode; int32_t _pad_ds;
} AitkAttnArgs;
int aitk_attn_fwd(const AitkAttnArgs* args, aitk_stream_t stream);
int aitk_attn_bwd(const AitkAttnArgs* args, aitk_stream_t stream);


/* ---- small-batch projection (Bm <= 8 rows): out[Bm,N] (+)= X W^T + bias + T Bl^T   (adaLN linears with LoRA, timestep /
 * guidance / pooled-text embedders).  Same math as AitkGemmArgs, shaped for weight streaming. */
typedef struct AitkGemvArgs {
  const aitk_bf16* X; int64_t ldx;
  const aitk_bf16* W; int64_t ldw;
  const aitk_bf16* bias;
  const aitk_bf16* T; int64_t ldt;
  const aitk_bf16* Bl; int64_t ldbl;
  aitk_bf16* out; int64_t ldo;
  int32_t Bm, N, K, R;
  int32_t accumulate; int32_t cols_per_group; /* cols_per_group: set by the library */
  const float* col_scale; /* may be NULL; DoRA: out = col_scale[n] * (X W^T + T Bl^T) + bias */
} AitkGemvArgs;
int aitk_gemv_nt(const AitkGemvArgs* args, aitk_stream_t stream);

/* ---- flow-matching noise mix + 2x2 patchify (toolkit/samplers/custom_flowmatch_sampler.py:91-102,
 * toolkit/stable_diffusion_model.py:2157-2163): noisy = (1-t/1000) x0 + (t/1000) eps, target = eps - x0, both packed
 * [B, (H/2)(W/2), 4C] bf16. */
typedef struct AitkNoisePackArgs {
  const aitk_bf16* latents; const aitk_bf16* noise; const float* t;
  aitk_bf16* noisy; aitk_bf16* target;
  int32_t B, C, H, W;
} AitkNoisePackArgs;
int aitk_flow_noise_pack(const AitkNoisePackArgs* args, aitk_stream_t stream);

/* ---- MSE loss (SDTrainer.py:916, 987-990, 1013) and its gradient wrt pred (bf16).  weight[b] (may be NULL) is the
 * per-sample loss multiplier (loss_multiplier / timestep weights, SDTrainer.py:932-944, 994).  mask (may be NULL) is the
 * reference's mask_multiplier (SDTrainer.py:959, 1484-1504: [B,1,h,w] broadcast over channels, already divided by its mean) in
 * the packed-token layout of pred: fp32 [B][n_per_sample / feat][4], element (token, feature f) uses entry f & 3 (the
 * (ph, pw) position inside the 2x2 patch); feat = features per token (64).  loss_b = mean_j (mask_j (pred_j - target_j)^2).
 * partial: aitk_mse_workspace_bytes() scratch. */
typedef struct AitkMseArgs {
  const aitk_bf16* pred; const aitk_bf16* target; const float* weight;
  aitk_bf16* dpred; float* partial; float* loss_per_sample; float* loss;
  int64_t n_per_sample; int32_t B, feat;
  const float* mask;
  int32_t loss_type; /* AITK_LOSS_MSE / _MAE / _PSEUDO_HUBER: train.loss_type (extensions_built_in/sd_trainer/SDTrainer.py:903-916) */
  float huber_c;     /* pseudo_huber: sqrt(d^2 + c^2) - c; the reference hard-codes c = 0.01; 0 selects that default */
  /* Failure handling of the reference's train loop, on the device (no host sync).  guard: int32[8] device buffer shared with
   * aitk_adamw_ema_step, NULL = off.  A non-finite loss is reported as 0 and its gradient zeroed (SDTrainer.py:2221-2224: `loss =
   * zeros_like(loss)`); with max_loss > 0 a loss above it is reported as max_loss and its gradient zeroed (SDTrainer.py:1049-1050:
   * torch.clamp(loss, max=max_loss) has derivative 0 there).  guard[0] = gated micro-batches of the current step (cleared by the optimizer
   * launch), [1] = non-finite losses so far, [2] = clamped losses so far, [3] = optimizer steps applied, [4] = optimizer steps skipped,
   * [5] = 1 if the last optimizer launch skipped, [6] = 1 if the last loss launch was gated. */
  float max_loss; int32_t _pad_guard;
  int32_t* guard;
} AitkMseArgs;
#define AITK_LOSS_MSE 0          /* (pred - target)^2 */
#define AITK_LOSS_MAE 1          /* |pred - target| (gradient sign(d), 0 at d = 0 like torch l1_loss) */
#define AITK_LOSS_PSEUDO_HUBER 2 /* sqrt(d^2 + c^2) - c */
int64_t aitk_mse_workspace_bytes(int32_t B, int64_t n_per_sample);
int aitk_mse_loss_grad(const AitkMseArgs* args, aitk_stream_t stream);

/* ---- clip_grad_norm_(max_norm) -> torch.optim.AdamW(eps=1e-6 by default in the toolkit) -> EMA over flat fp32 arenas
 * (SDTrainer.py:2278-2293, toolkit/optimizer.py:78-79, toolkit/ema.py:116-152).  g is multiplied by grad_scale before
 * clipping semantics (norm_out reports ||g * grad_scale||).  norm_partial: aitk_adamw_workspace_bytes(n) scratch. */
typedef struct AitkAdamWArgs {
  float* p; const float* g; float* m; float* v; float* ema;
  float* norm_partial; float* norm_partial2; float* norm_out;
  int64_t n;
  float lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2_sqrt, max_norm, ema_decay, grad_scale;
  /* toolkit/ema.py:126-152 options (train.ema_config.use_feedback / param_multiplier), applied to the parameter after the EMA update
   * in the reference's order: tmp = (1-d)(s - p); s -= tmp; p += ema_feedback * tmp (use_feedback: 10, else 0); p *= param_multiplier
   * (0 is read as 1).  Only with `ema`. */
  float ema_feedback, param_multiplier;
  /* guard (int32[8], see AitkMseArgs; NULL = off): the parameter update is SKIPPED — p, m, v untouched, the EMA still follows p like the
   * reference's ema.update() — when the gradient norm is not finite or all n_micro loss launches of the step were gated; the bias
   * corrections are then derived on the device from the number of applied steps (guard[3] + 1; bias_correction1 / bias_correction2_sqrt
   * of this struct are ignored), so a skipped step does not advance them — torch.optim.AdamW on parameters without .grad. */
  int32_t* guard; int32_t n_micro; int32_t _pad_micro;
  double beta1_d, beta2_d; /* the betas in double, as the host's `1 - beta ** step` uses them: with `guard` the device forms the same doubles */
} AitkAdamWArgs;
int64_t aitk_adamw_workspace_bytes(int64_t n);
int aitk_adamw_ema_step(const AitkAdamWArgs* args, aitk_stream_t stream);

/* toolkit/ema.py:126-152 (`ExponentialMovingAverage.update`) over the flat fp32 arenas in one launch, for callers that keep the reference's
 * call order — `optimizer.step()` (SDTrainer.py:2285) ... `self.ema.update()` (SDTrainer.py:2291-2293) — instead of the fused
 * aitk_adamw_ema_step: per element tmp = one_minus_decay * (s - p); s -= tmp; p += feedback * tmp (use_feedback: 10, else 0);
 * p *= param_multiplier (0 is read as 1), each operation rounded on its own like the reference's tensor ops.  p is only written when
 * feedback / param_multiplier ask for it.  Both pointers 16-byte aligned. */
int aitk_ema_update(float* p, float* ema, int64_t n, float one_minus_decay, float feedback, float param_multiplier, aitk_stream_t stream);

/* bf16 shadows of every adapter matrix of the fp32 arena (row-major [rows, cols] at src_off), refreshed after each optimizer
 * step.  hi = bf16(w), lo = bf16(w - hi).  Offsets are elements into `shadow`.
 *   kind 0 (plain; LoKr factors): d0 = hi [rows, cols], d1 = hi transposed [cols, rows].
 *   kind 1 (lora_down A [r, in]):  d0 = hi [r, in], d1 = lo [r, in] (P / P_lo of the forward aitk_lora_down),
 *                                  d2 = [in, 3r] rows = [A^T_hi | A^T_hi | A^T_lo] (B2 of the dgrad K-slab); aux > 0 = row stride of that
 *                                  block in elements (the adapters of a same-input group are column windows of ONE [in, 3R] matrix, the
 *                                  B2 operand of the group's K-concatenated data-gradient GEMM), aux = 0: 3r.
 *   kind 2 (lora_up B [out, r]):   d0 = [out, 3r] rows = [B_hi | B_hi | B_lo] (B2 of the forward K-slab),
 *                                  d1 = hi transposed [r, out], d2 = lo transposed [r, out] (P / P_lo of the backward aitk_lora_down).
 *   kind 3 (low-rank LoKr factor, toolkit/models/lokr.py:184-197): the arena holds a [rows, aux] followed by b [aux, cols];
 *                                  d0 = bf16(a @ b) [rows, cols] composed in fp32, d1 = its transpose [cols, rows].
 *   kind 4 (lora_down of a 3x3-conv adapter: Conv2d weight [r, Cin, 3, 3] = [rows, cols = 9 Cin], column cin*9 + tap; aux = Cin,
 *           toolkit/lora_special.py:95-104): d0 = [2r, 9 Cin] = [A_hi ; A_lo] with tap-major columns tap*Cin + cin (B operand of the
 *           implicit-GEMM lora_down, AITK_EPI_SPLIT_SLAB); d1 = [Cin, 9*3r], column (8 - tap)*3r + j = [A_hi | A_hi | A_lo] (the rotated
 *           filter of the data gradient, a 3x3 convolution over the dT slab image [hi | lo | hi]); d2 unused. */
typedef struct AitkShadowDesc { int64_t src_off; int64_t d0; int64_t d1; int64_t d2; int32_t rows, cols, kind, aux; } AitkShadowDesc;
int aitk_lora_refresh_shadows(const float* arena, aitk_bf16* shadow, const AitkShadowDesc* table, int32_t ntensors,
                              aitk_stream_t stream);

/* bf16 transport of the flat fp32 gradient arena for the data-parallel all-reduce (SURVEY.md section 8e "fp32 (parity) or bf16 (speed)"; the
 * reference's nominal DDP reduces whatever dtype the parameters have, jobs/process/BaseSDTrainProcess.py:1982-1983 keeps the network in fp32):
 * out[i] = bf16(g[i]) (round to nearest even) before the collective, g[i] = float(in[i]) after it.  Any element alignment; the 16-byte path
 * is taken when the two pointers reach a 16-byte boundary after the same number of elements (a transport buffer indexed like the arena). */
int aitk_grad_compress_bf16(const float* g, aitk_bf16* out, int64_t n, aitk_stream_t stream);
int aitk_grad_expand_bf16(const aitk_bf16* in, float* g, int64_t n, aitk_stream_t stream);

/* Low-rank LoKr: gradients of the pair lokr_w2_a [O, r], lokr_w2_b [r, I] from the gradient dW [O, I] of their product
 * (autograd of `lokr_w2_a @ lokr_w2_b`, toolkit/models/lokr.py:236-241, 331-339): ga (+)= dW b^T, gb (+)= a^T dW; all fp32. */
int aitk_lokr_lowrank_grad(const float* dW, const float* a, const float* b, float* ga, float* gb, int32_t O, int32_t I, int32_t r,
                           int32_t accumulate, aitk_stream_t stream);


/* ---- VAE encoder side kernels (NHWC bf16): GroupNorm(G groups, eps, gamma/beta [C]) with optional SiLU over
 * x [B, HW, C]; scratch from aitk_groupnorm_workspace_bytes.  Replaces nn.GroupNorm + SiLU of diffusers AutoencoderKL
 * (reached from toolkit/stable_diffusion_model.py:2567). */
typedef struct AitkGroupNormArgs {
  const aitk_bf16* x; int64_t ldx;
  aitk_bf16* y; int64_t ldy;
  const aitk_bf16* gamma; const aitk_bf16* beta;
  float* partial; float* stats; /* stats: set by the library (inside partial) */
  float eps; int32_t silu;
  int32_t B, HW, C, G;
  float* stats_out; /* may be NULL: fp32 [B][G][2] (mean, rstd) kept by the caller for aitk_groupnorm_bwd */
} AitkGroupNormArgs;
int64_t aitk_groupnorm_workspace_bytes(int32_t B, int32_t HW, int32_t C, int32_t G);
int aitk_groupnorm(const AitkGroupNormArgs* args, aitk_stream_t stream);
/* in-place softmax(scale * x) over the n columns of each row (VAE mid-block attention scores) */
int aitk_softmax_rows(aitk_bf16* x, int64_t ld, int32_t rows, int32_t n, float scale, aitk_stream_t stream);
/* image [B,3,H,W] fp32 -> NHWC bf16 with channels padded to 8 (conv_in operand) */
int aitk_image_to_nhwc8(const float* img, aitk_bf16* out, int32_t B, int32_t H, int32_t W, aitk_stream_t stream);
/* The same conversion behind a bilinear resize [Hs, Ws] -> [Hd, Wd] (align_corners = False, no antialiasing; pixels rounded to bf16 first, weights
 * in fp32): Wan21.encode_images' F.interpolate of inputs whose sides are not multiples of 8 (toolkit/models/wan21/wan21.py:652-657). */
int aitk_image_resize_to_nhwc8(const float* img, aitk_bf16* out, int32_t B, int32_t Hs, int32_t Ws, int32_t Hd, int32_t Wd, aitk_stream_t stream);
/* DiagonalGaussianDistribution.sample() + scaling_factor * (z - shift_factor): moments NHWC [B*hw, >=2L] -> NCHW [B,L,h,w]
 * (toolkit/stable_diffusion_model.py:2567-2573) */
int aitk_latent_sample(const aitk_bf16* moments, int64_t ldm, const float* eps, aitk_bf16* out, int32_t B, int32_t L, int32_t hw,
                       float scale, float shift, aitk_stream_t stream);
/* the same with a per-channel affine, out = ch_scale[c] * (z - ch_shift[c]) (ch_* fp32 [L] on the device): Wan21.encode_images'
 * `(latents - latents_mean) * (1 / latents_std)` (toolkit/models/wan21/wan21.py:661-670); with hw = T'*h*w the output is [B, L, T', h, w] */
int aitk_latent_sample_affine(const aitk_bf16* moments, int64_t ldm, const float* eps, aitk_bf16* out, int32_t B, int32_t L, int32_t hw,
                              const float* ch_shift, const float* ch_scale, aitk_stream_t stream);
/* WanRMS_norm over the channel axis of NHWC rows (+ optional SiLU): y = x / max(||x||_2, eps) * sqrt(C) * gamma[c]
 * (F.normalize(x, dim=1) * scale * gamma of diffusers' AutoencoderKLWan, reached from toolkit/models/wan21/wan21.py:659).
 * x, y [M, C] bf16 (may alias), gamma bf16 [C], C % 8 == 0, C <= 2048. */
int aitk_rmsnorm_rows(const aitk_bf16* x, int64_t ldx, aitk_bf16* y, int64_t ldy, const aitk_bf16* gamma, int64_t M, int32_t C,
                      float eps, int32_t silu, aitk_stream_t stream);


/* ---- RMSNorm across heads (weight [C], C = H*128) + optional rotary embedding, one row per token (Wan2.1 q/k path,
 * toolkit/models/wan21/wan_attn.py:32-54).  fwd: y = rope(norm(x)); bwd: y = d/dx given g = d/dy and the forward input x.
 * cos/sin [S,128] fp32 (NULL = no rope), row m uses position m % S. */
typedef struct AitkRmsFullArgs {
  const aitk_bf16* x; int64_t ldx;
  const aitk_bf16* g; int64_t ldg;
  aitk_bf16* y; int64_t ldy;
  const aitk_bf16* weight;
  const float* cos; const float* sin;
  float eps; int32_t S; int64_t M; int32_t C, _pad;
} AitkRmsFullArgs;
int aitk_rms_full_fwd(const AitkRmsFullArgs* args, aitk_stream_t stream);
int aitk_rms_full_bwd(const AitkRmsFullArgs* args, aitk_stream_t stream);

/* out[r][k] = bf16(e4m3(q[r][k]) * scale[mode == 1 ? r : k]): one layer's weight-only-fp8 base weight expanded into a reusable
 * bf16 scratch right before its GEMM (toolkit/util/quantize.py:43-75 semantics: weight-only, bf16 arithmetic). */
int aitk_dequant_fp8(const uint8_t* q, int64_t ldq, const float* scale, int32_t mode, aitk_bf16* out, int64_t ldo, int32_t rows,
                     int32_t cols, aitk_stream_t stream);

/* ---- LoKr (Kronecker adapter; reference toolkit/models/lokr.py:331-399 _call_forward_fast_linear, factor shapes 136-188).
 * Per token m:   out_m[a_out x b_out] = scale * A[a_out x a_in] . X_m[a_in x b_in] . B[b_out x b_in]^T
 * X_m = row m of x viewed (a_in, b_in) row-major (the reference's x.unflatten(-1, (in_m, in_n))), out row m viewed
 * (a_out, b_out) row-major — (b_out, a_out) when transpose_out.  A / B are dense row-major bf16 with leading dimension a_in /
 * b_in; NULL = identity (then a_out == a_in / b_out == b_in).  forward delta: A = lokr_w1, B = lokr_w2; data gradient: the
 * transposed factors; one factor NULL: the intermediates of the factor gradients (ai-toolkit_amd/graph.py _lokr_grads).
 * Only columns [col0, col0 + ncols) of each output row are written, to out + row + (j - col0) (ncols == 0: all); accumulate: +=.
 * x / out rows may be segmented like AitkGemmArgs.A / .C.  b_in % 8 == b_out % 8 == 0; any a_in, a_out. */
typedef struct AitkKronApplyArgs {
  const aitk_bf16* x; int64_t ldx; int64_t x_seg_stride;
  const aitk_bf16* A; const aitk_bf16* B;
  aitk_bf16* out; int64_t ldo; int64_t out_seg_stride;
  int32_t x_seg_rows, out_seg_rows;
  int32_t M, a_in, b_in, a_out, b_out;
  int32_t transpose_out, accumulate, col0, ncols;
  float scale;
} AitkKronApplyArgs;
int aitk_kron_apply(const AitkKronApplyArgs* args, aitk_stream_t stream);
/* W[a_rows*b_rows, a_cols*b_cols] (bf16, leading dimension ldw) += alpha * kron(A[a_rows,a_cols], B[b_rows,b_cols]) (fp32 factors):
 * LokrModule.merge_in (toolkit/models/lokr.py:261-309) on the base weight (A = lokr_w1, B = lokr_w2) or on its transposed copy
 * (A = lokr_w1^T, B = lokr_w2^T).  b_cols % 8 == 0. */
int aitk_kron_merge(aitk_bf16* W, int64_t ldw, const float* A, const float* B, int32_t a_rows, int32_t a_cols, int32_t b_rows,
                    int32_t b_cols, float alpha, aitk_stream_t stream);

/* ---- DoRA (toolkit/models/DoRA.py, network_mixins.py:323-339): y = c * (x W^T + s m x A^T B^T) + b with
 * c_j = magnitude_j / ||W_j + s B_j A||, the norm detached.
 * aitk_dora_colscale: c from ||W_j||^2 (w2), tw = W A^T [N,R] (aitk_lora_down with the weight as streamed operand), up = B fp32
 *   [N,R] (arena view), gram = A A^T fp32 [R,R] (aitk_lora_wgrad), magnitude:  n^2 = w2 + 2 s B.tw + s^2 B gram B^T.
 * aitk_dora_bwd: dz = c * dy (bf16) and magnitude.grad_j += (sum_m dy*y - bias_j sum_m dy) / magnitude_j with y = this step's
 *   linear output; partial = 2 * ceil(M / aitk_rows_per_block()) * N floats of scratch. */
typedef struct AitkDoraColscaleArgs {
  const float* w2; const aitk_bf16* tw; int64_t ldtw; const float* up; const float* gram; const float* mag; float* c;
  float s; int32_t N, R, _pad;
} AitkDoraColscaleArgs;
int aitk_dora_colscale(const AitkDoraColscaleArgs* args, aitk_stream_t stream);
typedef struct AitkDoraBwdArgs {
  const aitk_bf16* dy; int64_t ld_dy; const aitk_bf16* y; int64_t ld_y;
  const float* c; const aitk_bf16* bias; const float* mag;
  aitk_bf16* dz; int64_t ld_dz; float* dmag; float* partial;
  int32_t M, N;
} AitkDoraBwdArgs;
int aitk_dora_bwd(const AitkDoraBwdArgs* args, aitk_stream_t stream);

/* ---- UNet2DConditionModel side kernels (SD1.5 / SDXL; NHWC bf16 activations [B*H*W, C]).  Replace the diffusers UNet body reached
 * from toolkit/stable_diffusion_model.py:2049-2055 (SDXL) / 2260-2265 (SD1.5) and its autograd backward.
 * GroupNorm(+SiLU) backward: dx = d/dx act(GroupNorm(x) * gamma + beta) . dy (+ dres); gamma / beta frozen.  stats = the forward's
 * stats_out; partial = aitk_groupnorm_bwd_workspace_bytes scratch. */
typedef struct AitkGroupNormBwdArgs {
  const aitk_bf16* dy; int64_t ld_dy;
  const aitk_bf16* x; int64_t ldx;
  const aitk_bf16* gamma; const aitk_bf16* beta;
  const float* stats;
  const aitk_bf16* dres; int64_t ld_dres; /* may be NULL */
  aitk_bf16* dx; int64_t ld_dx;
  float* partial; float* red; /* red: set by the library (inside partial) */
  int32_t silu; int32_t B, HW, C, G, _pad;
} AitkGroupNormBwdArgs;
int64_t aitk_groupnorm_bwd_workspace_bytes(int32_t B, int32_t HW, int32_t C, int32_t G);
int aitk_groupnorm_bwd(const AitkGroupNormBwdArgs* args, aitk_stream_t stream);
/* diffusers GEGLU (FeedForward of BasicTransformerBlock): hg [M, 2F] = [hidden | gate]; out = hidden * gelu_erf(gate); backward writes
 * d[hidden | gate] */
int aitk_geglu_fwd(const aitk_bf16* hg, int64_t ld_hg, aitk_bf16* out, int64_t ld_out, int64_t M, int32_t F, aitk_stream_t stream);
int aitk_geglu_bwd(const aitk_bf16* dy, int64_t ld_dy, const aitk_bf16* hg, int64_t ld_hg, aitk_bf16* dhg, int64_t ld_dhg, int64_t M,
                   int32_t F, aitk_stream_t stream);
/* 2x resampling of a contiguous NHWC tensor, (H, W) = SOURCE size.  mode 0: nearest up (Upsample2D forward) -> [B,2H,2W,C];
 * mode 1: 2x2 sum (its backward) -> [B,H/2,W/2,C]; mode 2: zero insertion -> [B,2H,2W,C] (data gradient of a stride-2 conv as a stride-1
 * conv with the rotated filter). */
int aitk_resample2x(const aitk_bf16* src, aitk_bf16* dst, int32_t B, int32_t H, int32_t W, int32_t C, int32_t mode, aitk_stream_t stream);
/* dst [B, H+2, W+2, C] = src [B, H, W, C] inside a one-pixel zero border (contiguous NHWC, C % 8 == 0): operands of the nine per-tap
 * aitk_lora_wgrad launches that form lora_down.weight.grad of a 3x3-conv adapter (toolkit/lora_special.py:95-104). */
int aitk_pad_nhwc(const aitk_bf16* src, aitk_bf16* dst, int32_t B, int32_t H, int32_t W, int32_t C, aitk_stream_t stream);
/* per-head column copy with zero fill: dst[m][h*d_dst + j] = j < d_src ? src[m][h*d_src + j] : 0 (j < d_dst) — pads head_dim 40 / 64 / 80 to
 * the flash kernels' 128 (exact: zero columns change neither q.k nor softmax) and drops the padding again. */
int aitk_copy_heads(const aitk_bf16* src, int64_t ld_src, aitk_bf16* dst, int64_t ld_dst, int64_t M, int32_t H, int32_t d_src, int32_t d_dst,
                    aitk_stream_t stream);
/* generic attention for head_dim > 128 (SD1.5: 160 at its two coarsest levels, <= 256 tokens at 512^2) — same AitkAttnArgs, D = head_dim (any
 * multiple of 8 up to 256), heads at column h*D, S and Skv <= 8192; LSE = natural-log log-sum-exp of the scaled scores.  Plain fp32
 * formulation, deterministic; aitk_attn_fwd / aitk_attn_bwd (head_dim 128, MFMA) stay the path for everything that fits them. */
int aitk_attn_small_fwd(const AitkAttnArgs* args, aitk_stream_t stream);
int aitk_attn_small_bwd(const AitkAttnArgs* args, aitk_stream_t stream);
/* DDPMScheduler.add_noise (toolkit/stable_diffusion_model.py:1854-1876) into the NHWC conv_in operand + the loss target:
 * noisy [B*HW, Cp] (channels >= C zero), target [B*HW, C] = eps (mode 0, SDTrainer.py:650) or velocity (mode 1, 623-625);
 * a[b] = sqrt(alphas_cumprod[t_b]), s[b] = sqrt(1 - alphas_cumprod[t_b]) as fp32 values already rounded to the latent dtype. */
typedef struct AitkDdpmNoiseArgs {
  const aitk_bf16* latents; const aitk_bf16* noise; /* NCHW [B, C, HW] */
  const float* a; const float* s;
  aitk_bf16* noisy; aitk_bf16* target;
  int32_t B, C, HW, Cp, mode, _pad;
} AitkDdpmNoiseArgs;
int aitk_ddpm_noise_nhwc(const AitkDdpmNoiseArgs* args, aitk_stream_t stream);

/* ---- hardware probes (test infrastructure for layout assumptions; not on the product path) ---- */
/* host-side evaluation of the attention kernels' workgroup -> (row tile, head, batch) map for a 1-D grid of n blocks (XCD-grouped order) */
int aitk_probe_attn_wg_coords(int32_t n, int32_t id, int32_t ntiles, int32_t H, int32_t* out3);
int aitk_probe_tr16(int16_t* out /*[64*4]*/, int32_t pitch_elems, aitk_stream_t stream);
int aitk_probe_glds(const int32_t* src /*[1024]*/, int32_t* out /*[1024]*/, aitk_stream_t stream);
int aitk_probe_mfma32(const aitk_bf16* a /*[32*16]*/, const aitk_bf16* b /*[16*32]*/, float* d /*[32*32]*/, aitk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* AITK_MI355_H */
