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

#define AITK_ABI_VERSION 1

/* ---- GEMM epilogue flags ---- */
#define AITK_EPI_BIAS 1      /* + bias[n]                                                        */
#define AITK_EPI_ACCUM 2     /* + C_old[m][n]  (sum of several dgrads into one dX)               */
#define AITK_EPI_GELU 4      /* aux_out = u (pre-activation, bf16); C = gelu_tanh(u)             */
#define AITK_EPI_DGELU 8     /* C = val * gelu_tanh'(aux_in)                                     */
#define AITK_EPI_GATE_RES 16 /* aux_out = y; C = aux_in(residual) + gate[m / gate_rows][n] * y   */

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
  int32_t stage_mode; /* 0 = VGPR-staged, 1 = LDS-DMA (global_load_lds) */
} AitkGemmArgs;

int aitk_abi_version(void);
int aitk_sizeof(int32_t which); /* 0: AitkGemmArgs — struct-size handshake for FFI mirrors */
int aitk_gemm_nt(const AitkGemmArgs* args, aitk_stream_t stream);

/* ---- hardware probes (test infrastructure for layout assumptions; not on the product path) ---- */
int aitk_probe_tr16(int16_t* out /*[64*4]*/, int32_t pitch_elems, aitk_stream_t stream);
int aitk_probe_glds(const int32_t* src /*[1024]*/, int32_t* out /*[1024]*/, aitk_stream_t stream);
int aitk_probe_mfma32(const aitk_bf16* a /*[32*16]*/, const aitk_bf16* b /*[16*32]*/, float* d /*[32*32]*/, aitk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* AITK_MI355_H */
