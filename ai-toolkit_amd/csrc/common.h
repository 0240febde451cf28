// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels of the LoRA train-step hot path.
// wave = 64 lanes everywhere; bf16 storage, fp32 accumulate.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 bits in memory
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // MFMA A/B operand (8 bf16 = 4 VGPR)
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;   // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;     // 16x16 MFMA accumulator

#define AITK_OK 0
#define AITK_ERR_SHAPE (-1)
#define AITK_ERR_ALIGN (-2)
#define AITK_ERR_ARG (-3)

#define AITK_LAUNCH_CHECK()                                  \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return (int)e__;                  \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even, NaN preserved (same rule as torch's float->bfloat16)
__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// two fp32 -> packed bf16x2 with one v_cvt_pk_bf16_f32 (hardware round-to-nearest-even)
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bfround(float f) { return bf2f(f2bf(f)); }
// the two halves of a packed bf16x2 word back as fp32 (exact)
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// D[i][j] += sum_k A[i][k] * B[k][j]; lane l supplies A[i=l&31][k=8*(l>>5)..+8] and B[k=8*(l>>5)..+8][j=l&31];
// lane l receives D[i=(r&3)+8*(r>>2)+4*(l>>5)][j=l&31] in register r (guide: cdna_hip_programming.md §3).
__device__ __forceinline__ f32x16_t mfma32(s16x8_t a, s16x8_t b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// 16x16x32: lane l supplies A[i=l&15][k=8*(l>>4)..+8], B[k=8*(l>>4)..+8][j=l&15]; receives D[i=4*(l>>4)+r][j=l&15].
__device__ __forceinline__ f32x4_t mfma16(s16x8_t a, s16x8_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// tanh-approximate GELU, torch.nn.functional.gelu(approximate="tanh"):  0.5 x (1 + tanh(u)),  u = k0 (x + k1 x^3).
// Written through the logistic function (tanh(u) = 2 sigmoid(2u) - 1) so it costs one v_exp_f32 + one v_rcp_f32 instead
// of libm tanhf (~60 instructions): the GELU / dGELU GEMM epilogues were spending ~50 us per 256x256 tile round on it.
// |error| ~1e-6 relative, far below the bf16 rounding of the stored result.
__device__ __forceinline__ float sigmoid_fast_f(float z) { return __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }
// GELU(tanh) and its derivative on PAIRS (the two bf16 halves of a packed word): every fp32 step is a packed instruction (v_pk_mul_f32 /
// v_pk_fma_f32 / v_pk_add_f32: two elements per issue slot), only v_exp_f32 and v_rcp_f32 stay per element — 3 + 2 issue slots per element for
// the forward (was 7 + 2 with the scalar chain the compiler formed) and 5.5 + 2 for the derivative (12.5 + 2): the GELU / dGELU GEMM epilogues
// are VALU-issue bound (profiles/r04_gemm8_tile_switch.md).  s = sigmoid(2u) = 1 / (1 + 2^(c x (1 + k1 x^2))), c = -2 k0 log2(e) folded.
// The operation ORDER below is the definition: every user (GEMM epilogues, the lora_wgrad GELU re-evaluation, the scalar wrappers) goes
// through these two functions, explicit fma only, so all of them produce the same bits.
__device__ __forceinline__ f32x2_t gelu_sigmoid_2(f32x2_t x, f32x2_t x2) {
  const float k1 = 0.044715f, c = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
  const f32x2_t p = __builtin_elementwise_fma(x2, f32x2_t{k1, k1}, f32x2_t{1.0f, 1.0f});
  const f32x2_t z = (x * p) * f32x2_t{c, c};
  const f32x2_t d = f32x2_t{__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)} + f32x2_t{1.0f, 1.0f};
  return f32x2_t{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}
__device__ __forceinline__ f32x2_t gelu_tanh_2(f32x2_t x) { return x * gelu_sigmoid_2(x, x * x); }
// d/dx = 0.5 (1 + t) + 0.5 x (1 - t^2) u'(x)  with t = 2s - 1:  s + (s x (1 - s)) (2 k0 + 6 k0 k1 x^2)
__device__ __forceinline__ f32x2_t gelu_tanh_grad_2(f32x2_t x) {
  const float c1 = 2.0f * 0.7978845608028654f, c2 = 6.0f * 0.7978845608028654f * 0.044715f;
  const f32x2_t x2 = x * x;
  const f32x2_t s = gelu_sigmoid_2(x, x2);
  const f32x2_t q = __builtin_elementwise_fma(x2, f32x2_t{c2, c2}, f32x2_t{c1, c1});
  const f32x2_t w = x * (f32x2_t{1.0f, 1.0f} - s);
  return __builtin_elementwise_fma(s * w, q, s);
}
__device__ __forceinline__ float gelu_tanh_f(float x) { return gelu_tanh_2(f32x2_t{x, x}).x; }
__device__ __forceinline__ float gelu_tanh_grad_f(float x) { return gelu_tanh_grad_2(f32x2_t{x, x}).x; }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// bijective XCD-aware remap of a linear block id: blocks that land on one XCD (bid % 8) get a contiguous
// chunk of the logical tile order so neighbouring tiles share that XCD's L2 (guide §5.5 T1).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int NX = 8;
  if (nwg < NX) return bid;
  int xcd = bid % NX, idx = bid / NX;
  int q = nwg / NX, r = nwg % NX;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
