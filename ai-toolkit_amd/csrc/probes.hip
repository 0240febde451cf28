// Hardware-assumption probes (test infrastructure): they dump what gfx950 actually does for
//  (1) ds_read_b64_tr_b16 lane mapping, (2) global_load_lds destination order, (3) the 32x32x16 bf16 MFMA
//  operand / accumulator lane layout.  tests/ compare the dumps with the layouts the kernels assume.
#include "common.h"
#include "aitk_args.h"

__global__ void probe_tr16_kernel(int16_t* out, int pitch) {
  __shared__ __attribute__((aligned(16))) int16_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (int16_t)i;
  __syncthreads();
  const int lane = threadIdx.x;
  // 16-lane group g reads a 4(row) x 16(col) block starting at column 16*g... rows at `pitch` elements:
  // lane i in the group supplies the address of row (i>>2), cols (i&3)*4..+4.
  const int g = lane >> 4, i = lane & 15;
  const int16_t* src = lds + (i >> 2) * pitch + g * 16 + (i & 3) * 4;
  s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)src);
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}

__global__ void probe_glds_kernel(const int32_t* src, int32_t* out) {
  __shared__ __attribute__((aligned(16))) int32_t lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = -1;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // each wave DMA-copies 1 KiB: lane l sources 16 B at src + (wave*64 + (63-l))*4 ints (reversed on purpose),
  // destination = wave-uniform base; expected LDS image: lane-linear, i.e. reversed 16-B pieces per wave.
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (wave * 64 + (63 - lane)) * 4),
                                   (__attribute__((address_space(3))) void*)(lds + wave * 256), 16, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 256) out[i] = lds[i];
}

__global__ void probe_mfma32_kernel(const bf16_t* a, const bf16_t* b, float* d) {
  const int lane = threadIdx.x;
  s16x8_t af, bfr;
  for (int e = 0; e < 8; ++e) {
    af[e] = (short)a[(lane & 31) * 16 + 8 * (lane >> 5) + e];    // A[i][k], i = lane&31, k = 8*(lane>>5)+e
    bfr[e] = (short)b[(8 * (lane >> 5) + e) * 32 + (lane & 31)];  // B[k][j], j = lane&31
  }
  f32x16_t acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = mfma32(af, bfr, acc);
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    d[i * 32 + (lane & 31)] = acc[r];
  }
}

extern "C" int aitk_abi_version(void) { return AITK_ABI_VERSION; }

extern "C" int aitk_probe_tr16(int16_t* out, int32_t pitch_elems, aitk_stream_t stream) {
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, pitch_elems);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}
extern "C" int aitk_probe_glds(const int32_t* src, int32_t* out, aitk_stream_t stream) {
  hipLaunchKernelGGL(probe_glds_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, src, out);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}
extern "C" int aitk_probe_mfma32(const aitk_bf16* a, const aitk_bf16* b, float* d, aitk_stream_t stream) {
  hipLaunchKernelGGL(probe_mfma32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, d);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// struct-size handshake for the ctypes mirror (ai-toolkit_amd/_capi.py)
extern "C" int aitk_sizeof(int32_t which) {
  switch (which) {
    case 0: return (int)sizeof(AitkGemmArgs);
    case 1: return (int)sizeof(AitkLoraDownArgs);
    case 2: return (int)sizeof(AitkLoraWgradArgs);
    case 3: return (int)sizeof(AitkLnModArgs);
    case 4: return (int)sizeof(AitkLnModBwdArgs);
    case 5: return (int)sizeof(AitkGateBwdArgs);
    case 6: return (int)sizeof(AitkColsumFinishArgs);
    case 7: return (int)sizeof(AitkQkvPostArgs);
    case 8: return (int)sizeof(AitkEwArgs);
    case 9: return (int)sizeof(AitkAttnArgs);
    case 10: return (int)sizeof(AitkGemvArgs);
    case 11: return (int)sizeof(AitkNoisePackArgs);
    case 12: return (int)sizeof(AitkMseArgs);
    case 13: return (int)sizeof(AitkAdamWArgs);
    case 14: return (int)sizeof(AitkShadowDesc);
    case 15: return (int)sizeof(AitkGroupNormArgs);
    case 16: return (int)sizeof(AitkRmsFullArgs);
    case 17: return (int)sizeof(AitkDoraColscaleArgs);
    case 18: return (int)sizeof(AitkDoraBwdArgs);
    case 19: return (int)sizeof(AitkKronApplyArgs);
    case 20: return (int)sizeof(AitkGroupNormBwdArgs);
    case 21: return (int)sizeof(AitkDdpmNoiseArgs);
    case 22: return (int)sizeof(AitkQuantRowsArgs);
    case 23: return (int)sizeof(AitkWgradSrc2);
    default: return -1;
  }
}
