// Can the matrix pipe of a SIMD run one wave's MFMAs while the vector ALU runs ANOTHER wave's (or the same wave's) fp32 / exp instructions?
// Workgroup = 512 threads = 8 waves, waves w and w + 4 share a SIMD; one workgroup per CU, 256 workgroups.
// Each wave picks ONE role before its loop (scalar branch), so an iteration is 8 MFMAs (256 matrix cycles) and / or 32 vector instructions
// and nothing else; the modes are listed in main().
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mvo tools/probes/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define NIT 4096

__device__ __forceinline__ void mfma_block(f32x16& a0, f32x16& a1, f32x16& a2, f32x16& a3, const s16x8& x, const s16x8& y) {
  // 8 MFMAs, 4 independent accumulators (no back-to-back dependency)
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
  }
}
__device__ __forceinline__ void fma_block(float (&v)[8], float c) {
  // 32 v_fma_f32, 8 independent chains
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
}
__device__ __forceinline__ void exp_block(float (&v)[8]) {
  // 32 v_exp_f32, 8 independent chains
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
}

template <int MODE>
__global__ __launch_bounds__(512) void probe(float* out, int nit) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool hi = wave >= 4;
  f32x16 a0, a1, a2, a3;
  for (int r = 0; r < 16; ++r) a0[r] = a1[r] = a2[r] = a3[r] = 0.f;
  s16x8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (short)(threadIdx.x + i); y[i] = (short)(threadIdx.x * 3 + i); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.001f * (threadIdx.x + i);
  const float c = 0.999f;
  // one role per wave, chosen once (scalar branch), one tight loop per role
  const int role = MODE == 0 ? 0 : MODE == 1 ? 1 : MODE == 2 ? 2 : MODE == 3 ? (hi ? 1 : 0) : MODE == 4 ? (hi ? 2 : 0) : MODE == 5 ? 3
                   : MODE == 6 ? (hi ? 1 : 4) : MODE == 7 ? 5 : MODE == 8 ? (hi ? 6 : 0) : MODE == 9 ? 7 : 8;
  if (role == 0) {
    for (int it = 0; it < nit; ++it) mfma_block(a0, a1, a2, a3, x, y);
  } else if (role == 4) {
    __builtin_amdgcn_s_setprio(2);
    for (int it = 0; it < nit; ++it) mfma_block(a0, a1, a2, a3, x, y);
  } else if (role == 1) {
    for (int it = 0; it < nit; ++it) fma_block(v, c);
  } else if (role == 2) {
    for (int it = 0; it < nit; ++it) exp_block(v);
  } else if (role == 3) {  // same wave: 8 MFMA then 32 fma per iteration, independent streams, compiler's order
    for (int it = 0; it < nit; ++it) {
      mfma_block(a0, a1, a2, a3, x, y);
      fma_block(v, c);
    }
  } else if (role == 5) {  // same wave: 1 MFMA : 4 fma, hand-interleaved
    for (int it = 0; it < nit; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
#pragma unroll
        for (int i = 4; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
#pragma unroll
        for (int i = 4; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
      }
    }
  } else if (role == 7) {  // 8 MFMAs on ONE accumulator: every product waits for the previous one (the Q K^T chain of the attention kernels)
    for (int it = 0; it < nit; ++it) {
#pragma unroll
      for (int r = 0; r < 8; ++r) a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
    }
  } else if (role == 8) {  // two accumulators alternating
    for (int it = 0; it < nit; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
      }
    }
  } else {  // role 6: 16 fma + 16 exp per iteration (the softmax mix)
    for (int it = 0; it < nit; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
      }
    }
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 123.456f) out[threadIdx.x] = s;
}

template <int MODE>
static float run(float* out, int nit) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, out, 64);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, out, nit);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  float* out;
  hipMalloc(&out, 4096);
  const char* names[] = {"all 8 waves: 8 MFMA / iteration", "all 8 waves: 32 v_fma / iteration", "all 8 waves: 32 v_exp / iteration",
                         "waves 0-3 MFMA | waves 4-7 v_fma", "waves 0-3 MFMA | waves 4-7 v_exp", "every wave: 8 MFMA + 32 v_fma, compiler order",
                         "waves 0-3 MFMA at s_setprio 2 | waves 4-7 v_fma", "every wave: 8 MFMA + 32 v_fma hand-interleaved 1 : 4",
                         "waves 0-3 MFMA | waves 4-7 16 v_fma + 16 v_exp", "all 8 waves: 8 MFMA on ONE accumulator (dependent chain)",
                         "all 8 waves: 8 MFMA on two alternating accumulators"};
  float ms[11] = {run<0>(out, NIT), run<1>(out, NIT), run<2>(out, NIT), run<3>(out, NIT), run<4>(out, NIT), run<5>(out, NIT), run<6>(out, NIT),
                  run<7>(out, NIT), run<8>(out, NIT), run<9>(out, NIT), run<10>(out, NIT)};
  for (int m = 0; m < 11; ++m) printf("mode %d  %-60s %8.3f ms = %7.1f ns / iteration\n", m, names[m], ms[m], ms[m] * 1e6 / NIT);
  return 0;
}
