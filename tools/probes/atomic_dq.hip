// What does accumulating dQ with fp32 atomics cost on MI355X?  ("fold dQ into the dK/dV pass ... fp32 atomics")
// Emulates exactly the atomic traffic a fused attention backward would emit at the FLUX shape (S = 4608, d = 128): a workgroup owns a
// KV block (KVB = 128 or 256 rows) of one (batch, head) and walks all 144 query sub-tiles of 32 rows; per sub-tile each of its 4 waves
// adds a 32 x 32 fp32 block of the head's dQ accumulator [S][128] (lane = column, 16 registers = rows: one atomic instruction covers two
// 128-B row segments).  Modes: relaxed agent-scope atomics (hipcc emits the same plain global_atomic_add_f32 for workgroup scope on gfx950:
// there is no cheaper "XCD-local" flavour to choose) and plain fp32 stores of the same addresses as the no-RMW floor.  Heads are dealt to
// XCDs like attn_wg_coords (contiguous ranges per XCD).
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_dq tools/probes/atomic_dq.hip ; run: /tmp/atomic_dq
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define S 4608
#define D 128

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* acc, int nkvb, int nheads, float v) {
  const int n = gridDim.x, id = blockIdx.x;
  const int xcd = id & 7, slot = id >> 3;
  const int per = n >> 3, rem = n & 7;
  const int l = xcd * per + (xcd < rem ? xcd : rem) + slot;
  const int head = l / nkvb;
  float* base = acc + (size_t)head * S * D;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  for (int sub = 0; sub < S / 32; ++sub) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      float* p = base + (size_t)row * D + 32 * wave + l31;
      if (MODE == 0) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (MODE == 1) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (MODE == 2) *p = v;
      else asm volatile("global_atomic_pk_add_bf16 %0, %1, off" ::"v"(p), "v"(0x3f803f80) : "memory");  // same addresses, packed bf16 pairs
    }
  }
}

template <int MODE>
static void run(const char* name, float* acc, int nheads, int kvb) {
  const int nkvb = S / kvb;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipMemset(acc, 0, (size_t)nheads * S * D * 4);
  hipLaunchKernelGGL(probe<MODE>, dim3(nkvb * nheads), dim3(256), 0, 0, acc, nkvb, nheads, 1.0f);
  hipDeviceSynchronize();
  hipMemset(acc, 0, (size_t)nheads * S * D * 4);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(nkvb * nheads), dim3(256), 0, 0, acc, nkvb, nheads, 1.0f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  float first = 0;
  hipMemcpy(&first, acc + 5 * D + 7, 4, hipMemcpyDeviceToHost);
  const double bytes = (double)nheads * nkvb * S * D * 4;
  printf("%-34s KV block %3d: %7.3f ms per layer (%d heads), %6.2f GB of fp32 adds -> %6.2f TB/s ; acc[5][7] = %.1f (expect %d)\n", name, kvb, ms, nheads,
         bytes / 1e9, bytes / ms / 1e9, first, MODE == 2 ? 1 : nkvb);
}

int main() {
  const int nheads = 168;  // B = 7 x 24 heads: one FLUX layer
  float* acc;
  hipMalloc(&acc, (size_t)nheads * S * D * 4);
  for (int kvb : {128, 256}) {
    run<0>("device-scope atomic add f32", acc, nheads, kvb);
    run<2>("plain stores (no RMW floor)", acc, nheads, kvb);
  }
  return 0;
}
