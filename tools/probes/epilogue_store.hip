// What bounds the store burst of the persistent GEMM's epilogue?  (gemm8.hip: every CU writes its 256 x 256 bf16 tile = 128 KB at the same moment,
// 16 global_store_dwordx4 per wave, each covering 16 rows x 64 B; the s_memtime trace of round 4 puts the bias-only epilogue at ~12,300 cycles = ~10.7 B
// per cycle and CU, the same rate whether the eight XCDs are staggered against each other or not.)
// This probe replays exactly that store pattern — no MFMA, no LDS — from a chosen subset of the 256 workgroups and stamps it with s_memtime:
//   who = 0: all 256 workgroups            1: the 32 workgroups of XCD 0 only (blockIdx % 8 == 0)
//         2: 4 workgroups on each XCD      3: one workgroup
//   gap  = idle cycles between two bursts (the K loop of the real kernel: ~120,000), so that every burst starts from a drained write path
//   stag = workgroup j of an XCD (j = blockIdx / 8) starts its bursts j * stag cycles late (0 = all together)
//   pat  = 0: the kernel's pattern (lane quad = one 64-B row segment, 16 rows per instruction)   1: 8 lanes = one 128-B row segment, 8 rows per instruction
//   mode = 0: stores only (bias / GELU epilogues)   1: the gate-residual epilogue's chain — per row group one 16-B load of the residual (a second matrix,
//          HBM), then, dependent on it, two stores (y and x_new), sixteen such round trips one after the other   2: the same bytes with all sixteen loads
//          issued first (64 registers) and the stores behind them
// Prints mean / max burst cycles over workgroups and rounds, and the bytes per cycle and CU that makes.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/epilogue_store tools/probes/epilogue_store.hip ; run: /tmp/epilogue_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define M 32256
#define ROUNDS 6

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memtime(); }
__device__ __forceinline__ void spin(unsigned long long cycles) {
  const unsigned long long t0 = now();
  while (now() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
}

__global__ __launch_bounds__(512) void probe(uint4* C, const uint4* R, uint4* Y, int N, int who, int gap, int stag, int pat, int mode, unsigned* out) {
  const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
  const bool active = who == 0 || (who == 1 && xcd == 0) || (who == 2 && j < 4) || (who == 3 && id == 0);
  if (!active) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3;
  const int tiles_m = M / 256, tiles_n = N / 256;
  const uint4 v = {(unsigned)tid, 1u, 2u, 3u};
  const long ld16 = N / 8;  // row stride in 16-B units
  if (stag) spin((unsigned long long)j * stag);
  for (int r = 0; r < ROUNDS; ++r) {
    const int t = (id + r * 256) % (tiles_m * tiles_n);
    // the kernel's raster: groups of 8 tile rows, column-major inside a group
    const int gsz = 8 * tiles_n, gid = t / gsz, first = gid * 8, gm = min(tiles_m - first, 8);
    const int m0 = (first + (t % gsz) % gm) * 256, n0 = ((t % gsz) / gm) * 256;
    __syncthreads();
    const unsigned long long t0 = now();
    if (mode == 1) {
      const int rr = lane >> 2, c4 = lane & 3;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int m = m0 + wr * 128 + mi * 32 + it * 16 + rr, n = n0 + wc * 64 + ni * 32 + c4 * 8;
            uint4 x = R[(long)m * ld16 + n / 8];
            x.x += v.x;  // consumed before the stores, as in the kernel (C may alias the residual)
            Y[(long)m * ld16 + n / 8] = v;
            C[(long)m * ld16 + n / 8] = x;
          }
    } else if (mode == 2) {
      const int rr = lane >> 2, c4 = lane & 3;
      uint4 x[16];
#pragma unroll
      for (int b = 0; b < 16; ++b) {
        const int m = m0 + wr * 128 + (b & 7) * 16 + rr, n = n0 + wc * 64 + (b >> 3) * 32 + c4 * 8;
        x[b] = R[(long)m * ld16 + n / 8];
      }
#pragma unroll
      for (int b = 0; b < 16; ++b) {
        const int m = m0 + wr * 128 + (b & 7) * 16 + rr, n = n0 + wc * 64 + (b >> 3) * 32 + c4 * 8;
        x[b].x += v.x;
        Y[(long)m * ld16 + n / 8] = v;
        C[(long)m * ld16 + n / 8] = x[b];
      }
    } else if (pat == 0) {
      const int rr = lane >> 2, c4 = lane & 3;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int m = m0 + wr * 128 + mi * 32 + it * 16 + rr, n = n0 + wc * 64 + ni * 32 + c4 * 8;
            C[(long)m * ld16 + n / 8] = v;
          }
    } else {
      const int rr = lane >> 3, c8 = lane & 7;
#pragma unroll
      for (int mi = 0; mi < 16; ++mi) {
        const int m = m0 + wr * 128 + mi * 8 + rr, n = n0 + wc * 64 + c8 * 8;
        C[(long)m * ld16 + n / 8] = v;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = now();
    if (tid == 0) out[id * ROUNDS + r] = (unsigned)(t1 - t0);
    if (gap) spin(gap);
  }
}

int main() {
  uint4 *C, *R, *Y;
  unsigned* out;
  const size_t bytes = (size_t)M * 12288 * 2;
  hipMalloc(&C, bytes);
  hipMalloc(&R, bytes);
  hipMalloc(&Y, bytes);
  hipMemset(R, 1, bytes);
  hipMalloc(&out, 256 * ROUNDS * sizeof(unsigned));
  std::vector<unsigned> h(256 * ROUNDS);
  const char* names[] = {"all 256 workgroups", "32 workgroups of XCD 0", "4 workgroups on each XCD", "one workgroup"};
  for (int N : {3072, 12288})
   for (int mode = 0; mode < 3; ++mode)
    for (int pat = 0; pat < 2; ++pat)
      for (int who = 0; who < 4; ++who)
        for (int gap : {0, 120000})
          for (int stag : {0, 400, 1200, 3600}) {
            if (stag && who >= 2) continue;
            if (stag && !gap) continue;
            if (mode && pat) continue;
            hipMemset(out, 0, 256 * ROUNDS * sizeof(unsigned));
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, C, R, Y, N, who, gap, stag, pat, mode, out);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), out, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
            double sum = 0;
            unsigned mx = 0, cnt = 0;
            for (int i = 0; i < 256; ++i)
              for (int r = 1; r < ROUNDS; ++r)  // round 0 starts cold
                if (h[i * ROUNDS + r]) {
                  sum += h[i * ROUNDS + r];
                  mx = h[i * ROUNDS + r] > mx ? h[i * ROUNDS + r] : mx;
                  ++cnt;
                }
            const double mean = cnt ? sum / cnt : 0;
            printf("N %5d mode %d pat %d  %-26s gap %6d stagger %4d : burst mean %7.0f max %7u cycles  -> %5.1f B / cycle / CU (mean)\n", N, mode, pat, names[who], gap, stag, mean, mx,
                   mean > 0 ? (mode ? 3 : 1) * 131072.0 / mean : 0.0);
          }
  return 0;
}
