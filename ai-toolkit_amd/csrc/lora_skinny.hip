// Rank-r LoRA side kernels (gfx950): the skinny projections and the weight gradients.
//
//   aitk_lora_down : T[M,R]  = bf16( c[m] * (X[M,K] * P[R,K]^T) ),  c[m] = scale * mult[m / rows_per_batch]
//        forward : X = layer input,  P = lora_down.weight (bf16 shadow)  -> T feeds the GEMM K-slab
//        backward: X = dY,           P = lora_up.weight^T (bf16 shadow)  -> dT = c*(dY*B) feeds dgrad K-slab and dA
//        (reference: lora_down -> lora_up -> *scale -> *multiplier, toolkit/network_mixins.py:197-239, 309-321)
//   aitk_lora_wgrad: out[r][l] += sum_m S[m][r] * G[m][l]   (fp32, arbitrary output strides)
//        dA[r][k] = sum_m dT[m][r] * X[m][k]     (S = dT, G = X)
//        dB[n][r] = sum_m dY[m][n] * T[m][r]     (S = T,  G = dY, transposed output strides)
//        = what autograd produces for lora_down.weight / lora_up.weight in the reference.
//
// Split precision (the adapter branch of the reference is fp32: toolkit/network_mixins.py:309, BaseSDTrainProcess.py:1982-1983):
// every fp32 adapter matrix is shadowed as hi = bf16(w), lo = bf16(w - hi) (|w - hi - lo| <= 2^-17 |w|).  lora_down contracts
// X against BOTH (P and P_lo accumulate into the same fp32 accumulator), and with split_rp > 0 writes the fp32 result as a
// bf16 pair in the K-slab layout the GEMM consumes: per rank block of split_rp columns [T_hi | T_lo | T_hi] (3*split_rp
// columns), to be multiplied with [B_hi | B_hi | B_lo] — T_hi*B_hi + T_lo*B_hi + T_hi*B_lo, the lo*lo term (2^-18) dropped.
// lora_wgrad reads S from the same layout (hi and lo into one accumulator).  The extra MFMAs ride under the X / dY stream.
//
// Both are HBM-bound (they stream X / dY once); MFMA is used only because the contraction is matmul-shaped.
// The contraction of wgrad runs over the ROW index of row-major tiles, so both operands are consumed through
// ds_read_b64_tr_b16 (hardware transpose read; lane mapping verified by aitk_probe_tr16).
#include <cstdlib>
#include "common.h"
#include "aitk_args.h"

__device__ __forceinline__ const bf16_t* seg_row2(const bf16_t* base, long ld, int seg_rows, long seg_stride, int m) {
  if (seg_rows > 0) {
    int s = m / seg_rows;
    return base + (long)s * seg_stride + (long)(m - s * seg_rows) * ld;
  }
  return base + (long)m * ld;
}

// four consecutive ranks rr..rr+3 of row m: plain bf16, or the [hi | lo | hi] K-slab triple of the rank block (split_rp > 0)
__device__ __forceinline__ void store_t4(const AitkLoraDownArgs& p, int m, int rr, const float vin[4]) {
  float v[4] = {vin[0], vin[1], vin[2], vin[3]};
  if (p.tmask) {  // dropout / rank-dropout mask (already scaled by 1 / keep-probability) on the rank-space activation
    const float* tm = p.tmask + (long)(p.tmask_rows_per_batch > 0 ? m / p.tmask_rows_per_batch : m) * p.R + rr;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= tm[e];
  }
  uint2 hi;
  hi.x = pack2bf(v[0], v[1]);
  hi.y = pack2bf(v[2], v[3]);
  bf16_t* row = p.T + (long)m * p.ldt;
  if (p.split_rp <= 0) {
    *reinterpret_cast<uint2*>(row + rr) = hi;
    return;
  }
  const int blk = rr / p.split_rp, j = rr - blk * p.split_rp;
  row += (long)blk * 3 * p.split_rp + j;
  uint2 lo;
  lo.x = pack2bf(v[0] - bf_lo(hi.x), v[1] - bf_hi(hi.x));
  lo.y = pack2bf(v[2] - bf_lo(hi.y), v[3] - bf_hi(hi.y));
  *reinterpret_cast<uint2*>(row) = hi;
  *reinterpret_cast<uint2*>(row + p.split_rp) = lo;
  *reinterpret_cast<uint2*>(row + 2 * p.split_rp) = hi;
}

// ------------------------------------------------------------------------------------------------------------
// lora_down: one workgroup = 32 rows; wave w contracts K-quarter w straight from global (fragment-shaped loads,
// no LDS in the main loop), partials combined through LDS.  RB = number of 32-wide rank blocks (R <= 32*RB).
// ------------------------------------------------------------------------------------------------------------
template <int RB>
__global__ __launch_bounds__(256) void lora_down_kernel(AitkLoraDownArgs p) {
  __shared__ __attribute__((aligned(16))) float red[4 * RB * 16 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * 32;
  const int mrow = min(m0 + l31, p.M - 1);
  const bf16_t* xrow = seg_row2(p.X, p.ldx, p.x_seg_rows, p.x_seg_stride, mrow);
  const bf16_t* prow[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) prow[rb] = p.P + (long)min(rb * 32 + l31, p.R - 1) * p.ldp;
  const long lo_off = p.P_lo ? (p.P_lo - p.P) : 0;  // same row pitch, element offset between the hi and lo matrices

  f32x16_t acc[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;

  // K split: wave w owns k-steps [w*ksteps/4, (w+1)*ksteps/4) of 16 elements each (K % 16 == 0)
  const int ksteps = p.K / 16;
  const int kbeg = (ksteps * wave) / 4, kend = (ksteps * (wave + 1)) / 4;
  int ks = kbeg;
  for (; ks + 4 <= kend; ks += 4) {
    s16x8_t xa[4];
    s16x8_t pa[4][RB];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = (ks + u) * 16 + 8 * h;
      xa[u] = *reinterpret_cast<const s16x8_t*>(xrow + k);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) pa[u][rb] = *reinterpret_cast<const s16x8_t*>(prow[rb] + k);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma32(pa[u][rb], xa[u], acc[rb]);  // D rows = r, cols = m
    if (lo_off) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = (ks + u) * 16 + 8 * h;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma32(*reinterpret_cast<const s16x8_t*>(prow[rb] + lo_off + k), xa[u], acc[rb]);
      }
    }
  }
  for (; ks < kend; ++ks) {
    const int k = ks * 16 + 8 * h;
    s16x8_t xa = *reinterpret_cast<const s16x8_t*>(xrow + k);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      s16x8_t pa = *reinterpret_cast<const s16x8_t*>(prow[rb] + k);
      acc[rb] = mfma32(pa, xa, acc[rb]);
      if (lo_off) acc[rb] = mfma32(*reinterpret_cast<const s16x8_t*>(prow[rb] + lo_off + k), xa, acc[rb]);
    }
  }
  // partials -> LDS [wave][rb][reg][lane]
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wave * RB + rb) * 16 + r) * 64 + lane] = acc[rb][r];
  __syncthreads();
  // wave w finishes register group g = w (4 consecutive ranks) for every rank block
  const int m = m0 + l31;
  float c = p.scale;
  if (p.mult) c *= p.mult[min(m, p.M - 1) / p.rows_per_batch];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * wave + e;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) s += red[((w * RB + rb) * 16 + r) * 64 + lane];
      v[e] = s * c;
    }
    const int rr = rb * 32 + 8 * wave + 4 * h;  // rank index of v[0]
    if (m < p.M && rr < p.R) store_t4(p, m, rr, v);
  }
}

// ------------------------------------------------------------------------------------------------------------
// lora_down, K % 32 == 0 (every shape of the hot path): v_mfma_f32_16x16x32_bf16 so one P fragment (16 ranks x 32 k) serves
// TWO 16-row blocks of X — a third of the load instructions fetch the (L2-resident, WG-redundant) projection instead of half —
// each lane's loads are 16 B of a 64-B row segment, and 8 k-steps (24+ loads per lane) are in flight before the first MFMA.
// One workgroup = 32 rows, wave w contracts K-quarter w; partials combined through LDS in a fixed order.
// RB = number of 16-wide rank blocks (R <= 16*RB <= 64).
// ------------------------------------------------------------------------------------------------------------
// U = k-steps (of 32) whose loads are in flight before the first MFMA.  RB = 1 (one rank-16 adapter, 95 % of the launches of a FLUX
// step) runs with U = 6: 116 VGPRs -> 4 waves per SIMD -> all 1008 workgroups of a 32256-row launch are resident at once (U = 8 needs
// 132 VGPRs -> 3 per SIMD -> 768 slots -> a second, one-third-full round).
// RAW (aitk_lora_down_raw): the un-scaled fp32 sums go to raw[m][r] instead of T — one more tile of a partial-sum slab that aitk_lora_t_finish turns into T.
// NW = waves per workgroup = K slices: 4, or 8 for SHORT launches (below 16384 rows: B <= 3 at 1024^2) — there the chip is not full (144 workgroups at
// B = 1) and a launch is a latency chain of K / (NW * 32 * U) load batches per wave: eight waves halve the chain (the sum over K slices has another order:
// equal to the 4-wave kernel to fp32 rounding, chosen by row count only, so a sample's result does not depend on the other samples of a short batch);
// 16 for launches of ONE workgroup (M <= 32 rows: the adaLN adapters' B x 18432 backward operand was a 24-batch chain on four waves).
// NB = 16-row blocks per workgroup (2, or 4 for LONG launches of wide rank groups): every workgroup re-reads the whole projection (hi + lo) through its L1, and that
// L2 -> L1 traffic is what a launch costs beside its X stream (tools/gpu_lora_down_bench.py: 32 us + 11 us per 16-rank half at M = 32256, K = 3072 — 120 us for the
// 64-rank q,k,v,proj_mlp group); 64-row workgroups halve it.  Each (row, rank) sum keeps its own order of contributions for a given U.
template <int RB, int U, bool RAW = false, int NW = 4, int NB = 2>
__global__ __launch_bounds__(64 * NW, (RB == 1 && U <= 6 && NW == 4 && NB == 2) ? 4 : 1) void lora_down16_kernel(AitkLoraDownArgs p, float* raw = nullptr) {
  __shared__ __attribute__((aligned(16))) float red[NW * RB * NB * 4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * (16 * NB);
  const bf16_t* xrow[NB];
#pragma unroll
  for (int blk = 0; blk < NB; ++blk)
    xrow[blk] = seg_row2(p.X, p.ldx, p.x_seg_rows, p.x_seg_stride, min(m0 + blk * 16 + i16, p.M - 1)) + 8 * g;
  const bf16_t* prow[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) prow[rb] = p.P + (long)min(rb * 16 + i16, p.R - 1) * p.ldp + 8 * g;
  const long lo_off = p.P_lo ? (p.P_lo - p.P) : 0;
  f32x4_t acc[RB][NB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) acc[rb][blk] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // gridDim.y > 1 (aitk_lora_down_ksplit, RAW only): workgroup (x, y) contracts the y-th slice of K for its 32 rows into raw tile y
  const int ksteps_all = p.K / 32;
  const int sbeg = (int)(((long)ksteps_all * blockIdx.y) / gridDim.y), ksteps = (int)(((long)ksteps_all * (blockIdx.y + 1)) / gridDim.y) - sbeg;
  const int kbeg = sbeg + (ksteps * wave) / NW, kend = sbeg + (ksteps * (wave + 1)) / NW;
  if constexpr (RAW) raw += (long)blockIdx.y * p.M * p.R;
  int ks = kbeg;
  for (; ks + U <= kend; ks += U) {
    s16x8_t xa[U][NB], pa[U][RB];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = (ks + u) * 32;
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) xa[u][blk] = *reinterpret_cast<const s16x8_t*>(xrow[blk] + k);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) pa[u][rb] = *reinterpret_cast<const s16x8_t*>(prow[rb] + k);
    }
    __builtin_amdgcn_sched_barrier(0);  // all loads of the batch are issued before the first MFMA waits on one
    if (lo_off) {  // the lo halves of the (L2-resident) projection: loaded behind the X loads, contracted with the same X fragments
      s16x8_t pl[U][RB];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) pl[u][rb] = *reinterpret_cast<const s16x8_t*>(prow[rb] + lo_off + (ks + u) * 32);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int blk = 0; blk < NB; ++blk) acc[rb][blk] = mfma16(pa[u][rb], xa[u][blk], acc[rb][blk]);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int blk = 0; blk < NB; ++blk) acc[rb][blk] = mfma16(pl[u][rb], xa[u][blk], acc[rb][blk]);
      continue;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) acc[rb][blk] = mfma16(pa[u][rb], xa[u][blk], acc[rb][blk]);  // D rows = rank, cols = x row
  }
  for (; ks < kend; ++ks) {
    const int k = ks * 32;
    s16x8_t xt[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) xt[blk] = *reinterpret_cast<const s16x8_t*>(xrow[blk] + k);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const s16x8_t pf = *reinterpret_cast<const s16x8_t*>(prow[rb] + k);
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) acc[rb][blk] = mfma16(pf, xt[blk], acc[rb][blk]);
      if (lo_off) {
        const s16x8_t pl = *reinterpret_cast<const s16x8_t*>(prow[rb] + lo_off + k);
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) acc[rb][blk] = mfma16(pl, xt[blk], acc[rb][blk]);
      }
    }
  }
  // partials -> LDS [wave][rb][blk][reg][lane]; wave w then finishes the (rb, blk) pairs with (rb*2+blk) % 4 == w
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(((wave * RB + rb) * NB + blk) * 4 + r) * 64 + lane] = acc[rb][blk][r];
  __syncthreads();
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      if (((rb * NB + blk) & 3) != wave) continue;
      const int m = m0 + blk * 16 + i16;
      float c = p.scale;
      if (p.mult) c *= p.mult[min(m, p.M - 1) / p.rows_per_batch];
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += red[(((w * RB + rb) * NB + blk) * 4 + r) * 64 + lane];
        v[r] = RAW ? s : s * c;
      }
      const int rr = rb * 16 + 4 * g;  // lane holds ranks rr..rr+3 of row m (mfma16 D layout: row 4*(l>>4)+reg, col l&15)
      if constexpr (RAW) {
        if (m < p.M && rr < p.R) *reinterpret_cast<f32x4_t*>(raw + (long)m * p.R + rr) = f32x4_t{v[0], v[1], v[2], v[3]};
      } else {
        if (m < p.M && rr < p.R) store_t4(p, m, rr, v);
      }
    }
}

extern "C" int aitk_lora_down(const AitkLoraDownArgs* a, aitk_stream_t stream) {
  if (!a || a->M <= 0 || a->K <= 0 || a->R <= 0) return AITK_ERR_SHAPE;
  if ((a->K % 16) || (a->R % 4) || a->R > 64) return AITK_ERR_SHAPE;
  if ((a->ldx % 8) || (a->ldp % 8) || (a->ldt % 4)) return AITK_ERR_ALIGN;
  if (a->mult && a->rows_per_batch <= 0) return AITK_ERR_ARG;
  // split_rp >= R: the launch covers (a chunk of) ONE rank block — ranks above 64 go out in 64-rank chunks of the same slab
  if (a->split_rp < 0 || (a->split_rp > 0 && ((a->split_rp % 4) || (a->split_rp < a->R && (a->R % a->split_rp))))) return AITK_ERR_ARG;
  // tmask rows are R wide = the ranks of THIS launch: a 64-rank chunk of a wider slab (split_rp > R) brings its own contiguous [rows, R] mask
  const int grid = (a->M + 31) / 32;
  if (a->K % 32 == 0) {
    static int u1 = 0;  // AITK_LORA_DOWN_U=8 selects the 3-waves-per-SIMD variant (A/B measurements)
    if (!u1) {
      const char* e = getenv("AITK_LORA_DOWN_U");
      u1 = (e && atoi(e) == 8) ? 8 : 6;
    }
    static int short8 = -1;  // AITK_LORA_DOWN_SHORT8=0: the 4-wave kernel for short launches too (A/B)
    if (short8 < 0) {
      const char* e = getenv("AITK_LORA_DOWN_SHORT8");
      short8 = (e && atoi(e) == 0) ? 0 : 1;
    }
    // LONG launches (>= 16384 rows: B >= 4 at 1024^2): 64-row workgroups for the rank widths in the mask AITK_LORA_DOWN_NB4 (bit 0: <= 16 ranks, 1: <= 32, 2: <= 48,
    // 3: <= 64) — half the projection traffic per row (kernel comment)
    static int short8w = -1;
    if (short8w < 0) {
      const char* e = getenv("AITK_LORA_DOWN_SHORT8W");
      short8w = (e && atoi(e) == 0) ? 0 : 1;
    }
    static int nb4 = -1;
    if (nb4 < 0) {
      const char* e = getenv("AITK_LORA_DOWN_NB4");
      nb4 = e ? atoi(e) : 15;
    }
    const int grid64 = (a->M + 63) / 64;
    const bool lng = a->M >= 16384;
    if (lng && a->R > 48 && (nb4 & 8)) hipLaunchKernelGGL((lora_down16_kernel<4, 2, false, 4, 4>), dim3(grid64), dim3(256), 0, (hipStream_t)stream, *a);
    else if (lng && a->R > 32 && a->R <= 48 && (nb4 & 4)) hipLaunchKernelGGL((lora_down16_kernel<3, 2, false, 4, 4>), dim3(grid64), dim3(256), 0, (hipStream_t)stream, *a);
    else if (lng && a->R > 16 && a->R <= 32 && (nb4 & 2)) hipLaunchKernelGGL((lora_down16_kernel<2, 4, false, 4, 4>), dim3(grid64), dim3(256), 0, (hipStream_t)stream, *a);
    else if (lng && a->R <= 16 && (nb4 & 1)) hipLaunchKernelGGL((lora_down16_kernel<1, 6, false, 4, 4>), dim3(grid64), dim3(256), 0, (hipStream_t)stream, *a);
    // one workgroup for the whole launch (M <= 32: the adaLN adapters, B rows x K = 3 d / 6 d in their backward): 16 K slices
    else if (a->R <= 16 && short8 && a->M <= 32 && a->K >= 32 * 6 * 16) hipLaunchKernelGGL((lora_down16_kernel<1, 6, false, 16>), dim3(grid), dim3(1024), 0, (hipStream_t)stream, *a);
    else if (a->R <= 16 && short8 && a->M < 16384 && a->K >= 32 * 6 * 8) hipLaunchKernelGGL((lora_down16_kernel<1, 6, false, 8>), dim3(grid), dim3(512), 0, (hipStream_t)stream, *a);
    // the wider rank groups of a short launch (q,k,v[,proj_mlp] at B <= 3): eight K slices too (AITK_LORA_DOWN_SHORT8W=0: four, A/B)
    else if (short8w && a->M < 16384 && a->K >= 32 * 4 * 8 && a->R > 48) hipLaunchKernelGGL((lora_down16_kernel<4, 4, false, 8>), dim3(grid), dim3(512), 0, (hipStream_t)stream, *a);
    else if (short8w && a->M < 16384 && a->K >= 32 * 4 * 8 && a->R > 32) hipLaunchKernelGGL((lora_down16_kernel<3, 4, false, 8>), dim3(grid), dim3(512), 0, (hipStream_t)stream, *a);
    else if (short8w && a->M < 16384 && a->K >= 32 * 8 * 8 && a->R > 16) hipLaunchKernelGGL((lora_down16_kernel<2, 8, false, 8>), dim3(grid), dim3(512), 0, (hipStream_t)stream, *a);
    else if (a->R <= 16 && u1 == 6) hipLaunchKernelGGL((lora_down16_kernel<1, 6>), dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
    else if (a->R <= 16) hipLaunchKernelGGL((lora_down16_kernel<1, 8>), dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
    else if (a->R <= 32) hipLaunchKernelGGL((lora_down16_kernel<2, 8>), dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
    else if (a->R <= 48) hipLaunchKernelGGL((lora_down16_kernel<3, 4>), dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL((lora_down16_kernel<4, 4>), dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  } else if (a->R <= 32)
    hipLaunchKernelGGL(lora_down_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  else
    hipLaunchKernelGGL(lora_down_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

extern "C" int aitk_lora_down_raw(const AitkLoraDownArgs* a, float* raw, aitk_stream_t stream) {
  if (!a || !raw || a->M <= 0 || a->K <= 0 || (a->R != 16 && a->R != 32) || (a->K % 32)) return AITK_ERR_SHAPE;
  if ((a->ldx % 8) || (a->ldp % 8) || ((uintptr_t)raw & 15)) return AITK_ERR_ALIGN;
  if (!a->X || !a->P) return AITK_ERR_ARG;
  if (a->R == 16 && a->M >= 16384) hipLaunchKernelGGL((lora_down16_kernel<1, 6, true, 4, 4>), dim3((a->M + 63) / 64), dim3(256), 0, (hipStream_t)stream, *a, raw);  // 64-row workgroups (kernel comment)
  else if (a->R == 16) hipLaunchKernelGGL((lora_down16_kernel<1, 6, true>), dim3((a->M + 31) / 32), dim3(256), 0, (hipStream_t)stream, *a, raw);
  else hipLaunchKernelGGL((lora_down16_kernel<2, 8, true>), dim3((a->M + 31) / 32), dim3(256), 0, (hipStream_t)stream, *a, raw);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// ------------------------------------------------------------------------------------------------------------
// slab_rescale: T[m] = split( (T_hi + T_lo)[m][r] * rowf[m / rows_per_batch] * tmask[m / tmask_rows_per_batch][r] ) in place on a
// [hi(rp) | lo(rp) | hi(rp)] slab.  The rank-space activation of a 3x3-conv adapter leaves the implicit-GEMM epilogue with a uniform scale;
// per-sample multipliers (network.multiplier as a list) and the dropout / rank_dropout masks (toolkit/network_mixins.py:211-229) are row /
// element factors on lx, applied here on the fp32 value before it is split again (aitk_lora_down does the same inside its epilogue).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void slab_rescale_kernel(bf16_t* T, long ldt, int M, int rp, const float* rowf, int rows_per_batch, const float* tmask,
                                                           int tmask_rows_per_batch) {
  const int per_row = rp / 4;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)M * per_row) return;
  const int m = (int)(idx / per_row), rr = (int)(idx - (long)m * per_row) * 4;
  bf16_t* row = T + (long)m * ldt + rr;
  const uint2 hi = *reinterpret_cast<const uint2*>(row), lo = *reinterpret_cast<const uint2*>(row + rp);
  float v[4] = {bf_lo(hi.x) + bf_lo(lo.x), bf_hi(hi.x) + bf_hi(lo.x), bf_lo(hi.y) + bf_lo(lo.y), bf_hi(hi.y) + bf_hi(lo.y)};
  const float f = rowf ? rowf[m / rows_per_batch] : 1.f;
  const float* tm = tmask ? tmask + (long)(tmask_rows_per_batch > 0 ? m / tmask_rows_per_batch : m) * rp + rr : nullptr;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] *= tm ? f * tm[e] : f;
  uint2 h2, l2;
  h2.x = pack2bf(v[0], v[1]);
  h2.y = pack2bf(v[2], v[3]);
  l2.x = pack2bf(v[0] - bf_lo(h2.x), v[1] - bf_hi(h2.x));
  l2.y = pack2bf(v[2] - bf_lo(h2.y), v[3] - bf_hi(h2.y));
  *reinterpret_cast<uint2*>(row) = h2;
  *reinterpret_cast<uint2*>(row + rp) = l2;
  *reinterpret_cast<uint2*>(row + 2 * rp) = h2;
}

extern "C" int aitk_slab_rescale(aitk_bf16* T, int64_t ldt, int32_t M, int32_t rp, const float* rowf, int32_t rows_per_batch, const float* tmask,
                                 int32_t tmask_rows_per_batch, aitk_stream_t stream) {
  if (!T || M <= 0 || rp <= 0 || (rp % 4) || ldt < 3 * rp || (ldt % 4)) return AITK_ERR_SHAPE;
  if ((rowf && rows_per_batch <= 0) || tmask_rows_per_batch < 0 || (!rowf && !tmask)) return AITK_ERR_ARG;
  const long n = (long)M * (rp / 4);
  hipLaunchKernelGGL(slab_rescale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)T, (long)ldt, M, rp, rowf,
                     rows_per_batch, tmask, tmask_rows_per_batch);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// ------------------------------------------------------------------------------------------------------------
// lora_wgrad: grid = (L/128 column tiles, row chunks of WG_MC rows).  Per 64-row sub-tile the block stages
// G[64][128] and S[64][R] row-major into LDS and runs v_mfma_f32_16x16x32_bf16 with BOTH operands read through
// ds_read_b64_tr_b16 (the contraction index is the tile row).  Chunk partials go to `partial`
// [nchunks][R][L] fp32; aitk_lora_wgrad_finish adds them (deterministic order) into the gradient arena.
// ------------------------------------------------------------------------------------------------------------
#define WG_MC 256  /* minimum rows per chunk (workspace sizing); big problems use 512 (fewer partials to reduce) */
#define WG_LT 128
#define WG_GPITCH 144  // elements (288 B): 4 consecutive rows land on disjoint bank octets for the tr reads
#define WG_SPITCH 72   // elements (144 B), R <= 64

__device__ __forceinline__ s16x4_t tr16(const bf16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
}
// 8 contraction rows x 16 columns fragment for mfma16: lane l (g = l>>4, i = l&15) gets rows row0+8g+0..7 of
// column col0+i from a row-major LDS tile with `pitch` elements per row.
__device__ __forceinline__ s16x8_t load_frag_tr(const bf16_t* tile, int pitch, int row0, int col0, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const bf16_t* p = tile + (row0 + 8 * g + (i >> 2)) * pitch + col0 + (i & 3) * 4;
  s16x4_t lo = tr16(p);
  s16x4_t hi = tr16(p + 4 * pitch);
  s16x8_t f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

struct WgradFuse {  // operands of the fused dT partial (lora_wgrad_fused_kernel)
  const bf16_t* P; const bf16_t* P_lo; long ldp; float* dt_partial;
};

template <int RB16>
__global__ __launch_bounds__(256) void lora_wgrad_kernel(AitkLoraWgradArgs p, int mc) {
#define WG_SECOND 0
#define WG_FUSE 0
#include "lora_wgrad_body.inc"
#undef WG_FUSE
#undef WG_SECOND
}
template <int RB16>
__global__ __launch_bounds__(256) void lora_wgrad2_kernel(AitkLoraWgradArgs p, AitkWgradSrc2 s2, int mc) {
#define WG_SECOND 1
#define WG_FUSE 0
#include "lora_wgrad_body.inc"
#undef WG_FUSE
#undef WG_SECOND
}
// dB = dY^T T AND the column-tile partials of dT = dY (P + P_lo)^T from ONE pass over dY (aitk_lora_bwd_fused)
template <int RB16>
__global__ __launch_bounds__(256) void lora_wgrad_fused_kernel(AitkLoraWgradArgs p, WgradFuse fz, int mc) {
#define WG_SECOND 0
#define WG_FUSE 1
#include "lora_wgrad_body.inc"
#undef WG_FUSE
#undef WG_SECOND
}

// Second form of the fused pass: one workgroup walks CT consecutive 128-column tiles of its row chunk, so the dT partial of a 64-row sub-tile is
// accumulated over CT * 128 columns in registers before it is written: the partial traffic ([column group][M][R] fp32, written and read back by
// the finish pass) shrinks by CT — at 128 columns per partial it is HALF the bytes of dY itself at rank 16 (measured: lora_dt_finish 7.5 ms per
// step).  dB accumulators: one set per column tile (CT * RB16 * 8 registers); the tile's P / P_lo slices sit in LDS for the whole row chunk.
// Same register-prefetched staging as the wgrad body, linearised over (sub-tile, column tile).  dB stays bit-identical to aitk_lora_wgrad.
template <int RB16, int CT>
__global__ __launch_bounds__(256) void lora_bwd_fused_ct_kernel(AitkLoraWgradArgs p, WgradFuse fz, int mc) {
  constexpr int PP = CT * WG_LT + 8;  // LDS pitch of the P slices (elements): + 16 B per row spreads the rows over the banks
  __shared__ __attribute__((aligned(16))) bf16_t gt[64 * WG_GPITCH];
  __shared__ __attribute__((aligned(16))) bf16_t st[2 * 64 * WG_SPITCH];
  __shared__ __attribute__((aligned(16))) bf16_t pt[2 * RB16 * 16 * PP];  // [hi | lo][rank][CT * 128 columns]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g4 = lane >> 4;
  const int lbase = blockIdx.x * (CT * WG_LT);
  const bool split = p.split_rp > 0;
  const int mbeg = blockIdx.y * mc;
  const int mend = min(p.M, mbeg + mc);
  constexpr int R = RB16 * 16;

  f32x4_t acc[CT][RB16][2];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int a = 0; a < RB16; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[c][a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  {  // P / P_lo slices of this column group -> LDS (zero beyond L; P_lo may be absent)
    constexpr int CW8 = CT * WG_LT / 8;
    for (int q = tid; q < 2 * R * CW8; q += 256) {
      const int half = q / (R * CW8), rem = q - half * (R * CW8);
      const int r = rem / CW8, c8 = rem - r * CW8;
      const int col = lbase + c8 * 8;
      const bf16_t* src = half ? fz.P_lo : fz.P;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (src && col < p.L) v = *reinterpret_cast<const uint4*>(src + (long)r * fz.ldp + col);
      *reinterpret_cast<uint4*>(pt + (half * R + r) * PP + c8 * 8) = v;
    }
  }
  const int chunks_per_row = R / 8;
  uint4 rg[4], rs[2], rl[2];
  auto load_regs = [&](int ms, int l0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + 256 * i;
      const int row = q >> 4, ch = q & 15;
      const int m = ms + row, col = l0 + ch * 8;
      rg[i] = make_uint4(0, 0, 0, 0);
      if (m < mend && col < p.L) rg[i] = *reinterpret_cast<const uint4*>(seg_row2(p.G, p.ldg, p.g_seg_rows, p.g_seg_stride, m) + col);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = tid + 256 * i;
      const int row = q / chunks_per_row, ch = q - row * chunks_per_row;
      rs[i] = make_uint4(0, 0, 0, 0);
      rl[i] = make_uint4(0, 0, 0, 0);
      if (q < 64 * chunks_per_row && ms + row < mend) {
        const int r = ch * 8, blk = split ? r / p.split_rp : 0;
        const bf16_t* src = p.S + (long)(ms + row) * p.lds + r + 2 * blk * p.split_rp;
        rs[i] = *reinterpret_cast<const uint4*>(src);
        if (split) rl[i] = *reinterpret_cast<const uint4*>(src + p.split_rp);
      }
    }
  };
  auto write_lds = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + 256 * i;
      *reinterpret_cast<uint4*>(gt + (q >> 4) * WG_GPITCH + (q & 15) * 8) = rg[i];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = tid + 256 * i;
      const int row = q / chunks_per_row, ch = q - row * chunks_per_row;
      if (q < 64 * chunks_per_row) {
        *reinterpret_cast<uint4*>(st + row * WG_SPITCH + ch * 8) = rs[i];
        if (split) *reinterpret_cast<uint4*>(st + 64 * WG_SPITCH + row * WG_SPITCH + ch * 8) = rl[i];
      }
    }
  };
  load_regs(mbeg, lbase);
  for (int ms = mbeg; ms < mend; ms += 64) {
    f32x4_t dt[RB16];
#pragma unroll
    for (int rb = 0; rb < RB16; ++rb) dt[rb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      write_lds();
      __syncthreads();  // also publishes pt on the first pass
      if (ct + 1 < CT) load_regs(ms, lbase + (ct + 1) * WG_LT);
      else if (ms + 64 < mend) load_regs(ms + 64, lbase);
      // dT partial over this tile's 128 columns: D rows = ranks 4 g .. 4 g + 3, column = dY row i
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const s16x8_t yf = *reinterpret_cast<const s16x8_t*>(gt + (wave * 16 + i16) * WG_GPITCH + k4 * 32 + 8 * g4);
#pragma unroll
        for (int rb = 0; rb < RB16; ++rb) {
          const bf16_t* pr = pt + (rb * 16 + i16) * PP + ct * WG_LT + k4 * 32 + 8 * g4;
          dt[rb] = mfma16(*reinterpret_cast<const s16x8_t*>(pr), yf, dt[rb]);
          dt[rb] = mfma16(*reinterpret_cast<const s16x8_t*>(pr + R * PP), yf, dt[rb]);
        }
      }
      // dB of this column tile (the wgrad body's product, same operand order)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        s16x8_t bfr[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) bfr[cb] = load_frag_tr(gt, WG_GPITCH, kk * 32, (wave * 2 + cb) * 16, lane);
#pragma unroll
        for (int rb = 0; rb < RB16; ++rb) {
          s16x8_t af = load_frag_tr(st, WG_SPITCH, kk * 32, rb * 16, lane);
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) acc[ct][rb][cb] = mfma16(af, bfr[cb], acc[ct][rb][cb]);
          if (split) {
            s16x8_t al = load_frag_tr(st + 64 * WG_SPITCH, WG_SPITCH, kk * 32, rb * 16, lane);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[ct][rb][cb] = mfma16(al, bfr[cb], acc[ct][rb][cb]);
          }
        }
      }
      __syncthreads();
    }
    const int m = ms + wave * 16 + i16;
    if (m < mend) {
      float* dst = fz.dt_partial + ((long)blockIdx.x * p.M + m) * p.R + 4 * g4;
#pragma unroll
      for (int rb = 0; rb < RB16; ++rb) *reinterpret_cast<f32x4_t*>(dst + rb * 16) = dt[rb];
    }
  }
  float* part = p.partial + (long)blockIdx.y * p.R * p.L;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int rb = 0; rb < RB16; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int col = lbase + ct * WG_LT + (wave * 2 + cb) * 16 + i16;
        if (col < p.L) {
#pragma unroll
          for (int r = 0; r < 4; ++r) part[(long)(rb * 16 + 4 * g4 + r) * p.L + col] = acc[ct][rb][cb][r];
        }
      }
}

// dT[m][r..r+3] = c[m] * sum over the column tiles (fixed order: deterministic) of the partials, written like aitk_lora_down writes it
// (bf16, or the [hi | lo | hi] K-slab of the rank block; dropout mask applied on the fp32 value)
__device__ __forceinline__ void lora_dt_finish_body(const AitkLoraDownArgs& p, const float* part, int ntiles, long block) {
  const int per_row = p.R / 4;
  const long idx = block * 256 + threadIdx.x;
  if (idx >= (long)p.M * per_row) return;
  const int m = (int)(idx / per_row), rr = (int)(idx - (long)m * per_row) * 4;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  const float* src = part + (long)m * p.R + rr;
  const long tstride = (long)p.M * p.R;
  int t = 0;
  for (; t + 8 <= ntiles; t += 8) {  // eight tile loads in flight, added in tile order (the sum is the one-at-a-time loop's, bit for bit): the GELU-emitted T has 49 tiles
    f32x4_t q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) q[u] = *reinterpret_cast<const f32x4_t*>(src + (long)(t + u) * tstride);
#pragma unroll
    for (int u = 0; u < 8; ++u) { v[0] += q[u][0]; v[1] += q[u][1]; v[2] += q[u][2]; v[3] += q[u][3]; }
  }
  for (; t < ntiles; ++t) {
    const f32x4_t q = *reinterpret_cast<const f32x4_t*>(src + (long)t * tstride);
    v[0] += q[0]; v[1] += q[1]; v[2] += q[2]; v[3] += q[3];
  }
  float c = p.scale;
  if (p.mult) c *= p.mult[m / p.rows_per_batch];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] *= c;
  store_t4(p, m, rr, v);
}
__global__ __launch_bounds__(256) void lora_dt_finish_kernel(AitkLoraDownArgs p, const float* part, int ntiles) {
  lora_dt_finish_body(p, part, ntiles, (long)blockIdx.x);
}

extern "C" int aitk_lora_t_finish(const AitkLoraDownArgs* a, const float* partial, int32_t ntiles, aitk_stream_t stream) {
  if (!a || !partial || a->M <= 0 || a->R <= 0 || (a->R % 4) || ntiles <= 0) return AITK_ERR_SHAPE;
  if (!a->T || (a->ldt % 4) || ((uintptr_t)partial & 15)) return AITK_ERR_ALIGN;
  if (a->mult && a->rows_per_batch <= 0) return AITK_ERR_ARG;
  if (a->split_rp < 0 || (a->split_rp > 0 && ((a->split_rp % 4) || (a->split_rp < a->R && (a->R % a->split_rp))))) return AITK_ERR_ARG;
  const long nt = (long)a->M * (a->R / 4);
  hipLaunchKernelGGL(lora_dt_finish_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a, partial, (int)ntiles);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// aitk_lora_down for launches of a FEW rows over a LONG contraction (the adaLN adapters' backward: dT [B, 16] = dmod [B, 6 d] lora_up — one workgroup
// pulling 1.2 MB of projection through one CU: 35 - 63 us): the contraction is cut into `nsplit` slices, one workgroup each (eight waves), their raw
// fp32 tiles are summed in slice order by the finish pass that also applies scale / multiplier / mask and writes T like aitk_lora_down would.
extern "C" int64_t aitk_lora_down_ksplit_workspace_bytes(int32_t M, int32_t R, int32_t nsplit) { return (int64_t)nsplit * M * R * 4; }
extern "C" int aitk_lora_down_ksplit(const AitkLoraDownArgs* a, float* partial, int32_t nsplit, aitk_stream_t stream) {
  if (!a || a->M <= 0 || a->K <= 0 || a->R != 16 || (a->K % 32) || nsplit < 1 || nsplit > a->K / 32) return AITK_ERR_SHAPE;
  if ((a->ldx % 8) || (a->ldp % 8) || ((uintptr_t)partial & 15)) return AITK_ERR_ALIGN;
  if (!a->X || !a->P || !a->T || !partial) return AITK_ERR_ARG;
  if (a->split_rp < 0 || (a->split_rp > 0 && ((a->split_rp % 4) || (a->split_rp < a->R && (a->R % a->split_rp))))) return AITK_ERR_ARG;
  if (a->mult && a->rows_per_batch <= 0) return AITK_ERR_ARG;
  hipLaunchKernelGGL((lora_down16_kernel<1, 6, true, 8>), dim3((a->M + 31) / 32, nsplit), dim3(512), 0, (hipStream_t)stream, *a, partial);
  AITK_LAUNCH_CHECK();
  const long nt = (long)a->M * (a->R / 4);
  hipLaunchKernelGGL(lora_dt_finish_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a, (const float*)partial, nsplit);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}


// 64 outputs per 256-thread block: thread (j = tid & 63, k = tid >> 6) sums the chunks c = k, k+4, ... of output j, the four
// partial sums are combined through LDS in a fixed order (deterministic; 4x shorter dependent-load chain than one thread per
// output, which matters for the LoKr factor gradients whose row count M * factor runs into the millions).
__device__ __forceinline__ void lora_wgrad_finish_body(const AitkLoraWgradArgs& p, int nchunks, long block, float* red) {
  const int j = threadIdx.x & 63, k = threadIdx.x >> 6;
  const long idx = block * 64 + j;
  const long total = (long)p.R * p.L;
  float s = 0.f;
  if (idx < total) {
    int c = k;
    for (; c + 12 < nchunks; c += 16) {  // four chunk loads in flight, added in chunk order (the same sum, bit for bit): 63 chunks per launch at B = 7
      const float t0 = p.partial[(long)c * total + idx], t1 = p.partial[(long)(c + 4) * total + idx];
      const float t2 = p.partial[(long)(c + 8) * total + idx], t3 = p.partial[(long)(c + 12) * total + idx];
      s += t0; s += t1; s += t2; s += t3;
    }
    for (; c < nchunks; c += 4) s += p.partial[(long)c * total + idx];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (k != 0 || idx >= total) return;
  s = (red[j] + red[64 + j]) + (red[128 + j] + red[192 + j]);
  const int r = (int)(idx / p.L), l = (int)(idx - (long)r * p.L);
  float* o = p.out + (long)r * p.out_stride_r + (long)l * p.out_stride_l;
  *o = p.accumulate ? (*o + s) : s;
}
__global__ __launch_bounds__(256) void lora_wgrad_finish_kernel(AitkLoraWgradArgs p, int nchunks) {
  __shared__ float red[256];
  lora_wgrad_finish_body(p, nchunks, (long)blockIdx.x, red);
}
// the two finish passes of aitk_lora_bwd_fused (lora_up gradient: blocks [0, nblk_w); dT slab: the rest) as ONE launch: they are independent, and at short
// batches a launch of their size is mostly its own start-up
__global__ __launch_bounds__(256) void lora_bwd_finish2_kernel(AitkLoraWgradArgs w, int nchunks, int nblk_w, AitkLoraDownArgs d, const float* dt_part, int ntiles) {
  __shared__ float red[256];
  if ((int)blockIdx.x < nblk_w) lora_wgrad_finish_body(w, nchunks, (long)blockIdx.x, red);  // block-uniform branch: the barrier inside is reached by all or none
  else lora_dt_finish_body(d, dt_part, ntiles, (long)blockIdx.x - nblk_w);
}

extern "C" int64_t aitk_lora_wgrad_workspace_bytes(int32_t M, int32_t R, int32_t L) {
  const int64_t nchunks = (M + WG_MC - 1) / WG_MC;
  return nchunks * (int64_t)R * L * 4;
}

static int wgrad_row_chunk(int M) {
  int mc = M >= 8192 ? 2 * WG_MC : WG_MC;
  if ((M + mc - 1) / mc > 512) mc = ((M + 511) / 512 + 63) / 64 * 64;  // millions of rows (LoKr): at most 512 row chunks
  return mc;
}
// the producing launch of aitk_lora_wgrad (chunk partials only)
static int wgrad_main(const AitkLoraWgradArgs* a, hipStream_t s) {
  if (!a || a->M <= 0 || a->R <= 0 || a->L <= 0) return AITK_ERR_SHAPE;
  if ((a->R % 16) || a->R > 64 || (a->L % 8)) return AITK_ERR_SHAPE;
  if ((a->ldg % 8) || (a->lds % 8)) return AITK_ERR_ALIGN;
  if (!a->partial || !a->out) return AITK_ERR_ARG;
  if (a->split_rp < 0 || (a->split_rp > 0 && ((a->split_rp % 8) || (a->split_rp < a->R && (a->R % a->split_rp))))) return AITK_ERR_ARG;
  const int mc = wgrad_row_chunk(a->M);
  const int nchunks = (a->M + mc - 1) / mc;
  dim3 grid((a->L + WG_LT - 1) / WG_LT, nchunks);
  switch (a->R / 16) {
    case 1: hipLaunchKernelGGL(lora_wgrad_kernel<1>, grid, dim3(256), 0, s, *a, mc); break;
    case 2: hipLaunchKernelGGL(lora_wgrad_kernel<2>, grid, dim3(256), 0, s, *a, mc); break;
    case 3: hipLaunchKernelGGL(lora_wgrad_kernel<3>, grid, dim3(256), 0, s, *a, mc); break;
    default: hipLaunchKernelGGL(lora_wgrad_kernel<4>, grid, dim3(256), 0, s, *a, mc); break;
  }
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}
static int wgrad_finish(const AitkLoraWgradArgs* a, hipStream_t s) {
  const int mc = wgrad_row_chunk(a->M);
  const int nchunks = (a->M + mc - 1) / mc;
  const long total = (long)a->R * a->L;
  hipLaunchKernelGGL(lora_wgrad_finish_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, *a, nchunks);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}
extern "C" int aitk_lora_wgrad(const AitkLoraWgradArgs* a, aitk_stream_t stream) {
  const int rc = wgrad_main(a, (hipStream_t)stream);
  return rc != AITK_OK ? rc : wgrad_finish(a, (hipStream_t)stream);
}

// aitk_lora_wgrad with a two-part G operand: out[r][l] += sum_m S[m][r] * X[m][l],  X[m][l] = G[m][l] for l < split_col and
// act(G2[m][l - split_col]) beyond — the lora_down gradient of a layer whose input is [attention output | gelu(pre-activation)] (FLUX single
// blocks' proj_out) or gelu(pre-activation) alone (ff.net.2, split_col = 0) WITHOUT keeping the GELU output resident: 6.4 GB per image at 1024^2.
static int wgrad2_main(const AitkLoraWgradArgs* a, const AitkWgradSrc2* q, hipStream_t s) {
  if (!a || !q || a->M <= 0 || a->R <= 0 || a->L <= 0) return AITK_ERR_SHAPE;
  if ((a->R % 16) || a->R > 64 || (a->L % 8)) return AITK_ERR_SHAPE;
  if ((a->lds % 8) || (q->ldg2 % 8)) return AITK_ERR_ALIGN;
  if (!a->partial || !a->out || !q->G2) return AITK_ERR_ARG;
  if (q->split_col < 0 || q->split_col >= a->L || (q->split_col % WG_LT) || (q->act != 0 && q->act != 1)) return AITK_ERR_ARG;
  if (q->split_col > 0 && (!a->G || (a->ldg % 8))) return AITK_ERR_ARG;
  if (a->split_rp < 0 || (a->split_rp > 0 && ((a->split_rp % 8) || (a->split_rp < a->R && (a->R % a->split_rp))))) return AITK_ERR_ARG;
  const int mc = wgrad_row_chunk(a->M);
  const int nchunks = (a->M + mc - 1) / mc;
  dim3 grid((a->L + WG_LT - 1) / WG_LT, nchunks);
  switch (a->R / 16) {
    case 1: hipLaunchKernelGGL(lora_wgrad2_kernel<1>, grid, dim3(256), 0, s, *a, *q, mc); break;
    case 2: hipLaunchKernelGGL(lora_wgrad2_kernel<2>, grid, dim3(256), 0, s, *a, *q, mc); break;
    case 3: hipLaunchKernelGGL(lora_wgrad2_kernel<3>, grid, dim3(256), 0, s, *a, *q, mc); break;
    default: hipLaunchKernelGGL(lora_wgrad2_kernel<4>, grid, dim3(256), 0, s, *a, *q, mc); break;
  }
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}
extern "C" int aitk_lora_wgrad2(const AitkLoraWgradArgs* a, const AitkWgradSrc2* q, aitk_stream_t stream) {
  const int rc = wgrad2_main(a, q, (hipStream_t)stream);
  return rc != AITK_OK ? rc : wgrad_finish(a, (hipStream_t)stream);
}

// ---- deferred finish (ABI 12) ----
// aitk_lora_wgrad_main: the producing launch alone (src2 == NULL: aitk_lora_wgrad's, else aitk_lora_wgrad2's); `partial` then holds the chunk partials and `out`
// is untouched until aitk_lora_wgrad_finish_multi is handed the same argument block.  Nothing in a backward pass reads a weight gradient before the optimizer, so a
// trainer may collect up to AITK_WGRAD_FINISH_MAX finishes (each with its own `partial` buffer, outputs pairwise distinct) and run them as ONE launch: at short
// batches a finish launch is mostly its own start-up (5 us behind a 10-us producer, 380 of them per FLUX step).  Same arithmetic as the separate launch, bit for bit.
#define AITK_WGRAD_FINISH_MAX 8
struct WgradFinishJobs {
  AitkLoraWgradArgs p[AITK_WGRAD_FINISH_MAX];
  int nchunks[AITK_WGRAD_FINISH_MAX];
  int blk_end[AITK_WGRAD_FINISH_MAX];
  int n;
};
__global__ __launch_bounds__(256) void lora_wgrad_finish_multi_kernel(WgradFinishJobs j) {
  __shared__ float red[256];
  int i = 0;
  while (i + 1 < j.n && (int)blockIdx.x >= j.blk_end[i]) ++i;
  const int first = i ? j.blk_end[i - 1] : 0;
  lora_wgrad_finish_body(j.p[i], j.nchunks[i], (long)blockIdx.x - first, red);
}
extern "C" int aitk_lora_wgrad_main(const AitkLoraWgradArgs* a, const AitkWgradSrc2* src2, aitk_stream_t stream) {
  return src2 ? wgrad2_main(a, src2, (hipStream_t)stream) : wgrad_main(a, (hipStream_t)stream);
}
extern "C" int aitk_lora_wgrad_finish_multi(const AitkLoraWgradArgs* jobs, int32_t njobs, aitk_stream_t stream) {
  if (!jobs || njobs < 1 || njobs > AITK_WGRAD_FINISH_MAX) return AITK_ERR_ARG;
  WgradFinishJobs j;
  int blocks = 0;
  for (int i = 0; i < njobs; ++i) {
    const AitkLoraWgradArgs& a = jobs[i];
    if (a.M <= 0 || a.R <= 0 || a.L <= 0) return AITK_ERR_SHAPE;
    if (!a.partial || !a.out) return AITK_ERR_ARG;
    for (int k = 0; k < i; ++k)
      if (jobs[k].out == a.out) return AITK_ERR_ARG;  // two accumulations into one matrix inside one launch would race
    const int mc = wgrad_row_chunk(a.M);
    j.p[i] = a;
    j.nchunks[i] = (a.M + mc - 1) / mc;
    blocks += (int)(((long)a.R * a.L + 63) / 64);
    j.blk_end[i] = blocks;
  }
  j.n = njobs;
  hipLaunchKernelGGL(lora_wgrad_finish_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, j);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// ------------------------------------------------------------------------------------------------------------
// aitk_lora_bwd_fused: both adapter-side products of a layer's backward that stream dY — dT = c (dY (P + P_lo)^T) (aitk_lora_down with
// X = dY) and lora_up.weight.grad = dY^T T (aitk_lora_wgrad with G = dY) — from ONE read of dY.  Column-tile partials of dT
// ([L / 128][M][R] fp32, `dt_partial`) are summed in a fixed order by a finish pass that writes dT exactly as aitk_lora_down would.
// ------------------------------------------------------------------------------------------------------------
extern "C" int64_t aitk_lora_bwd_fused_workspace_bytes(int32_t M, int32_t R, int32_t L) {
  return (int64_t)((L + WG_LT - 1) / WG_LT) * M * R * 4;
}

extern "C" int aitk_lora_bwd_fused(const AitkLoraWgradArgs* a, const AitkLoraDownArgs* d, float* dt_partial, aitk_stream_t stream) {
  if (!a || !d || a->M <= 0 || a->R <= 0 || a->L <= 0) return AITK_ERR_SHAPE;
  if ((a->R != 16 && a->R != 32) || (a->L % 8)) return AITK_ERR_SHAPE;  // P fragments live in registers: 16 VGPRs per 16 ranks and precision half
  if ((a->ldg % 8) || (a->lds % 8) || (d->ldp % 8) || (d->ldt % 4)) return AITK_ERR_ALIGN;
  if (!a->partial || !a->out || !dt_partial || !d->P || !d->T) return AITK_ERR_ARG;
  // the two argument blocks must describe the same dY and the same rank block
  if ((const void*)d->X != (const void*)a->G || d->ldx != a->ldg || d->x_seg_rows != a->g_seg_rows || d->x_seg_stride != a->g_seg_stride) return AITK_ERR_ARG;
  if (d->M != a->M || d->K != a->L || d->R != a->R) return AITK_ERR_ARG;
  if (a->split_rp < 0 || (a->split_rp > 0 && ((a->split_rp % 8) || (a->split_rp < a->R && (a->R % a->split_rp))))) return AITK_ERR_ARG;
  if (d->split_rp < 0 || (d->split_rp > 0 && ((d->split_rp % 4) || (d->split_rp < d->R && (d->R % d->split_rp))))) return AITK_ERR_ARG;
  if (d->mult && d->rows_per_batch <= 0) return AITK_ERR_ARG;
  int mc = a->M >= 8192 ? 2 * WG_MC : WG_MC;
  if ((a->M + mc - 1) / mc > 512) mc = ((a->M + 511) / 512 + 63) / 64 * 64;
  const int nchunks = (a->M + mc - 1) / mc;
  const int ntiles1 = (a->L + WG_LT - 1) / WG_LT;
  // column tiles per workgroup (4 only at rank 16: the P slices must fit LDS beside the tiles); AITK_LORA_BWD_CT = 1 | 2 | 4 forces a form (A/B measurements)
  const char* e_ct = getenv("AITK_LORA_BWD_CT");  // read per launch: the micro-benchmark switches forms inside one process
  const int forced = e_ct ? atoi(e_ct) : 0;
  // measured (profiles/r05_lora_bwd_fused_bench_ct.json): two tiles per workgroup beat one at every shape of the step (L = 3072: 114 -> 101 us at
  // M = 32256, 30.6 -> 28.8 at M = 4608); four beat two from L = 12288 up (325 -> 271 us; also at M = 4608: 82 -> 72) and lose at L = 9216 (224 vs 236)
  int ct = ntiles1 >= 2 ? 2 : 1;
  if (a->R == 16 && ntiles1 >= 96) ct = 4;
  if (forced == 1 || forced == 2 || (forced == 4 && a->R == 16)) ct = forced;
  const int ntiles = (ntiles1 + ct - 1) / ct;
  dim3 grid(ntiles, nchunks);
  hipStream_t s = (hipStream_t)stream;
  WgradFuse fz{(const bf16_t*)d->P, (const bf16_t*)d->P_lo, (long)d->ldp, dt_partial};
  if (ct == 1) {
    if (a->R == 16) hipLaunchKernelGGL(lora_wgrad_fused_kernel<1>, grid, dim3(256), 0, s, *a, fz, mc);
    else hipLaunchKernelGGL(lora_wgrad_fused_kernel<2>, grid, dim3(256), 0, s, *a, fz, mc);
  } else if (ct == 2) {
    if (a->R == 16) hipLaunchKernelGGL((lora_bwd_fused_ct_kernel<1, 2>), grid, dim3(256), 0, s, *a, fz, mc);
    else hipLaunchKernelGGL((lora_bwd_fused_ct_kernel<2, 2>), grid, dim3(256), 0, s, *a, fz, mc);
  } else {
    hipLaunchKernelGGL((lora_bwd_fused_ct_kernel<1, 4>), grid, dim3(256), 0, s, *a, fz, mc);
  }
  AITK_LAUNCH_CHECK();
  const long total = (long)a->R * a->L;
  const long nt = (long)d->M * (d->R / 4);
  static int merged = -1;  // AITK_LORA_FINISH_MERGED=0: the two finish passes as two launches (A/B)
  if (merged < 0) {
    const char* e = getenv("AITK_LORA_FINISH_MERGED");
    merged = (e && atoi(e) == 0) ? 0 : 1;
  }
  if (merged) {
    const int nblk_w = (int)((total + 63) / 64);
    hipLaunchKernelGGL(lora_bwd_finish2_kernel, dim3((unsigned)(nblk_w + (nt + 255) / 256)), dim3(256), 0, s, *a, nchunks, nblk_w, *d, (const float*)dt_partial, ntiles);
    AITK_LAUNCH_CHECK();
    return AITK_OK;
  }
  hipLaunchKernelGGL(lora_wgrad_finish_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, *a, nchunks);
  AITK_LAUNCH_CHECK();
  hipLaunchKernelGGL(lora_dt_finish_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, *d, (const float*)dt_partial, ntiles);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}
