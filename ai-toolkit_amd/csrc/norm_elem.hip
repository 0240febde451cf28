// HBM-bound row kernels of the DiT block (gfx950): adaLN LayerNorm+modulate, gate/residual backward, per-head
// QK RMSNorm + RoPE, column-sum finish.  One wave (64 lanes) per token row, 16-B (8 x bf16) accesses, wave-level
// shuffle reductions only (no LDS in the row math); per-batch column sums are accumulated in registers across the
// rows a wave visits and combined once per workgroup.
//
// Reference math being restated (all in diffusers, called from toolkit/stable_diffusion_model.py:2192-2205):
//   AdaLayerNormZero / ZeroSingle / Continuous:  LN(x; eps=1e-6, no affine) * (1 + scale[b]) + shift[b]
//   gate residual:                               x + gate[b] * y
//   attention pre-processing:                    RMSNorm(head_dim, eps=1e-6, weight) on q,k then rotary embedding
//                                                (order: toolkit/models/flux_sage_attn.py:36-74;
//                                                 rotation: extensions_built_in/diffusion_models/chroma/src/math.py:47-51)
#include <cstdlib>
#include "common.h"
#include "aitk_args.h"

#define MAXI 8  // C <= 64 lanes * 8 elems * MAXI = 4096

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = bf2f(v.x & 0xffff); f[1] = bf2f(v.x >> 16);
  f[2] = bf2f(v.y & 0xffff); f[3] = bf2f(v.y >> 16);
  f[4] = bf2f(v.z & 0xffff); f[5] = bf2f(v.z >> 16);
  f[6] = bf2f(v.w & 0xffff); f[7] = bf2f(v.w >> 16);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]);
  v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
  return v;
}

// ---------------------------------------------------------------- LN + modulate forward
__global__ __launch_bounds__(256) void ln_mod_fwd_kernel(AitkLnModArgs p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = blockIdx.x * 4 + wave;
  if (m >= p.M) return;
  const int b = m / p.rows_per_batch;
  const bf16_t* xr = p.x + (long)m * p.ldx;
  float xv[MAXI][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < p.C) {
      uint4 v = *reinterpret_cast<const uint4*>(xr + c);
      unpack8(v, xv[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += xv[i][e];
    }
  }
  const float mean = wave_sum(sum) / (float)p.C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < p.C) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float d = xv[i][e] - mean;
        sq += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)p.C + p.eps);
  if (lane == 0 && p.mean) {
    p.mean[m] = mean;
    p.rstd[m] = rstd;
  }
  const bf16_t* sh = p.shift + (long)b * p.ld_mod;
  const bf16_t* sc = p.scale + (long)b * p.ld_mod;
  bf16_t* orow = p.out + (long)m * p.ld_out;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < p.C) {
      float s8[8], h8[8], o[8];
      unpack8(*reinterpret_cast<const uint4*>(sc + c), s8);
      unpack8(*reinterpret_cast<const uint4*>(sh + c), h8);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (xv[i][e] - mean) * rstd * (1.0f + s8[e]) + h8[e];
      *reinterpret_cast<uint4*>(orow + c) = pack8(o);
    }
  }
}

extern "C" int aitk_ln_mod_fwd(const AitkLnModArgs* a, aitk_stream_t stream) {
  if (!a || a->M <= 0 || a->C <= 0 || (a->C % 8) || a->C > 64 * 8 * MAXI) return AITK_ERR_SHAPE;
  if ((a->ldx % 8) || (a->ld_out % 8) || (a->ld_mod % 8) || a->rows_per_batch <= 0) return AITK_ERR_ALIGN;
  hipLaunchKernelGGL(ln_mod_fwd_kernel, dim3((a->M + 3) / 4), dim3(256), 0, (hipStream_t)stream, *a);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// ---------------------------------------------------------------- LN + modulate backward
// xn = xhat*(1+scale)+shift  =>  g = dxn*(1+scale);  dx = rstd*(g - mean(g) - xhat*mean(g*xhat)) (+ dres)
//                                dshift[b][c] = sum_s dxn ;  dscale[b][c] = sum_s dxn*xhat
// grid = (ceil(S/RPB), B).  Phase 1 (wave per row): the two row means.  Phase 2 (thread per 8-column chunk,
// walks the block's rows): dx and the column partials in registers -> deterministic, no atomics.
#define RPB 16
__global__ __launch_bounds__(256) void ln_mod_bwd_kernel(AitkLnModBwdArgs p) {
  __shared__ float rowstat[RPB][4];  // c1, c2, mean, rstd
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int s0 = blockIdx.x * RPB;
  const int nrows = min(RPB, p.S - s0);
  const long mbase = (long)b * p.S + s0;
  const bf16_t* sc = p.scale + (long)b * p.ld_mod;
  for (int r = wave; r < nrows; r += 4) {
    const long m = mbase + r;
    const float mean = p.mean[m], rstd = p.rstd[m];
    const bf16_t* xr = p.x + m * p.ldx;
    const bf16_t* gr = p.dxn + m * p.ld_dxn;
    float a1 = 0.f, a2 = 0.f;
    for (int c = lane * 8; c < p.C; c += 512) {
      float xv[8], gv[8], s8[8];
      unpack8(*reinterpret_cast<const uint4*>(xr + c), xv);
      unpack8(*reinterpret_cast<const uint4*>(gr + c), gv);
      unpack8(*reinterpret_cast<const uint4*>(sc + c), s8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float g = gv[e] * (1.0f + s8[e]);
        a1 += g;
        a2 += g * (xv[e] - mean) * rstd;
      }
    }
    a1 = wave_sum(a1);
    a2 = wave_sum(a2);
    if (lane == 0) {
      rowstat[r][0] = a1 / (float)p.C;
      rowstat[r][1] = a2 / (float)p.C;
      rowstat[r][2] = mean;
      rowstat[r][3] = rstd;
    }
  }
  __syncthreads();
  const int nch = p.C / 8;
  for (int ch = tid; ch < nch; ch += 256) {
    const int c = ch * 8;
    float s8[8], dsh[8], dsc[8];
    unpack8(*reinterpret_cast<const uint4*>(sc + c), s8);
#pragma unroll
    for (int e = 0; e < 8; ++e) { dsh[e] = 0.f; dsc[e] = 0.f; }
    for (int r = 0; r < nrows; ++r) {
      const long m = mbase + r;
      const float c1 = rowstat[r][0], c2 = rowstat[r][1], mean = rowstat[r][2], rstd = rowstat[r][3];
      float xv[8], gv[8], o[8];
      unpack8(*reinterpret_cast<const uint4*>(p.x + m * p.ldx + c), xv);
      unpack8(*reinterpret_cast<const uint4*>(p.dxn + m * p.ld_dxn + c), gv);
      if (p.dres) unpack8(*reinterpret_cast<const uint4*>(p.dres + m * p.ld_dres + c), o);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (xv[e] - mean) * rstd;
        const float g = gv[e] * (1.0f + s8[e]);
        o[e] += rstd * (g - c1 - xh * c2);
        dsh[e] += gv[e];
        dsc[e] += gv[e] * xh;
      }
      *reinterpret_cast<uint4*>(p.dx + m * p.ld_dx + c) = pack8(o);
    }
    if (p.partial) {
      float* pp = p.partial + (((long)b * gridDim.x + blockIdx.x) * 2) * p.C + c;
      *reinterpret_cast<float4*>(pp) = make_float4(dsh[0], dsh[1], dsh[2], dsh[3]);
      *reinterpret_cast<float4*>(pp + 4) = make_float4(dsh[4], dsh[5], dsh[6], dsh[7]);
      *reinterpret_cast<float4*>(pp + p.C) = make_float4(dsc[0], dsc[1], dsc[2], dsc[3]);
      *reinterpret_cast<float4*>(pp + p.C + 4) = make_float4(dsc[4], dsc[5], dsc[6], dsc[7]);
    }
  }
}

// Single-pass form for C = 1024 NCW (FLUX: 3072).  The two-phase kernel above reads x and dxn twice (row means, then dx): 6 instead of 4 row streams per launch,
// 2.85 TB/s of algorithmic traffic at 32256 x 3072.  Here a 16-row chunk belongs to the TWO waves of a 128-thread workgroup, wave w owning the 512-column chunks
// 2 i + w of every row: a wave keeps its half of the row of x and dxn in registers (NCW 16-byte chunks per lane each), so both are read once; the two halves of the
// row means meet through 16 bytes of LDS and one workgroup barrier per row (double-buffered by row parity); the column partials of the chunk's rows stay in registers
// and are written as the chunk's partial exactly as before (same chunk list, same order of summation: the column sums are bit-identical, aitk_rows_per_block
// unchanged).  A row's means are summed per-wave halves first, i.e. in a different order than above: same formulas, last-bit differences in dx (<= 1 bf16 ulp).
// 278 -> 200 us per launch at 32256 x 3072 (3.96 TB/s; profiles/r04_rowkernels_ab.log); one wave per chunk with the whole row in registers (320 of them, one wave
// per SIMD) reached 250 us and was dropped.
template <int NCW>
__global__ __launch_bounds__(128) void ln_mod_bwd_row2_kernel(AitkLnModBwdArgs p) {
  __shared__ float st[2][2][2];  // [row parity][wave][a1, a2]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.y;
  const int chunk = blockIdx.x;
  const int s0 = chunk * RPB;
  const int nrows = min(RPB, p.S - s0);
  const long mbase = (long)b * p.S + s0;
  const bf16_t* sc = p.scale + (long)b * p.ld_mod;
  const int col0 = lane * 8 + 512 * wave;
  uint4 spk[NCW];
  float dsh[NCW][8], dsc[NCW][8];
#pragma unroll
  for (int i = 0; i < NCW; ++i) {
    spk[i] = *reinterpret_cast<const uint4*>(sc + col0 + 1024 * i);
#pragma unroll
    for (int e = 0; e < 8; ++e) dsh[i][e] = dsc[i][e] = 0.f;
  }
  for (int r = 0; r < nrows; ++r) {
    const long m = mbase + r;
    const float mean = p.mean[m], rstd = p.rstd[m];
    uint4 xpk[NCW], gpk[NCW], rpk[NCW];
#pragma unroll
    for (int i = 0; i < NCW; ++i) {
      xpk[i] = *reinterpret_cast<const uint4*>(p.x + m * p.ldx + col0 + 1024 * i);
      gpk[i] = *reinterpret_cast<const uint4*>(p.dxn + m * p.ld_dxn + col0 + 1024 * i);
      if (p.dres) rpk[i] = *reinterpret_cast<const uint4*>(p.dres + m * p.ld_dres + col0 + 1024 * i);
    }
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCW; ++i) {
      float xv[8], gv[8], s8[8];
      unpack8(xpk[i], xv);
      unpack8(gpk[i], gv);
      unpack8(spk[i], s8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float g = gv[e] * (1.0f + s8[e]);
        a1 += g;
        a2 += g * (xv[e] - mean) * rstd;
      }
    }
    a1 = wave_sum(a1);
    a2 = wave_sum(a2);
    if (lane == 0) {
      st[r & 1][wave][0] = a1;
      st[r & 1][wave][1] = a2;
    }
    __syncthreads();
    const float c1 = (st[r & 1][0][0] + st[r & 1][1][0]) / (float)p.C, c2 = (st[r & 1][0][1] + st[r & 1][1][1]) / (float)p.C;
#pragma unroll
    for (int i = 0; i < NCW; ++i) {
      float xv[8], gv[8], s8[8], o[8];
      unpack8(xpk[i], xv);
      unpack8(gpk[i], gv);
      unpack8(spk[i], s8);
      if (p.dres) unpack8(rpk[i], o);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (xv[e] - mean) * rstd;
        const float g = gv[e] * (1.0f + s8[e]);
        o[e] += rstd * (g - c1 - xh * c2);
        dsh[i][e] += gv[e];
        dsc[i][e] += gv[e] * xh;
      }
      *reinterpret_cast<uint4*>(p.dx + m * p.ld_dx + col0 + 1024 * i) = pack8(o);
    }
  }
  if (p.partial) {
#pragma unroll
    for (int i = 0; i < NCW; ++i) {
      float* pp = p.partial + (((long)b * gridDim.x + chunk) * 2) * p.C + col0 + 1024 * i;
      *reinterpret_cast<float4*>(pp) = make_float4(dsh[i][0], dsh[i][1], dsh[i][2], dsh[i][3]);
      *reinterpret_cast<float4*>(pp + 4) = make_float4(dsh[i][4], dsh[i][5], dsh[i][6], dsh[i][7]);
      *reinterpret_cast<float4*>(pp + p.C) = make_float4(dsc[i][0], dsc[i][1], dsc[i][2], dsc[i][3]);
      *reinterpret_cast<float4*>(pp + p.C + 4) = make_float4(dsc[i][4], dsc[i][5], dsc[i][6], dsc[i][7]);
    }
  }
}

extern "C" int32_t aitk_rows_per_block(void) { return RPB; }

extern "C" int aitk_ln_mod_bwd(const AitkLnModBwdArgs* a, aitk_stream_t stream) {
  if (!a || a->S <= 0 || a->B <= 0 || a->C <= 0 || (a->C % 8)) return AITK_ERR_SHAPE;
  if ((a->ldx % 8) || (a->ld_dxn % 8) || (a->ld_dx % 8) || (a->ld_mod % 8) || (a->dres && (a->ld_dres % 8))) return AITK_ERR_ALIGN;
  dim3 grid((a->S + RPB - 1) / RPB, a->B);
  const bool two_phase = getenv("AITK_LN_BWD_TWO_PHASE") != nullptr;  // measurement knob: the two-phase kernel at every width
  if (a->C == 3072 && !two_phase) hipLaunchKernelGGL(ln_mod_bwd_row2_kernel<3>, grid, dim3(128), 0, (hipStream_t)stream, *a);
  else hipLaunchKernelGGL(ln_mod_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, *a);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// ---------------------------------------------------------------- gate/residual backward
// x_new = res + gate[b]*y :  dy = gate*dx_new ;  dgate[b][c] = sum_s dx_new*y      (dres = dx_new, same buffer)
// y == NULL: only dy (the gate has no trainable ancestor)
__global__ __launch_bounds__(256) void gate_bwd_kernel(AitkGateBwdArgs p) {
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int s0 = blockIdx.x * RPB;
  const int nrows = min(RPB, p.S - s0);
  const long mbase = (long)b * p.S + s0;
  const bf16_t* gt = p.gate + (long)b * p.ld_gate;
  const int nch = p.C / 8;
  for (int ch = tid; ch < nch; ch += 256) {
    const int c = ch * 8;
    float g8[8], acc[8];
    unpack8(*reinterpret_cast<const uint4*>(gt + c), g8);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    // four rows of loads in flight before the first store (the compiler cannot move a load across the dy store: the buffers may alias for all it knows, and
    // a row at a time is a chain of 16 round trips per thread at short batches); rows are added in order: the column sums are the one-row loop's, bit for bit
    int r = 0;
    for (; r + 4 <= nrows; r += 4) {
      uint4 dq[4], yq[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long m = mbase + r + u;
        dq[u] = *reinterpret_cast<const uint4*>(p.dx + m * p.ld_dx + c);
        yq[u] = p.y ? *reinterpret_cast<const uint4*>(p.y + m * p.ld_y + c) : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long m = mbase + r + u;
        float dv[8], yv[8], o[8];
        unpack8(dq[u], dv);
        unpack8(yq[u], yv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o[e] = g8[e] * dv[e];
          if (p.y) acc[e] += dv[e] * yv[e];
        }
        *reinterpret_cast<uint4*>(p.dy + m * p.ld_dy + c) = pack8(o);
      }
    }
    for (; r < nrows; ++r) {
      const long m = mbase + r;
      float dv[8], yv[8], o[8];
      unpack8(*reinterpret_cast<const uint4*>(p.dx + m * p.ld_dx + c), dv);
      if (p.y) unpack8(*reinterpret_cast<const uint4*>(p.y + m * p.ld_y + c), yv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o[e] = g8[e] * dv[e];
        if (p.y) acc[e] += dv[e] * yv[e];
      }
      *reinterpret_cast<uint4*>(p.dy + m * p.ld_dy + c) = pack8(o);
    }
    if (!p.y) continue;  // no d_gate wanted (frozen modulation, Wan2.1)
    float* pp = p.partial + ((long)b * gridDim.x + blockIdx.x) * p.C + c;
    *reinterpret_cast<float4*>(pp) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4*>(pp + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
}

extern "C" int aitk_gate_bwd(const AitkGateBwdArgs* a, aitk_stream_t stream) {
  if (!a || a->S <= 0 || a->B <= 0 || a->C <= 0 || (a->C % 8)) return AITK_ERR_SHAPE;
  if ((a->ld_dx % 8) || (a->ld_y % 8) || (a->ld_dy % 8) || (a->ld_gate % 8)) return AITK_ERR_ALIGN;
  dim3 grid((a->S + RPB - 1) / RPB, a->B);
  hipLaunchKernelGGL(gate_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, *a);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// ---------------------------------------------------------------- column-sum finish
// partial [B][nchunk][V][C] fp32 -> out_v[b*ld_out + c] (bf16), v < V <= 2
// block = 64 columns x 4 chunk-groups: each thread sums a quarter of the row-block partials, LDS combines (fixed order)
__global__ __launch_bounds__(256) void colsum_finish_kernel(AitkColsumFinishArgs p) {
  __shared__ float red[4][64];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const long idx = (long)blockIdx.x * 64 + col;
  const long total = (long)p.B * p.V * p.C;
  float s = 0.f;
  int c = 0, v = 0, b = 0;
  if (idx < total) {
    c = (int)(idx % p.C);
    v = (int)((idx / p.C) % p.V);
    b = (int)(idx / ((long)p.C * p.V));
    const float* src = p.partial + ((long)b * p.nchunk * p.V + v) * p.C + c;
    const int k0 = (p.nchunk * grp) / 4, k1 = (p.nchunk * (grp + 1)) / 4;
    for (int k = k0; k < k1; ++k) s += src[(long)k * p.V * p.C];
  }
  red[grp][col] = s;
  __syncthreads();
  if (grp == 0 && idx < total) {
    const float t = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
    bf16_t* o = (v == 0 ? p.out0 : p.out1) + (long)b * p.ld_out + c;
    *o = f2bf(t);
  }
}

// The same reduction for C % 4 == 0 and many row blocks (the FLUX graphs: 288 per sample): 16 column quads x 16 chunk groups per block, 16-byte loads issued six
// at a time — the 64 x 4 form above is a chain of nchunk / 4 dependent 4-byte loads per thread (14 us for 7 MB at B = 1, 2.1 TB/s at B = 7).  Each group sums
// its chunks in ascending order and the groups are combined in a fixed tree: deterministic (another association than the 4-group form: fp32 rounding).
__global__ __launch_bounds__(256) void colsum_finish16_kernel(AitkColsumFinishArgs p) {
  __shared__ float4 red[16][16];
  const int cq = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int cq_n = p.C / 4;
  const long q = (long)blockIdx.x * 16 + cq;
  const long totalq = (long)p.B * p.V * cq_n;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int c = 0, v = 0, b = 0;
  if (q < totalq) {
    c = (int)(q % cq_n) * 4;
    v = (int)((q / cq_n) % p.V);
    b = (int)(q / ((long)cq_n * p.V));
    const float* src = p.partial + ((long)b * p.nchunk * p.V + v) * p.C + c;
    const long step = (long)p.V * p.C;
    const int k0 = (p.nchunk * grp) / 16, k1 = (p.nchunk * (grp + 1)) / 16;
    int k = k0;
    for (; k + 6 <= k1; k += 6) {
      float4 t[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) t[u] = *reinterpret_cast<const float4*>(src + (long)(k + u) * step);
#pragma unroll
      for (int u = 0; u < 6; ++u) { s.x += t[u].x; s.y += t[u].y; s.z += t[u].z; s.w += t[u].w; }
    }
    for (; k < k1; ++k) {
      const float4 t = *reinterpret_cast<const float4*>(src + (long)k * step);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
  }
  red[grp][cq] = s;
  __syncthreads();
  if (grp == 0 && q < totalq) {
    float t[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float g[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) g[i] = reinterpret_cast<const float*>(&red[i][cq])[e];
#pragma unroll
      for (int w = 1; w < 16; w *= 2)
#pragma unroll
        for (int i = 0; i < 16; i += 2 * w) g[i] += g[i + w];
      t[e] = g[0];
    }
    bf16_t* o = (v == 0 ? p.out0 : p.out1) + (long)b * p.ld_out + c;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f2bf(t[e]);
  }
}

extern "C" int aitk_colsum_finish(const AitkColsumFinishArgs* a, aitk_stream_t stream) {
  if (!a || a->B <= 0 || a->C <= 0 || a->V <= 0 || a->V > 2 || a->nchunk <= 0) return AITK_ERR_SHAPE;
  if (!a->out0 || (a->V == 2 && !a->out1)) return AITK_ERR_ARG;
  const long total = (long)a->B * a->V * a->C;
  static int wide = -1;  // AITK_COLSUM_FINISH16=0: the 4-group form everywhere (A/B)
  if (wide < 0) {
    const char* e = getenv("AITK_COLSUM_FINISH16");
    wide = (e && atoi(e) == 0) ? 0 : 1;
  }
  if (wide && (a->C % 4) == 0 && a->nchunk >= 32 && ((uintptr_t)a->partial % 16) == 0) {
    hipLaunchKernelGGL(colsum_finish16_kernel, dim3((unsigned)((total / 4 + 15) / 16)), dim3(256), 0, (hipStream_t)stream, *a);
    AITK_LAUNCH_CHECK();
    return AITK_OK;
  }
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, (hipStream_t)stream, *a);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// ---------------------------------------------------------------- per-head RMSNorm + RoPE (q,k) / copy (v)
// One 16-lane group per (token, head); D = 128 = 16 lanes x 8.  Rounding points mirror diffusers' RMSNorm
// (normalise in fp32 -> cast to the bf16 weight dtype -> * weight) and apply_rotary_emb (fp32 -> bf16).
__device__ __forceinline__ float group16_sum(float v) {
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 1, 64);
  return v;
}

// Work split (round 4): one WAVE per token; lane = (hq = lane >> 4: one of four consecutive heads, sub = lane & 15: the 8-column chunk), the wave walks the
// token's heads four at a time — 1 KiB contiguous per load — and the token's cos / sin row and the norm weight are loaded once per lane instead of once per
// (token, head).  The round-3 form gave each 16-lane group one (token, head) pair of a flat pair list: two 64-bit divisions + two modulos per 16 bytes (several
// hundred vector instructions) and 64 B of fp32 rotary table per 16 B of data; it ran at 3.0 TB/s.  Same arithmetic per element: bit-identical.
__global__ __launch_bounds__(256) void qkv_post_fwd_kernel(AitkQkvPostArgs p) {
  const AitkQkvJob job = p.job[blockIdx.y];
  const int lane = threadIdx.x & 63, sub = lane & 15, hq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ntok = p.B * p.S_src;
  float w[8];
  if (job.weight) unpack8(*reinterpret_cast<const uint4*>(job.weight + sub * 8), w);
  for (int tok = blockIdx.x * 4 + wave; tok < ntok; tok += gridDim.x * 4) {
    const int b = tok / p.S_src, s = tok - b * p.S_src;
    const bf16_t* src = job.src + (long)tok * job.ld_src + sub * 8;
    bf16_t* dst = job.dst + ((long)b * p.S_dst + p.s_off + s) * job.ld_dst + sub * 8;
    if (!job.weight) {
      for (int hd = hq; hd < p.H; hd += 4) *reinterpret_cast<uint4*>(dst + hd * 128) = *reinterpret_cast<const uint4*>(src + hd * 128);
      continue;
    }
    float cs[8], sn[8];
    {
      const float* cp = p.cos + (long)(p.s_off + s) * 128 + sub * 8;
      const float* sp = p.sin + (long)(p.s_off + s) * 128 + sub * 8;
      const float4 c0 = *reinterpret_cast<const float4*>(cp), c1 = *reinterpret_cast<const float4*>(cp + 4);
      const float4 s0 = *reinterpret_cast<const float4*>(sp), s1 = *reinterpret_cast<const float4*>(sp + 4);
      cs[0] = c0.x; cs[1] = c0.y; cs[2] = c0.z; cs[3] = c0.w; cs[4] = c1.x; cs[5] = c1.y; cs[6] = c1.z; cs[7] = c1.w;
      sn[0] = s0.x; sn[1] = s0.y; sn[2] = s0.z; sn[3] = s0.w; sn[4] = s1.x; sn[5] = s1.y; sn[6] = s1.z; sn[7] = s1.w;
    }
    for (int hd = hq; hd < p.H; hd += 4) {
      float x[8], o[8], t[8];
      unpack8(*reinterpret_cast<const uint4*>(src + hd * 128), x);
      float ss = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) ss += x[e] * x[e];
      const float rstd = rsqrtf(group16_sum(ss) / 128.0f + p.eps);
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = bfround(bfround(x[e] * rstd) * w[e]);
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        o[e] = t[e] * cs[e] - t[e + 1] * sn[e];
        o[e + 1] = t[e + 1] * cs[e + 1] + t[e] * sn[e + 1];
      }
      *reinterpret_cast<uint4*>(dst + hd * 128) = pack8(o);
    }
  }
}

// backward: src = grad wrt the joint (roped) tensor, raw = the forward input, dst = grad wrt raw
__global__ __launch_bounds__(256) void qkv_post_bwd_kernel(AitkQkvPostArgs p) {
  const AitkQkvJob job = p.job[blockIdx.y];
  const int lane = threadIdx.x & 63, sub = lane & 15, hq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ntok = p.B * p.S_src;
  float w[8];
  if (job.weight) unpack8(*reinterpret_cast<const uint4*>(job.weight + sub * 8), w);
  for (int tok = blockIdx.x * 4 + wave; tok < ntok; tok += gridDim.x * 4) {
    const int b = tok / p.S_src, s = tok - b * p.S_src;
    // here "dst" layout = joint grad (read), "src" layout = raw-side (write); see ops.qkv_post_bwd
    const bf16_t* gj = job.dst + ((long)b * p.S_dst + p.s_off + s) * job.ld_dst + sub * 8;
    bf16_t* graw = const_cast<bf16_t*>(job.src) + (long)tok * job.ld_src + sub * 8;
    if (!job.weight) {
      for (int hd = hq; hd < p.H; hd += 4) *reinterpret_cast<uint4*>(graw + hd * 128) = *reinterpret_cast<const uint4*>(gj + hd * 128);
      continue;
    }
    const bf16_t* rawp = job.raw + (long)tok * job.ld_raw + sub * 8;
    float cs[8], sn[8];
    {
      const float* cp = p.cos + (long)(p.s_off + s) * 128 + sub * 8;
      const float* sp = p.sin + (long)(p.s_off + s) * 128 + sub * 8;
      const float4 c0 = *reinterpret_cast<const float4*>(cp), c1 = *reinterpret_cast<const float4*>(cp + 4);
      const float4 s0 = *reinterpret_cast<const float4*>(sp), s1 = *reinterpret_cast<const float4*>(sp + 4);
      cs[0] = c0.x; cs[1] = c0.y; cs[2] = c0.z; cs[3] = c0.w; cs[4] = c1.x; cs[5] = c1.y; cs[6] = c1.z; cs[7] = c1.w;
      sn[0] = s0.x; sn[1] = s0.y; sn[2] = s0.z; sn[3] = s0.w; sn[4] = s1.x; sn[5] = s1.y; sn[6] = s1.z; sn[7] = s1.w;
    }
    for (int hd = hq; hd < p.H; hd += 4) {
      float g[8], x[8], dt[8], o[8];
      unpack8(*reinterpret_cast<const uint4*>(gj + hd * 128), g);
      unpack8(*reinterpret_cast<const uint4*>(rawp + hd * 128), x);
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        // o_e = t_e c_e - t_{e+1} s_e ; o_{e+1} = t_{e+1} c_{e+1} + t_e s_{e+1}
        dt[e] = g[e] * cs[e] + g[e + 1] * sn[e + 1];
        dt[e + 1] = -g[e] * sn[e] + g[e + 1] * cs[e + 1];
      }
      float ss = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) ss += x[e] * x[e];
      const float rstd = rsqrtf(group16_sum(ss) / 128.0f + p.eps);
      float dot = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        dt[e] *= w[e];              // through "* weight"
        dot += dt[e] * x[e] * rstd; // sum_i dt1_i * xhat_i
      }
      dot = group16_sum(dot) / 128.0f;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = rstd * (dt[e] - x[e] * rstd * dot);
      *reinterpret_cast<uint4*>(graw + hd * 128) = pack8(o);
    }
  }
}

static int qkv_post_check(const AitkQkvPostArgs* a) {
  if (!a || a->njobs <= 0 || a->njobs > 3 || a->B <= 0 || a->H <= 0 || a->S_src <= 0) return AITK_ERR_SHAPE;
  if (a->D != 128 || (long)a->B * a->S_src > 0x7fffffffL) return AITK_ERR_SHAPE;
  for (int j = 0; j < a->njobs; ++j)
    if ((a->job[j].ld_src % 8) || (a->job[j].ld_dst % 8)) return AITK_ERR_ALIGN;
  return AITK_OK;
}
extern "C" int aitk_qkv_post_fwd(const AitkQkvPostArgs* a, aitk_stream_t stream) {
  int rc = qkv_post_check(a);
  if (rc) return rc;
  const long ntok = (long)a->B * a->S_src;
  dim3 grid((unsigned)min((ntok + 3) / 4, (long)16384), a->njobs);
  hipLaunchKernelGGL(qkv_post_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, *a);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}
extern "C" int aitk_qkv_post_bwd(const AitkQkvPostArgs* a, aitk_stream_t stream) {
  int rc = qkv_post_check(a);
  if (rc) return rc;
  const long ntok = (long)a->B * a->S_src;
  dim3 grid((unsigned)min((ntok + 3) / 4, (long)16384), a->njobs);
  hipLaunchKernelGGL(qkv_post_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, *a);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// ---------------------------------------------------------------- RMSNorm across heads (+ RoPE)  (Wan2.1 attention)
// y = rope( bf16(bf16(x * rsqrt(mean_C(x^2) + eps)) * w) ),  C = H*128 <= 4096, one wave per token; rope optional.
// Restates the q/k path of toolkit/models/wan21/wan_attn.py:32-54 (norm_q/norm_k = diffusers RMSNorm over the full
// projection, then view_as_complex(x) * freqs on (2i, 2i+1) pairs; we rotate in fp32 instead of fp64).
__global__ __launch_bounds__(256) void rms_full_fwd_kernel(AitkRmsFullArgs p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long m = (long)blockIdx.x * 4 + wave;
  if (m >= p.M) return;
  const bf16_t* xr = p.x + m * p.ldx;
  float xv[MAXI][8];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < p.C) {
      unpack8(*reinterpret_cast<const uint4*>(xr + c), xv[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) ss += xv[i][e] * xv[i][e];
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)p.C + p.eps);
  const int s_pos = (int)(m % p.S);
  bf16_t* yr = p.y + m * p.ldy;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < p.C) {
      float w[8], t[8], o[8];
      unpack8(*reinterpret_cast<const uint4*>(p.weight + c), w);
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = bfround(bfround(xv[i][e] * rstd) * w[e]);
      if (p.cos) {
        const float* cs = p.cos + (long)s_pos * 128 + (c & 127);
        const float* sn = p.sin + (long)s_pos * 128 + (c & 127);
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          o[e] = t[e] * cs[e] - t[e + 1] * sn[e];
          o[e + 1] = t[e + 1] * cs[e + 1] + t[e] * sn[e + 1];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = t[e];
      }
      *reinterpret_cast<uint4*>(yr + c) = pack8(o);
    }
  }
}
// backward: g = grad wrt y, x = forward input -> dx
__global__ __launch_bounds__(256) void rms_full_bwd_kernel(AitkRmsFullArgs p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long m = (long)blockIdx.x * 4 + wave;
  if (m >= p.M) return;
  const bf16_t* xr = p.x + m * p.ldx;
  const bf16_t* gr = p.g + m * p.ldg;
  const int s_pos = (int)(m % p.S);
  float xv[MAXI][8], dt[MAXI][8];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < p.C) {
      float g[8], w[8];
      unpack8(*reinterpret_cast<const uint4*>(xr + c), xv[i]);
      unpack8(*reinterpret_cast<const uint4*>(gr + c), g);
      unpack8(*reinterpret_cast<const uint4*>(p.weight + c), w);
#pragma unroll
      for (int e = 0; e < 8; ++e) ss += xv[i][e] * xv[i][e];
      if (p.cos) {
        const float* cs = p.cos + (long)s_pos * 128 + (c & 127);
        const float* sn = p.sin + (long)s_pos * 128 + (c & 127);
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          dt[i][e] = (g[e] * cs[e] + g[e + 1] * sn[e + 1]) * w[e];
          dt[i][e + 1] = (-g[e] * sn[e] + g[e + 1] * cs[e + 1]) * w[e + 1];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) dt[i][e] = g[e] * w[e];
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)p.C + p.eps);
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < p.C) {
#pragma unroll
      for (int e = 0; e < 8; ++e) dot += dt[i][e] * xv[i][e] * rstd;
    }
  }
  dot = wave_sum(dot) / (float)p.C;
  bf16_t* yr = p.y + m * p.ldy;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < p.C) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = rstd * (dt[i][e] - xv[i][e] * rstd * dot);
      *reinterpret_cast<uint4*>(yr + c) = pack8(o);
    }
  }
}
static int rms_full_check(const AitkRmsFullArgs* a) {
  if (!a || a->M <= 0 || a->C <= 0 || (a->C % 128) || a->C > 64 * 8 * MAXI || a->S <= 0) return AITK_ERR_SHAPE;
  if ((a->ldx % 8) || (a->ldy % 8)) return AITK_ERR_ALIGN;
  if (!a->x || !a->y || !a->weight) return AITK_ERR_ARG;
  return AITK_OK;
}
extern "C" int aitk_rms_full_fwd(const AitkRmsFullArgs* a, aitk_stream_t stream) {
  int rc = rms_full_check(a);
  if (rc) return rc;
  hipLaunchKernelGGL(rms_full_fwd_kernel, dim3((unsigned)((a->M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *a);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}
extern "C" int aitk_rms_full_bwd(const AitkRmsFullArgs* a, aitk_stream_t stream) {
  int rc = rms_full_check(a);
  if (rc) return rc;
  if (!a->g || (a->ldg % 8)) return AITK_ERR_ARG;
  hipLaunchKernelGGL(rms_full_bwd_kernel, dim3((unsigned)((a->M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *a);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// ---------------------------------------------------------------- small element-wise ops on [rows, C] bf16
// op 0: y = silu(x)                      (AdaLayerNorm*: linear(silu(temb)))
// op 1: y = x                            (copy / cast helper)
// op 2: y = a + x                        (sum of embedder outputs; a_rows_per_batch > 0: a row = m / a_rows_per_batch — the
//                                         ResnetBlock2D time-embedding add h + temb_proj[b] broadcast over the pixels of a sample)
// op 3: y = alpha * x                    (scaled lora_up for merge_in / merge_out)
__global__ void ew_kernel(AitkEwArgs p) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  const long total = (long)p.rows * p.C;
  if (i >= total) return;
  const long r = i / p.C;
  const int c = (int)(i - r * p.C);
  float x[8], o[8], a8[8];
  unpack8(*reinterpret_cast<const uint4*>(p.x + r * p.ldx + c), x);
  if (p.op == 2) unpack8(*reinterpret_cast<const uint4*>(p.a + (p.a_rows_per_batch > 0 ? r / p.a_rows_per_batch : r) * p.lda + c), a8);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (p.op == 0) o[e] = x[e] / (1.0f + expf(-x[e]));
    else if (p.op == 1) o[e] = x[e];
    else if (p.op == 2) o[e] = a8[e] + x[e];
    else o[e] = p.alpha * x[e];
  }
  *reinterpret_cast<uint4*>(p.y + r * p.ldy + c) = pack8(o);
}
extern "C" int aitk_ew(const AitkEwArgs* a, aitk_stream_t stream) {
  if (!a || a->rows <= 0 || a->C <= 0 || (a->C % 8)) return AITK_ERR_SHAPE;
  if ((a->ldx % 8) || (a->ldy % 8) || (a->op == 2 && (!a->a || (a->lda % 8)))) return AITK_ERR_ALIGN;
  const long n = (long)a->rows * a->C / 8;
  hipLaunchKernelGGL(ew_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// sinusoidal timestep projection (diffusers Timesteps(256, flip_sin_to_cos=True, shift 0)): out[b] = [cos | sin]
// restated from extensions_built_in/diffusion_models/chroma/src/layers.py:30-53
__global__ void timestep_embed_kernel(const float* t, bf16_t* out, int B, int dim, float tscale) {
  const int b = blockIdx.x;
  const int half = dim / 2;
  for (int i = threadIdx.x; i < half; i += blockDim.x) {
    const float freq = expf(-logf(10000.0f) * (float)i / (float)half);
    const float ang = t[b] * tscale * freq;
    out[(long)b * dim + i] = f2bf(cosf(ang));
    out[(long)b * dim + half + i] = f2bf(sinf(ang));
  }
}
extern "C" int aitk_timestep_embed(const float* t, aitk_bf16* out, int32_t B, int32_t dim, float tscale, aitk_stream_t stream) {
  if (B <= 0 || dim <= 0 || (dim & 1)) return AITK_ERR_SHAPE;
  hipLaunchKernelGGL(timestep_embed_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, t, out, B, dim, tscale);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// strided 2-D copy (device to device) for the few layout moves of the block graph (cat/split of text|image rows)
extern "C" int aitk_copy2d(void* dst, int64_t dst_pitch_bytes, const void* src, int64_t src_pitch_bytes,
                           int64_t width_bytes, int64_t rows, aitk_stream_t stream) {
  hipError_t e = hipMemcpy2DAsync(dst, (size_t)dst_pitch_bytes, src, (size_t)src_pitch_bytes, (size_t)width_bytes,
                                  (size_t)rows, hipMemcpyDeviceToDevice, (hipStream_t)stream);
  return (int)e;
}
