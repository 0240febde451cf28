// LoKr (Kronecker-product adapter) — per-token  out_m = scale * A . X_m . B^T  on MFMA, gfx950.
//
// Reference: toolkit/models/lokr.py:331-399 (_call_forward_fast_linear): the delta of a LoKr-wrapped Linear is
//   X = x.unflatten(-1, (in_m, in_n));  tmp = einsum('...qs,os->...qo', X, w2);  delta = einsum('...qo,pq->...po', tmp, w1*scale)
// i.e. for every token the [in_m x in_n] view of its feature row is multiplied by w2^T on the right and w1 on the left
// (kron(w1, w2) is never formed).  The same per-token product with transposed factors is the data gradient, and with one
// factor replaced by the identity it produces the intermediates of the two factor gradients (see ai-toolkit_amd/graph.py).
//
// One workgroup (4 waves) keeps A and B in LDS for its whole life and walks over tokens m = blockIdx.x, += gridDim.x:
//   stage 0  the token's feature row (a_in*b_in bf16, contiguous) is prefetched into registers one token ahead, then written
//            to LDS as Xs[q][s]
//   stage 1  Tt[o][q] = bf16( sum_s Xs[q][s] * Bs[o][s] )      v_mfma_f32_16x16x32_bf16, D rows = q so a lane owns 4 consecutive q
//   stage 2  Os[p][o] = scale * sum_q As[p][q] * Tt[o][q]      D rows = o so a lane owns 4 consecutive o of one output row
//   stage 3  the output row is copied out with 16-byte stores (optionally a column window of it, optionally += )
// All K extents are zero-padded to multiples of 32 in LDS (rows to multiples of 16), so any factor pair the reference's
// factorization() produces for dims that are multiples of 8 works (48x64, 96x128, 120x128, 128x144, 32x48, 80x112 ...).
// HBM-bound by design: 2 passes over an [M, features] bf16 tensor per call; the MFMA work is ~1/a_out of the base GEMM.
#include "aitk_args.h"
#include "common.h"

namespace {

// KRON_MAXCH (template): 16-byte chunks of the token row held in registers per thread — 2 for rows up to 4096 features (the
// common 3072-wide layers: fewer registers, one more resident workgroup per CU), 8 for rows up to 16384; longer rows are staged directly.

__device__ __forceinline__ const bf16_t* seg_row(const bf16_t* base, long ld, int seg_rows, long seg_stride, int m) {
  if (seg_rows > 0) {
    const int s = m / seg_rows;
    return base + (long)s * seg_stride + (long)(m - s * seg_rows) * ld;
  }
  return base + (long)m * ld;
}

struct KronLayout {
  int ai16, ao16, bi16, bo16;  // row counts padded to 16
  int xs;                      // row stride (elements) of Xs / Bs   (K = b_in padded to 32, +8)
  int as;                      // row stride of As / Tt              (K = a_in padded to 32, +8)
  int off_x, off_b, off_a, off_t, off_o, total;  // element offsets
};

__host__ __device__ inline KronLayout kron_layout(int a_in, int b_in, int a_out, int b_out, bool hasA, bool hasB) {
  KronLayout L;
  L.ai16 = (a_in + 15) / 16 * 16; L.ao16 = (a_out + 15) / 16 * 16;
  L.bi16 = (b_in + 15) / 16 * 16; L.bo16 = (b_out + 15) / 16 * 16;
  L.xs = (b_in + 31) / 32 * 32 + 8;
  L.as = (a_in + 31) / 32 * 32 + 8;
  int o = 0;
  L.off_x = o; o += hasB ? L.ai16 * L.xs : 0;
  L.off_b = o; o += hasB ? L.bo16 * L.xs : 0;
  L.off_a = o; o += hasA ? L.ao16 * L.as : 0;
  L.off_t = o; o += hasA ? L.bo16 * L.as : 0;
  L.off_o = o; o += (a_out * b_out + 7) / 8 * 8;
  L.total = o;
  return L;
}

template <int KRON_MAXCH>
__global__ __launch_bounds__(256) void kron_apply_kernel(AitkKronApplyArgs p) {
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const bool hasA = p.A != nullptr, hasB = p.B != nullptr;
  const KronLayout L = kron_layout(p.a_in, p.b_in, p.a_out, p.b_out, hasA, hasB);
  bf16_t* Xs = lds + L.off_x;
  bf16_t* Bs = lds + L.off_b;
  bf16_t* As = lds + L.off_a;
  bf16_t* Tt = lds + L.off_t;
  bf16_t* Os = lds + L.off_o;

  // ---- one-time: zero everything (padding must be exact zeros), then the two factors
  for (int i = tid * 8; i < L.total; i += 256 * 8) *reinterpret_cast<uint4*>(lds + i) = uint4{0u, 0u, 0u, 0u};
  __syncthreads();
  if (hasB)
    for (int i = tid; i < p.b_out * (p.b_in / 8); i += 256) {
      const int o = i / (p.b_in / 8), c = i - o * (p.b_in / 8);
      *reinterpret_cast<uint4*>(Bs + o * L.xs + c * 8) = *reinterpret_cast<const uint4*>(p.B + (long)o * p.b_in + c * 8);
    }
  if (hasA)
    for (int i = tid; i < p.a_out * p.a_in; i += 256) {
      const int r = i / p.a_in, q = i - r * p.a_in;
      As[r * L.as + q] = p.A[i];
    }

  // ---- per-thread chunk table of the token row (same for every token)
  const int nin = p.a_in * p.b_in, nch = nin / 8;
  int dst[KRON_MAXCH];
#pragma unroll
  for (int c = 0; c < KRON_MAXCH; ++c) {
    const int idx = tid + c * 256;
    dst[c] = -1;
    if (idx < nch) {
      const int q = (idx * 8) / p.b_in, s = idx * 8 - q * p.b_in;
      // hasB: Xs[q][s..s+7];  !hasB: the row IS tmp[q][o=s] -> Tt[o][q] (hasA) or the output (handled at store time)
      dst[c] = hasB ? q * L.xs + s : (q << 16) | s;
    }
  }
  const bool in_regs = nch <= KRON_MAXCH * 256;
  uint4 reg[KRON_MAXCH];
  auto xrow = [&](int m) -> const bf16_t* { return seg_row(p.x, p.ldx, p.x_seg_rows, p.x_seg_stride, m); };
  auto load_regs = [&](int m) {
    const bf16_t* xr = xrow(m);
#pragma unroll
    for (int c = 0; c < KRON_MAXCH; ++c)
      if (dst[c] >= 0) reg[c] = *reinterpret_cast<const uint4*>(xr + (long)(tid + c * 256) * 8);
  };
  // identity-B targets: element (q, o) of tmp
  auto put_tmp = [&](int q, int o, bf16_t v) {
    if (hasA) Tt[o * L.as + q] = v;
    else Os[p.transpose_out ? o * p.a_out + q : q * p.b_out + o] = v;  // both identity: a pure (transposing) copy
  };
  auto store_chunk = [&](int d, uint4 v) {
    if (hasB) { *reinterpret_cast<uint4*>(Xs + d) = v; return; }
    const int q = d >> 16, s = d & 0xffff;
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
    const float sc = hasA ? 1.f : p.scale;
#pragma unroll
    for (int j = 0; j < 8; ++j) put_tmp(q, s + j, sc == 1.f ? e[j] : f2bf(bf2f(e[j]) * sc));
  };

  int m = blockIdx.x;
  if (m < p.M && in_regs) load_regs(m);
  const int nout = p.a_out * p.b_out;
  const int col0 = p.col0, ncols = p.ncols > 0 ? p.ncols : nout;

  for (; m < p.M; m += gridDim.x) {
    // ---- stage 0
    if (in_regs) {
#pragma unroll
      for (int c = 0; c < KRON_MAXCH; ++c)
        if (dst[c] >= 0) store_chunk(dst[c], reg[c]);
      if (m + (int)gridDim.x < p.M) load_regs(m + gridDim.x);  // next token's row in flight during the MFMA stages
    } else {
      const bf16_t* xr = xrow(m);
      for (int idx = tid; idx < nch; idx += 256) {
        const int q = (idx * 8) / p.b_in, s = idx * 8 - q * p.b_in;
        store_chunk(hasB ? q * L.xs + s : (q << 16) | s, *reinterpret_cast<const uint4*>(xr + (long)idx * 8));
      }
    }
    __syncthreads();
    // ---- stage 1: tmp[q][o] = sum_s X[q][s] B[o][s]
    if (hasB) {
      const int tq = L.ai16 / 16, to = L.bo16 / 16, ksteps = (L.xs - 8) / 32;
      for (int t = wave; t < tq * to; t += 4) {
        const int q0 = (t % tq) * 16, o0 = (t / tq) * 16;
        const bf16_t* ap = Xs + (q0 + i16) * L.xs + 8 * g;
        const bf16_t* bp = Bs + (o0 + i16) * L.xs + 8 * g;
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < ksteps; ++ks)
          acc = mfma16(*reinterpret_cast<const s16x8_t*>(ap + ks * 32), *reinterpret_cast<const s16x8_t*>(bp + ks * 32), acc);
        // D: rows q0 + 4g + r, col o0 + i16
        const int o = o0 + i16, q = q0 + 4 * g;
        if (hasA) {
          uint2 w;
          w.x = pack2bf(acc[0], acc[1]);
          w.y = pack2bf(acc[2], acc[3]);
          *reinterpret_cast<uint2*>(Tt + o * L.as + q) = w;  // pad rows/cols hold exact zeros (zero operands)
        } else if (o < p.b_out) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (q + r < p.a_in) Os[p.transpose_out ? o * p.a_out + q + r : (q + r) * p.b_out + o] = f2bf(acc[r] * p.scale);
        }
      }
      __syncthreads();
    }
    // ---- stage 2: out[p][o] = scale * sum_q A[p][q] tmp[q][o]
    if (hasA) {
      const int to = L.bo16 / 16, tp = L.ao16 / 16, ksteps = (L.as - 8) / 32;
      for (int t = wave; t < to * tp; t += 4) {
        const int o0 = (t % to) * 16, p0 = (t / to) * 16;
        const bf16_t* ap = Tt + (o0 + i16) * L.as + 8 * g;
        const bf16_t* bp = As + (p0 + i16) * L.as + 8 * g;
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < ksteps; ++ks)
          acc = mfma16(*reinterpret_cast<const s16x8_t*>(ap + ks * 32), *reinterpret_cast<const s16x8_t*>(bp + ks * 32), acc);
        // D: rows o0 + 4g + r, col p0 + i16
        const int pp = p0 + i16, o = o0 + 4 * g;
        if (pp < p.a_out) {
          if (!p.transpose_out && o + 3 < p.b_out) {
            uint2 w;
            w.x = pack2bf(acc[0] * p.scale, acc[1] * p.scale);
            w.y = pack2bf(acc[2] * p.scale, acc[3] * p.scale);
            *reinterpret_cast<uint2*>(Os + pp * p.b_out + o) = w;  // b_out % 8 == 0: 8-byte aligned
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (o + r < p.b_out) Os[p.transpose_out ? (o + r) * p.a_out + pp : pp * p.b_out + o + r] = f2bf(acc[r] * p.scale);
          }
        }
      }
      __syncthreads();
    } else if (!hasB) {
      __syncthreads();
    }
    // ---- stage 3: copy the (window of the) output row out
    bf16_t* orow = const_cast<bf16_t*>(seg_row(p.out, p.ldo, p.out_seg_rows, p.out_seg_stride, m));
    for (int i = tid * 8; i < ncols; i += 256 * 8) {
      uint4 v = *reinterpret_cast<const uint4*>(Os + col0 + i);
      if (p.accumulate) {
        const uint4 old = *reinterpret_cast<const uint4*>(orow + i);
        const bf16_t* a = reinterpret_cast<const bf16_t*>(&v);
        const bf16_t* b = reinterpret_cast<const bf16_t*>(&old);
        uint4 r;
        r.x = pack2bf(bf2f(a[0]) + bf2f(b[0]), bf2f(a[1]) + bf2f(b[1]));
        r.y = pack2bf(bf2f(a[2]) + bf2f(b[2]), bf2f(a[3]) + bf2f(b[3]));
        r.z = pack2bf(bf2f(a[4]) + bf2f(b[4]), bf2f(a[5]) + bf2f(b[5]));
        r.w = pack2bf(bf2f(a[6]) + bf2f(b[6]), bf2f(a[7]) + bf2f(b[7]));
        v = r;
      }
      *reinterpret_cast<uint4*>(orow + i) = v;
    }
    // the next iteration's first LDS writes (Xs / Tt / Os) are ordered behind this token's readers by the barriers above:
    // Xs is last read before barrier B, Tt before barrier C; Os is rewritten only after the next token's barrier A or B,
    // which every wave reaches after finishing this copy-out.  (Os written directly in stage 0 — both factors identity —
    // needs its own fence.)
    if (!hasA && !hasB) __syncthreads();
  }
}

// W[r][c] += alpha * A[r / b_rows][c / b_cols] * B[r % b_rows][c % b_cols]   (W bf16 [a_rows*b_rows, a_cols*b_cols], A / B fp32):
// the LoKr delta kron(lokr_w1, lokr_w2) * scale merged into a base weight (toolkit/models/lokr.py:62-69, 261-309).
__global__ __launch_bounds__(256) void kron_merge_kernel(bf16_t* W, long ldw, const float* A, const float* B, int a_rows, int a_cols,
                                                          int b_rows, int b_cols, float alpha) {
  const long cols8 = (long)a_cols * b_cols / 8;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)a_rows * b_rows * cols8) return;
  const int r = (int)(i / cols8), c = (int)(i - (long)r * cols8) * 8;
  const int p = r / b_rows, o = r - p * b_rows;
  const int q = c / b_cols, s0 = c - q * b_cols;  // b_cols % 8 == 0: the 8 columns share one A entry
  const float a = alpha * A[(long)p * a_cols + q];
  uint4 v = *reinterpret_cast<uint4*>(W + (long)r * ldw + c);
  const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
  float f[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = bf2f(e[j]) + a * B[(long)o * b_cols + s0 + j];
  uint4 w;
  w.x = pack2bf(f[0], f[1]); w.y = pack2bf(f[2], f[3]); w.z = pack2bf(f[4], f[5]); w.w = pack2bf(f[6], f[7]);
  *reinterpret_cast<uint4*>(W + (long)r * ldw + c) = w;
}

}  // namespace

extern "C" int aitk_kron_merge(aitk_bf16* W, int64_t ldw, const float* A, const float* B, int32_t a_rows, int32_t a_cols, int32_t b_rows,
                               int32_t b_cols, float alpha, aitk_stream_t stream) {
  if (!W || !A || !B || a_rows <= 0 || a_cols <= 0 || b_rows <= 0 || b_cols <= 0 || (b_cols % 8)) return AITK_ERR_SHAPE;
  if ((ldw % 8) || ((uintptr_t)W & 15)) return AITK_ERR_ALIGN;
  const long n = (long)a_rows * b_rows * ((long)a_cols * b_cols / 8);
  hipLaunchKernelGGL(kron_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, (long)ldw, A, B, a_rows, a_cols,
                     b_rows, b_cols, alpha);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

extern "C" int aitk_kron_apply(const AitkKronApplyArgs* a, aitk_stream_t stream) {
  if (!a || a->M <= 0 || a->a_in <= 0 || a->b_in <= 0 || a->a_out <= 0 || a->b_out <= 0) return AITK_ERR_SHAPE;
  if ((a->b_in % 8) || (a->b_out % 8)) return AITK_ERR_SHAPE;
  if (!a->A && a->a_out != a->a_in) return AITK_ERR_SHAPE;
  if (!a->B && a->b_out != a->b_in) return AITK_ERR_SHAPE;
  if (a->a_in > 0x7fff || a->b_in > 0xffff) return AITK_ERR_SHAPE;
  if (!a->x || !a->out) return AITK_ERR_ARG;
  if ((a->ldx % 8) || (a->ldo % 8) || (a->x_seg_stride % 8) || (a->out_seg_stride % 8)) return AITK_ERR_ALIGN;
  if (((uintptr_t)a->x | (uintptr_t)a->out | (uintptr_t)a->B) & 15) return AITK_ERR_ALIGN;  // A is read element-wise
  const int nout = a->a_out * a->b_out;
  if (a->col0 < 0 || (a->col0 % 8) || a->ncols < 0 || (a->ncols % 8) || a->col0 + a->ncols > nout) return AITK_ERR_SHAPE;
  if (a->ncols == 0 && a->col0 != 0) return AITK_ERR_SHAPE;
  const KronLayout L = kron_layout(a->a_in, a->b_in, a->a_out, a->b_out, a->A != nullptr, a->B != nullptr);
  const size_t lds_bytes = (size_t)L.total * 2;
  if (lds_bytes > 160 * 1024) return AITK_ERR_SHAPE;
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
    for (const void* f : {reinterpret_cast<const void*>(kron_apply_kernel<2>), reinterpret_cast<const void*>(kron_apply_kernel<8>)})
      if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  int per_cu = (int)((160 * 1024) / (lds_bytes + 1024));
  per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
  long grid = (long)n_cu * per_cu;
  if (grid > a->M) grid = a->M;
  if (a->a_in * a->b_in <= 2 * 256 * 8)
    hipLaunchKernelGGL(kron_apply_kernel<2>, dim3((unsigned)grid), dim3(256), lds_bytes, (hipStream_t)stream, *a);
  else
    hipLaunchKernelGGL(kron_apply_kernel<8>, dim3((unsigned)grid), dim3(256), lds_bytes, (hipStream_t)stream, *a);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}
