// Flash-style attention forward / backward for the DiT joint attention (non-causal, no mask, head_dim 128),
// bf16 in/out, fp32 softmax statistics, v_mfma_f32_32x32x16_bf16 throughout (gfx950).
//
// Replaces F.scaled_dot_product_attention inside the diffusers Flux attention processor that the reference runs
// (call order restated in toolkit/models/flux_sage_attn.py:76-93; wan path toolkit/models/wan21/wan_attn.py:67-75)
// and its autograd backward.
//
// Layout: Q,K,V,O are [B, S, H, 128] views (row stride ld elements, head h at column h*128) so they can alias the
// GEMM outputs directly.  LSE is kept in the scaled log2 domain:  L2 = max2 + log2(sum exp2(s2 - max2)),
// s2 = (q.k) * softmax_scale * log2(e).
//
// Structure (guide Appendix B "Fused attention prefill"): swapped QK^T — each wave computes S^T = K Q^T so a lane owns
// ONE query row (column l&31 of the 32x32 accumulator); softmax statistics, rescale and normalisation are per-lane
// scalars, and P^T feeds the next MFMA straight from registers using a permuted contraction order
// (slot (h,e) <-> row 8*(e>>2)+4h+(e&3) of each 16-row step).  Operands whose contraction index is the row of a
// row-major LDS tile (V in PV, K in dQ, Q/dO in dK/dV) are consumed with ds_read_b64_tr_b16.
#include "common.h"
#include "aitk_args.h"

#define KP 136  // LDS pitch (elements) for tiles read with ds_read_b128 fragments (conflict-free)
#define VP 144  // LDS pitch for tiles read only through tr16

__device__ __forceinline__ s16x4_t tr16a(const bf16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
}
// MFMA operand whose row/col index is the tile row and contraction index the tile column
__device__ __forceinline__ s16x8_t frag_rm(const bf16_t* tile, int pitch, int row0, int k0, int lane) {
  return *reinterpret_cast<const s16x8_t*>(tile + (row0 + (lane & 31)) * pitch + k0 + 8 * (lane >> 5));
}
// MFMA operand whose contraction index is the tile ROW, in the permuted order that matches packed accumulator
// registers: lane (i = l&31, h = l>>5) receives rows {kb+4h+0..3, kb+8+4h+0..3} of column col0+i.
__device__ __forceinline__ s16x8_t frag_tr_perm(const bf16_t* tile, int pitch, int kb, int col0, int lane) {
  const int h = lane >> 5, gq = (lane >> 4) & 1, i = lane & 15;
  const bf16_t* p = tile + (kb + 4 * h + (i >> 2)) * pitch + col0 + gq * 16 + (i & 3) * 4;
  s16x4_t lo = tr16a(p);
  s16x4_t hi = tr16a(p + 8 * pitch);
  s16x8_t f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}
// accumulator registers [base, base+8) -> bf16 operand (slot e <-> register base+e)
__device__ __forceinline__ s16x8_t pack_acc8(const f32x16_t& a, int base) {
  uint4 u;
  u.x = pack2bf(a[base + 0], a[base + 1]);
  u.y = pack2bf(a[base + 2], a[base + 3]);
  u.z = pack2bf(a[base + 4], a[base + 5]);
  u.w = pack2bf(a[base + 6], a[base + 7]);
  return __builtin_bit_cast(s16x8_t, u);
}
__device__ __forceinline__ f32x16_t zero16() {
  f32x16_t z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}
// row index inside a 32-row accumulator block held by (register r, lane half h)
__device__ __forceinline__ int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// stage a [ROWS][128] bf16 tile (rows row0.. of one (b,h) slice) into registers / LDS, zero-filling rows >= S
template <int ROWS>
struct TileStager {
  static constexpr int CH = ROWS * 16 / 256;  // 16-B chunks per thread
  uint4 reg[CH];
  __device__ __forceinline__ void load(const bf16_t* base, long ld, int row0, int S, int tid, bool zero_oob) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int q = tid + 256 * i;
      const int row = q >> 4, ch = q & 15;
      int r = row0 + row;
      const bool oob = r >= S;
      if (oob) r = S - 1;
      uint4 v = *reinterpret_cast<const uint4*>(base + (long)r * ld + ch * 8);
      if (oob && zero_oob) v = make_uint4(0, 0, 0, 0);
      reg[i] = v;
    }
  }
  __device__ __forceinline__ void store(bf16_t* tile, int pitch, int tid) const {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int q = tid + 256 * i;
      const int row = q >> 4, ch = q & 15;
      *reinterpret_cast<uint4*>(tile + row * pitch + ch * 8) = reg[i];
    }
  }
};

// ============================================================================================ forward
// grid (ceil(S/128), H, B); 4 waves x 32 query rows; KV tiles of 64 rows, double-buffered in LDS.
__global__ __launch_bounds__(256) void attn_fwd_kernel(AitkAttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* kt[2];
  bf16_t* vt[2];
  kt[0] = reinterpret_cast<bf16_t*>(smem);
  vt[0] = kt[0] + 64 * KP;
  kt[1] = vt[0] + 64 * VP;
  vt[1] = kt[1] + 64 * KP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int hd = blockIdx.y, b = blockIdx.z;
  const int S = p.S;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const bf16_t* Qb = p.Q + (long)b * S * p.ldq + hd * 128;
  const bf16_t* Kb = p.K + (long)b * S * p.ldk + hd * 128;
  const bf16_t* Vb = p.V + (long)b * S * p.ldv + hd * 128;

  // Q^T operand fragments: lane holds Q[q0 + l31][16 ks + 8h .. +8]
  s16x8_t qf[8];
  {
    const int qr = min(q0 + l31, S - 1);
    const bf16_t* qp = Qb + (long)qr * p.ldq + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const s16x8_t*>(qp + 16 * ks);
  }
  f32x16_t o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) o[d] = zero16();
  float m_run = -INFINITY, l_run = 0.f;
  const float c2 = p.scale * 1.4426950408889634f;

  const int ntiles = (S + 63) / 64;
  TileStager<64> ks_, vs_;
  ks_.load(Kb, p.ldk, 0, S, tid, false);
  vs_.load(Vb, p.ldv, 0, S, tid, true);
  ks_.store(kt[0], KP, tid);
  vs_.store(vt[0], VP, tid);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntiles) {
      ks_.load(Kb, p.ldk, (t + 1) * 64, S, tid, false);
      vs_.load(Vb, p.ldv, (t + 1) * 64, S, tid, true);
    }
    // ---- S^T = K Q^T : two 32-kv blocks
    f32x16_t s[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      s[j] = zero16();
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) s[j] = mfma32(frag_rm(kt[cur], KP, 32 * j, 16 * ks, lane), qf[ks], s[j]);
    }
    // ---- online softmax (lane owns query row q0+l31; halves h hold disjoint kv subsets)
    const int kv0 = t * 64;
    float mt = -INFINITY;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = s[j][r] * c2;
        if (kv0 + 32 * j + crow(r, h) >= S) v = -INFINITY;
        s[j][r] = v;
        mt = fmaxf(mt, v);
      }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float ps = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float e = __builtin_amdgcn_exp2f(s[j][r] - m_new);
        s[j][r] = e;
        ps += e;
      }
    l_run = l_run * alpha + ps;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
    // ---- O^T += V^T P^T : 4 d-blocks x 4 k-steps of 16 kv
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const s16x8_t pf = pack_acc8(s[kk >> 1], 8 * (kk & 1));
#pragma unroll
      for (int d = 0; d < 4; ++d) o[d] = mfma32(frag_tr_perm(vt[cur], VP, 16 * kk, 32 * d, lane), pf, o[d]);
    }
    if (t + 1 < ntiles) {
      ks_.store(kt[cur ^ 1], KP, tid);
      vs_.store(vt[cur ^ 1], VP, tid);
    }
    __syncthreads();
  }
  // ---- epilogue
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int q = q0 + l31;
  if (q < S) {
    bf16_t* op = p.O + ((long)b * S + q) * p.ldo + hd * 128;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 u;
        u.x = pack2bf(o[d][4 * g + 0] * inv, o[d][4 * g + 1] * inv);
        u.y = pack2bf(o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(op + 32 * d + 8 * g + 4 * h) = u;
      }
    if (h == 0) p.LSE[((long)b * p.H + hd) * S + q] = m_run + log2f(l_tot);
  }
}

// ============================================================================================ delta = rowsum(dO * O)
__global__ __launch_bounds__(256) void attn_delta_kernel(AitkAttnArgs p) {
  const int sub = threadIdx.x & 15;
  const long pair = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const long npairs = (long)p.B * p.S * p.H;
  if (pair >= npairs) return;
  const int hd = (int)(pair % p.H);
  const long tok = pair / p.H;
  const int s = (int)(tok % p.S), b = (int)(tok / p.S);
  uint4 a = *reinterpret_cast<const uint4*>(p.O + tok * p.ldo + hd * 128 + sub * 8);
  uint4 g = *reinterpret_cast<const uint4*>(p.dO + tok * p.lddo + hd * 128 + sub * 8);
  float acc = 0.f;
  acc += bf2f(a.x & 0xffff) * bf2f(g.x & 0xffff) + bf2f(a.x >> 16) * bf2f(g.x >> 16);
  acc += bf2f(a.y & 0xffff) * bf2f(g.y & 0xffff) + bf2f(a.y >> 16) * bf2f(g.y >> 16);
  acc += bf2f(a.z & 0xffff) * bf2f(g.z & 0xffff) + bf2f(a.z >> 16) * bf2f(g.z >> 16);
  acc += bf2f(a.w & 0xffff) * bf2f(g.w & 0xffff) + bf2f(a.w >> 16) * bf2f(g.w >> 16);
  acc += __shfl_xor(acc, 8, 64);
  acc += __shfl_xor(acc, 4, 64);
  acc += __shfl_xor(acc, 2, 64);
  acc += __shfl_xor(acc, 1, 64);
  if (sub == 0) p.delta[((long)b * p.H + hd) * p.S + s] = acc;
}

// ============================================================================================ backward: dK, dV
// grid (ceil(S/128), H, B); wave w owns kv rows [kv0 + 32 w, +32) (K, V fragments in registers, dK/dV accumulators);
// loops over query tiles of 32 rows staged (Q, dO, L2, delta) in LDS.
// S[q][kv] = Q K^T (lane owns one kv column), P = exp2(S c2 - L2[q]), dP = dO V^T, dS = P (dP - delta[q]);
// dV += P^T dO, dK += scale * dS^T Q  (Q/dO consumed via tr16 with the permuted order of the packed P / dS registers).
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(AitkAttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* qt[2];
  bf16_t* dot_[2];
  float* lt[2];
  float* dt[2];
  qt[0] = reinterpret_cast<bf16_t*>(smem);
  dot_[0] = qt[0] + 32 * KP;
  qt[1] = dot_[0] + 32 * KP;
  dot_[1] = qt[1] + 32 * KP;
  lt[0] = reinterpret_cast<float*>(dot_[1] + 32 * KP);
  dt[0] = lt[0] + 32;
  lt[1] = dt[0] + 32;
  dt[1] = lt[1] + 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int hd = blockIdx.y, b = blockIdx.z;
  const int S = p.S;
  const int kvw = blockIdx.x * 128 + wave * 32;
  const bf16_t* Qb = p.Q + (long)b * S * p.ldq + hd * 128;
  const bf16_t* Kb = p.K + (long)b * S * p.ldk + hd * 128;
  const bf16_t* Vb = p.V + (long)b * S * p.ldv + hd * 128;
  const bf16_t* dOb = p.dO + (long)b * S * p.lddo + hd * 128;
  const float* Lb = p.LSE + ((long)b * p.H + hd) * S;
  const float* Db = p.delta + ((long)b * p.H + hd) * S;

  s16x8_t kf[8], vf[8];
  {
    const int kr = min(kvw + l31, S - 1);
    const bf16_t* kp = Kb + (long)kr * p.ldk + 8 * h;
    const bf16_t* vp = Vb + (long)kr * p.ldv + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      kf[ks] = *reinterpret_cast<const s16x8_t*>(kp + 16 * ks);
      vf[ks] = *reinterpret_cast<const s16x8_t*>(vp + 16 * ks);
    }
  }
  f32x16_t dk[4], dv[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    dk[d] = zero16();
    dv[d] = zero16();
  }
  const float c2 = p.scale * 1.4426950408889634f;
  const int ntiles = (S + 31) / 32;
  TileStager<32> qs_, ds_;
  float lreg = 0.f;
  auto load_stats = [&](int t) {
    if (tid < 64) {
      const int q = t * 32 + (tid & 31);
      if (tid < 32) lreg = q < S ? Lb[q] : INFINITY;
      else lreg = q < S ? Db[q] : 0.f;
    }
  };
  auto store_stats = [&](int buf) {
    if (tid < 32) lt[buf][tid] = lreg;
    else if (tid < 64) dt[buf][tid - 32] = lreg;
  };
  qs_.load(Qb, p.ldq, 0, S, tid, true);
  ds_.load(dOb, p.lddo, 0, S, tid, true);
  load_stats(0);
  qs_.store(qt[0], KP, tid);
  ds_.store(dot_[0], KP, tid);
  store_stats(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntiles) {
      qs_.load(Qb, p.ldq, (t + 1) * 32, S, tid, true);
      ds_.load(dOb, p.lddo, (t + 1) * 32, S, tid, true);
      load_stats(t + 1);
    }
    // S[q][kv] and dP[q][kv] : D rows = q (tile rows), cols = kv (lane l31)
    f32x16_t s = zero16(), dp = zero16();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      s = mfma32(frag_rm(qt[cur], KP, 0, 16 * ks, lane), kf[ks], s);
      dp = mfma32(frag_rm(dot_[cur], KP, 0, 16 * ks, lane), vf[ks], dp);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = crow(r, h);
      const float pr = __builtin_amdgcn_exp2f(s[r] * c2 - lt[cur][qi]);
      s[r] = pr;
      dp[r] = pr * (dp[r] - dt[cur][qi]);
    }
    // dV[kv][d] += sum_q P[q][kv] dO[q][d] ; dK[kv][d] += sum_q dS[q][kv] Q[q][d]
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const s16x8_t pf = pack_acc8(s, 8 * kk);
      const s16x8_t df = pack_acc8(dp, 8 * kk);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        dv[d] = mfma32(pf, frag_tr_perm(dot_[cur], KP, 16 * kk, 32 * d, lane), dv[d]);
        dk[d] = mfma32(df, frag_tr_perm(qt[cur], KP, 16 * kk, 32 * d, lane), dk[d]);
      }
    }
    if (t + 1 < ntiles) {
      qs_.store(qt[cur ^ 1], KP, tid);
      ds_.store(dot_[cur ^ 1], KP, tid);
      store_stats(cur ^ 1);
    }
    __syncthreads();
  }
  // dK/dV accumulators: D[i = kv (A rows = lane l31 of P^T)...]  -> rows = kv?  see layout note below
  // mfma32(a = P^T frag (rows kv = l31), b = dO frag (cols d = l31)) gives D[i = kv][j = d]:
  // lane holds column d = 32*blk + l31 and rows kv = crow(r, h).
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kv = kvw + crow(r, h);
      if (kv < S) {
        const long off = ((long)b * S + kv);
        p.dK[off * p.lddk + hd * 128 + 32 * d + l31] = f2bf(dk[d][r] * p.scale);
        p.dV[off * p.lddv + hd * 128 + 32 * d + l31] = f2bf(dv[d][r]);
      }
    }
}

// ============================================================================================ backward: dQ
// grid (ceil(S/128), H, B); wave w owns query rows [q0 + 32 w, +32) (Q, dO fragments in registers, dQ^T accumulators);
// loops over KV tiles of 64 rows (K, V row-major in LDS).  S^T = K Q^T, dP^T = V dO^T, dS^T = P (dP^T - delta[q]);
// dQ^T[d][q] += sum_kv K^T[d][kv] dS^T[kv][q]  (K through tr16, dS^T straight from registers).
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AitkAttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* kt[2];
  bf16_t* vt[2];
  kt[0] = reinterpret_cast<bf16_t*>(smem);
  vt[0] = kt[0] + 64 * KP;
  kt[1] = vt[0] + 64 * KP;
  vt[1] = kt[1] + 64 * KP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int hd = blockIdx.y, b = blockIdx.z;
  const int S = p.S;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const bf16_t* Qb = p.Q + (long)b * S * p.ldq + hd * 128;
  const bf16_t* Kb = p.K + (long)b * S * p.ldk + hd * 128;
  const bf16_t* Vb = p.V + (long)b * S * p.ldv + hd * 128;
  const bf16_t* dOb = p.dO + (long)b * S * p.lddo + hd * 128;
  const int qr = min(q0 + l31, S - 1);
  s16x8_t qf[8], gf[8];
  {
    const bf16_t* qp = Qb + (long)qr * p.ldq + 8 * h;
    const bf16_t* gp = dOb + (long)qr * p.lddo + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      qf[ks] = *reinterpret_cast<const s16x8_t*>(qp + 16 * ks);
      gf[ks] = *reinterpret_cast<const s16x8_t*>(gp + 16 * ks);
    }
  }
  const float L2 = p.LSE[((long)b * p.H + hd) * S + qr];
  const float dl = p.delta[((long)b * p.H + hd) * S + qr];
  f32x16_t dq[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) dq[d] = zero16();
  const float c2 = p.scale * 1.4426950408889634f;
  const int ntiles = (S + 63) / 64;
  TileStager<64> ks_, vs_;
  ks_.load(Kb, p.ldk, 0, S, tid, true);
  vs_.load(Vb, p.ldv, 0, S, tid, true);
  ks_.store(kt[0], KP, tid);
  vs_.store(vt[0], KP, tid);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntiles) {
      ks_.load(Kb, p.ldk, (t + 1) * 64, S, tid, true);
      vs_.load(Vb, p.ldv, (t + 1) * 64, S, tid, true);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x16_t s = zero16(), dp = zero16();
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        s = mfma32(frag_rm(kt[cur], KP, 32 * j, 16 * ks, lane), qf[ks], s);
        dp = mfma32(frag_rm(vt[cur], KP, 32 * j, 16 * ks, lane), gf[ks], dp);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kv = t * 64 + 32 * j + crow(r, h);
        const float pr = kv < S ? __builtin_amdgcn_exp2f(s[r] * c2 - L2) : 0.f;
        dp[r] = pr * (dp[r] - dl);
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const s16x8_t df = pack_acc8(dp, 8 * kk);
#pragma unroll
        for (int d = 0; d < 4; ++d) dq[d] = mfma32(frag_tr_perm(kt[cur], KP, 32 * j + 16 * kk, 32 * d, lane), df, dq[d]);
      }
    }
    if (t + 1 < ntiles) {
      ks_.store(kt[cur ^ 1], KP, tid);
      vs_.store(vt[cur ^ 1], KP, tid);
    }
    __syncthreads();
  }
  const int q = q0 + l31;
  if (q < S) {
    bf16_t* op = p.dQ + ((long)b * S + q) * p.lddq + hd * 128;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 u;
        u.x = pack2bf(dq[d][4 * g + 0] * p.scale, dq[d][4 * g + 1] * p.scale);
        u.y = pack2bf(dq[d][4 * g + 2] * p.scale, dq[d][4 * g + 3] * p.scale);
        *reinterpret_cast<uint2*>(op + 32 * d + 8 * g + 4 * h) = u;
      }
  }
}

static int attn_check(const AitkAttnArgs* a) {
  if (!a || a->B <= 0 || a->H <= 0 || a->S <= 0 || a->D != 128) return AITK_ERR_SHAPE;
  if ((a->ldq % 8) || (a->ldk % 8) || (a->ldv % 8) || (a->ldo % 8)) return AITK_ERR_ALIGN;
  return AITK_OK;
}

extern "C" int aitk_attn_fwd(const AitkAttnArgs* a, aitk_stream_t stream) {
  int rc = attn_check(a);
  if (rc) return rc;
  if (!a->Q || !a->K || !a->V || !a->O || !a->LSE) return AITK_ERR_ARG;
  const size_t lds = 2 * (64 * KP + 64 * VP) * sizeof(bf16_t);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  dim3 grid((a->S + 127) / 128, a->H, a->B);
  hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(256), lds, (hipStream_t)stream, *a);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

extern "C" int aitk_attn_bwd(const AitkAttnArgs* a, aitk_stream_t stream) {
  int rc = attn_check(a);
  if (rc) return rc;
  if (!a->Q || !a->K || !a->V || !a->O || !a->LSE || !a->dO || !a->dQ || !a->dK || !a->dV || !a->delta) return AITK_ERR_ARG;
  if ((a->lddo % 8) || (a->lddq % 8) || (a->lddk % 8) || (a->lddv % 8)) return AITK_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const long npairs = (long)a->B * a->S * a->H;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((npairs + 15) / 16)), dim3(256), 0, s, *a);
  AITK_LAUNCH_CHECK();
  dim3 grid((a->S + 127) / 128, a->H, a->B);
  const size_t lds1 = 4 * 32 * KP * sizeof(bf16_t) + 4 * 32 * sizeof(float);
  hipLaunchKernelGGL(attn_bwd_dkdv_kernel, grid, dim3(256), lds1, s, *a);
  AITK_LAUNCH_CHECK();
  const size_t lds2 = 4 * 64 * KP * sizeof(bf16_t);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    attr_set = true;
  }
  hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), lds2, s, *a);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}
