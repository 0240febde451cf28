// Flash-style attention forward / backward for the DiT joint attention (non-causal, no mask, head_dim 128),
// bf16 in/out, fp32 softmax statistics, v_mfma_f32_32x32x16_bf16 throughout (gfx950).
//
// Replaces F.scaled_dot_product_attention inside the diffusers Flux attention processor that the reference runs
// (call order restated in toolkit/models/flux_sage_attn.py:76-93; wan path toolkit/models/wan21/wan_attn.py:67-75)
// and its autograd backward.
//
// Layout: Q,K,V,O are [B, S, H, 128] views (row stride ld elements, head h at column h*128) so they can alias the
// GEMM outputs directly.  Heads of 64 / 96 columns can also be read where the projections wrote them (AitkAttnArgs.hstride = head
// width: head h at column h*hstride, no padded copies): tiles still fetch 128 columns per row — the surplus is the neighbouring head,
// never contracted (KS / DB below) — and the buffer descriptor ends each (batch, head) slice at its own last column.  LSE is kept in the scaled log2 domain:  L2 = max2 + log2(sum exp2(s2 - max2)),
// s2 = (q.k) * softmax_scale * log2(e).
//
// Structure (guide Appendix B "Fused attention prefill"): swapped QK^T — each wave computes S^T = K Q^T so a lane owns
// ONE query row (column l&31 of the 32x32 accumulator); softmax statistics, rescale and normalisation are per-lane
// scalars, and P^T feeds the next MFMA straight from registers using a permuted contraction order
// (slot (h,e) <-> row 8*(e>>2)+4h+(e&3) of each 16-row step).  Operands whose contraction index is the row of a
// row-major LDS tile (V in PV, K in dQ, Q/dO in dK/dV) are consumed with ds_read_b64_tr_b16.
#include <cstdlib>
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

// ---- swizzled, unpadded tiles: [rows][128] bf16, 256-B rows, physical 16-B chunk = logical chunk ^ (row & 15).
// Conflict-free for ds_read_b128 fragments and for tr16 reads, and lane-linear so LDS-DMA can fill them
// (the XOR goes on the global SOURCE address, guide rule 21).
// All tile pointers are address_space(3): a generic pointer (e.g. selected from an array of buffers) would turn the
// fragment loads into flat loads that wait on vmcnt and drain the LDS-DMA prefetch.
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ s16x8_t frag_rm_sw(const lds_char* tile, int row0, int k0, int lane) {
  const int row = row0 + (lane & 31);
  const int c = (k0 >> 3) + (lane >> 5);
  return *reinterpret_cast<const __attribute__((address_space(3))) s16x8_t*>(tile + row * 256 + ((c ^ (row & 15)) << 4));
}
__device__ __forceinline__ s16x4_t tr16l(const lds_char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
}
__device__ __forceinline__ s16x8_t frag_tr_perm_sw(const lds_char* tile, int kb, int col0, int lane) {
  const int h = lane >> 5, gq = (lane >> 4) & 1, i = lane & 15;
  const int row = kb + 4 * h + (i >> 2);
  const int col = col0 + gq * 16 + (i & 3) * 4;
  const int c = col >> 3, off = (col & 7) * 2;
  const lds_char* base = tile;
  s16x4_t lo = tr16l(base + row * 256 + ((c ^ (row & 15)) << 4) + off);
  const int row2 = row + 8;
  s16x4_t hi = tr16l(base + row2 * 256 + ((c ^ (row2 & 15)) << 4) + off);
  s16x8_t f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}
// ---- LDS-DMA through a raw buffer descriptor (buffer_load_dwordx4 ... lds, as in gemm8.hip): the (batch, head) slice [rows][128] of an
// operand is one buffer whose num_records ends with its last valid row, so rows >= nrows arrive as ZEROS (a zero K row scores 0 and is
// masked to -inf by the tail code, a zero Q / dO row meets L2 = +inf -> P = 0) and a piece costs one s_mov m0 + one VALU add + the load
// itself.  (The flat global_load_lds form needed a 64-bit per-lane address: ~10 VALU per 1-KiB piece, 80 per KV tile of the forward
// kernel, and the compiler drained vmcnt in front of every LDS read that followed a piece.)  The compiler does not see these loads:
// every tile is published by an explicit s_waitcnt vmcnt(0) in front of the workgroup barrier.
typedef int v4i_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4i_t slice_srd(const bf16_t* base, long ld, int nrows, int width = 128) {
  const unsigned long long a = (unsigned long long)base;
  v4i_t r;
  r.x = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
  r.y = __builtin_amdgcn_readfirstlane((int)(a >> 32));
  r.z = __builtin_amdgcn_readfirstlane((int)(((long)(nrows - 1) * ld + width) * 2));
  r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ v4i_t slice_srd_bytes(const void* base, long nbytes) {
  const unsigned long long a = (unsigned long long)base;
  v4i_t r;
  r.x = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
  r.y = __builtin_amdgcn_readfirstlane((int)(a >> 32));
  r.z = __builtin_amdgcn_readfirstlane((int)nbytes);
  r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ void dma16(unsigned lds_dst, unsigned voff, const v4i_t& srd) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_dst), "v"(voff), "s"(srd) : "memory");
}
// s_waitcnt vmcnt(0) as the BUILTIN (expcnt / lgkmcnt fields at their no-wait maxima): the waitcnt pass reads it and clears its own scoreboard
// of outstanding global loads (Q / K / V fragments loaded at kernel entry) — with an asm wait it kept counting them down inside the tile loop
// (vmcnt(7) .. vmcnt(0) in front of the first MFMAs), i.e. it drained the just-issued pieces of the next tile
#define DMA_WAIT_ALL() do { __builtin_amdgcn_s_waitcnt(0x0F70); asm volatile("" ::: "memory"); } while (0)
// per-lane byte offsets (relative to the tile's first row) of this wave's four 1-KiB pieces of a [64][128] tile in the swizzled
// row-major layout: piece ins = wave + 4 ii covers rows 4 ins .. 4 ins + 3, lane -> (row, physical chunk), source chunk = chunk ^ (row & 15)
__device__ __forceinline__ void rm_voff(unsigned (&v)[4], long ld, int wave, int lane) {
#pragma unroll
  for (int ii = 0; ii < 4; ++ii) {
    const int row = (wave + 4 * ii) * 4 + (lane >> 4);
    v[ii] = (unsigned)((row * ld + (((lane & 15) ^ (row & 15)) << 3)) * 2);
  }
}
__device__ __forceinline__ void dma_rm64(lds_char* tile, const unsigned (&v)[4], unsigned tile_off, const v4i_t& srd, int wave) {
#pragma unroll
  for (int ii = 0; ii < 4; ++ii)
    dma16(__builtin_amdgcn_readfirstlane((unsigned)(size_t)tile + (wave + 4 * ii) * 1024), v[ii] + tile_off, srd);
}


// ---- sub-tiled tiles for operands consumed through ds_read_b64_tr_b16 (PMC: row-major tiles cost ~6 conflict cycles per
// tr16 instruction, XOR swizzles do not help the transpose read — guide T10).  Layout [8 column blocks][64 rows][16 cols]:
// a 16-lane group's 4 rows x 32 B are 128 contiguous bytes, the two groups of a 32-lane half sit 2176 B (= 128 mod 256)
// apart -> conflict-free.  The two 16-B halves of a row are swapped for rows with bit 3 set so the ds_read_b128 row
// fragments of the same tile stay conflict-free too.  Filled by LDS-DMA (1 KiB = 32 rows of one column block).
#define SUBP 2176
#define SUBTILE_BYTES (8 * SUBP)
__device__ __forceinline__ s16x8_t frag_rm_st(const lds_char* tile, int row0, int k0, int lane) {
  const int r = row0 + (lane & 31);
  const int c = (k0 >> 3) + (lane >> 5);
  return *reinterpret_cast<const __attribute__((address_space(3))) s16x8_t*>(tile + (c >> 1) * SUBP + r * 32 + (((c & 1) ^ ((r >> 3) & 1)) << 4));
}
__device__ __forceinline__ s16x8_t frag_tr_perm_st(const lds_char* tile, int kb, int col0, int lane) {
  const int h = lane >> 5, gq = (lane >> 4) & 1, i = lane & 15;
  const int row = kb + 4 * h + (i >> 2);
  const int sub = (col0 >> 4) + gq;
  const int lh = (i & 3) >> 1, piece = (i & 1) * 8;
  const lds_char* base = tile + sub * SUBP + piece;
  s16x4_t lo = tr16l(base + row * 32 + ((lh ^ ((row >> 3) & 1)) << 4));
  const int row2 = row + 8;
  s16x4_t hi = tr16l(base + row2 * 32 + ((lh ^ ((row2 >> 3) & 1)) << 4));
  s16x8_t f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}
// sub-tiled 64-row tile: piece ins = wave + 4 ii is column block ins >> 1, row half ins & 1 (32 rows x 32 B)
__device__ __forceinline__ void st_voff(unsigned (&v)[4], long ld, int wave, int lane) {
#pragma unroll
  for (int ii = 0; ii < 4; ++ii) {
    const int ins = wave + 4 * ii;
    const int sub = ins >> 1, r = (ins & 1) * 32 + (lane >> 1);
    const int lh = (lane & 1) ^ ((r >> 3) & 1);
    v[ii] = (unsigned)((r * ld + (2 * sub + lh) * 8) * 2);
  }
}
__device__ __forceinline__ void dma_st64(lds_char* tile, const unsigned (&v)[4], unsigned tile_off, const v4i_t& srd, int wave) {
#pragma unroll
  for (int ii = 0; ii < 4; ++ii) {
    const int ins = wave + 4 * ii;
    dma16(__builtin_amdgcn_readfirstlane((unsigned)(size_t)tile + (ins >> 1) * SUBP + (ins & 1) * 1024), v[ii] + tile_off, srd);
  }
}

// ---- tr16 reads hipcc does not see (guide §5.7 form ii): the builtin makes SIInsertWaitcnts drain vmcnt (= the in-flight
// LDS-DMA prefetch of the NEXT tile) before the first transpose read of every iteration; the asm form leaves the prefetch in
// flight for the whole iteration.  Safe because the tile being read was published by the previous barrier.  The matching
// lgkmcnt wait names every destination register ("+v") and is followed by sched_barrier(0) (rule 18).
__device__ __forceinline__ void tr16_issue(s16x4_t& out, const lds_char* p) {
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(out) : "v"((unsigned)(unsigned long)p));
}
__device__ __forceinline__ void frag_tr_perm_st_issue(s16x4_t& lo, s16x4_t& hi, const lds_char* tile, int kb, int col0, int lane) {
  const int h = lane >> 5, gq = (lane >> 4) & 1, i = lane & 15;
  const int row = kb + 4 * h + (i >> 2);
  const int sub = (col0 >> 4) + gq;
  const int lh = (i & 3) >> 1, piece = (i & 1) * 8;
  const lds_char* base = tile + sub * SUBP + piece;
  tr16_issue(lo, base + row * 32 + ((lh ^ ((row >> 3) & 1)) << 4));
  const int row2 = row + 8;
  tr16_issue(hi, base + row2 * 32 + ((lh ^ ((row2 >> 3) & 1)) << 4));
}
template <int OFF>
__device__ __forceinline__ void tr16_issue_off(s16x4_t& out, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(out) : "v"(addr), "n"(OFF));
}
#define TR_PIN8(a) "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
__device__ __forceinline__ s16x8_t join_lohi(const s16x4_t& lo, const s16x4_t& hi) {
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}


// ---- workgroup -> (row tile, head, batch) with the row tiles of ONE (batch, head) kept on ONE XCD.  The grid is 1-D; the dispatcher places
// block i on XCD i % 8 (guide "Workgroups, grid, and XCD partitioning"), each XCD has its own 4-MiB L2, and every workgroup of a (batch,
// head) streams the same K / V (or Q / dO) slice: 2.4 MB at 4608 tokens.  In plain blockIdx order the 36 row tiles of a head are dealt round
// the 8 XCDs, every L2 sees every head in flight (14 heads x 2.4 MB against 4 MiB) and the slices are re-fetched from the fabric —
// PMC round 3: 526 MB fetched per B = 1 launch against 85 MB of operands.  Here XCD k works through a contiguous range of the
// (batch, head, tile) list: ~2 heads in flight per L2.  A performance mapping only: any bijection is correct.
__host__ __device__ __forceinline__ void attn_wg_coords_of(int n, int id, int ntiles, int H, int& tile, int& hd, int& b) {
  const int xcd = id & 7, slot = id >> 3;
  const int per = n >> 3, rem = n & 7;
  const int l = xcd * per + (xcd < rem ? xcd : rem) + slot;  // XCD k owns per + (k < rem) consecutive list entries
  tile = l % ntiles;
  const int r = l / ntiles;
  hd = r % H;
  b = r / H;
}
__device__ __forceinline__ void attn_wg_coords(int ntiles, int H, int& tile, int& hd, int& b) {
  attn_wg_coords_of((int)gridDim.x, (int)blockIdx.x, ntiles, H, tile, hd, b);
}
// the same function on the host (no launch): tests/test_capi_symbols.py checks that it is a bijection for ragged grid sizes
extern "C" int aitk_probe_attn_wg_coords(int32_t n, int32_t id, int32_t ntiles, int32_t H, int32_t* out3) {
  if (n <= 0 || id < 0 || id >= n || ntiles <= 0 || H <= 0 || !out3) return AITK_ERR_ARG;
  int t, h, b;
  attn_wg_coords_of(n, id, ntiles, H, t, h, b);
  out3[0] = t;
  out3[1] = h;
  out3[2] = b;
  return AITK_OK;
}

// ============================================================================================ forward
// grid = ceil(S/128) * H * B workgroups (attn_wg_coords); 4 waves x 32 query rows; KV tiles of 64 rows, LDS-DMA double buffer (64 KiB -> 2
// workgroups per CU, <=256 registers -> 2 waves per SIMD so one workgroup's softmax overlaps the other's MFMAs).
// Online softmax with deferred rescale (guide T13, threshold 2^8): O/l are only rescaled when the running max grows by
// more than 8 in the log2 domain; P is then bounded by 2^8 instead of 1, exact in fp32 accumulation.
#define ATTN_DEFER_THR 8.0f
// KS = 16-wide contraction steps of Q K^T that carry data, DB = 32-wide output column blocks of P V that carry data: heads narrower than
// the 128-column layout (UNet head_dim 40 / 64 / 80, stored zero-padded — AitkAttnArgs.Dv) skip the all-zero steps and blocks.  The
// tiles keep their 128-column layouts; the skipped output columns are written as zeros, exactly what the padded computation yields.
template <int KS, int DB>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AitkAttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* const sm = (lds_char*)smem;  // buffer b: K tile (row-major, swizzled) at b*FBUF, V tile (sub-tiled) at +16384
  constexpr int FBUF = 16384 + SUBTILE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int S = p.S;                        // query rows per batch
  const int Skv = p.Skv > 0 ? p.Skv : p.S;  // key/value rows per batch (cross-attention: Skv != S)
  int tile_x, hd, b;
  attn_wg_coords((S + 127) / 128, p.H, tile_x, hd, b);
  const int HS = p.hstride > 0 ? p.hstride : 128;  // elements between heads: 128 (padded layout) or the native head width
  const int q0 = tile_x * 128 + wave * 32;
  const bf16_t* Qb = p.Q + (long)b * S * p.ldq + hd * HS;
  const bf16_t* Kb = p.K + (long)b * Skv * p.ldk + hd * HS;
  const bf16_t* Vb = p.V + (long)b * Skv * p.ldv + hd * HS;

  s16x8_t qf[KS];
  {
    const int qr = min(q0 + l31, S - 1);
    const bf16_t* qp = Qb + (long)qr * p.ldq + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const s16x8_t*>(qp + 16 * ks);
  }
  f32x16_t o[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d) o[d] = zero16();
  float m_run = -INFINITY, l_run = 0.f;  // m_run in the scaled log2 domain
  const float c2 = p.scale * 1.4426950408889634f;

  const int ntiles = (Skv + 63) / 64;
  const v4i_t srdK = slice_srd(Kb, p.ldk, Skv, HS), srdV = slice_srd(Vb, p.ldv, Skv, HS);
  unsigned vK[4], vV[4];
  rm_voff(vK, p.ldk, wave, lane);
  st_voff(vV, p.ldv, wave, lane);
  const unsigned stepK = (unsigned)(64 * p.ldk * 2), stepV = (unsigned)(64 * p.ldv * 2);  // bytes per KV tile
  dma_rm64(sm, vK, 0, srdK, wave);
  dma_st64(sm + 16384, vV, 0, srdV, wave);
  DMA_WAIT_ALL();
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    const lds_char* ktc = sm + cur * FBUF;
    const lds_char* vtc = ktc + 16384;
    if (t + 1 < ntiles) {
      dma_rm64(sm + (cur ^ 1) * FBUF, vK, (t + 1) * stepK, srdK, wave);
      dma_st64(sm + (cur ^ 1) * FBUF + 16384, vV, (t + 1) * stepV, srdV, wave);
    }
    // LDS fragment reads are issued in groups ahead of the MFMAs that consume them (the compiler otherwise emits
    // read -> lgkmcnt(0) -> mfma one by one and every MFMA eats a full LDS round trip)
    f32x16_t s[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      s[j] = zero16();
#pragma unroll
      for (int hh = 0; hh < (KS + 3) / 4; ++hh) {
        constexpr int NU = 4;
        s16x8_t kfr[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u)
          if (4 * hh + u < KS) kfr[u] = frag_rm_sw(ktc, 32 * j, 16 * (4 * hh + u), lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NU; ++u)
          if (4 * hh + u < KS) s[j] = mfma32(kfr[u], qf[4 * hh + u], s[j]);
      }
    }
    const int kv0 = t * 64;
    if (kv0 + 64 > Skv) {  // wave-uniform: only the last tile masks
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kv0 + 32 * j + crow(r, h) >= Skv) s[j][r] = -INFINITY;
    }
    float mt = -INFINITY;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[j][r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64)) * c2;
    if (!__all(mt - m_run <= ATTN_DEFER_THR)) {
      const float m_new = fmaxf(m_run, mt);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
      m_run = m_new;
    }
    float ps = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(fmaf(s[j][r], c2, -m_run));
        s[j][r] = e;
        ps += e;
      }
    l_run += ps;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const s16x8_t pf = pack_acc8(s[kk >> 1], 8 * (kk & 1));
#pragma unroll
      for (int d = 0; d < DB; ++d) o[d] = mfma32(frag_tr_perm_st(vtc, 16 * kk, 32 * d, lane), pf, o[d]);
    }
    DMA_WAIT_ALL();  // the next tile has landed (this wave's pieces) before the barrier publishes it
    __syncthreads();
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int q = q0 + l31;
  if (q < S) {
    bf16_t* op = p.O + ((long)b * S + q) * p.ldo + hd * HS;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (d >= DB && 32 * d >= HS) continue;  // native head layout: the columns past this head belong to the next one
        uint2 u = make_uint2(0u, 0u);
        if (d < DB) {
          u.x = pack2bf(o[d < DB ? d : 0][4 * g + 0] * inv, o[d < DB ? d : 0][4 * g + 1] * inv);
          u.y = pack2bf(o[d < DB ? d : 0][4 * g + 2] * inv, o[d < DB ? d : 0][4 * g + 3] * inv);
        }
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
  const int HS = p.hstride > 0 ? p.hstride : 128;
  uint4 a = make_uint4(0, 0, 0, 0), g = a;
  if (sub * 8 < HS) {  // native head layout: the head is HS columns wide (the padded layout carries zeros up to 128)
    a = *reinterpret_cast<const uint4*>(p.O + tok * p.ldo + hd * HS + sub * 8);
    g = *reinterpret_cast<const uint4*>(p.dO + tok * p.lddo + hd * HS + sub * 8);
  }
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
// grid = ceil(Skv/128) * H * B workgroups (attn_wg_coords); wave w owns kv rows [kv0 + 32 w, +32) (K, V fragments in registers, dK/dV accumulators);
// loops over query tiles of 32 rows staged (Q, dO, L2, delta) in LDS.
// S[q][kv] = Q K^T (lane owns one kv column), P = exp2(S c2 - L2[q]), dP = dO V^T, dS = P (dP - delta[q]);
// dV += P^T dO, dK += scale * dS^T Q  (Q/dO consumed via tr16 with the permuted order of the packed P / dS registers).
template <int KS, int DB>
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(AitkAttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* const sm = (lds_char*)smem;  // buffer b: Q tile (sub-tiled) at b*2*ST, dO tile at +ST; stats at 4*ST + b*512
  constexpr int ST = SUBTILE_BYTES;
  typedef __attribute__((address_space(3))) float lds_float;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int S = p.S;
  const int Skv = p.Skv > 0 ? p.Skv : p.S;
  int tile_x, hd, b;
  attn_wg_coords((Skv + 127) / 128, p.H, tile_x, hd, b);
  const int HS = p.hstride > 0 ? p.hstride : 128;  // elements between heads: 128 (padded layout) or the native head width
  const int kvw = tile_x * 128 + wave * 32;
  const bf16_t* Qb = p.Q + (long)b * S * p.ldq + hd * HS;
  const bf16_t* Kb = p.K + (long)b * Skv * p.ldk + hd * HS;
  const bf16_t* Vb = p.V + (long)b * Skv * p.ldv + hd * HS;
  const bf16_t* dOb = p.dO + (long)b * S * p.lddo + hd * HS;
  const float* Lb = p.LSE + ((long)b * p.H + hd) * S;
  const float* Db = p.delta + ((long)b * p.H + hd) * S;

  s16x8_t kf[KS], vf[KS];
  {
    const int kr = min(kvw + l31, Skv - 1);
    const bf16_t* kp = Kb + (long)kr * p.ldk + 8 * h;
    const bf16_t* vp = Vb + (long)kr * p.ldv + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[ks] = *reinterpret_cast<const s16x8_t*>(kp + 16 * ks);
      vf[ks] = *reinterpret_cast<const s16x8_t*>(vp + 16 * ks);
    }
  }
  f32x16_t dk[DB], dv[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d) {
    dk[d] = zero16();
    dv[d] = zero16();
  }
  const float c2 = p.scale * 1.4426950408889634f;
  const int ntiles = (S + 63) / 64;  // query tiles of 64 rows, processed as two 32-row halves per barrier
  // rows >= S are clamped (finite data); their L2 = +inf makes P = 0 so they contribute nothing
  // The per-row statistics are ordinary global loads: they are issued BEFORE the tile DMAs and stored to LDS at the
  // END of the iteration, otherwise their vmcnt wait (in-order counter) would also drain the just-issued LDS-DMA
  // prefetch and make it synchronous.
  float stat = 0.f;
  const v4i_t srdQ = slice_srd(Qb, p.ldq, S, HS), srdD = slice_srd(dOb, p.lddo, S, HS);
  unsigned vQ[4], vD[4];
  st_voff(vQ, p.ldq, wave, lane);
  st_voff(vD, p.lddo, wave, lane);
  const unsigned stepQ = (unsigned)(64 * p.ldq * 2), stepD = (unsigned)(64 * p.lddo * 2);
  auto stage_load = [&](int t, int buf) {
    if (tid < 128) {
      const int q = t * 64 + (tid & 63);
      stat = tid < 64 ? (q < S ? Lb[q] : INFINITY) : (q < S ? Db[q] : 0.f);
    }
    dma_st64(sm + buf * 2 * ST, vQ, t * stepQ, srdQ, wave);
    dma_st64(sm + buf * 2 * ST + ST, vD, t * stepD, srdD, wave);
  };
  auto stage_store = [&](int buf) {
    if (tid < 128) ((lds_float*)(sm + 4 * ST + buf * 512))[tid] = stat;  // [0,64): L2, [64,128): delta
  };
  stage_load(0, 0);
  stage_store(0);
  DMA_WAIT_ALL();
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntiles) stage_load(t + 1, cur ^ 1);
    const lds_char* qtc = sm + cur * 2 * ST;
    const lds_char* dotc = qtc + ST;
    const lds_float* ltc = (const lds_float*)(sm + 4 * ST + cur * 512);
    const lds_float* dtc = ltc + 64;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      f32x16_t s, dp;
#pragma unroll
      for (int hk = 0; hk < (KS + 3) / 4; ++hk) {  // batches of 4 k-steps: 32 fragment registers instead of 64
        s16x8_t qa[4], da[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (4 * hk + ks >= KS) continue;
          qa[ks] = frag_rm_st(qtc, 32 * sub, 16 * (4 * hk + ks), lane);
          da[ks] = frag_rm_st(dotc, 32 * sub, 16 * (4 * hk + ks), lane);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (4 * hk + ks >= KS) continue;
          // inline asm pins the register classes: S / dP accumulate in arch VGPRs (the softmax VALU reads them), the
          // loop-invariant K / V fragments sit in AccVGPRs.  Left to the allocator (389 registers, 1 wave per SIMD) S / dP
          // land in AccVGPRs time-shared with dK: 128 v_accvgpr_read/write per iteration.
          if (hk == 0 && ks == 0) {  // first product of the chain: C = 0 (no zero-fill of the 32 accumulator registers)
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(s) : "v"(qa[ks]), "a"(kf[0]));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(dp) : "v"(da[ks]), "a"(vf[0]));
          } else {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(qa[ks]), "a"(kf[4 * hk + ks]));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(dp) : "v"(da[ks]), "a"(vf[4 * hk + ks]));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // MFMA results -> VALU reads: the hazard recogniser does not see inside the asm (18 wait states for a 16-pass result)
      asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(s), "+v"(dp));
      // transposed operands of the dV / dK products: issued now so their LDS latency hides behind the softmax VALU
      s16x4_t dlo[8], dhi[8], qlo[8], qhi[8];  // index = 4*kk + d
      {
        // lane (h, gq, i) reads rows 32 sub + 16 kk + 4h + (i>>2) (+8) of column block 2d + gq of the sub-tiled Q / dO tiles:
        // (row>>3)&1 is 0 for the first and 1 for the second read, so two per-lane bases + immediates address all 32 reads
        const int hq = lane >> 5, gq = (lane >> 4) & 1, i = lane & 15, lh = (i & 3) >> 1;
        const unsigned tlo = (unsigned)(size_t)qtc + gq * SUBP + (i & 1) * 8 + (4 * hq + (i >> 2)) * 32 + (lh << 4);
        const unsigned thi = tlo + 8 * 32 + ((lh ^ 1) - lh) * 16;
#define TRQ(KK, D)                                                                                     \
  if constexpr (D < DB) {                                                                              \
    tr16_issue_off<ST + (32 * SUBI + 16 * KK) * 32 + 2 * D * SUBP>(dlo[4 * KK + D], tlo);               \
    tr16_issue_off<ST + (32 * SUBI + 16 * KK) * 32 + 2 * D * SUBP>(dhi[4 * KK + D], thi);               \
    tr16_issue_off<(32 * SUBI + 16 * KK) * 32 + 2 * D * SUBP>(qlo[4 * KK + D], tlo);                    \
    tr16_issue_off<(32 * SUBI + 16 * KK) * 32 + 2 * D * SUBP>(qhi[4 * KK + D], thi);                    \
  }
#define TRQ8() TRQ(0, 0) TRQ(0, 1) TRQ(0, 2) TRQ(0, 3) TRQ(1, 0) TRQ(1, 1) TRQ(1, 2) TRQ(1, 3)
        if (sub == 0) {
#define SUBI 0
          TRQ8()
#undef SUBI
        } else {
#define SUBI 1
          TRQ8()
#undef SUBI
        }
#undef TRQ8
#undef TRQ
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4_t l4 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4_t*>(ltc + 32 * sub + 8 * g + 4 * h);
        const f32x4_t d4 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4_t*>(dtc + 32 * sub + 8 * g + 4 * h);
        const float ls[4] = {l4[0], l4[1], l4[2], l4[3]};
        const float ds[4] = {d4[0], d4[1], d4[2], d4[3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const float pr = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -ls[e]));
          s[r] = pr;
          dp[r] = pr * (dp[r] - ds[e]);
        }
      }
      // the transpose reads must have landed: wait names every destination so nothing that uses them is scheduled above
      asm volatile("s_waitcnt lgkmcnt(0)" : TR_PIN8(dlo), TR_PIN8(dhi));
      asm volatile("" : TR_PIN8(qlo), TR_PIN8(qhi));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const s16x8_t pf = pack_acc8(s, 8 * kk);
        const s16x8_t df = pack_acc8(dp, 8 * kk);
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          dv[d] = mfma32(pf, join_lohi(dlo[4 * kk + d], dhi[4 * kk + d]), dv[d]);
          dk[d] = mfma32(df, join_lohi(qlo[4 * kk + d], qhi[4 * kk + d]), dk[d]);
        }
      }
    }
    if (t + 1 < ntiles) stage_store(cur ^ 1);
    DMA_WAIT_ALL();
    __syncthreads();
  }
  // D[i = kv][j = d]: lane holds column d = 32*blk + l31 and rows kv = crow(r, h)
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kv = kvw + crow(r, h);
      if (kv < Skv && (d < DB || 32 * d < HS)) {
        const long off = ((long)b * Skv + kv);
        p.dK[off * p.lddk + hd * HS + 32 * d + l31] = d < DB ? f2bf(dk[d < DB ? d : 0][r] * p.scale) : (bf16_t)0;
        p.dV[off * p.lddv + hd * HS + 32 * d + l31] = d < DB ? f2bf(dv[d < DB ? d : 0][r]) : (bf16_t)0;
      }
    }
}

// ============================================================================================ backward: dK, dV — software-pipelined (head_dim 128)
// Same ownership as attn_bwd_dkdv_kernel (wave w owns kv rows [kv0 + 32 w, +32): K / V fragments and the dK / dV accumulators stay in
// registers).  That kernel runs ONE wave per SIMD (its register budget) and a wave walks its phases one after the other — fragment reads,
// S / dP MFMAs, softmax VALU, transpose reads, dV / dK MFMAs — so matrix time and everything-else time add up instead of overlapping
// (profiles/r02_notes_attention_ablation.md: 1570 us of non-matrix work + 1082 us of MFMA = the 2893 us launch).  Here the query tile is
// 128 rows = four 32-row sub-tiles and the wave pipelines them itself:
//     A(0) [C(3 of the previous tile) || B(0)] | A(1) [C(0) || B(1)] | A(2) [C(1) || B(2)] | A(3) [C(2) || B(3)] | barrier
// A(i) = S, dP products of sub-tile i, B(i) = its softmax / dS arithmetic (VALU), C(i) = its dV, dK products.  C(i-1) and B(i) are
// independent, so their instruction streams are interleaved one MFMA : one accumulator register's worth of VALU — the matrix pipe runs
// C(i-1) while the vector pipe runs B(i) (guide T15, done inside one wave).  The transpose reads of C(i-1) are issued inside A(i)'s MFMA
// block and land behind it.  P / dS live as packed bf16 operands in two alternating register sets.
#define DKDV_P_LDS (8 * SUBTILE_BYTES + 2 * 1024)
__global__ __launch_bounds__(256) void attn_bwd_dkdv_pipe_kernel(AitkAttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* const sm = (lds_char*)smem;  // buffer b at b*TB: Q sub-tiles (rows 0-63 | 64-127), then dO sub-tiles; stats at 2*TB + b*1024
  constexpr int KS = 8, DB = 4;
  constexpr int ST = SUBTILE_BYTES, TB = 4 * SUBTILE_BYTES;
  typedef __attribute__((address_space(3))) float lds_float;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int S = p.S;
  const int Skv = p.Skv > 0 ? p.Skv : p.S;
  int tile_x, hd, b;
  attn_wg_coords((Skv + 127) / 128, p.H, tile_x, hd, b);
  const int kvw = tile_x * 128 + wave * 32;
  const bf16_t* Qb = p.Q + (long)b * S * p.ldq + hd * 128;
  const bf16_t* Kb = p.K + (long)b * Skv * p.ldk + hd * 128;
  const bf16_t* Vb = p.V + (long)b * Skv * p.ldv + hd * 128;
  const bf16_t* dOb = p.dO + (long)b * S * p.lddo + hd * 128;
  const float* Lb = p.LSE + ((long)b * p.H + hd) * S;
  const float* Db = p.delta + ((long)b * p.H + hd) * S;

  s16x8_t kf[KS], vf[KS];
  {
    const int kr = min(kvw + l31, Skv - 1);
    const bf16_t* kp = Kb + (long)kr * p.ldk + 8 * h;
    const bf16_t* vp = Vb + (long)kr * p.ldv + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[ks] = *reinterpret_cast<const s16x8_t*>(kp + 16 * ks);
      vf[ks] = *reinterpret_cast<const s16x8_t*>(vp + 16 * ks);
    }
  }
  f32x16_t dk[DB], dv[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d) {
    dk[d] = zero16();
    dv[d] = zero16();
  }
  const float c2 = p.scale * 1.4426950408889634f;
  const int ntiles = (S + 127) / 128;
  float stat = 0.f;
  const v4i_t srdQ = slice_srd(Qb, p.ldq, S), srdD = slice_srd(dOb, p.lddo, S);
  unsigned vQ[4], vD[4];
  st_voff(vQ, p.ldq, wave, lane);
  st_voff(vD, p.lddo, wave, lane);
  const unsigned stepQ = (unsigned)(64 * p.ldq * 2), stepD = (unsigned)(64 * p.lddo * 2);  // bytes per 64-row quarter
  auto stage_load = [&](int t, int buf) {
    {  // statistics: plain global loads issued BEFORE the tile DMAs, stored to LDS at the end of the iteration (see attn_bwd_dkdv_kernel)
      const int q = t * 128 + (tid & 127);
      stat = tid < 128 ? (q < S ? Lb[q] : INFINITY) : (q < S ? Db[q] : 0.f);
    }
  };
  // one quarter (64 rows of Q or dO) of tile t's LDS-DMA: the four quarters of the NEXT tile are issued inside the MFMA streams of the
  // current one (an LDS-DMA piece costs 100-185 cycles among fragment reads, 25-60 in a gap of the matrix stream: backward 4.02 -> 3.96 ms)
  auto stage_quarter = [&](int t, int buf, int q) {
    lds_char* b0 = sm + buf * TB + q * ST;
    if (q < 2) dma_st64(b0, vQ, (2 * t + q) * stepQ, srdQ, wave);
    else dma_st64(b0, vD, (2 * t + q - 2) * stepD, srdD, wave);
  };
  auto stage_store = [&](int buf) { ((lds_float*)(sm + 2 * TB + buf * 1024))[tid] = stat; };  // [0,128): L2, [128,256): delta
  stage_load(0, 0);
  for (int q = 0; q < 4; ++q) stage_quarter(0, 0, q);
  stage_store(0);
  DMA_WAIT_ALL();
  __syncthreads();
  const int hq = lane >> 5, gq = (lane >> 4) & 1, i16 = lane & 15, lh = (i16 & 3) >> 1;
  f32x16_t s, dp;
  s16x8_t pf[2][2], df[2][2];                      // [sub-tile parity][kk]: packed P / dS operands
  s16x4_t dlo[8], dhi[8], qlo[8], qhi[8];          // transposed dO / Q fragments of ONE sub-tile, index 4*kk + d
  // the pipeline runs across tiles (C(3) of tile t - 1 rides under B(0) of tile t): in front of the first tile the "previous" operands are zeros
#pragma unroll
  for (int e = 0; e < 8; ++e) dlo[e] = dhi[e] = qlo[e] = qhi[e] = s16x4_t{0, 0, 0, 0};
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) pf[1][kk] = df[1][kk] = s16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    const bool more = t + 1 < ntiles;
    if (more) stage_load(t + 1, cur ^ 1);  // the statistics of the next tile (global loads, ahead of its LDS-DMA quarters)
    const lds_char* qt = sm + cur * TB;
    const lds_float* ltc = (const lds_float*)(sm + 2 * TB + cur * 1024);
    const lds_float* dtc = ltc + 128;
    // lane parts of the transpose-read addresses (rows with bit 3 clear / the rows 8 further), Q and dO tiles
    const unsigned tlo_q = (unsigned)(size_t)qt + gq * SUBP + (i16 & 1) * 8 + (4 * hq + (i16 >> 2)) * 32 + (lh << 4);
    const unsigned thi_q = tlo_q + 8 * 32 + ((lh ^ 1) - lh) * 16;
    const unsigned tlo_d = tlo_q + 2 * ST, thi_d = thi_q + 2 * ST;

#define DKDV_TR1(SUB, KK, D)                                                                                          \
  tr16_issue_off<((SUB) >> 1) * ST + (32 * ((SUB) & 1) + 16 * (KK)) * 32 + 2 * (D) * SUBP>(dlo[4 * (KK) + (D)], tlo_d); \
  tr16_issue_off<((SUB) >> 1) * ST + (32 * ((SUB) & 1) + 16 * (KK)) * 32 + 2 * (D) * SUBP>(dhi[4 * (KK) + (D)], thi_d); \
  tr16_issue_off<((SUB) >> 1) * ST + (32 * ((SUB) & 1) + 16 * (KK)) * 32 + 2 * (D) * SUBP>(qlo[4 * (KK) + (D)], tlo_q); \
  tr16_issue_off<((SUB) >> 1) * ST + (32 * ((SUB) & 1) + 16 * (KK)) * 32 + 2 * (D) * SUBP>(qhi[4 * (KK) + (D)], thi_q);
#define DKDV_TR(SUB)                                                                                                     \
  DKDV_TR1(SUB, 0, 0) DKDV_TR1(SUB, 0, 1) DKDV_TR1(SUB, 0, 2) DKDV_TR1(SUB, 0, 3) DKDV_TR1(SUB, 1, 0) DKDV_TR1(SUB, 1, 1) \
  DKDV_TR1(SUB, 1, 2) DKDV_TR1(SUB, 1, 3)
#define DKDV_TR_WAIT()                                                   \
  asm volatile("s_waitcnt lgkmcnt(0)" : TR_PIN8(dlo), TR_PIN8(dhi));    \
  asm volatile("" : TR_PIN8(qlo), TR_PIN8(qhi));                        \
  __builtin_amdgcn_sched_barrier(0);
    // A(SUB): S = Q K^T, dP = dO V^T of sub-tile SUB (asm MFMAs: S / dP in arch VGPRs, K / V fragments in AccVGPRs).  TRSUB >= 0: the
    // transpose reads of sub-tile TRSUB are issued after the first MFMA of the second k-batch, so they land behind the remaining seven pairs.
#define DKDV_A(SUB, TR_STMT)                                                                                             \
  {                                                                                                                      \
    const lds_char* qsub = qt + ((SUB) >> 1) * ST;                                                                       \
    const lds_char* dosub = qsub + 2 * ST;                                                                               \
    _Pragma("unroll") for (int hk = 0; hk < 2; ++hk) {                                                                   \
      s16x8_t qa[4], da[4];                                                                                              \
      _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                                 \
        qa[ks] = frag_rm_st(qsub, 32 * ((SUB) & 1), 16 * (4 * hk + ks), lane);                                           \
        da[ks] = frag_rm_st(dosub, 32 * ((SUB) & 1), 16 * (4 * hk + ks), lane);                                          \
      }                                                                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                                 \
      _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                                 \
        if (hk == 0 && ks == 0) {                                                                                        \
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(s) : "v"(qa[ks]), "a"(kf[0]));                    \
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(dp) : "v"(da[ks]), "a"(vf[0]));                   \
        } else {                                                                                                         \
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(qa[ks]), "a"(kf[4 * hk + ks]));         \
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(dp) : "v"(da[ks]), "a"(vf[4 * hk + ks]));        \
        }                                                                                                                \
        if (hk == 1 && ks == 0) { TR_STMT }                                                                              \
      }                                                                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                                 \
    }                                                                                                                    \
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(s), "+v"(dp));                                                   \
  }
    // C(prev) || B(SUB): sixteen steps of [one dV / dK MFMA of the previous sub-tile (operand set PAR ^ 1, transposed fragments in
    // dlo .. qhi)] + [the softmax / dS arithmetic of accumulator register r of sub-tile SUB]; the result is packed into operand set PAR
    // keep the dS arithmetic scalar: the SLP vectoriser otherwise pairs registers r, r + 1 into v_pk_add_f32 / v_pk_mul_f32, which cost ~26
    // cycles per MFMA gap beside the matrix stream (MI355X_MICROARCH.md; measured here: backward 4.16 -> 4.02 ms at B = 4, H = 24, S = 4608)
#define DKDV_OPAQUE(x) asm volatile("" : "+v"(x))
#define DKDV_CB(SUB, PAR, STEP_STMT)                                                                                     \
  {                                                                                                                      \
    float ls[16], ds[16];                                                                                                \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                      \
      const f32x4_t l4 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4_t*>(ltc + 32 * (SUB) + 8 * g + 4 * h); \
      const f32x4_t d4 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4_t*>(dtc + 32 * (SUB) + 8 * g + 4 * h); \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                                    \
        ls[4 * g + e] = l4[e];                                                                                           \
        ds[4 * g + e] = d4[e];                                                                                           \
      }                                                                                                                  \
    }                                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                                     \
      const int kk = r >> 3, d = (r >> 1) & 3;                                                                           \
      if ((r & 1) == 0)                                                                                                  \
        dv[d] = mfma32(pf[(PAR) ^ 1][kk], join_lohi(dlo[4 * kk + d], dhi[4 * kk + d]), dv[d]);                           \
      else                                                                                                               \
        dk[d] = mfma32(df[(PAR) ^ 1][kk], join_lohi(qlo[4 * kk + d], qhi[4 * kk + d]), dk[d]);                           \
      const float pr = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -ls[r]));                                                                 \
      float dd = dp[r] - ds[r];                                                                                          \
      DKDV_OPAQUE(dd);                                                                                                   \
      dd *= pr;                                                                                                          \
      DKDV_OPAQUE(dd);                                                                                                   \
      s[r] = pr;                                                                                                         \
      dp[r] = dd;                                                                                                        \
      STEP_STMT                                                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                                                 \
    }                                                                                                                    \
    pf[PAR][0] = pack_acc8(s, 0);                                                                                        \
    pf[PAR][1] = pack_acc8(s, 8);                                                                                        \
    df[PAR][0] = pack_acc8(dp, 0);                                                                                       \
    df[PAR][1] = pack_acc8(dp, 8);                                                                                       \
  }
#define DKDV_DMA(Q) if (r == 9 && more) stage_quarter(t + 1, cur ^ 1, Q);
    // across tiles: the last sub-tile's dV / dK products of tile t - 1 (operand set 1, transposed fragments read before that tile's
    // closing barrier) ride under the first softmax of tile t instead of running bare at the end of their own tile
    DKDV_A(0, )
    DKDV_CB(0, 0, )
    DKDV_A(1, DKDV_TR(0))
    DKDV_TR_WAIT()
    DKDV_CB(1, 1, DKDV_DMA(0))
    DKDV_A(2, DKDV_TR(1))
    DKDV_TR_WAIT()
    DKDV_CB(2, 0, DKDV_DMA(1))
    DKDV_A(3, DKDV_TR(2))
    DKDV_TR_WAIT()
#define DKDV_DMA23 if ((r == 4 || r == 12) && more) stage_quarter(t + 1, cur ^ 1, r == 4 ? 2 : 3);
    DKDV_CB(3, 1, DKDV_DMA23)
    DKDV_TR(3)
    DKDV_TR_WAIT()   // in registers before the barrier: the next tile's LDS-DMA may overwrite this buffer one tile later
#undef DKDV_DMA23
#undef DKDV_CB
#undef DKDV_A
#undef DKDV_TR_WAIT
#undef DKDV_TR
#undef DKDV_TR1
    if (t + 1 < ntiles) stage_store(cur ^ 1);
    DMA_WAIT_ALL();
    __syncthreads();
  }
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)  // the last tile's last sub-tile
#pragma unroll
    for (int d = 0; d < DB; ++d) {
      dv[d] = mfma32(pf[1][kk], join_lohi(dlo[4 * kk + d], dhi[4 * kk + d]), dv[d]);
      dk[d] = mfma32(df[1][kk], join_lohi(qlo[4 * kk + d], qhi[4 * kk + d]), dk[d]);
    }
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kv = kvw + crow(r, h);
      if (kv < Skv) {
        const long off = ((long)b * Skv + kv);
        p.dK[off * p.lddk + hd * 128 + 32 * d + l31] = f2bf(dk[d][r] * p.scale);
        p.dV[off * p.lddv + hd * 128 + 32 * d + l31] = f2bf(dv[d][r]);
      }
    }
}

// ============================================================================================ backward: dK, dV — wave-specialised (head_dim 128)
// The pipelined kernel above runs ONE wave per SIMD (394 registers): whatever a wave waits for — fragment reads, transpose reads, the
// softmax's quarter-rate exponentials — idles its SIMD's matrix pipe (PMC round 3: MFMA busy 0.37).  Here a workgroup is 8 waves, two per
// SIMD, with different jobs on the same 32 kv rows:
//   producer wave w (0-3):  S = Q K^T, dP = dO V^T of sub-tile i+1 (16 MFMAs) interleaved with the softmax / dS arithmetic of sub-tile i
//                           (K / V fragments in AccVGPRs, two S / dP register sets); P and dS leave as the packed bf16 operands the
//                           dV / dK products consume, through a lane-private 64-byte LDS slot — the accumulator layout of S IS the
//                           A-operand layout of P^T (permuted contraction order, see frag_tr_perm_st), so nothing is transposed;
//   consumer wave w + 4:    dV += P^T dO, dK += dS^T Q of sub-tile i-1 (16 MFMAs, transposed dO / Q fragments through tr16 reads),
//                           owns the dK / dV accumulators (128 AccVGPRs) and the statistics loads.
// Both fit 256 registers, so each SIMD always has a second wave to issue matrix work while the other waits or runs vector code — the
// cross-wave MFMA || VALU overlap that profiles/r03_probe_mfma_valu_overlap.txt shows is free.  One workgroup barrier per 32-row sub-tile
// hands the operand slots over (double-buffered); query tiles of 64 rows travel by LDS-DMA into a ring of three buffers (a tile is read
// by the producer one step before and by the consumer one step after its own two steps), issued two steps ahead of their first use.
// Same ownership, same products in the same order as attn_bwd_dkdv_pipe_kernel: the gradients are bit-identical to it.
#define DKDV_WS_TILE (2 * SUBTILE_BYTES)              /* Q (64 rows) then dO (64 rows), sub-tiled */
#define DKDV_WS_NSLOT 4                               /* tile ring */
#define DKDV_WS_XOFF (DKDV_WS_NSLOT * DKDV_WS_TILE)   /* operand slots: [pair 4][vector 4][lane 64] x 16 B (written in H2, read in the next H1) */
#define DKDV_WS_SOFF (DKDV_WS_XOFF + 4 * 4096)        /* statistics: [tile ring][L2 64 | delta 64] floats */
#define DKDV_WS_LDS (DKDV_WS_SOFF + DKDV_WS_NSLOT * 512)
// TRACE: workgroup 0 records s_memtime around every barrier of tiles 8-11 for one producer and one consumer wave (aitk_probe_attn_ws_trace)
__device__ unsigned long long g_ws_trace[2 * 4 * 8];
// DS (AitkAttnArgs.ds_mode): 0 = off; 1 = the producer waves also write the packed dS they hand to the consumers to global memory, their two
// operand vectors as they are: 2 KiB per wave and sub-tile in two coalesced non-temporal 16-byte stores per lane — the probe of the 5-matmul
// backward (dQ = dS K as its own product).  (A row-major [q][kv] form — sixteen 2-byte stores per lane, what a plain GEMM would read — was
// tried and dropped: it cannot be addressed without spilling 56-116 registers into the hand-scheduled loop.)
template <bool TRACE, int DS>
__global__ __launch_bounds__(512) void attn_bwd_dkdv_ws_kernel(AitkAttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* const sm = (lds_char*)smem;
  constexpr int ST = SUBTILE_BYTES, TILE = DKDV_WS_TILE, NSLOT = DKDV_WS_NSLOT;
  typedef __attribute__((address_space(3))) float lds_float;
  typedef __attribute__((address_space(3))) s16x8_t lds_s16x8;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave8 >> 2, w = wave8 & 3;  // role 0: producer, 1: consumer; pair w owns kv rows [kv0 + 32 w, +32)
  const int l31 = lane & 31, h = lane >> 5;
  const int S = p.S;
  const int Skv = p.Skv > 0 ? p.Skv : p.S;
  int tile_x, hd, b;
  attn_wg_coords((Skv + 127) / 128, p.H, tile_x, hd, b);
  const int kvw = tile_x * 128 + w * 32;
  const bf16_t* Qb = p.Q + (long)b * S * p.ldq + hd * 128;
  const bf16_t* Kb = p.K + (long)b * Skv * p.ldk + hd * 128;
  const bf16_t* Vb = p.V + (long)b * Skv * p.ldv + hd * 128;
  const bf16_t* dOb = p.dO + (long)b * S * p.lddo + hd * 128;
  const float* Lb = p.LSE + ((long)b * p.H + hd) * S;
  const float* Db = p.delta + ((long)b * p.H + hd) * S;
  const float c2 = p.scale * 1.4426950408889634f;
  const int ntile = (S + 63) / 64;  // query tiles of 64 rows = two 32-row sub-tiles (rows >= S: zero Q / dO, L2 = +inf -> P = dS = 0)

  // ---- tile staging.  Producers stage the Q half of a tile, consumers the dO half: four 1-KiB LDS-DMA pieces per wave and tile, issued two at a
  // time INSIDE the matrix streams (a piece costs ~100 cycles of issue among LDS reads, ~30 in a gap of a running MFMA block) over the two
  // steps after the ring slot became free, and retired by COUNTED waits: a wave always has exactly four younger pieces in flight when it needs
  // a tile, so `vmcnt(4)` is the landing condition and nothing ever drains the queue.  Tiles past the end are issued all the same (rows >= S
  // arrive as zeros through the buffer window) so that the counts stay exact.
  const v4i_t srd = role == 0 ? slice_srd(Qb, p.ldq, S) : slice_srd(dOb, p.lddo, S);
  unsigned vo[4];
  st_voff(vo, role == 0 ? p.ldq : p.lddo, w, lane);
  const unsigned step_bytes = (unsigned)(64 * (role == 0 ? p.ldq : p.lddo) * 2);
  auto issue_pieces = [&](int t, int ii0) {  // pieces ii0, ii0 + 1 of this wave's four for tile t
    lds_char* tb = sm + (t % NSLOT) * TILE + role * ST;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      const int ins = w + 4 * (ii0 + ii);
      dma16(__builtin_amdgcn_readfirstlane((unsigned)(size_t)tb + (ins >> 1) * SUBP + (ins & 1) * 1024), vo[ii0 + ii] + (unsigned)t * step_bytes, srd);
    }
  };
  // statistics of a tile (L2 and delta of its 64 query rows) also travel by LDS-DMA, one dword per lane: consumer wave w fetches L2 (w even) or
  // delta (w odd) — waves 2 / 3 repeat what 0 / 1 fetch so that every consumer wave has the same five pieces per tile in flight.  Rows >= S
  // arrive as zeros, which is as good as L2 = +inf there: their Q / dO rows are zeros too, so P^T dO = 0 and dS = P (0 - 0) = 0.
  const v4i_t srd_st = slice_srd_bytes((w & 1) ? (const void*)Db : (const void*)Lb, (long)S * 4);
  auto issue_stats = [&](int t) {
    const unsigned dst = (unsigned)(size_t)(sm + DKDV_WS_SOFF + (t % NSLOT) * 512 + (w & 1) * 256);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(__builtin_amdgcn_readfirstlane(dst)),
                 "v"((unsigned)((t * 64 + lane) * 4)), "s"(srd_st)
                 : "memory");
  };
  // prologue: tiles 0 and 1 complete, then the in-flight state the loop's counted waits assume: tile 2 — producers pieces 0-1 (2-3 follow in
  // H1 of step 0), consumers statistics + all four pieces
  for (int t = 0; t < 2; ++t) {
    if (role == 1) issue_stats(t);
    issue_pieces(t, 0);
    issue_pieces(t, 2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (role == 1) issue_stats(2);
  issue_pieces(2, 0);
  if (role == 1) issue_pieces(2, 2);
  __syncthreads();
  int ws_bar = 0;  // barrier index inside the current tile (trace only)
#define DKDV_WS_BAR(T)                                                                                                     \
  {                                                                                                                        \
    if (TRACE && blockIdx.x == 0 && w == 0 && lane == 0 && (T) >= 8 && (T) < 12) {                                         \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                   \
      g_ws_trace[(role * 4 + ((T) - 8)) * 8 + 2 * ws_bar] = __builtin_readcyclecounter();                                  \
    }                                                                                                                      \
    __syncthreads();                                                                                                       \
    if (TRACE && blockIdx.x == 0 && w == 0 && lane == 0 && (T) >= 8 && (T) < 12)                                           \
      g_ws_trace[(role * 4 + ((T) - 8)) * 8 + 2 * ws_bar + 1] = __builtin_readcyclecounter();                              \
    ws_bar = (ws_bar + 1) & 3;                                                                                             \
  }

  // Every 32-row step is two half-steps separated by workgroup barriers, and the two waves of a SIMD are in OPPOSITE phases (the ping-pong
  // of the 8-phase GEMM): in H1 the producer streams its 16 S / dP products of sub-tile i + 1 through the matrix pipe while the consumer
  // fetches its operands (the pair's slot + 32 transpose reads of sub-tile i - 1); in H2 the consumer streams its 16 dV / dK products while
  // the producer runs the softmax / dS arithmetic of sub-tile i on the vector ALU, hands P / dS over and fetches the Q / dO row fragments
  // of sub-tile i + 2 (into AccVGPRs: ds_read straight into the register class the MFMA reads them from).  A wave that meets a busy
  // matrix pipe stalls in order — interleaving the two waves' MFMAs instruction by instruction (version 1 of this kernel) serialised them.
  if (role == 0) {
    // ================================================================ producer
    s16x8_t kf[8], vf[8];
    {
      const int kr = min(kvw + l31, Skv - 1);
      const bf16_t* kp = Kb + (long)kr * p.ldk + 8 * h;
      const bf16_t* vp = Vb + (long)kr * p.ldv + 8 * h;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        kf[ks] = *reinterpret_cast<const s16x8_t*>(kp + 16 * ks);
        vf[ks] = *reinterpret_cast<const s16x8_t*>(vp + 16 * ks);
      }
    }
    f32x16_t sA[2], dpA[2];  // S / dP of sub-tile parity 0 / 1
    s16x8_t qa[8], da[8];    // row fragments of ONE sub-tile (AccVGPRs), fetched half a step ahead of their products
    f32x4_t lst[4], dst[4];  // statistics of the sub-tile whose arithmetic runs in the coming H2 (read at the end of H1)
    lds_char* const xw = sm + DKDV_WS_XOFF + w * 4096 + lane * 16;
    // DS: this lane's first destination element: block (kv block of 32 = 4 tile_x + w, q block SUB) of 1024 elements, lane-linear
    bf16_t* ds_base = nullptr;
    if (DS) {
      const long nq32 = (S + 31) / 32, nkv32 = (Skv + 31) / 32;
      ds_base = p.dS + (((long)b * p.H + hd) * nkv32 + (tile_x * 4 + w)) * nq32 * 1024 + lane * 8;
    }
    // frag_rm_st's address = tile + ks * SUBP + row * 32 + ((h ^ ((row >> 3) & 1)) << 4): the lane part is the same for rows l31 and 32 + l31
    const unsigned ln_rm = (unsigned)(l31 * 32 + ((h ^ ((l31 >> 3) & 1)) << 4));
#define DKDV_WS_FRAG1(RH, KS)                                                                                                       \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(qa[KS]) : "v"(fbase), "n"((RH) * 32 * 32 + (KS) * SUBP));                     \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(da[KS]) : "v"(fbase), "n"(ST + (RH) * 32 * 32 + (KS) * SUBP));
#define DKDV_WS_FRAGS(RH, TB)                                                                                                       \
  {                                                                                                                                 \
    const unsigned fbase = (unsigned)(size_t)(TB) + ln_rm;                                                                          \
    DKDV_WS_FRAG1(RH, 0) DKDV_WS_FRAG1(RH, 1) DKDV_WS_FRAG1(RH, 2) DKDV_WS_FRAG1(RH, 3)                                             \
    DKDV_WS_FRAG1(RH, 4) DKDV_WS_FRAG1(RH, 5) DKDV_WS_FRAG1(RH, 6) DKDV_WS_FRAG1(RH, 7)                                             \
  }
#define DKDV_WS_PIN8A(x) "+a"(x[0]), "+a"(x[1]), "+a"(x[2]), "+a"(x[3]), "+a"(x[4]), "+a"(x[5]), "+a"(x[6]), "+a"(x[7])
    // H1: S = Q K^T, dP = dO V^T of the sub-tile whose fragments are in qa / da, into register set SET; two LDS-DMA pieces (DMA_STMT) ride in
    // the matrix stream.  VALU = 1: the first two thirds of the CURRENT sub-tile's softmax / dS arithmetic (register set SET ^ 1, statistics of
    // tile TS row half RH: t = s c2 - L2, dd = dP - delta, p = exp2(t)) ride in the same stream, three or two registers per MFMA pair from the
    // third pair on — the matrix pipe belongs to this wave alone in H1 (the consumer is fetching operands), an MFMA runs 32 cycles and these
    // are 16 cycles of vector work, and by the third pair the statistics have returned.  What is left for H2 (p * dd, packing, the hand-over)
    // is short enough not to outlast the consumer's sixteen products: the vector ALU runs at ~0.6 of its rate beside the sibling's matrix
    // stream (profiles/r04_attn_ws_trace_*.txt: version 3b of this kernel spent 1000 cycles per H2 on 384 cycles of vector instructions).
    // Single fp32 instructions as asm statements: the scheduler may not pair them into v_pk_* forms (slower beside a matrix stream).
#define DKDV_WS_VALU3(CUR, R)                                                                                                       \
  {                                                                                                                                 \
    asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(sA[CUR][R]) : "v"(sA[CUR][R]), "v"(c2), "v"(lst[(R) >> 2][(R) & 3]));           \
    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dpA[CUR][R]) : "v"(dpA[CUR][R]), "v"(dst[(R) >> 2][(R) & 3]));                       \
    asm volatile("v_exp_f32 %0, %1" : "=v"(sA[CUR][R]) : "v"(sA[CUR][R]));                                                          \
  }
#define DKDV_WS_MMA(SET, VALU, DMA_STMT, TS, RH)                                                                                    \
  {                                                                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" : DKDV_WS_PIN8A(qa), DKDV_WS_PIN8A(da));                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                              \
    if (VALU) { /* the statistics of the current sub-tile: issued now, they return under the first two MFMA pairs */                 \
      const lds_float* ltc = (const lds_float*)(sm + DKDV_WS_SOFF + ((TS) % NSLOT) * 512) + 32 * (RH);                              \
      _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                               \
        lst[g] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4_t*>(ltc + 8 * g + 4 * h);                          \
        dst[g] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4_t*>(ltc + 64 + 8 * g + 4 * h);                     \
      }                                                                                                                             \
    }                                                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                                              \
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(sA[SET]) : "a"(qa[0]), "a"(kf[0]));                               \
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(dpA[SET]) : "a"(da[0]), "a"(vf[0]));                              \
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(sA[SET]) : "a"(qa[1]), "a"(kf[1]));                               \
    { DMA_STMT }                                                                                                                    \
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(dpA[SET]) : "a"(da[1]), "a"(vf[1]));                              \
    if (VALU) {                                                                                                                     \
      /* the statistics must be in registers before the asm instructions below read them: the compiler waits for loads it issued   */ \
      /* only in front of uses it can see, and it sees asm operands — pin them                                                     */ \
      asm volatile("" : "+v"(lst[0]), "+v"(lst[1]), "+v"(lst[2]), "+v"(lst[3]), "+v"(dst[0]), "+v"(dst[1]), "+v"(dst[2]), "+v"(dst[3])); \
    }                                                                                                                               \
    _Pragma("unroll") for (int ks = 2; ks < 8; ++ks) {                                                                              \
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(sA[SET]) : "a"(qa[ks]), "a"(kf[ks]));                           \
      if (VALU) {                                                                                                                   \
        const int r0 = ks < 6 ? 3 * (ks - 2) : 12 + 2 * (ks - 6);                                                                   \
        DKDV_WS_VALU3((SET) ^ 1, r0)                                                                                                \
        if (ks < 6) DKDV_WS_VALU3((SET) ^ 1, r0 + 1)                                                                                \
      }                                                                                                                             \
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(dpA[SET]) : "a"(da[ks]), "a"(vf[ks]));                          \
      if (VALU) {                                                                                                                   \
        const int r1 = ks < 6 ? 3 * (ks - 2) + 2 : 12 + 2 * (ks - 6) + 1;                                                           \
        DKDV_WS_VALU3((SET) ^ 1, r1)                                                                                                \
      }                                                                                                                             \
    }                                                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                                              \
  }
    // H2: the rest of the arithmetic of register set SET (dS = p * dd), packed operands into the pair's slot; SUB = index of the 32-row
    // query sub-tile these are the products of (DS: where its dS block goes)
#define DKDV_WS_SOFTMAX(SET, SUB)                                                                                                   \
  {                                                                                                                                 \
    _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                                                  \
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(dpA[SET][r]) : "v"(dpA[SET][r]), "v"(sA[SET][r]));                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                                              \
    *reinterpret_cast<lds_s16x8*>(xw) = pack_acc8(sA[SET], 0);                                                                      \
    *reinterpret_cast<lds_s16x8*>(xw + 1024) = pack_acc8(sA[SET], 8);                                                               \
    const s16x8_t ds_lo = pack_acc8(dpA[SET], 0), ds_hi = pack_acc8(dpA[SET], 8);                                                   \
    *reinterpret_cast<lds_s16x8*>(xw + 2048) = ds_lo;                                                                               \
    *reinterpret_cast<lds_s16x8*>(xw + 3072) = ds_hi;                                                                               \
    if (DS == 1) {                                                                                                                  \
      s16x8_t* dst = reinterpret_cast<s16x8_t*>(ds_base + (long)(SUB) * 1024);                                                      \
      __builtin_nontemporal_store(ds_lo, dst);                                                                                      \
      __builtin_nontemporal_store(ds_hi, dst + 64);                                                                                 \
    } else if (DS == 2) { /* the same blocks with plain (L2-allocating) stores: A/B of the store flavour */                         \
      s16x8_t* dst = reinterpret_cast<s16x8_t*>(ds_base + (long)(SUB) * 1024);                                                      \
      dst[0] = ds_lo;                                                                                                               \
      dst[64] = ds_hi;                                                                                                              \
    } else if (DS == 5) { /* probe: the same store instructions into ONE 2-KiB block per wave (stays in L2: issue cost without HBM) */ \
      s16x8_t* dst = reinterpret_cast<s16x8_t*>(ds_base);                                                                           \
      dst[0] = ds_lo;                                                                                                               \
      dst[64] = ds_hi;                                                                                                              \
    }                                                                                                                               \
  }
    // prologue: S / dP of sub-tile 0 + its statistics, fragments of sub-tile 1
    DKDV_WS_FRAGS(0, sm)
    DKDV_WS_MMA(0, 0, , 0, 0)
    DKDV_WS_FRAGS(1, sm)
    // state on entry of iteration t: sA[0] / dpA[0] = products of sub-tile 2t, qa / da = fragments of sub-tile 2t + 1
    for (int t = 0; t < ntile; ++t) {
      const lds_char* tn1 = sm + ((t + 1) % NSLOT) * TILE;
      // ---- step 2t.  H1: products of sub-tile 2t + 1 -> set 1 || t, dd, p of sub-tile 2t (set 0); pieces 2-3 of tile t + 2
      DKDV_WS_MMA(1, 1, issue_pieces(t + 2, 2);, t, 0)
      // this wave's pieces of tile t + 1 have landed (the four younger ones stay in flight).  DS: the four dS stores of the previous iteration
      // sit in the same in-order queue BEHIND those pieces (gfx9 counts stores in vmcnt): old -> new = [tile t+1 ...] [tile t+2 p0-1]
      // st st [tile t+2 p2-3]... i.e. eight younger operations, not four — with vmcnt(4) the wave would wait for the stores'
      // acknowledgements and for half of the next tile (measured: +0.66 ms per layer-launch instead of the stores' own cost)
      if (DS) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      DKDV_WS_BAR(t)
      // H2: dS of sub-tile 2t, hand-over; fragments of sub-tile 2t + 2 = rows 0-31 of tile t + 1
      DKDV_WS_FRAGS(0, tn1)  // issued first: they return under the arithmetic (the barrier's lgkmcnt(0) waits for every LDS operation of the wave)
      DKDV_WS_SOFTMAX(0, 2 * t)
      DKDV_WS_BAR(t)
      // ---- step 2t + 1.  H1: products of sub-tile 2t + 2 -> set 0 (past the last tile: zeros, never read); pieces 0-1 of tile t + 3;
      //      statistics of sub-tile 2t + 1
      DKDV_WS_MMA(0, 1, issue_pieces(t + 3, 0);, t, 1)
      DKDV_WS_BAR(t)
      // H2: dS of sub-tile 2t + 1, hand-over; fragments of sub-tile 2t + 3 = rows 32-63 of tile t + 1
      DKDV_WS_FRAGS(1, tn1)
      DKDV_WS_SOFTMAX(1, 2 * t + 1)
      DKDV_WS_BAR(t)
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : DKDV_WS_PIN8A(qa), DKDV_WS_PIN8A(da));  // the last (unused) fetches retire before the wave ends
#undef DKDV_WS_SOFTMAX
#undef DKDV_WS_MMA
#undef DKDV_WS_VALU3
#undef DKDV_WS_PIN8A
#undef DKDV_WS_FRAGS
#undef DKDV_WS_FRAG1
  } else {
    // ================================================================ consumer
    f32x16_t dk[4], dv[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      dk[d] = zero16();
      dv[d] = zero16();
    }
    const int gq = (lane >> 4) & 1, i16 = lane & 15, lh = (i16 & 3) >> 1;
    const unsigned ln_lo = gq * SUBP + (i16 & 1) * 8 + (4 * h + (i16 >> 2)) * 32 + (lh << 4);
    const unsigned ln_hi = ln_lo + 8 * 32 + ((lh ^ 1) - lh) * 16;
    const lds_char* const xr = sm + DKDV_WS_XOFF + w * 4096 + lane * 16;
    s16x8_t pf[2], df[2];
    s16x4_t dlo[8], dhi[8], qlo[8], qhi[8];
    // H1 of a consumer step: operands of the sub-tile in row half RH of the tile at TB into registers
#define DKDV_WS_TR1(RH, KK, D)                                                                             \
  tr16_issue_off<ST + (32 * (RH) + 16 * (KK)) * 32 + 2 * (D) * SUBP>(dlo[4 * (KK) + (D)], tlo);            \
  tr16_issue_off<ST + (32 * (RH) + 16 * (KK)) * 32 + 2 * (D) * SUBP>(dhi[4 * (KK) + (D)], thi);            \
  tr16_issue_off<(32 * (RH) + 16 * (KK)) * 32 + 2 * (D) * SUBP>(qlo[4 * (KK) + (D)], tlo);                 \
  tr16_issue_off<(32 * (RH) + 16 * (KK)) * 32 + 2 * (D) * SUBP>(qhi[4 * (KK) + (D)], thi);
#define DKDV_WS_CLOAD(RH, TB)                                                                                                   \
  {                                                                                                                             \
    pf[0] = *reinterpret_cast<const lds_s16x8*>(xr);                                                                            \
    pf[1] = *reinterpret_cast<const lds_s16x8*>(xr + 1024);                                                                     \
    df[0] = *reinterpret_cast<const lds_s16x8*>(xr + 2048);                                                                     \
    df[1] = *reinterpret_cast<const lds_s16x8*>(xr + 3072);                                                                     \
    const unsigned tlo = (unsigned)(size_t)(TB) + ln_lo, thi = (unsigned)(size_t)(TB) + ln_hi;                                  \
    DKDV_WS_TR1(RH, 0, 0) DKDV_WS_TR1(RH, 0, 1) DKDV_WS_TR1(RH, 0, 2) DKDV_WS_TR1(RH, 0, 3)                                     \
    DKDV_WS_TR1(RH, 1, 0) DKDV_WS_TR1(RH, 1, 1) DKDV_WS_TR1(RH, 1, 2) DKDV_WS_TR1(RH, 1, 3)                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" : TR_PIN8(dlo), TR_PIN8(dhi));                                                          \
    asm volatile("" : TR_PIN8(qlo), TR_PIN8(qhi));                                                                              \
    asm volatile("" : "+v"(pf[0]), "+v"(pf[1]), "+v"(df[0]), "+v"(df[1]));                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                                          \
  }
    // H2: dV += P^T dO, dK += dS^T Q with the operands in registers; two LDS-DMA pieces (DMA_STMT) ride in the matrix stream
#define DKDV_WS_CMMA(DMA_STMT)                                                                                                  \
  {                                                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                                          \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                                            \
      _Pragma("unroll") for (int d = 0; d < 4; ++d) {                                                                           \
        dv[d] = mfma32(pf[kk], join_lohi(dlo[4 * kk + d], dhi[4 * kk + d]), dv[d]);                                            \
        dk[d] = mfma32(df[kk], join_lohi(qlo[4 * kk + d], qhi[4 * kk + d]), dk[d]);                                            \
        if (kk == 0 && d == 1) {                                                                                                \
          __builtin_amdgcn_sched_barrier(0);                                                                                    \
          DMA_STMT                                                                                                              \
          __builtin_amdgcn_sched_barrier(0);                                                                                    \
        }                                                                                                                       \
      }                                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                          \
  }
    for (int t = 0; t < ntile; ++t) {
      const lds_char* tcur = sm + (t % NSLOT) * TILE;
      // ---- step 2t.  H1: operands of sub-tile 2t - 1 = rows 32-63 of tile t - 1
      if (t > 0) {
        const lds_char* tprev = sm + ((t + NSLOT - 1) % NSLOT) * TILE;
        DKDV_WS_CLOAD(1, tprev)
      }
      asm volatile("s_waitcnt vmcnt(5)" ::: "memory");  // this wave's pieces (and statistics) of tile t + 1 have landed; tile t + 2's five stay in flight
      DKDV_WS_BAR(t)
      // H2: its products; statistics + pieces 0-1 of tile t + 3
      if (t > 0) {
        DKDV_WS_CMMA(issue_stats(t + 3); issue_pieces(t + 3, 0);)
      } else {
        issue_stats(t + 3);
        issue_pieces(t + 3, 0);
      }
      DKDV_WS_BAR(t)
      // ---- step 2t + 1.  H1: operands of sub-tile 2t = rows 0-31 of tile t
      DKDV_WS_CLOAD(0, tcur)
      DKDV_WS_BAR(t)
      DKDV_WS_CMMA(issue_pieces(t + 3, 2);)
      DKDV_WS_BAR(t)
    }
    {  // the last sub-tile: rows 32-63 of the last tile
      const lds_char* tlast = sm + ((ntile - 1) % NSLOT) * TILE;
      DKDV_WS_CLOAD(1, tlast)
      DKDV_WS_CMMA()
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tiles issued past the end retire before the wave ends
#undef DKDV_WS_CMMA
#undef DKDV_WS_CLOAD
#undef DKDV_WS_TR1
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kv = kvw + crow(r, h);
        if (kv < Skv) {
          const long off = ((long)b * Skv + kv);
          p.dK[off * p.lddk + hd * 128 + 32 * d + l31] = f2bf(dk[d][r] * p.scale);
          p.dV[off * p.lddv + hd * 128 + 32 * d + l31] = f2bf(dv[d][r]);
        }
      }
  }
}
#undef DKDV_WS_BAR

// ============================================================================================ backward: dQ
// grid = ceil(S/128) * H * B workgroups (attn_wg_coords); wave w owns query rows [q0 + 32 w, +32) (Q, dO fragments in registers, dQ^T accumulators);
// loops over KV tiles of 64 rows (K, V row-major in LDS).  S^T = K Q^T, dP^T = V dO^T, dS^T = P (dP^T - delta[q]);
// dQ^T[d][q] += sum_kv K^T[d][kv] dS^T[kv][q]  (K through tr16, dS^T straight from registers).
template <int KS, int DB>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AitkAttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* const sm = (lds_char*)smem;  // buffer b: K tile (sub-tiled: b128 + tr16 reads) at b*DBUF, V tile (sub-tiled: b128 reads) after it
  constexpr int DBUF = 2 * SUBTILE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int S = p.S;
  const int Skv = p.Skv > 0 ? p.Skv : p.S;
  int tile_x, hd, b;
  attn_wg_coords((S + 127) / 128, p.H, tile_x, hd, b);
  const int HS = p.hstride > 0 ? p.hstride : 128;  // elements between heads: 128 (padded layout) or the native head width
  const int q0 = tile_x * 128 + wave * 32;
  const bf16_t* Qb = p.Q + (long)b * S * p.ldq + hd * HS;
  const bf16_t* Kb = p.K + (long)b * Skv * p.ldk + hd * HS;
  const bf16_t* Vb = p.V + (long)b * Skv * p.ldv + hd * HS;
  const bf16_t* dOb = p.dO + (long)b * S * p.lddo + hd * HS;
  const int qr = min(q0 + l31, S - 1);
  s16x8_t qf[KS], gf[KS];
  {
    const bf16_t* qp = Qb + (long)qr * p.ldq + 8 * h;
    const bf16_t* gp = dOb + (long)qr * p.lddo + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = *reinterpret_cast<const s16x8_t*>(qp + 16 * ks);
      gf[ks] = *reinterpret_cast<const s16x8_t*>(gp + 16 * ks);
    }
  }
  const float L2 = p.LSE[((long)b * p.H + hd) * S + qr];
  const float dl = p.delta[((long)b * p.H + hd) * S + qr];
  f32x16_t dq[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d) dq[d] = zero16();
  const float c2 = p.scale * 1.4426950408889634f;
  const int ntiles = (Skv + 63) / 64;
  // Per-lane parts of every LDS fragment address, computed once: inside the tile loop an address is (tile base + lane part), one add per
  // distinct lane part, plus a compile-time offset that goes into the instruction's immediate field.  (Left to the address helpers the
  // swizzle arithmetic was redone for each of the 64 fragment reads of a tile: 72 v_add per tile in a kernel that is vector-issue bound.)
  typedef const __attribute__((address_space(3))) s16x8_t* lds_v8;
  const int gq = (lane >> 4) & 1, i16 = lane & 15, lh = (i16 & 3) >> 1;
  const unsigned ln_rm = l31 * 32 + ((h ^ ((l31 >> 3) & 1)) << 4);                                     // frag_rm_st (K and V, sub-tiled)
  const unsigned ln_tr_lo = gq * SUBP + (i16 & 1) * 8 + (4 * h + (i16 >> 2)) * 32 + (lh << 4);          // frag_tr_perm_st, rows with bit 3 clear
  const unsigned ln_tr_hi = gq * SUBP + (i16 & 1) * 8 + (4 * h + (i16 >> 2) + 8) * 32 + ((lh ^ 1) << 4);  // ... and the rows 8 further
  const v4i_t srdK = slice_srd(Kb, p.ldk, Skv, HS), srdV = slice_srd(Vb, p.ldv, Skv, HS);
  unsigned vK[4], vV[4];
  st_voff(vK, p.ldk, wave, lane);
  st_voff(vV, p.ldv, wave, lane);
  const unsigned stepK = (unsigned)(64 * p.ldk * 2), stepV = (unsigned)(64 * p.ldv * 2);
  dma_st64(sm, vK, 0, srdK, wave);
  dma_st64(sm + SUBTILE_BYTES, vV, 0, srdV, wave);
  DMA_WAIT_ALL();
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    const lds_char* ktc = sm + cur * DBUF;
    const lds_char* vtc = ktc + SUBTILE_BYTES;
    if (t + 1 < ntiles) {
      dma_st64(sm + (cur ^ 1) * DBUF, vK, (t + 1) * stepK, srdK, wave);
      dma_st64(sm + (cur ^ 1) * DBUF + SUBTILE_BYTES, vV, (t + 1) * stepV, srdV, wave);
    }
    const lds_char* k_rm = ktc + ln_rm;
    const lds_char* v_rm = vtc + ln_rm;
    const lds_char* k_lo = ktc + ln_tr_lo;
    const lds_char* k_hi = ktc + ln_tr_hi;
    const bool tail = t * 64 + 64 > Skv;
    {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16_t s = zero16(), dp = zero16();
#pragma unroll
        for (int hh = 0; hh < (KS + 1) / 2; ++hh) {
          s16x8_t kfr[2], vfr[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (2 * hh + u >= KS) continue;
            kfr[u] = *reinterpret_cast<lds_v8>(k_rm + (2 * hh + u) * SUBP + 32 * j * 32);
            vfr[u] = *reinterpret_cast<lds_v8>(v_rm + (2 * hh + u) * SUBP + 32 * j * 32);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (2 * hh + u >= KS) continue;
            s = mfma32(kfr[u], qf[2 * hh + u], s);
            dp = mfma32(vfr[u], gf[2 * hh + u], dp);
          }
        }
        {
          const f32x2_t c2v = {c2, c2}, lv = {-L2, -L2}, dlv = {dl, dl};
#pragma unroll
          for (int r = 0; r < 16; r += 2) {  // register pairs (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32): this kernel is vector-issue bound
                                             // (profiles/r02_notes_attention_ablation.md); backward 4.29 -> 4.12 ms same-box at B = 4
            f32x2_t a = {s[r], s[r + 1]};
            a = __builtin_elementwise_fma(a, c2v, lv);
            a[0] = __builtin_amdgcn_exp2f(a[0]);
            a[1] = __builtin_amdgcn_exp2f(a[1]);
            if (tail) {
              if (t * 64 + 32 * j + crow(r, h) >= Skv) a[0] = 0.f;
              if (t * 64 + 32 * j + crow(r + 1, h) >= Skv) a[1] = 0.f;
            }
            f32x2_t d = {dp[r], dp[r + 1]};
            d = a * (d - dlv);
            dp[r] = d[0];
            dp[r + 1] = d[1];
          }
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const s16x8_t df = pack_acc8(dp, 8 * kk);
#pragma unroll
          for (int d = 0; d < DB; ++d) {
            // rows kb + 4h + (i>>2) (+8) of column block 2d + gq: kb = 32j + 16kk is a multiple of 16, so the first read sits in a row
            // with bit 3 clear (lane part ln_tr_lo), the second 8 rows further (ln_tr_hi)
            const s16x4_t lo = tr16l(k_lo + 2 * d * SUBP + (32 * j + 16 * kk) * 32);
            const s16x4_t hi = tr16l(k_hi + 2 * d * SUBP + (32 * j + 16 * kk) * 32);
            dq[d] = mfma32(join_lohi(lo, hi), df, dq[d]);
          }
        }
      }
    }
    DMA_WAIT_ALL();
    __syncthreads();
  }
  const int q = q0 + l31;
  if (q < S) {
    bf16_t* op = p.dQ + ((long)b * S + q) * p.lddq + hd * HS;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (d >= DB && 32 * d >= HS) continue;
        uint2 u = make_uint2(0u, 0u);
        if (d < DB) {
          u.x = pack2bf(dq[d < DB ? d : 0][4 * g + 0] * p.scale, dq[d < DB ? d : 0][4 * g + 1] * p.scale);
          u.y = pack2bf(dq[d < DB ? d : 0][4 * g + 2] * p.scale, dq[d < DB ? d : 0][4 * g + 3] * p.scale);
        }
        *reinterpret_cast<uint2*>(op + 32 * d + 8 * g + 4 * h) = u;
      }
  }
}

// ============================================================================================ backward: dQ as a product of its own (5-matmul backward)
// dQ = scale * dS K with the bf16 dS the dK/dV pass emitted (AitkAttnArgs.dS, ds_mode 1) instead of recomputing S and dP a second time: one
// matmul where attn_bwd_dq_kernel runs three, at the price of streaming the [S, Skv] dS of every (batch, head) through HBM once (42 MB per
// head at 4608 tokens) — an HBM-bound kernel whose matrix work hides under the stream.
// Same ownership and accumulation order as attn_bwd_dq_kernel (wave w owns query rows [q0 + 32 w, +32); kv tiles of 64 ascending, sub-blocks
// j, kk ascending; dQ^T[d][q] += sum_kv K^T[d][kv] dS^T[kv][q] with K^T through tr16 in the permuted contraction order), and the SAME bf16
// dS values (the dK/dV pass forms them with the same instructions), so the result is the recomputing kernel's.
// The dS blocks are accumulator-native (lane = kv row, slots = 8 q rows; see the header): the contraction index here is kv, i.e. each
// 32 x 32 block must be read TRANSPOSED — it is staged by LDS-DMA into a wave-private slot with the 16-byte chunks reordered on the way
// (DMA lane L fetches the chunk of dump-lane (L & 1) * 32 + (L >> 1): LDS chunk position = 2 kv + h'), which makes the 4-row x 16-column
// patch of every ds_read_b64_tr_b16 group 128 contiguous bytes (conflict-free), and read with two transpose reads per 16 kv rows.
// LDS per workgroup: K tile double buffer (2 x 17 KiB, sub-tiled like attn_bwd_dq_kernel's) + per wave 2 x 4 KiB of dS slots = 66 KiB -> 2 WG / CU.
#define DQDS_SLOT 4096                                  /* one wave's two blocks (j = 0, 1) of a kv tile */
#define DQDS_KOFF 0
#define DQDS_SOFF (2 * SUBTILE_BYTES)                  /* [buffer 2][wave 4] slots */
#define DQDS_LDS (DQDS_SOFF + 2 * 4 * DQDS_SLOT)
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_ds_kernel(AitkAttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* const sm = (lds_char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int S = p.S;
  const int Skv = p.Skv > 0 ? p.Skv : p.S;
  int tile_x, hd, b;
  attn_wg_coords((S + 127) / 128, p.H, tile_x, hd, b);
  const int q0 = tile_x * 128 + wave * 32;
  const bf16_t* Kb = p.K + (long)b * Skv * p.ldk + hd * 128;
  f32x16_t dq[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) dq[d] = zero16();
  const int ntiles = Skv / 64;  // whole tiles only (checked at launch)
  const int gq = (lane >> 4) & 1, i16 = lane & 15, lh = (i16 & 3) >> 1;
  const unsigned ln_tr_lo = gq * SUBP + (i16 & 1) * 8 + (4 * h + (i16 >> 2)) * 32 + (lh << 4);            // K^T: frag_tr_perm_st, rows with bit 3 clear
  const unsigned ln_tr_hi = gq * SUBP + (i16 & 1) * 8 + (4 * h + (i16 >> 2) + 8) * 32 + ((lh ^ 1) << 4);  // ... and the rows 8 further
  // dS^T: rows R0 + (i16 >> 2), R0 = 16 kk + 4 h (+ 8); LDS chunk position 2 kv + h' with h' = i16 & 1, 8-byte half (i16 >> 1) & 1; q column
  // half gq = the block's second 1-KiB vector
  const unsigned ln_ds = gq * 1024 + ((4 * h + (i16 >> 2)) * 2 + (i16 & 1)) * 16 + ((i16 >> 1) & 1) * 8;
  const v4i_t srdK = slice_srd(Kb, p.ldk, Skv, 128);
  unsigned vK[4];
  st_voff(vK, p.ldk, wave, lane);
  const unsigned stepK = (unsigned)(64 * p.ldk * 2);
  // this wave's dS blocks: block (kv32 = 2 t + j, q32 = q0 / 32) of (b, hd) at ((kv32 * nq32 + q32) * 2048 bytes; per DMA piece v (1 KiB)
  // lane L fetches 16 bytes at v * 1024 + ((L & 1) * 32 + (L >> 1)) * 16
  const long nq32 = S / 32, nkv32 = Skv / 32;
  const bf16_t* dSb = p.dS + ((long)b * p.H + hd) * nkv32 * nq32 * 1024;
  const v4i_t srdS = slice_srd_bytes(dSb, nkv32 * nq32 * 2048);
  const unsigned vS = (unsigned)((q0 >> 5) * 2048 + ((lane & 1) * 32 + (lane >> 1)) * 16);
  const unsigned stepS = (unsigned)(nq32 * 2048);  // one kv32 block further
  lds_char* const slot0 = sm + DQDS_SOFF + wave * DQDS_SLOT;
  auto stage = [&](int t, int buf) {
    dma_st64(sm + DQDS_KOFF + buf * SUBTILE_BYTES, vK, (unsigned)t * stepK, srdK, wave);
    lds_char* sl = slot0 + buf * 4 * DQDS_SLOT;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 2; ++v)
        dma16(__builtin_amdgcn_readfirstlane((unsigned)(size_t)sl + j * 2048 + v * 1024), vS + (unsigned)(2 * t + j) * stepS + v * 1024, srdS);
  };
  stage(0, 0);
  DMA_WAIT_ALL();
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntiles) stage(t + 1, cur ^ 1);
    const lds_char* ktc = sm + DQDS_KOFF + cur * SUBTILE_BYTES;
    const lds_char* k_lo = ktc + ln_tr_lo;
    const lds_char* k_hi = ktc + ln_tr_hi;
    const lds_char* ds_l = slot0 + cur * 4 * DQDS_SLOT + ln_ds;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const s16x4_t dlo = tr16l(ds_l + j * 2048 + (16 * kk) * 32);        // rows 16 kk + 4 h + 0..3 of q column l31
        const s16x4_t dhi = tr16l(ds_l + j * 2048 + (16 * kk + 8) * 32);    // rows 16 kk + 8 + 4 h + 0..3
        const s16x8_t df = join_lohi(dlo, dhi);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const s16x4_t lo = tr16l(k_lo + 2 * d * SUBP + (32 * j + 16 * kk) * 32);
          const s16x4_t hi = tr16l(k_hi + 2 * d * SUBP + (32 * j + 16 * kk) * 32);
          dq[d] = mfma32(join_lohi(lo, hi), df, dq[d]);
        }
      }
    DMA_WAIT_ALL();
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

// ---- the same product with more bytes in flight: 8 waves (256 query rows) per workgroup, rings of THREE tiles, counted waits.
// attn_bwd_dq_ds_kernel keeps 2 workgroups x 16 KiB of dS in flight per CU and drains the queue at every tile (5.0 TB/s on the 7.1-GB stream);
// here a workgroup's eight waves share one K tile, every wave has the dS blocks of TWO tiles in flight behind the one it computes (64 KiB
// per CU), and a tile is published by `vmcnt(6)` — a wave issues exactly six LDS-DMA pieces per tile (2 of K, 4 of dS), tiles past the end
// are issued all the same (they arrive as zeros through the buffer windows) so that the count stays exact.  Transpose reads in their asm form:
// the builtin would make the waitcnt pass drain the DMA queue in front of the first LDS read of every iteration (see tr16_issue).
// LDS: K ring 3 x 17 KiB + dS slots 3 x 8 x 4 KiB = 147 KiB -> 1 workgroup per CU.  Same products in the same order -> the same bits.
#define DQ8_KRING (3 * SUBTILE_BYTES)
#define DQ8_LDS (DQ8_KRING + 3 * 8 * DQDS_SLOT)
__global__ __launch_bounds__(512) void attn_bwd_dq_ds8_kernel(AitkAttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* const sm = (lds_char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int S = p.S;
  const int Skv = p.Skv > 0 ? p.Skv : p.S;
  int tile_x, hd, b;
  attn_wg_coords((S + 255) / 256, p.H, tile_x, hd, b);
  const int q0 = tile_x * 256 + wave * 32;
  const bf16_t* Kb = p.K + (long)b * Skv * p.ldk + hd * 128;
  f32x16_t dq[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) dq[d] = zero16();
  const int ntiles = Skv / 64;
  const int gq = (lane >> 4) & 1, i16 = lane & 15, lh = (i16 & 3) >> 1;
  const unsigned ln_tr_lo = gq * SUBP + (i16 & 1) * 8 + (4 * h + (i16 >> 2)) * 32 + (lh << 4);
  const unsigned ln_tr_hi = gq * SUBP + (i16 & 1) * 8 + (4 * h + (i16 >> 2) + 8) * 32 + ((lh ^ 1) << 4);
  const unsigned ln_ds = gq * 1024 + ((4 * h + (i16 >> 2)) * 2 + (i16 & 1)) * 16 + ((i16 >> 1) & 1) * 8;
  const v4i_t srdK = slice_srd(Kb, p.ldk, Skv, 128);
  // K tile [64][128], sub-tiled: 16 pieces of 1 KiB (column block ins >> 1, row half ins & 1), two per wave: ins = wave + 8 ii
  unsigned vK[2];
#pragma unroll
  for (int ii = 0; ii < 2; ++ii) {
    const int ins = wave + 8 * ii;
    const int sub = ins >> 1, r = (ins & 1) * 32 + (lane >> 1);
    const int lhk = (lane & 1) ^ ((r >> 3) & 1);
    vK[ii] = (unsigned)((r * p.ldk + (2 * sub + lhk) * 8) * 2);
  }
  const unsigned stepK = (unsigned)(64 * p.ldk * 2);
  const long nq32 = S / 32, nkv32 = Skv / 32;
  const bf16_t* dSb = p.dS + ((long)b * p.H + hd) * nkv32 * nq32 * 1024;
  const v4i_t srdS = slice_srd_bytes(dSb, nkv32 * nq32 * 2048);
  // a wave past the last query block (S % 256 != 0) points its dS pieces past the window: zeros, and it writes nothing at the end
  const unsigned vS = q0 < S ? (unsigned)((q0 >> 5) * 2048 + ((lane & 1) * 32 + (lane >> 1)) * 16) : 0xfff00000u;
  const unsigned stepS = (unsigned)(nq32 * 2048);
  lds_char* const slot0 = sm + DQ8_KRING + wave * DQDS_SLOT;
  auto stage = [&](int t) {  // six pieces, always
    const int buf = t % 3;
    lds_char* kt = sm + buf * SUBTILE_BYTES;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      const int ins = wave + 8 * ii;
      dma16(__builtin_amdgcn_readfirstlane((unsigned)(size_t)kt + (ins >> 1) * SUBP + (ins & 1) * 1024), vK[ii] + (unsigned)t * stepK, srdK);
    }
    lds_char* sl = slot0 + buf * 8 * DQDS_SLOT;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 2; ++v)
        dma16(__builtin_amdgcn_readfirstlane((unsigned)(size_t)sl + j * 2048 + v * 1024), vS + (unsigned)(2 * t + j) * stepS + v * 1024, srdS);
  };
  stage(0);
  stage(1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    stage(t + 2);
    const int cur = t % 3;
    const unsigned kt = (unsigned)(size_t)(sm + cur * SUBTILE_BYTES);
    const unsigned k_lo = kt + ln_tr_lo, k_hi = kt + ln_tr_hi;
    const unsigned ds_l = (unsigned)(size_t)(slot0 + cur * 8 * DQDS_SLOT) + ln_ds;
#define DQ8_STEP(J, KK)                                                                                              \
  {                                                                                                                  \
    s16x4_t dlo, dhi, klo[4], khi[4];                                                                                \
    tr16_issue_off<(J) * 2048 + (16 * (KK)) * 32>(dlo, ds_l);                                                        \
    tr16_issue_off<(J) * 2048 + (16 * (KK) + 8) * 32>(dhi, ds_l);                                                    \
    tr16_issue_off<0 * 2 * SUBP + (32 * (J) + 16 * (KK)) * 32>(klo[0], k_lo);                                        \
    tr16_issue_off<0 * 2 * SUBP + (32 * (J) + 16 * (KK)) * 32>(khi[0], k_hi);                                        \
    tr16_issue_off<1 * 2 * SUBP + (32 * (J) + 16 * (KK)) * 32>(klo[1], k_lo);                                        \
    tr16_issue_off<1 * 2 * SUBP + (32 * (J) + 16 * (KK)) * 32>(khi[1], k_hi);                                        \
    tr16_issue_off<2 * 2 * SUBP + (32 * (J) + 16 * (KK)) * 32>(klo[2], k_lo);                                        \
    tr16_issue_off<2 * 2 * SUBP + (32 * (J) + 16 * (KK)) * 32>(khi[2], k_hi);                                        \
    tr16_issue_off<3 * 2 * SUBP + (32 * (J) + 16 * (KK)) * 32>(klo[3], k_lo);                                        \
    tr16_issue_off<3 * 2 * SUBP + (32 * (J) + 16 * (KK)) * 32>(khi[3], k_hi);                                        \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                              \
                 : "+v"(dlo), "+v"(dhi), "+v"(klo[0]), "+v"(khi[0]), "+v"(klo[1]), "+v"(khi[1]), "+v"(klo[2]), "+v"(khi[2]), "+v"(klo[3]), \
                   "+v"(khi[3]));                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    const s16x8_t df = join_lohi(dlo, dhi);                                                                          \
    _Pragma("unroll") for (int d = 0; d < 4; ++d) dq[d] = mfma32(join_lohi(klo[d], khi[d]), df, dq[d]);              \
  }
    DQ8_STEP(0, 0)
    DQ8_STEP(0, 1)
    DQ8_STEP(1, 0)
    DQ8_STEP(1, 1)
#undef DQ8_STEP
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // tile t + 1 has landed; tile t + 2's six pieces stay in flight
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tiles issued past the end retire before the wave ends
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
  if (a->Dv < 0 || a->Dv > 128) return AITK_ERR_SHAPE;
  // native head layout (head h at column h * hstride): the head must fill whole contraction steps and output blocks, because the tiles
  // still fetch 128 columns per row and everything past the head is the NEXT head's data, not zeros
  // (64 and 96 are the widths whose (KS, DB) instantiation is exact; 32 would run the 48-column variant)
  if (a->hstride < 0 || (a->hstride > 0 && (a->hstride != a->Dv || (a->hstride != 64 && a->hstride != 96)))) return AITK_ERR_SHAPE;
  if ((a->ldq % 8) || (a->ldk % 8) || (a->ldv % 8) || (a->ldo % 8)) return AITK_ERR_ALIGN;
  return AITK_OK;
}

// (KS, DB) instantiation for a valid head width Dv inside the 128-column layout: KS = ceil(Dv / 16), DB = ceil(Dv / 32)
// 0: 128 (8, 4)   1: <= 96 (6, 3)   2: <= 80 (5, 3)   3: <= 64 (4, 2)   4: <= 48 (3, 2)
static int attn_variant(int Dv) {
  if (Dv <= 0 || Dv > 96) return 0;
  if (Dv > 80) return 1;
  if (Dv > 64) return 2;
  if (Dv > 48) return 3;
  return 4;
}

template <int KS, int DB>
static void launch_fwd(const AitkAttnArgs* a, hipStream_t s) {
  const size_t lds = 2 * (16384 + SUBTILE_BYTES);
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<KS, DB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  dim3 grid((unsigned)(((a->S + 127) / 128) * a->H * a->B));  // 1-D: attn_wg_coords deals the (batch, head, tile) list to the XCDs
  hipLaunchKernelGGL((attn_fwd_kernel<KS, DB>), grid, dim3(256), lds, s, *a);
}

// AITK_ATTN_DKDV_WS: 1 (default) = the wave-specialised dK / dV kernel for head_dim 128, 0 = the pipelined one-wave-per-SIMD kernel (same bits),
// 2 = the wave-specialised kernel with the barrier trace.  Read at every launch (57 per step): tests and A/B runs switch it inside one process.
static int dkdv_ws_mode() {
  const char* e = getenv("AITK_ATTN_DKDV_WS");
  return (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1;
}
static bool dkdv_ws_enabled() { return dkdv_ws_mode() != 0; }
// the s_memtime stamps the TRACE instantiation left (64 values: [role 2][tile 8-11][barrier 4][before, after])
extern "C" int aitk_probe_attn_ws_trace(uint64_t* out64) {
  if (!out64) return AITK_ERR_ARG;
  return (int)hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_ws_trace), sizeof(unsigned long long) * 64, 0, hipMemcpyDeviceToHost);
}
// AITK_ATTN_DKDV_PIPE=0 selects the un-pipelined dK / dV kernel for head_dim 128 (same-box A/B); default: the pipelined one
static bool dkdv_pipe_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("AITK_ATTN_DKDV_PIPE");
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v == 1;
}

template <int KS, int DB>
static void launch_dq(const AitkAttnArgs* a, dim3 grid, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<KS, DB>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * SUBTILE_BYTES);
    attr = true;
  }
  hipLaunchKernelGGL((attn_bwd_dq_kernel<KS, DB>), grid, dim3(256), 4 * SUBTILE_BYTES, s, *a);
}
template <int KS, int DB>
static void launch_bwd(const AitkAttnArgs* a, hipStream_t s) {
  const int Skv = a->Skv > 0 ? a->Skv : a->S;
  dim3 grid((unsigned)(((a->S + 127) / 128) * a->H * a->B));
  dim3 grid_kv((unsigned)(((Skv + 127) / 128) * a->H * a->B));
  if (KS == 8 && DB == 4 && dkdv_ws_enabled()) {
    static bool wattr = false;
    if (!wattr) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkdv_ws_kernel<false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DKDV_WS_LDS);
      hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkdv_ws_kernel<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DKDV_WS_LDS);
      hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkdv_ws_kernel<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DKDV_WS_LDS);
      hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkdv_ws_kernel<false, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DKDV_WS_LDS);
      hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkdv_ws_kernel<true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DKDV_WS_LDS);
      hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<KS, DB>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * SUBTILE_BYTES);
      wattr = true;
    }
    // dS modes cover whole tiles only, full-width heads in the padded layout (FLUX, Wan self-attention); everything else recomputes
    const int dsm = (a->dS && (a->S % 64) == 0 && (Skv % 128) == 0 && a->hstride == 0 && (a->Dv == 0 || a->Dv == 128)) ? a->ds_mode : 0;
    if (dkdv_ws_mode() == 2) hipLaunchKernelGGL((attn_bwd_dkdv_ws_kernel<true, 0>), grid_kv, dim3(512), DKDV_WS_LDS, s, *a);
    else if (dsm == 1 || dsm == 3) hipLaunchKernelGGL((attn_bwd_dkdv_ws_kernel<false, 1>), grid_kv, dim3(512), DKDV_WS_LDS, s, *a);
    else if (dsm == 2) hipLaunchKernelGGL((attn_bwd_dkdv_ws_kernel<false, 2>), grid_kv, dim3(512), DKDV_WS_LDS, s, *a);
    else if (dsm == 5) hipLaunchKernelGGL((attn_bwd_dkdv_ws_kernel<false, 5>), grid_kv, dim3(512), DKDV_WS_LDS, s, *a);
    else hipLaunchKernelGGL((attn_bwd_dkdv_ws_kernel<false, 0>), grid_kv, dim3(512), DKDV_WS_LDS, s, *a);
    if (dsm == 1) {  // the 5-matmul backward: dQ = dS K from the emitted blocks
      static bool dattr = false;
      static int dq_variant = 2;  // AITK_ATTN_DQDS=1: the 4-wave / drain-every-tile kernel (same-box A/B)
      if (!dattr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_ds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DQDS_LDS);
        hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_ds8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DQ8_LDS);
        const char* e = getenv("AITK_ATTN_DQDS");
        if (e && e[0] == '1') dq_variant = 1;
        dattr = true;
      }
      if (dq_variant == 1) hipLaunchKernelGGL(attn_bwd_dq_ds_kernel, grid, dim3(256), DQDS_LDS, s, *a);
      else hipLaunchKernelGGL(attn_bwd_dq_ds8_kernel, dim3((unsigned)(((a->S + 255) / 256) * a->H * a->B)), dim3(512), DQ8_LDS, s, *a);
      return;
    }
    launch_dq<KS, DB>(a, grid, s);
    return;
  }
  if (KS == 8 && DB == 4 && dkdv_pipe_enabled()) {
    static bool pattr = false;
    if (!pattr) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkdv_pipe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DKDV_P_LDS);
      hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<KS, DB>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * SUBTILE_BYTES);
      pattr = true;
    }
    hipLaunchKernelGGL(attn_bwd_dkdv_pipe_kernel, grid_kv, dim3(256), DKDV_P_LDS, s, *a);
    launch_dq<KS, DB>(a, grid, s);
    return;
  }
  const size_t lds1 = 4 * SUBTILE_BYTES + 4 * 64 * sizeof(float);
  const size_t lds2 = 4 * SUBTILE_BYTES;
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkdv_kernel<KS, DB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<KS, DB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    attr = true;
  }
  hipLaunchKernelGGL((attn_bwd_dkdv_kernel<KS, DB>), grid_kv, dim3(256), lds1, s, *a);
  launch_dq<KS, DB>(a, grid, s);
}

extern "C" int aitk_attn_fwd(const AitkAttnArgs* a, aitk_stream_t stream) {
  int rc = attn_check(a);
  if (rc) return rc;
  if (!a->Q || !a->K || !a->V || !a->O || !a->LSE) return AITK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  switch (attn_variant(a->Dv)) {
    case 1: launch_fwd<6, 3>(a, s); break;
    case 2: launch_fwd<5, 3>(a, s); break;
    case 3: launch_fwd<4, 2>(a, s); break;
    case 4: launch_fwd<3, 2>(a, s); break;
    default: launch_fwd<8, 4>(a, s); break;
  }
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
  switch (attn_variant(a->Dv)) {
    case 1: launch_bwd<6, 3>(a, s); break;
    case 2: launch_bwd<5, 3>(a, s); break;
    case 3: launch_bwd<4, 2>(a, s); break;
    case 4: launch_bwd<3, 2>(a, s); break;
    default: launch_bwd<8, 4>(a, s); break;
  }
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}
