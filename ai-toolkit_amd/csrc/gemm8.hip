// Persistent 8-phase bf16 MFMA GEMM with the LoRA K-slab (gfx950 / MI355X) — the big-problem path of aitk_gemm_nt.
//
//   C[M,N] = epi( A[M,K] * B[N,K]^T  +  A2[M,K2] * B2[N,K2]^T  + bias[N] )          (same contract as gemm.hip)
//
// One 512-thread workgroup per CU (2 x 4 waves, 256 x 256 x 64 tile, two 64-KiB LDS tile buffers + 32 KiB epilogue
// scratch) walks output tiles blockIdx.x, +gridDim.x, ... in the XCD-grouped order.
//
// K loop ("8-phase" schedule of the CDNA4 guide §5, two K-tiles = 8 phases): a K-tile is four phases, one 64x32
// C-quadrant per wave each (8 x v_mfma_f32_32x32x16_bf16).  A phase is
//     [ds_read this quadrant's new fragments | stage ONE 16-KiB half-tile by LDS-DMA | counted vmcnt]  s_barrier
//     [counted lgkmcnt per k-substep | 8 MFMA]                                                          s_barrier
// and waves 4-7 (the second wave of every SIMD) run one barrier behind waves 0-3, so on each SIMD one wave is in its
// MFMA segment while its partner is in its load segment.
//   phase j   reads (LDS)               computes     stages (K-tile, half)    waits (after staging)
//     0       A0 rows(8) + B0 cols(4)   Q(0,0)       B1 of tile t+1           vmcnt(8) -> B1(t) landed   (read in j=1)
//     1       B1 cols (4)               Q(0,1)       A1 of tile t+1           vmcnt(8) -> A1(t) landed   (read in j=2)
//     2       A1 rows (8)               Q(1,1)       A0 of tile t+2           -
//     3       -  (B0 kept in VGPRs)     Q(1,0)       B0 of tile t+2           vmcnt(8) -> A0,B0(t+1)     (read in next j=0)
// RAW: a half-tile is read one phase after the vmcnt that retires it (plus the barrier in between).  WAR: it is re-staged
// >= 2 phases after its last ds_read.  Every stage call issues exactly two DMAs per wave so the vmcnt immediates are
// exact; vmcnt(8) = "everything but the last four half-tiles has landed" (>= 4 phases old).
// The K-tile counter runs across output tiles: "tile t+1 / t+2" past the end of this output tile are the first K-tiles
// of the workgroup's NEXT output tile, so its prologue loads fly during this tile's last phases and epilogue.
//
// LDS half-tiles are interleaved so a wave owns a contiguous 128 x 64 block of C: LDS A row R = half*128 + wr*64 + r
// holds tile row wr*128 + half*64 + r; LDS B row R = half*128 + wc*32 + r holds tile column wc*64 + half*32 + r.
// Rows are 128 B, 16-B chunks XOR-swizzled by (row>>1)&7 (conflict-free ds_read_b128), written lane-linearly by
// global_load_lds_dwordx4 with the swizzle applied on the global source address.
//
// Epilogue: each 32x32 accumulator block goes through a wave-private 4-KiB LDS patch (fp32, swizzled) so that a lane ends
// up with 8 consecutive columns of one row: bias / residual / gate / GELU are applied in fp32 exactly as in gemm.hip and
// C, aux_out are written (aux_in, C read) with 16-byte accesses covering 64-B row segments.
#include "common.h"
#include "aitk_args.h"

#define BK 64
#define BM 256
#define BN 256
#define NT 512
#define A_BYTES (BM * BK * 2)
#define BUF_BYTES (2 * A_BYTES)
#define EPI_OFF (2 * BUF_BYTES)

template <int V>
struct IC { static constexpr int value = V; };

// ds_read_b128 the compiler does not see as an LDS access: lgkmcnt is counted by hand in the K loop (the waitcnt pass
// otherwise puts lgkmcnt(0) in front of the first MFMA of every phase).
template <int OFF>
__device__ __forceinline__ void lds_read128(s16x8_t& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}

__device__ __forceinline__ const bf16_t* seg_row8(const bf16_t* base, long ld, int seg_rows, long seg_stride, int m) {
  if (seg_rows > 0) {
    int s = m / seg_rows;
    int w = m - s * seg_rows;
    return base + (long)s * seg_stride + (long)w * ld;
  }
  return base + (long)m * ld;
}

// element offset of row m under the (seg_rows, seg_stride) row map (fp8 operands: elements are bytes)
__device__ __forceinline__ long seg_off(long ld, int seg_rows, long seg_stride, int m) {
  if (seg_rows > 0) {
    int s = m / seg_rows;
    int w = m - s * seg_rows;
    return (long)s * seg_stride + (long)w * ld;
  }
  return (long)m * ld;
}

// W8A8 base segment (b_scale_mode 3): both operands are OCP e4m3 bytes and the K-tile is 128 elements = the same 128 B per LDS row as
// 64 bf16, so staging, LDS layout, swizzle and fragment reads are byte-for-byte those of the bf16 kernel.  One
// v_mfma_scale_f32_32x32x64_f8f6f4 (unit E8M0 block scales: 0x7f = 2^0) consumes TWO of the bf16 kernel's 16-B fragment reads per
// operand: lane half h supplies chunks (4s + h) and (4s + 2 + h) of 64-byte contraction step s — the same (lane, byte) -> k map for A
// and B, which is all the dot product needs.  64 cycles per instruction and SIMD for 2 x 32 x 32 x 64 flop: twice the bf16 rate.
typedef int v8i_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16_t mfma32_f8(const s16x8_t& a_lo, const s16x8_t& a_hi, const s16x8_t& b_lo, const s16x8_t& b_hi, f32x16_t c) {
  const v8i_t a = __builtin_bit_cast(v8i_t, __builtin_shufflevector(a_lo, a_hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15));
  const v8i_t b = __builtin_bit_cast(v8i_t, __builtin_shufflevector(b_lo, b_hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15));
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

__device__ __forceinline__ void unpack8f(uint4 u, float* f) {
  f[0] = bf2f(u.x & 0xffff); f[1] = bf2f(u.x >> 16); f[2] = bf2f(u.y & 0xffff); f[3] = bf2f(u.y >> 16);
  f[4] = bf2f(u.z & 0xffff); f[5] = bf2f(u.z >> 16); f[6] = bf2f(u.w & 0xffff); f[7] = bf2f(u.w >> 16);
}
__device__ __forceinline__ uint4 pack8f(const float* f) {
  uint4 o;
  o.x = pack2bf(f[0], f[1]); o.y = pack2bf(f[2], f[3]); o.z = pack2bf(f[4], f[5]); o.w = pack2bf(f[6], f[7]);
  return o;
}

// GR: two problems in one launch (aitk_gemm_nt_grouped): same N, K, K2 and epilogue flags, different operands / M.  The tile list
// is problem 0's tiles followed by problem 1's; a workgroup switches operand bases (buffer descriptors, row clamp, slab pointers)
// when the tile it is STAGING changes problem, and epilogue arguments when the tile it is COMPUTING does.
static_assert(sizeof(AitkGemmArgs) % 8 == 0, "two AitkGemmArgs must be contiguous in the kernarg segment");
// CV: implicit-GEMM 3x3 convolution (AitkGemmArgs.conv_mode, 2-D form): A is the NHWC image [B, H, W, Cin], row m = output pixel (b, oy, ox),
// K runs over (tap, cin) = 9 Cin with Cin % 64 == 0, so a K-tile lies inside ONE tap: the tap's pixel shift ((ky W + kx) Cin elements) and
// the channel offset are wave-uniform and ride in the DMA's SGPR offset, exactly where the plain GEMM puts its K offset; what differs
// per lane is only whether the tap's pixel exists — a 9-bit mask per staged row, built once per output tile — and a lane whose pixel lies
// outside the image points its offset past the buffer window and receives zeros (the same mechanism as K tails and dead tiles).
// The descriptor base is shifted one image row + one pixel down so that the base offset of a border pixel is never negative.
// FE: the fast epilogue forms (see the epilogue).  TRACE: s_memtime at the tile-switch points of tiles 1-3 of workgroups 0 and 100, waves 0 and 4
// (aitk_probe_gemm8_trace).
__device__ unsigned g_gemm8_trace[2 * 2 * 3 * 5];
// ET: AITK_EPI_EMIT_T instantiations (own kernels: the product kernels' code is the text without the `if constexpr (EMT)` blocks).  The BIAS | GELU fast
// epilogue then also contracts the bf16 GELU values it stores against the consumer layer's lora_down rows: per 16-row group and 32-column block the
// lane's eight packed values are moved to the v_mfma_f32_16x16x32_bf16 operand layout (row = lane & 15, column chunk = lane >> 4: a lane
// permutation through ds_bpermute) and two matrix instructions (hi and lo shadow) add into 8 x 4 accumulator registers = the wave's 128 rows x 16
// ranks over its 64 columns; the four waves of a row half then sum their slabs through the (idle) epilogue patches in a fixed order and write ONE
// [128 rows][16] fp32 slab per workgroup row half and 256-column tile.  Needs whole tiles (every wave takes part in the workgroup barriers).
// SK: stream-K tail (own instantiations, OPT-IN: AITK_GEMM8_SK=1 picks them when the last tile round would leave a large part of the chip idle — small
// batches: 216 tiles on 256 CUs at B = 1 —, =2 whenever the contract allows).  Measured and NOT the default (profiles/r06_gemm8_stream_k.md): the K loop
// of these shapes runs at the 1,400-W cap, and a round that fills 216 of 256 CUs is answered with an ~18 % higher clock on the busy ones (1.47 us per
// K-tile at 216 active CUs against 1.67 at 256), so the "idle" 16 % is not there to be won — the balanced schedule ties at best (-2 % at K >= 9216,
// B = 3, 4) and loses 5-25 % at B = 1, where the 256-KiB partial exchange is not amortised.  What it does:  The workgroup's full rounds stay data-parallel (tiles w, w + G, ...); the R = ntiles % G tiles that are left are dealt
// round the eight XCDs exactly as the data-parallel order deals them (virtual tile ndp + 8 j + x belongs to XCD x: xcd_remap gives the tiles of one XCD
// neighbouring positions, so that its L2 sees few A / B panels), and PER XCD they form an iteration space of cnt_x * nsteps K-tiles that is cut into G / 8
// equal ranges, one per workgroup of that XCD (workgroup w = 8 wl + x takes [sk_bound(wl), sk_bound(wl + 1))).  A range is at most: the END of
// one tile's K loop (the FINAL chunk: it holds the K tail and the LoRA slab, sums the other chunks' partial accumulators and runs the epilogue), whole
// tiles, and the START (or a middle piece) of another tile — a chunk that does not reach the tile's last K-tile leaves its fp32 accumulators in the
// workspace slot of its workgroup (one slot each: only the last segment of a range can be open) and raises the slot's flag; a chunk that does not START
// at K-tile 0 first adds the slot of the workgroup below, which therefore holds everything the tile has accumulated so far (a chain: one contributor
// per chunk).  Every workgroup computes its open chunk FIRST, so that by the time anybody's final chunk reaches its fix-up the partial it needs is
// (nearly always) there: the spin is a formality, and it cannot deadlock (a workgroup only ever waits for a lower-numbered one — w - 8, w - 16, ... —
// dispatched before it, and the chain ends at a chunk that starts a tile).  The order of the fp32 sums is fixed by the schedule:
// results are deterministic, and differ from the data-parallel kernel's only by the order of that fp32 sum.  A boundary one K-tile away from a tile's
// end is snapped onto it so that no chunk is shorter than TWO K-tiles — all the K loop's software pipeline needs (K-tiles t + 1, t + 2 are staged during
// K-tile t): it runs unchanged inside [kb, ke), and "the next tile's first K-tiles" become K-tiles kbn, kbn + 1 of the next item; a chunk may begin or
// end anywhere, also inside the LoRA slab.
struct AitkSkArgs { float* ws; int* flags; };
__host__ __device__ __forceinline__ int sk_bound(int w, int G, int I, int ns) {
  const int b = (int)((unsigned)w * (unsigned)I / (unsigned)G);  // launcher: (G + 1) * I < 2^31
  const int kk = b % ns;
  return kk == 1 ? b - 1 : (kk == ns - 1 ? b + 1 : b);
}
// item s (s = 0, 1, ...) of the stream-K range [lo, hi): tile (relative to the first stream-K tile) and [kb, ke); false past the end.  Natural order = first
// tile's end, whole tiles, last tile's start; the last segment goes FIRST when it is open.
__host__ __device__ __forceinline__ bool sk_segment(int lo, int hi, int ns, int s, int& tile, int& kb, int& ke) {
  if (lo >= hi) return false;
  const int t0 = lo / ns, tl = (hi - 1) / ns;
  const int n = tl - t0 + 1;
  if (s >= n) return false;
  const bool open = hi - tl * ns < ns;
  const int idx = open ? (s == 0 ? n - 1 : s - 1) : s;
  tile = t0 + idx;
  kb = idx == 0 ? lo - t0 * ns : 0;
  ke = idx == n - 1 ? hi - tl * ns : ns;
  return true;
}
// THE schedule (kernel and host probe): item i of workgroup w of G (G % 8 == 0) -> virtual tile v, K-tiles [kb, ke); false past the workgroup's last item.
__host__ __device__ __forceinline__ bool sk_item_of(int G, int w, int ntiles, int ns, int i, int& v, int& kb, int& ke) {
  const int ndp = ntiles - ntiles % G, rounds = ndp / G;
  if (i < rounds) {
    v = w + i * G;
    kb = 0;
    ke = ns;
    return true;
  }
  const int x = w & 7, wl = w >> 3, Gl = G >> 3;
  const int cnt = (ntiles - ndp - x + 7) >> 3;  // tail tiles of this XCD
  int tile = 0;
  if (!sk_segment(sk_bound(wl, Gl, cnt * ns, ns), sk_bound(wl + 1, Gl, cnt * ns, ns), ns, i - rounds, tile, kb, ke)) return false;
  v = ndp + 8 * tile + x;
  return true;
}
// the workgroup whose slot holds what the tile of a chunk with kb > 0 has accumulated so far: the nearest one below on the same XCD with a non-empty range
__host__ __device__ __forceinline__ int sk_predecessor(int G, int w, int ntiles, int ns) {
  const int x = w & 7, Gl = G >> 3;
  const int cnt = (ntiles - (ntiles - ntiles % G) - x + 7) >> 3;
  int wl = (w >> 3) - 1;
  while (wl > 0 && sk_bound(wl, Gl, cnt * ns, ns) >= sk_bound(wl + 1, Gl, cnt * ns, ns)) --wl;
  return 8 * wl + x;
}
template <bool GR, bool F8, bool CV = false, bool FE = true, bool TRACE = false, int ET = 0, bool SK = false>  // ET = 16-rank blocks of the emitted product (0: no emission)
__device__ __forceinline__ void gemm8_body(const AitkGemmArgs& p, const AitkGemmArgs& p2) {
  static_assert(ET == 0 || (FE && !F8 && !CV && ET <= 2), "EMIT_T: bf16 fast-epilogue kernels only, rank 16 or 32");
  static_assert(!SK || (FE && !F8 && !CV && !TRACE), "stream-K tail: bf16 fast-epilogue kernels");
  static_assert(!(CV && (GR || F8)), "convolution mode: single bf16 problem");
  constexpr int KB = F8 ? 128 : BK;  // base-segment elements per K-tile (128 B per LDS row either way; the LoRA slab stays bf16, 64 wide)
  constexpr int CH = F8 ? 16 : 8;    // base-segment elements per 16-B chunk
  constexpr int ESH = F8 ? 0 : 1;    // log2(bytes per base-segment element)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int l31 = lane & 31, h = lane >> 5;
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int nt1 = tiles_m * tiles_n;
  const int tiles_m2 = GR ? (p2.M + BM - 1) / BM : 0;
  const int ntiles = nt1 + tiles_m2 * tiles_n;
  const int nk1 = (p.K + KB - 1) / KB;
  const int nk2 = p.K2 > 0 ? (p.K2 + BK - 1) / BK : 0;
  const int nsteps = nk1 + nk2;   // >= 2 (launcher)
  const int nfast = p.K / KB;     // K-tiles [0, nfast) are full tiles of the base segment
  // SK: item i of this workgroup: virtual tile v, K-tiles [kb, ke) (launcher: gridDim.x % 8 == 0, nsteps >= 8, every XCD's share >= 8 K-tiles per workgroup).
  // Recomputed at item switches and in the fix-up rather than kept in SGPRs across the K loop.
  auto sk_item = [&](int i, int& v, int& kb, int& ke) -> bool {
    int nt_ = ntiles;
    asm volatile("" : "+s"(nt_));
    return sk_item_of((int)gridDim.x, (int)blockIdx.x, nt_, nsteps, i, v, kb, ke);
  };

  // ---- staging geometry: thread owns physical 16-B chunk pc of LDS rows srow + 64 i (i = 0..3) of A and of B
  // (recomputed from an opaque copy of tid at each use: these are needed once per output tile, and values the compiler
  //  would otherwise hoist out of the tile loop cost VGPRs the K loop does not have)
  // (SK: rebuilt from the wave index — a scalar — and mbcnt, so that no copy of tid has to stay in a VGPR, or in scratch, across the K loop)
  const int wave_s = SK ? __builtin_amdgcn_readfirstlane(wave) : 0;
  auto opaque_tid = [&]() {
    int t_ = SK ? (wave_s << 6) | (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) : tid;
    asm volatile("" : "+v"(t_));
    return t_;
  };
#define SROW(t_) ((t_) >> 3)
#define CC(t_) (((t_) & 7) ^ ((SROW(t_) >> 1) & 7))  /* logical chunk at physical slot tid&7 (same key for rows srow + 64 i) */
  // kernel arguments re-read through an opaque pointer where they are needed rarely (K tails, epilogue): keeping all ~45
  // dwords of AitkGemmArgs live in SGPRs across the K loop spills
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const __attribute__((address_space(4))) AitkGemmArgs* KArgsPtr;
#else
  typedef const AitkGemmArgs* KArgsPtr;
#endif
  auto kargs = [&](int prob = 0) -> KArgsPtr {
#if defined(__HIP_DEVICE_COMPILE__)
    KArgsPtr q = (KArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();  // constant address space: scalar loads
    asm volatile("" : "+s"(q));
    return GR ? q + prob : q;
#else
    return prob ? &p2 : &p;
#endif
  };
  unsigned va[4], vb[4];  // byte offsets of this thread's chunk (K-tile 0) in A / B for the tile being STAGED
  unsigned va2[4], vb2[4];  // ... and in the LoRA slab operands A2 / B2 (K-tile 0 of the slab)
  unsigned cm01 = 0, cm23 = 0;  // CV: tap-validity masks of staged rows i = 0, 1 (bits 0-8, 9-17) and i = 2, 3
  // CV: wave-uniform constants of the tap walk (SGPRs for the whole kernel: scalar loads inside the K loop would share lgkmcnt with the
  // hand-counted fragment reads)
  const int cv_kpt = CV ? __builtin_amdgcn_readfirstlane(p.conv_Cin >> 6) : 1;                 // K-tiles per tap
  const int cv_magic = CV ? __builtin_amdgcn_readfirstlane((65536 + cv_kpt - 1) / cv_kpt) : 0;  // kt / cv_kpt = (kt * magic) >> 16 for kt <= 9 * 64
  const int cv_w = CV ? __builtin_amdgcn_readfirstlane(p.conv_W) : 0;
  const int cv_cin2 = CV ? __builtin_amdgcn_readfirstlane(p.conv_Cin * 2) : 0;                  // bytes per pixel
  int om0 = 0, on0 = 0;   // origin of that tile (the K-tail / LoRA-slab path recomputes its row offsets from it)
  int m0 = 0, n0 = 0;
  int sM = p.M;   // row count of the problem being STAGED (row clamp of the A operand)
  int sprob = 0;  // ... and its index (slab pointers)
  int cprob = 0;  // problem of the tile being COMPUTED (epilogue arguments, fp8 operand scales)
  auto tile_origin = [&](int v, int& tm0, int& tn0, int& prob) {
    int lid = xcd_remap(v, ntiles);
    int tmx = tiles_m;
    prob = 0;
    if (GR && lid >= nt1) {
      lid -= nt1;
      tmx = tiles_m2;
      prob = 1;
    }
    const int GROUP = 8;
    const int group_sz = GROUP * tiles_n;
    const int gid = lid / group_sz;
    const int first_m = gid * GROUP;
    const int gm = min(tmx - first_m, GROUP);
    tm0 = (first_m + (lid % group_sz) % gm) * BM;
    tn0 = ((lid % group_sz) / gm) * BN;
  };
  auto a_row = [&](int i, int srow) { return min(om0 + (i & 1) * 128 + (i >> 1) * 64 + srow, sM - 1); };
  auto b_row = [&](int i, int srow) { return min(on0 + ((srow >> 5) + 2 * (i & 1)) * 64 + (i >> 1) * 32 + (srow & 31), p.N - 1); };
  typedef int v4i __attribute__((ext_vector_type(4)));
  auto make_srd = [&](const void* ptr) {
    const unsigned long long a = (unsigned long long)ptr;
    v4i r;
    r.x = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
    r.y = __builtin_amdgcn_readfirstlane((int)(a >> 32));
    r.z = (int)0x80000000u;
    r.w = 0x00020000;
    return r;
  };
  v4i srdA = make_srd(CV ? (const void*)(p.A - (long)(p.conv_W + 1) * p.conv_Cin) : (const void*)p.A), srdB = make_srd(p.B);
  v4i srdA2 = make_srd(p.A2), srdB2 = make_srd(p.B2);
  // lanes whose 16-B chunk of the LAST slab K-tile lies beyond K2 (K2 = 48 for rank 16: chunks 6, 7): OR-ed into the lane offset, which then points
  // outside the buffer window and reads zeros
  constexpr bool SLAB_PRE = !F8 && !SK;  // the W8A8 instantiation (and the stream-K one: three more scalars live across the K loop) has no registers left for the eight precomputed offsets (it would spill 50 more)
  unsigned slab_tail_oob = 0;
  if (SLAB_PRE && nk2 > 0) {
    const int t_ = opaque_tid();
    slab_tail_oob = ((nk2 - 1) * BK + CC(t_) * 8 < p.K2) ? 0u : 0x80000000u;
  }
  auto set_offsets = [&](int tm0, int tn0, int prob) {
    om0 = tm0;
    on0 = tn0;
    KArgsPtr q = kargs(prob);
    if (GR) {
      sprob = prob;
      sM = q->M;
      srdA = make_srd(q->A);
      srdB = make_srd(q->B);
      if constexpr (SLAB_PRE) {
        srdA2 = make_srd(q->A2);
        srdB2 = make_srd(q->B2);
      }
    }
    const int t_ = opaque_tid();
    const int srow = SROW(t_), cc = CC(t_);
    if (SLAB_PRE && nk2 > 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        va2[i] = ((unsigned)((long)a_row(i, srow) * q->lda2) + cc * 8) * 2;
        vb2[i] = ((unsigned)((long)b_row(i, srow) * q->ldb2) + cc * 8) * 2;
      }
    }
    if constexpr (CV) {
      const int HoWo = q->conv_HoWo, Wo = q->conv_Wo, cst = q->conv_stride, H = q->conv_H, W = q->conv_W, Cin = q->conv_Cin;
      cm01 = cm23 = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = a_row(i, srow);
        const int b = m / HoWo, rem = m - b * HoWo;
        const int oy = rem / Wo, ox = rem - oy * Wo;
        const int iy0 = oy * cst - q->conv_pad_t, ix0 = ox * cst - q->conv_pad_l;  // input pixel of tap (0, 0): >= -1
        va[i] = (unsigned)(((((long)b * H + iy0 + 1) * W + ix0 + 1) * Cin + cc * 8) * 2);  // relative to the shifted base: never negative
        unsigned mk = 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int ky = tap / 3, kx = tap - 3 * ky;
          if ((unsigned)(iy0 + ky) < (unsigned)H && (unsigned)(ix0 + kx) < (unsigned)W) mk |= 1u << tap;
        }
        if (i < 2) cm01 |= mk << (9 * i);
        else cm23 |= mk << (9 * (i - 2));
        vb[i] = ((unsigned)((long)b_row(i, srow) * q->ldb) + cc * 8) * 2;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        va[i] = ((unsigned)seg_off(q->lda, q->a_seg_rows, q->a_seg_stride, a_row(i, srow)) + cc * CH) << ESH;
        vb[i] = ((unsigned)((long)b_row(i, srow) * q->ldb) + cc * CH) << ESH;
      }
    }
  };
  // CV: K-tile kt of the base segment -> (tap, byte offset of the tap's pixel shift + channel offset); all scalar
  auto conv_tap = [&](int kt, unsigned& soff) -> int {
    const int tap = (kt * cv_magic) >> 16;
    const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;  // tap / 3 for tap <= 8
    soff = (unsigned)((ky * cv_w + kx) * cv_cin2 + (kt - tap * cv_kpt) * (BK * 2));
    return tap;
  };
  auto conv_voff = [&](int i, int tap) -> unsigned {
    const unsigned m = (i < 2 ? cm01 : cm23) >> (tap + 9 * (i & 1));
    return (m & 1u) ? va[i] : 0x80000000u;
  };

  // Raw buffer descriptors (stride 0, 2 GiB window): LDS-DMA through buffer_load ... lds takes base (SGPRs) + 32-bit lane
  // offset (VGPR) + K offset (SGPR), and a lane whose offset is outside the window reads ZEROS — K tails, the narrow LoRA
  // slab and dead tiles need no zero page and no per-lane 64-bit pointers.
  const unsigned OOB = 0x80000000u;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);
  auto dma = [&](unsigned ldst, unsigned voff, const v4i& srd, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(ldst), "v"(voff), "s"(srd), "s"(soff) : "memory");
  };

  // one 16-KiB half-tile (opnd 0 = A, 1 = B; half 0/1) of K-tile kt of the tile whose offsets are loaded, into tile
  // buffer `buf`; dead = no such tile (past the workgroup's last output tile).  Always exactly two DMAs per wave.
  auto stage_half = [&](int kt, int buf, int opnd, int half, bool dead) {
    const unsigned ldst = lds_wave + buf * BUF_BYTES + (opnd ? A_BYTES : 0) + half * (2 * NT * 16);
    if (!dead && kt < nfast) {  // full base-segment K-tile: no VALU at all
      unsigned soff = (unsigned)kt * (BK * 2);
      if (CV && !opnd) {  // image operand: the tap's shift in the scalar offset, the tap's validity per staged row
        const int tap = conv_tap(kt, soff);
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) dma(ldst + ii * (NT * 16), conv_voff(2 * half + ii, tap), srdA, soff);
        return;
      }
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = 2 * half + ii;
        dma(ldst + ii * (NT * 16), opnd ? vb[i] : va[i], opnd ? srdB : srdA, soff);
      }
      return;
    }
    if (SLAB_PRE && !dead && kt >= nk1) {  // LoRA-slab K-tile: offsets precomputed per output tile, no scalar loads (they would share lgkmcnt with the fragment reads)
      const int ks = kt - nk1;
      const unsigned tail = ks == nk2 - 1 ? slab_tail_oob : 0u;
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = 2 * half + ii;
        dma(ldst + ii * (NT * 16), (opnd ? vb2[i] : va2[i]) | tail, opnd ? srdB2 : srdA2, (unsigned)ks * (BK * 2));
      }
      return;
    }
    // K tail of the base segment / dead tile (and, W8A8, the LoRA slab): lanes whose chunk lies beyond the segment read zeros
    const int t_ = opaque_tid();
    const int srow = SROW(t_), cc = CC(t_);
    const bool second = kt >= nk1;
    const int k0 = second ? (kt - nk1) * BK : kt * KB;
    const int Kseg = second ? p.K2 : p.K;
    const bool kvalid = !dead && (k0 + cc * (second ? 8 : CH)) < Kseg;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      const int i = 2 * half + ii;
      unsigned voff;
      KArgsPtr qs = kargs(sprob);  // slab operands of the problem being staged
      if (second) voff = opnd ? ((unsigned)((long)b_row(i, srow) * qs->ldb2) + cc * 8) * 2 : ((unsigned)((long)a_row(i, srow) * qs->lda2) + cc * 8) * 2;
      else voff = opnd ? vb[i] : va[i];
      if (!kvalid) voff = OOB;
      if (second) dma(ldst + ii * (NT * 16), voff, make_srd(opnd ? (const void*)qs->B2 : (const void*)qs->A2), (unsigned)k0 * 2);
      else dma(ldst + ii * (NT * 16), voff, opnd ? srdB : srdA, (unsigned)k0 << ESH);
    }
  };

  // ---- fragment read addresses (current tile buffer; bit 16 toggles per K-tile); other quadrant +16 KiB, B +32 KiB
  unsigned aaddr[4], baddr[4];  // (rows +32 keep the swizzle key: the second 32-row block is an immediate +4 KiB)
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int rowa = wr * 64 + l31;
    aaddr[ks] = lds0 + rowa * 128 + (((ks * 2 + h) ^ ((rowa >> 1) & 7)) << 4);
    const int row = wc * 32 + l31;
    baddr[ks] = lds0 + row * 128 + (((ks * 2 + h) ^ ((row >> 1) & 7)) << 4);
  }
  s16x8_t af[2][4], b0f[4], b1f[4];
  f32x16_t acc[4][2];

  auto read_frags = [&](auto qm_c, auto qn_c, s16x8_t (*bf)[4]) {  // ks-major: b[ks], a0[ks], a1[ks]
    constexpr int qm = decltype(qm_c)::value, qn = decltype(qn_c)::value;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if constexpr (qn >= 0) lds_read128<A_BYTES + (qn > 0 ? 128 * 128 : 0)>((*bf)[ks], baddr[ks]);
      if constexpr (qm >= 0) {
        lds_read128<(qm > 0 ? 128 * 128 : 0)>(af[0][ks], aaddr[ks]);
        lds_read128<(qm > 0 ? 128 * 128 : 0) + 32 * 128>(af[1][ks], aaddr[ks]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto wait_lgkm = [&](int n) {
    switch (n) {
      case 0: asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); break;
      default: asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory"); break;
    }
  };
  // per_ks = fragment reads this phase issued per ks group (3: b+a0+a1, 2: a0+a1, 1: b, 0: none)
  // f8: this K-tile belongs to the e4m3 base segment (F8 kernels; the LoRA-slab tiles that follow it are bf16)
  auto mma_quadrant = [&](int qm, int qn, const s16x8_t (&bf)[4], int per_ks, bool f8 = true) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if (F8 && f8) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        __builtin_amdgcn_sched_barrier(0);  // keeps the wait of step 1 behind step 0's matrix instructions
        if (per_ks > 0) wait_lgkm(per_ks * (2 - 2 * s2));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
          acc[qm * 2 + mm][qn] = mfma32_f8(bf[2 * s2], bf[2 * s2 + 1], af[mm][2 * s2], af[mm][2 * s2 + 1], acc[qm * 2 + mm][qn]);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (per_ks > 0) wait_lgkm(per_ks * (3 - ks));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) acc[qm * 2 + mm][qn] = mfma32(bf[ks], af[mm][ks], acc[qm * 2 + mm][qn]);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  // F8: the e4m3 products carry the quantisation scales of their operands: acc[m][n] *= a_scale[m] * b_scale[n] once the base segment
  // is complete and before the (unscaled, bf16) LoRA slab accumulates on top.  Lane owns row l31 of each 32-row block, register r
  // column (r & 3) + 8 (r >> 2) + 4 h.  The scales of a tile travel through the wave's private epilogue patch in LDS (idle during the K loop):
  // fetch_scales() issues the global loads where a vmcnt(0) follows anyway (kernel entry; end of the previous tile's epilogue),
  // park_scales() stores them to the patch behind that wait, scale_acc() reads them back with LDS latency — the first version loaded
  // them from global right here and paid one exposed L2 / HBM round trip per output tile (fixed cost 19.9 vs 9.7 us per tile round of the
  // bf16 kernel, profiles/r03_gemm_ksweep.log).
  f32x4_t sc_rows = {1.f, 1.f, 1.f, 1.f}, sc_cols = {1.f, 1.f, 1.f, 1.f};
  auto fetch_scales = [&](int tm0, int tn0, int prob) {
    KArgsPtr q = kargs(prob);
    const float* as = q->a_scale;
    const float* bs = q->b_scale;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    sc_rows = f32x4_t{1.f, 1.f, 1.f, 1.f};
    sc_cols = f32x4_t{1.f, 1.f, 1.f, 1.f};
    if (as && ln < 32) {  // rows tm0 + wr * 128 + 4 ln .. + 3 (a_scale is padded to a multiple of 4 by the contract: M rows rounded up are readable? no: clamp)
      const int r0 = tm0 + wr * 128 + 4 * ln;
#pragma unroll
      for (int e = 0; e < 4; ++e) sc_rows[e] = as[min(r0 + e, q->M - 1)];
    }
    if (bs && ln < 16) {
      const int c0 = tn0 + wc * 64 + 4 * ln;
      if (c0 < q->N) sc_cols = *reinterpret_cast<const f32x4_t*>(bs + c0);
    }
  };
  auto park_scales = [&]() {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    char* patch = smem + EPI_OFF + (tid >> 6) * 4096;
    if (ln < 32) *reinterpret_cast<f32x4_t*>(patch + ln * 16) = sc_rows;          // rows 4 ln .. 4 ln + 3 of the wave's 128
    if (ln < 16) *reinterpret_cast<f32x4_t*>(patch + 512 + ln * 16) = sc_cols;   // columns 4 ln .. of the wave's 64
  };
  auto scale_acc = [&]() {
    const char* patch = smem + EPI_OFF + (tid >> 6) * 4096;
    float rs[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) rs[mi] = *reinterpret_cast<const float*>(patch + (mi * 32 + l31) * 4);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4_t cs = *reinterpret_cast<const f32x4_t*>(patch + 512 + (ni * 8 + 2 * g + h) * 16);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[mi][ni][4 * g + e] *= rs[mi] * cs[e];
      }
  };
#define VMCNT8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
#define BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

  // ---- first output tile: offsets + the six prologue half-tiles (K-tile 0: A0 B0 B1 A1, K-tile 1: A0 B0)
  int vt = blockIdx.x;
  int item = 0, kb = 0, ke = nsteps;  // SK: the item being computed and its K-tile range (else: whole tiles)
  if constexpr (SK) {
    if (!sk_item(0, vt, kb, ke)) return;  // nothing for this workgroup (wave-uniform, before any barrier)
  }
  tile_origin(vt, m0, n0, cprob);
  set_offsets(m0, n0, cprob);
  if constexpr (F8) fetch_scales(m0, n0, cprob);  // ahead of the prologue DMAs: the first VMCNT8 covers them
  int gk = -kb;  // K-tile counter across output tiles: K-tile t of this output tile lives in buffer (gk + t) & 1
  stage_half(kb, 0, 0, 0, false);
  stage_half(kb, 0, 1, 0, false);
  stage_half(kb, 0, 1, 1, false);
  stage_half(kb, 0, 0, 1, false);
  stage_half(kb + 1, 1, 0, 0, false);
  stage_half(kb + 1, 1, 1, 0, false);
  VMCNT8();  // first tile: K-tile 0 landed (the later tiles wait at the bottom of the loop)
  int tix = 0;   // output tiles this workgroup has finished
  unsigned tr[3][5] = {};
#define STAMP(k_)                                                                                                   \
  if constexpr (TRACE) {                                                                                            \
    const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime();                                                     \
    if (tix == 1) tr[0][k_] = t_;                                                                                   \
    else if (tix == 2) tr[1][k_] = t_;                                                                              \
    else if (tix == 3) tr[2][k_] = t_;                                                                              \
  }
  while (true) {
    int vnext = vt + gridDim.x, kbn = 0, ken = nsteps;
    bool has_next;
    if constexpr (SK) has_next = sk_item(item + 1, vnext, kbn, ken);
    else has_next = vnext < ntiles;
    const int k_end = SK ? ke : nsteps;  // this item's K-tiles are [kb, k_end); the K-tiles staged past it are kbn, kbn + 1 of the next item
    int m0n = 0, n0n = 0, probn = 0;
    if (has_next) tile_origin(vnext, m0n, n0n, probn);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    STAMP(0);
    if constexpr (F8) park_scales();  // this tile's operand scales (fetched before the wait above) into the wave's idle epilogue patch
    BAR();
    if (wr == 1) BAR();  // second wave of every SIMD runs one barrier behind
    STAMP(1);
    // Steady part: K-tiles t+1, t+2 are full base-segment tiles of THIS output tile — straight-line fast staging, no
    // branches.  Tail part (last <= 2 + LoRA-slab iterations): K tail, slab, and the next output tile's first K-tiles.
    auto stage_fast = [&](int kt, unsigned ldst_buf, int opnd, int half) {
      unsigned soff = (unsigned)kt * (BK * 2);
      const unsigned ldst = ldst_buf + (opnd ? A_BYTES : 0) + half * (2 * NT * 16);
      if (CV && !opnd) {
        const int tap = conv_tap(kt, soff);
        dma(ldst, conv_voff(2 * half, tap), srdA, soff);
        dma(ldst + NT * 16, conv_voff(2 * half + 1, tap), srdA, soff);
        return;
      }
      dma(ldst, opnd ? vb[2 * half] : va[2 * half], opnd ? srdB : srdA, soff);
      dma(ldst + NT * 16, opnd ? vb[2 * half + 1] : va[2 * half + 1], opnd ? srdB : srdA, soff);
    };
    auto flip = [&]() {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        aaddr[ks] ^= BUF_BYTES;
        baddr[ks] ^= BUF_BYTES;
      }
    };
    const int nsteady = min(nfast, k_end) - 2;  // iterations t < nsteady stage only fast tiles (t + 2 < nfast)
    int t = SK ? kb : 0;
    for (; t < nsteady; ++t) {
      const unsigned lb1 = lds_wave + ((gk + t + 1) & 1) * BUF_BYTES, lb2 = lds_wave + ((gk + t) & 1) * BUF_BYTES;
      read_frags(IC<0>{}, IC<0>{}, &b0f);
      stage_fast(t + 1, lb1, 1, 1);
      VMCNT8();
      BAR();
      mma_quadrant(0, 0, b0f, 3);
      BAR();
      read_frags(IC<-1>{}, IC<1>{}, &b1f);
      stage_fast(t + 1, lb1, 0, 1);
      VMCNT8();
      BAR();
      mma_quadrant(0, 1, b1f, 1);
      BAR();
      read_frags(IC<1>{}, IC<-1>{}, (s16x8_t(*)[4]) nullptr);
      stage_fast(t + 2, lb2, 0, 0);
      BAR();
      mma_quadrant(1, 1, b1f, 2);
      BAR();
      stage_fast(t + 2, lb2, 1, 0);
      VMCNT8();
      BAR();
      mma_quadrant(1, 0, b0f, 0);
      flip();
      BAR();
    }
    STAMP(2);
    bool switched = false;  // offsets already describe the NEXT output tile
    // one tail iteration; f8_c: this K-tile belongs to the e4m3 base segment (compile-time so each loop below carries ONE matrix form)
    auto tail_iter = [&](int t, auto f8_c) {
      constexpr bool f8t = decltype(f8_c)::value != 0;
      // K-tiles t+1 / t+2 past this output tile are K-tiles 0 / 1 of the next one (or dead)
      const int t1 = t + 1, t2 = t + 2;
      const bool n1 = t1 >= k_end, n2 = t2 >= k_end;
      const int k1 = n1 ? kbn + t1 - k_end : t1, k2 = n2 ? kbn + t2 - k_end : t2;
      const int buf1 = (gk + t1) & 1, buf2 = (gk + t2) & 1;
      // phase 0
      read_frags(IC<0>{}, IC<0>{}, &b0f);
      stage_half(k1, buf1, 1, 1, n1 && !has_next);
      VMCNT8();
      BAR();
      mma_quadrant(0, 0, b0f, 3, f8t);
      BAR();
      // phase 1
      read_frags(IC<-1>{}, IC<1>{}, &b1f);
      stage_half(k1, buf1, 0, 1, n1 && !has_next);
      VMCNT8();
      BAR();
      mma_quadrant(0, 1, b1f, 1, f8t);
      BAR();
      // phase 2
      read_frags(IC<1>{}, IC<-1>{}, (s16x8_t(*)[4]) nullptr);
      if (n2 && !switched) {  // every staging from here on belongs to the next output tile
        if (has_next) set_offsets(m0n, n0n, probn);
        switched = true;
      }
      stage_half(k2, buf2, 0, 0, n2 && !has_next);
      BAR();
      mma_quadrant(1, 1, b1f, 2, f8t);
      BAR();
      // phase 3
      stage_half(k2, buf2, 1, 0, n2 && !has_next);
      VMCNT8();
      BAR();
      mma_quadrant(1, 0, b0f, 0, f8t);
      flip();
      BAR();
    };
    if constexpr (F8) {
      const int nbase = min(nk1, nsteps);
      for (; t < nbase; ++t) tail_iter(t, IC<1>{});
      if (nk2 > 0) scale_acc();  // base segment complete: the bf16 slab accumulates on top of the scaled products
      for (; t < nsteps; ++t) tail_iter(t, IC<0>{});
    } else {
      for (; t < k_end; ++t) tail_iter(t, IC<0>{});
    }
    if (wr == 0) BAR();  // pair the extra barrier of the second group
    gk += k_end - kbn;
    if (F8 && nk2 == 0) scale_acc();

    // ---------------- epilogue (tile m0, n0; wave block rows wr*128.., cols wc*64..) ----------------
    // Two forms.  The GENERIC one takes any flag combination, ragged rows / columns and any row map; it tests the flags at run time, which the compiler
    // turns into compute-both-and-select code, re-derives every row's address (a 64-bit multiply, or a division under a segmented row map) and re-reads
    // kernel arguments per row group: ~60 vector instructions per row group for a bias-only tile.  The s_memtime trace of round 4
    // (profiles/r04_gemm8_tile_switch.md) shows what that costs: 12,300 cycles per bias-only tile and 57,000 per gate-residual tile next to 2,456 per
    // K-tile, while the same stores replayed without the arithmetic (tools/probes/epilogue_store.hip) take 3,700 — the epilogue is bound by vector issue
    // and, where it reads, by sixteen SERIAL load -> use -> store round trips per wave (gfx9 retires vmcnt in order: waiting for a load issued behind a
    // store also waits for that store's acknowledgement).  The FAST form (FE) is one straight-line instantiation per flag set the graphs actually use, for
    // waves whose 128 x 64 block lies inside the matrix: flags are compile-time, a row group's address is a wave-uniform 64-bit base (walked incrementally
    // through the segmented row map) + one per-lane 32-bit offset, bf16 round trips reuse the packed words that are stored anyway, gate columns are loaded
    // once per sample instead of once per row group, and the operands a block READS (residual / pre-activation / old C) are requested two blocks ahead, in
    // front of the stores of the blocks in between.  Same arithmetic, same order: bit-identical to the generic form.
    STAMP(3);
    bool epi_done = false;
    if constexpr (SK) {
#if defined(__HIP_DEVICE_COMPILE__)
      // the workspace / flag pointers ride behind the problem descriptors in the kernarg segment; read where they are needed (scalar loads)
      auto skargs = [&]() {
        const __attribute__((address_space(4))) char* q = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(q));
        return (const __attribute__((address_space(4))) AitkSkArgs*)(q + (GR ? 2 : 1) * sizeof(AitkGemmArgs));
      };
      // A slot is [wave][64 register pairs][lane] fp32x2: wave w owns 32 KiB, one instruction moves 512 B contiguous.  Partials and flags travel as relaxed
      // agent-scope atomics (8-byte ones for the data: global_load / store_dwordx2 sc1 — the cache policy that is coherent across the eight XCDs' L2s)
      // instead of behind release / acquire fences: a fence here is buffer_wbl2 / buffer_inv over the whole L2, once per wave.
      const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
      int ln = lane;
      asm volatile("" : "+v"(ln));
      typedef unsigned long long u64_t;
      if (kb > 0) {
        // this chunk continues a tile: the workgroup below (the nearest one with a non-empty range) holds everything the tile has accumulated before K-tile
        // kb — its own chunk plus, if that was a continuation too, what IT was handed: one contributor, straight-line code (a loop over contributors
        // carries 128 accumulators round a back edge and the allocator answers with scratch)
        int nt_ = ntiles;
        asm volatile("" : "+s"(nt_));
        const int w2 = sk_predecessor((int)gridDim.x, (int)blockIdx.x, nt_, nsteps);
        if (tid == 0) {
          int* fl = skargs()->flags + w2;
          while (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(2);
          __hip_atomic_store(fl, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // consumed: the slot's next writer is a later launch
        }
        __syncthreads();
        const u64_t* wsp = reinterpret_cast<const u64_t*>(skargs()->ws + (long)w2 * (BM * BN) + wv * 8192) + ln;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r2 = 0; r2 < 8; ++r2) {
              const f32x2_t pv = __builtin_bit_cast(f32x2_t, __hip_atomic_load(wsp + ((mi * 2 + ni) * 8 + r2) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
              acc[mi][ni][2 * r2] += pv.x;
              acc[mi][ni][2 * r2 + 1] += pv.y;
            }
        // the staging offsets of the NEXT tile (set when the K loop switched) are a pure function of its origin: recomputed here, they are not live across
        // the fix-up (the allocator otherwise spills them for the whole tail of the K loop, and a reload in front of a DMA drains the pipeline)
        if (has_next) set_offsets(m0n, n0n, probn);
      }
      if (k_end < nsteps) {
        // open chunk: accumulators -> this workgroup's slot, then the flag
        u64_t* wsp = reinterpret_cast<u64_t*>(skargs()->ws + (long)blockIdx.x * (BM * BN) + wv * 8192) + ln;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r2 = 0; r2 < 8; ++r2)
              __hip_atomic_store(wsp + ((mi * 2 + ni) * 8 + r2) * 64, __builtin_bit_cast(u64_t, f32x2_t{acc[mi][ni][2 * r2], acc[mi][ni][2 * r2 + 1]}), __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(skargs()->flags + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        epi_done = true;
      }
#endif
    }
    if constexpr (FE) {
     if (!epi_done) {
      KArgsPtr q = kargs(cprob);
      const int flags = q->flags;
      const int mw = __builtin_amdgcn_readfirstlane(m0 + wr * 128), nw = __builtin_amdgcn_readfirstlane(n0 + wc * 64);
      const int csr = q->c_seg_rows, grows = q->gate_rows;
      bool ok = mw + 128 <= q->M && nw + 64 <= q->N && (csr == 0 || (csr >= 128 && (csr & 15) == 0));
      if (flags & AITK_EPI_GATE_RES) ok = ok && grows >= 128 && (grows & 15) == 0;
      auto fast_epi = [&](auto fl_c) {
        constexpr int FL = decltype(fl_c)::value;
        constexpr bool BIAS = (FL & AITK_EPI_BIAS) != 0, GELU = (FL & AITK_EPI_GELU) != 0, DGELU = (FL & AITK_EPI_DGELU) != 0;
        constexpr bool GATE = (FL & AITK_EPI_GATE_RES) != 0, ADDA = (FL & AITK_EPI_ADD_AUX) != 0, ACC = (FL & AITK_EPI_ACCUM) != 0;
        constexpr bool RD_IN = DGELU || GATE || ADDA;  // reads aux_in
        constexpr bool EMT = ET > 0 && (FL & AITK_EPI_EMIT_T) != 0;
        constexpr int ERB = ET > 0 ? ET : 1;  // 16-rank blocks of the emitted product
        static_assert(!EMT || (GELU && !RD_IN && !ACC), "EMIT_T rides on the BIAS | GELU form");
        int ln = lane;
        asm volatile("" : "+v"(ln));
        // EMT: the consumer's lora_down rows over this wave's 64 columns, in the A-operand layout (rank = lane & 15, 8-column chunk = lane >> 4), hi and lo
        s16x8_t awh[2][ERB], awl[2][ERB];
        f32x4_t tacc[8][ERB];
        int bperm_src = 0;
        const int m_rows = EMT ? q->M : 0;
        if constexpr (EMT) {
          const long ldp = q->t_ldp;
          const long aoff = (long)(ln & 15) * ldp + (n0 + wc * 64) + (ln >> 4) * 8;
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int rb = 0; rb < ERB; ++rb) {
              awh[ni][rb] = *reinterpret_cast<const s16x8_t*>(q->t_p + aoff + rb * 16 * ldp + ni * 32);
              awl[ni][rb] = *reinterpret_cast<const s16x8_t*>(q->t_p_lo + aoff + rb * 16 * ldp + ni * 32);
            }
#pragma unroll
          for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int rb = 0; rb < ERB; ++rb) tacc[g][rb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
          bperm_src = ((((ln & 15) << 2) | (ln >> 4)) << 2);  // ds_bpermute byte address: this lane takes the words of lane (row << 2 | chunk)
        }
        char* patch = smem + EPI_OFF + (tid >> 6) * 4096;
        const int r_w = ln & 31, hh = ln >> 5, rr = ln >> 2, c4 = ln & 3;
        const long ldc = q->ldc;
        const long css = q->c_seg_stride;
        const unsigned lofC = (unsigned)(rr * (int)ldc + c4 * 8) * 2u;  // byte offset of the lane's 16 B inside a 16-row group (< 2^31: 16 rows)
        unsigned lofI = 0, lofO = 0;
        const long ldi = RD_IN ? q->ld_aux_in : 0, ldo = (GELU || GATE) ? q->ld_aux_out : 0;
        if constexpr (RD_IN) lofI = (unsigned)(rr * (int)ldi + c4 * 8) * 2u;
        if constexpr (GELU || GATE) lofO = (unsigned)(rr * (int)ldo + c4 * 8) * 2u;
        const char* Cb = reinterpret_cast<const char*>(q->C) + (long)nw * 2;
        const char* Ib = RD_IN ? reinterpret_cast<const char*>(q->aux_in) + (long)nw * 2 : nullptr;
        char* Ob = (GELU || GATE) ? const_cast<char*>(reinterpret_cast<const char*>(q->aux_out)) : nullptr;
        const bool save_y = Ob != nullptr;  // gate-residual: y is only saved when asked
        if (Ob) Ob += (long)nw * 2;
        // wave-uniform walk through the segmented row map of C: group g = rows mw + 16 g .. + 15 sits in segment sg at row wg (at most one wrap per wave)
        int sg0 = 0, wg0 = mw;
        if (csr > 0) {
          sg0 = mw / csr;
          wg0 = mw - sg0 * csr;
        }
        auto c_group = [&](int g) -> char* {
          int sg = sg0, wg = wg0 + 16 * g;
          if (csr > 0 && wg >= csr) {
            wg -= csr;
            ++sg;
          }
          return const_cast<char*>(Cb) + ((long)sg * css + (long)wg * ldc) * 2;
        };
        // column operands: both halves requested now (packed), unpacked at the head of their column pass (ni outer: 8 + 4 live registers instead of 16)
        uint4 bpk[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
        if constexpr (BIAS) {
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) bpk[ni] = *reinterpret_cast<const uint4*>(q->bias + nw + ni * 32 + c4 * 8);
        }
        int gb0 = 0, gw0 = 0;
        const bf16_t* gate_p = GATE ? q->gate + nw + c4 * 8 : nullptr;
        const long ldg = GATE ? q->ld_gate : 0;
        if constexpr (GATE) {
          gb0 = mw / grows;
          gw0 = mw - gb0 * grows;
        }
        // operands a block reads, requested two blocks ahead: blk = 4 ni + mi, row groups 2 mi, 2 mi + 1
        uint4 pin[16], pacc[16];
        auto request = [&](int blk) {
          const int mi = blk & 3, ni = blk >> 2;
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int g = 2 * mi + it;
            if constexpr (RD_IN) pin[2 * blk + it] = *reinterpret_cast<const uint4*>(Ib + ((long)(mw + 16 * g) * ldi) * 2 + lofI + ni * 64);
            if constexpr (ACC) pacc[2 * blk + it] = *reinterpret_cast<const uint4*>(c_group(g) + lofC + ni * 64);
          }
        };
        constexpr int PFD = 2;  // blocks requested ahead (3 measured equal within noise and spills 8 registers: profiles/r04_gemm8_tile_switch.md)
        if constexpr (RD_IN || ACC) {
#pragma unroll
          for (int i = 0; i < PFD; ++i) request(i);
        }
        float b8[8], g8[8];
        int gb_loaded = -1;
#pragma unroll
        for (int blk = 0; blk < 8; ++blk) {
          const int mi = blk & 3, ni = blk >> 2;
          if (mi == 0) {
            if constexpr (BIAS) unpack8f(bpk[ni], b8);
            gb_loaded = -1;
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4_t v4 = {acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
            *reinterpret_cast<f32x4_t*>(patch + r_w * 128 + (((2 * g + hh) ^ (r_w & 7)) << 4)) = v4;
          }
          __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): same-wave LDS ops execute in order, no barrier needed
          if constexpr (RD_IN || ACC) {
            if (blk + PFD < 8) request(blk + PFD);  // ahead of this block's stores
          }
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int g = 2 * mi + it, r = it * 16 + rr;
            const bool row_in = !EMT || (mw + 16 * g + rr < m_rows);
            const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(patch + r * 128 + (((2 * c4) ^ (r & 7)) << 4));
            const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(patch + r * 128 + (((2 * c4 + 1) ^ (r & 7)) << 4));
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            char* crow = c_group(g) + lofC + ni * 64;
            if constexpr (BIAS) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += b8[e];
            }
            if constexpr (ADDA) {
              float a8[8];
              unpack8f(pin[2 * blk + it], a8);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += a8[e];
            }
            if constexpr (ACC) {
              float c8[8];
              unpack8f(pacc[2 * blk + it], c8);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += c8[e];
            }
            if constexpr (GELU) {
              // u = bf16(pre-activation) is saved for backward; h = gelu_tanh(u) on the rounded value: the packed words ARE the rounded values
              const uint4 w = pack8f(v);
              if (!EMT || row_in)  // EMT launches may end in a ragged row tile (every wave walks the whole epilogue: workgroup barriers): stores are per-row predicated
                *reinterpret_cast<uint4*>(Ob + ((long)(mw + 16 * g) * ldo) * 2 + lofO + ni * 64) = w;
              float u8[8];
              unpack8f(w, u8);
#pragma unroll
              for (int e = 0; e < 8; e += 2) {
                const f32x2_t h2 = gelu_tanh_2(f32x2_t{u8[e], u8[e + 1]});
                v[e] = h2.x;
                v[e + 1] = h2.y;
              }
            }
            if constexpr (DGELU) {
              float u8[8];
              unpack8f(pin[2 * blk + it], u8);
#pragma unroll
              for (int e = 0; e < 8; e += 2) {
                const f32x2_t g2 = f32x2_t{v[e], v[e + 1]} * gelu_tanh_grad_2(f32x2_t{u8[e], u8[e + 1]});
                v[e] = g2.x;
                v[e + 1] = g2.y;
              }
            }
            if constexpr (GATE) {
              // y = bf16(linear out) saved when asked (d_gate needs it); x_new = res + gate[sample] * y
              const uint4 w = pack8f(v);
              if (save_y) *reinterpret_cast<uint4*>(Ob + ((long)(mw + 16 * g) * ldo) * 2 + lofO + ni * 64) = w;
              int gb = gb0;
              if (gw0 + 16 * g >= grows) ++gb;
              if (gb != gb_loaded) {  // wave-uniform: the sample changes at most once inside a wave's 128 rows
                unpack8f(*reinterpret_cast<const uint4*>(gate_p + (long)gb * ldg + ni * 32), g8);
                gb_loaded = gb;
              }
              float y8[8], r8[8];
              unpack8f(w, y8);
              unpack8f(pin[2 * blk + it], r8);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = r8[e] + g8[e] * y8[e];
            }
            const uint4 cw = pack8f(v);
            if (!EMT || row_in) *reinterpret_cast<uint4*>(crow) = cw;
            if constexpr (EMT) {
              v4i hb;
              hb.x = __builtin_amdgcn_ds_bpermute(bperm_src, (int)cw.x);
              hb.y = __builtin_amdgcn_ds_bpermute(bperm_src, (int)cw.y);
              hb.z = __builtin_amdgcn_ds_bpermute(bperm_src, (int)cw.z);
              hb.w = __builtin_amdgcn_ds_bpermute(bperm_src, (int)cw.w);
              const s16x8_t hf = __builtin_bit_cast(s16x8_t, hb);
#pragma unroll
              for (int rb = 0; rb < ERB; ++rb) {
                tacc[g][rb] = mfma16(awh[ni][rb], hf, tacc[g][rb]);  // D[rank 16 rb + 4 (lane >> 4) + r][row lane & 15]
                tacc[g][rb] = mfma16(awl[ni][rb], hf, tacc[g][rb]);
              }
            }
          }
          __builtin_amdgcn_s_waitcnt(0xc07f);  // patch reads retired before the next block overwrites it
        }
        if constexpr (EMT) {
          // the four column waves of this row half add their [128][16] slabs through the patches (4 KiB each: two halves of 64 rows), wave wc finishing
          // row group wc of each half; lane (row = lane & 15, ranks 4 (lane >> 4) ..) writes 16 B, the wave 1 KiB contiguous
          constexpr int RT = 16 * ERB;  // ranks per row of the slab
          float* tp = q->t_partial + ((long)(q->t_tile0 + n0 / BN) * m_rows + (m0 + wr * 128)) * RT;
#pragma unroll
          for (int pass = 0; pass < 2 * ERB; ++pass) {  // (row half, rank block): 4 KiB of every wave's patch per pass
            const int half = pass / ERB, rb = pass % ERB;
            if (pass > 0) BAR();  // the previous pass's reads are done in every wave before the patches are rewritten
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) *reinterpret_cast<f32x4_t*>(patch + gg * 1024 + ln * 16) = tacc[half * 4 + gg][rb];
            __builtin_amdgcn_s_waitcnt(0xc07f);
            BAR();
            f32x4_t sum4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              const f32x4_t t4 = *reinterpret_cast<const f32x4_t*>(smem + EPI_OFF + (wr * 4 + w) * 4096 + wc * 1024 + ln * 16);
              sum4[0] += t4[0]; sum4[1] += t4[1]; sum4[2] += t4[2]; sum4[3] += t4[3];
            }
            if (m0 + wr * 128 + (half * 4 + wc) * 16 + (ln & 15) < m_rows)
              *reinterpret_cast<f32x4_t*>(tp + (long)((half * 4 + wc) * 16 + (ln & 15)) * RT + rb * 16 + 4 * (ln >> 4)) = sum4;
            __builtin_amdgcn_s_waitcnt(0xc07f);  // (after the last pass the next writer of a patch is the next tile's epilogue, behind the K loop's barriers)
          }
        }
      };
      if constexpr (ET > 0) {
        // launcher contract: whole column tiles, no row maps, flags == BIAS | GELU | EMIT_T — every wave of the workgroup is here (the emission has workgroup
        // barriers), also the waves of a ragged last row tile: their rows beyond M compute on clamped operand rows and store nothing
        epi_done = true;
        fast_epi(IC<(AITK_EPI_BIAS | AITK_EPI_GELU | AITK_EPI_EMIT_T)>{});
      } else if (ok) {
        epi_done = true;
        switch (flags) {
          case 0: fast_epi(IC<0>{}); break;
          case AITK_EPI_BIAS: fast_epi(IC<AITK_EPI_BIAS>{}); break;
          case AITK_EPI_BIAS | AITK_EPI_GELU: fast_epi(IC<(AITK_EPI_BIAS | AITK_EPI_GELU)>{}); break;
          case AITK_EPI_DGELU: fast_epi(IC<AITK_EPI_DGELU>{}); break;
          case AITK_EPI_BIAS | AITK_EPI_GATE_RES: fast_epi(IC<(AITK_EPI_BIAS | AITK_EPI_GATE_RES)>{}); break;
          case AITK_EPI_BIAS | AITK_EPI_ADD_AUX: fast_epi(IC<(AITK_EPI_BIAS | AITK_EPI_ADD_AUX)>{}); break;
          case AITK_EPI_ACCUM: fast_epi(IC<AITK_EPI_ACCUM>{}); break;
          default: epi_done = false; break;
        }
      }
     }
    }
    if (!epi_done) {
      KArgsPtr q = kargs(cprob);
      const int flags = q->flags;
      int ln = lane;
      asm volatile("" : "+v"(ln));  // keep the epilogue's lane geometry out of the K loop's live set
      char* patch = smem + EPI_OFF + (tid >> 6) * 4096;
      const int r_w = ln & 31, h = ln >> 5;       // write side: MFMA layout, lane owns row l31, 4 columns per (g, h)
      const int rr = ln >> 2, c4 = ln & 3;        // read side: 16 rows x 4 lanes, lane owns 8 columns c4*8..
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int n = n0 + wc * 64 + ni * 32 + c4 * 8;
        const bool ncol = n < q->N;
        float bias8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
        if ((flags & AITK_EPI_BIAS) && ncol) unpack8f(*reinterpret_cast<const uint4*>(q->bias + n), bias8);
        float cs8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) cs8[e] = 1.f;
        if ((flags & AITK_EPI_COL_SCALE) && ncol) {
          const f32x4_t c0 = *reinterpret_cast<const f32x4_t*>(q->col_scale + n), c1 = *reinterpret_cast<const f32x4_t*>(q->col_scale + n + 4);
          cs8[0] = c0[0]; cs8[1] = c0[1]; cs8[2] = c0[2]; cs8[3] = c0[3]; cs8[4] = c1[0]; cs8[5] = c1[1]; cs8[6] = c1[2]; cs8[7] = c1[3];
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          // acc block -> wave-private patch [32 rows][32 fp32], 16-B chunk (2g+h) swizzled by row&7
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4_t v4 = {acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
            *reinterpret_cast<f32x4_t*>(patch + r_w * 128 + (((2 * g + h) ^ (r_w & 7)) << 4)) = v4;
          }
          __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): same-wave LDS ops execute in order, no barrier needed
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int r = it * 16 + rr;
            const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(patch + r * 128 + (((2 * c4) ^ (r & 7)) << 4));
            const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(patch + r * 128 + (((2 * c4 + 1) ^ (r & 7)) << 4));
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            const int m = m0 + wr * 128 + mi * 32 + r;
            if (m < q->M && ncol) {
              bf16_t* crow = const_cast<bf16_t*>(seg_row8(q->C, q->ldc, q->c_seg_rows, q->c_seg_stride, m)) + n;
              if (flags & AITK_EPI_COL_SCALE) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= cs8[e];
              }
              if (flags & AITK_EPI_BIAS) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bias8[e];
              }
              if (flags & AITK_EPI_BIAS_ROW) {
                const float br = bf2f(q->bias[m]);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += br;
              }
              if (flags & AITK_EPI_ADD_AUX) {
                float a8[8];
                unpack8f(*reinterpret_cast<const uint4*>(q->aux_in + (long)m * q->ld_aux_in + n), a8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += a8[e];
              }
              if (flags & AITK_EPI_ACCUM) {
                float c8[8];
                unpack8f(*reinterpret_cast<const uint4*>(crow), c8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += c8[e];
              }
              if (flags & AITK_EPI_GELU) {
                // u = bf16(pre-activation) is saved for backward; h = gelu_tanh(u) (torch evaluates GELU on the bf16 value)
                *reinterpret_cast<uint4*>(q->aux_out + (long)m * q->ld_aux_out + n) = pack8f(v);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gelu_tanh_f(bfround(v[e]));
              }
              if (flags & AITK_EPI_DGELU) {
                float u8[8];
                unpack8f(*reinterpret_cast<const uint4*>(q->aux_in + (long)m * q->ld_aux_in + n), u8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= gelu_tanh_grad_f(u8[e]);
              }
              if (flags & AITK_EPI_GATE_RES) {
                // y = bf16(linear out) saved when asked (d_gate needs it); x_new = res + gate[b] * y
                if (q->aux_out) *reinterpret_cast<uint4*>(q->aux_out + (long)m * q->ld_aux_out + n) = pack8f(v);
                float r8[8], g8[8];
                unpack8f(*reinterpret_cast<const uint4*>(q->aux_in + (long)m * q->ld_aux_in + n), r8);
                unpack8f(*reinterpret_cast<const uint4*>(q->gate + (long)(m / q->gate_rows) * q->ld_gate + n), g8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = r8[e] + g8[e] * bfround(v[e]);
              }
              *reinterpret_cast<uint4*>(crow) = pack8f(v);
            }
          }
          __builtin_amdgcn_s_waitcnt(0xc07f);  // patch reads retired before the next block overwrites it
        }
      }
    }
    STAMP(4);
    ++tix;
    if constexpr (F8) {
      if (has_next) fetch_scales(m0n, n0n, probn);  // the next tile's operand scales ride under the wait below
    }
    // vmcnt(0): the prefetched K-tiles of the next output tile + this wave's epilogue traffic.  The BUILTIN, and on EVERY path out of the epilogue (in front
    // of the exit test: the compiler folds the `break` into the loop latch, and a path that reaches the latch without the wait counts as a back edge), so that
    // the waitcnt pass sees it and clears its scoreboard: with an asm wait it keeps the epilogue's global loads "possibly pending" round the back edge and,
    // whenever their destination registers are reused by the fragment reads, puts vmcnt(0) INSIDE the steady K loop — which drains the LDS-DMA pipeline every
    // K-tile (measured: +20 % per K-tile, profiles/r04_gemm8_tile_switch.md)
    __builtin_amdgcn_s_waitcnt(0x0f70);
    if (!has_next) break;
    vt = vnext;
    if constexpr (SK) {
      ++item;
      kb = kbn;
      ke = ken;
    }
    m0 = m0n;
    n0 = n0n;
    cprob = probn;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (TRACE) {
    if ((blockIdx.x == 0 || blockIdx.x == 100) && (wave == 0 || wave == 4) && lane == 0) {
      unsigned* dst = g_gemm8_trace + ((blockIdx.x ? 1 : 0) * 2 + (wave ? 1 : 0)) * 15;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 5; ++k) dst[i * 5 + k] = tr[i][k];
    }
  }
#undef STAMP
#undef VMCNT8
#undef SROW
#undef CC
#undef BAR
}

__global__ __launch_bounds__(NT) void gemm_nt_8phase_kernel(AitkGemmArgs p) { gemm8_body<false, false>(p, p); }
__global__ __launch_bounds__(NT) void gemm_nt_8phase_grouped_kernel(AitkGemmArgs p, AitkGemmArgs p2) { gemm8_body<true, false>(p, p2); }
// measurement instantiations: the generic epilogue only (AITK_GEMM8_FE=0: A/B and bit-exactness of the fast forms), the s_memtime stamps
// (AITK_GEMM8_TRACE=1, plain kernel)
__global__ __launch_bounds__(NT) void gemm_nt_8phase_ge_kernel(AitkGemmArgs p) { gemm8_body<false, false, false, false>(p, p); }
__global__ __launch_bounds__(NT) void gemm_nt_8phase_grouped_ge_kernel(AitkGemmArgs p, AitkGemmArgs p2) { gemm8_body<true, false, false, false>(p, p2); }
__global__ __launch_bounds__(NT) void gemm_nt_8phase_tr_kernel(AitkGemmArgs p) { gemm8_body<false, false, false, true, true>(p, p); }
__global__ __launch_bounds__(NT) void gemm_nt_8phase_tr_ge_kernel(AitkGemmArgs p) { gemm8_body<false, false, false, false, true>(p, p); }
// AITK_EPI_EMIT_T (BIAS | GELU launches that also leave the column-tile partials of the consumer's lora_down product)
__global__ __launch_bounds__(NT) void gemm_nt_8phase_et_kernel(AitkGemmArgs p) { gemm8_body<false, false, false, true, false, 1>(p, p); }
__global__ __launch_bounds__(NT) void gemm_nt_8phase_grouped_et_kernel(AitkGemmArgs p, AitkGemmArgs p2) { gemm8_body<true, false, false, true, false, 1>(p, p2); }
// ... for a rank-32 consumer (two 16-rank blocks: t_rank = 32)
__global__ __launch_bounds__(NT) void gemm_nt_8phase_et32_kernel(AitkGemmArgs p) { gemm8_body<false, false, false, true, false, 2>(p, p); }
__global__ __launch_bounds__(NT) void gemm_nt_8phase_grouped_et32_kernel(AitkGemmArgs p, AitkGemmArgs p2) { gemm8_body<true, false, false, true, false, 2>(p, p2); }
// stream-K tail (opt-in, AITK_GEMM8_SK; mode 1 picks these when the last tile round leaves >= AITK_GEMM8_SK_MIN_IDLE % of a tile time idle)
// Only the two-problem form is instantiated: it reads the problem descriptors through the kernarg pointer where it needs them and keeps its K loop free of
// scratch traffic (the one-problem form holds the descriptor in SGPRs and spilt staging offsets in the loop's tail: a reload in front of a DMA waits for
// vmcnt(0) and drains the pipeline — measured 2.5x slower); a single problem is launched with an empty second one (M = 0: no tiles).
__global__ __launch_bounds__(NT) void gemm_nt_8phase_grouped_sk_kernel(AitkGemmArgs p, AitkGemmArgs p2, AitkSkArgs sk) { gemm8_body<true, false, false, true, false, 0, true>(p, p2); }
__global__ __launch_bounds__(NT) void gemm_nt_8phase_grouped_et_sk_kernel(AitkGemmArgs p, AitkGemmArgs p2, AitkSkArgs sk) { gemm8_body<true, false, false, true, false, 1, true>(p, p2); }
// W8A8: e4m3 activations (per-row scale) x e4m3 weights (per-row-of-B scale) on the MX-scaled fp8 MFMA, bf16 LoRA slab on top
__global__ __launch_bounds__(NT) void gemm_nt_8phase_f8_kernel(AitkGemmArgs p) { gemm8_body<false, true, false, false>(p, p); }
__global__ __launch_bounds__(NT) void gemm_nt_8phase_f8_grouped_kernel(AitkGemmArgs p, AitkGemmArgs p2) { gemm8_body<true, true, false, false>(p, p2); }
// implicit-GEMM 3x3 convolution on the persistent 8-phase schedule (UNet / VAE convolutions and their data gradients)
__global__ __launch_bounds__(NT) void gemm_nt_8phase_conv_kernel(AitkGemmArgs p) { gemm8_body<false, false, true>(p, p); }

// Called by aitk_gemm_nt (gemm.hip) for big bf16 problems.  Returns AITK_OK after launching, or 1 when the shape is
// outside this kernel's contract (caller falls back to the 2-barrier kernels).
static int gemm8_contract(const AitkGemmArgs* a) {
  const bool f8 = a->b_scale_mode == 3;
  if (a->flags & AITK_EPI_EMIT_T) {  // whole column tiles (the wave's 64 columns), any row count (stores predicated per row), the BIAS | GELU form, plain rows
    if (a->flags != (AITK_EPI_BIAS | AITK_EPI_GELU | AITK_EPI_EMIT_T) || (a->N % BN) || a->c_seg_rows || a->conv_mode || a->b_scale_mode) return 1;
    if (a->t_rank != 0 && a->t_rank != 16 && a->t_rank != 32) return 1;
    if (!a->t_partial || !a->t_p || !a->t_p_lo || (a->t_ldp % 8) || a->t_tile0 < 0 || (((uintptr_t)a->t_p | (uintptr_t)a->t_p_lo | (uintptr_t)a->t_partial) & 15)) return 1;
  }
  if (a->b_scale_mode && !f8) return 1;
  if (a->conv_mode) {
    // 2-D 3x3 form only, K-tiles inside one tap, the split-slab epilogue stays on the 2-barrier kernel, and the whole image batch (plus
    // the one-row shift of the descriptor base and the largest tap shift) inside the 2-GiB buffer window
    if (f8 || a->conv_t3d || (a->conv_Cin % 64) || a->K != 9 * a->conv_Cin || (a->flags & AITK_EPI_SPLIT_SLAB) || a->a_seg_rows) return 1;
    if (a->conv_pad_t < 0 || a->conv_pad_t > 1 || a->conv_pad_l < 0 || a->conv_pad_l > 1 || a->conv_stride < 1) return 1;
    const long nb = ((long)a->M + a->conv_HoWo - 1) / a->conv_HoWo;
    if (((nb * a->conv_H + 3) * a->conv_W + 4) * a->conv_Cin * 2 >= 0x7ff00000L) return 1;
  }
  if ((a->K % 16) || (a->K2 % 16) || (a->N % 8) || (a->ldc % 8)) return 1;
  if (f8 && ((a->lda % 16) || (a->ldb % 16) || (a->a_seg_stride % 16) || (((uintptr_t)a->A | (uintptr_t)a->B) & 15) ||
             (((uintptr_t)a->a_scale | (uintptr_t)a->b_scale) & 15)))
    return 1;
  const int kb = f8 ? 128 : BK;
  const int nsteps = (a->K + kb - 1) / kb + (a->K2 > 0 ? (a->K2 + BK - 1) / BK : 0);
  if (nsteps < 2) return 1;
  if ((a->flags & (AITK_EPI_ADD_AUX | AITK_EPI_DGELU | AITK_EPI_GATE_RES)) && (a->ld_aux_in % 8)) return 1;
  if ((a->flags & AITK_EPI_GELU) && (a->ld_aux_out % 8)) return 1;
  if ((a->flags & AITK_EPI_GATE_RES) && ((a->ld_gate % 8) || (a->aux_out && (a->ld_aux_out % 8)))) return 1;
  if ((a->flags & AITK_EPI_BIAS) && ((uintptr_t)a->bias & 15)) return 1;
  if (((uintptr_t)a->aux_in | (uintptr_t)a->aux_out | (uintptr_t)a->gate | (uintptr_t)a->C) & 15) return 1;
  return 0;
}
// measurement knobs, re-read on every launch (the A/B tools flip them inside one process)
static int gemm8_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
static int gemm8_cus() {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const void* kernels[] = {(const void*)gemm_nt_8phase_kernel,        (const void*)gemm_nt_8phase_grouped_kernel,    (const void*)gemm_nt_8phase_f8_kernel,
                             (const void*)gemm_nt_8phase_f8_grouped_kernel, (const void*)gemm_nt_8phase_conv_kernel,       (const void*)gemm_nt_8phase_ge_kernel,
                             (const void*)gemm_nt_8phase_grouped_ge_kernel, (const void*)gemm_nt_8phase_tr_kernel,         (const void*)gemm_nt_8phase_tr_ge_kernel,
                             (const void*)gemm_nt_8phase_et_kernel,         (const void*)gemm_nt_8phase_grouped_et_kernel,     (const void*)gemm_nt_8phase_et32_kernel,
                             (const void*)gemm_nt_8phase_grouped_et32_kernel,   (const void*)gemm_nt_8phase_grouped_sk_kernel,     (const void*)gemm_nt_8phase_grouped_et_sk_kernel};
    for (const void* k : kernels)
      if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, EPI_OFF + 32768) != hipSuccess) {
        n_cu = 0;
        return 0;
      }
  }
  return n_cu;
}
// ---- stream-K tail: when it pays, and its workspace (one 256-KiB accumulator slot + one flag per CU, per (device, stream): launches on one stream are
// ordered, two streams must not share slots).  Allocated on first use outside graph capture; a launch that cannot have one stays data-parallel.
static bool gemm8_sk_plan(const AitkGemmArgs* a, int tiles, int n_cu) {
  const int mode = gemm8_env("AITK_GEMM8_SK", 0);  // 0 off (default: see the SK note at gemm8_body), 1 when the last round is poorly filled, 2 whenever the contract allows
  if (!mode || a->b_scale_mode || a->conv_mode || tiles % n_cu == 0 || (n_cu & 7)) return false;
  if ((a->flags & AITK_EPI_EMIT_T) && a->t_rank == 32) return false;
  if (!gemm8_env("AITK_GEMM8_FE", 1) || gemm8_env("AITK_GEMM8_TRACE", 0)) return false;
  const int nk1 = (a->K + BK - 1) / BK, nk2 = a->K2 > 0 ? (a->K2 + BK - 1) / BK : 0, nsteps = nk1 + nk2;
  if (nsteps < 8) return false;
  const long R = tiles % n_cu;
  if ((R / 8) * nsteps / (n_cu / 8) < 8 || (long)(n_cu + 1) * R * nsteps >= 0x7fffffffL) return false;  // every workgroup's share of its XCD's tail: >= 8 K-tiles
  if (mode >= 2) return true;
  const int rounds = tiles / n_cu + 1;
  const double idle = rounds - (double)tiles / n_cu;  // tile times the data-parallel schedule leaves idle (chip average)
  return idle * 100.0 >= gemm8_env("AITK_GEMM8_SK_MIN_IDLE", 15);
}
struct Gemm8SkWs { int dev; hipStream_t st; AitkSkArgs args; };
static bool gemm8_sk_workspace(hipStream_t st, int n_cu, AitkSkArgs* out) {
  static Gemm8SkWs table[16];
  static int n = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  for (int i = 0; i < n; ++i)
    if (table[i].dev == dev && table[i].st == st) {
      *out = table[i].args;
      return true;
    }
  if (n == 16) return false;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false;
  AitkSkArgs w = {nullptr, nullptr};
  if (hipMalloc((void**)&w.ws, (size_t)n_cu * BM * BN * sizeof(float)) != hipSuccess) return false;
  if (hipMalloc((void**)&w.flags, (size_t)n_cu * sizeof(int)) != hipSuccess || hipMemsetAsync(w.flags, 0, (size_t)n_cu * sizeof(int), st) != hipSuccess) {
    hipFree(w.ws);
    if (w.flags) hipFree(w.flags);
    return false;
  }
  table[n].dev = dev;
  table[n].st = st;
  table[n].args = w;
  ++n;
  *out = w;
  return true;
}
// host view of the schedule (tests/test_capi_symbols.py: every K-tile of every tail tile is covered exactly once, chunk lengths respect the pipeline's minima)
extern "C" int aitk_probe_gemm8_sk_item(int32_t G, int32_t w, int32_t ntiles, int32_t nsteps, int32_t i, int32_t* out3) {
  if (G <= 0 || w < 0 || w >= G || ntiles <= 0 || nsteps <= 0 || i < 0 || !out3) return AITK_ERR_ARG;
  int v = 0, kb = 0, ke = 0;
  if ((G & 7) || !sk_item_of(G, w, ntiles, nsteps, i, v, kb, ke)) return 1;
  out3[0] = v;
  out3[1] = kb;
  out3[2] = ke;
  return AITK_OK;
}
extern "C" int aitk_probe_gemm8_sk_predecessor(int32_t G, int32_t w, int32_t ntiles, int32_t nsteps) { return sk_predecessor(G, w, ntiles, nsteps); }
extern "C" int aitk_gemm8_try_launch(const AitkGemmArgs* a, hipStream_t st) {
  if (gemm8_contract(a)) return 1;
  const int n_cu = gemm8_cus();
  if (!n_cu) return 1;
  const int tiles = ((a->M + BM - 1) / BM) * ((a->N + BN - 1) / BN);
  const int grid = tiles < n_cu ? tiles : n_cu;
  AitkSkArgs sk;
  if (gemm8_sk_plan(a, tiles, n_cu) && gemm8_sk_workspace(st, n_cu, &sk)) {
    AitkGemmArgs none = *a;  // the empty second problem
    none.M = 0;
    if (a->flags & AITK_EPI_EMIT_T) hipLaunchKernelGGL(gemm_nt_8phase_grouped_et_sk_kernel, dim3(n_cu), dim3(NT), EPI_OFF + 32768, st, *a, none, sk);
    else hipLaunchKernelGGL(gemm_nt_8phase_grouped_sk_kernel, dim3(n_cu), dim3(NT), EPI_OFF + 32768, st, *a, none, sk);
    return AITK_OK;
  }
  if ((a->flags & AITK_EPI_EMIT_T) && a->t_rank == 32) hipLaunchKernelGGL(gemm_nt_8phase_et32_kernel, dim3(grid), dim3(NT), EPI_OFF + 32768, st, *a);
  else if (a->flags & AITK_EPI_EMIT_T) hipLaunchKernelGGL(gemm_nt_8phase_et_kernel, dim3(grid), dim3(NT), EPI_OFF + 32768, st, *a);
  else if (a->conv_mode) hipLaunchKernelGGL(gemm_nt_8phase_conv_kernel, dim3(grid), dim3(NT), EPI_OFF + 32768, st, *a);
  else if (a->b_scale_mode == 3) hipLaunchKernelGGL(gemm_nt_8phase_f8_kernel, dim3(grid), dim3(NT), EPI_OFF + 32768, st, *a);
  else {
    const int fe = gemm8_env("AITK_GEMM8_FE", 1);
    if (gemm8_env("AITK_GEMM8_TRACE", 0)) {
      if (fe) hipLaunchKernelGGL(gemm_nt_8phase_tr_kernel, dim3(grid), dim3(NT), EPI_OFF + 32768, st, *a);
      else hipLaunchKernelGGL(gemm_nt_8phase_tr_ge_kernel, dim3(grid), dim3(NT), EPI_OFF + 32768, st, *a);
    } else if (fe) hipLaunchKernelGGL(gemm_nt_8phase_kernel, dim3(grid), dim3(NT), EPI_OFF + 32768, st, *a);
    else hipLaunchKernelGGL(gemm_nt_8phase_ge_kernel, dim3(grid), dim3(NT), EPI_OFF + 32768, st, *a);
  }
  return AITK_OK;
}
// the s_memtime stamps the TRACE instantiation left: [workgroup 0 / 100][wave 0 / 4][tile 1-3][top, after the top wait + barrier, end of the steady K loop,
// end of the K loop, end of the epilogue]
extern "C" int aitk_probe_gemm8_trace(unsigned* out60) {
  return hipMemcpyFromSymbol(out60, HIP_SYMBOL(g_gemm8_trace), sizeof(unsigned) * 60) == hipSuccess ? AITK_OK : 1;
}
// Two problems with equal N, K, K2 and flags in one persistent launch (aitk_gemm_nt_grouped); 1 = outside the contract.
extern "C" int aitk_gemm8_try_launch_grouped(const AitkGemmArgs* a, const AitkGemmArgs* b, hipStream_t st) {
  if (gemm8_contract(a) || gemm8_contract(b) || a->conv_mode || b->conv_mode) return 1;
  if (a->N != b->N || a->K != b->K || a->K2 != b->K2 || a->flags != b->flags || a->b_scale_mode != b->b_scale_mode) return 1;
  const int n_cu = gemm8_cus();
  if (!n_cu) return 1;
  const int tn = (a->N + BN - 1) / BN;
  const int tiles = ((a->M + BM - 1) / BM + (b->M + BM - 1) / BM) * tn;
  const int grid = tiles < n_cu ? tiles : n_cu;
  if ((a->flags & AITK_EPI_EMIT_T) && (a->t_rank == 32) != (b->t_rank == 32)) return 1;
  AitkSkArgs sk;
  if (gemm8_sk_plan(a, tiles, n_cu) && gemm8_sk_workspace(st, n_cu, &sk)) {
    if (a->flags & AITK_EPI_EMIT_T) hipLaunchKernelGGL(gemm_nt_8phase_grouped_et_sk_kernel, dim3(n_cu), dim3(NT), EPI_OFF + 32768, st, *a, *b, sk);
    else hipLaunchKernelGGL(gemm_nt_8phase_grouped_sk_kernel, dim3(n_cu), dim3(NT), EPI_OFF + 32768, st, *a, *b, sk);
    return AITK_OK;
  }
  if ((a->flags & AITK_EPI_EMIT_T) && a->t_rank == 32) hipLaunchKernelGGL(gemm_nt_8phase_grouped_et32_kernel, dim3(grid), dim3(NT), EPI_OFF + 32768, st, *a, *b);
  else if (a->flags & AITK_EPI_EMIT_T) hipLaunchKernelGGL(gemm_nt_8phase_grouped_et_kernel, dim3(grid), dim3(NT), EPI_OFF + 32768, st, *a, *b);
  else if (a->b_scale_mode == 3) hipLaunchKernelGGL(gemm_nt_8phase_f8_grouped_kernel, dim3(grid), dim3(NT), EPI_OFF + 32768, st, *a, *b);
  else if (gemm8_env("AITK_GEMM8_FE", 1)) hipLaunchKernelGGL(gemm_nt_8phase_grouped_kernel, dim3(grid), dim3(NT), EPI_OFF + 32768, st, *a, *b);
  else hipLaunchKernelGGL(gemm_nt_8phase_grouped_ge_kernel, dim3(grid), dim3(NT), EPI_OFF + 32768, st, *a, *b);
  return AITK_OK;
}
