// bf16 MFMA GEMM with the rank-r LoRA path fused into the same output tile (gfx950 / MI355X).
//
//   C[M,N] = epi( A[M,K] * B[N,K]^T  +  A2[M,K2] * B2[N,K2]^T  + bias[N] )
//
// A/B are the frozen-base operands (activation x, weight W in torch Linear layout [out,in]); A2/B2 is the
// LoRA K-slab: A2 = T = s*m_b*(x*lora_down^T) (bf16, produced by aitk_lora_down) and B2 = lora_up (bf16 shadow
// [out,r]).  Concatenating the rank-r slab onto the K loop is the MFMA-native form of the reference's
//   org_forward(x) + lora_up(lora_down(x.float())) * scale * multiplier      (toolkit/network_mixins.py:304-342)
// and, for the backward data-gradient, of  dX = dY*W + (s*m*dY*B)*A  (SURVEY.md §3.2).
//
// Tile: 128x128x64 per 256-thread workgroup (4 waves, 2x2), each wave 64x64 = 2x2 v_mfma_f32_32x32x16_bf16.
// LDS: double-buffered 2 x (16 KiB A + 16 KiB B), rows of 128 B XOR-swizzled at 16-B granularity
// (chunk ^= (row>>1)&7) so the ds_read_b128 fragment reads are conflict-free (guide §5.5 T2).
// Staging: STAGE=0 global->VGPR->LDS with the global loads issued before the MFMA phase (T14);
//          STAGE=1 global_load_lds_dwordx4 (LDS-DMA, lane-linear destination, swizzle on the source address).
// blockIdx -> tile: bijective XCD remap + grouped (8 row-tiles) ordering for per-XCD L2 reuse (T1).
#include <cstdlib>
#include "common.h"
#include "aitk_args.h"

#define BK 64

// 128 B of zeros: source of LDS-DMA chunks that lie beyond K (A operand), so every K-step can run all four 16-wide
// sub-steps branch-free (zeros x finite = 0) instead of branching on the K tail.
__device__ __attribute__((aligned(128))) bf16_t g_zero_page[64];

__device__ __forceinline__ const bf16_t* seg_row(const bf16_t* base, long ld, int seg_rows, long seg_stride, int m) {
  if (seg_rows > 0) {
    int s = m / seg_rows;
    int w = m - s * seg_rows;
    return base + (long)s * seg_stride + (long)w * ld;
  }
  return base + (long)m * ld;
}

// Tile configs: <128,128,2,2> (4 waves, 64x64 per wave, 64 KiB LDS, 2 workgroups/CU) for small / ragged problems and
// <256,256,2,4> (8 waves, 128x64 per wave, 128 KiB LDS, 1 workgroup/CU): twice the operand reuse per LDS byte.
// CONV: A is an NHWC image [B, H, W, Cin] and the K axis runs over (3x3 tap, Cin): implicit-GEMM convolution — row m is
// output pixel (b, oy, ox), the 16-B chunk at k = tap*Cin + cin is fetched from input pixel (oy*stride+ky-pad_t,
// ox*stride+kx-pad_l), out-of-image chunks come from a zero page.  Replaces nn.Conv2d inside the VAE encoder the reference
// runs at toolkit/stable_diffusion_model.py:2567 (diffusers AutoencoderKL).
// conv_t3d != 0 (kt | tstride << 8 | ks << 16): the batch index is a FRAME and K also runs over kt temporal taps, k = ((dt*ks + ky)*ks +
// kx)*Cin + cin, read from input frame b*tstride + dt — the causal 3-D convolutions of the Wan2.1 video VAE (AutoencoderKLWan, reached
// from toolkit/models/wan21/wan21.py:659).  No bounds check in time: the caller's buffer starts with the causal zero frames.
template <int STAGE, int BM, int BN, int WM, int WN, bool CONV = false>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_kernel(AitkGemmArgs p) {
  constexpr int NT = 64 * WM * WN;           // threads
  constexpr int RPP = NT / 8;                // tile rows staged per pass (8 x 16-B chunks per 128-B row)
  constexpr int PA = BM / RPP, PB = BN / RPP;  // staging passes per operand
  constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, BUF_BYTES = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN;
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  const int lid = xcd_remap(blockIdx.x, nwg);
  const int GROUP = 8;
  const int group_sz = GROUP * tiles_n;
  const int gid = lid / group_sz;
  const int first_m = gid * GROUP;
  const int gm = min(tiles_m - first_m, GROUP);
  const int tm = first_m + (lid % group_sz) % gm;
  const int tn = (lid % group_sz) / gm;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- staging geometry: thread owns physical chunk pc of rows r_i = (tid>>3) + 32 i, i = 0..3 ----
  const int srow = tid >> 3;
  const int pc = tid & 7;
  // logical 16-B chunk held at physical slot pc.  Key = (row>>1)&7: two 128-B rows share one 256-B bank line, so the
  // 16 rows of a ds_read_b128 lane group land on 16 distinct 16-B slots (conflict-free); same key for every pass i.
  const int cc = pc ^ ((srow >> 1) & 7);
  // 32-bit element offsets from the (wave-uniform, SGPR) base pointers: half the address registers of 64-bit pointers
  // and the loads can use the saddr+voffset form.  The launcher checks every operand spans < 2^32 elements.
  long pa[PA];  // CONV: signed element offset (can be negative at the image border); else unsigned offsets below
  unsigned ao[PA], bo[PB], ao2[PA], bo2[PB];
  int iy0[CONV ? PA : 1], ix0[CONV ? PA : 1];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    int ra = min(m0 + srow + RPP * i, p.M - 1);
    if constexpr (CONV) {
      const int b = ra / p.conv_HoWo, rem = ra - b * p.conv_HoWo;
      const int oy = rem / p.conv_Wo, ox = rem - oy * p.conv_Wo;
      iy0[i] = oy * p.conv_stride - p.conv_pad_t;
      ix0[i] = ox * p.conv_stride - p.conv_pad_l;
      const int fr = p.conv_t3d ? b * ((p.conv_t3d >> 8) & 255) : b;  // first input frame of output frame b
      pa[i] = (((long)fr * p.conv_H + iy0[i]) * p.conv_W + ix0[i]) * p.conv_Cin;  // may lie outside; only used when valid
      ao[i] = 0;
      ao2[i] = p.K2 > 0 ? (unsigned)((long)ra * p.lda2) : 0u;  // LoRA K-slab of a convolution adapter: plain rows of A2 [M, K2]
    } else {
      pa[i] = 0;
      ao[i] = (unsigned)(seg_row(p.A, p.lda, p.a_seg_rows, p.a_seg_stride, ra) - p.A);
      ao2[i] = p.K2 > 0 ? (unsigned)((long)ra * p.lda2) : 0u;
    }
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    int rb = min(n0 + srow + RPP * i, p.N - 1);
    bo[i] = (unsigned)((long)rb * p.ldb);
    bo2[i] = p.K2 > 0 ? (unsigned)((long)rb * p.ldb2) : 0u;
  }
  const int nk1 = (p.K + BK - 1) / BK;
  const int nk2 = p.K2 > 0 ? (p.K2 + BK - 1) / BK : 0;
  const int nsteps = nk1 + nk2;

  f32x16_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  uint4 ra_reg[PA], rb_reg[PB];

  auto step_info = [&](int s, int& k0, int& Kseg, bool& second) {
    second = s >= nk1;
    k0 = (second ? s - nk1 : s) * BK;
    Kseg = second ? p.K2 : p.K;
  };

  // ---- STAGE 3: weight-only fp8 (OCP e4m3) base operand.  B holds bytes [N, K] plus a per-output-channel fp32 scale; the
  // chunk is dequantised to bf16 (w = bf16(fp8 * scale), the value a weight-only-quantised Linear multiplies with — reference:
  // optimum-quanto qfloat8 / torchao Float8WeightOnly, toolkit/util/quantize.py:43-75) while it moves VGPR -> LDS.  b_scale_mode 1:
  // scale indexed by the B row (forward, rows = output channels); 2: by the contraction index (dgrad on W^T).
  auto dequant8 = [&](uint2 q, const float* sc8, float sc_row) -> uint4 {
    float f[8];
    f[0] = __builtin_amdgcn_cvt_f32_fp8(q.x, 0); f[1] = __builtin_amdgcn_cvt_f32_fp8(q.x, 1);
    f[2] = __builtin_amdgcn_cvt_f32_fp8(q.x, 2); f[3] = __builtin_amdgcn_cvt_f32_fp8(q.x, 3);
    f[4] = __builtin_amdgcn_cvt_f32_fp8(q.y, 0); f[5] = __builtin_amdgcn_cvt_f32_fp8(q.y, 1);
    f[6] = __builtin_amdgcn_cvt_f32_fp8(q.y, 2); f[7] = __builtin_amdgcn_cvt_f32_fp8(q.y, 3);
    if (sc8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] *= sc8[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] *= sc_row;
    }
    uint4 o;
    o.x = pack2bf(f[0], f[1]); o.y = pack2bf(f[2], f[3]); o.z = pack2bf(f[4], f[5]); o.w = pack2bf(f[6], f[7]);
    return o;
  };
  float brow_scale[STAGE == 3 ? PB : 1];
  if constexpr (STAGE == 3) {
#pragma unroll
    for (int i = 0; i < PB; ++i) brow_scale[i] = p.b_scale_mode == 1 ? p.b_scale[min(n0 + srow + RPP * i, p.N - 1)] : 1.0f;
  }
  uint2 rb8_reg[STAGE == 3 ? PB : 1];
  auto load_b8 = [&](int s) {  // issue the global loads of the B tile of step s (fp8 bytes, or bf16 for the LoRA slab)
    int k0, Kseg;
    bool second;
    step_info(s, k0, Kseg, second);
    const int kk = k0 + cc * 8;
    const bool valid = kk < Kseg;
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      if (second) {
        rb_reg[i] = valid ? *reinterpret_cast<const uint4*>(p.B2 + bo2[i] + kk) : make_uint4(0, 0, 0, 0);
      } else {
        const uint8_t* b8 = reinterpret_cast<const uint8_t*>(p.B) + bo[i];
        rb8_reg[i] = valid ? *reinterpret_cast<const uint2*>(b8 + kk) : make_uint2(0, 0);
      }
    }
  };
  auto write_b8 = [&](int s, int buf) {
    int k0, Kseg;
    bool second;
    step_info(s, k0, Kseg, second);
    const int kk = k0 + cc * 8;
    char* sb = smem + buf * BUF_BYTES + A_BYTES;
    float sc8[8];
    const bool per_k = (!second) && p.b_scale_mode == 2;
    if (per_k) {
#pragma unroll
      for (int e = 0; e < 8; ++e) sc8[e] = (kk + e < Kseg) ? p.b_scale[kk + e] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const uint4 v = second ? rb_reg[i] : dequant8(rb8_reg[i], per_k ? sc8 : nullptr, brow_scale[i]);
      *reinterpret_cast<uint4*>(sb + (tid + NT * i) * 16) = v;
    }
  };

  auto load_regs = [&](int s) {
    int k0, Kseg;
    bool second;
    step_info(s, k0, Kseg, second);
    const int kk = k0 + cc * 8;
    const bool valid = kk < Kseg;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const bf16_t* a = second ? p.A2 + ao2[i] : p.A + ao[i];
      ra_reg[i] = valid ? *reinterpret_cast<const uint4*>(a + kk) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const bf16_t* b = second ? p.B2 + bo2[i] : p.B + bo[i];
      rb_reg[i] = valid ? *reinterpret_cast<const uint4*>(b + kk) : make_uint4(0, 0, 0, 0);
    }
  };
  auto write_lds = [&](int buf) {
    char* sa = smem + buf * BUF_BYTES;
    char* sb = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < PA; ++i) *reinterpret_cast<uint4*>(sa + (tid + NT * i) * 16) = ra_reg[i];
#pragma unroll
    for (int i = 0; i < PB; ++i) *reinterpret_cast<uint4*>(sb + (tid + NT * i) * 16) = rb_reg[i];
  };
  auto issue_glds = [&](int s, int buf) {
    int k0, Kseg;
    bool second;
    step_info(s, k0, Kseg, second);
    int kk = k0 + cc * 8;
    const bool kvalid_chunk = kk < Kseg;
    if (!kvalid_chunk) kk = 0;  // B: in-bounds finite data (multiplied by the zero A chunk); A: zero page below
    char* sa = smem + buf * BUF_BYTES;
    char* sb = sa + A_BYTES;
    // destination = wave-uniform base + lane*16 (LDS-DMA is lane-linear)
    if (CONV && second) {  // K-slab steps of a convolution with an adapter: A2 is an ordinary row-major matrix
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const bf16_t* a = kvalid_chunk ? p.A2 + ao2[i] + kk : g_zero_page;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a,
                                         (__attribute__((address_space(3))) void*)(sa + (wave * 64 + NT * i) * 16), 16, 0, 0);
      }
    } else if constexpr (CONV) {
      const int kc = k0 + cc * 8;
      const bool kin = kc < p.K;
      const int tap3 = kin ? kc / p.conv_Cin : 0;
      const int cin = kc - tap3 * p.conv_Cin;
      const int ks = p.conv_t3d ? (p.conv_t3d >> 16) & 255 : 3;  // spatial kernel size (3, or 1 for the (3,1,1) time convolution)
      const int dt = tap3 / (ks * ks), tap = tap3 - dt * ks * ks;  // temporal tap (always 0 for a 2-D convolution)
      const int ky = tap / ks, kx = tap - ks * ky;
      const long toff = ((long)(dt * p.conv_H + ky) * p.conv_W + kx) * p.conv_Cin + cin;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const bool ok = kin && (unsigned)(iy0[i] + ky) < (unsigned)p.conv_H && (unsigned)(ix0[i] + kx) < (unsigned)p.conv_W;
        const bf16_t* src = ok ? p.A + (pa[i] + toff) : p.zero_page;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(sa + (wave * 64 + NT * i) * 16), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const bf16_t* a = kvalid_chunk ? (second ? p.A2 + ao2[i] : p.A + ao[i]) + kk : g_zero_page;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a,
                                         (__attribute__((address_space(3))) void*)(sa + (wave * 64 + NT * i) * 16), 16, 0, 0);
      }
    }
    if constexpr (STAGE != 3) {
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const bf16_t* b = second ? p.B2 + bo2[i] : p.B + bo[i];
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b + kk),
                                         (__attribute__((address_space(3))) void*)(sb + (wave * 64 + NT * i) * 16), 16, 0, 0);
      }
    }
  };
  auto compute = [&](int s, int buf) {
    // Fragments are double-buffered in registers: the ds_read_b128 group of k-substep ks+1 is issued before the MFMA
    // group of ks, so an MFMA only waits (counted lgkmcnt) for reads issued one group earlier.  K tails need no branch:
    // A chunks beyond K are zero in LDS (zero-filled by the VGPR path / DMA'd from g_zero_page).
    const char* sa = smem + buf * BUF_BYTES;
    const char* sb = sa + A_BYTES;
    const int l31 = lane & 31, h = lane >> 5;
    s16x8_t af[2][MI], bfr[2][NI];
    auto ldfrag = [&](int ks, int slot) {
      const int c = ks * 2 + h;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        int row = wr * (32 * MI) + mi * 32 + l31;
        af[slot][mi] = *reinterpret_cast<const s16x8_t*>(sa + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        int row = wc * (32 * NI) + ni * 32 + l31;
        bfr[slot][ni] = *reinterpret_cast<const s16x8_t*>(sb + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
      }
    };
    ldfrag(0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks + 1 < 4) ldfrag(ks + 1, (ks + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = mfma32(bfr[ks & 1][ni], af[ks & 1][mi], acc[mi][ni]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if (STAGE == 0) {
    load_regs(0);
    write_lds(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
      if (s + 1 < nsteps) load_regs(s + 1);
      compute(s, s & 1);
      if (s + 1 < nsteps) write_lds((s + 1) & 1);
      __syncthreads();
    }
  } else if (STAGE == 1) {
    issue_glds(0, 0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
      if (s + 1 < nsteps) issue_glds(s + 1, (s + 1) & 1);
      compute(s, s & 1);
      __syncthreads();
    }
  } else if (STAGE == 3) {
    issue_glds(0, 0);
    load_b8(0);
    write_b8(0, 0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
      if (s + 1 < nsteps) {
        issue_glds(s + 1, (s + 1) & 1);
        load_b8(s + 1);
      }
      compute(s, s & 1);
      if (s + 1 < nsteps) write_b8(s + 1, (s + 1) & 1);
      __syncthreads();
    }
  }

  // ---------------- epilogue ----------------
  const int l31 = lane & 31, h = lane >> 5;
  const int flags = p.flags;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = m0 + wr * (32 * MI) + mi * 32 + l31;
    if (m >= p.M) continue;
    bf16_t* crow = const_cast<bf16_t*>(seg_row(p.C, p.ldc, p.c_seg_rows, p.c_seg_stride, m));
    const bf16_t* gate_row = (flags & AITK_EPI_GATE_RES) ? p.gate + (long)(m / p.gate_rows) * p.ld_gate : nullptr;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nb = n0 + wc * (32 * NI) + ni * 32 + 8 * g + 4 * h;
        if (nb >= p.N) continue;
        float v[4];
        if (flags & AITK_EPI_SPLIT_SLAB) {
          // B rows = 16-rank blocks [P_hi(16) ; P_lo(16)] of a rank-rp projection (rp = N / 2): t = A P_hi^T + A P_lo^T in fp32 — columns
          // n and n + 16 of a 32-column block sit in register groups g and g + 2 of the same lane — written as the K-slab triple
          // [hi(rp) | lo(rp) | hi(rp)] like aitk_lora_down(split_rp = rp)
          if (g >= 2) continue;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][4 * g + e] + acc[mi][ni][4 * (g + 2) + e];
          if (flags & AITK_EPI_COL_SCALE) {
            const f32x4_t cs = *reinterpret_cast<const f32x4_t*>(p.col_scale + nb);
            v[0] *= cs[0]; v[1] *= cs[1]; v[2] *= cs[2]; v[3] *= cs[3];
          }
          uint2 hi, lo;
          hi.x = pack2bf(v[0], v[1]); hi.y = pack2bf(v[2], v[3]);
          lo.x = pack2bf(v[0] - bf_lo(hi.x), v[1] - bf_hi(hi.x));
          lo.y = pack2bf(v[2] - bf_lo(hi.y), v[3] - bf_hi(hi.y));
          const int rp = p.N >> 1, rk = ((nb >> 5) << 4) + (nb & 31);  // rank of v[0]
          *reinterpret_cast<uint2*>(crow + rk) = hi;
          *reinterpret_cast<uint2*>(crow + rp + rk) = lo;
          *reinterpret_cast<uint2*>(crow + 2 * rp + rk) = hi;
          continue;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][4 * g + e];
        if (flags & AITK_EPI_COL_SCALE) {
          const f32x4_t cs = *reinterpret_cast<const f32x4_t*>(p.col_scale + nb);
          v[0] *= cs[0]; v[1] *= cs[1]; v[2] *= cs[2]; v[3] *= cs[3];
        }
        if (flags & AITK_EPI_BIAS) {
          uint2 bb = *reinterpret_cast<const uint2*>(p.bias + nb);
          v[0] += bf2f(bb.x & 0xffff); v[1] += bf2f(bb.x >> 16);
          v[2] += bf2f(bb.y & 0xffff); v[3] += bf2f(bb.y >> 16);
        }
        if (flags & AITK_EPI_BIAS_ROW) {
          const float br = bf2f(p.bias[m]);
          v[0] += br; v[1] += br; v[2] += br; v[3] += br;
        }
        if (flags & AITK_EPI_ADD_AUX) {
          uint2 rr = *reinterpret_cast<const uint2*>(p.aux_in + (long)m * p.ld_aux_in + nb);
          v[0] += bf2f(rr.x & 0xffff); v[1] += bf2f(rr.x >> 16);
          v[2] += bf2f(rr.y & 0xffff); v[3] += bf2f(rr.y >> 16);
        }
        if (flags & AITK_EPI_ACCUM) {
          uint2 cc2 = *reinterpret_cast<const uint2*>(crow + nb);
          v[0] += bf2f(cc2.x & 0xffff); v[1] += bf2f(cc2.x >> 16);
          v[2] += bf2f(cc2.y & 0xffff); v[3] += bf2f(cc2.y >> 16);
        }
        if (flags & AITK_EPI_GELU) {
          // u = bf16(pre-activation) is saved for backward; h = gelu_tanh(u) (torch evaluates GELU on the bf16 value)
          uint2 uo;
          uo.x = pack2bf(v[0], v[1]); uo.y = pack2bf(v[2], v[3]);
          *reinterpret_cast<uint2*>(p.aux_out + (long)m * p.ld_aux_out + nb) = uo;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f(bfround(v[e]));
        }
        if (flags & AITK_EPI_DGELU) {
          uint2 uu = *reinterpret_cast<const uint2*>(p.aux_in + (long)m * p.ld_aux_in + nb);
          v[0] *= gelu_tanh_grad_f(bf2f(uu.x & 0xffff)); v[1] *= gelu_tanh_grad_f(bf2f(uu.x >> 16));
          v[2] *= gelu_tanh_grad_f(bf2f(uu.y & 0xffff)); v[3] *= gelu_tanh_grad_f(bf2f(uu.y >> 16));
        }
        if (flags & AITK_EPI_GATE_RES) {
          // y = bf16(linear out) saved (d_gate needs it); x_new = res + gate[b] * y
          uint2 yo;
          yo.x = pack2bf(v[0], v[1]); yo.y = pack2bf(v[2], v[3]);
          if (p.aux_out) *reinterpret_cast<uint2*>(p.aux_out + (long)m * p.ld_aux_out + nb) = yo;
          uint2 rr = *reinterpret_cast<const uint2*>(p.aux_in + (long)m * p.ld_aux_in + nb);
          uint2 gg = *reinterpret_cast<const uint2*>(gate_row + nb);
          v[0] = bf2f(rr.x & 0xffff) + bf2f(gg.x & 0xffff) * bfround(v[0]);
          v[1] = bf2f(rr.x >> 16) + bf2f(gg.x >> 16) * bfround(v[1]);
          v[2] = bf2f(rr.y & 0xffff) + bf2f(gg.y & 0xffff) * bfround(v[2]);
          v[3] = bf2f(rr.y >> 16) + bf2f(gg.y >> 16) * bfround(v[3]);
        }
        uint2 o;
        o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
        *reinterpret_cast<uint2*>(crow + nb) = o;
      }
    }
  }
}

// gemm8.hip: persistent 8-phase kernel for big bf16 problems (returns 1 when the shape is outside its contract)
extern "C" int aitk_gemm8_try_launch(const AitkGemmArgs* a, hipStream_t st);
extern "C" int aitk_gemm8_try_launch_grouped(const AitkGemmArgs* a, const AitkGemmArgs* b, hipStream_t st);

static int gemm_check(const AitkGemmArgs* a);
// smallest number of 256x256 output tiles for which the persistent 8-phase kernel (one workgroup per CU) is chosen over the 128x128
// kernel (two workgroups per CU); AITK_BIG_TILES_MIN overrides it for A/B measurements.  128 = half the chip: measured on the SDXL
// step (its 8192 x 1280 token GEMMs are 160 tiles) 579 -> 688 TFLOP/s over all GEMM + conv launches against the old value 192, and
// +1.6 % on the FLUX step without stream pairing (text-stream GEMMs, 168 tiles); 96 is no better (profiles/r02_notes_grouped_gemm.md)
static long big_tiles_min() {
  static long v = -1;
  if (v < 0) {
    const char* e = getenv("AITK_BIG_TILES_MIN");
    v = (e && atol(e) > 0) ? atol(e) : 128;
  }
  return v;
}

// Two independent problems C_i = epi(A_i B_i^T + A2_i B2_i^T + ...) with the same N, K, K2 and epilogue flags (e.g. the image- and
// text-stream projections of a FLUX double block: different weights, adapters and row counts).  When together they fill the chip
// they run as ONE persistent 8-phase launch whose tile list is the concatenation of both, so the small problem's tiles fill the
// last, partly empty tile round of the big one; otherwise (or outside the 8-phase contract) the two are launched back to back.
extern "C" int aitk_gemm_nt_grouped(const AitkGemmArgs* a, const AitkGemmArgs* b, aitk_stream_t stream_) {
  int rc = gemm_check(a);
  if (rc) return rc;
  rc = gemm_check(b);
  if (rc) return rc;
  const bool same = a->N == b->N && a->K == b->K && a->K2 == b->K2 && a->flags == b->flags;
  const bool w8a8 = a->b_scale_mode == 3 && b->b_scale_mode == 3;
  const bool plain = !a->conv_mode && !b->conv_mode && ((!a->b_scale_mode && !b->b_scale_mode) || w8a8) && a->tile_mode == 0 && b->tile_mode == 0 &&
                     a->stage_mode == 1 && b->stage_mode == 1;
  const long t256 = (long)((a->M + 255) / 256 + (b->M + 255) / 256) * ((a->N + 255) / 256);
  if (same && plain && a->N >= 512 && (w8a8 || t256 >= big_tiles_min()) && aitk_gemm8_try_launch_grouped(a, b, (hipStream_t)stream_) == AITK_OK) {
    AITK_LAUNCH_CHECK();
    return AITK_OK;
  }
  rc = aitk_gemm_nt(a, stream_);
  return rc ? rc : aitk_gemm_nt(b, stream_);
}

static int gemm_check(const AitkGemmArgs* a) {
  if (!a || a->M <= 0 || a->N <= 0 || a->K <= 0) return AITK_ERR_SHAPE;
  if ((a->K % 8) || (a->K2 % 8) || (a->N % 4)) return AITK_ERR_SHAPE;
  if ((a->lda % 8) || (a->ldb % 8) || (a->ldc % 4)) return AITK_ERR_ALIGN;
  if (a->b_scale_mode < 0 || a->b_scale_mode > 3) return AITK_ERR_ARG;
  if (a->b_scale_mode == 3 && (a->conv_mode || (a->K % 16) || (a->lda % 16) || (a->ldb % 16))) return AITK_ERR_ARG;
  if (a->K2 > 0 && (!a->A2 || !a->B2 || (a->lda2 % 8) || (a->ldb2 % 8))) return AITK_ERR_ARG;
  if ((a->flags & (AITK_EPI_BIAS | AITK_EPI_BIAS_ROW)) && !a->bias) return AITK_ERR_ARG;
  if ((a->flags & AITK_EPI_ADD_AUX) && !a->aux_in) return AITK_ERR_ARG;
  if ((a->flags & AITK_EPI_COL_SCALE) && (!a->col_scale || ((uintptr_t)a->col_scale & 15))) return AITK_ERR_ARG;
  if ((a->flags & AITK_EPI_GELU) && !a->aux_out) return AITK_ERR_ARG;  // GATE_RES: aux_out optional (only d_gate needs y)
  if ((a->flags & (AITK_EPI_DGELU | AITK_EPI_GATE_RES)) && !a->aux_in) return AITK_ERR_ARG;
  if ((a->flags & AITK_EPI_GATE_RES) && (!a->gate || a->gate_rows <= 0)) return AITK_ERR_ARG;
  if (a->flags & AITK_EPI_SPLIT_SLAB) {  // [P_hi ; P_lo] blocks -> [hi | lo | hi] slab of rank N / 2 <= 64, no other epilogue but the column scale
    if ((a->N % 32) || a->N > 128 || a->ldc < 3 * (a->N / 2) || a->c_seg_rows != 0 || (a->flags & ~(AITK_EPI_SPLIT_SLAB | AITK_EPI_COL_SCALE)))
      return AITK_ERR_ARG;
  }
  if (((uintptr_t)a->A | (uintptr_t)a->C) & 15) return AITK_ERR_ALIGN;
  if ((uintptr_t)a->B & (a->b_scale_mode ? 7 : 15)) return AITK_ERR_ALIGN;
  return AITK_OK;
}

extern "C" int aitk_gemm_nt(const AitkGemmArgs* a, aitk_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int chk = gemm_check(a);
  if (chk) return chk;
  hipStream_t st = stream;
  if (a->conv_mode) {
    const int c_kt = a->conv_t3d ? (a->conv_t3d & 255) : 1, c_ts = a->conv_t3d ? ((a->conv_t3d >> 8) & 255) : 1;
    const int c_ks = a->conv_t3d ? ((a->conv_t3d >> 16) & 255) : 3;
    if (!a->zero_page || a->conv_Cin <= 0 || (a->conv_Cin % 8) || a->K != c_kt * c_ks * c_ks * a->conv_Cin || a->a_seg_rows != 0 ||
        a->conv_stride <= 0 || a->conv_Wo <= 0 || a->conv_HoWo <= 0 || c_kt < 1 || c_kt > 3 || c_ts < 1 || c_ts > 2 || (c_ks != 1 && c_ks != 3) ||
        (a->conv_t3d >> 24))
      return AITK_ERR_ARG;
    static bool cattr = false;
    if (!cattr) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<1, 256, 256, 2, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      cattr = true;
    }
    const long t256 = (long)((a->M + 255) / 256) * ((a->N + 255) / 256);
    // big convolutions: the persistent 8-phase kernel (gemm8.hip, conv mode); stage_mode 5 keeps the 2-barrier kernel (same-box A/B)
    static long conv8_min_n = -1;  // AITK_CONV8_MIN_N: smallest Cout routed to the 8-phase kernel (A/B; a 256-wide tile is half empty at 128)
    if (conv8_min_n < 0) {
      const char* e = getenv("AITK_CONV8_MIN_N");
      conv8_min_n = (e && atol(e) > 0) ? atol(e) : 256;
    }
    if (a->stage_mode != 5 && a->stage_mode != 0 && ((a->tile_mode == 0 && a->N >= conv8_min_n && t256 >= big_tiles_min()) || a->stage_mode == 4) &&
        aitk_gemm8_try_launch(a, st) == AITK_OK) {
      AITK_LAUNCH_CHECK();
      return AITK_OK;
    }
    if (a->tile_mode == 2 || (a->tile_mode == 0 && a->N >= 256 && t256 >= 192)) {
      hipLaunchKernelGGL((gemm_nt_kernel<1, 256, 256, 2, 4, true>), dim3((unsigned)t256), dim3(512), 131072, st, *a);
    } else {
      const int tiles = ((a->M + 127) / 128) * ((a->N + 127) / 128);
      hipLaunchKernelGGL((gemm_nt_kernel<1, 128, 128, 2, 2, true>), dim3(tiles), dim3(256), 65536, st, *a);
    }
    AITK_LAUNCH_CHECK();
    return AITK_OK;
  }
  if (a->flags & AITK_EPI_EMIT_T) {  // the emitting epilogue exists on the persistent 8-phase kernel only: outside its contract the caller keeps aitk_lora_down
    if (aitk_gemm8_try_launch(a, st) != AITK_OK) return AITK_ERR_SHAPE;
    AITK_LAUNCH_CHECK();
    return AITK_OK;
  }
  AitkGemmArgs tmp = *a;
  if (a->b_scale_mode == 3) {  // W8A8 on the MX-scaled fp8 MFMA: the persistent 8-phase kernel is the only implementation (no fallback)
    if (aitk_gemm8_try_launch(a, st) != AITK_OK) return AITK_ERR_SHAPE;
    AITK_LAUNCH_CHECK();
    return AITK_OK;
  }
  if (a->b_scale_mode) {
    if (!a->b_scale || (a->ldb % 8) || (a->K % 16) || (a->K2 % 16)) return AITK_ERR_ARG;
    static bool f8attr = false;
    if (!f8attr) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<3, 256, 256, 2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      f8attr = true;
    }
    const long t256f = (long)((a->M + 255) / 256) * ((a->N + 255) / 256);
    if (a->tile_mode == 2 || (a->tile_mode == 0 && a->M >= 1024 && a->N >= 512 && t256f >= 192)) {
      hipLaunchKernelGGL((gemm_nt_kernel<3, 256, 256, 2, 4>), dim3((unsigned)t256f), dim3(512), 131072, st, *a);
    } else {
      const int tiles = ((a->M + 127) / 128) * ((a->N + 127) / 128);
      hipLaunchKernelGGL((gemm_nt_kernel<3, 128, 128, 2, 2>), dim3(tiles), dim3(256), 65536, st, *a);
    }
    AITK_LAUNCH_CHECK();
    return AITK_OK;
  }
  if (tmp.stage_mode >= 1 && ((a->K % 16) || (a->K2 % 16))) tmp.stage_mode = 0;  // LDS-DMA cannot zero-fill an 8-wide K tail
  a = &tmp;
  // tile choice: 256x256 when the problem fills the chip with big tiles (>= 1 full round of 256 CUs) or is overridden
  int big = a->tile_mode == 2 ? 1 : 0;
  if (a->tile_mode == 0) {
    const long t256 = (long)((a->M + 255) / 256) * ((a->N + 255) / 256);
    big = (a->M >= 1024 && a->N >= 512 && t256 >= big_tiles_min()) ? 1 : 0;
  }
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<0, 256, 256, 2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<1, 256, 256, 2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    attr_set = true;
  }
  // stage_mode: 0 VGPR-staged, 1 auto (8-phase persistent kernel when the problem is big, else LDS-DMA 2-barrier),
  //             4 force the 8-phase kernel, 5 force the 2-barrier LDS-DMA kernels (A/B reference)
  if ((a->stage_mode == 1 && big) || a->stage_mode == 4) {
    if (aitk_gemm8_try_launch(a, st) == AITK_OK) {
      AITK_LAUNCH_CHECK();
      return AITK_OK;
    }
  }
  if (a->stage_mode >= 4) tmp.stage_mode = 1;
  if (big) {
    const int tiles = ((a->M + 255) / 256) * ((a->N + 255) / 256);
    if (a->stage_mode == 1)
      hipLaunchKernelGGL((gemm_nt_kernel<1, 256, 256, 2, 4>), dim3(tiles), dim3(512), 131072, st, *a);
    else
      hipLaunchKernelGGL((gemm_nt_kernel<0, 256, 256, 2, 4>), dim3(tiles), dim3(512), 131072, st, *a);
  } else {
    const int tiles = ((a->M + 127) / 128) * ((a->N + 127) / 128);
    if (a->stage_mode == 1)
      hipLaunchKernelGGL((gemm_nt_kernel<1, 128, 128, 2, 2>), dim3(tiles), dim3(256), 65536, st, *a);
    else
      hipLaunchKernelGGL((gemm_nt_kernel<0, 128, 128, 2, 2>), dim3(tiles), dim3(256), 65536, st, *a);
  }
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}
