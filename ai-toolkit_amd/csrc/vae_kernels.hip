// VAE-encoder side kernels (gfx950), NHWC bf16 activations: GroupNorm(+SiLU), row softmax for the single-head mid-block
// attention (scores materialised by the GEMM: the 16384 x 16384 bf16 matrix is 512 MB of 288 GB), NCHW image -> NHWC
// with channels padded to 8, and the Gaussian latent sample + shift/scale.  Convolutions run on gemm_nt_kernel<CONV>.
//
// Reference being replaced: `self.vae.encode(images).latent_dist.sample()` and `scaling_factor * (latents - shift)`
// (toolkit/stable_diffusion_model.py:2533-2575); the layer bodies are diffusers' AutoencoderKL (oracle/vae_ref.py).
#include "common.h"
#include "aitk_args.h"

__device__ __forceinline__ void unpack8v(const uint4& v, float* f) {
  f[0] = bf2f(v.x & 0xffff); f[1] = bf2f(v.x >> 16);
  f[2] = bf2f(v.y & 0xffff); f[3] = bf2f(v.y >> 16);
  f[4] = bf2f(v.z & 0xffff); f[5] = bf2f(v.z >> 16);
  f[6] = bf2f(v.w & 0xffff); f[7] = bf2f(v.w >> 16);
}
__device__ __forceinline__ uint4 pack8v(const float* f) {
  uint4 v;
  v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]);
  v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
  return v;
}

// ---------------------------------------------------------------- GroupNorm statistics
// x [B, HW, C]; per-channel (sum, sumsq) partials over row chunks of GN_ROWS: partial [B][nchunk][2][C] fp32.
#define GN_ROWS 64
__global__ __launch_bounds__(256) void gn_stats_kernel(AitkGroupNormArgs p) {
  const int b = blockIdx.y;
  const int r0 = blockIdx.x * GN_ROWS;
  const int nrows = min(GN_ROWS, p.HW - r0);
  const int nch = p.C / 8;
  for (int ch = threadIdx.x; ch < nch; ch += 256) {
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    const bf16_t* src = p.x + ((long)b * p.HW + r0) * p.ldx + ch * 8;
    for (int r = 0; r < nrows; ++r) {
      float v[8];
      unpack8v(*reinterpret_cast<const uint4*>(src + (long)r * p.ldx), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] += v[e]; q[e] += v[e] * v[e]; }
    }
    float* pp = p.partial + (((long)b * gridDim.x + blockIdx.x) * 2) * p.C + ch * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { pp[e] = s[e]; pp[p.C + e] = q[e]; }
  }
}
// one wave per (b, group): lanes stride over the (chunk, channel) partials, fp64 accumulation, wave reduction -> mean, rstd
// (one THREAD per group walked nchunk * C/G partials serially: 175 us per call at 128x128 x 320 channels)
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__global__ __launch_bounds__(64) void gn_finish_kernel(AitkGroupNormArgs p, int nchunk) {
  const int idx = blockIdx.x;
  const int b = idx / p.G, g = idx - b * p.G;
  const int cg = p.C / p.G;
  double s = 0.0, q = 0.0;
  for (int i = threadIdx.x; i < nchunk * cg; i += 64) {
    const int k = i / cg, c = i - k * cg;
    const float* pp = p.partial + (((long)b * nchunk + k) * 2) * p.C + g * cg + c;
    s += pp[0];
    q += pp[p.C];
  }
  s = wave_sum_f64(s);
  q = wave_sum_f64(q);
  if (threadIdx.x != 0) return;
  const double n = (double)p.HW * cg;
  const double mean = s / n;
  const double var = fmax(q / n - mean * mean, 0.0);
  p.stats[2 * idx] = (float)mean;
  p.stats[2 * idx + 1] = (float)(1.0 / sqrt(var + (double)p.eps));
  if (p.stats_out) {  // kept by the caller for aitk_groupnorm_bwd
    p.stats_out[2 * idx] = p.stats[2 * idx];
    p.stats_out[2 * idx + 1] = p.stats[2 * idx + 1];
  }
}
// y = (x - mean) * rstd * gamma + beta, optionally SiLU
__global__ __launch_bounds__(256) void gn_apply_kernel(AitkGroupNormArgs p) {
  const long chunk = (long)blockIdx.x * 256 + threadIdx.x;
  const int nch = p.C / 8;
  const long total = (long)p.B * p.HW * nch;
  if (chunk >= total) return;
  const int ch = (int)(chunk % nch);
  const long row = chunk / nch;
  const int b = (int)(row / p.HW);
  const int cg = p.C / p.G;
  float v[8], ga[8], be[8];
  unpack8v(*reinterpret_cast<const uint4*>(p.x + row * p.ldx + ch * 8), v);
  unpack8v(*reinterpret_cast<const uint4*>(p.gamma + ch * 8), ga);
  unpack8v(*reinterpret_cast<const uint4*>(p.beta + ch * 8), be);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int g = (ch * 8 + e) / cg;
    const float mean = p.stats[2 * (b * p.G + g)], rstd = p.stats[2 * (b * p.G + g) + 1];
    float y = (v[e] - mean) * rstd * ga[e] + be[e];
    if (p.silu) y = y / (1.0f + expf(-y));
    v[e] = y;
  }
  *reinterpret_cast<uint4*>(p.y + row * p.ldy + ch * 8) = pack8v(v);
}
extern "C" int64_t aitk_groupnorm_workspace_bytes(int32_t B, int32_t HW, int32_t C, int32_t G) {
  const int64_t nchunk = (HW + GN_ROWS - 1) / GN_ROWS;
  return (B * nchunk * 2 * (int64_t)C + 2 * (int64_t)B * G) * 4;
}
extern "C" int aitk_groupnorm(const AitkGroupNormArgs* a, aitk_stream_t stream) {
  if (!a || a->B <= 0 || a->HW <= 0 || a->C <= 0 || a->G <= 0 || (a->C % 8) || (a->C % a->G)) return AITK_ERR_SHAPE;
  if ((a->ldx % 8) || (a->ldy % 8) || !a->partial) return AITK_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const int nchunk = (a->HW + GN_ROWS - 1) / GN_ROWS;
  AitkGroupNormArgs args = *a;
  args.stats = a->partial + (long)a->B * nchunk * 2 * a->C;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunk, a->B), dim3(256), 0, s, args);
  AITK_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_finish_kernel, dim3(a->B * a->G), dim3(64), 0, s, args, nchunk);
  AITK_LAUNCH_CHECK();
  const long total = (long)a->B * a->HW * (a->C / 8);
  hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, args);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// ---------------------------------------------------------------- row softmax (in place), x [rows, n] bf16, scaled
__global__ __launch_bounds__(256) void softmax_rows_kernel(bf16_t* x, long ld, int n, float scale) {
  __shared__ float red[4];
  bf16_t* row = x + (long)blockIdx.x * ld;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float m = -INFINITY;
  for (int c = threadIdx.x * 8; c < n; c += 256 * 8) {
    float v[8];
    unpack8v(*reinterpret_cast<const uint4*>(row + c), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, v[e]);
  }
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale;
  __syncthreads();
  float s = 0.f;
  for (int c = threadIdx.x * 8; c < n; c += 256 * 8) {
    float v[8];
    unpack8v(*reinterpret_cast<const uint4*>(row + c), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += expf(v[e] * scale - m);
  }
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
  for (int c = threadIdx.x * 8; c < n; c += 256 * 8) {
    float v[8];
    unpack8v(*reinterpret_cast<const uint4*>(row + c), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = expf(v[e] * scale - m) * inv;
    *reinterpret_cast<uint4*>(row + c) = pack8v(v);
  }
}
extern "C" int aitk_softmax_rows(aitk_bf16* x, int64_t ld, int32_t rows, int32_t n, float scale, aitk_stream_t stream) {
  if (!x || rows <= 0 || n <= 0 || (n % 8) || (ld % 8)) return AITK_ERR_SHAPE;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, (long)ld, n, scale);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// ---------------------------------------------------------------- image [B,3,H,W] fp32 in [-1,1] -> NHWC bf16, C padded to 8
__global__ void image_to_nhwc8_kernel(const float* img, bf16_t* out, int B, int H, int W) {
  const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * H * W;
  if (pix >= total) return;
  const long hw = (long)H * W;
  const int b = (int)(pix / hw);
  const long r = pix - (long)b * hw;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  v[0] = img[((long)b * 3 + 0) * hw + r];
  v[1] = img[((long)b * 3 + 1) * hw + r];
  v[2] = img[((long)b * 3 + 2) * hw + r];
  *reinterpret_cast<uint4*>(out + pix * 8) = pack8v(v);
}
extern "C" int aitk_image_to_nhwc8(const float* img, aitk_bf16* out, int32_t B, int32_t H, int32_t W, aitk_stream_t stream) {
  if (!img || !out || B <= 0 || H <= 0 || W <= 0) return AITK_ERR_SHAPE;
  const long total = (long)B * H * W;
  hipLaunchKernelGGL(image_to_nhwc8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img, out, B, H, W);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// ---------------------------------------------------------------- bilinear resize + NHWC conversion (Wan21.encode_images resizes inputs whose
// sides are not multiples of 8 with F.interpolate(mode="bilinear", align_corners=False) in the VAE dtype, toolkit/models/wan21/wan21.py:652-657):
// src = (in / out) * (dst + 0.5) - 0.5 clamped at 0, neighbours i0 = floor(src), i1 = min(i0 + 1, in - 1), weights in fp32 on the
// bf16-rounded pixels, summed as  h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11)  like ATen's upsample_bilinear2d kernel.
__global__ void image_resize_to_nhwc8_kernel(const float* img, bf16_t* out, int B, int Hs, int Ws, int Hd, int Wd, float sh, float sw) {
  const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * Hd * Wd;
  if (pix >= total) return;
  const int x = (int)(pix % Wd);
  const int y = (int)((pix / Wd) % Hd);
  const int b = (int)(pix / ((long)Wd * Hd));
  const float fy = fmaxf(sh * ((float)y + 0.5f) - 0.5f, 0.f), fx = fmaxf(sw * ((float)x + 0.5f) - 0.5f, 0.f);
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = min(y0 + 1, Hs - 1), x1 = min(x0 + 1, Ws - 1);
  const float h1 = fy - (float)y0, w1 = fx - (float)x0;
  const float h0 = 1.0f - h1, w0 = 1.0f - w1;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* pl = img + ((long)b * 3 + c) * Hs * Ws;
    const float v00 = bfround(pl[(long)y0 * Ws + x0]), v01 = bfround(pl[(long)y0 * Ws + x1]);
    const float v10 = bfround(pl[(long)y1 * Ws + x0]), v11 = bfround(pl[(long)y1 * Ws + x1]);
    v[c] = h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11);
  }
  *reinterpret_cast<uint4*>(out + pix * 8) = pack8v(v);
}
extern "C" int aitk_image_resize_to_nhwc8(const float* img, aitk_bf16* out, int32_t B, int32_t Hs, int32_t Ws, int32_t Hd, int32_t Wd,
                                          aitk_stream_t stream) {
  if (!img || !out || B <= 0 || Hs <= 0 || Ws <= 0 || Hd <= 0 || Wd <= 0) return AITK_ERR_SHAPE;
  const long total = (long)B * Hd * Wd;
  hipLaunchKernelGGL(image_resize_to_nhwc8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img, out, B, Hs, Ws,
                     Hd, Wd, (float)Hs / (float)Hd, (float)Ws / (float)Wd);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// ---------------------------------------------------------------- latent sample: moments NHWC [B*hw, 2L] -> latents NCHW [B, L, h, w]
// z = mean + exp(0.5 clamp(logvar,-30,20)) * eps ; out = scale * (z - shift)     (eps NCHW fp32, like randn_tensor(mean.shape))
__global__ void latent_sample_kernel(const bf16_t* mom, long ldm, const float* eps, bf16_t* out, int B, int L, int hw,
                                     float scale, float shift) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * L * hw;
  if (idx >= total) return;
  const int r = (int)(idx % hw);
  const int c = (int)((idx / hw) % L);
  const int b = (int)(idx / ((long)hw * L));
  const bf16_t* mrow = mom + ((long)b * hw + r) * ldm;
  const float mean = bf2f(mrow[c]);
  const float logvar = fminf(fmaxf(bf2f(mrow[L + c]), -30.0f), 20.0f);
  const float z = mean + expf(0.5f * logvar) * eps[idx];
  out[idx] = f2bf(scale * (z - shift));
}
extern "C" int aitk_latent_sample(const aitk_bf16* moments, int64_t ldm, const float* eps, aitk_bf16* out, int32_t B, int32_t L,
                                  int32_t hw, float scale, float shift, aitk_stream_t stream) {
  if (!moments || !eps || !out || B <= 0 || L <= 0 || hw <= 0) return AITK_ERR_SHAPE;
  const long total = (long)B * L * hw;
  hipLaunchKernelGGL(latent_sample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, moments, (long)ldm,
                     eps, out, B, L, hw, scale, shift);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// ---------------------------------------------------------------- Wan2.1 video VAE (AutoencoderKLWan encoder, ai-toolkit_amd/wan_vae.py)
// the same sample with the per-channel affine of Wan21.encode_images (toolkit/models/wan21/wan21.py:661-670)
__global__ void latent_sample_affine_kernel(const bf16_t* mom, long ldm, const float* eps, bf16_t* out, int B, int L, int hw,
                                            const float* ch_shift, const float* ch_scale) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * L * hw;
  if (idx >= total) return;
  const int r = (int)(idx % hw);
  const int c = (int)((idx / hw) % L);
  const int b = (int)(idx / ((long)hw * L));
  const bf16_t* mrow = mom + ((long)b * hw + r) * ldm;
  const float mean = bf2f(mrow[c]);
  const float logvar = fminf(fmaxf(bf2f(mrow[L + c]), -30.0f), 20.0f);
  const float z = mean + expf(0.5f * logvar) * eps[idx];
  out[idx] = f2bf(ch_scale[c] * (z - ch_shift[c]));
}
extern "C" int aitk_latent_sample_affine(const aitk_bf16* moments, int64_t ldm, const float* eps, aitk_bf16* out, int32_t B, int32_t L,
                                         int32_t hw, const float* ch_shift, const float* ch_scale, aitk_stream_t stream) {
  if (!moments || !eps || !out || !ch_shift || !ch_scale || B <= 0 || L <= 0 || hw <= 0) return AITK_ERR_SHAPE;
  const long total = (long)B * L * hw;
  hipLaunchKernelGGL(latent_sample_affine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, moments,
                     (long)ldm, eps, out, B, L, hw, ch_shift, ch_scale);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// WanRMS_norm over the channel axis of NHWC rows: y = x / max(||x||, eps) * sqrt(C) * gamma (+ SiLU).  HBM-bound (4 B / element):
// G lanes per row (16 B each; G = 16 / 32 / 64 for C <= 128 / 256 / more), up to four 16-B chunks per lane, group reduction by
// shuffles.  Rows are independent, so x and y may alias.
template <int G, int NJ>
__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(const bf16_t* x, long ldx, bf16_t* y, long ldy, const bf16_t* gamma, long M,
                                                          int C, float eps, int silu) {
  constexpr int RPB = 256 / G;
  const long row = (long)blockIdx.x * RPB + threadIdx.x / G;
  const int l = threadIdx.x % G;
  if (row >= M) return;  // whole lane groups leave together (the shuffles below stay inside a group)
  float v[NJ][8];
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = (l + j * G) * 8;
    if (c < C) {
      unpack8v(*reinterpret_cast<const uint4*>(x + row * ldx + c), v[j]);
#pragma unroll
      for (int e = 0; e < 8; ++e) ss += v[j][e] * v[j][e];
    }
  }
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
  const float k = sqrtf((float)C) / fmaxf(sqrtf(ss), eps);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = (l + j * G) * 8;
    if (c < C) {
      float g[8], o[8];
      unpack8v(*reinterpret_cast<const uint4*>(gamma + c), g);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = v[j][e] * k * g[e];
        o[e] = silu ? t / (1.0f + __expf(-t)) : t;
      }
      *reinterpret_cast<uint4*>(y + row * ldy + c) = pack8v(o);
    }
  }
}
extern "C" int aitk_rmsnorm_rows(const aitk_bf16* x, int64_t ldx, aitk_bf16* y, int64_t ldy, const aitk_bf16* gamma, int64_t M, int32_t C,
                                 float eps, int32_t silu, aitk_stream_t stream) {
  if (!x || !y || !gamma || M <= 0 || C <= 0 || (C % 8) || C > 2048 || (ldx % 8) || (ldy % 8)) return AITK_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (C <= 128) {
    hipLaunchKernelGGL((rmsnorm_rows_kernel<16, 1>), dim3((unsigned)((M + 15) / 16)), dim3(256), 0, s, x, (long)ldx, y, (long)ldy, gamma,
                       (long)M, C, eps, silu);
  } else if (C <= 256) {
    hipLaunchKernelGGL((rmsnorm_rows_kernel<32, 1>), dim3((unsigned)((M + 7) / 8)), dim3(256), 0, s, x, (long)ldx, y, (long)ldy, gamma,
                       (long)M, C, eps, silu);
  } else {
    hipLaunchKernelGGL((rmsnorm_rows_kernel<64, 4>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, x, (long)ldx, y, (long)ldy, gamma,
                       (long)M, C, eps, silu);
  }
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}
