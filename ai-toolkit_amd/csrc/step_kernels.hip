// Step-level kernels around the DiT (gfx950): small-batch projections (adaLN / embedders), flow-matching noise mix +
// 2x2 patchify, MSE loss + gradient, and the fused clip -> AdamW -> EMA -> bf16-shadow update over the flat LoRA
// parameter arena.  All HBM-bound; 16-B accesses, wave64 shuffles, deterministic two-stage reductions.
//
// Reference behaviour being replaced:
//   add_noise ............ toolkit/samplers/custom_flowmatch_sampler.py:91-102
//   pack / unpack ........ toolkit/stable_diffusion_model.py:2157-2163, 2210-2219
//   loss ................. extensions_built_in/sd_trainer/SDTrainer.py:644-646, 916, 987-990, 1013
//   clip / AdamW / EMA ... SDTrainer.py:2278-2293; toolkit/optimizer.py:78-79 (torch.optim.AdamW, eps=1e-6);
//                          toolkit/ema.py:116-152
#include "common.h"
#include "aitk_args.h"

// ------------------------------------------------------------------------------------------------ small-M GEMV
// out[Bm,N] (+)= X[Bm,K] W[N,K]^T + bias[N] + T[Bm,R] Bl[N,R]^T ; Bm <= 8.  One 16-lane group per output column,
// W rows streamed once with 16-B loads (weights dominate the bytes), X staged in LDS.
#define GEMV_MAXB 8
__global__ __launch_bounds__(256) void gemv_nt_kernel(AitkGemvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);  // [Bm][K]
  const int tid = threadIdx.x;
  const int kch = p.K / 8;
  for (int q = tid; q < p.Bm * kch; q += 256) {
    const int bb = q / kch, c = q - bb * kch;
    *reinterpret_cast<uint4*>(xs + (long)bb * p.K + c * 8) = *reinterpret_cast<const uint4*>(p.X + (long)bb * p.ldx + c * 8);
  }
  __syncthreads();
  const int grp = tid >> 4, sub = tid & 15;
  const int ncol_per_block = 16 * p.cols_per_group;
  for (int ci = 0; ci < p.cols_per_group; ++ci) {
    const int n = blockIdx.x * ncol_per_block + ci * 16 + grp;
    if (n >= p.N) continue;  // whole 16-lane group leaves together
    const bf16_t* wrow = p.W + (long)n * p.ldw;
    float acc[GEMV_MAXB];
#pragma unroll
    for (int bb = 0; bb < GEMV_MAXB; ++bb) acc[bb] = 0.f;
    for (int c0 = sub; c0 < kch; c0 += 64) {  // four 16-B weight loads in flight per lane before the first FMA
      uint4 wq[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + 16 * u;
        wq[u] = c < kch ? *reinterpret_cast<const uint4*>(wrow + c * 8) : uint4{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + 16 * u;
        if (c >= kch) break;
        const uint4 wv = wq[u];
        float w[8];
        w[0] = bf2f(wv.x & 0xffff); w[1] = bf2f(wv.x >> 16); w[2] = bf2f(wv.y & 0xffff); w[3] = bf2f(wv.y >> 16);
        w[4] = bf2f(wv.z & 0xffff); w[5] = bf2f(wv.z >> 16); w[6] = bf2f(wv.w & 0xffff); w[7] = bf2f(wv.w >> 16);
#pragma unroll
        for (int bb = 0; bb < GEMV_MAXB; ++bb) {
          if (bb < p.Bm) {
            const uint4 xv = *reinterpret_cast<const uint4*>(xs + (long)bb * p.K + c * 8);
            acc[bb] += w[0] * bf2f(xv.x & 0xffff) + w[1] * bf2f(xv.x >> 16) + w[2] * bf2f(xv.y & 0xffff) + w[3] * bf2f(xv.y >> 16) +
                       w[4] * bf2f(xv.z & 0xffff) + w[5] * bf2f(xv.z >> 16) + w[6] * bf2f(xv.w & 0xffff) + w[7] * bf2f(xv.w >> 16);
          }
        }
      }
    }
#pragma unroll
    for (int bb = 0; bb < GEMV_MAXB; ++bb) {
      float v = acc[bb];
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 4, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 1, 64);
      acc[bb] = v;
    }
    if (sub < p.Bm) {
      const int bb = sub;
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < GEMV_MAXB; ++k)
        if (k == bb) v = acc[k];
      if (p.R > 0) {
        float lv = 0.f;
        for (int r = 0; r < p.R; ++r) lv += bf2f(p.T[(long)bb * p.ldt + r]) * bf2f(p.Bl[(long)n * p.ldbl + r]);
        v += lv;
      }
      if (p.col_scale) v *= p.col_scale[n];  // DoRA: c * (x W^T + T B^T) + b
      if (p.bias) v += bf2f(p.bias[n]);
      bf16_t* o = p.out + (long)bb * p.ldo + n;
      if (p.accumulate) v += bf2f(*o);
      *o = f2bf(v);
    }
  }
}

extern "C" int aitk_gemv_nt(const AitkGemvArgs* a, aitk_stream_t stream) {
  if (!a || a->Bm <= 0 || a->Bm > GEMV_MAXB || a->N <= 0 || a->K <= 0 || (a->K % 8)) return AITK_ERR_SHAPE;
  if ((a->ldx % 8) || (a->ldw % 8)) return AITK_ERR_ALIGN;
  if (a->R > 0 && (!a->T || !a->Bl)) return AITK_ERR_ARG;
  const size_t lds = (size_t)a->Bm * a->K * 2;
  if (lds > 64 * 1024) return AITK_ERR_SHAPE;
  AitkGemvArgs args = *a;
  args.cols_per_group = 1;  // 16 columns per workgroup: N / 16 workgroups (> 4 per CU for the adaLN projections) keep enough loads in flight
  const int ncol = 16 * args.cols_per_group;
  hipLaunchKernelGGL(gemv_nt_kernel, dim3((a->N + ncol - 1) / ncol), dim3(256), lds, (hipStream_t)stream, args);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// ------------------------------------------------------------------------------------------------ noise mix + patchify
// latents/noise [B, C, H, W] bf16; t [B] fp32 in [0,1000].
//   noisy[b, (h/2)(w/2), c*4 + 2*ph + pw] = bf16((1 - t/1000) x0 + (t/1000) eps)          (fp32 math like the reference)
//   target (same packing)                 = bf16(eps - x0)
__global__ void flow_noise_pack_kernel(AitkNoisePackArgs p) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)p.B * p.C * p.H * p.W;
  if (idx >= total) return;
  const int w = (int)(idx % p.W);
  const int hh = (int)((idx / p.W) % p.H);
  const int c = (int)((idx / ((long)p.W * p.H)) % p.C);
  const int b = (int)(idx / ((long)p.W * p.H * p.C));
  const float x0 = bf2f(p.latents[idx]);
  const float e = bf2f(p.noise[idx]);
  const float t01 = p.t[b] / 1000.0f;
  const long tok = (long)(hh >> 1) * (p.W >> 1) + (w >> 1);
  const int ch = c * 4 + ((hh & 1) << 1) + (w & 1);
  const long o = ((long)b * (p.H >> 1) * (p.W >> 1) + tok) * (p.C * 4) + ch;
  p.noisy[o] = f2bf((1.0f - t01) * x0 + t01 * e);
  p.target[o] = f2bf(e - x0);
}
extern "C" int aitk_flow_noise_pack(const AitkNoisePackArgs* a, aitk_stream_t stream) {
  if (!a || a->B <= 0 || a->C <= 0 || a->H <= 0 || a->W <= 0 || (a->H & 1) || (a->W & 1)) return AITK_ERR_SHAPE;
  const long total = (long)a->B * a->C * a->H * a->W;
  hipLaunchKernelGGL(flow_noise_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// ------------------------------------------------------------------------------------------------ MSE loss + gradient
// per-sample loss_b = mean_j l(pred - target) ; loss = mean_b (w_b * loss_b) ; dpred = l'(pred - target) w_b / (n B)
//   LT 0 (mse): l = d^2, l' = 2d     LT 1 (mae): l = |d|, l' = sign(d)     LT 2 (pseudo_huber): l = sqrt(d^2 + c^2) - c, l' = d / sqrt(d^2 + c^2)
#define LOSS_CHUNK 8192
template <int LT>
__global__ __launch_bounds__(256) void mse_partial_kernel(AitkMseArgs p, int nchunk) {
  __shared__ float red[4];
  const int b = blockIdx.y;
  const long base = (long)b * p.n_per_sample + (long)blockIdx.x * LOSS_CHUNK;
  const long end = min((long)(b + 1) * p.n_per_sample, base + LOSS_CHUNK);
  const float wb = p.weight ? p.weight[b] : 1.0f;
  const float gs = (LT == 0 ? 2.0f : 1.0f) * wb / ((float)p.n_per_sample * (float)p.B);
  const float c2 = p.huber_c * p.huber_c;
  float acc = 0.f;
  for (long i = base + threadIdx.x * 8; i < end; i += 256 * 8) {
    const uint4 pv = *reinterpret_cast<const uint4*>(p.pred + i);
    const uint4 tv = *reinterpret_cast<const uint4*>(p.target + i);
    float d[8];
    d[0] = bf2f(pv.x & 0xffff) - bf2f(tv.x & 0xffff); d[1] = bf2f(pv.x >> 16) - bf2f(tv.x >> 16);
    d[2] = bf2f(pv.y & 0xffff) - bf2f(tv.y & 0xffff); d[3] = bf2f(pv.y >> 16) - bf2f(tv.y >> 16);
    d[4] = bf2f(pv.z & 0xffff) - bf2f(tv.z & 0xffff); d[5] = bf2f(pv.z >> 16) - bf2f(tv.z >> 16);
    d[6] = bf2f(pv.w & 0xffff) - bf2f(tv.w & 0xffff); d[7] = bf2f(pv.w >> 16) - bf2f(tv.w >> 16);
    float mk[4] = {1.f, 1.f, 1.f, 1.f};
    if (p.mask) {  // 8 consecutive features of one token: patch positions 0,1,2,3,0,1,2,3
      const long j = i - (long)b * p.n_per_sample;
      const f32x4_t m4 = *reinterpret_cast<const f32x4_t*>(p.mask + ((long)b * (p.n_per_sample / p.feat) + j / p.feat) * 4);
      mk[0] = m4[0]; mk[1] = m4[1]; mk[2] = m4[2]; mk[3] = m4[3];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (LT == 0) {
        acc += mk[e & 3] * d[e] * d[e];
        d[e] *= mk[e & 3];
      } else if (LT == 1) {
        acc += mk[e & 3] * fabsf(d[e]);
        d[e] = mk[e & 3] * (d[e] > 0.f ? 1.0f : (d[e] < 0.f ? -1.0f : 0.f));
      } else {
        const float r = sqrtf(d[e] * d[e] + c2);
        acc += mk[e & 3] * (r - p.huber_c);
        d[e] = mk[e & 3] * d[e] / r;
      }
    }
    uint4 g;
    g.x = pack2bf(d[0] * gs, d[1] * gs); g.y = pack2bf(d[2] * gs, d[3] * gs);
    g.z = pack2bf(d[4] * gs, d[5] * gs); g.w = pack2bf(d[6] * gs, d[7] * gs);
    *reinterpret_cast<uint4*>(p.dpred + i) = g;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) p.partial[(long)b * nchunk + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void mse_finish_kernel(AitkMseArgs p, int nchunk) {
  // one thread per sample, then thread 0 averages: tiny
  const int b = threadIdx.x;
  __shared__ float ls[64];
  float s = 0.f;
  if (b < p.B) {
    for (int k = 0; k < nchunk; ++k) s += p.partial[(long)b * nchunk + k];
    s /= (float)p.n_per_sample;
    p.loss_per_sample[b] = s;
    ls[b] = s * (p.weight ? p.weight[b] : 1.0f);
  }
  __syncthreads();
  if (b == 0) {
    float t = 0.f;
    for (int k = 0; k < p.B; ++k) t += ls[k];
    t /= (float)p.B;
    if (p.guard) {
      // SDTrainer.py:1049-1050: loss = clamp(loss, max=max_loss) — above the bound the clamp's derivative is 0: autograd still runs and
      //   every parameter receives a ZERO gradient, so the optimizer steps (weight decay, moment decay, step count) on g = 0;
      // SDTrainer.py:2221-2224: a non-finite loss is replaced by a fresh zero WITHOUT a graph: backward reaches no parameter, .grad stays
      //   None and torch.optim.AdamW skips the parameters altogether.
      // Both zero this micro-batch's dpred (guard[6]); only the non-finite kind counts toward the all-gated skip of the optimizer launch:
      // guard[0] = micro-batches of the current step that left no gradient at all (the optimizer reads and clears it), [1] / [2] running totals.
      int gate = 0;
      if (!isfinite(t)) { t = 0.f; gate = 1; p.guard[1] += 1; p.guard[0] += 1; }
      else if (p.max_loss > 0.f && t > p.max_loss) { t = p.max_loss; gate = 1; p.guard[2] += 1; }
      p.guard[6] = gate;
    }
    p.loss[0] = t;
  }
}
// zeroes dpred when the finish kernel gated this micro-batch (every block reads one flag; 1 launch, no host round trip)
__global__ __launch_bounds__(256) void mse_gate_kernel(AitkMseArgs p, long n16) {
  if (p.guard[6] == 0) return;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) reinterpret_cast<uint4*>(p.dpred)[i] = make_uint4(0u, 0u, 0u, 0u);
}
extern "C" int64_t aitk_mse_workspace_bytes(int32_t B, int64_t n_per_sample) {
  return (int64_t)B * ((n_per_sample + LOSS_CHUNK - 1) / LOSS_CHUNK) * 4;
}
extern "C" int aitk_mse_loss_grad(const AitkMseArgs* a, aitk_stream_t stream) {
  if (!a || a->B <= 0 || a->B > 64 || a->n_per_sample <= 0 || (a->n_per_sample % 8)) return AITK_ERR_SHAPE;
  if (!a->pred || !a->target || !a->dpred || !a->partial || !a->loss || !a->loss_per_sample) return AITK_ERR_ARG;
  if (a->mask && (a->feat <= 0 || (a->feat % 8) || (a->n_per_sample % a->feat))) return AITK_ERR_ARG;
  if (a->loss_type < AITK_LOSS_MSE || a->loss_type > AITK_LOSS_PSEUDO_HUBER || a->huber_c < 0.f) return AITK_ERR_ARG;
  const int nchunk = (int)((a->n_per_sample + LOSS_CHUNK - 1) / LOSS_CHUNK);
  AitkMseArgs args = *a;
  if (args.loss_type == AITK_LOSS_PSEUDO_HUBER && args.huber_c == 0.f) args.huber_c = 0.01f;  // SDTrainer.py:905
  const dim3 grid(nchunk, a->B);
  if (args.loss_type == AITK_LOSS_MAE) hipLaunchKernelGGL(mse_partial_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, args, nchunk);
  else if (args.loss_type == AITK_LOSS_PSEUDO_HUBER) hipLaunchKernelGGL(mse_partial_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, args, nchunk);
  else hipLaunchKernelGGL(mse_partial_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, args, nchunk);
  AITK_LAUNCH_CHECK();
  hipLaunchKernelGGL(mse_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, *a, nchunk);
  AITK_LAUNCH_CHECK();
  if (a->guard) {
    if (reinterpret_cast<uintptr_t>(a->dpred) & 15) return AITK_ERR_ALIGN;  // the gate clears dpred in 16-byte stores (n_per_sample % 8 == 0 is checked above)
    const long n16 = (long)a->B * a->n_per_sample / 8;
    hipLaunchKernelGGL(mse_gate_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a, n16);
    AITK_LAUNCH_CHECK();
  }
  return AITK_OK;
}

// ------------------------------------------------------------------------------------------------ optimizer
// flat fp32 arenas p, g, m, v (, ema).  Stage 1: per-block sum of squares.  Stage 2: every block re-reduces the (few)
// partials -> total norm -> clip coefficient (torch.nn.utils.clip_grad_norm_: coef = min(1, max_norm/(norm+1e-6)))
// -> AdamW (decoupled weight decay, bias correction as torch.optim.AdamW) -> EMA (s -= (1-d)(s-p)).
#define OPT_BLOCK_ELEMS 4096
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* g, long n, float* partial) {
  __shared__ float red[4];
  const long base = (long)blockIdx.x * OPT_BLOCK_ELEMS;
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < OPT_BLOCK_ELEMS / (256 * 4); ++i) {
    const long j = base + (long)(i * 256 + threadIdx.x) * 4;
    if (j + 3 < n) {
      const float4 v = *reinterpret_cast<const float4*>(g + j);
      acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    } else {
      for (long k = j; k < n; ++k) acc += g[k] * g[k];
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// second-level reduce so the update kernel only sums <= 1024 values
__global__ __launch_bounds__(256) void sumsq_level2_kernel(const float* partial, int n1, float* out2) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = blockIdx.x * 1024 + threadIdx.x; i < min(n1, (blockIdx.x + 1) * 1024); i += 256) acc += partial[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out2[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// One thread decides the step: total norm -> clip coefficient; with a guard buffer also whether the update happens at all and which bias
// corrections apply.  ctl = {coef * grad_scale, skip, bias_correction1, bias_correction2_sqrt} (the tail of the norm workspace).
//   skip <=> the gradient norm is not finite (a NaN / Inf batch reached the arena), or every micro-batch of the step was gated by the loss
//   guard (non-finite loss / max_loss): the reference then has no gradient on any parameter and torch.optim.AdamW skips them all — no weight
//   decay, no moment decay, no step count (SDTrainer.py:2221-2224 + torch/optim/adamw.py "if p.grad is None: continue").  The bias
//   corrections follow the number of APPLIED steps, kept on the device (guard[3]): no host sync, and a skipped step does not advance them.
__global__ void adamw_decide_kernel(AitkAdamWArgs p, int n2, float* ctl) {
  float tot = 0.f;
  for (int i = 0; i < n2; ++i) tot += p.norm_partial2[i];
  const float norm = sqrtf(tot) * fabsf(p.grad_scale);  // norm of the scaled (e.g. rank-averaged) gradient
  float coef = 1.0f;
  if (p.max_norm > 0.f) coef = fminf(1.0f, p.max_norm / (norm + 1e-6f));
  float bc1 = p.bias_correction1, bc2s = p.bias_correction2_sqrt;
  int skip = 0;
  if (p.guard) {
    skip = !isfinite(norm) || (p.n_micro > 0 && p.guard[0] >= p.n_micro);
    p.guard[0] = 0;
    p.guard[5] = skip;
    if (skip) p.guard[4] += 1;
    else p.guard[3] += 1;
    const double step = (double)(p.guard[3] + (skip ? 1 : 0));  // the step this update is (or would have been)
    bc1 = (float)(1.0 - pow(p.beta1_d, step));  // the doubles the host's `1 - beta ** step` is made of (float(beta) would not reproduce them)
    bc2s = (float)sqrt(1.0 - pow(p.beta2_d, step));
  }
  ctl[0] = coef * p.grad_scale;
  ctl[1] = skip ? 1.0f : 0.0f;
  ctl[2] = bc1;
  ctl[3] = bc2s;
  // 1 - beta as torch forms it: in double from the Python float, THEN rounded to fp32 (lerp weight / addcmul value).  1.0f - 0.999f is
  // 1.29e-5 off that value — a systematic scale on v, i.e. 6.5e-6 on every update (tools/gpu_trainer_fusion_diag.py, round 6)
  ctl[4] = p.beta1_d != 0.0 ? (float)(1.0 - p.beta1_d) : 1.0f - p.beta1;
  ctl[5] = p.beta2_d != 0.0 ? (float)(1.0 - p.beta2_d) : 1.0f - p.beta2;
  if (p.norm_out) p.norm_out[0] = norm;
}

__global__ __launch_bounds__(256) void adamw_ema_kernel(AitkAdamWArgs p, const float* ctl) {
  const float coef = ctl[0];
  const bool skip = ctl[1] != 0.f;
  const float bc1 = ctl[2], bc2s = ctl[3], omb1 = ctl[4], omb2 = ctl[5];
  const long base = (long)blockIdx.x * OPT_BLOCK_ELEMS;
#pragma unroll
  for (int i = 0; i < OPT_BLOCK_ELEMS / 256; ++i) {
    const long j = base + i * 256 + threadIdx.x;
    if (j < p.n) {
      float w = p.p[j];
      if (!skip) {
        const float g = p.g[j] * coef;
        w -= p.lr * p.weight_decay * w;
        const float m = p.beta1 * p.m[j] + omb1 * g;
        const float v = p.beta2 * p.v[j] + omb2 * g * g;
        const float denom = sqrtf(v) / bc2s + p.eps;
        w -= (p.lr / bc1) * (m / denom);
        p.m[j] = m;
        p.v[j] = v;
      } else if (!p.ema) {
        continue;  // skipped step without EMA: nothing is touched
      }
      if (p.ema) {  // on a skipped step too: the reference's ema.update() runs after every train-loop iteration (SDTrainer.py:2291-2293)  // toolkit/ema.py:135-143: tmp = (1-d)(s - p); s -= tmp; p += 10 tmp (use_feedback); p *= param_multiplier
        const float s = p.ema[j];
        const float tmp = (1.0f - p.ema_decay) * (s - w);
        p.ema[j] = s - tmp;
        if (p.ema_feedback != 0.f) w += p.ema_feedback * tmp;
        if (p.param_multiplier != 0.f && p.param_multiplier != 1.0f) w *= p.param_multiplier;
      }
      p.p[j] = w;
    }
  }
}

extern "C" int64_t aitk_adamw_workspace_bytes(int64_t n) {
  const int64_t n1 = (n + OPT_BLOCK_ELEMS - 1) / OPT_BLOCK_ELEMS;
  const int64_t n2 = (n1 + 1023) / 1024;
  return (n1 + n2 + 8) * 4;  // two levels of norm partials + the 4-float control block of adamw_decide_kernel
}

extern "C" int aitk_adamw_ema_step(const AitkAdamWArgs* a, aitk_stream_t stream) {
  if (!a || a->n <= 0) return AITK_ERR_SHAPE;
  if (!a->p || !a->g || !a->m || !a->v || !a->norm_partial) return AITK_ERR_ARG;
  const long n1 = (a->n + OPT_BLOCK_ELEMS - 1) / OPT_BLOCK_ELEMS;
  const int n2 = (int)((n1 + 1023) / 1024);
  if (n2 > 4096) return AITK_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  AitkAdamWArgs args = *a;
  args.norm_partial2 = a->norm_partial + n1;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3((unsigned)n1), dim3(256), 0, s, a->g, (long)a->n, a->norm_partial);
  AITK_LAUNCH_CHECK();
  hipLaunchKernelGGL(sumsq_level2_kernel, dim3(n2), dim3(256), 0, s, a->norm_partial, (int)n1, args.norm_partial2);
  AITK_LAUNCH_CHECK();
  float* ctl = a->norm_partial + n1 + n2;
  hipLaunchKernelGGL(adamw_decide_kernel, dim3(1), dim3(1), 0, s, args, n2, ctl);
  AITK_LAUNCH_CHECK();
  hipLaunchKernelGGL(adamw_ema_kernel, dim3((unsigned)n1), dim3(256), 0, s, args, (const float*)ctl);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// EMA alone, for trainers that call the optimizer and the EMA separately (the reference: `self.optimizer.step()` ... `self.ema.update()`,
// extensions_built_in/sd_trainer/SDTrainer.py:2284-2293): toolkit/ema.py:126-152 over the flat arenas in ONE launch instead of its Python loop
// over every parameter (>= 3 tiny kernels each).  Same arithmetic, same order as the tail of adamw_ema_kernel:
//   tmp = (1 - d)(s - p); s -= tmp; p += feedback * tmp (use_feedback: 10); p *= param_multiplier.
// a product that must be ROUNDED before it is added: the file is built with -ffp-contract=fast, under which the backend fuses a multiply into
// the following add whatever pragma the source carries; an empty asm the value passes through keeps the two instructions apart
static __device__ __forceinline__ float rounded(float x) {
  asm volatile("" : "+v"(x));
  return x;
}
__global__ __launch_bounds__(256) void ema_update_kernel(float* __restrict__ p, float* __restrict__ ema, long n, float one_minus_decay,
                                                         float feedback, float mult) {
  const long base = (long)blockIdx.x * OPT_BLOCK_ELEMS;
  const bool writes_p = feedback != 0.f || (mult != 0.f && mult != 1.0f);
#pragma unroll
  for (int i = 0; i < OPT_BLOCK_ELEMS / (256 * 4); ++i) {
    const long j = base + (long)(i * 256 + threadIdx.x) * 4;
    if (j + 3 < n) {
      float4 w = *reinterpret_cast<const float4*>(p + j);
      float4 s = *reinterpret_cast<const float4*>(ema + j);
      float* wv = reinterpret_cast<float*>(&w);
      float* sv = reinterpret_cast<float*>(&s);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float tmp = rounded(one_minus_decay * (sv[k] - wv[k]));
        sv[k] = sv[k] - tmp;
        if (feedback != 0.f) wv[k] = wv[k] + rounded(feedback * tmp);
        if (mult != 0.f && mult != 1.0f) wv[k] = wv[k] * mult;
      }
      *reinterpret_cast<float4*>(ema + j) = s;
      if (writes_p) *reinterpret_cast<float4*>(p + j) = w;
    } else {
      for (long k = j; k < n && k < j + 4; ++k) {
        float w = p[k];
        const float s = ema[k];
        const float tmp = rounded(one_minus_decay * (s - w));
        ema[k] = s - tmp;
        if (feedback != 0.f) w = w + rounded(feedback * tmp);
        if (mult != 0.f && mult != 1.0f) w = w * mult;
        if (writes_p) p[k] = w;
      }
    }
  }
}

extern "C" int aitk_ema_update(float* p, float* ema, int64_t n, float one_minus_decay, float feedback, float param_multiplier,
                               aitk_stream_t stream) {
  if (n <= 0) return AITK_ERR_SHAPE;
  if (!p || !ema) return AITK_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(ema)) & 15) return AITK_ERR_ALIGN;
  const long n1 = (n + OPT_BLOCK_ELEMS - 1) / OPT_BLOCK_ELEMS;
  hipLaunchKernelGGL(ema_update_kernel, dim3((unsigned)n1), dim3(256), 0, (hipStream_t)stream, p, ema, (long)n, one_minus_decay, feedback,
                     param_multiplier);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// ------------------------------------------------------------------------------------------------ bf16 shadows
// For every adapter matrix in the fp32 arena (row-major [rows, cols]) write its bf16 shadows in the layouts the skinny kernels
// and the GEMM K-slab read (AitkShadowDesc in the header): hi = bf16(w), lo = bf16(w - hi) — the split representation that keeps
// the adapter branch at fp32-class precision (|w - hi - lo| <= 2^-17 |w|) on bf16 MFMA.
// TILED (default): the plain LoRA matrices (kind 1 / 2, rank <= 64) go through LDS in tiles of 64 columns (A) / 64 rows (B), so that the transposed
// layouts are written with consecutive lanes on consecutive addresses — the element-per-thread form below scatters 2-byte stores 96 B (the [in, 3 R] block)
// or a whole row (the transposes) apart: 1.8 ms for the 988 matrices of FLUX r16, once per step (and once per adapter-active forward of the trainer path).
// Same values to the same places: bit-identical.
template <bool TILED>
__global__ __launch_bounds__(256) void refresh_shadows_kernel(const float* arena, bf16_t* shadow, const AitkShadowDesc* table) {
  const AitkShadowDesc d = table[blockIdx.y];
  const long n = (long)d.rows * d.cols;
  const float* src = arena + d.src_off;
  if constexpr (TILED) {
    __shared__ bf16_t sh[2][64][66];  // [hi | lo][tile row][tile column] (+ 2: the transposed reads walk the banks)
    const int tid = threadIdx.x;
    if (d.kind == 1 && d.rows <= 64) {  // A [R, in]: tile = R ranks x 64 columns
      const int R = d.rows, w3 = 3 * R;
      const long stride = d.aux > 0 ? (long)d.aux : (long)w3;
      for (int c0 = blockIdx.x * 64; c0 < d.cols; c0 += gridDim.x * 64) {
        const int nc = min(64, d.cols - c0);
        for (int t = tid; t < R * 64; t += 256) {
          const int r = t >> 6, cc = t & 63;
          if (cc < nc) {
            const long i = (long)r * d.cols + c0 + cc;
            const float w = src[i];
            const bf16_t hi = f2bf(w), lo = f2bf(w - bf2f(hi));
            sh[0][r][cc] = hi;
            sh[1][r][cc] = lo;
            shadow[d.d0 + i] = hi;
            shadow[d.d1 + i] = lo;
          }
        }
        __syncthreads();
        for (int t = tid; t < nc * w3; t += 256) {
          const int col = t / w3, j = t - col * w3;
          const int blk = j / R, rr = j - blk * R;
          shadow[d.d2 + (long)(c0 + col) * stride + j] = sh[blk == 2][rr][col];
        }
        __syncthreads();
      }
      return;
    }
    if (d.kind == 2 && d.cols <= 64) {  // B [out, R]: tile = 64 rows x R ranks
      const int R = d.cols, w3 = 3 * R;
      for (int r0 = blockIdx.x * 64; r0 < d.rows; r0 += gridDim.x * 64) {
        const int nr = min(64, d.rows - r0);
        for (int t = tid; t < nr * R; t += 256) {
          const int rr = t / R, c = t - rr * R;
          const float w = src[(long)r0 * R + t];
          const bf16_t hi = f2bf(w), lo = f2bf(w - bf2f(hi));
          sh[0][rr][c] = hi;
          sh[1][rr][c] = lo;
        }
        __syncthreads();
        for (int t = tid; t < nr * w3; t += 256) {  // [out, 3 R] rows [hi | hi | lo]: one contiguous block per tile
          const int rr = t / w3, j = t - rr * w3;
          const int blk = j / R, c = j - blk * R;
          shadow[d.d0 + (long)(r0 + rr) * w3 + j] = sh[blk == 2][rr][c];
        }
        for (int t = tid; t < R * 64; t += 256) {  // the two transposes [R, out]
          const int c = t >> 6, rr = t & 63;
          if (rr < nr) {
            shadow[d.d1 + (long)c * d.rows + r0 + rr] = sh[0][rr][c];
            shadow[d.d2 + (long)c * d.rows + r0 + rr] = sh[1][rr][c];
          }
        }
        __syncthreads();
      }
      return;
    }
  }
  if (d.kind == 3) {  // low-rank LoKr factor: W2 = a [rows, r] @ b [r, cols] composed in fp32 (b follows a in the arena)
    const int r = d.aux;
    const float* a = src;
    const float* b = src + (long)d.rows * r;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
      const long ro = i / d.cols, c = i - ro * d.cols;
      float acc = 0.f;
      for (int k = 0; k < r; ++k) acc = fmaf(a[ro * r + k], b[(long)k * d.cols + c], acc);
      const bf16_t hi = f2bf(acc);
      shadow[d.d0 + i] = hi;
      shadow[d.d1 + c * d.rows + ro] = hi;
    }
    return;
  }
  if (d.kind == 4) {  // lora_down of a 3x3-conv adapter: Conv2d weight [rank, Cin, 3, 3] (columns c = cin*9 + tap), Cin = aux
    const int cin_n = d.aux;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
      const float w = src[i];
      const bf16_t hi = f2bf(w);
      const bf16_t lo = f2bf(w - bf2f(hi));
      const long r = i / d.cols, c = i - r * d.cols;
      const long cin = c / 9, tap = c - cin * 9;
      // d0: [2 rank, 9 Cin], tap-major columns, rows in 16-rank blocks [A_hi(16) ; A_lo(16)] — B operand of the implicit-GEMM lora_down
      // (AITK_EPI_SPLIT_SLAB adds columns n and n + 16 of every 32-column block)
      const long rh = (r >> 4) * 32 + (r & 15);
      shadow[d.d0 + rh * d.cols + tap * cin_n + cin] = hi;
      shadow[d.d0 + (rh + 16) * d.cols + tap * cin_n + cin] = lo;
      // d1: [Cin, 9 * 3 rank]: data-gradient filter over the dT slab image [hi | lo | hi] = rotated taps, [A_hi | A_hi | A_lo] per tap
      bf16_t* t3 = shadow + d.d1 + cin * (27 * d.rows) + (8 - tap) * 3 * d.rows + r;
      t3[0] = hi;
      t3[d.rows] = hi;
      t3[2 * d.rows] = lo;
    }
    return;
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float w = src[i];
    const bf16_t hi = f2bf(w);
    const long r = i / d.cols, c = i - r * d.cols;
    if (d.kind == 0) {
      shadow[d.d0 + i] = hi;
      shadow[d.d1 + c * d.rows + r] = hi;
      continue;
    }
    const bf16_t lo = f2bf(w - bf2f(hi));
    if (d.kind == 1) {  // A [rank, in]
      shadow[d.d0 + i] = hi;
      shadow[d.d1 + i] = lo;
      // aux > 0: row stride of the [in, 3 rank] block — the adapters of a same-input group share ONE [in, 3 R] matrix (column windows), the B2
      // operand of the group's K-concatenated data-gradient GEMM
      bf16_t* t3 = shadow + d.d2 + c * (d.aux > 0 ? (long)d.aux : 3L * d.rows) + r;
      t3[0] = hi;
      t3[d.rows] = hi;
      t3[2 * d.rows] = lo;
    } else {  // B [out, rank]
      bf16_t* d3 = shadow + d.d0 + r * 3 * d.cols + c;
      d3[0] = hi;
      d3[d.cols] = hi;
      d3[2 * d.cols] = lo;
      shadow[d.d1 + c * d.rows + r] = hi;
      shadow[d.d2 + c * d.rows + r] = lo;
    }
  }
}
extern "C" int aitk_lora_refresh_shadows(const float* arena, aitk_bf16* shadow, const AitkShadowDesc* table, int32_t ntensors,
                                         aitk_stream_t stream) {
  if (!arena || !shadow || !table || ntensors <= 0) return AITK_ERR_ARG;
  const char* e = getenv("AITK_REFRESH_TILED");  // read per launch (one launch per step): =0 -> the element-per-thread form for every matrix (A/B, bit-equality test)
  if (e && atoi(e) == 0) hipLaunchKernelGGL(refresh_shadows_kernel<false>, dim3(16, ntensors), dim3(256), 0, (hipStream_t)stream, arena, shadow, table);
  else hipLaunchKernelGGL(refresh_shadows_kernel<true>, dim3(16, ntensors), dim3(256), 0, (hipStream_t)stream, arena, shadow, table);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}


// ------------------------------------------------------------------------------------------------ bf16 gradient transport (DP all-reduce)
// the LoRA-gradient all-reduce over xGMI may run in fp32 (parity) or bf16 (half the bytes on the links).  The flat fp32
// gradient arena is rounded to bf16 into a transport buffer (one rounding per rank, round-to-nearest-even), RCCL sums the bf16 buffers,
// and the sum is expanded back over the fp32 arena.  HBM-bound: 6 B per element each way, 16-B accesses, grid-stride.
// `lead` scalar elements bring both pointers to a 16-byte boundary (the transport buffer is indexed like the arena, so one lead serves both);
// lead = n: no common alignment, everything goes through the scalar path.
__global__ __launch_bounds__(256) void grad_compress_bf16_kernel(const float* __restrict__ g, bf16_t* __restrict__ out, long n, int lead) {
  const long tid = (long)blockIdx.x * 256 + threadIdx.x, nth = (long)gridDim.x * 256;
  const long n8 = (n - lead) / 8;
  const float* ga = g + lead;
  bf16_t* oa = out + lead;
  for (long i = tid; i < n8; i += nth) {
    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(ga + 8 * i), b = *reinterpret_cast<const f32x4_t*>(ga + 8 * i + 4);
    uint4 o;
    o.x = pack2bf(a[0], a[1]); o.y = pack2bf(a[2], a[3]); o.z = pack2bf(b[0], b[1]); o.w = pack2bf(b[2], b[3]);
    *reinterpret_cast<uint4*>(oa + 8 * i) = o;
  }
  for (long i = tid; i < lead; i += nth) out[i] = f2bf(g[i]);
  for (long i = lead + 8 * n8 + tid; i < n; i += nth) out[i] = f2bf(g[i]);
}
__global__ __launch_bounds__(256) void grad_expand_bf16_kernel(const bf16_t* __restrict__ in, float* __restrict__ g, long n, int lead) {
  const long tid = (long)blockIdx.x * 256 + threadIdx.x, nth = (long)gridDim.x * 256;
  const long n8 = (n - lead) / 8;
  float* ga = g + lead;
  const bf16_t* ia = in + lead;
  for (long i = tid; i < n8; i += nth) {
    const uint4 u = *reinterpret_cast<const uint4*>(ia + 8 * i);
    f32x4_t a = {bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y)};
    f32x4_t b = {bf_lo(u.z), bf_hi(u.z), bf_lo(u.w), bf_hi(u.w)};
    *reinterpret_cast<f32x4_t*>(ga + 8 * i) = a;
    *reinterpret_cast<f32x4_t*>(ga + 8 * i + 4) = b;
  }
  for (long i = tid; i < lead; i += nth) g[i] = bf2f(in[i]);
  for (long i = lead + 8 * n8 + tid; i < n; i += nth) g[i] = bf2f(in[i]);
}
static int grad_lead(const void* f32p, const void* bf16p, int64_t n) {
  const int lead = (int)(((16 - ((uintptr_t)bf16p & 15)) & 15) / 2);  // bf16 elements up to the next 16-byte boundary
  if (lead >= n) return (int)n;
  return (((uintptr_t)((const float*)f32p + lead)) & 15) ? (int)(n > 0x7fffffff ? 0x7fffffff : n) : lead;
}
extern "C" int aitk_grad_compress_bf16(const float* g, aitk_bf16* out, int64_t n, aitk_stream_t stream) {
  if (!g || !out || n <= 0) return AITK_ERR_ARG;
  if (((uintptr_t)g & 3) || ((uintptr_t)out & 1) || n > 0x7fffffffLL * 8) return AITK_ERR_ALIGN;
  const int lead = grad_lead(g, out, n);
  const long blocks = (n / 8 + 255) / 256;
  hipLaunchKernelGGL(grad_compress_bf16_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream, g,
                     (bf16_t*)out, (long)n, lead);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}
extern "C" int aitk_grad_expand_bf16(const aitk_bf16* in, float* g, int64_t n, aitk_stream_t stream) {
  if (!g || !in || n <= 0) return AITK_ERR_ARG;
  if (((uintptr_t)g & 3) || ((uintptr_t)in & 1) || n > 0x7fffffffLL * 8) return AITK_ERR_ALIGN;
  const int lead = grad_lead(g, in, n);
  const long blocks = (n / 8 + 255) / 256;
  hipLaunchKernelGGL(grad_expand_bf16_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)in, g, (long)n, lead);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// ------------------------------------------------------------------------------------------------ low-rank LoKr factor gradients
// W2 = a [O, r] @ b [r, I] (toolkit/models/lokr.py:184-197): from the gradient dW [O, I] of the composed factor,
// ga (+)= dW b^T, gb (+)= a^T dW.  O, I <= a few hundred, r <= 64: one workgroup, fp32 VALU, fixed summation order.
__global__ __launch_bounds__(256) void lokr_lowrank_grad_kernel(const float* dW, const float* a, const float* b, float* ga, float* gb,
                                                                int O, int I, int r, int accumulate) {
  const int na = O * r, nb = r * I;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < na + nb; e += gridDim.x * 256) {
    float acc = 0.f;
    if (e < na) {
      const int o = e / r, k = e - o * r;
      for (int i = 0; i < I; ++i) acc = fmaf(dW[(long)o * I + i], b[(long)k * I + i], acc);
      ga[e] = accumulate ? ga[e] + acc : acc;
    } else {
      const int f = e - na;
      const int k = f / I, i = f - k * I;
      for (int o = 0; o < O; ++o) acc = fmaf(a[(long)o * r + k], dW[(long)o * I + i], acc);
      gb[f] = accumulate ? gb[f] + acc : acc;
    }
  }
}
extern "C" int aitk_lokr_lowrank_grad(const float* dW, const float* a, const float* b, float* ga, float* gb, int32_t O, int32_t I,
                                      int32_t r, int32_t accumulate, aitk_stream_t stream) {
  if (!dW || !a || !b || !ga || !gb) return AITK_ERR_ARG;
  if (O <= 0 || I <= 0 || r <= 0) return AITK_ERR_SHAPE;
  const int n = O * r + r * I;
  hipLaunchKernelGGL(lokr_lowrank_grad_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dW, a, b, ga, gb, O, I, r, accumulate);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// ------------------------------------------------------------------------------------------------ DoRA
// c_j = magnitude_j / sqrt(||W_j||^2 + 2 s B_j.(W A^T)_j + s^2 B_j (A A^T) B_j^T); one thread per output channel.  The Gram matrix sits in LDS
// up to rank 64 (16 KiB); larger ranks (toolkit/models/DoRA.py:126-148 has no rank limit) read it through the caches — the same sums in the same order.
template <bool GRAM_LDS>
__global__ void dora_colscale_kernel(AitkDoraColscaleArgs p) {
  __shared__ float g_lds[GRAM_LDS ? 64 * 64 : 1];
  if constexpr (GRAM_LDS) {
    for (int i = threadIdx.x; i < p.R * p.R; i += blockDim.x) g_lds[i] = p.gram[i];
    __syncthreads();
  }
  const float* g = GRAM_LDS ? g_lds : p.gram;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= p.N) return;
  const float* b = p.up + (long)j * p.R;
  float cross = 0.f, quad = 0.f;
  for (int r = 0; r < p.R; ++r) {
    const float br = b[r];
    cross += br * bf2f(p.tw[(long)j * p.ldtw + r]);
    float t = 0.f;
    for (int q = 0; q < p.R; ++q) t += g[r * p.R + q] * b[q];
    quad += br * t;
  }
  const float n2 = p.w2[j] + 2.0f * p.s * cross + p.s * p.s * quad;
  p.c[j] = p.mag[j] / sqrtf(n2);
}
extern "C" int aitk_dora_colscale(const AitkDoraColscaleArgs* a, aitk_stream_t stream) {
  if (!a || a->N <= 0 || a->R <= 0 || a->R > 1024) return AITK_ERR_SHAPE;
  if (!a->w2 || !a->tw || !a->up || !a->gram || !a->mag || !a->c) return AITK_ERR_ARG;
  if (a->R <= 64) hipLaunchKernelGGL(dora_colscale_kernel<true>, dim3((a->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, *a);
  else hipLaunchKernelGGL(dora_colscale_kernel<false>, dim3((a->N + 63) / 64), dim3(64), 0, (hipStream_t)stream, *a);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// dz = c * dy; per-row-block partial column sums of dy and dy*y (deterministic two-stage reduction like aitk_gate_bwd)
#define DORA_RPB 16
__global__ __launch_bounds__(256) void dora_bwd_kernel(AitkDoraBwdArgs p) {
  const int m0 = blockIdx.x * DORA_RPB;
  const int nrows = min(DORA_RPB, p.M - m0);
  for (int ch = threadIdx.x; ch < p.N / 8; ch += 256) {
    const int c = ch * 8;
    float cs[8], s0[8], s1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      cs[e] = p.c[c + e];
      s0[e] = s1[e] = 0.f;
    }
    for (int r = 0; r < nrows; ++r) {
      const long m = m0 + r;
      const uint4 dv = *reinterpret_cast<const uint4*>(p.dy + m * p.ld_dy + c);
      const uint4 yv = *reinterpret_cast<const uint4*>(p.y + m * p.ld_y + c);
      float d[8], y[8];
      d[0] = bf2f(dv.x & 0xffff); d[1] = bf2f(dv.x >> 16); d[2] = bf2f(dv.y & 0xffff); d[3] = bf2f(dv.y >> 16);
      d[4] = bf2f(dv.z & 0xffff); d[5] = bf2f(dv.z >> 16); d[6] = bf2f(dv.w & 0xffff); d[7] = bf2f(dv.w >> 16);
      y[0] = bf2f(yv.x & 0xffff); y[1] = bf2f(yv.x >> 16); y[2] = bf2f(yv.y & 0xffff); y[3] = bf2f(yv.y >> 16);
      y[4] = bf2f(yv.z & 0xffff); y[5] = bf2f(yv.z >> 16); y[6] = bf2f(yv.w & 0xffff); y[7] = bf2f(yv.w >> 16);
      uint4 o;
      o.x = pack2bf(d[0] * cs[0], d[1] * cs[1]); o.y = pack2bf(d[2] * cs[2], d[3] * cs[3]);
      o.z = pack2bf(d[4] * cs[4], d[5] * cs[5]); o.w = pack2bf(d[6] * cs[6], d[7] * cs[7]);
      *reinterpret_cast<uint4*>(p.dz + m * p.ld_dz + c) = o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s0[e] += d[e];
        s1[e] += d[e] * y[e];
      }
    }
    float* pp = p.partial + ((long)blockIdx.x * 2) * p.N + c;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      pp[e] = s0[e];
      pp[p.N + e] = s1[e];
    }
  }
}
__global__ void dora_bwd_finish_kernel(AitkDoraBwdArgs p, int nchunk) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= p.N) return;
  float s0 = 0.f, s1 = 0.f;
  for (int k = 0; k < nchunk; ++k) {
    s0 += p.partial[((long)k * 2) * p.N + j];
    s1 += p.partial[((long)k * 2 + 1) * p.N + j];
  }
  const float b = p.bias ? bf2f(p.bias[j]) : 0.f;
  p.dmag[j] += (s1 - b * s0) / p.mag[j];
}
extern "C" int aitk_dora_bwd(const AitkDoraBwdArgs* a, aitk_stream_t stream) {
  if (!a || a->M <= 0 || a->N <= 0 || (a->N % 8)) return AITK_ERR_SHAPE;
  if ((a->ld_dy % 8) || (a->ld_y % 8) || (a->ld_dz % 8)) return AITK_ERR_ALIGN;
  if (!a->dy || !a->y || !a->c || !a->mag || !a->dz || !a->dmag || !a->partial) return AITK_ERR_ARG;
  const int nchunk = (a->M + DORA_RPB - 1) / DORA_RPB;
  hipLaunchKernelGGL(dora_bwd_kernel, dim3(nchunk), dim3(256), 0, (hipStream_t)stream, *a);
  AITK_LAUNCH_CHECK();
  hipLaunchKernelGGL(dora_bwd_finish_kernel, dim3((a->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, *a, nchunk);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}


// ------------------------------------------------------------------------------------------------ per-token fp8 quantisation
// Q[m][k] = e4m3( x[m][k] * col_mul[k] * (1 / s_m) ),  s_m = max_k |x[m][k] * col_mul[k]| / 448  (>= 2^-126): the A operand of the W8A8
// GEMM (AitkGemmArgs.b_scale_mode 3).  One wave per row: pass 1 row maximum, pass 2 re-reads the row (L2) and converts with
// v_cvt_pk_fp8_f32 (OCP e4m3 on gfx950, round-to-nearest-even); 16-B loads, 8-B stores.
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(AitkQuantRowsArgs p) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= p.M) return;
  long off;
  if (p.seg_rows > 0) {
    const int sgi = m / p.seg_rows;
    off = (long)sgi * p.seg_stride + (long)(m - sgi * p.seg_rows) * p.ldx;
  } else {
    off = (long)m * p.ldx;
  }
  const bf16_t* x = p.X + off;
  const int nch = p.K >> 3;
  float amax = 0.f;
  for (int c = lane; c < nch; c += 64) {
    const uint4 v = *reinterpret_cast<const uint4*>(x + c * 8);
    float f[8] = {bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y), bf_lo(v.z), bf_hi(v.z), bf_lo(v.w), bf_hi(v.w)};
    if (p.col_mul) {
      const f32x4_t c0 = *reinterpret_cast<const f32x4_t*>(p.col_mul + c * 8), c1 = *reinterpret_cast<const f32x4_t*>(p.col_mul + c * 8 + 4);
      f[0] *= c0[0]; f[1] *= c0[1]; f[2] *= c0[2]; f[3] *= c0[3]; f[4] *= c1[0]; f[5] *= c1[1]; f[6] *= c1[2]; f[7] *= c1[3];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(f[e]));
  }
  amax = wave_max(amax);
  const float sc = fmaxf(amax / 448.0f, 1.17549435e-38f);
  const float inv = 1.0f / sc;
  if (lane == 0) p.row_scale[m] = sc;
  uint8_t* q = p.Q + (long)m * p.ldq;
  for (int c = lane; c < nch; c += 64) {
    const uint4 v = *reinterpret_cast<const uint4*>(x + c * 8);
    float f[8] = {bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y), bf_lo(v.z), bf_hi(v.z), bf_lo(v.w), bf_hi(v.w)};
    if (p.col_mul) {
      const f32x4_t c0 = *reinterpret_cast<const f32x4_t*>(p.col_mul + c * 8), c1 = *reinterpret_cast<const f32x4_t*>(p.col_mul + c * 8 + 4);
      f[0] *= c0[0]; f[1] *= c0[1]; f[2] *= c0[2]; f[3] *= c0[3]; f[4] *= c1[0]; f[5] *= c1[1]; f[6] *= c1[2]; f[7] *= c1[3];
    }
    int w0 = 0, w1 = 0;
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, w1, true);
    *reinterpret_cast<uint2*>(q + c * 8) = make_uint2((unsigned)w0, (unsigned)w1);
  }
}
extern "C" int aitk_quant_rows_fp8(const AitkQuantRowsArgs* a, aitk_stream_t stream) {
  if (!a || a->M <= 0 || a->K <= 0 || (a->K % 16)) return AITK_ERR_SHAPE;
  if (!a->X || !a->Q || !a->row_scale) return AITK_ERR_ARG;
  if ((a->ldx % 8) || (a->ldq % 16) || (a->seg_stride % 8) || (((uintptr_t)a->X | (uintptr_t)a->Q | (uintptr_t)a->col_mul) & 15)) return AITK_ERR_ALIGN;
  hipLaunchKernelGGL(quant_rows_fp8_kernel, dim3((unsigned)((a->M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *a);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}

// ------------------------------------------------------------------------------------------------ fp8 weight dequantisation
// out[r][k] = bf16(e4m3(q[r][k]) * scale) with scale indexed by the row (mode 1: q = W [out,in]) or by the column (mode 2:
// q = W^T [in,out]) — the value a weight-only-quantised Linear multiplies with (optimum-quanto qfloat8 / torchao
// Float8WeightOnly, toolkit/util/quantize.py:43-75).  Used to expand one layer's weight into a reusable bf16 scratch right
// before its GEMM: 3 bytes of traffic per weight element, then the bf16 8-phase kernel runs at full speed.
__global__ __launch_bounds__(256) void dequant_fp8_kernel(const uint8_t* q, long ldq, const float* scale, int mode, bf16_t* out, long ldo,
                                                           int rows, int cols) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
  const long total = (long)rows * cols;
  if (i >= total) return;
  const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
  const uint2 v = *reinterpret_cast<const uint2*>(q + (long)r * ldq + c);
  float f[8];
  f[0] = __builtin_amdgcn_cvt_f32_fp8(v.x, 0); f[1] = __builtin_amdgcn_cvt_f32_fp8(v.x, 1);
  f[2] = __builtin_amdgcn_cvt_f32_fp8(v.x, 2); f[3] = __builtin_amdgcn_cvt_f32_fp8(v.x, 3);
  f[4] = __builtin_amdgcn_cvt_f32_fp8(v.y, 0); f[5] = __builtin_amdgcn_cvt_f32_fp8(v.y, 1);
  f[6] = __builtin_amdgcn_cvt_f32_fp8(v.y, 2); f[7] = __builtin_amdgcn_cvt_f32_fp8(v.y, 3);
  if (mode == 1) {
    const float sc = scale[r];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] *= sc;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] *= scale[c + e];
  }
  uint4 o;
  o.x = pack2bf(f[0], f[1]); o.y = pack2bf(f[2], f[3]); o.z = pack2bf(f[4], f[5]); o.w = pack2bf(f[6], f[7]);
  *reinterpret_cast<uint4*>(out + (long)r * ldo + c) = o;
}
extern "C" int aitk_dequant_fp8(const uint8_t* q, int64_t ldq, const float* scale, int32_t mode, aitk_bf16* out, int64_t ldo,
                                int32_t rows, int32_t cols, aitk_stream_t stream) {
  if (!q || !scale || !out || rows <= 0 || cols <= 0 || (cols % 8) || (mode != 1 && mode != 2)) return AITK_ERR_ARG;
  if ((ldq % 8) || (ldo % 8)) return AITK_ERR_ALIGN;
  const long n = (long)rows * cols / 8;
  hipLaunchKernelGGL(dequant_fp8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q, (long)ldq, scale, mode, out,
                     (long)ldo, rows, cols);
  AITK_LAUNCH_CHECK();
  return AITK_OK;
}
