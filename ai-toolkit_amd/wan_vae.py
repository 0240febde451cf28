"""Wan2.1 video VAE encoder (`AutoencoderKLWan`) on the gfx950 kernels (host graph): `Wan21.encode_images`.

Replaces `self.vae.encode(images).latent_dist.sample()` + `(latents - latents_mean) * (1 / latents_std)` of
toolkit/models/wan21/wan21.py:618-672.  Module / parameter names and shapes are diffusers' (`encoder.conv_in.weight` [96,3,3,3,3],
`encoder.down_blocks.N.norm1.gamma` [C,1,1,1], `....resample.1.weight`, `....time_conv.weight` [C,C,3,1,1],
`encoder.mid_block.attentions.0.to_qkv.weight`, `quant_conv.weight` ...) so a diffusers checkpoint loads by key.

Formulation: diffusers evaluates the encoder in chunks (first frame, then 4 frames at a time) with a 2-frame feature cache in front of
every causal convolution; that is the same function as ONE pass over the whole clip with every 3x3x3 convolution causal (two zero frames
in front) and the temporal down-sampler defined as  y[0] = x[0],  y[j] = time_conv(x[2j-2], x[2j-1], x[2j])  (its first chunk skips the
time convolution) — tests/test_wan_vae_cpu.py checks this graph against the chunked restatement in oracle/wan_vae_ref.py.

Data layout: a clip is [T*H*W, C] bf16 rows (frames are the batch axis of the NHWC kernels).  Every buffer that feeds a causal 3x3x3
convolution is allocated with two zero frames in front, so the convolution is a single implicit-GEMM launch whose K axis runs over
(dt, ky, kx, Cin) = 27*Cin (`ops.conv3d`, `conv_t3d` mode of gemm_nt_kernel<CONV>: temporal tap dt of output frame t reads frame t + dt of
the padded buffer; fp32 accumulation over all 27 taps, one rounding).  WanRMS_norm (+SiLU) is the row kernel `aitk_rmsnorm_rows`, which
writes straight into the next convolution's padded buffer; residual adds ride the convolution epilogue; the (3,1,1) stride-2 time
convolution is the same kernel with ks = 1, tstride = 2; the per-frame single-head mid attention materialises its scores with the GEMM.
"""
import torch
import torch.nn as nn

LATENTS_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508, 0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497,
                0.2503, -0.2921]
LATENTS_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743, 3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251,
               1.9160]


class _Conv(nn.Module):
    """frozen convolution holder; `shape` = kernel dims, e.g. (3,3,3), (1,1,1), (3,1,1), (3,3), (1,1)"""

    def __init__(self, cin, cout, shape, dtype, device):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, *shape, dtype=dtype, device=device), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(cout, dtype=dtype, device=device), requires_grad=False)
        self.wk = None


class _RMS(nn.Module):
    def __init__(self, c, images, dtype, device):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones((c, 1, 1) if images else (c, 1, 1, 1), dtype=dtype, device=device), requires_grad=False)
        self.g = None


class _Res(nn.Module):
    def __init__(self, cin, cout, dtype, device):
        super().__init__()
        self.norm1 = _RMS(cin, False, dtype, device)
        self.conv1 = _Conv(cin, cout, (3, 3, 3), dtype, device)
        self.norm2 = _RMS(cout, False, dtype, device)
        self.conv2 = _Conv(cout, cout, (3, 3, 3), dtype, device)
        self.conv_shortcut = _Conv(cin, cout, (1, 1, 1), dtype, device) if cin != cout else nn.Identity()


class _Resample(nn.Module):
    def __init__(self, c, mode, dtype, device):
        super().__init__()
        self.mode = mode
        self.resample = nn.Sequential(nn.Identity(), _Conv(c, c, (3, 3), dtype, device))  # index 1 like (ZeroPad2d, Conv2d)
        if mode == "downsample3d":
            self.time_conv = _Conv(c, c, (3, 1, 1), dtype, device)


class _Attn(nn.Module):
    def __init__(self, c, dtype, device):
        super().__init__()
        self.norm = _RMS(c, True, dtype, device)
        self.to_qkv = _Conv(c, 3 * c, (1, 1), dtype, device)
        self.proj = _Conv(c, c, (1, 1), dtype, device)


class _Holder(nn.Module):
    pass


class AutoencoderKLWanEncoder(nn.Module):
    def _apply(self, fn, recurse=True):
        """`.to()` / `.cpu()` / `.float()` are no-ops: the reference's trainer parks the VAE on the CPU and re-casts it (jobs/process/
        BaseSDTrainProcess.py:1902); the native encoder stays on its device in its dtype (kernel-layout buffers are not nn.Parameters)."""
        return self

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return self.dt

    def __init__(self, base_dim=96, z_dim=16, dim_mult=(1, 2, 4, 4), num_res_blocks=2, temperal_downsample=(False, True, True),
                 latents_mean=None, latents_std=None, dtype=torch.bfloat16, device=None, ops=None):
        super().__init__()
        self.ops, self.dt, self.z_dim = ops, dtype, z_dim
        dims = [base_dim * u for u in (1,) + tuple(dim_mult)]
        enc = _Holder()
        enc.conv_in = _Conv(3, dims[0], (3, 3, 3), dtype, device)
        blocks = []
        for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
            for _ in range(num_res_blocks):
                blocks.append(_Res(cin, cout, dtype, device))
                cin = cout
            if i != len(dim_mult) - 1:
                blocks.append(_Resample(cout, "downsample3d" if temperal_downsample[i] else "downsample2d", dtype, device))
        enc.down_blocks = nn.ModuleList(blocks)
        mid = _Holder()
        mid.resnets = nn.ModuleList([_Res(dims[-1], dims[-1], dtype, device), _Res(dims[-1], dims[-1], dtype, device)])
        mid.attentions = nn.ModuleList([_Attn(dims[-1], dtype, device)])
        enc.mid_block = mid
        enc.norm_out = _RMS(dims[-1], False, dtype, device)
        enc.conv_out = _Conv(dims[-1], 2 * z_dim, (3, 3, 3), dtype, device)
        self.encoder = enc
        self.quant_conv = _Conv(2 * z_dim, 2 * z_dim, (1, 1, 1), dtype, device)
        self.latents_mean = list(LATENTS_MEAN[:z_dim] if latents_mean is None else latents_mean)
        self.latents_std = list(LATENTS_STD[:z_dim] if latents_std is None else latents_std)
        self._prepared = False

    # ------------------------------------------------------------------------------------------------ kernel layouts
    def prepare(self):
        for m in self.modules():
            if isinstance(m, _Conv):
                w = m.weight.data
                cout, cin = w.shape[0], w.shape[1]
                cpad = (cin + 7) // 8 * 8
                taps = w.reshape(cout, cin, -1).permute(0, 2, 1)  # [Cout, kt*ky*kx (row-major), Cin]
                wk = torch.zeros(cout, taps.shape[1], cpad, dtype=w.dtype, device=w.device)
                wk[..., :cin] = taps
                m.wk = wk.reshape(cout, -1).contiguous()
            elif isinstance(m, _RMS):
                m.g = m.gamma.data.reshape(-1).contiguous()
        dev = self.encoder.conv_in.weight.device
        self._ch_shift = torch.tensor(self.latents_mean, dtype=torch.float32, device=dev)
        self._ch_scale = 1.0 / torch.tensor(self.latents_std, dtype=torch.float32, device=dev)
        self._prepared = True
        return self

    def _new(self, rows, c):
        return torch.empty(rows, c, dtype=self.dt, device=self.encoder.conv_in.weight.device)

    def _padded(self, T, HW, c):
        """buffer of T frames with the two causal zero frames in front -> (whole buffer, view of the T frames)"""
        buf = self._new((T + 2) * HW, c)
        buf[:2 * HW].zero_()
        return buf, buf[2 * HW:]

    # ------------------------------------------------------------------------------------------------ blocks
    def _res(self, r, x, T, H, W):
        ops = self.ops
        HW, cout = H * W, r.conv1.weight.shape[0]
        a, a_body = self._padded(T, HW, x.shape[1])
        ops.rmsnorm_rows(x, r.norm1.g, a_body, silu=True)
        h1 = self._new(T * HW, cout)
        ops.conv3d(a, r.conv1.wk, h1, T=T, H=H, W=W, bias=r.conv1.bias)
        b, b_body = self._padded(T, HW, cout)
        ops.rmsnorm_rows(h1, r.norm2.g, b_body, silu=True)
        res = x
        if isinstance(r.conv_shortcut, _Conv):
            res = self._new(T * HW, cout)
            ops.gemm_nt(x, r.conv_shortcut.wk, res, bias=r.conv_shortcut.bias)
        out = h1  # conv1's output is dead once normalised
        ops.conv3d(b, r.conv2.wk, out, T=T, H=H, W=W, bias=r.conv2.bias, flags=ops.EPI_ADD_AUX, aux_in=res)
        return out

    def _resample(self, rs, x, T, H, W):
        ops = self.ops
        Ho, Wo, c = H // 2, W // 2, x.shape[1]
        conv = rs.resample[1]
        y = self._new(T * Ho * Wo, c)
        ops.conv3x3(x, conv.wk, y, B=T, H=H, W=W, stride=2, pad_t=0, pad_l=0, Ho=Ho, Wo=Wo, bias=conv.bias)  # ZeroPad2d((0,1,0,1))
        if rs.mode != "downsample3d" or T == 1:
            return y, T, Ho, Wo
        n = (T - 1) // 2
        z = self._new((1 + n) * Ho * Wo, c)
        ops.copy_rows(z[:Ho * Wo], y[:Ho * Wo])  # first frame: no time convolution
        ops.conv3d(y, rs.time_conv.wk, z[Ho * Wo:], T=n, H=Ho, W=Wo, kt=3, ks=1, tstride=2, pad_t=0, pad_l=0, bias=rs.time_conv.bias)
        return z, 1 + n, Ho, Wo

    def _attention(self, a, x, T, HW):
        ops = self.ops
        c = x.shape[1]
        wq, wk_, wv = a.to_qkv.wk[:c], a.to_qkv.wk[c:2 * c], a.to_qkv.wk[2 * c:]
        bq, bk, bv = a.to_qkv.bias[:c], a.to_qkv.bias[c:2 * c], a.to_qkv.bias[2 * c:]
        n = self._new(T * HW, c)
        ops.rmsnorm_rows(x, a.norm.g, n, silu=False)
        q, k = self._new(T * HW, c), self._new(T * HW, c)
        ops.gemm_nt(n, wq, q, bias=bq)
        ops.gemm_nt(n, wk_, k, bias=bk)
        o = self._new(T * HW, c)
        scores = self._new(HW, HW)
        vT = self._new(c, HW)
        for t in range(T):
            sl = slice(t * HW, (t + 1) * HW)
            ops.gemm_nt(wv, n[sl], vT, bias=bv, flags=ops.EPI_BIAS_ROW)  # V^T = W_v x^T + b_v
            ops.gemm_nt(q[sl], k[sl], scores)
            ops.softmax_rows(scores, c ** -0.5)
            ops.gemm_nt(scores, vT, o[sl])
        out = self._new(T * HW, c)
        ops.gemm_nt(o, a.proj.wk, out, bias=a.proj.bias, flags=ops.EPI_ADD_AUX, aux_in=x)
        return out

    # ------------------------------------------------------------------------------------------------ encoder
    def moments(self, clip):
        """clip [T,3,H,W] fp32 in [-1,1] -> moments rows [T'*h*w, 2*z_dim] (mean | logvar), (T', h, w); frames past 1 + 4k are dropped
        like the published chunk loop does."""
        if not self._prepared:
            self.prepare()
        ops, enc = self.ops, self.encoder
        T, _, Hs, Ws = clip.shape
        H, W = Hs // 8 * 8, Ws // 8 * 8  # sides that are not multiples of 8 are resized bilinearly, every frame (wan21.py:652-657)
        if H == 0 or W == 0:
            raise ValueError(f"Wan VAE encode: image side below 8 pixels ({Hs}x{Ws})")
        T = 1 + 4 * ((T - 1) // 4)
        HW = H * W
        x8, x8_body = self._padded(T, HW, 8)
        if (H, W) != (Hs, Ws):
            ops.image_resize_to_nhwc8(clip[:T].float().contiguous(), x8_body, Hd=H, Wd=W)
        else:
            ops.image_to_nhwc8(clip[:T].float().contiguous(), x8_body)
        x = self._new(T * HW, enc.conv_in.weight.shape[0])
        ops.conv3d(x8, enc.conv_in.wk, x, T=T, H=H, W=W, bias=enc.conv_in.bias)
        for blk in enc.down_blocks:
            if isinstance(blk, _Res):
                x = self._res(blk, x, T, H, W)
            else:
                x, T, H, W = self._resample(blk, x, T, H, W)
        x = self._res(enc.mid_block.resnets[0], x, T, H, W)
        x = self._attention(enc.mid_block.attentions[0], x, T, H * W)
        x = self._res(enc.mid_block.resnets[1], x, T, H, W)
        a, a_body = self._padded(T, H * W, x.shape[1])
        ops.rmsnorm_rows(x, enc.norm_out.g, a_body, silu=True)
        mom = self._new(T * H * W, 2 * self.z_dim)
        ops.conv3d(a, enc.conv_out.wk, mom, T=T, H=H, W=W, bias=enc.conv_out.bias)
        mq = self._new(T * H * W, 2 * self.z_dim)
        ops.gemm_nt(mom, self.quant_conv.wk, mq, bias=self.quant_conv.bias)
        return mq, (T, H, W)

    @torch.no_grad()
    def encode_images(self, image_list, eps=None, generator=None):
        """list of [C,H,W] images / [T,C,H,W] clips in [-1,1] -> normalised latents [B, z_dim, T', H/8, W/8] (model dtype), as
        Wan21.encode_images returns (toolkit/models/wan21/wan21.py:618-672)."""
        clips = []
        for im in image_list:
            if im.ndim == 3:
                clips.append(im.unsqueeze(0))  # one frame
            elif im.ndim == 4:
                clips.append(im)               # [T,C,H,W] is already frame-major (the reference permutes to [C,T,H,W] for Conv3d)
            else:
                raise ValueError(f"Invalid image shape: {im.shape}")
        if len({tuple(c.shape) for c in clips}) != 1:
            raise ValueError("all clips of a batch must have one shape")
        out = None
        for b, clip in enumerate(clips):
            mom, (T, h, w) = self.moments(clip)
            if out is None:
                out = torch.empty(len(clips), self.z_dim, T, h, w, dtype=self.dt, device=mom.device)
                if eps is None:
                    eps = torch.randn(out.shape, device=mom.device, dtype=torch.float32, generator=generator)
                eps = eps.float().contiguous()
            self.ops.latent_sample_affine(mom, eps[b:b + 1], out[b:b + 1], ch_shift=self._ch_shift, ch_scale=self._ch_scale)
        return out
