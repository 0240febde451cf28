"""Wan2.1 T2V video DiT forward / backward as an explicit op graph over the gfx950 kernels (host logic only).

Module tree and parameter names are those of diffusers' `WanTransformer3DModel` — the class the reference loads at
toolkit/models/wan21/wan21.py:343-420 and calls at :578-603 with (hidden_states [B,16,F,H,W], timestep 0..1000,
encoder_hidden_states [B,512,4096]) — so diffusers checkpoints load by key and the LoRA network produces the
reference's keys (`transformer.blocks.N.attn1.to_q.lora_A.weight` ...; target module 'WanTransformer3DModel',
block filter ['blocks'], wan21.py:330,735: ten adapters per block).  The attention follows the reference's own
processor (toolkit/models/wan21/wan_attn.py:12-103).  Same computation as oracle/wan_ref.py, every arithmetic step a
C-ABI kernel call:

  patch_embedding Conv3d k=s=(1,2,2)  -> gemm_nt on 2x2-packed tokens (K = 64)
  scale_shift_table + time_proj       -> gemv_nt + ew(add): frozen, so no gradient flows into the modulation
  norm1/3 (+modulate), norm2 (affine) -> ln_mod_fwd (norm2: scale = gamma-1, shift = beta, one virtual batch)
  attn1                               -> LoRA-fused q/k/v GEMMs, rms_full_fwd (RMSNorm across heads + 3-D RoPE), attn_fwd
  attn2 (text cross-attention)        -> q from the video tokens, k/v from the 512 text tokens (attn_fwd with Skv=512)
  to_out / ffn                        -> gemm_nt with gate-residual / residual-add / GELU-tanh epilogues
and an explicit backward (no autograd inside, no gradient checkpointing; activations stay resident in HBM).

Token order is (frame, row, col); the packed feature order is the Conv3d weight's (c, ph, pw).  `proj_out` emits
(ph, pw, c) in diffusers; prepare() keeps a row-permuted copy so the prediction comes out in the SAME (c, ph, pw)
order as the packed flow-matching target (aitk_flow_noise_pack) and the loss needs no unpatchify.
"""
import math

import torch
import torch.nn as nn

from .graph import (EPI_ADD_AUX, EPI_DGELU, EPI_GATE_RES, EPI_GELU, FusedGraphBase, Linear, RMSNormW, _Holder)


def _wan_attention(dim, heads, dtype, device):
    a = _Holder()
    a.to_q = Linear(dim, dim, True, dtype, device)
    a.to_k = Linear(dim, dim, True, dtype, device)
    a.to_v = Linear(dim, dim, True, dtype, device)
    a.to_out = nn.ModuleList([Linear(dim, dim, True, dtype, device), nn.Identity()])
    a.norm_q = RMSNormW(dim, dtype, device)
    a.norm_k = RMSNormW(dim, dtype, device)
    return a


class _AffineNorm(nn.Module):
    def __init__(self, dim, dtype, device):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim, dtype=dtype, device=device), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(dim, dtype=dtype, device=device), requires_grad=False)


class WanTransformerBlock(nn.Module):
    def __init__(self, dim, ffn_dim, heads, dtype, device):
        super().__init__()
        self.attn1 = _wan_attention(dim, heads, dtype, device)
        self.attn2 = _wan_attention(dim, heads, dtype, device)
        self.norm2 = _AffineNorm(dim, dtype, device)
        f, g = _Holder(), _Holder()
        g.proj = Linear(dim, ffn_dim, True, dtype, device)
        f.net = nn.ModuleList([g, nn.Identity(), Linear(ffn_dim, dim, True, dtype, device)])
        self.ffn = f
        self.scale_shift_table = nn.Parameter(torch.zeros(1, 6, dim, dtype=dtype, device=device), requires_grad=False)


class _PatchEmbed(nn.Module):
    def __init__(self, cin, dim, patch, dtype, device):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(dim, cin, *patch, dtype=dtype, device=device), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(dim, dtype=dtype, device=device), requires_grad=False)


class WanTransformer3DModel(FusedGraphBase):
    def __init__(self, patch_size=(1, 2, 2), num_attention_heads=12, attention_head_dim=128, in_channels=16, out_channels=16,
                 text_dim=4096, freq_dim=256, ffn_dim=8960, num_layers=30, eps=1e-6, dtype=torch.bfloat16, device=None, ops=None):
        super().__init__()
        assert attention_head_dim == 128, "attention kernels are specialised for head_dim 128"
        assert tuple(patch_size) == (1, 2, 2) and in_channels == out_channels
        self.config = dict(patch_size=tuple(patch_size), num_attention_heads=num_attention_heads,
                           attention_head_dim=attention_head_dim, in_channels=in_channels, out_channels=out_channels,
                           text_dim=text_dim, freq_dim=freq_dim, ffn_dim=ffn_dim, num_layers=num_layers, eps=eps)
        self.heads, self.dim = num_attention_heads, num_attention_heads * attention_head_dim
        self.in_channels, self.freq_dim, self.eps = in_channels, freq_dim, eps
        d = self.dim
        self.patch_embedding = _PatchEmbed(in_channels, d, patch_size, dtype, device)
        ce = _Holder()
        te = _Holder()
        te.linear_1 = Linear(freq_dim, d, True, dtype, device)
        te.linear_2 = Linear(d, d, True, dtype, device)
        ce.time_embedder = te
        ce.time_proj = Linear(d, 6 * d, True, dtype, device)
        tx = _Holder()
        tx.linear_1 = Linear(text_dim, d, True, dtype, device)
        tx.linear_2 = Linear(d, d, True, dtype, device)
        ce.text_embedder = tx
        self.condition_embedder = ce
        self.blocks = nn.ModuleList([WanTransformerBlock(d, ffn_dim, num_attention_heads, dtype, device) for _ in range(num_layers)])
        self.proj_out = Linear(d, out_channels * 4, True, dtype, device)
        self.scale_shift_table = nn.Parameter(torch.zeros(1, 2, d, dtype=dtype, device=device), requires_grad=False)
        self._init_graph(ops, dtype)  # grad_ready_hook pieces: 'late' (second half of the blocks) then 'early'
        self._rope_cache = {}

    # ------------------------------------------------------------------ setup
    def _token_linears(self):
        out = []
        for blk in self.blocks:
            for a in (blk.attn1, blk.attn2):
                out += [a.to_q, a.to_k, a.to_v, a.to_out[0]]
            out += [blk.ffn.net[0].proj, blk.ffn.net[2]]
        return out

    def _dgrad_linears(self):
        # attn2.to_k / to_v read the (frozen) text states: no data gradient, so no transposed copy
        skip = set()
        for blk in self.blocks:
            skip.update((id(blk.attn2.to_k), id(blk.attn2.to_v)))
        return [l for l in self._token_linears() if id(l) not in skip]

    def _dgrad_groups(self):
        return [(blk.attn1.to_q, blk.attn1.to_k, blk.attn1.to_v) for blk in self.blocks]

    def prepare(self):
        super().prepare()
        C4 = self.proj_out.out_features
        C = C4 // 4
        # proj_out rows (ph, pw, c) -> (c, ph, pw)
        perm = torch.arange(C4, device=self.proj_out.weight.device).view(4, C).t().reshape(-1)
        self._proj_w = self.proj_out.weight.data[perm].contiguous()
        self._proj_b = self.proj_out.bias.data[perm].contiguous()
        self._proj_wt = self._proj_w.t().contiguous()
        self._patch_w = self.patch_embedding.weight.data.reshape(self.dim, -1).contiguous()
        for blk in self.blocks:
            blk._table = blk.scale_shift_table.data.reshape(1, 6 * self.dim).to(self.dt).contiguous()
            blk._n2_scale = (blk.norm2.weight.data.float() - 1.0).to(self.dt).reshape(1, -1).contiguous()
            blk._n2_shift = blk.norm2.bias.data.to(self.dt).reshape(1, -1).contiguous()
        self._table = self.scale_shift_table.data.reshape(1, 2 * self.dim).to(self.dt).contiguous()
        self._prepared = True
        return self

    def rope_tables(self, grid):
        """WanRotaryPosEmbed in float64 on the host, cached per (frames, rows, cols); fp32 [S,128] cos/sin, each angle
        repeated for the (2i, 2i+1) pair.  Axis dims (44, 42, 42) for head_dim 128."""
        hit = self._rope_cache.get(tuple(grid))
        if hit is not None:
            return hit
        F_, H_, W_ = grid
        hd = 128
        h_dim = w_dim = 2 * (hd // 6)
        t_dim = hd - h_dim - w_dim
        parts = []
        for dim, n, shape in ((t_dim, F_, (F_, 1, 1)), (h_dim, H_, (1, H_, 1)), (w_dim, W_, (1, 1, W_))):
            freqs = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float64)[: dim // 2] / dim))
            ang = torch.outer(torch.arange(n, dtype=torch.float64), freqs)
            parts.append(ang.view(*shape, dim // 2).expand(F_, H_, W_, dim // 2))
        ang = torch.cat(parts, dim=-1).reshape(F_ * H_ * W_, hd // 2)
        dev = self._device()
        out = (ang.cos().repeat_interleave(2, dim=1).float().contiguous().to(dev),
               ang.sin().repeat_interleave(2, dim=1).float().contiguous().to(dev))
        self._rope_cache[tuple(grid)] = out
        return out

    def lora_groups(self):
        """Same-input adapters: (q, k, v) of the self-attention, (k, v) of the cross-attention (text states)."""
        groups = []
        for blk in self.blocks:
            for lins in ((blk.attn1.to_q, blk.attn1.to_k, blk.attn1.to_v), (blk.attn2.to_k, blk.attn2.to_v)):
                mods = [l.lora for l in lins]
                if all(m is not None for m in mods):
                    groups.append(mods)
        return groups

    # ------------------------------------------------------------------ layout helpers (pure data movement)
    @staticmethod
    def pack_tokens(x):
        """[B,C,F,H,W] -> [B, F*(H/2)*(W/2), 4C] in (c, ph, pw) feature order."""
        B, C, Fr, Hh, W = x.shape
        x = x.view(B, C, Fr, Hh // 2, 2, W // 2, 2).permute(0, 2, 3, 5, 1, 4, 6)
        return x.reshape(B, Fr * (Hh // 2) * (W // 2), C * 4)

    @staticmethod
    def unpack_tokens(x, grid):
        B = x.shape[0]
        Fr, h2, w2 = grid
        C = x.shape[2] // 4
        x = x.view(B, Fr, h2, w2, C, 2, 2).permute(0, 4, 1, 2, 5, 3, 6)
        return x.reshape(B, C, Fr, h2 * 2, w2 * 2)

    # ------------------------------------------------------------------ forward
    def forward(self, hidden_states, timestep, encoder_hidden_states, return_dict=False, **kwargs):
        self._resolve_network()  # adopts / syncs a network the reference built itself (adopt.py)
        B, C, Fr, Hh, W = hidden_states.shape
        grid = (Fr, Hh // 2, W // 2)
        tokens = self.pack_tokens(hidden_states.to(self.dt)).contiguous()
        pred = self.forward_native(tokens, timestep, encoder_hidden_states, grid, save_for_backward=torch.is_grad_enabled())
        if torch.is_grad_enabled() and self.network is not None and self.network.is_active:
            from .flux import _FluxGraphFn

            pred = _FluxGraphFn.apply(pred.detach(), self, self.network.arena_p.requires_grad_(True))  # detach: the explicit graph is the only history (a torch-backed kernel table would otherwise leave autograd history of its own on pred)
        return (self.unpack_tokens(pred, grid),)

    def forward_native(self, tokens, timestep, encoder_hidden_states, grid, save_for_backward=True):
        """tokens [B,S,64] packed noisy latents, timestep [B] in 0..1000, encoder_hidden_states [B,St,text_dim];
        returns the packed prediction [B,S,64] in (c, ph, pw) order."""
        ops, dt, d, H = self.ops, self.dt, self.dim, self.heads
        if not self._prepared:
            self.prepare()
        B, S, Cin = tokens.shape
        assert S == grid[0] * grid[1] * grid[2]
        St = encoder_hidden_states.shape[1]
        M, Mt = B * S, B * St
        cos, sin = self.rope_tables(grid)
        scale = 1.0 / math.sqrt(128.0)
        eps = self.eps
        ctx = {"B": B, "S": S, "St": St, "cos": cos, "sin": sin, "blk": []} if save_for_backward else None

        # ---- conditioning (frozen; nothing saved)
        ce = self.condition_embedder
        proj = self._new(B, self.freq_dim)
        ops.timestep_embed(timestep.float().contiguous(), proj)
        h1 = self._new(B, d)
        ops.gemv_nt(proj, ce.time_embedder.linear_1.weight, h1, bias=ce.time_embedder.linear_1.bias)
        ops.ew(0, h1, h1)
        temb = self._new(B, d)
        ops.gemv_nt(h1, ce.time_embedder.linear_2.weight, temb, bias=ce.time_embedder.linear_2.bias)
        silu_temb = self._new(B, d)
        ops.ew(0, temb, silu_temb)
        tproj = self._new(B, 6 * d)
        ops.gemv_nt(silu_temb, ce.time_proj.weight, tproj, bias=ce.time_proj.bias)
        text = encoder_hidden_states.to(dt).reshape(Mt, -1).contiguous()
        e1 = self._new(Mt, d)
        ops.gemm_nt(text, ce.text_embedder.linear_1.weight, e1, bias=ce.text_embedder.linear_1.bias, flags=EPI_GELU,
                    aux_out=self._new(Mt, d))
        enc = self._new(Mt, d)
        ops.gemm_nt(e1, ce.text_embedder.linear_2.weight, enc, bias=ce.text_embedder.linear_2.bias)

        # ---- patch embedding
        x = self._new(M, d)
        ops.gemm_nt(tokens.to(dt).reshape(M, Cin), self._patch_w, x, bias=self.patch_embedding.bias)

        for blk in self.blocks:
            r = {}
            a1, a2 = blk.attn1, blk.attn2
            mod = self._new(B, 6 * d)
            ops.ew(2, tproj, mod, a=blk._table.expand(B, 6 * d))
            # 1. self-attention
            mean1, rstd1 = self._new(M, dtype=torch.float32), self._new(M, dtype=torch.float32)
            xn = self._new(M, d)
            ops.ln_mod_fwd(x, mod[:, 0:d], mod[:, d:2 * d], xn, rows_per_batch=S, mean=mean1, rstd=rstd1, eps=eps)
            qk_raw = self._new(M, 2 * d)
            qkv = self._new(M, 3 * d)
            lins = (a1.to_q, a1.to_k, a1.to_v)
            Tg = self._group_down(lins, xn, M=M, rows_per_batch=S, B=B)
            outs = (qk_raw[:, 0:d], qk_raw[:, d:2 * d], qkv[:, 2 * d:])
            r["T_qkv"] = [self._lin_fwd(lin, xn, o, M=M, rows_per_batch=S, B=B, T=Tg.get(id(lin))) for lin, o in zip(lins, outs)]
            ops.rms_full_fwd(qk_raw[:, 0:d], a1.norm_q.weight, qkv[:, 0:d], S=S, cos=cos, sin=sin, eps=eps)
            ops.rms_full_fwd(qk_raw[:, d:2 * d], a1.norm_k.weight, qkv[:, d:2 * d], S=S, cos=cos, sin=sin, eps=eps)
            o1 = self._new(M, d)
            lse1 = self._new(B, H, S, dtype=torch.float32)
            ops.attn_fwd(qkv[:, 0:d], qkv[:, d:2 * d], qkv[:, 2 * d:], o1, lse1, B=B, H=H, S=S, scale=scale)
            x1 = self._new(M, d)
            r["T_o1"] = self._lin_fwd(a1.to_out[0], o1, x1, M=M, rows_per_batch=S, B=B, flags=EPI_GATE_RES, aux_in=x,
                                      gate=mod[:, 2 * d:3 * d], gate_rows=S)
            # 2. text cross-attention
            mean2, rstd2 = self._new(M, dtype=torch.float32), self._new(M, dtype=torch.float32)
            xn2 = self._new(M, d)
            ops.ln_mod_fwd(x1, blk._n2_shift, blk._n2_scale, xn2, rows_per_batch=M, mean=mean2, rstd=rstd2, eps=eps)
            q2_raw, q2 = self._new(M, d), self._new(M, d)
            r["T_q2"] = self._lin_fwd(a2.to_q, xn2, q2_raw, M=M, rows_per_batch=S, B=B)
            ops.rms_full_fwd(q2_raw, a2.norm_q.weight, q2, S=S, eps=eps)
            k2_raw = self._new(Mt, d)
            kv2 = self._new(Mt, 2 * d)
            Tg = self._group_down((a2.to_k, a2.to_v), enc, M=Mt, rows_per_batch=St, B=B)
            r["T_k2"] = self._lin_fwd(a2.to_k, enc, k2_raw, M=Mt, rows_per_batch=St, B=B, T=Tg.get(id(a2.to_k)))
            r["T_v2"] = self._lin_fwd(a2.to_v, enc, kv2[:, d:], M=Mt, rows_per_batch=St, B=B, T=Tg.get(id(a2.to_v)))
            ops.rms_full_fwd(k2_raw, a2.norm_k.weight, kv2[:, 0:d], S=St, eps=eps)
            o2 = self._new(M, d)
            lse2 = self._new(B, H, S, dtype=torch.float32)
            ops.attn_fwd(q2, kv2[:, 0:d], kv2[:, d:], o2, lse2, B=B, H=H, S=S, scale=scale, Skv=St)
            x2 = self._new(M, d)
            r["T_o2"] = self._lin_fwd(a2.to_out[0], o2, x2, M=M, rows_per_batch=S, B=B, flags=EPI_ADD_AUX, aux_in=x1)
            # 3. feed-forward
            mean3, rstd3 = self._new(M, dtype=torch.float32), self._new(M, dtype=torch.float32)
            xn3 = self._new(M, d)
            ops.ln_mod_fwd(x2, mod[:, 3 * d:4 * d], mod[:, 4 * d:5 * d], xn3, rows_per_batch=S, mean=mean3, rstd=rstd3, eps=eps)
            ffn_dim = blk.ffn.net[0].proj.out_features
            u, hbuf = self._new(M, ffn_dim), self._new(M, ffn_dim)
            em = self._emit_t_plan(blk.ffn.net[0].proj, blk.ffn.net[2], M=M, N=ffn_dim)  # ffn.net.2's T = gelu(u) A^T from inside the GELU launch (graph.py)
            r["T_ff1"] = self._lin_fwd(blk.ffn.net[0].proj, xn3, hbuf, M=M, rows_per_batch=S, B=B, flags=EPI_GELU, aux_out=u,
                                       emit_t=None if em is None else em["args"])
            x3 = self._new(M, d)
            T2 = None if em is None else self._emit_t_finish(em, blk.ffn.net[2], M=M, rows_per_batch=S, B=B, ntiles=em["ntiles"])
            r["T_ff2"] = self._lin_fwd(blk.ffn.net[2], hbuf, x3, M=M, rows_per_batch=S, B=B, flags=EPI_GATE_RES, aux_in=x2,
                                       gate=mod[:, 5 * d:6 * d], gate_rows=S, T=T2)
            if ctx is not None:
                r.update(mod=mod, x=x, mean1=mean1, rstd1=rstd1, xn=xn, qk_raw=qk_raw, qkv=qkv, o1=o1, lse1=lse1, x1=x1,
                         mean2=mean2, rstd2=rstd2, xn2=xn2, q2_raw=q2_raw, q2=q2, k2_raw=k2_raw, kv2=kv2, o2=o2, lse2=lse2,
                         x2=x2, mean3=mean3, rstd3=rstd3, xn3=xn3, u=u, h=hbuf)
                ctx["blk"].append(r)
            x = x3

        # ---- head (frozen)
        mod_out = self._new(B, 2 * d)
        ops.ew(2, temb, mod_out[:, 0:d], a=self._table[:, 0:d].expand(B, d))
        ops.ew(2, temb, mod_out[:, d:], a=self._table[:, d:].expand(B, d))
        mean, rstd = self._new(M, dtype=torch.float32), self._new(M, dtype=torch.float32)
        xn = self._new(M, d)
        ops.ln_mod_fwd(x, mod_out[:, 0:d], mod_out[:, d:], xn, rows_per_batch=S, mean=mean, rstd=rstd, eps=eps)
        pred = self._new(M, Cin)
        ops.gemm_nt(xn, self._proj_w, pred, bias=self._proj_b)
        if ctx is not None:
            ctx.update(enc=enc, x_out=x, mod_out=mod_out, mean_out=mean, rstd_out=rstd)
            self.ctx = ctx
        return pred.view(B, S, Cin)

    # ------------------------------------------------------------------ backward
    def backward_native(self, dpred):
        """dpred [B,S,64] (c, ph, pw order): accumulates every adapter gradient into network.arena_g.  Frees the graph."""
        ops, d, H = self.ops, self.dim, self.heads
        ctx = self.ctx
        assert ctx is not None, "forward_native(save_for_backward=True) must run first"
        B, S, St = ctx["B"], ctx["S"], ctx["St"]
        M, Mt = B * S, B * St
        cos, sin, enc = ctx["cos"], ctx["sin"], ctx["enc"]
        scale = 1.0 / math.sqrt(128.0)
        eps = self.eps
        wdefer = getattr(ops, "wgrad_defer_begin", None) is not None and ops.wgrad_defer_begin(dpred.device)  # weight-gradient finishes eight at a time (flux.py)

        dxn = self._new(M, d)
        ops.gemm_nt(dpred.to(self.dt).reshape(M, -1).contiguous(), self._proj_wt, dxn)
        dx = self._new(M, d)
        ops.ln_mod_bwd(dxn, ctx["x_out"], ctx["mean_out"], ctx["rstd_out"], ctx["mod_out"][:, d:], dx, B=B, S=S)

        nblk = len(self.blocks)
        for i, (blk, r) in enumerate(zip(reversed(self.blocks), reversed(ctx["blk"]))):
            a1, a2 = blk.attn1, blk.attn2
            mod = r["mod"]
            # 3. feed-forward: x3 = x2 + c_gate * ffn(...)
            dy = self._new(M, d)
            ops.gate_bwd(dx, None, mod[:, 5 * d:6 * d], dy, None, B=B, S=S)
            du = self._new(M, r["u"].shape[1])
            self._lin_bwd(blk.ffn.net[2], dy, r["T_ff2"], r["h"], du, M=M, rows_per_batch=S, B=B, flags=EPI_DGELU, aux_in=r["u"])
            dxn3 = self._new(M, d)
            self._lin_bwd(blk.ffn.net[0].proj, du, r["T_ff1"], r["xn3"], dxn3, M=M, rows_per_batch=S, B=B)
            dx2 = self._new(M, d)
            ops.ln_mod_bwd(dxn3, r["x2"], r["mean3"], r["rstd3"], mod[:, 4 * d:5 * d], dx2, B=B, S=S, dres=dx)
            # 2. cross-attention: x2 = x1 + to_out(attn(q2, k2, v2))
            do2 = self._new(M, d)
            self._lin_bwd(a2.to_out[0], dx2, r["T_o2"], r["o2"], do2, M=M, rows_per_batch=S, B=B)
            dq2 = self._new(M, d)
            dkv2 = self._new(Mt, 2 * d)
            kv2 = r["kv2"]
            ops.attn_bwd(r["q2"], kv2[:, 0:d], kv2[:, d:], r["o2"], r["lse2"], do2, dq2, dkv2[:, 0:d], dkv2[:, d:],
                         B=B, H=H, S=S, scale=scale, Skv=St)
            dk2_raw = self._new(Mt, d)
            ops.rms_full_bwd(dkv2[:, 0:d], r["k2_raw"], a2.norm_k.weight, dk2_raw, S=St, eps=eps)
            # attn2.to_k / to_v read the frozen text states: adapter weight gradients only
            self._wgrad_only((a2.to_k, a2.to_v), (dk2_raw, dkv2[:, d:]), (r["T_k2"], r["T_v2"]), enc, Mt, St, B)
            dq2_raw = self._new(M, d)
            ops.rms_full_bwd(dq2, r["q2_raw"], a2.norm_q.weight, dq2_raw, S=S, eps=eps)
            dxn2 = self._new(M, d)
            self._lin_bwd(a2.to_q, dq2_raw, r["T_q2"], r["xn2"], dxn2, M=M, rows_per_batch=S, B=B)
            dx1 = self._new(M, d)
            ops.ln_mod_bwd(dxn2, r["x1"], r["mean2"], r["rstd2"], blk._n2_scale, dx1, B=1, S=M, dres=dx2)
            # 1. self-attention: x1 = x + gate_msa * to_out(attn(q, k, v))
            ops.gate_bwd(dx1, None, mod[:, 2 * d:3 * d], dy, None, B=B, S=S)
            do1 = self._new(M, d)
            self._lin_bwd(a1.to_out[0], dy, r["T_o1"], r["o1"], do1, M=M, rows_per_batch=S, B=B)
            qkv = r["qkv"]
            dqkv = self._new(M, 2 * d)
            dgrp = self._new(M, 3 * d)  # d[q_raw | k_raw | v] side by side: one K-concatenated data-gradient GEMM for the group (graph._group_bwd)
            ops.attn_bwd(qkv[:, 0:d], qkv[:, d:2 * d], qkv[:, 2 * d:], r["o1"], r["lse1"], do1,
                         dqkv[:, 0:d], dqkv[:, d:2 * d], dgrp[:, 2 * d:], B=B, H=H, S=S, scale=scale)
            dqk_raw = dgrp[:, :2 * d]
            qk_raw = r["qk_raw"]
            ops.rms_full_bwd(dqkv[:, 0:d], qk_raw[:, 0:d], a1.norm_q.weight, dqk_raw[:, 0:d], S=S, cos=cos, sin=sin, eps=eps)
            ops.rms_full_bwd(dqkv[:, d:2 * d], qk_raw[:, d:2 * d], a1.norm_k.weight, dqk_raw[:, d:2 * d], S=S, cos=cos, sin=sin, eps=eps)
            qkv_lins = (a1.to_q, a1.to_k, a1.to_v)
            dys = [dqk_raw[:, 0:d], dqk_raw[:, d:2 * d], dgrp[:, 2 * d:]]
            if i == nblk - 1:  # the first block's input (patch embedding) has no trainable ancestor: weight grads only
                self._wgrad_only(qkv_lins, dys, r["T_qkv"], r["xn"], M, S, B)
                r.clear()
                break
            dxn = self._new(M, d)
            self._group_bwd(qkv_lins, dys, r["T_qkv"], r["xn"], dxn, M=M, rows_per_batch=S, B=B)
            dx0 = self._new(M, d)
            ops.ln_mod_bwd(dxn, r["x"], r["mean1"], r["rstd1"], mod[:, d:2 * d], dx0, B=B, S=S, dres=dx1)
            dx = dx0
            r.clear()
            if self.grad_ready_hook is not None and i == nblk // 2 - 1:
                if wdefer:
                    ops.wgrad_defer_flush()
                self.grad_ready_hook("late")
        if wdefer:
            ops.wgrad_defer_end()
        if self.grad_ready_hook is not None:
            self.grad_ready_hook("early")
        self.ctx = None

    def grad_split_offset(self, network):
        """Arena offset from which the adapters of the 'late' piece (second half of the blocks) start."""
        first_late = len(self.blocks) - len(self.blocks) // 2
        tag = f"blocks$${first_late}$$"
        offs = [min(m.off_down, m.off_up) for m in network.unet_loras if tag in m.lora_name]
        return min(offs) if offs and len(self.blocks) > 1 else network.arena_p.numel()

    def _wgrad_only(self, lins, dys, Ts, enc, Mt, St, B):
        """Adapter weight gradients of same-input Linears whose input needs no data gradient."""
        grp = getattr(lins[0].lora, "group", None) if all(t is not None for t in Ts) else None
        if grp is not None and [id(m) for m in grp["mods"]] != [id(l.lora) for l in lins]:
            grp = None
        dTcat = self._new(Mt, 3 * grp["R"]) if grp is not None else None  # split slab layout per adapter (aitk_lora_down)
        for lin, dy, T in zip(lins, dys, Ts):
            dy = self._dora_dz(lin, dy, Mt)
            dT_out = None
            if grp is not None:
                c0 = 3 * grp["col"][id(lin.lora)]
                dT_out = dTcat[:, c0:c0 + 3 * lin.lora.rank_pad]
            self._lora_grads(lin, dy, T, enc, M=Mt, rows_per_batch=St, B=B, dT_out=dT_out)
        if grp is not None:
            self.ops.lora_wgrad(dTcat, enc, grp["g_down"], accumulate=True, M=Mt, split=grp["rp"])
