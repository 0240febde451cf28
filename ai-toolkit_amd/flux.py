"""FLUX.1 DiT forward / backward as an explicit op graph over the gfx950 kernels (host logic only).

Module tree, class names and parameter names are those of diffusers' FluxTransformer2DModel (the class the reference
loads at toolkit/stable_diffusion_model.py:667-673 and calls at 2192-2205), so diffusers checkpoints load by key and
the LoRA network produces the reference's state-dict keys.  The forward below restates the same computation as
oracle/flux_ref.py, but every arithmetic step is one C-ABI kernel call (ai-toolkit_amd/ops.py):

  Linear (+LoRA)  -> lora_down (skinny) + gemm_nt with the rank-r K-slab and fused epilogue (bias / GELU / gate-residual)
  adaLN           -> gemv_nt (B rows) + ln_mod_fwd
  attention       -> qkv_post_fwd (per-head RMSNorm + RoPE into the joint [txt|img] buffer) + attn_fwd
and the backward is written out explicitly (no autograd inside): dgrad GEMMs against pre-transposed frozen weights
(2x weight memory — 48 GB of the 288 GB HBM — buys a single NT kernel for fwd and dgrad), LoRA weight gradients by
lora_wgrad straight into the flat fp32 gradient arena, all activations kept resident (no gradient checkpointing).

`ops` is the kernel table: ai_toolkit_amd.ops on MI355X.  Tests inject oracle/ref_ops.py (same signatures, plain
torch) to check this host logic against autograd of the oracle on CPU; the product never does.
"""
import functools
import math

import torch
import torch.nn as nn



from .graph import (EPI_ACCUM, EPI_DGELU, EPI_GATE_RES, EPI_GELU, FusedGraphBase, Linear, RMSNormW, _ActInput, _Holder)  # noqa: F401


def _ada(dim, mult, dtype, device):
    h = _Holder()
    h.linear = Linear(dim, mult * dim, True, dtype, device)
    return h


def _attention(dim, heads, dim_head, added_kv, pre_only, dtype, device):
    a = _Holder()
    inner = heads * dim_head
    a.norm_q = RMSNormW(dim_head, dtype, device)
    a.norm_k = RMSNormW(dim_head, dtype, device)
    a.to_q = Linear(dim, inner, True, dtype, device)
    a.to_k = Linear(dim, inner, True, dtype, device)
    a.to_v = Linear(dim, inner, True, dtype, device)
    if added_kv:
        a.add_k_proj = Linear(dim, inner, True, dtype, device)
        a.add_v_proj = Linear(dim, inner, True, dtype, device)
        a.add_q_proj = Linear(dim, inner, True, dtype, device)
        a.norm_added_q = RMSNormW(dim_head, dtype, device)
        a.norm_added_k = RMSNormW(dim_head, dtype, device)
    if not pre_only:
        a.to_out = nn.ModuleList([Linear(inner, dim, True, dtype, device), nn.Identity()])
    if added_kv:
        a.to_add_out = Linear(inner, dim, True, dtype, device)
    return a


def _ff(dim, dtype, device):
    f = _Holder()
    g = _Holder()
    g.proj = Linear(dim, 4 * dim, True, dtype, device)
    f.net = nn.ModuleList([g, nn.Identity(), Linear(4 * dim, dim, True, dtype, device)])
    return f


class FluxTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, dtype, device):
        super().__init__()
        self.norm1 = _ada(dim, 6, dtype, device)
        self.norm1_context = _ada(dim, 6, dtype, device)
        self.attn = _attention(dim, heads, dim_head, True, False, dtype, device)
        self.ff = _ff(dim, dtype, device)
        self.ff_context = _ff(dim, dtype, device)


class FluxSingleTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, dtype, device):
        super().__init__()
        self.norm = _ada(dim, 3, dtype, device)
        self.proj_mlp = Linear(dim, 4 * dim, True, dtype, device)
        self.proj_out = Linear(5 * dim, dim, True, dtype, device)
        self.attn = _attention(dim, heads, dim_head, False, True, dtype, device)


class _TimestepEmbedding(nn.Module):
    def __init__(self, cin, dim, dtype, device):
        super().__init__()
        self.linear_1 = Linear(cin, dim, True, dtype, device)
        self.linear_2 = Linear(dim, dim, True, dtype, device)


class _TextProj(nn.Module):
    def __init__(self, cin, dim, dtype, device):
        super().__init__()
        self.linear_1 = Linear(cin, dim, True, dtype, device)
        self.linear_2 = Linear(dim, dim, True, dtype, device)


class FluxTransformer2DModel(FusedGraphBase):
    def __init__(self, in_channels=64, num_layers=19, num_single_layers=38, attention_head_dim=128,
                 num_attention_heads=24, joint_attention_dim=4096, pooled_projection_dim=768, guidance_embeds=True,
                 axes_dims_rope=(16, 56, 56), dtype=torch.bfloat16, device=None, ops=None):
        super().__init__()
        assert attention_head_dim == 128, "attention kernels are specialised for head_dim 128"
        self.config = dict(in_channels=in_channels, num_layers=num_layers, num_single_layers=num_single_layers,
                           attention_head_dim=attention_head_dim, num_attention_heads=num_attention_heads,
                           joint_attention_dim=joint_attention_dim, pooled_projection_dim=pooled_projection_dim,
                           guidance_embeds=guidance_embeds, axes_dims_rope=tuple(axes_dims_rope))
        self.heads, self.dim = num_attention_heads, num_attention_heads * attention_head_dim
        self.in_channels = in_channels
        d = self.dim
        tte = _Holder()
        tte.timestep_embedder = _TimestepEmbedding(256, d, dtype, device)
        if guidance_embeds:  # FLUX.1-schnell checkpoints (config.json: guidance_embeds false) have no guidance embedder
            tte.guidance_embedder = _TimestepEmbedding(256, d, dtype, device)
        tte.text_embedder = _TextProj(pooled_projection_dim, d, dtype, device)
        self.time_text_embed = tte
        self.context_embedder = Linear(joint_attention_dim, d, True, dtype, device)
        self.x_embedder = Linear(in_channels, d, True, dtype, device)
        self.transformer_blocks = nn.ModuleList(
            [FluxTransformerBlock(d, num_attention_heads, attention_head_dim, dtype, device) for _ in range(num_layers)])
        self.single_transformer_blocks = nn.ModuleList(
            [FluxSingleTransformerBlock(d, num_attention_heads, attention_head_dim, dtype, device) for _ in range(num_single_layers)])
        self.norm_out = _ada(d, 2, dtype, device)
        self.proj_out = Linear(d, in_channels, True, dtype, device)
        self._init_graph(ops, dtype)  # grad_ready_hook pieces: 'single' then 'double'
        self._rope_cache = {}
        self.res_dt = dtype  # storage type of the residual stream and of its gradient (set_precision)
        # recompute_gelu: the GELU outputs (the inputs of ff.net.2 / the mlp part of proj_out's input) are not kept for the backward pass —
        # their only reader there, the lora_down gradient, rebuilds them from the saved pre-activation inside aitk_lora_wgrad2 (bit for bit
        # the same values).  6.4 GB less per 1024^2 image; off until it has been timed on the GPU
        self.recompute_gelu = False

    def set_precision(self, precision="default"):
        """precision="high": hidden_states / encoder_hidden_states and their gradients are carried across the 57 blocks in fp32 (every
        other activation stays in the model dtype).  The reference has no such switch (its stream is the model dtype); this is the
        experiment of DESIGN.md section 7 against north_star's 1e-3 LoRA-delta bound."""
        assert precision in ("default", "high")
        if precision == "high" and getattr(self.ops, "__name__", "").split(".")[-1] == "ops" and hasattr(self.ops, "_capi"):
            # the HIP kernels keep the bf16 stream (the fp32 stream buys 1.7x, not the bound); the switch exists for the
            # experiment on the oracle kernel table (tools/residual_stream_experiment.py)
            raise NotImplementedError("set_precision('high'): fp32 residual stream is implemented on the oracle kernel table only")
        self.res_dt = torch.float32 if precision == "high" else self.dt
        return self

    def _newr(self, *shape):
        return torch.empty(*shape, dtype=self.res_dt, device=self._device())

    def _drop_gelu_output(self, lin):
        """may the GELU output that feeds `lin` be dropped after the forward pass?  Yes without an adapter (nothing reads it in backward) and
        with a plain LoRA adapter (aitk_lora_wgrad2); DoRA / LoKr read their input tensor elsewhere too."""
        if not self.recompute_gelu:
            return False
        lo = lin.lora
        return lo is None or not self._lora_active(lin) or (not lo.is_lokr and lo.magnitude is None and lo.rank_pad <= 64)

    # ------------------------------------------------------------------ setup
    def _token_linears(self):
        out = []
        for blk in self.transformer_blocks:
            a = blk.attn
            out += [a.to_q, a.to_k, a.to_v, a.add_q_proj, a.add_k_proj, a.add_v_proj, a.to_out[0], a.to_add_out,
                    blk.ff.net[0].proj, blk.ff.net[2], blk.ff_context.net[0].proj, blk.ff_context.net[2]]
        for blk in self.single_transformer_blocks:
            a = blk.attn
            out += [a.to_q, a.to_k, a.to_v, blk.proj_mlp, blk.proj_out]
        return out

    def _dgrad_linears(self):
        return self._token_linears() + [self.proj_out]

    def _dgrad_groups(self):
        out = []
        for blk in self.transformer_blocks:
            a = blk.attn
            out += [(a.to_q, a.to_k, a.to_v), (a.add_q_proj, a.add_k_proj, a.add_v_proj)]
        for blk in self.single_transformer_blocks:
            a = blk.attn
            out.append((a.to_q, a.to_k, a.to_v, blk.proj_mlp))
        return out

    def grad_split_offset(self, network):
        """Arena offset of the first single-stream adapter: [split, n) is final first during backward ('single')."""
        for m in network.unet_loras:
            if "single_transformer_blocks" in m.lora_name:
                return m.off_down
        return network.arena_p.numel()

    def rope_tables(self, img_ids, txt_ids):
        """FluxPosEmbed in float64 on the host, cached per bucket; fp32 [S, 128] cos/sin on device.  The cache key is the
        position grid itself: `_aitk_grid` = (h2, w2, n_txt) stamped on img_ids by make_ids / the plug-in (no device sync), or —
        for ids built elsewhere — the bytes of the ids (one device-to-host copy).  Transposed buckets (96x168 vs 168x96 latents)
        therefore never share a table."""
        grid = getattr(img_ids, "_aitk_grid", None)
        if grid is not None and (grid[0] * grid[1], grid[2]) == (img_ids.shape[0], txt_ids.shape[0]):
            key = ("grid",) + tuple(int(g) for g in grid)
        else:
            key = ("ids", tuple(img_ids.shape), tuple(txt_ids.shape),
                   img_ids.detach().to("cpu", torch.float32).contiguous().numpy().tobytes(),
                   txt_ids.detach().to("cpu", torch.float32).contiguous().numpy().tobytes())
        hit = self._rope_cache.get(key)
        if hit is not None:
            return hit
        ids = torch.cat((txt_ids, img_ids), dim=0).to("cpu", torch.float64)
        cos_out, sin_out = [], []
        for i, dd in enumerate(self.config["axes_dims_rope"]):
            freqs = 1.0 / (10000.0 ** (torch.arange(0, dd, 2, dtype=torch.float64)[: dd // 2] / dd))
            ang = torch.outer(ids[:, i], freqs)
            cos_out.append(ang.cos().repeat_interleave(2, dim=1).float())
            sin_out.append(ang.sin().repeat_interleave(2, dim=1).float())
        dev = self._device()
        out = (torch.cat(cos_out, -1).contiguous().to(dev), torch.cat(sin_out, -1).contiguous().to(dev))
        self._rope_cache[key] = out
        return out

    def lora_groups(self):
        """Adapters whose base layers read the same activation: (q, k, v) of each stream / (q, k, v, proj_mlp) of a
        single block.  Passed to FusedLoRANetwork.build_arena so their lora_down matrices are adjacent."""
        groups = []

        def add(lins):
            mods = [l.lora for l in lins]
            if all(m is not None for m in mods):
                groups.append(mods)

        for blk in self.transformer_blocks:
            a = blk.attn
            add((a.to_q, a.to_k, a.to_v))
            add((a.add_q_proj, a.add_k_proj, a.add_v_proj))
        for blk in self.single_transformer_blocks:
            a = blk.attn
            add((a.to_q, a.to_k, a.to_v, blk.proj_mlp))
        return groups

    # ------------------------------------------------------------------ forward
    def forward(self, hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids, guidance,
                return_dict=False, **kwargs):
        self._resolve_network()  # adopts / syncs a network the reference built itself (adopt.py)
        pred = self.forward_native(hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids,
                                   guidance, save_for_backward=torch.is_grad_enabled())
        if torch.is_grad_enabled() and self.network is not None and self.network.is_active:
            pred = _FluxGraphFn.apply(pred.detach(), self, self.network.arena_p.requires_grad_(True))  # detach: the explicit graph is the only history (a torch-backed kernel table would otherwise leave autograd history of its own on pred)
        return (pred,)

    def forward_native(self, hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids,
                       guidance, save_for_backward=True):
        ops, dt, d, H = self.ops, self.dt, self.dim, self.heads
        if not self._prepared:
            self.prepare()
        B, Si, Cin = hidden_states.shape
        St = encoder_hidden_states.shape[1]
        S = St + Si
        Mi, Mt, Mj = B * Si, B * St, B * S
        cos, sin = self.rope_tables(img_ids, txt_ids)
        scale = 1.0 / math.sqrt(128.0)
        ctx = {"B": B, "Si": Si, "St": St, "cos": cos, "sin": sin, "dbl": [], "sgl": []} if save_for_backward else None

        # ---- conditioning vector (no trainable ancestors; nothing saved)
        tte = self.time_text_embed
        # the pinned diffusers casts timestep/guidance to the model dtype BEFORE the *1000
        t_eff = (timestep.to(dt) * 1000).float().contiguous()
        # guidance = None: the reference's bypass_flux_guidance (toolkit/models/flux.py:9-35, the FLUX.1-schnell training adapter path,
        # stable_diffusion_model.py:2182-2183): the conditioning vector is timestep + pooled text only
        embedders = [(tte.timestep_embedder, t_eff)]
        if guidance is not None and self.config["guidance_embeds"]:  # diffusers ignores `guidance` without the embedder (CombinedTimestepTextProjEmbeddings)
            embedders.append((tte.guidance_embedder, (guidance.to(dt) * 1000).float().contiguous()))
        temb = self._new(B, d)
        first = True
        for emb, src in embedders:
            proj = self._new(B, 256)
            ops.timestep_embed(src, proj)
            h1 = self._new(B, d)
            ops.gemv_nt(proj, emb.linear_1.weight, h1, bias=emb.linear_1.bias)
            ops.ew(0, h1, h1)
            ops.gemv_nt(h1, emb.linear_2.weight, temb, bias=emb.linear_2.bias, accumulate=not first)
            first = False
        h1 = self._new(B, d)
        pooled = pooled_projections.to(dt).contiguous()
        ops.gemv_nt(pooled, tte.text_embedder.linear_1.weight, h1, bias=tte.text_embedder.linear_1.bias)
        ops.ew(0, h1, h1)
        ops.gemv_nt(h1, tte.text_embedder.linear_2.weight, temb, bias=tte.text_embedder.linear_2.bias, accumulate=True)
        silu_temb = self._new(B, d)
        ops.ew(0, temb, silu_temb)

        # ---- token embedders
        x_img = self._newr(Mi, d)
        ops.gemm_nt(hidden_states.to(dt).reshape(Mi, Cin), self.x_embedder.weight, x_img, bias=self.x_embedder.bias)
        x_txt = self._newr(Mt, d)
        ops.gemm_nt(encoder_hidden_states.to(dt).reshape(Mt, -1), self.context_embedder.weight, x_txt,
                    bias=self.context_embedder.bias)

        # ---- double-stream blocks
        for blk in self.transformer_blocks:
            self._q8_reset()
            rec = {}
            qkv_j = self._new(Mj, 3 * d)
            o_j = self._new(Mj, d)
            lse = self._new(B, H, S, dtype=torch.float32)
            streams = (("img", x_img, Mi, Si, St, blk.norm1, (blk.attn.to_q, blk.attn.to_k, blk.attn.to_v),
                        (blk.attn.norm_q, blk.attn.norm_k)),
                       ("txt", x_txt, Mt, St, 0, blk.norm1_context,
                        (blk.attn.add_q_proj, blk.attn.add_k_proj, blk.attn.add_v_proj),
                        (blk.attn.norm_added_q, blk.attn.norm_added_k)))

            def qkv_stream(name, x, M, Ss, s_off, norm1, qkv_lins, qk_norms):
                r = {}
                mod, r["T_mod"] = self._ada_fwd(norm1.linear, silu_temb, B)
                r["mod"] = mod
                mean, rstd = self._new(M, dtype=torch.float32), self._new(M, dtype=torch.float32)
                xn = self._new(M, d)
                ops.ln_mod_fwd(x, mod[:, 0:d], mod[:, d:2 * d], xn, rows_per_batch=Ss, mean=mean, rstd=rstd)
                qkv_raw = self._new(M, 3 * d)
                Tg = self._group_down(qkv_lins, xn, M=M, rows_per_batch=Ss, B=B)
                r["T_qkv"] = [self._lin_fwd(lin, xn, qkv_raw[:, j * d:(j + 1) * d], M=M, rows_per_batch=Ss, B=B,
                                            T=Tg.get(id(lin))) for j, lin in enumerate(qkv_lins)]
                jobs = [dict(src=qkv_raw[:, 0:d], dst=qkv_j[:, 0:d], weight=qk_norms[0].weight),
                        dict(src=qkv_raw[:, d:2 * d], dst=qkv_j[:, d:2 * d], weight=qk_norms[1].weight),
                        dict(src=qkv_raw[:, 2 * d:], dst=qkv_j[:, 2 * d:], weight=None)]
                ops.qkv_post_fwd(jobs, cos, sin, B=B, H=H, S_src=Ss, S_dst=S, s_off=s_off)
                r.update(x=x, mean1=mean, rstd1=rstd, xn=xn, qkv_raw=qkv_raw)
                rec[name] = r

            # image and text stream are independent up to the joint attention: their launches are merged, equal-shape GEMMs grouped
            self._paired([functools.partial(qkv_stream, *st) for st in streams])
            ops.attn_fwd(qkv_j[:, 0:d], qkv_j[:, d:2 * d], qkv_j[:, 2 * d:], o_j, lse, B=B, H=H, S=S, scale=scale)
            rec.update(qkv_j=qkv_j, o_j=o_j, lse=lse)
            outs = {}
            def mlp_stream(name, M, Ss, s_off, out_lin, ff):
                r = rec[name]
                mod, x = r["mod"], r["x"]
                seg = (Ss, S * d)
                o_view = o_j[s_off:]
                y_attn = self._new(M, d)
                x1 = self._newr(M, d)
                r["T_o"] = self._lin_fwd(out_lin, o_view, x1, M=M, rows_per_batch=Ss, B=B, flags=EPI_GATE_RES,
                                         aux_out=y_attn, aux_in=x, gate=mod[:, 2 * d:3 * d], gate_rows=Ss, a_seg=seg)
                mean, rstd = self._new(M, dtype=torch.float32), self._new(M, dtype=torch.float32)
                xn2 = self._new(M, d)
                ops.ln_mod_fwd(x1, mod[:, 3 * d:4 * d], mod[:, 4 * d:5 * d], xn2, rows_per_batch=Ss, mean=mean, rstd=rstd)
                u = self._new(M, 4 * d)
                hbuf = self._new(M, 4 * d)
                em = self._emit_t_plan(ff.net[0].proj, ff.net[2], M=M, N=4 * d)  # ff.net.2's T = gelu(u) A^T from inside the GELU launch
                r["T_ff1"] = self._lin_fwd(ff.net[0].proj, xn2, hbuf, M=M, rows_per_batch=Ss, B=B, flags=EPI_GELU, aux_out=u,
                                           emit_t=None if em is None else em["args"])
                y_ff = self._new(M, d)
                x2 = self._newr(M, d)
                T2 = None if em is None else self._emit_t_finish(em, ff.net[2], M=M, rows_per_batch=Ss, B=B, ntiles=em["ntiles"])
                r["T_ff2"] = self._lin_fwd(ff.net[2], hbuf, x2, M=M, rows_per_batch=Ss, B=B, flags=EPI_GATE_RES,
                                           aux_out=y_ff, aux_in=x1, gate=mod[:, 5 * d:6 * d], gate_rows=Ss, T=T2)
                r.update(y_attn=y_attn, x1=x1, mean2=mean, rstd2=rstd, xn2=xn2, u=u, h=None if self._drop_gelu_output(ff.net[2]) else hbuf,
                         y_ff=y_ff)
                outs[name] = x2

            self._paired([functools.partial(mlp_stream, "img", Mi, Si, St, blk.attn.to_out[0], blk.ff),
                          functools.partial(mlp_stream, "txt", Mt, St, 0, blk.attn.to_add_out, blk.ff_context)])
            x_img, x_txt = outs["img"], outs["txt"]
            if ctx is not None:
                ctx["dbl"].append(rec)

        # ---- joint stream
        x = self._newr(Mj, d)
        xv = x.view(B, S, d)
        for b in range(B):
            ops.copy_rows(xv[b, :St], x_txt.view(B, St, d)[b])
            ops.copy_rows(xv[b, St:], x_img.view(B, Si, d)[b])
        for blk in self.single_transformer_blocks:
            self._q8_reset()
            r = {}
            mod, r["T_mod"] = self._ada_fwd(blk.norm.linear, silu_temb, B)
            mean, rstd = self._new(Mj, dtype=torch.float32), self._new(Mj, dtype=torch.float32)
            xn = self._new(Mj, d)
            ops.ln_mod_fwd(x, mod[:, 0:d], mod[:, d:2 * d], xn, rows_per_batch=S, mean=mean, rstd=rstd)
            qkv_raw = self._new(Mj, 3 * d)
            a = blk.attn
            Tg = self._group_down((a.to_q, a.to_k, a.to_v, blk.proj_mlp), xn, M=Mj, rows_per_batch=S, B=B)
            r["T_qkv"] = [self._lin_fwd(lin, xn, qkv_raw[:, j * d:(j + 1) * d], M=Mj, rows_per_batch=S, B=B, T=Tg.get(id(lin)))
                          for j, lin in enumerate((a.to_q, a.to_k, a.to_v))]
            cat = self._new(Mj, 5 * d)
            u = self._new(Mj, 4 * d)
            # proj_out's T over its [attn | gelu(mlp)] input: the GELU columns from inside this launch, the attention columns as one more tile below
            em = self._emit_t_plan(blk.proj_mlp, blk.proj_out, M=Mj, N=4 * d, col0=d, extra_tiles=1)
            r["T_mlp"] = self._lin_fwd(blk.proj_mlp, xn, cat[:, d:], M=Mj, rows_per_batch=S, B=B, flags=EPI_GELU, aux_out=u,
                                       T=Tg.get(id(blk.proj_mlp)), emit_t=None if em is None else em["args"])
            # q, k: RMSNorm + RoPE into qkv_j; v needs neither and already sits in joint row order: attention reads it where the
            # projection wrote it (the third, copy-only job of qkv_post is gone: 12 KB of HBM traffic per token and layer, both directions)
            qkv_j = self._new(Mj, 2 * d)
            jobs = [dict(src=qkv_raw[:, 0:d], dst=qkv_j[:, 0:d], weight=a.norm_q.weight),
                    dict(src=qkv_raw[:, d:2 * d], dst=qkv_j[:, d:2 * d], weight=a.norm_k.weight)]
            ops.qkv_post_fwd(jobs, cos, sin, B=B, H=H, S_src=S, S_dst=S, s_off=0)
            lse = self._new(B, H, S, dtype=torch.float32)
            drop = self._drop_gelu_output(blk.proj_out)
            if drop:  # the attention output outlives `cat`: it gets its own buffer and is copied into the GEMM operand
                o_buf = self._new(Mj, d)
                ops.attn_fwd(qkv_j[:, 0:d], qkv_j[:, d:2 * d], qkv_raw[:, 2 * d:], o_buf, lse, B=B, H=H, S=S, scale=scale)
                ops.copy_rows(cat[:, 0:d], o_buf)
            else:
                o_buf = cat[:, 0:d]
                ops.attn_fwd(qkv_j[:, 0:d], qkv_j[:, d:2 * d], qkv_raw[:, 2 * d:], o_buf, lse, B=B, H=H, S=S, scale=scale)
            y = self._new(Mj, d)
            x_new = self._newr(Mj, d)
            T_out = None
            if em is not None:
                lo_out = blk.proj_out.lora
                ops.lora_down_raw(cat[:, 0:d], lo_out.sh_down[:, 0:d], em["partial"][em["ntiles"]], p_lo=lo_out.sh_down_lo[:, 0:d], M=Mj)
                T_out = self._emit_t_finish(em, blk.proj_out, M=Mj, rows_per_batch=S, B=B, ntiles=em["ntiles"] + 1)
            r["T_out"] = self._lin_fwd(blk.proj_out, cat, x_new, M=Mj, rows_per_batch=S, B=B, flags=EPI_GATE_RES,
                                       aux_out=y, aux_in=x, gate=mod[:, 2 * d:3 * d], gate_rows=S, T=T_out)
            r.update(mod=mod, x=x, mean=mean, rstd=rstd, xn=xn, qkv_raw=qkv_raw, cat=None if drop else cat, o=o_buf, u=u, qkv_j=qkv_j, lse=lse, y=y)
            if ctx is not None:
                ctx["sgl"].append(r)
            x = x_new

        # ---- output head (frozen): AdaLayerNormContinuous ([scale, shift]) + proj_out on the image tokens
        self._q8_reset()
        x_out = self._newr(Mi, d)
        xv = x.view(B, S, d)
        for b in range(B):
            ops.copy_rows(x_out.view(B, Si, d)[b], xv[b, St:])
        mod_out = self._new(B, 2 * d)
        ops.gemv_nt(silu_temb, self.norm_out.linear.weight, mod_out, bias=self.norm_out.linear.bias)
        mean, rstd = self._new(Mi, dtype=torch.float32), self._new(Mi, dtype=torch.float32)
        xn = self._new(Mi, d)
        ops.ln_mod_fwd(x_out, mod_out[:, d:2 * d], mod_out[:, 0:d], xn, rows_per_batch=Si, mean=mean, rstd=rstd)
        pred = self._new(Mi, Cin)
        ops.gemm_nt(xn, self.proj_out.weight, pred, bias=self.proj_out.bias)
        if ctx is not None:
            ctx.update(silu_temb=silu_temb, x_out=x_out, mod_out=mod_out, mean_out=mean, rstd_out=rstd)
            self.ctx = ctx
        return pred.view(B, Si, Cin)

    # ------------------------------------------------------------------ backward
    def backward_native(self, dpred):
        """dpred [B, Si, Cin]: accumulates every adapter gradient into network.arena_g (fp32).  Frees the saved graph."""
        ops, d, H = self.ops, self.dim, self.heads
        ctx = self.ctx
        assert ctx is not None, "forward_native(save_for_backward=True) must run first"
        B, Si, St = ctx["B"], ctx["Si"], ctx["St"]
        S = St + Si
        Mi, Mt, Mj = B * Si, B * St, B * S
        cos, sin, silu_temb = ctx["cos"], ctx["sin"], ctx["silu_temb"]
        scale = 1.0 / math.sqrt(128.0)
        Cin = self.in_channels

        # the finish passes of the lora_down / adaLN weight gradients are collected and go out eight at a time (ops.wgrad_defer_begin); flushed before
        # the gradient is handed on (all-reduce pieces, optimizer)
        wdefer = getattr(ops, "wgrad_defer_begin", None) is not None and ops.wgrad_defer_begin(dpred.device)
        # ---- head
        dxn = self._new(Mi, d)
        ops.gemm_nt(dpred.to(self.dt).reshape(Mi, Cin).contiguous(), self.proj_out.weight_t, dxn)
        dx_out = self._newr(Mi, d)
        ops.ln_mod_bwd(dxn, ctx["x_out"], ctx["mean_out"], ctx["rstd_out"], ctx["mod_out"][:, 0:d], dx_out, B=B, S=Si)
        dx = torch.zeros(Mj, d, dtype=self.res_dt, device=dx_out.device)
        dxv = dx.view(B, S, d)
        for b in range(B):
            ops.copy_rows(dxv[b, St:], dx_out.view(B, Si, d)[b])

        # ---- single-stream blocks (reverse)
        for blk, r in zip(reversed(self.single_transformer_blocks), reversed(ctx["sgl"])):
            self._q8_reset()
            mod = r["mod"]
            dmod = self._new(B, 3 * d)
            dy = self._new(Mj, d)
            ops.gate_bwd(dx, r["y"], mod[:, 2 * d:3 * d], dy, dmod[:, 2 * d:3 * d], B=B, S=S)
            dcat_o = self._new(Mj, d)
            # d[q | k | v | mlp pre-activation] side by side: the four layers read the same xn, so their data gradient is ONE GEMM over the
            # concatenated 7 d output channels (graph._group_bwd)
            dgrp = self._new(Mj, 7 * d)
            du = dgrp[:, 3 * d:]
            # proj_out: adapter grads once, then the two column ranges of d[attn | mlp]
            dy = self._dora_dz(blk.proj_out, dy, Mj)
            cat_in = r["cat"] if r["cat"] is not None else _ActInput(r["o"], r["u"], "gelu")
            dT = self._lora_grads(blk.proj_out, dy, r["T_out"], cat_in, M=Mj, rows_per_batch=S, B=B)
            self._lin_dgrad(blk.proj_out, dy, dT, dcat_o, M=Mj, w_rows=(0, d))
            self._lin_dgrad(blk.proj_out, dy, dT, du, M=Mj, w_rows=(d, 5 * d), flags=EPI_DGELU, aux_in=r["u"])
            qkv_j = r["qkv_j"]
            qkv_raw = r["qkv_raw"]
            dqkv_j = self._new(Mj, 2 * d)
            dqkv_raw = dgrp[:, :3 * d]
            # dV goes straight to the raw-side gradient buffer (v was never copied), dQ / dK through the RoPE / RMSNorm backward
            ops.attn_bwd(qkv_j[:, 0:d], qkv_j[:, d:2 * d], qkv_raw[:, 2 * d:], r["o"], r["lse"], dcat_o,
                         dqkv_j[:, 0:d], dqkv_j[:, d:2 * d], dqkv_raw[:, 2 * d:], B=B, H=H, S=S, scale=scale)
            a = blk.attn
            jobs = [dict(src=dqkv_raw[:, 0:d], dst=dqkv_j[:, 0:d], weight=a.norm_q.weight, raw=qkv_raw[:, 0:d]),
                    dict(src=dqkv_raw[:, d:2 * d], dst=dqkv_j[:, d:2 * d], weight=a.norm_k.weight, raw=qkv_raw[:, d:2 * d])]
            ops.qkv_post_bwd(jobs, cos, sin, B=B, H=H, S_src=S, S_dst=S, s_off=0)
            dxn = self._new(Mj, d)
            self._group_bwd((a.to_q, a.to_k, a.to_v, blk.proj_mlp),
                            [dqkv_raw[:, 0:d], dqkv_raw[:, d:2 * d], dqkv_raw[:, 2 * d:], du],
                            r["T_qkv"] + [r["T_mlp"]], r["xn"], dxn, M=Mj, rows_per_batch=S, B=B)
            dx_prev = self._newr(Mj, d)
            ops.ln_mod_bwd(dxn, r["x"], r["mean"], r["rstd"], mod[:, d:2 * d], dx_prev, B=B, S=S, dres=dx,
                           dshift=dmod[:, 0:d], dscale=dmod[:, d:2 * d])
            self._ada_bwd(blk.norm.linear, dmod, r["T_mod"], silu_temb, B)
            dx = dx_prev
            r.clear()

        if self.grad_ready_hook is not None:
            if wdefer:
                ops.wgrad_defer_flush()
            self.grad_ready_hook("single")

        # ---- split the joint gradient
        dx_img, dx_txt = self._newr(Mi, d), self._newr(Mt, d)
        dxv = dx.view(B, S, d)
        for b in range(B):
            ops.copy_rows(dx_txt.view(B, St, d)[b], dxv[b, :St])
            ops.copy_rows(dx_img.view(B, Si, d)[b], dxv[b, St:])

        # ---- double-stream blocks (reverse)
        for blk, rec in zip(reversed(self.transformer_blocks), reversed(ctx["dbl"])):
            self._q8_reset()
            do_j = self._new(Mj, d)
            grads = {"img": dx_img, "txt": dx_txt}
            dmods, dx1s = {}, {}
            def mlp_stream_bwd(name, M, Ss, s_off, out_lin, ff):
                r = rec[name]
                mod = r["mod"]
                dx2 = grads[name]
                dmod = self._new(B, 6 * d)
                dy = self._new(M, d)
                ops.gate_bwd(dx2, r["y_ff"], mod[:, 5 * d:6 * d], dy, dmod[:, 5 * d:6 * d], B=B, S=Ss)
                du = self._new(M, 4 * d)
                h_in = r["h"] if r["h"] is not None else _ActInput(None, r["u"], "gelu")
                self._lin_bwd(ff.net[2], dy, r["T_ff2"], h_in, du, M=M, rows_per_batch=Ss, B=B, flags=EPI_DGELU, aux_in=r["u"])
                dxn2 = self._new(M, d)
                self._lin_bwd(ff.net[0].proj, du, r["T_ff1"], r["xn2"], dxn2, M=M, rows_per_batch=Ss, B=B)
                dx1 = self._newr(M, d)
                ops.ln_mod_bwd(dxn2, r["x1"], r["mean2"], r["rstd2"], mod[:, 4 * d:5 * d], dx1, B=B, S=Ss, dres=dx2,
                               dshift=dmod[:, 3 * d:4 * d], dscale=dmod[:, 4 * d:5 * d])
                ops.gate_bwd(dx1, r["y_attn"], mod[:, 2 * d:3 * d], dy, dmod[:, 2 * d:3 * d], B=B, S=Ss)
                seg = (Ss, S * d)
                self._lin_bwd(out_lin, dy, r["T_o"], rec["o_j"][s_off:], do_j[s_off:], M=M, rows_per_batch=Ss, B=B,
                              x_seg=seg, dx_seg=seg)
                dmods[name], dx1s[name] = dmod, dx1

            self._paired([functools.partial(mlp_stream_bwd, "img", Mi, Si, St, blk.attn.to_out[0], blk.ff),
                          functools.partial(mlp_stream_bwd, "txt", Mt, St, 0, blk.attn.to_add_out, blk.ff_context)])
            qkv_j = rec["qkv_j"]
            dqkv_j = self._new(Mj, 3 * d)
            ops.attn_bwd(qkv_j[:, 0:d], qkv_j[:, d:2 * d], qkv_j[:, 2 * d:], rec["o_j"], rec["lse"], do_j,
                         dqkv_j[:, 0:d], dqkv_j[:, d:2 * d], dqkv_j[:, 2 * d:], B=B, H=H, S=S, scale=scale)
            new_grads = {}
            def qkv_stream_bwd(name, M, Ss, s_off, norm1, qkv_lins, qk_norms):
                r = rec[name]
                mod, dmod = r["mod"], dmods[name]
                qkv_raw = r["qkv_raw"]
                dqkv_raw = self._new(M, 3 * d)
                jobs = [dict(src=dqkv_raw[:, 0:d], dst=dqkv_j[:, 0:d], weight=qk_norms[0].weight, raw=qkv_raw[:, 0:d]),
                        dict(src=dqkv_raw[:, d:2 * d], dst=dqkv_j[:, d:2 * d], weight=qk_norms[1].weight, raw=qkv_raw[:, d:2 * d]),
                        dict(src=dqkv_raw[:, 2 * d:], dst=dqkv_j[:, 2 * d:], weight=None)]
                ops.qkv_post_bwd(jobs, cos, sin, B=B, H=H, S_src=Ss, S_dst=S, s_off=s_off)
                dxn = self._new(M, d)
                self._group_bwd(qkv_lins, [dqkv_raw[:, j * d:(j + 1) * d] for j in range(3)], r["T_qkv"], r["xn"], dxn,
                                M=M, rows_per_batch=Ss, B=B)
                dx0 = self._newr(M, d)
                ops.ln_mod_bwd(dxn, r["x"], r["mean1"], r["rstd1"], mod[:, d:2 * d], dx0, B=B, S=Ss, dres=dx1s[name],
                               dshift=dmod[:, 0:d], dscale=dmod[:, d:2 * d])
                self._ada_bwd(norm1.linear, dmod, r["T_mod"], silu_temb, B)
                new_grads[name] = dx0

            self._paired([functools.partial(qkv_stream_bwd, "img", Mi, Si, St, blk.norm1, (blk.attn.to_q, blk.attn.to_k, blk.attn.to_v),
                                            (blk.attn.norm_q, blk.attn.norm_k)),
                          functools.partial(qkv_stream_bwd, "txt", Mt, St, 0, blk.norm1_context,
                                            (blk.attn.add_q_proj, blk.attn.add_k_proj, blk.attn.add_v_proj),
                                            (blk.attn.norm_added_q, blk.attn.norm_added_k))])
            dx_img, dx_txt = new_grads["img"], new_grads["txt"]
            rec.clear()
        if wdefer:
            ops.wgrad_defer_end()
        if self.grad_ready_hook is not None:
            self.grad_ready_hook("double")
        self._q8_reset()
        self.ctx = None


class _FluxGraphFn(torch.autograd.Function):
    """Lets reference-style code call loss.backward(): the explicit backward runs when autograd reaches pred.
    Adapter gradients are written into the arena views that back every lora_down / lora_up Parameter's .grad."""

    @staticmethod
    def forward(ctx_, pred, model, arena_p):
        ctx_.model = model
        # this forward's saved graph travels with this node: a second grad-enabled forward before loss.backward() (the trainer's preservation
        # prediction, SDTrainer.py:2182-2219) must not replace it
        ctx_.graph = model._take_graph_state()
        return pred.clone()

    @staticmethod
    def backward(ctx_, dpred):
        net = ctx_.model.network
        if net.grads_dropped():
            # the trainer's optimizer.zero_grad(set_to_none=True) (SDTrainer.py:2249, 2288) dropped the .grad views: "none" means
            # zero, so the arena they alias is cleared and the views are re-attached before this backward accumulates into it
            net.zero_grad_arena()
        arm = getattr(net, "before_backward", None)
        if arm is not None:
            arm(ctx_.model)  # adopted reference networks under data parallelism: the gradient all-reduce goes out in pieces during this backward
        if ctx_.graph is None:
            raise RuntimeError("backward through the same native prediction twice: its saved graph was released by the first pass (retain_graph is not supported)")
        ctx_.model._put_graph_state(ctx_.graph)
        ctx_.graph = None
        ctx_.model.backward_native(dpred)
        hook = getattr(net, "after_backward", None)
        if hook is not None:
            hook()  # adopted reference networks: the data-parallel average of the gradient arena (adopt.AdoptedNetwork.after_backward)
        return None, None, None
