"""Shared host-side machinery of the fused DiT graphs (FLUX.1: flux.py, Wan2.1: wan.py).

`FusedGraphBase` owns everything that is the same for every transformer the LoRA hot path runs on: the frozen `Linear`
holder (bf16 weight [out,in] + transposed copy for dgrad, or weight-only fp8), the LoRA-fused forward of one Linear
(lora_down + gemm_nt with the rank-r K-slab), its backward (adapter weight gradients into the flat fp32 arena + dgrad
GEMM), the same-input group launches, and the small-batch (GEMV) modulation projections.  Sub-classes write only the
model's op graph.  `ops` is the kernel table: ai_toolkit_amd.ops on MI355X; tests inject oracle/ref_ops.py (same
signatures, plain torch) to check the host logic against autograd of the oracle on CPU; the product never does.
"""
import os

import torch
import torch.nn as nn

EPI_ACCUM, EPI_GELU, EPI_DGELU, EPI_GATE_RES, EPI_ADD_AUX = 2, 4, 8, 16, 64


class Linear(nn.Module):
    """Frozen base projection: holds weight [out,in] / bias in the model dtype; `lora` is set by LoRAModule.apply_to."""

    def __init__(self, in_features, out_features, bias=True, dtype=torch.bfloat16, device=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features, dtype=dtype, device=device), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(out_features, dtype=dtype, device=device), requires_grad=False) if bias else None
        object.__setattr__(self, "lora", None)
        self.weight_t = None  # [in,out] copy for the data-gradient GEMM (built by prepare())
        self._dgroup = None   # (W^T_cat [in, sum(out)], column offset, ids of the group's Linears) when laid out with its same-input group
        self.qweight = self.qweight_t = self.wscale = None  # weight-only fp8 base (quantize_base_fp8)
        self._register_load_state_dict_pre_hook(_note_loaded_keys, with_module=True)
        self.register_load_state_dict_post_hook(_linear_weights_loaded)

    def forward(self, x):
        raise RuntimeError("fused path: Linear is executed inside the model's explicit graph (forward_native)")

    def __setattr__(self, name, value):
        # The reference attaches an adapter by swapping the wrapped layer's forward (`LoRAModule.apply_to`, toolkit/lora_special.py:132-135:
        # `self.org_forward = org.forward; org.forward = self.forward`).  The explicit graph never calls Linear.forward, so the assignment is
        # taken as what it means: the module that owns the new forward becomes this layer's adapter (adopt.register_foreign_adapter validates it
        # and raises for anything the fused graph cannot run — never a silent base-only model).
        if name == "forward":
            from .adopt import register_foreign_adapter

            register_foreign_adapter(self, value)
            return
        super().__setattr__(name, value)


def _note_loaded_keys(module, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
    """load_state_dict pre-hook (it knows the layer's key prefix, the post-hook does not): did THIS layer's weight / bias arrive?  A parent
    `load_state_dict(strict=False)` that carries nothing for a layer must leave its derived operands — and an fp8-quantised layer — alone."""
    object.__setattr__(module, "_sd_touched", (prefix + "weight") in state_dict or (prefix + "bias") in state_dict)


def _linear_weights_loaded(module, incompatible_keys):
    """`load_state_dict` on a prepared layer (the reference's `merge_in` writes the merged weight back this way, toolkit/network_mixins.py:
    452-462): the transposed copy the data-gradient GEMM reads follows the new weight."""
    if not module.__dict__.pop("_sd_touched", True):
        return
    if getattr(module, "qweight", None) is not None:
        raise NotImplementedError("load_state_dict into a weight-only fp8 Linear: merge through FusedLoRANetwork.merge_in (re-quantises), "
                                  "or reload the base model")
    wt = getattr(module, "weight_t", None)
    if wt is not None:
        with torch.no_grad():
            wt.copy_(module.weight.data.t())


class RMSNormW(nn.Module):
    def __init__(self, dim, dtype, device):
        super().__init__()
        self.eps = 1e-6
        self.weight = nn.Parameter(torch.ones(dim, dtype=dtype, device=device), requires_grad=False)


class _Holder(nn.Module):
    pass


class _DoraPS:
    """Forward state of a DoRA layer under per-sample multipliers (slider-style batches; toolkit/network_mixins.py:313-340): the reference
    output is c*(W x + m_mean B A x) + b + (m_b - m_mean) B A x, i.e. the column-scaled product carries the MEAN multiplier and a second,
    un-scaled rank-r term carries each sample's deviation.  bar / delta = the two rank-space activations (split slabs)."""

    def __init__(self, bar, delta, mbar, dvec, rpb):
        self.bar, self.delta, self.mbar, self.dvec, self.rpb = bar, delta, mbar, dvec, rpb


class _ActInput:
    """A layer input the backward pass does not hold as one tensor: [g | act(g2)] (g may be None).  With `recompute_gelu` the GELU outputs
    are dropped after the forward pass; aitk_lora_wgrad2 forms lora_down.weight.grad from the saved pre-activation instead."""

    def __init__(self, g, g2, act):
        self.g, self.g2, self.act = g, g2, act


@torch.no_grad()
def quantize_linear_fp8(lin, w):
    """lin.qweight / qweight_t / wscale from a [out, in] weight: OCP e4m3 bytes with one fp32 scale per output channel
    (amax / 448) — used at load time (quantize_base_fp8) and when an adapter is merged into a quantised base
    (toolkit/network_mixins.py:452-459: merged weights are re-quantised so the model stays quantised)."""
    w = w.float()
    amax = w.abs().amax(dim=1).clamp_min(1e-12)
    # tensor / tensor: an IEEE division on every device (a Python-scalar divisor becomes a multiply by the rounded reciprocal on the
    # GPU, 1 ulp off on about half the rows, which moves exact ties of bf16 weights to the neighbouring code)
    scale = (amax / torch.full_like(amax, 448.0)).contiguous()
    q = (w / scale[:, None]).to(torch.float8_e4m3fn)
    lin.qweight = q.view(torch.uint8).contiguous()
    lin.qweight_t = q.view(torch.uint8).t().contiguous()
    lin.wscale = scale


_KRON = object()  # marker travelling where the rank-r activation T / its gradient dT travel for LoRA layers


class FusedGraphBase(nn.Module):
    """Sub-classes set: self.ops, self.dt (model dtype), and implement _token_linears() (every Linear that runs as a
    token GEMM and needs the transposed / fp8 copies)."""

    def _init_graph(self, ops, dtype):
        self.ops = ops
        self.dt = dtype
        object.__setattr__(self, "network", None)
        self._prepared = False
        self.ctx = None
        self.grad_ready_hook = None  # called with a piece name when that part of the adapter grads is final

    def _device(self):
        return next(self.parameters()).device

    # ---- what ONE forward saved for its backward, as an object the autograd bridge can hold.  The reference's trainer may run several
    # grad-enabled predictions before one loss.backward() (diff_output_preservation / blank_prompt_preservation: the training prediction and a
    # preservation prediction with other embeddings, extensions_built_in/sd_trainer/SDTrainer.py:2182-2219): each bridge node keeps its own
    # forward's state and puts it back in front of its backward_native.  `self.ctx` (FLUX, Wan) / `self.tape` (UNets) stay the slot
    # forward_native / backward_native use, so the fused step (one forward, one backward) is untouched.
    _graph_slots = ("ctx",)

    def _dora_modules(self):
        net = self.network
        return [m for m in net.get_all_modules() if getattr(m, "magnitude", None) is not None] if net is not None else []

    def _take_graph_state(self):
        state = {k: getattr(self, k, None) for k in self._graph_slots}
        for k in self._graph_slots:
            setattr(self, k, None)
        # DoRA layers park this forward's linear output on the adapter (d magnitude needs it): it belongs to the forward, not to the layer
        state["_dora_y"] = [(m, m.y_lin) for m in self._dora_modules() if getattr(m, "y_lin", None) is not None]
        for m, _ in state["_dora_y"]:
            m.y_lin = None
        return state

    def _put_graph_state(self, state):
        for k in self._graph_slots:
            setattr(self, k, state.get(k))
        for m, y in state.get("_dora_y", ()):
            m.y_lin = y

    # diffusers' ModelMixin surface the reference reads off the denoiser (toolkit/models/base_model.py:961-968: `self.unet.device`, `.dtype`)
    @property
    def device(self):
        return self._device()

    @property
    def dtype(self):
        return self.dt

    def _apply(self, fn, recurse=True):
        """`.to()` / `.cuda()` / `.cpu()` / `.float()` / `.half()` are no-ops on a native graph.  The reference's trainer and its device-state
        presets shuttle models between cpu and cuda and re-cast them (`unet.to(self.device_torch, dtype=dtype)`, jobs/process/
        BaseSDTrainProcess.py:1899; BaseModel.set_device_state; `self.sd.unet.to('cpu')`, SDTrainer.py:298) to make room on 24-80 GB cards; here the
        frozen base, its transposed / quantised copies and every kernel-layout buffer stay where they were built (288 GB of HBM: nothing needs
        to leave) and in the dtype the kernels are written for.  Build the model on the device and in the dtype you want."""
        return self

    def enable_gradient_checkpointing(self):
        """accepted and ignored (BaseSDTrainProcess.py:1864-1866): the explicit backward keeps every activation it needs resident, there is
        nothing to recompute (the `recompute_gelu` switch of the FLUX graph is the one memory lever)"""

    def disable_gradient_checkpointing(self):
        pass

    def set_ops(self, ops):
        self.ops = ops

    def attach_network(self, network):
        # not a registered sub-module: the adapter must stay out of the base model's state_dict / parameters()
        if self.fp8_mfma:
            self._refuse_w8a8_adapters(network)
        object.__setattr__(self, "network", network)

    @staticmethod
    def _refuse_w8a8_adapters(network):
        """W8A8 base GEMMs (quantize_base_fp8(mfma=True)) are written for plain LoRA adapters: a DoRA layer's column scale has no place in
        the scaled fp8 epilogue and a LoKr layer would silently run on the weight-only bf16 path, i.e. the model would mix two
        arithmetics.  Refused up front instead of failing inside the first step."""
        if network is None:
            return
        for m in network.get_all_modules():
            if getattr(m, "is_lokr", False) or getattr(m, "magnitude", None) is not None:
                raise NotImplementedError(f"W8A8 fp8 base (mfma=True) with a {'LoKr' if getattr(m, 'is_lokr', False) else 'DoRA'} adapter "
                                          f"({m.lora_name}): use the weight-only fp8 base (mfma=False) or a bf16 base")

    def _resolve_network(self):
        """The adapter network of this forward: the FusedLoRANetwork attached with attach_network, or — when the reference's trainer built its
        own LoRASpecialNetwork over this model and called apply_to (jobs/process/BaseSDTrainProcess.py:1949-1993) — the AdoptedNetwork around
        it (adopt.py: arena re-pointing on first use, pointer check + shadow refresh on every use)."""
        from .adopt import resolve_network

        return resolve_network(self)

    def _token_linears(self):
        raise NotImplementedError

    def _dgrad_linears(self):
        """Linears whose data gradient is needed (default: all token linears)."""
        return self._token_linears()

    def _dgrad_groups(self):
        """Tuples of Linears that read the SAME activation (q, k, v[, proj_mlp]): their transposed copies are laid out as column
        windows of one [in, sum(out)] matrix so that the group's data gradient is ONE GEMM contracting over the concatenated outputs
        (_group_bwd).  Default: none."""
        return []

    concat_dgrad = os.environ.get("AITK_CONCAT_DGRAD", "1") != "0"
    # one pass over dY for dT and lora_up.weight.grad (aitk_lora_bwd_fused) instead of aitk_lora_down + aitk_lora_wgrad; 0 = the two launches
    lora_bwd_fused = os.environ.get("AITK_LORA_BWD_FUSED", "1") != "0"

    def prepare(self):
        """Build the transposed weight copies used by the data-gradient GEMMs (frozen => one-time)."""
        grouped = set()
        for lins in (self._dgrad_groups() if self.concat_dgrad else []):
            if any(l.qweight is not None for l in lins) or len({l.in_features for l in lins}) != 1:
                continue
            cat = torch.empty(lins[0].in_features, sum(l.out_features for l in lins), dtype=lins[0].weight.dtype, device=lins[0].weight.device)
            c0 = 0
            for l in lins:
                l.weight_t = cat[:, c0:c0 + l.out_features]  # column window (row stride = sum(out)): every single-layer use reads it through its stride
                l.weight_t.copy_(l.weight.data.t())
                l._dgroup = (cat, c0, tuple(id(x) for x in lins))
                c0 += l.out_features
                grouped.add(id(l))
        for lin in self._dgrad_linears():
            if lin.qweight is None and id(lin) not in grouped:
                lin.weight_t = lin.weight.data.t().contiguous()
                lin._dgroup = None
        self._prepared = True
        return self

    @torch.no_grad()
    def quantize_base_fp8(self, release_bf16=False, mfma=False):
        """Weight-only fp8 (OCP e4m3, per-output-channel scale) for every token-GEMM Linear of the blocks — BASELINE config 5;
        the reference does this with optimum-quanto qfloat8 / torchao Float8WeightOnly (toolkit/util/quantize.py:43-75,
        toolkit/stable_diffusion_model.py:794-801).  Activations and the LoRA adapter stay bf16 / fp32.  Each layer's weight is
        expanded to bf16(fp8 * scale) into a shared scratch right before its GEMM (_dequant): forward from `qweight` [out,in], dgrad
        from `qweight_t` [in,out]; aitk_gemm_nt can also consume the fp8 bytes directly (b_scale, slower).  adaLN / embedder
        projections (B rows, weight streaming) keep bf16 weights.

        mfma=True (opt-in; BASELINE config 5's "CDNA4 fp8 MFMA base"): the token GEMMs run W8A8 on the MX-scaled fp8 matrix instruction
        (aitk_gemm_nt b_scale_mode 3, twice the bf16 MFMA rate): the activation operand of every base GEMM — x in forward, dY (times the
        weight's per-channel scale, which runs along the contraction there) in the data gradient — is quantised per token to e4m3 by
        aitk_quant_rows_fp8 right before its GEMM; the rank-r adapter slab, its operands and every adapter gradient stay bf16 / fp32.
        This is NOT the reference's arithmetic (its quantisers are weight-only): DESIGN.md states the deviation and the measured parity.
        Outlier statistics (a few residual channels at 10^2 - 10^3 x the rest, as real DiT weights produce) do not break the per-token e4m3
        operand: adapter-gradient error 1.0e-2 at full depth with planted outliers vs 1.75e-2 without (profiles/r04_outlier_parity.log,
        tests/test_gpu_outlier_parity.py) — the row scale follows the outlier and e4m3's exponent range covers the spread below it."""
        if mfma:
            self._refuse_w8a8_adapters(self.network)
        self.fp8_mfma = bool(mfma)
        for lin in self._token_linears():
            quantize_linear_fp8(lin, lin.weight.data)
            lin.weight_t = None
            lin._dgroup = None
            if release_bf16:
                lin.weight.data = torch.empty(0, dtype=lin.weight.dtype, device=lin.weight.device)
        if not mfma and self.concat_dgrad:
            # weight-only base: a same-input group's data gradient still runs as ONE K-concatenated GEMM — the members' e4m3 W^T are expanded
            # side by side into one [in, sum(out)] bf16 scratch right before it (_concat_dgrad_operand) instead of one scratch + one GEMM each
            for lins in self._dgrad_groups():
                if all(l.qweight is not None for l in lins) and len({l.in_features for l in lins}) == 1:
                    c0 = 0
                    for l in lins:
                        l._dgroup = ("fp8", c0, tuple(id(x) for x in lins))
                        c0 += l.out_features
        self._prepared = True
        self.is_quantized = True
        return self

    def _dequant(self, q, scale, mode):
        """Weight-only fp8 base: expand ONE layer's e4m3 weight into the shared bf16 scratch right before its GEMM (in-order on
        the stream, so the scratch is reused by every layer).  +3 B/element of traffic (~10-40 us per layer) buys the full-speed
        bf16 8-phase GEMM instead of a GEMM that dequantises in its load segment."""
        n = q.shape[0] * q.shape[1]
        slot = getattr(self, "_dq_slot", 0)  # 0: sequential launches; 1 / 2: the two merged streams of _paired
        pool = self.__dict__.setdefault("_dq_scratch", {})
        buf = pool.get(slot)
        if buf is None or buf.numel() < n:
            buf = pool[slot] = torch.empty(n, dtype=self.dt, device=q.device)
        out = buf[:n].view(q.shape[0], q.shape[1])
        self.ops.dequant_fp8(q, scale, mode, out)
        return out

    fp8_mfma = False

    def _q8(self, x, M, x_seg=None, col_mul=None, tag=None):
        """(e4m3 bytes [M, K], fp32 row scales [M]) of a GEMM activation operand.  The most recent result is kept: the q / k / v
        (/ proj_mlp) projections read the same normalised activation back to back, and the two column ranges of a single block's
        proj_out data gradient the same dY."""
        key = (id(x), x.data_ptr(), M, x_seg, None if col_mul is None else col_mul.data_ptr(), tag)
        last = self.__dict__.get("_q8_last")
        if last is not None and last[0] == key:
            return last[2], last[3]
        q = torch.empty(M, x.shape[1], dtype=torch.uint8, device=x.device)
        rs = torch.empty(M, dtype=torch.float32, device=x.device)
        self.ops.quant_rows_fp8(x, q, rs, col_mul=col_mul, x_seg=x_seg, M=M)
        self._q8_last = (key, x, q, rs)  # holds x: its id cannot be re-used while the entry lives
        return q, rs

    def _q8_reset(self):
        """Forget the kept quantisation: called at every block boundary and at the end of forward / backward, so the entry never outlives the
        group of launches it was made for (kernels rewrite buffers in place, which the identity key cannot see) and does not pin an
        activation + its codes across steps."""
        self.__dict__.pop("_q8_last", None)

    def dequantized_weight(self, lin):
        return (lin.qweight.view(torch.float8_e4m3fn).float() * lin.wscale[:, None]).to(self.dt)

    # ------------------------------------------------------------------ paired launches of two independent streams
    pair_streams = os.environ.get("AITK_PAIR_STREAMS", "1") != "0"

    def _paired(self, bodies):
        """Run two independent op sequences (callables; e.g. the image and the text stream of a double block).  On the MI355X table
        their kernel launches are recorded and merged so that GEMMs of equal shape go out as one grouped persistent launch
        (ops.replay_paired); each sequence keeps its own order.  The fp8 base expands each layer's weight into a bf16 scratch right
        before its GEMM: under the merged order the two sequences must not share that scratch, so each gets its own (_dq_slot)."""
        if not self.pair_streams or len(bodies) != 2 or not hasattr(self.ops, "recording"):
            return [b() for b in bodies]
        outs, recs = [], []
        try:
            for i, b in enumerate(bodies):
                self._dq_slot = i + 1
                with self.ops.recording() as launches:
                    outs.append(b())
                recs.append(launches)
        finally:
            self._dq_slot = 0
        self.ops.replay_paired(*recs)
        return outs

    # ------------------------------------------------------------------ helpers
    def _new(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or self.dt, device=self._device())

    def _defer_kw(self):
        """{'defer': True} on a kernel table that can collect weight-gradient finishes (the MI355X table), {} on the others."""
        return {"defer": True} if hasattr(self.ops, "wgrad_defer_begin") else {}

    def _lora_active(self, lin):
        net = self.network
        return (lin.lora is not None and net is not None and net.is_active and not net.is_merged_in
                and net._multiplier != 0)

    def _mult(self, rows_per_batch, B):
        """per-sample multiplier vector (network.torch_multiplier, shape [1] or [B]) -> (tensor|None, rows_per_batch)."""
        tm = self.network.torch_multiplier
        if tm.numel() == 1:
            mv = self.network._multiplier
            if float(mv[0] if isinstance(mv, (list, tuple)) else mv) == 1.0:
                return None, 0
            tm = tm.expand(B).contiguous()
        elif tm.numel() != B:
            tm = tm.repeat_interleave(B // tm.numel()).contiguous()
        return tm, rows_per_batch

    def _dora_ps(self, lin, rows_per_batch, B):
        """(m_mean, per-row deviation vector fp32 [B], rows_per_batch) when `lin` carries a DoRA adapter and the network multiplier differs
        between samples; None otherwise.  Keeps the column scale c in step with the current mean."""
        net = self.network
        if lin.lora is None or lin.lora.magnitude is None:
            return None
        mbar = net.multiplier_mean()
        if getattr(net, "_dora_mbar", None) != mbar:  # the multiplier changed since the column scales were computed (uniform values too)
            net.refresh_dora(self.ops)
        if not net.multiplier_is_per_sample():
            return None
        tm = net.torch_multiplier
        if tm.numel() != B:
            tm = tm.repeat_interleave(B // tm.numel())
        return mbar, (tm - mbar).contiguous(), rows_per_batch

    def _group_down(self, lins, x, *, M, rows_per_batch, B):
        """One skinny launch for every adapter of a same-input group: returns {id(lin): T view [M, r]} (or {} when the
        group is not laid out adjacently / inactive)."""
        if not all(self._lora_active(l) for l in lins):
            return {}
        if self.network.training and self.network.has_dropout:
            return {}  # dropout decisions are per adapter: each layer draws its own mask in _lin_fwd
        if any(self._dora_ps(l, rows_per_batch, B) is not None for l in lins):
            return {}  # DoRA under per-sample multipliers: two rank-space activations per layer (_DoraPS), built in _lin_fwd
        grp = getattr(lins[0].lora, "group", None)
        if grp is None or [id(m) for m in grp["mods"]] != [id(l.lora) for l in lins]:
            return {}
        Tcat = self._new(M, 3 * grp["R"])  # per adapter [T_hi | T_lo | T_hi] (split precision, see aitk_lora_down)
        mult, rpb = self._mult(rows_per_batch, B)
        self.ops.lora_down(x, grp["sh_down"], Tcat, scale=grp["scale"], mult=mult, rows_per_batch=rpb, M=M,
                           p_lo=grp["sh_down_lo"], split=grp["rp"])
        out = {}
        for l in lins:
            c0 = 3 * grp["col"][id(l.lora)]
            out[id(l)] = Tcat[:, c0:c0 + 3 * l.lora.rank_pad]
        return out

    # ------------------------------------------------------------------ T = x A^T of a GELU-fed adapter from inside the producing GEMM (AITK_EPI_EMIT_T)
    # The inputs of ff.net.2 / ff_context.net.2 / the single blocks' proj_out are GELU outputs: 792 MB per launch at B = 7 that aitk_lora_down reads back
    # right after the GEMM wrote them (14 ms of the 1.2-s step).  With emit_t the BIAS | GELU launch leaves the per-column-tile partial products instead
    # (+6 % bytes written next to u and gelu(u)) and aitk_lora_t_finish sums them: step 1198.5 -> 1190.1 ms same box (profiles/r06_ab_emit_t.txt).  Default;
    # AITK_EMIT_T=0 / model.emit_t = False turns it off.  Plain LoRA consumers of rank <= 32 without dropout behind plain / LoRA producers with a bias, N % 256 == 0
    # (any row count on the HIP kernel; the oracle table keeps whole 256-row tiles so that the committed CPU fixtures stay put); everything else keeps aitk_lora_down.
    emit_t = os.environ.get("AITK_EMIT_T", "1") != "0"

    def _emit_t_plan(self, producer, consumer, *, M, N, col0=0, extra_tiles=0):
        """None, or the state of one emission: consumer's lora_down product over its input columns [col0, col0 + N) is left by `producer`'s GELU launch as
        N / 256 tiles of a [tiles + extra_tiles, M, rank_pad] fp32 slab (extra tiles: parts of the consumer's input that come from elsewhere, aitk_lora_down_raw)."""
        ops, net = self.ops, self.network
        if not self.emit_t or not hasattr(ops, "lora_t_finish") or not self._lora_active(consumer):
            return None
        lo = consumer.lora
        rt = getattr(ops, "EMIT_T_ROW_TILE", 256)
        if lo.is_lokr or lo.magnitude is not None or lo.rank_pad not in (16, 32) or (net.training and net.has_dropout) or M % rt or N % 256:
            return None
        pl = producer.lora if self._lora_active(producer) else None
        if producer.bias is None or (pl is not None and (pl.is_lokr or pl.magnitude is not None)) or (producer.qweight is not None and self.fp8_mfma):
            return None
        ntiles = N // 256
        nbytes = (ntiles + extra_tiles) * M * lo.rank_pad * 4
        ws = ops.workspace(nbytes, self._device(), f"emit_t{getattr(self, '_dq_slot', 0)}")  # the two merged streams of _paired must not share it
        partial = ws[:nbytes // 4].view(ntiles + extra_tiles, M, lo.rank_pad)
        return {"partial": partial, "ntiles": ntiles, "args": (lo.sh_down[:, col0:col0 + N], lo.sh_down_lo[:, col0:col0 + N], partial, 0)}

    def _emit_t_finish(self, plan, consumer, *, M, rows_per_batch, B, ntiles):
        lo = consumer.lora
        T = self._new(M, 3 * lo.rank_pad)
        mult, rpb = self._mult(rows_per_batch, B)
        self.ops.lora_t_finish(plan["partial"], ntiles, T, scale=lo.scale, mult=mult, rows_per_batch=rpb, split=lo.rank_pad, M=M)
        return T

    def _lin_fwd(self, lin, x, out, *, M, rows_per_batch, B, flags=0, aux_out=None, aux_in=None, gate=None, gate_rows=0,
                 a_seg=None, c_seg=None, T=None, emit_t=None):
        """out = epi(x W^T + b + T B^T); returns T (saved for the weight gradient) or None.  A precomputed T (group
        launch) may be passed in.  emit_t: _emit_t_plan(...)["args"] — the launch also emits the consumer layer's lora_down partials."""
        ops = self.ops
        kw = {}
        if emit_t is not None:
            kw["emit_t"] = emit_t
        if self._lora_active(lin) and lin.lora.is_lokr:
            # LoKr: the Kronecker delta is written into the destination first, the base GEMM then accumulates onto it BEFORE its
            # activation / gate epilogue (ACCUM precedes GELU / GATE_RES in the epilogue order)
            assert not (flags & EPI_ACCUM)
            self._kron(lin.lora, x, out, M=M, x_seg=a_seg, out_seg=c_seg)
            w = lin.weight if lin.qweight is None else self._dequant(lin.qweight, lin.wscale, 1)
            ops.gemm_nt(x, w, out, bias=lin.bias, flags=flags | EPI_ACCUM, aux_out=aux_out, aux_in=aux_in, gate=gate,
                        gate_rows=gate_rows, a_seg=a_seg, c_seg=c_seg, M=M)
            return _KRON
        ps = self._dora_ps(lin, rows_per_batch, B) if self._lora_active(lin) else None
        if ps is not None:
            lo = lin.lora
            assert T is None and c_seg is None and not (lin.qweight is not None and self.fp8_mfma)
            mbar, dvec, rpb = ps
            rp = lo.rank_pad
            Tb, Td = self._new(M, 3 * rp), self._new(M, 3 * rp)
            ops.lora_down(x, lo.sh_down, Tb, scale=lo.scale * mbar, x_seg=a_seg, M=M, p_lo=lo.sh_down_lo, split=rp)
            ops.lora_down(x, lo.sh_down, Td, scale=lo.scale, mult=dvec, rows_per_batch=rpb, x_seg=a_seg, M=M, p_lo=lo.sh_down_lo, split=rp)
            ylin = self._new(M, lin.out_features)  # c * (x W^T + T_mean B^T) + b: kept for d magnitude
            wps = lin.weight if lin.qweight is None else self._dequant(lin.qweight, lin.wscale, 1)
            ops.gemm_nt(x, wps, ylin, bias=lin.bias, a_seg=a_seg, M=M, a2=Tb, b2=lo.sh_up3, col_scale=lo.c)
            lo.y_lin = ylin
            ops.ew(1, ylin, out[:M])
            # out = epi(ylin + T_dev B^T): the accumulate epilogue precedes GELU / gate-residual in the epilogue order
            ops.gemm_nt(Td, lo.sh_up3, out, flags=flags | EPI_ACCUM, aux_out=aux_out, aux_in=aux_in, gate=gate, gate_rows=gate_rows, M=M)
            return _DoraPS(Tb, Td, mbar, dvec, rpb)
        plan = self.network.dropout_plan(lin.lora, M=M, rows_per_batch=rows_per_batch, B=B) if (T is None and self._lora_active(lin) and not lin.lora.is_lokr) else None
        if self._lora_active(lin) and plan != "skip":
            lo = lin.lora
            if T is None:
                T = self._new(M, 3 * lo.rank_pad)
                mult, rpb = self._mult(rows_per_batch, B)
                tm, tm_rpb = plan if plan is not None else (None, 0)
                ops.lora_down(x, lo.sh_down, T, scale=lo.scale, mult=mult, rows_per_batch=rpb, x_seg=a_seg, M=M,
                              p_lo=lo.sh_down_lo, split=lo.rank_pad, tmask=tm, tmask_rows_per_batch=tm_rpb)
                if plan is not None:
                    T._tmask = plan  # the gradient of the masked activation takes the same mask (_lora_grads)
            kw.update(a2=T, b2=lo.sh_up3)  # [T_hi | T_lo | T_hi] . [B_hi | B_hi | B_lo]^T: the fp32 adapter product to 2^-17
            if lo.magnitude is not None:  # DoRA: y = c * (x W^T + T B^T) + b; the linear output is kept for d magnitude
                kw["col_scale"] = lo.c
                if (flags & EPI_GATE_RES) and aux_out is None:
                    aux_out = self._new(M, lin.out_features)
                assert c_seg is None
                lo.y_lin = aux_out if (flags & (EPI_GELU | EPI_GATE_RES)) else out
        else:
            T = None
        if lin.qweight is not None and self.fp8_mfma:  # W8A8 on the MX-scaled fp8 MFMA: x quantised per token, e4m3 weight codes as they are
            assert "col_scale" not in kw
            xq, xs = self._q8(x, M, x_seg=a_seg)
            ops.gemm_nt(xq, lin.qweight, out, bias=lin.bias, flags=flags, aux_out=aux_out, aux_in=aux_in, gate=gate, gate_rows=gate_rows,
                        c_seg=c_seg, M=M, a_scale=xs, b_scale=lin.wscale, b_scale_mode=3, **kw)
            return T
        w = lin.weight if lin.qweight is None else self._dequant(lin.qweight, lin.wscale, 1)
        if "col_scale" in kw and (flags & EPI_ADD_AUX):
            # DoRA needs the bare linear output for d magnitude and the residual-add epilogue does not store it: product first,
            # residual in a second pass (Wan cross-attention out-projection only)
            y = self._new(M, lin.out_features)
            ops.gemm_nt(x, w, y, bias=lin.bias, flags=flags & ~EPI_ADD_AUX, a_seg=a_seg, M=M, **kw)
            ops.ew(2, y, out, a=aux_in)
            lin.lora.y_lin = y
            return T
        ops.gemm_nt(x, w, out, bias=lin.bias, flags=flags, aux_out=aux_out,
                    aux_in=aux_in, gate=gate, gate_rows=gate_rows, a_seg=a_seg, c_seg=c_seg, M=M, **kw)
        return T

    # ------------------------------------------------------------------ LoKr (toolkit/models/lokr.py:331-399)
    def _lokr_scale(self, lo):
        """runtime scale * mean(network multiplier): the reference averages per-sample multipliers for LoKr (lokr.py:394-395)."""
        mv = self.network._multiplier
        vals = [float(v) for v in mv] if isinstance(mv, (list, tuple)) else [float(mv)]
        return lo.scale * sum(vals) / len(vals)

    def _kron(self, lo, x, out, *, M, x_seg=None, out_seg=None):
        """out = scale * per-token lokr_w1 . X . lokr_w2^T  (the LoKr delta of one layer)."""
        self._kron_any(x, lo.sh_up, lo.sh_down, out, a_in=lo.in_m, b_in=lo.in_n, a_out=lo.out_l, b_out=lo.out_k, scale=self._lokr_scale(lo),
                       x_seg=x_seg, out_seg=out_seg, M=M, two_stage=getattr(lo, "kron_two_stage", False))

    def _rows_view(self, t, a, b, seg, M):
        """[rows, a * b] (rows of one token) -> ([rows * a, b] view, segment map in those rows) for a GEMM over (token x factor-index) rows; a column
        window of a wider buffer (row stride != a * b) cannot be viewed that way and is gathered into a contiguous copy first."""
        if t.stride(0) != a * b:
            c = self._new(M, a * b)
            self.ops.kron_apply(t, None, None, c, a_in=a, b_in=b, a_out=a, b_out=b, x_seg=seg, M=M)
            t, seg = c, None
        return t.view(-1, b), (None if seg is None else (seg[0] * a, seg[1]))

    def _kron_any(self, x, A, Bm, out, *, a_in, b_in, a_out, b_out, scale, M, x_seg=None, out_seg=None, accumulate=False, two_stage=False,
                  col0=0, ncols=0):
        """out (+)= scale * (A kron Bm) applied per token.  two_stage (lora.check_kron_fits: Bm too large for the per-token kernel's LDS, an explicit
        small network.lokr_factor on a wide layer): a plain GEMM with Bm over (token x factor-index) rows + the per-token mix with the small factor
        A on the NARROWER side of Bm, so that the mix's tiles fit:
            b_in <= b_out:  V_m = A X_m  [a_out, b_in]   then  out[(m, p), :] = V[(m, p), :] Bm^T
            b_in >  b_out:  T[(m, q), :] = X[(m, q), :] Bm^T   then  out_m = A T_m
        The intermediate is bf16, as the per-token kernel's LDS intermediate is."""
        ops = self.ops
        if not two_stage:
            ops.kron_apply(x, A, Bm, out, a_in=a_in, b_in=b_in, a_out=a_out, b_out=b_out, scale=scale, accumulate=accumulate, col0=col0, ncols=ncols,
                           x_seg=x_seg, out_seg=out_seg, M=M)
            return
        if ncols:  # a column window of the product (the single blocks' proj_out data gradient goes out in two windows): the whole product into a scratch, the window copied / added
            tmp = self._new(M, a_out * b_out)
            self._kron_any(x, A, Bm, tmp, a_in=a_in, b_in=b_in, a_out=a_out, b_out=b_out, scale=scale, M=M, x_seg=x_seg, two_stage=True)
            ops.kron_apply(tmp[:, col0:col0 + ncols], None, None, out, a_in=1, b_in=ncols, a_out=1, b_out=ncols, accumulate=accumulate, out_seg=out_seg, M=M)
            return
        if b_in <= b_out:
            V = self._new(M, a_out * b_in)
            ops.kron_apply(x, A, None, V, a_in=a_in, b_in=b_in, a_out=a_out, b_out=b_in, scale=scale, x_seg=x_seg, M=M)
            if out.stride(0) != a_out * b_out:  # destination is a column window: product into a contiguous scratch, then a (transposing-free) row copy
                tmp = self._new(M, a_out * b_out)
                ops.gemm_nt(V.view(M * a_out, b_in), Bm, tmp.view(M * a_out, b_out), M=M * a_out)
                ops.kron_apply(tmp, None, None, out, a_in=a_out, b_in=b_out, a_out=a_out, b_out=b_out, accumulate=accumulate, out_seg=out_seg, M=M)
                return
            o2, c_seg = out.view(-1, b_out), (None if out_seg is None else (out_seg[0] * a_out, out_seg[1]))
            ops.gemm_nt(V.view(M * a_out, b_in), Bm, o2, flags=EPI_ACCUM if accumulate else 0, c_seg=c_seg, M=M * a_out)
        else:
            x2, a_seg = self._rows_view(x, a_in, b_in, x_seg, M)
            T = self._new(M, a_in * b_out)
            ops.gemm_nt(x2, Bm, T.view(M * a_in, b_out), a_seg=a_seg, M=M * a_in)
            ops.kron_apply(T, A, None, out, a_in=a_in, b_in=b_out, a_out=a_out, b_out=b_out, scale=scale, accumulate=accumulate, out_seg=out_seg, M=M)

    def _lokr_grads(self, lo, dy, x_in, *, M, x_seg=None):
        """Factor gradients into the fp32 arena (autograd of the reference's two einsums):
          d lokr_w1[p,q] = sum_{m,o} dY[m,p,o] * (X_m w2^T)[q,o]        d lokr_w2[o,s] = sum_{m,q} (w1^T dY_m)[q,o] * X[m,q,s]
        Both are skinny weight-gradient contractions over (token x factor-index) rows once the per-token intermediates are laid
        out with the contracted index leading: aitk_kron_apply writes them (transposed where needed), aitk_lora_wgrad reduces."""
        ops, sc = self.ops, self._lokr_scale(lo)
        a_in, b_in, a_out, b_out = lo.in_m, lo.in_n, lo.out_l, lo.out_k
        if getattr(lo, "kron_two_stage", False):
            return self._lokr_grads_two_stage(lo, dy, x_in, M=M, x_seg=x_seg)
        tmpT = self._new(M, b_out * a_in)   # [m, o, q] = (X_m w2^T)^T
        ops.kron_apply(x_in, None, lo.sh_down, tmpT, a_in=a_in, b_in=b_in, a_out=a_in, b_out=b_out, transpose_out=True, x_seg=x_seg, M=M)
        dyT = self._new(M, b_out * a_out)   # [m, o, p] = scale * dY_m^T
        ops.kron_apply(dy, None, None, dyT, a_in=a_out, b_in=b_out, a_out=a_out, b_out=b_out, transpose_out=True, scale=sc, M=M)
        self._skinny_tn(dyT.view(M * b_out, a_out), tmpT.view(M * b_out, a_in), lo.g_up, rows=M * b_out)
        U = self._new(M, a_in * b_out)      # [m, q, o] = scale * (w1^T dY_m)
        ops.kron_apply(dy, lo.sh_upT, None, U, a_in=a_out, b_in=b_out, a_out=a_in, b_out=b_out, scale=sc, M=M)
        if x_in.stride(0) != a_in * b_in:   # column window of a wider buffer: gather the rows first
            xc = self._new(M, a_in * b_in)
            ops.kron_apply(x_in, None, None, xc, a_in=a_in, b_in=b_in, a_out=a_in, b_out=b_in, x_seg=x_seg, M=M)
            x_in, x_seg = xc, None
        rows = M if x_seg is None else x_seg[0]
        xg = x_in[:rows].view(rows * a_in, b_in)
        g_seg = None if x_seg is None else (x_seg[0] * a_in, x_seg[1])
        if lo.use_w2:
            self._skinny_tn(U.view(M * a_in, b_out), xg, lo.g_down, rows=M * a_in, g_seg=g_seg)
        else:  # low-rank W2 = a @ b: gradient of the composed factor into its scratch, then d a = dW2 b^T, d b = a^T dW2 into the arena
            self._skinny_tn(U.view(M * a_in, b_out), xg, lo.g_down, rows=M * a_in, g_seg=g_seg, accumulate=False)
            ops.lokr_lowrank_grad(lo.g_down, lo.lokr_w2_a.data, lo.lokr_w2_b.data, lo.g_w2a, lo.g_w2b, accumulate=True)

    def _lokr_grads_two_stage(self, lo, dy, x_in, *, M, x_seg=None):
        """_lokr_grads when W2 does not fit the per-token kernel (lora.check_kron_fits "two_stage"): the same two contractions, with every per-token
        product that involves W2 replaced by a GEMM over (token x factor-index) rows and every mix with lokr_w1 kept on the narrower side.
          b_out <= b_in ("out" side narrow):  T = X w2^T [M, a_in, b_out];  d w1[p,q] = sum_{m,o} dY[m,p,o] T[m,q,o]
                                              U = w1^T dY [M, a_in, b_out];  d w2[o,s] = sum_{m,q} U[m,q,o] X[m,q,s]
          b_in  <  b_out ("in" side narrow):  dV = dY w2 [M, a_out, b_in];  d w1[p,q] = sum_{m,s} dV[m,p,s] X[m,q,s]
                                              V = w1 X [M, a_out, b_in];    d w2[o,s] = sum_{m,p} dY[m,p,o] V[m,p,s]"""
        ops, sc = self.ops, self._lokr_scale(lo)
        a_in, b_in, a_out, b_out = lo.in_m, lo.in_n, lo.out_l, lo.out_k

        def w2_grad(s_rows, g_rows, rows, g_seg=None):
            if lo.use_w2:
                self._skinny_tn(s_rows, g_rows, lo.g_down, rows=rows, g_seg=g_seg)
            else:  # low-rank W2 = a @ b: gradient of the composed factor into its scratch, then the pair's gradients into the arena
                self._skinny_tn(s_rows, g_rows, lo.g_down, rows=rows, g_seg=g_seg, accumulate=False)
                ops.lokr_lowrank_grad(lo.g_down, lo.lokr_w2_a.data, lo.lokr_w2_b.data, lo.g_w2a, lo.g_w2b, accumulate=True)

        if b_out <= b_in:
            x2, a_seg = self._rows_view(x_in, a_in, b_in, x_seg, M)  # contiguous (token x q) rows of X (gathered when x_in is a column window)
            T = self._new(M, a_in * b_out)
            ops.gemm_nt(x2, lo.sh_down, T.view(M * a_in, b_out), a_seg=a_seg, M=M * a_in)
            TT = self._new(M, b_out * a_in)   # [m, o, q]
            ops.kron_apply(T, None, None, TT, a_in=a_in, b_in=b_out, a_out=a_in, b_out=b_out, transpose_out=True, M=M)
            dyT = self._new(M, b_out * a_out)  # [m, o, p] = scale * dY_m^T
            ops.kron_apply(dy, None, None, dyT, a_in=a_out, b_in=b_out, a_out=a_out, b_out=b_out, transpose_out=True, scale=sc, M=M)
            self._skinny_tn(dyT.view(M * b_out, a_out), TT.view(M * b_out, a_in), lo.g_up, rows=M * b_out)
            U = self._new(M, a_in * b_out)     # [m, q, o] = scale * (w1^T dY_m)
            ops.kron_apply(dy, lo.sh_upT, None, U, a_in=a_out, b_in=b_out, a_out=a_in, b_out=b_out, scale=sc, M=M)
            w2_grad(U.view(M * a_in, b_out), x2, M * a_in, g_seg=a_seg)
        else:
            dy2, d_seg = self._rows_view(dy, a_out, b_out, None, M)
            dV = self._new(M, a_out * b_in)    # [m, p, s] = dY_m w2
            ops.gemm_nt(dy2, lo.sh_downT, dV.view(M * a_out, b_in), a_seg=d_seg, M=M * a_out)
            dVT = self._new(M, b_in * a_out)   # [m, s, p] (scaled)
            ops.kron_apply(dV, None, None, dVT, a_in=a_out, b_in=b_in, a_out=a_out, b_out=b_in, transpose_out=True, scale=sc, M=M)
            XT = self._new(M, b_in * a_in)     # [m, s, q]
            ops.kron_apply(x_in, None, None, XT, a_in=a_in, b_in=b_in, a_out=a_in, b_out=b_in, transpose_out=True, x_seg=x_seg, M=M)
            self._skinny_tn(dVT.view(M * b_in, a_out), XT.view(M * b_in, a_in), lo.g_up, rows=M * b_in)
            V = self._new(M, a_out * b_in)     # [m, p, s] = scale * (w1 X_m)
            ops.kron_apply(x_in, lo.sh_up, None, V, a_in=a_in, b_in=b_in, a_out=a_out, b_out=b_in, scale=sc, x_seg=x_seg, M=M)
            w2_grad(dy2, V.view(M * a_out, b_in), M * a_out)

    def _skinny_tn(self, s, g, out, *, rows, g_seg=None, accumulate=True):
        """out[R, L] (fp32) += s[rows, R]^T @ g[rows, L] through aitk_lora_wgrad (rank blocks of <= 64 columns, R % 16 == 0);
        when R is not a multiple of 16 but L is (and L <= 64, unsegmented) the roles are swapped and the result written transposed; anything
        else goes through zero-padded copies (small factors: network.lokr_factor 4 / 8)."""
        R, L = s.shape[1], g.shape[1]
        if R % 16 == 0:
            for c0 in range(0, R, 64):
                c1 = min(R, c0 + 64)
                self.ops.lora_wgrad(s[:, c0:c1], g, out[c0:c1], accumulate=accumulate, g_seg=g_seg, M=rows)
        elif L % 16 == 0 and L <= 64 and g_seg is None and R % 8 == 0:
            self.ops.lora_wgrad(g, s, out, transpose_out=True, accumulate=accumulate, M=rows)
        else:
            # small / odd Kronecker factors (network.lokr_factor 4 or 8: lokr_w1 is 4 x 4 / 8 x 8; a b_out such as 24): zero-pad the operands to the
            # kernel's granules (R -> 16, L -> 8), reduce into a scratch, add the valid block.  The padded copy of `s` is rows x 16 bf16 (up to 4x
            # the layer's dY for factor 4): the slow path of an unusual configuration, not a tuned one — it runs and it is exact.
            if L % 8 and g_seg is not None:
                raise NotImplementedError(f"LoKr factor gradient {R}x{L} over a segmented row map: the operand width must be a multiple of 8")
            R16, L8 = -(-R // 16) * 16, -(-L // 8) * 8
            ops = self.ops
            sp = torch.empty(s.shape[0], R16, dtype=s.dtype, device=s.device) if (R16 != R or not s.is_contiguous()) else s
            gp = torch.empty(g.shape[0], L8, dtype=g.dtype, device=g.device) if L8 != L else g
            scratch = torch.empty(R16, L8, dtype=torch.float32, device=out.device)

            def fill():  # torch-side, but in launch order (ops.host_call): under the paired image / text launch lists the producers of s / g are deferred too
                if sp is not s:
                    sp.zero_()
                    sp[:, :R].copy_(s)
                if gp is not g:
                    gp.zero_()
                    gp[:, :L].copy_(g)

            ops.host_call(fill)
            for c0 in range(0, R16, 64):
                c1 = min(R16, c0 + 64)
                ops.lora_wgrad(sp[:, c0:c1], gp, scratch[c0:c1], accumulate=False, g_seg=g_seg, M=rows)
            ops.host_call((lambda: out.add_(scratch[:R, :L])) if accumulate else (lambda: out.copy_(scratch[:R, :L])))

    def _lora_grads(self, lin, dy, T, x_in, *, M, rows_per_batch, B, x_seg=None, dT_out=None):
        """Adapter weight gradients into the fp32 arena; returns dT = c * (dy B) ([M, 3r] split slab layout) or None.
        With dT_out (a column slice of a group's dT buffer) the lora_down gradient is left to _group_wgrad."""
        if T is None:
            return None
        if T is _KRON:
            self._lokr_grads(lin.lora, dy, x_in, M=M, x_seg=x_seg)
            return _KRON
        ops = self.ops
        lo = lin.lora
        assert lo.magnitude is None or getattr(dy, "_dora_dz", False), "DoRA: pass dy through _dora_dz() first"
        rp = lo.rank_pad
        if isinstance(T, _DoraPS):
            # mean term: gradient dz = c*dy through (W, A, B) at the mean multiplier; deviation term: the raw dy at (m_b - m_mean)
            dT = dT_out if dT_out is not None else self._new(M, 3 * rp)
            ops.lora_down(dy, lo.sh_upT, dT, scale=lo.scale * T.mbar, M=M, p_lo=lo.sh_upT_lo, split=rp)
            ops.lora_wgrad(T.bar, dy, lo.g_up, transpose_out=True, accumulate=True, M=M, split=rp)
            if dT_out is None:
                ops.lora_wgrad(dT, x_in, lo.g_down, accumulate=True, g_seg=x_seg, M=M, split=rp)
            raw = dy._raw
            dTd = self._new(M, 3 * rp)
            ops.lora_down(raw, lo.sh_upT, dTd, scale=lo.scale, mult=T.dvec, rows_per_batch=T.rpb, M=M, p_lo=lo.sh_upT_lo, split=rp)
            ops.lora_wgrad(T.delta, raw, lo.g_up, transpose_out=True, accumulate=True, M=M, split=rp)
            ops.lora_wgrad(dTd, x_in, lo.g_down, accumulate=True, g_seg=x_seg, M=M, split=rp)
            dT._extra = dTd  # _lin_dgrad adds dT_dev A to the data gradient
            return dT
        dT = dT_out if dT_out is not None else self._new(M, 3 * rp)
        mult, rpb = self._mult(rows_per_batch, B)
        tm, tm_rpb = getattr(T, "_tmask", (None, 0))
        if self.lora_bwd_fused and rp in (16, 32) and M >= 2048 and hasattr(ops, "lora_bwd_fused") and dy.dim() == 2 and dy.stride(1) == 1:
            # dT = c (dy B) and lora_up.weight.grad = dy^T T from ONE read of dy (aitk_lora_bwd_fused) instead of one read each
            ops.lora_bwd_fused(dy, T, lo.sh_upT, lo.sh_upT_lo, dT, lo.g_up, scale=lo.scale, mult=mult, rows_per_batch=rpb, M=M, split=rp,
                               tmask=tm, tmask_rows_per_batch=tm_rpb)
        else:
            ops.lora_down(dy, lo.sh_upT, dT, scale=lo.scale, mult=mult, rows_per_batch=rpb, M=M, p_lo=lo.sh_upT_lo, split=rp, tmask=tm,
                          tmask_rows_per_batch=tm_rpb)
            ops.lora_wgrad(T, dy, lo.g_up, transpose_out=True, accumulate=True, M=M, split=rp, **self._defer_kw())  # only the finish waits: dy is consumed now
        if dT_out is None:
            # nothing reads lora_down.weight.grad before the optimizer: its finish pass may be collected (ops.wgrad_defer_begin)
            if isinstance(x_in, _ActInput):  # the input is [g | gelu(pre-activation)] and only the pre-activation was kept
                assert x_seg is None
                ops.lora_wgrad(dT, x_in.g, lo.g_down, accumulate=True, M=M, split=rp, g2=x_in.g2, g2_act=x_in.act, **self._defer_kw())
            else:
                ops.lora_wgrad(dT, x_in, lo.g_down, accumulate=True, g_seg=x_seg, M=M, split=rp, **self._defer_kw())
        return dT

    def _dora_dz(self, lin, dy, M):
        """DoRA layers: the gradient entering the (W, A, B) products is dz = c * dy, and d magnitude is accumulated from dy
        and this step's linear output.  Plain LoRA layers: dy unchanged."""
        if not self._lora_active(lin) or lin.lora.magnitude is None:
            return dy
        lo = lin.lora
        dz = self._new(M, lin.out_features)
        self.ops.dora_bwd(dy, lo.y_lin, lo.c, lin.bias, lo.magnitude.data, dz, lo.g_mag, M=M)
        lo.y_lin = None
        dz._dora_dz = True
        dz._raw = dy  # per-sample multipliers: the deviation term (m_b - m_mean) B A x sits outside the column scale (_DoraPS)
        return dz

    def _group_bwd(self, lins, dys, Ts, x_in, dx, *, M, rows_per_batch, B, first_flags=0):
        """Backward of several adapters+linears that read the same x_in: dx (+)= sum_j dy_j W_j + dT_j A_j, up-grads per
        layer, ONE lora_down-gradient launch for the whole group when it is laid out adjacently."""
        grp = getattr(lins[0].lora, "group", None) if all(t is not None for t in Ts) else None
        if grp is not None and [id(m) for m in grp["mods"]] != [id(l.lora) for l in lins]:
            grp = None
        dTcat = self._new(M, 3 * grp["R"]) if grp is not None else None
        cat = self._concat_dgrad_operand(lins, dys, Ts, grp, M)
        # census of the K-concatenated path: a group whose transposed weights prepare() laid out together but whose output
        # gradients did not arrive as adjacent windows of one buffer falls back to per-layer GEMMs — a different summation order, silently.
        # Counted so that tests can assert the production graphs never take that branch (dgrad_census()).
        if getattr(lins[0], "_dgroup", None) is not None and lins[0]._dgroup[2] == tuple(id(l) for l in lins):
            c = self.__dict__.setdefault("_dgrad_census", {"concat": 0, "fallback": 0})
            c["concat" if cat is not None else "fallback"] += 1
        if cat is not None:
            # ONE data-gradient GEMM for the group: dx = [dy_0 | dy_1 | ...] [W_0^T | W_1^T | ...]^T + [dT_0 | dT_1 | ...] [A_0^T3 | A_1^T3 | ...]^T
            # (contraction over the concatenated output channels; fp32 accumulation across the whole group instead of bf16
            # read-modify-write of dx between per-layer GEMMs)
            for lin, dy, T in zip(lins, dys, Ts):
                if grp is not None:
                    c0 = 3 * grp["col"][id(lin.lora)]
                    self._lora_grads(lin, dy, T, x_in, M=M, rows_per_batch=rows_per_batch, B=B, dT_out=dTcat[:, c0:c0 + 3 * lin.lora.rank_pad])
            kw = dict(a2=dTcat, b2=grp["sh_downT3"]) if grp is not None else {}
            self.ops.gemm_nt(cat[0], cat[1], dx, flags=first_flags, M=M, **kw)
            if grp is not None:
                self.ops.lora_wgrad(dTcat, x_in, grp["g_down"], accumulate=True, M=M, split=grp["rp"], **self._defer_kw())
            return
        for j, (lin, dy, T) in enumerate(zip(lins, dys, Ts)):
            dy = self._dora_dz(lin, dy, M)
            dT_out = None
            if grp is not None:
                c0 = 3 * grp["col"][id(lin.lora)]
                dT_out = dTcat[:, c0:c0 + 3 * lin.lora.rank_pad]
            dT = self._lora_grads(lin, dy, T, x_in, M=M, rows_per_batch=rows_per_batch, B=B, dT_out=dT_out)
            self._lin_dgrad(lin, dy, dT, dx, M=M, flags=(first_flags if j == 0 else EPI_ACCUM))
        if grp is not None:
            self.ops.lora_wgrad(dTcat, x_in, grp["g_down"], accumulate=True, M=M, split=grp["rp"], **self._defer_kw())

    def dgrad_census(self, reset=False):
        """{'concat': n, 'fallback': n} of the same-input groups seen by _group_bwd since the last reset (groups laid out for the concatenated
        data-gradient GEMM only)."""
        c = dict(self.__dict__.get("_dgrad_census", {"concat": 0, "fallback": 0}))
        if reset:
            self.__dict__["_dgrad_census"] = {"concat": 0, "fallback": 0}
        return c

    def _concat_dgrad_operand(self, lins, dys, Ts, grp, M):
        """(dY_cat [M, sum(out)], W^T_cat [in, sum(out)]) when the group's data gradient can run as one K-concatenated GEMM: the transposed
        weights were laid out together by prepare(), the output gradients are adjacent column windows of one buffer in the same order,
        and every layer is either a plain LoRA layer of one laid-out adapter group or has no active adapter; None otherwise."""
        dg = getattr(lins[0], "_dgroup", None)
        if dg is None or dg[2] != tuple(id(l) for l in lins) or self.fp8_mfma:
            return None
        if any(l.qweight is not None for l in lins) and not (isinstance(dg[0], str) and all(l.qweight is not None for l in lins)):
            return None
        if grp is None and any(t is not None for t in Ts):
            return None
        if grp is not None and any((t is _KRON or isinstance(t, _DoraPS) or l.lora.magnitude is not None) for l, t in zip(lins, Ts)):
            return None
        d0 = dys[0]
        if d0.dim() != 2 or d0.stride(1) != 1:
            return None
        ld, esz, p = d0.stride(0), d0.element_size(), d0.data_ptr()
        for l, dy in zip(lins, dys):
            if dy.dim() != 2 or dy.stride(1) != 1 or dy.stride(0) != ld or dy.data_ptr() != p or dy.shape[1] != l.out_features or dy.shape[0] < M:
                return None
            p += l.out_features * esz
        tot = sum(l.out_features for l in lins)
        if ld < tot:
            return None
        wt_cat = dg[0]
        if isinstance(wt_cat, str):  # weight-only fp8 group: expand the members' W^T (scale along the contraction axis) into adjacent column windows
            slot = getattr(self, "_dq_slot", 0)
            pool = self.__dict__.setdefault("_dq_scratch", {})
            n = lins[0].in_features * tot
            buf = pool.get(("cat", slot))
            if buf is None or buf.numel() < n:
                buf = pool[("cat", slot)] = torch.empty(n, dtype=self.dt, device=d0.device)
            wt_cat = buf[:n].view(lins[0].in_features, tot)
            c0 = 0
            for l in lins:
                self.ops.dequant_fp8(l.qweight_t, l.wscale, 2, wt_cat[:, c0:c0 + l.out_features])
                c0 += l.out_features
        return torch.as_strided(d0, (d0.shape[0], tot), (ld, 1), d0.storage_offset()), wt_cat

    def _lin_dgrad(self, lin, dy, dT, dx, *, M, flags=0, aux_in=None, dx_seg=None, w_rows=None):
        """dx (+)= dy W + dT A; w_rows = (r0, r1) restricts to input columns [r0, r1) (rows of W^T / A^T)."""
        kw = {}
        if dT is _KRON:  # dX_m = w1^T . dY_m . w2, written (or added) into dx ahead of the base dgrad GEMM
            lo = lin.lora
            c0, nc = (0, 0) if w_rows is None else (w_rows[0], w_rows[1] - w_rows[0])
            self._kron_any(dy, lo.sh_upT, lo.sh_downT, dx, a_in=lo.out_l, b_in=lo.out_k, a_out=lo.in_m, b_out=lo.in_n, scale=self._lokr_scale(lo),
                           accumulate=bool(flags & EPI_ACCUM), col0=c0, ncols=nc, out_seg=dx_seg, M=M, two_stage=getattr(lo, "kron_two_stage", False))
            flags |= EPI_ACCUM
            dT = None
        if dT is not None:
            shT = lin.lora.sh_downT3  # [in, 3r] = [A^T_hi | A^T_hi | A^T_lo] against dT = [hi | lo | hi]
            shT = shT if w_rows is None else shT[w_rows[0]:w_rows[1]]
            extra = getattr(dT, "_extra", None)
            if extra is not None:  # DoRA under per-sample multipliers: dx (+)= dT_dev A first, the main product then accumulates onto it
                self.ops.gemm_nt(extra, shT, dx, flags=flags & EPI_ACCUM, c_seg=dx_seg, M=M)
                flags |= EPI_ACCUM
            kw = dict(a2=dT, b2=shT)
        if lin.qweight is not None and self.fp8_mfma:
            # dX = dY W contracts over the output channels: the weight's per-channel scale multiplies dY before the per-token quantisation
            qt = lin.qweight_t if w_rows is None else lin.qweight_t[w_rows[0]:w_rows[1]]
            dyq, dys = self._q8(dy, M, col_mul=lin.wscale, tag=id(lin))
            self.ops.gemm_nt(dyq, qt, dx, flags=flags, aux_in=aux_in, c_seg=dx_seg, M=M, a_scale=dys, b_scale=None, b_scale_mode=3, **kw)
            return
        if lin.qweight is not None:  # rows of W^T = input columns; scale runs along the contraction (out) axis
            qt = lin.qweight_t if w_rows is None else lin.qweight_t[w_rows[0]:w_rows[1]]
            wt = self._dequant(qt, lin.wscale, 2)
        else:
            wt = lin.weight_t if w_rows is None else lin.weight_t[w_rows[0]:w_rows[1]]
        self.ops.gemm_nt(dy, wt, dx, flags=flags, aux_in=aux_in, c_seg=dx_seg, M=M, **kw)

    def _lin_bwd(self, lin, dy, T, x_in, dx, *, M, rows_per_batch, B, flags=0, aux_in=None, x_seg=None, dx_seg=None):
        dy = self._dora_dz(lin, dy, M)
        dT = self._lora_grads(lin, dy, T, x_in, M=M, rows_per_batch=rows_per_batch, B=B, x_seg=x_seg)
        self._lin_dgrad(lin, dy, dT, dx, M=M, flags=flags, aux_in=aux_in, dx_seg=dx_seg)

    def _ada_fwd(self, ada_lin, silu_temb, B):
        """mod[B, k*dim] = linear(silu(temb)) (+LoRA): small-batch GEMV; returns (mod, T)."""
        ops = self.ops
        mod = self._new(B, ada_lin.out_features)
        T = None
        kw = {}
        if self._lora_active(ada_lin) and ada_lin.lora.is_lokr:
            self._kron(ada_lin.lora, silu_temb, mod, M=B)
            ops.gemv_nt(silu_temb, ada_lin.weight, mod, bias=ada_lin.bias, accumulate=True)
            return mod, _KRON
        ps = self._dora_ps(ada_lin, 1, B) if self._lora_active(ada_lin) else None
        if ps is not None:
            lo = ada_lin.lora
            mbar, dvec, rpb = ps
            rp = lo.rank_pad
            Tb, Td = self._new(B, 3 * rp), self._new(B, 3 * rp)
            ops.lora_down(silu_temb, lo.sh_down, Tb, scale=lo.scale * mbar, M=B, p_lo=lo.sh_down_lo, split=rp)
            ops.lora_down(silu_temb, lo.sh_down, Td, scale=lo.scale, mult=dvec, rows_per_batch=rpb, M=B, p_lo=lo.sh_down_lo, split=rp)
            ylin = self._new(B, ada_lin.out_features)
            ops.gemv_nt(silu_temb, ada_lin.weight, ylin, bias=ada_lin.bias, t=Tb, bl=lo.sh_up3, col_scale=lo.c)
            lo.y_lin = ylin
            ops.ew(1, ylin, mod)
            ops.gemm_nt(Td, lo.sh_up3, mod, flags=EPI_ACCUM, M=B)
            return mod, _DoraPS(Tb, Td, mbar, dvec, rpb)
        plan = self.network.dropout_plan(ada_lin.lora, M=B, rows_per_batch=1, B=B) if self._lora_active(ada_lin) else None
        if self._lora_active(ada_lin) and plan != "skip":
            lo = ada_lin.lora
            T = self._new(B, 3 * lo.rank_pad)
            mult, rpb = self._mult(1, B)
            tm, tm_rpb = plan if plan is not None else (None, 0)
            ops.lora_down(silu_temb, lo.sh_down, T, scale=lo.scale, mult=mult, rows_per_batch=rpb, M=B, p_lo=lo.sh_down_lo,
                          split=lo.rank_pad, tmask=tm, tmask_rows_per_batch=tm_rpb)
            if plan is not None:
                T._tmask = plan
            kw = dict(t=T, bl=lo.sh_up3)
            if lo.magnitude is not None:
                kw["col_scale"] = lo.c
                lo.y_lin = mod
        ops.gemv_nt(silu_temb, ada_lin.weight, mod, bias=ada_lin.bias, **kw)
        return mod, T

    def _ada_bwd(self, ada_lin, dmod, T, silu_temb, B):
        """Only the adapter gradients: temb has no trainable ancestor, so no data gradient is propagated."""
        if T is None:
            return
        if T is _KRON:
            self._lokr_grads(ada_lin.lora, dmod, silu_temb, M=B)
            return
        ops = self.ops
        lo = ada_lin.lora
        dmod = self._dora_dz(ada_lin, dmod, B)
        rp = lo.rank_pad
        if isinstance(T, _DoraPS):
            self._lora_grads(ada_lin, dmod, T, silu_temb, M=B, rows_per_batch=1, B=B)
            return
        dT = self._new(B, 3 * rp)
        mult, rpb = self._mult(1, B)
        tm, tm_rpb = getattr(T, "_tmask", (None, 0))
        ops.lora_down(dmod, lo.sh_upT, dT, scale=lo.scale, mult=mult, rows_per_batch=rpb, M=B, p_lo=lo.sh_upT_lo, split=rp, tmask=tm,
                      tmask_rows_per_batch=tm_rpb)
        ops.lora_wgrad(T, dmod, lo.g_up, transpose_out=True, accumulate=True, M=B, split=rp, **self._defer_kw())
        ops.lora_wgrad(dT, silu_temb, lo.g_down, accumulate=True, M=B, split=rp, **self._defer_kw())
