"""Adoption of a LoRA network the REFERENCE built itself — the drop-in boundary without a trainer patch.

The reference's trainer constructs its own `LoRASpecialNetwork` over `sd.get_model_to_train()` (jobs/process/BaseSDTrainProcess.py:1932-1978),
moves it to the device in fp32 (1983), hands it to the model plug-in (`self.sd.network = self.network`, 1985) and calls `apply_to` (1988), which
for every wrapped layer runs `LoRAModule.apply_to` (toolkit/lora_special.py:132-135):

    self.org_forward = self.org_module[0].forward
    self.org_module[0].forward = self.forward

It then builds the optimizer over `network.prepare_optimizer_params()` (toolkit/kohya_lora.py:1030-1074), the EMA over the same Parameters
(toolkit/ema.py) and activates the adapter with `with network:` (toolkit/network_mixins.py:849-853).  None of this can be replaced from a model
plug-in, and the native graphs never call `Linear.forward` — so instead of asking the trainer to build another class, the fused path ADOPTS the
objects the reference made:

  * `graph.Linear.__setattr__` (and the UNet's convolution holders) see the `forward` swap and call `register_foreign_adapter`: the module that owns
    the new forward becomes the layer's adapter (`lin.lora`), after a check that the fused graph can run it.  Anything it cannot run — LoRM,
    full-rank / FullModule layers, `use_bias`, decomposed LoKr factors, DoRA above rank 64 — raises HERE, at `apply_to`, never a base-only model.
  * On the first forward of the native model (`FusedGraphBase._resolve_network`), `AdoptedNetwork` lays the flat fp32 arenas out exactly as
    `FusedLoRANetwork.build_arena` does and RE-POINTS the storage of the reference's own Parameters at views of them (`param.data = view`,
    `param.grad = grad view`).  Parameter identity is untouched, so the optimizer, `clip_grad_norm_`, `toolkit/ema.py`, `accelerator.prepare`,
    `network.state_dict()` / `get_state_dict` / `save_weights` / `load_weights` / `merge_in` of the REFERENCE keep working on the same memory the
    HIP kernels read (through the bf16 split shadows, refreshed at every adapter-active forward: the trainer has no "weights changed" hook).
  * Activation, multipliers and dropout are read from the reference's network at launch time: `is_active` / `is_merged_in` / `_multiplier == 0`
    (network_mixins.py:274-296), `torch_multiplier` incl. per-sample vectors (313-321), `scale`, `module.dropout / rank_dropout / module_dropout`
    with the reference's draw order (197-239), `training`.
  * If something replaced a Parameter's storage behind our back (`network.to(...)`, a rank-changing `load_weights`: network_mixins.py:737-775),
    the next forward notices the pointer mismatch, copies the current values into the arena and re-points again (`AdoptedNetwork.sync`).
"""
import os as _os
import sys as _sys
import weakref

import torch
import torch.nn as nn

from .lora import FusedLoRANetwork, _ParamProxy, check_kron_fits

# class names of the reference's adapter modules (toolkit/lora_special.py:46, toolkit/models/DoRA.py:36, toolkit/models/lokr.py:79) and of the
# oracle's restatements of them (oracle/lora_ref.py: tests drive the adoption without importing the reference)
_LORA_CLASSES = ("LoRAModule", "RefLoRAModule")
_DORA_CLASSES = ("DoRAModule", "RefDoRAModule")
_LOKR_CLASSES = ("LokrModule", "RefLokrModule")


class AdoptionError(NotImplementedError):
    """The reference built an adapter the fused graph cannot execute; raised where the reference attaches it (`apply_to`)."""


def _kind(module):
    name = module.__class__.__name__
    if name in _LORA_CLASSES:
        return "lora"
    if name in _DORA_CLASSES:
        return "dora"
    if name in _LOKR_CLASSES:
        return "lokr"
    return None


def _network_of(module):
    ref = getattr(module, "network_ref", None)
    net = ref() if callable(ref) else None
    if net is None:
        net = getattr(module, "network", None)
        if isinstance(net, (list, tuple)):  # oracle modules keep the network in a list (out of the module tree)
            net = net[0]
    return net


def check_foreign_network(net):
    """Network-level options of a reference-built network (`sd.network = network`, BaseSDTrainProcess.py:1985) that have no fused
    counterpart.  A FusedLoRANetwork passes untouched."""
    if isinstance(net, FusedLoRANetwork):
        return
    if getattr(net, "is_lorm", False):
        raise AdoptionError("network.type 'lorm' is not on the fused path")
    if getattr(net, "full_rank", False) or str(getattr(net, "network_type", "lora")).lower() == "fullrank":
        raise AdoptionError("network.type 'fullrank' is not on the fused path")
    if getattr(net, "full_train_in_out", False):
        raise AdoptionError("full_train_in_out retrains base layers: not on the fused path (frozen base)")
    if len(getattr(net, "text_encoder_loras", None) or []):
        raise AdoptionError("text-encoder adapters (train_text_encoder) are not on the fused path: text embeddings are cached")
    cfg = getattr(net, "network_config", None)
    if cfg is not None and getattr(cfg, "all_layers", False):
        raise AdoptionError("network.all_layers (FullModule weights on norms / embeddings) is not on the fused path")
    if getattr(net, "full_if_contains", None):
        raise AdoptionError("network.full_if_contains (FullModule layers) is not on the fused path")


def register_foreign_adapter(layer, new_forward):
    """`layer.forward = new_forward` as issued by the reference's `apply_to`: validate the adapter module behind `new_forward` and make it the
    layer's adapter.  Raises AdoptionError / TypeError instead of leaving a layer that would silently run base-only."""
    owner = getattr(new_forward, "__self__", None)
    kind = _kind(owner) if owner is not None else None
    if kind is None:
        raise TypeError(f"{layer.__class__.__name__}.forward cannot be replaced by {new_forward!r}: the fused graph never calls it. Only the "
                        "reference's LoRAModule / DoRAModule / LokrModule (whose apply_to swaps forward) are adopted as adapters")
    m = owner
    name = getattr(m, "lora_name", "?")
    net = _network_of(m)
    if net is not None and getattr(net, "is_lorm", False):
        raise AdoptionError(f"{name}: LoRM networks are not on the fused path")
    if getattr(m, "full_rank", False):
        raise AdoptionError(f"{name}: network.type 'fullrank' is not on the fused path")
    is3 = bool(getattr(layer, "is_conv3x3", False))
    if kind in ("dora", "lokr") and (is3 or getattr(layer, "is_conv1x1", False)):
        raise AdoptionError(f"{name}: {kind} adapters on convolutions are not on the fused path (plain LoRA only)")
    if kind in ("lora", "dora"):
        up = getattr(m, "lora_up", None)
        if getattr(up, "bias", None) is not None:
            raise AdoptionError(f"{name}: use_bias adapters (LoRM) are not on the fused path")
        if getattr(m, "lora_mid", None) is not None or hasattr(m, "scalar"):
            raise AdoptionError(f"{name}: tucker / trainable-scalar (LoCon) adapters are not on the fused path")
        r = int(m.lora_dim)
        has_dropout = any(getattr(m, k, None) for k in ("dropout", "rank_dropout", "module_dropout"))
        if isinstance(getattr(m, "dropout", None), nn.Module):
            raise AdoptionError(f"{name}: nn.Dropout-module dropout is not on the fused path (float probabilities are)")
        if kind == "dora" and has_dropout:
            raise AdoptionError(f"{name}: DoRA with dropout: the fused path runs DoRA without dropout")
        if is3 and r > 64:
            raise AdoptionError(f"{name}: 3x3-conv adapters above rank 64 are not on the fused path")
        if is3 and (layer.cin_pad != layer.in_channels or layer.cout_pad != layer.out_channels):
            raise AdoptionError(f"{name}: channel-padded convolutions (conv_in / conv_out) cannot carry an adapter on the fused path")
    else:
        if not hasattr(m, "lokr_w1") or hasattr(m, "lokr_w1_a") or getattr(m, "cp", False):
            raise AdoptionError(f"{name}: LoKr with decompose_both / tucker factors is not on the fused path")
        if getattr(m, "rank_dropout", 0) or getattr(m, "module_dropout", 0) or getattr(m, "dropout", 0):
            raise AdoptionError(f"{name}: LoKr dropout variants are not on the fused path")
        in_n = getattr(m, "_in_n", getattr(m, "in_n", None))
        out_k = getattr(m, "_out_k", getattr(m, "out_k", None))
        if in_n is None or out_k is None or in_n % 8 or out_k % 8:
            raise AdoptionError(f"{name}: LoKr factor {out_k}x{in_n}: the kron kernel needs multiples of 8")
        try:
            check_kron_fits(name, int(layer.in_features) // int(in_n), int(in_n), int(layer.out_features) // int(out_k), int(out_k))
        except NotImplementedError as e:
            raise AdoptionError(str(e)) from None
    object.__setattr__(layer, "lora", m)
    object.__setattr__(layer, "_foreign_adapter", True)


def _graft(m, layer):
    """Kernel-facing bookkeeping attributes of `lora.LoRAModule` / `DoRAModule` / `LoKrModule` on a module the reference constructed (plain
    instance attributes: nothing registered, nothing that shows up in its state_dict)."""
    kind = _kind(m)
    put = lambda k, v: object.__setattr__(m, k, v)  # noqa: E731
    put("is_lokr", kind == "lokr")
    put("is_conv3x3", bool(getattr(layer, "is_conv3x3", False)))
    put("is_conv1x1", bool(getattr(layer, "is_conv1x1", False)))
    if m.is_conv3x3:
        put("conv_cin", layer.in_channels)
        put("conv_stride", layer.stride)
        put("in_features", layer.in_channels * 9)
        put("out_features", layer.out_channels)
    else:
        put("in_features", layer.in_features)
        put("out_features", layer.out_features)
    if kind != "dora" and "magnitude" not in m.__dict__ and "magnitude" not in m._parameters:
        put("magnitude", None)
    for k in ("dropout", "rank_dropout", "module_dropout"):
        if not hasattr(m, k):
            put(k, None)
    if kind == "lokr":
        put("in_m", getattr(m, "_in_m", getattr(m, "in_m", None)))
        put("in_n", getattr(m, "_in_n", getattr(m, "in_n", None)))
        put("out_l", getattr(m, "_out_l", getattr(m, "out_l", None)))
        put("out_k", getattr(m, "_out_k", getattr(m, "out_k", None)))
        put("kron_two_stage", check_kron_fits(m.lora_name, int(m.in_m), int(m.in_n), int(m.out_l), int(m.out_k)) == "two_stage")
        put("_in", layer.in_features)
        put("_out", layer.out_features)
        put("lora_up", _ParamProxy(m, "lokr_w1"))
        if m.use_w2:
            put("lora_down", _ParamProxy(m, "lokr_w2"))
        put("g_w2a", None)
        put("g_w2b", None)
        put("factor_params", lambda m=m: ([("lokr_w1", m.lokr_w1), ("lokr_w2", m.lokr_w2)] if m.use_w2 else
                                          [("lokr_w1", m.lokr_w1), ("lokr_w2_a", m.lokr_w2_a), ("lokr_w2_b", m.lokr_w2_b)]))
        put("composed_w2", lambda m=m: (m.lokr_w2.data if m.use_w2 else m.lokr_w2_a.data @ m.lokr_w2_b.data))
    if kind == "dora":
        put("c", None)
        put("w2", None)
        put("y_lin", None)
        put("off_mag", -1)
        put("g_mag", None)
    if not hasattr(m, "org_module"):
        put("org_module", [layer])
    for k in ("off_down", "off_up"):
        put(k, -1)
    for k in ("sh_down", "sh_down_lo", "sh_downT3", "sh_downT", "sh_up", "sh_up3", "sh_upT", "sh_upT_lo", "sh_down_stack", "sh_down_dgrad",
              "g_down", "g_up", "group"):
        put(k, None)


def _trainable(m):
    if m.is_lokr:
        return [p for _, p in m.factor_params()]
    out = [m.lora_down.weight, m.lora_up.weight]
    if m.magnitude is not None:
        out.append(m.magnitude)
    return out


class AdoptedNetwork(FusedLoRANetwork):
    """The kernel-facing half of `FusedLoRANetwork` (arenas, split shadows, same-input groups, dropout plans, DoRA column scales) built AROUND a
    network object the reference constructed; everything the trainer talks to stays the reference's own object."""

    _repoint = True
    fuse_trainer_step = True  # the trainer's optimizer.step() / ema.update() on the arena kernels when that is the same computation (below)

    def __init__(self, foreign, model, ops, device=None):
        nn.Module.__init__(self)
        self._fusions = []  # weak references to the _OptimizerFusion / _EmaFusion objects whose state lives in this network's arenas
        object.__setattr__(self, "_foreign", foreign)          # not a registered sub-module: its Parameters have one owner, the reference's network
        object.__setattr__(self, "_model_ref", weakref.ref(model))
        mods = list(foreign.get_all_modules() if hasattr(foreign, "get_all_modules") else foreign.unet_loras)
        if len(getattr(foreign, "text_encoder_loras", None) or []):
            raise AdoptionError("text-encoder adapters (train_text_encoder) are not on the fused path: text embeddings are cached")
        if not mods:
            raise AdoptionError("the network holds no adapter module")
        layers = {id(l.lora): l for l in model.modules() if getattr(l, "_foreign_adapter", False) and getattr(l, "lora", None) is not None}
        missing = [getattr(m, "lora_name", "?") for m in mods if id(m) not in layers]
        if missing:
            raise AdoptionError(f"{len(missing)} adapter module(s) of the network are not attached to a layer of the native model "
                                f"(apply_to not called, or they wrap modules outside it): {missing[:4]}")
        stray = [getattr(l.lora, "lora_name", "?") for l in layers.values() if id(l.lora) not in {id(m) for m in mods}]
        if stray:
            raise AdoptionError(f"layers of the native model carry adapters of ANOTHER network: {stray[:4]}")
        for m in mods:
            _graft(m, layers[id(m)])
        kinds = {_kind(m) for m in mods}
        if len(kinds) != 1:
            raise AdoptionError(f"mixed adapter types in one network are not on the fused path: {sorted(kinds)}")
        self.network_type = kinds.pop()
        if self.network_type != "lora" and any(m.is_conv3x3 or m.is_conv1x1 for m in mods):
            raise AdoptionError("DoRA / LoKr with convolution adapters are not on the fused path")
        self.unet_loras = mods                                 # plain list attribute (nn.Module.__setattr__ leaves lists alone)
        self._has_dropout = self.network_type == "lora" and any(bool(m.dropout or m.rank_dropout or m.module_dropout) for m in mods)
        self.text_encoder_loras = []
        self.lora_dim = int(getattr(foreign, "lora_dim", mods[0].lora_dim))
        self.conv_lora_dim = getattr(foreign, "conv_lora_dim", None)
        self.conv_alpha = getattr(foreign, "conv_alpha", None)
        self.alpha = getattr(foreign, "alpha", self.lora_dim)
        self.peft_format = bool(getattr(foreign, "peft_format", True))
        self.is_transformer = bool(getattr(foreign, "is_transformer", True))
        self.base_model_version = getattr(foreign, "base_model_version", None)
        self.base_model_ref = getattr(foreign, "base_model_ref", None)
        self.is_lorm = False
        # the reference draws its dropout masks with torch.rand on the global generators (network_mixins.py:200, 220): same provider as
        # FusedLoRANetwork's default
        self.mask_provider = lambda name, kind, shape, dev: torch.rand(shape, device="cpu" if kind == "module" else dev)
        self._arena_built = False
        dev = device if device is not None else _trainable(mods[0])[0].device
        if torch.device(dev).type != next(model.parameters()).device.type:
            raise AdoptionError(f"the adapter lives on {dev} but the native model on {next(model.parameters()).device}: call "
                                "network.force_to(device, torch.float32) first (BaseSDTrainProcess.py:1983)")
        for m in mods:
            for p in _trainable(m):
                if p.dtype != torch.float32:
                    raise AdoptionError(f"{m.lora_name}: adapter weights must be fp32 (network.force_to(device, torch.float32)), got {p.dtype}")
        groups = model.lora_groups() if hasattr(model, "lora_groups") else None
        self.build_arena(dev, ema=False, groups=groups)
        self._ops = ops
        self._expect = None
        self._record_pointers()
        self.refresh_shadows(ops)
        _LIVE.add(self)
        _GEN[0] += 1
        install_trainer_fusion()

    # ---- state that lives on the reference's network object
    @property
    def foreign(self):
        return self._foreign

    @property
    def training(self):
        return bool(self._foreign.training)

    @training.setter
    def training(self, v):
        pass

    @property
    def is_active(self):
        return bool(self._foreign.is_active)

    @is_active.setter
    def is_active(self, v):
        self._foreign.is_active = bool(v)

    @property
    def is_merged_in(self):
        return bool(getattr(self._foreign, "is_merged_in", False))

    @is_merged_in.setter
    def is_merged_in(self, v):
        self._foreign.is_merged_in = bool(v)

    @property
    def _multiplier(self):
        f = self._foreign
        v = f._multiplier if hasattr(f, "_multiplier") else (f.multiplier if hasattr(f, "multiplier") else f.torch_multiplier)
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().tolist()
        if isinstance(v, (list, tuple)) and len(v) == 1:
            v = v[0]
        return v

    @_multiplier.setter
    def _multiplier(self, v):
        self._foreign.multiplier = v

    @property
    def torch_multiplier(self):
        tm = getattr(self._foreign, "torch_multiplier", None)
        if tm is None:
            v = self._multiplier
            vals = [float(x) for x in v] if isinstance(v, (list, tuple)) else [float(v)]
            return torch.tensor(vals, dtype=torch.float32, device=self.arena_p.device)
        return tm.to(self.arena_p.device, torch.float32).reshape(-1)

    @torch_multiplier.setter
    def torch_multiplier(self, v):
        pass  # FusedLoRANetwork._update_torch_multiplier: the reference's own tensor is the source

    def _update_torch_multiplier(self):
        pass

    @property
    def dropout(self):
        return getattr(self._foreign, "dropout", None)

    @property
    def rank_dropout(self):
        return getattr(self._foreign, "rank_dropout", None)

    @property
    def module_dropout(self):
        return getattr(self._foreign, "module_dropout", None)

    @property
    def has_dropout(self):
        return self._has_dropout

    # ---- Parameter <-> arena aliasing
    def _record_pointers(self):
        self._expect = [(p, p.data_ptr()) for m in self.unet_loras for p in _trainable(m)]
        self._expect_by_id = {id(p): p for p, _ in self._expect}

    def aliasing_intact(self):
        return all(p.data_ptr() == ptr for p, ptr in self._expect)

    def sync(self, refresh=True):
        """Called at every forward of the native model.  Re-adopts Parameters whose storage was replaced since the arena was built (a
        `.to()`, a rank-changing `load_weights`) and — with the adapter active — refreshes the bf16 split shadows (and DoRA's column scales)
        from the fp32 arena: the reference's optimizer / EMA / load_state_dict write the Parameters without telling anyone."""
        if not self.aliasing_intact():
            dev = self.arena_p.device
            with torch.no_grad():
                for m in self.unet_loras:
                    for p in _trainable(m):
                        if p.device != dev or p.dtype != torch.float32:
                            raise AdoptionError(f"{m.lora_name}: adapter moved to {p.device}/{p.dtype}; the fused path needs it on {dev} in fp32")
            # same module set, possibly new shapes: rebuild (values are taken from the Parameters as they are now)
            grads = [(p, None if p.grad is None else p.grad.detach().clone()) for p, _ in self._expect]
            for ref in self._fusions:  # AdamW moments / EMA shadows that live in the arenas about to be replaced become tensors of their own
                fus = ref()
                if fus is not None:
                    fus.materialise()
            model = self._model_ref()
            self.build_arena(dev, ema=False, groups=model.lora_groups() if hasattr(model, "lora_groups") else None)
            with torch.no_grad():
                for p, g in grads:
                    if g is not None and g.shape == p.grad.shape:
                        p.grad.copy_(g)
            self._record_pointers()
        if refresh:
            self.refresh_shadows(self._ops)

    def grads_dropped(self):
        return self._expect[0][0].grad is None

    # ---- data parallelism under the reference's trainer
    # The reference wraps the network with accelerate / DDP (BaseSDTrainProcess.py:765-767), whose gradient all-reduce hangs off autograd's
    # per-parameter accumulation hooks.  The explicit backward writes the adapter gradients straight into the arena, so those hooks never
    # fire: under `accelerate launch` with N processes the replicas would silently train on their own shards only.  With a process group
    # initialised, every backward ends with ONE all-reduce(average) of the flat gradient arena (the same collective the fused train step
    # issues, RCCL over xGMI on the GPU).  Under gradient accumulation the arena holds avg(g_1) + local g_2 after the second backward and
    # averaging that again leaves avg(g_1) + avg(g_2): correct without a no_sync protocol.  dp_allreduce = False turns it off.
    # Round 6: issued like the fused train step's (trainer._on_grads_ready) — in TWO pieces, asynchronously, as soon as the explicit backward
    # declares a part of the arena final (FLUX: the single-stream blocks' adapters first, whose collective then runs behind the double-stream
    # blocks' backward; model.grad_ready_hook), joined at the end of the backward.  dp_overlap = False: one blocking all-reduce at the end.
    dp_allreduce = True
    dp_overlap = True

    def _dp_world(self):
        if not self.dp_allreduce:
            return 0
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()):
            return 0
        w = dist.get_world_size()
        return w if w > 1 else 0

    def before_backward(self, model):
        """Called by the autograd bridge in front of the explicit backward: arms the model's gradient-ready hook under data parallelism."""
        self._pending = []
        model.grad_ready_hook = self._on_grads_ready if (self._dp_world() and self.dp_overlap) else None

    def _on_grads_ready(self, which):
        import torch.distributed as dist

        model = self._model_ref()
        g = self.arena_g
        split = model.grad_split_offset(self) if hasattr(model, "grad_split_offset") else 0
        n_mat = getattr(self, "n_mat", g.numel())  # DoRA magnitude vectors sit at [n_mat, n): final only at the end
        ranges = [(split, n_mat)] if which in ("single", "late") else [(0, split), (n_mat, g.numel())]
        for a, b in ranges:
            if b > a:
                self._pending.append(dist.all_reduce(g[a:b], op=dist.ReduceOp.SUM, async_op=True))
                self.dp_pieces_issued += 1

    dp_pieces_issued = 0  # collectives issued from inside a backward (tests)

    def after_backward(self):
        world = self._dp_world()
        model = self._model_ref()
        if model is not None:
            model.grad_ready_hook = None
        if not world:
            return
        import torch.distributed as dist

        pending, self._pending = getattr(self, "_pending", []), []
        if pending:
            for w in pending:
                w.wait()
        else:
            dist.all_reduce(self.arena_g, op=dist.ReduceOp.SUM)
        self.arena_g.div_(world)

    def parameters(self, recurse=True):
        for p, _ in self._expect:
            yield p

    def attach_grad_views(self):
        super().attach_grad_views()
        # a Parameter whose .grad the trainer replaced by its own tensor (not a view of the arena) would swallow the kernels' output
        if self.network_type == "lokr":
            return
        for par, want, ptr in self._grad_views():
            if par.grad.data_ptr() != ptr:
                with torch.no_grad():
                    want.copy_(par.grad.reshape(want.shape))
                par.grad = want

    # ---- the reference's own object does the I/O: these are here so that code written against FusedLoRANetwork keeps working
    def get_state_dict(self, *a, **k):
        return self._foreign.get_state_dict(*a, **k)

    def save_weights(self, *a, **k):
        return self._foreign.save_weights(*a, **k)

    def load_weights(self, *a, **k):
        out = self._foreign.load_weights(*a, **k)
        self.sync()
        return out

    def prepare_optimizer_params(self, *a, **k):
        return self._foreign.prepare_optimizer_params(*a, **k)

    def __enter__(self):
        self._foreign.__enter__()

    def __exit__(self, *a):
        self._foreign.__exit__(*a)


def foreign_network_of(model):
    """The reference-built network whose adapters are attached to layers of `model` (None: no foreign adapter attached).  The layer that
    answered last time is asked first, so the steady-state cost is one attribute walk."""
    probe = model.__dict__.get("_foreign_probe")
    layers = [probe] if probe is not None and getattr(probe, "_foreign_adapter", False) and probe.lora is not None else model.modules()
    for l in layers:
        if getattr(l, "_foreign_adapter", False) and getattr(l, "lora", None) is not None:
            net = _network_of(l.lora)
            if net is None:
                raise AdoptionError(f"{getattr(l.lora, 'lora_name', '?')}: the adapter's network is gone (weak reference dead)")
            object.__setattr__(model, "_foreign_probe", l)
            return net
    return None


def resolve_network(model):
    """The network object the explicit graph consults for this forward: a FusedLoRANetwork attached with `attach_network`, or the
    AdoptedNetwork around the reference's own network (built on first use, synced on every use)."""
    net = model.__dict__.get("network")
    if net is not None and not isinstance(net, AdoptedNetwork):
        return net
    foreign = foreign_network_of(model)
    if foreign is None:
        return net
    if net is None or net.foreign is not foreign:
        net = AdoptedNetwork(foreign, model, model.ops)
        model.attach_network(net)
        return net  # shadows are fresh
    net.sync(refresh=net.is_active)
    return net


# ------------------------------------------------------------------------------------------------------------------------------------------
# The trainer's optimizer tail on the arena kernels.
#
# Per step the reference's trainer runs (extensions_built_in/sd_trainer/SDTrainer.py:2278-2293)
#
#     self.accelerator.clip_grad_norm_(self.params, max_grad_norm)     torch foreach ops over the 988 .grad views
#     self.optimizer.step()                                            torch.optim.AdamW(eps=1e-6) (toolkit/optimizer.py:78-79): foreach over 988 tensors
#     self.optimizer.zero_grad(set_to_none=True)
#     self.ema.update()                                                toolkit/ema.py:126-152: a Python loop, >= 3 tiny kernels per parameter
#
# over Parameters that, after adoption, are views of ONE flat arena.  Without touching the trainer, the two objects it calls are served from
# the arena kernels when — and only when — that is the same computation:
#
#   * `torch.optim.Optimizer` step hooks (torch.optim.optimizer.register_optimizer_step_pre_hook / _post_hook, process-global, installed when
#     the first network is adopted): a plain `torch.optim.AdamW` whose parameter set is exactly the adopted network's, all groups with the same
#     hyper-parameters, no amsgrad / maximize / closure, every parameter with its arena-view gradient -> ONE `aitk_adamw_ema_step` over the
#     arenas (no clip: the trainer's clip_grad_norm_ already ran on the same memory), its moments living in `arena_m` / `arena_v` with the
#     optimizer's own `state[p]['exp_avg' / 'exp_avg_sq']` re-pointed at views of them (so `optimizer.state_dict()`, the reference's
#     optimizer.pt, resumes and `load_state_dict` keep working), and torch's own step finds no gradients (they are hidden for the duration of
#     the call and handed back by the post-hook).  Anything else (8-bit Adam, Adafactor, Prodigy, per-group learning rates, a partial
#     gradient set) runs torch's / the optimizer's own code as before.
#   * `install_ema_fusion(cls)` wraps `cls.update` of the reference's `ExponentialMovingAverage` (called with toolkit.ema's class by
#     integration/extensions/aitk_mi355, or found in sys.modules at adoption): an EMA over exactly the adopted parameters, fp32 shadows on the
#     arena's device -> shadows re-pointed at views of `arena_ema`, `update()` = the class's own decay bookkeeping + ONE `aitk_ema_update`.
#
# AITK_FUSE_TRAINER_STEP=0 (or AdoptedNetwork.fuse_trainer_step = False) leaves both objects alone.
_LIVE = weakref.WeakSet()  # AdoptedNetworks alive in this process
_GEN = [0]                 # bumped whenever a network is adopted: negative matches cached on optimizers / EMAs are re-examined
_HOOKS_INSTALLED = [False]
STATS = {"adamw_fused": 0, "adamw_fallback": 0, "ema_fused": 0, "ema_fallback": 0}  # calls served by the arena kernels / left to torch (process-wide)


def fusion_enabled():
    return bool(AdoptedNetwork.fuse_trainer_step) and _os.environ.get("AITK_FUSE_TRAINER_STEP", "1") != "0"


def _arena_twin(net, par, arena):
    """The view of `arena` (arena_m / arena_v / arena_ema) with the geometry `par` has inside arena_p."""
    off = par.storage_offset() - net.arena_p.storage_offset()
    return torch.as_strided(arena, par.shape, par.stride(), arena.storage_offset() + off)


def _match_network(param_ids):
    for net in list(_LIVE):
        if net._expect is not None and len(net._expect) == len(param_ids) and {id(p) for p, _ in net._expect} == param_ids:
            return net
    return None


class _OptimizerFusion:
    """One torch.optim.AdamW instance served by `aitk_adamw_ema_step` over the arenas of the network whose parameters it holds."""

    def __init__(self, opt, net):
        self.opt = weakref.ref(opt)
        self.net = weakref.ref(net)
        self.params = [p for g in opt.param_groups for p in g["params"]]
        self.step_count = 0
        self.steps_current = True   # state[p]['step'] tensors hold step_count (they are only written when somebody looks)
        self.arena_m = None         # the arena objects the state views were cut from (a rebuilt arena is a new object)
        self.hidden = None
        self.fused_steps = 0
        # state[p]['step'] is written lazily — right before somebody reads the state (state_dict(), torch's own step); a torch without the
        # state_dict pre-hook gets it written after every fused step instead (988 host-side fills)
        self.eager_steps = not hasattr(opt, "register_state_dict_pre_hook")
        if not self.eager_steps:
            opt.register_state_dict_pre_hook(lambda o: self.flush_steps())
        net._fusions.append(weakref.ref(self))

    # -- eligibility, per call (cheap: group hyper-parameters and a pointer spot check)
    def _hyper(self):
        opt = self.opt()
        if type(opt) is not torch.optim.AdamW:
            return None
        g0 = opt.param_groups[0]
        n = 0
        for g in opt.param_groups:
            if g.get("amsgrad") or g.get("maximize") or g.get("differentiable") or g.get("capturable"):
                return None
            for k in ("lr", "betas", "eps", "weight_decay"):
                a, b = g[k], g0[k]
                if isinstance(a, torch.Tensor) or isinstance(b, torch.Tensor):
                    a, b = (float(x) if not isinstance(x, tuple) else x for x in (a, b))
                if a != b:
                    return None  # per-group hyper-parameters: torch's own step
            n += len(g["params"])
        if n != len(self.params):
            return None          # add_param_group since the match
        return float(g0["lr"]), float(g0["betas"][0]), float(g0["betas"][1]), float(g0["eps"]), float(g0["weight_decay"])

    def _state_in_arena(self, net):
        if self.arena_m is not net.arena_m:
            return False
        st = self.opt().state
        for p in (self.params[0], self.params[-1]):
            s = st.get(p)
            if not s or "exp_avg" not in s or s["exp_avg"].data_ptr() != _arena_twin(net, p, net.arena_m).data_ptr():
                return False
        return True

    def _adopt_state(self, net):
        """Moments into arena_m / arena_v, `state[p]` re-pointed at views of them.  Returns False (nothing changed) when the optimizer's
        state cannot be expressed by one step count."""
        opt = self.opt()
        steps = set()
        for p in self.params:
            s = opt.state.get(p)
            steps.add(float(s["step"]) if s and "step" in s else 0.0)
        if len(steps) != 1:
            return False
        with torch.no_grad():
            for p in self.params:
                s = opt.state.get(p)
                mv, vv = _arena_twin(net, p, net.arena_m), _arena_twin(net, p, net.arena_v)
                if s and "exp_avg" in s:
                    if s["exp_avg"].data_ptr() != mv.data_ptr():
                        mv.copy_(s["exp_avg"])
                        vv.copy_(s["exp_avg_sq"])
                    step_t = s["step"] if isinstance(s.get("step"), torch.Tensor) and s["step"].device.type == "cpu" else torch.tensor(0.0, dtype=torch.float32)
                else:
                    mv.zero_()
                    vv.zero_()
                    step_t = torch.tensor(0.0, dtype=torch.float32)
                opt.state[p] = {"step": step_t, "exp_avg": mv, "exp_avg_sq": vv}
        self.step_count = int(steps.pop())
        self.steps_current = False
        self.flush_steps()
        self.arena_m = net.arena_m
        return True

    def _fallback(self):
        STATS["adamw_fallback"] += 1
        return self.flush_steps()

    def flush_steps(self):
        """state[p]['step'] <- the number of steps applied (before anybody reads the state: state_dict(), torch's own step)."""
        if self.steps_current:
            return
        opt = self.opt()
        if opt is None:
            return
        for p in self.params:
            s = opt.state.get(p)
            if s and "step" in s:
                s["step"].fill_(float(self.step_count))
        self.steps_current = True

    def materialise(self):
        """Before the arenas are rebuilt (AdoptedNetwork.sync): the moments become tensors of their own again; the next fused step moves them
        into the new arena."""
        opt = self.opt()
        if opt is None or self.arena_m is None:
            return
        self.flush_steps()
        for p in self.params:
            s = opt.state.get(p)
            if s and "exp_avg" in s:
                s["exp_avg"], s["exp_avg_sq"] = s["exp_avg"].clone(), s["exp_avg_sq"].clone()
        self.arena_m = None

    # -- the hooks
    def pre(self, args, kwargs):
        self.hidden = None
        net, opt = self.net(), self.opt()
        if net is None or not fusion_enabled():
            return self._fallback()
        if (len(args) > 1 and args[1] is not None) or kwargs.get("closure") is not None:  # args = (optimizer, closure?) as torch's step wrapper passes them
            return self._fallback()
        hp = self._hyper()
        if hp is None or not net.aliasing_intact():
            return self._fallback()
        grads = [p.grad for p in self.params]
        n_none = sum(g is None for g in grads)
        if n_none == len(grads):
            return None              # nothing to step (torch skips parameters without .grad): no state change either way
        if n_none or any(grads[i].data_ptr() != _arena_twin(net, self.params[i], net.arena_g).data_ptr() for i in (0, -1)):
            return self._fallback()  # a partial gradient set / gradients that are not the arena's: torch's own step
        if not self._state_in_arena(net):
            if not self._adopt_state(net):
                return self._fallback()
        elif self.steps_current:     # torch's own step may have run since (a fallback call): the tensors are the truth
            self.step_count = int(float(opt.state[self.params[0]]["step"]))
        lr, b1, b2, eps, wd = hp
        self.step_count += 1
        self.steps_current = False
        with torch.no_grad():  # step hooks run outside the optimizer's own no_grad region
            net._ops.adamw_ema_step(net.arena_p, net.arena_g, net.arena_m, net.arena_v, lr=lr, beta1=b1, beta2=b2, eps=eps, weight_decay=wd,
                                    step=self.step_count, max_norm=0.0, ema=None)
        self.fused_steps += 1
        STATS["adamw_fused"] += 1
        if self.eager_steps:
            self.flush_steps()
        for p in self.params:        # torch's own step now finds nothing to do
            p.grad = None
        self.hidden = grads

    def post(self):
        if self.hidden is not None:
            for p, g in zip(self.params, self.hidden):
                if p.grad is None:
                    p.grad = g
            self.hidden = None


def _fusion_of(opt):
    ent = opt.__dict__.get("_aitk_fusion")
    if ent is not None and ent[0] is not None and ent[0].net() is None:
        ent = None  # the network this optimizer was matched with is gone (the model adopted the trainer's network anew): match again
    if ent is not None and (ent[0] is not None or ent[1] == _GEN[0]):
        return ent[0]
    fus = None
    if type(opt) is torch.optim.AdamW and _LIVE:
        net = _match_network({id(p) for g in opt.param_groups for p in g["params"]})
        if net is not None:
            fus = _OptimizerFusion(opt, net)
    opt.__dict__["_aitk_fusion"] = (fus, _GEN[0])
    return fus


def _optimizer_pre_hook(opt, args, kwargs):
    if not _LIVE:
        return None
    fus = _fusion_of(opt)
    if fus is not None:
        fus.pre(args, kwargs)
    return None


def _optimizer_post_hook(opt, args, kwargs):
    ent = opt.__dict__.get("_aitk_fusion")
    if ent is not None and ent[0] is not None:
        ent[0].post()
    return None


class _EmaFusion:
    """One `ExponentialMovingAverage` (toolkit/ema.py) whose shadows live in `arena_ema` of the network whose parameters it averages."""

    def __init__(self, ema, net):
        self.ema = weakref.ref(ema)
        self.net = weakref.ref(net)
        self.arena = None
        self.fused_updates = 0
        net._fusions.append(weakref.ref(self))

    def _in_arena(self, net, ema, params):
        if self.arena is None or self.arena is not net.arena_ema or len(ema.shadow_params) != len(params):
            return False
        return all(ema.shadow_params[i].data_ptr() == _arena_twin(net, params[i], net.arena_ema).data_ptr() for i in (0, -1))

    def _adopt(self, net, ema, params):
        dev = net.arena_p.device
        if any(s.dtype != torch.float32 or s.device != dev or s.shape != p.shape for s, p in zip(ema.shadow_params, params)):
            return False
        if net.arena_ema is None or net.arena_ema.numel() != net.arena_p.numel():
            net.arena_ema = torch.zeros_like(net.arena_p)
        with torch.no_grad():
            for i, p in enumerate(params):
                view = _arena_twin(net, p, net.arena_ema)
                if ema.shadow_params[i].data_ptr() != view.data_ptr():
                    view.copy_(ema.shadow_params[i])
                ema.shadow_params[i] = view
        self.arena = net.arena_ema
        return True

    def materialise(self):
        ema = self.ema()
        if ema is None or self.arena is None:
            return
        ema.shadow_params = [s.clone() for s in ema.shadow_params]
        self.arena = None

    def update(self, ema, params):
        """True when the arena kernel did the update."""
        net = self.net()
        if net is None or not fusion_enabled() or not net.aliasing_intact():
            return False
        if any(p.dtype != torch.float32 for p in (params[0], params[-1])):
            return False
        if not self._in_arena(net, ema, params) and not self._adopt(net, ema, params):
            return False
        decay = ema.decay  # toolkit/ema.py:117-125
        if ema.num_updates is not None:
            ema.num_updates += 1
            decay = min(decay, (1 + ema.num_updates) / (10 + ema.num_updates))
        with torch.no_grad():
            net._ops.ema_update(net.arena_p, net.arena_ema, decay=decay, ema_feedback=10.0 if getattr(ema, "use_feedback", False) else 0.0,
                                param_multiplier=float(getattr(ema, "param_multiplier", 1.0)))
        self.fused_updates += 1
        STATS["ema_fused"] += 1
        return True


def install_ema_fusion(cls):
    """Wrap `cls.update` (the reference's toolkit.ema.ExponentialMovingAverage, or a class with its attributes: shadow_params, decay,
    num_updates, use_feedback, param_multiplier, _get_parameters).  Idempotent."""
    if getattr(cls.update, "_aitk_wrapped", False):
        return cls
    orig = cls.update

    def update(self, parameters=None):
        if _LIVE and fusion_enabled():
            ent = self.__dict__.get("_aitk_fusion")
            params = None
            if ent is None or (ent[0] is None and ent[1] != _GEN[0]):
                params = list(self._get_parameters(parameters))
                net = _match_network({id(p) for p in params}) if params else None
                ent = (_EmaFusion(self, net) if net is not None else None, _GEN[0])
                self.__dict__["_aitk_fusion"] = ent
            if ent[0] is not None:
                params = params if params is not None else list(self._get_parameters(parameters))
                net = ent[0].net()
                if net is not None and len(params) == len(net._expect) and params[0] is net._expect_by_id.get(id(params[0])) \
                        and params[-1] is net._expect_by_id.get(id(params[-1])) and ent[0].update(self, params):
                    return None
            STATS["ema_fallback"] += 1
        return orig(self, parameters)

    update._aitk_wrapped = True
    update._aitk_orig = orig
    cls.update = update
    return cls


def install_trainer_fusion():
    """Process-global optimizer step hooks (once) + the EMA wrap when the reference's toolkit.ema is loaded."""
    if not _HOOKS_INSTALLED[0]:
        try:
            from torch.optim.optimizer import register_optimizer_step_post_hook, register_optimizer_step_pre_hook
        except ImportError:  # a torch without global optimizer step hooks (< 2.0): the trainer's optimizer runs its own code
            _HOOKS_INSTALLED[0] = True
        else:
            register_optimizer_step_pre_hook(_optimizer_pre_hook)
            register_optimizer_step_post_hook(_optimizer_post_hook)
            _HOOKS_INSTALLED[0] = True
    mod = _sys.modules.get("toolkit.ema")
    cls = getattr(mod, "ExponentialMovingAverage", None) if mod is not None else None
    if cls is not None:
        install_ema_fusion(cls)
