"""One LoRA training step for FLUX.1 on MI355X (host orchestration of the C-ABI kernels).

Mirrors, in order, what the reference does per step:
  process_general_training_batch .... jobs/process/BaseSDTrainProcess.py:1036-1478 (timesteps, noise, add_noise)
  predict_noise (flux branch) ....... toolkit/stable_diffusion_model.py:2154-2222 (pack, ids, guidance = 1.0, unpack)
  calculate_loss .................... extensions_built_in/sd_trainer/SDTrainer.py:644-646, 916, 987-1013
  backward / clip / AdamW / EMA ..... SDTrainer.py:2238, 2278-2293; toolkit/optimizer.py:78-79; toolkit/ema.py:116-152

Data parallelism: one process per GPU, identical frozen base + adapter state on every rank, each rank
steps its own shard of the bucket batch, ONE all-reduce(mean) of the flat fp32 adapter-gradient arena (RCCL over
xGMI), issued in two pieces so the single-stream half overlaps the double-stream blocks' backward; clip/AdamW/EMA then
run redundantly on every rank (bit-identical adapter weights across ranks).
"""
import torch

from .flowmatch import FlowMatchTrainSchedule, get_noise


class _LoRATrainStepBase:
    """State and the model-independent tail of a step: gradient all-reduce pieces, clip + AdamW + EMA, shadow refresh."""

    def __init__(self, model, network, ops, *, lr=1e-4, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-6,
                 max_grad_norm=1.0, ema_decay=0.0, timestep_type="linear", guidance=1.0, process_group=None,
                 seed=None, schedule=None, lr_scheduler=None, noise_options=None, linear_timesteps=False, linear_timesteps2=False,
                 latent_multiplier=1.0, adaptive_scaling_factor=False, noisy_latent_multiplier=1.0, loss_type="mse",
                 ema_use_feedback=False, ema_param_multiplier=1.0, ema_use_num_updates=False, allreduce_dtype="fp32",
                 nonfinite_guard=True, max_loss=None):
        self.model, self.network, self.ops = model, network, ops
        # Failure handling of the reference's loop, on the device (no host sync; the reference reads loss.item(), we never do):
        #   * non-finite loss -> the micro-batch contributes no gradient and reports 0 (SDTrainer.py:2221-2224: the replacement has no graph,
        #     so no parameter receives a .grad);
        #   * train.max_loss -> clamp(loss, max=max_loss): above it the gradient is ZERO but present (SDTrainer.py:1049-1050: autograd runs
        #     through the clamp), so the optimizer still steps on g = 0 — weight decay, moment decay, step count — like torch.optim.AdamW does;
        #   * a step whose every micro-batch had a non-finite loss, or whose gradient arena holds a NaN / Inf (norm not finite), is SKIPPED: p,
        #     m, v and the AdamW step count stay as they are — what torch.optim.AdamW does when no parameter has a .grad — and one bad batch
        #     cannot poison the moments, the EMA or the bf16 shadows.  `guard_counters()` reads the tallies back (a host sync, call it rarely).
        #     Under data parallelism the count of gradient-less micro-batches is summed over the ranks before the decision (one 4-byte
        #     all-reduce): the gradients every rank applies are the all-reduced ones, so every rank must take the SAME decision.
        self.max_loss = float(max_loss) if max_loss else None
        self.guard = torch.zeros(8, dtype=torch.int32, device=network.arena_p.device) if (nonfinite_guard or self.max_loss) else None
        self._n_micro = 0
        # train.loss_type (SDTrainer.py:903-916): mse (default) / mae / pseudo_huber run as modes of aitk_mse_loss_grad; wavelet,
        # stepped, pixelspace and mean_flow are other losses of the reference and are refused
        if loss_type not in ("mse", "mae", "pseudo_huber"):
            raise ValueError(f"loss_type {loss_type!r}: the fused step implements mse, mae and pseudo_huber")
        self.loss_type = loss_type
        # train.ema_config.use_feedback / param_multiplier (BaseSDTrainProcess.py:798-803 -> toolkit/ema.py:135-143) and the class's
        # use_num_updates warm-up (ema.py:118-123: decay = min(decay, (1 + n) / (10 + n)))
        self.ema_feedback = 10.0 if ema_use_feedback else 0.0
        self.ema_param_multiplier = float(ema_param_multiplier)
        self.ema_num_updates = 0 if ema_use_num_updates else None
        # jobs/process/BaseSDTrainProcess.py:1393-1401 (latents * latent_multiplier, or 1 / (per-channel std + 1e-6) with
        # adaptive_scaling_factor) and 1467-1470 (noisy latents * noisy_latent_multiplier)
        self.latent_multiplier, self.adaptive_scaling_factor = float(latent_multiplier), bool(adaptive_scaling_factor)
        self.noisy_latent_multiplier = float(noisy_latent_multiplier)
        self.lr, self.weight_decay, self.betas, self.eps = lr, weight_decay, betas, eps
        self.max_grad_norm, self.ema_decay = max_grad_norm, ema_decay
        self.timestep_type, self.guidance = timestep_type, guidance
        # SDTrainer.py:923-944: linear_timesteps / linear_timesteps2 / timestep_type 'weighted' multiply the per-sample loss by
        # the scheduler's weight of each sampled timestep
        self.linear_timesteps, self.linear_timesteps2 = bool(linear_timesteps), bool(linear_timesteps2)
        self.schedule = schedule or FlowMatchTrainSchedule()
        self.step_num = 0
        self.noise_options = dict(noise_options or {})  # keywords of flowmatch.get_noise (noise_offset, noise_multiplier, ...)
        self.lr_scheduler = lr_scheduler  # LRSchedule or None (constant): stepped once per train-loop iteration
        if lr_scheduler is not None:
            self.lr = lr_scheduler.get_last_lr()[0]
        # the gradient all-reduce runs in fp32 (default: DP(P) == one rank on the concatenated batch up to fp32 summation
        # order) or in bf16 (half the bytes on the xGMI links: each rank's gradient is rounded once to bf16, RCCL sums in bf16, the sum is
        # expanded over the fp32 arena — a stated deviation of ~2^-9 relative per element, tests/test_train_step_cpu.py)
        if allreduce_dtype not in ("fp32", "bf16"):
            raise ValueError(f"allreduce_dtype {allreduce_dtype!r}: fp32 or bf16")
        self.allreduce_dtype = allreduce_dtype
        self._transport = None  # bf16 transport buffer, one per arena (allocated on first use)
        self.pg = process_group
        self.world = 1
        self.dp = process_group is not None  # a 1-rank group still walks the all-reduce path (RCCL smoke on one GPU)
        self._pending = []
        self._graphs, self._graph_pool = {}, None
        self.collect_dp_timing = False  # bench.py: events around the all-reduce wait (what of the collective is NOT hidden)
        self.dp_wait_events = []
        if process_group is not None:
            import torch.distributed as dist

            self.world = dist.get_world_size(process_group)
        dev = network.arena_p.device
        self.device = dev
        self.gen = None
        if seed is not None:
            self.gen = torch.Generator(device=dev)
            self.gen.manual_seed(seed)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.loss_per_sample = None
        self._loss_per_sample_by_B = {}
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        if ema_decay > 0 and network.arena_ema is None:
            network.arena_ema = network.arena_p.clone()
        # arena split point: adapters at [split, n) get their final gradients first during backward
        self._split = model.grad_split_offset(network)
        model.grad_ready_hook = None  # set per backward pass (only the last micro-batch issues the all-reduce)

    def set_step_count(self, n):
        """Number of optimizer steps applied so far (AdamW's bias corrections count from it).  With the failure guard the count lives on the
        device (guard[3]: a skipped step does not advance it), so resuming from a checkpoint or replaying a step sets both copies."""
        self.step_num = int(n)
        if self.guard is not None:
            self.guard[3:4].fill_(int(n))

    def guard_counters(self):
        """{'nonfinite_losses', 'clamped_losses', 'steps_applied', 'steps_skipped', 'last_step_skipped'} read back from the device (host sync)."""
        if self.guard is None:
            return None
        g = self.guard.tolist()
        return {"nonfinite_losses": g[1], "clamped_losses": g[2], "steps_applied": g[3], "steps_skipped": g[4], "last_step_skipped": bool(g[5])}

    def _scale_latents(self, latents):
        """batch.latents of the reference: the cached / encoded latents times latent_multiplier (BaseSDTrainProcess.py:1393-1411)."""
        if self.adaptive_scaling_factor:
            if latents.dim() != 4:
                raise ValueError("adaptive_scaling_factor: image latents only (the reference reduces over dims (2, 3))")
            return latents * (1 / (latents.std(dim=(2, 3), keepdim=True) + 1e-6))
        return latents if self.latent_multiplier == 1.0 else latents * self.latent_multiplier

    def _scale_noisy(self, noisy):
        if self.noisy_latent_multiplier != 1.0:
            self.ops.ew(3, noisy.view(-1, noisy.shape[-1]), noisy.view(-1, noisy.shape[-1]), alpha=self.noisy_latent_multiplier)
        return noisy

    # ------------------------------------------------------------------ DP
    def _on_grads_ready(self, which):
        import torch.distributed as dist

        g = self.network.arena_g
        n_mat = getattr(self.network, "n_mat", g.numel())  # DoRA magnitude vectors sit at [n_mat, n): final only at the end
        ranges = [(self._split, n_mat)] if which in ("single", "late") else [(0, self._split), (n_mat, g.numel())]
        for a, b in ranges:
            if b <= a:
                continue
            if self.allreduce_dtype == "bf16":
                if self._transport is None:  # indexed like the arena: piece [a, b) travels as transport[a:b]
                    self._transport = torch.empty(g.numel(), dtype=torch.bfloat16, device=g.device)
                buf = self._transport[a:b]
                self.ops.grad_compress_bf16(g[a:b], buf)
                self._pending.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), buf, (a, b)))
            else:
                self._pending.append((dist.all_reduce(g[a:b], op=dist.ReduceOp.SUM, group=self.pg, async_op=True), None, None))

    def _table_type(self):
        """timestep table the scheduler is set to: `linear_timesteps` / `linear_timesteps2` force 'linear' whatever timestep_type says
        (jobs/process/BaseSDTrainProcess.py:1196-1203); timestep_type itself still selects the 'weighted' loss-weight lookup."""
        return "linear" if (self.linear_timesteps or self.linear_timesteps2) else self.timestep_type

    def _timestep_loss_weight(self, timesteps, loss_weight):
        """per-sample loss weights of the flow-matching timestep weighting (None when off), folded into `loss_weight`."""
        if not (self.linear_timesteps or self.linear_timesteps2 or self.timestep_type == "weighted"):
            return loss_weight
        tw = self.schedule.get_weights_for_timesteps(timesteps, v2=self.linear_timesteps2, timestep_type=self.timestep_type).float()
        return tw if loss_weight is None else loss_weight * tw

    def _finish_allreduce(self):
        for w, buf, rng in self._pending:
            w.wait()
            if buf is not None:
                self.ops.grad_expand_bf16(buf, self.network.arena_g[rng[0]:rng[1]])
        self._pending = []

    @staticmethod
    def pack_mask(mask):
        """reference mask_multiplier [B,1,H,W] (or [B,1,F,H,W]) at latent resolution, already divided by its mean
        (SDTrainer.py:1484-1504) -> fp32 [B, tokens, 4] in the packed 2x2-patch token order of the prediction."""
        if mask.dim() == 4:
            mask = mask[:, :, None]
        B, _, Fr, Hh, W = mask.shape
        m = mask.float().reshape(B, Fr, Hh // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5)
        return m.reshape(B, Fr * (Hh // 2) * (W // 2), 4).contiguous()

    # ------------------------------------------------------------------ hipGraph replay of the launch sequence
    # A micro-batch is split into `_prepare` (host work: timestep sampling, noise draw, table look-ups, loss weights -> a dict of
    # device tensors) and `_run` (nothing but kernel launches on the current stream: noise mix, forward, loss, backward).  `_run`
    # of one bucket shape is captured once into a hipGraph (torch.cuda.CUDAGraph on ROCm) and replayed with the prepared tensors
    # copied into its static inputs: ~5 000 launches per step leave the Python host path, which is what bounds the small-batch
    # and UNet steps (the reference's default batch size is 1).  All graphs share one memory pool (they replay one at a time), so
    # the activations of every bucket shape alias the same HBM.  Clip / AdamW / EMA and the DP all-reduce stay outside the graph.
    def _graph_key(self, prepared):
        return tuple((k, tuple(v.shape), str(v.dtype)) if torch.is_tensor(v) else (k, repr(v)) for k, v in sorted(prepared.items()))

    def capture(self, **batch):
        """Capture the launch sequence of this batch's bucket shape; returns the graph entry (also cached by shape)."""
        return self._capture_prepared(self._prepare(**batch))

    def _capture_prepared(self, prepared):
        key = self._graph_key(prepared)
        if key in self._graphs:
            return self._graphs[key]
        if self.network.training and self.network.has_dropout and not getattr(self.network, "dropout_is_capturable", lambda: False)():
            # the module_dropout coin is a HOST decision made while the launch list is built (an adapter is launched or not): a captured graph
            # would replay ONE decision for ever (toolkit/network_mixins.py:197-239 draws per call); the same holds for a custom mask provider.
            # dropout / rank_dropout alone are device-side torch.rand draws on the default generator, whose offset every replay advances:
            # those graphs draw fresh masks (round 6)
            raise NotImplementedError("hipGraph replay with LoRA module_dropout (or a custom mask provider): use step() (eager launches)")
        static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in prepared.items()}
        # warm-up on a side stream (workspaces, function attributes, RoPE tables are created here, not under capture)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.network.zero_grad_arena()
            self._run(static, final=False)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()  # the warm-up's activations go back to the driver: the graph pool takes their place
        g = torch.cuda.CUDAGraph()
        self.network.zero_grad_arena()
        n0 = self._n_micro
        with torch.cuda.graph(g, pool=self._graph_pool):
            loss = self._run(static, final=False)
        n_loss = self._n_micro - n0  # loss launches inside the graph: what a replay adds to the step's micro-batch count
        self._n_micro = 0
        if self.guard is not None:
            self.guard[0:1].zero_()  # the warm-up run is not a step: its gate count must not reach the next optimizer launch
        if self._graph_pool is None:
            self._graph_pool = g.pool()
        # the graph holds raw pointers: every grow-only kernel workspace as it was during capture (ops.workspace replaces a buffer
        # when a larger bucket asks for more) and this batch size's per-sample loss buffer stay referenced by the entry, so a later
        # reallocation cannot hand their memory to another tensor while the graph can still be replayed
        ent = {"graph": g, "static": static, "loss": loss, "n_loss": n_loss, "keepalive": (list(getattr(self.ops, "_ws", {}).values()),
                                                                                          dict(self._loss_per_sample_by_B))}
        self._graphs[key] = ent
        return ent

    def step_graphed(self, **batch):
        """`step` through the captured graph of the batch's bucket shape (captured on first use)."""
        prepared = self._prepare(**batch)
        ent = self._graphs.get(self._graph_key(prepared)) or self._capture_prepared(prepared)
        for k, v in prepared.items():
            if torch.is_tensor(v):
                ent["static"][k].copy_(v)
        self.network.zero_grad_arena()
        ent["graph"].replay()
        self._n_micro = ent.get("n_loss", 0)
        if self.dp:  # replay finishes every gradient at once: both pieces go out back to back (not overlapped with backward)
            self._on_grads_ready("single")
            self._on_grads_ready("double")
        self._optimizer_step()
        if self.lr_scheduler is not None:
            self.lr = self.lr_scheduler.step()
        return ent["loss"]  # what _run returned under capture (with `preservation`: the SUM of both terms, not the last mse launch)

    def _single(self, final=True, **batch):
        return self._run(self._prepare(**batch), final=final)

    def step_list(self, batches):
        """The reference's `gradient_accumulation` batch list (SDTrainer.py:2243-2293): gradients are zeroed once, every
        micro-batch (any bucket resolution) runs forward + backward into the same fp32 arena, the micro-batch losses are SUMMED
        (no 1/len scaling, like the reference), then one clip / AdamW / EMA.  The DP all-reduce is issued only by the last
        micro-batch's backward.  `batches` is a list of keyword dicts for `step`."""
        self.network.zero_grad_arena()
        total = None
        for i, b in enumerate(batches):
            loss = self._single(final=(i == len(batches) - 1), **b)
            if len(batches) > 1:
                total = loss.clone() if total is None else total.add_(loss)
            else:
                total = loss
        self._optimizer_step()
        if self.lr_scheduler is not None:  # SDTrainer.py:2298-2300
            self.lr = self.lr_scheduler.step()
        return total

    def _loss_backward(self, pred, target, loss_weight, loss_mask=None, final=True):
        """MSE loss + explicit backward accumulating into the gradient arena (caller holds `with network`)."""
        ops, model, net = self.ops, self.model, self.network
        B = pred.shape[0]
        dpred = torch.empty_like(pred)
        self.loss_per_sample = self._loss_per_sample_by_B.get(B)  # one buffer per batch size, never re-created (captured graphs point at it)
        if self.loss_per_sample is None:
            self.loss_per_sample = self._loss_per_sample_by_B[B] = torch.zeros(B, dtype=torch.float32, device=pred.device)
        kw = {} if self.loss_type == "mse" else {"loss_type": self.loss_type}
        if self.guard is not None:
            kw.update(guard=self.guard, max_loss=self.max_loss)
            self._n_micro += 1
        ops.mse_loss_grad(pred, target, dpred, self.loss_per_sample, self.loss, weight=loss_weight,
                          mask=loss_mask, **kw)
        model.grad_ready_hook = self._on_grads_ready if (final and self.dp) else None
        model.backward_native(dpred)  # inside `with network` like the reference (SDTrainer.py:2229-2238)
        return self.loss

    def _optimizer_step(self):
        ops, net = self.ops, self.network
        self.step_num += 1
        grad_scale = 1.0
        if self.dp:
            if self.collect_dp_timing:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()  # on the launch stream, behind the last backward kernel
            self._finish_allreduce()
            if self.collect_dp_timing:
                e1.record()  # behind the stream-wait on the collective: e1 - e0 = exposed all-reduce time
                self.dp_wait_events.append((e0, e1))
            grad_scale = 1.0 / self.world
        decay, kw = self.ema_decay, {}
        if self.ema_decay > 0:
            if self.ema_num_updates is not None:
                self.ema_num_updates += 1
                decay = min(decay, (1 + self.ema_num_updates) / (10 + self.ema_num_updates))
            if self.ema_feedback or self.ema_param_multiplier != 1.0:
                kw = dict(ema_feedback=self.ema_feedback, param_multiplier=self.ema_param_multiplier)
        if self.guard is not None:
            n_micro = self._n_micro
            if self.dp:  # the skip decision must be the same on every rank (they all apply the same averaged gradient): sum the local counts
                import torch.distributed as dist

                dist.all_reduce(self.guard[0:1], op=dist.ReduceOp.SUM, group=self.pg)
                n_micro *= self.world
            kw.update(guard=self.guard, n_micro=n_micro)
            self._n_micro = 0
        ops.adamw_ema_step(net.arena_p, net.arena_g, net.arena_m, net.arena_v, lr=self.lr, beta1=self.betas[0],
                           beta2=self.betas[1], eps=self.eps, weight_decay=self.weight_decay, step=self.step_num,
                           max_norm=self.max_grad_norm, ema=net.arena_ema if self.ema_decay > 0 else None,
                           ema_decay=decay, grad_scale=grad_scale, norm_out=self.grad_norm, **kw)
        net.refresh_shadows(ops)
        return self.loss


class FluxLoRATrainStep(_LoRATrainStepBase):
    # ------------------------------------------------------------------ one step
    def step(self, latents, prompt_embeds, pooled_embeds, **kw):
        """latents [B,16,H,W] (scaled VAE latents), prompt_embeds [B,512,4096], pooled_embeds [B,768]; keywords of `_single`.
        Returns the device-resident loss tensor (no host sync)."""
        return self.step_list([dict(latents=latents, prompt_embeds=prompt_embeds, pooled_embeds=pooled_embeds, **kw)])

    def _prepare(self, latents, prompt_embeds, pooled_embeds, *, noise=None, timesteps=None, loss_weight=None, loss_mask=None,
                 preservation=None, preservation_multiplier=1.0):
        """Host side of one micro-batch: timestep / noise draws and loss weights -> device tensors for `_run`."""
        dt = self.model.dt
        B = latents.shape[0]
        dev = latents.device
        latents = latents.to(dt)
        self.schedule.set_train_timesteps(1000, dev, self._table_type(), latents=latents, patch_size=2)
        if timesteps is None:
            timesteps, _ = self.schedule.sample_timesteps(B, dev, generator=self.gen)
        loss_weight = self._timestep_loss_weight(timesteps, loss_weight)
        if noise is None:  # randn in fp32 on device, then cast (toolkit/stable_diffusion_model.py:1803-1812) + the noise options;
            # drawn and shaped from the UNSCALED latents: latent_multiplier is applied after it (BaseSDTrainProcess.py:1323-1401)
            noise = get_noise(latents, self.gen, dtype=dt, **self.noise_options)
        latents = self._scale_latents(latents).contiguous()
        p = dict(latents=latents, noise=noise.to(dt).contiguous(), timesteps=timesteps.float().contiguous(),
                 prompt_embeds=prompt_embeds, pooled_embeds=pooled_embeds, loss_weight=loss_weight,
                 loss_mask=self.pack_mask(loss_mask) if loss_mask is not None else None)
        if preservation is not None:
            p.update(pres_embeds=preservation[0], pres_pooled=preservation[1],
                     pres_weight=torch.full((B,), float(preservation_multiplier), dtype=torch.float32, device=dev))
        return p

    def _run(self, p, final=True):
        """Device side of one micro-batch: noise / pack, forward, loss, backward — kernel launches only (capturable).  With
        `pres_embeds` / `pres_pooled` it adds the reference's diff-output / blank-prompt preservation term (SDTrainer.py:1229-1247,
        2182-2220): the base model's prediction on those embeds (adapter inactive, nothing saved) is the target of a second
        adapter-active pass on the same noisy latents, `mse(pred, prior) * preservation_multiplier` joins the loss.  The two
        backward passes accumulate into the same arena, which equals the reference's single backward of the summed loss."""
        ops, model, net = self.ops, self.model, self.network
        latents, timesteps = p["latents"], p["timesteps"]
        dt = model.dt
        B, Cc, Hh, W = latents.shape
        dev = latents.device
        n_tok = (Hh // 2) * (W // 2)
        noisy = torch.empty(B, n_tok, Cc * 4, dtype=dt, device=dev)
        target = torch.empty_like(noisy)
        ops.flow_noise_pack(latents, p["noise"], timesteps, noisy, target)
        self._scale_noisy(noisy)
        img_ids, txt_ids = make_ids(Hh, W, p["prompt_embeds"].shape[1], dev)
        guidance = torch.full((B,), float(self.guidance), device=dev)
        prior = None
        if p.get("pres_embeds") is not None:  # adapter inactive (outside `with net`): base-model prediction, no graph kept
            prior = model.forward_native(noisy, p["pres_embeds"], p["pres_pooled"], timesteps / 1000, img_ids, txt_ids, guidance,
                                         save_for_backward=False)
        with net:
            pred = model.forward_native(noisy, p["prompt_embeds"], p["pooled_embeds"], timesteps / 1000, img_ids, txt_ids, guidance)
            loss = self._loss_backward(pred, target, p.get("loss_weight"), p.get("loss_mask"), final=final and prior is None)
            if prior is not None:
                loss = loss.clone()
                pres = model.forward_native(noisy, p["pres_embeds"], p["pres_pooled"], timesteps / 1000, img_ids, txt_ids, guidance)
                loss.add_(self._loss_backward(pres, prior, p["pres_weight"], None, final=final))
        return loss


class WanLoRATrainStep(_LoRATrainStepBase):
    """Wan2.1 T2V LoRA step (BASELINE config 4).  Per step the reference runs the same trainer with the Wan model class:
    flow-matching scheduler with static shift 3.0 (toolkit/models/wan21/wan21.py:80-84, 338-342), noisy latents
    (1-t)x0 + t*noise, `self.model(hidden_states, timestep 0..1000, encoder_hidden_states)` (wan21.py:578-603), target
    noise - latents (wan21.py:717-724), MSE, backward, clip, AdamW."""

    def __init__(self, model, network, ops, **kw):
        kw.setdefault("schedule", FlowMatchTrainSchedule(shift=3.0, use_dynamic_shifting=False))
        super().__init__(model, network, ops, **kw)

    def step(self, latents, prompt_embeds, **kw):
        """latents [B,16,F,H,W] (normalised Wan-VAE latents), prompt_embeds [B,512,4096] (UMT5).  Returns the loss tensor."""
        return self.step_list([dict(latents=latents, prompt_embeds=prompt_embeds, **kw)])

    def _prepare(self, latents, prompt_embeds, *, noise=None, timesteps=None, loss_weight=None, loss_mask=None):
        dt = self.model.dt
        B, Cc, Fr, Hh, W = latents.shape
        dev = latents.device
        self.schedule.set_train_timesteps(1000, dev, self._table_type(), latents=latents, patch_size=2)
        if timesteps is None:
            timesteps, _ = self.schedule.sample_timesteps(B, dev, generator=self.gen)
        loss_weight = self._timestep_loss_weight(timesteps, loss_weight)
        if noise is None:  # from the unscaled latents (BaseSDTrainProcess.py:1323-1401)
            noise = get_noise(latents, self.gen, dtype=dt, **self.noise_options)
        latents = self._scale_latents(latents)
        if loss_mask is not None:
            if loss_mask.dim() == 4:  # [B,1,H,W] -> repeated over frames (SDTrainer.py:955-958)
                loss_mask = loss_mask[:, :, None].expand(-1, -1, Fr, -1, -1)
            loss_mask = self.pack_mask(loss_mask)
        return dict(latents=latents, noise=noise, timesteps=timesteps.float().contiguous(), prompt_embeds=prompt_embeds,
                    loss_weight=loss_weight, loss_mask=loss_mask)

    def _run(self, p, final=True):
        ops, model, net = self.ops, self.model, self.network
        latents, timesteps = p["latents"], p["timesteps"]
        dt = model.dt
        B, Cc, Fr, Hh, W = latents.shape
        dev = latents.device
        # frame-major copies ([B*F, C, H, W]) so the 2x2 pack kernel emits tokens in (frame, row, col) order
        lat_f = latents.to(dt).permute(0, 2, 1, 3, 4).reshape(B * Fr, Cc, Hh, W).contiguous()
        noi_f = p["noise"].to(dt).permute(0, 2, 1, 3, 4).reshape(B * Fr, Cc, Hh, W).contiguous()
        n_tok = (Hh // 2) * (W // 2)
        noisy = torch.empty(B * Fr, n_tok, Cc * 4, dtype=dt, device=dev)
        target = torch.empty_like(noisy)
        ops.flow_noise_pack(lat_f, noi_f, timesteps.repeat_interleave(Fr).contiguous(), noisy, target)
        self._scale_noisy(noisy)
        grid = (Fr, Hh // 2, W // 2)
        with net:
            pred = model.forward_native(noisy.view(B, Fr * n_tok, Cc * 4), timesteps, p["prompt_embeds"], grid)
            return self._loss_backward(pred, target.view(B, Fr * n_tok, Cc * 4), p.get("loss_weight"), p.get("loss_mask"), final=final)


class UNetLoRATrainStep(_LoRATrainStepBase):
    """SD1.5 / SDXL LoRA step (BASELINE configs 1-2).  Per step the reference runs: DDPM timesteps + add_noise
    (jobs/process/BaseSDTrainProcess.py:1301-1323, toolkit/stable_diffusion_model.py:1854-1876), the UNet on (noisy latents, timestep,
    CLIP states[, pooled text + time_ids for SDXL: 1824-1852, 1985-1990]) (2049-2055 / 2260-2265), target = eps (SDTrainer.py:650) or the
    velocity (623-625), MSE with optional min-SNR weighting (1005-1011), backward, clip, AdamW."""

    def __init__(self, model, network, ops, *, min_snr_gamma=None, snr_gamma=None, **kw):
        from .ddpm import DDPMTrainSchedule

        sched = kw.pop("schedule", None) or DDPMTrainSchedule()
        super().__init__(model, network, ops, schedule=sched, **kw)
        self.min_snr_gamma, self.snr_gamma = min_snr_gamma, snr_gamma
        self.is_xl = model.config["addition_embed_type"] == "text_time"

    def step(self, latents, prompt_embeds, pooled_embeds=None, **kw):
        """latents [B,4,h,w] (scaled VAE latents), prompt_embeds [B,77,768|2048], pooled_embeds [B,1280] (SDXL)."""
        return self.step_list([dict(latents=latents, prompt_embeds=prompt_embeds, pooled_embeds=pooled_embeds, **kw)])

    def _prepare(self, latents, prompt_embeds, pooled_embeds=None, *, noise=None, timesteps=None, loss_weight=None, time_ids=None):
        dt = self.model.dt
        B, Cc, Hh, W = latents.shape
        dev = latents.device
        latents = latents.to(dt)
        if timesteps is None:
            timesteps, _ = self.schedule.sample_timesteps(B, dev, generator=self.gen)
        timesteps = timesteps.to(dev).long()
        if noise is None:  # from the unscaled latents (BaseSDTrainProcess.py:1323-1401)
            noise = get_noise(latents, self.gen, dtype=dt, **self.noise_options)
        latents = self._scale_latents(latents).contiguous()
        a, s = self.schedule.noise_coefficients(timesteps, dt)
        p = dict(latents=latents, noise=noise.to(dt).contiguous(), t_float=timesteps.float(), alpha=a, sigma=s,
                 prompt_embeds=prompt_embeds)
        if self.is_xl:
            if time_ids is None:  # StableDiffusion.get_time_ids_from_latents: (H, W, 0, 0, H, W) in pixels
                time_ids = torch.tensor([[Hh * 8, W * 8, 0, 0, Hh * 8, W * 8]], dtype=dt, device=dev).repeat(B, 1)
            p.update(pooled_embeds=pooled_embeds, time_ids=time_ids)
        w = loss_weight
        gamma = self.min_snr_gamma if (self.min_snr_gamma is not None and self.min_snr_gamma > 1e-6) else None
        fixed = self.snr_gamma is not None and self.snr_gamma > 1e-6
        if fixed or gamma is not None:  # SDTrainer.py:1003-1011: snr_gamma (fixed) takes precedence over min_snr_gamma
            sw = self.schedule.snr_weights(timesteps, self.snr_gamma if fixed else gamma, fixed=fixed)
            w = sw if w is None else w * sw
        p["loss_weight"] = w
        return p

    def _run(self, p, final=True):
        ops, model, net = self.ops, self.model, self.network
        latents = p["latents"]
        dt = model.dt
        B, Cc, Hh, W = latents.shape
        dev = latents.device
        noisy = torch.empty(B * Hh * W, 8, dtype=dt, device=dev)
        target = torch.empty(B * Hh * W, Cc, dtype=dt, device=dev)
        ops.ddpm_noise_nhwc(latents, p["noise"], p["alpha"], p["sigma"], noisy, target,
                            v_prediction=self.schedule.prediction_type == "v_prediction")
        self._scale_noisy(noisy)
        added = dict(text_embeds=p["pooled_embeds"], time_ids=p["time_ids"]) if self.is_xl else None
        with net:
            pred = model.forward_native(noisy, p["t_float"], p["prompt_embeds"], added, B=B, H=Hh, W=W)
            return self._loss_backward(pred.view(B, Hh * W, Cc), target.view(B, Hh * W, Cc), p.get("loss_weight"), None, final=final)


_IDS_CACHE = {}


def make_ids(Hh, W, n_txt, device):
    """img_ids (0,row,col) for the 2x2-packed grid, txt_ids zeros (toolkit/stable_diffusion_model.py:2165-2170); cached per bucket
    (the host-to-device copy must not happen under graph capture)."""
    key = (Hh, W, n_txt, str(device))
    if key not in _IDS_CACHE:
        _IDS_CACHE[key] = _make_ids(Hh, W, n_txt, device)
    return _IDS_CACHE[key]


def _make_ids(Hh, W, n_txt, device):
    h2, w2 = Hh // 2, W // 2
    img_ids = torch.zeros(h2, w2, 3)
    img_ids[..., 1] = img_ids[..., 1] + torch.arange(h2)[:, None]
    img_ids[..., 2] = img_ids[..., 2] + torch.arange(w2)[None, :]
    img_ids = img_ids.reshape(h2 * w2, 3).to(device)
    img_ids._aitk_grid = (h2, w2, n_txt)  # RoPE-table cache key of the fused graph (no device sync per step)
    return img_ids, torch.zeros(n_txt, 3, device=device)
