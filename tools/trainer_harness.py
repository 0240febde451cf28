"""Measurement / test harness: what the reference's trainer does AROUND the model plug-in in one train-loop iteration, written out once so
that `bench.py` (leg `trainer_path`), `tools/gpu_trainer_path.py` (rocprof target) and `tests/test_gpu_trainer_path.py` can walk the boundary
the reference really uses on a box that has no reference tree (the GPU box):

    SDTrainer.hook_train_loop (extensions_built_in/sd_trainer/SDTrainer.py:2243-2318)
      optimizer.zero_grad()                                                  2250
      train_single_accumulation (1523-2240):
        noise / timesteps / add_noise of process_general_training_batch     jobs/process/BaseSDTrainProcess.py:1301-1478 (torch, latent-sized)
        with self.network: noise_pred = sd.predict_noise(...) -> sd.get_noise_prediction(...)     2152-2160
        calculate_loss: mse(pred.float(), target.float(), 'none').mean([1,2,3]).mean()            903-1050
        `if not torch.isfinite(loss)` (a host sync in the middle of the step)                     2221-2224
        self.accelerator.backward(loss)                                                             2238
      self.accelerator.clip_grad_norm_(self.params, max_grad_norm)           2278-2283 (= torch.nn.utils.clip_grad_norm_)
      self.optimizer.step(); self.optimizer.zero_grad(set_to_none=True)      2285-2288
      self.ema.update()                                                      2291-2293 (toolkit/ema.py:104-152)
      self.lr_scheduler.step(); loss_dict = {'loss': loss.item()}            2299, 2311-2313

The trainer-side OBJECTS are stand-ins with the reference's protocol and nothing else: `LoRAModule` (toolkit/lora_special.py:46-135: lora_down /
lora_up nn.Linear in fp32, alpha, scale, apply_to = forward swap), `TrainerLoRANetwork` (the slice of LoRASpecialNetwork / ToolkitNetworkMixin the
loop touches: module discovery, force_to, prepare_optimizer_params, `with network:`, multiplier), `ExponentialMovingAverage` (toolkit/ema.py's
attributes and update loop).  None of them computes anything for the model: the native graph never calls `Linear.forward`, and `LoRAModule.forward`
here RAISES if anything does.  All arithmetic of the measured step is the HIP library plus the torch ops listed above — the same split as under
the reference's real trainer (run on the CPU kernel table by tests/golden/make_golden.py golden_trainer_loop).
"""
import math
import weakref
from types import SimpleNamespace

import torch
import torch.nn as nn


class LoRAModule(nn.Module):
    """Holder with the attributes of toolkit/lora_special.py:46-135 (Linear case)."""

    def __init__(self, lora_name, org_module, lora_dim, alpha, network):
        super().__init__()
        self.lora_name = lora_name
        self.lora_dim = lora_dim
        self.lora_down = nn.Linear(org_module.in_features, lora_dim, bias=False)
        self.lora_up = nn.Linear(lora_dim, org_module.out_features, bias=False)
        alpha = lora_dim if alpha is None or alpha == 0 else alpha
        self.scale = float(alpha) / lora_dim
        self.register_buffer("alpha", torch.tensor(float(alpha)))
        nn.init.kaiming_uniform_(self.lora_down.weight, a=math.sqrt(5))
        nn.init.zeros_(self.lora_up.weight)
        self.multiplier = 1.0
        self.org_module = [org_module]
        self.network_ref = weakref.ref(network)
        self.dropout = self.rank_dropout = self.module_dropout = None

    def apply_to(self):
        self.org_forward = self.org_module[0].forward
        self.org_module[0].forward = self.forward

    def forward(self, x, *args, **kwargs):
        raise RuntimeError(f"{self.lora_name}: Linear.forward was called — the native graph must run this adapter inside its own kernels")


class TrainerLoRANetwork(nn.Module):
    """Discovery + naming of toolkit/lora_special.py:457-647 for a PEFT-format transformer network (Linear children of `target` modules whose
    dotted name passes the transformer-block filter), and the methods the train loop calls on the network."""
    is_merged_in = False
    is_lorm = False
    text_encoder_loras = ()
    peft_format = True
    is_transformer = True

    def __init__(self, unet, lora_dim, multiplier=1.0, alpha=None, target=("FluxTransformer2DModel",),
                 block_names=("transformer_blocks", "single_transformer_blocks")):
        super().__init__()
        self.lora_dim = lora_dim
        self.alpha = lora_dim if alpha is None else alpha
        self.is_active = False
        self._multiplier = float(multiplier)
        self.torch_multiplier = torch.tensor([float(multiplier)])
        self.unet_loras = []
        for name, module in unet.named_modules():
            if module.__class__.__name__ not in target:
                continue
            for child_name, child in module.named_modules():
                if child.__class__.__name__ != "Linear":
                    continue
                clean = ".".join(x for x in ("transformer", name, child_name) if x)
                if not any(b in clean for b in block_names):
                    continue
                self.unet_loras.append(LoRAModule(clean.replace(".", "$$"), child, lora_dim, self.alpha, self))
        for lo in self.unet_loras:
            self.add_module(lo.lora_name, lo)

    def get_all_modules(self):
        return list(self.unet_loras)

    def force_to(self, device, dtype):  # network_mixins.py:855-866
        self.to(device, dtype)
        self.torch_multiplier = self.torch_multiplier.to(device, torch.float32)

    def apply_to(self, text_encoder=None, unet=None, apply_text_encoder=True, apply_unet=True):  # kohya_lora.py:952-965
        for lo in self.unet_loras:
            lo.apply_to()

    def prepare_grad_etc(self, *a, **k):
        self.requires_grad_(True)

    def prepare_optimizer_params(self, text_encoder_lr=None, unet_lr=None, default_lr=None):  # kohya_lora.py:1030-1074: one group for the unet loras
        params = [p for lo in self.unet_loras for p in lo.parameters()]
        group = {"params": params}
        if unet_lr is not None:
            group["lr"] = unet_lr
        return [group]

    @property
    def multiplier(self):
        return self._multiplier

    @multiplier.setter
    def multiplier(self, value):
        self._multiplier = value
        vals = [float(v) for v in value] if isinstance(value, (list, tuple)) else [float(value)]
        self.torch_multiplier = torch.tensor(vals, dtype=torch.float32, device=self.torch_multiplier.device)

    def __enter__(self):  # network_mixins.py:849-853
        self.is_active = True

    def __exit__(self, *a):
        self.is_active = False


class ExponentialMovingAverage:
    """toolkit/ema.py:16-152: attributes and the update loop (fp32 parameters: the `.float()` calls of the loop are no-ops and are left out)."""

    def __init__(self, parameters, decay=0.995, use_num_updates=False, use_feedback=False, param_multiplier=1.0):
        parameters = list(parameters)
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.use_feedback = use_feedback
        self.param_multiplier = param_multiplier
        self.shadow_params = [p.clone().detach() for p in parameters]
        self._params_refs = [weakref.ref(p) for p in parameters]

    def _get_parameters(self, parameters):
        return [p() for p in self._params_refs] if parameters is None else list(parameters)

    def update(self, parameters=None):
        parameters = self._get_parameters(parameters)
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))
        one_minus_decay = 1.0 - decay
        with torch.no_grad():
            for s_param, param in zip(self.shadow_params, parameters):
                tmp = s_param - param
                tmp.mul_(one_minus_decay)
                s_param.sub_(tmp)
                if self.use_feedback:
                    param.add_(tmp * 10)
                if self.param_multiplier != 1.0:
                    param.mul_(self.param_multiplier)


class TrainerLoop:
    """The trainer's set-up sequence (jobs/process/BaseSDTrainProcess.py:1949-2039: network over sd.get_model_to_train(), force_to(device,
    fp32), `sd.network = network`, apply_to, prepare_optimizer_params -> toolkit/optimizer.py:78-79 torch.optim.AdamW(eps=1e-6), EMA over the
    same parameters: 798-803) and its per-step sequence (module docstring) over a model plug-in `sd`."""

    def __init__(self, sd, *, rank=16, lr=1e-4, weight_decay=0.01, max_grad_norm=1.0, ema_decay=0.99, device="cuda", seed=0, fuse_ema=True):
        self.sd = sd
        torch.manual_seed(seed)
        net = TrainerLoRANetwork(sd.get_model_to_train(), rank, 1.0)
        net.force_to(torch.device(device), torch.float32)
        sd.network = net
        net.apply_to(None, sd.unet, False, True)
        net.prepare_grad_etc(None, sd.unet)
        groups = net.prepare_optimizer_params(None, lr, lr)
        self.params = [p for g in groups for p in g["params"]]
        self.optimizer = torch.optim.AdamW(groups, lr=lr, eps=1e-6, weight_decay=weight_decay)
        self.ema = ExponentialMovingAverage(self.params, decay=ema_decay) if ema_decay else None
        if self.ema is not None and fuse_ema:
            from ai_toolkit_amd.adopt import install_ema_fusion

            install_ema_fusion(ExponentialMovingAverage)  # what integration/extensions/aitk_mi355 does with toolkit.ema's class
        self.network = net
        self.max_grad_norm = max_grad_norm
        self.gen = torch.Generator(device=device).manual_seed(1000 + seed)

    def process_batch(self, latents):
        """noise, timesteps (uniform indices into the 1000-entry linear table), flow-matching add_noise, loss target — torch ops on
        latent-sized tensors like the reference's process_general_training_batch + toolkit/samplers/custom_flowmatch_sampler.py:91-102"""
        B = latents.shape[0]
        noise = torch.randn(latents.shape, device=latents.device, dtype=torch.float32, generator=self.gen).to(latents.dtype)
        idx = torch.randint(0, 1000, (B,), device=latents.device, generator=self.gen)
        timesteps = (1000.0 - idx.float()).clamp(1.0, 1000.0)
        t01 = (timesteps / 1000.0).view(-1, 1, 1, 1).to(latents.dtype)
        noisy = (1.0 - t01) * latents + t01 * noise
        return noisy, timesteps, (noise - latents).detach()

    def hook_train_loop(self, latents, prompt_embeds, pooled_embeds):
        opt = self.optimizer
        opt.zero_grad()
        noisy, timesteps, target = self.process_batch(latents)
        pe = SimpleNamespace(text_embeds=prompt_embeds, pooled_embeds=pooled_embeds)
        with self.network:
            pred = self.sd.get_noise_prediction(noisy, timesteps, pe, guidance_embedding_scale=1.0)
            loss = torch.nn.functional.mse_loss(pred.float(), target.float(), reduction="none").mean([1, 2, 3]).mean()
            if not torch.isfinite(loss):  # the reference's mid-step host sync
                loss = torch.zeros_like(loss).requires_grad_(True)
            loss.backward()
        torch.nn.utils.clip_grad_norm_(self.params, self.max_grad_norm)
        opt.step()
        opt.zero_grad(set_to_none=True)
        if self.ema is not None:
            self.ema.update()
        return loss.detach().item()
