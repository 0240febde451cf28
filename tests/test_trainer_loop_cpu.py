"""The reference's REAL trainer over the plug-in, end to end (tests/golden/make_golden.py golden_trainer_loop): `SDTrainer.run()` of
extensions_built_in/sd_trainer — unmodified — trained a LoRA for three steps over `integration/extensions/aitk_mi355` on the CPU (native FLUX
graph on the oracle kernel table; the reference's own LoRASpecialNetwork adopted; its own flow-match scheduler, optimizer factory, accelerate,
EMA, text-encoder unloader and checkpoint writer; a synthetic dataloader and tiny `transformers` text encoders as the only stand-ins).  The fixture
holds what every `get_noise_prediction` call received, the loss target the trainer formed, the losses it logged, and the LoRA file / optimizer
state / EMA it ended with.  Here a FusedLoRANetwork twin replays the recorded calls with torch's AdamW and must land on the SAME saved LoRA,
AdamW moments and EMA, bit for bit — i.e. everything between the recorded inputs and the saved file went through the native graph exactly as the
fused network does it."""
import json
import os
from types import SimpleNamespace

import torch
from safetensors import safe_open
from safetensors.torch import load_file

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd.flux import FluxTransformer2DModel
from ai_toolkit_amd.lora import FusedLoRANetwork
from ai_toolkit_amd.plugin import Flux1MI355Model, StableDiffusionMI355Model, Wan21MI355Model
from ai_toolkit_amd.unet import UNet2DConditionModel
from ai_toolkit_amd.wan import WanTransformer3DModel
from oracle import flux_ref, ref_ops, unet_ref, wan_ref
from tests.test_unet_cpu import TINY_SD15, TINY_SDXL

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CFG = dict(in_channels=64, num_layers=2, num_single_layers=2, attention_head_dim=128, num_attention_heads=2,
           joint_attention_dim=64, pooled_projection_dim=32)
WAN_CFG = dict(num_attention_heads=2, attention_head_dim=128, in_channels=16, out_channels=16, text_dim=48, freq_dim=256, ffn_dim=320, num_layers=3)
# kind -> (fixture, native class, mirror class, oracle module, its config, synthetic-weight seed, what BaseModel.predict_noise passed down, key prefix)
KINDS = {
    "flux": (Flux1MI355Model, FluxTransformer2DModel, flux_ref.FluxTransformer2DModel, flux_ref.init_synthetic_, CFG, 1234,
             {"guidance_embedding_scale": 1.0, "bypass_guidance_embedding": False}, "transformer."),
    # Wan2.1 (BASELINE config 4): video latents [B, 16, F, H, W], UMT5 text states, no pooled vector; the LoRA file is written with the
    # `diffusion_model.` prefix of the reference's Wan convert_lora_weights_before_save (extensions_built_in/diffusion_models/wan22/…, toolkit/models/wan21)
    "wan": (Wan21MI355Model, WanTransformer3DModel, wan_ref.WanTransformer3DModel, wan_ref.init_synthetic_, WAN_CFG, 99, {}, "diffusion_model."),
    # SD1.5 UNet (BASELINE config 1): eps-prediction over the DDPM schedule; the trainer calls the legacy-StableDiffusion `predict_noise` surface with
    # its guidance arguments (all neutral in training); the LoRA file is in the kohya layout (lora_unet_* names, lora_down / lora_up / alpha)
    "sd15": (StableDiffusionMI355Model, UNet2DConditionModel, unet_ref.UNet2DConditionModel, unet_ref.init_synthetic_, TINY_SD15, 5,
             {"bypass_guidance_embedding": False, "detach_unconditional": False, "guidance_embedding_scale": 1.0, "guidance_scale": 1.0, "rescale_cfg": 1.0},
             "lora_unet_"),
    # SDXL UNet (BASELINE config 2): + pooled text embedding and the (H, W, 0, 0, H, W) time ids the mirror derives from the latents
    "sdxl": (StableDiffusionMI355Model, UNet2DConditionModel, unet_ref.UNet2DConditionModel, unet_ref.init_synthetic_, TINY_SDXL, 5,
             {"bypass_guidance_embedding": False, "detach_unconditional": False, "guidance_embedding_scale": 1.0, "guidance_scale": 1.0, "rescale_cfg": 1.0},
             "lora_unet_"),
}
KINDS["flux_accum2"] = KINDS["flux"]  # train.gradient_accumulation: 2 (two micro-batches per optimizer step; zero_grad(set_to_none) drops the grad views)
KINDS["flux_bf16"] = KINDS["flux"]    # train.dtype: bf16 (the reference's default): bf16 base + activations, fp32 network
KINDS["flux_uncached"] = KINDS["flux"]      # images instead of cached latents: the trainer calls the plug-in's encode_images (native VAE encoder)
KINDS["flux_lora_partial"] = KINDS["flux"]  # network_kwargs.only_if_contains: same-input groups with members missing
KINDS["flux_dora"] = KINDS["flux"]          # network.type: dora (light fixture: losses + saved file)
KINDS["flux_lokr_lowrank"] = KINDS["flux"]  # network.type: lokr, lokr_full_rank: false
KINDS["flux_bf16_fp8base"] = KINDS["flux"]  # model.quantize: true — e4m3 weight-only base under the adopted network (BASELINE config 5's base)
# train.blank_prompt_preservation: per step a prior prediction (network off, no_grad), the training prediction and a second grad-enabled prediction with the blank
# embeddings; ONE loss.backward() through two native graphs (SDTrainer.py:1983-2016, 2182-2219)
KINDS["flux_blankpp"] = KINDS["flux"]
SCHEDULER = {"flux_blankpp": "CustomFlowMatchEulerDiscreteScheduler", "flux_lora_partial": "CustomFlowMatchEulerDiscreteScheduler", "flux_uncached": "CustomFlowMatchEulerDiscreteScheduler", "flux_dora": "CustomFlowMatchEulerDiscreteScheduler", "flux_lokr_lowrank": "CustomFlowMatchEulerDiscreteScheduler",
             "flux_bf16_fp8base": "CustomFlowMatchEulerDiscreteScheduler", "flux_accum2": "CustomFlowMatchEulerDiscreteScheduler", "flux_bf16": "CustomFlowMatchEulerDiscreteScheduler", "flux": "CustomFlowMatchEulerDiscreteScheduler", "wan": "CustomFlowMatchEulerDiscreteScheduler",
             # diffusers is not installed where the fixture is generated: toolkit/sampler.py returns an import stub there and the plug-in falls back to
             # its native DDPM table (a working DDPMScheduler is kept: tests/test_plugin_cpu.py)
             "sd15": "DDPMTrainSchedule", "sdxl": "DDPMTrainSchedule"}


def gold(kind):
    return os.path.join(HERE, "golden", f"trainer_loop_{kind}_tiny.safetensors")


@pytest.mark.parametrize("kind", list(KINDS))
def test_the_run_was_the_references_own_trainer_network_and_scheduler(kind):
    Mirror, _, _, _, _, _, kw, prefix = KINDS[kind]
    with safe_open(gold(kind), "pt") as fh:
        meta = json.loads(fh.metadata()["meta"])
    assert meta["trainer"] == "SDTrainer" and meta["network"] == "LoRASpecialNetwork" and meta["scheduler"] == SCHEDULER[kind]
    assert meta["model_mro"][1] == Mirror.__name__ and meta["model_mro"][0] == Mirror.__name__[:-len("Model")]  # the real BaseModel subclass of the extension, hooks from the mirror
    accum = meta.get("accum", 1)
    calls_per_micro = 3 if meta.get("preservation") else 1  # preservation runs: prior + training + preservation prediction
    assert meta["steps"] == 5 and meta["resume_at"] == 3 and meta["n_predict_calls"] == 5 * accum * calls_per_micro  # three steps, then a second process resumed for two more
    assert meta["opt_group"] == {"betas": [0.9, 0.999], "eps": 1e-06, "lr": 0.001, "weight_decay": 0.01}  # toolkit/optimizer.py:78-79 defaults
    assert {"aitk_trainer_run.safetensors", "optimizer.pt", "aitk_trainer_run_000000002.safetensors", "aitk_trainer_run_000000004.safetensors"} <= set(meta["files"])
    assert meta["kw"] == kw
    tails = ("lora_down.weight", "lora_up.weight", "alpha") if kind in ("sd15", "sdxl") else ("lora_A.weight", "lora_B.weight")
    tails = {"flux_dora": tails + ("magnitude",), "flux_lokr_lowrank": ("lokr_w1", "lokr_w2_a", "lokr_w2_b", "alpha")}.get(kind, tails)
    assert all(k.startswith(prefix) and k.endswith(tails) for k in meta["saved_keys"])


@pytest.mark.parametrize("kind", list(KINDS))
def test_fused_twin_replaying_the_trainers_calls_reproduces_its_saved_lora_optimizer_state_and_ema(kind):
    Mirror, Native, Ref, init_, cfg, seed, _, prefix = KINDS[kind]
    g = load_file(gold(kind))
    with safe_open(gold(kind), "pt") as fh:
        meta = json.loads(fh.metadata()["meta"])
    accum = meta.get("accum", 1)
    torch.manual_seed(0)
    ref = Ref(**cfg)
    init_(ref, seed=seed, std=0.05)
    dt = {"fp32": torch.float32, "bf16": torch.bfloat16}[meta.get("dtype", "fp32")]
    nat = Native(**cfg, dtype=dt, device="cpu", ops=ref_ops)
    nat.load_state_dict({k: v.to(dt) for k, v in ref.state_dict().items()}, strict=True)
    nat.prepare()
    if meta.get("quantize"):
        assert meta["base_is_quantized"]
        nat.quantize_base_fp8(release_bf16=True)  # what load_model does for model.quantize (plugin.py)
    sd = Mirror("cpu", model=nat, dtype=dt, **({"is_xl": True} if kind == "sdxl" else {}))
    # what BaseSDTrainProcess passes per family (jobs/process/BaseSDTrainProcess.py:1937-1990)
    extra = {"flux": {}, "wan": dict(target_lin_modules=tuple(sd.target_lora_modules), base_model_version="wan_2.1"),
             "sd15": dict(target_lin_modules=tuple(sd.target_lora_modules), is_transformer=False, peft_format=False, transformer_only=False, base_model_version="sd1"),
             "sdxl": dict(target_lin_modules=tuple(sd.target_lora_modules), is_transformer=False, peft_format=False, transformer_only=False, base_model_version="sdxl")}[kind.split("_")[0]]
    nk = meta.get("network_kind", "lora")
    rank = 4 if nk == "lokr_lowrank" else 8
    if nk == "lora_partial":
        extra = dict(extra, only_if_contains=meta["only_if_contains"])
    elif nk != "lora":
        extra = dict(extra, network_type={"dora": "dora", "lokr_lowrank": "lokr"}[nk])
    net = FusedLoRANetwork(nat, lora_dim=rank, alpha=rank, transformer_block_names=sd.get_transformer_block_names(), base_model=sd, **extra)
    init = {k[len("init/"):]: v for k, v in g.items() if k.startswith("init/")}
    with torch.no_grad():  # the adapter as the trainer's RNG stream initialised it
        for m in net.unet_loras:
            for pname, p_ in m.named_parameters():  # lora_down / lora_up (+ magnitude), or the LoKr factors; 1x1-conv adapters: [r, in, 1, 1] in the reference
                p_.copy_(init[f"{m.lora_name}.{pname}"].reshape(p_.shape))
    net.apply_to()
    net.build_arena("cpu", groups=nat.lora_groups())
    net.refresh_shadows(ref_ops)
    nat.attach_network(net)
    plist = net.prepare_optimizer_params(default_lr=1e-3)[0]["params"]
    og = meta["opt_group"]
    opt = torch.optim.AdamW(plist, lr=og["lr"], betas=tuple(og["betas"]), eps=og["eps"], weight_decay=og["weight_decay"])
    ema = [p.detach().clone() for p in plist]
    losses = []
    for i in range(meta["steps"]):
        if i == meta["resume_at"]:
            # the second trainer process resumed here (BaseSDTrainProcess.py:2057-2066, 2190-2215): the network's own load_weights read the saved
            # file — which holds the EMA weights (fp32 save dtype in this run) — optimizer.pt restored the AdamW moments and step counts, and a
            # fresh EMA started from the loaded parameters
            with torch.no_grad():
                for p_, e_ in zip(plist, ema):
                    p_.copy_(e_)
                if nk == "dora":
                    # the reference's PEFT-format load path un-escapes only the lora_down / lora_up (and LoKr) keys (toolkit/network_mixins.py:702-718):
                    # `...to_k.magnitude` becomes `...to_k$$magnitude`, matches nothing and is dropped ("Missing keys") — a resumed DoRA run restarts
                    # from the INITIAL magnitudes (row norms of the base weight) with the restored AdamW moments.  The adopted network holds whatever
                    # the reference's objects hold, so the twin does the same.
                    for m in net.unet_loras:
                        m.magnitude.copy_(init[f"{m.lora_name}.magnitude"])
            ema = [p_.detach().clone() for p_ in plist]
            net.refresh_shadows(ref_ops)
        opt.zero_grad()
        step_loss = 0.0
        for a in range(accum):  # SDTrainer.hook_train_loop: one backward per micro-batch, gradients accumulate (SDTrainer.py:2246-2271)
            j = i * accum + a
            pe = SimpleNamespace(text_embeds=g[f"step{j}/text"], pooled_embeds=g[f"step{j}/pooled"] if g[f"step{j}/pooled"].numel() else None)
            with net:
                pres = meta.get("preservation")
                if pres:  # get_prior_prediction (SDTrainer.py:1211-1339): the network switched off, no_grad, the preservation embeddings
                    ppe = SimpleNamespace(text_embeds=g[f"step{j}/pres_text"], pooled_embeds=g[f"step{j}/pres_pooled"])
                    net.is_active = False
                    with torch.no_grad():
                        prior = sd.get_noise_prediction(g[f"step{j}/latent_model_input"], g[f"step{j}/timestep"], ppe, **meta["kw"])
                    net.is_active = True
                pred = sd.get_noise_prediction(g[f"step{j}/latent_model_input"], g[f"step{j}/timestep"], pe, **meta["kw"])
                # SDTrainer.calculate_loss default branch (SDTrainer.py:903-1013): mse(reduction none) -> mean over all but the batch axis (:987-990; 5-D for video) -> * loss_multiplier (1) -> mean
                loss = torch.nn.functional.mse_loss(pred.float(), g[f"step{j}/target"].float(), reduction="none")
                loss = loss.mean(list(range(1, loss.dim()))).mean()
                if pres:  # SDTrainer.py:2182-2219: a second grad-enabled prediction, held to the prior; one backward through both graphs
                    pres_pred = sd.get_noise_prediction(g[f"step{j}/latent_model_input"], g[f"step{j}/timestep"], ppe, **meta["kw"])
                    loss = loss + torch.nn.functional.mse_loss(pres_pred, prior) * pres["multiplier"]
                loss.backward()
            step_loss += loss.item()
            if nk == "lora_partial" and j == 0:
                first_grads = {m.lora_name: (m.lora_down.weight.grad.clone(), m.lora_up.weight.grad.clone()) for m in net.unet_loras}
        torch.nn.utils.clip_grad_norm_(plist, meta["max_grad_norm"])
        opt.step()
        opt.zero_grad(set_to_none=True)
        with torch.no_grad():
            for s, p in zip(ema, plist):
                tmp = s - p
                tmp.mul_(1.0 - meta["ema_decay"])
                s.sub_(tmp)
        net.refresh_shadows(ref_ops)
        losses.append(step_loss / accum)  # the trainer logs the mean over its micro-batches
    assert torch.allclose(torch.tensor(losses, dtype=torch.float64), g["losses"], rtol=1e-6, atol=0)
    saved = sd.convert_lora_weights_before_save(net.get_state_dict(dtype=torch.float32))
    assert sorted(saved) == sorted(meta["saved_keys"])
    # with train.ema_config.use_ema the trainer's save() writes the EMA weights (BaseSDTrainProcess.save: ema.store / copy_to around the save):
    # the file must equal the twin's EMA shadows, parameter by parameter
    live_sd = {k: v.clone() for k, v in sd.convert_lora_weights_before_save(net.get_state_dict(dtype=torch.float32)).items()}
    with torch.no_grad():  # ema.store / copy_to: the parameters hold the EMA while the file is written
        for p_, e_ in zip(plist, ema):
            p_.copy_(e_)
    ema_sd = sd.convert_lora_weights_before_save(net.get_state_dict(dtype=torch.float32))  # Wan: the file's own key names
    assert sorted(ema_sd) == sorted(meta["saved_keys"])
    for k, v in ema_sd.items():
        assert torch.equal(v.reshape(g[f"saved/{k}"].shape), g[f"saved/{k}"]), k
    ups = [k for k in live_sd if k.endswith(("lora_B.weight", "lora_up.weight", "lokr_w2_b"))]
    differ = sum(not torch.equal(live_sd[k].reshape(g[f"saved/{k}"].shape), g[f"saved/{k}"]) for k in ups)
    assert ups and differ >= 0.9 * len(ups), (differ, len(ups))  # ... and not the live weights (an adapter whose gradient is exactly 0 stays at its zero init in both)
    if nk == "lora_partial":
        # ... and the partially covered graph itself against autograd: the oracle network over the oracle model, the same seven adapters, call 0
        from oracle import lora_ref

        names = [m.lora_name for m in net.unet_loras]
        assert len(names) == 7 and any("to_q" in n for n in names) and not any("attn$$to_k" in n for n in names)
        ref_net = lora_ref.RefLoRANetwork(ref, 8)
        keep_mods = [m for m in ref_net.unet_loras if m.lora_name in names]
        for m in ref_net.unet_loras:
            if m.lora_name not in names:
                delattr(ref_net, m.lora_name)
        ref_net.unet_loras = keep_mods
        assert [m.lora_name for m in keep_mods] == names
        with torch.no_grad():
            for m in keep_mods:
                m.lora_down.weight.copy_(init[f"{m.lora_name}.lora_down.weight"])
                m.lora_up.weight.copy_(init[f"{m.lora_name}.lora_up.weight"])
        ref_net.apply_to()
        lat0, ts0 = g["step0/latent_model_input"], g["step0/timestep"]
        img_ids, txt_ids = flux_ref.make_ids(lat0.shape[2], lat0.shape[3], g["step0/text"].shape[1])
        with ref_net:
            p_ref = flux_ref.unpack_latents(ref(flux_ref.pack_latents(lat0), g["step0/text"], g["step0/pooled"], ts0 / 1000, img_ids, txt_ids,
                                                torch.full((lat0.shape[0],), 1.0)), lat0.shape[2], lat0.shape[3])
            l_ref = torch.nn.functional.mse_loss(p_ref.float(), g["step0/target"].float(), reduction="none").mean([1, 2, 3]).mean()
            l_ref.backward()
        assert abs(l_ref.item() - g["losses"][0].item()) <= 1e-5 * abs(l_ref.item()), (l_ref.item(), g["losses"][0].item())
        for m in keep_mods:  # lora_up starts at zero: its gradient is the informative one at step 0
            assert torch.allclose(first_grads[m.lora_name][1], m.lora_up.weight.grad, rtol=5e-4, atol=1e-7), m.lora_name
            assert torch.allclose(first_grads[m.lora_name][0], m.lora_down.weight.grad, rtol=5e-4, atol=1e-7), m.lora_name
    if meta.get("uncached"):
        # what the trainer got from `sd.encode_images(images)`: the native AutoencoderKL encoder over the pipeline directory's vae/ weights, sampled with the
        # generator state the trainer's process had at that call
        from ai_toolkit_amd.vae import AutoencoderKLEncoder

        vae = AutoencoderKLEncoder(latent_channels=16, block_out_channels=(32, 64), layers_per_block=1, dtype=torch.float32, device="cpu", ops=ref_ops)
        torch.manual_seed(11)
        with torch.no_grad():
            for p_ in vae.parameters():
                p_.copy_(torch.randn_like(p_) * 0.05)
        sdv = Mirror("cpu", model=nat, vae=vae, dtype=dt)
        keep_rng = torch.get_rng_state()
        torch.set_rng_state(g["encode/rng_state"])
        lat = sdv.encode_images(list(g["encode/images"]))
        torch.set_rng_state(keep_rng)
        assert tuple(lat.shape) == tuple(g["encode/latents"].shape) == (2, 16, 8, 4) and torch.equal(lat, g["encode/latents"])
    if nk != "lora" or meta.get("uncached"):
        return  # light fixtures (adapter-type variants): losses + the saved file (= the EMA weights) carry the comparison
    for i, p in enumerate(plist):  # (reshape: the reference's 1x1-conv adapters keep [.., 1, 1] axes)
        assert torch.equal(opt.state[p]["exp_avg"], g[f"opt/{i}/exp_avg"].reshape(p.shape)) and torch.equal(opt.state[p]["exp_avg_sq"], g[f"opt/{i}/exp_avg_sq"].reshape(p.shape)), i
        assert torch.equal(ema[i], g[f"ema/{i}"].reshape(p.shape)), i


import subprocess  # noqa: E402
import sys  # noqa: E402


RUN_KW = {"flux": dict(kind="flux"), "wan": dict(kind="wan"), "sd15": dict(kind="sd15"), "sdxl": dict(kind="sdxl"), "flux_accum2": dict(kind="flux", accum=2),
          "flux_bf16": dict(kind="flux", dtype="bf16"), "flux_bf16_fp8base": dict(kind="flux", dtype="bf16", quantize=True), "flux_dora": dict(kind="flux", network="dora"),
          "flux_lokr_lowrank": dict(kind="flux", network="lokr_lowrank"), "flux_uncached": dict(kind="flux", uncached=True),
          "flux_lora_partial": dict(kind="flux", network="lora_partial"), "flux_blankpp": dict(kind="flux", preservation="blank")}


@pytest.mark.skipif(not os.path.isdir("/root/reference/toolkit"), reason="the reference tree is not mounted here")
def test_committed_trainer_loop_fixtures_are_what_the_references_trainer_produces_today(tmp_path):
    """run the reference's SDTrainer over the plug-in again — every committed run, one after the other in ONE separate process (the import shims and
    accelerate's state stay out of this one) — and compare with the committed fixtures, tensor for tensor"""
    assert set(RUN_KW) == set(KINDS)
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys, io, contextlib; sys.argv=['make_golden.py']; sys.path.insert(0, %r); import runpy; "
            "g = runpy.run_path(%r, run_name='not_main')\n"
            "for kw in %r:\n"
            "    g['golden_trainer_loop'](%r, **kw)\n"
            % (os.path.join(here, "golden"), os.path.join(here, "golden", "make_golden.py"), list(RUN_KW.values()), str(tmp_path)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-3000:]
    for kind in KINDS:
        new, old = load_file(str(tmp_path / f"trainer_loop_{kind}_tiny.safetensors")), load_file(gold(kind))
        assert set(new) == set(old), kind
        for k in old:
            assert torch.equal(new[k], old[k]), (kind, k)
