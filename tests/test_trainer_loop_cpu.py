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
from ai_toolkit_amd.plugin import Flux1MI355Model
from oracle import flux_ref, ref_ops

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trainer_loop_flux_tiny.safetensors")
CFG = dict(in_channels=64, num_layers=2, num_single_layers=2, attention_head_dim=128, num_attention_heads=2,
           joint_attention_dim=64, pooled_projection_dim=32)


def test_the_run_was_the_references_own_trainer_network_and_scheduler():
    with safe_open(GOLD, "pt") as fh:
        meta = json.loads(fh.metadata()["meta"])
    assert meta["trainer"] == "SDTrainer" and meta["network"] == "LoRASpecialNetwork" and meta["scheduler"] == "CustomFlowMatchEulerDiscreteScheduler"
    assert meta["model_mro"][:2] == ["Flux1MI355", "Flux1MI355Model"]  # the real BaseModel subclass of the extension, hooks from the mirror
    assert meta["steps"] == 5 and meta["resume_at"] == 3 and meta["n_predict_calls"] == 5  # three steps, then a second process resumed for two more
    assert meta["opt_group"] == {"betas": [0.9, 0.999], "eps": 1e-06, "lr": 0.001, "weight_decay": 0.01}  # toolkit/optimizer.py:78-79 defaults
    assert {"aitk_trainer_run.safetensors", "optimizer.pt", "aitk_trainer_run_000000002.safetensors", "aitk_trainer_run_000000004.safetensors"} <= set(meta["files"])
    assert meta["kw"] == {"guidance_embedding_scale": 1.0, "bypass_guidance_embedding": False}  # what BaseModel.predict_noise passed down
    assert all(k.startswith("transformer.") and (k.endswith("lora_A.weight") or k.endswith("lora_B.weight")) for k in meta["saved_keys"])


def test_fused_twin_replaying_the_trainers_calls_reproduces_its_saved_lora_optimizer_state_and_ema():
    g = load_file(GOLD)
    with safe_open(GOLD, "pt") as fh:
        meta = json.loads(fh.metadata()["meta"])
    torch.manual_seed(0)
    ref = flux_ref.FluxTransformer2DModel(**CFG)
    flux_ref.init_synthetic_(ref, seed=1234, std=0.05)
    nat = FluxTransformer2DModel(**CFG, dtype=torch.float32, device="cpu", ops=ref_ops)
    nat.load_state_dict(ref.state_dict(), strict=True)
    nat.prepare()
    sd = Flux1MI355Model("cpu", model=nat, dtype=torch.float32)
    net = FusedLoRANetwork(nat, lora_dim=8, alpha=8, transformer_block_names=sd.get_transformer_block_names(), base_model=sd)
    init = {k[len("init/"):]: v for k, v in g.items() if k.startswith("init/")}
    with torch.no_grad():  # the adapter as the trainer's RNG stream initialised it
        for m in net.unet_loras:
            m.lora_down.weight.copy_(init[f"{m.lora_name}.lora_down.weight"])
            m.lora_up.weight.copy_(init[f"{m.lora_name}.lora_up.weight"])
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
            ema = [p_.detach().clone() for p_ in plist]
            net.refresh_shadows(ref_ops)
        pe = SimpleNamespace(text_embeds=g[f"step{i}/text"], pooled_embeds=g[f"step{i}/pooled"])
        opt.zero_grad()
        with net:
            pred = sd.get_noise_prediction(g[f"step{i}/latent_model_input"], g[f"step{i}/timestep"], pe, **meta["kw"])
            # SDTrainer.calculate_loss default branch (SDTrainer.py:903-1013): mse(reduction none) -> mean(1,2,3) -> * loss_multiplier (1) -> mean
            loss = torch.nn.functional.mse_loss(pred.float(), g[f"step{i}/target"].float(), reduction="none").mean([1, 2, 3]).mean()
            loss.backward()
        torch.nn.utils.clip_grad_norm_(plist, meta["max_grad_norm"])
        opt.step()
        opt.zero_grad(set_to_none=True)
        with torch.no_grad():
            for s, p in zip(ema, plist):
                tmp = s - p
                tmp.mul_(1.0 - meta["ema_decay"])
                s.sub_(tmp)
        net.refresh_shadows(ref_ops)
        losses.append(loss.item())
    assert torch.allclose(torch.tensor(losses, dtype=torch.float64), g["losses"], rtol=1e-6, atol=0)
    saved = net.get_state_dict(dtype=torch.float32)
    assert sorted(saved) == sorted(meta["saved_keys"])
    # with train.ema_config.use_ema the trainer's save() writes the EMA weights (BaseSDTrainProcess.save: ema.store / copy_to around the save):
    # the file must equal the twin's EMA shadows, parameter by parameter
    by_param = {id(p): e for p, e in zip(plist, ema)}
    for m in net.unet_loras:
        base = m.lora_name.replace("$$", ".")
        assert torch.equal(by_param[id(m.lora_down.weight)], g[f"saved/{base}.lora_A.weight"]), base
        assert torch.equal(by_param[id(m.lora_up.weight)], g[f"saved/{base}.lora_B.weight"]), base
        assert not torch.equal(m.lora_up.weight.detach(), g[f"saved/{base}.lora_B.weight"])  # ... and not the live weights
    for i, p in enumerate(plist):
        assert torch.equal(opt.state[p]["exp_avg"], g[f"opt/{i}/exp_avg"]) and torch.equal(opt.state[p]["exp_avg_sq"], g[f"opt/{i}/exp_avg_sq"]), i
        assert torch.equal(ema[i], g[f"ema/{i}"]), i


import subprocess  # noqa: E402
import sys  # noqa: E402

import pytest  # noqa: E402


@pytest.mark.skipif(not os.path.isdir("/root/reference/toolkit"), reason="the reference tree is not mounted here")
def test_committed_trainer_loop_fixture_is_what_the_references_trainer_produces_today(tmp_path):
    """run the reference's SDTrainer over the plug-in again (separate process: the import shims and accelerate's state stay out of this one)
    and compare with the committed fixture"""
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.argv=['make_golden.py']; sys.path.insert(0, %r); import runpy; "
            "g = runpy.run_path(%r, run_name='not_main'); g['golden_trainer_loop'](%r)"
            % (os.path.join(here, "golden"), os.path.join(here, "golden", "make_golden.py"), str(tmp_path)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    new, old = load_file(str(tmp_path / "trainer_loop_flux_tiny.safetensors")), load_file(GOLD)
    assert set(new) == set(old)
    for k in old:
        assert torch.equal(new[k], old[k]), k
