"""Generates the golden fixtures in this directory by EXECUTING THE REFERENCE'S OWN CODE (read-only, under import
shims) in this container:  python tests/golden/make_golden.py

  lora_flux_tiny.safetensors   reference toolkit.lora_special.LoRASpecialNetwork attached to the oracle's tiny
                               FluxTransformer2DModel (class / attribute names = diffusers'): adapter names, the init
                               values it draws under torch.manual_seed(99), one forward output and every adapter
                               gradient for fixed inputs, per-sample multiplier case, and the PEFT-format state dict
                               it would save (keys + values).
  flowmatch.safetensors        reference toolkit/samplers/custom_flowmatch_sampler.py executed with a stub diffusers
                               base class: linear / sigmoid(seed) timestep tables and add_noise on a fixed batch.
The reference is never imported at test or bench time; only these vectors travel.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_shims  # noqa: E402

ref_shims.install()

import torch  # noqa: E402
from safetensors.torch import save_file  # noqa: E402

from oracle import flux_ref  # noqa: E402

TINY = dict(in_channels=64, num_layers=1, num_single_layers=1, attention_head_dim=128, num_attention_heads=2,
            joint_attention_dim=64, pooled_projection_dim=32)


def tiny_inputs():
    g = torch.Generator().manual_seed(3)
    Hl, Wl, n_txt, B = 4, 4, 5, 2
    hidden = torch.randn(B, (Hl // 2) * (Wl // 2), 64, generator=g)
    enc = torch.randn(B, n_txt, 64, generator=g)
    pooled = torch.randn(B, 32, generator=g)
    t = torch.tensor([0.3, 0.8])
    img_ids, txt_ids = flux_ref.make_ids(Hl, Wl, n_txt)
    return hidden, enc, pooled, t, img_ids, txt_ids, torch.ones(B)


def golden_lora():
    from toolkit.lora_special import LoRASpecialNetwork

    torch.manual_seed(0)
    model = flux_ref.FluxTransformer2DModel(**TINY)
    flux_ref.init_synthetic_(model, seed=1234, std=0.05)
    torch.manual_seed(99)
    net = LoRASpecialNetwork(text_encoder=None, unet=model, lora_dim=8, alpha=1.0, multiplier=1.0, train_text_encoder=False,
                             train_unet=True, is_flux=True, target_lin_modules=["FluxTransformer2DModel"], transformer_only=True)
    out = {}
    names = [m.lora_name for m in net.unet_loras]
    for m in net.unet_loras:
        out[f"init/{m.lora_name}/down"] = m.lora_down.weight.detach().clone()
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for m in net.unet_loras:
            m.lora_up.weight.copy_(torch.randn(m.lora_up.weight.shape, generator=g) * 0.05)
            out[f"warm/{m.lora_name}/up"] = m.lora_up.weight.detach().clone()
    net.force_to("cpu", torch.float32)  # order of jobs/process/BaseSDTrainProcess.py:1983-1993
    net._update_torch_multiplier()
    net.apply_to(None, model, False, True)
    inp = tiny_inputs()
    for tag, mult in (("m1", 1.0), ("mvec", [0.5, -1.5])):
        net.multiplier = mult
        for p in net.parameters():
            p.grad = None
        with net:
            pred = model(*inp)
            pred.square().sum().backward()
        out[f"{tag}/pred"] = pred.detach().clone()
        for m in net.unet_loras:
            out[f"{tag}/grad/{m.lora_name}/down"] = m.lora_down.weight.grad.detach().clone()
            out[f"{tag}/grad/{m.lora_name}/up"] = m.lora_up.weight.grad.detach().clone()
    net.multiplier = 1.0
    sd = net.get_state_dict(dtype=torch.float32)
    for k, v in sd.items():
        out[f"saved/{k}"] = v.clone()
    meta = {"names": json.dumps(names), "saved_keys": json.dumps(list(sd.keys())), "scale": json.dumps(net.unet_loras[0].scale),
            "alpha": json.dumps(float(net.unet_loras[0].alpha)), "peft_format": json.dumps(bool(net.peft_format)),
            "state_dict_keys_first": json.dumps(list(net.state_dict().keys())[:6])}
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "lora_flux_tiny.safetensors"), meta)
    print("lora golden:", len(names), "adapters;", len(sd), "saved tensors; scale", net.unet_loras[0].scale)


def golden_flowmatch():
    import diffusers

    class _Cfg(dict):
        __getattr__ = dict.get

    class FlowMatchEulerDiscreteScheduler:  # minimal stand-in for the un-vendored diffusers base class
        def __init__(self, num_train_timesteps=1000, shift=1.0, use_dynamic_shifting=False, **kw):
            self.config = _Cfg(num_train_timesteps=num_train_timesteps, shift=shift, use_dynamic_shifting=use_dynamic_shifting, **kw)
            self.shift = shift

    diffusers.FlowMatchEulerDiscreteScheduler = FlowMatchEulerDiscreteScheduler
    sys.modules.pop("toolkit.samplers.custom_flowmatch_sampler", None)
    from toolkit.samplers.custom_flowmatch_sampler import CustomFlowMatchEulerDiscreteScheduler, calculate_shift

    s = CustomFlowMatchEulerDiscreteScheduler()
    out = {"linear": s.set_train_timesteps(1000, "cpu", "linear").clone()}
    torch.manual_seed(123)
    out["sigmoid_seed123"] = s.set_train_timesteps(1000, "cpu", "sigmoid").clone()
    torch.manual_seed(321)
    out["lognorm_blend_seed321"] = s.set_train_timesteps(1000, "cpu", "lognorm_blend").clone()
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(3, 16, 8, 8, generator=g)
    eps = torch.randn(3, 16, 8, 8, generator=g)
    ts = torch.tensor([1000.0, 417.25, 1.0])
    out["x0"], out["eps"], out["ts"] = x0, eps, ts
    noisy = torch.cat([s.add_noise(x0[i:i + 1], eps[i:i + 1], ts[i:i + 1]) for i in range(3)], 0)  # per-sample loop like
    out["noisy"] = noisy  # toolkit/stable_diffusion_model.py:1861-1875
    out["calc_shift"] = torch.tensor([calculate_shift(n) for n in (256, 1024, 4096, 3952)], dtype=torch.float64)
    # bell-shaped timestep weights of the reference scheduler for a few table entries (linear table)
    s.set_train_timesteps(1000, "cpu", "linear")
    tw = s.timesteps[torch.tensor([0, 17, 499, 500, 730, 999])].clone()
    out["tw_ts"] = tw
    out["tw_v1"] = s.get_weights_for_timesteps(tw, v2=False, timestep_type="linear").clone()
    out["tw_v2"] = s.get_weights_for_timesteps(tw, v2=True, timestep_type="linear").clone()
    s.set_train_timesteps(1000, "cpu", "weighted")  # timestep_type 'weighted': the linear table + the empirical per-index weights
    out["tw_weighted_table"] = s.timesteps.clone()
    out["tw_weighted"] = s.get_weights_for_timesteps(tw, timestep_type="weighted").clone()
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "flowmatch.safetensors"))
    print("flowmatch golden written")


def golden_dora():
    """Reference LoRASpecialNetwork(network_type='dora') on the tiny FLUX oracle: init draws, forward, every gradient
    (magnitude / lora_up / lora_down) and the PEFT-format state dict it saves."""
    from toolkit.lora_special import LoRASpecialNetwork

    torch.manual_seed(0)
    model = flux_ref.FluxTransformer2DModel(**TINY)
    flux_ref.init_synthetic_(model, seed=1234, std=0.05)
    torch.manual_seed(99)
    net = LoRASpecialNetwork(text_encoder=None, unet=model, lora_dim=8, alpha=1.0, multiplier=1.0, train_text_encoder=False,
                             train_unet=True, is_flux=True, target_lin_modules=["FluxTransformer2DModel"], transformer_only=True,
                             network_type="dora")
    out = {}
    for m in net.unet_loras:
        out[f"init/{m.lora_name}/down"] = m.lora_down.weight.detach().clone()
        out[f"init/{m.lora_name}/magnitude"] = m.magnitude.detach().clone()
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for m in net.unet_loras:
            m.lora_up.weight.copy_(torch.randn(m.lora_up.weight.shape, generator=g) * 0.05)
            m.magnitude.mul_(1 + 0.05 * torch.randn(m.magnitude.shape, generator=g))
            out[f"set/{m.lora_name}/up"] = m.lora_up.weight.detach().clone()
            out[f"set/{m.lora_name}/magnitude"] = m.magnitude.detach().clone()
    net.force_to("cpu", torch.float32)
    net._update_torch_multiplier()
    net.apply_to(None, model, False, True)
    net.is_active = True
    pred = model(*tiny_inputs())
    w = torch.randn(pred.shape, generator=torch.Generator().manual_seed(11))
    (pred * w).sum().backward()
    out["fwd/pred"] = pred.detach().clone()
    out["fwd/w"] = w
    for m in net.unet_loras:
        out[f"grad/{m.lora_name}/down"] = m.lora_down.weight.grad.clone()
        out[f"grad/{m.lora_name}/up"] = m.lora_up.weight.grad.clone()
        out[f"grad/{m.lora_name}/magnitude"] = m.magnitude.grad.clone()
    sd = net.get_state_dict(dtype=torch.float32)
    for k, v in sd.items():
        out["saved/" + k] = v.clone()
    # per-sample multipliers (slider-style batch; toolkit/network_mixins.py:313-340: the LoRA term takes each sample's multiplier, the
    # DoRA weight the mean): second forward / backward of the same network
    for m in net.unet_loras:
        m.lora_down.weight.grad = m.lora_up.weight.grad = m.magnitude.grad = None
    net.multiplier = [1.0, 0.4]
    net._update_torch_multiplier()
    pred2 = model(*tiny_inputs())
    (pred2 * w).sum().backward()
    out["ps/pred"] = pred2.detach().clone()
    for m in net.unet_loras:
        out[f"ps/grad/{m.lora_name}/down"] = m.lora_down.weight.grad.clone()
        out[f"ps/grad/{m.lora_name}/up"] = m.lora_up.weight.grad.clone()
        out[f"ps/grad/{m.lora_name}/magnitude"] = m.magnitude.grad.clone()
    meta = {"names": json.dumps([m.lora_name for m in net.unet_loras]), "saved_keys": json.dumps(list(sd.keys())),
            "param_order": json.dumps([n for n, _ in net.unet_loras[0].named_parameters()]), "ps_multiplier": json.dumps([1.0, 0.4])}
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "dora_flux_tiny.safetensors"), meta)
    print("dora golden:", len(net.unet_loras), "adapters;", len(sd), "saved tensors")


def golden_lokr():
    """Reference LoRASpecialNetwork(network_type='lokr') (full factors, factor -1) on the tiny FLUX oracle: factor shapes, the
    init it draws under the same seed, forward, every factor gradient and the saved state dict (keys + values, alpha kept)."""
    from types import SimpleNamespace

    from toolkit.lora_special import LoRASpecialNetwork

    torch.manual_seed(0)
    model = flux_ref.FluxTransformer2DModel(**TINY)
    flux_ref.init_synthetic_(model, seed=1234, std=0.05)
    torch.manual_seed(99)
    big = 9999999999  # toolkit/config_modules.py:204-209 (lokr_full_rank)
    net = LoRASpecialNetwork(text_encoder=None, unet=model, lora_dim=big, alpha=big, multiplier=1.0, train_text_encoder=False,
                             train_unet=True, is_flux=True, target_lin_modules=["FluxTransformer2DModel"], transformer_only=True,
                             network_type="lokr", network_config=SimpleNamespace(lokr_factor=-1, old_lokr_format=False))
    out = {}
    shapes = {}
    for m in net.unet_loras:
        out[f"init/{m.lora_name}/w1"] = m.lokr_w1.detach().clone()
        assert float(m.lokr_w2.abs().max()) == 0.0
        shapes[m.lora_name] = [list(m.lokr_w1.shape), list(m.lokr_w2.shape)]
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for m in net.unet_loras:
            m.lokr_w2.copy_(torch.randn(m.lokr_w2.shape, generator=g) * 0.05)
            out[f"set/{m.lora_name}/w2"] = m.lokr_w2.detach().clone()
    net.force_to("cpu", torch.float32)
    net._update_torch_multiplier()
    net.apply_to(None, model, False, True)
    net.is_active = True
    pred = model(*tiny_inputs())
    w = torch.randn(pred.shape, generator=torch.Generator().manual_seed(11))
    (pred * w).sum().backward()
    out["fwd/pred"] = pred.detach().clone()
    out["fwd/w"] = w
    for m in net.unet_loras:
        out[f"grad/{m.lora_name}/w1"] = m.lokr_w1.grad.clone()
        out[f"grad/{m.lora_name}/w2"] = m.lokr_w2.grad.clone()
    sd = net.get_state_dict(dtype=torch.float32)
    for k, v in sd.items():
        out["saved/" + k] = v.clone()
    meta = {"names": json.dumps([m.lora_name for m in net.unet_loras]), "saved_keys": json.dumps(list(sd.keys())),
            "shapes": json.dumps(shapes), "scale": json.dumps(net.unet_loras[0].scale),
            "param_order": json.dumps([n for n, _ in net.unet_loras[0].named_parameters()])}
    # LokrModule.merge_in(0.7) (toolkit/models/lokr.py:261-309) into two base weights (first 24 rows kept)
    for m in net.unet_loras:
        if m.lora_name.endswith("transformer_blocks$$0$$attn$$to_q") or m.lora_name.endswith("single_transformer_blocks$$0$$proj_out"):
            w_before = m.org_module[0].weight.detach().clone()
            m.merge_in(0.7)
            out[f"merged/{m.lora_name}"] = m.org_module[0].weight.detach()[:24].clone()
            assert not torch.equal(w_before, m.org_module[0].weight)
    from toolkit.models.lokr import factorization

    dims = (64, 127, 128, 250, 256, 360, 512, 768, 1024, 1280, 1536, 3072, 8960, 9216, 12288, 15360, 18432)
    meta["factorization"] = json.dumps({f"{d}:{f}": list(factorization(d, f)) for d in dims for f in (-1, 2, 4, 8, 16)})
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "lokr_flux_tiny.safetensors"), meta)
    print("lokr golden:", len(net.unet_loras), "adapters;", len(sd), "saved tensors; scale", net.unet_loras[0].scale)


def golden_lokr_lowrank():
    """Reference LoRASpecialNetwork(network_type='lokr', lora_dim=4, alpha=2) on the tiny FLUX oracle: every layer has
    max(out_k, in_n) / 2 > 4, so lokr_w2 is the low-rank pair lokr_w2_a [out_k, 4] @ lokr_w2_b [4, in_n] (toolkit/models/lokr.py:184-197),
    scale = alpha / lora_dim = 0.5.  Shapes, the init drawn under the same seed (w2_a then w1; w2_b = 0), forward, every factor gradient,
    the saved state dict and merge_in."""
    from types import SimpleNamespace

    from toolkit.lora_special import LoRASpecialNetwork

    torch.manual_seed(0)
    model = flux_ref.FluxTransformer2DModel(**TINY)
    flux_ref.init_synthetic_(model, seed=1234, std=0.05)
    torch.manual_seed(99)
    net = LoRASpecialNetwork(text_encoder=None, unet=model, lora_dim=4, alpha=2, multiplier=1.0, train_text_encoder=False,
                             train_unet=True, is_flux=True, target_lin_modules=["FluxTransformer2DModel"], transformer_only=True,
                             network_type="lokr", network_config=SimpleNamespace(lokr_factor=-1, old_lokr_format=False))
    out = {}
    shapes = {}
    for m in net.unet_loras:
        assert not m.use_w2 and m.use_w1
        out[f"init/{m.lora_name}/w1"] = m.lokr_w1.detach().clone()
        out[f"init/{m.lora_name}/w2_a"] = m.lokr_w2_a.detach().clone()
        assert float(m.lokr_w2_b.abs().max()) == 0.0
        shapes[m.lora_name] = [list(m.lokr_w1.shape), list(m.lokr_w2_a.shape), list(m.lokr_w2_b.shape)]
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for m in net.unet_loras:
            m.lokr_w2_b.copy_(torch.randn(m.lokr_w2_b.shape, generator=g) * 0.2)
            out[f"set/{m.lora_name}/w2_b"] = m.lokr_w2_b.detach().clone()
    net.force_to("cpu", torch.float32)
    net._update_torch_multiplier()
    net.apply_to(None, model, False, True)
    net.is_active = True
    pred = model(*tiny_inputs())
    w = torch.randn(pred.shape, generator=torch.Generator().manual_seed(11))
    (pred * w).sum().backward()
    out["fwd/pred"] = pred.detach().clone()
    out["fwd/w"] = w
    for m in net.unet_loras:
        out[f"grad/{m.lora_name}/w1"] = m.lokr_w1.grad.clone()
        out[f"grad/{m.lora_name}/w2_a"] = m.lokr_w2_a.grad.clone()
        out[f"grad/{m.lora_name}/w2_b"] = m.lokr_w2_b.grad.clone()
    sd = net.get_state_dict(dtype=torch.float32)
    for k, v in sd.items():
        out["saved/" + k] = v.clone()
    meta = {"names": json.dumps([m.lora_name for m in net.unet_loras]), "saved_keys": json.dumps(list(sd.keys())),
            "shapes": json.dumps(shapes), "scale": json.dumps(net.unet_loras[0].scale),
            "param_order": json.dumps([n for n, _ in net.unet_loras[0].named_parameters()])}
    for m in net.unet_loras:
        if m.lora_name.endswith("transformer_blocks$$0$$attn$$to_q") or m.lora_name.endswith("single_transformer_blocks$$0$$proj_out"):
            m.merge_in(0.7)
            out[f"merged/{m.lora_name}"] = m.org_module[0].weight.detach()[:24].clone()
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "lokr_lowrank_flux_tiny.safetensors"), meta)
    print("lokr low-rank golden:", len(net.unet_loras), "adapters;", len(sd), "saved tensors; scale", net.unet_loras[0].scale,
          "; params", meta["param_order"])


def golden_unet_keymap_keys():
    """Parameter names of the diffusers UNet2DConditionModel (SD1.5: 686, SDXL: 1680) and of the AutoencoderKL encoder (+ quant_conv) as the
    reference's own LDM<->diffusers key maps list them (toolkit/keymaps/stable_diffusion_sd1.json / _sdxl.json, `ldm_diffusers_keymap` values with
    the `unet_` / `vae_` prefix of toolkit/saving.py stripped): the un-vendored diffusers classes' module structure, pinned on an in-tree artefact."""
    out = {}
    for tag, f in (("sd1", "stable_diffusion_sd1.json"), ("sdxl", "stable_diffusion_sdxl.json")):
        km = json.load(open(os.path.join(os.environ.get("AITK_REFERENCE", "/root/reference"), "toolkit", "keymaps", f)))["ldm_diffusers_keymap"]
        out[tag] = sorted(v[len("unet_"):] for v in km.values() if v.startswith("unet_"))
        out[tag + "_vae_encoder"] = sorted(v[len("vae_"):] for v in km.values() if v.startswith("vae_encoder") or v.startswith("vae_quant_conv"))
    assert out["sd1_vae_encoder"] == out["sdxl_vae_encoder"]
    json.dump(out, open(os.path.join(HERE, "unet_keymap_keys.json"), "w"))
    print("unet keymap keys:", {k: len(v) for k, v in out.items()})


def golden_flux_blocks():
    """FLUX block arithmetic pinned on the reference's OWN in-tree restatement of the BFL FLUX blocks — the Chroma model
    (extensions_built_in/diffusion_models/chroma/src/layers.py: DoubleStreamBlock 471-607, SingleStreamBlock 610-681, LastLayer
    684-720, QKNorm / RMSNorm 72-89, 417-427, EmbedND 13-27, timestep_embedding 30-53, MLPEmbedder 56-69; math.py: attention 13-31,
    rope 34-43, apply_rope 46-51), executed here on the oracle's tiny model weights mapped through the reference's diffusers<->BFL
    key map (scripts/convert_diffusers_to_comfy.py:67-285: q/k/v concatenated into qkv, norm_q/norm_k -> query_norm/key_norm,
    to_out.0 -> proj, ff.net.0.proj / ff.net.2 -> mlp.0 / mlp.2, single: q,k,v,proj_mlp -> linear1, proj_out -> linear2,
    norm_out.linear [scale, shift] -> adaLN_modulation [shift, scale]).  Chroma feeds the modulation vectors from outside
    (ModulationOut(shift, scale, gate)); they are taken from the oracle's adaLN Linear chunked in that order."""
    import importlib
    import types

    src = "/root/reference/extensions_built_in/diffusion_models/chroma/src"
    pkg = types.ModuleType("chroma_src")
    pkg.__path__ = [src]
    sys.modules["chroma_src"] = pkg
    L = importlib.import_module("chroma_src.layers")
    M = importlib.import_module("chroma_src.math")
    assert not M._HAS_FLASH

    torch.manual_seed(0)
    model = flux_ref.FluxTransformer2DModel(**TINY)
    flux_ref.init_synthetic_(model, seed=1234, std=0.05)
    with torch.no_grad():  # non-trivial biases / norm scales so every term is exercised
        gi = torch.Generator().manual_seed(77)
        for n, p_ in model.named_parameters():
            if n.endswith("bias"):
                p_.copy_(torch.randn(p_.shape, generator=gi) * 0.05)
            if "norm_" in n and n.endswith("weight"):
                p_.copy_(1 + 0.2 * torch.randn(p_.shape, generator=gi))
    d, H = 256, 2
    g = torch.Generator().manual_seed(21)
    B, Hl, Wl, n_txt = 2, 4, 6, 5
    Si = (Hl // 2) * (Wl // 2)
    img = torch.randn(B, Si, d, generator=g)
    txt = torch.randn(B, n_txt, d, generator=g)
    temb = torch.randn(B, d, generator=g)
    img_ids, txt_ids = flux_ref.make_ids(Hl, Wl, n_txt)
    ids = torch.cat((txt_ids, img_ids), 0)[None].repeat(B, 1, 1)
    out = {"in/img": img, "in/txt": txt, "in/temb": temb, "in/ids": ids[0]}
    # the weights are regenerated by the test with the same three lines (seeded); a checksum guards the regeneration
    out["w_checksum"] = torch.stack([v.double().abs().sum() for v in model.state_dict().values()]).float()

    def cat(*ws):
        return torch.cat([w.detach() for w in ws], 0)

    with torch.no_grad():
        pe = L.EmbedND(dim=128, theta=10000, axes_dim=[16, 56, 56])(ids)
        out["ref/pe"] = pe.clone()
        # ---- double block
        blk = model.transformer_blocks[0]
        ref = L.DoubleStreamBlock(d, H, mlp_ratio=4.0, qkv_bias=True)
        a = blk.attn
        sd = {"img_attn.qkv.weight": cat(a.to_q.weight, a.to_k.weight, a.to_v.weight), "img_attn.qkv.bias": cat(a.to_q.bias, a.to_k.bias, a.to_v.bias),
              "img_attn.norm.query_norm.scale": a.norm_q.weight, "img_attn.norm.key_norm.scale": a.norm_k.weight,
              "img_attn.proj.weight": a.to_out[0].weight, "img_attn.proj.bias": a.to_out[0].bias,
              "img_mlp.0.weight": blk.ff.net[0].proj.weight, "img_mlp.0.bias": blk.ff.net[0].proj.bias,
              "img_mlp.2.weight": blk.ff.net[2].weight, "img_mlp.2.bias": blk.ff.net[2].bias,
              "txt_attn.qkv.weight": cat(a.add_q_proj.weight, a.add_k_proj.weight, a.add_v_proj.weight),
              "txt_attn.qkv.bias": cat(a.add_q_proj.bias, a.add_k_proj.bias, a.add_v_proj.bias),
              "txt_attn.norm.query_norm.scale": a.norm_added_q.weight, "txt_attn.norm.key_norm.scale": a.norm_added_k.weight,
              "txt_attn.proj.weight": a.to_add_out.weight, "txt_attn.proj.bias": a.to_add_out.bias,
              "txt_mlp.0.weight": blk.ff_context.net[0].proj.weight, "txt_mlp.0.bias": blk.ff_context.net[0].proj.bias,
              "txt_mlp.2.weight": blk.ff_context.net[2].weight, "txt_mlp.2.bias": blk.ff_context.net[2].bias}
        ref.load_state_dict({k: v.detach().clone() for k, v in sd.items()}, strict=True)

        def mods(lin, n):
            c = lin(torch.nn.functional.silu(temb))[:, None, :].chunk(n, dim=-1)
            return [L.ModulationOut(*c[i:i + 3]) for i in range(0, n, 3)]

        r_img, r_txt = ref(img=img, txt=txt, pe=pe, distill_vec=[mods(blk.norm1.linear, 6), mods(blk.norm1_context.linear, 6)], mask=None)
        out["ref/double/img"], out["ref/double/txt"] = r_img.clone(), r_txt.clone()
        # ---- single block on the concatenated stream
        sb = model.single_transformer_blocks[0]
        refs = L.SingleStreamBlock(d, H, mlp_ratio=4.0)
        a = sb.attn
        refs.load_state_dict({"linear1.weight": cat(a.to_q.weight, a.to_k.weight, a.to_v.weight, sb.proj_mlp.weight),
                              "linear1.bias": cat(a.to_q.bias, a.to_k.bias, a.to_v.bias, sb.proj_mlp.bias),
                              "linear2.weight": sb.proj_out.weight.detach().clone(), "linear2.bias": sb.proj_out.bias.detach().clone(),
                              "norm.query_norm.scale": a.norm_q.weight.detach().clone(), "norm.key_norm.scale": a.norm_k.weight.detach().clone()},
                             strict=True)
        x = torch.cat((r_txt, r_img), 1)
        r_x = refs(x, pe=pe, distill_vec=mods(sb.norm.linear, 3)[0], mask=None)
        out["ref/single/x"] = r_x.clone()
        # ---- last layer on the image tokens: BFL order [shift, scale] = diffusers norm_out.linear [scale, shift] swapped
        last = L.LastLayer(d, 1, TINY["in_channels"])
        last.load_state_dict({"linear.weight": model.proj_out.weight.detach().clone(), "linear.bias": model.proj_out.bias.detach().clone()})
        scale, shift = model.norm_out.linear(torch.nn.functional.silu(temb)).chunk(2, dim=1)
        out["ref/last"] = last(r_x[:, n_txt:], [shift[:, None], scale[:, None]]).clone()
        # ---- conditioning embedders
        tt = torch.tensor([0.3, 0.8])
        out["in/t"] = tt
        out["ref/t_sinusoid"] = L.timestep_embedding(tt, 256).clone()  # time_factor 1000: the model is called with timestep / 1000
        emb = L.MLPEmbedder(256, d)
        te = model.time_text_embed.timestep_embedder
        emb.load_state_dict({"in_layer.weight": te.linear_1.weight.detach().clone(), "in_layer.bias": te.linear_1.bias.detach().clone(),
                             "out_layer.weight": te.linear_2.weight.detach().clone(), "out_layer.bias": te.linear_2.bias.detach().clone()})
        out["ref/t_mlp"] = emb(out["ref/t_sinusoid"]).clone()
        # embedder composition: the reference's guidance_embed_bypass_forward (toolkit/models/flux.py:8-14) executed on the
        # oracle's time_text_embed module = timestep_embedder(time_proj(t)) + text_embedder(pooled)  (no guidance term)
        import importlib.util as _ilu

        spec = _ilu.spec_from_file_location("flux_bypass_ref", "/root/reference/toolkit/models/flux.py")
        fb = _ilu.module_from_spec(spec)
        spec.loader.exec_module(fb)
        pooled = torch.randn(2, TINY["pooled_projection_dim"], generator=g)
        t1000 = tt * 1000  # the transformer multiplies its timestep / 1000 input by 1000 before time_text_embed
        out["in/pooled"] = pooled
        out["ref/cond_no_guidance"] = fb.guidance_embed_bypass_forward(model.time_text_embed, t1000, None, pooled).clone()
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "flux_blocks_chroma.safetensors"),
              {"cfg": json.dumps(TINY), "shape": json.dumps([B, Hl, Wl, n_txt])})
    print("flux block golden (reference Chroma blocks) written:", {k: tuple(v.shape) for k, v in out.items() if k.startswith("ref/")})


def golden_wan_attn():
    """The reference's Wan attention processor (toolkit/models/wan21/wan_attn.py:8-84: projection -> q/k norm -> heads -> complex
    RoPE in float64 -> SDPA -> to_out) executed on the Wan oracle's Attention module (same attribute names as the diffusers
    Attention it is written for), self-attention with rotary_emb and text cross-attention without."""
    import importlib.util

    from oracle import wan_ref

    spec = importlib.util.spec_from_file_location("wan_attn_ref", "/root/reference/toolkit/models/wan21/wan_attn.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    proc = m.WanAttnProcessor2_0()
    torch.manual_seed(31)
    attn = wan_ref.Attention(256, 2, 128)
    with torch.no_grad():
        for p_ in attn.parameters():
            p_.copy_(torch.randn(p_.shape) * 0.05)
        attn.norm_q.weight.copy_(1 + 0.2 * torch.randn(256))
        attn.norm_k.weight.copy_(1 + 0.2 * torch.randn(256))
    attn.add_k_proj = None
    g = torch.Generator().manual_seed(32)
    Fr, Hh, W = 2, 3, 4
    x = torch.randn(2, Fr * Hh * W, 256, generator=g)
    enc = torch.randn(2, 7, 256, generator=g)
    ang = wan_ref.wan_rope_freqs(Fr, Hh, W)
    freqs = torch.polar(torch.ones_like(ang), ang)[None, None]  # complex128 [1, 1, S, 64] (WanRotaryPosEmbed output form)
    with torch.no_grad():
        out = {"x": x, "enc": enc, "self": proc(attn, x, None, None, freqs).clone(), "cross": proc(attn, x, enc, None, None).clone(),
               "w_checksum": torch.stack([v.double().abs().sum() for v in attn.state_dict().values()]).float()}
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "wan_attn.safetensors"), {"grid": json.dumps([Fr, Hh, W])})
    print("wan attention golden written")


def golden_optimizer_ema():
    """The per-step tail of SDTrainer.hook_train_loop (2273-2293): clip_grad_norm_ -> torch.optim.AdamW(eps=1e-6) as built by
    toolkit/optimizer.py:78-79 -> the reference's own toolkit/ema.py ExponentialMovingAverage.update(), run for 3 steps on a
    small flat parameter; the fused aitk_adamw_ema_step must land on the same parameters and EMA shadow."""
    from toolkit.ema import ExponentialMovingAverage

    g = torch.Generator().manual_seed(41)
    p = torch.nn.Parameter(torch.randn(4096, generator=g) * 0.1)
    p0 = p.detach().clone()
    opt = torch.optim.AdamW([p], lr=3e-3, eps=1e-6, weight_decay=0.01)
    ema = ExponentialMovingAverage([p], decay=0.9)
    grads = torch.randn(3, 4096, generator=g) * torch.tensor([0.02, 3.0, 0.5])[:, None]  # below / above / near the clip threshold
    norms = []
    for k in range(3):
        p.grad = grads[k].clone()
        norms.append(torch.nn.utils.clip_grad_norm_([p], 1.0).clone())
        opt.step()
        opt.zero_grad(set_to_none=True)
        ema.update()
    out = {"p0": p0, "grads": grads, "p3": p.detach().clone(), "ema3": ema.shadow_params[0].clone(), "norms": torch.stack(norms)}
    save_file(out, os.path.join(HERE, "optimizer_ema.safetensors"))
    print("optimizer + EMA golden written; grad norms", [round(float(n), 3) for n in norms])


def golden_ema_options():
    """toolkit/ema.py options the trainer passes (jobs/process/BaseSDTrainProcess.py:798-803): use_feedback (the parameter is pulled
    10 x the EMA step towards the shadow) + param_multiplier, and the class's use_num_updates warm-up — three AdamW steps each, run on
    the reference's own ExponentialMovingAverage."""
    from toolkit.ema import ExponentialMovingAverage

    out = {}
    for tag, kw in (("fb", dict(use_feedback=True, param_multiplier=0.999)), ("nu", dict(use_num_updates=True)), ("pm", dict(param_multiplier=1.002))):
        g = torch.Generator().manual_seed(43)
        p = torch.nn.Parameter(torch.randn(4096, generator=g) * 0.1)
        out[f"{tag}/p0"] = p.detach().clone()
        opt = torch.optim.AdamW([p], lr=3e-3, eps=1e-6, weight_decay=0.01)
        ema = ExponentialMovingAverage([p], decay=0.9, **kw)
        grads = torch.randn(3, 4096, generator=g) * torch.tensor([0.02, 3.0, 0.5])[:, None]
        for k in range(3):
            p.grad = grads[k].clone()
            torch.nn.utils.clip_grad_norm_([p], 1.0)
            opt.step()
            opt.zero_grad(set_to_none=True)
            ema.update()
        out[f"{tag}/grads"] = grads
        out[f"{tag}/p3"] = p.detach().clone()
        out[f"{tag}/ema3"] = ema.shadow_params[0].clone()
    save_file(out, os.path.join(HERE, "ema_options.safetensors"))
    print("EMA options golden written")


def golden_model_hash():
    """sshs_model_hash / sshs_legacy_hash the reference stamps on saved adapters (toolkit/metadata.py:32-48), computed by its own
    add_model_hash_to_meta on a small and on a > 1 MiB state dict (the legacy hash reads bytes at offset 0x100000)."""
    import types
    from collections import OrderedDict

    sys.modules.setdefault("info", types.SimpleNamespace(software_meta={"name": "ai-toolkit"}))
    from toolkit.metadata import add_model_hash_to_meta

    g = torch.Generator().manual_seed(51)
    res = {}
    for tag, shape in (("small", (8, 16)), ("big", (600, 512))):
        sd = OrderedDict((f"transformer.blocks.{i}.lora_A.weight", torch.randn(shape, generator=g).to(torch.float16)) for i in range(3))
        meta = add_model_hash_to_meta(sd, OrderedDict(ss_output_name="x", ss_base_model_version="flux1", name="not hashed"))
        res[tag] = {"shape": list(shape), "sshs_model_hash": meta["sshs_model_hash"], "sshs_legacy_hash": meta["sshs_legacy_hash"]}
    json.dump(res, open(os.path.join(HERE, "model_hash.json"), "w"), indent=1)
    print("model hash golden written", res)


def golden_merge():
    """Reference LoRASpecialNetwork.merge_in(0.7) (toolkit/network_mixins.py:370-462, 894-899) on the tiny FLUX oracle with the
    warm adapter of golden_lora: the merged base weights of three wrapped Linears."""
    from toolkit.lora_special import LoRASpecialNetwork

    torch.manual_seed(0)
    model = flux_ref.FluxTransformer2DModel(**TINY)
    flux_ref.init_synthetic_(model, seed=1234, std=0.05)
    torch.manual_seed(99)
    net = LoRASpecialNetwork(text_encoder=None, unet=model, lora_dim=8, alpha=1.0, multiplier=1.0, train_text_encoder=False,
                             train_unet=True, is_flux=True, target_lin_modules=["FluxTransformer2DModel"], transformer_only=True)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for m in net.unet_loras:
            m.lora_up.weight.copy_(torch.randn(m.lora_up.weight.shape, generator=g) * 0.05)
    net.force_to("cpu", torch.float32)
    net.apply_to(None, model, False, True)
    net.merge_in(merge_weight=0.7)
    assert net.is_merged_in
    keys = ["transformer_blocks.0.attn.to_q.weight", "transformer_blocks.0.ff.net.0.proj.weight", "single_transformer_blocks.0.proj_out.weight"]
    sd = model.state_dict()
    save_file({k: sd[k][:24].clone().contiguous() for k in keys}, os.path.join(HERE, "merge_flux_tiny.safetensors"))  # first 24 rows
    print("merge golden written")


def golden_vae_encoder():
    """VAE encoder arithmetic pinned on the reference's in-tree LDM-style `Encoder`
    (extensions_built_in/diffusion_models/flux2/src/autoencoder.py:36-233: ResnetBlock, Downsample with the (0,1,0,1) pad,
    AttnBlock, mid block, norm_out -> swish -> conv_out), the architecture of the FLUX.1 AutoencoderKL encoder, executed on the
    VAE oracle's weights mapped diffusers -> LDM names (down_blocks.i.resnets.j -> down.i.block.j, conv_shortcut -> nin_shortcut,
    downsamplers.0.conv -> downsample.conv, mid_block.resnets.0/1 -> mid.block_1/2, attention Linear -> 1x1 conv,
    conv_norm_out -> norm_out); its extra quant_conv is set to the identity (FLUX.1's VAE has none)."""
    import importlib.util

    from oracle import vae_ref

    spec = importlib.util.spec_from_file_location("flux2_autoencoder_ref", "/root/reference/extensions_built_in/diffusion_models/flux2/src/autoencoder.py")
    m = importlib.util.module_from_spec(spec)
    sys.modules["flux2_autoencoder_ref"] = m  # dataclasses resolve the module through sys.modules
    spec.loader.exec_module(m)
    chans, zc = (32, 64, 64), 4
    torch.manual_seed(0)
    mine = vae_ref.Encoder(3, zc, chans, 2, 32)
    with torch.no_grad():
        g = torch.Generator().manual_seed(61)
        for n, p_ in mine.named_parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * (0.08 if p_.dim() > 1 else 0.05) + (1.0 if ("norm" in n and n.endswith("weight")) else 0.0))
    ref = m.Encoder(resolution=32, in_channels=3, ch=32, ch_mult=[1, 2, 2], num_res_blocks=2, z_channels=zc)
    sd = {}
    for k, v in mine.state_dict().items():
        k2 = k
        for i in range(3):
            for j in range(2):
                k2 = k2.replace(f"down_blocks.{i}.resnets.{j}.", f"down.{i}.block.{j}.")
            k2 = k2.replace(f"down_blocks.{i}.downsamplers.0.conv.", f"down.{i}.downsample.conv.")
        k2 = k2.replace("mid_block.resnets.0.", "mid.block_1.").replace("mid_block.resnets.1.", "mid.block_2.")
        k2 = k2.replace("mid_block.attentions.0.group_norm.", "mid.attn_1.norm.")
        for a, b in (("to_q", "q"), ("to_k", "k"), ("to_v", "v"), ("to_out.0", "proj_out")):
            k2 = k2.replace(f"mid_block.attentions.0.{a}.", f"mid.attn_1.{b}.")
        k2 = k2.replace("conv_shortcut.", "nin_shortcut.").replace("conv_norm_out.", "norm_out.")
        if "mid.attn_1." in k2 and k2.endswith("weight") and v.dim() == 2:
            v = v[:, :, None, None]
        sd[k2] = v.clone()
    sd["quant_conv.weight"] = torch.eye(2 * zc)[:, :, None, None].clone()
    sd["quant_conv.bias"] = torch.zeros(2 * zc)
    ref.load_state_dict(sd, strict=True)
    x = torch.randn(2, 3, 32, 24, generator=torch.Generator().manual_seed(62))
    with torch.no_grad():
        out = {"x": x, "moments": ref(x).clone(),
               "w_checksum": torch.stack([v.double().abs().sum() for v in mine.state_dict().values()]).float()}
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "vae_encoder_ldm.safetensors"),
              {"chans": json.dumps(list(chans)), "zc": json.dumps(zc)})
    print("vae encoder golden written", tuple(out["moments"].shape))


def golden_latent_cache_paths():
    """`_latent_cache/<stem>_<hash>.safetensors` names computed by the reference's own FileItemDTO mixin methods
    (toolkit/dataloader_mixins.py:1779-1842 get_latent_info_dict / get_latent_path) for a few crop plans / flips."""
    import types

    sys.modules.setdefault("info", types.SimpleNamespace(software_meta={"name": "ai-toolkit"}))
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    import toolkit.dataloader_mixins as dm

    cases = []
    for path, geo, flips, lsv in (("/data/set/cat 01.JPG", (1536, 1024, 0, 0, 1536, 1024), (False, False), "flux1"),
                                  ("/data/x/img.png", (1024, 1540, 0, 2, 1024, 1536), (True, False), "flux1"),
                                  ("/data/x/a.b.webp", (832, 1250, 0, 17, 832, 1216), (False, True), "sdxl"),
                                  ("rel/dir/q.jpeg", (1216, 832, 0, 0, 1216, 832), (True, True), "wan21")):
        it = types.SimpleNamespace(path=path, scale_to_width=geo[0], scale_to_height=geo[1], crop_x=geo[2], crop_y=geo[3],
                                   crop_width=geo[4], crop_height=geo[5], latent_space_version=lsv, latent_version=1,
                                   flip_x=flips[0], flip_y=flips[1], is_video=False, is_audio_model=False, _latent_path=None,
                                   dataset_config=types.SimpleNamespace(cache_tensors_to_disk=False, auto_frame_count=False, num_frames=1))
        it.get_latent_info_dict = types.MethodType(dm.LatentCachingFileItemDTOMixin.get_latent_info_dict, it)
        out_path = dm.LatentCachingFileItemDTOMixin.get_latent_path(it)
        cases.append({"path": path, "geometry": list(geo), "flip_x": flips[0], "flip_y": flips[1], "latent_space_version": lsv,
                      "latent_path": out_path, "info": it.get_latent_info_dict()})
    # text-embedding cache names (2120-2163) for plain captions, and a PromptEmbeds file written by the reference's own
    # PromptEmbeds.save (toolkit/prompt_utils.py:119-141)
    te = []
    for path, caption, space in (("/data/set/cat 01.JPG", "a photo of a cat, [trigger]", "flux1"), ("/d/x.png", "", "wan21"),
                                 ("rel/q.jpeg", "ünïcode caption \u2603", "flux1")):
        it = types.SimpleNamespace(path=path, caption=caption, text_embedding_space_version=space, text_embedding_version=1,
                                   encode_control_in_text_embeddings=False, control_path=None, is_video=False,
                                   dataset_config=types.SimpleNamespace(do_i2v=False), _text_embedding_path=None)
        it.get_text_embedding_info_dict = types.MethodType(dm.TextEmbeddingFileItemDTOMixin.get_text_embedding_info_dict, it)
        it._build_text_embedding_path = types.MethodType(dm.TextEmbeddingFileItemDTOMixin._build_text_embedding_path, it)
        te.append({"path": path, "caption": caption, "space": space, "te_path": dm.TextEmbeddingFileItemDTOMixin.get_text_embedding_path(it)})
    from toolkit.prompt_utils import PromptEmbeds

    g = torch.Generator().manual_seed(71)
    pe = PromptEmbeds(None)
    pe.text_embeds = torch.randn(1, 6, 8, generator=g).to(torch.bfloat16)
    pe.pooled_embeds = torch.randn(1, 4, generator=g).to(torch.bfloat16)
    pe.attention_mask = None
    pe.save(os.path.join(HERE, "prompt_embeds_ref.safetensors"))
    json.dump({"latent": cases, "text": te}, open(os.path.join(HERE, "latent_cache_paths.json"), "w"), indent=1)
    print("cache path golden written:", [os.path.basename(c["latent_path"]) for c in cases], [os.path.basename(c["te_path"]) for c in te])


def golden_kohya_to_peft():
    """The reference's scripts/convert_lora_to_peft_format.py (a CLI script) run on a kohya-format file holding the FLUX adapter
    names of golden_lora (alpha = rank, as ai-toolkit writes): the PEFT keys it produces, in order."""
    import subprocess
    import tempfile

    from safetensors import safe_open
    from safetensors.torch import load_file

    src = os.path.join(HERE, "lora_flux_tiny.safetensors")
    with safe_open(src, "pt") as f:
        names = json.loads(f.metadata()["names"])
    g = torch.Generator().manual_seed(81)
    kohya = {}
    for n in names:  # transformer$$transformer_blocks$$0$$attn$$to_q -> lora_transformer_transformer_blocks_0_attn_to_q
        base = "lora_" + n.replace("$$", "_")
        kohya[base + ".lora_down.weight"] = torch.randn(8, 4, generator=g)
        kohya[base + ".lora_up.weight"] = torch.randn(4, 8, generator=g)
        kohya[base + ".alpha"] = torch.tensor(8.0)
    with tempfile.TemporaryDirectory() as td:
        a, b = os.path.join(td, "kohya.safetensors"), os.path.join(td, "peft.safetensors")
        save_file(kohya, a)
        subprocess.check_call([sys.executable, "/root/reference/scripts/convert_lora_to_peft_format.py", a, b], stdout=subprocess.DEVNULL)
        out = load_file(b)
    json.dump({"kohya_keys": list(kohya), "peft_keys": sorted(out)}, open(os.path.join(HERE, "kohya_to_peft_keys.json"), "w"), indent=0)
    chk = {k: float(v.double().sum()) for k, v in out.items()}
    json.dump(chk, open(os.path.join(HERE, "kohya_to_peft_sums.json"), "w"), indent=0)
    print("kohya -> peft golden written:", len(out), "tensors")


def golden_wan_lora_keys():
    """Key names written by the reference's Wan adapter converter (toolkit/models/wan21/wan_lora_convert.py)."""
    import importlib.util
    import json

    spec = importlib.util.spec_from_file_location("wlc", "/root/reference/toolkit/models/wan21/wan_lora_convert.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    keys = []
    for n in range(3):
        for a in ("attn1", "attn2"):
            for pr in ("to_q", "to_k", "to_v", "to_out.0", "add_k_proj", "add_v_proj"):
                keys.append(f"transformer.blocks.{n}.{a}.{pr}")
        keys += [f"transformer.blocks.{n}.ffn.net.0.proj", f"transformer.blocks.{n}.ffn.net.2"]
    sd = {k + sfx: i for i, k in enumerate(keys) for sfx in (".lora_A.weight", ".lora_B.weight")}
    orig = m.convert_to_original(sd)
    assert list(m.convert_to_diffusers(orig)) == list(sd)
    json.dump({"diffusers": list(sd), "original": list(orig)}, open(os.path.join(HERE, "wan_lora_keys.json"), "w"), indent=0)
    print("wan lora keys golden written")


def golden_unet_lora():
    """Reference LoRASpecialNetwork on the UNet oracle trees (kohya format, the SD1.5 / SDXL branch of toolkit/lora_special.py:457-647):
    tiny SD1.5-like (1x1-conv proj_in / proj_out adapters) and SDXL-like models -> adapter names, init draws under manual_seed(99),
    one forward + every adapter gradient through the reference's own LoRAModule.forward, and the state dict it saves; plus the adapter
    name lists of the FULL SD1.5 (192) and SDXL (722) trees (built on the meta device) as count + sha256."""
    import hashlib

    from toolkit.lora_special import LoRASpecialNetwork

    from oracle import unet_ref
    from tests.test_unet_cpu import TINY_SD15, TINY_SDXL, _inputs

    out, meta = {}, {}
    for tag, cfg, is_xl in (("sd15", TINY_SD15, False), ("sdxl", TINY_SDXL, True)):
        torch.manual_seed(0)
        model = unet_ref.UNet2DConditionModel(**cfg)
        unet_ref.init_synthetic_(model, seed=11)
        torch.manual_seed(99)
        net = LoRASpecialNetwork(text_encoder=None, unet=model, lora_dim=4, alpha=2.0, multiplier=1.0, train_text_encoder=False,
                                 train_unet=True, is_sdxl=is_xl)
        names = [m.lora_name for m in net.unet_loras]
        for m in net.unet_loras:
            out[f"{tag}/init/{m.lora_name}/down"] = m.lora_down.weight.detach().clone()
        g = torch.Generator().manual_seed(7)
        with torch.no_grad():
            for m in net.unet_loras:
                m.lora_up.weight.copy_(torch.randn(m.lora_up.weight.shape, generator=g) * 0.05)
                out[f"{tag}/warm/{m.lora_name}/up"] = m.lora_up.weight.detach().clone()
        net.force_to("cpu", torch.float32)
        net._update_torch_multiplier()
        net.apply_to(None, model, False, True)
        lat, ts, ctx, added = _inputs(cfg)
        with net:
            pred = model(lat, ts, ctx, added)
            wgt = torch.randn(pred.shape, generator=torch.Generator().manual_seed(11))
            (pred * wgt).sum().backward()
        out[f"{tag}/pred"], out[f"{tag}/wgt"] = pred.detach().clone(), wgt
        for m in net.unet_loras:
            out[f"{tag}/grad/{m.lora_name}/down"] = m.lora_down.weight.grad.detach().clone()
            out[f"{tag}/grad/{m.lora_name}/up"] = m.lora_up.weight.grad.detach().clone()
        sd = net.get_state_dict(dtype=torch.float32)
        for k, v in sd.items():
            out[f"{tag}/saved/{k}"] = v.clone()
        meta[tag] = {"names": names, "saved_keys": list(sd.keys()), "scale": net.unet_loras[0].scale, "peft_format": bool(net.peft_format)}
        print(f"unet lora golden [{tag}]:", len(names), "adapters;", len(sd), "saved tensors; scale", net.unet_loras[0].scale)
    for tag, cfg, is_xl in (("sd15_full", unet_ref.SD15, False), ("sdxl_full", unet_ref.SDXL, True)):
        with torch.device("meta"):
            model = unet_ref.UNet2DConditionModel(**cfg)
        net = LoRASpecialNetwork(text_encoder=None, unet=model, lora_dim=4, alpha=4.0, multiplier=1.0, train_text_encoder=False,
                                 train_unet=True, is_sdxl=is_xl)
        names = [m.lora_name for m in net.unet_loras]
        shapes = [[list(m.lora_down.weight.shape), list(m.lora_up.weight.shape)] for m in net.unet_loras]
        meta[tag] = {"count": len(names), "names_sha256": hashlib.sha256("\n".join(names).encode()).hexdigest(), "first": names[:3], "last": names[-3:],
                     "shapes_sha256": hashlib.sha256(json.dumps(shapes).encode()).hexdigest(),
                     "params": int(sum(m.lora_down.weight.numel() + m.lora_up.weight.numel() for m in net.unet_loras))}
        print(f"unet lora golden [{tag}]:", meta[tag]["count"], "adapters,", meta[tag]["params"], "parameters")
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "unet_lora_tiny.safetensors"), {"meta": json.dumps(meta)})


def golden_unet_conv_lora():
    """The same as golden_unet_lora with `network.conv` set (conv_lora_dim / conv_alpha: toolkit/lora_special.py:585-587, 678-681): the
    reference then also wraps every Linear / Conv2d child of ResnetBlock2D, Downsample2D and Upsample2D (toolkit/kohya_lora.py:750-751) —
    3x3 convolutions at the conv rank (lora_down = Conv2d(in, r, 3, stride, padding), lora_up = Conv2d(r, out, 1): lora_special.py:95-104),
    `time_emb_proj` and the 1x1 `conv_shortcut` at the linear rank."""
    import hashlib

    from toolkit.lora_special import LoRASpecialNetwork

    from oracle import unet_ref
    from tests.test_unet_cpu import TINY_SD15, TINY_SDXL, _inputs

    out, meta = {}, {}
    for tag, cfg, is_xl in (("sd15", TINY_SD15, False), ("sdxl", TINY_SDXL, True)):
        torch.manual_seed(0)
        model = unet_ref.UNet2DConditionModel(**cfg)
        unet_ref.init_synthetic_(model, seed=11)
        torch.manual_seed(99)
        net = LoRASpecialNetwork(text_encoder=None, unet=model, lora_dim=4, alpha=2.0, conv_lora_dim=2, conv_alpha=1.0, multiplier=1.0,
                                 train_text_encoder=False, train_unet=True, is_sdxl=is_xl)
        names = [m.lora_name for m in net.unet_loras]
        for m in net.unet_loras:
            out[f"{tag}/init/{m.lora_name}/down"] = m.lora_down.weight.detach().clone()
        g = torch.Generator().manual_seed(7)
        with torch.no_grad():
            for m in net.unet_loras:
                m.lora_up.weight.copy_(torch.randn(m.lora_up.weight.shape, generator=g) * 0.05)
                out[f"{tag}/warm/{m.lora_name}/up"] = m.lora_up.weight.detach().clone()
        net.force_to("cpu", torch.float32)
        net._update_torch_multiplier()
        net.apply_to(None, model, False, True)
        lat, ts, ctx, added = _inputs(cfg)
        with net:
            pred = model(lat, ts, ctx, added)
            wgt = torch.randn(pred.shape, generator=torch.Generator().manual_seed(11))
            (pred * wgt).sum().backward()
        out[f"{tag}/pred"], out[f"{tag}/wgt"] = pred.detach().clone(), wgt
        for m in net.unet_loras:
            out[f"{tag}/grad/{m.lora_name}/down"] = m.lora_down.weight.grad.detach().clone()
            out[f"{tag}/grad/{m.lora_name}/up"] = m.lora_up.weight.grad.detach().clone()
        sd = net.get_state_dict(dtype=torch.float32)
        for k, v in sd.items():
            out[f"{tag}/saved/{k}"] = v.clone()
        meta[tag] = {"names": names, "saved_keys": list(sd.keys()), "scales": [m.scale for m in net.unet_loras],
                     "dims": [m.lora_dim for m in net.unet_loras], "peft_format": bool(net.peft_format)}
        print(f"unet conv-lora golden [{tag}]:", len(names), "adapters;", len(sd), "saved tensors")
    for tag, cfg, is_xl in (("sd15_full", unet_ref.SD15, False), ("sdxl_full", unet_ref.SDXL, True)):
        with torch.device("meta"):
            model = unet_ref.UNet2DConditionModel(**cfg)
        net = LoRASpecialNetwork(text_encoder=None, unet=model, lora_dim=4, alpha=4.0, conv_lora_dim=4, conv_alpha=4.0, multiplier=1.0,
                                 train_text_encoder=False, train_unet=True, is_sdxl=is_xl)
        names = [m.lora_name for m in net.unet_loras]
        shapes = [[list(m.lora_down.weight.shape), list(m.lora_up.weight.shape)] for m in net.unet_loras]
        meta[tag] = {"count": len(names), "names_sha256": hashlib.sha256("\n".join(names).encode()).hexdigest(), "first": names[:3], "last": names[-3:],
                     "shapes_sha256": hashlib.sha256(json.dumps(shapes).encode()).hexdigest(),
                     "params": int(sum(m.lora_down.weight.numel() + m.lora_up.weight.numel() for m in net.unet_loras))}
        print(f"unet conv-lora golden [{tag}]:", meta[tag]["count"], "adapters,", meta[tag]["params"], "parameters")
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "unet_conv_lora_tiny.safetensors"), {"meta": json.dumps(meta)})


def golden_unet_conv_lora_highrank():
    """golden_unet_conv_lora at ranks beyond one 16-wide rank block / one 64-rank skinny launch: conv rank 24 with linear rank 80 (SDXL-like
    tree) and conv rank 40 with linear rank 4 (SD1.5-like tree) through the reference's LoRASpecialNetwork / LoRAModule.forward."""
    from toolkit.lora_special import LoRASpecialNetwork

    from oracle import unet_ref
    from tests.test_unet_cpu import TINY_SD15, TINY_SDXL, _inputs

    out, meta = {}, {}
    for tag, cfg, is_xl, lin_r, conv_r in (("sdxl", TINY_SDXL, True, 80, 24), ("sd15", TINY_SD15, False, 4, 40)):
        torch.manual_seed(0)
        model = unet_ref.UNet2DConditionModel(**cfg)
        unet_ref.init_synthetic_(model, seed=11)
        torch.manual_seed(99)
        net = LoRASpecialNetwork(text_encoder=None, unet=model, lora_dim=lin_r, alpha=lin_r / 2, conv_lora_dim=conv_r, conv_alpha=conv_r / 4,
                                 multiplier=1.0, train_text_encoder=False, train_unet=True, is_sdxl=is_xl)
        g = torch.Generator().manual_seed(7)
        with torch.no_grad():
            for m in net.unet_loras:  # the test repeats these draws (and the constructor's kaiming draws under seed 99, pinned bit for bit by
                m.lora_up.weight.copy_(torch.randn(m.lora_up.weight.shape, generator=g) * 0.05)  # golden_unet_conv_lora) instead of storing them
        net.force_to("cpu", torch.float32)
        net._update_torch_multiplier()
        net.apply_to(None, model, False, True)
        lat, ts, ctx, added = _inputs(cfg)
        with net:
            pred = model(lat, ts, ctx, added)
            wgt = torch.randn(pred.shape, generator=torch.Generator().manual_seed(11))
            (pred * wgt).sum().backward()
        out[f"{tag}/pred"], out[f"{tag}/wgt"] = pred.detach().clone(), wgt
        # every gradient matrix G [rows, cols] is stored through two fixed random projections (G v and u G, fp32) and its norm: the full
        # matrices of a rank-80 network would be 25 MB of fixture
        gp = torch.Generator().manual_seed(123)
        for m in net.unet_loras:
            for nm, w in (("down", m.lora_down.weight), ("up", m.lora_up.weight)):
                G = w.grad.detach().reshape(w.shape[0], -1)
                v, u = torch.randn(G.shape[1], generator=gp), torch.randn(G.shape[0], generator=gp)
                out[f"{tag}/grad/{m.lora_name}/{nm}"] = torch.cat((G @ v, u @ G, G.norm().reshape(1)))
        meta[tag] = {"names": [m.lora_name for m in net.unet_loras], "lin_rank": lin_r, "conv_rank": conv_r,
                     "dims": [m.lora_dim for m in net.unet_loras], "scales": [m.scale for m in net.unet_loras]}
        print(f"unet conv-lora high-rank golden [{tag}]:", len(meta[tag]["names"]), "adapters")
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "unet_conv_lora_highrank.safetensors"), {"meta": json.dumps(meta)})



def golden_base_model_contract():
    """The plug-in contract of the reference, read off its own classes by introspection (imported under the shims, nothing executed):
    toolkit/models/base_model.py `BaseModel` — every public method with its parameter names, which of them are the "must be implemented in
    child classes" hooks (their body raises NotImplementedError, base_model.py:306-360), the properties, and the class attribute `arch`;
    and what the two in-tree plug-ins that our models mirror define themselves (extensions_built_in/diffusion_models/flux_kontext/
    flux_kontext.py:41-419 `FluxKontextModel`, toolkit/models/wan21/wan21.py:310-736 `Wan21`), plus how a plug-in is registered and
    selected (toolkit/util/get_model.py:20-50: module attribute AI_TOOLKIT_MODELS, match on `arch`).  tests/test_plugin_contract_cpu.py
    holds ai_toolkit_amd.plugin to it."""
    import inspect
    import textwrap

    ref_shims.install_stub_finder(("controlnet_aux", "PIL", "imageio", "librosa", "soundfile", "pytorch_wavelets", "torchdiffeq", "gguf",
                                   "huggingface_hub", "accelerate", "flatten_json", "pytorch_fid", "clip", "scipy", "tqdm", "yaml",
                                   "ftfy", "sentencepiece", "omegaconf", "moviepy", "decord"))
    from toolkit.models.base_model import BaseModel

    def describe(cls, own_only):
        out = {}
        for name, fn in (vars(cls).items() if own_only else inspect.getmembers(cls)):
            if isinstance(fn, staticmethod):
                fn, kind = fn.__func__, "static"
            elif isinstance(fn, property):
                out[name] = {"kind": "property", "settable": fn.fset is not None}
                continue
            else:
                kind = "method"
            if not inspect.isfunction(fn) or (name.startswith("__") and name != "__init__"):
                continue
            sig = inspect.signature(fn)
            params = [{"name": p.name, "kind": p.kind.name, "has_default": p.default is not inspect._empty} for p in sig.parameters.values()]
            try:
                body = textwrap.dedent(inspect.getsource(fn))
            except OSError:
                body = ""
            must = "raise NotImplementedError" in body and body.count("\n") < 16
            out[name] = {"kind": kind, "params": params, "must_implement": bool(must)}
        return out

    contract = {"BaseModel": describe(BaseModel, own_only=True), "BaseModel_arch_default": BaseModel.arch}
    from extensions_built_in.diffusion_models.flux_kontext.flux_kontext import FluxKontextModel
    from toolkit.models.wan21.wan21 import Wan21

    for cls in (FluxKontextModel, Wan21):
        contract[cls.__name__] = {"arch": cls.arch, "bases": [b.__name__ for b in cls.__mro__[1:-1]], "defines": describe(cls, own_only=True)}
    import toolkit.util.get_model as gm

    src = inspect.getsource(gm.get_all_models) + inspect.getsource(gm.get_model_class)
    contract["registration"] = {"module_attribute": "AI_TOOLKIT_MODELS" if "AI_TOOLKIT_MODELS" in src else None,
                                "selected_by": "arch" if "ModelClass.arch == config.arch" in src else None,
                                "extension_folders": ["extensions", "extensions_built_in"] if "extensions_built_in" in src else None}
    with open(os.path.join(HERE, "base_model_contract.json"), "w") as f:
        json.dump(contract, f, indent=1, sort_keys=True)
    must = sorted(k for k, v in contract["BaseModel"].items() if v.get("must_implement"))
    print("base_model_contract.json: BaseModel methods", len(contract["BaseModel"]), "must-implement:", must)



def golden_plugin_registration():
    """integration/extensions/aitk_mi355 executed against the reference's OWN classes (under the shims): every entry of its AI_TOOLKIT_MODELS
    is a real `BaseModel` subclass whose hooks resolve to the MI355X mirror, it can be constructed with the reference's ModelConfig, and
    the reference's selection logic (toolkit/util/get_model.py:44-50: first class whose `arch` equals `config.arch`) returns it."""
    import importlib.util

    ref_shims.install_stub_finder(("controlnet_aux", "PIL", "imageio", "librosa", "soundfile", "pytorch_wavelets", "torchdiffeq", "gguf",
                                   "huggingface_hub", "accelerate", "flatten_json", "pytorch_fid", "clip", "scipy", "tqdm", "yaml",
                                   "ftfy", "sentencepiece", "omegaconf", "moviepy", "decord"))
    import toolkit.util.get_model as gm
    from toolkit.config_modules import ModelConfig
    from toolkit.models.base_model import BaseModel

    spec = importlib.util.spec_from_file_location("aitk_mi355_ext", os.path.join(ROOT, "integration", "extensions", "aitk_mi355", "__init__.py"))
    ext = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ext)
    builtin = list(gm.BUILT_IN_MODELS)
    gm.get_all_models = lambda: builtin + list(ext.AI_TOOLKIT_MODELS)
    out = {"classes": []}
    for cls in ext.AI_TOOLKIT_MODELS:
        cfg = ModelConfig(name_or_path="/nonexistent", arch=cls.arch)
        picked = gm.get_model_class(cfg)
        obj = cls("cpu", cfg, dtype="bf16")
        hooks = {}
        for h in ("load_model", "get_noise_prediction", "get_model_has_grad", "get_te_has_grad", "get_prompt_embeds", "save_model",
                  "get_generation_pipeline", "generate_single_image", "get_loss_target", "encode_images", "get_train_scheduler",
                  "prepare_optimizer_params", "set_device_state_preset", "get_bucket_divisibility"):
            fn = getattr(cls, h)
            owner = next(k.__name__ for k in cls.__mro__ if h in vars(k))
            hooks[h] = owner
        out["classes"].append({"name": cls.__name__, "arch": cls.arch, "is_BaseModel_subclass": issubclass(cls, BaseModel),
                               "selected_by_get_model_class": picked is cls, "mro": [k.__name__ for k in cls.__mro__],
                               "constructed": type(obj).__name__, "torch_dtype": str(obj.torch_dtype), "hook_owner": hooks,
                               # read off the INSTANCE, after BaseModel.__init__ (which resets them, base_model.py:158-185) and the mirror's
                               # constructor: what BaseSDTrainProcess.py:1976 / 1699 / 1730 and lora_special.py:412-423 actually see
                               "instance_flags": {"is_flow_matching": bool(obj.is_flow_matching), "is_transformer": bool(obj.is_transformer),
                                                  "use_old_lokr_format": bool(obj.use_old_lokr_format)}})
    with open(os.path.join(HERE, "plugin_registration.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("plugin_registration.json:", [(c["name"], c["is_BaseModel_subclass"], c["selected_by_get_model_class"]) for c in out["classes"]])


ADOPT_CFG = dict(in_channels=64, num_layers=2, num_single_layers=2, attention_head_dim=128, num_attention_heads=2,
                 joint_attention_dim=64, pooled_projection_dim=32)


def adoption_batches(n, seed=9, B=2, Hl=8, Wl=4, n_txt=6):
    """the synthetic batches of the adoption run (shared with tests/test_adoption_cpu.py, which imports this function's twin)"""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        lat = torch.randn(B, 16, Hl, Wl, generator=g)
        emb = torch.randn(B, n_txt, ADOPT_CFG["joint_attention_dim"], generator=g) * 0.5
        pooled = torch.randn(B, ADOPT_CFG["pooled_projection_dim"], generator=g) * 0.5
        target = torch.randn(B, 16, Hl, Wl, generator=g)
        out.append((lat, emb, pooled, torch.tensor([700.0, 250.0]), target))
    return out


def golden_adoption(out_dir=None):
    """THE boundary test vector: the reference's OWN trainer sequence (jobs/process/BaseSDTrainProcess.py:1932-2007 network construction with
    its literal keyword set, force_to, `sd.network = network`, apply_to, prepare_grad_etc, prepare_optimizer_params; then
    extensions_built_in/sd_trainer/SDTrainer.py:2243-2293: zero_grad, `with network:` prediction through the plug-in, mse, backward,
    clip_grad_norm_, torch.optim.AdamW(eps=1e-6), toolkit/ema.py ExponentialMovingAverage) executed with the reference's own
    LoRASpecialNetwork / LoRAModule / DoRAModule / LokrModule / ExponentialMovingAverage classes over the REAL plug-in class of
    integration/extensions/aitk_mi355 wrapping a native FluxTransformer2DModel (oracle kernel table, fp32, CPU).  Nothing of the
    reference is patched: the native model adopts the network the reference built (ai_toolkit_amd/adopt.py).  Recorded: per-step losses, the
    gradients of the last step, final parameters, EMA shadows, the state dict the reference's get_state_dict returns and the file its
    save_weights writes (model hash).  tests/test_adoption_cpu.py holds a FusedLoRANetwork twin to these bit for bit."""
    import importlib.util
    import tempfile
    from collections import OrderedDict
    from types import SimpleNamespace

    from safetensors import safe_open

    ref_shims.install_stub_finder(("controlnet_aux", "PIL", "imageio", "librosa", "soundfile", "pytorch_wavelets", "torchdiffeq", "gguf",
                                   "huggingface_hub", "accelerate", "flatten_json", "pytorch_fid", "clip", "scipy", "tqdm", "yaml",
                                   "ftfy", "sentencepiece", "omegaconf", "moviepy", "decord"))
    import types

    sys.modules.setdefault("info", types.SimpleNamespace(software_meta={"name": "ai-toolkit"}))
    from toolkit.config_modules import ModelConfig, NetworkConfig
    from toolkit.ema import ExponentialMovingAverage
    from toolkit.lora_special import LoRASpecialNetwork

    from ai_toolkit_amd.adopt import AdoptedNetwork, AdoptionError
    from ai_toolkit_amd.flux import FluxTransformer2DModel
    from oracle import ref_ops

    spec = importlib.util.spec_from_file_location("aitk_mi355_ext", os.path.join(ROOT, "integration", "extensions", "aitk_mi355", "__init__.py"))
    ext = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ext)

    def plugin():
        torch.manual_seed(0)
        ref = flux_ref.FluxTransformer2DModel(**ADOPT_CFG)
        flux_ref.init_synthetic_(ref, seed=1234, std=0.05)
        nat = FluxTransformer2DModel(**ADOPT_CFG, dtype=torch.float32, device="cpu", ops=ref_ops)
        nat.load_state_dict(ref.state_dict(), strict=True)
        nat.prepare()
        cfg = ModelConfig(name_or_path="/nonexistent", arch="flux_mi355")
        return cfg, ext.Flux1MI355("cpu", cfg, dtype="fp32", model=nat), nat

    def build_network(cfg, sd, ncfg, **over):
        """jobs/process/BaseSDTrainProcess.py:1932-1978, keyword for keyword"""
        network_kwargs = dict(ncfg.network_kwargs)
        is_lorm = ncfg.type.lower() == "lorm"
        if hasattr(sd, "target_lora_modules"):
            network_kwargs["target_lin_modules"] = sd.target_lora_modules
        kw = dict(text_encoder=None, unet=sd.get_model_to_train(), lora_dim=ncfg.linear, multiplier=1.0, alpha=ncfg.linear_alpha, train_unet=True,
                  train_text_encoder=False, conv_lora_dim=ncfg.conv, conv_alpha=ncfg.conv_alpha, is_sdxl=cfg.is_xl or cfg.is_ssd, is_v2=cfg.is_v2,
                  is_v3=cfg.is_v3, is_pixart=cfg.is_pixart, is_auraflow=cfg.is_auraflow, is_flux=cfg.is_flux, is_lumina2=cfg.is_lumina2,
                  is_ssd=cfg.is_ssd, is_vega=cfg.is_vega, dropout=ncfg.dropout, use_text_encoder_1=cfg.use_text_encoder_1,
                  use_text_encoder_2=cfg.use_text_encoder_2, use_bias=is_lorm, is_lorm=is_lorm, network_config=ncfg, network_type=ncfg.type,
                  transformer_only=ncfg.transformer_only, is_transformer=sd.is_transformer, base_model=sd, **network_kwargs)
        kw.update(over)
        return LoRASpecialNetwork(**kw)

    def run(tag, ncfg, steps, out, multiplier=None, warm=False):
        cfg, sd, nat = plugin()
        assert sd.is_transformer and sd.is_flow_matching and not sd.use_old_lokr_format
        torch.manual_seed(99)
        net = build_network(cfg, sd, ncfg)
        net.force_to(torch.device("cpu"), dtype=torch.float32)   # 1983
        sd.network = net                                         # 1985
        net._update_torch_multiplier()                           # 1986
        net.apply_to(None, sd.unet, False, True)                 # 1988-1993
        net.prepare_grad_etc(None, sd.unet)                      # 2021
        if warm:  # non-zero lora_up / w2 so that every gradient family is exercised from step 1 (load_state_dict: in place, like a resume)
            g = torch.Generator().manual_seed(7)
            warm_sd = {k: (torch.randn(v.shape, generator=g) * 0.05 if ("lora_up" in k or "lokr_w2" in k) else v.clone())
                       for k, v in net.state_dict().items()}
            net.load_state_dict(warm_sd)
        params = net.prepare_optimizer_params(text_encoder_lr=1e-3, unet_lr=1e-3, default_lr=1e-3)  # 2027-2039
        plist = [p for grp in params for p in grp["params"]]
        opt = torch.optim.AdamW(params, lr=1e-3, eps=1e-6, weight_decay=0.01)  # toolkit/optimizer.py:78-79
        ema = ExponentialMovingAverage(plist, decay=0.99)                       # BaseSDTrainProcess.py:798-803
        if multiplier is not None:
            net.multiplier = multiplier
        losses = []
        for k, (lat, emb, pooled, ts, target) in enumerate(adoption_batches(steps)):
            pe = SimpleNamespace(text_embeds=emb, pooled_embeds=pooled)
            opt.zero_grad()
            with net:
                pred = sd.get_noise_prediction(lat, ts, pe, guidance_embedding_scale=1.0, bypass_guidance_embedding=False)
                loss = torch.nn.functional.mse_loss(pred.float(), target.float(), reduction="none").mean([1, 2, 3]).mean()
                loss.backward()
            if k == steps - 1:
                for i, p in enumerate(plist):
                    out[f"{tag}/last_grad/{i}"] = p.grad.detach().clone()
            torch.nn.utils.clip_grad_norm_(plist, 1.0)
            opt.step()
            opt.zero_grad(set_to_none=True)
            ema.update()
            losses.append(loss.detach().clone())
        assert isinstance(nat.network, AdoptedNetwork) and nat.network.foreign is net and nat.network.aliasing_intact()
        out[f"{tag}/losses"] = torch.stack(losses)
        for i, p in enumerate(plist):
            out[f"{tag}/param/{i}"] = p.detach().clone()
            out[f"{tag}/ema/{i}"] = ema.shadow_params[i].detach().clone()
        sdict = net.get_state_dict(dtype=torch.float32)
        for k2, v in sdict.items():
            out[f"{tag}/saved/{k2}"] = v.clone()
        with tempfile.TemporaryDirectory() as td:
            f = os.path.join(td, "lora.safetensors")
            net.save_weights(f, dtype=torch.float16, metadata=OrderedDict(name="adoption", step=str(steps), format="pt"))
            with safe_open(f, "pt") as fh:
                fmeta = dict(fh.metadata())
                fkeys = list(fh.keys())
            if tag == "lora":
                # resume (BaseSDTrainProcess.py:2060-2066): a FRESH plug-in + network, the reference's own load_weights(file) BEFORE the first
                # forward (values land in the reference's tensors, adoption then takes them over), same rank and a rank-8 file into a rank-4
                # network (shrink: network_mixins.py:737-775)
                lat, emb, pooled, ts, _ = adoption_batches(1, seed=21)[0]
                pe = SimpleNamespace(text_embeds=emb, pooled_embeds=pooled)
                for sub, rank in (("loaded", 8), ("loaded_shrunk", 4)):
                    cfg2, sd2, nat2 = plugin()
                    torch.manual_seed(123)
                    net2 = build_network(cfg2, sd2, NetworkConfig(type="lora", linear=rank, linear_alpha=rank, transformer_only=True))
                    net2.force_to(torch.device("cpu"), dtype=torch.float32)
                    sd2.network = net2
                    net2._update_torch_multiplier()
                    net2.apply_to(None, sd2.unet, False, True)
                    net2.prepare_grad_etc(None, sd2.unet)
                    extra = net2.load_weights(f)
                    assert extra is None
                    with torch.no_grad(), net2:
                        out[f"{tag}/pred_{sub}"] = sd2.get_noise_prediction(lat, ts, pe, 1.0, False).clone()
                    assert isinstance(nat2.network, AdoptedNetwork) and nat2.network.aliasing_intact()
        # the adapter-inactive prediction == base model (no_grad: sampling / prior prediction path)
        lat, emb, pooled, ts, _ = adoption_batches(1, seed=21)[0]
        with torch.no_grad():
            out[f"{tag}/pred_inactive"] = sd.get_noise_prediction(lat, ts, SimpleNamespace(text_embeds=emb, pooled_embeds=pooled), 1.0, False).clone()
            with net:
                out[f"{tag}/pred_active"] = sd.get_noise_prediction(lat, ts, SimpleNamespace(text_embeds=emb, pooled_embeds=pooled), 1.0, False).clone()
        if tag == "lora":
            # the reference's own merge_in / merge_out on the adopted network (toolkit/network_mixins.py:370-462, 894-906): every wrapped layer's
            # weight is rewritten through org_module.load_state_dict, which the native Linear answers by refreshing the transposed copy its
            # data-gradient GEMM reads; merged + skipped adapter == active adapter, merge_out restores the base
            with torch.no_grad():
                net.merge_in(1.0)
                assert net.is_merged_in
                with net:
                    out[f"{tag}/pred_merged"] = sd.get_noise_prediction(lat, ts, SimpleNamespace(text_embeds=emb, pooled_embeds=pooled), 1.0, False).clone()
                lin = nat.single_transformer_blocks[0].attn.to_k
                assert torch.equal(lin.weight_t, lin.weight.data.t()), "weight_t must follow a merged weight"
                net.merge_out(1.0)
                out[f"{tag}/pred_after_merge_out"] = sd.get_noise_prediction(lat, ts, SimpleNamespace(text_embeds=emb, pooled_embeds=pooled), 1.0, False).clone()
            e_m = ((out[f"{tag}/pred_merged"] - out[f"{tag}/pred_active"]).norm() / out[f"{tag}/pred_active"].norm()).item()
            e_o = ((out[f"{tag}/pred_after_merge_out"] - out[f"{tag}/pred_inactive"]).norm() / out[f"{tag}/pred_inactive"].norm()).item()
            assert e_m < 1e-5 and e_o < 1e-5, (e_m, e_o)
        return {"names": [m.lora_name for m in net.unet_loras], "saved_keys": list(sdict.keys()), "n_params": len(plist),
                "file_keys": fkeys, "sshs_model_hash": fmeta.get("sshs_model_hash"), "sshs_legacy_hash": fmeta.get("sshs_legacy_hash"),
                "peft_format": bool(net.peft_format), "module_class": type(net.unet_loras[0]).__name__, "steps": steps}

    out, meta = {}, {}
    meta["lora"] = run("lora", NetworkConfig(type="lora", linear=8, linear_alpha=8, transformer_only=True), 3, out)
    meta["lora_mvec"] = run("lora_mvec", NetworkConfig(type="lora", linear=4, linear_alpha=4), 2, out, multiplier=[0.5, -1.5], warm=True)
    meta["dora"] = run("dora", NetworkConfig(type="dora", linear=4, linear_alpha=4), 2, out, warm=True)
    meta["lokr"] = run("lokr", NetworkConfig(type="lokr", lokr_full_rank=True, lokr_factor=-1), 2, out, warm=True)
    meta["lokr_lowrank"] = run("lokr_lowrank", NetworkConfig(type="lokr", lokr_full_rank=False, linear=4, linear_alpha=4, lokr_factor=-1), 2, out, warm=True)

    # ---- the UNet (kohya-format) branch with network.conv over the StableDiffusion-style plug-in: Conv2d-shaped adapters
    def run_unet(tag, xl, out):
        from ai_toolkit_amd.unet import UNet2DConditionModel
        from oracle import unet_ref
        from tests.test_unet_cpu import TINY_SD15, TINY_SDXL

        ucfg = TINY_SDXL if xl else TINY_SD15
        torch.manual_seed(0)
        ref = unet_ref.UNet2DConditionModel(**ucfg)
        unet_ref.init_synthetic_(ref, seed=11)
        nat = UNet2DConditionModel(**ucfg, dtype=torch.float32, device="cpu", ops=ref_ops)
        nat.load_state_dict(ref.state_dict(), strict=True)
        nat.prepare()
        cfg = ModelConfig(name_or_path="/nonexistent", arch="sd_mi355")
        sd = ext.StableDiffusionMI355("cpu", cfg, dtype="fp32", model=nat, is_xl=xl)
        assert not sd.is_transformer and not sd.is_flow_matching
        ncfg = NetworkConfig(type="lora", linear=4, linear_alpha=2.0, conv=2, conv_alpha=1.0)
        torch.manual_seed(99)
        net = build_network(cfg, sd, ncfg, is_sdxl=xl)
        net.force_to(torch.device("cpu"), dtype=torch.float32)
        sd.network = net
        net._update_torch_multiplier()
        net.apply_to(None, sd.unet, False, True)
        net.prepare_grad_etc(None, sd.unet)
        g = torch.Generator().manual_seed(7)
        with torch.no_grad():
            for m in net.unet_loras:
                m.lora_up.weight.copy_((torch.randn(m.lora_up.weight.shape[:2], generator=g) * 0.05).reshape(m.lora_up.weight.shape))
        params = net.prepare_optimizer_params(text_encoder_lr=1e-3, unet_lr=1e-3, default_lr=1e-3)
        plist = [p for grp in params for p in grp["params"]]
        opt = torch.optim.AdamW(params, lr=1e-3, eps=1e-6, weight_decay=0.01)
        gen = torch.Generator().manual_seed(4)
        B, H, W = 2, 16, 8
        pooled_dim = ucfg["projection_class_embeddings_input_dim"] - 6 * ucfg["addition_time_embed_dim"] if xl else 8
        losses = []
        for k in range(2):
            lat, tgt = torch.randn(B, 4, H, W, generator=gen), torch.randn(B, 4, H, W, generator=gen)
            pe = SimpleNamespace(text_embeds=torch.randn(B, 7, ucfg["cross_attention_dim"], generator=gen),
                                 pooled_embeds=torch.randn(B, pooled_dim, generator=gen))
            opt.zero_grad()
            with net:
                pred = sd.predict_noise(lat, text_embeddings=pe, timestep=torch.tensor([640, 17]))
                loss = torch.nn.functional.mse_loss(pred.float(), tgt.float(), reduction="none").mean([1, 2, 3]).mean()
                loss.backward()
            torch.nn.utils.clip_grad_norm_(plist, 1.0)
            opt.step()
            opt.zero_grad(set_to_none=True)
            losses.append(loss.detach().clone())
        assert isinstance(nat.network, AdoptedNetwork) and nat.network.aliasing_intact()
        out[f"{tag}/losses"] = torch.stack(losses)
        for i, p in enumerate(plist):
            out[f"{tag}/param/{i}"] = p.detach().clone()
        sdict = net.get_state_dict(dtype=torch.float32)
        for k2, v in sdict.items():
            out[f"{tag}/saved/{k2}"] = v.clone()
        return {"names": [m.lora_name for m in net.unet_loras], "saved_keys": list(sdict.keys()), "n_params": len(plist),
                "peft_format": bool(net.peft_format), "module_class": type(net.unet_loras[0]).__name__,
                "n_conv3x3": sum(1 for m in net.unet_loras if tuple(m.lora_down.weight.shape[2:]) == (3, 3)),
                "n_conv1x1": sum(1 for m in net.unet_loras if tuple(m.lora_down.weight.shape[2:]) == (1, 1))}

    # ---- Wan2.1 (BASELINE config 4): block filter from the plug-in's get_transformer_block_names(), keys converted to the original repo's names
    # on save through the plug-in's convert_lora_weights_before_save hook (toolkit/models/wan21/wan21.py:726-730)
    def run_wan(tag, out):
        from ai_toolkit_amd.wan import WanTransformer3DModel
        from oracle import wan_ref
        from tests.test_wan_cpu import CFG as WCFG

        torch.manual_seed(0)
        ref = wan_ref.WanTransformer3DModel(**WCFG)
        wan_ref.init_synthetic_(ref, seed=99, std=0.05)
        nat = WanTransformer3DModel(**WCFG, dtype=torch.float32, device="cpu", ops=ref_ops)
        nat.load_state_dict(ref.state_dict(), strict=True)
        nat.prepare()
        cfg = ModelConfig(name_or_path="/nonexistent", arch="wan21_mi355")
        sd = ext.Wan21MI355("cpu", cfg, dtype="fp32", model=nat)
        ncfg = NetworkConfig(type="lora", linear=8, linear_alpha=8)
        torch.manual_seed(99)
        net = build_network(cfg, sd, ncfg)
        net.force_to(torch.device("cpu"), dtype=torch.float32)
        sd.network = net
        net._update_torch_multiplier()
        net.apply_to(None, sd.unet, False, True)
        net.prepare_grad_etc(None, sd.unet)
        g = torch.Generator().manual_seed(7)
        with torch.no_grad():
            for m in net.unet_loras:
                m.lora_up.weight.copy_(torch.randn(m.lora_up.weight.shape, generator=g) * 0.05)
        params = net.prepare_optimizer_params(text_encoder_lr=1e-3, unet_lr=1e-3, default_lr=1e-3)
        plist = [p for grp in params for p in grp["params"]]
        opt = torch.optim.AdamW(params, lr=1e-3, eps=1e-6, weight_decay=0.01)
        gen = torch.Generator().manual_seed(3)
        losses = []
        for k in range(2):
            lat, tgt = torch.randn(2, 16, 3, 8, 4, generator=gen), torch.randn(2, 16, 3, 8, 4, generator=gen)
            pe = SimpleNamespace(text_embeds=torch.randn(2, 5, WCFG["text_dim"], generator=gen), pooled_embeds=None)
            opt.zero_grad()
            with net:
                pred = sd.get_noise_prediction(lat, torch.tensor([310.0, 845.0]), pe)
                loss = torch.nn.functional.mse_loss(pred.float(), tgt.float(), reduction="none").mean([1, 2, 3, 4]).mean()
                loss.backward()
            torch.nn.utils.clip_grad_norm_(plist, 1.0)
            opt.step()
            opt.zero_grad(set_to_none=True)
            losses.append(loss.detach().clone())
        assert isinstance(nat.network, AdoptedNetwork) and nat.network.aliasing_intact()
        out[f"{tag}/losses"] = torch.stack(losses)
        for i, p in enumerate(plist):
            out[f"{tag}/param/{i}"] = p.detach().clone()
        sdict = net.get_state_dict(dtype=torch.float32)
        for k2, v in sdict.items():
            out[f"{tag}/saved/{k2}"] = v.clone()
        return {"names": [m.lora_name for m in net.unet_loras], "saved_keys": list(sdict.keys()), "n_params": len(plist),
                "peft_format": bool(net.peft_format), "module_class": type(net.unet_loras[0]).__name__}

    meta["wan"] = run_wan("wan", out)
    meta["unet_sd15_conv"] = run_unet("unet_sd15_conv", False, out)
    meta["unet_sdxl_conv"] = run_unet("unet_sdxl_conv", True, out)

    # what cannot be adopted raises where the reference attaches it (apply_to), never a base-only model
    refused = {}
    for tag, ncfg, over in (("lorm_use_bias", NetworkConfig(type="lora", linear=4, linear_alpha=4), dict(use_bias=True)),
                            ("fullrank", NetworkConfig(type="fullrank", linear=4, linear_alpha=4), {}),
                            ("full_if_contains", NetworkConfig(type="lora", linear=4, linear_alpha=4), dict(full_if_contains=["attn.to_q"]))):
        cfg, sd, nat = plugin()
        net = build_network(cfg, sd, ncfg, **over)
        net.force_to(torch.device("cpu"), dtype=torch.float32)
        try:
            sd.network = net
            net._update_torch_multiplier()
            net.apply_to(None, sd.unet, False, True)
            refused[tag] = "NOT REFUSED"
        except (AdoptionError, TypeError) as e:
            refused[tag] = type(e).__name__
    meta["refused"] = refused
    assert all(v != "NOT REFUSED" for v in refused.values()), refused
    out_dir = out_dir or HERE
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(out_dir, "adoption_flux_tiny.safetensors"),
              {"meta": json.dumps(meta, sort_keys=True)})
    print("adoption golden:", {k: (v if k == "refused" else (v["module_class"], v["n_params"], (v.get("sshs_model_hash") or "")[:12])) for k, v in meta.items()})


def golden_flux_glue():
    """Pins the FLUX glue that only diffusers' (un-vendored) classes spell out — adaLN chunk order, the final layer's (scale, shift) order, the
    timestep-embedding layout and the embedder composition — against code the reference DOES hold:

      * scripts/convert_diffusers_to_comfy.py: its `diffusers_map` (diffusers key -> BFL key, which tensors are concatenated in which order) and
        `swap_scale_shift` — extracted from the script by AST and EXECUTED (the script itself is a CLI that runs at import);
      * extensions_built_in/diffusion_models/flux2/src/model.py: the in-tree BFL-lineage `Modulation` (lin(silu(vec)).chunk -> shift, scale, gate
        [x2]), `LastLayer` (shift, scale = chunk(2); (1 + scale) * norm(x) + shift), `MLPEmbedder`, `timestep_embedding`;
      * toolkit/models/flux.py: `guidance_embed_bypass_forward` (conditioning = timestep_embedder(time_proj(t)) + text_embedder(pooled)).

    Recorded: inputs, the weights in diffusers naming, and the outputs of the REFERENCE-side modules loaded through the converter's mapping.
    tests/test_flux_glue_golden.py runs the oracle's diffusers-named restatement (and the native graph's pieces) on the same inputs."""
    import ast
    import importlib.util

    src = open(os.path.join(ref_shims.REFERENCE, "scripts", "convert_diffusers_to_comfy.py")).read()
    tree = ast.parse(src)
    ns = {"torch": torch}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == "swap_scale_shift":
            exec(compile(ast.Module([node], []), "convert_diffusers_to_comfy.py", "exec"), ns)
        if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id == "diffusers_map" for t in node.targets):
            exec(compile(ast.Module([node], []), "convert_diffusers_to_comfy.py", "exec"), ns)
    dmap, swap = ns["diffusers_map"], ns["swap_scale_shift"]
    spec = importlib.util.spec_from_file_location("flux2_model_ref", os.path.join(ref_shims.REFERENCE, "extensions_built_in", "diffusion_models", "flux2", "src", "model.py"))
    f2 = importlib.util.module_from_spec(spec)
    sys.modules["flux2_model_ref"] = f2
    spec.loader.exec_module(f2)
    from toolkit.models.flux import guidance_embed_bypass_forward

    d, B, S = 64, 2, 5
    g = torch.Generator().manual_seed(123)
    rnd = lambda *sh: torch.randn(*sh, generator=g)  # noqa: E731
    out = {}
    # ---- the map itself: which diffusers tensors make which BFL tensor, in which order
    meta = {"img_mod": dmap["double_blocks.().img_mod.lin.weight"], "txt_mod": dmap["double_blocks.().txt_mod.lin.weight"],
            "single_mod": dmap["single_blocks.().modulation.lin.weight"], "img_qkv": dmap["double_blocks.().img_attn.qkv.weight"],
            "txt_qkv": dmap["double_blocks.().txt_attn.qkv.weight"], "single_linear1": dmap["single_blocks.().linear1.weight"],
            "final_mod": dmap["final_layer.adaLN_modulation.1.weight"], "final_linear": dmap["final_layer.linear.weight"],
            "time_in": [dmap["time_in.in_layer.weight"], dmap["time_in.out_layer.weight"]],
            "vector_in": [dmap["vector_in.in_layer.weight"], dmap["vector_in.out_layer.weight"]],
            "guidance_in": [dmap["guidance_in.in_layer.weight"], dmap["guidance_in.out_layer.weight"]]}
    # ---- AdaLayerNormZero (norm1 / norm1_context) == BFL Modulation(double) with the SAME weight (the converter copies it un-swapped)
    vec, x = rnd(B, d), rnd(B, S, d)
    w6, b6 = rnd(6 * d, d) * 0.2, rnd(6 * d) * 0.1
    mod = f2.Modulation(d, double=True)
    with torch.no_grad():
        mod.lin.weight.copy_(w6)
        mod.lin.bias.copy_(b6)
        (sh1, sc1, g1), (sh2, sc2, g2) = mod(vec)
        ln = torch.nn.functional.layer_norm(x, (d,), eps=1e-6)
        out.update({"zero/vec": vec, "zero/x": x, "zero/linear.weight": w6, "zero/linear.bias": b6, "zero/x_mod": (1 + sc1) * ln + sh1,
                    "zero/gate_msa": g1[:, 0], "zero/shift_mlp": sh2[:, 0], "zero/scale_mlp": sc2[:, 0], "zero/gate_mlp": g2[:, 0]})
        # ---- AdaLayerNormZeroSingle == Modulation(single)
        w3, b3 = rnd(3 * d, d) * 0.2, rnd(3 * d) * 0.1
        ms = f2.Modulation(d, double=False)
        ms.lin.weight.copy_(w3)
        ms.lin.bias.copy_(b3)
        (sh, sc, gt), _none = ms(vec)
        out.update({"single/linear.weight": w3, "single/linear.bias": b3, "single/x_mod": (1 + sc) * ln + sh, "single/gate": gt[:, 0]})
        # ---- norm_out + proj_out == LastLayer with adaLN weight = swap_scale_shift(norm_out.linear.weight)
        w2, wp = rnd(2 * d, d) * 0.2, rnd(16, d) * 0.2
        last = f2.LastLayer(d, 16)
        last.adaLN_modulation[1].weight.copy_(swap(w2))
        last.linear.weight.copy_(wp)
        out.update({"final/norm_out.linear.weight": w2, "final/proj_out.weight": wp, "final/out": last(x, vec)})
        # ---- timestep embedding + embedder MLP (time_in <- time_text_embed.timestep_embedder.linear_1 / linear_2)
        t = torch.tensor([0.3, 0.8])
        out["temb/t"] = t
        out["temb/proj"] = f2.timestep_embedding(t, 256)  # time_factor 1000 inside: diffusers is handed t * 1000
        emb = f2.MLPEmbedder(256, d)
        for p_ in emb.parameters():
            p_.copy_(rnd(*p_.shape) * 0.1)
        out.update({"temb/linear_1.weight": emb.in_layer.weight.detach().clone(), "temb/linear_1.bias": emb.in_layer.bias.detach().clone(),
                    "temb/linear_2.weight": emb.out_layer.weight.detach().clone(), "temb/linear_2.bias": emb.out_layer.bias.detach().clone(),
                    "temb/out": emb(out["temb/proj"])})
        # ---- the embedder composition of the bypass path, executed on the oracle's diffusers-named module by the reference's own function
        torch.manual_seed(5)
        tte = flux_ref.CombinedTimestepGuidanceTextProjEmbeddings(d, 32)
        pooled = rnd(B, 32)
        out["bypass/pooled"] = pooled
        for k, v in tte.state_dict().items():
            out[f"bypass/sd/{k}"] = v.clone()
        out["bypass/conditioning"] = guidance_embed_bypass_forward(tte, t * 1000, None, pooled)
    save_file({k: v.detach().contiguous() for k, v in out.items()}, os.path.join(HERE, "flux_glue.safetensors"), {"meta": json.dumps(meta, sort_keys=True)})
    print("flux glue golden:", len(out), "tensors;", {k: v for k, v in meta.items() if k in ("img_qkv", "single_linear1", "final_mod")})


class HashTokenizer:
    """Deterministic stand-in for a `transformers` tokenizer (no vocabulary files in this image): words -> ids by a fixed hash, padded /
    truncated to max_length, with the call signature and return fields encode_prompts_flux uses.  Shared with tests/test_text_encoders_cpu.py."""

    def __init__(self, vocab_size, model_max_length, pad_id=0, eos_id=1):
        self.vocab_size, self.model_max_length, self.pad_id, self.eos_id = vocab_size, model_max_length, pad_id, eos_id

    def __call__(self, prompts, padding=None, max_length=None, truncation=True, return_tensors="pt", **kw):
        import hashlib
        from types import SimpleNamespace

        ids, mask = [], []
        for p in prompts:
            toks = [2 + int(hashlib.sha256(w.encode()).hexdigest(), 16) % (self.vocab_size - 2) for w in p.split()][: max_length - 1] + [self.eos_id]
            mask.append([1] * len(toks) + [0] * (max_length - len(toks)))
            ids.append(toks + [self.pad_id] * (max_length - len(toks)))
        out = SimpleNamespace(input_ids=torch.tensor(ids), attention_mask=torch.tensor(mask))
        out.__getitem__ = None
        return _TokOut(out.input_ids, out.attention_mask)


class _TokOut(dict):
    def __init__(self, input_ids, attention_mask):
        super().__init__(input_ids=input_ids, attention_mask=attention_mask)
        self.input_ids, self.attention_mask = input_ids, attention_mask


def tiny_text_encoders(seed=31, t5_dim=24):
    from transformers import CLIPTextConfig, CLIPTextModel, T5Config, T5EncoderModel

    torch.manual_seed(seed)
    clip = CLIPTextModel(CLIPTextConfig(vocab_size=99, hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2,
                                        max_position_embeddings=16, eos_token_id=1, pad_token_id=0, bos_token_id=2)).eval()
    t5 = T5EncoderModel(T5Config(vocab_size=101, d_model=t5_dim, d_kv=8, d_ff=48, num_layers=2, num_heads=3, is_encoder_decoder=False, use_cache=False)).eval()
    toks = [HashTokenizer(99, 16), HashTokenizer(101, 512)]
    return toks, [clip, t5]


PROMPTS = ["a photo of a red fox in the snow", "", "two words"]


def golden_text_encoders():
    """Row a17 (toolkit/train_tools.py:510-574): the reference's own encode_prompts_flux executed on tiny random CLIP / T5 text encoders of the
    `transformers` classes it uses, with a deterministic stand-in tokenizer — pins the restatement in ai_toolkit_amd/plugin.py (which pads CLIP to
    tokenizer.model_max_length, T5 to 512, takes CLIP's pooler_output and T5's last hidden state, optional attention masking)."""
    sys.modules.setdefault("info", __import__("types").SimpleNamespace(software_meta={"name": "ai-toolkit"}))
    ref_shims.install_stub_finder(("controlnet_aux", "PIL", "imageio", "librosa", "soundfile", "pytorch_wavelets", "torchdiffeq", "gguf",
                                   "huggingface_hub", "accelerate", "flatten_json", "pytorch_fid", "clip", "scipy", "tqdm", "yaml",
                                   "ftfy", "sentencepiece", "omegaconf", "moviepy", "decord"))
    from toolkit.train_tools import encode_prompts_flux

    toks, tes = tiny_text_encoders()
    out = {}
    with torch.no_grad():
        for tag, kw in (("plain", {}), ("masked", {"attn_mask": True}), ("len64", {"max_length": 64})):
            emb, pooled = encode_prompts_flux(toks, tes, list(PROMPTS), **kw)
            out[f"{tag}/embeds"], out[f"{tag}/pooled"] = emb.clone(), pooled.clone()
    # ---- SD1.x / SDXL (toolkit/train_tools.py:192-323, 379-422): CLIP hidden states, long prompts in windows, SDXL's two encoders + pooled
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection

    from toolkit.train_tools import encode_prompts, encode_prompts_xl

    torch.manual_seed(41)
    ccfg = dict(vocab_size=99, intermediate_size=64, num_hidden_layers=3, num_attention_heads=2, max_position_embeddings=16, eos_token_id=1, pad_token_id=0, bos_token_id=2)
    c1 = CLIPTextModel(CLIPTextConfig(hidden_size=32, **ccfg)).eval()
    c2 = CLIPTextModelWithProjection(CLIPTextConfig(hidden_size=48, projection_dim=40, **ccfg)).eval()
    tk = [HashTokenizer(99, 16), HashTokenizer(99, 16)]
    long_prompts = [" ".join(f"w{i}" for i in range(40)), "short one"]
    with torch.no_grad():
        out["sd/plain"] = encode_prompts(tk[0], c1, list(PROMPTS)).clone()
        for tag, kw in (("plain", {}), ("no_te1", {"use_text_encoder_1": False}), ("two_images", {"num_images_per_prompt": 2})):
            e, p_ = encode_prompts_xl(tk, [c1, c2], list(PROMPTS), None, **kw)
            out[f"sdxl/{tag}/embeds"], out[f"sdxl/{tag}/pooled"] = e.clone(), p_.clone()
        e, p_ = encode_prompts_xl(tk, [c1, c2], long_prompts, None, truncate=False, max_length=16 * 4)
        out["sdxl/long/embeds"], out["sdxl/long/pooled"] = e.clone(), p_.clone()
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "text_encoders_flux.safetensors"))
    print("text encoder golden:", {k: tuple(v.shape) for k, v in out.items()})


PARTIAL_ONLY = ["transformer_blocks.0.attn.to_q", "transformer_blocks.0.attn.add_k_proj", "single_transformer_blocks.1.proj_mlp", "single_transformer_blocks.0.attn.to_v",
                "transformer_blocks.1.ff.net.2", "transformer_blocks.1.ff_context.net.0.proj"]


def golden_trainer_loop(out_dir=None, kind="flux", accum=1, dtype="fp32", quantize=False, network="lora", uncached=False, preservation=None):
    """THE END-TO-END BOUNDARY RUN: the reference's real `SDTrainer` (extensions_built_in/sd_trainer/SDTrainer.py) — its `run()`, unmodified —
    trains a LoRA for 3 steps over the plug-in of integration/extensions/aitk_mi355 on CPU: job / process config parsing, `get_model_class`
    picking `flux_mi355`, `ModelClass.get_train_scheduler()`, `sd.load_model()` (native FluxTransformer2DModel streamed from a diffusers-format
    checkpoint directory; oracle kernel table, fp32), the freeze / move / gradient-checkpointing calls on unet / vae / text encoders, its own
    `LoRASpecialNetwork` (adopted by the native graph), `prepare_optimizer_params` -> `toolkit.optimizer.get_optimizer('adamw')`, accelerate,
    EMA, `hook_before_train_loop` (static prompts encoded, text encoders unloaded through toolkit/unloader.py), then per step
    `process_general_training_batch` (its scheduler: timesteps, noise, add_noise), `BaseModel.predict_noise` -> our `get_noise_prediction`,
    `calculate_loss`, `accelerator.backward`, clip, optimizer step, EMA, and its periodic `save()`.
    Stand-ins, all on the DATA side: the dataloader function is replaced by a synthetic iterator of duck-typed batches carrying cached latents
    and cached prompt embeddings (no image files in this image), the text encoders are tiny random `transformers` CLIP / T5 models with the
    hash tokenizer.  Recorded: what every `get_noise_prediction` call received and the loss target the trainer formed, the losses it logged, the
    LoRA file and optimizer state it saved.  tests/test_trainer_loop_cpu.py replays the recorded calls through a FusedLoRANetwork twin."""
    import importlib.util
    import tempfile
    import types
    from collections import OrderedDict

    from safetensors.torch import load_file

    ref_shims.install_stub_finder(("controlnet_aux", "PIL", "imageio", "librosa", "soundfile", "pytorch_wavelets", "torchdiffeq", "gguf",
                                   "huggingface_hub", "flatten_json", "pytorch_fid", "clip", "scipy", "tqdm", "yaml",
                                   "ftfy", "sentencepiece", "omegaconf", "moviepy", "decord"))
    sys.modules.setdefault("info", types.SimpleNamespace(software_meta={"name": "ai-toolkit"}))
    import jobs.process.BaseSDTrainProcess  # noqa: F401
    import toolkit.util.get_model as gm
    from extensions_built_in.sd_trainer.SDTrainer import SDTrainer
    from toolkit.prompt_utils import PromptEmbeds

    bsp = sys.modules["jobs.process.BaseSDTrainProcess"]
    spec = importlib.util.spec_from_file_location("aitk_mi355_ext", os.path.join(ROOT, "integration", "extensions", "aitk_mi355", "__init__.py"))
    ext = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ext)
    builtin = list(gm.BUILT_IN_MODELS)
    gm.get_all_models = lambda: builtin + list(ext.AI_TOOLKIT_MODELS)  # = dropping the package into <ai-toolkit>/extensions/

    from ai_toolkit_amd import loader, plugin
    from ai_toolkit_amd.adopt import AdoptedNetwork
    from ai_toolkit_amd.flux import FluxTransformer2DModel
    from oracle import ref_ops

    plugin._native_ops = lambda: ref_ops  # CPU: the oracle kernel table behind the native graph

    def tiny_te(self, path=None):
        toks, tes = tiny_text_encoders(t5_dim=ADOPT_CFG["joint_attention_dim"])
        self.tokenizer, self.text_encoder = toks, [t.requires_grad_(False) for t in tes]
        return self.text_encoder

    def tiny_umt5(self, path=None):
        from transformers import UMT5Config, UMT5EncoderModel

        torch.manual_seed(31)
        self.text_encoder = UMT5EncoderModel(UMT5Config(vocab_size=101, d_model=48, d_kv=8, d_ff=48, num_layers=2, num_heads=3)).eval().requires_grad_(False)
        self.tokenizer = HashTokenizer(101, 512)
        return self.text_encoder

    def tiny_clip(self, path=None):
        from transformers import CLIPTextConfig, CLIPTextModel

        torch.manual_seed(31)
        self.text_encoder = CLIPTextModel(CLIPTextConfig(vocab_size=99, hidden_size=24, intermediate_size=48, num_hidden_layers=2, num_attention_heads=2,
                                                         max_position_embeddings=16, eos_token_id=1, pad_token_id=0, bos_token_id=2)).eval().requires_grad_(False)
        self.tokenizer = HashTokenizer(99, 16)
        return self.text_encoder

    def tiny_clip_xl(self, path=None):
        from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection

        torch.manual_seed(31)
        ccfg = dict(vocab_size=99, intermediate_size=48, num_hidden_layers=2, num_attention_heads=2, max_position_embeddings=16, eos_token_id=1, pad_token_id=0, bos_token_id=2)
        self.text_encoder = [CLIPTextModel(CLIPTextConfig(hidden_size=8, **ccfg)).eval().requires_grad_(False),
                             CLIPTextModelWithProjection(CLIPTextConfig(hidden_size=16, projection_dim=16, **ccfg)).eval().requires_grad_(False)]
        self.tokenizer = [HashTokenizer(99, 16), HashTokenizer(99, 16)]
        return self.text_encoder

    tag = kind + (f"_accum{accum}" if accum > 1 else "") + (f"_{dtype}" if dtype != "fp32" else "") + ("_fp8base" if quantize else "") + (f"_{network}" if network != "lora" else "") + ("_uncached" if uncached else "") + (f"_{preservation}pp" if preservation else "")
    Plug = {"flux": ext.Flux1MI355, "wan": ext.Wan21MI355, "sd15": ext.StableDiffusionMI355, "sdxl": ext.StableDiffusionMI355}[kind]
    Plug.load_text_encoders = {"flux": tiny_te, "wan": tiny_umt5, "sd15": tiny_clip, "sdxl": tiny_clip_xl}[kind]
    Plug._load_text_side = lambda self, path: self.load_text_encoders(path)

    rec = {"calls": [], "targets": []}
    orig_pred, orig_target = Plug.get_noise_prediction, Plug.get_loss_target

    import functools

    @functools.wraps(orig_pred)  # keeps the signature BaseModel.predict_noise inspects (guidance_embedding_scale / bypass_guidance_embedding)
    def rec_pred(self, latent_model_input, timestep, text_embeddings, **kw):
        pooled = text_embeddings.pooled_embeds
        rec["calls"].append((latent_model_input.detach().clone(), timestep.detach().clone(), text_embeddings.text_embeds.detach().clone(),
                             torch.zeros(0) if pooled is None else pooled.detach().clone(), bool(torch.is_grad_enabled()),
                             {k: v for k, v in kw.items() if isinstance(v, (int, float, bool))}))
        return orig_pred(self, latent_model_input, timestep, text_embeddings, **kw)

    def rec_target(self, *a, **kw):
        t = orig_target(self, *a, **kw)
        rec["targets"].append(t.detach().clone())
        return t

    if kind in ("sd15", "sdxl"):  # the legacy-StableDiffusion mirror answers `predict_noise` itself (stable_diffusion_model.py:1878), one level above
        orig_pn = Plug.predict_noise

        @functools.wraps(orig_pn)
        def rec_pn(self, latents, text_embeddings=None, timestep=1, **kw):
            te = text_embeddings if text_embeddings is not None else kw.get("conditional_embeddings")
            pooled = getattr(te, "pooled_embeds", None)
            rec["calls"].append((latents.detach().clone(), torch.as_tensor(timestep).detach().clone(), te.text_embeds.detach().clone(),
                                 torch.zeros(0) if pooled is None else pooled.detach().clone(), bool(torch.is_grad_enabled()), {k: v for k, v in kw.items() if isinstance(v, (int, float, bool))}))
            return orig_pn(self, latents, text_embeddings=text_embeddings, timestep=timestep, **kw)

        Plug.predict_noise, Plug.get_loss_target = rec_pn, rec_target
    else:
        Plug.get_noise_prediction, Plug.get_loss_target = rec_pred, rec_target

    tmp = tempfile.mkdtemp()
    torch.manual_seed(0)
    if kind == "flux":
        ref = flux_ref.FluxTransformer2DModel(**ADOPT_CFG)
        flux_ref.init_synthetic_(ref, seed=1234, std=0.05)
        nat = FluxTransformer2DModel(**ADOPT_CFG, dtype=torch.float32, device="cpu", ops=ref_ops)
        cfg_json = dict(ADOPT_CFG, guidance_embeds=True, _class_name="FluxTransformer2DModel")
        lat_shape, txt_dim, pooled_dim = (16, 8, 4), ADOPT_CFG["joint_attention_dim"], ADOPT_CFG["pooled_projection_dim"]
    elif kind in ("sd15", "sdxl"):  # SD1.5 / SDXL UNet (BASELINE configs 1, 2): eps-prediction over the DDPM schedule, CLIP hidden states (+ pooled, time ids)
        from ai_toolkit_amd.unet import UNet2DConditionModel
        from oracle import unet_ref
        from tests.test_unet_cpu import TINY_SD15, TINY_SDXL

        ucfg = TINY_SDXL if kind == "sdxl" else TINY_SD15
        ref = unet_ref.UNet2DConditionModel(**ucfg)
        unet_ref.init_synthetic_(ref, seed=5, std=0.05)
        nat = UNet2DConditionModel(**ucfg, dtype=torch.float32, device="cpu", ops=ref_ops)
        cfg_json = dict({k: (list(v) if isinstance(v, tuple) else v) for k, v in ucfg.items()}, _class_name="UNet2DConditionModel")
        lat_shape, txt_dim = (4, 8, 8), ucfg["cross_attention_dim"]
        pooled_dim = ucfg["projection_class_embeddings_input_dim"] - 6 * ucfg["addition_time_embed_dim"] if kind == "sdxl" else None
    else:  # Wan2.1 (BASELINE config 4): video latents [B, 16, F, H, W], UMT5 text states, no pooled vector
        from ai_toolkit_amd.wan import WanTransformer3DModel
        from oracle import wan_ref
        from tests.test_wan_cpu import CFG as WCFG

        ref = wan_ref.WanTransformer3DModel(**WCFG)
        wan_ref.init_synthetic_(ref, seed=99, std=0.05)
        nat = WanTransformer3DModel(**WCFG, dtype=torch.float32, device="cpu", ops=ref_ops)
        cfg_json = dict(WCFG, patch_size=[1, 2, 2], _class_name="WanTransformer3DModel")
        lat_shape, txt_dim, pooled_dim = (16, 3, 8, 4), WCFG["text_dim"], None
    nat.load_state_dict(ref.state_dict())
    if uncached:  # a VAE in the pipeline directory: the trainer gets images and calls the plug-in's encode_images (native AutoencoderKL encoder)
        from ai_toolkit_amd.vae import AutoencoderKLEncoder

        assert kind == "flux"
        vsrc = AutoencoderKLEncoder(latent_channels=16, block_out_channels=(32, 64), layers_per_block=1, dtype=torch.float32, device="cpu", ops=ref_ops)
        torch.manual_seed(11)
        with torch.no_grad():
            for p_ in vsrc.parameters():
                p_.copy_(torch.randn_like(p_) * 0.05)
        loader.save_component(vsrc, os.path.join(tmp, "ckpt", "vae"))
        with open(os.path.join(tmp, "ckpt", "vae", "config.json"), "w") as f:
            json.dump(dict(latent_channels=16, block_out_channels=[32, 64], layers_per_block=1, scaling_factor=0.3611, shift_factor=0.1159, norm_num_groups=32,
                           use_quant_conv=False, _class_name="AutoencoderKL"), f)
        orig_enc = Plug.encode_images

        def rec_enc(self, image_list, *a, **k):
            rng = torch.get_rng_state()  # latent_dist.sample() draws from the global CPU generator: the state in front of the call makes the draw replayable
            out_ = orig_enc(self, image_list, *a, **k)
            rec.setdefault("encode", []).append((torch.stack([im.detach().clone() for im in image_list]) if isinstance(image_list, (list, tuple)) else image_list.detach().clone(),
                                                 out_.detach().clone(), rng.clone()))
            return out_

        Plug.encode_images = rec_enc
    comp = "unet" if kind in ("sd15", "sdxl") else "transformer"
    loader.save_component(nat, os.path.join(tmp, "ckpt", comp))
    with open(os.path.join(tmp, "ckpt", comp, "config.json"), "w") as f:
        json.dump(cfg_json, f)

    class Duck:
        def __init__(self, **kw):
            self.__dict__.update(kw)

        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return None

    class DuckBatch(Duck):  # the attribute set of toolkit/data_transfer_object/data_loader.py DataLoaderBatchDTO the FLUX LoRA step reads
        def __init__(self, B, seed):
            g = torch.Generator().manual_seed(seed)
            pe = PromptEmbeds(torch.randn(B, 6, txt_dim, generator=g) * 0.5)
            if pooled_dim is not None:
                pe.pooled_embeds = torch.randn(B, pooled_dim, generator=g) * 0.5
            # (uncached: images instead of latents; the tiny VAE halves H and W once)
            super().__init__(latents=None if uncached else torch.randn(B, *lat_shape, generator=g),
                             tensor=(torch.rand(B, 3, 2 * lat_shape[1], 2 * lat_shape[2], generator=g) * 2 - 1) if uncached else None, prompt_embeds=pe, loss_multiplier_list=[1.0] * B,
                             file_items=[Duck(path=f"img{i}.png", dataset_config=Duck(), is_reg=False, prior_reg=False, network_weight=1.0) for i in range(B)])
            self.B = B

        def get_caption_list(self, *a, **k):
            return ["a photo"] * self.B

        def get_caption_short_list(self, *a, **k):
            return ["a"] * self.B

        def get_is_reg_list(self):
            return [False] * self.B

        def get_network_weight_list(self):
            return [1.0] * self.B

        def cleanup(self):
            pass

    class DuckLoader:
        dataset = types.SimpleNamespace(datasets=[])

        def __init__(self):
            self.n = 0

        def __iter__(self):
            return self

        def __next__(self):
            self.n += 1
            return DuckBatch(2, 100 + self.n)

        def __len__(self):
            return 1000

    bsp.get_dataloader_from_datasets = lambda *a, **k: DuckLoader()
    bsp.trigger_dataloader_setup_epoch = lambda *a, **k: None
    os.makedirs(os.path.join(tmp, "data"), exist_ok=True)
    steps = 3
    config = OrderedDict(type="sd_trainer", training_folder=os.path.join(tmp, "out"), device="cpu",
                         network={"lora": dict(type="lora", linear=8, linear_alpha=8), "dora": dict(type="dora", linear=8, linear_alpha=8),
                                  "lokr": dict(type="lokr", linear=8, linear_alpha=8),  # lokr_full_rank default: both Kronecker factors full
                                  "lokr_lowrank": dict(type="lokr", linear=4, linear_alpha=4, lokr_full_rank=False),
                                  # adapters on a subset of the Linears: same-input groups (q / k / v, the single blocks' fused projections) with members missing
                                  "lora_partial": dict(type="lora", linear=8, linear_alpha=8, network_kwargs=dict(only_if_contains=PARTIAL_ONLY))}[network],
                         save=dict(dtype="float32", save_every=2, max_step_saves_to_keep=2),
                         datasets=[dict(folder_path=os.path.join(tmp, "data"), cache_latents_to_disk=True, resolution=[64])],
                         train=dict(batch_size=2, steps=steps, gradient_accumulation=accum, train_unet=True, train_text_encoder=False,
                                    gradient_checkpointing=True, noise_scheduler="ddpm" if kind in ("sd15", "sdxl") else "flowmatch", optimizer="adamw", lr=1e-3, dtype=dtype,
                                    disable_sampling=True, skip_first_sample=True, cache_text_embeddings=True,
                                    ema_config=dict(use_ema=True, ema_decay=0.99), timestep_type="sigmoid",
                                    # preservation="blank": train.blank_prompt_preservation (SDTrainer.py:1983-2016, 2182-2219) — per step a prior prediction
                                    # (network off, no_grad, the cached "" embeddings), the training prediction and a SECOND grad-enabled prediction with the
                                    # blank embeddings; loss + multiplier * mse(preservation_pred, prior_pred), one backward through both native graphs
                                    **(dict(blank_prompt_preservation=True, blank_prompt_preservation_multiplier=0.5) if preservation == "blank" else {})),
                         model=dict(arch={"flux": "flux_mi355", "wan": "wan21_mi355", "sd15": "sd_mi355", "sdxl": "sd_mi355"}[kind], **({"is_xl": True} if kind == "sdxl" else {}), name_or_path=os.path.join(tmp, "ckpt"), quantize=quantize),
                         sample=dict(sample_every=10 ** 9, prompts=[]))
    job = types.SimpleNamespace(name="aitk_trainer_run", training_folder=os.path.join(tmp, "out"), device="cpu", meta=OrderedDict(),
                                raw_config={"config": {"name": "aitk_trainer_run"}}, log_dir=None, training_seed=7,
                                config=OrderedDict(name="aitk_trainer_run"), gpu_id=0)
    losses = []
    keep = {}

    def run_trainer(cfg):
        tr = SDTrainer(0, job, cfg)
        orig_loop = tr.hook_train_loop

        def loop(batch):
            if "init" not in keep:  # the adapter as the trainer initialised it (its RNG stream), before the first step
                keep["init"] = OrderedDict((k, v.detach().clone()) for k, v in tr.network.state_dict().items())
            d = orig_loop(batch)
            losses.append(float(d["loss"]))
            keep.update(sd=tr.sd, network=tr.network, ema=tr.ema, step=tr.step_num)  # run() deletes self.sd / self.network on its way out
            return d

        tr.hook_train_loop = loop
        tr.run()
        return tr

    tr = run_trainer(config)
    # ---- RESUME (jobs/process/BaseSDTrainProcess.py:2057-2066, 2190-2215): a second process over the same training folder finds the latest LoRA
    # file, loads it with the network's own load_weights (BEFORE the native graph's first forward: the adoption then takes those values over),
    # restores optimizer.pt and the step count from the file's metadata, and trains on to step 5
    import copy

    resumed = copy.deepcopy(config)
    resumed["train"]["steps"] = steps + 2
    n_before = len(losses)
    tr = run_trainer(resumed)
    assert len(losses) == n_before + 2, losses
    steps = steps + 2
    sd_, net_, ema_ = keep["sd"], keep["network"], keep["ema"]
    assert len(losses) == steps and isinstance(sd_.unet.network, AdoptedNetwork) and sd_.unet.network.aliasing_intact()
    assert sd_.unet.network.foreign is net_ and type(net_).__name__ == "LoRASpecialNetwork"
    # sd15: diffusers is not installed here, toolkit/sampler.py hands back an import stub, and the plug-in falls back to its native DDPM schedule
    # (a real DDPMScheduler is kept: tests/test_plugin_cpu.py); FLUX / Wan train on the reference's own flow-match scheduler object
    assert type(sd_.noise_scheduler).__name__ == ("DDPMTrainSchedule" if kind in ("sd15", "sdxl") else "CustomFlowMatchEulerDiscreteScheduler"), type(sd_.noise_scheduler)
    tes_ = sd_.text_encoder if isinstance(sd_.text_encoder, (list, tuple)) else [sd_.text_encoder]
    assert all(type(t).__name__ == "FakeTextEncoder" for t in tes_)  # unloaded by toolkit/unloader.py after the static prompts
    train_calls = [c for c in rec["calls"] if c[4]]
    out = {"losses": torch.tensor(losses, dtype=torch.float64)}
    if preservation:
        # three calls per micro-batch, in this order: prior (no_grad, network off), training, preservation (both grad-enabled)
        assert len(rec["calls"]) == 3 * steps * accum and [c[4] for c in rec["calls"]] == [False, True, True] * (steps * accum), [c[4] for c in rec["calls"]]
        pres_calls = train_calls[1::2]
        priors = [c for c in rec["calls"] if not c[4]]
        train_calls = train_calls[0::2]
        for i, (pc, qc, tc) in enumerate(zip(pres_calls, priors, train_calls)):
            assert torch.equal(pc[0], tc[0]) and torch.equal(pc[1], tc[1]) and torch.equal(qc[0], tc[0]) and torch.equal(qc[2], pc[2])  # same noisy latents / timesteps; prior and preservation share the embeddings
            out[f"step{i}/pres_text"], out[f"step{i}/pres_pooled"] = pc[2], pc[3]
    assert len(train_calls) == steps * accum and len(rec["targets"]) == steps * accum, (len(rec["calls"]), len(train_calls), len(rec["targets"]))
    for i, ((lat, ts, emb, pooled, _, kw), tgt) in enumerate(zip(train_calls, rec["targets"])):
        out[f"step{i}/latent_model_input"], out[f"step{i}/timestep"], out[f"step{i}/text"], out[f"step{i}/pooled"], out[f"step{i}/target"] = lat, ts, emb, pooled, tgt
    save_root = os.path.join(tmp, "out", "aitk_trainer_run")
    sd_final = load_file(os.path.join(save_root, "aitk_trainer_run.safetensors"))
    for k, v in sd_final.items():
        out[f"saved/{k}"] = v
    opt_sd = torch.load(os.path.join(save_root, "optimizer.pt"), weights_only=True)
    if uncached:
        assert len(rec["encode"]) == steps * accum
        out["encode/images"], out["encode/latents"], out["encode/rng_state"] = rec["encode"][0]
    if network == "lora" and not uncached:  # the adapter-type variants keep the fixture small: the saved file (= the EMA weights) and the losses carry the comparison
        for i, st in opt_sd["state"].items():
            out[f"opt/{i}/exp_avg"], out[f"opt/{i}/exp_avg_sq"] = st["exp_avg"], st["exp_avg_sq"]
    for k, v in keep["init"].items():
        out[f"init/{k}"] = v
    if network == "lora" and not uncached:
        for i, sp in enumerate(ema_.shadow_params):
            out[f"ema/{i}"] = sp.detach().clone()
    meta = {"steps": steps, "accum": accum, "dtype": dtype, "preservation": ({"kind": preservation, "multiplier": 0.5} if preservation else None), "network_kind": network, "only_if_contains": PARTIAL_ONLY if network == "lora_partial" else None, "uncached": bool(uncached), "light": bool(network != "lora" or uncached), "quantize": bool(quantize), "base_is_quantized": bool(getattr(sd_.unet, "is_quantized", False)), "resume_at": n_before, "kw": train_calls[0][5], "opt_group": {k: v for k, v in opt_sd["param_groups"][0].items() if k in ("lr", "betas", "eps", "weight_decay")},
            "max_grad_norm": tr.train_config.max_grad_norm, "ema_decay": tr.train_config.ema_config.ema_decay, "saved_keys": list(sd_final.keys()),
            "files": sorted(os.listdir(save_root)), "n_predict_calls": len(rec["calls"]), "trainer": type(tr).__name__, "network": type(net_).__name__,
            "scheduler": type(sd_.noise_scheduler).__name__, "model": type(sd_).__name__, "model_mro": [k.__name__ for k in type(sd_).__mro__][:3]}
    out_dir = out_dir or HERE
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(out_dir, f"trainer_loop_{tag}_tiny.safetensors"), {"meta": json.dumps(meta, sort_keys=True)})
    print("trainer loop golden:", meta["trainer"], meta["model"], meta["network"], meta["scheduler"], "losses", [round(x, 5) for x in losses], meta["files"])


def golden_wan_vae_flow(out_dir=None):
    """The part of the Wan2.1 video-VAE encoder the reference holds IN TREE (toolkit/models/wan21/autoencoder_kl_wan.py): its own copy of the
    encoder forward (`_wan_encoder_forward`, :32-77: conv_in -> down blocks -> mid block -> norm_out -> nonlinearity -> conv_out) and the
    cache-free temporal down-sampler it states to be EXACTLY the published chunked semantics (`_wan_resample_forward`, :132-141:
    out = cat([x[:, :, :1], time_conv(x)], dim=2)).  Those two functions are loaded from the reference file and executed here, cache-free over the
    whole clip, with the oracle's blocks as `self` (the block classes come from diffusers, which is not in this image: the module is imported
    with placeholder classes of those names).  The result pins (a) the encoder's stage order and (b) the oracle's chunked evaluation — first frame
    alone, four frames at a time, 2-frame feature caches, first chunk past the time convolution — against the reference's statement of what that
    evaluation equals.  Block internals (residual block, RMS norm, attention, causal padding) stay diffusers-only: unpinned."""
    import importlib.util
    import types

    from oracle import wan_vae_ref

    ph = types.ModuleType("diffusers.models.autoencoders.autoencoder_kl_wan")
    ph.CACHE_T = wan_vae_ref.CACHE_T
    for name in ("AutoencoderKLWan", "WanDecoder3d", "WanEncoder3d", "WanResample"):
        setattr(ph, name, type(name, (), {"forward": lambda self, *a, **k: (_ for _ in ()).throw(RuntimeError("placeholder"))}))
    ph.patchify = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("placeholder"))
    saved = {k: sys.modules.get(k) for k in ("diffusers", "diffusers.models", "diffusers.models.autoencoders", ph.__name__)}
    for k in ("diffusers", "diffusers.models", "diffusers.models.autoencoders"):
        if k not in sys.modules:
            m = types.ModuleType(k)
            m.__path__ = []
            sys.modules[k] = m
    sys.modules[ph.__name__] = ph
    try:
        spec = importlib.util.spec_from_file_location("ref_autoencoder_kl_wan", os.path.join(ref_shims.REFERENCE, "toolkit", "models", "wan21", "autoencoder_kl_wan.py"))
        refmod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(refmod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v

    class Down(torch.nn.Module):  # an oracle WanResample behind the reference's cache-free forward
        def __init__(self, blk):
            super().__init__()
            self.blk = blk

        def forward(self, x, feat_cache=None, feat_idx=None):
            assert feat_cache is None
            if self.blk.mode == "downsample3d":
                return refmod._wan_resample_forward(self.blk, x)
            return self.blk(x)  # downsample2d: no temporal part (the reference routes it to the stock forward)

    out, meta = {}, {"cases": []}
    for tag, cfg, seed, T, hw in (("tiny", dict(base_dim=32, z_dim=4, dim_mult=(1, 2, 4, 4), num_res_blocks=1, temperal_downsample=(False, True, True)), 0, 9, (32, 32)),
                                  ("two_res", dict(base_dim=16, z_dim=4, dim_mult=(1, 2, 4, 4), num_res_blocks=2, temperal_downsample=(False, True, True)), 5, 13, (16, 24)),
                                  # (a single frame cannot take this route: the reference's cache-free time_conv needs >= 3 frames)
                                  ("five_frames", dict(base_dim=16, z_dim=4, dim_mult=(1, 2, 4, 4), num_res_blocks=1, temperal_downsample=(False, True, True)), 2, 5, (16, 16))):
        vae = wan_vae_ref.AutoencoderKLWanEncoder(**cfg)
        wan_vae_ref.init_synthetic_(vae, seed)
        enc = vae.encoder
        duck = types.SimpleNamespace(gradient_checkpointing=False, conv_in=enc.conv_in, mid_block=enc.mid_block, norm_out=enc.norm_out,
                                     nonlinearity=torch.nn.SiLU(), conv_out=enc.conv_out,
                                     down_blocks=[Down(b) if isinstance(b, wan_vae_ref.WanResample) else b for b in enc.down_blocks])
        g = torch.Generator().manual_seed(100 + seed)
        x = torch.rand(1, 3, T, *hw, generator=g) * 2 - 1
        with torch.no_grad():
            y = refmod._wan_encoder_forward(duck, x)  # cache-free, whole clip: the reference's own forward
        out[f"{tag}/x"], out[f"{tag}/encoder_out"] = x, y
        meta["cases"].append({"tag": tag, "cfg": {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}, "seed": seed, "T": T})
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(out_dir or HERE, "wan_vae_flow.safetensors"), {"meta": json.dumps(meta, sort_keys=True)})
    print("wan vae flow golden:", {c["tag"]: list(out[c["tag"] + "/encoder_out"].shape) for c in meta["cases"]})


def golden_trainer_loop_wan(out_dir=None):
    golden_trainer_loop(out_dir, kind="wan")


def golden_trainer_loop_sd15(out_dir=None):
    golden_trainer_loop(out_dir, kind="sd15")


def golden_trainer_loop_sdxl(out_dir=None):
    golden_trainer_loop(out_dir, kind="sdxl")


def golden_trainer_loop_flux_bf16(out_dir=None):
    """train.dtype: bf16 — the reference's default training dtype: base model and activations bf16, the network forced to fp32
    (BaseSDTrainProcess.py:1983), latents / embeddings cast by the trainer before predict_noise."""
    golden_trainer_loop(out_dir, kind="flux", dtype="bf16")


def golden_trainer_loop_flux_bf16_fp8base(out_dir=None):
    """model.quantize: true (BASELINE config 5's base): load_model quantises the block Linears to e4m3 + per-channel scale and releases the bf16
    weights; the trainer's network is adopted over the quantised base."""
    golden_trainer_loop(out_dir, kind="flux", dtype="bf16", quantize=True)


def golden_trainer_loop_flux_dora(out_dir=None):
    """network.type: dora — the reference's DoRAModule (magnitude vector) built, trained, saved and resumed by its own trainer over the plug-in."""
    golden_trainer_loop(out_dir, kind="flux", network="dora")


def golden_trainer_loop_flux_lokr_lowrank(out_dir=None):
    """network.type: lokr with lokr_full_rank: false — LokrModule with W2 = lokr_w2_a @ lokr_w2_b."""
    golden_trainer_loop(out_dir, kind="flux", network="lokr_lowrank")


def golden_trainer_loop_flux_lora_partial(out_dir=None):
    """network_kwargs.only_if_contains: adapters on a few Linears only — members of the native graph's same-input groups with and without an adapter side by side."""
    golden_trainer_loop(out_dir, kind="flux", network="lora_partial")


def golden_trainer_loop_flux_uncached(out_dir=None):
    """datasets without cached latents: the batch carries images, process_general_training_batch calls `sd.encode_images` (BaseSDTrainProcess.py:1106-1140) — the
    plug-in's native AutoencoderKL encoder, loaded by load_model from the pipeline directory's vae/ folder."""
    golden_trainer_loop(out_dir, kind="flux", uncached=True)


def golden_trainer_loop_flux_blankpp(out_dir=None):
    """train.blank_prompt_preservation: true — the trainer's prior / training / preservation predictions per step and ONE backward through two native graphs."""
    golden_trainer_loop(out_dir, kind="flux", preservation="blank")


def golden_trainer_loop_flux_accum2(out_dir=None):
    """train.gradient_accumulation: 2 — two micro-batches per hook_train_loop call: `optimizer.zero_grad()` (set_to_none) at its top drops the
    adopted parameters' .grad views, two backward passes accumulate, one clip / step / EMA (SDTrainer.py:2246-2293)."""
    golden_trainer_loop(out_dir, kind="flux", accum=2)


if __name__ == "__main__":
    if len(sys.argv) > 1:  # python tests/golden/make_golden.py golden_ema_options ...: only the named generators
        for name in sys.argv[1:]:
            globals()[name]()
        raise SystemExit(0)
    golden_ema_options()
    golden_unet_lora()
    golden_unet_conv_lora()
    golden_unet_conv_lora_highrank()
    golden_unet_keymap_keys()
    golden_lora()
    golden_dora()
    golden_lokr()
    golden_lokr_lowrank()
    golden_flux_blocks()
    golden_wan_attn()
    golden_optimizer_ema()
    golden_model_hash()
    golden_merge()
    golden_vae_encoder()
    golden_latent_cache_paths()
    golden_kohya_to_peft()
    golden_flowmatch()
    golden_wan_lora_keys()
    golden_adoption()
    golden_flux_glue()
    golden_text_encoders()
    golden_wan_vae_flow()
    golden_trainer_loop()
    golden_trainer_loop(kind="wan")
    golden_trainer_loop(kind="sd15")
    golden_trainer_loop(kind="sdxl")
    golden_trainer_loop(kind="flux", accum=2)
    golden_trainer_loop(kind="flux", dtype="bf16")
    golden_trainer_loop(kind="flux", dtype="bf16", quantize=True)
    golden_trainer_loop(kind="flux", network="dora")
    golden_trainer_loop(kind="flux", uncached=True)
    golden_trainer_loop(kind="flux", network="lora_partial")
    golden_trainer_loop(kind="flux", network="lokr_lowrank")
    golden_trainer_loop(kind="flux", preservation="blank")
