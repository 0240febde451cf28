"""UNet adapters pinned to the reference's own LoRASpecialNetwork (kohya format, SD1.5 / SDXL branch), executed by
tests/golden/make_golden.py::golden_unet_lora on the oracle UNet trees: adapter discovery + names (Linear and 1x1-Conv2d children of
every Transformer2DModel), init draws under the same seed, the adapter forward / every gradient through the reference's own
LoRAModule.forward, and the kohya-format state dict it saves (keys, shapes incl. [r, in, 1, 1] conv adapters, values, alpha)."""
import hashlib
import json
import os

import pytest
import torch
from safetensors import safe_open
from safetensors.torch import load_file

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd.lora import FusedLoRANetwork
from ai_toolkit_amd.unet import UNet2DConditionModel
from oracle import lora_ref, ref_ops, unet_ref
from tests.test_unet_cpu import TINY_SD15, TINY_SDXL, _inputs, _nhwc8

G = os.path.join(os.path.dirname(__file__), "golden", "unet_lora_tiny.safetensors")


def _golden():
    with safe_open(G, "pt") as f:
        meta = json.loads(f.metadata()["meta"])
    return load_file(G), meta


def _fused(cfg, seed=99):
    torch.manual_seed(0)
    ref = unet_ref.UNet2DConditionModel(**cfg)
    unet_ref.init_synthetic_(ref, seed=11)
    nat = UNet2DConditionModel(**cfg, dtype=torch.float32, device="cpu", ops=ref_ops)
    nat.load_state_dict(ref.state_dict(), strict=True)
    torch.manual_seed(seed)
    net = FusedLoRANetwork(nat, lora_dim=4, alpha=2.0, target_lin_modules=("Transformer2DModel",), is_transformer=False, peft_format=False,
                           transformer_only=False)
    return ref, nat, net


@pytest.mark.parametrize("tag,cfg", [("sd15", TINY_SD15), ("sdxl", TINY_SDXL)])
def test_fused_unet_network_matches_reference_network(tag, cfg, tmp_path):
    t, meta = _golden()
    m = meta[tag]
    ref, nat, net = _fused(cfg)
    assert [x.lora_name for x in net.unet_loras] == m["names"]
    assert net.unet_loras[0].scale == m["scale"] == 0.5 and net.peft_format == m["peft_format"] is False
    for x in net.unet_loras:  # same construction order and shapes => same RNG consumption => same kaiming draws
        want = t[f"{tag}/init/{x.lora_name}/down"]
        assert torch.equal(x.lora_down.weight, want.reshape(x.lora_down.weight.shape)), x.lora_name
        with torch.no_grad():
            x.lora_up.weight.copy_(t[f"{tag}/warm/{x.lora_name}/up"].reshape(x.lora_up.weight.shape))
    net.apply_to()
    net.build_arena("cpu", groups=nat.lora_groups())
    net.refresh_shadows(ref_ops)
    nat.attach_network(net)
    nat.prepare()
    lat, ts, ctx, added = _inputs(cfg)
    B, _, H, W = lat.shape
    with net:
        pred = nat.forward_native(_nhwc8(lat), ts, ctx, added, B=B, H=H, W=W)
        got = pred.view(B, H, W, 4).permute(0, 3, 1, 2)
        assert torch.allclose(got, t[f"{tag}/pred"], rtol=2e-4, atol=2e-5), (got - t[f"{tag}/pred"]).abs().max()
        net.zero_grad_arena()
        nat.backward_native(t[f"{tag}/wgt"].permute(0, 2, 3, 1).reshape(B * H * W, 4).contiguous())
    for x in net.unet_loras:
        for nm, p_ in (("down", x.lora_down.weight), ("up", x.lora_up.weight)):
            want = t[f"{tag}/grad/{x.lora_name}/{nm}"].reshape(p_.shape)
            err = ((p_.grad - want).norm() / (want.norm() + 1e-12)).item()
            assert err < 5e-4, (x.lora_name, nm, err)
    # the saved file: exactly the reference's get_state_dict (kohya keys incl. alpha, Conv2d-shaped 1x1 adapters)
    f = tmp_path / "unet.safetensors"
    net.save_weights(str(f), dtype=torch.float32)
    sd = load_file(str(f))
    assert sorted(sd.keys()) == sorted(m["saved_keys"])
    for k in m["saved_keys"]:
        want = t[f"{tag}/saved/{k}"]
        assert sd[k].shape == want.shape and torch.equal(sd[k], want), k
    assert list(net.get_state_dict(dtype=torch.float32).keys()) == m["saved_keys"]  # same key ORDER as the reference's state_dict


@pytest.mark.parametrize("tag,cfg", [("sd15", TINY_SD15), ("sdxl", TINY_SDXL)])
def test_oracle_unet_lora_matches_reference_network(tag, cfg):
    t, meta = _golden()
    torch.manual_seed(0)
    ref = unet_ref.UNet2DConditionModel(**cfg)
    unet_ref.init_synthetic_(ref, seed=11)
    torch.manual_seed(99)
    net = lora_ref.RefLoRANetwork(ref, 4, target=("Transformer2DModel",), kohya_unet=True, alpha=2.0)
    assert [x.lora_name for x in net.unet_loras] == meta[tag]["names"]
    with torch.no_grad():
        for x in net.unet_loras:
            assert torch.equal(x.lora_down.weight, t[f"{tag}/init/{x.lora_name}/down"])
            x.lora_up.weight.copy_(t[f"{tag}/warm/{x.lora_name}/up"])
    net.apply_to()
    lat, ts, ctx, added = _inputs(cfg)
    with net:
        pred = ref(lat, ts, ctx, added)
        (pred * t[f"{tag}/wgt"]).sum().backward()
    assert torch.allclose(pred, t[f"{tag}/pred"], rtol=1e-5, atol=1e-6)
    for x in net.unet_loras:
        assert torch.allclose(x.lora_up.weight.grad, t[f"{tag}/grad/{x.lora_name}/up"], rtol=1e-4, atol=1e-7), x.lora_name


@pytest.mark.parametrize("tag,cfg,count", [("sd15_full", unet_ref.SD15, 192), ("sdxl_full", unet_ref.SDXL, 722)])
def test_full_size_adapter_inventory_equals_the_reference(tag, cfg, count):
    """BASELINE config 1 = 192 adapters (SD1.5), config 2 = 722 (SDXL): names and shapes of the network the reference builds on the
    full trees (hashed) equal what FusedLoRANetwork builds on the native model."""
    _, meta = _golden()
    m = meta[tag]
    assert m["count"] == count
    with torch.device("meta"):
        nat = UNet2DConditionModel(**cfg, dtype=torch.float32)
    net = FusedLoRANetwork(nat, lora_dim=4, alpha=4.0, target_lin_modules=("Transformer2DModel",), is_transformer=False, peft_format=False,
                           transformer_only=False)
    names = [x.lora_name for x in net.unet_loras]
    assert len(names) == count and names[:3] == m["first"] and names[-3:] == m["last"]
    assert hashlib.sha256("\n".join(names).encode()).hexdigest() == m["names_sha256"]
    shapes = []
    for x in net.unet_loras:
        d, u = list(x.lora_down.weight.shape), list(x.lora_up.weight.shape)
        if x.is_conv1x1:
            d, u = d + [1, 1], u + [1, 1]
        shapes.append([d, u])
    assert hashlib.sha256(json.dumps(shapes).encode()).hexdigest() == m["shapes_sha256"]
    assert sum(x.lora_down.weight.numel() + x.lora_up.weight.numel() for x in net.unet_loras) == m["params"]


def test_unet_parameter_names_equal_the_reference_keymaps():
    """The diffusers UNet2DConditionModel is not vendored, but the reference's LDM<->diffusers key maps (toolkit/keymaps/stable_diffusion_sd1.json,
    stable_diffusion_sdxl.json) list every one of its parameter names: 686 for SD1.5, 1680 for SDXL.  Oracle and fused model must carry exactly
    those state-dict keys (tests/golden/unet_keymap_keys.json, written by make_golden.golden_unet_keymap_keys from the reference tree)."""
    import json
    import os

    import torch

    from ai_toolkit_amd.unet import SD15_CONFIG, SDXL_CONFIG, UNet2DConditionModel
    from oracle import ref_ops, unet_ref

    keys = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "unet_keymap_keys.json")))
    assert (len(keys["sd1"]), len(keys["sdxl"])) == (686, 1680)
    for tag, ocfg, ncfg in (("sd1", unet_ref.SD15, SD15_CONFIG), ("sdxl", unet_ref.SDXL, SDXL_CONFIG)):
        with torch.device("meta"):
            ref = unet_ref.UNet2DConditionModel(**ocfg)
        nat = UNet2DConditionModel(**ncfg, dtype=torch.float32, device="meta", ops=ref_ops)
        assert sorted(ref.state_dict().keys()) == keys[tag], tag
        assert sorted(nat.state_dict().keys()) == keys[tag], tag
