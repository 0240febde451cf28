"""PEFT <-> kohya LoRA key conversion round trip on the adapter names the reference itself produced (golden)."""
import json
import os

import torch
from safetensors import safe_open
from safetensors.torch import load_file

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd import convert
from ai_toolkit_amd.flux import FluxTransformer2DModel

G = os.path.join(os.path.dirname(__file__), "golden", "lora_flux_tiny.safetensors")
TINY = dict(in_channels=64, num_layers=1, num_single_layers=1, attention_head_dim=128, num_attention_heads=2,
            joint_attention_dim=64, pooled_projection_dim=32)


def test_peft_kohya_roundtrip_on_reference_saved_keys():
    t = load_file(G)
    with safe_open(G, "pt") as f:
        keys = json.loads(f.metadata()["saved_keys"])
    peft = {k: t[f"saved/{k}"] for k in keys}
    kohya = convert.peft_to_kohya(peft)
    assert "lora_transformer_transformer_blocks_0_attn_to_q.lora_down.weight" in kohya
    assert float(kohya["lora_transformer_transformer_blocks_0_attn_to_q.alpha"]) == 8.0
    assert len(kohya) == len(peft) + len(peft) // 2
    model = FluxTransformer2DModel(**TINY, dtype=torch.float32, device="cpu")
    paths = [n for n, m in model.named_modules() if m.__class__.__name__ == "Linear"]
    back = convert.kohya_to_peft(kohya, paths)
    assert sorted(back) == sorted(peft) and all(torch.equal(back[k], peft[k]) for k in peft)


def test_alpha_is_folded_into_lora_up():
    sd = {"lora_transformer_x.lora_down.weight": torch.ones(4, 8), "lora_transformer_x.lora_up.weight": torch.ones(6, 4),
          "lora_transformer_x.alpha": torch.tensor(2.0)}
    out = convert.scale_for_alpha(sd)
    assert torch.allclose(out["lora_transformer_x.lora_up.weight"], torch.full((6, 4), 0.5)) and float(out["lora_transformer_x.alpha"]) == 4.0


def test_wan_key_conversion_matches_reference_converter_golden():
    """tests/golden/wan_lora_keys.json was produced by the reference's toolkit/models/wan21/wan_lora_convert.py."""
    import json
    import os

    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "wan_lora_keys.json")))
    sd = {k: i for i, k in enumerate(gold["diffusers"])}
    orig = convert.wan_lora_to_original(sd)
    assert list(orig) == gold["original"]
    assert list(convert.wan_lora_to_diffusers(orig)) == gold["diffusers"]


def test_saved_file_hashes_equal_the_reference_add_model_hash_to_meta():
    """sshs_model_hash / sshs_legacy_hash (toolkit/metadata.py:32-48) against values computed by the reference's own function."""
    import json
    import os
    from collections import OrderedDict

    import torch

    from ai_toolkit_amd.lora import add_model_hash_to_meta

    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "model_hash.json")))
    g = torch.Generator().manual_seed(51)
    for tag in ("small", "big"):
        shape = tuple(gold[tag]["shape"])
        sd = OrderedDict((f"transformer.blocks.{i}.lora_A.weight", torch.randn(shape, generator=g).to(torch.float16)) for i in range(3))
        meta = add_model_hash_to_meta(sd, OrderedDict(ss_output_name="x", ss_base_model_version="flux1", name="not hashed"))
        assert meta["sshs_model_hash"] == gold[tag]["sshs_model_hash"] and meta["sshs_legacy_hash"] == gold[tag]["sshs_legacy_hash"], tag


def test_kohya_to_peft_agrees_with_the_reference_script_where_the_script_is_well_formed():
    """scripts/convert_lora_to_peft_format.py (executed by make_golden.py on a kohya-format file with the FLUX adapter names,
    alpha = rank) rewrites names with string replacements ("currently only works with flux as support is not quite there yet"): for the
    attention / proj_mlp / proj_out / single-block norm.linear adapters it yields real module paths and the converter here — which
    resolves names against the module tree — gives exactly those keys and values; for ff / ff_context / norm1(_context).linear its
    replacements produce names that are not module paths (`ff.net_0.proj`, `ff.context.net.0.proj`, `norm1_linear`), which the
    module-tree lookup gets right."""
    d = os.path.dirname(__file__)
    gold = json.load(open(os.path.join(d, "golden", "kohya_to_peft_keys.json")))
    sums = json.load(open(os.path.join(d, "golden", "kohya_to_peft_sums.json")))
    g = torch.Generator().manual_seed(81)
    kohya = {}
    for k in gold["kohya_keys"]:
        kohya[k] = torch.tensor(8.0) if k.endswith(".alpha") else (torch.randn(8, 4, generator=g) if "lora_down" in k else torch.randn(4, 8, generator=g))
    model = FluxTransformer2DModel(**TINY, dtype=torch.float32, device="cpu")
    paths = [n for n, m in model.named_modules() if m.__class__.__name__ == "Linear"]
    mine = convert.kohya_to_peft(convert.scale_for_alpha(kohya), paths)
    ref_keys = set(gold["peft_keys"])
    well_formed = {k for k in ref_keys if k.rsplit(".lora_", 1)[0][len("transformer."):] in paths}
    assert len(well_formed) == 28 and well_formed <= set(mine)           # 14 adapters: q/k/v/out/add_*/to_add_out, single q/k/v/proj_mlp/proj_out/norm.linear
    for k in well_formed:
        assert abs(float(mine[k].double().sum()) - sums[k]) < 1e-9, k
    bad = sorted(k for k in ref_keys - well_formed)
    assert all(("ff." in k or "norm1" in k) for k in bad) and len(bad) == 12
    assert len(mine) == 40 and all(k.rsplit(".lora_", 1)[0][len("transformer."):] in paths for k in mine)
