"""The FLUX glue that only diffusers' un-vendored classes spell out, pinned to code the reference holds (VERDICT r4 item 2;
tests/golden/make_golden.py golden_flux_glue EXECUTES it): the converter script's diffusers -> BFL key map and `swap_scale_shift`
(scripts/convert_diffusers_to_comfy.py:56-290), the in-tree BFL-lineage `Modulation` / `LastLayer` / `MLPEmbedder` / `timestep_embedding`
(extensions_built_in/diffusion_models/flux2/src/model.py:243-279, 446-499) and `guidance_embed_bypass_forward` (toolkit/models/flux.py:9-15).

  * adaLN chunk order: diffusers' `norm1.linear` IS BFL's `img_mod.lin` (copied un-swapped), whose forward chunks (shift, scale, gate) x 2;
  * final layer: diffusers' `norm_out.linear` is BFL's `final_layer.adaLN_modulation.1` with the halves SWAPPED -> diffusers order (scale, shift);
  * timestep embedding: [cos | sin], frequencies exp(-ln(10000) i / 128), of 1000 t; embedder = linear_2(silu(linear_1(.)));
  * conditioning (bypass) = timestep_embedder(time_proj(t)) + text_embedder(pooled).

The oracle's diffusers-named restatement (oracle/flux_ref.py) must reproduce the recorded reference outputs; the native graph is held to that
oracle by tests/test_host_graph_cpu.py, which closes the chain reference-held code -> oracle -> HIP graph."""
import json
import os

import torch
from safetensors import safe_open
from safetensors.torch import load_file

from oracle import flux_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "flux_glue.safetensors")


def _close(a, b, tol=2e-6):
    return torch.allclose(a, b, rtol=tol, atol=tol)


def test_key_map_orders_follow_the_converter_script():
    with safe_open(GOLD, "pt") as fh:
        meta = json.loads(fh.metadata()["meta"])
    assert meta["img_mod"] == ["norm1.linear.weight"] and meta["txt_mod"] == ["norm1_context.linear.weight"] and meta["single_mod"] == ["norm.linear.weight"]
    assert meta["img_qkv"] == ["attn.to_q.weight", "attn.to_k.weight", "attn.to_v.weight"]
    assert meta["txt_qkv"] == ["attn.add_q_proj.weight", "attn.add_k_proj.weight", "attn.add_v_proj.weight"]
    assert meta["single_linear1"] == ["attn.to_q.weight", "attn.to_k.weight", "attn.to_v.weight", "proj_mlp.weight"]
    assert meta["final_mod"] == ["norm_out.linear.weight"] and meta["final_linear"] == ["proj_out.weight"]
    assert meta["time_in"] == [["time_text_embed.timestep_embedder.linear_1.weight"], ["time_text_embed.timestep_embedder.linear_2.weight"]]
    assert meta["guidance_in"] == [["time_text_embed.guidance_embedder.linear_1.weight"], ["time_text_embed.guidance_embedder.linear_2.weight"]]
    assert meta["vector_in"] == [["time_text_embed.text_embedder.linear_1.weight"], ["time_text_embed.text_embedder.linear_2.weight"]]
    # the same-input groups of the native graph are laid out in exactly this order (one K-concatenated data-gradient GEMM per group)
    import ai_toolkit_amd  # noqa: F401
    from ai_toolkit_amd.flux import FluxTransformer2DModel
    from oracle import ref_ops
    from tests.test_host_graph_cpu import CFG

    nat = FluxTransformer2DModel(**CFG, dtype=torch.float32, device="cpu", ops=ref_ops)
    blk = nat.single_transformer_blocks[0]
    grp = [g for g in nat._dgrad_groups() if g[0] is blk.attn.to_q][0]
    assert [id(l) for l in grp] == [id(blk.attn.to_q), id(blk.attn.to_k), id(blk.attn.to_v), id(blk.proj_mlp)]


def test_adaln_zero_and_single_chunk_order_is_shift_scale_gate():
    g = load_file(GOLD)
    d = g["zero/vec"].shape[1]
    m = flux_ref.AdaLayerNormZero(d)
    m.load_state_dict({"linear.weight": g["zero/linear.weight"], "linear.bias": g["zero/linear.bias"]})
    with torch.no_grad():
        x_mod, gate_msa, shift_mlp, scale_mlp, gate_mlp = m(g["zero/x"], g["zero/vec"])
    for got, key in ((x_mod, "x_mod"), (gate_msa, "gate_msa"), (shift_mlp, "shift_mlp"), (scale_mlp, "scale_mlp"), (gate_mlp, "gate_mlp")):
        assert _close(got, g[f"zero/{key}"]), key
    s = flux_ref.AdaLayerNormZeroSingle(d)
    s.load_state_dict({"linear.weight": g["single/linear.weight"], "linear.bias": g["single/linear.bias"]})
    with torch.no_grad():
        x_mod, gate = s(g["zero/x"], g["zero/vec"])
    assert _close(x_mod, g["single/x_mod"]) and _close(gate, g["single/gate"])
    # a wrong chunk order is caught: swapping shift and scale of the first triple changes the result
    w = g["zero/linear.weight"].clone()
    w[:d], w[d:2 * d] = g["zero/linear.weight"][d:2 * d], g["zero/linear.weight"][:d]
    m.load_state_dict({"linear.weight": w, "linear.bias": g["zero/linear.bias"]})
    with torch.no_grad():
        assert not _close(m(g["zero/x"], g["zero/vec"])[0], g["zero/x_mod"], 1e-3)


def test_final_layer_is_scale_then_shift_in_diffusers_naming():
    g = load_file(GOLD)
    d = g["zero/vec"].shape[1]
    n = flux_ref.AdaLayerNormContinuous(d, d)
    n.load_state_dict({"linear.weight": g["final/norm_out.linear.weight"], "linear.bias": torch.zeros(2 * d)})
    with torch.no_grad():
        out = torch.nn.functional.linear(n(g["zero/x"], g["zero/vec"]), g["final/proj_out.weight"])
    assert _close(out, g["final/out"], 1e-5)
    sw = torch.cat(g["final/norm_out.linear.weight"].chunk(2, 0)[::-1], 0)  # un-swapped BFL weight in the diffusers module: must NOT match
    n.load_state_dict({"linear.weight": sw, "linear.bias": torch.zeros(2 * d)})
    with torch.no_grad():
        assert not _close(torch.nn.functional.linear(n(g["zero/x"], g["zero/vec"]), g["final/proj_out.weight"]), g["final/out"], 1e-3)


def test_timestep_embedding_layout_and_embedder_mlp():
    g = load_file(GOLD)
    proj = flux_ref.get_timestep_embedding(g["temb/t"] * 1000, 256)
    assert _close(proj, g["temb/proj"], 1e-5) and proj.shape == (2, 256)
    d = g["temb/linear_1.weight"].shape[0]
    e = flux_ref.TimestepEmbedding(256, d)
    e.load_state_dict({k: g[f"temb/{k}"] for k in ("linear_1.weight", "linear_1.bias", "linear_2.weight", "linear_2.bias")})
    with torch.no_grad():
        assert _close(e(proj), g["temb/out"], 1e-5)
    # the kernel table's timestep_embed (what the native graph launches) writes the same layout
    from oracle import ref_ops

    out = torch.empty(2, 256)
    ref_ops.timestep_embed((g["temb/t"] * 1000).float().contiguous(), out)
    assert _close(out, g["temb/proj"], 1e-5)


def test_bypass_conditioning_is_timestep_plus_pooled_text():
    g = load_file(GOLD)
    d = g["bypass/sd/timestep_embedder.linear_1.weight"].shape[0]
    tte = flux_ref.CombinedTimestepGuidanceTextProjEmbeddings(d, g["bypass/pooled"].shape[1])
    tte.load_state_dict({k[len("bypass/sd/"):]: v for k, v in g.items() if k.startswith("bypass/sd/")})
    with torch.no_grad():
        got = tte(g["temb/t"] * 1000, None, g["bypass/pooled"])
        assert _close(got, g["bypass/conditioning"], 1e-5)
        with_guidance = tte(g["temb/t"] * 1000, torch.ones(2) * 1000, g["bypass/pooled"])
    assert not _close(with_guidance, got, 1e-3)
