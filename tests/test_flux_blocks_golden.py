"""FLUX block arithmetic of the oracle (oracle/flux_ref.py, diffusers naming) pinned on outputs of the REFERENCE'S OWN in-tree
FLUX blocks — the Chroma model's DoubleStreamBlock / SingleStreamBlock / LastLayer / EmbedND / timestep_embedding / MLPEmbedder
(extensions_built_in/diffusion_models/chroma/src/layers.py, math.py), executed by tests/golden/make_golden.py on the same weights
mapped through the reference's diffusers<->BFL key map.  Then the native host graph (oracle kernel table) on the same inputs."""
import json
import os

import torch
from safetensors import safe_open
from safetensors.torch import load_file

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd.flux import FluxTransformer2DModel
from oracle import flux_ref, ref_ops

G = os.path.join(os.path.dirname(__file__), "golden", "flux_blocks_chroma.safetensors")


def blocks_model(cfg):
    torch.manual_seed(0)
    model = flux_ref.FluxTransformer2DModel(**cfg)
    flux_ref.init_synthetic_(model, seed=1234, std=0.05)
    with torch.no_grad():
        gi = torch.Generator().manual_seed(77)
        for n, p_ in model.named_parameters():
            if n.endswith("bias"):
                p_.copy_(torch.randn(p_.shape, generator=gi) * 0.05)
            if "norm_" in n and n.endswith("weight"):
                p_.copy_(1 + 0.2 * torch.randn(p_.shape, generator=gi))
    return model


def load():
    with safe_open(G, "pt") as f:
        meta = {k: json.loads(v) for k, v in f.metadata().items()}
    t = load_file(G)
    model = blocks_model(meta["cfg"])
    chk = torch.stack([v.double().abs().sum() for v in model.state_dict().values()]).float()
    assert torch.allclose(chk, t["w_checksum"], rtol=1e-6), "seeded weight regeneration drifted from the golden's"
    return t, meta, model


def test_oracle_blocks_equal_reference_chroma_blocks():
    t, meta, model = load()
    B, Hl, Wl, n_txt = meta["shape"]
    img, txt, temb, ids = t["in/img"], t["in/txt"], t["in/temb"], t["in/ids"]
    with torch.no_grad():
        # RoPE table: reference pe[..., d, i, j] = [[cos, -sin], [sin, cos]] per frequency pair
        cos, sin = flux_ref.rope_freqs(ids)
        pe = t["ref/pe"][0, 0]  # [S, 64, 2, 2]
        assert torch.allclose(cos[:, 0::2], pe[..., 0, 0], atol=1e-6) and torch.allclose(sin[:, 0::2], pe[..., 1, 0], atol=1e-6)
        assert torch.allclose(cos[:, 1::2], pe[..., 1, 1], atol=1e-6) and torch.allclose(-sin[:, 1::2], pe[..., 0, 1], atol=1e-6)
        rot = (cos, sin)
        e, h = model.transformer_blocks[0](img, txt, temb, rot)
        assert torch.allclose(h, t["ref/double/img"], rtol=1e-4, atol=2e-5), (h - t["ref/double/img"]).abs().max()
        assert torch.allclose(e, t["ref/double/txt"], rtol=1e-4, atol=2e-5), (e - t["ref/double/txt"]).abs().max()
        x = model.single_transformer_blocks[0](torch.cat((t["ref/double/txt"], t["ref/double/img"]), 1), temb, rot)
        assert torch.allclose(x, t["ref/single/x"], rtol=1e-4, atol=2e-5), (x - t["ref/single/x"]).abs().max()
        last = model.proj_out(model.norm_out(t["ref/single/x"][:, n_txt:], temb))
        assert torch.allclose(last, t["ref/last"], rtol=1e-4, atol=2e-5), (last - t["ref/last"]).abs().max()
        # conditioning: the model is called with timestep / 1000 and multiplies by 1000 (Chroma: time_factor = 1000)
        sinus = flux_ref.get_timestep_embedding(t["in/t"] * 1000, 256)
        assert torch.allclose(sinus, t["ref/t_sinusoid"], rtol=1e-5, atol=1e-6)
        assert torch.allclose(model.time_text_embed.timestep_embedder(t["ref/t_sinusoid"]), t["ref/t_mlp"], rtol=1e-5, atol=1e-6)
        # embedder sum: the reference's guidance-bypass forward (toolkit/models/flux.py:8-14) = oracle forward minus the guidance term
        tte = model.time_text_embed
        t1000, guid = t["in/t"] * 1000, torch.tensor([3.5, 1.0]) * 1000
        full = tte(t1000, guid, t["in/pooled"])
        g_term = tte.guidance_embedder(flux_ref.get_timestep_embedding(guid, 256))
        assert torch.allclose(full - g_term, t["ref/cond_no_guidance"], rtol=1e-5, atol=1e-5)


def test_native_host_graph_blocks_equal_reference_chroma_blocks():
    """the hand-written forward graph (ai_toolkit_amd/flux.py, oracle kernel table in fp32): whole tiny model = embedders +
    1 double + 1 single + head, against the composition of the reference blocks' outputs is covered block-wise through the
    oracle above; here the same weights / inputs go through forward_native and must reproduce the oracle model end to end."""
    t, meta, model = load()
    B, Hl, Wl, n_txt = meta["shape"]
    cfg = meta["cfg"]
    nat = FluxTransformer2DModel(**cfg, dtype=torch.float32, device="cpu", ops=ref_ops)
    nat.load_state_dict(model.state_dict(), strict=True)
    g = torch.Generator().manual_seed(4)
    hidden = torch.randn(B, (Hl // 2) * (Wl // 2), cfg["in_channels"], generator=g)
    enc = torch.randn(B, n_txt, cfg["joint_attention_dim"], generator=g)
    pooled = torch.randn(B, cfg["pooled_projection_dim"], generator=g)
    ts = torch.tensor([0.3, 0.8])
    img_ids, txt_ids = flux_ref.make_ids(Hl, Wl, n_txt)
    with torch.no_grad():
        want = model(hidden, enc, pooled, ts, img_ids, txt_ids, torch.ones(B))
        got = nat.forward_native(hidden, enc, pooled, ts, img_ids, txt_ids, torch.ones(B), save_for_backward=False)
    assert torch.allclose(got, want, rtol=2e-4, atol=2e-5), (got - want).abs().max()
