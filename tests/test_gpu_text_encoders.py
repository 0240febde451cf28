"""Row a17 on the GPU: the text-encoder LIBRARY path (`transformers` modules on PyTorch-ROCm, once per caption, off the
per-step path — by decision not a HIP kernel: DESIGN.md section 2) run on cuda:0 against the committed vectors of the reference's own
functions (tests/golden/text_encoders_flux.safetensors: toolkit/train_tools.py:510-574, 192-323, 379-422 executed on tiny encoders).
fp32 on two backends: rocBLAS / MIOpen kernels against the CPU's — equal to fp32 rounding, not bit for bit."""
import os

import pytest
import torch
from safetensors.torch import load_file

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "text_encoders_flux.safetensors")


def _close(a, b, tag):
    assert a.shape == b.shape, tag
    err = (a.float().cpu() - b.float()).abs().max().item()
    assert err <= 2e-5 * max(1.0, b.float().abs().max().item()), (tag, err)


def test_flux_sd_sdxl_prompt_encoding_on_rocm_match_the_reference_vectors():
    from ai_toolkit_amd import plugin
    from tests.test_text_encoders_cpu import PROMPTS, _clips, _encoders

    gold = load_file(GOLD)
    toks, tes = _encoders()
    tes = [t.cuda() for t in tes]
    with torch.no_grad():
        for tag, kw in (("plain", {}), ("masked", {"attn_mask": True}), ("len64", {"max_length": 64})):
            emb, pooled = plugin.encode_prompts_flux(toks, tes, list(PROMPTS), **kw)
            assert emb.is_cuda and pooled.is_cuda
            _close(emb, gold[f"{tag}/embeds"], tag)
            _close(pooled, gold[f"{tag}/pooled"], tag)
    # the plug-in hook on the device the trainer gives it
    plug = plugin.Flux1MI355Model("cuda", dtype=torch.float32)
    plug.tokenizer, plug.text_encoder = toks, tes
    pe = plug.get_prompt_embeds(list(PROMPTS))
    _close(pe.text_embeds, gold["plain/embeds"], "hook")
    _close(pe.pooled_embeds, gold["plain/pooled"], "hook")
    tk, (c1, c2) = _clips()
    c1, c2 = c1.cuda(), c2.cuda()
    with torch.no_grad():
        _close(plugin.encode_prompts(tk[0], c1, list(PROMPTS)), gold["sd/plain"], "sd")
        for tag, kw in (("plain", {}), ("no_te1", {"use_text_encoder_1": False}), ("two_images", {"num_images_per_prompt": 2})):
            e, p = plugin.encode_prompts_xl(tk, [c1, c2], list(PROMPTS), None, **kw)
            _close(e, gold[f"sdxl/{tag}/embeds"], tag)
            _close(p, gold[f"sdxl/{tag}/pooled"], tag)
