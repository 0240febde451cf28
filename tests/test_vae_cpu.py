"""VAE-encoder host graph (implicit-GEMM conv layout, GroupNorm, materialised mid-block attention, latent sampling)
driven by the oracle's torch kernels in fp32 vs the oracle AutoencoderKL encoder; latent-cache path/format."""
import base64
import hashlib
import json
import os

import torch
from safetensors.torch import load_file

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd import buckets as bk
from ai_toolkit_amd import vae as nvae
from oracle import ref_ops, vae_ref

CFG = dict(latent_channels=4, block_out_channels=(32, 64, 64), layers_per_block=1, groups=32)


def build(dtype=torch.float32, device="cpu", ops=ref_ops):
    torch.manual_seed(0)
    ref = vae_ref.AutoencoderKLEncoder(**CFG)
    vae_ref.init_synthetic_(ref)
    nat = nvae.AutoencoderKLEncoder(**CFG, dtype=dtype, device=device, ops=ops)
    nat.load_state_dict({k: v.to(dtype) for k, v in ref.state_dict().items()}, strict=True)
    nat.prepare()
    return ref, nat


def test_encoder_graph_matches_oracle_fp32():
    ref, nat = build()
    g = torch.Generator().manual_seed(1)
    img = torch.rand(2, 3, 32, 24, generator=g) * 2 - 1
    eps = torch.randn(2, 4, 8, 6, generator=g)
    want = ref.encode_images(img, eps)
    got = nat.encode_images(img, eps=eps)
    assert got.shape == want.shape == (2, 4, 8, 6)
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-5), (got - want).abs().max()
    # moments themselves (pre-sampling)
    mom, (h, w) = nat.moments(img)
    m_ref = ref.moments(img).permute(0, 2, 3, 1).reshape(-1, 8)
    assert torch.allclose(mom, m_ref, rtol=1e-4, atol=1e-5)


def test_sd_vae_variant_with_quant_conv_and_the_reference_keymap_names():
    """SD1.5 / SDXL AutoencoderKL: 4 latent channels, a 1x1 quant_conv on the moments, no shift.  Host graph vs oracle, and the parameter
    names of the full-width encoder (+ quant_conv) against the list the reference's own key maps carry (toolkit/keymaps/
    stable_diffusion_sd1.json / _sdxl.json via tests/golden/unet_keymap_keys.json)."""
    torch.manual_seed(0)
    cfg = dict(CFG, scaling_factor=0.13025, shift_factor=0.0, use_quant_conv=True)
    ref = vae_ref.AutoencoderKLEncoder(**cfg)
    vae_ref.init_synthetic_(ref)
    nat = nvae.AutoencoderKLEncoder(**cfg, dtype=torch.float32, device="cpu", ops=ref_ops)
    nat.load_state_dict(ref.state_dict(), strict=True)
    nat.prepare()
    g = torch.Generator().manual_seed(1)
    img = torch.rand(2, 3, 32, 24, generator=g) * 2 - 1
    eps = torch.randn(2, 4, 8, 6, generator=g)
    want, got = ref.encode_images(img, eps), nat.encode_images(img, eps=eps)
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-5), (got - want).abs().max()
    no_q = vae_ref.AutoencoderKLEncoder(**dict(cfg, use_quant_conv=False))
    no_q.load_state_dict({k: v for k, v in ref.state_dict().items() if not k.startswith("quant_conv")})
    assert not torch.allclose(no_q.encode_images(img, eps), want, atol=1e-3)  # the 1x1 convolution is not a no-op
    keys = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "unet_keymap_keys.json")))
    with torch.device("meta"):
        full = vae_ref.AutoencoderKLEncoder(latent_channels=4, use_quant_conv=True)
    full_nat = nvae.AutoencoderKLEncoder(latent_channels=4, use_quant_conv=True, dtype=torch.float32, device="meta", ops=ref_ops)
    assert sorted(full.state_dict().keys()) == keys["sd1_vae_encoder"] == keys["sdxl_vae_encoder"] == sorted(full_nat.state_dict().keys())
    assert len(keys["sd1_vae_encoder"]) == 108


def test_latent_cache_path_and_roundtrip(tmp_path):
    plan = bk.plan_crop(2048, 1365, resolution=1024, bucket_tolerance=64)
    info = nvae.latent_info_dict("/data/set/cat 01.JPG", plan, latent_space_version="flux1")
    path = nvae.latent_cache_path(str(tmp_path / "cat 01.JPG"), info)
    expect_hash = base64.urlsafe_b64encode(hashlib.md5(json.dumps(info, sort_keys=True).encode()).digest()).decode().replace("=", "")
    assert path == os.path.join(str(tmp_path), "_latent_cache", f"cat 01_{expect_hash}.safetensors")
    lat = torch.randn(16, plan.crop_height // 8, plan.crop_width // 8).to(torch.bfloat16)
    nvae.save_latent_cache(path, lat)
    assert list(load_file(path).keys()) == ["latent"]  # toolkit/dataloader_mixins.py:2076-2080
    assert torch.equal(nvae.load_latent_cache(path), lat)


def test_oracle_encoder_equals_reference_ldm_encoder():
    """oracle/vae_ref.Encoder (diffusers names) against the moments computed by the reference's in-tree LDM-style Encoder
    (extensions_built_in/diffusion_models/flux2/src/autoencoder.py) executed by make_golden.py on the same seeded weights mapped to
    its names: ResnetBlock, asymmetric-pad Downsample, single-head mid attention, norm_out -> SiLU -> conv_out."""
    from safetensors import safe_open

    path = os.path.join(os.path.dirname(__file__), "golden", "vae_encoder_ldm.safetensors")
    t = load_file(path)
    with safe_open(path, "pt") as f:
        chans, zc = tuple(json.loads(f.metadata()["chans"])), json.loads(f.metadata()["zc"])
    torch.manual_seed(0)
    enc = vae_ref.Encoder(3, zc, chans, 2, 32)
    with torch.no_grad():
        g = torch.Generator().manual_seed(61)
        for n, p_ in enc.named_parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * (0.08 if p_.dim() > 1 else 0.05) + (1.0 if ("norm" in n and n.endswith("weight")) else 0.0))
        chk = torch.stack([v.double().abs().sum() for v in enc.state_dict().values()]).float()
        assert torch.allclose(chk, t["w_checksum"], rtol=1e-6)
        got = enc(t["x"])
    assert torch.allclose(got, t["moments"], rtol=1e-4, atol=1e-5), (got - t["moments"]).abs().max()


def test_latent_cache_paths_equal_reference_file_item_methods():
    """`_latent_cache` file names against the reference's own FileItemDTO.get_latent_info_dict / get_latent_path
    (toolkit/dataloader_mixins.py:1779-1842), executed by make_golden.py: an existing cache made by the reference is found."""
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "latent_cache_paths.json")))
    cases = gold["latent"]
    assert len(cases) == 4
    for c in cases:
        plan = bk.CropPlan(*c["geometry"])
        info = nvae.latent_info_dict(c["path"], plan, c["latent_space_version"], flip_x=c["flip_x"], flip_y=c["flip_y"])
        assert dict(info) == c["info"] and list(info) == list(c["info"]), c["path"]
        assert nvae.latent_cache_path(c["path"], info) == c["latent_path"], c["path"]


def test_text_embedding_cache_names_and_prompt_embeds_file_equal_reference():
    """`_t_e_cache` names against the reference's get_text_embedding_path, and a file written by the reference's own
    PromptEmbeds.save read back by load_prompt_embeds (toolkit/prompt_utils.py:119-190)."""
    from ai_toolkit_amd import batches

    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "latent_cache_paths.json")))
    assert len(gold["text"]) == 3
    for c in gold["text"]:
        assert batches.text_embedding_cache_path(c["path"], c["caption"], c["space"]) == c["te_path"], c["path"]
    text, pooled, mask = batches.load_prompt_embeds(os.path.join(os.path.dirname(__file__), "golden", "prompt_embeds_ref.safetensors"))
    g = torch.Generator().manual_seed(71)
    assert torch.equal(text, torch.randn(1, 6, 8, generator=g).to(torch.bfloat16))
    assert torch.equal(pooled, torch.randn(1, 4, generator=g).to(torch.bfloat16)) and mask is None
