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
