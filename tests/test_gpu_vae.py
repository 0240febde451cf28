"""GPU parity of the VAE-encoder kernels (implicit-GEMM conv on MFMA, GroupNorm+SiLU, row softmax, latent sample) and
of the whole encoder graph vs the oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


def R(*shape, s=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * s


@pytest.mark.parametrize("case", [
    dict(B=2, H=20, W=12, Cin=8, Cout=128, stride=1),      # conv_in shape class (3->8 padded channels, K = 72)
    dict(B=1, H=33, W=31, Cin=64, Cout=96, stride=1),      # ragged pixel count
    dict(B=2, H=16, W=24, Cin=128, Cout=128, stride=2),    # downsampler: stride 2, pad (0,1,0,1)
    dict(B=1, H=64, W=64, Cin=256, Cout=512, stride=1),    # 256x256 tile path
    dict(B=1, H=16, W=16, Cin=512, Cout=32, stride=1),     # conv_out (N < tile)
])
def test_conv3x3_implicit_gemm(case):
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    B, H, W, Cin, Cout, stride = (case[k] for k in ("B", "H", "W", "Cin", "Cout", "stride"))
    x = R(B * H * W, Cin, seed=1).to(bf).cuda()
    w = R(Cout, 9 * Cin, s=(9 * Cin) ** -0.5, seed=2).to(bf).cuda()
    bias = R(Cout, s=0.1, seed=3).to(bf).cuda()
    kw = dict(B=B, H=H, W=W, bias=bias)
    if stride == 2:
        kw.update(stride=2, pad_t=0, pad_l=0, Ho=H // 2, Wo=W // 2)
    Ho, Wo = kw.get("Ho", H), kw.get("Wo", W)
    res = R(B * Ho * Wo, Cout, seed=4).to(bf).cuda()
    out = torch.full((B * Ho * Wo, Cout), float("nan"), dtype=bf, device="cuda")
    ref = torch.empty(B * Ho * Wo, Cout, dtype=torch.float32, device="cuda")
    ops.conv3x3(x, w, out, flags=ops.EPI_ADD_AUX, aux_in=res, **kw)
    ref_ops.conv3x3(x, w, ref, flags=ref_ops.EPI_ADD_AUX, aux_in=res, **kw)
    torch.cuda.synchronize()
    assert rel(out, ref) < 5e-3, rel(out, ref)


def test_groupnorm_softmax_sample_kernels():
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    for (B, HW, Cc, silu) in ((2, 300, 128, True), (1, 1000, 512, False), (2, 77, 256, True)):
        x = (R(B * HW, Cc, seed=5) * 2 + 0.5).to(bf).cuda()
        ga, be = (1 + 0.1 * R(Cc, seed=6)).to(bf).cuda(), (0.1 * R(Cc, seed=7)).to(bf).cuda()
        o1, o2 = torch.empty_like(x), torch.empty(B * HW, Cc, device="cuda")
        ops.groupnorm(x, ga, be, o1, B=B, HW=HW, silu=silu)
        ref_ops.groupnorm(x, ga, be, o2, B=B, HW=HW, silu=silu)
        assert rel(o1, o2) < 5e-3
    s = R(130, 1024, s=20.0, seed=8).to(bf).cuda()
    s2 = s.clone().float()
    ops.softmax_rows(s, 0.125)
    ref_ops.softmax_rows(s2, 0.125)
    assert rel(s, s2) < 5e-3
    mom = R(2 * 48, 32, seed=9).to(bf).cuda()
    eps = R(2, 16, 8, 6, seed=10).cuda()
    l1, l2 = torch.empty(2, 16, 8, 6, dtype=bf, device="cuda"), torch.empty(2, 16, 8, 6, device="cuda")
    ops.latent_sample(mom, eps, l1, scale=0.3611, shift=0.1159)
    ref_ops.latent_sample(mom, eps, l2, scale=0.3611, shift=0.1159)
    assert rel(l1, l2) < 5e-3
    img = torch.rand(2, 3, 10, 6, device="cuda") * 2 - 1
    a, b = torch.empty(120, 8, dtype=bf, device="cuda"), torch.empty(120, 8, dtype=bf, device="cuda")
    ops.image_to_nhwc8(img, a)
    ref_ops.image_to_nhwc8(img, b)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


def test_vae_encoder_graph_vs_oracle():
    from ai_toolkit_amd import ops
    from ai_toolkit_amd import vae as nvae
    from oracle import vae_ref

    cfg = dict(latent_channels=16, block_out_channels=(64, 128, 256, 256), layers_per_block=2, groups=32)
    torch.manual_seed(0)
    ref = vae_ref.AutoencoderKLEncoder(**cfg)
    vae_ref.init_synthetic_(ref)
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(p.to(bf).float())
    ref = ref.cuda()
    nat = nvae.AutoencoderKLEncoder(**cfg, dtype=bf, device="cuda", ops=ops)
    nat.load_state_dict({k: v.to(bf) for k, v in ref.state_dict().items()}, strict=True)
    g = torch.Generator().manual_seed(1)
    img = (torch.rand(2, 3, 128, 96, generator=g) * 2 - 1).cuda()
    eps = torch.randn(2, 16, 16, 12, generator=g).cuda()
    with torch.no_grad():
        want = ref.encode_images(img, eps)
        ref16 = ref.to(bf).encode_images(img.to(bf), eps).float()
    got = nat.encode_images(img, eps=eps)
    e_ours, e_ref16 = rel(got, want), rel(ref16, want)
    print(f"VAE latents rel err vs fp32 oracle: ours {e_ours:.4e}, oracle-in-bf16 {e_ref16:.4e}")
    assert e_ours < max(2.0 * e_ref16, 2e-2), (e_ours, e_ref16)
