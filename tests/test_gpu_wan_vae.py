"""GPU parity of the Wan2.1 video-VAE kernels (3-D implicit-GEMM convolution on MFMA, WanRMS_norm rows, per-channel latent sample) and of
the whole encoder graph vs the chunked fp32 oracle."""
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
    dict(T=3, H=12, W=20, Cin=8, Cout=96),                                   # conv_in class (3 -> 8 padded channels, K = 216)
    dict(T=5, H=17, W=15, Cin=96, Cout=96),                                  # ragged pixel count, K = 2592 (40.5 K-tiles)
    dict(T=2, H=16, W=16, Cin=192, Cout=384),                                # 3 N-tiles of 128
    dict(T=1, H=8, W=8, Cin=384, Cout=32),                                   # conv_out (N < tile), a single frame (both pad frames read)
    dict(T=3, H=10, W=6, Cin=192, Cout=192, kt=3, ks=1, tstride=2, pad=0),   # (3,1,1) stride-2 time convolution: 7 input frames
    dict(T=9, H=64, W=64, Cin=96, Cout=256),                                 # 256x256 tile path
])
def test_conv3d_implicit_gemm(case):
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    T, H, W, Cin, Cout = (case[k] for k in ("T", "H", "W", "Cin", "Cout"))
    kt, ks, ts, pad = case.get("kt", 3), case.get("ks", 3), case.get("tstride", 1), case.get("pad", 1)
    Tin = (T - 1) * ts + kt
    x = R(Tin * H * W, Cin, seed=1).to(bf).cuda()
    if ts == 1:
        x[:2 * H * W].zero_()  # the causal zero frames (any data is legal for the kernel; zeros make it the causal convolution)
    w = R(Cout, kt * ks * ks * Cin, s=(kt * ks * ks * Cin) ** -0.5, seed=2).to(bf).cuda()
    bias = R(Cout, s=0.1, seed=3).to(bf).cuda()
    res = R(T * H * W, Cout, seed=4).to(bf).cuda()
    out = torch.full((T * H * W, Cout), float("nan"), dtype=bf, device="cuda")
    ref = torch.empty(T * H * W, Cout, dtype=torch.float32, device="cuda")
    kw = dict(T=T, H=H, W=W, kt=kt, ks=ks, tstride=ts, pad_t=pad, pad_l=pad, bias=bias)
    ops.conv3d(x, w, out, flags=ops.EPI_ADD_AUX, aux_in=res, **kw)
    ref_ops.conv3d(x, w, ref, flags=ref_ops.EPI_ADD_AUX, aux_in=res, **kw)
    torch.cuda.synchronize()
    assert rel(out, ref) < 5e-3, rel(out, ref)


def test_rmsnorm_rows_and_affine_sample_kernels():
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    for (M, Cc, silu) in ((1000, 96, True), (333, 192, True), (130, 384, False), (64, 640, True), (7, 2048, False), (50, 8, True)):
        x = (R(M, Cc, seed=5) * 2 + 0.3).to(bf).cuda()
        ga = (1 + 0.1 * R(Cc, seed=6)).to(bf).cuda()
        o1, o2 = torch.full_like(x, float("nan")), torch.empty(M, Cc, device="cuda")
        ops.rmsnorm_rows(x, ga, o1, silu=silu)
        ref_ops.rmsnorm_rows(x, ga, o2, silu=silu)
        assert rel(o1, o2) < 4e-3, (M, Cc, rel(o1, o2))
    # in place, into a row-offset view (how the graph fills a padded buffer)
    x = R(40, 96, seed=8).to(bf).cuda()
    ga = torch.ones(96, dtype=bf, device="cuda")
    buf = torch.zeros(60, 96, dtype=bf, device="cuda")
    want = torch.empty(40, 96, device="cuda")
    ops.rmsnorm_rows(x, ga, buf[20:], silu=True)
    ref_ops.rmsnorm_rows(x, ga, want, silu=True)
    assert rel(buf[20:], want) < 4e-3 and float(buf[:20].abs().max()) == 0.0
    # zero rows stay zero (eps clamp of F.normalize)
    z = torch.zeros(4, 96, dtype=bf, device="cuda")
    ops.rmsnorm_rows(z, ga, z, silu=False)
    assert float(z.abs().max()) == 0.0
    mom = R(2 * 3 * 20, 32, seed=9).to(bf).cuda()
    eps = R(2, 16, 3, 5, 4, seed=10).cuda()
    sh, sc = R(16, seed=11).cuda(), (R(16, seed=12).abs() + 0.5).cuda()
    l1, l2 = torch.empty(2, 16, 3, 5, 4, dtype=bf, device="cuda"), torch.empty(2, 16, 3, 5, 4, device="cuda")
    ops.latent_sample_affine(mom, eps, l1, ch_shift=sh, ch_scale=sc)
    ref_ops.latent_sample_affine(mom, eps, l2, ch_shift=sh, ch_scale=sc)
    torch.cuda.synchronize()
    assert rel(l1, l2) < 5e-3


def test_wan_vae_encoder_graph_vs_chunked_oracle():
    from ai_toolkit_amd import ops
    from ai_toolkit_amd import wan_vae as nwv
    from oracle import wan_vae_ref

    cfg = dict(base_dim=48, z_dim=16, dim_mult=(1, 2, 4, 4), num_res_blocks=2, temperal_downsample=(False, True, True))
    ref = wan_vae_ref.AutoencoderKLWanEncoder(**cfg)
    wan_vae_ref.init_synthetic_(ref)
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(p.to(bf).float())
    ref = ref.cuda()
    nat = nwv.AutoencoderKLWanEncoder(**cfg, dtype=bf, device="cuda", ops=ops)
    nat.load_state_dict({k: v.to(bf) for k, v in ref.state_dict().items()}, strict=True)
    g = torch.Generator().manual_seed(1)
    clips = [(torch.rand(9, 3, 64, 96, generator=g) * 2 - 1).cuda() for _ in range(2)]
    eps = torch.randn(2, 16, 3, 8, 12, generator=g).cuda()
    with torch.no_grad():
        want = ref.encode_images(clips, eps)
        ref16 = ref.to(bf).encode_images([c.to(bf) for c in clips], eps).float()
    got = nat.encode_images(clips, eps=eps)
    assert got.shape == want.shape == (2, 16, 3, 8, 12) and got.dtype == bf
    e_ours, e_ref16 = rel(got, want), rel(ref16, want)
    print(f"Wan VAE latents rel err vs fp32 chunked oracle: ours {e_ours:.4e}, oracle-in-bf16 {e_ref16:.4e}")
    assert e_ours < max(2.0 * e_ref16, 2e-2), (e_ours, e_ref16)
