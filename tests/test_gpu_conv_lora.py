"""`network.conv` adapters on the MI355X: the kernel pieces (3x3-conv lora_down with the split-slab epilogue, K-slab inside the implicit-GEMM
convolution, accumulate epilogue of the slab data gradient, zero frame, strided per-tap weight gradient, shadow kind 4) against the oracle
table, and a train step of mid-size SD1.5-like / SDXL-like UNets with conv adapters: the HIP path vs the SAME host graph driven by the oracle's
torch kernels in fp32 (that graph is pinned on the CPU to the reference's own LoRASpecialNetwork run, tests/test_unet_conv_lora_golden.py)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


def R(*shape, s=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * s


@pytest.mark.parametrize("case", [dict(B=2, H=16, W=12, Cin=64, stride=1), dict(B=1, H=33, W=31, Cin=32, stride=1),
                                  dict(B=2, H=16, W=24, Cin=128, stride=2), dict(B=1, H=64, W=64, Cin=320, stride=1),
                                  dict(B=2, H=16, W=12, Cin=64, stride=1, rp=32), dict(B=1, H=33, W=31, Cin=32, stride=1, rp=48),
                                  dict(B=2, H=16, W=24, Cin=128, stride=2, rp=64)])
def test_conv_lora_down_split_slab(case):
    """rp = padded rank of the adapter: the stacked filter is rp / 16 blocks [A_hi(16) ; A_lo(16)] (N = 2 rp GEMM columns), the slab
    [hi(rp) | lo(rp) | hi(rp)]; checked against the oracle AND against the plain fp32 convolution with the un-split filter."""
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    B, H, W, Cin, stride = (case[k] for k in ("B", "H", "W", "Cin", "stride"))
    rp = case.get("rp", 16)
    x = R(B * H * W, Cin, seed=1).to(bf).cuda()
    a32 = R(rp, 9 * Cin, s=(9 * Cin) ** -0.5, seed=2)
    hi = a32.to(bf)
    lo = (a32 - hi.float()).to(bf)
    w = torch.stack((hi.view(rp // 16, 16, -1), lo.view(rp // 16, 16, -1)), dim=1).reshape(2 * rp, -1).contiguous().cuda()
    cs = torch.full((2 * rp,), 0.37, dtype=torch.float32, device="cuda")
    kw = dict(B=B, H=H, W=W)
    if stride == 2:
        kw.update(stride=2, Ho=H // 2, Wo=W // 2)
    M = B * kw.get("Ho", H) * kw.get("Wo", W)
    out = torch.full((M, 3 * rp), float("nan"), dtype=bf, device="cuda")
    ref = torch.empty(M, 3 * rp, dtype=torch.float32, device="cuda")
    ops.conv3x3(x, w, out, split_slab=True, col_scale=cs, **kw)
    ref_ops.conv3x3(x, w, ref, split_slab=True, col_scale=cs, **kw)
    torch.cuda.synchronize()
    assert torch.equal(out[:, :rp], out[:, 2 * rp:])
    t_ours = out[:, :rp].float() + out[:, rp:2 * rp].float()
    t_ref = ref[:, :rp] + ref[:, rp:2 * rp]  # the fp32 value before the split (hi + lo of the oracle is exact to 2^-17)
    assert rel(t_ours, t_ref) < 5e-5, rel(t_ours, t_ref)
    assert rel(out[:, :rp], ref[:, :rp]) < 4e-3
    plain = torch.empty(M, rp, dtype=torch.float32, device="cuda")  # rank r of the slab = filter r of the adapter, whatever the block order
    ref_ops.conv3x3(x.float(), (hi.float() + lo.float()).cuda(), plain, **kw)
    assert rel(t_ours, 0.37 * plain) < 5e-5, rel(t_ours, 0.37 * plain)


def test_conv_with_lora_up_slab_accumulate_pad_and_strided_wgrad():
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    B, H, W, Cin, Cout = 2, 20, 12, 64, 96
    x = R(B * H * W, Cin, seed=1).to(bf).cuda()
    w = R(Cout, 9 * Cin, s=(9 * Cin) ** -0.5, seed=2).to(bf).cuda()
    bias = R(Cout, s=0.1, seed=3).to(bf).cuda()
    res = R(B * H * W, Cout, seed=4).to(bf).cuda()
    T = R(B * H * W, 48, s=0.5, seed=5).to(bf).cuda()
    b3 = R(Cout, 48, s=0.2, seed=6).to(bf).cuda()
    out = torch.full((B * H * W, Cout), float("nan"), dtype=bf, device="cuda")
    ref = torch.empty(B * H * W, Cout, dtype=torch.float32, device="cuda")
    kw = dict(B=B, H=H, W=W, bias=bias, flags=ops.EPI_ADD_AUX, aux_in=res, a2=T, b2=b3)
    ops.conv3x3(x, w, out, **kw)
    ref_ops.conv3x3(x, w, ref, **kw)
    assert rel(out, ref) < 5e-3, rel(out, ref)
    # data gradient of the adapter: 3x3 convolution over the 48-channel slab image, accumulated onto the base data gradient
    wd = R(Cin, 9 * 48, s=0.05, seed=7).to(bf).cuda()
    dx = R(B * H * W, Cin, seed=8).to(bf).cuda()
    dref = dx.float().clone()
    ops.conv3x3(T, wd, dx, B=B, H=H, W=W, flags=2)
    ref_ops.conv3x3(T, wd, dref, B=B, H=H, W=W, flags=2)
    assert rel(dx, dref) < 5e-3, rel(dx, dref)
    # zero frame
    xp, xr = torch.full((B * (H + 2) * (W + 2), Cin), float("nan"), dtype=bf, device="cuda"), torch.empty(B * (H + 2) * (W + 2), Cin, dtype=bf, device="cuda")
    ops.pad_nhwc(x, xp, B=B, H=H, W=W)
    ref_ops.pad_nhwc(x, xr, B=B, H=H, W=W)
    assert torch.equal(xp, xr)
    # one tap of the [r, Cin, 3, 3] gradient: strided destination, shifted row windows of the framed grids
    Tp = torch.empty(B * (H + 2) * (W + 2), 48, dtype=bf, device="cuda")
    ops.pad_nhwc(T, Tp, B=B, H=H, W=W)
    g1 = torch.zeros(16 * Cin * 9, dtype=torch.float32, device="cuda")
    g2 = torch.zeros(16 * Cin * 9, dtype=torch.float32, device="cuda")
    Np, Wp = Tp.shape[0], W + 2
    for tap in range(9):
        sh = (tap // 3 - 1) * Wp + (tap % 3 - 1)
        j0, j1 = max(0, -sh), min(Np, Np - sh)
        for table, g in ((ops, g1), (ref_ops, g2)):
            table.lora_wgrad(Tp[j0:j1], xp[j0 + sh:j1 + sh], g[tap:], accumulate=True, M=j1 - j0, split=16, out_strides=(9 * Cin, 9))
    torch.cuda.synchronize()
    assert rel(g1, g2) < 2e-4, rel(g1, g2)
    # ... and it IS the Conv2d weight gradient: autograd of conv2d(x, A) . dT with dT = hi + lo of the slab
    A = torch.zeros(16, Cin, 3, 3, device="cuda", requires_grad=True)
    xi = x.float().view(B, H, W, Cin).permute(0, 3, 1, 2)
    dT = (T[:, :16].float() + T[:, 16:32].float()).view(B, H, W, 16).permute(0, 3, 1, 2)
    (torch.nn.functional.conv2d(xi, A, padding=1) * dT).sum().backward()
    assert rel(g1.view(16, Cin, 3, 3), A.grad) < 2e-4


@pytest.mark.parametrize("rp", [16, 48])
def test_shadow_kind4_layouts_bit_equal(rp):
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    cin = 24
    arena = R(rp * cin * 9 + 7, s=0.3, seed=11).cuda()
    ent = [(0, rp, cin * 9, 4, 0, 2 * rp * cin * 9, 0, cin)]
    n = 5 * rp * cin * 9
    s1 = torch.zeros(n, dtype=bf, device="cuda")
    s2 = torch.zeros(n, dtype=bf, device="cuda")
    ops.refresh_shadows(arena, s1, ops.make_shadow_table(ent, "cuda"))
    ref_ops.refresh_shadows(arena, s2, ref_ops.make_shadow_table(ent, "cuda"))
    torch.cuda.synchronize()
    assert torch.equal(s1, s2)


MID_SD15 = dict(block_out_channels=(64, 128, 256, 256), cross_attention_dim=96, attention_head_dim=4, layers_per_block=1, norm_num_groups=16)
MID_SDXL = dict(block_out_channels=(64, 128, 256), cross_attention_dim=96, attention_head_dim=(2, 2, 4), transformer_layers_per_block=(1, 1, 2),
                projection_class_embeddings_input_dim=32 + 6 * 16, addition_time_embed_dim=16, norm_num_groups=16)


def _rel_lists(a, b):
    num = sum(((x.float().reshape(-1) - y.float().reshape(-1)) ** 2).sum().item() for x, y in zip(a, b))
    den = sum((y.float() ** 2).sum().item() for y in b)
    return math.sqrt(num / max(den, 1e-300))


@pytest.mark.parametrize("M,rp,rpb,tm_rpb", [(1000, 16, 250, 0), (777, 48, 0, 111), (640, 32, 160, 160)])
def test_slab_rescale_vs_oracle(M, rp, rpb, tm_rpb):
    """aitk_slab_rescale: per-sample row factor and dropout / rank masks on a conv adapter's [hi | lo | hi] slab, in place."""
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    v = R(M, rp, s=0.7, seed=3)
    hi = v.to(bf)
    lo = (v - hi.float()).to(bf)
    T0 = torch.cat((hi, lo, hi, torch.full((M, 8), 5.0).to(bf)), 1).cuda()  # 8 guard columns behind the slab
    mult = torch.tensor([1.0, -0.5, 2.0, 0.25, 3.0][: (M + rpb - 1) // rpb], device="cuda") if rpb else None
    nrows = (M + tm_rpb - 1) // tm_rpb if tm_rpb else M
    tmask = ((R(nrows, rp, seed=4) > 0).float() / 0.5).cuda() if (tm_rpb or rpb == 250) else None
    a, b = T0.clone(), T0.clone()
    ops.slab_rescale(a, rp, mult=mult, rows_per_batch=rpb, tmask=tmask, tmask_rows_per_batch=tm_rpb, M=M)
    ref_ops.slab_rescale(b, rp, mult=mult, rows_per_batch=rpb, tmask=tmask, tmask_rows_per_batch=tm_rpb, M=M)
    torch.cuda.synchronize()
    assert torch.equal(a[:, 3 * rp:], T0[:, 3 * rp:])
    va, vb = a[:, :rp].float() + a[:, rp:2 * rp].float(), b[:, :rp].float() + b[:, rp:2 * rp].float()
    assert rel(va, vb) < 1e-6 and torch.equal(a[:, :rp], a[:, 2 * rp:3 * rp]) and rel(a[:, :rp], b[:, :rp]) < 1e-6


@pytest.mark.parametrize("sdxl,lin_r,conv_r,drop", [(False, 8, 4, False), (True, 8, 4, False), (True, 80, 24, False), (False, 8, 40, False), (True, 8, 8, True)],
                         ids=["sd15", "sdxl", "sdxl-lin80-conv24", "sd15-conv40", "sdxl-dropout-per-sample-multipliers"])
def test_unet_step_with_conv_adapters(sdxl, lin_r, conv_r, drop):
    import ai_toolkit_amd  # noqa: F401
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.lora import FusedLoRANetwork
    from ai_toolkit_amd.trainer import UNetLoRATrainStep
    from ai_toolkit_amd.unet import UNet2DConditionModel
    from oracle import ref_ops, unet_ref
    from tests.test_gpu_unet import _batch

    dev = "cuda"
    cfg = dict(unet_ref.SDXL if sdxl else unet_ref.SD15, **(MID_SDXL if sdxl else MID_SD15))
    torch.manual_seed(0)
    ref = unet_ref.UNet2DConditionModel(**cfg)
    unet_ref.init_synthetic_(ref, seed=11)
    sd = {k: v.to(bf) for k, v in ref.state_dict().items()}

    def make(table, dtype, shadow_dtype=None):
        nat = UNet2DConditionModel(**cfg, dtype=dtype, device=dev, ops=table)
        nat.load_state_dict({k: v.to(dtype) for k, v in sd.items()}, strict=True)
        torch.manual_seed(99)
        dkw = dict(dropout=0.1, rank_dropout=0.25, module_dropout=0.1) if drop else {}
        net = FusedLoRANetwork(nat, lora_dim=lin_r, alpha=lin_r / 2, conv_lora_dim=conv_r, conv_alpha=conv_r / 2, target_lin_modules=("Transformer2DModel",),
                               is_transformer=False, peft_format=False, transformer_only=False, **dkw)
        if drop:  # one keyed source of uniforms for both graphs; training mode; one multiplier per sample (slider training)
            import hashlib

            net.mask_provider = lambda name, kind, shape, device: torch.rand(
                shape, generator=torch.Generator().manual_seed(int(hashlib.sha256(f"{name}/{kind}".encode()).hexdigest()[:8], 16))).to(
                "cpu" if kind == "module" else device)
            net.train()
        g = torch.Generator().manual_seed(7)
        with torch.no_grad():
            for m in net.unet_loras:
                m.lora_up.weight.copy_(torch.randn(m.lora_up.weight.shape, generator=g) * 0.03)
        net.apply_to()
        net.build_arena(dev, groups=nat.lora_groups(), shadow_dtype=shadow_dtype)
        net.refresh_shadows(table)
        nat.attach_network(net)
        nat.prepare()
        return nat, net

    nat, net = make(ops, bf)
    r32, r32_net = make(ref_ops, torch.float32, shadow_dtype=torch.float32)
    assert any(m.is_conv3x3 for m in net.unet_loras) and [m.lora_name for m in net.unet_loras] == [m.lora_name for m in r32_net.unet_loras]
    lat, ctx, pooled, noise, ts = _batch(cfg)
    if drop:
        mult = [1.0, -0.5, 0.75, 1.5][: lat.shape[0]]
        net.multiplier = mult
        r32_net.multiplier = mult
    kw = dict(lr=0.0, weight_decay=0.0, max_grad_norm=0.0, min_snr_gamma=5.0)
    l32 = UNetLoRATrainStep(r32, r32_net, ref_ops, **kw).step(lat.float(), ctx.float(), pooled.float(), noise=noise.float(), timesteps=ts).item()
    lo = UNetLoRATrainStep(nat, net, ops, **kw).step(lat, ctx, pooled, noise=noise, timesteps=ts).item()

    def grads(n, pred):
        return [p.grad.detach().clone() for m in n.unet_loras if pred(m) for p in (m.lora_down.weight, m.lora_up.weight)]

    kinds = {"all": lambda m: True, "conv3x3": lambda m: m.is_conv3x3, "time_emb_proj": lambda m: "time_emb_proj" in m.lora_name,
             "conv_shortcut": lambda m: "conv_shortcut" in m.lora_name, "transformer": lambda m: "attentions" in m.lora_name}
    e = {k: _rel_lists(grads(net, f), grads(r32_net, f)) for k, f in kinds.items()}
    print(f"CONV-LORA {'sdxl' if sdxl else 'sd15'}-mid: loss ours {lo:.6f} host-graph fp32 {l32:.6f}; adapter-gradient rel err vs fp32 " +
          " ".join(f"{k}={v:.3e}" for k, v in e.items()))
    assert math.isfinite(lo) and abs(lo - l32) <= 2e-3 * abs(l32), (lo, l32)
    # the bf16 floor of this graph (profiles/r02_pytest_gpu_e.log: 2.0e-2 / 2.1e-2 for the Transformer2DModel-only networks) + slack
    assert e["all"] <= 4e-2 and e["transformer"] <= 4e-2, e
    assert e["conv3x3"] <= 5e-2 and e["time_emb_proj"] <= 5e-2 and e["conv_shortcut"] <= 5e-2, e
