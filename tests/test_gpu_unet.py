"""UNet path (SD1.5 / SDXL, BASELINE configs 1-2) on the MI355X through the C ABI: per-kernel parity of the UNet-side kernels, a
train step of scaled-down SD1.5-like / SDXL-like models against the fp32 oracle (next to the rounding-matched oracle and the
reference's bf16 arithmetic, as tests/test_gpu_parity_r2.py does for FLUX), and the full-size SDXL UNet at 1024^2 against the eager
oracle."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
bf = torch.bfloat16

MID_SD15 = dict(block_out_channels=(64, 128, 256, 256), cross_attention_dim=96, attention_head_dim=4, layers_per_block=1, norm_num_groups=16)
MID_SDXL = dict(block_out_channels=(64, 128, 256), cross_attention_dim=96, attention_head_dim=(2, 2, 4), transformer_layers_per_block=(1, 1, 2),
                projection_class_embeddings_input_dim=32 + 6 * 16, addition_time_embed_dim=16, norm_num_groups=16)


def _ok(res):
    assert res.get("ok"), res


def test_unet_side_kernels():
    from tools import gpu_check4 as g

    _ok(g.t_groupnorm())
    _ok(g.t_groupnorm(1, 1024, 1920, 32, False))
    _ok(g.t_groupnorm(3, 300, 64, 16, True))
    _ok(g.t_geglu())
    _ok(g.t_resample())
    _ok(g.t_copy_heads())
    _ok(g.t_copy_heads(100, 8, 40))
    _ok(g.t_ddpm())
    _ok(g.t_ddpm(v=True))
    _ok(g.t_ew_broadcast())


def test_attention_at_unet_head_dims():
    """head_dim 64 (SDXL) / 40 / 80 (SD1.5) through the head_dim-128 flash kernels by exact zero padding, text cross-attention with
    77 keys, and head_dim 160 (SD1.5's coarse levels) through the generic kernels."""
    from tools import gpu_check4 as g

    _ok(g.t_attn_padded())                      # D 64, 1000 queries x 77 keys
    _ok(g.t_attn_padded(1, 8, 1024, 40, 0))     # D 40 self-attention
    _ok(g.t_attn_padded(2, 8, 300, 80, 77))
    _ok(g.t_attn_padded(1, 10, 4096, 64, 0))    # SDXL level-1 self-attention shape
    _ok(g.t_attn_small())                       # D 160, S 256
    _ok(g.t_attn_small(Skv=77))
    _ok(g.t_attn_small(1, 8, 64, 160, 0))


def test_conv_forward_and_data_gradient():
    from tools import gpu_check4 as g

    _ok(g.t_conv(dgrad=True))
    _ok(g.t_conv(stride=2, dgrad=True))
    _ok(g.t_conv(res=True))
    _ok(g.t_conv(2, 64, 64, 8, 320))                    # conv_in (4 latent channels padded to 8)
    _ok(g.t_conv(2, 64, 64, 320, 4, dgrad=True))        # conv_out (N = 4)
    _ok(g.t_conv(1, 32, 32, 1920, 640, dgrad=True))     # up-block resnet on a concatenated skip
    _ok(g.t_conv(1, 128, 128, 320, 320, dgrad=True))    # SDXL 1024^2 first level


def _pair(cfg_over, sdxl, dev="cuda", rank=8, alpha=4.0):
    import ai_toolkit_amd  # noqa: F401
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.lora import FusedLoRANetwork
    from ai_toolkit_amd.unet import UNet2DConditionModel
    from oracle import lora_ref, unet_ref

    cfg = dict(unet_ref.SDXL if sdxl else unet_ref.SD15, **cfg_over)
    torch.manual_seed(0)
    ref = unet_ref.UNet2DConditionModel(**cfg)
    unet_ref.init_synthetic_(ref, seed=11)
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(p.to(bf).float())  # bf16-representable base so every path sees identical weights
    ref = ref.to(dev)

    def native(table, shadow_dtype=None):
        nat = UNet2DConditionModel(**cfg, dtype=bf, device=dev, ops=table)
        nat.load_state_dict({k: v.to(bf) for k, v in ref.state_dict().items()}, strict=True)
        net = FusedLoRANetwork(nat, lora_dim=rank, alpha=alpha, target_lin_modules=("Transformer2DModel",), is_transformer=False,
                               peft_format=False, transformer_only=False)
        return nat, net

    ref_net = lora_ref.RefLoRANetwork(ref, rank, target=("Transformer2DModel",), kohya_unet=True, alpha=alpha).to(dev)
    ref_net.torch_multiplier = ref_net.torch_multiplier.to(dev)
    g = torch.Generator().manual_seed(7)
    ups = [torch.randn(m.lora_up.weight.shape, generator=g) * 0.03 for m in ref_net.unet_loras]
    with torch.no_grad():
        for m, u in zip(ref_net.unet_loras, ups):
            m.lora_up.weight.copy_(u)
    ref_net.apply_to()

    def finish(nat, net, table, shadow_dtype=None):
        with torch.no_grad():
            for a, b in zip(net.unet_loras, ref_net.unet_loras):
                a.lora_down.weight.copy_(b.lora_down.weight.detach().cpu().reshape(a.lora_down.weight.shape))
                a.lora_up.weight.copy_(b.lora_up.weight.detach().cpu().reshape(a.lora_up.weight.shape))
        net.apply_to()
        net.build_arena(dev, groups=nat.lora_groups(), shadow_dtype=shadow_dtype)
        net.refresh_shadows(table)
        nat.attach_network(net)
        nat.prepare()
        return nat, net

    return cfg, ref, ref_net, native, finish, ops


def _batch(cfg, B=2, h=32, w=24, n_txt=77, seed=5, dev="cuda"):
    g = torch.Generator().manual_seed(seed)
    lat = torch.randn(B, 4, h, w, generator=g).to(bf)
    ctx = (torch.randn(B, n_txt, cfg["cross_attention_dim"], generator=g) * 0.7).to(bf)
    pooled = (torch.randn(B, 32, generator=g) * 0.7).to(bf)
    noise = torch.randn(B, 4, h, w, generator=g).to(bf)
    ts = torch.tensor([640, 17, 998, 2][:B])
    return [t.to(dev) for t in (lat, ctx, pooled, noise, ts)]


def _rel_lists(a, b):
    num = sum(((x.float().reshape(-1) - y.float().reshape(-1)) ** 2).sum().item() for x, y in zip(a, b))
    den = sum((y.float() ** 2).sum().item() for y in b)
    return math.sqrt(num / max(den, 1e-300))


@pytest.mark.parametrize("sdxl", [False, True], ids=["sd15", "sdxl"])
def test_unet_step_four_way_parity(sdxl):
    from ai_toolkit_amd.trainer import UNetLoRATrainStep
    from oracle import ref_ops, train_ref

    cfg, ref, ref_net, native, finish, ops = _pair(MID_SDXL if sdxl else MID_SD15, sdxl)
    nat, net = finish(*native(ops), ops)
    rm, rm_net = finish(*native(ref_ops), ref_ops, shadow_dtype=torch.float32)
    lat, ctx, pooled, noise, ts = _batch(cfg)
    kw = dict(lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    oracle = train_ref.RefUNetTrainStep(ref, ref_net, min_snr_gamma=5.0, **kw)
    l32 = oracle.step(lat.float(), ctx.float(), pooled.float(), noise.float(), ts).item()
    g32 = [p.grad.clone() for p in oracle.params]
    ref.to(bf)
    l16 = oracle.step(lat, ctx, pooled, noise, ts, dtype=bf).item()
    g16 = [p.grad.clone() for p in oracle.params]
    ref.float()

    def grads(n):
        return [p.grad.detach().clone() for m in n.unet_loras for p in (m.lora_down.weight, m.lora_up.weight)]

    lrm = UNetLoRATrainStep(rm, rm_net, ref_ops, min_snr_gamma=5.0, **kw).step(lat, ctx, pooled, noise=noise, timesteps=ts).item()
    grm = grads(rm_net)
    lo = UNetLoRATrainStep(nat, net, ops, min_snr_gamma=5.0, **kw).step(lat, ctx, pooled, noise=noise, timesteps=ts).item()
    go = grads(net)
    e = {"ours_vs_fp32": _rel_lists(go, g32), "ref16_vs_fp32": _rel_lists(g16, g32), "rm16_vs_fp32": _rel_lists(grm, g32),
         "ours_vs_rm16": _rel_lists(go, grm), "ref16_vs_rm16": _rel_lists(g16, grm)}
    print(f"PARITY4 unet {'sdxl' if sdxl else 'sd15'}-mid: loss ours {lo:.6f} rm16 {lrm:.6f} ref16 {l16:.6f} fp32 {l32:.6f}; adapter-gradient rel err " +
          " ".join(f"{k}={v:.3e}" for k, v in e.items()))
    assert math.isfinite(lo) and abs(lo - l32) <= 2e-3 * abs(l32), (lo, l32, l16)
    assert e["ours_vs_fp32"] <= 1.3 * max(e["ref16_vs_fp32"], e["rm16_vs_fp32"]) + 1e-3, e
    assert e["ours_vs_rm16"] <= max(e["ref16_vs_rm16"], 6e-3), e


def test_unet_three_steps_and_kohya_file(tmp_path):
    """three AdamW steps on the device follow the oracle's losses; the saved kohya file carries the trained weights."""
    from safetensors.torch import load_file

    from ai_toolkit_amd.trainer import UNetLoRATrainStep
    from oracle import train_ref

    cfg, ref, ref_net, native, finish, ops = _pair(MID_SD15, False)
    nat, net = finish(*native(ops), ops)
    kw = dict(lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
    oracle = train_ref.RefUNetTrainStep(ref, ref_net, **kw)
    ours = UNetLoRATrainStep(nat, net, ops, **kw)
    for k in range(3):
        lat, ctx, pooled, noise, ts = _batch(cfg, seed=30 + k)
        l32 = oracle.step(lat.float(), ctx.float(), pooled.float(), noise.float(), ts).item()
        l = ours.step(lat, ctx, pooled, noise=noise, timesteps=ts).item()
        assert abs(l - l32) <= 3e-3 * abs(l32), (k, l, l32)
    f = tmp_path / "sd15_lora.safetensors"
    net.save_weights(str(f), dtype=torch.float16, metadata={"name": "t"})
    sd = load_file(str(f))
    m0 = net.unet_loras[0]
    assert m0.lora_name == "lora_unet_down_blocks_0_attentions_0_proj_in" and sd[f"{m0.lora_name}.lora_down.weight"].dim() == 4
    assert torch.equal(sd[f"{m0.lora_name}.lora_up.weight"][:, :, 0, 0], m0.lora_up.weight.detach().cpu().to(torch.float16))


def test_full_size_sdxl_unet_at_1024_vs_eager_oracle():
    """BASELINE config 2 itself: the 2.57 B-parameter SDXL UNet, 128x128 latents (1024^2), 77 text tokens, LoRA r8 on 722 layers, B = 1.
    ours -> loss and every adapter gradient, then the eager oracle on the same weights in bf16 (the reference's arithmetic, fp32 adapter)
    and in fp32."""
    import ai_toolkit_amd  # noqa: F401
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.lora import FusedLoRANetwork
    from ai_toolkit_amd.trainer import UNetLoRATrainStep
    from ai_toolkit_amd.unet import SDXL_CONFIG, UNet2DConditionModel
    from oracle import lora_ref, train_ref, unet_ref

    dev = "cuda"
    torch.set_default_dtype(bf)
    try:
        with torch.device(dev):
            ref = unet_ref.UNet2DConditionModel(**unet_ref.SDXL)
    finally:
        torch.set_default_dtype(torch.float32)
    g = torch.Generator(device=dev).manual_seed(1234)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, (torch.nn.Linear, torch.nn.Conv2d)):
                fan = m.weight[0].numel()
                m.weight.copy_((torch.randn(m.weight.shape, device=dev, generator=g) / math.sqrt(fan)).to(bf))
                if m.bias is not None:
                    m.bias.copy_((torch.randn(m.bias.shape, device=dev, generator=g) * 0.01).to(bf))
    for p in ref.parameters():
        p.requires_grad_(False)
    nat = UNet2DConditionModel(**SDXL_CONFIG, dtype=bf, device=dev, ops=ops)
    nat.load_state_dict(ref.state_dict(), strict=True)
    torch.manual_seed(5)
    net = FusedLoRANetwork(nat, lora_dim=8, alpha=8.0, target_lin_modules=("Transformer2DModel",), is_transformer=False, peft_format=False,
                           transformer_only=False)
    assert len(net.unet_loras) == 722
    with torch.no_grad():
        for m in net.unet_loras:
            m.lora_up.weight.normal_(0, 2e-3)
    net.apply_to()
    net.build_arena(dev, groups=nat.lora_groups())
    net.refresh_shadows(ops)
    nat.attach_network(net)
    nat.prepare()
    ref_net = lora_ref.RefLoRANetwork(ref, 8, target=("Transformer2DModel",), kohya_unet=True, alpha=8.0).to(dev)
    ref_net.torch_multiplier = ref_net.torch_multiplier.to(dev)
    with torch.no_grad():
        for a, b in zip(net.unet_loras, ref_net.unet_loras):
            assert a.lora_name == b.lora_name
            b.lora_down.weight.copy_(a.lora_down.weight.reshape(b.lora_down.weight.shape))
            b.lora_up.weight.copy_(a.lora_up.weight.reshape(b.lora_up.weight.shape))
    ref_net.apply_to()
    gg = torch.Generator(device=dev).manual_seed(42)
    lat = torch.randn(1, 4, 128, 128, device=dev, generator=gg).to(bf)
    ctx = (torch.randn(1, 77, 2048, device=dev, generator=gg) * 0.5).to(bf)
    pooled = (torch.randn(1, 1280, device=dev, generator=gg) * 0.5).to(bf)
    noise = torch.randn(1, 4, 128, 128, device=dev, generator=gg).to(bf)
    ts = torch.tensor([500], device=dev)
    kw = dict(lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    lo = UNetLoRATrainStep(nat, net, ops, **kw).step(lat, ctx, pooled, noise=noise, timesteps=ts).item()
    go = [p.grad.detach().clone() for m in net.unet_loras for p in (m.lora_down.weight, m.lora_up.weight)]
    assert math.isfinite(lo)
    del nat
    torch.cuda.empty_cache()
    oracle = train_ref.RefUNetTrainStep(ref, ref_net, **kw)
    l16 = oracle.step(lat, ctx, pooled, noise, ts, dtype=bf).item()
    g16 = [p.grad.clone() for p in oracle.params]
    ref.float()
    l32 = oracle.step(lat.float(), ctx.float(), pooled.float(), noise.float(), ts).item()
    g32 = [p.grad.clone() for p in oracle.params]
    e_o, e_16, e_o16 = _rel_lists(go, g32), _rel_lists(g16, g32), _rel_lists(go, g16)
    print(f"PARITY full-size SDXL @1024^2 B=1: loss ours {lo:.6f} ref16 {l16:.6f} fp32 {l32:.6f} (rel ours {abs(lo - l32) / l32:.2e}, ref16 "
          f"{abs(l16 - l32) / l32:.2e}); adapter-gradient rel err ours_vs_fp32 {e_o:.3e} ref16_vs_fp32 {e_16:.3e} ours_vs_ref16 {e_o16:.3e}; "
          f"peak mem {torch.cuda.max_memory_allocated() / 2 ** 30:.0f} GiB")
    assert abs(lo - l32) <= 2e-3 * abs(l32), (lo, l32, l16)
    assert e_o <= 1.3 * e_16 + 2e-3, (e_o, e_16)
