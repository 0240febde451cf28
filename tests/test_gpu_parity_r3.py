"""Round-3 parity ledger: every BASELINE config at its own size on the HIP path, against the eager oracle on identical weights / inputs.

  config 4   Wan2.1-T2V-1.3B: 30 blocks, d = 1536, 13 x 64 x 64 latents (13 312 tokens), 512 text tokens, LoRA r16 on 300 Linears, B = 1
             (reference toolkit/models/wan21/wan21.py:578-603, 717-724)
  config 5   FLUX.1-dev fp8 (e4m3 weight-only) base, LoRA r32, 19 + 38 blocks @1024^2 (toolkit/util/quantize.py:43-75)
  config 1   SD1.5 UNet (859.5 M parameters), 64 x 64 latents (512^2), LoRA r4 on 192 layers — the architecture of the reference's own
             CPU-runnable config, on the GPU
  config 3   three AdamW steps at full FLUX depth: the LoRA delta  B'A' - BA  per module, four ways (ours / rm16 / ref16 / fp32)

Assertions are the ones of tests/test_gpu_parity_r2.py: loss within north_star's 1e-3 of the fp32 oracle; adapter gradients / deltas not
worse than the reference's own bf16 arithmetic (ref16) against the same fp32 truth — bf16 storage bounds any bf16 execution at ~1e-2 on
gradients (profiles/r03_residual_stream_experiment_*.json); every number is printed as a `PARITY` line."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


def _rel_lists(a, b):
    num = sum(((x.float() - y.float()) ** 2).sum().item() for x, y in zip(a, b))
    den = sum((y.float() ** 2).sum().item() for y in b)
    return math.sqrt(num / max(den, 1e-300))


def _ckpt(blocks):
    from torch.utils.checkpoint import checkpoint

    for blk in blocks:
        f = blk.forward
        blk.forward = (lambda *a, _f=f: checkpoint(_f, *a, use_reentrant=False))


def _grads(net):
    out = []
    for m in net.unet_loras:
        out += [m.lora_down.weight.grad.detach().clone(), m.lora_up.weight.grad.detach().clone()]
    return out


# ------------------------------------------------------------------------------------------------------------------ config 4
def test_wan21_config4_geometry_vs_eager_oracle():
    import ai_toolkit_amd  # noqa: F401
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.lora import FusedLoRANetwork
    from ai_toolkit_amd.trainer import WanLoRATrainStep
    from ai_toolkit_amd.wan import WanTransformer3DModel
    from oracle import lora_ref, wan_ref
    from tests.test_gpu_wan import _oracle_step

    dev = "cuda"
    torch.manual_seed(0)
    torch.set_default_dtype(bf)
    try:
        with torch.device(dev):
            ref = wan_ref.WanTransformer3DModel()  # 1.3B defaults: 12 heads x 128, ffn 8960, 30 layers, text 4096
    finally:
        torch.set_default_dtype(torch.float32)
    g = torch.Generator(device=dev).manual_seed(99)
    with torch.no_grad():
        for name, p in ref.named_parameters():
            if "norm_q" in name or "norm_k" in name or name.endswith("norm2.weight"):
                p.copy_((1 + 0.1 * torch.randn(p.shape, device=dev, generator=g)).to(bf))
            elif name.endswith("norm2.bias"):
                p.copy_((0.05 * torch.randn(p.shape, device=dev, generator=g)).to(bf))
            elif "scale_shift_table" in name:
                p.copy_((torch.randn(p.shape, device=dev, generator=g) / p.shape[-1] ** 0.5).to(bf))
            elif p.ndim >= 2:
                p.copy_((torch.randn(p.shape, device=dev, generator=g) * min(0.06, (1.0 / p[0].numel()) ** 0.5)).to(bf))
            else:
                p.copy_((torch.randn(p.shape, device=dev, generator=g) * 0.02).to(bf))
    for p in ref.parameters():
        p.requires_grad_(False)
    nat = WanTransformer3DModel(dtype=bf, device=dev, ops=ops)
    nat.load_state_dict(ref.state_dict(), strict=True)
    torch.manual_seed(7)
    net = FusedLoRANetwork(nat, lora_dim=16, target_lin_modules=("WanTransformer3DModel",), transformer_block_names=["blocks"],
                           base_model_version="wan_2.1")
    assert len(net.unet_loras) == 300
    with torch.no_grad():
        for m in net.unet_loras:
            m.lora_up.weight.normal_(0, 2e-3)
    net.apply_to()
    net.build_arena(dev, groups=nat.lora_groups())
    net.refresh_shadows(ops)
    nat.attach_network(net)
    nat.prepare()
    ref_net = lora_ref.RefLoRANetwork(ref, 16, target=("WanTransformer3DModel",), block_names=("blocks",)).to(dev)
    ref_net.torch_multiplier = ref_net.torch_multiplier.to(dev)
    with torch.no_grad():
        for a, b in zip(net.unet_loras, ref_net.unet_loras):
            assert a.lora_name == b.lora_name
            b.lora_down.weight.copy_(a.lora_down.weight)
            b.lora_up.weight.copy_(a.lora_up.weight)
    ref_net.apply_to()
    gg = torch.Generator(device=dev).manual_seed(5)
    lat = torch.randn(1, 16, 13, 64, 64, device=dev, generator=gg).to(bf)
    noise = torch.randn(1, 16, 13, 64, 64, device=dev, generator=gg).to(bf)
    txt = (torch.randn(1, 512, 4096, device=dev, generator=gg) * 0.3).to(bf)
    ts = torch.tensor([500.0], device=dev)
    lo = WanLoRATrainStep(nat, net, ops, lr=0.0, weight_decay=0.0, max_grad_norm=0.0).step(lat, txt, noise=noise, timesteps=ts).item()
    go = _grads(net)
    assert math.isfinite(lo)
    del nat
    torch.cuda.empty_cache()
    _ckpt(ref.blocks)
    l16, g16 = _oracle_step(ref, ref_net, lat, txt, noise, ts, bf)
    ref.float()
    torch.cuda.empty_cache()
    l32, g32 = _oracle_step(ref, ref_net, lat, txt, noise, ts, torch.float32)
    e_o, e_16 = _rel_lists(go, g32), _rel_lists(g16, g32)
    worst_o = max(_rel_lists([a], [b]) for a, b in zip(go, g32))
    worst_16 = max(_rel_lists([a], [b]) for a, b in zip(g16, g32))
    print(f"PARITY Wan2.1-1.3B config 4 (30 blocks, 13x64x64 latents = 13312 tokens, B=1): loss ours {lo:.6f} ref16 {l16:.6f} fp32 {l32:.6f} "
          f"(rel ours {abs(lo - l32) / l32:.2e}, ref16 {abs(l16 - l32) / l32:.2e}); adapter-gradient rel err ours_vs_fp32 {e_o:.3e} "
          f"ref16_vs_fp32 {e_16:.3e}; worst module ours {worst_o:.3e} ref16 {worst_16:.3e}; peak mem {torch.cuda.max_memory_allocated() / 2 ** 30:.0f} GiB")
    assert abs(lo - l32) <= 1e-3 * abs(l32), (lo, l32, l16)
    assert e_o <= 1.25 * e_16 + 1e-3, (e_o, e_16)
    assert worst_o <= 1.5 * worst_16 + 5e-3, (worst_o, worst_16)


# ------------------------------------------------------------------------------------------------------------------ config 5
def test_flux_fp8_base_r32_full_depth_at_1024_vs_eager_oracle():
    """The weights the oracle multiplies with are bf16(e4m3 code x per-channel scale) of the product's codes; the codes / scales
    themselves are pinned to the independent numpy quantiser at small size (tests/test_gpu_fp8.py) and spot-checked here."""
    import numpy as np

    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from oracle import flux_ref, fp8_ref, lora_ref, train_ref
    from tests.test_gpu_fullsize import _batch, _flux
    from tests.test_gpu_parity_r2 import _checkpoint_blocks

    model, net, ops = _flux(19, 38, rank=32)
    assert len(net.unet_loras) == 494 and net.unet_loras[0].lora_dim == 32
    lin0 = model.transformer_blocks[3].attn.to_q
    w0 = lin0.weight.detach().float().cpu().numpy()
    model.quantize_base_fp8()
    codes, scale = fp8_ref.quantize_per_channel(w0)  # spot check of one full-size layer against the numpy restatement
    assert np.allclose(lin0.wscale.cpu().numpy(), scale, rtol=2.5e-7, atol=0)
    assert (lin0.qweight.cpu().numpy() != codes).mean() <= 1e-2
    lat, emb, pooled, noise, ts = _batch(1)
    kw = dict(lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    lo = FluxLoRATrainStep(model, net, ops, **kw).step(lat, emb, pooled, noise=noise, timesteps=ts).item()
    go = _grads(net)
    assert math.isfinite(lo)
    sd = model.state_dict()
    for lin_name, lin in model.named_modules():
        if getattr(lin, "qweight", None) is not None:
            sd[lin_name + ".weight"] = model.dequantized_weight(lin)
    torch.set_default_dtype(bf)
    try:
        with torch.device("cuda"):
            ref = flux_ref.FluxTransformer2DModel()
    finally:
        torch.set_default_dtype(torch.float32)
    ref.load_state_dict(sd, strict=True)
    del sd
    for p in ref.parameters():
        p.requires_grad_(False)
    ref_net = lora_ref.RefLoRANetwork(ref, 32).cuda()
    ref_net.torch_multiplier = ref_net.torch_multiplier.cuda()
    with torch.no_grad():
        for a, b in zip(net.unet_loras, ref_net.unet_loras):
            b.lora_down.weight.copy_(a.lora_down.weight)
            b.lora_up.weight.copy_(a.lora_up.weight)
    ref_net.apply_to()
    del model, net
    torch.cuda.empty_cache()
    _checkpoint_blocks(ref)
    oracle = train_ref.RefTrainStep(ref, ref_net, **kw)
    l16 = oracle.step(lat, emb, pooled, noise, ts, dtype=bf).item()
    g16 = [p.grad.clone() for p in oracle.params]
    ref.float()
    torch.cuda.empty_cache()
    l32 = oracle.step(lat.float(), emb.float(), pooled.float(), noise.float(), ts).item()
    g32 = [p.grad.clone() for p in oracle.params]
    e_o, e_16 = _rel_lists(go, g32), _rel_lists(g16, g32)
    print(f"PARITY FLUX fp8-base r32 full depth 19+38 @1024^2 B=1: loss ours {lo:.6f} ref16 {l16:.6f} fp32 {l32:.6f} (rel ours "
          f"{abs(lo - l32) / l32:.2e}, ref16 {abs(l16 - l32) / l32:.2e}); adapter-gradient rel err ours_vs_fp32 {e_o:.3e} ref16_vs_fp32 {e_16:.3e}; "
          f"peak mem {torch.cuda.max_memory_allocated() / 2 ** 30:.0f} GiB")
    assert abs(lo - l32) <= 1e-3 * abs(l32), (lo, l32, l16)
    assert e_o <= 1.25 * e_16 + 1e-3, (e_o, e_16)


# ------------------------------------------------------------------------------------------------------------------ config 1
def test_full_size_sd15_unet_at_512_vs_eager_oracle():
    import ai_toolkit_amd  # noqa: F401
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.lora import FusedLoRANetwork
    from ai_toolkit_amd.trainer import UNetLoRATrainStep
    from ai_toolkit_amd.unet import SD15_CONFIG, UNet2DConditionModel
    from oracle import lora_ref, train_ref, unet_ref

    dev = "cuda"
    torch.set_default_dtype(bf)
    try:
        with torch.device(dev):
            ref = unet_ref.UNet2DConditionModel(**unet_ref.SD15)
    finally:
        torch.set_default_dtype(torch.float32)
    g = torch.Generator(device=dev).manual_seed(1234)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, (torch.nn.Linear, torch.nn.Conv2d)):
                m.weight.copy_((torch.randn(m.weight.shape, device=dev, generator=g) / math.sqrt(m.weight[0].numel())).to(bf))
                if m.bias is not None:
                    m.bias.copy_((torch.randn(m.bias.shape, device=dev, generator=g) * 0.01).to(bf))
    for p in ref.parameters():
        p.requires_grad_(False)
    assert sum(p.numel() for p in ref.parameters()) == 859_520_964
    nat = UNet2DConditionModel(**SD15_CONFIG, dtype=bf, device=dev, ops=ops)
    nat.load_state_dict(ref.state_dict(), strict=True)
    torch.manual_seed(5)
    net = FusedLoRANetwork(nat, lora_dim=4, alpha=4.0, target_lin_modules=("Transformer2DModel",), is_transformer=False, peft_format=False,
                           transformer_only=False, base_model_version="sd1")
    assert len(net.unet_loras) == 192
    with torch.no_grad():
        for m in net.unet_loras:
            m.lora_up.weight.normal_(0, 2e-3)
    net.apply_to()
    net.build_arena(dev, groups=nat.lora_groups())
    net.refresh_shadows(ops)
    nat.attach_network(net)
    nat.prepare()
    ref_net = lora_ref.RefLoRANetwork(ref, 4, target=("Transformer2DModel",), kohya_unet=True, alpha=4.0).to(dev)
    ref_net.torch_multiplier = ref_net.torch_multiplier.to(dev)
    with torch.no_grad():
        for a, b in zip(net.unet_loras, ref_net.unet_loras):
            assert a.lora_name == b.lora_name
            b.lora_down.weight.copy_(a.lora_down.weight.reshape(b.lora_down.weight.shape))
            b.lora_up.weight.copy_(a.lora_up.weight.reshape(b.lora_up.weight.shape))
    ref_net.apply_to()
    gg = torch.Generator(device=dev).manual_seed(42)
    B = 2
    lat = torch.randn(B, 4, 64, 64, device=dev, generator=gg).to(bf)
    ctx = (torch.randn(B, 77, 768, device=dev, generator=gg) * 0.5).to(bf)
    noise = torch.randn(B, 4, 64, 64, device=dev, generator=gg).to(bf)
    ts = torch.tensor([500, 120], device=dev)
    kw = dict(lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    lo = UNetLoRATrainStep(nat, net, ops, **kw).step(lat, ctx, None, noise=noise, timesteps=ts).item()
    go = [p.grad.detach().clone() for m in net.unet_loras for p in (m.lora_down.weight, m.lora_up.weight)]
    assert math.isfinite(lo)
    oracle = train_ref.RefUNetTrainStep(ref, ref_net, **kw)
    l16 = oracle.step(lat, ctx, None, noise, ts, dtype=bf).item()
    g16 = [p.grad.clone() for p in oracle.params]
    ref.float()
    l32 = oracle.step(lat.float(), ctx.float(), None, noise.float(), ts).item()
    g32 = [p.grad.clone() for p in oracle.params]
    go = [a.reshape(b.shape) for a, b in zip(go, g32)]
    e_o, e_16 = _rel_lists(go, g32), _rel_lists(g16, g32)
    print(f"PARITY full-size SD1.5 UNet @512^2 B={B}: loss ours {lo:.6f} ref16 {l16:.6f} fp32 {l32:.6f} (rel ours {abs(lo - l32) / l32:.2e}, "
          f"ref16 {abs(l16 - l32) / l32:.2e}); adapter-gradient rel err ours_vs_fp32 {e_o:.3e} ref16_vs_fp32 {e_16:.3e}")
    assert abs(lo - l32) <= 1e-3 * abs(l32), (lo, l32, l16)
    assert e_o <= 1.3 * e_16 + 2e-3, (e_o, e_16)


# ------------------------------------------------------------------------------------------------------------------ config 3, deltas
def test_three_adamw_steps_lora_delta_four_way_at_full_flux_depth():
    """LoRA deltas as the model sees them after three AdamW steps (lr 1e-3, wd 0.01, clip 1.0) on the benchmarked model itself
    (19 + 38 blocks, 4096 + 512 tokens, r16, 494 adapters, B = 1): dW = B'A' - BA per module, relative Frobenius over all modules,
    for ours (HIP), rm16 (our graph on the oracle's torch kernels, bf16), ref16 (the reference's arithmetic) and fp32 (truth)."""
    from ai_toolkit_amd.flux import FluxTransformer2DModel
    from ai_toolkit_amd.lora import FusedLoRANetwork
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from oracle import flux_ref, lora_ref, ref_ops, train_ref
    from tests.test_gpu_fullsize import _batch, _flux
    from tests.test_gpu_parity_r2 import _checkpoint_blocks

    kw = dict(lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
    NSTEP = 3
    model, net, ops = _flux(19, 38)
    init = [(m.lora_down.weight.detach().clone(), m.lora_up.weight.detach().clone()) for m in net.unet_loras]
    batches = [_batch(1, seed=100 + k) for k in range(NSTEP)]

    def deltas(pairs):
        """dW = B'A' - BA = [B' | -B] @ [A' ; A] kept in factored form (U [out, 2r], V [2r, in]): the full matrices of all 494 modules
        would be 48 GB per path; Frobenius inner products of factored matrices need only 2r x 2r products (_lowrank_rel)."""
        return [(torch.cat((b1.double(), -b0.double().to(b1.device)), 1).cpu(), torch.cat((a1.double(), a0.double().to(a1.device)), 0).cpu())
                for (a1, b1), (a0, b0) in zip(pairs, init)]

    def _inner(x, y):
        (ux, vx), (uy, vy) = x, y
        return float(((ux.t() @ uy) * (vx @ vy.t())).sum())

    def _lowrank_rel(xs, ys):
        num = sum(_inner(x, x) - 2 * _inner(x, y) + _inner(y, y) for x, y in zip(xs, ys))
        den = sum(_inner(y, y) for y in ys)
        return math.sqrt(max(num, 0.0) / max(den, 1e-300))

    st = FluxLoRATrainStep(model, net, ops, **kw)
    lo = [st.step(lat, emb, pooled, noise=noise, timesteps=ts).item() for lat, emb, pooled, noise, ts in batches]
    d_o = deltas([(m.lora_down.weight.detach(), m.lora_up.weight.detach()) for m in net.unet_loras])
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    del st, model, net
    torch.cuda.empty_cache()

    # rm16: our op graph on the oracle's kernel table (one rounding per op output), fp32 adapter shadows
    rm = FluxTransformer2DModel(dtype=bf, device="cuda", ops=ref_ops)
    rm.load_state_dict(sd, strict=True)
    rm_net = FusedLoRANetwork(rm, lora_dim=16, alpha=16)
    with torch.no_grad():
        for m, (a0, b0) in zip(rm_net.unet_loras, init):
            m.lora_down.weight.copy_(a0.cpu())
            m.lora_up.weight.copy_(b0.cpu())
    rm_net.apply_to()
    rm_net.build_arena("cuda", groups=rm.lora_groups(), shadow_dtype=torch.float32)
    rm_net.refresh_shadows(ref_ops)
    rm.attach_network(rm_net)
    rm.prepare()
    st = FluxLoRATrainStep(rm, rm_net, ref_ops, **kw)
    lrm = [st.step(lat, emb, pooled, noise=noise, timesteps=ts).item() for lat, emb, pooled, noise, ts in batches]
    d_rm = deltas([(m.lora_down.weight.detach(), m.lora_up.weight.detach()) for m in rm_net.unet_loras])
    del st, rm, rm_net
    torch.cuda.empty_cache()

    torch.set_default_dtype(bf)
    try:
        with torch.device("cuda"):
            ref = flux_ref.FluxTransformer2DModel()
    finally:
        torch.set_default_dtype(torch.float32)
    ref.load_state_dict(sd, strict=True)
    del sd
    for p in ref.parameters():
        p.requires_grad_(False)
    ref_net = lora_ref.RefLoRANetwork(ref, 16).cuda()
    ref_net.torch_multiplier = ref_net.torch_multiplier.cuda()
    ref_net.apply_to()
    _checkpoint_blocks(ref)

    def run_oracle(dtype):
        with torch.no_grad():
            for m, (a0, b0) in zip(ref_net.unet_loras, init):
                m.lora_down.weight.copy_(a0)
                m.lora_up.weight.copy_(b0)
        o = train_ref.RefTrainStep(ref, ref_net, **kw)
        losses = [o.step(lat.to(dtype), emb.to(dtype), pooled.to(dtype), noise.to(dtype), ts, dtype=dtype).item()
                  for lat, emb, pooled, noise, ts in batches]
        return losses, deltas([(m.lora_down.weight.detach(), m.lora_up.weight.detach()) for m in ref_net.unet_loras])

    l16, d_16 = run_oracle(bf)
    ref.float()
    torch.cuda.empty_cache()
    l32, d_32 = run_oracle(torch.float32)
    e = {"ours_vs_fp32": _lowrank_rel(d_o, d_32), "ref16_vs_fp32": _lowrank_rel(d_16, d_32), "rm16_vs_fp32": _lowrank_rel(d_rm, d_32),
         "ours_vs_rm16": _lowrank_rel(d_o, d_rm), "ref16_vs_rm16": _lowrank_rel(d_16, d_rm)}
    print("PARITY4 full depth 19+38 @1024^2, 3 AdamW steps: losses ours", [f"{x:.5f}" for x in lo], "rm16", [f"{x:.5f}" for x in lrm],
          "ref16", [f"{x:.5f}" for x in l16], "fp32", [f"{x:.5f}" for x in l32], "; LoRA delta-W rel err " + " ".join(f"{k}={v:.3e}" for k, v in e.items()))
    # the first step's loss sees identical adapters; later steps see adapters that have already moved apart by AdamW's lr * sign(g) on
    # entries whose gradient is inside the bf16 noise — in EVERY bf16 execution (ref16's distance to fp32 is the yardstick)
    assert abs(lo[0] - l32[0]) <= 1e-3 * abs(l32[0]), (lo, l32)
    for a, r, b in zip(lo, l16, l32):
        assert abs(a - b) <= max(2e-3 * abs(b), 1.25 * abs(r - b)), (lo, l16, l32)
    assert e["ours_vs_fp32"] <= 1.25 * max(e["ref16_vs_fp32"], e["rm16_vs_fp32"]), e
    assert e["ours_vs_rm16"] <= 1.1 * e["ref16_vs_rm16"], e
