"""UNet (SD1.5 / SDXL, BASELINE configs 1-2) host logic on CPU: the explicit NHWC op graph of ai_toolkit_amd.unet driven by the
oracle's plain-torch kernel table in fp32 must reproduce autograd of oracle/unet_ref.py + the oracle LoRA layer — prediction, every
adapter gradient (Linear and 1x1-conv adapters), skip connections, stride-2 / upsample paths, cross-attention on 77-like text
tokens, GEGLU, the text_time embedding of SDXL — and the architecture restatement is pinned by the published parameter counts."""
import pytest
import torch

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd.lora import FusedLoRANetwork
from ai_toolkit_amd.unet import SD15_CONFIG, SDXL_CONFIG, UNet2DConditionModel
from oracle import lora_ref, ref_ops, unet_ref

TINY_SD15 = dict(unet_ref.SD15, block_out_channels=(32, 64, 64, 64), cross_attention_dim=24, attention_head_dim=2, layers_per_block=1,
                 norm_num_groups=8)
TINY_SDXL = dict(unet_ref.SDXL, block_out_channels=(32, 64, 128), cross_attention_dim=24, attention_head_dim=(2, 4, 8),
                 transformer_layers_per_block=(1, 1, 2), projection_class_embeddings_input_dim=16 + 6 * 8, addition_time_embed_dim=8,
                 norm_num_groups=8)


def test_architecture_matches_the_published_checkpoints():
    """SD1.5 UNet: 859,520,964 parameters; SDXL UNet: 2,567,463,684 — both configs, oracle and native tree, same keys and shapes."""
    for cfg, n_params, n_lora in ((unet_ref.SD15, 859_520_964, 192), (unet_ref.SDXL, 2_567_463_684, 722)):
        with torch.device("meta"):
            ref = unet_ref.UNet2DConditionModel(**cfg)
            nat = UNet2DConditionModel(**cfg, dtype=torch.float32)
        assert sum(p.numel() for p in ref.parameters()) == n_params
        rs, ns = ref.state_dict(), nat.state_dict()
        assert list(rs.keys()) == list(ns.keys())
        assert all(rs[k].shape == ns[k].shape for k in rs)
        # adapters: 12 per BasicTransformerBlock-bearing Transformer2DModel in SD1.5 (16 x 12), 70 x 10 + 11 x 2 in SDXL
        net = lora_ref.RefLoRANetwork(ref, 4, target=("Transformer2DModel",), kohya_unet=True)
        assert len(net.unet_loras) == n_lora
    assert SD15_CONFIG["block_out_channels"] == unet_ref.SD15["block_out_channels"] and SDXL_CONFIG["transformer_layers_per_block"] == (1, 2, 10)


def build_pair(cfg, rank=4, alpha=2.0, seed=0):
    torch.manual_seed(seed)
    ref = unet_ref.UNet2DConditionModel(**cfg)
    unet_ref.init_synthetic_(ref, seed=11)
    nat = UNet2DConditionModel(**cfg, dtype=torch.float32, device="cpu", ops=ref_ops)
    nat.load_state_dict(ref.state_dict(), strict=True)
    ref_net = lora_ref.RefLoRANetwork(ref, rank, target=("Transformer2DModel",), kohya_unet=True, alpha=alpha)
    net = FusedLoRANetwork(nat, lora_dim=rank, alpha=alpha, target_lin_modules=("Transformer2DModel",), is_transformer=False,
                           peft_format=False, transformer_only=False, base_model_version="sd1")
    assert [m.lora_name for m in net.unet_loras] == [m.lora_name for m in ref_net.unet_loras]
    assert net.unet_loras[0].lora_name.startswith("lora_unet_down_blocks_") and net.unet_loras[0].scale == alpha / rank
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for a, b in zip(net.unet_loras, ref_net.unet_loras):
            b.lora_up.weight.copy_(torch.randn(b.lora_up.weight.shape, generator=g) * 0.05)
            a.lora_down.weight.copy_(b.lora_down.weight.reshape(a.lora_down.weight.shape))
            a.lora_up.weight.copy_(b.lora_up.weight.reshape(a.lora_up.weight.shape))
    ref_net.apply_to()
    net.apply_to()
    net.build_arena("cpu", groups=nat.lora_groups())
    net.refresh_shadows(ref_ops)
    nat.attach_network(net)
    nat.prepare()
    return ref, ref_net, nat, net


def _inputs(cfg, B=2, H=16, W=8, n_txt=7, seed=3):
    g = torch.Generator().manual_seed(seed)
    lat = torch.randn(B, 4, H, W, generator=g)
    ctx = torch.randn(B, n_txt, cfg["cross_attention_dim"], generator=g)
    ts = torch.tensor([10.0, 500.0, 999.0][:B])
    added = None
    if cfg["addition_embed_type"] == "text_time":
        added = dict(text_embeds=torch.randn(B, cfg["projection_class_embeddings_input_dim"] - 6 * cfg["addition_time_embed_dim"], generator=g),
                     time_ids=unet_ref.time_ids_from_latents(lat))
    return lat, ts, ctx, added


def _nhwc8(lat):
    B, Cc, H, W = lat.shape
    x = torch.zeros(B * H * W, 8)
    x[:, :Cc] = lat.permute(0, 2, 3, 1).reshape(B * H * W, Cc)
    return x


@pytest.mark.parametrize("cfg", [TINY_SD15, TINY_SDXL], ids=["sd15", "sdxl"])
def test_forward_and_adapter_gradients_match_oracle_autograd(cfg):
    ref, ref_net, nat, net = build_pair(cfg)
    lat, ts, ctx, added = _inputs(cfg)
    B, _, H, W = lat.shape
    # inactive network == base model (nothing recorded)
    p0 = nat.forward_native(_nhwc8(lat), ts, ctx, added, B=B, H=H, W=W)
    p0_ref = ref(lat, ts, ctx, added)
    assert nat.tape is None
    assert torch.allclose(p0.view(B, H, W, 4).permute(0, 3, 1, 2), p0_ref, rtol=1e-4, atol=1e-5)
    with ref_net:
        pred_ref = ref(lat, ts, ctx, added)
        wgt = torch.randn(pred_ref.shape, generator=torch.Generator().manual_seed(11))
        (pred_ref * wgt).sum().backward()
    with net:
        pred = nat.forward_native(_nhwc8(lat), ts, ctx, added, B=B, H=H, W=W)
        got = pred.view(B, H, W, 4).permute(0, 3, 1, 2)
        assert torch.allclose(got, pred_ref, rtol=1e-4, atol=1e-5), (got - pred_ref).abs().max()
        net.zero_grad_arena()
        nat.backward_native(wgt.permute(0, 2, 3, 1).reshape(B * H * W, 4).contiguous())
    worst = 0.0
    for a, b in zip(net.unet_loras, ref_net.unet_loras):
        for x, y, nm in ((a.lora_down.weight.grad, b.lora_down.weight.grad, "down"), (a.lora_up.weight.grad, b.lora_up.weight.grad, "up")):
            y = y.reshape(x.shape)
            err = ((x - y).norm() / (y.norm() + 1e-12)).item()
            worst = max(worst, err)
            assert err < 3e-4, (a.lora_name, nm, err)
    assert worst > 0
    # diffusers-signature call + autograd bridge (what a reference-style trainer drives)
    with net:
        net.zero_grad_arena()
        (out,) = nat(lat, ts, ctx, added)
        assert torch.allclose(out, pred_ref, rtol=1e-4, atol=1e-5)
        (out * wgt).sum().backward()
    m0 = net.unet_loras[0]
    # atol = 1e-5 of entries up to ~1.2: fp32 summation order (the q/k/v data gradient contracts over the concatenated 3 x dim channels in one product)
    assert torch.allclose(m0.lora_up.weight.grad, ref_net.unet_loras[0].lora_up.weight.grad.reshape(m0.lora_up.weight.shape), rtol=2e-3, atol=1e-5)


def test_kohya_state_dict_round_trip_and_shapes(tmp_path):
    from safetensors.torch import load_file

    ref, ref_net, nat, net = build_pair(TINY_SD15)
    f = tmp_path / "sd15.safetensors"
    net.save_weights(str(f), dtype=torch.float32)
    sd = load_file(str(f))
    want = ref_net.state_dict()
    assert sorted(sd.keys()) == sorted(want.keys())  # kohya format = the network state_dict (toolkit/network_mixins.py:590-598)
    for k, v in want.items():
        assert sd[k].shape == v.shape and torch.allclose(sd[k], v.float()), k
    conv_keys = [k for k in sd if k.endswith("proj_in.lora_down.weight")]
    assert conv_keys and sd[conv_keys[0]].dim() == 4 and sd[conv_keys[0]].shape[2:] == (1, 1)
    assert float(sd["lora_unet_down_blocks_0_attentions_0_proj_in.alpha"]) == 2.0
    with torch.no_grad():
        net.arena_p.zero_()
    assert net.load_weights(str(f)) is None
    for a, b in zip(net.unet_loras, ref_net.unet_loras):
        assert torch.equal(a.lora_up.weight, b.lora_up.weight.reshape(a.lora_up.weight.shape))


TINY_SD15_D160 = dict(unet_ref.SD15, block_out_channels=(32, 64, 320, 320), cross_attention_dim=24, attention_head_dim=2, layers_per_block=1,
                      norm_num_groups=8)


def test_head_dim_160_uses_the_generic_attention_path():
    """SD1.5's two coarsest levels have head_dim 1280 / 8 = 160 > the flash kernels' 128: the generic attention ops carry them."""
    ref, ref_net, nat, net = build_pair(TINY_SD15_D160)
    lat, ts, ctx, added = _inputs(TINY_SD15_D160, B=1, H=16, W=16)
    calls = []
    orig = ref_ops.attn_small_fwd
    ref_ops.attn_small_fwd = lambda *a, **k: (calls.append(k["D"]), orig(*a, **k))[1]
    try:
        with ref_net:
            pred_ref = ref(lat, ts, ctx, added)
            pred_ref.square().sum().backward()
        with net:
            pred = nat.forward_native(_nhwc8(lat), ts, ctx, added, B=1, H=16, W=16)
            assert torch.allclose(pred.view(1, 16, 16, 4).permute(0, 3, 1, 2), pred_ref, rtol=1e-4, atol=1e-5)
            net.zero_grad_arena()
            nat.backward_native((2 * pred).detach())
    finally:
        ref_ops.attn_small_fwd = orig
    assert calls and set(calls) == {160}
    for a, b in zip(net.unet_loras, ref_net.unet_loras):
        y = b.lora_up.weight.grad.reshape(a.lora_up.weight.shape)
        assert ((a.lora_up.weight.grad - y).norm() / (y.norm() + 1e-12)).item() < 3e-4, a.lora_name


@pytest.mark.parametrize("cfg,kw", [(TINY_SD15, dict(min_snr_gamma=5.0)), (TINY_SDXL, dict()), (TINY_SD15, dict(prediction_type="v_prediction", snr_gamma=5.0))],
                         ids=["sd15-minsnr", "sdxl-eps", "sd15-vpred-snr"])
def test_train_steps_match_oracle_optimizer_sequence(cfg, kw):
    from ai_toolkit_amd.ddpm import DDPMTrainSchedule
    from ai_toolkit_amd.trainer import UNetLoRATrainStep
    from oracle import train_ref

    ref, ref_net, nat, net = build_pair(cfg)
    opt = dict(lr=1e-3, weight_decay=0.01, max_grad_norm=0.5)
    pt = kw.get("prediction_type", "epsilon")
    oracle = train_ref.RefUNetTrainStep(ref, ref_net, **opt, **kw)
    ours = UNetLoRATrainStep(nat, net, ref_ops, schedule=DDPMTrainSchedule(prediction_type=pt), min_snr_gamma=kw.get("min_snr_gamma"),
                             snr_gamma=kw.get("snr_gamma"), **opt)
    g = torch.Generator().manual_seed(5)
    for k in range(2):
        lat = torch.randn(2, 4, 16, 8, generator=g)
        ctx = torch.randn(2, 7, cfg["cross_attention_dim"], generator=g)
        pooled = torch.randn(2, 16, generator=g)
        noise = torch.randn(2, 4, 16, 8, generator=g)
        ts = torch.tensor([[17, 640], [998, 2]][k])
        l_ref = oracle.step(lat, ctx, pooled, noise, ts)
        l = ours.step(lat, ctx, pooled, noise=noise, timesteps=ts)
        assert abs(l.item() - l_ref.item()) <= 1e-4 * abs(l_ref.item()), (k, l.item(), l_ref.item())
    for a, b in zip(net.unet_loras, ref_net.unet_loras):
        for pa, pb in ((a.lora_down.weight, b.lora_down.weight), (a.lora_up.weight, b.lora_up.weight)):
            assert torch.allclose(pa, pb.reshape(pa.shape), rtol=2e-3, atol=1e-5), (a.lora_name, (pa - pb.reshape(pa.shape)).abs().max())


def test_ddpm_schedule_tables():
    """scaled-linear betas, [999..0] timestep table, balanced indices in [1, 997] (randint(min+1, max-1)), min-SNR weights."""
    from ai_toolkit_amd.ddpm import DDPMTrainSchedule

    s = DDPMTrainSchedule()
    assert torch.allclose(s.alphas_cumprod, unet_ref.ddpm_alphas_cumprod())
    assert abs(float(s.alphas_cumprod[0]) - (1 - 0.00085)) < 1e-7 and abs(float(s.alphas_cumprod[-1]) - 0.0047) < 2e-4
    tt = s.set_timesteps(1000, "cpu")
    assert tt[0] == 999 and tt[-1] == 0 and len(tt) == 1000
    g1, g2 = torch.Generator().manual_seed(3), torch.Generator().manual_seed(3)
    t, idx = s.sample_timesteps(64, "cpu", generator=g1)
    assert torch.equal(idx, torch.randint(1, 998, (64,), generator=g2)) and torch.equal(t, 999 - idx)
    ts = torch.tensor([2, 500, 998])
    assert torch.allclose(s.snr_weights(ts, 5.0), unet_ref.min_snr_weight(ts, s.alphas_cumprod, 5.0))
    assert torch.allclose(s.snr_weights(ts, 5.0, fixed=True), unet_ref.min_snr_weight(ts, s.alphas_cumprod, 5.0, fixed=True))
    a, sg = s.noise_coefficients(ts, torch.bfloat16)
    assert a.dtype == torch.float32 and torch.equal(a, (s.alphas_cumprod.to(torch.bfloat16)[ts] ** 0.5).float())


def _unet_dp_worker(rank, world, port, out):
    import datetime
    import os

    import torch.distributed as dist

    from ai_toolkit_amd.trainer import UNetLoRATrainStep

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    torch.set_num_threads(2)
    ref, ref_net, nat, net = build_pair(TINY_SDXL)
    step = UNetLoRATrainStep(nat, net, ref_ops, lr=1e-3, max_grad_norm=0.5, min_snr_gamma=5.0, process_group=dist.group.WORLD)
    for k in range(2):
        lat, ctx, pooled, noise, ts = _dp_batch(4, seed=40 + k)
        sl = slice(rank * 2, rank * 2 + 2)  # disjoint shard of the bucket batch
        step.step(lat[sl], ctx[sl], pooled[sl], noise=noise[sl], timesteps=ts[sl])
    torch.save(net.arena_p.clone(), os.path.join(out, f"p{rank}.pt"))
    dist.destroy_process_group()


def _dp_batch(B, seed):
    g = torch.Generator().manual_seed(seed)
    cfg = TINY_SDXL
    lat = torch.randn(B, 4, 16, 8, generator=g)
    ctx = torch.randn(B, 7, cfg["cross_attention_dim"], generator=g)
    pooled = torch.randn(B, cfg["projection_class_embeddings_input_dim"] - 6 * cfg["addition_time_embed_dim"], generator=g)
    noise = torch.randn(B, 4, 16, 8, generator=g)
    ts = torch.tensor([10, 500, 998, 250][:B])
    return lat, ctx, pooled, noise, ts


def test_unet_dp2_gloo_equals_single_rank_on_concatenated_batch(tmp_path):
    """SURVEY.md §8e for the UNet step: two ranks on disjoint halves of the batch, one all-reduce(mean) of the flat gradient arena
    (up-block adapters first), then clip / AdamW redundantly == one rank on the whole batch; ranks end bit-identical."""
    import torch.multiprocessing as mp

    from ai_toolkit_amd.trainer import UNetLoRATrainStep
    from tests.conftest import free_port

    mp.spawn(_unet_dp_worker, args=(2, free_port(), str(tmp_path)), nprocs=2, join=True)
    p0, p1 = torch.load(tmp_path / "p0.pt"), torch.load(tmp_path / "p1.pt")
    assert torch.equal(p0, p1), "ranks must hold bit-identical adapter weights"
    ref, ref_net, nat, net = build_pair(TINY_SDXL)
    step = UNetLoRATrainStep(nat, net, ref_ops, lr=1e-3, max_grad_norm=0.5, min_snr_gamma=5.0)
    for k in range(2):
        lat, ctx, pooled, noise, ts = _dp_batch(4, seed=40 + k)
        step.step(lat, ctx, pooled, noise=noise, timesteps=ts)
    assert torch.allclose(net.arena_p, p0, rtol=1e-3, atol=1e-6), (net.arena_p - p0).abs().max()
