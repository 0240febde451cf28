"""`network.conv` (conv-LoRA, toolkit/lora_special.py:95-104, 585-590, 678-681; toolkit/kohya_lora.py:751) pinned to the reference's own
LoRASpecialNetwork(conv_lora_dim=...) executed by tests/golden/make_golden.py::golden_unet_conv_lora on the oracle UNet trees: the wider
adapter inventory (Linear / Conv2d children of ResnetBlock2D, Downsample2D, Upsample2D besides every Transformer2DModel), per-kind ranks and
scales, init draws under the same seed, one forward + every adapter gradient through the reference's LoRAModule.forward — incl. the 3x3
(stride 1 and 2) convolution adapters, the 1x1 conv_shortcut and the small-batch time_emb_proj adapters — and the kohya file it saves
(lora_down [r, in, 3, 3], lora_up [out, r, 1, 1])."""
import hashlib
import json
import os

import pytest
import torch
from safetensors import safe_open
from safetensors.torch import load_file

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd.lora import FusedLoRANetwork
from ai_toolkit_amd.unet import UNet2DConditionModel
from oracle import ref_ops, unet_ref
from tests.test_unet_cpu import TINY_SD15, TINY_SDXL, _inputs, _nhwc8

G = os.path.join(os.path.dirname(__file__), "golden", "unet_conv_lora_tiny.safetensors")
KW = dict(target_lin_modules=("Transformer2DModel",), is_transformer=False, peft_format=False, transformer_only=False)


def _golden():
    with safe_open(G, "pt") as f:
        meta = json.loads(f.metadata()["meta"])
    return load_file(G), meta


def build(cfg, dtype=torch.float32, device="cpu", ops=ref_ops):
    torch.manual_seed(0)
    ref = unet_ref.UNet2DConditionModel(**cfg)
    unet_ref.init_synthetic_(ref, seed=11)
    nat = UNet2DConditionModel(**cfg, dtype=dtype, device=device, ops=ops)
    nat.load_state_dict({k: v.to(dtype) for k, v in ref.state_dict().items()}, strict=True)
    torch.manual_seed(99)
    net = FusedLoRANetwork(nat, lora_dim=4, alpha=2.0, conv_lora_dim=2, conv_alpha=1.0, **KW)
    return nat, net


def warm_and_attach(nat, net, t, tag, ops, device="cpu"):
    with torch.no_grad():
        for x in net.unet_loras:
            x.lora_up.weight.copy_(t[f"{tag}/warm/{x.lora_name}/up"].reshape(x.lora_up.weight.shape))
    net.apply_to()
    net.build_arena(device, groups=nat.lora_groups())
    net.refresh_shadows(ops)
    nat.attach_network(net)
    nat.prepare()


@pytest.mark.parametrize("tag,cfg", [("sd15", TINY_SD15), ("sdxl", TINY_SDXL)])
def test_fused_conv_lora_network_matches_reference_network(tag, cfg, tmp_path):
    t, meta = _golden()
    m = meta[tag]
    nat, net = build(cfg)
    assert [x.lora_name for x in net.unet_loras] == m["names"]
    assert [x.lora_dim for x in net.unet_loras] == m["dims"] and [x.scale for x in net.unet_loras] == m["scales"]
    kinds = {"conv3x3": sum(x.is_conv3x3 for x in net.unet_loras), "conv1x1": sum(bool(x.is_conv1x1) for x in net.unet_loras)}
    assert kinds["conv3x3"] > 0 and any("time_emb_proj" in x.lora_name for x in net.unet_loras) and any("conv_shortcut" in x.lora_name for x in net.unet_loras)
    for x in net.unet_loras:  # same construction order and fan-in => same RNG consumption => the reference's kaiming draws, bit for bit
        want = t[f"{tag}/init/{x.lora_name}/down"]
        assert torch.equal(x.lora_down.weight, want.reshape(x.lora_down.weight.shape)), x.lora_name
    warm_and_attach(nat, net, t, tag, ref_ops)
    lat, ts, ctx, added = _inputs(cfg)
    B, _, H, W = lat.shape
    with net:
        pred = nat.forward_native(_nhwc8(lat), ts, ctx, added, B=B, H=H, W=W)
        got = pred.view(B, H, W, 4).permute(0, 3, 1, 2)
        assert torch.allclose(got, t[f"{tag}/pred"], rtol=2e-4, atol=2e-5), (got - t[f"{tag}/pred"]).abs().max()
        net.zero_grad_arena()
        nat.backward_native(t[f"{tag}/wgt"].permute(0, 2, 3, 1).reshape(B * H * W, 4).contiguous())
    for x in net.unet_loras:
        for nm, p_ in (("down", x.lora_down.weight), ("up", x.lora_up.weight)):
            want = t[f"{tag}/grad/{x.lora_name}/{nm}"].reshape(p_.shape)
            err = ((p_.grad - want).norm() / (want.norm() + 1e-12)).item()
            assert err < 5e-4, (x.lora_name, nm, err)
    f = tmp_path / "unet_conv.safetensors"
    net.save_weights(str(f), dtype=torch.float32)
    sd = load_file(str(f))
    assert sorted(sd.keys()) == sorted(m["saved_keys"])
    for k in m["saved_keys"]:
        want = t[f"{tag}/saved/{k}"]
        assert sd[k].shape == want.shape and torch.equal(sd[k], want), k
    assert list(net.get_state_dict(dtype=torch.float32).keys()) == m["saved_keys"]
    # round trip through load_weights into a fresh network (4-D conv shapes back into the flat arena blocks)
    nat2, net2 = build(cfg)
    net2.apply_to()
    net2.build_arena("cpu", groups=nat2.lora_groups())
    assert net2.load_weights(str(f)) is None
    for a, b in zip(net.unet_loras, net2.unet_loras):
        assert torch.equal(a.lora_down.weight, b.lora_down.weight) and torch.equal(a.lora_up.weight, b.lora_up.weight)


def test_merge_in_of_conv_adapters_equals_the_active_network():
    """ToolkitModuleMixin.merge_in on Conv2d adapters (toolkit/network_mixins.py:424-433): merged-weight forward == adapter-active forward."""
    t, _ = _golden()
    nat, net = build(TINY_SDXL)
    warm_and_attach(nat, net, t, "sdxl", ref_ops)
    lat, ts, ctx, added = _inputs(TINY_SDXL)
    B, _, H, W = lat.shape
    with net:
        active = nat.forward_native(_nhwc8(lat), ts, ctx, added, B=B, H=H, W=W, save_for_backward=False).clone()
    base = nat.forward_native(_nhwc8(lat), ts, ctx, added, B=B, H=H, W=W, save_for_backward=False).clone()
    net.merge_in(1.0, ops=ref_ops)
    merged = nat.forward_native(_nhwc8(lat), ts, ctx, added, B=B, H=H, W=W, save_for_backward=False).clone()
    net.merge_out(1.0, ops=ref_ops)
    restored = nat.forward_native(_nhwc8(lat), ts, ctx, added, B=B, H=H, W=W, save_for_backward=False).clone()
    assert (active - base).abs().max() > 1e-4
    assert torch.allclose(merged, active, rtol=1e-4, atol=1e-5) and torch.allclose(restored, base, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("tag,cfg,count", [("sd15_full", unet_ref.SD15, 278), ("sdxl_full", unet_ref.SDXL, 788)])
def test_full_size_conv_lora_inventory_equals_the_reference(tag, cfg, count):
    _, meta = _golden()
    m = meta[tag]
    assert m["count"] == count
    with torch.device("meta"):
        nat = UNet2DConditionModel(**cfg, dtype=torch.float32)
    net = FusedLoRANetwork(nat, lora_dim=4, alpha=4.0, conv_lora_dim=4, conv_alpha=4.0, **KW)
    names = [x.lora_name for x in net.unet_loras]
    assert len(names) == count and names[:3] == m["first"] and names[-3:] == m["last"]
    assert hashlib.sha256("\n".join(names).encode()).hexdigest() == m["names_sha256"]
    shapes = []
    for x in net.unet_loras:
        d, u = list(x.lora_down.weight.shape), list(x.lora_up.weight.shape)
        if x.is_conv1x1:
            d, u = d + [1, 1], u + [1, 1]
        elif x.is_conv3x3:
            d, u = [d[0], x.conv_cin, 3, 3], u + [1, 1]
        shapes.append([d, u])
    assert hashlib.sha256(json.dumps(shapes).encode()).hexdigest() == m["shapes_sha256"]
    assert sum(x.lora_down.weight.numel() + x.lora_up.weight.numel() for x in net.unet_loras) == m["params"]


def test_refusals():
    with torch.device("meta"):
        nat = UNet2DConditionModel(**TINY_SD15, dtype=torch.float32)
    with pytest.raises(NotImplementedError):
        FusedLoRANetwork(nat, lora_dim=4, conv_lora_dim=2, network_type="dora", **KW)
    with pytest.raises(NotImplementedError):
        FusedLoRANetwork(nat, lora_dim=4, conv_lora_dim=96, **KW)  # above one 128-column slab tile
    # conv_alpha None: the module falls back to alpha = rank (toolkit/lora_special.py:113-115), i.e. scale 1
    net = FusedLoRANetwork(nat, lora_dim=4, alpha=2.0, conv_lora_dim=8, **KW)
    conv = next(x for x in net.unet_loras if x.is_conv3x3)
    assert conv.scale == 1.0 and conv.lora_dim == 8 and float(conv.alpha) == 8.0


def _conv_dp_net():
    t, _ = _golden()
    nat, net = build(TINY_SDXL)
    warm_and_attach(nat, net, t, "sdxl", ref_ops)
    return nat, net


def _conv_dp_worker(rank, world, port, out):
    import datetime

    import torch.distributed as dist

    from ai_toolkit_amd.trainer import UNetLoRATrainStep
    from tests.test_unet_cpu import _dp_batch

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    torch.set_num_threads(2)
    nat, net = _conv_dp_net()
    step = UNetLoRATrainStep(nat, net, ref_ops, lr=1e-3, max_grad_norm=0.5, min_snr_gamma=5.0, process_group=dist.group.WORLD)
    for k in range(2):
        lat, ctx, pooled, noise, ts = _dp_batch(4, seed=40 + k)
        sl = slice(rank * 2, rank * 2 + 2)
        step.step(lat[sl], ctx[sl], pooled[sl], noise=noise[sl], timesteps=ts[sl])
    torch.save(net.arena_p.clone(), os.path.join(out, f"p{rank}.pt"))
    dist.destroy_process_group()


def test_conv_lora_dp2_gloo_equals_single_rank_on_concatenated_batch(tmp_path):
    """SURVEY.md §8e with `network.conv`: the conv / time_emb_proj / conv_shortcut adapters live in the same flat gradient arena, so two ranks
    on disjoint halves of the batch + one all-reduce(mean) == one rank on the whole batch; ranks end bit-identical."""
    import torch.multiprocessing as mp

    from ai_toolkit_amd.trainer import UNetLoRATrainStep
    from tests.conftest import free_port
    from tests.test_unet_cpu import _dp_batch

    mp.spawn(_conv_dp_worker, args=(2, free_port(), str(tmp_path)), nprocs=2, join=True)
    p0, p1 = torch.load(tmp_path / "p0.pt"), torch.load(tmp_path / "p1.pt")
    assert torch.equal(p0, p1)
    nat, net = _conv_dp_net()
    step = UNetLoRATrainStep(nat, net, ref_ops, lr=1e-3, max_grad_norm=0.5, min_snr_gamma=5.0)
    for k in range(2):
        lat, ctx, pooled, noise, ts = _dp_batch(4, seed=40 + k)
        step.step(lat, ctx, pooled, noise=noise, timesteps=ts)
    assert torch.allclose(net.arena_p, p0, rtol=1e-3, atol=1e-6), (net.arena_p - p0).abs().max()


def test_optimizer_state_of_conv_adapters_loads_into_the_reference_shapes():
    """optimizer.pt (BaseSDTrainProcess.py:701-714): the reference's torch.optim.AdamW holds Conv2d-shaped moments for conv adapters
    ([r, in, 3, 3] / [r, in, 1, 1] down, [out, r, 1, 1] up).  The exported state loads into an AdamW over parameters of exactly those shapes
    (the shapes of the file the reference's network saves) and comes back unchanged."""
    t, meta = _golden()
    nat, net = build(TINY_SD15)
    warm_and_attach(nat, net, t, "sd15", ref_ops)
    net.arena_m.copy_(torch.randn(net.arena_m.shape, generator=torch.Generator().manual_seed(1)))
    net.arena_v.copy_(torch.rand(net.arena_v.shape, generator=torch.Generator().manual_seed(2)))
    sd = net.optimizer_state_dict(step=4, lr=1e-4)
    ref_params = []
    for x in net.unet_loras:  # the reference module's parameters: lora_down.weight, lora_up.weight with the saved-file shapes
        ref_params += [torch.nn.Parameter(torch.zeros_like(t[f"sd15/saved/{x.lora_name}.lora_down.weight"])),
                       torch.nn.Parameter(torch.zeros_like(t[f"sd15/saved/{x.lora_name}.lora_up.weight"]))]
    assert any(p.dim() == 4 and p.shape[2:] == (3, 3) for p in ref_params)
    opt = torch.optim.AdamW(ref_params, lr=1e-4, eps=1e-6)
    opt.load_state_dict(sd)
    for i, p in enumerate(ref_params):
        assert opt.state[p]["exp_avg"].shape == p.shape, (i, opt.state[p]["exp_avg"].shape, p.shape)
    m0, v0 = net.arena_m.clone(), net.arena_v.clone()
    net.arena_m.zero_()
    net.arena_v.zero_()
    assert net.load_optimizer_state_dict(opt.state_dict()) == 4
    # the rank padding of the arena blocks is not part of any parameter: compare through the logical views
    for a, b in zip(net._opt_slices(net.arena_m), net._opt_slices(m0)):
        assert torch.equal(a, b)
    for a, b in zip(net._opt_slices(net.arena_v), net._opt_slices(v0)):
        assert torch.equal(a, b)


GH = os.path.join(os.path.dirname(__file__), "golden", "unet_conv_lora_highrank.safetensors")


@pytest.mark.parametrize("tag,cfg", [("sdxl", TINY_SDXL), ("sd15", TINY_SD15)])
def test_conv_rank_above_16_and_linear_rank_above_64_match_the_reference_network(tag, cfg):
    """Conv adapters of rank 24 / 40 (two / three 16-rank blocks of the split-slab epilogue, slab of rank_pad 32 / 48) and Linear adapters of
    rank 80 (two skinny launches per product: 64 + 16 ranks of one slab) against the reference's LoRASpecialNetwork run
    (tests/golden/make_golden.py::golden_unet_conv_lora_highrank): prediction, and every adapter gradient through two fixed random
    projections + its norm."""
    with safe_open(GH, "pt") as f:
        meta = json.loads(f.metadata()["meta"])[tag]
    t = load_file(GH)
    lin_r, conv_r = meta["lin_rank"], meta["conv_rank"]
    torch.manual_seed(0)
    ref = unet_ref.UNet2DConditionModel(**cfg)
    unet_ref.init_synthetic_(ref, seed=11)
    nat = UNet2DConditionModel(**cfg, dtype=torch.float32, device="cpu", ops=ref_ops)
    nat.load_state_dict(ref.state_dict(), strict=True)
    torch.manual_seed(99)
    net = FusedLoRANetwork(nat, lora_dim=lin_r, alpha=lin_r / 2, conv_lora_dim=conv_r, conv_alpha=conv_r / 4, **KW)
    assert [x.lora_name for x in net.unet_loras] == meta["names"] and [x.lora_dim for x in net.unet_loras] == meta["dims"]
    assert [x.scale for x in net.unet_loras] == meta["scales"]
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for x in net.unet_loras:
            x.lora_up.weight.copy_((torch.randn(tuple(x.lora_up.weight.shape) + ((1, 1) if (x.is_conv1x1 or x.is_conv3x3) else ()), generator=g)
                                    * 0.05).reshape(x.lora_up.weight.shape))
    net.apply_to()
    net.build_arena("cpu", groups=nat.lora_groups())
    net.refresh_shadows(ref_ops)
    nat.attach_network(net)
    nat.prepare()
    conv = [x for x in net.unet_loras if x.is_conv3x3]
    assert conv and all(x.rank_pad == (conv_r + 15) // 16 * 16 for x in conv)
    lat, ts, ctx, added = _inputs(cfg)
    B, _, H, W = lat.shape
    with net:
        pred = nat.forward_native(_nhwc8(lat), ts, ctx, added, B=B, H=H, W=W)
        got = pred.view(B, H, W, 4).permute(0, 3, 1, 2)
        assert torch.allclose(got, t[f"{tag}/pred"], rtol=2e-4, atol=2e-5), (got - t[f"{tag}/pred"]).abs().max()
        net.zero_grad_arena()
        nat.backward_native(t[f"{tag}/wgt"].permute(0, 2, 3, 1).reshape(B * H * W, 4).contiguous())
    gp = torch.Generator().manual_seed(123)
    for x in net.unet_loras:
        for nm, p_ in (("down", x.lora_down.weight), ("up", x.lora_up.weight)):
            G = p_.grad.reshape(p_.shape[0], -1)
            v, u = torch.randn(G.shape[1], generator=gp), torch.randn(G.shape[0], generator=gp)
            want = t[f"{tag}/grad/{x.lora_name}/{nm}"]
            mine = torch.cat((G @ v, u @ G, G.norm().reshape(1)))
            err = ((mine - want).norm() / (want.norm() + 1e-12)).item()
            assert err < 5e-4, (x.lora_name, nm, err)


@pytest.mark.parametrize("per_sample", [False, True], ids=["dropout", "dropout+per-sample-multipliers"])
def test_conv_adapters_with_dropout_variants_and_per_sample_multipliers_match_the_oracle(per_sample):
    """network.conv adapters under dropout / rank_dropout / module_dropout (toolkit/network_mixins.py:198-229: the rank mask of a Conv2d
    activation is [B, r, 1, 1]) and under a per-sample multiplier list (slider training): the convolution epilogue carries the runtime scale
    only, aitk_slab_rescale applies the row factor and the masks to the rank-space activation, aitk_lora_down applies them to its gradient.
    Oracle = the restated LoRAModule on nn.Conv2d (oracle/lora_ref.py) under autograd; both sides draw from one keyed provider."""
    import hashlib

    from oracle import lora_ref

    def provider(name, kind, shape, device):
        seed = int(hashlib.sha256(f"{name}/{kind}".encode()).hexdigest()[:8], 16)
        return torch.rand(shape, generator=torch.Generator().manual_seed(seed))

    cfg = TINY_SDXL
    torch.manual_seed(0)
    ref = unet_ref.UNet2DConditionModel(**cfg)
    unet_ref.init_synthetic_(ref, seed=11)
    nat = UNet2DConditionModel(**cfg, dtype=torch.float32, device="cpu", ops=ref_ops)
    nat.load_state_dict(ref.state_dict(), strict=True)
    drop = dict(dropout=0.1, rank_dropout=0.25, module_dropout=0.15)
    torch.manual_seed(99)
    net = FusedLoRANetwork(nat, lora_dim=4, alpha=2.0, conv_lora_dim=8, conv_alpha=4.0, **drop, **KW)
    net.mask_provider = provider
    ref_net = lora_ref.RefLoRANetwork(ref, 4, target=("Transformer2DModel",), kohya_unet=True, alpha=2.0, conv_lora_dim=8, conv_alpha=4.0)
    ref_net.dropout_cfg, ref_net.mask_provider = drop, provider
    assert [m.lora_name for m in net.unet_loras] == [m.lora_name for m in ref_net.unet_loras]
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for a, b in zip(net.unet_loras, ref_net.unet_loras):
            b.lora_up.weight.copy_(torch.randn(b.lora_up.weight.shape, generator=g) * 0.05)
            b.lora_down.weight.copy_(a.lora_down.weight.reshape(b.lora_down.weight.shape))
            a.lora_up.weight.copy_(b.lora_up.weight.reshape(a.lora_up.weight.shape))
    ref_net.apply_to()
    net.apply_to()
    net.build_arena("cpu", groups=nat.lora_groups())
    net.refresh_shadows(ref_ops)
    nat.attach_network(net)
    nat.prepare()
    lat, ts, ctx, added = _inputs(cfg)
    B, _, H, W = lat.shape
    if per_sample:
        mult = [1.0, -0.5][:B] if B <= 2 else [1.0, -0.5] * (B // 2)
        net.multiplier = mult
        ref_net.torch_multiplier = torch.tensor(mult)
    conv3 = [m for m in net.unet_loras if m.is_conv3x3]
    skipped = [m.lora_name for m in conv3 if float(provider(m.lora_name, "module", (1,), "cpu")) < drop["module_dropout"]]
    assert conv3 and 0 < len(skipped) < len(conv3)
    wgt = torch.randn(B, 4, H, W, generator=torch.Generator().manual_seed(11))
    preds = {}
    for mode in ("train", "eval"):
        getattr(ref_net, mode)()
        getattr(net, mode)()
        for p_ in ref_net.parameters():
            p_.grad = None
        with ref_net:
            pred_ref = ref(lat, ts, ctx, added)
            (pred_ref * wgt).sum().backward()
        with net:
            pred = nat.forward_native(_nhwc8(lat), ts, ctx, added, B=B, H=H, W=W)
            got = pred.view(B, H, W, 4).permute(0, 3, 1, 2)
            assert torch.allclose(got, pred_ref, rtol=2e-4, atol=2e-5), (mode, (got - pred_ref).abs().max())
            net.zero_grad_arena()
            nat.backward_native(wgt.permute(0, 2, 3, 1).reshape(B * H * W, 4).contiguous())
        preds[mode] = got.detach().clone()
        for a, b in zip(net.unet_loras, ref_net.unet_loras):
            if mode == "train" and b.lora_up.weight.grad is None:  # module_dropout fired
                assert float(a.lora_up.weight.grad.abs().max()) == 0.0 and float(a.lora_down.weight.grad.abs().max()) == 0.0, a.lora_name
                continue
            for x, y, nm in ((a.lora_down.weight.grad, b.lora_down.weight.grad, "down"), (a.lora_up.weight.grad, b.lora_up.weight.grad, "up")):
                err = ((x.reshape(-1) - y.reshape(-1)).norm() / (y.norm() + 1e-12)).item()
                assert err < 5e-4, (mode, a.lora_name, nm, err)
    assert not torch.allclose(preds["train"], preds["eval"], rtol=1e-3, atol=1e-4)  # the masks are live in training mode
