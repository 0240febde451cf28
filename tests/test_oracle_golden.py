"""Pins the oracle (and the native host logic driven by the oracle's kernel table) to vectors produced by the
REFERENCE'S OWN CODE (tests/golden/make_golden.py executed toolkit.lora_special / custom_flowmatch_sampler under shims)."""
import json
import os

import pytest
import torch
from safetensors import safe_open
from safetensors.torch import load_file

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd.flowmatch import FlowMatchTrainSchedule, calculate_shift
from ai_toolkit_amd.flux import FluxTransformer2DModel
from ai_toolkit_amd.lora import FusedLoRANetwork
from oracle import flux_ref, lora_ref, ref_ops

G = os.path.join(os.path.dirname(__file__), "golden")
TINY = dict(in_channels=64, num_layers=1, num_single_layers=1, attention_head_dim=128, num_attention_heads=2,
            joint_attention_dim=64, pooled_projection_dim=32)


@pytest.fixture(scope="module")
def gold():
    path = os.path.join(G, "lora_flux_tiny.safetensors")
    with safe_open(path, "pt") as f:
        meta = {k: json.loads(v) for k, v in f.metadata().items()}
    return load_file(path), meta


def tiny_inputs():
    g = torch.Generator().manual_seed(3)
    Hl, Wl, n_txt, B = 4, 4, 5, 2
    hidden = torch.randn(B, (Hl // 2) * (Wl // 2), 64, generator=g)
    enc = torch.randn(B, n_txt, 64, generator=g)
    pooled = torch.randn(B, 32, generator=g)
    t = torch.tensor([0.3, 0.8])
    img_ids, txt_ids = flux_ref.make_ids(Hl, Wl, n_txt)
    return hidden, enc, pooled, t, img_ids, txt_ids, torch.ones(B)


def oracle_model():
    torch.manual_seed(0)
    model = flux_ref.FluxTransformer2DModel(**TINY)
    flux_ref.init_synthetic_(model, seed=1234, std=0.05)
    return model


def test_oracle_lora_matches_reference_names_init_forward_grads_and_saved_file(gold):
    t, meta = gold
    model = oracle_model()
    torch.manual_seed(99)
    net = lora_ref.RefLoRANetwork(model, 8)
    assert [m.lora_name for m in net.unet_loras] == meta["names"]
    assert meta["peft_format"] is True and meta["scale"] == 1.0  # alpha forced to rank (toolkit/lora_special.py:428-433)
    for m in net.unet_loras:
        assert torch.equal(m.lora_down.weight, t[f"init/{m.lora_name}/down"]), m.lora_name  # same RNG consumption
        with torch.no_grad():
            m.lora_up.weight.copy_(t[f"warm/{m.lora_name}/up"])
    net.apply_to()
    for tag, mult in (("m1", [1.0]), ("mvec", [0.5, -1.5])):
        net.torch_multiplier = torch.tensor(mult)
        net.zero_grad()
        with net:
            pred = model(*tiny_inputs())
            pred.square().sum().backward()
        assert torch.allclose(pred, t[f"{tag}/pred"], rtol=1e-5, atol=1e-6)
        for m in net.unet_loras:
            assert torch.allclose(m.lora_down.weight.grad, t[f"{tag}/grad/{m.lora_name}/down"], rtol=1e-4, atol=1e-6)
            assert torch.allclose(m.lora_up.weight.grad, t[f"{tag}/grad/{m.lora_name}/up"], rtol=1e-4, atol=1e-6)
    sd = net.peft_state_dict(torch.float32)
    assert list(sd.keys()) == meta["saved_keys"]
    for k, v in sd.items():
        assert torch.equal(v, t[f"saved/{k}"]), k


def test_native_network_and_host_graph_match_reference_vectors(gold, tmp_path):
    t, meta = gold
    ref = oracle_model()
    nat = FluxTransformer2DModel(**TINY, dtype=torch.float32, device="cpu", ops=ref_ops)
    nat.load_state_dict(ref.state_dict(), strict=True)
    torch.manual_seed(99)
    net = FusedLoRANetwork(nat, lora_dim=8, alpha=1.0)
    assert [m.lora_name for m in net.unet_loras] == meta["names"]
    for m in net.unet_loras:
        assert torch.equal(m.lora_down.weight, t[f"init/{m.lora_name}/down"]), m.lora_name
        assert m.scale == 1.0 and float(m.alpha) == meta["alpha"]
        with torch.no_grad():
            m.lora_up.weight.copy_(t[f"warm/{m.lora_name}/up"])
    net.apply_to(None, nat, False, True)
    net.force_to("cpu", torch.float32)
    net.refresh_shadows(ref_ops)
    nat.attach_network(net)
    for tag, mult in (("m1", 1.0), ("mvec", [0.5, -1.5])):
        net.multiplier = mult
        with net:
            pred = nat.forward_native(*tiny_inputs())
            assert torch.allclose(pred, t[f"{tag}/pred"], rtol=1e-4, atol=1e-5)
            net.zero_grad_arena()
            nat.backward_native((2 * pred).detach())
        for m in net.unet_loras:
            assert torch.allclose(m.lora_down.weight.grad, t[f"{tag}/grad/{m.lora_name}/down"], rtol=2e-4, atol=1e-5), m.lora_name
            assert torch.allclose(m.lora_up.weight.grad, t[f"{tag}/grad/{m.lora_name}/up"], rtol=2e-4, atol=1e-5), m.lora_name
    # saved file: same keys, same values, loadable back (PEFT format, toolkit/network_mixins.py:607-624)
    f = str(tmp_path / "lora.safetensors")
    net.save_weights(f, dtype=torch.float32, metadata={"training_info": {"step": 3, "epoch": 0}, "name": "x"})
    sd = load_file(f)
    assert sorted(sd.keys()) == sorted(meta["saved_keys"])
    for k, v in sd.items():
        assert torch.equal(v, t[f"saved/{k}"]), k
    with safe_open(f, "pt") as fh:
        md = fh.metadata()
    assert md["format"] == "pt" and json.loads(md["training_info"]) == {"step": 3, "epoch": 0}
    before = net.arena_p.clone()
    net.arena_p.zero_()
    assert net.load_weights(f) is None
    assert torch.equal(net.arena_p, before)
    # state_dict surface: <lora_name>.lora_down.weight / .lora_up.weight / .alpha ; _runtime_scale not persisted
    keys = list(net.state_dict().keys())
    assert keys[:3] == meta["state_dict_keys_first"][:3]
    assert not any("_runtime_scale" in k for k in keys)


def test_flowmatch_schedule_matches_reference_scheduler():
    t = load_file(os.path.join(G, "flowmatch.safetensors"))
    s = FlowMatchTrainSchedule()
    assert torch.equal(s.set_train_timesteps(1000, "cpu", "linear"), t["linear"])
    torch.manual_seed(123)
    assert torch.equal(s.set_train_timesteps(1000, "cpu", "sigmoid"), t["sigmoid_seed123"])
    torch.manual_seed(321)
    assert torch.equal(s.set_train_timesteps(1000, "cpu", "lognorm_blend"), t["lognorm_blend_seed321"])
    x0, eps, ts = t["x0"], t["eps"], t["ts"]
    B, Cc, Hh, W = x0.shape
    noisy = torch.empty(B, Hh * W // 4, Cc * 4)
    target = torch.empty_like(noisy)
    ref_ops.flow_noise_pack(x0, eps, ts, noisy, target)
    assert torch.allclose(flux_ref.unpack_latents(noisy, Hh, W), t["noisy"], rtol=1e-6, atol=1e-6)
    assert torch.allclose(flux_ref.unpack_latents(target, Hh, W), eps - x0, rtol=1e-6, atol=1e-6)
    cs = torch.tensor([calculate_shift(n) for n in (256, 1024, 4096, 3952)], dtype=torch.float64)
    assert torch.allclose(cs, t["calc_shift"])
    # bell-shaped per-timestep loss weights (linear_timesteps / linear_timesteps2) of the reference scheduler
    s.set_train_timesteps(1000, "cpu", "linear")
    assert torch.allclose(s.get_weights_for_timesteps(t["tw_ts"], v2=False), t["tw_v1"], rtol=1e-6, atol=1e-7)
    assert torch.allclose(s.get_weights_for_timesteps(t["tw_ts"], v2=True), t["tw_v2"], rtol=1e-6, atol=1e-7)
    # timestep_type 'weighted' (custom_flowmatch_sampler.py:65-70, 116): linear table, empirical per-index weights
    assert torch.equal(s.set_train_timesteps(1000, "cpu", "weighted"), t["tw_weighted_table"])
    w = s.get_weights_for_timesteps(t["tw_ts"], timestep_type="weighted")
    assert w.dtype == t["tw_weighted"].dtype and torch.equal(w, t["tw_weighted"])


def test_timestep_index_sampling_modes():
    """content / style cubic sampling and the degenerate balanced case (BaseSDTrainProcess.py:1275-1318)."""
    s = FlowMatchTrainSchedule()
    s.set_train_timesteps(1000, "cpu", "linear")
    g = torch.Generator().manual_seed(4)
    u = torch.rand((64,), generator=torch.Generator().manual_seed(4))
    ts, idx = s.sample_timesteps(64, "cpu", generator=g, content_or_style="content")
    want = ((u ** 3 * 1000) * 999 / 999).long().clamp(0, 999)  # value_map(0..999 -> min_idx..max_idx) then clamp
    assert torch.equal(idx, want) and torch.equal(ts, s.timesteps[want])
    _, idx_s = s.sample_timesteps(64, "cpu", generator=torch.Generator().manual_seed(4), content_or_style="style")
    assert torch.equal(idx_s, (((1 - u ** 3) * 1000) * 999 / 999).long().clamp(0, 999))
    assert idx.float().mean() < idx_s.float().mean()  # content favours early table entries (high noise), style late ones
    _, idx_b = s.sample_timesteps(5, "cpu", min_idx=7, max_idx=7)
    assert idx_b.tolist() == [7] * 5


def test_pack_unpack_roundtrip_and_ids():
    x = torch.randn(2, 16, 8, 6)
    p = flux_ref.pack_latents(x)
    assert p.shape == (2, 12, 64)
    assert torch.equal(flux_ref.unpack_latents(p, 8, 6), x)
    img_ids, txt_ids = flux_ref.make_ids(8, 6, 5)
    assert img_ids.shape == (12, 3) and txt_ids.shape == (5, 3)
    assert img_ids[4].tolist() == [0.0, 1.0, 1.0]


def test_oracle_dora_matches_reference_dora_network():
    """oracle/lora_ref.RefDoRAModule vs the reference's DoRAModule (tests/golden/dora_flux_tiny.safetensors, produced by
    LoRASpecialNetwork(network_type='dora') under the import shims): init draws under the same seed, forward, all gradients
    (magnitude / lora_up / lora_down) and saved keys."""
    path = os.path.join(G, "dora_flux_tiny.safetensors")
    with safe_open(path, "pt") as f:
        meta = {k: json.loads(v) for k, v in f.metadata().items()}
    t = load_file(path)
    model = oracle_model()
    torch.manual_seed(99)
    net = lora_ref.RefLoRANetwork(model, 8, network_type="dora")
    assert [m.lora_name for m in net.unet_loras] == meta["names"]
    assert [n for n, _ in net.unet_loras[0].named_parameters()] == meta["param_order"]
    for m in net.unet_loras:
        assert torch.equal(m.lora_down.weight, t[f"init/{m.lora_name}/down"]), m.lora_name
        assert torch.allclose(m.magnitude, t[f"init/{m.lora_name}/magnitude"], rtol=1e-6), m.lora_name
        with torch.no_grad():
            m.lora_up.weight.copy_(t[f"set/{m.lora_name}/up"])
            m.magnitude.copy_(t[f"set/{m.lora_name}/magnitude"])
    net.apply_to()
    with net:
        pred = model(*tiny_inputs())
        (pred * t["fwd/w"]).sum().backward()
    assert torch.allclose(pred, t["fwd/pred"], rtol=1e-5, atol=1e-6)
    for m in net.unet_loras:
        for nm, p_ in (("down", m.lora_down.weight), ("up", m.lora_up.weight), ("magnitude", m.magnitude)):
            assert torch.allclose(p_.grad, t[f"grad/{m.lora_name}/{nm}"], rtol=2e-4, atol=2e-6), (m.lora_name, nm)
    sd = net.peft_state_dict(torch.float32)
    assert sorted(sd.keys()) == sorted(meta["saved_keys"])
    for k, v in sd.items():
        assert torch.allclose(v, t[f"saved/{k}"]), k


def test_fused_optimizer_tail_matches_torch_adamw_and_reference_ema_class():
    """clip_grad_norm_ -> torch.optim.AdamW(eps=1e-6, wd=0.01) -> the reference's toolkit/ema.py ExponentialMovingAverage.update(),
    executed by make_golden.py for three steps (grad norm below / far above / above the clip threshold): the oracle's
    adamw_ema_step (what the HIP kernel is tested against on the GPU) lands on the same parameters and EMA shadow."""
    t = load_file(os.path.join(G, "optimizer_ema.safetensors"))
    p, ema = t["p0"].clone(), t["p0"].clone()
    m, v, norm = torch.zeros_like(p), torch.zeros_like(p), torch.zeros(1)
    for k in range(3):
        ref_ops.adamw_ema_step(p, t["grads"][k].clone(), m, v, lr=3e-3, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=0.01, step=k + 1,
                               max_norm=1.0, ema=ema, ema_decay=0.9, norm_out=norm)
        assert torch.allclose(norm[0], t["norms"][k], rtol=1e-5)
    assert torch.allclose(p, t["p3"], rtol=1e-5, atol=1e-7), (p - t["p3"]).abs().max()
    assert torch.allclose(ema, t["ema3"], rtol=1e-5, atol=1e-7), (ema - t["ema3"]).abs().max()


def test_merge_in_equals_reference_merge_in(gold):
    """FusedLoRANetwork.merge_in(0.7) (rank-r accumulate GEMM on the base weight) against base weights merged by the reference's
    own LoRASpecialNetwork.merge_in(0.7) (tests/golden/merge_flux_tiny.safetensors); merge_out restores the originals."""
    t, meta = gold
    want = load_file(os.path.join(G, "merge_flux_tiny.safetensors"))
    ref = oracle_model()
    nat = FluxTransformer2DModel(**TINY, dtype=torch.float32, device="cpu", ops=ref_ops)
    nat.load_state_dict(ref.state_dict(), strict=True)
    torch.manual_seed(99)
    net = FusedLoRANetwork(nat, lora_dim=8, alpha=1.0)
    with torch.no_grad():
        for m in net.unet_loras:
            m.lora_up.weight.copy_(t[f"warm/{m.lora_name}/up"])
    net.apply_to(None, nat, False, True)
    net.force_to("cpu", torch.float32)
    nat.attach_network(net)
    nat.prepare()
    before = {k: nat.state_dict()[k].clone() for k in want}
    net.merge_in(0.7, ops=ref_ops)
    assert net.is_merged_in
    for k, v in want.items():  # the golden keeps the first 24 rows of each merged weight
        got = nat.state_dict()[k][:24]
        assert torch.allclose(got, v, rtol=1e-5, atol=1e-6) and not torch.allclose(got, before[k][:24], atol=1e-4), (k, (got - v).abs().max())
    net.merge_out(0.7, ops=ref_ops)
    for k, v in before.items():
        assert torch.allclose(nat.state_dict()[k], v, rtol=1e-5, atol=1e-6), k


def test_n_step_timestep_types_follow_the_trainer():
    """timestep_type one_step / two_step / four_step / eight_step (jobs/process/BaseSDTrainProcess.py:1196-1203, 1254-1272): a linear table,
    indices from a fixed list drawn with Python's `random.choices` (same global-RNG consumption as the reference)."""
    import random

    from ai_toolkit_amd.flowmatch import FlowMatchTrainSchedule

    s = FlowMatchTrainSchedule()
    lin = torch.linspace(1000, 1, 1000)
    for name, choices in (("two_step", [0, 499]), ("four_step", [0, 250, 500, 750]), ("eight_step", [0, 125, 250, 375, 500, 625, 750, 875])):
        s.set_train_timesteps(1000, "cpu", name)
        assert torch.equal(s.timesteps, lin)
        random.seed(5)
        t, idx = s.sample_timesteps(16, "cpu")
        random.seed(5)
        want = torch.tensor(random.choices(choices, k=16))
        assert torch.equal(idx, want) and torch.equal(t, lin[want])
    s.set_train_timesteps(1000, "cpu", "one_step")
    t, idx = s.sample_timesteps(4, "cpu")
    assert torch.equal(idx, torch.zeros(4, dtype=torch.long)) and torch.equal(t, torch.full((4,), 1000.0))


def test_signal_and_batch_noise_correction_follow_the_trainer():
    """do_signal_correction_noise / do_batch_noise_correction (jobs/process/BaseSDTrainProcess.py:1353-1376) restated line by line next to
    flowmatch.get_noise under the same seeds."""
    from ai_toolkit_amd.flowmatch import get_noise

    lat = torch.randn(3, 4, 6, 5, generator=torch.Generator().manual_seed(1))
    torch.manual_seed(11)
    got = get_noise(lat, None, noise_multiplier=0.9, signal_correction_noise_scale=0.3, batch_noise_correction_scale=0.2)
    torch.manual_seed(11)
    noise = torch.randn(lat.shape) * 0.9
    batch_noise = lat.clone()
    noise = noise + batch_noise * (torch.randn(3, 4, 1, 1) * 0.3)
    batch_noise = lat.clone().roll(shifts=torch.randint(1, 3, (1,)).item(), dims=0)
    noise = noise + batch_noise * (torch.randn(3, 4, 1, 1) * 0.2)
    assert torch.equal(got, noise)
    # batch of one: the batch correction is skipped like the reference does
    torch.manual_seed(3)
    a = get_noise(lat[:1], None, batch_noise_correction_scale=0.2)
    torch.manual_seed(3)
    assert torch.equal(a, torch.randn(lat[:1].shape))
