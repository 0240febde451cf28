"""DoRA through the fused graph (host logic on CPU, oracle kernel table): network_type='dora' must reproduce
 (a) the reference's own DoRA network on the tiny FLUX model — golden vectors written by tests/golden/make_golden.py from
     toolkit.lora_special.LoRASpecialNetwork(network_type='dora') — init draws, prediction, every gradient, saved file;
 (b) autograd of the oracle DoRA module on a larger configuration, through a full train step with AdamW."""
import json
import os

import pytest
import torch
from safetensors import safe_open
from safetensors.torch import load_file

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd.flux import FluxTransformer2DModel
from ai_toolkit_amd.lora import FusedLoRANetwork
from ai_toolkit_amd.trainer import FluxLoRATrainStep
from oracle import flux_ref, lora_ref, ref_ops
from tests.test_oracle_golden import TINY, oracle_model, tiny_inputs

G = os.path.join(os.path.dirname(__file__), "golden")


def test_fused_dora_matches_reference_golden(tmp_path):
    path = os.path.join(G, "dora_flux_tiny.safetensors")
    with safe_open(path, "pt") as f:
        meta = {k: json.loads(v) for k, v in f.metadata().items()}
    t = load_file(path)
    ref = oracle_model()
    nat = FluxTransformer2DModel(**TINY, dtype=torch.float32, device="cpu", ops=ref_ops)
    nat.load_state_dict(ref.state_dict(), strict=True)
    torch.manual_seed(99)
    net = FusedLoRANetwork(nat, lora_dim=8, network_type="dora")
    assert [m.lora_name for m in net.unet_loras] == meta["names"]
    for m in net.unet_loras:
        assert torch.equal(m.lora_down.weight, t[f"init/{m.lora_name}/down"]), m.lora_name  # same RNG consumption
        assert torch.allclose(m.magnitude, t[f"init/{m.lora_name}/magnitude"], rtol=1e-6)
        with torch.no_grad():
            m.lora_up.weight.copy_(t[f"set/{m.lora_name}/up"])
            m.magnitude.copy_(t[f"set/{m.lora_name}/magnitude"])
    net.apply_to()
    net.build_arena("cpu", groups=nat.lora_groups())
    net.refresh_shadows(ref_ops)
    nat.attach_network(net)
    nat.prepare()
    with net:
        pred = nat.forward_native(*tiny_inputs())
        assert torch.allclose(pred, t["fwd/pred"], rtol=2e-4, atol=2e-5), (pred - t["fwd/pred"]).abs().max()
        net.zero_grad_arena()
        nat.backward_native(t["fwd/w"])
    for m in net.unet_loras:
        for nm, p_ in (("down", m.lora_down.weight), ("up", m.lora_up.weight), ("magnitude", m.magnitude)):
            ref_g = t[f"grad/{m.lora_name}/{nm}"]
            err = ((p_.grad - ref_g).norm() / (ref_g.norm() + 1e-12)).item()
            assert err < 5e-4, (m.lora_name, nm, err)
    # saved file: same keys / values as the reference writes (order of keys inside a module differs: sorted compare)
    f = tmp_path / "dora.safetensors"
    net.save_weights(str(f), dtype=torch.float32, metadata={"name": "x"})
    sd = load_file(str(f))
    assert sorted(sd.keys()) == sorted(meta["saved_keys"])
    for k, v in sd.items():
        assert torch.allclose(v, t[f"saved/{k}"], rtol=1e-6), k
    # optimizer parameter order of the reference module: magnitude, lora_up, lora_down
    m0 = net.unet_loras[0]
    first3 = net.prepare_optimizer_params()[0]["params"][:3]
    assert first3[0] is m0.magnitude and first3[1] is m0.lora_up.weight and first3[2] is m0.lora_down.weight


CFG = dict(in_channels=64, num_layers=1, num_single_layers=2, attention_head_dim=128, num_attention_heads=2,
           joint_attention_dim=64, pooled_projection_dim=32)


@pytest.mark.parametrize("rank", [4, 80], ids=["r4", "r80_above_one_rank_chunk"])
def test_dora_train_steps_match_oracle_autograd_adamw(rank):
    """rank 80: above the 64 ranks one skinny launch contracts (toolkit/models/DoRA.py:126-148 has no rank limit) — the rank-space operands go out
    in 64-rank chunks like plain LoRA's and aitk_dora_colscale reads the 80 x 80 Gram matrix through the caches."""
    torch.manual_seed(0)
    ref = flux_ref.FluxTransformer2DModel(**CFG)
    flux_ref.init_synthetic_(ref, seed=1234, std=0.05)
    nat = FluxTransformer2DModel(**CFG, dtype=torch.float32, device="cpu", ops=ref_ops)
    nat.load_state_dict(ref.state_dict(), strict=True)
    torch.manual_seed(5)
    ref_net = lora_ref.RefLoRANetwork(ref, rank, network_type="dora")
    torch.manual_seed(5)
    net = FusedLoRANetwork(nat, lora_dim=rank, network_type="dora")
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for a, b in zip(net.unet_loras, ref_net.unet_loras):
            assert torch.equal(a.lora_down.weight, b.lora_down.weight)
            up = torch.randn(b.lora_up.weight.shape, generator=g) * 0.05
            a.lora_up.weight.copy_(up)
            b.lora_up.weight.copy_(up)
    ref_net.apply_to()
    net.apply_to()
    net.build_arena("cpu", groups=nat.lora_groups())
    net.refresh_shadows(ref_ops)
    nat.attach_network(net)
    nat.prepare()
    kw = dict(lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
    ours = FluxLoRATrainStep(nat, net, ref_ops, **kw)
    params = [p for m in ref_net.unet_loras for p in (m.magnitude, m.lora_up.weight, m.lora_down.weight)]
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=0.01, eps=1e-6)
    for it in range(3):
        gi = torch.Generator().manual_seed(20 + it)
        lat = torch.randn(2, 16, 8, 4, generator=gi)
        emb = torch.randn(2, 6, 64, generator=gi)
        pooled = torch.randn(2, 32, generator=gi)
        noise = torch.randn(2, 16, 8, 4, generator=gi)
        ts = torch.tensor([310.0, 845.0])
        loss = ours.step(lat, emb, pooled, noise=noise, timesteps=ts)
        tt = (ts / 1000).view(-1, 1, 1, 1)
        noisy = flux_ref.pack_latents((1 - tt) * lat + tt * noise)
        img_ids, txt_ids = flux_ref.make_ids(8, 4, 6)
        opt.zero_grad()
        with ref_net:
            pred = ref(noisy, emb, pooled, ts / 1000, img_ids, txt_ids, torch.ones(2))
            loss_ref = (pred - flux_ref.pack_latents(noise - lat)).pow(2).mean()
            loss_ref.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        assert abs(loss.item() - loss_ref.item()) < 2e-4 * max(1.0, abs(loss_ref.item())), (it, loss.item(), loss_ref.item())
    for a, b in zip(net.unet_loras, ref_net.unet_loras):
        assert torch.allclose(a.magnitude, b.magnitude, rtol=2e-3, atol=2e-6), a.lora_name
        assert torch.allclose(a.lora_up.weight, b.lora_up.weight, rtol=2e-3, atol=2e-6), a.lora_name
        assert torch.allclose(a.lora_down.weight, b.lora_down.weight, rtol=2e-3, atol=2e-6), a.lora_name


def test_dora_per_sample_multipliers_match_reference_golden():
    """Slider-style batch: network.multiplier = [1.0, 0.4].  The reference adds each sample's multiplier to the LoRA term and scales the DoRA
    weight by the MEAN multiplier (toolkit/network_mixins.py:313-340): y = c (W x + m_mean B A x) + b + (m_b - m_mean) B A x.  The fused graph
    runs the mean part through the column-scale epilogue and the deviation as a second, un-scaled rank-r term (graph._DoraPS); prediction and
    every gradient (magnitude / lora_up / lora_down) against the reference's own run."""
    path = os.path.join(G, "dora_flux_tiny.safetensors")
    with safe_open(path, "pt") as f:
        meta = {k: json.loads(v) for k, v in f.metadata().items()}
    t = load_file(path)
    ref = oracle_model()
    nat = FluxTransformer2DModel(**TINY, dtype=torch.float32, device="cpu", ops=ref_ops)
    nat.load_state_dict(ref.state_dict(), strict=True)
    torch.manual_seed(99)
    net = FusedLoRANetwork(nat, lora_dim=8, network_type="dora")
    with torch.no_grad():
        for m in net.unet_loras:
            m.lora_up.weight.copy_(t[f"set/{m.lora_name}/up"])
            m.magnitude.copy_(t[f"set/{m.lora_name}/magnitude"])
    net.apply_to()
    net.build_arena("cpu", groups=nat.lora_groups())
    net.refresh_shadows(ref_ops)
    nat.attach_network(net)
    nat.prepare()
    net.multiplier = meta["ps_multiplier"]
    assert net.multiplier_is_per_sample() and abs(net.multiplier_mean() - 0.7) < 1e-12
    with net:
        pred = nat.forward_native(*tiny_inputs())
        assert torch.allclose(pred, t["ps/pred"], rtol=2e-4, atol=2e-5), (pred - t["ps/pred"]).abs().max()
        assert not torch.allclose(pred, t["fwd/pred"], atol=1e-2)  # the uniform-multiplier prediction is a different one
        net.zero_grad_arena()
        nat.backward_native(t["fwd/w"])
    for m in net.unet_loras:
        for nm, p_ in (("down", m.lora_down.weight), ("up", m.lora_up.weight), ("magnitude", m.magnitude)):
            ref_g = t[f"ps/grad/{m.lora_name}/{nm}"]
            err = ((p_.grad - ref_g).norm() / (ref_g.norm() + 1e-12)).item()
            assert err < 5e-4, (m.lora_name, nm, err)
    # back to a uniform multiplier: the column scale follows the new mean, the plain path reproduces the uniform golden
    net.multiplier = 1.0
    with net:
        pred = nat.forward_native(*tiny_inputs())
    assert torch.allclose(pred, t["fwd/pred"], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("per_sample", [False, True], ids=["uniform_multiplier", "per_sample_multipliers"])
def test_dora_over_the_weight_only_fp8_base_matches_autograd_on_the_dequantised_weights(per_sample):
    """network.type dora with model.quantize: the reference's DoRAModule takes its norm over the DEQUANTISED weight (toolkit/models/DoRA.py:105-109,
    126-148: get_orig_weight -> weight.dequantize()).  The fused path expands the e4m3 codes for the skinny pass of refresh_dora and takes ||W_j||^2
    from the same codes (the quantised layer has released its bf16 copy); forward / backward multiply with the expansion the base GEMM uses anyway.
    Oracle: autograd over the restated DoRA module on a model holding the dequantised weights."""
    torch.manual_seed(0)
    ref = flux_ref.FluxTransformer2DModel(**CFG)
    flux_ref.init_synthetic_(ref, seed=1234, std=0.05)
    nat = FluxTransformer2DModel(**CFG, dtype=torch.float32, device="cpu", ops=ref_ops)
    nat.load_state_dict(ref.state_dict(), strict=True)
    nat.prepare()
    nat.quantize_base_fp8(release_bf16=True)  # before the network exists, as in the reference's load order; the plug-in releases the bf16 copies
    with torch.no_grad():
        mods = dict(ref.named_modules())
        nq = 0
        for n, lin in nat.named_modules():
            if getattr(lin, "qweight", None) is not None:
                assert lin.weight.numel() == 0  # the bf16 copy is gone: nothing but the codes can feed the norm
                mods[n].weight.copy_(nat.dequantized_weight(lin).float())
                nq += 1
        assert nq > 0
    torch.manual_seed(5)
    ref_net = lora_ref.RefLoRANetwork(ref, 4, network_type="dora")
    torch.manual_seed(5)
    net = FusedLoRANetwork(nat, lora_dim=4, network_type="dora")
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for a, b in zip(net.unet_loras, ref_net.unet_loras):
            assert torch.equal(a.lora_down.weight, b.lora_down.weight)
            up = torch.randn(b.lora_up.weight.shape, generator=g) * 0.05
            a.lora_up.weight.copy_(up)
            b.lora_up.weight.copy_(up)
            assert torch.allclose(a.magnitude, b.magnitude, rtol=1e-6, atol=0)  # both built over the dequantised weight (DoRA.py:105-109)
            a.magnitude.copy_(b.magnitude)
    ref_net.apply_to()
    net.apply_to()
    net.build_arena("cpu", groups=nat.lora_groups())
    net.refresh_shadows(ref_ops)
    nat.attach_network(net)
    mult = [0.5, 1.25] if per_sample else 1.0
    net.multiplier = mult
    ref_net.multiplier = mult
    net.refresh_dora(ref_ops)
    kw = dict(lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
    ours = FluxLoRATrainStep(nat, net, ref_ops, **kw)
    params = [p for m in ref_net.unet_loras for p in (m.magnitude, m.lora_up.weight, m.lora_down.weight)]
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=0.01, eps=1e-6)
    for it in range(2):
        gi = torch.Generator().manual_seed(20 + it)
        lat = torch.randn(2, 16, 8, 4, generator=gi)
        emb = torch.randn(2, 6, 64, generator=gi)
        pooled = torch.randn(2, 32, generator=gi)
        noise = torch.randn(2, 16, 8, 4, generator=gi)
        ts = torch.tensor([310.0, 845.0])
        loss = ours.step(lat, emb, pooled, noise=noise, timesteps=ts)
        tt = (ts / 1000).view(-1, 1, 1, 1)
        noisy = flux_ref.pack_latents((1 - tt) * lat + tt * noise)
        img_ids, txt_ids = flux_ref.make_ids(8, 4, 6)
        opt.zero_grad()
        with ref_net:
            pred = ref(noisy, emb, pooled, ts / 1000, img_ids, txt_ids, torch.ones(2))
            loss_ref = (pred - flux_ref.pack_latents(noise - lat)).pow(2).mean()
            loss_ref.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        assert abs(loss.item() - loss_ref.item()) < 2e-4 * max(1.0, abs(loss_ref.item())), (it, loss.item(), loss_ref.item())
    for a, b in zip(net.unet_loras, ref_net.unet_loras):
        assert torch.allclose(a.magnitude, b.magnitude, rtol=2e-3, atol=2e-6), a.lora_name
        assert torch.allclose(a.lora_up.weight, b.lora_up.weight, rtol=2e-3, atol=2e-6), a.lora_name
        assert torch.allclose(a.lora_down.weight, b.lora_down.weight, rtol=2e-3, atol=2e-6), a.lora_name
