"""Regression tests for the round-2 review findings (host logic on CPU, oracle kernel table)."""
import pytest
import torch

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd.flowmatch import get_noise
from ai_toolkit_amd.trainer import FluxLoRATrainStep
from oracle import lora_ref, ref_ops
from tests.test_host_graph_cpu import CFG, build_pair
from tests.test_lokr_lowrank_cpu import R, _ready, native_pair
from tests.test_oracle_golden import tiny_inputs


def test_lowrank_lokr_through_the_autograd_bridge_after_set_to_none():
    """`_FluxGraphFn.backward` asked `lora_down.weight.grad` whether the optimizer had dropped the .grad views —
    a low-rank LoKr module has no `lora_down`.  model.forward + loss.backward() + zero_grad(set_to_none=True), twice, must equal
    the oracle network's autograd gradients."""
    ref, nat, net = native_pair()
    torch.manual_seed(99)
    ref_net = lora_ref.RefLoRANetwork(ref, R, network_type="lokr")
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for b in ref_net.unet_loras:
            b.lokr_w2_b.copy_(torch.randn(b.lokr_w2_b.shape, generator=g) * 0.2)
    ref_net.apply_to()
    _ready(net, nat)
    with torch.no_grad():
        for a, b in zip(net.unet_loras, ref_net.unet_loras):
            a.lokr_w1.copy_(b.lokr_w1)
            a.lokr_w2_a.copy_(b.lokr_w2_a)
            a.lokr_w2_b.copy_(b.lokr_w2_b)
    net.refresh_shadows(ref_ops)
    inputs = tiny_inputs()
    w = torch.randn(ref(*inputs).shape, generator=g)
    params = net.prepare_optimizer_params(default_lr=1e-3)[0]["params"]
    opt = torch.optim.SGD(params, lr=0.0)
    for rep in range(2):
        for m in ref_net.unet_loras:
            for p in m.parameters():
                p.grad = None
        with ref_net:
            (ref(*inputs) * w).sum().backward()
        opt.zero_grad(set_to_none=True)
        assert net.grads_dropped()
        with net:
            pred = nat(*inputs)[0]
            (pred * w).sum().backward()
        assert not net.grads_dropped()
        for a, b in zip(net.unet_loras, ref_net.unet_loras):
            for k in ("lokr_w1", "lokr_w2_a", "lokr_w2_b"):
                assert torch.allclose(getattr(a, k).grad, getattr(b, k).grad, rtol=3e-4, atol=1e-5), (rep, a.lora_name, k)


def _step(**kw):
    ref, ref_net, nat, net = build_pair(rank=4)
    return FluxLoRATrainStep(nat, net, ref_ops, lr=0.0, weight_decay=0.0, max_grad_norm=0.0, seed=11, **kw)


def test_noise_is_drawn_from_the_unscaled_latents():
    """latent_multiplier is applied AFTER the noise has been drawn and shaped from the latents
    (jobs/process/BaseSDTrainProcess.py:1323-1401): with a latent-derived noise term (dynamic_noise_offset) the scaled-latents
    order gave noise off by the multiplier."""
    opts = dict(dynamic_noise_offset=True)
    st = _step(latent_multiplier=3.0, noise_options=opts)
    g = torch.Generator().manual_seed(2)
    lat = torch.randn(2, 16, 8, 4, generator=g) + 0.5
    emb = torch.randn(2, 6, CFG["joint_attention_dim"], generator=g)
    pooled = torch.randn(2, CFG["pooled_projection_dim"], generator=g)
    p = st._prepare(lat, emb, pooled)
    gen = torch.Generator().manual_seed(11)
    st2 = _step()  # same seed: identical timestep draw ahead of the noise draw
    st2.gen = gen
    st2.schedule.set_train_timesteps(1000, "cpu", "linear", latents=lat, patch_size=2)
    st2.schedule.sample_timesteps(2, "cpu", generator=gen)
    want = get_noise(lat, gen, dtype=torch.float32, **opts)
    assert torch.equal(p["noise"], want)
    assert torch.equal(p["latents"], lat * 3.0)
    # and it differs from what the scaled latents would have produced
    gen3 = torch.Generator().manual_seed(11)
    st2.schedule.sample_timesteps(2, "cpu", generator=gen3)
    assert not torch.equal(get_noise(lat * 3.0, gen3, dtype=torch.float32, **opts), want)


def test_linear_timesteps_force_the_linear_table():
    """linear_timesteps / linear_timesteps2 make the reference set the scheduler to 'linear' whatever timestep_type
    says (BaseSDTrainProcess.py:1196-1203); the bell weights then index that table."""
    st = _step(timestep_type="sigmoid", linear_timesteps=True)
    assert st._table_type() == "linear"
    g = torch.Generator().manual_seed(2)
    lat = torch.randn(2, 16, 8, 4, generator=g)
    st._prepare(lat, torch.randn(2, 6, CFG["joint_attention_dim"], generator=g), torch.randn(2, CFG["pooled_projection_dim"], generator=g))
    lin = _step(timestep_type="linear")
    lin.schedule.set_train_timesteps(1000, "cpu", "linear", latents=lat, patch_size=2)
    assert torch.equal(st.schedule.timesteps, lin.schedule.timesteps)
    sig = _step(timestep_type="sigmoid")
    assert sig._table_type() == "sigmoid"
    w = _step(timestep_type="weighted")
    assert w._table_type() == "weighted"


def test_graph_capture_refuses_dropout_and_keeps_per_batch_loss_buffers():
    """ADVICE r2 (low + medium): host-drawn dropout decisions cannot be replayed from a captured graph; the per-sample loss buffer of
    a batch size is created once (a captured graph points at it).  Round 6: only the module_dropout coin (and a custom mask provider) is a
    host decision — dropout / rank_dropout alone are device-side draws and capturable (tests/test_gpu_graph.py)."""
    ref, ref_net, nat, net = build_pair(rank=4)
    net.dropout = 0.25
    for m in net.unet_loras:
        m.dropout = 0.25
    assert net.has_dropout and net.dropout_is_capturable()
    net.mask_provider = lambda *a, **k: None  # a custom provider: draws are no longer known to be on the graph-safe generator
    assert not net.dropout_is_capturable()
    from ai_toolkit_amd.lora import _default_mask_provider
    net.mask_provider = _default_mask_provider
    net.module_dropout = 0.25
    for m in net.unet_loras:
        m.module_dropout = 0.25
    assert not net.dropout_is_capturable()
    st = FluxLoRATrainStep(nat, net, ref_ops, lr=0.0, weight_decay=0.0, max_grad_norm=0.0, seed=1)
    net.train()
    g = torch.Generator().manual_seed(2)
    batch = dict(latents=torch.randn(2, 16, 8, 4, generator=g), prompt_embeds=torch.randn(2, 6, CFG["joint_attention_dim"], generator=g),
                 pooled_embeds=torch.randn(2, CFG["pooled_projection_dim"], generator=g))
    with pytest.raises(NotImplementedError):
        st.capture(**batch)
    st2 = _step()
    st2.step(**batch)
    buf2 = st2.loss_per_sample
    one = {k: v[:1] for k, v in batch.items()}
    st2.step(**one)
    assert st2.loss_per_sample.numel() == 1
    st2.step(**batch)
    assert st2.loss_per_sample is buf2
