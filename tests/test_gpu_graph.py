"""hipGraph replay of the step's launch sequence (trainer.capture / step_graphed) must be the SAME computation as the eager launch
sequence: every kernel is deterministic, so gradients, loss and the adapter after AdamW are compared bit for bit, on inputs that
differ from the ones the graph was captured with, for two bucket shapes sharing one memory pool."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _flux_pair():
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from tests.test_gpu_e2e import _build

    kw = dict(lr=1e-3, weight_decay=0.01, max_grad_norm=1.0, ema_decay=0.99)
    _, _, nat_a, net_a = _build()
    _, _, nat_b, net_b = _build()
    assert torch.equal(net_a.arena_p, net_b.arena_p)
    return FluxLoRATrainStep(nat_a, net_a, ops, **kw), FluxLoRATrainStep(nat_b, net_b, ops, **kw)


def test_flux_graph_replay_is_bitwise_the_eager_step():
    from tests.test_gpu_e2e import _batch

    eager, graphed = _flux_pair()
    shapes = [dict(B=2, Hl=16, Wl=12), dict(B=2, Hl=12, Wl=16), dict(B=2, Hl=16, Wl=12), dict(B=2, Hl=12, Wl=16)]
    for k, shp in enumerate(shapes):
        lat, emb, pooled, noise, ts = _batch(seed=50 + k, **shp)
        ts = ts + 3.0 * k
        le = eager.step(lat, emb, pooled, noise=noise, timesteps=ts).clone()
        ge = eager.network.arena_g.clone()
        lg = graphed.step_graphed(latents=lat, prompt_embeds=emb, pooled_embeds=pooled, noise=noise, timesteps=ts).clone()
        assert torch.equal(le, lg), (k, le.item(), lg.item())
        assert torch.equal(ge, graphed.network.arena_g), k
        assert torch.equal(eager.network.arena_p, graphed.network.arena_p), k
        assert torch.equal(eager.network.arena_ema, graphed.network.arena_ema), k
    assert len(graphed._graphs) == 2  # one graph per bucket shape, replayed twice each


def test_flux_graph_with_loss_weights_and_sampled_inputs_runs():
    """timesteps / noise drawn by the step itself (host sampling stays outside the graph) + per-sample loss weights."""
    from tests.test_gpu_e2e import _batch

    eager, graphed = _flux_pair()
    for st in (eager, graphed):
        st.gen = torch.Generator(device="cuda")
        st.gen.manual_seed(99)
        st.linear_timesteps = True
    for k in range(2):
        lat, emb, pooled, _, _ = _batch(2, seed=70 + k)
        le = eager.step(lat, emb, pooled).clone()
        lg = graphed.step_graphed(latents=lat, prompt_embeds=emb, pooled_embeds=pooled).clone()
        assert torch.equal(le, lg), (le.item(), lg.item())
        assert torch.equal(eager.network.arena_p, graphed.network.arena_p)


@pytest.mark.parametrize("sdxl", [False, True], ids=["sd15", "sdxl"])
def test_unet_graph_replay_is_bitwise_the_eager_step(sdxl):
    from ai_toolkit_amd.trainer import UNetLoRATrainStep
    from tests.test_gpu_unet import MID_SD15, MID_SDXL, _batch, _pair

    cfg, ref, ref_net, native, finish, ops = _pair(MID_SDXL if sdxl else MID_SD15, sdxl)
    kw = dict(lr=1e-3, weight_decay=0.01, max_grad_norm=1.0, min_snr_gamma=5.0)
    steps = [UNetLoRATrainStep(*finish(*native(ops), ops), ops, **kw) for _ in range(2)]
    eager, graphed = steps
    for k in range(3):
        lat, ctx, pooled, noise, ts = _batch(cfg, seed=80 + k)
        ts = (ts + 7 * k) % 1000
        le = eager.step(lat, ctx, pooled if sdxl else None, noise=noise, timesteps=ts).clone()
        lg = graphed.step_graphed(latents=lat, prompt_embeds=ctx, pooled_embeds=pooled if sdxl else None, noise=noise, timesteps=ts).clone()
        assert torch.equal(le, lg), (k, le.item(), lg.item())
        assert torch.equal(eager.network.arena_g, graphed.network.arena_g), k
        assert torch.equal(eager.network.arena_p, graphed.network.arena_p), k


def test_flux_graph_replay_with_dropout_draws_fresh_masks_every_replay():
    """Round 6 (one more refusal lifted): dropout / rank_dropout masks are device-side torch.rand draws on the default CUDA generator, whose offset
    every replay of a captured graph advances — so `step_graphed` may capture a network with them (module_dropout, a host-side coin that changes
    the launch list, is still refused).  Checked: the first replay equals the eager step started from the same generator state bit for bit (same
    draws in the same order), consecutive replays on the same batch give different losses (fresh masks), eval mode replays are mask-free."""
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from oracle.pairs import batch, build

    cfg = dict(dropout=0.1, rank_dropout=0.25)
    kw = dict(lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
    _, _, nat_a, net_a = build(16, dropout_cfg=cfg, mask_provider=None)
    _, _, nat_b, net_b = build(16, dropout_cfg=cfg, mask_provider=None)
    from ai_toolkit_amd.lora import _default_mask_provider

    net_a.mask_provider = net_b.mask_provider = _default_mask_provider
    net_a.train()
    net_b.train()
    assert net_a.dropout_is_capturable()
    eager, graphed = FluxLoRATrainStep(nat_a, net_a, ops, **kw), FluxLoRATrainStep(nat_b, net_b, ops, **kw)
    lat, emb, pooled, noise, ts = batch(2, seed=90)
    graphed.capture(latents=lat, prompt_embeds=emb, pooled_embeds=pooled, noise=noise, timesteps=ts)  # warm-up + capture consume draws of their own
    p0 = net_b.arena_p.clone()
    assert torch.equal(net_a.arena_p, p0)  # capture does not step
    torch.cuda.manual_seed(1234)
    l_e = eager.step(lat, emb, pooled, noise=noise, timesteps=ts).clone()
    torch.cuda.manual_seed(1234)
    l_g = graphed.step_graphed(latents=lat, prompt_embeds=emb, pooled_embeds=pooled, noise=noise, timesteps=ts).clone()
    assert torch.equal(l_e, l_g), (l_e.item(), l_g.item())
    assert torch.equal(net_a.arena_g, net_b.arena_g) and torch.equal(net_a.arena_p, net_b.arena_p)
    losses = [graphed.step_graphed(latents=lat, prompt_embeds=emb, pooled_embeds=pooled, noise=noise, timesteps=ts).item() for _ in range(3)]
    assert len({round(x, 7) for x in losses}) == 3, losses
    # module_dropout is a host decision: still refused
    _, _, nat_c, net_c = build(16, dropout_cfg=dict(module_dropout=0.2), mask_provider=None)
    net_c.mask_provider = _default_mask_provider
    net_c.train()
    with pytest.raises(NotImplementedError, match="module_dropout"):
        FluxLoRATrainStep(nat_c, net_c, ops, **kw).capture(latents=lat, prompt_embeds=emb, pooled_embeds=pooled, noise=noise, timesteps=ts)
