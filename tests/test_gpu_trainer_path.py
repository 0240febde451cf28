"""The reference trainer's per-step sequence over the plug-in ON THE HIP KERNELS (tools/trainer_harness.TrainerLoop: SDTrainer.hook_train_loop's
calls over an adopted network), with its optimizer.step() / ema.update() served by the arena kernels (ai_toolkit_amd/adopt.py) — against the
same loop left to torch's foreach AdamW and the EMA class's Python loop.  CPU twin (bit-for-bit, kernel table = oracle): tests/test_trainer_fusion_cpu.py."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


@pytest.fixture(autouse=True)
def _env():
    old = os.environ.get("AITK_FUSE_TRAINER_STEP")
    yield
    if old is None:
        os.environ.pop("AITK_FUSE_TRAINER_STEP", None)
    else:
        os.environ["AITK_FUSE_TRAINER_STEP"] = old


@pytest.mark.parametrize("opts", [dict(), dict(feedback=10.0), dict(mult=0.999), dict(feedback=10.0, mult=1.001)], ids=["plain", "feedback", "mult", "both"])
def test_ema_update_kernel_is_the_reference_loop_bit_for_bit(opts):
    """aitk_ema_update against the tensor ops of toolkit/ema.py:126-152 on the same device (every op rounded on its own in both), odd length
    (scalar tail), 16-byte aligned arenas."""
    from ai_toolkit_amd import ops

    g = torch.Generator(device="cuda").manual_seed(3)
    n = 4096 * 37 + 13
    p = torch.randn(n, device="cuda", generator=g)
    s = p + 0.01 * torch.randn(n, device="cuda", generator=g)
    p2, s2 = p.clone(), s.clone()
    decay = 0.9937
    ops.ema_update(p, s, decay=decay, ema_feedback=opts.get("feedback", 0.0), param_multiplier=opts.get("mult", 1.0))
    tmp = s2 - p2
    tmp.mul_(1.0 - decay)
    s2.sub_(tmp)
    if opts.get("feedback"):
        p2.add_(tmp * 10)
    if opts.get("mult", 1.0) != 1.0:
        p2.mul_(opts["mult"])
    assert torch.equal(s, s2)
    assert torch.equal(p, p2)


def _loop(fuse, rank=16):
    from ai_toolkit_amd.plugin import Flux1MI355Model
    from oracle.pairs import build
    from tools.trainer_harness import TrainerLoop

    os.environ["AITK_FUSE_TRAINER_STEP"] = "1" if fuse else "0"
    _, _, nat, none = build(attach=False)
    assert none is None
    sd = Flux1MI355Model("cuda", model=nat, dtype=bf)
    loop = TrainerLoop(sd, rank=rank, lr=1e-3, device="cuda", seed=0)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for m in loop.network.unet_loras:
            m.lora_up.weight.copy_((torch.randn(m.lora_up.weight.shape, generator=g) * 0.02).cuda())
    return loop, nat


def test_trainer_loop_on_the_hip_kernels_arena_tail_against_torchs_loops():
    """Three iterations of the trainer loop on the HIP kernels with the optimizer / EMA served by the arena kernels.  Beside it, torch's own
    AdamW + the EMA class's loop run on CLONES fed with the same clipped gradients (a second full loop would not do: one ulp of difference in
    the parameters moves a bf16 prediction — and with it the next gradient — by ~3e-3, tools/gpu_trainer_fusion_diag.py): the two tails
    are two fp32 formulations of one update and must agree to rounding, step after step."""
    from ai_toolkit_amd import adopt
    from ai_toolkit_amd.adopt import AdoptedNetwork
    from oracle.pairs import batch
    a, nat_a = _loop(True)
    os.environ["AITK_FUSE_TRAINER_STEP"] = "1"
    twin_p = None
    for k in range(3):
        lat, emb, pooled, _, _ = batch(2, seed=60 + k)
        if twin_p is None:  # first step: adoption happens inside; the twin starts from the same values
            twin_p = [torch.nn.Parameter(p.detach().clone()) for p in a.params]
            twin_opt = torch.optim.AdamW(twin_p, lr=1e-3, eps=1e-6, weight_decay=0.01)
            twin_shadow = [s_.detach().clone() for s_ in a.ema.shadow_params]  # as the EMA was constructed (before _loop's warm start of lora_up)
        # the trainer's sequence by hand, so that the clipped gradients can be handed to the twin
        from types import SimpleNamespace

        s0 = dict(adopt.STATS)
        a.optimizer.zero_grad()
        noisy, ts, target = a.process_batch(lat)
        with a.network:
            pred = a.sd.get_noise_prediction(noisy, ts, SimpleNamespace(text_embeds=emb, pooled_embeds=pooled), guidance_embedding_scale=1.0)
            loss = torch.nn.functional.mse_loss(pred.float(), target.float(), reduction="none").mean([1, 2, 3]).mean()
            loss.backward()
        torch.nn.utils.clip_grad_norm_(a.params, 1.0)
        for tp, p in zip(twin_p, a.params):
            tp.grad = p.grad.detach().clone()
        a.optimizer.step()
        a.optimizer.zero_grad(set_to_none=True)
        a.ema.update()
        assert {k2: adopt.STATS[k2] - s0[k2] for k2 in s0} == {"adamw_fused": 1, "adamw_fallback": 0, "ema_fused": 1, "ema_fallback": 0}
        twin_opt.step()
        with torch.no_grad():  # toolkit/ema.py:126-139
            for s_, p_ in zip(twin_shadow, twin_p):
                tmp = s_ - p_
                tmp.mul_(1.0 - 0.99)
                s_.sub_(tmp)
        flat = lambda ts_: torch.cat([t.detach().reshape(-1) for t in ts_])  # noqa: E731
        assert _rel(flat(a.params), flat(twin_p)) < 2e-6 * (k + 1), (k, _rel(flat(a.params), flat(twin_p)))
        assert _rel(flat(a.ema.shadow_params), flat(twin_shadow)) < 2e-6 * (k + 1)
        sa = a.optimizer.state_dict()["state"]
        assert {float(v["step"]) for v in sa.values()} == {float(k + 1)}
        assert _rel(flat([sa[i]["exp_avg"] for i in range(len(a.params))]), flat([twin_opt.state[p]["exp_avg"] for p in twin_p])) < 2e-6
        assert _rel(flat([sa[i]["exp_avg_sq"] for i in range(len(a.params))]), flat([twin_opt.state[p]["exp_avg_sq"] for p in twin_p])) < 2e-6
    ad = nat_a.network
    assert isinstance(ad, AdoptedNetwork) and ad.aliasing_intact()
    # moments and shadows live inside the arenas
    lo, hi = ad.arena_m.data_ptr(), ad.arena_m.data_ptr() + ad.arena_m.numel() * 4
    assert all(lo <= a.optimizer.state[p]["exp_avg"].data_ptr() < hi for p in a.params)
    lo, hi = ad.arena_ema.data_ptr(), ad.arena_ema.data_ptr() + ad.arena_ema.numel() * 4
    assert all(lo <= s.data_ptr() < hi for s in a.ema.shadow_params)


def test_arena_tail_equals_the_same_kernels_called_directly_bit_for_bit():
    """hook path = [torch clip on the arena views] -> aitk_adamw_ema_step(max_norm = 0) -> aitk_ema_update: a twin that calls those entries
    itself on clones of the arenas lands on the same bits (nothing else touches the parameters)."""
    from ai_toolkit_amd import ops
    from oracle.pairs import batch

    a, nat_a = _loop(True)
    os.environ["AITK_FUSE_TRAINER_STEP"] = "1"
    lat, emb, pooled, _, _ = batch(2, seed=70)
    a.hook_train_loop(lat, emb, pooled)  # adoption, state moved into the arenas
    ad = nat_a.network
    p, m, v, e = (t.clone() for t in (ad.arena_p, ad.arena_m, ad.arena_v, ad.arena_ema))
    # second step by hand up to the gradients, then both tails
    lat, emb, pooled, _, _ = batch(2, seed=71)
    a.optimizer.zero_grad()
    noisy, ts, target = a.process_batch(lat)
    from types import SimpleNamespace

    with a.network:
        pred = a.sd.get_noise_prediction(noisy, ts, SimpleNamespace(text_embeds=emb, pooled_embeds=pooled), guidance_embedding_scale=1.0)
        loss = torch.nn.functional.mse_loss(pred.float(), target.float(), reduction="none").mean([1, 2, 3]).mean()
        loss.backward()
    torch.nn.utils.clip_grad_norm_(a.params, 1.0)
    g = ad.arena_g.clone()
    a.optimizer.step()
    a.ema.update()
    ops.adamw_ema_step(p, g, m, v, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=0.01, step=2, max_norm=0.0, ema=None)
    ops.ema_update(p, e, decay=0.99)
    assert torch.equal(p, ad.arena_p) and torch.equal(m, ad.arena_m) and torch.equal(v, ad.arena_v) and torch.equal(e, ad.arena_ema)


def test_harness_network_is_adopted_exactly_like_the_oracles_restatement_of_the_reference_network():
    """tools/trainer_harness.TrainerLoRANetwork (bench / profiling stand-in) against oracle/lora_ref.RefLoRANetwork (pinned to the reference's
    LoRASpecialNetwork by tests/golden/make_golden.py): same discovery order and names, and — same weights in — the same prediction and
    gradients bit for bit through the adopted HIP graph."""
    from types import SimpleNamespace

    from ai_toolkit_amd.plugin import Flux1MI355Model
    from oracle import lora_ref
    from oracle.pairs import batch, build

    os.environ["AITK_FUSE_TRAINER_STEP"] = "0"
    a, nat_a = _loop(False)
    _, _, nat_r, _ = build(attach=False)
    sd_r = Flux1MI355Model("cuda", model=nat_r, dtype=bf)
    net_r = lora_ref.RefLoRANetwork(sd_r.get_model_to_train(), 16, 1.0, block_names=("transformer_blocks", "single_transformer_blocks"))
    assert [m.lora_name for m in net_r.unet_loras] == [m.lora_name for m in a.network.unet_loras]
    with torch.no_grad():
        for x, y in zip(net_r.unet_loras, a.network.unet_loras):
            x.lora_down.weight.copy_(y.lora_down.weight.cpu())
            x.lora_up.weight.copy_(y.lora_up.weight.cpu())
    net_r.force_to(torch.device("cuda"), torch.float32)
    sd_r.network = net_r
    net_r._update_torch_multiplier()
    net_r.apply_to(None, sd_r.unet, False, True)
    net_r.prepare_grad_etc(None, sd_r.unet)
    lat, emb, pooled, noise, ts = batch(2, seed=80)
    pe = SimpleNamespace(text_embeds=emb, pooled_embeds=pooled)
    outs = []
    for sd, net in ((a.sd, a.network), (sd_r, net_r)):
        with net:
            pred = sd.get_noise_prediction(lat, ts, pe, guidance_embedding_scale=1.0)
            loss = torch.nn.functional.mse_loss(pred.float(), noise.float())
            loss.backward()
        outs.append((pred.detach().clone(), sd.get_model_to_train().network.arena_g.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
