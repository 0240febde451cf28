"""train.loss_type mae / pseudo_huber (extensions_built_in/sd_trainer/SDTrainer.py:903-916) and the EMA options use_feedback /
param_multiplier / use_num_updates (toolkit/ema.py:116-152): the oracle kernel-table entries against torch autograd of the reference's
formulas / the reference's own EMA class (tests/golden/ema_options.safetensors, made by executing toolkit/ema.py), and the trainer
plumbing on the host graph."""
import os

import pytest
import torch
from safetensors.torch import load_file

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd.trainer import FluxLoRATrainStep
from oracle import ref_ops
from tests.test_host_graph_cpu import CFG, build_pair

G = os.path.join(os.path.dirname(__file__), "golden")
EMA_CASES = {"fb": dict(ema_feedback=10.0, param_multiplier=0.999), "pm": dict(param_multiplier=1.002)}


def reference_loss(pred, target, loss_type, weight=None, mask=None):
    """SDTrainer.calculate_loss: elementwise loss (903-916) [* mask_multiplier 959] -> mean over all but batch (987-990) -> * per-sample
    multiplier (994) -> mean over batch (1013)."""
    if loss_type == "pseudo_huber":
        diff = pred.float() - target.float()
        c = 0.01
        loss = torch.sqrt(diff.pow(2) + c ** 2) - c
    elif loss_type == "mae":
        loss = torch.nn.functional.l1_loss(pred.float(), target.float(), reduction="none")
    else:
        loss = torch.nn.functional.mse_loss(pred.float(), target.float(), reduction="none")
    if mask is not None:
        loss = loss * mask
    loss = loss.mean(list(range(1, loss.dim())))
    if weight is not None:
        loss = loss * weight
    return loss.mean()


@pytest.mark.parametrize("loss_type", ["mse", "mae", "pseudo_huber"])
def test_oracle_loss_types_match_autograd_of_the_reference_formulas(loss_type):
    g = torch.Generator().manual_seed(3)
    B, T, Fd = 3, 10, 64
    pred = (torch.randn(B, T, Fd, generator=g) * 0.3).requires_grad_(True)
    target = torch.randn(B, T, Fd, generator=g) * 0.3
    with torch.no_grad():
        target[0, 0, :8] = pred[0, 0, :8]  # exact zeros of the difference: sign(0) = 0 for mae
    weight = torch.tensor([1.0, 0.5, 2.0])
    m4 = torch.rand(B, T, 4, generator=g)
    mask_full = m4.reshape(B, T, 1, 4).expand(B, T, Fd // 4, 4).reshape(B, T, Fd)
    want = reference_loss(pred, target, loss_type, weight, mask_full)
    want.backward()
    dpred, lps, loss = torch.empty(B, T, Fd), torch.zeros(B), torch.zeros(1)
    ref_ops.mse_loss_grad(pred.detach(), target, dpred, lps, loss, weight=weight, mask=m4, loss_type=loss_type)
    assert torch.allclose(loss[0], want.detach(), rtol=1e-6)
    assert torch.allclose(dpred, pred.grad, rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("tag", ["fb", "pm"])
def test_oracle_ema_options_match_the_reference_class(tag):
    t = load_file(os.path.join(G, "ema_options.safetensors"))
    p, ema = t[f"{tag}/p0"].clone(), t[f"{tag}/p0"].clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for k in range(3):
        ref_ops.adamw_ema_step(p, t[f"{tag}/grads"][k].clone(), m, v, lr=3e-3, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=0.01, step=k + 1,
                               max_norm=1.0, ema=ema, ema_decay=0.9, **EMA_CASES[tag])
    assert torch.allclose(p, t[f"{tag}/p3"], rtol=1e-5, atol=1e-7), (p - t[f"{tag}/p3"]).abs().max()
    assert torch.allclose(ema, t[f"{tag}/ema3"], rtol=1e-5, atol=1e-7)


def _batch(seed=2):
    g = torch.Generator().manual_seed(seed)
    return dict(latents=torch.randn(2, 16, 8, 4, generator=g), prompt_embeds=torch.randn(2, 6, CFG["joint_attention_dim"], generator=g),
                pooled_embeds=torch.randn(2, CFG["pooled_projection_dim"], generator=g), noise=torch.randn(2, 16, 8, 4, generator=g),
                timesteps=torch.tensor([700.0, 250.0]))


def test_trainer_use_num_updates_and_feedback_follow_the_reference_class():
    """The step object drives the same three updates as the reference class on the flat arena: decay warm-up min(d, (1+n)/(10+n))
    (golden 'nu') and the feedback / multiplier pair (golden 'fb')."""
    t = load_file(os.path.join(G, "ema_options.safetensors"))
    for tag, kw in (("nu", dict(ema_use_num_updates=True)), ("fb", dict(ema_use_feedback=True, ema_param_multiplier=0.999))):
        ref, ref_net, nat, net = build_pair(rank=4)
        st = FluxLoRATrainStep(nat, net, ref_ops, lr=3e-3, weight_decay=0.01, max_grad_norm=1.0, ema_decay=0.9, **kw)
        n = 4096
        assert net.arena_p.numel() >= n
        # drive the optimizer tail directly on a slice-shaped stand-in: same arithmetic, golden-sized vectors
        class _N:
            pass
        fake = _N()
        fake.arena_p, fake.arena_g = t[f"{tag}/p0"].clone(), None
        fake.arena_m, fake.arena_v, fake.arena_ema = torch.zeros(n), torch.zeros(n), t[f"{tag}/p0"].clone()
        fake.refresh_shadows = lambda ops: None
        st.network = fake
        for k in range(3):
            fake.arena_g = t[f"{tag}/grads"][k].clone()
            st._optimizer_step()
        assert torch.allclose(fake.arena_p, t[f"{tag}/p3"], rtol=1e-5, atol=1e-7), tag
        assert torch.allclose(fake.arena_ema, t[f"{tag}/ema3"], rtol=1e-5, atol=1e-7), tag


@pytest.mark.parametrize("loss_type", ["mae", "pseudo_huber"])
def test_train_step_with_loss_type_matches_autograd_oracle(loss_type):
    from oracle import flux_ref

    ref, ref_net, nat, net = build_pair(rank=4)
    b = _batch()
    st = FluxLoRATrainStep(nat, net, ref_ops, lr=0.0, weight_decay=0.0, max_grad_norm=0.0, loss_type=loss_type)
    loss = st.step(**b).item()
    # oracle: same noisy latents / target, autograd through the oracle model, the reference's loss formula
    t01 = (b["timesteps"] / 1000).view(-1, 1, 1, 1)
    noisy = (1 - t01) * b["latents"] + t01 * b["noise"]
    target = b["noise"] - b["latents"]
    img_ids, txt_ids = flux_ref.make_ids(8, 4, 6)
    for m in ref_net.unet_loras:
        for p in m.parameters():
            p.grad = None
    with ref_net:
        pred = flux_ref.unpack_latents(ref(flux_ref.pack_latents(noisy), b["prompt_embeds"], b["pooled_embeds"], b["timesteps"] / 1000,
                                           img_ids, txt_ids, torch.ones(2)), 8, 4)
        want = reference_loss(pred, target, loss_type)
        want.backward()
    assert abs(loss - want.item()) <= 1e-4 * abs(want.item())
    for a, r in zip(net.unet_loras, ref_net.unet_loras):
        assert torch.allclose(a.lora_up.weight.grad, r.lora_up.weight.grad, rtol=2e-3, atol=1e-7), a.lora_name
        assert torch.allclose(a.lora_down.weight.grad, r.lora_down.weight.grad, rtol=2e-3, atol=1e-7), a.lora_name


def test_unknown_loss_type_is_refused():
    ref, ref_net, nat, net = build_pair(rank=4)
    with pytest.raises(ValueError):
        FluxLoRATrainStep(nat, net, ref_ops, loss_type="wavelet")
