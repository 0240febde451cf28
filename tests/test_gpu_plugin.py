"""The drop-in surface itself on the GPU: the reference's trainer would drive the fused graphs exactly like this
(extensions_built_in/diffusion_models/flux_kontext/flux_kontext.py:243-352, extensions_built_in/sd_trainer/SDTrainer.py:2226-2293):

    optimizer.zero_grad()                                     # 2249
    with network:
        pred = sd.get_noise_prediction(...) / sd.predict_noise(...)
        loss = mse_loss(pred.float(), target.float()).mean(); loss.backward()          # 916-1013, 2238
    torch.nn.utils.clip_grad_norm_(params, max_grad_norm)     # 2278-2283
    optimizer.step(); optimizer.zero_grad(set_to_none=True)   # 2285-2288  (torch.optim.AdamW(eps=1e-6), toolkit/optimizer.py:78-79)
    ema.update()                                              # 2291-2293  (toolkit/ema.py:126-139)

with the HIP kernels behind `pred` (forward) and behind `loss.backward()` (the autograd bridge -> explicit backward graph), torch's own
optimizer on the arena-view Parameters, and must land where the fused `*LoRATrainStep` lands on the same inputs."""
import math
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


@pytest.fixture(autouse=True)
def _torch_optimizer_tail(monkeypatch):
    """These tests compare an ADOPTED network with a FusedLoRANetwork twin bit for bit, both stepped by torch.optim.AdamW: the adopted side's
    optimizer must then run torch's own step too (by default it is served by aitk_adamw_ema_step — a second fp32 formulation of the same
    update, tests/test_gpu_trainer_path.py — and would differ from the twin in the last bit)."""
    monkeypatch.setenv("AITK_FUSE_TRAINER_STEP", "0")


def _rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


def _unpack(t, B, h, w):
    return t.reshape(B, h // 2, w // 2, 16, 2, 2).permute(0, 3, 1, 4, 2, 5).reshape(B, 16, h, w).contiguous()


def _torch_trainer_step(net, params, opt, ema, ops, predict, target, max_norm=1.0, decay=0.99):
    opt.zero_grad()
    with net:
        pred = predict()
        loss = torch.nn.functional.mse_loss(pred.float(), target.float(), reduction="none").mean([1, 2, 3]).mean()
        loss.backward()
    grad = net.arena_g.clone()
    torch.nn.utils.clip_grad_norm_(params, max_norm)
    opt.step()
    opt.zero_grad(set_to_none=True)
    with torch.no_grad():  # toolkit/ema.py:126-139
        for s, p in zip(ema, params):
            s.sub_((s - p) * (1.0 - decay))
    net.refresh_shadows(ops)  # the weights-changed hook: bf16 shadows follow the fp32 masters
    return loss.detach(), grad


def test_flux_plugin_trainer_loop_equals_fused_train_step():
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.plugin import Flux1MI355Model
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from tests.test_gpu_e2e import _batch, _build

    _, _, nat_a, net_a = _build()
    _, _, nat_b, net_b = _build()
    assert torch.equal(net_a.arena_p, net_b.arena_p)
    plug = Flux1MI355Model("cuda", model=nat_a, dtype=bf)
    params = net_a.prepare_optimizer_params(default_lr=1e-3)[0]["params"]
    assert all(p.is_cuda and p.dtype == torch.float32 for p in params)
    opt = torch.optim.AdamW(params, lr=1e-3, eps=1e-6, weight_decay=0.01)
    ema = [p.detach().clone() for p in params]
    fused = FluxLoRATrainStep(nat_b, net_b, ops, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0, ema_decay=0.99)
    for k in range(3):
        lat, emb, pooled, noise, ts = _batch(2, seed=20 + k)
        # what process_general_training_batch hands the trainer: noisy latents from the scheduler's add_noise, target = noise - latents
        # (custom_flowmatch_sampler.py:91-102: fp32 mix, then the latent dtype).  The mix is taken from aitk_flow_noise_pack — un-packed back
        # to [B,16,H,W] — so both paths see bit-identical inputs: an fp32-ulp difference of a torch mix flips a few bf16 roundings of the
        # noisy latents, which this random-weight model amplifies to 1e-4 of the loss, hiding what the test is about (the boundary)
        B_, _, h_, w_ = lat.shape
        npk = torch.empty(B_, (h_ // 2) * (w_ // 2), 64, dtype=bf, device="cuda")
        tpk = torch.empty_like(npk)
        ops.flow_noise_pack(lat, noise, ts.float().contiguous(), npk, tpk)
        noisy, target = _unpack(npk, B_, h_, w_), _unpack(tpk, B_, h_, w_)
        t01 = (ts / 1000).view(-1, 1, 1, 1)
        assert _rel(noisy, ((1 - t01) * lat.float() + t01 * noise.float())) < 3e-3  # = the reference's formula up to the bf16 rounding
        pe = SimpleNamespace(text_embeds=emb, pooled_embeds=pooled)
        loss_a, g_a = _torch_trainer_step(net_a, params, opt, ema, ops,
                                          lambda: plug.get_noise_prediction(noisy, ts, pe, guidance_embedding_scale=1.0), target)
        loss_b = fused.step(lat, emb, pooled, noise=noise, timesteps=ts)
        assert abs(loss_a.item() - loss_b.item()) <= (2e-5 if k == 0 else 1e-3) * abs(loss_b.item()), (k, loss_a.item(), loss_b.item())  # k > 0: the adapters have taken AdamW steps (lr * sign(g) on near-zero gradients differs)
        # identical kernels behind both paths: gradients agree to the rounding of the loss gradient (torch fp32 -> bf16 vs the mse kernel)
        # (later steps start from adapters that already differ by AdamW's lr * sign(g) on entries whose gradient is inside that rounding:
        # the trajectories stay close, not identical)
        assert _rel(g_a, net_b.arena_g) < (2e-3 if k == 0 else 3e-2), (k, _rel(g_a, net_b.arena_g))
        assert _rel(net_a.arena_p, net_b.arena_p) < (2e-3 if k == 0 else 1e-2), (k, _rel(net_a.arena_p, net_b.arena_p))
    ema_a = torch.cat([e.reshape(-1) for e in ema])
    # arena order = optimizer parameter order for plain LoRA up to the rank padding: compare module by module
    off = 0
    for m in net_b.unet_loras:
        for which, par in (("down", m.lora_down.weight), ("up", m.lora_up.weight)):
            n = par.numel()
            want = net_b.arena_view(net_b.arena_ema, m, which)
            got = ema_a[off:off + n].view_as(want)
            assert _rel(got, want) < 1e-2, (m.lora_name, which, _rel(got, want))
            off += n
    m0 = net_a.unet_loras[0]
    assert m0.lora_up.weight.grad is None  # set_to_none dropped the views; the next backward re-attaches them


@pytest.mark.parametrize("sdxl", [False, True], ids=["sd15", "sdxl"])
def test_stable_diffusion_wrapper_trainer_loop_equals_fused_train_step(sdxl):
    from ai_toolkit_amd.plugin import StableDiffusionMI355Model
    from ai_toolkit_amd.trainer import UNetLoRATrainStep
    from tests.test_gpu_unet import MID_SD15, MID_SDXL, _batch, _pair

    cfg, ref, ref_net, native, finish, ops = _pair(MID_SDXL if sdxl else MID_SD15, sdxl)
    nat_a, net_a = finish(*native(ops), ops)
    nat_b, net_b = finish(*native(ops), ops)
    sd = StableDiffusionMI355Model("cuda", model=nat_a, dtype=bf, is_xl=sdxl)
    params = net_a.prepare_optimizer_params(default_lr=1e-3)[0]["params"]
    opt = torch.optim.AdamW(params, lr=1e-3, eps=1e-6, weight_decay=0.01)
    ema = [p.detach().clone() for p in params]
    fused = UNetLoRATrainStep(nat_b, net_b, ops, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0, ema_decay=0.99)
    for k in range(2):
        lat, ctx, pooled, noise, ts = _batch(cfg, seed=40 + k)
        noisy = sd.add_noise(lat, noise, ts)  # toolkit/stable_diffusion_model.py:1854-1876 (latent dtype)
        target = sd.get_loss_target(noise=noise)
        pe = SimpleNamespace(text_embeds=ctx, pooled_embeds=pooled)
        loss_a, g_a = _torch_trainer_step(net_a, params, opt, ema, ops, lambda: sd.predict_noise(noisy, text_embeddings=pe, timestep=ts), target)
        loss_b = fused.step(lat, ctx, pooled if sdxl else None, noise=noise, timesteps=ts)
        assert math.isfinite(loss_a.item())
        # first step: same adapters, inputs equal up to the rounding of the noise mix (torch bf16 arithmetic vs aitk_ddpm_noise_nhwc)
        assert abs(loss_a.item() - loss_b.item()) <= (1e-3 if k == 0 else 5e-3) * abs(loss_b.item()), (k, loss_a.item(), loss_b.item())
        assert _rel(g_a, net_b.arena_g) < (1e-2 if k == 0 else 5e-2), (k, _rel(g_a, net_b.arena_g))
        assert _rel(net_a.arena_p, net_b.arena_p) < (5e-3 if k == 0 else 2e-2), (k, _rel(net_a.arena_p, net_b.arena_p))


def test_reference_built_network_adopted_on_the_hip_kernels_equals_the_fused_network_bit_for_bit():
    """The boundary without a trainer patch, on the GPU (CPU twin + the run with the reference's own classes: tests/test_adoption_cpu.py): the
    trainer's sequence — network constructed over sd.get_model_to_train(), force_to(device, fp32), `sd.network = network`, apply_to (forward
    swap), prepare_optimizer_params -> torch.optim.AdamW, `with network:` prediction, loss.backward(), clip_grad_norm_, step, zero_grad —
    with the oracle's restatement of the reference network protocol (oracle/lora_ref.py) over the native model.  The native graph adopts
    that network (ai_toolkit_amd/adopt.py): same arena layout, same kernels as a FusedLoRANetwork, so everything must be BIT-IDENTICAL, with the
    optimizer still holding the network's own Parameter objects."""
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.adopt import AdoptedNetwork
    from ai_toolkit_amd.plugin import Flux1MI355Model
    from oracle import lora_ref
    from oracle.pairs import batch as _batch, build as _build

    _, _, nat_f, net_f = _build()
    _, _, nat_a, none = _build(attach=False)
    assert none is None
    sd_f = Flux1MI355Model("cuda", model=nat_f, dtype=bf)
    sd_a = Flux1MI355Model("cuda", model=nat_a, dtype=bf)
    # --- the trainer's side (BaseSDTrainProcess.py:1949-2039)
    net_a = lora_ref.RefLoRANetwork(sd_a.get_model_to_train(), 16, 1.0)
    with torch.no_grad():
        for a, b in zip(net_a.unet_loras, net_f.unet_loras):
            assert a.lora_name == b.lora_name
            a.lora_down.weight.copy_(b.lora_down.weight.cpu())
            a.lora_up.weight.copy_(b.lora_up.weight.cpu())
    net_a.force_to(torch.device("cuda"), torch.float32)
    sd_a.network = net_a
    net_a._update_torch_multiplier()
    net_a.apply_to(None, sd_a.unet, False, True)
    net_a.prepare_grad_etc(None, sd_a.unet)
    groups_a = net_a.prepare_optimizer_params(text_encoder_lr=1e-3, unet_lr=1e-3, default_lr=1e-3)
    params_a = [p for g in groups_a for p in g["params"]]
    ids = [id(p) for p in params_a]
    opt_a = torch.optim.AdamW(groups_a, lr=1e-3, eps=1e-6, weight_decay=0.01)
    params_f = net_f.prepare_optimizer_params(default_lr=1e-3)[0]["params"]
    opt_f = torch.optim.AdamW(params_f, lr=1e-3, eps=1e-6, weight_decay=0.01)
    ema_a, ema_f = [p.detach().clone() for p in params_a], [p.detach().clone() for p in params_f]

    for k in range(3):
        lat, emb, pooled, noise, ts = _batch(2, seed=30 + k)
        B_, _, h_, w_ = lat.shape
        npk = torch.empty(B_, (h_ // 2) * (w_ // 2), 64, dtype=bf, device="cuda")
        tpk = torch.empty_like(npk)
        ops.flow_noise_pack(lat, noise, ts.float().contiguous(), npk, tpk)
        noisy, target = _unpack(npk, B_, h_, w_), _unpack(tpk, B_, h_, w_)
        pe = SimpleNamespace(text_embeds=emb, pooled_embeds=pooled)
        opt_a.zero_grad()
        with net_a:
            pred = sd_a.get_noise_prediction(noisy, ts, pe, guidance_embedding_scale=1.0)
            loss_a = torch.nn.functional.mse_loss(pred.float(), target.float(), reduction="none").mean([1, 2, 3]).mean()
            loss_a.backward()
        ad = nat_a.network
        assert isinstance(ad, AdoptedNetwork) and ad.foreign is net_a and ad.aliasing_intact()
        g_a = ad.arena_g.clone()
        torch.nn.utils.clip_grad_norm_(params_a, 1.0)
        opt_a.step()
        opt_a.zero_grad(set_to_none=True)
        with torch.no_grad():
            for s, p in zip(ema_a, params_a):
                s.sub_((s - p) * (1.0 - 0.99))
        loss_f, g_f = _torch_trainer_step(net_f, params_f, opt_f, ema_f, ops, lambda: sd_f.get_noise_prediction(noisy, ts, pe, guidance_embedding_scale=1.0), target)
        assert torch.equal(loss_a.detach(), loss_f), (k, loss_a.item(), loss_f.item())
        assert torch.equal(g_a, g_f), (k, _rel(g_a, g_f))
        assert torch.equal(ad.arena_p, net_f.arena_p), (k, _rel(ad.arena_p, net_f.arena_p))
    assert [id(p) for g in opt_a.param_groups for p in g["params"]] == ids
    assert all(torch.equal(a, b) for a, b in zip(ema_a, ema_f))
    # what the reference-side object saves is the arena content
    sd_ref = net_a.peft_state_dict(dtype=torch.float32)
    sd_fus = net_f.get_state_dict(dtype=torch.float32)
    assert list(sd_ref) == list(sd_fus) and all(torch.equal(sd_ref[k2], sd_fus[k2]) for k2 in sd_ref)
    # inactive network == base model, bit for bit with the fused network's inactive pass
    with torch.no_grad():
        assert torch.equal(sd_a.get_noise_prediction(noisy, ts, pe, 1.0), sd_f.get_noise_prediction(noisy, ts, pe, 1.0))


@pytest.mark.parametrize("sdxl", [False, True], ids=["sd15", "sdxl"])
def test_reference_built_conv_network_adopted_on_the_unet_hip_kernels_bit_for_bit(sdxl):
    """The UNet leg of the adoption on the GPU (CPU twin + the reference's own LoRASpecialNetwork run: tests/test_adoption_cpu.py): a network the
    "trainer" built by class-name discovery over the native UNet — Linear, 1x1-conv and (network.conv) 3x3-conv adapters with diffusers-shaped
    4-D Conv2d weights, kohya names — is adopted: its Parameters are re-pointed at [rank_pad, in*k*k] / [out, rank_pad] arena blocks and the HIP
    kernels (implicit-GEMM conv + conv-adapter epilogues) run it.  Bit-identical to a FusedLoRANetwork over the same model."""
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.adopt import AdoptedNetwork
    from ai_toolkit_amd.lora import FusedLoRANetwork
    from ai_toolkit_amd.plugin import StableDiffusionMI355Model
    from ai_toolkit_amd.unet import UNet2DConditionModel
    from oracle import lora_ref, unet_ref
    from tests.test_gpu_unet import MID_SD15, MID_SDXL, _batch

    cfg = dict(unet_ref.SDXL if sdxl else unet_ref.SD15, **(MID_SDXL if sdxl else MID_SD15))
    torch.manual_seed(0)
    ref = unet_ref.UNet2DConditionModel(**cfg)
    unet_ref.init_synthetic_(ref, seed=11)
    state = {k: v.to(bf) for k, v in ref.state_dict().items()}

    def native():
        nat = UNet2DConditionModel(**cfg, dtype=bf, device="cuda", ops=ops)
        nat.load_state_dict(state, strict=True)
        nat.prepare()
        return nat, StableDiffusionMI355Model("cuda", model=nat, dtype=bf, is_xl=sdxl)

    nat_a, sd_a = native()
    nat_f, sd_f = native()
    torch.manual_seed(99)
    net_a = lora_ref.RefLoRANetwork(sd_a.get_model_to_train(), 8, 1.0, target=tuple(sd_a.target_lora_modules), kohya_unet=True, alpha=4.0,
                                    conv_lora_dim=4, conv_alpha=2.0)
    torch.manual_seed(99)
    net_f = FusedLoRANetwork(nat_f, lora_dim=8, alpha=4.0, conv_lora_dim=4, conv_alpha=2.0, target_lin_modules=("Transformer2DModel",),
                             is_transformer=False, peft_format=False, transformer_only=False, base_model_version="sdxl" if sdxl else "sd1")
    assert [m.lora_name for m in net_a.unet_loras] == [m.lora_name for m in net_f.unet_loras]
    assert any(m.lora_down.weight.dim() == 4 and m.lora_down.weight.shape[2:] == (3, 3) for m in net_a.unet_loras)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for a, b in zip(net_a.unet_loras, net_f.unet_loras):
            b.lora_down.weight.copy_(a.lora_down.weight.reshape(b.lora_down.weight.shape))
            up = torch.randn(b.lora_up.weight.shape, generator=g) * 0.03
            a.lora_up.weight.copy_(up.reshape(a.lora_up.weight.shape))
            b.lora_up.weight.copy_(up)
    # --- the trainer's side
    net_a.force_to(torch.device("cuda"), torch.float32)
    sd_a.network = net_a
    net_a._update_torch_multiplier()
    net_a.apply_to(None, sd_a.unet, False, True)
    pa = [p for grp in net_a.prepare_optimizer_params(None, 1e-3, 1e-3) for p in grp["params"]]
    # --- the fused network
    net_f.apply_to()
    net_f.build_arena("cuda", groups=nat_f.lora_groups())
    net_f.refresh_shadows(ops)
    nat_f.attach_network(net_f)
    pf = net_f.prepare_optimizer_params(default_lr=1e-3)[0]["params"]
    oa = torch.optim.AdamW(pa, lr=1e-3, eps=1e-6, weight_decay=0.01)
    of = torch.optim.AdamW(pf, lr=1e-3, eps=1e-6, weight_decay=0.01)
    for k in range(2):
        lat, ctx, pooled, noise, ts = _batch(cfg, seed=50 + k)
        noisy = sd_f.add_noise(lat, noise, ts)
        pe = SimpleNamespace(text_embeds=ctx, pooled_embeds=pooled)
        out = []
        for net, sd, opt, plist in ((net_a, sd_a, oa, pa), (net_f, sd_f, of, pf)):
            opt.zero_grad()
            with net:
                pred = sd.predict_noise(noisy, text_embeddings=pe, timestep=ts)
                loss = torch.nn.functional.mse_loss(pred.float(), noise.float(), reduction="none").mean([1, 2, 3]).mean()
                loss.backward()
            grads = (nat_a.network if net is net_a else net_f).arena_g.clone()
            torch.nn.utils.clip_grad_norm_(plist, 1.0)
            opt.step()
            opt.zero_grad(set_to_none=True)
            out.append((loss.detach(), grads))
        net_f.refresh_shadows(ops)
        assert math.isfinite(out[0][0].item()) and torch.equal(out[0][0], out[1][0]), (k, out[0][0].item(), out[1][0].item())
        assert torch.equal(out[0][1], out[1][1]), (k, _rel(out[0][1], out[1][1]))
        assert out[0][1].abs().sum().item() > 0
    ad = nat_a.network
    assert isinstance(ad, AdoptedNetwork) and ad.foreign is net_a and ad.aliasing_intact() and torch.equal(ad.arena_p, net_f.arena_p)
    for a, b in zip(pa, pf):
        assert a.is_cuda and torch.equal(a.detach().reshape(b.shape), b.detach())
    # what the reference-side object would save: Conv2d-shaped tensors with the arena's values, the fused network's kohya keys
    sa, sf = net_a.state_dict(), net_f.get_state_dict(dtype=torch.float32)
    for key, v in sf.items():
        assert torch.equal(sa[key].detach().reshape(v.shape).float().cpu(), v.cpu()), key


def test_reference_built_network_adopted_on_the_wan_hip_kernels_bit_for_bit():
    """The Wan2.1 leg of the adoption on the GPU (BASELINE config 4; CPU twin + the reference's own LoRASpecialNetwork run:
    tests/test_adoption_cpu.py): a network built the trainer's way over the native video DiT (block filter `blocks`, 5-D latents, UMT5 text
    states, no pooled vector) is adopted and must be bit-identical to a FusedLoRANetwork over the same model on the HIP kernels."""
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.adopt import AdoptedNetwork
    from ai_toolkit_amd.lora import FusedLoRANetwork
    from ai_toolkit_amd.plugin import Wan21MI355Model
    from ai_toolkit_amd.wan import WanTransformer3DModel
    from oracle import lora_ref, wan_ref
    from tests.test_gpu_wan import CFG as WCFG

    torch.manual_seed(0)
    ref = wan_ref.WanTransformer3DModel(**WCFG)
    wan_ref.init_synthetic_(ref, seed=99, std=0.03)
    state = {k: v.to(bf) for k, v in ref.state_dict().items()}

    def native():
        nat = WanTransformer3DModel(**WCFG, dtype=bf, device="cuda", ops=ops)
        nat.load_state_dict(state, strict=True)
        nat.prepare()
        return nat, Wan21MI355Model("cuda", model=nat, dtype=bf)

    nat_a, sd_a = native()
    nat_f, sd_f = native()
    net_a = lora_ref.RefLoRANetwork(sd_a.get_model_to_train(), 16, 1.0, target=tuple(sd_a.target_lora_modules), block_names=tuple(sd_a.get_transformer_block_names()))
    net_f = FusedLoRANetwork(nat_f, lora_dim=16, target_lin_modules=tuple(sd_f.target_lora_modules), transformer_block_names=sd_f.get_transformer_block_names(),
                             base_model_version="wan_2.1", base_model=sd_f)
    assert [m.lora_name for m in net_a.unet_loras] == [m.lora_name for m in net_f.unet_loras] and len(net_f.unet_loras) == 10 * WCFG["num_layers"]
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for a, b in zip(net_a.unet_loras, net_f.unet_loras):
            b.lora_down.weight.copy_(a.lora_down.weight)
            up = torch.randn(b.lora_up.weight.shape, generator=g) * 0.02
            a.lora_up.weight.copy_(up)
            b.lora_up.weight.copy_(up)
    net_a.force_to(torch.device("cuda"), torch.float32)
    sd_a.network = net_a
    net_a._update_torch_multiplier()
    net_a.apply_to(None, sd_a.unet, False, True)
    pa = [p for grp in net_a.prepare_optimizer_params(None, 1e-3, 1e-3) for p in grp["params"]]
    net_f.apply_to()
    net_f.build_arena("cuda", groups=nat_f.lora_groups())
    net_f.refresh_shadows(ops)
    nat_f.attach_network(net_f)
    pf = net_f.prepare_optimizer_params(default_lr=1e-3)[0]["params"]
    oa = torch.optim.AdamW(pa, lr=1e-3, eps=1e-6, weight_decay=0.01)
    of = torch.optim.AdamW(pf, lr=1e-3, eps=1e-6, weight_decay=0.01)
    gen = torch.Generator().manual_seed(3)
    for k in range(2):
        lat = torch.randn(2, 16, 3, 16, 8, generator=gen).to(bf).cuda()
        tgt = torch.randn(2, 16, 3, 16, 8, generator=gen).to(bf).cuda()
        pe = SimpleNamespace(text_embeds=(torch.randn(2, 24, WCFG["text_dim"], generator=gen) * 0.5).to(bf).cuda(), pooled_embeds=None)
        ts = torch.tensor([310.0, 845.0], device="cuda")
        out = []
        for net, sd, opt, plist in ((net_a, sd_a, oa, pa), (net_f, sd_f, of, pf)):
            opt.zero_grad()
            with net:
                pred = sd.get_noise_prediction(lat, ts, pe)
                loss = torch.nn.functional.mse_loss(pred.float(), tgt.float(), reduction="none").mean([1, 2, 3, 4]).mean()
                loss.backward()
            grads = (nat_a.network if net is net_a else net_f).arena_g.clone()
            torch.nn.utils.clip_grad_norm_(plist, 1.0)
            opt.step()
            opt.zero_grad(set_to_none=True)
            out.append((loss.detach(), grads))
        net_f.refresh_shadows(ops)
        assert math.isfinite(out[0][0].item()) and torch.equal(out[0][0], out[1][0]), (k, out[0][0].item(), out[1][0].item())
        assert torch.equal(out[0][1], out[1][1]) and out[0][1].abs().sum().item() > 0, (k, _rel(out[0][1], out[1][1]))
    ad = nat_a.network
    assert isinstance(ad, AdoptedNetwork) and ad.foreign is net_a and ad.aliasing_intact() and torch.equal(ad.arena_p, net_f.arena_p)
    # the file the reference-side object would write, through the plug-in's key converter (diffusion_model.* names): the fused network's
    sa = sd_a.convert_lora_weights_before_save(net_a.peft_state_dict(dtype=torch.float32))
    sf = net_f.get_state_dict(dtype=torch.float32)
    assert sorted(sa) == sorted(sf) and next(iter(sf)).startswith("diffusion_model.blocks.0.")
    assert all(torch.equal(sa[key].cpu(), sf[key].cpu()) for key in sf)


@pytest.mark.parametrize("network_type", ["lora", "dora"])
def test_preservation_step_two_grad_predictions_one_backward_on_the_hip_kernels(network_type):
    """blank_prompt_preservation / diff_output_preservation on the HIP kernels (SDTrainer.py:1983-2016, 2182-2219): prior prediction with the network off
    under no_grad, the training prediction and a second grad-enabled prediction, ONE loss.backward() through two native graphs.  The arena must end
    with the sum of what the two predictions give when each is back-propagated alone (same kernels, same inputs), and the prior prediction must be the
    base model's (tests/test_plugin_cpu.py pins the same sequence to autograd over the oracle network; tests/test_trainer_loop_cpu.py to the
    reference's real trainer)."""
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.plugin import Flux1MI355Model
    from tests.test_gpu_e2e import _batch, _build

    if network_type == "lora":
        _, _, nat, net = _build()
    else:  # DoRA: the forward parks its linear outputs on the adapters (d magnitude needs them) — they must travel with each graph too
        from ai_toolkit_amd.flux import FluxTransformer2DModel
        from ai_toolkit_amd.lora import FusedLoRANetwork
        from tests.test_gpu_e2e import CFG

        _, _, base, _ = _build(attach=False)
        nat = FluxTransformer2DModel(**CFG, dtype=bf, device="cuda", ops=ops)
        nat.load_state_dict(base.state_dict(), strict=True)
        torch.manual_seed(5)
        net = FusedLoRANetwork(nat, lora_dim=16, network_type="dora")
        g_ = torch.Generator().manual_seed(7)
        with torch.no_grad():
            for a in net.unet_loras:
                a.lora_up.weight.copy_(torch.randn(a.lora_up.weight.shape, generator=g_) * 0.02)
                a.magnitude.copy_(a.magnitude.cpu() * (1 + 0.03 * torch.randn(a.magnitude.shape, generator=g_)))
        net.apply_to()
        net.build_arena("cuda", groups=nat.lora_groups())
        net.refresh_shadows(ops)
        nat.attach_network(net)
        nat.prepare()
    plug = Flux1MI355Model("cuda", model=nat, dtype=bf)
    lat, emb, pooled, noise, ts = _batch(2, seed=31)
    _, emb2, pooled2, _, _ = _batch(2, seed=32)
    pe, pe_blank = SimpleNamespace(text_embeds=emb, pooled_embeds=pooled), SimpleNamespace(text_embeds=emb2, pooled_embeds=pooled2)
    tt = (ts / 1000).view(-1, 1, 1, 1)
    noisy = ((1 - tt) * lat.float() + tt * noise.float()).to(bf)
    target = (noise.float() - lat.float())
    mult = 0.5

    def loss_a(p):
        return torch.nn.functional.mse_loss(p.float(), target, reduction="none").mean([1, 2, 3]).mean()

    with net:
        net.is_active = False
        with torch.no_grad():
            prior = plug.get_noise_prediction(noisy, ts, pe_blank, 1.0, False)
        net.is_active = True
        with torch.no_grad():
            active_blank = plug.get_noise_prediction(noisy, ts, pe_blank, 1.0, False)
        assert _rel(active_blank, prior) > 1e-3  # the adapter is not a no-op on this network: "network off" changed the prediction

        def loss_b(p):
            return torch.nn.functional.mse_loss(p, prior) * mult

        # each prediction alone
        net.zero_grad_arena()
        loss_a(plug.get_noise_prediction(noisy, ts, pe, 1.0, False)).backward()
        g_a = net.arena_g.clone()
        net.zero_grad_arena()
        loss_b(plug.get_noise_prediction(noisy, ts, pe_blank, 1.0, False)).backward()
        g_b = net.arena_g.clone()
        assert g_a.abs().max() > 0 and g_b.abs().max() > 0
        # the trainer's sequence
        net.zero_grad_arena()
        pred = plug.get_noise_prediction(noisy, ts, pe, 1.0, False)
        pres = plug.get_noise_prediction(noisy, ts, pe_blank, 1.0, False)
        assert nat.ctx is None
        (loss_a(pred) + loss_b(pres)).backward()
        g = net.arena_g.clone()
    assert torch.isfinite(g).all()
    assert _rel(g, g_a + g_b) < 1e-6, _rel(g, g_a + g_b)
