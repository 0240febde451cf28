"""The drop-in boundary without a trainer patch: the reference's trainer builds its OWN `LoRASpecialNetwork` over the
plug-in's native model (jobs/process/BaseSDTrainProcess.py:1932-1993) and the native graph adopts it (ai_toolkit_amd/adopt.py).

  * tests/golden/adoption_flux_tiny.safetensors was produced by EXECUTING the reference's own classes (LoRASpecialNetwork with LoRAModule /
    DoRAModule / LokrModule, toolkit/ema.py, the trainer's literal construction keywords) over the real plug-in class of
    integration/extensions/aitk_mi355 — `tests/golden/make_golden.py golden_adoption`.  Here a FusedLoRANetwork twin walks the same sequence
    and must land on the same numbers BIT FOR BIT: losses, last gradients, parameters, EMA shadows, saved state dict, model hash.
  * the same adoption driven by the oracle's restatement of the reference protocol (oracle/lora_ref.py: forward swap in apply_to, `with
    network:`, multiplier vectors) — runs anywhere, covers the behaviours around it (inactive network, storage replaced behind our back,
    load_state_dict, gradient accumulation, what is refused and where).
  * when /root/reference is present the fixture is regenerated in a scratch directory and compared with the committed one (freshness).
"""
import json
import os
import subprocess
import sys
from types import SimpleNamespace

import pytest
import torch
from safetensors import safe_open
from safetensors.torch import load_file

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd.adopt import AdoptedNetwork, AdoptionError
from ai_toolkit_amd.flux import FluxTransformer2DModel
from ai_toolkit_amd.lora import FusedLoRANetwork
from ai_toolkit_amd.plugin import Flux1MI355Model
from oracle import flux_ref, lora_ref, ref_ops

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "adoption_flux_tiny.safetensors")
CFG = dict(in_channels=64, num_layers=2, num_single_layers=2, attention_head_dim=128, num_attention_heads=2,
           joint_attention_dim=64, pooled_projection_dim=32)


def batches(n, seed=9, B=2, Hl=8, Wl=4, n_txt=6):  # twin of make_golden.adoption_batches
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        lat = torch.randn(B, 16, Hl, Wl, generator=g)
        emb = torch.randn(B, n_txt, CFG["joint_attention_dim"], generator=g) * 0.5
        pooled = torch.randn(B, CFG["pooled_projection_dim"], generator=g) * 0.5
        target = torch.randn(B, 16, Hl, Wl, generator=g)
        out.append((lat, emb, pooled, torch.tensor([700.0, 250.0]), target))
    return out


def native_plugin():
    torch.manual_seed(0)
    ref = flux_ref.FluxTransformer2DModel(**CFG)
    flux_ref.init_synthetic_(ref, seed=1234, std=0.05)
    nat = FluxTransformer2DModel(**CFG, dtype=torch.float32, device="cpu", ops=ref_ops)
    nat.load_state_dict(ref.state_dict(), strict=True)
    nat.prepare()
    return ref, nat, Flux1MI355Model("cpu", model=nat, dtype=torch.float32)


def trainer_step(net, plist, opt, ema, sd, batch, after_step=None, decay=0.99):
    """extensions_built_in/sd_trainer/SDTrainer.py:2243-2293 as the golden generator walks it; EMA = toolkit/ema.py:116-139"""
    lat, emb, pooled, ts, target = batch
    pe = SimpleNamespace(text_embeds=emb, pooled_embeds=pooled)
    opt.zero_grad()
    with net:
        pred = sd.get_noise_prediction(lat, ts, pe, guidance_embedding_scale=1.0, bypass_guidance_embedding=False)
        loss = torch.nn.functional.mse_loss(pred.float(), target.float(), reduction="none").mean([1, 2, 3]).mean()
        loss.backward()
    grads = [p.grad.detach().clone() for p in plist]
    torch.nn.utils.clip_grad_norm_(plist, 1.0)
    opt.step()
    opt.zero_grad(set_to_none=True)
    with torch.no_grad():
        for s, p in zip(ema, plist):
            tmp = (s - p)
            tmp.mul_(1.0 - decay)
            s.sub_(tmp)
    if after_step is not None:
        after_step()
    return loss.detach(), grads


TWINS = {
    "lora": dict(kw=dict(lora_dim=8, alpha=8), steps=3),
    "lora_mvec": dict(kw=dict(lora_dim=4, alpha=4), steps=2, multiplier=[0.5, -1.5], warm=True),
    "dora": dict(kw=dict(lora_dim=4, alpha=4, network_type="dora"), steps=2, warm=True),
    "lokr": dict(kw=dict(lora_dim=9999999999, alpha=9999999999, network_type="lokr", lokr_factor=-1), steps=2, warm=True),
    "lokr_lowrank": dict(kw=dict(lora_dim=4, alpha=4, network_type="lokr", lokr_factor=-1), steps=2, warm=True),
}


def _warm(net):
    """the golden run's warm start: state_dict order of the reference network, randn for every lora_up / lokr_w2* tensor"""
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for m in net.unet_loras:
            if m.magnitude is not None:      # DoRAModule.state_dict(): magnitude, lora_up.weight, lora_down.weight
                m.lora_up.weight.copy_(torch.randn(m.lora_up.weight.shape, generator=g) * 0.05)
            elif m.is_lokr:                  # LokrModule: lokr_w1, lokr_w2 | lokr_w2_a, lokr_w2_b
                for key, par in m.factor_params():
                    if key.startswith("lokr_w2"):
                        par.copy_(torch.randn(par.shape, generator=g) * 0.05)
            else:                            # LoRAModule: lora_down.weight, lora_up.weight
                m.lora_up.weight.copy_(torch.randn(m.lora_up.weight.shape, generator=g) * 0.05)


@pytest.mark.parametrize("tag", list(TWINS))
def test_fused_twin_lands_bit_for_bit_where_the_reference_network_landed_over_the_adopting_plugin(tag):
    spec = TWINS[tag]
    gold = load_file(GOLD)
    with safe_open(GOLD, "pt") as fh:
        meta = json.loads(fh.metadata()["meta"])[tag]
    _, nat, sd = native_plugin()
    torch.manual_seed(99)
    net = FusedLoRANetwork(nat, multiplier=1.0, transformer_block_names=sd.get_transformer_block_names(), base_model=sd, **spec["kw"])
    assert [m.lora_name for m in net.unet_loras] == meta["names"]
    net.apply_to()
    net.build_arena("cpu", groups=nat.lora_groups())
    if spec.get("warm"):
        _warm(net)
    net.refresh_shadows(ref_ops)
    nat.attach_network(net)
    params = net.prepare_optimizer_params(default_lr=1e-3)
    plist = [p for g in params for p in g["params"]]
    assert len(plist) == meta["n_params"]
    opt = torch.optim.AdamW(params, lr=1e-3, eps=1e-6, weight_decay=0.01)
    ema = [p.detach().clone() for p in plist]
    if spec.get("multiplier") is not None:
        net.multiplier = spec["multiplier"]
    losses = []
    for b in batches(spec["steps"]):
        loss, grads = trainer_step(net, plist, opt, ema, sd, b, after_step=lambda: net.refresh_shadows(ref_ops))
        losses.append(loss)
    assert torch.equal(torch.stack(losses), gold[f"{tag}/losses"]), (torch.stack(losses), gold[f"{tag}/losses"])
    for i, p in enumerate(plist):
        assert torch.equal(grads[i], gold[f"{tag}/last_grad/{i}"]), (tag, "grad", i)
        assert torch.equal(p.detach(), gold[f"{tag}/param/{i}"]), (tag, "param", i)
        assert torch.equal(ema[i], gold[f"{tag}/ema/{i}"]), (tag, "ema", i)
    sdict = net.get_state_dict(dtype=torch.float32)
    assert list(sdict) == meta["saved_keys"]
    for k, v in sdict.items():
        assert torch.equal(v, gold[f"{tag}/saved/{k}"]), k
    # the file: same keys, and the sd-webui model hash (sha256 over the tensor section of the fp16 file) the reference's save_weights stamped
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "lora.safetensors")
        net.save_weights(f, dtype=torch.float16, metadata={"name": "adoption", "step": str(spec["steps"]), "format": "pt"})
        with safe_open(f, "pt") as fh:
            assert sorted(fh.keys()) == sorted(meta["file_keys"])
            assert fh.metadata()["sshs_model_hash"] == meta["sshs_model_hash"]
            assert fh.metadata()["sshs_legacy_hash"] == meta["sshs_legacy_hash"]
        if tag == "lora":
            # resume: the golden holds what the reference's own load_weights produced in a fresh adopted network — same rank, and the rank-8
            # file into a rank-4 network (shrink, network_mixins.py:737-775); the fused network's load_weights must land on the same prediction
            lat_, emb_, pooled_, ts_, _ = batches(1, seed=21)[0]
            pe_ = SimpleNamespace(text_embeds=emb_, pooled_embeds=pooled_)
            for sub, rank in (("loaded", 8), ("loaded_shrunk", 4)):
                _, nat2, sd2 = native_plugin()
                torch.manual_seed(123)
                net2 = FusedLoRANetwork(nat2, lora_dim=rank, alpha=rank, transformer_block_names=sd2.get_transformer_block_names(), base_model=sd2)
                net2.apply_to()
                net2.build_arena("cpu", groups=nat2.lora_groups())
                net2.refresh_shadows(ref_ops)
                nat2.attach_network(net2)
                assert net2.load_weights(f) is None
                with torch.no_grad(), net2:
                    assert torch.equal(sd2.get_noise_prediction(lat_, ts_, pe_, 1.0, False), gold[f"lora/pred_{sub}"]), sub
            assert not torch.equal(gold["lora/pred_loaded"], gold["lora/pred_loaded_shrunk"])
    lat, emb, pooled, ts, _ = batches(1, seed=21)[0]
    pe = SimpleNamespace(text_embeds=emb, pooled_embeds=pooled)
    with torch.no_grad():
        assert torch.equal(sd.get_noise_prediction(lat, ts, pe, 1.0, False), gold[f"{tag}/pred_inactive"])
        with net:
            assert torch.equal(sd.get_noise_prediction(lat, ts, pe, 1.0, False), gold[f"{tag}/pred_active"])
    assert not torch.equal(gold[f"{tag}/pred_inactive"], gold[f"{tag}/pred_active"])
    if tag == "lora":
        lat_, emb_, pooled_, ts_, pe_ = lat, emb, pooled, ts, pe
        # the reference's own merge_in / merge_out over the adopted network (recorded by the generator, which also checks that the native
        # layer's transposed copy followed the merged weight): merged == active, merge_out == base; the fused network's merge agrees
        def r(a, b):
            return ((a - b).norm() / b.norm()).item()

        assert r(gold["lora/pred_merged"], gold["lora/pred_active"]) < 1e-5 and r(gold["lora/pred_after_merge_out"], gold["lora/pred_inactive"]) < 1e-5
        with torch.no_grad():
            net.merge_in(1.0, ops=ref_ops)
            with net:
                assert r(sd.get_noise_prediction(lat_, ts_, pe_, 1.0, False), gold["lora/pred_merged"]) < 1e-5
            net.merge_out(1.0, ops=ref_ops)


def test_golden_records_where_unsupported_reference_networks_are_refused():
    with safe_open(GOLD, "pt") as fh:
        refused = json.loads(fh.metadata()["meta"])["refused"]
    assert refused == {"lorm_use_bias": "AdoptionError", "fullrank": "AdoptionError", "full_if_contains": "AdoptionError"}


@pytest.mark.skipif(not os.path.isdir("/root/reference/toolkit"), reason="the reference tree is not mounted here")
def test_committed_adoption_fixture_is_what_the_reference_produces_today(tmp_path):
    """regenerate with the reference's own classes (separate process: the import shims stay out of this one) and compare"""
    code = ("import sys; sys.argv=['make_golden.py']; sys.path.insert(0, %r); import runpy; "
            "g = runpy.run_path(%r, run_name='not_main'); g['golden_adoption'](%r)"
            % (os.path.join(HERE, "golden"), os.path.join(HERE, "golden", "make_golden.py"), str(tmp_path)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    new, old = load_file(str(tmp_path / "adoption_flux_tiny.safetensors")), load_file(GOLD)
    assert set(new) == set(old)
    for k in old:
        assert torch.equal(new[k], old[k]), k


# ---------------------------------------------------------------- the protocol, driven by the oracle's restatement of the reference classes
def ref_sequence(nat, sd, rank=4, network_type="lora", seed=99):
    """BaseSDTrainProcess.py:1949-2039 on the oracle's restated network: construct over sd.get_model_to_train(), force_to, sd.network =,
    apply_to (forward swap), prepare_optimizer_params"""
    torch.manual_seed(seed)
    net = lora_ref.RefLoRANetwork(sd.get_model_to_train(), rank, 1.0, block_names=tuple(sd.get_transformer_block_names()), network_type=network_type)
    net.force_to(torch.device("cpu"), torch.float32)
    sd.network = net
    net._update_torch_multiplier()
    net.apply_to(None, sd.unet, False, True)
    net.prepare_grad_etc(None, sd.unet)
    params = net.prepare_optimizer_params(text_encoder_lr=1e-3, unet_lr=1e-3, default_lr=1e-3)
    return net, params, [p for g in params for p in g["params"]]


def test_adopted_oracle_network_trains_like_the_fused_twin_and_keeps_parameter_identity():
    _, nat, sd = native_plugin()
    net, params, plist = ref_sequence(nat, sd, rank=8)
    ids = [id(p) for p in plist]
    opt = torch.optim.AdamW(params, lr=1e-3, eps=1e-6, weight_decay=0.01)
    ema = [p.detach().clone() for p in plist]
    _, nat2, sd2 = native_plugin()
    torch.manual_seed(99)
    net2 = FusedLoRANetwork(nat2, lora_dim=8, alpha=8, transformer_block_names=sd2.get_transformer_block_names(), base_model=sd2)
    net2.apply_to()
    net2.build_arena("cpu", groups=nat2.lora_groups())
    net2.refresh_shadows(ref_ops)
    nat2.attach_network(net2)
    params2 = net2.prepare_optimizer_params(default_lr=1e-3)
    plist2 = [p for g in params2 for p in g["params"]]
    opt2 = torch.optim.AdamW(params2, lr=1e-3, eps=1e-6, weight_decay=0.01)
    ema2 = [p.detach().clone() for p in plist2]
    for b in batches(3):
        la, ga = trainer_step(net, plist, opt, ema, sd, b)
        lb, gb = trainer_step(net2, plist2, opt2, ema2, sd2, b, after_step=lambda: net2.refresh_shadows(ref_ops))
        assert torch.equal(la, lb)
        assert all(torch.equal(x, y) for x, y in zip(ga, gb))
    ad = nat.network
    assert isinstance(ad, AdoptedNetwork) and ad.foreign is net and ad.aliasing_intact()
    assert [id(p) for g in opt.param_groups for p in g["params"]] == ids  # the optimizer still holds the reference's own Parameter objects
    assert all(torch.equal(a, b) for a, b in zip(plist, plist2)) and all(torch.equal(a, b) for a, b in zip(ema, ema2))
    assert torch.equal(ad.arena_p, net2.arena_p)
    # every Parameter (and, while attached, its .grad) is a view of the flat arenas the kernels and a DP all-reduce treat as one tensor
    lo, hi = ad.arena_p.data_ptr(), ad.arena_p.data_ptr() + ad.arena_p.numel() * 4
    assert all(lo <= p.data_ptr() < hi for p in plist)
    assert plist[0].grad is None  # zero_grad(set_to_none=True) dropped the views; the next backward re-attaches them
    # the reference-side state dict IS the arena content (what its get_state_dict / save_weights read)
    sdict = net.peft_state_dict(dtype=torch.float32)
    sd2d = net2.get_state_dict(dtype=torch.float32)
    assert list(sdict) == list(sd2d) and all(torch.equal(sdict[k], sd2d[k]) for k in sdict)


def test_inactive_zero_multiplier_and_merged_networks_run_the_base_model():
    ref, nat, sd = native_plugin()
    net, params, plist = ref_sequence(nat, sd)
    with torch.no_grad():
        for m in net.unet_loras:
            m.lora_up.weight.normal_(0, 0.05)
    lat, emb, pooled, ts, _ = batches(1)[0]
    pe = SimpleNamespace(text_embeds=emb, pooled_embeds=pooled)
    with torch.no_grad():
        base = sd.get_noise_prediction(lat, ts, pe, 1.0, False)            # adapter attached but `with network:` not entered
        with net:
            act = sd.get_noise_prediction(lat, ts, pe, 1.0, False)
            net.multiplier = 0.0                                           # network_mixins.py:293-295
            assert torch.equal(sd.get_noise_prediction(lat, ts, pe, 1.0, False), base)
            net.multiplier = 1.0
            net.is_merged_in = True                                        # network_mixins.py:289-291
            assert torch.equal(sd.get_noise_prediction(lat, ts, pe, 1.0, False), base)
            net.is_merged_in = False
            assert torch.equal(sd.get_noise_prediction(lat, ts, pe, 1.0, False), act)
    assert not torch.equal(act, base)
    # base == the oracle model without any adapter
    img_ids, txt_ids = flux_ref.make_ids(lat.shape[2], lat.shape[3], emb.shape[1])
    with torch.no_grad():
        p_ref = flux_ref.unpack_latents(ref(flux_ref.pack_latents(lat), emb, pooled, ts / 1000, img_ids, txt_ids, torch.full((lat.shape[0],), 1.0)),
                                        lat.shape[2], lat.shape[3])
    assert torch.allclose(base, p_ref, rtol=2e-4, atol=2e-5)


def test_adopted_prediction_and_gradients_match_oracle_autograd_with_a_per_sample_multiplier():
    """the adoption carries torch_multiplier vectors (slider-style batches, network_mixins.py:313-321): compare with autograd of the SAME oracle
    network on the oracle (pure torch) model"""
    ref, nat, sd = native_plugin()
    net, params, plist = ref_sequence(nat, sd, rank=4)
    torch.manual_seed(99)
    rnet = lora_ref.RefLoRANetwork(ref, 4, 1.0, block_names=("transformer_blocks", "single_transformer_blocks"))
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for a, b in zip(net.unet_loras, rnet.unet_loras):
            assert a.lora_name == b.lora_name
            a.lora_up.weight.copy_(torch.randn(a.lora_up.weight.shape, generator=g) * 0.05)
            b.lora_up.weight.copy_(a.lora_up.weight)
            b.lora_down.weight.copy_(a.lora_down.weight)
    rnet.apply_to()
    net.multiplier = [0.5, -1.5]
    rnet.multiplier = [0.5, -1.5]
    lat, emb, pooled, ts, target = batches(1)[0]
    pe = SimpleNamespace(text_embeds=emb, pooled_embeds=pooled)
    img_ids, txt_ids = flux_ref.make_ids(lat.shape[2], lat.shape[3], emb.shape[1])
    with rnet:
        p_ref = flux_ref.unpack_latents(ref(flux_ref.pack_latents(lat), emb, pooled, ts / 1000, img_ids, txt_ids, torch.full((2,), 1.0)), lat.shape[2], lat.shape[3])
        torch.nn.functional.mse_loss(p_ref, target).backward()
    with net:
        pred = sd.get_noise_prediction(lat, ts, pe, 1.0, False)
        assert torch.allclose(pred, p_ref, rtol=2e-4, atol=2e-5)
        torch.nn.functional.mse_loss(pred, target).backward()
    for a, b in zip(net.unet_loras, rnet.unet_loras):
        for x, y in ((a.lora_down.weight.grad, b.lora_down.weight.grad), (a.lora_up.weight.grad, b.lora_up.weight.grad)):
            assert (x - y).norm() <= 3e-4 * y.norm() + 1e-7, a.lora_name


def test_storage_replaced_behind_our_back_is_re_adopted_and_load_state_dict_is_seen():
    _, nat, sd = native_plugin()
    net, params, plist = ref_sequence(nat, sd)
    lat, emb, pooled, ts, target = batches(1)[0]
    pe = SimpleNamespace(text_embeds=emb, pooled_embeds=pooled)
    with torch.no_grad(), net:
        p0 = sd.get_noise_prediction(lat, ts, pe, 1.0, False)
    ad = nat.network
    # (a) in-place load (the reference's load_weights ends in load_state_dict): values land in the arena, shadows follow at the next forward
    g = torch.Generator().manual_seed(3)
    new_sd = {k: (torch.randn(v.shape, generator=g) * 0.05 if "lora_up" in k else v.clone()) for k, v in net.state_dict().items()}
    net.load_state_dict(new_sd)
    assert ad.aliasing_intact()
    with torch.no_grad(), net:
        p1 = sd.get_noise_prediction(lat, ts, pe, 1.0, False)
    assert not torch.equal(p0, p1)
    # (b) a Parameter's storage replaced (rank-preserving `.data =`, e.g. network.to() round trip): noticed, copied in, re-pointed
    m = net.unet_loras[3]
    m.lora_up.weight.data = m.lora_up.weight.data.clone() * 2.0
    assert not ad.aliasing_intact()
    with torch.no_grad(), net:
        p2 = sd.get_noise_prediction(lat, ts, pe, 1.0, False)
    assert nat.network is ad and ad.aliasing_intact() and not torch.equal(p1, p2)
    lo, hi = ad.arena_p.data_ptr(), ad.arena_p.data_ptr() + ad.arena_p.numel() * 4
    assert lo <= m.lora_up.weight.data_ptr() < hi
    # and training continues on the same Parameter objects
    opt = torch.optim.AdamW(params, lr=1e-3, eps=1e-6)
    before = plist[0].detach().clone()
    trainer_step(net, plist, opt, [p.detach().clone() for p in plist], sd, (lat, emb, pooled, ts, target))
    assert not torch.equal(before, plist[0].detach())


def test_gradient_accumulation_over_two_backwards_sums_into_the_same_views():
    _, nat, sd = native_plugin()
    net, params, plist = ref_sequence(nat, sd)
    with torch.no_grad():
        for m in net.unet_loras:
            m.lora_up.weight.normal_(0, 0.05)
    b1, b2 = batches(2)
    opt = torch.optim.AdamW(params, lr=1e-3)

    def bwd(b):
        lat, emb, pooled, ts, target = b
        with net:
            pred = sd.get_noise_prediction(lat, ts, SimpleNamespace(text_embeds=emb, pooled_embeds=pooled), 1.0, False)
            torch.nn.functional.mse_loss(pred, target).backward()
        return [p.grad.detach().clone() for p in plist]

    opt.zero_grad()
    g1 = bwd(b1)
    opt.zero_grad()
    g2 = bwd(b2)
    opt.zero_grad()
    bwd(b1)
    g12 = bwd(b2)  # SDTrainer.py:2226-2238 under gradient_accumulation: no zero_grad between the micro-batches
    for a, b, c in zip(g1, g2, g12):
        assert torch.allclose(a + b, c, rtol=1e-5, atol=1e-8)


def test_what_cannot_be_adopted_raises_at_apply_to_never_base_only():
    _, nat, sd = native_plugin()
    lin = nat.transformer_blocks[0].attn.to_q

    class LoRAModule(torch.nn.Module):  # the reference's class name, with use_bias (LoRM: BaseSDTrainProcess.py:1969)
        def __init__(self, lin, bias):
            super().__init__()
            self.lora_name, self.lora_dim = "x", 4
            self.lora_down = torch.nn.Linear(lin.in_features, 4, bias=False)
            self.lora_up = torch.nn.Linear(4, lin.out_features, bias=bias)
            self.org_module = [lin]

        def apply_to(self):
            self.org_forward = self.org_module[0].forward
            self.org_module[0].forward = self.forward

        def forward(self, x):
            return x

    with pytest.raises(AdoptionError, match="use_bias"):
        LoRAModule(lin, True).apply_to()
    assert lin.lora is None

    class FullModule(LoRAModule):
        pass

    with pytest.raises(TypeError, match="cannot be replaced"):
        FullModule(lin, False).apply_to()
    with pytest.raises(TypeError):
        lin.forward = lambda x: x
    with pytest.raises(AdoptionError, match="lorm"):
        sd.network = SimpleNamespace(is_lorm=True)
    with pytest.raises(AdoptionError, match="text-encoder"):
        sd.network = SimpleNamespace(text_encoder_loras=[object()])
    # a network whose modules were never attached (apply_to skipped) cannot silently train nothing
    _, nat2, sd2 = native_plugin()
    net = lora_ref.RefLoRANetwork(nat2, 4, 1.0, block_names=("transformer_blocks",))
    net.unet_loras[0].apply_to()
    lat, emb, pooled, ts, _ = batches(1)[0]
    with pytest.raises(AdoptionError, match="not attached"):
        with net:
            sd2.get_noise_prediction(lat, ts, SimpleNamespace(text_embeds=emb, pooled_embeds=pooled), 1.0, False)


def test_plugin_instance_flags_survive_basemodel_init_order():
    """BaseModel.__init__ runs first in the real plug-in and sets instance attributes is_flow_matching = is_transformer =
    False, use_old_lokr_format = True; the mirror's constructor must set them again (here: simulate that order)."""

    class FakeBase:
        def __init__(self, *a, **k):
            self.is_flow_matching, self.is_transformer, self.use_old_lokr_format, self.network = False, False, True, None

    def init(self, device, model_config=None, dtype="bf16", **kw):
        FakeBase.__init__(self)
        Flux1MI355Model.__init__(self, device, model_config, dtype, **kw)

    Real = type("Flux1MI355", (Flux1MI355Model, FakeBase), {"__init__": init})
    obj = Real("cpu")
    assert obj.is_flow_matching is True and obj.is_transformer is True and obj.use_old_lokr_format is False
    with open(os.path.join(HERE, "golden", "plugin_registration.json")) as f:
        reg = json.load(f)
    flags = {c["name"]: c["instance_flags"] for c in reg["classes"]}
    assert flags["Flux1MI355"] == {"is_flow_matching": True, "is_transformer": True, "use_old_lokr_format": False}
    assert flags["Wan21MI355"] == {"is_flow_matching": True, "is_transformer": True, "use_old_lokr_format": False}
    assert flags["StableDiffusionMI355"] == {"is_flow_matching": False, "is_transformer": False, "use_old_lokr_format": False}


def _adopted_dp_worker(rank, world, port, out, overlap=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    AdoptedNetwork.dp_overlap = bool(overlap)
    import datetime

    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    torch.set_num_threads(2)
    _, nat, sd = native_plugin()
    net, params, plist = ref_sequence(nat, sd, rank=4)
    with torch.no_grad():
        g = torch.Generator().manual_seed(5)
        for m in net.unet_loras:
            m.lora_up.weight.copy_(torch.randn(m.lora_up.weight.shape, generator=g) * 0.05)
    opt = torch.optim.AdamW(params, lr=1e-3, eps=1e-6, weight_decay=0.01)
    ema = [p.detach().clone() for p in plist]
    for lat, emb, pooled, ts, target in batches(2, B=4):
        sl = slice(rank * 2, rank * 2 + 2)
        tsr = torch.tensor([700.0, 250.0, 999.0, 31.0])[sl]
        trainer_step(net, plist, opt, ema, sd, (lat[sl], emb[sl], pooled[sl], tsr, target[sl]))
    torch.save(nat.network.arena_p.clone(), os.path.join(out, f"p{rank}.pt"))
    torch.save(int(nat.network.dp_pieces_issued), os.path.join(out, f"pieces{rank}.pt"))
    dist.destroy_process_group()


def test_adopted_network_under_two_ranks_averages_gradients_like_one_rank_on_the_whole_batch(tmp_path):
    """`accelerate launch` with 2 processes over the unmodified trainer: DDP's autograd hooks never see the explicit backward, so the adopted
    network all-reduces its gradient arena itself at the end of every backward — replicas bit-identical, equal to one rank on the
    concatenated batch."""
    import torch.multiprocessing as mp

    from tests.conftest import free_port

    mp.spawn(_adopted_dp_worker, args=(2, free_port(), str(tmp_path)), nprocs=2, join=True)
    p0, p1 = torch.load(tmp_path / "p0.pt"), torch.load(tmp_path / "p1.pt")
    assert torch.equal(p0, p1)
    # round 6: the collective went out in two pieces per backward from INSIDE the explicit backward (single-stream adapters first), like the
    # fused train step's; one blocking all-reduce at the end (dp_overlap = False) gives the same bits
    assert torch.load(tmp_path / "pieces0.pt") == 2 * 2
    (tmp_path / "blocking").mkdir()
    mp.spawn(_adopted_dp_worker, args=(2, free_port(), str(tmp_path / "blocking"), False), nprocs=2, join=True)
    assert torch.load(tmp_path / "blocking" / "pieces0.pt") == 0
    assert torch.equal(torch.load(tmp_path / "blocking" / "p0.pt"), p0)
    _, nat, sd = native_plugin()
    net, params, plist = ref_sequence(nat, sd, rank=4)
    with torch.no_grad():
        g = torch.Generator().manual_seed(5)
        for m in net.unet_loras:
            m.lora_up.weight.copy_(torch.randn(m.lora_up.weight.shape, generator=g) * 0.05)
    p_init = torch.cat([p.detach().reshape(-1) for p in plist]).clone()
    opt = torch.optim.AdamW(params, lr=1e-3, eps=1e-6, weight_decay=0.01)
    ema = [p.detach().clone() for p in plist]
    for lat, emb, pooled, ts, target in batches(2, B=4):
        trainer_step(net, plist, opt, ema, sd, (lat, emb, pooled, torch.tensor([700.0, 250.0, 999.0, 31.0]), target))
    one = nat.network.arena_p
    assert torch.allclose(one, p0, rtol=1e-3, atol=2e-6), (one - p0).abs().max()
    assert not torch.equal(torch.cat([p.detach().reshape(-1) for p in plist]), p_init)


@pytest.mark.parametrize("xl", [False, True], ids=["sd15", "sdxl"])
def test_unet_kohya_network_with_conv_adapters_is_adopted_bit_for_bit(xl):
    """SD1.5 / SDXL UNets (BASELINE configs 1-2): the reference finds the adapters by CLASS NAME (`Linear`, `Conv2d` with kernel_size (1, 1) or —
    with network.conv — (3, 3); toolkit/lora_special.py:488-490, 585-590), builds Conv2d-shaped lora_down / lora_up for the convolutions
    (lora_special.py:95-104) and swaps their forward.  The native UNet's holders carry that class name; the adopted network views the 4-D
    Conv2d weights inside the same [rank_pad, in*k*k] / [out, rank_pad] arena blocks a FusedLoRANetwork uses.  Same trainer sequence, bit for bit."""
    from ai_toolkit_amd.plugin import StableDiffusionMI355Model
    from ai_toolkit_amd.unet import UNet2DConditionModel
    from oracle import unet_ref
    from tests.test_unet_cpu import TINY_SD15, TINY_SDXL

    cfg = TINY_SDXL if xl else TINY_SD15

    def native():
        torch.manual_seed(0)
        ref = unet_ref.UNet2DConditionModel(**cfg)
        unet_ref.init_synthetic_(ref, seed=11)
        nat = UNet2DConditionModel(**cfg, dtype=torch.float32, device="cpu", ops=ref_ops)
        nat.load_state_dict(ref.state_dict(), strict=True)
        nat.prepare()
        return nat, StableDiffusionMI355Model("cpu", model=nat, dtype=torch.float32, is_xl=xl)

    nat_a, sd_a = native()
    nat_f, sd_f = native()
    # the discovery the reference runs: class names only
    names = {m.__class__.__name__ for m in nat_a.modules()}
    assert "Conv2d" in names and "Linear" in names and not ({"Conv1x1", "Conv3x3"} & names)
    torch.manual_seed(99)
    net_a = lora_ref.RefLoRANetwork(sd_a.get_model_to_train(), 4, 1.0, target=tuple(sd_a.target_lora_modules), kohya_unet=True, alpha=2.0,
                                    conv_lora_dim=2, conv_alpha=1.0)
    torch.manual_seed(99)
    net_f = FusedLoRANetwork(nat_f, lora_dim=4, alpha=2.0, conv_lora_dim=2, conv_alpha=1.0, target_lin_modules=("Transformer2DModel",),
                             is_transformer=False, peft_format=False, transformer_only=False, base_model_version="sdxl" if xl else "sd1")
    assert [m.lora_name for m in net_a.unet_loras] == [m.lora_name for m in net_f.unet_loras]
    n3 = sum(1 for m in net_a.unet_loras if m.lora_down.weight.dim() == 4 and m.lora_down.weight.shape[2:] == (3, 3))
    n1 = sum(1 for m in net_a.unet_loras if m.lora_down.weight.dim() == 4 and m.lora_down.weight.shape[2:] == (1, 1))
    assert n3 > 0 and (n1 > 0 or xl)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for a, b in zip(net_a.unet_loras, net_f.unet_loras):
            assert torch.equal(a.lora_down.weight.reshape(b.lora_down.weight.shape), b.lora_down.weight), a.lora_name  # same init draws
            up = torch.randn(b.lora_up.weight.shape, generator=g) * 0.05
            a.lora_up.weight.copy_(up.reshape(a.lora_up.weight.shape))
            b.lora_up.weight.copy_(up)
    net_a.force_to(torch.device("cpu"), torch.float32)
    sd_a.network = net_a
    net_a._update_torch_multiplier()
    net_a.apply_to(None, sd_a.unet, False, True)
    pa = [p for grp in net_a.prepare_optimizer_params(None, 1e-3, 1e-3) for p in grp["params"]]
    net_f.apply_to()
    net_f.build_arena("cpu", groups=nat_f.lora_groups())
    net_f.refresh_shadows(ref_ops)
    nat_f.attach_network(net_f)
    pf = net_f.prepare_optimizer_params(default_lr=1e-3)[0]["params"]
    oa, of = torch.optim.AdamW(pa, lr=1e-3, eps=1e-6, weight_decay=0.01), torch.optim.AdamW(pf, lr=1e-3, eps=1e-6, weight_decay=0.01)
    gen = torch.Generator().manual_seed(4)
    B, H, W = 2, 16, 8
    pooled_dim = cfg["projection_class_embeddings_input_dim"] - 6 * cfg["addition_time_embed_dim"] if xl else 8
    losses = []
    for k in range(2):
        lat, tgt = torch.randn(B, 4, H, W, generator=gen), torch.randn(B, 4, H, W, generator=gen)
        pe = SimpleNamespace(text_embeds=torch.randn(B, 7, cfg["cross_attention_dim"], generator=gen), pooled_embeds=torch.randn(B, pooled_dim, generator=gen))
        ts = torch.tensor([640, 17])
        out = []
        for net, sd, opt, plist in ((net_a, sd_a, oa, pa), (net_f, sd_f, of, pf)):
            opt.zero_grad()
            with net:
                pred = sd.predict_noise(lat, text_embeddings=pe, timestep=ts)
                loss = torch.nn.functional.mse_loss(pred.float(), tgt.float(), reduction="none").mean([1, 2, 3]).mean()
                loss.backward()
            torch.nn.utils.clip_grad_norm_(plist, 1.0)
            opt.step()
            opt.zero_grad(set_to_none=True)
            out.append(loss.detach())
        net_f.refresh_shadows(ref_ops)
        assert torch.equal(out[0], out[1]), (k, out)
        losses.append(out[1])
    ad = nat_a.network
    assert isinstance(ad, AdoptedNetwork) and ad.aliasing_intact() and torch.equal(ad.arena_p, net_f.arena_p)
    for a, b in zip(pa, pf):
        assert torch.equal(a.detach().reshape(b.shape), b.detach())
    # ... and both equal the run of the reference's OWN LoRASpecialNetwork (Conv2d LoRAModules, kohya keys) over the real plug-in class
    tag = "unet_sdxl_conv" if xl else "unet_sd15_conv"
    gold = load_file(GOLD)
    with safe_open(GOLD, "pt") as fh:
        meta = json.loads(fh.metadata()["meta"])[tag]
    assert meta["names"] == [m.lora_name for m in net_f.unet_loras] and meta["n_conv3x3"] == n3 and meta["n_conv1x1"] == n1 and not meta["peft_format"]
    assert torch.equal(torch.stack(losses), gold[f"{tag}/losses"])
    for i, b in enumerate(pf):
        assert torch.equal(gold[f"{tag}/param/{i}"].reshape(b.shape), b.detach()), i
    sf_all = net_f.get_state_dict(dtype=torch.float32)
    assert list(sf_all) == meta["saved_keys"]
    for k, v in sf_all.items():
        assert torch.equal(v, gold[f"{tag}/saved/{k}"]), k
    # the reference-side state dict (what its get_state_dict walks): Conv2d-shaped tensors with the arena's values
    sa = {k: v for k, v in net_a.state_dict().items()}
    sf = net_f.get_state_dict(dtype=torch.float32)
    for k, v in sf.items():
        assert torch.equal(sa[k].detach().reshape(v.shape), v), k
        if k.endswith("lora_down.weight") and v.dim() == 4:
            assert tuple(sa[k].shape) == tuple(v.shape)


def test_wan_network_built_by_the_reference_over_the_wan_plugin_equals_the_fused_twin():
    """BASELINE config 4: the golden `wan` run is the reference's LoRASpecialNetwork over the real Wan21MI355 plug-in (block filter from
    get_transformer_block_names(), save keys through convert_lora_weights_before_save: toolkit/models/wan21/wan21.py:726-735)."""
    from ai_toolkit_amd.plugin import Wan21MI355Model
    from ai_toolkit_amd.wan import WanTransformer3DModel
    from oracle import wan_ref
    from tests.test_wan_cpu import CFG as WCFG

    gold = load_file(GOLD)
    with safe_open(GOLD, "pt") as fh:
        meta = json.loads(fh.metadata()["meta"])["wan"]
    torch.manual_seed(0)
    ref = wan_ref.WanTransformer3DModel(**WCFG)
    wan_ref.init_synthetic_(ref, seed=99, std=0.05)
    nat = WanTransformer3DModel(**WCFG, dtype=torch.float32, device="cpu", ops=ref_ops)
    nat.load_state_dict(ref.state_dict(), strict=True)
    nat.prepare()
    sd = Wan21MI355Model("cpu", model=nat, dtype=torch.float32)
    torch.manual_seed(99)
    net = FusedLoRANetwork(nat, lora_dim=8, alpha=8, target_lin_modules=tuple(sd.target_lora_modules), transformer_block_names=sd.get_transformer_block_names(),
                           base_model_version="wan_2.1", base_model=sd)
    assert [m.lora_name for m in net.unet_loras] == meta["names"] and len(net.unet_loras) == 10 * WCFG["num_layers"]
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for m in net.unet_loras:
            m.lora_up.weight.copy_(torch.randn(m.lora_up.weight.shape, generator=g) * 0.05)
    net.apply_to()
    net.build_arena("cpu", groups=nat.lora_groups())
    net.refresh_shadows(ref_ops)
    nat.attach_network(net)
    plist = net.prepare_optimizer_params(default_lr=1e-3)[0]["params"]
    opt = torch.optim.AdamW(plist, lr=1e-3, eps=1e-6, weight_decay=0.01)
    gen = torch.Generator().manual_seed(3)
    losses = []
    for k in range(2):
        lat, tgt = torch.randn(2, 16, 3, 8, 4, generator=gen), torch.randn(2, 16, 3, 8, 4, generator=gen)
        pe = SimpleNamespace(text_embeds=torch.randn(2, 5, WCFG["text_dim"], generator=gen), pooled_embeds=None)
        opt.zero_grad()
        with net:
            pred = sd.get_noise_prediction(lat, torch.tensor([310.0, 845.0]), pe)
            loss = torch.nn.functional.mse_loss(pred.float(), tgt.float(), reduction="none").mean([1, 2, 3, 4]).mean()
            loss.backward()
        torch.nn.utils.clip_grad_norm_(plist, 1.0)
        opt.step()
        opt.zero_grad(set_to_none=True)
        net.refresh_shadows(ref_ops)
        losses.append(loss.detach())
    assert torch.equal(torch.stack(losses), gold["wan/losses"])
    for i, p in enumerate(plist):
        assert torch.equal(p.detach(), gold[f"wan/param/{i}"]), i
    sdict = net.get_state_dict(dtype=torch.float32)
    assert list(sdict) == meta["saved_keys"] and next(iter(sdict)).startswith("diffusion_model.blocks.0.")
    for k, v in sdict.items():
        assert torch.equal(v, gold[f"wan/saved/{k}"]), k
