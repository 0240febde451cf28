"""LoKr (Kronecker adapter, reference toolkit/models/lokr.py): the oracle module against vectors produced by the reference's own
LokrModule / LoRASpecialNetwork(network_type='lokr') (tests/golden/lokr_flux_tiny.safetensors), then the native network + host
graph (oracle kernel table, fp32) against both, the saved-file surface, and a full train step against the autograd oracle."""
import json
import os

import pytest
import torch
from safetensors import safe_open
from safetensors.torch import load_file

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd.flux import FluxTransformer2DModel
from ai_toolkit_amd.lora import FusedLoRANetwork, factorization
from ai_toolkit_amd.trainer import FluxLoRATrainStep
from oracle import lora_ref, ref_ops, train_ref
from tests.test_oracle_golden import G, TINY, oracle_model, tiny_inputs

BIG = 9999999999  # lokr_full_rank (toolkit/config_modules.py:204-209)


@pytest.fixture(scope="module")
def gold():
    path = os.path.join(G, "lokr_flux_tiny.safetensors")
    with safe_open(path, "pt") as f:
        meta = {k: json.loads(v) for k, v in f.metadata().items()}
    return load_file(path), meta


def test_factorization_equals_the_reference_function(gold):
    """toolkit/models/lokr.py:22-59 executed by make_golden.py over model dims and factors (its docstring table is not what
    the code returns, e.g. 250 -> (10, 25))."""
    table = gold[1]["factorization"]
    assert len(table) == 85
    for key, want in table.items():
        d, f = (int(v) for v in key.split(":"))
        assert list(factorization(d, f)) == want == list(lora_ref.factorization(d, f)), key
    assert factorization(3072) == (48, 64) and factorization(12288) == (96, 128) and factorization(15360) == (120, 128)


def test_oracle_lokr_matches_reference_lokr_network(gold):
    t, meta = gold
    model = oracle_model()
    torch.manual_seed(99)
    net = lora_ref.RefLoRANetwork(model, BIG, network_type="lokr")
    assert [m.lora_name for m in net.unet_loras] == meta["names"]
    assert [n for n, _ in net.unet_loras[0].named_parameters()] == meta["param_order"] and meta["scale"] == 1.0
    for m in net.unet_loras:
        assert [list(m.lokr_w1.shape), list(m.lokr_w2.shape)] == meta["shapes"][m.lora_name]
        assert torch.equal(m.lokr_w1, t[f"init/{m.lora_name}/w1"]), m.lora_name  # same RNG consumption
        assert float(m.lokr_w2.abs().max()) == 0.0
        with torch.no_grad():
            m.lokr_w2.copy_(t[f"set/{m.lora_name}/w2"])
    net.apply_to()
    with net:
        pred = model(*tiny_inputs())
        (pred * t["fwd/w"]).sum().backward()
    assert torch.allclose(pred, t["fwd/pred"], rtol=1e-5, atol=1e-6)
    for m in net.unet_loras:
        assert torch.allclose(m.lokr_w1.grad, t[f"grad/{m.lora_name}/w1"], rtol=2e-4, atol=2e-6), m.lora_name
        assert torch.allclose(m.lokr_w2.grad, t[f"grad/{m.lora_name}/w2"], rtol=2e-4, atol=2e-6), m.lora_name
    sd = net.peft_state_dict(torch.float32)
    assert list(sd.keys()) == meta["saved_keys"]
    for k, v in sd.items():
        assert torch.equal(v, t[f"saved/{k}"]), k


def native_pair(factor=-1):
    ref = oracle_model()
    nat = FluxTransformer2DModel(**TINY, dtype=torch.float32, device="cpu", ops=ref_ops)
    nat.load_state_dict(ref.state_dict(), strict=True)
    torch.manual_seed(99)
    net = FusedLoRANetwork(nat, lora_dim=BIG, alpha=BIG, network_type="lokr", lokr_factor=factor)
    return ref, nat, net


def test_native_lokr_network_and_host_graph_match_reference_vectors(gold, tmp_path):
    t, meta = gold
    ref, nat, net = native_pair()
    assert [m.lora_name for m in net.unet_loras] == meta["names"]
    for m in net.unet_loras:
        assert [list(m.lokr_w1.shape), list(m.lokr_w2.shape)] == meta["shapes"][m.lora_name]
        assert torch.equal(m.lokr_w1, t[f"init/{m.lora_name}/w1"]), m.lora_name
        assert m.scale == 1.0
    net.apply_to(None, nat, False, True)
    net.force_to("cpu", torch.float32)
    with torch.no_grad():
        for m in net.unet_loras:
            m.lokr_w2.copy_(t[f"set/{m.lora_name}/w2"])
    net.refresh_shadows(ref_ops)
    nat.attach_network(net)
    with net:
        pred = nat.forward_native(*tiny_inputs())
        assert torch.allclose(pred, t["fwd/pred"], rtol=1e-4, atol=1e-5)
        net.zero_grad_arena()
        nat.backward_native(t["fwd/w"].clone())
    for m in net.unet_loras:
        assert torch.allclose(m.lokr_w1.grad, t[f"grad/{m.lora_name}/w1"], rtol=3e-4, atol=1e-5), m.lora_name
        assert torch.allclose(m.lokr_w2.grad, t[f"grad/{m.lora_name}/w2"], rtol=3e-4, atol=1e-5), m.lora_name
    # saved file: the reference's keys and values (alpha kept for LoKr), loadable back
    f = str(tmp_path / "lokr.safetensors")
    net.save_weights(f, dtype=torch.float32)
    sd = load_file(f)
    assert sorted(sd.keys()) == sorted(meta["saved_keys"])
    for k, v in sd.items():
        assert torch.equal(v, t[f"saved/{k}"]), k
    before = net.arena_p.clone()
    net.arena_p.zero_()
    assert net.load_weights(f) is None
    assert torch.equal(net.arena_p, before)
    params = net.prepare_optimizer_params(default_lr=1e-4)[0]["params"]
    assert params[0] is net.unet_loras[0].lokr_w1 and params[1] is net.unet_loras[0].lokr_w2


@pytest.fixture
def force_two_stage(monkeypatch):
    """every LoKr layer takes the two-stage form (W2 through a GEMM over (token x factor-index) rows + the small-factor mix on the narrower side) —
    what lora.check_kron_fits selects when W2 does not fit the per-token kernel's LDS (explicit small lokr_factor on a wide layer)"""
    from ai_toolkit_amd import lora as L

    real = L.check_kron_fits

    def forced(name, in_m, in_n, out_l, out_k):
        real(name, in_m, in_n, out_l, out_k)
        return "two_stage"

    monkeypatch.setattr(L, "check_kron_fits", forced)
    import ai_toolkit_amd.adopt as A

    monkeypatch.setattr(A, "check_kron_fits", forced)


@pytest.mark.parametrize("factor", [-1, 4, 8])
def test_two_stage_lokr_train_steps_match_autograd_oracle(force_two_stage, factor):
    """the double blocks' segmented joint buffers, the single blocks' windowed proj_out data gradient, adaLN projections and both narrow-side cases
    (b_in <= b_out, b_in > b_out) all occur in the tiny FLUX"""
    test_lokr_train_steps_match_autograd_oracle(factor, expect_two_stage=True)


@pytest.mark.parametrize("factor", [-1, 4, 8])
def test_lokr_train_steps_match_autograd_oracle(factor, expect_two_stage=False):
    """factor -1: the reference's default factorisation (factors near sqrt(dim), multiples of 16 on the model sizes); 4 / 8: `network.lokr_factor`
    — lokr_w1 is 4 x 4 / 8 x 8, below the granule of the weight-gradient kernel: graph._skinny_tn's zero-padded path."""
    ref, nat, net = native_pair(factor)
    assert all(bool(m.kron_two_stage) == expect_two_stage for m in net.unet_loras)
    torch.manual_seed(99)
    ref_net = lora_ref.RefLoRANetwork(ref, BIG, network_type="lokr", lokr_factor=factor)
    if factor > 0:
        assert all(tuple(m.lokr_w1.shape) == (factor, factor) for m in net.unet_loras if m.lokr_w1.shape[0] == m.lokr_w1.shape[1]) and any(
            tuple(m.lokr_w1.shape) == (factor, factor) for m in net.unet_loras)
    g = torch.Generator().manual_seed(5)
    for a, b in zip(net.unet_loras, ref_net.unet_loras):
        with torch.no_grad():
            b.lokr_w2.copy_(torch.randn(b.lokr_w2.shape, generator=g) * 0.05)
    ref_net.apply_to()
    net.apply_to(None, nat, False, True)
    net.force_to("cpu", torch.float32)
    with torch.no_grad():
        for a, b in zip(net.unet_loras, ref_net.unet_loras):
            a.lokr_w1.copy_(b.lokr_w1)
            a.lokr_w2.copy_(b.lokr_w2)
    net.refresh_shadows(ref_ops)
    nat.attach_network(net)
    kw = dict(lr=1e-3, weight_decay=0.01, max_grad_norm=0.5)
    oracle = train_ref.RefTrainStep(ref, ref_net, **kw)
    ours = FluxLoRATrainStep(nat, net, ref_ops, **kw)
    CFGI = dict(joint=TINY["joint_attention_dim"], pooled=TINY["pooled_projection_dim"])
    for k in range(2):
        gg = torch.Generator().manual_seed(70 + k)
        lat = torch.randn(2, 16, 8, 4, generator=gg)
        emb = torch.randn(2, 6, CFGI["joint"], generator=gg) * 0.5
        pooled = torch.randn(2, CFGI["pooled"], generator=gg) * 0.5
        noise = torch.randn(2, 16, 8, 4, generator=gg)
        ts = torch.tensor([700.0, 250.0])
        l_ref = oracle.step(lat, emb, pooled, noise, ts)
        l = ours.step(lat, emb, pooled, noise=noise, timesteps=ts)
        assert abs(l.item() - l_ref.item()) <= 1e-4 * abs(l_ref.item()), (k, l.item(), l_ref.item())
    for a, b in zip(net.unet_loras, ref_net.unet_loras):
        assert torch.allclose(a.lokr_w1, b.lokr_w1, rtol=2e-3, atol=2e-6), a.lora_name
        assert torch.allclose(a.lokr_w2, b.lokr_w2, rtol=2e-3, atol=2e-6), a.lora_name


def test_lokr_merge_in_equals_reference_merge_in(gold):
    """FusedLoRANetwork.merge_in(0.7) for LoKr (aitk_kron_merge: W += kron(w1, w2) * scale * merge_weight, on the weight and its
    transposed copy) against base weights merged by the reference's LokrModule.merge_in(0.7); merge_out restores."""
    t, meta = gold
    ref, nat, net = native_pair()
    net.apply_to(None, nat, False, True)
    net.force_to("cpu", torch.float32)
    with torch.no_grad():
        for m in net.unet_loras:
            m.lokr_w2.copy_(t[f"set/{m.lora_name}/w2"])
    net.refresh_shadows(ref_ops)
    nat.attach_network(net)
    nat.prepare()
    keys = [k[len("merged/"):] for k in t if k.startswith("merged/")]
    assert len(keys) == 3
    mods = {m.lora_name: m for m in net.unet_loras}
    before = {k: (mods[k].org_module[0].weight.detach().clone(), mods[k].org_module[0].weight_t.clone()) for k in keys}
    net.merge_in(0.7, ops=ref_ops)
    assert net.is_merged_in
    for k in keys:
        lin = mods[k].org_module[0]
        assert torch.allclose(lin.weight[:24], t[f"merged/{k}"], rtol=1e-5, atol=1e-6), k
        assert torch.allclose(lin.weight_t, lin.weight.t(), rtol=1e-5, atol=1e-6), k  # the dgrad copy is merged consistently
        assert not torch.allclose(lin.weight, before[k][0], atol=1e-4)
    net.merge_out(0.7, ops=ref_ops)
    for k in keys:
        lin = mods[k].org_module[0]
        assert torch.allclose(lin.weight, before[k][0], rtol=1e-5, atol=1e-6) and torch.allclose(lin.weight_t, before[k][1], rtol=1e-5, atol=1e-6)


def test_lokr_factor_modes_token_two_stage_and_refusal():
    """lora.check_kron_fits: every layer size of FLUX / Wan / the UNets runs in the per-token kernel under the reference's default factorisation; an explicit
    small `network.lokr_factor` (the reference UI offers 4 / 8 / 16 / 32) on a wide layer makes W2 a GEMM-sized matrix (factor 4 on 3072 x 3072: 768 x 768)
    and selects the two-stage form; what fits neither is refused where the adapter is attached (construction / apply_to), with the numbers."""
    from types import SimpleNamespace

    from ai_toolkit_amd.adopt import AdoptionError, register_foreign_adapter
    from ai_toolkit_amd.graph import Linear
    from ai_toolkit_amd.lora import LoKrModule, check_kron_fits, kron_lds_bytes

    sizes = ((3072, 3072), (3072, 9216), (3072, 12288), (12288, 3072), (15360, 3072), (3072, 21504), (1536, 8960), (320, 320), (1280, 10240), (2048, 2048))
    for d_in, d_out in sizes:
        im, inn = factorization(d_in)
        ol, ok = factorization(d_out)
        if inn % 8 == 0 and ok % 8 == 0:
            assert check_kron_fits("default", im, inn, ol, ok) == "token", (d_in, d_out)
    assert kron_lds_bytes(48, 64, 48, 64) == 38400  # = 2 * kron_layout(48, 64, 48, 64).total of csrc/kron.hip
    for f in (4, 8, 16, 32):  # the FLUX layer sizes under every factor the reference's UI offers
        for d_in, d_out in sizes[:6]:
            im, inn = factorization(d_in, f)
            ol, ok = factorization(d_out, f)
            assert check_kron_fits(f"f{f}", im, inn, ol, ok) in ("token", "two_stage"), (f, d_in, d_out)
    mk = lambda i, o: Linear(i, o, bias=False, dtype=torch.float32, device="meta")  # noqa: E731
    assert LoKrModule("wide", mk(3072, 3072), lora_dim=BIG, alpha=BIG, factor=4).kron_two_stage
    assert LoKrModule("mlp", mk(3072, 12288), lora_dim=BIG, alpha=BIG, factor=4).kron_two_stage
    assert not LoKrModule("wide", mk(3072, 3072), lora_dim=BIG, alpha=BIG, factor=-1).kron_two_stage
    assert not LoKrModule("narrow", mk(256, 256), lora_dim=BIG, alpha=BIG, factor=4).kron_two_stage
    with pytest.raises(NotImplementedError, match="KiB of LDS"):  # the small-factor mix itself does not fit: 8192-wide sides
        LoKrModule("huge", mk(16384, 16384), lora_dim=BIG, alpha=BIG, factor=2)
    # the same check on a LokrModule the reference built (adoption): the oracle's restatement stands in for it
    huge = mk(16384, 16384)
    m = lora_ref.RefLokrModule("lokr_huge", huge, BIG, BIG, SimpleNamespace(is_lorm=False), factor=2)
    with pytest.raises(AdoptionError, match="KiB of LDS"):
        register_foreign_adapter(huge, m.forward)


def test_lokr_merge_into_a_weight_only_fp8_base_requantises(gold):
    """Round 6 (first refusal lifted): LoKr merge_in over the e4m3 weight-only base = dequantise, W += kron(w1, w2) * scale *
    merge_weight, re-quantise with a fresh per-output-channel scale (the reference's merge ends in the same write-back + requantise for any
    quantised org_module, toolkit/models/lokr.py:261-309 / toolkit/network_mixins.py:452-459); the layer stays quantised, merge_out returns
    to within two roundings of the original."""
    t, meta = gold
    ref, nat, net = native_pair()
    net.apply_to(None, nat, False, True)
    net.force_to("cpu", torch.float32)
    with torch.no_grad():
        for m in net.unet_loras:
            m.lokr_w2.copy_(t[f"set/{m.lora_name}/w2"])
    net.refresh_shadows(ref_ops)
    nat.attach_network(net)
    nat.quantize_base_fp8()
    nat.prepare()
    quantised = [x for x in net.unet_loras if getattr(x.org_module[0], "qweight", None) is not None]
    assert quantised  # the block Linears are quantised (quantize_base_fp8 leaves embedders / adaLN projections in bf16)
    m = quantised[0]
    lin = m.org_module[0]
    deq = lambda: lin.qweight.view(torch.float8_e4m3fn).float() * lin.wscale[:, None]  # noqa: E731
    w0 = deq()
    delta = 0.7 * m.scale * torch.kron(m.lokr_w1.detach().float(), m.composed_w2().float())
    net.merge_in(0.7, ops=ref_ops)
    assert net.is_merged_in and lin.qweight.dtype == torch.uint8
    q_ref = ((w0 + delta) / lin.wscale[:, None]).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(lin.qweight, q_ref) and torch.equal(lin.qweight_t, lin.qweight.t())
    assert not torch.equal(deq(), w0)
    step = (w0 + delta).abs().amax(dim=1, keepdim=True) / 448.0 * 32
    net.merge_out(0.7, ops=ref_ops)
    assert ((deq() - w0).abs() <= step).all()
