"""LoRA dropout / rank_dropout / module_dropout (toolkit/network_mixins.py:197-239) through the whole fused FLUX step on the HIP kernels, at rank 16
and at rank 80 (two 64-rank chunk launches per skinny product, each with its own slice of the mask — lifted in round 5).  Both sides draw their
uniforms from the same keyed provider (the reference draws torch.rand in module-call order, which no two graphs share); CPU twin:
tests/test_advice_r2_cpu.py::test_lora_dropout_rank_dropout_and_module_dropout_match_the_oracle."""
import hashlib
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _provider(name, kind, shape, device):
    seed = int(hashlib.sha256(f"{name}/{kind}".encode()).hexdigest()[:8], 16)
    return torch.rand(shape, generator=torch.Generator().manual_seed(seed)).to("cpu" if kind == "module" else device)


@pytest.mark.parametrize("rank", [16, 80])
def test_flux_step_with_dropout_variants_matches_the_oracle(rank):
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from oracle import train_ref
    from oracle.pairs import batch, build

    cfg = dict(dropout=0.1, rank_dropout=0.25, module_dropout=0.2)
    ref, ref_net, nat, net = build(rank, dropout_cfg=cfg, mask_provider=_provider)
    skipped = [m.lora_name for m in net.unet_loras if float(_provider(m.lora_name, "module", (1,), "cpu")) < cfg["module_dropout"]]
    assert 0 < len(skipped) < len(net.unet_loras)
    lat, emb, pooled, noise, ts = batch(2)
    kw = dict(lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    ref_net.train()
    net.train()
    oracle = train_ref.RefTrainStep(ref, ref_net, **kw)
    l32 = oracle.step(lat.float(), emb.float(), pooled.float(), noise.float(), ts).item()
    g32 = [None if p.grad is None else p.grad.clone() for p in oracle.params]
    ours = FluxLoRATrainStep(nat, net, ops, **kw)
    lo = ours.step(lat, emb, pooled, noise=noise, timesteps=ts).item()
    assert abs(lo - l32) <= 2e-3 * abs(l32), (lo, l32)
    mine = []
    for m in net.unet_loras:
        mine += [m.lora_down.weight.grad, m.lora_up.weight.grad]
    num = den = 0.0
    for a, b, m in zip(mine, g32, [m for m in net.unet_loras for _ in (0, 1)]):
        if m.lora_name in skipped:  # module_dropout fired: the adapter saw no gradient on either side
            assert float(a.abs().max()) == 0.0 and (b is None or float(b.abs().max()) == 0.0), m.lora_name
            continue
        num += ((a - b) ** 2).sum().item()
        den += (b ** 2).sum().item()
    err = math.sqrt(num / den)
    print(f"dropout variants rank {rank}: loss ours {lo:.6f} fp32 {l32:.6f}; adapter-gradient rel err {err:.3e}; {len(skipped)} modules dropped")
    assert err < 1.5e-2, err
    # eval mode: no masks, the plain step
    ref_net.eval()
    net.eval()
    l32e = oracle.step(lat.float(), emb.float(), pooled.float(), noise.float(), ts).item()
    loe = ours.step(lat, emb, pooled, noise=noise, timesteps=ts).item()
    assert abs(loe - l32e) <= 1e-3 * abs(l32e) and abs(l32e - l32) > 1e-6
