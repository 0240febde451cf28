"""`recompute_gelu` on the GPU: aitk_lora_wgrad2 — the lora_down gradient from a two-part operand
[g | gelu(pre-activation)] — against the oracle op and against aitk_lora_wgrad on the materialised operand at the FLUX shapes
(ff.net.2: 12288 GELU columns; single-block proj_out: 3072 attention + 12288 GELU columns; ranks 16 and 32), and the train step with
the flag on == the flag off, bit for bit, at full width.  The reference's own memory lever is full block recompute
(toolkit/config_modules.py:413, SDTrainer.py:2226-2238); this one drops only the tensors whose single backward reader is the
lora_down gradient (network_mixins.py:309-321 keeps the layer input for exactly that product)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


def _rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


@pytest.mark.parametrize("M,R,d_o,d_u", [(32256, 16, 0, 12288), (32256, 16, 3072, 12288), (9216, 32, 3072, 12288), (4608 + 37, 32, 0, 12288)])
def test_lora_wgrad2_flux_shapes_vs_oracle_and_materialised(M, R, d_o, d_u):
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    g = torch.Generator(device="cuda").manual_seed(M + R)
    S = (torch.randn(M, 3 * R, device="cuda", generator=g) * 0.3).to(bf)
    S[:, 2 * R:] = S[:, :R]  # [hi | lo | hi] slab
    u = torch.randn(M, d_u, device="cuda", generator=g).to(bf)
    o = torch.randn(M, d_o, device="cuda", generator=g).to(bf) if d_o else None
    L = d_o + d_u
    a, b, c = (torch.zeros(R, L, device="cuda") for _ in range(3))
    ops.lora_wgrad(S, o, a, M=M, split=R, g2=u, g2_act="gelu")
    ref_ops.lora_wgrad(S, o, b, M=M, split=R, g2=u, g2_act="gelu")
    h = torch.nn.functional.gelu(u.float(), approximate="tanh").to(bf)
    full = h if o is None else torch.cat((o, h), 1).contiguous()
    ops.lora_wgrad(S, full, c, M=M, split=R)
    assert _rel(a, b) < 2e-4, _rel(a, b)      # fp32 accumulation order only
    assert _rel(a, c) < 2e-3, _rel(a, c)      # the kernel's exp/rcp GELU vs torch's tanh: <= 1 bf16 ulp on a few operand entries
    # accumulate mode adds onto what is there
    ops.lora_wgrad(S, o, a, M=M, split=R, g2=u, g2_act="gelu", accumulate=True)
    assert _rel(a, 2 * b) < 2e-4


def test_recompute_gelu_step_bit_identical_full_width():
    """One double + two single blocks at the real width (d = 3072, 24 heads, 4096 + 512 tokens, r16): loss and the whole gradient
    arena with `recompute_gelu` equal the default graph's bit for bit, and the peak memory of the step is lower."""
    from bench import build_flux, make_batch
    from ai_toolkit_amd.trainer import FluxLoRATrainStep

    dev = torch.device("cuda", 0)
    res = {}
    for flag in (False, True):
        model, net, ops = build_flux(dev, rank=16, num_layers=1, num_single=2, ema=False)
        model.recompute_gelu = flag
        st = FluxLoRATrainStep(model, net, ops, lr=0.0, weight_decay=0.0, max_grad_norm=0.0, timestep_type="linear", seed=5)
        lat, emb, pooled = make_batch(dev, 2, seed=3)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        loss = st.step(lat, emb, pooled).item()
        torch.cuda.synchronize()
        res[flag] = (loss, net.arena_g.clone(), torch.cuda.max_memory_allocated() - base)
        del model, net, st
        torch.cuda.empty_cache()
    assert res[False][0] == res[True][0]
    assert torch.equal(res[False][1], res[True][1])
    assert float(res[True][1].abs().max()) > 0
    # 2 images x (2 streams of ff GELU outputs 4608 x 12288 + 2 single blocks x 4608 x 12288) x 2 B = 0.45 GB less at least
    assert res[True][2] < res[False][2] - 300 * 2 ** 20, (res[True][2], res[False][2])
