"""Round-3 kernel modes on the GPU, through the C ABI, against the oracle table / the reference-made goldens:
loss_type mae / pseudo_huber in aitk_mse_loss_grad, EMA use_feedback / param_multiplier in aitk_adamw_ema_step."""
import os

import pytest
import torch
from safetensors.torch import load_file

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("loss_type", ["mse", "mae", "pseudo_huber"])
@pytest.mark.parametrize("masked", [False, True])
def test_loss_type_kernel_vs_oracle(loss_type, masked):
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    g = torch.Generator().manual_seed(9)
    B, T, F = 3, 520, 64
    pred = (torch.randn(B, T, F, generator=g) * 0.2).to(torch.bfloat16).cuda()
    tgt = (torch.randn(B, T, F, generator=g) * 0.2).to(torch.bfloat16).cuda()
    tgt[0, 0, :16] = pred[0, 0, :16]  # exact zeros of the difference
    mask = torch.rand(B, T, 4, generator=g).cuda() if masked else None
    w = torch.tensor([1.0, 0.25, 3.0]).cuda()
    outs = []
    for o_ in (ops, ref_ops):
        dp = torch.empty_like(pred)
        lps, loss = torch.zeros(B, device="cuda"), torch.zeros(1, device="cuda")
        o_.mse_loss_grad(pred, tgt, dp, lps, loss, weight=w, mask=mask, loss_type=loss_type)
        outs.append((dp.float(), lps, loss))
    assert torch.allclose(outs[0][1], outs[1][1], rtol=2e-5) and torch.allclose(outs[0][2], outs[1][2], rtol=2e-5)
    assert (outs[0][0] - outs[1][0]).abs().max() <= 2 ** -8 * outs[1][0].abs().max()
    assert float(outs[0][0][0, 0, :16].abs().max()) == 0.0


@pytest.mark.parametrize("tag,kw", [("fb", dict(ema_feedback=10.0, param_multiplier=0.999)), ("pm", dict(param_multiplier=1.002))])
def test_ema_option_kernel_vs_reference_class_golden(tag, kw):
    from ai_toolkit_amd import ops

    t = load_file(os.path.join(G, "ema_options.safetensors"))
    p, ema = t[f"{tag}/p0"].cuda(), t[f"{tag}/p0"].cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for k in range(3):
        ops.adamw_ema_step(p, t[f"{tag}/grads"][k].cuda().contiguous(), m, v, lr=3e-3, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=0.01,
                           step=k + 1, max_norm=1.0, ema=ema, ema_decay=0.9, **kw)
    assert torch.allclose(p.cpu(), t[f"{tag}/p3"], rtol=2e-5, atol=2e-7), (p.cpu() - t[f"{tag}/p3"]).abs().max()
    assert torch.allclose(ema.cpu(), t[f"{tag}/ema3"], rtol=2e-5, atol=2e-7)


@pytest.mark.parametrize("B,Hs,Ws", [(2, 37, 52), (1, 515, 509), (3, 64, 71)])
def test_bilinear_resize_to_nhwc8_kernel_vs_torch_interpolate(B, Hs, Ws):
    """aitk_image_resize_to_nhwc8 against F.interpolate(bilinear, align_corners=False) in bf16 on the same GPU (what Wan21.encode_images runs,
    toolkit/models/wan21/wan21.py:652-657): equal to one bf16 rounding of the result."""
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    g = torch.Generator().manual_seed(Hs)
    img = (torch.rand(B, 3, Hs, Ws, generator=g) * 2 - 1).cuda()
    Hd, Wd = Hs // 8 * 8, Ws // 8 * 8
    a = torch.full((B * Hd * Wd, 8), 7.0, dtype=torch.bfloat16, device="cuda")
    b = torch.empty_like(a)
    ops.image_resize_to_nhwc8(img, a, Hd=Hd, Wd=Wd)
    ref_ops.image_resize_to_nhwc8(img, b, Hd=Hd, Wd=Wd)
    assert float(a[:, 3:].abs().max()) == 0.0
    d = (a.float() - b.float()).abs()
    assert d.max().item() <= 2 ** -7, d.max().item()         # values in [-1, 1]: at most one bf16 ulp of the largest magnitude
    assert (d > 0).float().mean().item() < 0.02              # and almost everywhere identical
