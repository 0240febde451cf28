"""Round-3 kernel modes on the GPU, through the C ABI, against the oracle table / the reference-made goldens:
loss_type mae / pseudo_huber in aitk_mse_loss_grad, EMA use_feedback / param_multiplier in aitk_adamw_ema_step."""
import os

import pytest
import torch
from safetensors.torch import load_file

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


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


@pytest.mark.parametrize("B,H,S,Skv,d,pad", [(2, 10, 200, 0, 64, 0), (1, 20, 1024, 77, 64, 0), (2, 3, 130, 77, 64, 64), (1, 2, 96, 0, 96, 0),
                                              (1, 3, 257, 100, 96, 32)])
def test_attention_native_head_layout_vs_oracle(B, H, S, Skv, d, pad):
    """AitkAttnArgs.hstride: heads of 64 / 96 columns read and written in the projections' own [tokens, H*d] layout (SDXL's 64-wide
    heads: no padded copies).  Everything the kernels write is compared, plus `pad` guard columns behind the last head (row stride
    H*d + pad) that must come back untouched — the padded-layout kernels zero-fill up to 128 columns per head."""
    import math

    from ai_toolkit_amd import ops
    from oracle import ref_ops

    g = torch.Generator().manual_seed(S + 3 * d)
    n = Skv or S
    ld = H * d + pad

    def mk(rows):
        return torch.randn(rows, ld, generator=g).to(torch.bfloat16).cuda()

    q, k, v, do = mk(B * S), mk(B * n), mk(B * n), mk(B * S)
    sc = 1 / math.sqrt(d)
    outs = []
    for o_ in (ops, ref_ops):
        o = torch.full_like(q, 7.0)
        lse = torch.empty(B, H, S, dtype=torch.float32, device="cuda")
        o_.attn_fwd(q, k, v, o, lse, B=B, H=H, S=S, scale=sc, Skv=Skv, dv=d, hstride=d)
        dq, dk, dv_ = torch.full_like(q, 7.0), torch.full_like(k, 7.0), torch.full_like(v, 7.0)
        o_.attn_bwd(q, k, v, o, lse, do, dq, dk, dv_, B=B, H=H, S=S, scale=sc, Skv=Skv, dvalid=d, hstride=d)
        outs.append((o, dq, dk, dv_, lse))
    for a, b, nm in zip(outs[0], outs[1], ("o", "dq", "dk", "dv", "lse")):
        if nm == "lse":
            assert (a - b).abs().max().item() < 2e-2, nm
            continue
        assert torch.equal(a[:, H * d:], torch.full_like(a[:, H * d:], 7.0)), nm  # guard columns untouched
        assert _rel(a[:, :H * d], b[:, :H * d]) < 8e-3, (nm, _rel(a[:, :H * d], b[:, :H * d]))
        for hh in (0, H - 1):  # per head: a head that picked up its neighbour's columns would pass the global norm at H = 20
            sl = slice(hh * d, (hh + 1) * d)
            assert _rel(a[:, sl], b[:, sl]) < 1e-2, (nm, hh)


@pytest.mark.parametrize("R", [80, 128])
def test_skinny_kernels_above_64_ranks_go_out_in_chunks_of_one_slab(R):
    """aitk_lora_down / aitk_lora_wgrad contract up to 64 ranks per launch; ranks 80 / 128 (one rank block = one [hi | lo | hi] slab of 3 R
    columns) are 64-rank chunks of that slab: same results as the oracle on the whole rank."""
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    g = torch.Generator().manual_seed(R)
    M, K, L = 1000, 256, 192
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    p32 = torch.randn(R, K, generator=g) * K ** -0.5
    hi = p32.to(torch.bfloat16)
    lo = (p32 - hi.float()).to(torch.bfloat16)
    hi, lo = hi.cuda(), lo.cuda()
    mult = torch.tensor([0.5, 1.5], device="cuda")
    outs = []
    for tb in (ops, ref_ops):
        T = torch.full((M, 3 * R), float("nan"), dtype=torch.bfloat16, device="cuda")
        tb.lora_down(x, hi, T, scale=0.7, mult=mult, rows_per_batch=500, M=M, p_lo=lo, split=R)
        Tp = torch.full((M, R), float("nan"), dtype=torch.bfloat16, device="cuda")
        tb.lora_down(x, hi, Tp, scale=0.7, M=M)
        outs.append((T, Tp))
    (T, Tp), (Tr, Tpr) = outs
    assert torch.equal(T[:, :R], T[:, 2 * R:])
    assert _rel(T[:, :R].float() + T[:, R:2 * R].float(), Tr[:, :R].float() + Tr[:, R:2 * R].float()) < 5e-5
    assert _rel(Tp, Tpr) < 4e-3
    gy = torch.randn(M, L, generator=g).to(torch.bfloat16).cuda()
    for transpose in (False, True):
        shape = (L, R) if transpose else (R, L)
        a, b = torch.ones(shape, device="cuda"), torch.ones(shape, device="cuda")
        ops.lora_wgrad(Tr, gy, a, transpose_out=transpose, accumulate=True, M=M, split=R)
        ref_ops.lora_wgrad(Tr, gy, b, transpose_out=transpose, accumulate=True, M=M, split=R)
        assert _rel(a, b) < 2e-4, (transpose, _rel(a, b))
    a, b = torch.zeros(R, L, device="cuda"), torch.zeros(R, L, device="cuda")
    ops.lora_wgrad(Tpr, gy, a, M=M)
    ref_ops.lora_wgrad(Tpr, gy, b, M=M)
    assert _rel(a, b) < 2e-4
