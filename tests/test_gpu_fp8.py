"""Weight-only fp8 (e4m3) base operand of the GEMM (BASELINE config 5): in-flight dequant vs the oracle's dequantised-weight matmul."""
import pytest
import torch

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


def quantize_rows(w):
    """per-output-channel symmetric e4m3: scale[n] = amax|w[n,:]| / 448"""
    scale = (w.float().abs().amax(dim=1).clamp_min(1e-12) / 448.0)
    q = (w.float() / scale[:, None]).to(torch.float8_e4m3fn)
    return q, scale


@pytest.mark.parametrize("shape", [(512, 768, 256, 16), (300, 208, 128, 0), (2048, 3072, 3072, 32), (4608, 3072, 3072, 16)])
def test_fp8_base_forward_and_dgrad_layouts(shape):
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    M, N, K, r = shape
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, K, generator=g).to(bf).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).cuda()
    bias = torch.randn(N, generator=g).to(bf).cuda()
    q, scale = quantize_rows(w)
    kw = {}
    if r:
        kw = dict(a2=torch.randn(M, r, generator=g).to(bf).cuda(), b2=(torch.randn(N, r, generator=g) * 0.1).to(bf).cuda())
    out = torch.full((M, N), float("nan"), dtype=bf, device="cuda")
    ref = torch.empty(M, N, device="cuda")
    ops.gemm_nt(x, q.view(torch.uint8), out, bias=bias, b_scale=scale, b_scale_mode=1, **kw)
    ref_ops.gemm_nt(x, q.view(torch.uint8), ref, bias=bias, b_scale=scale, b_scale_mode=1, **kw)
    torch.cuda.synchronize()
    e = ((out.float() - ref).norm() / ref.norm()).item()
    assert e < 5e-3, e
    # dgrad orientation: B = W^T bytes [K, N], scale indexed by the contraction index
    dy = torch.randn(M, N, generator=g).to(bf).cuda()
    qt = q.view(torch.uint8).t().contiguous()
    dx = torch.full((M, K), float("nan"), dtype=bf, device="cuda")
    dref = torch.empty(M, K, device="cuda")
    ops.gemm_nt(dy, qt, dx, b_scale=scale, b_scale_mode=2)
    ref_ops.gemm_nt(dy, qt, dref, b_scale=scale, b_scale_mode=2)
    torch.cuda.synchronize()
    e2 = ((dx.float() - dref).norm() / dref.norm()).item()
    assert e2 < 5e-3, e2


def test_fp8_base_train_step_vs_oracle_with_dequantised_weights():
    """BASELINE config 5 in miniature: e4m3 base weights + fp32 adapter; the oracle multiplies with the dequantised weights."""
    import math

    from ai_toolkit_amd import ops
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from oracle import train_ref
    from tests.test_gpu_e2e import _batch, _build

    import numpy as np

    from oracle import fp8_ref

    ref, ref_net, nat, net = _build(rank=32)
    orig = {n: lin.weight.detach().float().cpu().numpy() for n, lin in nat.named_modules() if lin.__class__.__name__ == "Linear"}
    nat.quantize_base_fp8()
    with torch.no_grad():
        mods = dict(ref.named_modules())
        for n, lin in nat.named_modules():
            if getattr(lin, "qweight", None) is not None:
                # the oracle's weights come from the independent numpy restatement of the quantiser (oracle/fp8_ref.py) applied to the
                # ORIGINAL weights — not from the product's own dequantisation — and the product's codes / scales must equal it
                codes, scale = fp8_ref.quantize_per_channel(orig[n])
                # scales must agree to 1 ulp (device division); bf16 weights over such a scale produce many EXACT rounding ties, so a
                # 1-ulp scale difference may move a tie to the neighbouring code: bounded, and only ever to the adjacent grid point
                got_c, got_s = lin.qweight.cpu().numpy(), lin.wscale.cpu().numpy()
                assert np.allclose(got_s, scale, rtol=2.5e-7, atol=0), n
                bad = got_c != codes
                assert bad.mean() <= 1e-2, (n, bad.mean())
                if bad.any():
                    a, b = fp8_ref.e4m3fn_decode(got_c[bad]), fp8_ref.e4m3fn_decode(codes[bad])
                    assert np.all(np.abs(a - b) <= 0.126 * np.maximum(np.abs(a), np.abs(b))), n  # adjacent grid points (step <= 1/8 of the value)
                mods[n].weight.copy_(torch.from_numpy(fp8_ref.dequantize(codes, scale)).to(torch.bfloat16).float())
    lat, emb, pooled, noise, ts = _batch(2)
    oracle = train_ref.RefTrainStep(ref, ref_net, lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    loss32 = oracle.step(lat.float(), emb.float(), pooled.float(), noise.float(), ts).item()
    g32 = [p.grad.clone() for p in oracle.params]
    ours = FluxLoRATrainStep(nat, net, ops, lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    loss = ours.step(lat, emb, pooled, noise=noise, timesteps=ts).item()
    assert abs(loss - loss32) <= 1e-3 * abs(loss32), (loss, loss32)
    mine = []
    for m in net.unet_loras:
        mine += [m.lora_down.weight.grad, m.lora_up.weight.grad]
    num = sum(((a - b) ** 2).sum().item() for a, b in zip(mine, g32))
    den = sum((b ** 2).sum().item() for b in g32)
    assert math.sqrt(num / den) < 1.5e-2, math.sqrt(num / den)
