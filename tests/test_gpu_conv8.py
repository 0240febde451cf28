"""Implicit-GEMM 3x3 convolution on the persistent 8-phase kernel (gemm8.hip conv mode; VERDICT r3 item 4, north_star "VAE conv as implicit-GEMM on
MFMA"): against the oracle op (F.conv2d in fp32) and against the 2-barrier kernel it replaces for big problems, over the geometries the UNet /
VAE graphs produce — stride 1 and 2, the (1, 1) and the asymmetric (0, 0) padding of diffusers' Downsample2D, borders on every side, ragged M,
channel counts that put several K-tiles inside one tap, bias / residual-add / accumulate epilogues and a conv adapter's lora_up K-slab
(toolkit/lora_special.py:95-104).  The reference runs nn.Conv2d (stable_diffusion_model.py:2533-2575 for the VAE, the UNet's ResnetBlock2D)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


def _rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


CASES = [
    # B, H, W, Cin, Cout, stride, pad, Ho, Wo
    (2, 32, 32, 128, 256, 1, 1, 32, 32),
    (3, 19, 27, 64, 256, 1, 1, 19, 27),      # ragged M (1539 rows), narrow image: every tile row block crosses image rows
    (1, 64, 64, 320, 320, 1, 1, 64, 64),     # SDXL level 0: 5 K-tiles per tap, N = 320 (ragged last N tile)
    (2, 32, 32, 640, 1280, 1, 1, 32, 32),
    (2, 33, 33, 128, 256, 2, 0, 16, 16),     # Downsample2D: stride 2 with the (0, 1, 0, 1) padding = pad_t = pad_l = 0
    (2, 32, 32, 256, 512, 2, 1, 16, 16),     # stride 2, symmetric padding
    (1, 128, 128, 128, 128, 1, 1, 128, 128), # VAE level 0 shape class (N = 128 < 256: stays on the 2-barrier kernel unless forced)
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,pad,Ho,Wo", CASES)
def test_conv8_vs_oracle_and_two_barrier_kernel(B, H, W, Cin, Cout, stride, pad, Ho, Wo):
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    g = torch.Generator(device="cuda").manual_seed(H * W + Cin)
    x = torch.randn(B * H * W, Cin, device="cuda", generator=g).to(bf)
    w = (torch.randn(Cout, 9 * Cin, device="cuda", generator=g) * (9 * Cin) ** -0.5).to(bf)
    bias = (torch.randn(Cout, device="cuda", generator=g) * 0.1).to(bf)
    M = B * Ho * Wo
    kw = dict(B=B, H=H, W=W, stride=stride, pad_t=pad, pad_l=pad, Ho=Ho, Wo=Wo)
    ref = torch.empty(M, Cout, dtype=bf, device="cuda")
    ref_ops.conv3x3(x, w, ref, bias=bias, **kw)
    out8 = torch.full((M, Cout), float("nan"), dtype=bf, device="cuda")
    ops.conv3x3(x, w, out8, bias=bias, stage_mode=4, **kw)
    out2 = torch.full((M, Cout), float("nan"), dtype=bf, device="cuda")
    ops.conv3x3(x, w, out2, bias=bias, stage_mode=5, **kw)
    torch.cuda.synchronize()
    assert torch.isfinite(out8.float()).all()
    assert _rel(out8, ref) < 3e-3, _rel(out8, ref)   # one bf16 rounding of the output
    # same products in the same K order, fp32 accumulation: the two kernels agree to the last bit or to one rounding of a few entries
    assert _rel(out8, out2) < 1e-3, _rel(out8, out2)
    print(f"conv8 {B}x{H}x{W}x{Cin}->{Cout} s{stride}p{pad}: vs oracle {_rel(out8, ref):.2e}, bitwise equal to the 2-barrier kernel: {torch.equal(out8, out2)}")


def test_conv8_epilogues_and_adapter_slab():
    """residual add (ResnetBlock2D / VAE), accumulate (the dx += slab convolution of a conv adapter's data gradient) and the lora_up K-slab"""
    from ai_toolkit_amd import _capi, ops
    from oracle import ref_ops

    B, H, W, Cin, Cout = 2, 40, 24, 128, 320
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(B * H * W, Cin, device="cuda", generator=g).to(bf)
    w = (torch.randn(Cout, 9 * Cin, device="cuda", generator=g) * (9 * Cin) ** -0.5).to(bf)
    bias = (torch.randn(Cout, device="cuda", generator=g) * 0.1).to(bf)
    res = torch.randn(B * H * W, Cout, device="cuda", generator=g).to(bf)
    a2 = (torch.randn(B * H * W, 48, device="cuda", generator=g) * 0.3).to(bf)
    b2 = (torch.randn(Cout, 48, device="cuda", generator=g) * 0.05).to(bf)
    kw = dict(B=B, H=H, W=W)
    for name, extra in (("add_aux", dict(flags=_capi.EPI_ADD_AUX, aux_in=res)), ("slab", dict(a2=a2, b2=b2)), ("accum", dict(flags=_capi.EPI_ACCUM))):
        outs = []
        for o_, sm in ((ops, dict(stage_mode=4)), (ref_ops, {})):
            out = res.clone() if name == "accum" else torch.empty(B * H * W, Cout, dtype=bf, device="cuda")
            o_.conv3x3(x, w, out, bias=bias, **kw, **extra, **sm)
            outs.append(out)
        assert _rel(outs[0], outs[1]) < 3e-3, (name, _rel(outs[0], outs[1]))


def test_big_convolutions_take_the_8phase_kernel_by_default(monkeypatch):
    """routing: the default (stage_mode 1) result of a big convolution is bit-identical to the forced 8-phase launch, and AITK_CONV8=0's
    stage_mode 5 to the 2-barrier kernel — so the default path IS the new kernel where its contract holds"""
    from ai_toolkit_amd import ops

    B, H, W, Cin, Cout = 4, 64, 64, 256, 512   # 16384 x 512: 128 tiles of 256^2
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B * H * W, Cin, device="cuda", generator=g).to(bf)
    w = (torch.randn(Cout, 9 * Cin, device="cuda", generator=g) * (9 * Cin) ** -0.5).to(bf)
    outs = {}
    for sm in (None, 4, 5):
        o = torch.empty(B * H * W, Cout, dtype=bf, device="cuda")
        ops.conv3x3(x, w, o, B=B, H=H, W=W, stage_mode=sm)
        outs[sm] = o
    assert torch.equal(outs[None], outs[4])
    assert _rel(outs[4], outs[5]) < 1e-3
