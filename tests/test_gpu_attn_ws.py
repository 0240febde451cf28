"""Wave-specialised dK / dV kernel (attn_bwd_dkdv_ws_kernel: 8 waves per workgroup, producer waves compute S / dP + softmax, consumer waves the
dV / dK products; VERDICT r3 item 2 "two waves per SIMD"): same ownership and product order as the pipelined kernel, so the gradients must be
bit-identical to it, at the FLUX shape and at ragged sequence lengths (tiles past the end, odd tile counts, one tile)."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


def _bwd(q, k, v, o, lse, do, B, H, S, ws):
    from ai_toolkit_amd import ops

    os.environ["AITK_ATTN_DKDV_WS"] = "1" if ws else "0"
    try:
        dq, dk, dv = (torch.full_like(q, float("nan")) for _ in range(3))
        ops.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, B=B, H=H, S=S, scale=1 / math.sqrt(128))
        torch.cuda.synchronize()
    finally:
        os.environ.pop("AITK_ATTN_DKDV_WS", None)
    return dq, dk, dv


@pytest.mark.parametrize("B,H,S", [(1, 24, 4608), (2, 3, 200), (1, 2, 64), (1, 2, 33), (2, 2, 129), (1, 4, 1000)])
def test_ws_dkdv_bit_identical_to_pipelined_kernel_and_close_to_oracle(B, H, S):
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    g = torch.Generator(device="cuda").manual_seed(S + H)
    HD = H * 128
    q, k, v = ((torch.randn(B * S, HD, device="cuda", generator=g) * 0.7).to(bf) for _ in range(3))
    do = (torch.randn(B * S, HD, device="cuda", generator=g) * 0.5).to(bf)
    o = torch.empty_like(q)
    lse = torch.empty(B, H, S, device="cuda")
    sc = 1 / math.sqrt(128)
    ops.attn_fwd(q, k, v, o, lse, B=B, H=H, S=S, scale=sc)
    a = _bwd(q, k, v, o, lse, do, B, H, S, ws=False)
    b = _bwd(q, k, v, o, lse, do, B, H, S, ws=True)
    for x, y, nm in zip(a, b, ("dq", "dk", "dv")):
        assert torch.isfinite(y.float()).all(), nm
        assert torch.equal(x, y), (nm, (x.float() - y.float()).abs().max().item())
    if S <= 1000:  # the oracle materialises S x S scores
        ro, rl = torch.empty_like(q), torch.empty_like(lse)
        ref_ops.attn_fwd(q, k, v, ro, rl, B=B, H=H, S=S, scale=sc)
        rq, rk, rv = (torch.empty_like(q) for _ in range(3))
        ref_ops.attn_bwd(q, k, v, ro, rl, do, rq, rk, rv, B=B, H=H, S=S, scale=sc)
        for x, y in zip(b[1:], (rk, rv)):
            assert ((x.float() - y.float()).norm() / y.float().norm()).item() < 3.5e-3


@pytest.mark.parametrize("B,H,S,Skv", [(1, 24, 4608, 0), (2, 3, 1024, 0), (1, 2, 128, 0), (2, 2, 64, 128), (2, 3, 256, 512), (2, 3, 200, 0), (1, 2, 192, 320)])
def test_five_matmul_backward_is_bit_identical_to_the_recomputing_backward(B, H, S, Skv):
    """Round 6: the dK/dV pass emits its bf16 dS (accumulator-native 2-KiB blocks) and dQ = dS K runs as a product of its own
    (attn_bwd_dq_ds_kernel: blocks transposed by LDS-DMA chunk reordering + tr16 reads) instead of recomputing S and dP in a second pass.
    Same dS bits, same accumulation order -> dQ, dK, dV must be the recomputing backward's BIT FOR BIT: at the FLUX shape, at small whole-tile
    shapes, with separate key / value lengths (Wan cross-attention), and at shapes the mode does not cover (ragged lengths: it must fall back)."""
    from ai_toolkit_amd import ops

    g = torch.Generator(device="cuda").manual_seed(S + H + Skv)
    HD = H * 128
    kv = Skv or S
    q = (torch.randn(B * S, HD, device="cuda", generator=g) * 0.7).to(bf)
    k, v = ((torch.randn(B * kv, HD, device="cuda", generator=g) * 0.7).to(bf) for _ in range(2))
    do = (torch.randn(B * S, HD, device="cuda", generator=g) * 0.5).to(bf)
    o = torch.empty_like(q)
    lse = torch.empty(B, H, S, device="cuda")
    sc = 1 / math.sqrt(128)
    ops.attn_fwd(q, k, v, o, lse, B=B, H=H, S=S, scale=sc, Skv=Skv)
    outs = []
    for mode in (0, 1):
        dq = torch.full_like(q, float("nan"))
        dk, dv = torch.full_like(k, float("nan")), torch.full_like(k, float("nan"))
        ds = torch.full((B * H * ((S + 31) // 32 * 32) * ((kv + 31) // 32 * 32),), float("nan"), dtype=bf, device="cuda")
        # ds_mode 4 = "off, whatever the default": an explicit recomputing backward
        ops.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, B=B, H=H, S=S, scale=sc, Skv=Skv, ds=ds, ds_mode=1 if mode else 4)
        torch.cuda.synchronize()
        outs.append((dq, dk, dv, ds))
    for x, y, nm in zip(outs[0][:3], outs[1][:3], ("dq", "dk", "dv")):
        assert torch.isfinite(y.float()).all(), nm
        assert torch.equal(x, y), (nm, (x.float() - y.float()).abs().max().item())
    eligible = ops.attn_ds_eligible(S, kv)
    assert eligible == (S % 64 == 0 and kv % 128 == 0)
    # the scratch was written exactly when the mode applies (every element: the blocks tile [S, Skv] completely)
    assert bool(torch.isfinite(outs[1][3].float()).all()) == eligible
    assert torch.isnan(outs[0][3].float()).all()
