"""aitk_lora_refresh_shadows: the tiled form (plain LoRA matrices staged through LDS so that the transposed layouts are written coalesced) puts the same
bf16 values in the same places as the element-per-thread form — every byte of the shadow arena, for ranks 16 / 32 / 48, same-input groups (shared
[in, 3 R] blocks with a row stride) and widths that are not multiples of the 64-wide tile."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _refresh(net, ops, tiled):
    old = os.environ.get("AITK_REFRESH_TILED")
    os.environ["AITK_REFRESH_TILED"] = "1" if tiled else "0"
    try:
        net.arena_shadow.fill_(0)
        net.refresh_shadows(ops)
        torch.cuda.synchronize()
        return net.arena_shadow.clone()
    finally:
        if old is None:
            os.environ.pop("AITK_REFRESH_TILED", None)
        else:
            os.environ["AITK_REFRESH_TILED"] = old


@pytest.mark.parametrize("rank", [16, 32, 48])
def test_tiled_refresh_writes_the_same_shadow_arena(rank):
    from ai_toolkit_amd import ops
    from oracle.pairs import build

    _, _, nat, net = build(rank=rank)
    g = torch.Generator(device="cuda").manual_seed(rank)
    net.arena_p.copy_(torch.randn(net.arena_p.shape, generator=g, device="cuda") * 0.05)
    a = _refresh(net, ops, tiled=False)
    b = _refresh(net, ops, tiled=True)
    assert a.view(torch.int16).ne(0).any()
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))


def test_tiled_refresh_on_ragged_widths():
    """One A and one B matrix of a width that is no multiple of 64 (and fewer columns than one tile), straight through the C ABI."""
    from ai_toolkit_amd import ops

    dev = "cuda"
    for R, width in ((16, 200), (24, 40), (64, 130)):
        nA, nB = R * width, width * R
        arena = torch.randn(nA + nB, device=dev)
        # shadow layout: A: d0 [R, in] hi, d1 [R, in] lo, d2 [in, 3R];  B: d0 [out, 3R], d1 [R, out] hi^T, d2 [R, out] lo^T
        offs, tot = [], 0
        for n in (nA, nA, 3 * nA, 3 * nB, nB, nB):
            offs.append(tot)
            tot += n + 8
        entries = [(0, R, width, 1, offs[0], offs[1], offs[2], 0), (nA, width, R, 2, offs[3], offs[4], offs[5], 0)]  # (src_off, rows, cols, kind, d0, d1, d2, aux)
        table = ops.make_shadow_table(entries, torch.device(dev))
        outs = []
        for tiled in (False, True):
            os.environ["AITK_REFRESH_TILED"] = "1" if tiled else "0"
            try:
                sh = torch.zeros(tot, dtype=torch.bfloat16, device=dev)
                ops.refresh_shadows(arena, sh, table)
                torch.cuda.synchronize()
                outs.append(sh)
            finally:
                os.environ.pop("AITK_REFRESH_TILED", None)
        assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)), (R, width)
        A = arena[:nA].view(R, width)
        hi = A.to(torch.bfloat16)
        lo = (A - hi.float()).to(torch.bfloat16)
        assert torch.equal(outs[1][offs[0]:offs[0] + nA].view(R, width), hi)
        assert torch.equal(outs[1][offs[2]:offs[2] + 3 * nA].view(width, 3 * R), torch.cat([hi.t(), hi.t(), lo.t()], dim=1))
        Bm = arena[nA:].view(width, R)
        bh = Bm.to(torch.bfloat16)
        bl = (Bm - bh.float()).to(torch.bfloat16)
        assert torch.equal(outs[1][offs[3]:offs[3] + 3 * nB].view(width, 3 * R), torch.cat([bh, bh, bl], dim=1))
        assert torch.equal(outs[1][offs[4]:offs[4] + nB].view(R, width), bh.t())
        assert torch.equal(outs[1][offs[5]:offs[5] + nB].view(R, width), bl.t())
