"""Fast epilogue forms of the persistent 8-phase GEMM (gemm8.hip, FE) against its generic epilogue (AITK_GEMM8_FE=0): same arithmetic in the same order, so every
output — C, the saved pre-activation / y — must be bit-identical, for every flag set the graphs use, on ragged shapes (edge waves fall back to the generic form inside
the same launch), under the segmented row map of the joint attention buffers, and in the two-problem launch.  Also the single-pass LN-modulate backward against the
two-phase kernel (same formulas; the row means are summed in a different order: <= 1 bf16 ulp on dx, column sums bit-identical)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


def _run(fe, fn):
    os.environ["AITK_GEMM8_FE"] = str(fe)
    try:
        out = fn()
        torch.cuda.synchronize()
    finally:
        os.environ.pop("AITK_GEMM8_FE", None)
    return out


@pytest.mark.parametrize("M,N,K", [(4608, 3072, 512), (4700, 3080, 336), (9216, 3072, 128)])
def test_fast_epilogue_forms_bit_identical_to_generic(M, N, K):
    from ai_toolkit_amd import ops
    from tools.gpu_gemm8_ev import cases

    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x, w, cs = cases(M, N, K, g)
    for name, mk in cs.items():
        def once():
            o = torch.full((M, N), 0.25, dtype=bf, device="cuda")
            kw = mk(o)
            ops.gemm_nt(x, w, o, **kw)
            return o, kw.get("aux_out")
        ref, got = _run(0, once), _run(1, once)
        assert torch.isfinite(got[0].float()).all(), name
        assert torch.equal(ref[0], got[0]), (name, (ref[0].float() - got[0].float()).abs().max().item())
        assert ref[1] is None or torch.equal(ref[1], got[1]), name


def test_fast_epilogue_under_segmented_row_map_and_grouped_launch():
    from ai_toolkit_amd import ops
    from tools.gpu_gemm8_ev import cases

    g = torch.Generator(device="cuda").manual_seed(11)
    B, Si, St, N, K = 3, 2048, 512, 3072, 256
    for rows, off in ((Si, St), (St, 0)):
        M = B * rows
        x, w, cs = cases(M, N, K, g)
        gate_b = torch.randn(B, N, device="cuda", generator=g).to(bf)
        for name in ("bias+slab", "gate_res+bias+slab", "slab"):
            def once():
                joint = torch.full((B * (Si + St), N), 0.5, dtype=bf, device="cuda")
                kw = cs[name](torch.empty(M, N, dtype=bf, device="cuda"))
                if "gate_rows" in kw:
                    kw["gate_rows"], kw["gate"] = rows, gate_b
                ops.gemm_nt(x, w, joint[off:off + rows], c_seg=(rows, (Si + St) * N), M=M, **kw)
                return joint, kw.get("aux_out")
            ref, got = _run(0, once), _run(1, once)
            assert torch.equal(ref[0], got[0]), (name, rows)
            assert ref[1] is None or torch.equal(ref[1], got[1]), (name, rows)
    xs = [torch.randn(m, K, device="cuda", generator=g).to(bf) for m in (8192, 1024)]
    ws = [(torch.randn(N, K, device="cuda", generator=g) * 0.02).to(bf) for _ in range(2)]
    bs = [torch.randn(N, device="cuda", generator=g).to(bf) for _ in range(2)]

    def grouped():
        outs = [torch.zeros(x_.shape[0], N, dtype=bf, device="cuda") for x_ in xs]
        with ops.recording() as la:
            ops.gemm_nt(xs[0], ws[0], outs[0], bias=bs[0])
        with ops.recording() as lb:
            ops.gemm_nt(xs[1], ws[1], outs[1], bias=bs[1])
        ops.replay_paired(la, lb)
        return outs
    ref, got = _run(0, grouped), _run(1, grouped)
    assert all(torch.equal(a, b) for a, b in zip(ref, got))


@pytest.mark.parametrize("B,S", [(2, 201), (1, 16), (3, 1000)])
def test_single_pass_ln_mod_bwd_matches_two_phase_kernel(B, S):
    from ai_toolkit_amd import ops

    C = 3072
    g = torch.Generator(device="cuda").manual_seed(S)
    M = B * S
    x = torch.randn(M, C, device="cuda", generator=g).to(bf)
    dxn = (torch.randn(M, C, device="cuda", generator=g) * 0.1).to(bf)
    dres = (torch.randn(M, C, device="cuda", generator=g) * 0.1).to(bf)
    scale = (torch.randn(B, C, device="cuda", generator=g) * 0.2).to(bf)
    mean = x.float().mean(1)
    rstd = torch.rsqrt(x.float().var(1, unbiased=False) + 1e-6)
    res = []
    for two_phase in (True, False):
        if two_phase:
            os.environ["AITK_LN_BWD_TWO_PHASE"] = "1"
        try:
            dx = torch.full((M, C), float("nan"), dtype=bf, device="cuda")
            dsh, dsc = torch.empty(B, C, dtype=bf, device="cuda"), torch.empty(B, C, dtype=bf, device="cuda")
            ops.ln_mod_bwd(dxn, x, mean, rstd, scale, dx, B=B, S=S, dres=dres, dshift=dsh, dscale=dsc)
            torch.cuda.synchronize()
        finally:
            os.environ.pop("AITK_LN_BWD_TWO_PHASE", None)
        res.append((dx, dsh, dsc))
    (dx0, sh0, sc0), (dx1, sh1, sc1) = res
    assert torch.isfinite(dx1.float()).all()
    assert torch.equal(sh0, sh1) and torch.equal(sc0, sc1)  # the column sums: same chunks, same order
    # dx = dres + rstd (g - c1 - xhat c2): a last-bit difference in the row means moves the sum by at most an ulp of its larger operand
    d = (dx0.float() - dx1.float()).abs()
    assert d.max().item() <= 2.0 ** -7 * dx0.float().abs().max().item()
    assert (d > 0).float().mean().item() < 0.25
    assert ((dx0.float() - dx1.float()).norm() / dx0.float().norm()).item() < 1e-3
