"""Stream-K tail of the persistent 8-phase GEMM (gemm8.hip, opt-in through AITK_GEMM8_SK): same results as the data-parallel kernel up to the
order of one fp32 sum per split tile, bit-identical from run to run (the flag protocol has no race), across epilogue forms, grouped launches,
ragged rows, a K tail and the emitting epilogue.  The schedule itself is checked on the host in tests/test_capi_symbols.py."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def sk_env():
    old = os.environ.get("AITK_GEMM8_SK")
    yield lambda mode: os.environ.__setitem__("AITK_GEMM8_SK", str(mode))
    if old is None:
        os.environ.pop("AITK_GEMM8_SK", None)
    else:
        os.environ["AITK_GEMM8_SK"] = old


def _tool():
    import importlib

    return importlib.import_module("tools.gpu_gemm8_sk")


@pytest.mark.parametrize("name,M,N,K,r,flag,emit", [
    ("b1_n3072", 4608, 3072, 3072, 16, "0", False),
    ("b1_gelu_emit", 4608, 12288, 3072, 16, "GELU", True),
    ("b1_k12288_gate", 4608, 3072, 12288, 16, "GATE_RES", False),
    ("b1_dgelu", 4608, 12288, 3072, 0, "DGELU", False),
    ("b2_acc_r48", 9216, 3072, 3072, 48, "ACCUM", False),
    ("ragged_rows_k_tail", 4500, 3072, 3088, 16, "0", False),
])
def test_stream_k_tail_equals_the_data_parallel_kernel_and_is_deterministic(sk_env, name, M, N, K, r, flag, emit):
    from ai_toolkit_amd import ops

    t = _tool()
    flags = 0 if flag == "0" else getattr(ops, "EPI_" + flag)
    a, b, kw, c0 = t.operands(M, N, K, r, flags, 0, emit)
    sk_env(0)
    base = t.run(a, b, kw, c0, flags)
    sk_env(2)
    first = t.run(a, b, kw, c0, flags)
    for _ in range(4):
        again = t.run(a, b, kw, c0, flags)
        assert all(torch.equal(x, y) for x, y in zip(first, again)), name
    for x, y in zip(first, base):
        assert not torch.isnan(x.float()).any()
        assert t.rel(x, y) < 2e-4, name  # one fp32 sum per element in another order, then the bf16 rounding: a few elements move by one ulp
    assert float((first[0] != base[0]).float().mean()) < 5e-3


def test_stream_k_tail_in_a_grouped_launch(sk_env):
    from ai_toolkit_amd import ops

    t = _tool()
    N, K = 3072, 3072
    ops1, ops2 = t.operands(4096, N, K, 16, 0, 1), t.operands(512, N, K, 16, 0, 2)

    def go():
        outs, lists = [], []
        for (a, b, kw, _) in (ops1, ops2):
            out = torch.full((a.shape[0], N), float("nan"), dtype=torch.bfloat16, device="cuda")
            with ops.recording() as rec:
                ops.gemm_nt(a, b, out, stage_mode=4, **kw)
            lists.append(rec)
            outs.append(out)
        ops.replay_paired(lists[0], lists[1])
        torch.cuda.synchronize()
        return outs

    sk_env(0)
    base = go()
    sk_env(2)
    first = go()
    again = go()
    for x, y, z in zip(first, base, again):
        assert torch.equal(x, z) and t.rel(x, y) < 2e-4
