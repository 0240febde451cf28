"""Parity under "massive activation" statistics (the reference's quantisers are weight-only, toolkit/util/quantize.py:43-75,
for exactly this reason): a few hidden channels of the residual stream planted at 10^3 x the typical magnitude (tools/gpu_outlier_parity.py),
small FLUX, every mode against its fp32 truth.  Measured round 4 (profiles/r04_outlier_parity.log; full depth 19 + 38 @1024^2 in the same file):
bf16 and the weight-only fp8 base stay at or below the reference arithmetic's own error, W8A8 (per-token e4m3 activations) stays at 1-2e-2 on
adapter gradients — the per-token scale follows the outlier and e4m3's exponent range covers the remaining 10^3 spread, so the mode does not
break; it remains opt-in because it is not the reference's arithmetic."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("factor", [1.0, 1024.0])
def test_small_flux_with_planted_outlier_channels(factor):
    from tests.test_gpu_e2e import _batch
    from tools.gpu_outlier_parity import build_small, run_case

    r = run_case("test", build_small, _batch(2), factor, channels=[7, 100, 301])
    if factor > 1:
        assert r["stream_max_over_median"] > 300
    b = r["bf16"]
    assert b["loss_rel"] < 3e-3 and b["grad_rel"] <= 1.15 * b["ref16_grad_rel"] + 1e-3, b
    assert r["fp8_weight_only"]["loss_rel"] < 3e-3 and r["fp8_weight_only"]["grad_rel"] <= 1.15 * b["ref16_grad_rel"] + 1e-3, r["fp8_weight_only"]
    assert r["fp8_w8a8"]["loss_rel"] < 3e-3 and r["fp8_w8a8"]["grad_rel"] < 3e-2, r["fp8_w8a8"]
    torch.cuda.empty_cache()
