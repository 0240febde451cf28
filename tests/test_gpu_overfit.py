"""The fused step trains: a fixed batch, a fixed noise / timestep draw, AdamW on the adapters only — the flow-matching loss must fall, on the
HIP path as on the oracle (same hyper-parameters, same inputs), and the two loss curves must stay together."""
import math

import pytest

pytestmark = pytest.mark.gpu


def test_lora_overfits_a_fixed_batch_like_the_oracle():
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from oracle import train_ref
    from tests.test_gpu_e2e import _batch, _build

    ref, ref_net, nat, net = _build()
    kw = dict(lr=2e-3, weight_decay=0.0, max_grad_norm=1.0)
    oracle = train_ref.RefTrainStep(ref, ref_net, **kw)
    ours = FluxLoRATrainStep(nat, net, ops, **kw)
    lat, emb, pooled, noise, ts = _batch(2, seed=77)
    lo, lr_ = [], []
    for _ in range(40):
        lr_.append(oracle.step(lat.float(), emb.float(), pooled.float(), noise.float(), ts).item())
        lo.append(ours.step(lat, emb, pooled, noise=noise, timesteps=ts).item())
    print("overfit: ours", [round(v, 4) for v in lo[::8]], "oracle", [round(v, 4) for v in lr_[::8]])
    assert all(math.isfinite(v) for v in lo)
    assert lo[-1] < 0.97 * lo[0] and lr_[-1] < 0.97 * lr_[0], (lo[0], lo[-1], lr_[0], lr_[-1])  # both fall
    assert lo[-1] <= min(lo[:5])  # and ours keeps falling after the first steps
    # the curves stay together: same drop to within a tenth of it
    assert abs((lo[0] - lo[-1]) - (lr_[0] - lr_[-1])) <= 0.1 * (lr_[0] - lr_[-1]) + 2e-3 * lr_[0], (lo[0] - lo[-1], lr_[0] - lr_[-1])
