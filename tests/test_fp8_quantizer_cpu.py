"""Weight-only fp8 (e4m3fn, per-output-channel absmax / 448) quantiser of BASELINE config 5: the product quantiser
(ai_toolkit_amd.graph.quantize_linear_fp8, built on torch's float8 cast) against an independent bit-level numpy restatement of the
published optimum-quanto / torchao recipe (oracle/fp8_ref.py) — codes, scales, transposed copy, dequantised values."""
import numpy as np
import torch

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd.graph import Linear, quantize_linear_fp8
from oracle import fp8_ref


def test_e4m3fn_codec_round_trips_every_code_and_matches_torch():
    codes = np.arange(256, dtype=np.uint8)
    vals = fp8_ref.e4m3fn_decode(codes)
    t = torch.from_numpy(codes.copy()).view(torch.float8_e4m3fn).float().numpy()
    finite = ~np.isnan(vals)
    assert np.array_equal(np.isnan(vals), np.isnan(t)) and np.array_equal(vals[finite], t[finite])
    assert vals[0x7E] == 448.0 and vals[0x01] == 2.0 ** -9
    back = fp8_ref.e4m3fn_encode(vals[finite])
    keep = codes[finite]
    assert np.array_equal(back[keep != 0x80], keep[keep != 0x80])  # -0 encodes with the sign bit from signbit(): also equal
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-448, 448, 20000), rng.normal(0, 1, 20000), rng.normal(0, 1e-3, 5000), [0.0, 448.0, -448.0, 1e-9]]).astype(np.float32)
    ours = fp8_ref.e4m3fn_encode(x)
    tq = torch.from_numpy(x).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    assert np.array_equal(ours, tq), np.nonzero(ours != tq)[0][:10]


def test_product_quantizer_equals_the_restated_recipe():
    g = torch.Generator().manual_seed(3)
    w = torch.randn(96, 160, generator=g) * 0.02
    w[5] = 0.0            # an all-zero output channel: scale clamps, codes are zero
    w[7, 3] = 3.0         # an outlier row
    lin = Linear(160, 96, bias=False, dtype=torch.bfloat16)
    quantize_linear_fp8(lin, w.to(torch.bfloat16))
    codes, scale = fp8_ref.quantize_per_channel(w.to(torch.bfloat16).float().numpy())
    assert np.array_equal(lin.qweight.numpy(), codes)
    assert np.array_equal(lin.wscale.numpy(), scale)
    assert np.array_equal(lin.qweight_t.numpy(), codes.T)
    deq = (lin.qweight.view(torch.float8_e4m3fn).float() * lin.wscale[:, None]).numpy()
    assert np.array_equal(deq, fp8_ref.dequantize(codes, scale))
    # worst-case relative error of the representation: half an e4m3 step (2^-4) of the row maximum
    err = np.abs(deq - w.to(torch.bfloat16).float().numpy()).max(axis=1)
    assert (err <= 2.0 ** -4 * np.abs(w.to(torch.bfloat16).float().numpy()).max(axis=1) + 1e-12).all()
