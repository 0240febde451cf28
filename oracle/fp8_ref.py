"""ORACLE (test infrastructure): weight-only fp8 (OCP e4m3fn) quantiser of BASELINE config 5, restated in numpy.

The reference quantises the frozen base through third-party libraries that are not in this image and not vendored:
  * optimum-quanto `qfloat8` (toolkit/util/quantize.py:65-75, 213-228; stable_diffusion_model.py:794-801): weight-only,
    qtype float8_e4m3fn, per-output-channel (axis 0) scale = absmax(row) / qtype.qmax with qmax = 448, data = (w / scale) cast to e4m3fn
    (round-to-nearest-even, saturating), dequantised weight = data * scale in the compute dtype;
  * torchao `Float8WeightOnlyConfig` (quantize.py:43-49): the same per-row absmax / 448 affine recipe.
Neither is importable here => **parity unpinned** against their code; this file restates the published recipe independently of torch's
float8 cast (explicit bit-level e4m3fn encode / decode), and tests hold the product quantiser (ai_toolkit_amd.graph.quantize_linear_fp8)
and the HIP dequantisation to it bit for bit.
"""
import numpy as np

E4M3_MAX = 448.0


def e4m3fn_decode(b):
    """uint8 -> float32 (OCP e4m3fn: 1 sign, 4 exponent bits (bias 7), 3 mantissa bits; no inf, S.1111.111 = NaN)."""
    b = np.asarray(b, dtype=np.uint8).astype(np.int32)
    s = np.where(b & 0x80, -1.0, 1.0)
    e = (b >> 3) & 0xF
    m = b & 0x7
    sub = (m / 8.0) * 2.0 ** -6
    nor = (1.0 + m / 8.0) * np.exp2(e.astype(np.float64) - 7)
    v = np.where(e == 0, sub, nor)
    v = np.where((e == 15) & (m == 7), np.nan, v)
    return (s * v).astype(np.float32)


_TABLE = None


def _table():
    global _TABLE
    if _TABLE is None:
        codes = np.arange(0, 0x7F, dtype=np.uint8)  # non-negative finite codes 0x00 .. 0x7E (0x7F = NaN)
        _TABLE = (codes, e4m3fn_decode(codes).astype(np.float64))
    return _TABLE


def e4m3fn_encode(x):
    """float -> uint8 with round-to-nearest-even on the e4m3fn grid, saturating at +-448 (what a float8_e4m3fn cast of an in-range
    value does; |x| <= 448 always holds after division by absmax / 448)."""
    x = np.asarray(x, dtype=np.float64)
    codes, vals = _table()
    a = np.minimum(np.abs(x), E4M3_MAX)
    hi = np.searchsorted(vals, a, side="left").clip(0, len(vals) - 1)
    lo = (hi - 1).clip(0, len(vals) - 1)
    dlo, dhi = a - vals[lo], vals[hi] - a
    pick_hi = (dhi < dlo) | ((dhi == dlo) & ((codes[hi] & 1) == 0))  # ties to the even mantissa
    c = np.where(pick_hi, codes[hi], codes[lo]).astype(np.uint8)
    return np.where(np.signbit(x), c | 0x80, c).astype(np.uint8)


def quantize_per_channel(w):
    """w float [out, in] -> (codes uint8 [out, in], scale float32 [out]): scale = max(absmax(row), 1e-12) / 448."""
    w = np.asarray(w, dtype=np.float32)
    scale = (np.maximum(np.abs(w).max(axis=1), 1e-12) / np.float32(E4M3_MAX)).astype(np.float32)
    return e4m3fn_encode((w / scale[:, None]).astype(np.float32)), scale


def dequantize(codes, scale, axis=0):
    s = scale[:, None] if axis == 0 else scale[None, :]
    return e4m3fn_decode(codes) * s.astype(np.float32)
