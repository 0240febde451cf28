"""Tensor-level wrappers over the C ABI: they only translate torch tensors into (pointer, stride, size) and
enqueue on the current HIP stream.  No arithmetic happens in Python."""
import ctypes as C
import os

import torch

from . import _capi
from ._capi import EPI_ACCUM, EPI_BIAS, EPI_DGELU, EPI_GATE_RES, EPI_GELU  # noqa: F401

BF16 = torch.bfloat16
STAGE_MODE = int(os.environ.get("AITK_GEMM_STAGE", "1"))  # 1 = LDS-DMA staging (faster, profiles/r01_gpu_check_01)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _row_major(t, name):
    assert t.dtype == BF16, f"{name}: expected bf16, got {t.dtype}"
    assert t.dim() == 2 and t.stride(1) == 1, f"{name}: need 2-D row-major view, got {tuple(t.shape)} strides {t.stride()}"
    return t.stride(0)


def gemm_nt(a, b, out, *, bias=None, a2=None, b2=None, flags=0, aux_out=None, aux_in=None, gate=None,
            gate_rows=0, a_seg=None, c_seg=None, M=None, stage_mode=None):
    """out[M,N] = epi(a[M,K] @ b[N,K]^T + a2[M,K2] @ b2[N,K2]^T + bias).

    a_seg / c_seg = (seg_rows, seg_stride_elems): logical row m lives at base + (m // seg_rows) * seg_stride
    + (m % seg_rows) * ld (used for the image/text halves of joint attention buffers); then `a`/`out` is the 2-D
    view of the FIRST segment and M must be given.
    """
    g = _capi.GemmArgs()
    g.lda = _row_major(a, "a")
    g.ldb = _row_major(b, "b")
    g.ldc = _row_major(out, "out")
    N, K = b.shape
    assert a.shape[1] == K
    if M is None:
        M = a.shape[0]
    g.A, g.B, g.C = _ptr(a), _ptr(b), _ptr(out)
    if a_seg is not None:
        g.a_seg_rows, g.a_seg_stride = a_seg
    if c_seg is not None:
        g.c_seg_rows, g.c_seg_stride = c_seg
    if a2 is not None:
        g.lda2 = _row_major(a2, "a2")
        g.ldb2 = _row_major(b2, "b2")
        assert a2.shape[1] == b2.shape[1] and b2.shape[0] == N and a2.shape[0] == M
        g.A2, g.B2, g.K2 = _ptr(a2), _ptr(b2), a2.shape[1]
    if bias is not None:
        assert bias.dtype == BF16 and bias.numel() == N and bias.is_contiguous()
        flags |= EPI_BIAS
        g.bias = _ptr(bias)
    if aux_out is not None:
        g.ld_aux_out = _row_major(aux_out, "aux_out")
        g.aux_out = _ptr(aux_out)
    if aux_in is not None:
        g.ld_aux_in = _row_major(aux_in, "aux_in")
        g.aux_in = _ptr(aux_in)
    if gate is not None:
        g.ld_gate = _row_major(gate, "gate")
        g.gate, g.gate_rows = _ptr(gate), gate_rows
    g.M, g.N, g.K, g.flags = M, N, K, flags
    g.stage_mode = STAGE_MODE if stage_mode is None else stage_mode
    _capi.check(_capi.lib().aitk_gemm_nt(C.byref(g), _capi.stream_ptr()), "aitk_gemm_nt")
    return out
