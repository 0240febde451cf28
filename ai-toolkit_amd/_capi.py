"""ctypes binding of include/aitk_mi355.h.  The product path has NO fallback: if the HIP library is missing
or a call returns non-zero, a RuntimeError is raised."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libaitk_mi355.so")
_lib = None

vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", vp), ("lda", i64), ("a_seg_rows", i32), ("_pad0", i32), ("a_seg_stride", i64),
        ("B", vp), ("ldb", i64),
        ("A2", vp), ("lda2", i64),
        ("B2", vp), ("ldb2", i64),
        ("C", vp), ("ldc", i64), ("c_seg_rows", i32), ("_pad1", i32), ("c_seg_stride", i64),
        ("bias", vp),
        ("aux_out", vp), ("ld_aux_out", i64),
        ("aux_in", vp), ("ld_aux_in", i64),
        ("gate", vp), ("ld_gate", i64), ("gate_rows", i32),
        ("M", i32), ("N", i32), ("K", i32), ("K2", i32),
        ("flags", i32), ("stage_mode", i32),
    ]


EPI_BIAS, EPI_ACCUM, EPI_GELU, EPI_DGELU, EPI_GATE_RES = 1, 2, 4, 8, 16

_STRUCTS = {0: GemmArgs}


def lib():
    """Load libaitk_mi355.so (import torch first so its bundled HIP runtime is the one in the process)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the MI355X path has no CPU/PyTorch fallback)"
        )
    import torch  # noqa: F401  (loads libamdhip64 from torch/lib before our library resolves it)

    L = C.CDLL(LIB_PATH)
    L.aitk_abi_version.restype = C.c_int
    L.aitk_sizeof.restype = C.c_int
    L.aitk_sizeof.argtypes = [i32]
    for which, st in _STRUCTS.items():
        n = L.aitk_sizeof(which)
        if n != C.sizeof(st):
            raise RuntimeError(f"ABI mismatch: struct {st.__name__} is {C.sizeof(st)} B in Python, {n} B in C")
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with status {rc} ({'AITK_ERR' if rc < 0 else 'hipError'})")


def stream_ptr():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
