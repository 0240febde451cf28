"""CPU-side checks of the C-ABI boundary: the in-tree library loads, exports every entry point declared in
include/aitk_mi355.h, and the ctypes mirrors agree with the C struct sizes (no compute calls: no GPU here)."""
import ctypes
import os
import re

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd import _capi


import shutil

import pytest


@pytest.fixture(scope="module", autouse=True)
def _built_library():
    """A fresh checkout has no libaitk_mi355.so (built artefacts are git-ignored): cross-compile it once (hipcc, no GPU needed)."""
    if not os.path.exists(_capi.LIB_PATH) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        import __graft_entry__

        __graft_entry__.build()


def _declared(repo_root):
    src = open(os.path.join(repo_root, "include", "aitk_mi355.h")).read()
    return sorted(set(re.findall(r"\b(?:int|int32_t|int64_t)\s+(aitk_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(repo_root):
    names = _declared(repo_root)
    assert len(names) >= 25, names
    assert os.path.exists(_capi.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_capi.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_struct_size_handshake_and_abi_version():
    lib = _capi.lib()  # raises on any Python/C struct size mismatch
    assert lib.aitk_abi_version() == _capi.ABI_VERSION == 10
    assert lib.aitk_sizeof(0) == ctypes.sizeof(_capi.GemmArgs)
    assert lib.aitk_sizeof(99) == -1


def test_argument_validation_returns_error_codes_without_touching_the_gpu():
    lib = _capi.lib()
    g = _capi.GemmArgs()
    g.M, g.N, g.K = 0, 128, 64
    assert lib.aitk_gemm_nt(ctypes.byref(g), None) == -1  # AITK_ERR_SHAPE
    g.M, g.K = 128, 60
    assert lib.aitk_gemm_nt(ctypes.byref(g), None) == -1
    a = _capi.AttnArgs()
    a.B, a.H, a.S, a.D = 1, 2, 64, 64
    assert lib.aitk_attn_fwd(ctypes.byref(a), None) == -1  # head_dim must be 128
    assert lib.aitk_lora_wgrad_workspace_bytes(1000, 16, 3072) == 4 * 16 * 3072 * 4


def test_missing_library_fails_loudly(monkeypatch):
    import pytest

    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", "/nonexistent/libaitk_mi355.so")
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        _capi.lib()


def test_round2_entry_points_validate_their_arguments_before_launching():
    """Every new C entry point returns an AITK_ERR_* code on bad arguments (no GPU is touched on these paths)."""
    lib = _capi.lib()
    a, b = _capi.GemmArgs(), _capi.GemmArgs()
    assert lib.aitk_gemm_nt_grouped(ctypes.byref(a), ctypes.byref(b), None) == -1      # empty problems: AITK_ERR_SHAPE
    assert lib.aitk_gemm_nt_grouped(None, ctypes.byref(b), None) == -1
    at = _capi.AttnArgs()
    at.B, at.H, at.S, at.D, at.Dv = 1, 2, 64, 128, 200                                 # valid width beyond the 128-column layout
    assert lib.aitk_attn_fwd(ctypes.byref(at), None) == -1
    at.Dv = 64
    at.ldq = at.ldk = at.ldv = at.ldo = 256
    assert lib.aitk_attn_fwd(ctypes.byref(at), None) == -3                             # null operands: AITK_ERR_ARG
    at.hstride = 64
    assert lib.aitk_attn_fwd(ctypes.byref(at), None) == -3                             # native 64-wide heads: accepted (operands still null)
    for w in (40, 32):                                                                 # 40: not whole contraction steps / output blocks; 32: no exact instantiation
        at.Dv, at.hstride = w, w
        assert lib.aitk_attn_fwd(ctypes.byref(at), None) == -1
    at.Dv, at.hstride = 64, 96                                                         # native layout means hstride == Dv
    assert lib.aitk_attn_bwd(ctypes.byref(at), None) == -1
    at.Dv, at.hstride = 64, 0
    assert lib.aitk_lokr_lowrank_grad(None, None, None, None, None, 8, 8, 4, 1, None) == -3
    sd = _capi.ShadowDesc()
    assert ctypes.sizeof(sd) == 48 and hasattr(sd, "aux")                              # kind 3 (composed low-rank LoKr factor) carries its rank here
    assert lib.aitk_lora_refresh_shadows(None, None, None, 1, None) == -3


def test_attention_workgroup_order_is_a_bijection_that_keeps_a_head_on_one_xcd():
    """aitk_probe_attn_wg_coords evaluates, on the host, the map the attention kernels apply to blockIdx.x (1-D grid): every (tile, head,
    batch) exactly once for ragged grid sizes, and the row tiles of one (batch, head) on as few XCDs (= block id mod 8) as their count allows."""
    import collections

    lib = _capi.lib()
    out = (ctypes.c_int32 * 3)()
    for ntiles, H, B in ((36, 24, 7), (18, 24, 4), (5, 3, 2), (1, 1, 1), (9, 2, 3), (2, 10, 1)):
        n = ntiles * H * B
        seen = set()
        xcds = collections.defaultdict(set)
        for i in range(n):
            assert lib.aitk_probe_attn_wg_coords(n, i, ntiles, H, out) == 0
            t, h, b = out[0], out[1], out[2]
            assert 0 <= t < ntiles and 0 <= h < H and 0 <= b < B
            seen.add((t, h, b))
            xcds[(h, b)].add(i % 8)
        assert len(seen) == n
        if n >= 8 * ntiles:  # an XCD's share is at least one head long: a head spans at most two XCDs (one boundary)
            assert max(len(v) for v in xcds.values()) <= 2, (ntiles, H, B)
    assert lib.aitk_probe_attn_wg_coords(8, 8, 2, 2, out) == -3 and lib.aitk_probe_attn_wg_coords(8, 0, 2, 2, None) == -3
