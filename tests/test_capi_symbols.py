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
    assert lib.aitk_abi_version() == _capi.ABI_VERSION == 12
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


def test_stream_k_tail_schedule_covers_every_k_tile_once_and_only_waits_downwards():
    """aitk_probe_gemm8_sk_item evaluates, on the host, the work list the stream-K instantiations of the persistent GEMM walk (gemm8.hip, sk_item_of): full
    tile rounds data-parallel, the remaining tiles cut per XCD into one K-tile range per workgroup.  Every (tile, K-tile) exactly once; no chunk shorter
    than the two K-tiles the software pipeline needs; a workgroup's open chunk (one that does not reach its tile's last K-tile) comes first and is the only
    one; a chunk that continues a tile finds the previous chunk of that tile as the last item of its predecessor (a lower workgroup on the same XCD); the
    longest work list is within three K-tiles of the ideal share."""
    lib = _capi.lib()
    item = lib.aitk_probe_gemm8_sk_item
    pred = lib.aitk_probe_gemm8_sk_predecessor
    out = (ctypes.c_int32 * 3)()
    for G, ntiles, ns in ((256, 216, 49), (256, 216, 51), (256, 648, 49), (256, 864, 193), (256, 432, 241), (256, 1512, 49), (256, 100, 145), (256, 297, 64),
                          (256, 220, 49), (64, 54, 17), (8, 13, 40)):
        cover = {}
        lists = {}
        ndp = ntiles - ntiles % G
        for w in range(G):
            lst = []
            while item(G, w, ntiles, ns, len(lst), out) == 0:
                lst.append((out[0], out[1], out[2]))
                assert len(lst) <= ntiles // G + 3
            lists[w] = lst
            tail = [x for x in lst if x[0] >= ndp]
            assert lst[:len(lst) - len(tail)] == [(w + r * G, 0, ns) for r in range(ndp // G)]
            opens = [j for j, (_, _, ke) in enumerate(tail) if ke < ns]
            assert opens in ([], [0]), (G, ntiles, ns, w, tail)
            for v, kb, ke in lst:
                assert 0 <= v < ntiles and 0 <= kb and kb + 2 <= ke <= ns
                assert v < ndp or (v - ndp) % 8 == w % 8  # tail tiles stay on the XCD the data-parallel order would give them
                for k in range(kb, ke):
                    assert (v, k) not in cover
                    cover[(v, k)] = w
        assert len(cover) == ntiles * ns
        for w, lst in lists.items():
            for v, kb, ke in lst:
                if kb > 0:
                    p = pred(G, w, ntiles, ns)
                    assert 0 <= p < w and p % 8 == w % 8
                    pv, pkb, pke = lists[p][ndp // G]  # the predecessor's open chunk is the first item of its tail
                    assert pv == v and pke == kb and pke < ns, (w, (v, kb, ke), p, lists[p])
        work = [sum(ke - kb for _, kb, ke in lst) for lst in lists.values()]
        cnt_max = ((ntiles - ndp) + 7) // 8  # tail tiles of the fullest XCD
        assert max(work) <= (ndp // G) * ns + cnt_max * ns / (G // 8) + 3, (G, ntiles, ns, max(work))
    assert item(0, 0, 1, 1, 0, out) == -3 and item(12, 0, 20, 9, 0, out) == 1  # needs a grid that is a multiple of the XCD count
