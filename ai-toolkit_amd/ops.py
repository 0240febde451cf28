"""Tensor-level wrappers over the C ABI: they only translate torch tensors into (pointer, stride, size) and
enqueue on the current HIP stream.  No arithmetic happens in Python."""
import ctypes as C
import os

import math

import torch

from . import _capi
from ._capi import EPI_ACCUM, EPI_BIAS, EPI_DGELU, EPI_GATE_RES, EPI_GELU  # noqa: F401

BF16 = torch.bfloat16
TILE_MODE = int(os.environ.get("AITK_GEMM_TILE", "0"))  # 0 auto, 1 = 128x128, 2 = 256x256
STAGE_MODE = int(os.environ.get("AITK_GEMM_STAGE", "1"))  # 1 = LDS-DMA staging (faster, profiles/r01_gpu_check_01)


def _ptr(t):
    if t is None:
        return C.c_void_p(0)
    if _REC is not None:
        # deferred launch: the tensor must outlive the recording — a temporary freed inside one stream's body could otherwise be
        # handed by the allocator to the OTHER stream's body, whose kernels may run first in the merged order
        _REC.append(("_keepalive", (t,)))
    return C.c_void_p(t.data_ptr())


# ---------------------------------------------------------------------------------------------------------- launch plumbing
# Every wrapper ends in _call(name, *args): the C entry point is invoked on the current stream — or, while `recording()`, appended to
# a launch list.  Two launch lists of independent work (the image and the text stream of a FLUX double block) are merged by
# `replay_paired`: each list keeps its own order, and whenever both are about to launch a GEMM the two go out as ONE
# aitk_gemm_nt_grouped call (one persistent 8-phase launch when they are compatible, back-to-back launches otherwise).
_REC = None
_gemm_hook = None  # bench.py: fn(e0, e1, flops, shapes) called with events around every aitk_gemm_nt / aitk_gemm_nt_grouped launch


def _gemm_shape(ref):
    g = ref._obj
    return (g.M, g.N, g.K, g.K2, g.flags)


_attn_hook = None  # bench.py: fn(name, e0, e1, flops, (B, H, S, Skv, Dv)) with events around every aitk_attn_fwd / aitk_attn_bwd call


def _emit(name, args):
    if name == "_keepalive":
        return
    if name == "_host":
        args[0]()
        return
    if name == "_wgrad_defer":
        _emit_wgrad_defer(*args)
        return
    timed = _gemm_hook is not None and name in ("aitk_gemm_nt", "aitk_gemm_nt_grouped")
    timed_attn = _attn_hook is not None and name in ("aitk_attn_fwd", "aitk_attn_bwd")
    if timed or timed_attn:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _capi.check(getattr(_capi.lib(), name)(*args, _capi.stream_ptr()), name)
    if timed:
        e1.record()
        shapes = tuple(_gemm_shape(a) for a in args)
        _gemm_hook(e0, e1, sum(2.0 * m * n * (k + k2) for m, n, k, k2, _ in shapes), shapes if len(shapes) > 1 else shapes[0])
    elif timed_attn:
        e1.record()
        a = args[0]._obj
        skv, dv = a.Skv or a.S, a.Dv or 128
        # algorithmic matmuls of 2 S Skv Dv flop per (batch, head): forward 2 (Q K^T, P V), backward 4 (dV, dP, dK, dQ; the recomputation
        # of S the flash form needs is not algorithmic work — BASELINE.md section 3: 14.9 + 29.7 TFLOP per FLUX image)
        _attn_hook(name, e0, e1, (2 if name == "aitk_attn_fwd" else 4) * 2.0 * a.B * a.H * a.S * skv * dv, (a.B, a.H, a.S, skv, dv))


def _call(name, *args):
    if _REC is not None:
        _REC.append((name, args))
        return
    _emit(name, args)


def host_call(fn):
    """Torch-side work (a pad, an add into the arena) that must keep its place among the launches: run now, or — while `recording()` — when the
    launch list is replayed, in list order.  Tensors the closure holds live as long as the list."""
    if _REC is not None:
        _REC.append(("_host", (fn,)))
        return
    fn()


# ---------------------------------------------------------------------------------------------------------- deferred weight-gradient finishes
# aitk_lora_wgrad = a producing launch (chunk partials) + a finish launch (sum of the chunks into the gradient arena).  Nothing in a backward pass reads a weight
# gradient, so between wgrad_defer_begin() and wgrad_defer_end() the finishes of launches flagged `defer` are collected — each producer writes into its own slot of a
# ring of partial buffers — and go out eight at a time as ONE launch (aitk_lora_wgrad_finish_multi, ABI 12; same sums, bit for bit).  At B = 1 a finish launch is
# 5 us of start-up behind a 10-us producer, 380 times per step.  Everything is decided when a launch is EMITTED (ring slot, batch boundaries), so the merged image /
# text launch lists of the double blocks keep one consistent order.
WGRAD_DEFER = os.environ.get("AITK_WGRAD_DEFER", "1") != "0"
WGRAD_DEFER_MAX = 8
_wdefer = None  # {"jobs": [(args, keep)], "slot": int} while a backward pass collects


def wgrad_defer_begin(device):
    global _wdefer
    if _wdefer is not None:  # a backward pass that raised half way: its gradients are void anyway
        _wdefer = None
    if not WGRAD_DEFER or _REC is not None or torch.device(device).type != "cuda" or not hasattr(_capi.lib(), "aitk_lora_wgrad_finish_multi"):
        return False
    _wdefer = {"jobs": [], "slot": 0}
    return True


def _wgrad_defer_flush():
    st = _wdefer
    if st is None or not st["jobs"]:
        return
    n = len(st["jobs"])
    arr = (_capi.LoraWgradArgs * n)()
    for i, (a, _) in enumerate(st["jobs"]):
        C.memmove(C.byref(arr[i]), C.byref(a), C.sizeof(_capi.LoraWgradArgs))
    _capi.check(_capi.lib().aitk_lora_wgrad_finish_multi(arr, n, _capi.stream_ptr()), "aitk_lora_wgrad_finish_multi")
    st["jobs"] = []


def wgrad_defer_flush():
    """Every collected finish goes out now (in launch order while recording): call before anything reads the gradient arena."""
    if _REC is not None:
        _REC.append(("_host", (_wgrad_defer_flush,)))
        return
    _wgrad_defer_flush()


def wgrad_defer_end():
    global _wdefer
    assert _REC is None
    _wgrad_defer_flush()
    _wdefer = None


def _emit_wgrad_defer(a, q, nbytes, device, keep):
    st = _wdefer
    lib = _capi.lib()
    if st is None:  # no backward pass is collecting: the plain two-launch call
        ws = workspace(nbytes, device, "wgrad")
        a.partial = C.c_void_p(ws.data_ptr())
        if q is None:
            _capi.check(lib.aitk_lora_wgrad(C.byref(a), _capi.stream_ptr()), "aitk_lora_wgrad")
        else:
            _capi.check(lib.aitk_lora_wgrad2(C.byref(a), C.byref(q), _capi.stream_ptr()), "aitk_lora_wgrad2")
        return
    if any(j[0].out == a.out for j in st["jobs"]):  # a second accumulation into the same matrix: not inside one finish launch
        _wgrad_defer_flush()
    ws = workspace(nbytes, device, f"wgrad_ring{st['slot']}")
    st["slot"] = (st["slot"] + 1) % WGRAD_DEFER_MAX
    a.partial = C.c_void_p(ws.data_ptr())
    _capi.check(lib.aitk_lora_wgrad_main(C.byref(a), C.byref(q) if q is not None else None, _capi.stream_ptr()), "aitk_lora_wgrad_main")
    st["jobs"].append((a, keep))
    if len(st["jobs"]) == WGRAD_DEFER_MAX:
        _wgrad_defer_flush()


class recording:
    """`with ops.recording() as launches:` — wrappers called inside enqueue nothing; `launches` is handed to replay_paired."""

    def __enter__(self):
        global _REC
        assert _REC is None, "recording() does not nest"
        self.launches = _REC = []
        return self.launches

    def __exit__(self, *exc):
        global _REC
        _REC = None
        return False


def replay_paired(a, b):
    """Launch two recorded lists of mutually independent work; GEMMs that meet are grouped.  `_keepalive` entries hold every tensor a
    recorded launch points to until the caller drops the lists, i.e. until everything has been enqueued on the stream."""
    a_all, b_all = a, b
    a = [e for e in a_all if e[0] != "_keepalive"]
    b = [e for e in b_all if e[0] != "_keepalive"]
    i = j = 0
    while i < len(a) or j < len(b):
        while i < len(a) and a[i][0] != "aitk_gemm_nt":
            _emit(*a[i])
            i += 1
        while j < len(b) and b[j][0] != "aitk_gemm_nt":
            _emit(*b[j])
            j += 1
        if i < len(a) and j < len(b):
            _emit("aitk_gemm_nt_grouped", (a[i][1][0], b[j][1][0]))
            i, j = i + 1, j + 1
        elif i < len(a):
            _emit(*a[i])
            i += 1
        elif j < len(b):
            _emit(*b[j])
            j += 1


def _row_major(t, name):
    assert t.dtype == BF16, f"{name}: expected bf16, got {t.dtype}"
    assert t.dim() == 2 and t.stride(1) == 1, f"{name}: need 2-D row-major view, got {tuple(t.shape)} strides {t.stride()}"
    return t.stride(0)


def gemm_nt(a, b, out, *, bias=None, a2=None, b2=None, flags=0, aux_out=None, aux_in=None, gate=None,
            gate_rows=0, a_seg=None, c_seg=None, M=None, stage_mode=None, tile_mode=None, b_scale=None, b_scale_mode=0,
            col_scale=None, a_scale=None, emit_t=None):
    """out[M,N] = epi(a[M,K] @ b[N,K]^T + a2[M,K2] @ b2[N,K2]^T + bias).
    b_scale_mode 3 (W8A8 on the MX-scaled fp8 MFMA): a AND b are e4m3 bytes, out = epi(a_scale[m] b_scale[n] (a b^T) + a2 b2^T + bias);
    a_scale fp32 [M] comes from quant_rows_fp8, b_scale fp32 [N] (or None = 1) is the weight's per-channel scale.

    a_seg / c_seg = (seg_rows, seg_stride_elems): logical row m lives at base + (m // seg_rows) * seg_stride
    + (m % seg_rows) * ld (used for the image/text halves of joint attention buffers); then `a`/`out` is the 2-D
    view of the FIRST segment and M must be given.
    """
    g = _capi.GemmArgs()
    if b_scale_mode == 3:
        assert a.element_size() == 1 and b.element_size() == 1 and a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
        assert a_scale is not None and a_scale.dtype == torch.float32 and a_scale.is_contiguous() and a_seg is None
        assert b_scale is None or (b_scale.dtype == torch.float32 and b_scale.is_contiguous() and b_scale.numel() == b.shape[0])
        g.lda, g.ldb = a.stride(0), b.stride(0)
        g.a_scale, g.b_scale, g.b_scale_mode = _ptr(a_scale), _ptr(b_scale), 3
    else:
        g.lda = _row_major(a, "a")
    if b_scale_mode == 3:
        pass
    elif b_scale is not None:  # weight-only fp8 base: b is uint8/float8 bytes [N, K]
        assert b.element_size() == 1 and b.dim() == 2 and b.stride(1) == 1 and b_scale.dtype == torch.float32 and b_scale_mode in (1, 2)
        g.ldb = b.stride(0)
        g.b_scale, g.b_scale_mode = _ptr(b_scale), b_scale_mode
    else:
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
        assert bias.dtype == BF16 and bias.is_contiguous()
        if flags & _capi.EPI_BIAS_ROW:
            assert bias.numel() == M
        else:
            assert bias.numel() == N
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
    if col_scale is not None:  # DoRA: product * col_scale[n] before the bias
        assert col_scale.dtype == torch.float32 and col_scale.is_contiguous() and col_scale.numel() == N
        g.col_scale = _ptr(col_scale)
        flags |= _capi.EPI_COL_SCALE
    if emit_t is not None:
        # AITK_EPI_EMIT_T: (p_hi [16 | 32, >= N] view, p_lo likewise, partial fp32 [tiles, M, 16 | 32], first tile slot) — the BIAS | GELU launch also leaves
        # the column-tile partials of `gelu output @ (p_hi + p_lo)^T` (the consumer layer's lora_down product); lora_t_finish turns them into T
        p_hi, p_lo, partial, tile0 = emit_t
        assert flags == (EPI_BIAS | EPI_GELU) and N % 256 == 0 and a_seg is None and c_seg is None and not b_scale_mode
        Rt = p_hi.shape[0]
        assert p_hi.dtype == BF16 and p_lo.dtype == BF16 and Rt in (16, 32) and p_hi.shape[1] >= N and p_hi.stride(1) == 1
        assert p_lo.shape == p_hi.shape and p_lo.stride() == p_hi.stride()
        assert partial.dtype == torch.float32 and partial.is_contiguous() and partial.dim() == 3 and partial.shape[1] == M and partial.shape[2] == Rt
        assert partial.shape[0] >= tile0 + N // 256
        g.t_partial, g.t_p, g.t_p_lo, g.t_ldp, g.t_tile0, g.t_rank = _ptr(partial), _ptr(p_hi), _ptr(p_lo), p_hi.stride(0), int(tile0), Rt
        flags |= _capi.EPI_EMIT_T
    g.M, g.N, g.K, g.flags = M, N, K, flags
    g.stage_mode = STAGE_MODE if stage_mode is None else stage_mode
    g.tile_mode = TILE_MODE if tile_mode is None else tile_mode
    _call("aitk_gemm_nt", C.byref(g))
    return out


# ----------------------------------------------------------------------------------------------------------
_ws = {}


def workspace(nbytes, device, tag="ws"):
    """Grow-only fp32 scratch owned by torch (the C ABI never allocates)."""
    key = (tag, str(device))
    n = (nbytes + 3) // 4
    t = _ws.get(key)
    if t is None or t.numel() < n:
        if t is not None and _REC is not None:
            _REC.append(("_keepalive", (t,)))  # recorded launches may still point into the buffer being replaced
        t = torch.empty(max(n, 1), dtype=torch.float32, device=device)
        _ws[key] = t
    return t


def rows_per_block():
    return _capi.lib().aitk_rows_per_block()


def quant_rows_fp8(x, q, row_scale, *, col_mul=None, x_seg=None, M=None):
    """q[M,K] (uint8 = OCP e4m3 bytes) = e4m3(x * col_mul / row_scale[:, None]), row_scale[m] = max_k |x col_mul| / 448 — the per-token
    dynamic quantisation of a W8A8 GEMM's activation operand (gemm_nt b_scale_mode 3)."""
    a = _capi.QuantRowsArgs()
    a.ldx = _row_major(x, "x")
    K = x.shape[1]
    if M is None:
        M = x.shape[0]
    assert q.dtype == torch.uint8 and q.shape[1] == K and q.stride(1) == 1 and q.shape[0] >= M
    assert row_scale.dtype == torch.float32 and row_scale.is_contiguous() and row_scale.numel() >= M
    a.X, a.Q, a.row_scale, a.ldq = _ptr(x), _ptr(q), _ptr(row_scale), q.stride(0)
    if x_seg is not None:
        a.seg_rows, a.seg_stride = x_seg
    if col_mul is not None:
        assert col_mul.dtype == torch.float32 and col_mul.is_contiguous() and col_mul.numel() == K
        a.col_mul = _ptr(col_mul)
    a.M, a.K = M, K
    _call("aitk_quant_rows_fp8", C.byref(a))
    return q, row_scale


def lora_down(x, pmat, out, *, scale=1.0, mult=None, rows_per_batch=0, x_seg=None, M=None, p_lo=None, split=0, tmask=None,
              tmask_rows_per_batch=0):
    """out = scale * mult[m // rows_per_batch] * (x[M,K] @ (pmat + p_lo)[R,K]^T): [M,R] bf16, or — split = rank-block width —
    the [M,3R] K-slab layout [hi | lo | hi] per rank block (AitkLoraDownArgs in the header)."""
    R, K = pmat.shape
    if R > 64:  # the kernel contracts up to 64 ranks per launch: larger ranks go out in 64-rank chunks of the same slab (split >= R: one
        #         rank block, the chunk's hi / lo / hi columns sit at its rank offset inside it)
        if split and split < R:
            raise NotImplementedError("lora_down: ranks above 64 in several rank blocks per launch")
        assert out.shape[1] == (3 * split if split else R)
        for c0 in range(0, R, 64):
            c1 = min(R, c0 + 64)
            # the dropout mask of a chunk: its own [rows, chunk ranks] matrix (the kernel indexes the mask by the rank inside the launch)
            tm = None if tmask is None else tmask[:, c0:c1].contiguous()
            _lora_down_launch(x, pmat[c0:c1], out[:, c0:], scale, mult, rows_per_batch, x_seg, M, None if p_lo is None else p_lo[c0:c1],
                              split, tm, tmask_rows_per_batch)
        return out
    assert out.shape[1] == (3 * R if split else R)
    return _lora_down_launch(x, pmat, out, scale, mult, rows_per_batch, x_seg, M, p_lo, split, tmask, tmask_rows_per_batch)


def _lora_down_launch(x, pmat, out, scale, mult, rows_per_batch, x_seg, M, p_lo, split, tmask, tmask_rows_per_batch):
    a = _capi.LoraDownArgs()
    a.ldx = _row_major(x, "x")
    a.ldp = _row_major(pmat, "pmat")
    a.ldt = _row_major(out, "out")
    R, K = pmat.shape
    assert x.shape[1] == K
    a.X, a.P, a.T = _ptr(x), _ptr(pmat), _ptr(out)
    if p_lo is not None:
        assert p_lo.shape == pmat.shape and _row_major(p_lo, "p_lo") == a.ldp
        a.P_lo = _ptr(p_lo)
    a.split_rp = int(split)
    if tmask is not None:  # dropout / rank_dropout multipliers on T, fp32 [rows, R]
        assert tmask.dtype == torch.float32 and tmask.is_contiguous() and tmask.shape[1] == R
        a.tmask, a.tmask_rows_per_batch = _ptr(tmask), int(tmask_rows_per_batch)
    if x_seg is not None:
        a.x_seg_rows, a.x_seg_stride = x_seg
    if mult is not None:
        assert mult.dtype == torch.float32 and mult.is_contiguous()
        a.mult, a.rows_per_batch = _ptr(mult), rows_per_batch
    a.scale = float(scale)
    a.M, a.K, a.R = (x.shape[0] if M is None else M), K, R
    if KSPLIT and a.M <= 32 and R == 16 and K >= 6144 and K % 1536 == 0:
        # a few rows over a long contraction (adaLN adapters' backward): K slices of 1536 across workgroups + one finish pass (aitk_lora_down_ksplit)
        nsplit = K // 1536
        ws = workspace(_capi.lib().aitk_lora_down_ksplit_workspace_bytes(a.M, R, nsplit), x.device, "down_ksplit")
        _call("aitk_lora_down_ksplit", C.byref(a), _ptr(ws), nsplit)
        return out
    _call("aitk_lora_down", C.byref(a))
    return out


KSPLIT = os.environ.get("AITK_LORA_DOWN_KSPLIT", "1") != "0"  # 0: single-workgroup launches stay single-workgroup launches (A/B)
EMIT_T_ROW_TILE = 1  # AITK_EPI_EMIT_T takes any row count on the HIP kernel (the graphs ask the kernel table: oracle/ref_ops.py keeps whole 256-row tiles)


def lora_down_raw(x, pmat, raw, *, p_lo=None, x_seg=None, M=None):
    """raw [M, R] fp32 = x[M, K] @ (pmat + p_lo)[R, K]^T (R = 16 or 32), un-scaled: one tile of a partial-sum slab (aitk_lora_down_raw)."""
    a = _capi.LoraDownArgs()
    a.ldx, a.ldp = _row_major(x, "x"), _row_major(pmat, "pmat")
    R, K = pmat.shape
    assert R in (16, 32) and x.shape[1] == K and K % 32 == 0
    M = x.shape[0] if M is None else M
    assert raw.dtype == torch.float32 and raw.is_contiguous() and tuple(raw.shape) == (M, R)
    a.X, a.P = _ptr(x), _ptr(pmat)
    if p_lo is not None:
        assert p_lo.shape == pmat.shape and _row_major(p_lo, "p_lo") == a.ldp
        a.P_lo = _ptr(p_lo)
    if x_seg is not None:
        a.x_seg_rows, a.x_seg_stride = x_seg
    a.scale, a.M, a.K, a.R = 1.0, M, K, R
    _call("aitk_lora_down_raw", C.byref(a), _ptr(raw))
    return raw


def lora_t_finish(partial, ntiles, out, *, scale=1.0, mult=None, rows_per_batch=0, split=0, tmask=None, tmask_rows_per_batch=0, M=None):
    """out = what lora_down writes (scale, mult, tmask; plain [M, R] or the [hi | lo | hi] slab) from the sum of the first `ntiles` tiles of
    partial fp32 [tiles, M, R], in tile order (aitk_lora_t_finish)."""
    a = _capi.LoraDownArgs()
    R = partial.shape[2]
    M = partial.shape[1] if M is None else M
    assert partial.dtype == torch.float32 and partial.is_contiguous() and partial.shape[1] == M and 0 < ntiles <= partial.shape[0]
    assert out.shape[1] == (3 * R if split else R)
    a.ldt, a.T = _row_major(out, "out"), _ptr(out)
    a.split_rp, a.scale = int(split), float(scale)
    if tmask is not None:
        assert tmask.dtype == torch.float32 and tmask.is_contiguous() and tmask.shape[1] == R
        a.tmask, a.tmask_rows_per_batch = _ptr(tmask), int(tmask_rows_per_batch)
    if mult is not None:
        assert mult.dtype == torch.float32 and mult.is_contiguous()
        a.mult, a.rows_per_batch = _ptr(mult), rows_per_batch
    a.M, a.K, a.R = M, 0, R
    _call("aitk_lora_t_finish", C.byref(a), _ptr(partial), int(ntiles))
    return out


def lora_bwd_fused(dy, T, pmat, p_lo, dT, g_up, *, scale=1.0, mult=None, rows_per_batch=0, M=None, split=0, tmask=None, tmask_rows_per_batch=0,
                   g_seg=None):
    """One pass over dy [M, L] for BOTH adapter-side products of a layer's backward that stream it (aitk_lora_bwd_fused):
         dT [M, 3R] (slab) = scale * mult * (dy @ (pmat + p_lo)[R, L]^T) [* tmask]      — what lora_down(dy, pmat, dT, ...) writes
         g_up [L, R] (fp32) += dy^T @ T                                                   — what lora_wgrad(T, dy, g_up, transpose_out=True, accumulate=True) adds
    R = 16 or 32 ranks in one rank block layout (`split`)."""
    R, L = pmat.shape
    assert R in (16, 32) and dy.shape[1] == L and tuple(g_up.shape) == (L, R) and g_up.dtype == torch.float32 and g_up.is_contiguous()
    assert T.shape[1] == (3 * R if split else R) and dT.shape[1] == (3 * R if split else R)
    M = dy.shape[0] if M is None else M
    w, d = _capi.LoraWgradArgs(), _capi.LoraDownArgs()
    w.lds, w.ldg, w.split_rp = _row_major(T, "T"), _row_major(dy, "dy"), int(split)
    w.out_stride_r, w.out_stride_l = 1, R
    if g_seg is not None:
        w.g_seg_rows, w.g_seg_stride = g_seg
        d.x_seg_rows, d.x_seg_stride = g_seg
    ws = workspace(_capi.lib().aitk_lora_wgrad_workspace_bytes(M, R, L), dy.device, "wgrad")
    wsd = workspace(_capi.lib().aitk_lora_bwd_fused_workspace_bytes(M, R, L), dy.device, "bwd_fused")
    w.S, w.G, w.partial, w.out = _ptr(T), _ptr(dy), _ptr(ws), _ptr(g_up)
    w.accumulate, w.M, w.R, w.L = 1, M, R, L
    d.X, d.ldx, d.P, d.ldp, d.T, d.ldt = _ptr(dy), w.ldg, _ptr(pmat), _row_major(pmat, "pmat"), _ptr(dT), _row_major(dT, "dT")
    if p_lo is not None:
        assert p_lo.shape == pmat.shape and _row_major(p_lo, "p_lo") == d.ldp
        d.P_lo = _ptr(p_lo)
    d.split_rp, d.scale = int(split), float(scale)
    if tmask is not None:
        assert tmask.dtype == torch.float32 and tmask.is_contiguous() and tmask.shape[1] == R
        d.tmask, d.tmask_rows_per_batch = _ptr(tmask), int(tmask_rows_per_batch)
    if mult is not None:
        assert mult.dtype == torch.float32 and mult.is_contiguous()
        d.mult, d.rows_per_batch = _ptr(mult), rows_per_batch
    d.M, d.K, d.R = M, L, R
    _call("aitk_lora_bwd_fused", C.byref(w), C.byref(d), _ptr(wsd))
    return dT


def slab_rescale(T, rp, *, mult=None, rows_per_batch=0, tmask=None, tmask_rows_per_batch=0, M=None):
    """In place on the [hi | lo | hi] slab T [M, 3 rp]: value * mult[m // rows_per_batch] * tmask[m // tmask_rows_per_batch][r], split again
    (a conv adapter's per-sample multiplier / dropout masks: its lora_down comes out of the convolution epilogue with a uniform scale)."""
    assert T.dtype == BF16 and T.shape[1] >= 3 * rp
    if mult is not None:
        assert mult.dtype == torch.float32 and mult.is_contiguous() and rows_per_batch > 0
    if tmask is not None:
        assert tmask.dtype == torch.float32 and tmask.is_contiguous() and tmask.shape[1] == rp
    _call("aitk_slab_rescale", _ptr(T), _row_major(T, "T"), T.shape[0] if M is None else M, rp, _ptr(mult), int(rows_per_batch), _ptr(tmask),
          int(tmask_rows_per_batch))
    return T


def lora_wgrad(s, g, out, *, transpose_out=False, accumulate=False, g_seg=None, M=None, split=0, out_strides=None, g2=None, g2_act=None, defer=False):
    """out (fp32) (+)= s[M,R]^T @ g[M,L]:  out is [R,L], or [L,R] when transpose_out (lora_up.weight.grad).
    defer: `out` is read by nobody before wgrad_defer_flush / wgrad_defer_end — the finish pass may be collected (wgrad_defer_begin).
    g2 [M, L2] (aitk_lora_wgrad2): the operand is [g | act(g2)] — `g` may be None (L = L2) — with act "gelu" = tanh-GELU of a saved
    pre-activation, so that the GELU output itself need not be kept for the backward pass.
    split = rank-block width: s is the [M,3R] slab layout written by lora_down(split=...) and is read as hi + lo.
    out_strides = (stride_r, stride_l): element (r, l) goes to out.flatten()[r*stride_r + l*stride_l] (`out` = fp32 view starting at the
    first element; one tap of a conv adapter's [r, Cin, 3, 3] gradient: strides (9 Cin, 9))."""
    if g2 is not None:
        assert g_seg is None and g2_act in (None, "gelu") and g2.dtype == BF16
        split_col = 0 if g is None else g.shape[1]
        assert split_col % 128 == 0, "the second part of the operand starts on a 128-column tile boundary"
        R, L = (s.shape[1] // 3 if split else s.shape[1]), split_col + g2.shape[1]
        if R > 64:
            raise NotImplementedError("lora_wgrad with a two-part operand: ranks above 64")
        return _lora_wgrad_launch(s, g, out, R, L, accumulate, None, M, split, out_strides, transpose_out,
                                  second=(g2, split_col, 1 if g2_act == "gelu" else 0), defer=defer)
    R, L = (s.shape[1] // 3 if split else s.shape[1]), g.shape[1]
    if R > 64:  # 64-rank chunks of one slab (see lora_down): rank r of the output at r * stride_r
        if split and split < R:
            raise NotImplementedError("lora_wgrad: ranks above 64 in several rank blocks per launch")
        assert out.dtype == torch.float32 and out.is_contiguous()
        sr, sl = out_strides if out_strides is not None else ((1, R) if transpose_out else (L, 1))
        if out_strides is None:
            assert tuple(out.shape) == ((L, R) if transpose_out else (R, L))
        flat = out.view(-1)
        for c0 in range(0, R, 64):
            c1 = min(R, c0 + 64)
            sc = s[:, c0:] if split else s[:, c0:c1]
            _lora_wgrad_launch(sc, g, flat[c0 * sr:], c1 - c0, L, accumulate, g_seg, M, split, (sr, sl), False)
        return out
    return _lora_wgrad_launch(s, g, out, R, L, accumulate, g_seg, M, split, out_strides, transpose_out, defer=defer)


def _lora_wgrad_launch(s, g, out, R, L, accumulate, g_seg, M, split, out_strides, transpose_out, second=None, defer=False):
    a = _capi.LoraWgradArgs()
    a.lds = _row_major(s, "s")
    a.ldg = _row_major(g, "g") if g is not None else 0
    a.split_rp = int(split)
    M = s.shape[0] if M is None else M
    assert out.dtype == torch.float32 and out.is_contiguous()
    if out_strides is not None:
        a.out_stride_r, a.out_stride_l = out_strides
        assert out.numel() > (R - 1) * out_strides[0] + (L - 1) * out_strides[1]
    elif transpose_out:
        assert tuple(out.shape) == (L, R)
        a.out_stride_r, a.out_stride_l = 1, R
    else:
        assert tuple(out.shape) == (R, L)
        a.out_stride_r, a.out_stride_l = L, 1
    if g_seg is not None:
        a.g_seg_rows, a.g_seg_stride = g_seg
    nbytes = _capi.lib().aitk_lora_wgrad_workspace_bytes(M, R, L)
    a.accumulate = int(accumulate)
    a.M, a.R, a.L = M, R, L
    if defer and (_wdefer is not None or _REC is not None):
        # the finish may wait (wgrad_defer_begin): the partial buffer (a ring slot) and the batch the finish joins are chosen when the launch is emitted
        a.S, a.G, a.out = _ptr(s), _ptr(g), _ptr(out)
        q = None
        keep = [s, g, out]
        if second is not None:
            g2, split_col, act = second
            q = _capi.WgradSrc2()
            q.G2, q.ldg2, q.split_col, q.act = _ptr(g2), _row_major(g2, "g2"), split_col, act
            keep.append(g2)
        _call("_wgrad_defer", a, q, nbytes, s.device, keep)
        return out
    ws = workspace(nbytes, s.device, "wgrad")
    a.S, a.G, a.partial, a.out = _ptr(s), _ptr(g), _ptr(ws), _ptr(out)
    if second is not None:
        g2, split_col, act = second
        q = _capi.WgradSrc2()
        q.G2, q.ldg2, q.split_col, q.act = _ptr(g2), _row_major(g2, "g2"), split_col, act
        _call("aitk_lora_wgrad2", C.byref(a), C.byref(q))
        return out
    _call("aitk_lora_wgrad", C.byref(a))
    return out


def ln_mod_fwd(x, shift, scale, out, *, rows_per_batch, mean=None, rstd=None, eps=1e-6):
    """out = LayerNorm(x) * (1 + scale[b]) + shift[b]; shift/scale are [B, C] views sharing one row stride."""
    a = _capi.LnModArgs()
    a.ldx = _row_major(x, "x")
    a.ld_out = _row_major(out, "out")
    a.ld_mod = _row_major(shift, "shift")
    assert _row_major(scale, "scale") == a.ld_mod
    a.x, a.shift, a.scale, a.out = _ptr(x), _ptr(shift), _ptr(scale), _ptr(out)
    a.mean, a.rstd = _ptr(mean), _ptr(rstd)
    a.eps, a.rows_per_batch = eps, rows_per_batch
    a.M, a.C = x.shape
    _call("aitk_ln_mod_fwd", C.byref(a))
    return out


def colsum_finish(partial, nchunk, B, V, Cc, out0, out1=None):
    a = _capi.ColsumFinishArgs()
    a.partial, a.out0, a.out1 = _ptr(partial), _ptr(out0), _ptr(out1)
    a.ld_out = _row_major(out0, "out0")
    if out1 is not None:
        assert _row_major(out1, "out1") == a.ld_out
    a.B, a.nchunk, a.V, a.C = B, nchunk, V, Cc
    _call("aitk_colsum_finish", C.byref(a))


def ln_mod_bwd(dxn, x, mean, rstd, scale, dx, *, B, S, dres=None, dshift=None, dscale=None):
    """dx = LN-modulate backward (+ dres);  dshift/dscale [B,C] bf16 views (column sums over the S rows of a batch)."""
    a = _capi.LnModBwdArgs()
    a.ld_dxn, a.ldx, a.ld_dx = _row_major(dxn, "dxn"), _row_major(x, "x"), _row_major(dx, "dx")
    a.ld_mod = _row_major(scale, "scale")
    Cc = x.shape[1]
    a.dxn, a.x, a.mean, a.rstd, a.scale, a.dx = _ptr(dxn), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(scale), _ptr(dx)
    if dres is not None:
        a.dres, a.ld_dres = _ptr(dres), _row_major(dres, "dres")
    nchunk = (S + rows_per_block() - 1) // rows_per_block()
    part = None
    if dshift is not None:
        part = workspace(B * nchunk * 2 * Cc * 4, x.device, "colsum")
        a.partial = _ptr(part)
    a.S, a.B, a.C = S, B, Cc
    _call("aitk_ln_mod_bwd", C.byref(a))
    if dshift is not None:
        colsum_finish(part, nchunk, B, 2, Cc, dshift, dscale)
    return dx


def gate_bwd(dx, y, gate, dy, dgate, *, B, S):
    """dy = gate[b] * dx ; dgate[b] = sum_s dx * y  (y = dgate = None: dy only)."""
    a = _capi.GateBwdArgs()
    a.ld_dx, a.ld_dy, a.ld_gate = _row_major(dx, "dx"), _row_major(dy, "dy"), _row_major(gate, "gate")
    Cc = dx.shape[1]
    a.dx, a.gate, a.dy = _ptr(dx), _ptr(gate), _ptr(dy)
    a.S, a.B, a.C = S, B, Cc
    if y is None:
        assert dgate is None
        a.ld_y = 8
        _call("aitk_gate_bwd", C.byref(a))
        return dy
    nchunk = (S + rows_per_block() - 1) // rows_per_block()
    part = workspace(B * nchunk * Cc * 4, dx.device, "colsum")
    a.y, a.ld_y, a.partial = _ptr(y), _row_major(y, "y"), _ptr(part)
    _call("aitk_gate_bwd", C.byref(a))
    colsum_finish(part, nchunk, B, 1, Cc, dgate)
    return dy


def _qkv_args(jobs, cos, sin, B, H, S_src, S_dst, s_off, eps):
    a = _capi.QkvPostArgs()
    assert 1 <= len(jobs) <= 3
    for i, j in enumerate(jobs):
        src, dst, weight = j["src"], j["dst"], j.get("weight")
        a.job[i].src, a.job[i].ld_src = _ptr(src), _row_major(src, "src")
        a.job[i].dst, a.job[i].ld_dst = _ptr(dst), _row_major(dst, "dst")
        a.job[i].weight = _ptr(weight)
        raw = j.get("raw")
        if raw is not None:
            a.job[i].raw, a.job[i].ld_raw = _ptr(raw), _row_major(raw, "raw")
    assert cos.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous() and cos.shape == (S_dst, 128)
    a.cos, a.sin = _ptr(cos), _ptr(sin)
    a.eps, a.njobs = eps, len(jobs)
    a.B, a.H, a.D, a.S_src, a.S_dst, a.s_off = B, H, 128, S_src, S_dst, s_off
    return a


def qkv_post_fwd(jobs, cos, sin, *, B, H, S_src, S_dst, s_off, eps=1e-6):
    """jobs: dicts {src [B*S_src, >=H*128], dst [B*S_dst, >=H*128], weight [128] or None (= plain copy)}."""
    a = _qkv_args(jobs, cos, sin, B, H, S_src, S_dst, s_off, eps)
    _call("aitk_qkv_post_fwd", C.byref(a))


def qkv_post_bwd(jobs, cos, sin, *, B, H, S_src, S_dst, s_off, eps=1e-6):
    """jobs: {src: raw-side grad (written), dst: joint-side grad (read), weight, raw: forward input}."""
    a = _qkv_args(jobs, cos, sin, B, H, S_src, S_dst, s_off, eps)
    _call("aitk_qkv_post_bwd", C.byref(a))


def ew(op, x, y, a=None, alpha=1.0, a_rows_per_batch=0):
    """op 0: y = silu(x); 1: y = x; 2: y = a + x (a_rows_per_batch > 0: row m // a_rows_per_batch of `a`); 3: y = alpha * x
    ([rows, C] bf16 views)."""
    g = _capi.EwArgs()
    g.alpha = float(alpha)
    g.a_rows_per_batch = int(a_rows_per_batch)
    g.x, g.ldx, g.y, g.ldy = _ptr(x), _row_major(x, "x"), _ptr(y), _row_major(y, "y")
    if a is not None:
        g.a, g.lda = _ptr(a), _row_major(a, "a")
    g.rows, g.C = x.shape
    g.op = op
    _call("aitk_ew", C.byref(g))
    return y


def timestep_embed(t, out, tscale=1.0):
    assert t.dtype == torch.float32 and out.dtype == BF16 and out.is_contiguous()
    B, dim = out.shape
    _call("aitk_timestep_embed", _ptr(t), _ptr(out), B, dim, float(tscale))
    return out


def copy_rows(dst, src):
    """dst[:, :] = src[:, :] for 2-D views with unit inner stride (strided rows) via hipMemcpy2DAsync."""
    assert dst.shape == src.shape and dst.dtype == src.dtype and dst.stride(1) == 1 and src.stride(1) == 1
    es = dst.element_size()
    _call("aitk_copy2d", _ptr(dst), dst.stride(0) * es, _ptr(src), src.stride(0) * es,
                                        dst.shape[1] * es, dst.shape[0])
    return dst


def _attn_args(q, k, v, o, lse, B, H, S, scale, Skv=0, dv=0, hstride=0):
    a = _capi.AttnArgs()
    a.Skv = Skv
    a.Dv = dv  # valid head width inside the 128-column layout (0 = 128)
    a.hstride = hstride  # elements between heads (0 = 128); == dv: heads read where the projections wrote them
    a.Q, a.K, a.V, a.O, a.LSE = _ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(lse)
    a.ldq, a.ldk, a.ldv, a.ldo = _row_major(q, "q"), _row_major(k, "k"), _row_major(v, "v"), _row_major(o, "o")
    assert lse.dtype == torch.float32 and lse.numel() == B * H * S
    a.scale, a.B, a.H, a.S, a.D = scale, B, H, S, 128
    return a


def attn_fwd(q, k, v, o, lse, *, B, H, S, scale, Skv=0, dv=0, hstride=0):
    """q,o: [B*S, >=H*128]; k,v: [B*Skv, >=H*128] bf16 views (row stride = token stride); lse [B,H,S] fp32.  dv: heads narrower than
    128 are stored zero-padded to 128 columns; the kernels then skip the all-zero parts (exactly the padded result).  hstride == dv
    (64 / 96): native [B*S, H*dv] layout, no padding."""
    a = _attn_args(q, k, v, o, lse, B, H, S, scale, Skv, dv, hstride)
    _call("aitk_attn_fwd", C.byref(a))
    return o


ATTN_DS = os.environ.get("AITK_ATTN_DS", "1") != "0"  # 5-matmul attention backward (dS emitted by the dK/dV pass, product-form dQ; bit-identical to the
# recomputing backward, step +2.7 % same box: profiles/r06_ab_attn_ds_*.json); AITK_ATTN_DS=0: the 7-matmul backward, no [B, H, S, Skv] scratch


def attn_ds_eligible(S, Skv, dvalid=0, hstride=0):
    """shapes the dS-emitting dK/dV pass and the product-form dQ kernel cover: whole tiles, full-width heads in the padded layout"""
    return S % 64 == 0 and Skv % 128 == 0 and hstride == 0 and dvalid in (0, 128)


def attn_bwd(q, k, v, o, lse, do, dq, dk, dv, *, B, H, S, scale, Skv=0, dvalid=0, hstride=0, ds=None, ds_mode=0):
    """ds / ds_mode (AitkAttnArgs.dS): 1 = the 5-matmul backward — the wave-specialised dK/dV pass also writes its bf16 dS (accumulator-native
    2-KiB blocks) and dQ = dS K runs as a product of its own (attn_bwd_dq_ds_kernel); 3 = emit dS but keep the recomputing dQ kernel, 2 = the
    same with plain instead of non-temporal stores (probes), 4 = off whatever the default.  Default (ds_mode 0, ds None): AITK_ATTN_DS."""
    a = _attn_args(q, k, v, o, lse, B, H, S, scale, Skv, dvalid, hstride)
    if ds is None and ds_mode == 0 and ATTN_DS and attn_ds_eligible(S, Skv or S, dvalid, hstride):
        # the 5-matmul backward (AITK_ATTN_DS=1): the dK/dV pass emits dS, dQ = dS K is a product of its own; the [B, H, S, Skv] bf16 scratch
        # (7.1 GB at 7 x 24 x 4608^2) is one grow-only workspace shared by every layer
        ds, ds_mode = workspace(B * H * S * (Skv or S) * 2, q.device, "attn_ds").view(BF16), 1
    if ds_mode == 4:  # explicitly off (tests: the recomputing backward whatever the default is)
        ds, ds_mode = None, 0
    if ds is not None and ds_mode:
        assert ds.dtype == BF16 and ds.is_contiguous() and ds.numel() >= B * H * ((S + 31) // 32 * 32) * (((Skv or S) + 31) // 32 * 32)
        a.dS, a.ds_mode = _ptr(ds), int(ds_mode)
    a.dO, a.lddo = _ptr(do), _row_major(do, "do")
    a.dQ, a.dK, a.dV = _ptr(dq), _ptr(dk), _ptr(dv)
    a.lddq, a.lddk, a.lddv = _row_major(dq, "dq"), _row_major(dk, "dk"), _row_major(dv, "dv")
    delta = workspace(B * H * S * 4, q.device, "attn_delta")
    a.delta = _ptr(delta)
    _call("aitk_attn_bwd", C.byref(a))


def gemv_nt(x, w, out, *, bias=None, t=None, bl=None, accumulate=False, col_scale=None):
    """out[Bm,N] (+)= x[Bm,K] @ w[N,K]^T + bias + t[Bm,R] @ bl[N,R]^T.  The kernel takes Bm <= 8 rows per launch (weight streaming);
    more rows go out in chunks of 8."""
    if x.shape[0] > 8:
        for r0 in range(0, x.shape[0], 8):
            gemv_nt(x[r0:r0 + 8], w, out[r0:r0 + 8], bias=bias, t=None if t is None else t[r0:r0 + 8], bl=bl, accumulate=accumulate,
                    col_scale=col_scale)
        return out
    a = _capi.GemvArgs()
    a.ldx, a.ldw, a.ldo = _row_major(x, "x"), _row_major(w, "w"), _row_major(out, "out")
    a.X, a.W, a.out, a.bias = _ptr(x), _ptr(w), _ptr(out), _ptr(bias)
    a.Bm, a.K = x.shape
    a.N = w.shape[0]
    assert w.shape[1] == a.K and out.shape == (a.Bm, a.N)
    if t is not None:
        a.T, a.ldt, a.Bl, a.ldbl, a.R = _ptr(t), _row_major(t, "t"), _ptr(bl), _row_major(bl, "bl"), t.shape[1]
    a.accumulate = int(accumulate)
    if col_scale is not None:
        assert col_scale.dtype == torch.float32 and col_scale.is_contiguous() and col_scale.numel() == a.N
        a.col_scale = _ptr(col_scale)
    _call("aitk_gemv_nt", C.byref(a))
    return out


def flow_noise_pack(latents, noise, t, noisy, target):
    """latents/noise [B,C,H,W] bf16, t [B] fp32 (0..1000) -> noisy/target packed [B,(H/2)(W/2),4C] bf16."""
    a = _capi.NoisePackArgs()
    assert latents.is_contiguous() and noise.is_contiguous() and noisy.is_contiguous() and target.is_contiguous()
    assert latents.dtype == BF16 and noise.dtype == BF16 and t.dtype == torch.float32
    a.latents, a.noise, a.t, a.noisy, a.target = _ptr(latents), _ptr(noise), _ptr(t), _ptr(noisy), _ptr(target)
    a.B, a.C, a.H, a.W = latents.shape
    _call("aitk_flow_noise_pack", C.byref(a))


LOSS_TYPES = {"mse": 0, "mae": 1, "pseudo_huber": 2}


def mse_loss_grad(pred, target, dpred, loss_per_sample, loss, weight=None, mask=None, loss_type="mse", huber_c=0.01, guard=None,
                  max_loss=None):
    """loss_b = mean(mask*l(pred-target)), l = d^2 | |d| | sqrt(d^2+c^2)-c (train.loss_type mse / mae / pseudo_huber), loss = mean_b(w_b loss_b);
    dpred = dloss/dpred (bf16).  mask: fp32 [B, tokens, 4] (per 2x2-patch position), pred [B, tokens, feat].  guard: int32[8] device buffer
    (AitkMseArgs.guard): a non-finite loss -> 0 and a loss above max_loss -> max_loss, each with a zero gradient, decided on the device."""
    a = _capi.MseArgs()
    B = pred.shape[0]
    n = pred[0].numel()
    assert pred.is_contiguous() and target.is_contiguous() and dpred.is_contiguous()
    ws = workspace(_capi.lib().aitk_mse_workspace_bytes(B, n), pred.device, "mse")
    a.pred, a.target, a.weight, a.dpred, a.partial = _ptr(pred), _ptr(target), _ptr(weight), _ptr(dpred), _ptr(ws)
    a.loss_per_sample, a.loss = _ptr(loss_per_sample), _ptr(loss)
    a.n_per_sample, a.B = n, B
    if mask is not None:
        assert mask.dtype == torch.float32 and mask.is_contiguous() and mask.numel() == B * (n // pred.shape[-1]) * 4
        a.mask, a.feat = _ptr(mask), pred.shape[-1]
    a.loss_type, a.huber_c = LOSS_TYPES[loss_type], huber_c
    if guard is not None:
        assert guard.dtype == torch.int32 and guard.numel() >= 8 and guard.is_contiguous()
        a.guard, a.max_loss = _ptr(guard), float(max_loss or 0.0)
    elif max_loss:
        raise ValueError("max_loss needs the guard buffer (the clamp is decided on the device)")
    _call("aitk_mse_loss_grad", C.byref(a))


def adamw_ema_step(p, g, m, v, *, lr, beta1, beta2, eps, weight_decay, step, max_norm=0.0, ema=None, ema_decay=0.0,
                   grad_scale=1.0, norm_out=None, ema_feedback=0.0, param_multiplier=1.0, guard=None, n_micro=1):
    """In-place clip_grad_norm_ -> AdamW -> EMA over flat fp32 arenas (step is the 1-based AdamW step count); ema_feedback /
    param_multiplier: toolkit/ema.py's use_feedback (10) and param_multiplier applied to the parameter after the EMA update.  guard
    (int32[8], AitkAdamWArgs.guard): the update is skipped on the device when the gradient norm is not finite or all n_micro loss launches
    of the step were gated; the step count of the bias corrections then lives in guard[3] (`step` is ignored)."""
    a = _capi.AdamWArgs()
    n = p.numel()
    for t_ in (p, g, m, v):
        assert t_.dtype == torch.float32 and t_.is_contiguous() and t_.numel() == n
    ws = workspace(_capi.lib().aitk_adamw_workspace_bytes(n), p.device, "adamw")
    a.p, a.g, a.m, a.v, a.ema = _ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(ema)
    a.norm_partial, a.norm_out, a.n = _ptr(ws), _ptr(norm_out), n
    a.lr, a.beta1, a.beta2, a.eps, a.weight_decay = lr, beta1, beta2, eps, weight_decay
    a.bias_correction1 = 1.0 - beta1 ** step
    a.bias_correction2_sqrt = math.sqrt(1.0 - beta2 ** step)
    a.beta1_d, a.beta2_d = float(beta1), float(beta2)
    a.max_norm, a.ema_decay, a.grad_scale = max_norm, ema_decay, grad_scale
    a.ema_feedback, a.param_multiplier = ema_feedback, param_multiplier
    if guard is not None:
        assert guard.dtype == torch.int32 and guard.numel() >= 8 and guard.is_contiguous()
        a.guard, a.n_micro = _ptr(guard), int(n_micro)
    _call("aitk_adamw_ema_step", C.byref(a))


def ema_update(p, ema, *, decay, ema_feedback=0.0, param_multiplier=1.0):
    """toolkit/ema.py:126-152 over the flat arenas in one launch (`decay` already min'ed with the num_updates warm-up by the caller):
    ema -= (1 - decay)(ema - p); p += ema_feedback * that; p *= param_multiplier."""
    assert p.dtype == torch.float32 and ema.dtype == torch.float32 and p.is_contiguous() and ema.is_contiguous() and p.numel() == ema.numel()
    _call("aitk_ema_update", _ptr(p), _ptr(ema), p.numel(), 1.0 - decay, float(ema_feedback), float(param_multiplier))


def make_shadow_table(entries, device):
    """entries: list of (src_off, rows, cols, kind, d0, d1, d2[, aux]) (AitkShadowDesc) -> device table for refresh_shadows."""
    arr = (_capi.ShadowDesc * len(entries))()
    for i, (so, r, c, kind, d0, d1, d2, *aux) in enumerate(entries):
        arr[i].src_off, arr[i].rows, arr[i].cols, arr[i].kind, arr[i].d0, arr[i].d1, arr[i].d2 = so, r, c, kind, d0, d1, d2
        arr[i].aux = aux[0] if aux else 0
    raw = bytes(arr)
    t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
    return t, len(entries)


def lokr_lowrank_grad(dw, a, b, ga, gb, *, accumulate=True):
    """ga [O, r] (+)= dw [O, I] @ b[r, I]^T, gb [r, I] (+)= a[O, r]^T @ dw: gradients of the low-rank LoKr pair from the gradient of
    their product (all fp32, contiguous)."""
    O, I = dw.shape
    r = a.shape[1]
    for t in (dw, a, b, ga, gb):
        assert t.dtype == torch.float32 and t.is_contiguous()
    assert tuple(a.shape) == (O, r) == tuple(ga.shape) and tuple(b.shape) == (r, I) == tuple(gb.shape)
    _call("aitk_lokr_lowrank_grad", _ptr(dw), _ptr(a), _ptr(b), _ptr(ga), _ptr(gb), O, I, r, int(accumulate))


def grad_compress_bf16(g, out):
    """out (bf16, flat) = round(g) for a slice of the fp32 gradient arena: transport format of the bf16 DP all-reduce"""
    assert g.dtype == torch.float32 and out.dtype == BF16 and g.is_contiguous() and out.is_contiguous() and out.numel() >= g.numel()
    _call("aitk_grad_compress_bf16", _ptr(g), _ptr(out), g.numel())
    return out


def grad_expand_bf16(src, g):
    """g (fp32 arena slice) = float(src) after the collective"""
    assert g.dtype == torch.float32 and src.dtype == BF16 and g.is_contiguous() and src.is_contiguous() and src.numel() >= g.numel()
    _call("aitk_grad_expand_bf16", _ptr(src), _ptr(g), g.numel())
    return g


def refresh_shadows(arena, shadow, table):
    tab, n = table
    _call("aitk_lora_refresh_shadows", _ptr(arena), _ptr(shadow), _ptr(tab), n)


# ---------------------------------------------------------------------------------------------------------- VAE encoder
EPI_BIAS_ROW, EPI_ADD_AUX = _capi.EPI_BIAS_ROW, _capi.EPI_ADD_AUX
_zero_pages = {}


def _zero_page(device):
    z = _zero_pages.get(str(device))
    if z is None:
        z = torch.zeros(64, dtype=BF16, device=device)
        _zero_pages[str(device)] = z
    return z


CONV_STAGE = 5 if os.environ.get("AITK_CONV8", "1") == "0" else 1  # AITK_CONV8=0: 3x3 convolutions stay on the 2-barrier kernel (same-box A/B)


def conv3x3(x, w, out, *, B, H, W, stride=1, pad_t=1, pad_l=1, Ho=None, Wo=None, bias=None, flags=0, aux_in=None, a2=None, b2=None,
            split_slab=False, col_scale=None, stage_mode=None):
    """Implicit-GEMM 3x3 convolution on NHWC: x [B*H*W, Cin], w [Cout, 9*Cin] (k = (ky*3+kx)*Cin + cin), out [B*Ho*Wo, Cout].
    a2 [M, K2] / b2 [Cout, K2]: LoRA K-slab added to the product (the lora_up of a conv adapter, fused like a Linear's).
    split_slab: w = 16-rank blocks [A_hi ; A_lo] of a rank-rp projection (2 rp rows, rp <= 64) and out is the [M, 3 rp] slab
    [hi | lo | hi] of the fp32 sum (the conv adapter's lora_down), scaled by the fp32 vector col_scale [2 rp]."""
    g = _capi.GemmArgs()
    Cin = x.shape[1]
    assert x.is_contiguous() and x.dtype == BF16 and x.shape[0] == B * H * W
    Ho = H if Ho is None else Ho
    Wo = W if Wo is None else Wo
    g.lda, g.ldb, g.ldc = Cin, _row_major(w, "w"), _row_major(out, "out")
    N, K = w.shape
    assert K == 9 * Cin and out.shape == (B * Ho * Wo, 3 * (N // 2) if split_slab else N)
    g.A, g.B, g.C = _ptr(x), _ptr(w), _ptr(out)
    if split_slab:
        assert N % 32 == 0 and N <= 128 and bias is None and aux_in is None and a2 is None
        flags |= _capi.EPI_SPLIT_SLAB
    if col_scale is not None:
        assert col_scale.dtype == torch.float32 and col_scale.is_contiguous() and col_scale.numel() == N
        flags |= _capi.EPI_COL_SCALE
        g.col_scale = _ptr(col_scale)
    if a2 is not None:
        assert a2.shape[0] >= B * Ho * Wo and b2.shape[0] == N and a2.shape[1] == b2.shape[1]
        g.A2, g.lda2, g.B2, g.ldb2, g.K2 = _ptr(a2), _row_major(a2, "a2"), _ptr(b2), _row_major(b2, "b2"), a2.shape[1]
    if bias is not None:
        flags |= EPI_BIAS
        g.bias = _ptr(bias)
    if aux_in is not None:
        g.aux_in, g.ld_aux_in = _ptr(aux_in), _row_major(aux_in, "aux_in")
    g.M, g.N, g.K, g.flags = B * Ho * Wo, N, K, flags
    # stage_mode 1: the persistent 8-phase kernel's conv mode for big problems (Cin % 64 == 0, >= half a chip of 256^2 tiles), else the
    # 2-barrier kernel; 4 forces the former where its contract allows, 5 the latter
    g.stage_mode, g.tile_mode = (CONV_STAGE if stage_mode is None else stage_mode), TILE_MODE
    g.conv_mode, g.conv_H, g.conv_W, g.conv_Cin = 1, H, W, Cin
    g.conv_Wo, g.conv_HoWo, g.conv_stride, g.conv_pad_t, g.conv_pad_l = Wo, Ho * Wo, stride, pad_t, pad_l
    g.zero_page = _ptr(_zero_page(x.device))
    _call("aitk_gemm_nt", C.byref(g))
    return out


def conv3d(x, w, out, *, T, H, W, kt=3, ks=3, tstride=1, stride=1, pad_t=1, pad_l=1, Ho=None, Wo=None, bias=None, flags=0, aux_in=None):
    """Implicit-GEMM 3-D convolution on frames of NHWC rows (Wan2.1 video VAE): x [>= ((T-1)*tstride + kt)*H*W, Cin] whose frame
    t*tstride + dt feeds temporal tap dt of output frame t (a causal convolution passes a buffer that starts with its kt-1 zero frames),
    w [Cout, kt*ks*ks*Cin] with k = ((dt*ks + ky)*ks + kx)*Cin + cin, out [T*Ho*Wo, Cout]."""
    g = _capi.GemmArgs()
    Cin = x.shape[1]
    Ho = H if Ho is None else Ho
    Wo = W if Wo is None else Wo
    assert x.is_contiguous() and x.dtype == BF16 and x.shape[0] >= ((T - 1) * tstride + kt) * H * W
    g.lda, g.ldb, g.ldc = Cin, _row_major(w, "w"), _row_major(out, "out")
    N, K = w.shape
    assert K == kt * ks * ks * Cin and out.shape == (T * Ho * Wo, N)
    g.A, g.B, g.C = _ptr(x), _ptr(w), _ptr(out)
    if bias is not None:
        flags |= EPI_BIAS
        g.bias = _ptr(bias)
    if aux_in is not None:
        g.aux_in, g.ld_aux_in = _ptr(aux_in), _row_major(aux_in, "aux_in")
    g.M, g.N, g.K, g.flags = T * Ho * Wo, N, K, flags
    g.stage_mode, g.tile_mode = 1, TILE_MODE
    g.conv_mode, g.conv_H, g.conv_W, g.conv_Cin = 1, H, W, Cin
    g.conv_Wo, g.conv_HoWo, g.conv_stride, g.conv_pad_t, g.conv_pad_l = Wo, Ho * Wo, stride, pad_t, pad_l
    g.conv_t3d = kt | (tstride << 8) | (ks << 16)
    g.zero_page = _ptr(_zero_page(x.device))
    _call("aitk_gemm_nt", C.byref(g))
    return out


def rmsnorm_rows(x, gamma, out, *, eps=1e-12, silu=False):
    """out = x / max(||x||_2, eps) * sqrt(C) * gamma (+ SiLU) per row of x [M, C] (WanRMS_norm on NHWC rows); out may be x."""
    assert x.dtype == BF16 and out.dtype == BF16 and gamma.dtype == BF16 and gamma.is_contiguous() and gamma.numel() == x.shape[1]
    _call("aitk_rmsnorm_rows", _ptr(x), _row_major(x, "x"), _ptr(out), _row_major(out, "out"), _ptr(gamma), x.shape[0], x.shape[1],
          float(eps), int(silu))
    return out


def latent_sample_affine(moments, eps, out, *, ch_shift, ch_scale):
    """out [B, L, ...] = ch_scale[c] * (mean + exp(0.5 clamp(logvar)) * eps - ch_shift[c]); moments rows ordered like out's trailing dims."""
    B, L = out.shape[0], out.shape[1]
    hw = out[0, 0].numel()
    assert eps.dtype == torch.float32 and eps.is_contiguous() and eps.shape == out.shape and out.is_contiguous()
    for t in (ch_shift, ch_scale):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == L
    _call("aitk_latent_sample_affine", _ptr(moments), _row_major(moments, "moments"), _ptr(eps), _ptr(out), B, L, hw, _ptr(ch_shift),
          _ptr(ch_scale))
    return out


def groupnorm(x, gamma, beta, out, *, B, HW, G=32, eps=1e-6, silu=False, stats_out=None):
    """out = GroupNorm_G(x [B*HW, C]) * gamma + beta (+ SiLU); stats_out fp32 [B, G, 2] (mean, rstd) for groupnorm_bwd."""
    a = _capi.GroupNormArgs()
    if stats_out is not None:
        assert stats_out.dtype == torch.float32 and stats_out.is_contiguous() and stats_out.numel() == 2 * B * G
        a.stats_out = _ptr(stats_out)
    Cc = x.shape[1]
    a.x, a.ldx, a.y, a.ldy = _ptr(x), _row_major(x, "x"), _ptr(out), _row_major(out, "out")
    a.gamma, a.beta = _ptr(gamma), _ptr(beta)
    ws = workspace(_capi.lib().aitk_groupnorm_workspace_bytes(B, HW, Cc, G), x.device, "groupnorm")
    a.partial = _ptr(ws)
    a.eps, a.silu, a.B, a.HW, a.C, a.G = eps, int(silu), B, HW, Cc, G
    _call("aitk_groupnorm", C.byref(a))
    return out


def softmax_rows(x, scale):
    ld = _row_major(x, "x")
    _call("aitk_softmax_rows", _ptr(x), ld, x.shape[0], x.shape[1], float(scale))
    return x


def image_to_nhwc8(img, out):
    B, Cc, H, W = img.shape
    assert Cc == 3 and img.dtype == torch.float32 and img.is_contiguous() and out.shape == (B * H * W, 8)
    _call("aitk_image_to_nhwc8", _ptr(img), _ptr(out), B, H, W)
    return out


def image_resize_to_nhwc8(img, out, *, Hd, Wd):
    """bilinear (align_corners=False) resize of [B,3,Hs,Ws] fp32 images to [Hd, Wd], written as NHWC rows with 8 channels (bf16)."""
    B, Cc, Hs, Ws = img.shape
    assert Cc == 3 and img.dtype == torch.float32 and img.is_contiguous() and out.shape == (B * Hd * Wd, 8)
    _call("aitk_image_resize_to_nhwc8", _ptr(img), _ptr(out), B, Hs, Ws, Hd, Wd)
    return out


def latent_sample(moments, eps, out, *, scale, shift):
    B, L, h, w = out.shape
    assert eps.dtype == torch.float32 and eps.is_contiguous() and eps.shape == out.shape and out.is_contiguous()
    _call("aitk_latent_sample", _ptr(moments), _row_major(moments, "moments"), _ptr(eps), _ptr(out), B, L, h * w,
                                               float(scale), float(shift))
    return out


# ---------------------------------------------------------------------------------------------------------- Wan2.1
def _rms_full_args(x, weight, y, cos, sin, S, eps):
    a = _capi.RmsFullArgs()
    a.x, a.ldx, a.y, a.ldy = _ptr(x), _row_major(x, "x"), _ptr(y), _row_major(y, "y")
    a.weight, a.cos, a.sin = _ptr(weight), _ptr(cos), _ptr(sin)
    a.eps, a.S, a.M, a.C = eps, S, x.shape[0], x.shape[1]
    return a


def rms_full_fwd(x, weight, y, *, S, cos=None, sin=None, eps=1e-6):
    """y = rope(RMSNorm_C(x) * weight): norm across all heads of a token (C = H*128), rope on (2i,2i+1) pairs per head."""
    a = _rms_full_args(x, weight, y, cos, sin, S, eps)
    _call("aitk_rms_full_fwd", C.byref(a))
    return y


def rms_full_bwd(g, x, weight, dx, *, S, cos=None, sin=None, eps=1e-6):
    a = _rms_full_args(x, weight, dx, cos, sin, S, eps)
    a.g, a.ldg = _ptr(g), _row_major(g, "g")
    _call("aitk_rms_full_bwd", C.byref(a))
    return dx


# ---------------------------------------------------------------------------------------------------------- DoRA
def dora_colscale(w2, tw, up, gram, mag, s, c_out):
    """c = magnitude / ||W + s*B A||_row from ||W||^2, tw = W A^T [N,R] (bf16), B fp32 [N,R], gram = A A^T fp32 [R,R]."""
    a = _capi.DoraColscaleArgs()
    N, R = up.shape
    assert up.dtype == torch.float32 and up.is_contiguous() and gram.dtype == torch.float32 and gram.is_contiguous()
    assert w2.dtype == torch.float32 and mag.dtype == torch.float32 and c_out.dtype == torch.float32 and tw.dtype == BF16
    a.w2, a.tw, a.ldtw, a.up, a.gram, a.mag, a.c = _ptr(w2), _ptr(tw), tw.stride(0), _ptr(up), _ptr(gram), _ptr(mag), _ptr(c_out)
    a.s, a.N, a.R = float(s), N, R
    _call("aitk_dora_colscale", C.byref(a))
    return c_out


def dora_bwd(dy, y, c, bias, mag, dz, dmag, *, M):
    """dz = c * dy;  dmag += (colsum(dy*y) - bias * colsum(dy)) / mag   (y = c*z + bias is this step's linear output)."""
    a = _capi.DoraBwdArgs()
    N = dz.shape[1]
    a.dy, a.ld_dy, a.y, a.ld_y, a.dz, a.ld_dz = _ptr(dy), _row_major(dy, "dy"), _ptr(y), _row_major(y, "y"), _ptr(dz), _row_major(dz, "dz")
    assert c.dtype == torch.float32 and mag.dtype == torch.float32 and dmag.dtype == torch.float32 and dmag.is_contiguous()
    nchunk = (M + 15) // 16
    part = workspace(2 * nchunk * N * 4, dz.device, "dora_bwd")
    a.c, a.bias, a.mag, a.dmag, a.partial = _ptr(c), _ptr(bias), _ptr(mag), _ptr(dmag), _ptr(part)
    a.M, a.N = M, N
    _call("aitk_dora_bwd", C.byref(a))
    return dz


def kron_apply(x, A, Bm, out, *, a_in, b_in, a_out, b_out, scale=1.0, transpose_out=False, accumulate=False, col0=0, ncols=0,
               x_seg=None, out_seg=None, M=None):
    """LoKr per-token product: out[m] (viewed [a_out, b_out], or [b_out, a_out] when transpose_out) (+)=
    scale * A[a_out,a_in] @ x[m].view(a_in, b_in) @ Bm[b_out,b_in]^T;  A / Bm None = identity.  Only columns
    [col0, col0+ncols) of each output row are written (ncols 0: all) — `out` is then the destination window."""
    a = _capi.KronApplyArgs()
    a.ldx = _row_major(x, "x")
    a.ldo = _row_major(out, "out")
    assert x.shape[1] == a_in * b_in, (tuple(x.shape), a_in, b_in)
    for mat, r, c in ((A, a_out, a_in), (Bm, b_out, b_in)):
        if mat is not None:
            assert mat.dtype == BF16 and mat.is_contiguous() and tuple(mat.shape) == (r, c), (tuple(mat.shape), r, c)
    a.x, a.A, a.B, a.out = _ptr(x), _ptr(A), _ptr(Bm), _ptr(out)
    if x_seg is not None:
        a.x_seg_rows, a.x_seg_stride = x_seg
    if out_seg is not None:
        a.out_seg_rows, a.out_seg_stride = out_seg
    a.M = x.shape[0] if M is None else M
    a.a_in, a.b_in, a.a_out, a.b_out = a_in, b_in, a_out, b_out
    assert out.shape[1] == (ncols if ncols else a_out * b_out)
    a.transpose_out, a.accumulate, a.col0, a.ncols, a.scale = int(transpose_out), int(accumulate), col0, ncols, float(scale)
    _call("aitk_kron_apply", C.byref(a))
    return out


def kron_merge(W, A, Bm, alpha):
    """W (bf16 [a_rows*b_rows, a_cols*b_cols]) += alpha * kron(A, Bm) with fp32 factors (LoKr merge_in)."""
    assert W.dtype == BF16 and W.dim() == 2 and W.stride(1) == 1
    assert A.dtype == torch.float32 and Bm.dtype == torch.float32 and A.is_contiguous() and Bm.is_contiguous()
    assert W.shape[0] == A.shape[0] * Bm.shape[0] and W.shape[1] == A.shape[1] * Bm.shape[1]
    _call("aitk_kron_merge", _ptr(W), W.stride(0), _ptr(A), _ptr(Bm), A.shape[0], A.shape[1], Bm.shape[0], Bm.shape[1],
                                            float(alpha))
    return W


def dequant_fp8(q, scale, mode, out):
    """out (bf16 [rows, cols]) = e4m3(q) * scale (per row: mode 1, per column: mode 2)."""
    assert q.element_size() == 1 and q.dim() == 2 and q.stride(1) == 1 and out.dtype == BF16 and out.shape == q.shape and out.stride(1) == 1
    assert scale.dtype == torch.float32 and scale.numel() == (q.shape[0] if mode == 1 else q.shape[1])
    _call("aitk_dequant_fp8", _ptr(q), q.stride(0), _ptr(scale), mode, _ptr(out), out.stride(0), q.shape[0], q.shape[1])
    return out


# ---------------------------------------------------------------------------------------------------------- UNet (SD1.5 / SDXL)
def groupnorm_bwd(dy, x, gamma, beta, stats, dx, *, B, HW, G=32, silu=False, dres=None):
    """dx = backward of act(GroupNorm_G(x) * gamma + beta) (+ dres); stats = the forward's stats_out."""
    a = _capi.GroupNormBwdArgs()
    Cc = x.shape[1]
    a.dy, a.ld_dy, a.x, a.ldx, a.dx, a.ld_dx = _ptr(dy), _row_major(dy, "dy"), _ptr(x), _row_major(x, "x"), _ptr(dx), _row_major(dx, "dx")
    assert stats.dtype == torch.float32 and stats.is_contiguous() and stats.numel() == 2 * B * G
    a.gamma, a.beta, a.stats = _ptr(gamma), _ptr(beta), _ptr(stats)
    if dres is not None:
        a.dres, a.ld_dres = _ptr(dres), _row_major(dres, "dres")
    ws = workspace(_capi.lib().aitk_groupnorm_bwd_workspace_bytes(B, HW, Cc, G), x.device, "groupnorm_bwd")
    a.partial = _ptr(ws)
    a.silu, a.B, a.HW, a.C, a.G = int(silu), B, HW, Cc, G
    _call("aitk_groupnorm_bwd", C.byref(a))
    return dx


def geglu_fwd(hg, out):
    """out[M, F] = hg[:, :F] * gelu_erf(hg[:, F:])  (diffusers GEGLU)."""
    M, F2 = hg.shape
    assert out.shape == (M, F2 // 2)
    _call("aitk_geglu_fwd", _ptr(hg), _row_major(hg, "hg"), _ptr(out), _row_major(out, "out"), M, F2 // 2)
    return out


def geglu_bwd(dy, hg, dhg):
    M, F2 = hg.shape
    assert dy.shape == (M, F2 // 2) and dhg.shape == hg.shape
    _call("aitk_geglu_bwd", _ptr(dy), _row_major(dy, "dy"), _ptr(hg), _row_major(hg, "hg"), _ptr(dhg), _row_major(dhg, "dhg"),
                                           M, F2 // 2)
    return dhg


def resample2x(src, dst, *, B, H, W, mode):
    """contiguous NHWC [B*H*W, C] -> mode 0 nearest up [B*2H*2W, C]; 1: 2x2 sum [B*(H/2)*(W/2), C]; 2: zero insertion [B*2H*2W, C]."""
    Cc = src.shape[1]
    assert src.is_contiguous() and dst.is_contiguous() and src.dtype == BF16 and dst.dtype == BF16 and src.shape[0] == B * H * W
    assert dst.shape == ((B * (H // 2) * (W // 2), Cc) if mode == 1 else (B * 4 * H * W, Cc))
    _call("aitk_resample2x", _ptr(src), _ptr(dst), B, H, W, Cc, mode)
    return dst


def pad_nhwc(src, dst, *, B, H, W):
    """dst [B*(H+2)*(W+2), C] = src [B*H*W, C] inside a one-pixel zero border."""
    Cc = src.shape[1]
    assert src.is_contiguous() and dst.is_contiguous() and src.shape[0] == B * H * W and dst.shape == (B * (H + 2) * (W + 2), Cc)
    _call("aitk_pad_nhwc", _ptr(src), _ptr(dst), B, H, W, Cc)
    return dst


def copy_heads(src, dst, *, H, d_src, d_dst):
    """dst[m, h*d_dst + j] = src[m, h*d_src + j] (j < d_src) else 0, j < d_dst."""
    M = src.shape[0]
    assert dst.shape[0] == M and src.shape[1] >= H * d_src and dst.shape[1] >= H * d_dst
    _call("aitk_copy_heads", _ptr(src), _row_major(src, "src"), _ptr(dst), _row_major(dst, "dst"), M, H, d_src, d_dst)
    return dst


def ddpm_noise_nhwc(latents, noise, a, s, noisy, target, *, v_prediction=False):
    """latents / noise NCHW [B,C,h,w] bf16; a, s fp32 [B]; noisy [B*h*w, Cp] (zero-padded channels), target [B*h*w, C]."""
    g = _capi.DdpmNoiseArgs()
    B, Cc, h, w = latents.shape
    assert latents.is_contiguous() and noise.is_contiguous() and noisy.is_contiguous() and target.is_contiguous()
    assert latents.dtype == BF16 and noise.dtype == BF16 and a.dtype == torch.float32 and s.dtype == torch.float32
    assert noisy.shape[0] == B * h * w and target.shape == (B * h * w, Cc)
    g.latents, g.noise, g.a, g.s, g.noisy, g.target = _ptr(latents), _ptr(noise), _ptr(a), _ptr(s), _ptr(noisy), _ptr(target)
    g.B, g.C, g.HW, g.Cp, g.mode = B, Cc, h * w, noisy.shape[1], int(v_prediction)
    _call("aitk_ddpm_noise_nhwc", C.byref(g))


def attn_small_fwd(q, k, v, o, lse, *, B, H, S, D, scale, Skv=0):
    """generic attention (head_dim D > 128): q, o [B*S, >= H*D]; k, v [B*Skv, >= H*D]; lse [B,H,S] natural log."""
    a = _attn_args(q, k, v, o, lse, B, H, S, scale, Skv)
    a.D = D
    _call("aitk_attn_small_fwd", C.byref(a))
    return o


def attn_small_bwd(q, k, v, o, lse, do, dq, dk, dv, *, B, H, S, D, scale, Skv=0):
    a = _attn_args(q, k, v, o, lse, B, H, S, scale, Skv)
    a.D = D
    a.dO, a.lddo = _ptr(do), _row_major(do, "do")
    a.dQ, a.dK, a.dV = _ptr(dq), _ptr(dk), _ptr(dv)
    a.lddq, a.lddk, a.lddv = _row_major(dq, "dq"), _row_major(dk, "dk"), _row_major(dv, "dv")
    a.delta = _ptr(workspace(B * H * S * 4, q.device, "attn_delta"))
    _call("aitk_attn_small_bwd", C.byref(a))
