"""Epilogue forms of the persistent 8-phase GEMM (gemm8.hip; AITK_GEMM8_FE = 0: the generic epilogue only, 1: the fast forms, default): bit-exactness of the
fast forms against the generic one over the epilogues, ragged shapes and segmented row maps, time(K) fits (fixed cost per tile round), the FLUX shapes with their
epilogues, and the s_memtime trace of the tile switch (AITK_GEMM8_TRACE=1).  Prints JSON lines.   python tools/gpu_gemm8_ev.py [check] [sweep] [trace]
(In the JSON keys "ev" is the variant: 0 generic epilogue, 1 fast epilogue forms.)"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ai_toolkit_amd  # noqa: E402,F401
from ai_toolkit_amd import _capi, ops  # noqa: E402

bf = torch.bfloat16
EVS = [int(x) for x in os.environ.get("AITK_EVS", "0,1").split(",")]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def set_ev(ev, trace=0):
    """variant 0: generic epilogue; 1: fast epilogue forms (the default build)"""
    os.environ["AITK_GEMM8_FE"] = "0" if ev == 0 else "1"
    os.environ["AITK_GEMM8_TRACE"] = str(trace)


def cases(M, N, K, g):
    """(name, kwargs builder) for every epilogue the FLUX / UNet graphs use on this kernel."""
    x = torch.randn(M, K, device="cuda", generator=g).to(bf)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(bf)
    a2 = (torch.randn(M, 48, device="cuda", generator=g) * 0.1).to(bf)
    b2 = (torch.randn(N, 48, device="cuda", generator=g) * 0.1).to(bf)
    bias = torch.randn(N, device="cuda", generator=g).to(bf)
    aux = torch.randn(M, N, device="cuda", generator=g).to(bf)
    gate = torch.randn((M + 4607) // 4608, N, device="cuda", generator=g).to(bf)
    cs = torch.rand(N, device="cuda", generator=g) + 0.5
    slab = dict(a2=a2, b2=b2)
    return x, w, {
        "plain": lambda o: dict(),
        "slab": lambda o: dict(**slab),
        "bias+slab": lambda o: dict(bias=bias, **slab),
        "gelu+bias+slab": lambda o: dict(bias=bias, flags=_capi.EPI_GELU, aux_out=torch.empty_like(o), **slab),
        "dgelu+slab": lambda o: dict(flags=_capi.EPI_DGELU, aux_in=aux, **slab),
        "gate_res+bias+slab": lambda o: dict(bias=bias, flags=_capi.EPI_GATE_RES, aux_in=aux, aux_out=torch.empty_like(o), gate=gate, gate_rows=4608, **slab),
        "accum+slab": lambda o: dict(flags=_capi.EPI_ACCUM, **slab),
        "add_aux+bias": lambda o: dict(bias=bias, flags=_capi.EPI_ADD_AUX, aux_in=aux),
        "col_scale+bias+slab": lambda o: dict(bias=bias, col_scale=cs, **slab),
    }


def check():
    g = torch.Generator(device="cuda").manual_seed(3)
    bad = 0
    for (M, N, K) in ((32256, 3072, 3072), (32100, 3080, 1040), (9216, 3072, 256), (32256, 12288, 3072)):
        x, w, cs = cases(M, N, K, g)
        for name, mk in cs.items():
            outs = {}
            for ev in EVS:
                set_ev(ev)
                o = torch.full((M, N), 0.25, dtype=bf, device="cuda")
                kw = mk(o)
                ops.gemm_nt(x, w, o, **kw)
                torch.cuda.synchronize()
                outs[ev] = (o, kw.get("aux_out"))
            for ev in EVS[1:]:
                same = torch.equal(outs[ev][0], outs[EVS[0]][0]) and (outs[ev][1] is None or torch.equal(outs[ev][1], outs[EVS[0]][1]))
                if not same:
                    bad += 1
                    d = (outs[ev][0].float() - outs[EVS[0]][0].float()).abs()
                    print(json.dumps({"check": name, "shape": [M, N, K], "ev": ev, "equal": False, "max_abs": d.max().item(), "n_diff": int((d > 0).sum())}))
        del x, w, cs
        torch.cuda.empty_cache()
        print(json.dumps({"check_shape": [M, N, K], "mismatches_so_far": bad}), flush=True)
    # segmented row maps (the image / text halves of the joint attention buffers): out rows of sample b live at b * (Si + St) rows of a joint buffer
    g = torch.Generator(device="cuda").manual_seed(6)
    B, Si, St, N, K = 7, 4096, 512, 3072, 3072
    for (rows, off) in ((Si, St), (St, 0)):
        M = B * rows
        x, w, cs = cases(M, N, K, g)
        gate_b = torch.randn(B, N, device="cuda", generator=g).to(bf)
        for name in ("bias+slab", "gate_res+bias+slab", "slab"):
            outs = {}
            for ev in EVS:
                set_ev(ev)
                joint = torch.full((B * (Si + St), N), 0.5, dtype=bf, device="cuda")
                o = joint[off:off + rows]
                kw = cs[name](torch.empty(M, N, dtype=bf, device="cuda"))
                if "gate_rows" in kw:
                    kw["gate_rows"], kw["gate"] = rows, gate_b
                ops.gemm_nt(x, w, o, c_seg=(rows, (Si + St) * N), M=M, **kw)
                torch.cuda.synchronize()
                outs[ev] = (joint, kw.get("aux_out"))
            for ev in EVS[1:]:
                same = torch.equal(outs[ev][0], outs[EVS[0]][0]) and (outs[ev][1] is None or torch.equal(outs[ev][1], outs[EVS[0]][1]))
                bad += 0 if same else 1
                print(json.dumps({"check": "c_seg " + name, "rows": rows, "ev": ev, "equal": same}), flush=True)
        del x, w, cs
        torch.cuda.empty_cache()
    # the two-problem launch (image + text stream)
    g = torch.Generator(device="cuda").manual_seed(4)
    N, K = 3072, 3072
    xs = [torch.randn(m, K, device="cuda", generator=g).to(bf) for m in (28672, 3584)]
    w2 = [(torch.randn(N, K, device="cuda", generator=g) * 0.02).to(bf) for _ in range(2)]
    bs = [torch.randn(N, device="cuda", generator=g).to(bf) for _ in range(2)]
    res = {}
    for ev in EVS:
        set_ev(ev)
        os_ = [torch.zeros(x.shape[0], N, dtype=bf, device="cuda") for x in xs]
        with ops.recording() as la:
            ops.gemm_nt(xs[0], w2[0], os_[0], bias=bs[0])
        with ops.recording() as lb:
            ops.gemm_nt(xs[1], w2[1], os_[1], bias=bs[1])
        ops.replay_paired(la, lb)
        torch.cuda.synchronize()
        res[ev] = os_
    for ev in EVS[1:]:
        ok = all(torch.equal(res[ev][i], res[EVS[0]][i]) for i in range(2))
        bad += 0 if ok else 1
        print(json.dumps({"check": "grouped bias", "ev": ev, "equal": ok}))
    print(json.dumps({"check_total_mismatches": bad}), flush=True)
    return bad


def sweep():
    g = torch.Generator(device="cuda").manual_seed(1)
    M, N = 32256, 3072
    a2 = (torch.randn(M, 48, device="cuda", generator=g) * 0.1).to(bf)
    b2 = (torch.randn(N, 48, device="cuda", generator=g) * 0.1).to(bf)
    o = torch.empty(M, N, dtype=bf, device="cuda")
    res = {ev: {} for ev in EVS}
    ks = (1024, 2048, 3072, 6144, 12288)
    for K in ks:
        x = torch.randn(M, K, device="cuda", generator=g).to(bf)
        w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(bf)
        for ev in EVS:
            set_ev(ev)
            res[ev][f"K{K}"] = round(timeit(lambda: ops.gemm_nt(x, w, o, a2=a2, b2=b2)), 4)
            res[ev][f"K{K}_noslab"] = round(timeit(lambda: ops.gemm_nt(x, w, o)), 4)
        del x, w
    for ev in EVS:
        for tag in ("", "_noslab"):
            ts = [res[ev][f"K{k}{tag}"] for k in ks]
            n = len(ks)
            mx, my = sum(ks) / n, sum(ts) / n
            slope = sum((k - mx) * (t - my) for k, t in zip(ks, ts)) / sum((k - mx) ** 2 for k in ks)
            icpt = my - slope * mx
            res[ev]["fit" + tag] = {"fixed_us_per_tile_round": round(1e3 * icpt / 6, 2), "us_per_ktile": round(1e3 * slope * 64 / 6, 3)}
        print(json.dumps({"ksweep_ev": ev, **res[ev]}), flush=True)
    # the FLUX K = 3072 launches with their epilogues (B = 7: M = 32256) and the two big-K ones
    for (M, N, K, names) in ((32256, 3072, 3072, ("bias+slab", "gate_res+bias+slab", "accum+slab")), (32256, 12288, 3072, ("gelu+bias+slab", "dgelu+slab")),
                             (32256, 3072, 12288, ("gate_res+bias+slab",)), (32256, 3072, 9216, ("slab",)), (32256, 3072, 21504, ("slab",))):
        x, w, cs = cases(M, N, K, g)
        o = torch.empty(M, N, dtype=bf, device="cuda")
        for name in names:
            kw = cs[name](o)
            row = {"shape": [M, N, K], "epi": name}
            for ev in EVS:
                set_ev(ev)
                ms = timeit(lambda: ops.gemm_nt(x, w, o, **kw))
                row[f"ev{ev}_ms"] = round(ms, 4)
                row[f"ev{ev}_tflops"] = round(2.0 * M * N * (K + 48) / ms / 1e9, 1)
            print(json.dumps(row), flush=True)
        del x, w, cs, o
        torch.cuda.empty_cache()


def trace():
    g = torch.Generator(device="cuda").manual_seed(2)
    M, N, K = 32256, 3072, 3072
    x, w, cs = cases(M, N, K, g)
    o = torch.empty(M, N, dtype=bf, device="cuda")
    L = _capi.lib()
    L.aitk_probe_gemm8_trace.argtypes = [C.c_void_p]
    names = ["top wait + barrier", "steady K loop (46 K-tiles)", "tail K loop (2 K-tiles + slab)", "epilogue", "-> next top"]
    for epi in ("bias+slab", "gate_res+bias+slab"):
        kw = cs[epi](o)
        for ev in EVS:
            set_ev(ev, 1)
            for _ in range(3):
                ops.gemm_nt(x, w, o, **kw)
            torch.cuda.synchronize()
            buf = (C.c_uint32 * 60)()
            assert L.aitk_probe_gemm8_trace(buf) == 0
            v = list(buf)
            for wg in range(2):
                for wv in range(2):
                    base = (wg * 2 + wv) * 15
                    rows = []
                    for t in range(3):
                        s = v[base + t * 5: base + t * 5 + 5]
                        d = [(s[i + 1] - s[i]) & 0xffffffff for i in range(4)]
                        nxt = ((v[base + (t + 1) * 5] - s[4]) & 0xffffffff) if t < 2 else None
                        rows.append(d + [nxt])
                    print(json.dumps({"trace_epi": epi, "ev": ev, "wg": [0, 100][wg], "wave": [0, 4][wv], "ticks_10ns": rows, "legend": names}), flush=True)
    set_ev(0, 0)


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "sweep", "trace"]
    rc = 0
    if "check" in what:
        rc = check()
    if "sweep" in what:
        sweep()
    if "trace" in what:
        trace()
    sys.exit(1 if rc else 0)
