"""GEMM tile-config A/B (interleaved rounds in one process, guide rule 24) + correctness of the 256x256 config."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ai_toolkit_amd  # noqa: E402,F401
from ai_toolkit_amd import ops  # noqa: E402
from tools import gpu_check as g  # noqa: E402

dev = "cuda"
OUT = {}


def correctness():
    import functools
    res = {}
    orig = ops.gemm_nt
    try:
      for tag, part in (("t256", dict(tile_mode=2)), ("ring3", dict(stage_mode=2))):
        def patched(a, b, out, **kw):
            kw.pop("stage_mode", None)
            kw.update(part)
            return orig(a, b, out, **kw)
        ops.gemm_nt = patched
        for name, fn in {
            "c256_plain": lambda: g.gemm_case(512, 512, 128, stage=1),
            "c256_k64": lambda: g.gemm_case(512, 512, 64, stage=1),
            "c256_k64_r16": lambda: g.gemm_case(300, 200, 64, r=16, stage=1),
            "c256_ragged": lambda: g.gemm_case(700, 1000, 192, r=16, stage=1),
            "c256_ragged_s0": lambda: g.gemm_case(700, 1000, 192, r=16, stage=0),
            "c256_gelu": lambda: g.gemm_case(512, 768, 128, r=16, flags=ops.EPI_GELU, stage=1),
            "c256_gate": lambda: g.gemm_case(512, 768, 128, r=16, flags=ops.EPI_GATE_RES, stage=1),
            "c256_dgelu_acc": lambda: g.gemm_case(512, 768, 128, r=48, flags=ops.EPI_ACCUM, stage=1),
            "c256_seg": lambda: g.gemm_case(600, 512, 128, r=16, seg=True, stage=1),
            "c256_big": lambda: g.gemm_case(2048, 3072, 3072, r=16, stage=1),
        }.items():
            try:
                res[tag + "_" + name] = fn()
            except Exception as e:  # noqa: BLE001
                res[tag + "_" + name] = {"ok": False, "error": repr(e)}
            print(tag, name, res[tag + "_" + name], flush=True)
    finally:
        ops.gemm_nt = orig
    return res


def ab(M, N, K, rounds=5, iters=10, r=16):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    a2 = torch.randn(M, r, device=dev).to(torch.bfloat16)
    b2 = torch.randn(N, r, device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    variants = {"t128": dict(tile_mode=1, stage_mode=1), "t256": dict(tile_mode=2, stage_mode=1), "ring3_256x128": dict(stage_mode=2)}
    res = {k: [] for k in variants}
    for k, kw in variants.items():
        for _ in range(3):
            ops.gemm_nt(a, b, out, bias=bias, a2=a2, b2=b2, **kw)
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, kw in variants.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                ops.gemm_nt(a, b, out, bias=bias, a2=a2, b2=b2, **kw)
            e1.record()
            torch.cuda.synchronize()
            res[k].append(2.0 * M * N * K / (e0.elapsed_time(e1) / iters) / 1e9)
    out_ = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    out_["ok"] = True
    return out_


if __name__ == "__main__":
    OUT["correctness"] = correctness()
    # race screen for the counted-vmcnt ring: repeat larger cases, every run must be clean
    import functools
    orig = ops.gemm_nt
    def ring(a, b, out, **kw):
        kw.pop("stage_mode", None)
        return orig(a, b, out, stage_mode=2, **kw)
    ops.gemm_nt = ring
    try:
        for i in range(6):
            for (M, N, K) in ((4608, 3072, 3072), (1024, 12288, 3072), (3000, 3072, 1024)):
                r_ = g.gemm_case(M, N, K, r=16, stage=1, seed=i)
                OUT["correctness"][f"race_{i}_{M}x{N}x{K}"] = r_
                if not r_["ok"]:
                    print("RACE/ERR", i, M, N, K, r_, flush=True)
    finally:
        ops.gemm_nt = orig
    for (M, N, K) in ((4608, 3072, 3072), (4608, 12288, 3072), (4608, 3072, 12288), (18432, 3072, 3072), (18432, 12288, 3072),
                      (18432, 3072, 12288), (18432, 3072, 15360), (4096, 3072, 3072), (2048, 3072, 3072), (9216, 3072, 3072)):
        OUT[f"ab_{M}x{N}x{K}"] = ab(M, N, K)
        print(M, N, K, OUT[f"ab_{M}x{N}x{K}"], flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(OUT, open("gpurun_out/gemm_ab.json", "w"), indent=1)
    print("BAD:", [k for k, v in OUT["correctness"].items() if not v.get("ok")])
