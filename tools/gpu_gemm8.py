"""8-phase GEMM schedule (stage_mode=4): correctness + race screen vs fp32 reference, then interleaved A/B vs the 2-barrier
256x256 kernel (stage_mode=1) and torch.matmul (hipBLASLt) on the step's shapes."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ai_toolkit_amd  # noqa: E402,F401
from ai_toolkit_amd import ops  # noqa: E402
from tools import gpu_check as g  # noqa: E402

dev = "cuda"
OUT = {"correctness": {}}


def run_cases(stage):
    orig = ops.gemm_nt

    def patched(a, b, out, **kw):
        kw.pop("stage_mode", None)
        return orig(a, b, out, stage_mode=stage, tile_mode=2, **kw)

    ops.gemm_nt = patched
    bad = []
    try:
        cases = {
            "k64": lambda s: g.gemm_case(512, 512, 64, stage=1, seed=s),
            "k128": lambda s: g.gemm_case(512, 512, 128, stage=1, seed=s),
            "k192_r16": lambda s: g.gemm_case(300, 200, 192, r=16, stage=1, seed=s),
            "k80tail_r16": lambda s: g.gemm_case(700, 1000, 80, r=16, stage=1, seed=s),
            "k64_r48": lambda s: g.gemm_case(260, 516, 64, r=48, stage=1, seed=s),
            "ragged": lambda s: g.gemm_case(700, 1000, 208, r=16, stage=1, seed=s),
            "gelu": lambda s: g.gemm_case(512, 768, 128, r=16, flags=ops.EPI_GELU, stage=1, seed=s),
            "gate": lambda s: g.gemm_case(512, 768, 128, r=16, flags=ops.EPI_GATE_RES, stage=1, seed=s),
            "acc": lambda s: g.gemm_case(512, 768, 128, r=48, flags=ops.EPI_ACCUM, stage=1, seed=s),
            "seg": lambda s: g.gemm_case(600, 512, 128, r=16, seg=True, stage=1, seed=s),
            "big": lambda s: g.gemm_case(2048, 3072, 3072, r=16, stage=1, seed=s),
            "bigk": lambda s: g.gemm_case(1024, 3072, 12288, r=16, stage=1, seed=s),
            "wide": lambda s: g.gemm_case(4608, 3072, 1024, r=16, stage=1, seed=s),
            "persist_gelu": lambda s: g.gemm_case(8192, 4096, 256, r=16, flags=ops.EPI_GELU, stage=1, seed=s),
            "persist_gate": lambda s: g.gemm_case(8190, 4104, 192, r=16, flags=ops.EPI_GATE_RES, stage=1, seed=s),
            "persist_dgelu_acc": lambda s: g.gemm_case(6000, 5000, 128, r=32, flags=ops.EPI_DGELU | ops.EPI_ACCUM, stage=1, seed=s),
            "persist_seg": lambda s: g.gemm_case(9216, 3072, 128, r=16, seg=True, stage=1, seed=s),
            "persist_odd_steps": lambda s: g.gemm_case(9000, 3072, 192, stage=1, seed=s),
        }
        for rep in range(3):
            for name, fn in cases.items():
                try:
                    r_ = fn(rep)
                except Exception as e:  # noqa: BLE001
                    r_ = {"ok": False, "error": repr(e)}
                OUT["correctness"][f"s{stage}_{name}_{rep}"] = r_
                if not r_.get("ok"):
                    bad.append((name, rep, r_))
                    print("BAD", stage, name, rep, r_, flush=True)
    finally:
        ops.gemm_nt = orig
    return bad


def ab(M, N, K, rounds=5, iters=10, r=16):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    a2 = torch.randn(M, r, device=dev).to(torch.bfloat16)
    b2 = torch.randn(N, r, device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    variants = {"2bar_256": lambda: ops.gemm_nt(a, b, out, bias=bias, a2=a2, b2=b2, tile_mode=2, stage_mode=5),
                "8phase": lambda: ops.gemm_nt(a, b, out, bias=bias, a2=a2, b2=b2, tile_mode=2, stage_mode=4),
                "hipblaslt": lambda: torch.matmul(a, b.t(), out=out)}
    res = {k: [] for k in variants}
    for fn in variants.values():
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, fn in variants.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[k].append(2.0 * M * N * K / (e0.elapsed_time(e1) / iters) / 1e9)
    return {k: round(sorted(v)[len(v) // 2], 1) for k, v in res.items()}


if __name__ == "__main__":
    bad = run_cases(4)
    print("correctness bad:", bad, flush=True)
    for (M, N, K) in ((18432, 3072, 3072), (18432, 12288, 3072), (18432, 3072, 12288), (18432, 3072, 15360), (16384, 3072, 3072),
                      (4608, 3072, 3072), (26624, 1536, 1536), (26624, 8960, 1536), (26624, 1536, 8960)):
        OUT[f"ab_{M}x{N}x{K}"] = ab(M, N, K)
        print(M, N, K, OUT[f"ab_{M}x{N}x{K}"], flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(OUT, open("gpurun_out/gemm8.json", "w"), indent=1)
