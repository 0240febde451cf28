"""Stream-K tail of the persistent 8-phase GEMM (gemm8.hip, SK instantiations): correctness against the data-parallel kernel and the fp32
reference, run-to-run determinism (a race screen for the flag protocol), and alternating A/B timings on the step's small-batch shapes.

AITK_GEMM8_SK is re-read by the launcher on every call: 0 = data-parallel, 2 = stream-K tail whenever the contract allows."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ai_toolkit_amd  # noqa: E402,F401
from ai_toolkit_amd import ops  # noqa: E402

dev = "cuda"
OUT = {"correctness": {}, "timing": {}}


def sk(mode):
    os.environ["AITK_GEMM8_SK"] = str(mode)


def operands(M, N, K, r, flags, seed, emit=False):
    g = torch.Generator(device="cpu").manual_seed(seed)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    b = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    kw = {"bias": torch.randn(N, generator=g).to(torch.bfloat16).to(dev)}
    if r:
        kw["a2"] = torch.randn(M, r, generator=g).to(torch.bfloat16).to(dev)
        kw["b2"] = (torch.randn(N, r, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    c0 = None
    if flags & ops.EPI_ACCUM:
        c0 = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
    if flags & ops.EPI_GELU:
        kw["aux_out"] = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    if flags & ops.EPI_DGELU:
        kw["aux_in"] = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
    if flags & ops.EPI_GATE_RES:
        nb = 2 if M % 2 == 0 and (M // 2) % 16 == 0 and M // 2 >= 128 else 1
        kw.update(gate=torch.randn(nb, N, generator=g).to(torch.bfloat16).to(dev), gate_rows=M // nb,
                  aux_in=torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev), aux_out=torch.zeros(M, N, dtype=torch.bfloat16, device=dev))
    if emit:
        p = (torch.randn(16, N, generator=g) * 0.1)
        hi = p.to(torch.bfloat16)
        lo = (p - hi.float()).to(torch.bfloat16)
        kw["emit_t"] = (hi.to(dev), lo.to(dev), torch.zeros(N // 256, M, 16, dtype=torch.float32, device=dev), 0)
    return a, b, kw, c0


def run(a, b, kw, c0, flags):
    out = c0.clone() if c0 is not None else torch.full((a.shape[0], b.shape[0]), float("nan"), dtype=torch.bfloat16, device=dev)
    if "aux_out" in kw:
        kw["aux_out"].zero_()
    if "emit_t" in kw:
        kw["emit_t"][2].zero_()
    ops.gemm_nt(a, b, out, flags=flags, stage_mode=4, **kw)
    torch.cuda.synchronize()
    res = [out.clone()]
    if "aux_out" in kw:
        res.append(kw["aux_out"].clone())
    if "emit_t" in kw:
        res.append(kw["emit_t"][2].clone())
    return res


def rel(x, y):
    x, y = x.float(), y.float()
    return float((x - y).norm() / y.norm().clamp_min(1e-30))


def check(name, M, N, K, r=16, flags=0, emit=False, reps=6):
    a, b, kw, c0 = operands(M, N, K, r, flags, 0, emit)
    sk(0)
    base = run(a, b, kw, c0, flags)
    sk(2)
    first = run(a, b, kw, c0, flags)
    same_runs = True
    for _ in range(reps):
        again = run(a, b, kw, c0, flags)
        same_runs = same_runs and all(torch.equal(x, y) for x, y in zip(first, again))
    d = [rel(x, y) for x, y in zip(first, base)]
    differing = float((first[0] != base[0]).float().mean())
    nan = any(bool(torch.isnan(x.float()).any()) for x in first)
    ok = same_runs and not nan and all(e < 2e-3 for e in d)
    OUT["correctness"][name] = {"M": M, "N": N, "K": K, "r": r, "flags": flags, "emit": emit, "rel_vs_data_parallel": d, "frac_elements_differing": differing,
                                "deterministic": same_runs, "nan": nan, "ok": ok}
    print(("ok " if ok else "BAD"), name, OUT["correctness"][name], flush=True)
    return ok


def check_grouped(name, M1, M2, N, K, r=16, flags=0, reps=6):
    ops1 = operands(M1, N, K, r, flags, 1)
    ops2 = operands(M2, N, K, r, flags, 2)

    def go():
        outs = []
        lists = []
        for (a, b, kw, c0) in (ops1, ops2):
            out = c0.clone() if c0 is not None else torch.full((a.shape[0], N), float("nan"), dtype=torch.bfloat16, device=dev)
            if "aux_out" in kw:
                kw["aux_out"].zero_()
            with ops.recording() as rec:
                ops.gemm_nt(a, b, out, flags=flags, stage_mode=4, **kw)
            lists.append(rec)
            outs.append(out)
        ops.replay_paired(lists[0], lists[1])
        torch.cuda.synchronize()
        return [o.clone() for o in outs] + [kw["aux_out"].clone() for (_, _, kw, _) in (ops1, ops2) if "aux_out" in kw]

    sk(0)
    base = go()
    sk(2)
    first = go()
    same = True
    for _ in range(reps):
        again = go()
        same = same and all(torch.equal(x, y) for x, y in zip(first, again))
    d = [rel(x, y) for x, y in zip(first, base)]
    nan = any(bool(torch.isnan(x.float()).any()) for x in first)
    ok = same and not nan and all(e < 2e-3 for e in d)
    OUT["correctness"][name] = {"M": (M1, M2), "N": N, "K": K, "flags": flags, "rel_vs_data_parallel": d, "deterministic": same, "nan": nan, "ok": ok}
    print(("ok " if ok else "BAD"), name, OUT["correctness"][name], flush=True)
    return ok


def fp32_check(name, M, N, K, r=16):
    a, b, kw, _ = operands(M, N, K, r, 0, 3)
    ref = a.float() @ b.float().t() + kw["bias"].float() + kw["a2"].float() @ kw["b2"].float().t()
    sk(2)
    out = run(a, b, kw, None, 0)[0]
    sk(0)
    out0 = run(a, b, kw, None, 0)[0]
    OUT["correctness"][name] = {"rel_sk_vs_fp32": rel(out, ref), "rel_dp_vs_fp32": rel(out0, ref)}
    print(name, OUT["correctness"][name], flush=True)
    return OUT["correctness"][name]["rel_sk_vs_fp32"] < 4e-3


def ab(M, N, K, r=16, flags=0, rounds=5, iters=20):
    a, b, kw, c0 = operands(M, N, K, r, flags, 5)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    res = {0: [], 1: [], 2: []}
    for mode in (0, 2):
        sk(mode)
        for _ in range(3):
            ops.gemm_nt(a, b, out, flags=flags, stage_mode=4, **kw)
    torch.cuda.synchronize()
    for _ in range(rounds):
        for mode in (0, 1, 2):
            sk(mode)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                ops.gemm_nt(a, b, out, flags=flags, stage_mode=4, **kw)
            e1.record()
            torch.cuda.synchronize()
            res[mode].append(e0.elapsed_time(e1) / iters * 1e3)
    med = {m: sorted(v)[len(v) // 2] for m, v in res.items()}
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    row = {"tiles": tiles, "rounds": round(tiles / 256, 3), "us_data_parallel": round(med[0], 1), "us_auto": round(med[1], 1), "us_stream_k": round(med[2], 1),
           "stream_k_over_dp": round(med[2] / med[0], 4), "tflops_dp": round(2.0 * M * N * (K + r) / med[0] / 1e6, 1), "tflops_sk": round(2.0 * M * N * (K + r) / med[2] / 1e6, 1)}
    OUT["timing"][f"{M}x{N}x{K}_f{flags}"] = row
    print(M, N, K, flags, row, flush=True)


if __name__ == "__main__":
    ok = True
    G, GA, DG, AC = ops.EPI_GELU, ops.EPI_GATE_RES, ops.EPI_DGELU, ops.EPI_ACCUM
    ok &= fp32_check("fp32_4608x3072x3072", 4608, 3072, 3072)
    ok &= check("b1_n3072", 4608, 3072, 3072)
    ok &= check("b1_n9216", 4608, 9216, 3072)
    ok &= check("b1_gelu", 4608, 12288, 3072, flags=G)
    ok &= check("b1_gelu_emit", 4608, 12288, 3072, flags=G, emit=True)
    ok &= check("b1_k12288_gate", 4608, 3072, 12288, flags=GA)
    ok &= check("b1_k15360", 4608, 3072, 15360)
    ok &= check("b1_dgelu", 4608, 12288, 3072, flags=DG, r=0)
    ok &= check("b2_acc_r48", 9216, 3072, 3072, flags=AC, r=48)
    ok &= check("ragged_rows", 4500, 3072, 3072)
    ok &= check("k_tail", 4608, 3072, 3088, r=16)
    ok &= check("few_tiles", 1024, 3072, 3072)
    ok &= check("short_k", 4608, 3072, 1024)
    ok &= check("wan_b1", 13312, 1536, 1536)
    ok &= check_grouped("grouped_b1", 4096, 512, 3072, 3072)
    ok &= check_grouped("grouped_b1_gelu", 4096, 512, 12288, 3072, flags=G)
    ok &= check_grouped("grouped_b2_gate", 8192, 1024, 3072, 12288, flags=GA)
    print("correctness all ok:", ok, flush=True)
    OUT["all_ok"] = bool(ok)
    for B in (1, 2, 3, 4, 7):
        M = 4608 * B
        for (N, K, fl) in ((3072, 3072, 0), (9216, 3072, 0), (12288, 3072, G), (3072, 12288, GA), (3072, 15360, 0), (3072, 9216, 0)):
            ab(M, N, K, flags=fl)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(OUT, open("gpurun_out/gemm8_sk.json", "w"), indent=1)
    sys.exit(0 if ok else 1)
