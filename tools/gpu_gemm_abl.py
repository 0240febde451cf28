"""Ablation of the 8-phase GEMM loop: one subprocess per debug library (ai-toolkit_amd/libaitk_abl_*.so)."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch

    import ai_toolkit_amd  # noqa: F401
    from ai_toolkit_amd import ops

    out = {}
    for (M, N, K) in ((18432, 3072, 3072), (18432, 12288, 3072), (18432, 3072, 12288)):
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        b = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
        a2 = torch.randn(M, 16, device="cuda").to(torch.bfloat16)
        b2 = torch.randn(N, 16, device="cuda").to(torch.bfloat16)
        c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        for _ in range(3):
            ops.gemm_nt(a, b, c, a2=a2, b2=b2, tile_mode=2, stage_mode=4)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gemm_nt(a, b, c, a2=a2, b2=b2, tile_mode=2, stage_mode=4)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1000)
        out[f"{M}x{N}x{K}"] = round(sorted(ts)[2], 1)
    print("RESULT", json.dumps(out))
else:
    res = {}
    for lib in sorted(glob.glob(os.path.join(ROOT, "ai-toolkit_amd", "libaitk_abl_*.so"))):
        env = dict(os.environ, AITK_LIB_PATH=lib)
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True, timeout=200)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        res[os.path.basename(lib)] = json.loads(line[0][7:]) if line else r.stderr[-300:]
        print(os.path.basename(lib), res[os.path.basename(lib)], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "gemm_abl.json"), "w"), indent=1)
