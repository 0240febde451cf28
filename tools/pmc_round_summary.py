"""profiles/rNN_pmc/summary.json from the counter_collection CSVs of separate rocprofv3 --pmc passes over tools/gpu_pmc_target.py:
    python tools/pmc_round_summary.py <dir with *_counter_collection.csv> <out.json> M K2
Per kernel: mean of every counter per dispatch.  For the dominant GEMM shape M x 3072 x 3072 (+K2 slab): memory-side bytes per launch =
FETCH_SIZE*1024*2 (gfx950 half-count correction for 16-B/lane reads, MI355X_MICROARCH.md §HBM) + WRITE_SIZE*1024 (uncalibrated),
next to the algorithmic bytes 2*(M*(K+K2) + N*(K+K2) + M*N)."""
import csv
import glob
import json
import os
import sys

KEEP = ("gemm_nt_8phase_f8_kernel", "gemm_nt_8phase_kernel", "quant_rows_fp8_kernel", "attn_fwd_kernel", "attn_bwd_dkdv_ws_kernel", "attn_bwd_dkdv_pipe_kernel",
        "attn_bwd_dkdv_kernel", "attn_bwd_dq_ds8_kernel", "attn_bwd_dq_ds_kernel", "attn_bwd_dq_kernel")


def main():
    d, out, M, K2 = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    acc = {}
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            name = next((k for k in KEEP if k in r["Kernel_Name"]), None)
            if name is None:
                continue
            a = acc.setdefault(name, {}).setdefault(r["Counter_Name"], [0.0, 0])
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    means = {k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in acc.items()}
    res = {"workload": f"tools/gpu_pmc_target.py: 4 launches each; GEMM {M}x3072x3072 + LoRA K-slab {K2} + bias (bf16), attention B=1 H=24 S=4608",
           "correction": "bytes = FETCH_SIZE*1024*2 (gfx950: 64 B counted per 128-B request for 16-B/lane reads) + WRITE_SIZE*1024 (uncalibrated)",
           "raw_means": means}
    g = means.get("gemm_nt_8phase_kernel", {})
    if "FETCH_SIZE" in g and "WRITE_SIZE" in g:
        N = K = 3072
        res["gemm_nt_8phase_kernel_by_M"] = {str(M): {
            "fetch_bytes_per_launch": g["FETCH_SIZE"] * 1024 * 2, "write_bytes_per_launch": g["WRITE_SIZE"] * 1024,
            "hbm_bytes_per_launch": g["FETCH_SIZE"] * 1024 * 2 + g["WRITE_SIZE"] * 1024,
            "algorithmic_bytes": 2 * (M * (K + K2) + N * (K + K2) + M * N), "source": f"{d} (separate --pmc FETCH_SIZE / WRITE_SIZE passes)"}}
    f8 = means.get("gemm_nt_8phase_f8_kernel", {})
    if "FETCH_SIZE" in f8 and "WRITE_SIZE" in f8:
        N = K = 3072
        res["gemm_nt_8phase_f8_kernel_by_M"] = {str(M): {
            "fetch_bytes_per_launch": f8["FETCH_SIZE"] * 1024 * 2, "write_bytes_per_launch": f8["WRITE_SIZE"] * 1024,
            "hbm_bytes_per_launch": f8["FETCH_SIZE"] * 1024 * 2 + f8["WRITE_SIZE"] * 1024,
            "algorithmic_bytes": M * K + N * K + 2 * (M * K2 + N * K2 + M * N) + 4 * (M + N), "source": f"{d} (separate --pmc FETCH_SIZE / WRITE_SIZE passes)"}}
    for k, c in means.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"] > 0:
            res.setdefault("mfma_busy", {})[k] = {"SQ_VALU_MFMA_BUSY_CYCLES": c["SQ_VALU_MFMA_BUSY_CYCLES"], "SQ_BUSY_CYCLES": c["SQ_BUSY_CYCLES"],
                                                  "ratio": c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"]}
    for k, c in means.items():
        if "SQ_WAVE_CYCLES" in c:
            wc = c["SQ_WAVE_CYCLES"]
            res.setdefault("derived", {})[k] = {x: c[y] / wc for x, y in (("wait_any_frac", "SQ_WAIT_ANY"), ("wait_inst_any_frac", "SQ_WAIT_INST_ANY"),
                                                                         ("active_inst_any_frac", "SQ_ACTIVE_INST_ANY")) if y in c}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "raw_means"}, indent=1))


if __name__ == "__main__":
    main()
