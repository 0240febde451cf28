"""Per-kernel means of rocprofv3 --pmc counter_collection CSVs:  python tools/pmc_summary.py out.json pass1.csv [pass2.csv ...]
Only kernels whose name contains one of KEEP are kept (the workload of tools/gpu_pmc_target.py)."""
import csv
import json
import sys

KEEP = ("gemm_nt_8phase_kernel", "attn_fwd_kernel", "attn_bwd_dkdv_kernel", "attn_bwd_dq_kernel")


def main():
    out, files = sys.argv[1], sys.argv[2:]
    acc = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            name = next((k for k in KEEP if k in r["Kernel_Name"]), None)
            if name is None:
                continue
            a = acc.setdefault(name, {}).setdefault(r["Counter_Name"], [0.0, 0])
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    res = {k: {c: v[0] / v[1] for c, v in d.items()} for k, d in acc.items()}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
