"""From a rocprofv3 --kernel-trace CSV of a few steps: for every launch of the HBM-bound kernels, its duration grouped by (kernel, grid size,
the kernel that ran just before it) — which producer -> consumer pairs run cold?"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
want = ("lora_down16_kernel", "lora_wgrad_kernel", "ln_mod_fwd", "ln_mod_bwd", "qkv_post", "gate_bwd", "quant_rows")


def short(n):
    n = n.split("(")[0]
    for p in ("void ", ):
        n = n.replace(p, "")
    return n[:34]


acc = collections.defaultdict(list)
prev = None
for r in rows:
    name = r["Kernel_Name"]
    if any(w in name for w in want):
        gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
        wx = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        acc[(short(name), gx // max(wx, 1), short(prev) if prev else "-")].append(dur)
    prev = name
print(f"{'kernel':36s} {'WGs':>7s} {'previous kernel':36s} {'n':>5s} {'mean us':>9s} {'min':>8s} {'total ms':>9s}")
for (k, g, p), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) < 2000:
        continue
    print(f"{k:36s} {g:7d} {p:36s} {len(v):5d} {sum(v) / len(v):9.1f} {min(v):8.1f} {sum(v) / 1e3:9.2f}")
