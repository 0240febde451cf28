"""Kernel-stats CSV (Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs) from a rocprofv3 rocpd database
(`rocprofv3 --kernel-trace --stats` writes <name>_results.db when no csv output format is requested):
    python tools/rocpd_kernel_stats.py gpurun_out/r2e_prof/r2e_results.db profiles/r02_rocprof_kernel_stats_e_b7.csv"""
import csv
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = c.execute(f"select {name}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels "
                     f"group by {name} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, calls, total, avg, mn, mx in rows:
            w.writerow([n, calls, total, f"{avg:.1f}", f"{100.0 * total / tot:.4f}", mn, mx])
    print(f"{len(rows)} kernels, {tot / 1e6:.1f} ms of GPU time -> {out}")


if __name__ == "__main__":
    main()
