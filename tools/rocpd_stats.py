"""Kernel statistics (rocprofv3 --stats equivalent) from a rocpd SQLite database written by `rocprofv3 --kernel-trace`:
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [out.csv] [skip_first_fraction]
skip_first_fraction drops the leading part of the timeline (model construction / warm-up) from the totals."""
import csv
import sqlite3
import sys


def main():
    db = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else None
    skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    scol = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else scol[-1])
    rows = cur.execute(f"select s.{name_col}, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id").fetchall()
    t0 = min(r[1] for r in rows)
    t1 = max(r[2] for r in rows)
    cut = t0 + skip * (t1 - t0)
    agg = {}
    for n, s, e in rows:
        if s < cut:
            continue
        a = agg.setdefault(n, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += e - s
        a[2] = min(a[2], e - s)
        a[3] = max(a[3], e - s)
    tot = sum(a[1] for a in agg.values())
    table = sorted(((n, a[0], a[1], a[1] / a[0], 100.0 * a[1] / tot, a[2], a[3]) for n, a in agg.items()), key=lambda r: -r[2])
    if out:
        with open(out, "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            w.writerows(table)
    for r in table[:25]:
        print(f"{r[4]:6.2f}%  {r[2] / 1e6:9.2f} ms  {r[1]:6d} x {r[3] / 1e3:9.1f} us  {r[0][:110]}")
    print(f"total kernel time {tot / 1e6:.1f} ms over {cols and len(rows)} dispatches (cols: {cols[:6]}...)")


if __name__ == "__main__":
    main()
