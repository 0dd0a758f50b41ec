#!/usr/bin/env python3
"""Per-kernel summary (calls, average / min / max duration) of a rocprofv3 kernel trace stored as
a rocpd sqlite database (rocprofv3's default output on ROCm 7.2):

    python profiles/db_kernel_stats.py gpurun_out/prof/x_results.db [out.csv]
"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {name}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                      f"from kernels group by {name} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = [("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")]
    for n, c, t, a, lo, hi in rows:
        out.append((n, c, t, round(a, 1), round(100.0 * t / total, 2), lo, hi))
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w", newline="") as f:
            csv.writer(f).writerows(out)
    for r in out[:40]:
        print(f"{str(r[0])[:90]:90s} {r[1]:>7} {r[3]:>12} {r[4]:>6}")


if __name__ == "__main__":
    main()
