#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 --kernel-trace run, grouped by kernel name AND launch grid, so the launches of
bench.py's timed region (568 frames each) stand apart from its side measurements.  Input: the rocpd database
(*_results.db, rocprofv3's default output) or the CSV trace (*_kernel_trace.csv, --output-format csv).

usage: python tools/kernel_stats.py gpurun_out/<run>/<name>_results.db | <name>_kernel_trace.csv > profiles/<round>_bench_kernel_stats.csv
"""
import csv
import math
import sqlite3
import sys


def main(path):
    if path.endswith(".csv"):
        with open(path, newline="") as fh:
            rows = [(r["Kernel_Name"], int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                    for r in csv.DictReader(fh)]
    else:
        db = sqlite3.connect(path)
        rows = db.execute(
            "SELECT S.display_name, K.grid_size_x, K.grid_size_y, K.end - K.start "
            "FROM rocpd_kernel_dispatch K JOIN rocpd_info_kernel_symbol S ON S.id = K.kernel_id AND S.guid = K.guid").fetchall()
    groups = {}
    for name, gx, gy, ns in rows:
        groups.setdefault((name, gx, gy), []).append(ns)
    total = sum(ns for *_k, ns in rows) or 1
    w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "GridX", "GridY", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for (name, gx, gy), d in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        n, s = len(d), sum(d)
        avg = s / n
        sd = math.sqrt(sum((x - avg) ** 2 for x in d) / n)
        w.writerow([name, gx, gy, n, s, round(avg, 3), round(100.0 * s / total, 2), min(d), max(d), round(sd, 3)])


if __name__ == "__main__":
    main(sys.argv[1])
