#!/usr/bin/env python3
"""Average counter values per kernel from rocprofv3 --pmc CSV output (one *_counter_collection.csv per pass).
usage: python tools/pmc_summary.py <dir-or-csv> [kernel-substring]"""
import collections
import csv
import sys
from pathlib import Path

root = Path(sys.argv[1])
want = sys.argv[2] if len(sys.argv) > 2 else ""
files = [root] if root.is_file() else sorted(root.rglob("*counter_collection.csv"))
for f in files:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if want in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        print(f"{k:42s} {c:32s} dispatches {len(v):4d} avg {sum(v) / len(v):16.1f}")
