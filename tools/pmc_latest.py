#!/usr/bin/env python3
"""profiles/latest_pmc.json from the rocprofv3 --pmc passes of tools/pmc_passes.sh: per kernel the average counter values
per launch, the frames one launch processed, and a hash of the kernel sources the passes ran (bench.py compares it with the
tree it runs from and says "stale" when they differ).

usage (on the GPU box, after tools/pmc_passes.sh <dir>): python tools/pmc_latest.py <dir> <frames per launch> [out.json]
"""
import collections
import csv
import hashlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
KERNELS = ("fast_cells_kernel", "gauss7_kernel", "orient_describe_kernel", "pyr_resize_kernel", "quadtree_kernel")


def csrc_hash() -> str:
    """sha256 over the kernel and host sources of liborbfe.so, in name order"""
    h = hashlib.sha256()
    src = ROOT / "orb_slam2_amd" / "csrc"
    for f in sorted(p for p in src.iterdir() if p.suffix in (".h", ".hip", ".inc")):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()


def main():
    root, frames = Path(sys.argv[1]), int(sys.argv[2])
    out = Path(sys.argv[3]) if len(sys.argv) > 3 else ROOT / "profiles" / "latest_pmc.json"
    agg = collections.defaultdict(list)
    for f in sorted(root.rglob("*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            for k in KERNELS:
                if k in r["Kernel_Name"]:
                    agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    kernels = {}
    for (k, c), v in sorted(agg.items()):
        kernels.setdefault(k, {"launches_profiled": len(v)})[c] = round(sum(v) / len(v), 1)
    # the Hamming kNN-2 call (tools/knn_time.py: 2000 queries x 1 M references per call): counters summed over ALL kernels of
    # a call (sample pre-pass, main scan, folds) and divided by the calls = dispatches of the main scan knn2_scan_kernel<true>
    knn_sum, knn_calls, knn_kernels = collections.defaultdict(float), collections.defaultdict(int), collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(root.rglob("*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "knn2_" not in name:
                continue
            c = r["Counter_Name"]
            knn_sum[c] += float(r["Counter_Value"])
            short = name.split("(")[0].replace("void ", "").replace("orbfe::", "")
            knn_kernels[short][c].append(float(r["Counter_Value"]))
            if "knn2_scan_kernel<true>" in name:
                knn_calls[c] += 1
    knn = None
    if knn_calls:
        knn = {"queries": 2000, "references": 1000000, "pairs_per_call": 2000 * 1000000, "calls_profiled": max(knn_calls.values()),
               "per_call": {c: round(v / knn_calls[c], 1) for c, v in sorted(knn_sum.items()) if knn_calls.get(c)},
               "per_launch_by_kernel": {k: {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())} for k, cs in sorted(knn_kernels.items())}}
    doc = {"csrc_sha256": csrc_hash(), "frames_per_launch": frames, "workload": "tools/stage_times.py: KITTI-size frames, 2000 features",
           "units": {"FETCH_SIZE": "KB, halve-corrected by the reader: gfx950 tallies 128-B requests at 64 B (MI355X_MICROARCH.md)",
                     "WRITE_SIZE": "KB", "SQ_*": "summed over the 8 XCDs", "GRBM_GUI_ACTIVE": "cycles x 8 XCDs"},
           "kernels": kernels}
    if knn:
        doc["knn2"] = knn
    out.write_text(json.dumps(doc, indent=1) + "\n")
    print(f"wrote {out}: {len(kernels)} kernels, csrc {doc['csrc_sha256'][:12]}")


if __name__ == "__main__":
    main()
