#!/usr/bin/env python3
"""Throughput of orbfe_extract_batch against the frames one launch carries (KITTI-size frames, 2000 features, frames resident
in HBM, launches queued back to back on one handle): the curve bench.py's launch split is chosen from.
usage: python tools/launch_curve.py [out.json]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from orb_slam2_amd import capi, synth  # noqa: E402

W, H = 1241, 376
stride = (W + 63) // 64 * 64
NMAX = 1280
uniq = synth.frame_sequence(16, W, H)
ud = torch.zeros((16, H, stride), dtype=torch.uint8)
ud[:, :, :W] = torch.from_numpy(np.ascontiguousarray(uniq))
ud = ud.cuda()
frames = ud[torch.arange(NMAX, device="cuda") % 16].contiguous()
cap = 2100
kps = torch.zeros((NMAX, cap, 7), dtype=torch.float32, device="cuda")
desc = torch.zeros((NMAX, cap, 32), dtype=torch.uint8, device="cuda")
cnt = torch.zeros(NMAX, dtype=torch.int32, device="cuda")
ex = capi.ORBextractor(2000, 1.2, 8, 20, 7)
rows = []
for n in (16, 32, 64, 96, 128, 192, 256, 284, 320, 379, 384, 448, 505, 512, 568, 640, 757, 768, 1024, 1136, 1280):
    def run(launches):
        for _ in range(launches):
            ex.extract_batch_device(frames.data_ptr(), n, W, H, stride, stride * H, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap,
                                    sync=False)
        ex.synchronize()
    launches = max(4, 4096 // n)
    run(2)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        run(launches)
        best = min(best, time.perf_counter() - t0)
    fps = n * launches / best
    rows.append({"frames_per_launch": n, "launches": launches, "frames_per_s": round(fps, 1), "us_per_frame": round(1e6 / fps, 3)})
    print(rows[-1], flush=True)
if len(sys.argv) > 1:
    Path(sys.argv[1]).write_text(json.dumps({"workload": "1241x376, 2000 features, 8 levels; one handle, launches queued back to back",
                                             "curve": rows}, indent=1) + "\n")
