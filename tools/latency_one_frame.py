#!/usr/bin/env python3
"""Single-frame latency of the extractor (TUM 640x480, 1000 features): device-resident frame through orbfe_extract_batch
(sync) and host image -> host result through orbfe_extract.  usage: [ORBFE_GRAPH=0] python tools/latency_one_frame.py"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from orb_slam2_amd import capi, synth
W, H, NF = 640, 480, 1000
img = synth.textured_image(W, H, synth.BASE_SEED + 1)
stride = (W + 63) // 64 * 64
host = np.zeros((1, H, stride), np.uint8); host[0, :, :W] = img
frames = torch.from_numpy(host).cuda()
cap = NF + 64
kps = torch.zeros((1, cap, 7), dtype=torch.float32, device="cuda")
desc = torch.zeros((1, cap, 32), dtype=torch.uint8, device="cuda")
cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
ex = capi.ORBextractor(NF, 1.2, 8, 20, 7)
def dev():
    ex.extract_batch_device(frames.data_ptr(), 1, W, H, stride, stride * H, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap, sync=True)
def med(fn, n=300):
    for _ in range(20): fn()
    t = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
    return float(np.median(t)) * 1e3
print({"device_resident_ms": round(med(dev), 4), "host_to_host_ms": round(med(lambda: ex(img)), 4), "keypoints": int(cnt[0])})
