#!/usr/bin/env python3
"""Throughput with the batch split over N extractor handles (each its own HIP stream) vs one handle."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from orb_slam2_amd import capi, synth
W, H = 1241, 376
stride = (W + 63) // 64 * 64
uniq = synth.frame_sequence(8, W, H)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
host = np.zeros((B, H, stride), np.uint8)
for i in range(B):
    host[i, :, :W] = uniq[i % 8]
frames = torch.from_numpy(host).cuda()
cap = 2100
kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
for nh in (1, 2, 4):
    exs = [capi.ORBextractor(2000, 1.2, 8, 20, 7) for _ in range(nh)]
    per = B // nh
    def step():
        for k, ex in enumerate(exs):
            f0 = k * per
            ex.extract_batch_device(frames[f0].data_ptr(), per, W, H, stride, stride * H, kps[f0].data_ptr(), desc[f0].data_ptr(),
                                    cnt[f0:].data_ptr(), cap, sync=False)
    for _ in range(5):
        step()
    for ex in exs:
        ex.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        step()
    for ex in exs:
        ex.synchronize()
    dt = time.perf_counter() - t0
    print(f"handles={nh} frames/s={B * n / dt:.0f} ms/step={dt / n * 1e3:.3f}")
    del exs
