#!/usr/bin/env python3
"""Per-stage device times of the batched extractor (HIP events inside the library), for A/B work on kernels.
usage: [ORBFE_LIB=path/to/variant.so] python tools/stage_times.py [batch] [steps]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from orb_slam2_amd import capi, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
W, H = 1241, 376
stride = (W + 63) // 64 * 64
uniq = synth.frame_sequence(8, W, H)
host = np.zeros((B, H, stride), np.uint8)
for i in range(B):
    host[i, :, :W] = uniq[i % 8]
frames = torch.from_numpy(host).cuda()
cap = 2100
kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
ex = capi.ORBextractor(2000, 1.2, 8, 20, 7)
for _ in range(3):
    ex.extract_batch_device(frames.data_ptr(), B, W, H, stride, stride * H, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap)
ex.set_profiling(True)
for _ in range(steps):
    ex.extract_batch_device(frames.data_ptr(), B, W, H, stride, stride * H, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap)
p = ex.profile()
import zlib  # noqa: E402
n = cnt.cpu().numpy()
crc = 0
for f in range(min(B, 8)):      # output checksum of the distinct frames: A/B variants must print the same value
    crc = zlib.crc32(desc[f, :n[f]].cpu().numpy().tobytes(), zlib.crc32(kps[f, :n[f]].cpu().numpy().tobytes(), crc))
print("crc %08x" % crc, "variant", ex.fast_variant(), end=" ")
print(str(capi.LIB_PATH).split("/")[-1], {k: round(v / steps, 4) for k, v in p["stage_ms"].items()}, "sum",
      round(sum(p["stage_ms"].values()) / steps, 4), "kps/frame", p["keypoints"] / p["frames"])
