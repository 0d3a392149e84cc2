#!/usr/bin/env python3
"""Time orbfe_match_knn2 (2000 x 1M) with HIP events on the matcher stream via torch events is not possible (own stream),
so use wall time around synchronous calls.  usage: [ORBFE_LIB=...] python tools/knn_time.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from orb_slam2_amd import capi, synth
NQ, NDB = 2000, 1_000_000
db = torch.from_numpy(synth.random_descriptors(NDB, 7000)).cuda()
q = torch.from_numpy(synth.random_descriptors(NQ, 42)).cuda()
res = torch.zeros((3, NQ), dtype=torch.int32, device="cuda")
m = capi.ORBmatcher()
for _ in range(3):
    m.knn2_device(q.data_ptr(), NQ, db.data_ptr(), NDB, res[0].data_ptr(), res[1].data_ptr(), res[2].data_ptr())
t0 = time.perf_counter()
n = 20
for _ in range(n):
    m.knn2_device(q.data_ptr(), NQ, db.data_ptr(), NDB, res[0].data_ptr(), res[1].data_ptr(), res[2].data_ptr(), sync=False)
m.synchronize()
dt = (time.perf_counter() - t0) / n
print(str(capi.LIB_PATH).split("/")[-1], "ms", round(dt * 1e3, 3), "Tpairs/s", round(NQ * NDB / dt / 1e12, 3))
