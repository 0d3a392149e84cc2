#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native ORB front end.

Metric (BASELINE.json): ORB extract frames/sec @1241x376, 2000 keypoints, 8 levels; plus Hamming matches/sec.

A "step" is one pass of ORBextractor::operator() over BASELINE.json configs[3]: the 4541 synthetic KITTI-sized frames of a
KITTI-00-length sequence, already resident in HBM, split over the ranks in contiguous blocks of ceil(4541 / G) frames
 and run through orbfe_extract_batch (the C-ABI) in launches of --batch frames; keypoints and
descriptors are left in HBM.  Independent frames: no data-path collective; the total work is fixed, so the scaling is
"strong"; `value` = 4541 x steps / max-over-ranks time.  (--total-frames 0 gives the round-1 weak-scaling step: --batch
frames per rank and step.)

Extra objects in the JSON line:
  roofline     -- the dominant kernel (FAST): algorithmic bytes per launch / its average duration, measured live with HIP
                  events recorded on the library's own stream around that kernel inside the timed region.
  cpu_baseline -- the CPU oracle (a port of the reference's path: the reference itself needs OpenCV and cannot be built
                  here) timed on the host cores of this box on a bounded sample of the same frames (rank 0, N=1 only).
  hamming      -- brute-force kNN-2 of 2000 queries against a 1M-descriptor database sharded over the ranks
                  (BASELINE.json config 5): orbfe_knn2_sharded_run -- RCCL broadcast / all-gather and the fold inside liborbfe.

Launch: python bench.py [--gpus N --steps K --warmup W]; for N>1 the driver uses torch.distributed.run.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

W, H, NFEATURES, NLEVELS, SCALE, INI_TH, MIN_TH = 1241, 376, 2000, 8, 1.2, 20, 7
LEVEL_PIXELS = 1444097                      # sum of the 8 level areas at 1241x376
HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
N_UNIQUE = 16                               # distinct synthetic frames, tiled to the batch size


def host_cores() -> dict:
    """The host cores this process may use: the affinity mask, cut by the cgroup CPU quota where one is set (a 1-GPU box shows
    the whole machine's CPUs in its mask but owns a 16-CPU share of them).  `threads` is what the CPU baselines run on."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if a != "max":
            quota = float(a) / float(b)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    threads = aff if quota is None else max(1, min(aff, int(round(quota))))
    env = os.environ.get("OMP_NUM_THREADS", "")
    if quota is None and env.isdigit() and 0 < int(env) < threads:
        threads = int(env)      # no quota visible: the box's own sizing hint (set to its CPU share)
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"threads": threads, "affinity_cpus": aff, "cgroup_cpu_quota": quota, "cpu_model": model}


def cpu_side(W, H, nfeat, seed_first, budget_s=2.5) -> dict:
    """The CPU oracle on one of the other configurations: per-frame time on one core and the rate on every core of the box."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib as O
    from orb_slam2_amd import synth
    hc = host_cores()
    frames = np.ascontiguousarray(synth.frame_sequence(8, W, H, first=seed_first))
    ex = O.OracleExtractor(nfeat, SCALE, NLEVELS, INI_TH, MIN_TH)
    per = []
    for rep in range(2):
        for f in frames:
            t0 = time.perf_counter()
            ex.extract(np.ascontiguousarray(f))
            per.append(time.perf_counter() - t0)
    per = np.array(per[len(frames):])
    t1, _ = O.time_extract(frames, nfeat, SCALE, NLEVELS, INI_TH, MIN_TH, hc["threads"], reps=1)
    reps = int(max(1, min(200, budget_s / max(t1, 1e-3))))
    t, _ = O.time_extract(frames, nfeat, SCALE, NLEVELS, INI_TH, MIN_TH, hc["threads"], reps=reps)
    return {"kind": "port", "per_frame_ms_one_core": {"median": round(float(np.median(per)) * 1e3, 3), "mean": round(float(per.mean()) * 1e3, 3)},
            "frames_per_s_all_cores": round(len(frames) * reps / t, 1), "cores": hc["threads"],
            "sample": f"{len(frames) * reps} extractions of 8 distinct {W}x{H} frames in {t:.1f} s"}


def cpu_baseline(frames: np.ndarray, budget_s: float = 12.0) -> dict:
    """Time the CPU oracle (tests/oracle_lib.py -> oracle/liborb_oracle.so) on the box's host cores."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib as O

    hc = host_cores()
    threads = hc["threads"]                             # every core this box gives the process (count and model in the result)
    t, _ = O.time_extract(frames, NFEATURES, SCALE, NLEVELS, INI_TH, MIN_TH, threads, reps=1)
    per_pass = max(t, 1e-3)
    reps = int(max(1, min(1000, budget_s / per_pass)))
    t, kps = O.time_extract(frames, NFEATURES, SCALE, NLEVELS, INI_TH, MIN_TH, threads, reps=reps)
    n = len(frames) * reps
    # per-frame wall clock on one core, the way the reference's drivers time a frame (steady_clock around the call, then
    # median and mean over the sequence: Examples/Stereo/stereo_kitti.cc:80-97, :113-122)
    ex = O.OracleExtractor(NFEATURES, SCALE, NLEVELS, INI_TH, MIN_TH)
    per_frame = []
    for rep in range(2):
        for f in frames:
            t0 = time.perf_counter()
            ex.extract(np.ascontiguousarray(f))
            per_frame.append(time.perf_counter() - t0)
    per_frame = np.array(per_frame[len(frames):])       # second pass: buffers warm
    return {"value": round(n / t, 2), "unit": "frames/s", "cores": threads, "kind": "port", "cpu_model": hc["cpu_model"],
            "affinity_cpus": hc["affinity_cpus"], "cgroup_cpu_quota": hc["cgroup_cpu_quota"],
            "per_frame_ms_one_core": {"median": round(float(np.median(per_frame)) * 1e3, 3), "mean": round(float(per_frame.mean()) * 1e3, 3),
                                      "frames": len(per_frame)},
            "sample": f"{n} extractions ({len(frames)} distinct 1241x376 frames x {reps}) in {t:.1f} s, one frame per thread, "
                      f"oracle built -O3 -march=x86-64-v3 -ffp-contract=off; {kps / n:.0f} keypoints/frame"}


def plan_rank(total_frames: int, world: int, rank: int, batch: int):
    """(first frame, frames per step, frames per launch) of one rank.  total_frames > 0 (strong scaling): contiguous blocks of
    ceil(total / G) frames, the last ranks may get fewer or none; 0 (weak): `batch` frames per rank.  The block is cut into
    EQUAL orbfe_extract_batch launches of at most `batch` frames (4541 -> 8 x 568, 568 -> 1 x 568), never a full launch plus
    a small ragged one (512 + 56 ran the tail at the 64-frame rate)."""
    if total_frames > 0:
        block = -(-total_frames // world)
        f_first = min(rank * block, total_frames)
        B = max(min(block, total_frames - f_first), 0)
    else:
        f_first, B = rank * batch, batch
    n_launches = max(1, -(-max(B, 1) // max(batch, 1)))
    launch = max(1, -(-max(B, 1) // n_launches))
    return f_first, B, launch


def csrc_hash() -> str:
    """sha256 over the sources of liborbfe.so in name order (the same rule as tools/pmc_latest.py)"""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(p for p in (ROOT / "orb_slam2_amd" / "csrc").iterdir() if p.suffix in (".h", ".hip", ".inc")):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()


def latest_pmc():
    """profiles/latest_pmc.json (tools/pmc_passes.sh + tools/pmc_latest.py): counter values per launch of each kernel and the
    hash of the sources they were taken from; stale = the tree has other kernel sources than the profiled ones."""
    f = ROOT / "profiles" / "latest_pmc.json"
    if not f.exists():
        return None, True
    doc = json.loads(f.read_text())
    return doc, doc.get("csrc_sha256") != csrc_hash()


def other_configs(local_rank, dev, with_cpu=True):
    """Side measurements for BASELINE.json configs[1] (TUM 640x480, per-frame latency), configs[2] (KITTI stereo pair +
    row-band search) and the EuRoC resolution.  Rank 0, one GPU; a few hundred milliseconds each; not part of `value`."""
    import torch
    from orb_slam2_amd import capi, synth
    out = {}

    def batch_fps(W, H, nfeat, B, steps=6):
        stride = (W + 63) // 64 * 64
        uniq = synth.frame_sequence(8, W, H, first=900)
        host = np.zeros((B, H, stride), np.uint8)
        for i in range(B):
            host[i, :, :W] = uniq[i % 8]
        frames = torch.from_numpy(host).to(dev)
        cap = nfeat + 64
        kps = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
        desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
        cnt = torch.zeros(B, dtype=torch.int32, device=dev)
        ex = capi.ORBextractor(nfeat, SCALE, NLEVELS, INI_TH, MIN_TH, device=local_rank)
        call = lambda n, sync: ex.extract_batch_device(frames.data_ptr(), n, W, H, stride, stride * H, kps.data_ptr(),
                                                       desc.data_ptr(), cnt.data_ptr(), cap, sync=sync)
        for _ in range(2):
            call(B, True)
        t0 = time.perf_counter()
        for _ in range(steps):
            call(B, False)
        ex.synchronize()
        fps = B * steps / (time.perf_counter() - t0)
        lat = []
        for _ in range(30):          # one frame already in HBM, synchronous call
            t0 = time.perf_counter()
            call(1, True)
            lat.append(time.perf_counter() - t0)
        host_lat = []
        img = np.ascontiguousarray(uniq[0])
        ex(img)
        for _ in range(30):          # host image in, host keypoints + descriptors out (PCIe both ways)
            t0 = time.perf_counter()
            ex(img)
            host_lat.append(time.perf_counter() - t0)
        return {"frames_per_s_batched": round(fps, 1), "batch": B, "features": nfeat,
                "latency_ms_device_resident_frame": round(float(np.median(lat)) * 1e3, 4),
                "latency_ms_host_image_to_host_result": round(float(np.median(host_lat)) * 1e3, 4),
                "keypoints_per_frame": round(float(cnt.float().mean().item()), 1)}

    out["tum_640x480_1000kp"] = batch_fps(640, 480, 1000, 256)
    out["euroc_752x480_1200kp"] = batch_fps(752, 480, 1200, 256)       # Examples/Stereo/EuRoC.yaml:88
    if with_cpu:
        out["tum_640x480_1000kp"]["cpu_oracle"] = cpu_side(640, 480, 1000, 900)
        out["euroc_752x480_1200kp"]["cpu_oracle"] = cpu_side(752, 480, 1200, 900)
    # BASELINE.json configs[1] through the C++ drop-in (include/ORBextractor.h + adapters/ORBextractor_orbfe.cc): wall clock
    # around ORBextractor::operator() on a host image, with and without the mvImagePyramid mirror
    exe = ROOT / "tests" / "cpp" / "adapter_main"
    if exe.exists():
        import subprocess
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            raw = Path(td) / "tum.raw"
            raw.write_bytes(np.ascontiguousarray(synth.frame_sequence(1, 640, 480, first=900)[0]).tobytes())
            r = subprocess.run([str(exe), "latency", str(raw), "640", "480", "1000", "200"], capture_output=True, text=True, timeout=120)
        try:
            out["tum_640x480_1000kp"]["cpp_adapter_operator_call"] = json.loads(r.stdout.strip().splitlines()[-1])
        except (ValueError, IndexError):
            out["tum_640x480_1000kp"]["cpp_adapter_operator_call"] = {"error": (r.stdout + r.stderr)[-300:]}

    # KITTI stereo pair: left + right extraction as one 2-frame batch, then the row-band Hamming search on the device
    W, H, nfeat = 1241, 376, NFEATURES
    stride = (W + 63) // 64 * 64
    left, right = synth.stereo_pair(W, H, synth.BASE_SEED + 50)
    host = np.zeros((2, H, stride), np.uint8)
    host[0, :, :W], host[1, :, :W] = left, right
    frames = torch.from_numpy(host).to(dev)
    cap = nfeat + 64
    kps = torch.zeros((2, cap, 7), dtype=torch.float32, device=dev)
    desc = torch.zeros((2, cap, 32), dtype=torch.uint8, device=dev)
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    ur, dep = (torch.zeros(cap, dtype=torch.float32, device=dev) for _ in range(2))
    ex = capi.ORBextractor(nfeat, SCALE, NLEVELS, INI_TH, MIN_TH, device=local_rank)
    m = capi.ORBmatcher(device=local_rank)
    bf, b = 386.1448, 0.5371657                                       # KITTI00-02.yaml: Camera.bf, bf / fx
    lat = []
    for it in range(33):
        t0 = time.perf_counter()
        ex.extract_batch_device(frames.data_ptr(), 2, W, H, stride, stride * H, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(),
                                cap, sync=True)
        n = cnt.cpu().numpy()        # the matcher needs the counts on the host, as Frame::ComputeStereoMatches does
        # Frame::ComputeStereoMatches in full: row band, SAD refinement on the device pyramids, median gate
        m.stereo_match_device(ex, 0, ex, 1, kps[0].data_ptr(), desc[0].data_ptr(), int(n[0]), kps[1].data_ptr(),
                              desc[1].data_ptr(), int(n[1]), bf, b, ur.data_ptr(), dep.data_ptr(), sync=True)
        if it >= 3:
            lat.append(time.perf_counter() - t0)
    out["kitti_stereo_pair_1241x376_2000kp"] = {
        "latency_ms_pair_extract_plus_stereo_matches": round(float(np.median(lat)) * 1e3, 4),
        "pairs_per_s_sequential": round(1.0 / float(np.median(lat)), 1),
        "left_keypoints": int(n[0]), "right_keypoints": int(n[1]),
        "points_with_depth": int((dep[:int(n[0])] > 0).sum().item())}
    if with_cpu:
        # the same pair on the CPU oracle: two extractions + ComputeStereoMatches (src/Frame.cc:466-640), one core; and pairs in
        # parallel, one pair per core, estimated from the per-pair time (the oracle's stereo step has no threaded driver)
        sys.path.insert(0, str(ROOT / "tests"))
        import oracle_lib as O
        el, er = (O.OracleExtractor(nfeat, SCALE, NLEVELS, INI_TH, MIN_TH) for _ in range(2))
        tp = []
        for it in range(6):
            t0 = time.perf_counter()
            kl_, dl_ = el.extract(left)
            kr_, dr_ = er.extract(right)
            t1 = time.perf_counter()
            O.stereo_full(el, er, kl_, dl_, kr_, dr_, bf, b)
            t2 = time.perf_counter()
            if it:
                tp.append((t2 - t0, t2 - t1))
        tp = np.array(tp)
        out["kitti_stereo_pair_1241x376_2000kp"]["cpu_oracle"] = {
            "kind": "port", "cores": 1, "latency_ms_pair_extract_plus_stereo_matches": round(float(np.median(tp[:, 0])) * 1e3, 3),
            "of_which_stereo_matches_ms": round(float(np.median(tp[:, 1])) * 1e3, 3),
            "note": "the reference runs the two extractions on two threads (src/Frame.cc:78-81): halve the extraction part for that"}
    # DBoW2 transform (N2): tree descent of ORB descriptors down a synthetic 10-ary, 6-level vocabulary (the ORBvoc.txt
    # shape: 1.1 M nodes, 10^6 words), one frame's worth and a 512-frame batch
    co, ci, nd, _wi, _w = synth.regular_vocabulary(10, 6, 31337)
    voc = capi.Vocabulary(co, ci, nd, 6, device=local_rank)
    nbig = 512 * 2000
    feats = torch.from_numpy(synth.random_descriptors(nbig, 99)).to(dev)
    leaf = torch.zeros(nbig, dtype=torch.int32, device=dev)
    nid = torch.zeros(nbig, dtype=torch.int32, device=dev)
    for _ in range(2):
        m.bow_transform_device(voc, feats.data_ptr(), nbig, 4, leaf.data_ptr(), nid.data_ptr(), sync=True)
    t0 = time.perf_counter()
    for _ in range(5):
        m.bow_transform_device(voc, feats.data_ptr(), nbig, 4, leaf.data_ptr(), nid.data_ptr(), sync=False)
    m.synchronize()
    big = (time.perf_counter() - t0) / 5
    lat = []
    for _ in range(30):
        t0 = time.perf_counter()
        m.bow_transform_device(voc, feats.data_ptr(), 2000, 4, leaf.data_ptr(), nid.data_ptr(), sync=True)
        lat.append(time.perf_counter() - t0)
    out["bow_transform_k10_L6"] = {"descriptors_per_s_batched": round(nbig / big, 1), "batch_descriptors": nbig,
                                   "latency_ms_2000_descriptors": round(float(np.median(lat)) * 1e3, 4),
                                   "hamming_distances_per_descriptor": 60}
    # Frame grid + windowed search (N4): the SearchByProjection(Frame&, vpMapPoints, th) candidate loop for 1500 projected
    # map points against the left frame's keypoints; host queries in, host results out (synchronous C call)
    nl = int(n[0])
    khost = kps[0, :nl].cpu().numpy()
    dhost = desc[0, :nl].cpu().numpy()
    kx, ky = np.ascontiguousarray(khost[:, 0]), np.ascontiguousarray(khost[:, 1])
    koct = np.ascontiguousarray(khost[:, 5].view(np.int32))
    urh = ur[:nl].cpu().numpy()
    grid = capi.FrameGrid(kx, ky, koct, dhost, urh, 0.0, 0.0, float(np.float32(64) / np.float32(W)),
                          float(np.float32(48) / np.float32(H)), device=local_rank)
    rng = np.random.default_rng(7)
    nq = 1500
    src = rng.integers(0, nl, nq)
    q = np.zeros(nq, capi.AREA_QUERY_DTYPE)
    q["x"], q["y"] = kx[src] + rng.uniform(-3, 3, nq), ky[src] + rng.uniform(-3, 3, nq)
    q["r"] = (4.0 * 1.2 ** koct[src]).astype(np.float32)
    q["min_level"], q["max_level"] = koct[src] - 1, koct[src]
    q["proj_xr"], q["er_max"] = q["x"] - 20, q["r"]
    qd = np.ascontiguousarray(dhost[src])
    res = grid.search(m, q, qd, None)
    lat = []
    for _ in range(30):
        t0 = time.perf_counter()
        res = grid.search(m, q, qd, None)
        lat.append(time.perf_counter() - t0)
    out["search_by_projection_1500_windows"] = {"latency_ms_host_queries_to_host_results": round(float(np.median(lat)) * 1e3, 4),
                                                "frame_keypoints": nl, "candidates_per_window": round(float(res["n_candidates"].mean()), 2),
                                                "windows_with_match": int((res["best_idx"] >= 0).sum())}
    # SearchByBoW loops (a-11): 1500 key-frame features against 2000 frame features spread over 100 shared vocabulary nodes
    # (ORBvoc: 10-ary, 6 levels, FeatureVector at level 2), host arrays in, host matches out
    n1, n2, nn = 1500, 2000, 100
    d2 = synth.random_descriptors(n2, 123)
    srcb = rng.integers(0, n2, n1)
    d1 = d2[srcb].copy()
    d1[:, :3] ^= rng.integers(0, 256, (n1, 3)).astype(np.uint8)
    node2 = rng.integers(0, nn, n2)
    node1 = node2[srcb]

    def csr(node):
        order = [np.nonzero(node == k)[0] for k in range(nn)]
        return (np.concatenate([[0], np.cumsum([len(o) for o in order])]).astype(np.int32), np.concatenate(order).astype(np.int32))
    (o1, i1), (o2, i2) = csr(node1), csr(node2)
    mt = m.search_by_bow(d1, None, d2, None, o1, i1, o2, i2, 50, 0.7)
    lat = []
    for _ in range(30):
        t0 = time.perf_counter()
        mt = m.search_by_bow(d1, None, d2, None, o1, i1, o2, i2, 50, 0.7)
        lat.append(time.perf_counter() - t0)
    out["search_by_bow_1500x2000_100_nodes"] = {"latency_ms_host_arrays_to_host_matches": round(float(np.median(lat)) * 1e3, 4),
                                                "matches": int((mt >= 0).sum()),
                                                "hamming_distances": int(sum((o1[k + 1] - o1[k]) * (o2[k + 1] - o2[k]) for k in range(nn)))}
    # BASELINE.json configs[3] end to end: a KITTI-00-length sequence (4541 frames) in pinned host memory -> keypoints,
    # descriptors and counts in pinned host memory (orbfe_extract_sequence: PCIe both ways, two chunks in flight)
    nseq, capq = 4541, NFEATURES + 32
    stride = (W + 63) // 64 * 64
    uniq = synth.frame_sequence(8, W, H, first=700)
    seq = torch.zeros((nseq, H, stride), dtype=torch.uint8).pin_memory()
    sv = seq.numpy()
    for i in range(nseq):
        sv[i, :, :W] = uniq[i % 8]
    skp = torch.zeros((nseq, capq, 7), dtype=torch.float32).pin_memory()
    sde = torch.zeros((nseq, capq, 32), dtype=torch.uint8).pin_memory()
    scn = torch.zeros(nseq, dtype=torch.int32).pin_memory()
    exs = capi.ORBextractor(NFEATURES, SCALE, NLEVELS, INI_TH, MIN_TH, device=local_rank)
    seq_call = lambda: exs.extract_sequence_host(seq.data_ptr(), nseq, W, H, stride, stride * H, skp.data_ptr(), sde.data_ptr(),
                                                 scn.data_ptr(), capq, 256)
    seq_call()
    lat = []
    for _ in range(3):
        t0 = time.perf_counter()
        seq_call()
        lat.append(time.perf_counter() - t0)
    tseq = float(np.median(lat))
    out["sequence_4541_frames_pinned_host_to_pinned_host"] = {
        "frames_per_s": round(nseq / tseq, 1), "seconds": round(tseq, 4), "chunk_frames": 256,
        "host_to_device_GBps": round(nseq * stride * H / tseq / 1e9, 2), "device_to_host_GBps": round(nseq * capq * 60 / tseq / 1e9, 2),
        "keypoints_per_frame": round(float(scn.float().mean()), 1)}
    del seq, skp, sde, scn, exs
    # Fuse(KeyFrame*, vpMapPoints, th) selection (a-12): the same 1500 windows with th = 3 and the reprojection gate on the device
    sf = np.ones(8, np.float32)
    for i in range(1, 8):
        sf[i] = sf[i - 1] * np.float32(1.2)
    sigma2 = (sf * sf).astype(np.float32)
    q["r"] = (3.0 * sf[koct[src]]).astype(np.float32)
    q["er_max"] = -1
    inv_s2 = (np.float32(1) / sigma2).astype(np.float32)
    res = grid.search_ex(m, q, qd, inv_level_sigma2=inv_s2)
    lat = []
    for _ in range(30):
        t0 = time.perf_counter()
        res = grid.search_ex(m, q, qd, inv_level_sigma2=inv_s2)
        lat.append(time.perf_counter() - t0)
    out["fuse_selection_1500_windows"] = {"latency_ms_host_queries_to_host_results": round(float(np.median(lat)) * 1e3, 4),
                                          "fused": int((res["best_dist"] <= 50).sum())}
    # SearchForTriangulation loops (a-12): the same two feature sets and nodes, epipolar geometry of a sideways camera move
    xy2 = rng.uniform([0, 0], [W, H], (n2, 2)).astype(np.float32)
    xy1 = (xy2[srcb] + np.stack([rng.uniform(5, 60, n1), rng.normal(0, 1.0, n1)], 1)).astype(np.float32)
    oct2 = rng.integers(0, 8, n2).astype(np.int32)
    F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
    z1, z2 = np.zeros(n1, np.uint8), np.zeros(n2, np.uint8)
    u1, u2 = np.full(n1, -1, np.float32), np.full(n2, -1, np.float32)
    tri = lambda: m.search_for_triangulation(d1, z1, u1, xy1, d2, z2, u2, xy2, oct2, o1, i1, o2, i2, F12, -500.0, 180.0, sf, sigma2)
    mt = tri()
    lat = []
    for _ in range(30):
        t0 = time.perf_counter()
        mt = tri()
        lat.append(time.perf_counter() - t0)
    out["search_for_triangulation_1500x2000_100_nodes"] = {"latency_ms_host_arrays_to_host_matches": round(float(np.median(lat)) * 1e3, 4),
                                                           "matches": int((mt >= 0).sum())}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=640,
                    help="most frames one launch (orbfe_extract_batch call) carries; a rank's block is cut into EQUAL launches of "
                         "at most this many frames (profiles/r3_launch_curve.json: throughput is flat from ~320 frames per launch "
                         "up, 568 measured best, and falls below: 284 -> -2 %%, 64 -> -11 %%)")
    ap.add_argument("--total-frames", type=int, default=4541,
                    help="frames per step over ALL ranks, split in contiguous blocks of ceil(total / G) (BASELINE.json configs[3], "
                         "strong scaling); 0: --batch frames per rank and step (weak scaling)")
    ap.add_argument("--handles", type=int, default=1,
                    help="extractor handles (one HIP stream each) the step's frames are split over; the reference itself "
                         "runs two extractor objects concurrently (src/Frame.cc:78-81)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hamming", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the TUM / EuRoC / stereo side measurements")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from orb_slam2_amd import capi, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        print("bench.py: --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)", file=sys.stderr)
        sys.exit(2)
    # ORBFE_BENCH_BACKEND=gloo rehearses the N > 1 code path on a box with fewer GPUs than ranks: the ranks share the
    # GPUs round-robin and the (tiny) collectives go through host tensors.  The driver's runs use the default, RCCL.
    backend = os.environ.get("ORBFE_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if backend == "nccl" else torch.device("cpu")      # where collective operands live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- synthetic frames, resident in HBM before anything is timed
    strong = args.total_frames > 0
    f_first, B, launch = plan_rank(args.total_frames, world, rank, args.batch)
    stride = (W + 63) // 64 * 64
    uniq = synth.frame_sequence(N_UNIQUE, W, H, first=0)
    uniq_dev = torch.zeros((N_UNIQUE, H, stride), dtype=torch.uint8)
    uniq_dev[:, :, :W] = torch.from_numpy(np.ascontiguousarray(uniq))
    uniq_dev = uniq_dev.to(dev)
    # frame k of the sequence is synthetic frame k mod 16 (tiled on the device: 4541 x 481 280 B = 2.2 GB resident)
    frames = uniq_dev[(torch.arange(max(B, 1), device=dev) + f_first) % N_UNIQUE].contiguous()
    B = max(B, 1) if not strong else B
    if B == 0:
        frames = frames[:0]
    cap = NFEATURES + 4 * NLEVELS + 32
    nb = max(B, 1)
    kps = torch.zeros((nb, cap, 7), dtype=torch.float32, device=dev)
    desc = torch.zeros((nb, cap, 32), dtype=torch.uint8, device=dev)
    cnt = torch.zeros(nb, dtype=torch.int32, device=dev)
    # The step's B frames are split over `handles` extractor objects.  Each owns a HIP stream, so the latency-bound
    # stages of one half (pyramid chain, quadtree) run under the VALU-bound stages of the other (FAST, blur, describe).
    nh = max(1, min(args.handles, nb))
    bounds = [B * k // nh for k in range(nh + 1)]
    exs = [capi.ORBextractor(NFEATURES, SCALE, NLEVELS, INI_TH, MIN_TH, device=local_rank) for _ in range(nh)]

    def step():
        # asynchronous on the library's streams: the launches of all K steps queue back to back, as an offline extraction
        # job would; every frame of the rank's block is extracted once per step, its outputs in its own slot
        for k, ex in enumerate(exs):
            for f0 in range(bounds[k], bounds[k + 1], launch):
                n = min(launch, bounds[k + 1] - f0)
                ex.extract_batch_device(frames[f0].data_ptr(), n, W, H, stride, stride * H, kps[f0].data_ptr(),
                                        desc[f0].data_ptr(), cnt[f0:].data_ptr(), cap, sync=False)

    def sync_all():
        for ex in exs:
            ex.synchronize()    # the library's own streams (torch.cuda.synchronize does not cover non-blocking streams)

    for _ in range(args.warmup):
        step()
    sync_all()
    for ex in exs:
        ex.set_profiling(2)     # two HIP events per step on the library's stream, around the FAST kernel; never blocks
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    barrier()
    dt = time.perf_counter() - t0
    profs = [ex.profile() for ex in exs]
    # stage table: a short un-timed pass with events around every stage
    for ex in exs:
        ex.set_profiling(1)
    for _ in range(5):
        step()
    sync_all()
    stage_profs = [ex.profile() for ex in exs]
    for ex in exs:
        ex.set_profiling(0)
    stages = {k: round(sum(p["stage_ms"][k] for p in stage_profs) / 5, 4) for k in stage_profs[0]["stage_ms"]}
    prof = {"stage_ms": {k: sum(p["stage_ms"][k] for p in profs) for k in profs[0]["stage_ms"]},
            "stage_launches": {k: sum(p["stage_launches"][k] for p in profs) for k in profs[0]["stage_launches"]},
            "frames": sum(p["frames"] for p in profs), "candidates": sum(p["candidates"] for p in profs),
            "keypoints": sum(p["keypoints"] for p in profs)}
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    total_frames = (args.total_frames if strong else B * world) * args.steps
    fps = total_frames / dt
    kp_per_frame = float(cnt.float().mean().item()) if B else 0.0

    # ---- roofline of the dominant kernel (FAST): algorithmic bytes = level pixels read + 8 B per candidate written
    fast_ms = prof["stage_ms"]["fast"] / max(prof["stage_launches"]["fast"], 1)
    cand_per_frame = sum(p["candidates"] for p in stage_profs) / max(sum(p["frames"] for p in stage_profs), 1)
    frames_per_launch = prof["frames"] / max(prof["stage_launches"]["fast"], 1)
    fast_bytes = frames_per_launch * (LEVEL_PIXELS + 8 * cand_per_frame)
    achieved = fast_bytes / (fast_ms * 1e-3) / 1e9 if fast_ms > 0 else 0.0
    # traffic and instruction count: the rocprofv3 --pmc passes recorded in profiles/latest_pmc.json (tools/pmc_passes.sh,
    # tools/pmc_latest.py), per frame, scaled to the frames one launch of this run processes.  FETCH_SIZE x 2: gfx950 tallies
    # 128-byte fabric requests at 64 bytes (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact; both in KB.
    pmc, pmc_stale = latest_pmc()
    fk = (pmc or {}).get("kernels", {}).get("fast_cells_kernel", {})
    pmc_frames = max((pmc or {}).get("frames_per_launch", 1), 1)
    traffic = None
    if "FETCH_SIZE" in fk and "WRITE_SIZE" in fk:
        traffic = int(frames_per_launch * (2 * fk["FETCH_SIZE"] + fk["WRITE_SIZE"]) * 1024 / pmc_frames)
    roofline = {"bound": "hbm", "kernel": "fast_cells_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "ms_per_launch": round(fast_ms, 4), "algorithmic_bytes_per_launch": int(fast_bytes),
                "frames_per_launch": round(frames_per_launch, 2),
                "pmc": {"file": "profiles/latest_pmc.json", "stale": bool(pmc_stale),
                        "csrc_sha256": (pmc or {}).get("csrc_sha256", "")[:16]},
                "note": "FAST is integer-VALU-bound, not HBM-bound (valu_issue below: SQ_INSTS_VALU x 4.15 cycles against the kernel's "
                        "SIMD cycles); ms_per_launch is the HIP-event time around the kernel inside the timed region (with "
                        "--handles > 1 the other handle's kernels run concurrently and inflate it); stage_ms_per_step "
                        "comes from 5 extra un-timed steps with events around every stage",
                "stage_ms_per_step": stages}
    # the resource FAST is actually bound by: VALU issue slots.  SQ_INSTS_VALU per frame from the same PMC file (the count
    # depends on the frames, not on the run), a packed-integer instruction occupies a SIMD for 4.15 cycles
    # (profiles/r1_valu_rates_ubench.txt), 1024 SIMDs at 2.4 GHz.
    if "SQ_INSTS_VALU" in fk:
        valu_instr = fk["SQ_INSTS_VALU"] / pmc_frames * frames_per_launch
        roofline["valu_issue"] = {"wave_instructions_per_launch": int(valu_instr), "cycles_per_instruction": 4.15,
                                  "frac": round(valu_instr * 4.15 / (1024 * 2.4e9 * fast_ms * 1e-3), 4) if fast_ms > 0 else None,
                                  "note": "instruction count from profiles/latest_pmc.json, duration live; 2.4 GHz nominal clock"}
    # every dense stage against HBM (algorithmic bytes per frame, each datum moved once), from the
    # un-timed per-stage pass; the descriptor stage by the unique bytes it gathers (749 B disc + <= 1369 B window + 60 B out)
    spf = sum(p["frames"] for p in stage_profs) / 5
    kp_pf = sum(p["keypoints"] for p in stage_profs) / max(sum(p["frames"] for p in stage_profs), 1)
    stage_bytes = {"pyramid": 2385248, "fast": LEVEL_PIXELS + 8 * cand_per_frame, "blur": 2 * LEVEL_PIXELS,
                   "describe": (749 + 1369 + 60) * kp_pf, "quadtree": 8 * cand_per_frame}
    roofline["stage_hbm"] = {k: {"algorithmic_GBps": round(stage_bytes[k] * spf / (stages[k] * 1e-3) / 1e9, 1),
                                 "frac_of_8TBps": round(stage_bytes[k] * spf / (stages[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                             for k in stage_bytes if stages.get(k, 0) > 0}
    # the same stages by the HBM bytes the PMC passes counted (FETCH_SIZE x 2 + WRITE_SIZE, per launch of 512 frames in
    # profiles/latest_pmc.json; the pyramid is nlevels - 1 launches), against this run's stage times
    stage_kernel = {"pyramid": ("pyr_resize_kernel", NLEVELS - 1), "fast": ("fast_cells_kernel", 1), "blur": ("gauss7_kernel", 1),
                    "describe": ("orient_describe_kernel", 1), "quadtree": ("quadtree_kernel", 1)}
    for k, (kn, nl) in stage_kernel.items():
        c = (pmc or {}).get("kernels", {}).get(kn, {})
        if k in roofline["stage_hbm"] and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            per_frame = nl * (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / pmc_frames
            gbs = per_frame * spf / (stages[k] * 1e-3) / 1e9
            roofline["stage_hbm"][k].update({"pmc_bytes_per_frame": int(per_frame), "pmc_traffic_GBps": round(gbs, 1),
                                             "pmc_frac_of_8TBps": round(gbs / HBM_PEAK_GBS, 4)})

    # ---- Hamming kNN-2: 2000 queries vs a 1M database sharded over the ranks, RCCL gather of per-shard results
    hamming = None
    if not args.no_hamming:
        NQ, NDB = 2000, 1_000_000
        shard = NDB // world
        db = torch.from_numpy(synth.random_descriptors(shard, 7000 + rank)).to(dev)
        q_host = synth.random_descriptors(NQ, 42)
        q = torch.from_numpy(q_host).to(dev)
        res = torch.zeros((3, NQ), dtype=torch.int32, device=dev)
        m = capi.ORBmatcher(device=local_rank)
        # N > 1: the exchange runs inside liborbfe (orbfe_knn2_sharded_run: ncclBroadcast of the queries, local scan,
        # ncclAllGather of the per-shard triples, fold kernel); torch.distributed only ships the 128-byte RCCL id.
        # ORBFE_BENCH_BACKEND=gloo (rehearsal on fewer GPUs than ranks, where RCCL cannot form a communicator) keeps the
        # round-1 path: host-side collectives + orbfe_merge_knn2.
        sharded = None
        gathered = None
        if world > 1 and backend == "nccl":
            # Agree on the path BEFORE anything that can block on another rank: a rank that cannot bind librccl (or rank 0
            # failing to make the id) must not leave the others waiting in a broadcast or inside ncclCommInitRank.
            # orbfe_knn2_sharded_available only binds the library; the flag goes through torch.distributed (MIN).
            err = ""
            idb = b""
            ok_local = 1
            try:
                if not capi.sharded_available():
                    ok_local, err = 0, capi.lib().orbfe_last_error().decode()
                elif rank == 0:
                    idb = capi.sharded_unique_id()
            except Exception as e:
                ok_local, err = 0, str(e)
            ok = torch.tensor([ok_local], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                idt = torch.zeros(128, dtype=torch.uint8, device=dev)
                if rank == 0:
                    idt.copy_(torch.frombuffer(bytearray(idb), dtype=torch.uint8))
                dist.broadcast(idt, src=0)
                torch.cuda.synchronize()
                try:
                    # collective (ncclCommInitRank): every rank is known to have a working librccl at this point; a failure
                    # from here on is RCCL's own (symmetric: the communicator does not form on any rank)
                    sharded = capi.ShardedKnn2(m, bytes(idt.cpu().numpy().tobytes()), rank, world, NQ)
                except Exception as e:
                    err = str(e)
                ok = torch.tensor([1 if sharded is not None else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                errs = [None] * world
                dist.all_gather_object(errs, err)
                if rank == 0:
                    why = "; ".join(f"rank {r}: {e}" for r, e in enumerate(errs) if e) or "unknown"
                    print(f"[bench] RCCL inside liborbfe is not available on every rank ({why}); falling back to "
                          "torch.distributed + orbfe_merge_knn2", file=sys.stderr, flush=True)
                sharded = None
        if world > 1 and sharded is None:
            gathered = torch.zeros((world * 3, NQ), dtype=torch.int32, device=cdev)

        def hstep(sync=True):
            if sharded is not None:
                sharded.run(q.data_ptr(), NQ, db.data_ptr(), shard, rank * shard, res[0].data_ptr(), res[1].data_ptr(),
                            res[2].data_ptr(), sync=sync)
                return None
            if world > 1:
                qc = q if cdev.type == "cuda" else q.cpu()
                dist.broadcast(qc, src=0)
                if qc is not q:
                    q.copy_(qc)
                torch.cuda.synchronize()      # the matcher runs on its own non-blocking stream: order it after the broadcast
            m.knn2_device(q.data_ptr(), NQ, db.data_ptr(), shard, res[0].data_ptr(), res[1].data_ptr(), res[2].data_ptr(),
                          idx_base=rank * shard, sync=sync or world > 1)
            if world > 1:
                torch.cuda.synchronize()
                dist.all_gather_into_tensor(gathered, res if cdev.type == "cuda" else res.cpu())
                g = gathered.view(world, 3, NQ).cpu().numpy()
                return capi.merge_knn2(g[:, 0], g[:, 1], g[:, 2])
            return None

        def htime(sync, hsteps):
            """hsteps query batches: with a host synchronisation after every batch (a tracker that needs each answer before it
            goes on), or queued back to back on the matcher's stream and drained once (an offline job: several query batches
            in flight, so the exchange of one batch runs under the scan of the next)"""
            for _ in range(3):
                hstep(sync)
            m.synchronize()
            barrier()
            h0 = time.perf_counter()
            for _ in range(hsteps):
                hstep(sync)
            m.synchronize()
            barrier()
            t = time.perf_counter() - h0
            if world > 1:
                tt = torch.tensor([t], dtype=torch.float64, device=cdev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                t = float(tt.item())
            return t

        hsteps = 10
        hdt = htime(True, hsteps)
        # back to back only where the whole exchange is stream-ordered (one GPU, or RCCL inside the library)
        hdt_b2b = htime(False, 4 * hsteps) / 4 if (world == 1 or sharded is not None) else None
        pairs = NQ * shard * world * hsteps
        # Integer-VALU roofline of the pair loop, from THIS repository's measurements (no literals):
        #   ceiling  -- the scan's own instruction pair, v_xor_b32 v, s, v (reference dword from an SGPR: half rate, 4.15 cycles,
        #               not the 2.7 of the all-VGPR form) feeding v_bcnt_u32_b32 (accumulating), 16 instructions per wave and
        #               reference, at the cycles per instruction tools/ubench/valu_rates.hip measured for exactly that chain
        #               (profiles/valu_rates.json, 8 waves per SIMD): 1024 SIMDs x 2.4 GHz x 64 lanes / (16 x cycles)
        #   achieved -- pairs / time; instructions per pair from SQ_INSTS_VALU of the kNN kernels (profiles/latest_pmc.json,
        #               hash-checked against the sources like the FAST counters)
        rates = {}
        try:
            rates = json.loads((ROOT / "profiles" / "valu_rates.json").read_text()).get("rates", {})
        except (OSError, ValueError):
            pass
        chain_cyc = rates.get("xor(s)+bcnt chain @8 waves/SIMD")
        peak_pairs = world * 1024 * 2.4e9 * 64 / (16 * chain_cyc) if chain_cyc else None
        knn_pmc = (pmc or {}).get("knn2") or {}
        instr_per_call = (knn_pmc.get("per_call") or {}).get("SQ_INSTS_VALU")
        instr_per_pair = instr_per_call * 64 / knn_pmc["pairs_per_call"] if instr_per_call else None
        best = max(pairs / hdt, (pairs / hdt_b2b) if hdt_b2b else 0.0)
        hamming = {"metric": "hamming_pair_distances_per_sec", "value": round(pairs / hdt, 1), "ms_per_query_batch": round(hdt / hsteps * 1e3, 3),
                   "queries": NQ, "database": shard * world, "shards": world,
                   "host_sync_per_batch": {"pairs_per_s": round(pairs / hdt, 1), "ms_per_query_batch": round(hdt / hsteps * 1e3, 3)},
                   "back_to_back": None if hdt_b2b is None else {"pairs_per_s": round(pairs / hdt_b2b, 1),
                                                                 "ms_per_query_batch": round(hdt_b2b / hsteps * 1e3, 3),
                                                                 "batches_queued": 4 * hsteps},
                   # "matches/sec" of BASELINE.json's metric: queries matched (best + second-best over the whole database, the
                   # input of the ratio test) per second, in the faster of the two modes
                   "matches_per_s": round(NQ * hsteps / min(hdt, hdt_b2b or hdt), 1),
                   "exchange": "none (one shard)" if world == 1 else ("RCCL inside liborbfe (orbfe_knn2_sharded_run)" if sharded is not None
                                                                      else "host collectives (rehearsal backend) + orbfe_merge_knn2"),
                   "valu": {"achieved_Tpairs": round(best / 1e12, 3),
                            "peak_Tpairs": round(peak_pairs / 1e12, 3) if peak_pairs else None,
                            "frac": round(best / peak_pairs, 4) if peak_pairs else None,
                            "of": "back_to_back" if hdt_b2b and pairs / hdt_b2b >= pairs / hdt else "host_sync_per_batch",
                            "ceiling": {"instructions_per_pair": 16, "cycles_per_instruction": chain_cyc,
                                        "source": "profiles/valu_rates.json: 'xor(s)+bcnt chain @8 waves/SIMD' (tools/ubench/valu_rates.hip, "
                                                  "k_hamming_chain: the scan's v_xor_b32 v,s,v + v_bcnt_u32_b32 sequence)"},
                            "valu_instructions_per_pair": round(instr_per_pair, 3) if instr_per_pair else None,
                            "pmc": {"file": "profiles/latest_pmc.json", "stale": bool(pmc_stale), "calls_profiled": knn_pmc.get("calls_profiled")},
                            "note": "8 v_xor (SGPR operand) + 8 v_bcnt_u32_b32 (accumulating) per pair + v_min3/v_min/v_cmp per 4 "
                                    "references + the sample pre-pass and the folds; the best/second selection runs only where a "
                                    "reference can still matter (sample-seeded threshold)"}}

        if sharded is not None:
            # the library's RCCL communicator goes before torch's process group does: every rank has passed the barrier of the
            # last htime() and drained its stream
            barrier()
            sharded.close()
            sharded = None

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(uniq)
        if hamming is not None:
            # the matcher's CPU side: the reference's SWAR popcount loop (src/ORBmatcher.cc:1654-1660) and a
            # __builtin_popcount variant of the same best / second-best scan, one block of queries per host thread
            sys.path.insert(0, str(ROOT / "tests"))
            import oracle_lib as O
            threads = host_cores()["threads"]
            db_host = db.cpu().numpy()
            ns = len(db_host)       # 2e9 pairs: a few tenths of a second on the GPU box's 16 host threads
            t_swar = O.time_knn2(q_host, db_host[:ns], threads, swar=True)
            t_pop = O.time_knn2(q_host, db_host, threads, swar=False)
            hamming["cpu_baseline"] = {"value": round(NQ * ns / t_swar, 1), "unit": "pair distances/s", "cores": threads, "kind": "port",
                                       "sample": f"{NQ} queries x {ns} descriptors with the reference's SWAR DescriptorDistance in {t_swar:.2f} s",
                                       "popcnt_variant": round(NQ * len(db_host) / t_pop, 1),
                                       "popcnt_sample": f"{NQ} x {len(db_host)} with __builtin_popcount in {t_pop:.2f} s"}

    extras = None
    if rank == 0 and world == 1 and not args.no_extras:
        extras = other_configs(local_rank, dev, with_cpu=not args.no_cpu_baseline)
        if nh == 1 and B > 2 * launch:
            # the same step with its launches dealt round-robin to three handles (three HIP streams): the latency-bound stages of
            # one launch run under the issue-bound stages of another.  Not the headline: the event time around FAST, and with it
            # roofline.frac, is stretched by whatever else is resident.
            exs3 = exs + [capi.ORBextractor(NFEATURES, SCALE, NLEVELS, INI_TH, MIN_TH, device=local_rank) for _ in range(2)]

            def step3():
                for i, f0 in enumerate(range(0, B, launch)):
                    n = min(launch, B - f0)
                    exs3[i % 3].extract_batch_device(frames[f0].data_ptr(), n, W, H, stride, stride * H, kps[f0].data_ptr(),
                                                     desc[f0].data_ptr(), cnt[f0:].data_ptr(), cap, sync=False)

            step3()
            for ex in exs3:
                ex.synchronize()
            t3 = time.perf_counter()
            for _ in range(5):
                step3()
            for ex in exs3:
                ex.synchronize()
            t3 = (time.perf_counter() - t3) / 5
            extras["same_step_on_3_handles"] = {"frames_per_s": round(B / t3, 1), "ms_per_step": round(t3 * 1e3, 3), "handles": 3}

    if rank == 0:
        out = {
            "metric": "orb_extract_frames_per_sec_1241x376_2000kp_8lvl", "value": round(fps, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[3]: batched offline ORB extraction of 4541 synthetic 1241x376 frames (KITTI-00 "
                                    "length) resident in HBM, 2000 features, 8 levels, scale 1.2, FAST 20/7, contiguous blocks of "
                                    "ceil(4541 / G) frames per GPU" if strong and args.total_frames == 4541 else
                                    "batched ORB extraction, 1241x376 KITTI-size frames resident in HBM, 2000 features, 8 levels, "
                                    "scale 1.2, FAST 20/7"),
                       "frames_per_step": args.total_frames if strong else B * world, "frames_per_step_rank0": B,
                       "frames_per_launch": launch, "handles": nh, "keypoints_per_frame": round(kp_per_frame, 1),
                       "parallelism": f"frames sharded over {world} gpu(s), no collective"},
            "roofline": roofline, "cpu_baseline": cpu, "hamming": hamming, "other_configs": extras,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
