"""Golden fixtures of the ORB front end.

The reference has no tests and cannot be built here (OpenCV is absent), so the fixtures under tests/golden/ are outputs
of the CPU oracle on seeded synthetic inputs, written by tests/golden/make_golden.py in the build container and
committed ("parity unpinned": they pin the oracle and the HIP path to each other and to history, not to a run of the
reference).  This module defines the cases and how raw outputs are reduced to digests; the two providers (oracle, HIP
through the C-ABI) live in make_golden.py / test_golden.py.  Nothing here imports the oracle or the library."""
import hashlib
import json
import zlib
from pathlib import Path

import numpy as np

from orb_slam2_amd import synth

GOLDEN = Path(__file__).resolve().parent / "golden" / "front_end.json"

EXTRACT_CASES = {            # BASELINE.json configs: KITTI stereo, TUM mono / RGB-D, EuRoC
    "kitti_1241x376_2000": dict(w=1241, h=376, nfeatures=2000, seed=11),
    "tum_640x480_1000": dict(w=640, h=480, nfeatures=1000, seed=12),
    "euroc_752x480_1200": dict(w=752, h=480, nfeatures=1200, seed=13),
    "small_233x229_300": dict(w=233, h=229, nfeatures=300, seed=14),   # level 7 is 65x64: one FAST cell
}
STEREO = dict(w=1241, h=376, nfeatures=2000, seed=15, bf=386.1448, b=0.5371657)   # KITTI00-02.yaml
KNN = dict(nq=300, nr=5000, seed=16)
BOW = dict(k=10, L=4, seed=17, n=3000, levelsup=2)
DISTINCTIVE = dict(seed=18, points=400)
AREA = dict(seed=19, nq=800)
BOWMATCH = dict(seed=20, n1=1200, n2=1500, nodes=90, max_dist=50, ratio=0.7)
FUSE = dict(seed=21, nq=1200)
TRIANGULATION = dict(seed=22, n1=1500, nodes=80)


def crc(a) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def extract_inputs(name):
    c = EXTRACT_CASES[name]
    return synth.textured_image(c["w"], c["h"], synth.BASE_SEED + c["seed"]), c


def extract_digest(img, levels, blurred, candidates, selected, kps, desc):
    """levels / blurred: 8 u8 planes (blurred[l] may be None for a level without keypoints); candidates: 8 float32 [n,3]
    (x, y, response in the reference's order); selected: 8 arrays of kept keypoints; kps: 28-byte records; desc [n,32]."""
    per_level = []
    for l in range(len(levels)):
        per_level.append({
            "size": [int(levels[l].shape[1]), int(levels[l].shape[0])],
            "pyramid_crc32": crc(levels[l]),
            "blurred_crc32": None if blurred[l] is None else crc(blurred[l]),
            "fast_candidates": int(len(candidates[l])),
            "fast_candidates_crc32": crc(np.asarray(candidates[l], np.float32)),
            "kept": int(len(selected[l])),
        })
    raw = np.ascontiguousarray(kps).view(np.uint8).reshape(-1, 28)
    return {"image_crc32": crc(img), "levels": per_level, "keypoints": int(len(raw)),
            "first_32_keypoints_hex": [bytes(r).hex() for r in raw[:32]],
            "keypoints_sha256": sha(raw), "descriptors_sha256": sha(desc)}


def stereo_inputs():
    return synth.stereo_pair(STEREO["w"], STEREO["h"], synth.BASE_SEED + STEREO["seed"])


def stereo_digest(kl, kr, uright, depth):
    return {"left_keypoints": int(len(kl)), "right_keypoints": int(len(kr)), "with_depth": int((depth > 0).sum()),
            "mvuRight_sha256": sha(np.asarray(uright, np.float32)), "mvDepth_sha256": sha(np.asarray(depth, np.float32)),
            "first_matches": [[int(i), float(uright[i]), float(depth[i])] for i in np.nonzero(depth > 0)[0][:8]]}


def knn_inputs():
    db = synth.random_descriptors(KNN["nr"], 1000 + KNN["seed"])
    q = synth.planted_queries(db, KNN["nq"], 2000 + KNN["seed"], planted_frac=0.3)
    db[KNN["nr"] // 2] = db[3]
    q[0] = db[3]
    return q, db


def knn_digest(bd, bi, sd):
    out = np.stack([np.asarray(bd, np.int32), np.asarray(bi, np.int32), np.asarray(sd, np.int32)])
    return {"sha256": sha(out), "first_8": out[:, :8].T.tolist()}


def bow_inputs():
    co, ci, nd, wi, w = synth.random_vocabulary(BOW["k"], BOW["L"], 3000 + BOW["seed"])
    f = synth.random_descriptors(BOW["n"], 4000 + BOW["seed"])
    f[:200] = nd[(np.arange(200) * 11) % len(nd)]
    return (co, ci, nd, wi, w), f


def bow_digest(leaf, nid):
    out = np.stack([np.asarray(leaf, np.int32), np.asarray(nid, np.int32)])
    return {"sha256": sha(out), "first_8": out[:, :8].T.tolist()}


def distinctive_inputs():
    rng = np.random.default_rng(DISTINCTIVE["seed"])
    sizes = np.concatenate([rng.integers(0, 30, DISTINCTIVE["points"]), [64, 65, 100, 1, 0, 2]])
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    desc = synth.random_descriptors(int(off[-1]), 5000 + DISTINCTIVE["seed"])
    for p in range(0, len(sizes), 2):
        if sizes[p] > 1:
            base = desc[off[p]].copy()
            for i in range(off[p], off[p + 1]):
                d = base.copy()
                for b in rng.integers(0, 256, int(rng.integers(0, 25))):
                    d[b >> 3] ^= np.uint8(1 << (b & 7))
                desc[i] = d
    return desc, off


def distinctive_digest(best):
    best = np.asarray(best, np.int32)
    return {"sha256": sha(best), "first_16": best[:16].tolist()}


def area_inputs(kps, desc):
    """Windows around the keypoints of the KITTI extraction case (taken from the provider's own extraction)."""
    c = EXTRACT_CASES["kitti_1241x376_2000"]
    rng = np.random.default_rng(AREA["seed"])
    n, nq = len(kps), AREA["nq"]
    kx, ky, koct = kps["x"].copy(), kps["y"].copy(), kps["octave"].astype(np.int32)
    ur = np.where(rng.uniform(size=n) < 0.5, kx - rng.uniform(1, 60, n), -1).astype(np.float32)
    src = rng.integers(0, n, nq)
    q = {"x": (kx[src] + rng.uniform(-5, 5, nq)).astype(np.float32), "y": (ky[src] + rng.uniform(-5, 5, nq)).astype(np.float32)}
    q["r"] = (rng.choice([2.5, 4.0, 12.0], nq) * (1.2 ** koct[src])).astype(np.float32)
    q["min_level"], q["max_level"] = (koct[src] - 1).astype(np.int32), koct[src].astype(np.int32)
    q["proj_xr"] = (q["x"] - rng.uniform(1, 60, nq)).astype(np.float32)
    q["er_max"] = np.where(rng.uniform(size=nq) < 0.5, q["r"], -1).astype(np.float32)
    qd = desc[src].copy()
    for i in range(nq):
        for b in rng.integers(0, 256, 12):
            qd[i, b >> 3] ^= np.uint8(1 << (b & 7))
    blocked = (rng.uniform(size=n) < 0.2).astype(np.uint8)
    grid = dict(kx=kx, ky=ky, koct=koct, ur=ur, min_x=0.0, min_y=0.0,
                w_inv=float(np.float32(64) / np.float32(c["w"])), h_inv=float(np.float32(48) / np.float32(c["h"])))
    return grid, q, qd, blocked


def area_digest(rows):
    """rows: int32 [nq, 7] = best_dist, best_idx, best_level, best_dist2, best_level2, n_candidates, best_idx2"""
    rows = np.asarray(rows, np.int32)
    return {"sha256": sha(rows), "first_8": rows[:8].tolist(), "with_match": int((rows[:, 1] >= 0).sum())}


def bowmatch_inputs():
    c = BOWMATCH
    rng = np.random.default_rng(c["seed"])
    d2 = synth.random_descriptors(c["n2"], 6000 + c["seed"])
    src = rng.integers(0, c["n2"], c["n1"])
    d1 = d2[src].copy()
    for i in range(c["n1"]):
        for b in rng.integers(0, 256, int(rng.integers(0, 40))):
            d1[i, b >> 3] ^= np.uint8(1 << (b & 7))
    node2 = rng.integers(0, c["nodes"], c["n2"])
    node1 = node2[src].copy()
    stray = rng.uniform(size=c["n1"]) < 0.1
    node1[stray] = rng.integers(0, c["nodes"], int(stray.sum()))

    def csr(node):
        order = [np.nonzero(node == k)[0] for k in range(c["nodes"])]
        return (np.concatenate([[0], np.cumsum([len(o) for o in order])]).astype(np.int32), np.concatenate(order).astype(np.int32))
    valid1 = (rng.uniform(size=c["n1"]) < 0.85).astype(np.uint8)
    blocked2 = (rng.uniform(size=c["n2"]) < 0.1).astype(np.uint8)
    return d1, valid1, d2, blocked2, csr(node1), csr(node2)


def bowmatch_digest(match1):
    match1 = np.asarray(match1, np.int32)
    return {"sha256": sha(match1), "matches": int((match1 >= 0).sum()), "first_16": match1[:16].tolist()}


def scale_tables():
    sf = np.ones(8, np.float32)
    for i in range(1, 8):
        sf[i] = sf[i - 1] * np.float32(1.2)
    sigma2 = (sf * sf).astype(np.float32)
    return sf, sigma2, (np.float32(1.0) / sigma2).astype(np.float32)


def fuse_inputs(kps, desc):
    """MapPoints projected near the keypoints of the KITTI extraction case, for the selection loop of
    Fuse(KeyFrame*, vpMapPoints, th) (reference src/ORBmatcher.cc:889-952) and, with the limit array, the skip rule of
    SearchForInitialization (:444)."""
    c = EXTRACT_CASES["kitti_1241x376_2000"]
    rng = np.random.default_rng(FUSE["seed"])
    n, nq = len(kps), FUSE["nq"]
    kx, ky, koct = kps["x"].copy(), kps["y"].copy(), kps["octave"].astype(np.int32)
    sf, _, inv_sigma2 = scale_tables()
    ur = np.where(rng.uniform(size=n) < 0.5, kx - rng.uniform(1, 60, n), -1).astype(np.float32)
    src = rng.integers(0, n, nq)
    q = {"x": (kx[src] + rng.uniform(-3, 3, nq)).astype(np.float32), "y": (ky[src] + rng.uniform(-3, 3, nq)).astype(np.float32)}
    level = np.clip(koct[src] + rng.integers(0, 2, nq), 0, 7).astype(np.int32)
    q["r"] = (np.float32(3.0) * sf[level]).astype(np.float32)
    q["min_level"], q["max_level"] = (level - 1).astype(np.int32), level
    q["proj_xr"] = np.where(ur[src] >= 0, ur[src] + rng.uniform(-2, 2, nq), q["x"] - 20).astype(np.float32)
    q["er_max"] = np.full(nq, -1, np.float32)
    qd = desc[src].copy()
    for i in range(nq):
        for b in rng.integers(0, 256, 12):
            qd[i, b >> 3] ^= np.uint8(1 << (b & 7))
    limit = rng.choice(np.array([0, 15, 40, 2 ** 31 - 1], np.int32), n).astype(np.int32)
    grid = dict(kx=kx, ky=ky, koct=koct, ur=ur, min_x=0.0, min_y=0.0,
                w_inv=float(np.float32(64) / np.float32(c["w"])), h_inv=float(np.float32(48) / np.float32(c["h"])))
    return grid, q, qd, inv_sigma2, limit


def fuse_digest(best_idx_gate, rows_limit):
    """best_idx_gate: int32 [nq] keypoint fused into or -1; rows_limit: int32 [nq, 4] = best_dist, best_idx, best_dist2, best_idx2
    of the search with the vMatchedDistance-style limit"""
    a, b = np.asarray(best_idx_gate, np.int32), np.asarray(rows_limit, np.int32)
    return {"gate_sha256": sha(a), "fused": int((a >= 0).sum()), "gate_first_16": a[:16].tolist(),
            "limit_sha256": sha(b), "limit_first_8": b[:8].tolist()}


def triangulation_inputs(kps, desc):
    """SearchForTriangulation (:657-823): KF2 = the KITTI extraction case, KF1 = its keypoints seen from a camera shifted
    along x (same row up to noise), features spread over vocabulary nodes."""
    t = TRIANGULATION
    rng = np.random.default_rng(t["seed"])
    n2, n1 = len(kps), t["n1"]
    sf, sigma2, _ = scale_tables()
    x2, y2, oct2 = kps["x"].copy(), kps["y"].copy(), kps["octave"].astype(np.int32)
    src = rng.integers(0, n2, n1)
    x1 = (x2[src] + rng.uniform(5, 60, n1)).astype(np.float32)
    y1 = (y2[src] + rng.normal(0, 1.2, n1) * sf[oct2[src]]).astype(np.float32)
    d1 = desc[src].copy()
    for i in range(n1):
        for b in rng.integers(0, 256, int(rng.integers(0, 45))):
            d1[i, b >> 3] ^= np.uint8(1 << (b & 7))
    F12 = (np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32) + rng.normal(0, 2e-5, (3, 3))).astype(np.float32)
    node2 = rng.integers(0, t["nodes"], n2)
    node1 = node2[src].copy()
    stray = rng.uniform(size=n1) < 0.1
    node1[stray] = rng.integers(0, t["nodes"], int(stray.sum()))

    def csr(node):
        order = [np.nonzero(node == k)[0] for k in range(t["nodes"])]
        return (np.concatenate([[0], np.cumsum([len(o) for o in order])]).astype(np.int32), np.concatenate(order).astype(np.int32))
    return dict(d1=d1, mp1=(rng.uniform(size=n1) < 0.3).astype(np.uint8), ur1=np.where(rng.uniform(size=n1) < 0.5, x1 - 10, -1).astype(np.float32),
                xy1=np.stack([x1, y1], 1).astype(np.float32), d2=desc, mp2=(rng.uniform(size=n2) < 0.3).astype(np.uint8),
                ur2=np.where(rng.uniform(size=n2) < 0.5, x2 - 10, -1).astype(np.float32), xy2=np.stack([x2, y2], 1).astype(np.float32),
                oct2=oct2, csr1=csr(node1), csr2=csr(node2), F12=F12, ex=float(x2[0]) + 0.5, ey=float(y2[0]) - 0.5, sf=sf, sigma2=sigma2)


def triangulation_digest(m_all, m_stereo):
    a, b = np.asarray(m_all, np.int32), np.asarray(m_stereo, np.int32)
    return {"sha256": sha(np.stack([a, b])), "matches": int((a >= 0).sum()), "matches_only_stereo": int((b >= 0).sum()),
            "first_pairs": [[int(i), int(a[i])] for i in np.nonzero(a >= 0)[0][:8]]}


def load():
    with open(GOLDEN) as fh:
        return json.load(fh)
