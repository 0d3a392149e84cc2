"""The C++ adapter (include/ORBextractor.h + adapters/ORBextractor_orbfe.cc: the reference's class interface on the
C-ABI), driven like Frame::ExtractORB by tests/cpp/adapter_main, must reproduce the oracle bit for bit."""
import os
import subprocess
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as O
from orb_slam2_amd import capi, synth

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
CPP_BIN = Path(os.environ.get("ORBFE_TEST_BIN_DIR", ROOT / "tests" / "cpp"))   # tools/run_asan.sh points this at tests/cpp/_asan


def test_cpp_adapter_matches_oracle(tmp_path):
    exe = CPP_BIN / "adapter_main"
    assert exe.exists(), "build it with make -C orb_slam2_amd/csrc (done by __graft_entry__.build())"
    w, h, nf = 752, 480, 1200
    img = synth.textured_image(w, h, synth.BASE_SEED + 60)
    raw, out = tmp_path / "in.raw", tmp_path / "out.bin"
    raw.write_bytes(img.tobytes())
    r = subprocess.run([str(exe), str(raw), str(w), str(h), str(nf), str(out)], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stdout + r.stderr
    buf = out.read_bytes()
    n = int(np.frombuffer(buf, np.int32, 1)[0])
    kps = np.frombuffer(buf, capi.KP_DTYPE, n, 4)
    desc = np.frombuffer(buf, np.uint8, n * 32, 4 + 28 * n).reshape(n, 32)
    o = O.OracleExtractor(nf, 1.2, 8, 20, 7)
    ko, do = o.extract(img)
    assert n == len(ko) and np.array_equal(kps, ko) and np.array_equal(desc, do)
    off = 4 + 60 * n
    pw, ph = np.frombuffer(buf, np.int32, 2, off)
    lvl = np.frombuffer(buf, np.uint8, (pw + 38) * (ph + 38), off + 8).reshape(ph + 38, pw + 38)
    assert np.array_equal(lvl, np.pad(o.level(3), 19, mode="reflect"))
    d01 = int(np.frombuffer(buf, np.int32, 1, off + 8 + (pw + 38) * (ph + 38))[0])
    assert d01 == O.hamming(do[0], do[1])


def test_cpp_stereo_matcher_matches_oracle(tmp_path):
    """orbfe::StereoMatcher (include/ORBmatcher_orbfe.h) behind two adapter extractors == Frame::ComputeStereoMatches."""
    exe = CPP_BIN / "adapter_main"
    assert exe.exists()
    w, h, nf = 1241, 376, 2000
    bf, b = 386.1448, 0.5371657
    left, right = synth.stereo_pair(w, h, synth.BASE_SEED + 91)
    fl, fr, out = tmp_path / "l.raw", tmp_path / "r.raw", tmp_path / "out.bin"
    fl.write_bytes(left.tobytes())
    fr.write_bytes(right.tobytes())
    r = subprocess.run([str(exe), "stereo", str(fl), str(fr), str(w), str(h), str(nf), repr(bf), repr(b), str(out)],
                       capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stdout + r.stderr
    buf = out.read_bytes()
    n = int(np.frombuffer(buf, np.int32, 1)[0])
    ur = np.frombuffer(buf, np.float32, n, 4)
    dep = np.frombuffer(buf, np.float32, n, 4 + 4 * n)
    ol, orr = O.OracleExtractor(nf, 1.2, 8, 20, 7), O.OracleExtractor(nf, 1.2, 8, 20, 7)
    kl, dl = ol.extract(left)
    kr, dr = orr.extract(right)
    ref_ur, ref_dep = O.stereo_full(ol, orr, kl, dl, kr, dr, bf, b)
    assert n == len(kl) and (ref_dep > 0).sum() > 200
    assert np.array_equal(ur.view(np.uint32), ref_ur.view(np.uint32))
    assert np.array_equal(dep.view(np.uint32), ref_dep.view(np.uint32))


def test_cpp_vocabulary_transform_matches_oracle(tmp_path):
    """orbfe::Vocabulary (include/ORBVocabulary_orbfe.h): loadFromTextFile in the ORBvoc.txt format, then
    transform(descriptors, BowVector, FeatureVector, 4) as Frame::ComputeBoW calls it; doubles bit-exact."""
    exe = CPP_BIN / "adapter_main"
    assert exe.exists()
    k, L, levelsup = 10, 4, 2
    co, ci, nd, wi, w = synth.random_vocabulary(k, L, 4242)
    parent = np.zeros(len(nd), np.int64)
    for p in range(len(nd)):
        parent[ci[co[p]:co[p + 1]]] = p
    voc = tmp_path / "voc.txt"
    with open(voc, "w") as fh:
        fh.write(f"{k} {L} 0 0\n")                             # L1_NORM, TF_IDF: what ORBvoc.txt declares
        for i in range(1, len(nd)):
            fh.write(f"{parent[i]} {1 if wi[i] >= 0 else 0} " + " ".join(str(int(b)) for b in nd[i]) + f" {float(w[i])!r}\n")
    img = synth.textured_image(752, 480, synth.BASE_SEED + 17)
    _, desc = O.OracleExtractor(1200, 1.2, 8, 20, 7).extract(img)      # real ORB descriptors as features
    n = len(desc)
    raw, out = tmp_path / "desc.raw", tmp_path / "out.bin"
    raw.write_bytes(np.ascontiguousarray(desc).tobytes())
    r = subprocess.run([str(exe), "bow", str(voc), str(raw), str(n), str(levelsup), str(out)], capture_output=True, text=True,
                       timeout=100)
    assert r.returncode == 0, r.stdout + r.stderr
    buf = out.read_bytes()
    nb, nf = np.frombuffer(buf, np.int32, 2)
    ids = np.frombuffer(buf, np.int32, nb, 8)
    vals = np.frombuffer(buf, np.float64, nb, 8 + 4 * nb)
    fv = np.frombuffer(buf, np.int32, 2 * nf, 8 + 12 * nb).reshape(nf, 2)
    (rids, rvals), (rfn, rff) = O.bow_transform(co, ci, nd, wi, w, L, 0, 1, desc, levelsup)
    assert nb == len(rids) > 50 and np.array_equal(ids, rids)
    assert np.array_equal(vals.view(np.uint64), rvals.view(np.uint64))  # bit-exact doubles
    assert np.array_equal(fv[:, 0], rfn) and np.array_equal(fv[:, 1], rff)
    assert abs(vals.sum() - 1.0) < 1e-12                                # L1-normalised


def test_cpp_projection_search_replays_the_reference_loop(tmp_path):
    """orbfe::ProjectionSearch (include/ORBmatcher_orbfe.h) == ORBmatcher::SearchByProjection(Frame&, vpMapPoints, th)
    (src/ORBmatcher.cc:45-129) run sequentially on the oracle: same association per map point, same nmatches, including
    map points whose best keypoint was taken earlier in the same call (those are searched again on the device)."""
    exe = CPP_BIN / "adapter_main"
    assert exe.exists()
    W, H = 1241, 376
    img = synth.textured_image(W, H, synth.BASE_SEED + 321)
    o = O.OracleExtractor(2000, 1.2, 8, 20, 7)
    k, d = o.extract(img)
    n = len(k)
    rng = np.random.default_rng(19)
    kx, ky, koct = k["x"].copy(), k["y"].copy(), k["octave"].astype(np.int32)
    ur = np.where(rng.uniform(size=n) < 0.5, kx - rng.uniform(1, 60, n), -1).astype(np.float32)
    w_inv, h_inv = np.float32(64) / np.float32(W), np.float32(48) / np.float32(H)
    sf = np.array([np.float32(1.2) ** 0] * 8, np.float32)
    for i in range(1, 8):
        sf[i] = sf[i - 1] * np.float32(1.2)                      # ORBextractor.cc:417-421
    nq = 2500                                                    # more map points than free keypoints nearby: contention
    src = rng.integers(0, n, nq)
    px = (kx[src] + rng.uniform(-3, 3, nq)).astype(np.float32)
    py = (ky[src] + rng.uniform(-3, 3, nq)).astype(np.float32)
    pxr = np.where(ur[src] > 0, ur[src] + rng.uniform(-2, 2, nq), px - 20).astype(np.float32)
    lvl = np.clip(koct[src] + rng.integers(0, 2, nq), 0, 7).astype(np.int32)
    vcos = rng.choice([0.9, 0.9995], nq).astype(np.float32)
    qd = d[src].copy()
    flip = rng.integers(0, 256, (nq, 10))
    for i in range(nq):
        for b in flip[i]:
            qd[i, b >> 3] ^= np.uint8(1 << (b & 7))
    occ = (rng.uniform(size=n) < 0.1).astype(np.uint8)
    th, nnratio = np.float32(3.0), np.float32(0.8)
    case, out = tmp_path / "case.bin", tmp_path / "out.bin"
    with open(case, "wb") as fh:
        fh.write(np.array([n, nq], np.int32).tobytes())
        fh.write(np.array([0.0, 0.0, w_inv, h_inv, th, nnratio], np.float32).tobytes())
        fh.write(np.array([8], np.int32).tobytes())
        for a in (sf, kx, ky, koct, ur, d, occ, px, py, pxr, lvl, vcos, qd):
            fh.write(np.ascontiguousarray(a).tobytes())
    r = subprocess.run([str(exe), "project", str(case), str(out)], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stdout + r.stderr
    buf = out.read_bytes()
    nmatches, requeried = np.frombuffer(buf, np.int32, 2)
    match_idx = np.frombuffer(buf, np.int32, nq, 8)
    occ_out = np.frombuffer(buf, np.uint8, n, 8 + 4 * nq)
    # the reference's loop, one map point after the other, on the oracle
    og = O.OracleGrid(kx, ky, koct, d, ur, 0.0, 0.0, float(w_inv), float(h_inv))
    blocked = occ.copy()
    want = np.full(nq, -1, np.int32)
    for i in range(nq):
        rad = np.float32(2.5) if float(vcos[i]) > 0.998 else np.float32(4.0)
        rad = rad * th
        rs = float(np.float32(rad * sf[lvl[i]]))
        res = og.search(float(px[i]), float(py[i]), rs, int(lvl[i]) - 1, int(lvl[i]), float(pxr[i]), rs, qd[i], blocked)
        if res[5] == 0:
            continue
        if res[0] <= 100:
            if res[2] == res[4] and np.float32(res[0]) > nnratio * np.float32(res[3]):
                continue
            want[i] = res[1]
            blocked[res[1]] = 1
    assert np.array_equal(match_idx, want)
    assert nmatches == (want >= 0).sum() and np.array_equal(occ_out, blocked)
    assert nmatches > 500 and requeried > 10                     # contention really occurred and was replayed


@pytest.mark.parametrize("max_dist,check_ori", [(50, True), (49, True), (50, False)])
def test_cpp_bow_matcher_matches_oracle(tmp_path, max_dist, check_ori):
    """orbfe::BowMatcher::SearchByBoW (include/ORBmatcher_orbfe.h): merge walk over two DBoW2-style FeatureVectors, device
    loops, rotation histogram + ComputeThreeMaxima == the oracle's restatement of src/ORBmatcher.cc:159-288 / :522-655."""
    exe = CPP_BIN / "adapter_main"
    assert exe.exists()
    rng = np.random.default_rng(77 + max_dist)
    n1, n2 = 1800, 2000
    d2 = synth.random_descriptors(n2, 314)
    src = rng.integers(0, n2, n1)
    d1 = d2[src].copy()
    for i in range(n1):
        for b in rng.integers(0, 256, int(rng.integers(0, 40))):
            d1[i, b >> 3] ^= np.uint8(1 << (b & 7))
    # node ids: sparse and different on both sides, so the merge walk has to skip (lower_bound branches)
    node2 = rng.integers(0, 120, n2) * 7
    node1 = node2[src].copy()
    stray = rng.uniform(size=n1) < 0.15
    node1[stray] = rng.integers(0, 900, int(stray.sum()))
    ang2 = rng.uniform(0, 360, n2).astype(np.float32)
    # most true matches share one rotation (the camera rolled by ~20 degrees); the rest are spread: the histogram removes them
    ang1 = np.where(rng.uniform(size=n1) < 0.8, ang2[src] + 20 + rng.normal(0, 3, n1), rng.uniform(0, 360, n1)).astype(np.float32)
    ang1 = np.mod(ang1, np.float32(360)).astype(np.float32)
    valid1 = (rng.uniform(size=n1) < 0.9).astype(np.uint8)
    blocked2 = (rng.uniform(size=n2) < 0.05).astype(np.uint8)
    ent1 = np.stack([node1, np.arange(n1)], 1).astype(np.int32)      # features are added in index order (FeatureVector::addFeature)
    ent2 = np.stack([node2, np.arange(n2)], 1).astype(np.int32)
    case, out = tmp_path / "case.bin", tmp_path / "out.bin"
    with open(case, "wb") as fh:
        fh.write(np.array([n1, n2, n1, n2, max_dist, int(check_ori)], np.int32).tobytes())
        fh.write(np.array([0.7], np.float32).tobytes())
        for a in (d1, d2, valid1, blocked2, ang1, ang2, ent1, ent2):
            fh.write(np.ascontiguousarray(a).tobytes())
    r = subprocess.run([str(exe), "bowmatch", str(case), str(out)], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stdout + r.stderr
    buf = out.read_bytes()
    nmatches = int(np.frombuffer(buf, np.int32, 1)[0])
    match12 = np.frombuffer(buf, np.int32, n1, 4)
    # reference: std::map order = ascending node id, features of a node in insertion order; shared nodes only
    shared = sorted(set(node1.tolist()) & set(node2.tolist()))
    l1 = [np.nonzero(node1 == k)[0] for k in shared]
    l2 = [np.nonzero(node2 == k)[0] for k in shared]
    off1 = np.concatenate([[0], np.cumsum([len(x) for x in l1])]).astype(np.int32)
    off2 = np.concatenate([[0], np.cumsum([len(x) for x in l2])]).astype(np.int32)
    ref = O.search_by_bow(d1, valid1, d2, blocked2, off1, np.concatenate(l1), off2, np.concatenate(l2), max_dist, 0.7)
    before = int((ref >= 0).sum())
    if check_ori:
        ref, n_ref = O.rotation_filter(ang1, ang2, ref)
    else:
        n_ref = before
    assert np.array_equal(match12, ref) and nmatches == n_ref
    assert n_ref > 300 and (not check_ori or n_ref < before)        # the histogram really removed something


@pytest.mark.parametrize("check_ori", [True, False])
def test_cpp_track_last_frame_replays_the_reference_loop(tmp_path, check_ori):
    """orbfe::ProjectionSearch::searchLastFrame == SearchByProjection(CurrentFrame, LastFrame, th, bMono)
    (src/ORBmatcher.cc:1328-1470) from the window search on, replayed sequentially on the oracle: forward / backward / normal
    level ranges, the right-coordinate gate, unobserved (temporal) points whose keypoint can be overwritten, the histogram."""
    exe = CPP_BIN / "adapter_main"
    assert exe.exists()
    W, H = 1241, 376
    img = synth.textured_image(W, H, synth.BASE_SEED + 654)
    k, d = O.OracleExtractor(2000, 1.2, 8, 20, 7).extract(img)
    n = len(k)
    rng = np.random.default_rng(23)
    kx, ky, koct, kang = k["x"].copy(), k["y"].copy(), k["octave"].astype(np.int32), k["angle"].copy()
    kur = np.where(rng.uniform(size=n) < 0.5, kx - rng.uniform(1, 60, n), -1).astype(np.float32)
    w_inv, h_inv = np.float32(64) / np.float32(W), np.float32(48) / np.float32(H)
    sf = np.ones(8, np.float32)
    for i in range(1, 8):
        sf[i] = sf[i - 1] * np.float32(1.2)
    nq = 2200
    src = rng.integers(0, n, nq)
    pu = (kx[src] + rng.uniform(-4, 4, nq)).astype(np.float32)
    pv = (ky[src] + rng.uniform(-4, 4, nq)).astype(np.float32)
    pur = np.where(kur[src] > 0, kur[src] + rng.uniform(-3, 3, nq), pu - 25).astype(np.float32)
    last_oct = np.clip(koct[src] + rng.integers(-1, 2, nq), 0, 7).astype(np.int32)
    prad = (np.float32(15.0) * sf[last_oct]).astype(np.float32)             # th = 15 (stereo) as Tracking uses it
    mode = rng.integers(0, 3, nq)                                           # forward / backward / neither (a call has one mode;
    pmin = np.where(mode == 0, last_oct, np.where(mode == 1, 0, last_oct - 1)).astype(np.int32)   # mixing them covers all three)
    pmax = np.where(mode == 0, -1, np.where(mode == 1, last_oct, last_oct + 1)).astype(np.int32)
    # the scene rolled by ~12 degrees between the frames; some points have unrelated angles
    pang = np.mod(np.where(rng.uniform(size=nq) < 0.85, kang[src] + 12 + rng.normal(0, 2, nq), rng.uniform(0, 360, nq)),
                  360).astype(np.float32)
    pobs = (rng.uniform(size=nq) < 0.7).astype(np.uint8)
    qd = d[src].copy()
    for i in range(nq):
        for b in rng.integers(0, 256, 10):
            qd[i, b >> 3] ^= np.uint8(1 << (b & 7))
    occ = (rng.uniform(size=n) < 0.1).astype(np.uint8)
    case, out = tmp_path / "case.bin", tmp_path / "out.bin"
    with open(case, "wb") as fh:
        fh.write(np.array([n, nq, int(check_ori)], np.int32).tobytes())
        fh.write(np.array([0.0, 0.0, w_inv, h_inv], np.float32).tobytes())
        for a in (kx, ky, koct, kur, kang, d, occ, pu, pv, pur, prad, pmin, pmax, pang, pobs, qd):
            fh.write(np.ascontiguousarray(a).tobytes())
    r = subprocess.run([str(exe), "track", str(case), str(out)], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stdout + r.stderr
    buf = out.read_bytes()
    nmatches, requeried = np.frombuffer(buf, np.int32, 2)
    assigned = np.frombuffer(buf, np.int32, n, 8)
    occ_out = np.frombuffer(buf, np.uint8, n, 8 + 4 * n)
    # the reference's loop on the oracle
    og = O.OracleGrid(kx, ky, koct, d, kur, 0.0, 0.0, float(w_inv), float(h_inv))
    blocked = occ.copy()
    want = np.full(n, -1, np.int32)
    hist = [[] for _ in range(30)]
    nm = 0
    overwritten = 0
    for i in range(nq):
        res = og.search(float(pu[i]), float(pv[i]), float(prad[i]), int(pmin[i]), int(pmax[i]), float(pur[i]), float(prad[i]), qd[i], blocked)
        if res[5] == 0 or res[0] > 100:
            continue
        b = int(res[1])
        overwritten += want[b] >= 0
        want[b] = i
        nm += 1
        if pobs[i]:
            blocked[b] = 1
        if check_ori:
            rot = np.float32(pang[i]) - np.float32(kang[b])
            if rot < 0:
                rot = np.float32(rot + np.float32(360))
            v = float(np.float32(rot * np.float32(1.0 / 30)))
            bn = int(np.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)        # C round(): half away from zero
            hist[0 if bn == 30 else bn].append(b)
    if check_ori:
        keep = set(O.three_maxima([len(h) for h in hist]))
        for bn in range(30):
            if bn not in keep:
                for b in hist[bn]:
                    want[b] = -1
                    blocked[b] = 0
                    nm -= 1
    assert np.array_equal(assigned, want) and nmatches == nm
    assert np.array_equal(occ_out, blocked)
    assert nm > 800 and requeried > 10 and overwritten > 0
