"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/orbfe.h declares, and its host-only
entry points (orbfe_hamming, orbfe_merge_knn2) agree with the oracle.  No compute call needs a GPU here."""
import re
from pathlib import Path

import numpy as np

import oracle_lib as O
from orb_slam2_amd import capi

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    hdr = (ROOT / "include" / "orbfe.h").read_text()
    declared = set(re.findall(r"\b(orbfe_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    L = capi.lib()
    for name in declared:
        assert hasattr(L, name), name


def test_keypoint_layout_is_cv_keypoint():
    assert capi.KP_DTYPE.itemsize == 28
    assert [capi.KP_DTYPE.fields[n][1] for n in ("x", "y", "size", "angle", "response", "octave", "class_id")] == \
        [0, 4, 8, 12, 16, 20, 24]


def test_hamming_host_matches_oracle():
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (300, 32)).astype(np.uint8)
    b = rng.integers(0, 256, (300, 32)).astype(np.uint8)
    for i in range(300):
        assert capi.ORBmatcher.DescriptorDistance(a[i], b[i]) == O.hamming(a[i], b[i])
    z = np.zeros(32, np.uint8)
    assert capi.ORBmatcher.DescriptorDistance(z, np.full(32, 255, np.uint8)) == 256


def test_merge_knn2_equals_unsharded_scan():
    from orb_slam2_amd import synth
    db = synth.random_descriptors(4000, 5)
    db[1000] = db[10]          # duplicate rows in different shards: lowest global index must win
    db[3000] = db[10]
    q = synth.planted_queries(db, 64, 9, planted_frac=0.5)
    q[0] = db[10]
    ref = O.knn2(q, db)
    for shards in (1, 2, 3, 8):
        bounds = np.linspace(0, len(db), shards + 1).astype(int)
        parts = []
        for s in range(shards):
            bd, bi, sd = O.knn2(q, db[bounds[s]:bounds[s + 1]])
            parts.append((bd, np.where(bi >= 0, bi + bounds[s], -1), sd))
        got = capi.merge_knn2(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]),
                              np.stack([p[2] for p in parts]))
        for g, r in zip(got, ref):
            assert np.array_equal(g, r)


def test_no_gpu_means_loud_failure():
    if capi.lib().orbfe_device_count() > 0:
        return
    try:
        capi.ORBextractor()
    except capi.OrbfeError as e:
        assert e.code == capi.ERR_DEVICE
    else:
        raise AssertionError("extractor creation must fail without a GPU (no CPU fallback)")
