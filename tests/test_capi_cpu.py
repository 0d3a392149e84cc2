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


def test_missing_rccl_is_an_error_code_not_a_crash():
    """orbfe_knn2_sharded_* with a librccl that cannot be loaded (ORBFE_RCCL_LIB forces the name): ORBFE_ERR_DEVICE and the
    loader's message from every entry point -- the path the bench's fallback and extraction-only installs depend on.  Own
    process: the binding is resolved once per process."""
    import subprocess
    import sys
    code = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
from orb_slam2_amd import capi
L = capi.lib()
assert capi.sharded_available() is False
msg = L.orbfe_last_error().decode()
assert "no_such_rccl" in msg, msg
buf = (C.c_uint8 * 128)()
assert L.orbfe_knn2_sharded_unique_id(buf) == capi.ERR_DEVICE
assert "no_such_rccl" in L.orbfe_last_error().decode()
h = C.c_void_p()
# a non-null matcher pointer that is never dereferenced: the binding is checked before the matcher is touched
assert L.orbfe_knn2_sharded_create(C.c_void_p(8), buf, 0, 1, 16, C.byref(h)) == capi.ERR_DEVICE
assert not h.value
print("ok")
""" % str(ROOT)
    import os
    env = dict(os.environ, ORBFE_RCCL_LIB="/nonexistent/libno_such_rccl.so")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_adapters_parse_against_the_reference_class_declarations():
    """tools/refcheck: -fsyntax-only of adapters/ORBmatcher_orbfe.cc and adapters/ORBextractor_orbfe.cc with the reference's
    own include/Frame.h, KeyFrame.h, MapPoint.h (+ Map.h, KeyFrameDatabase.h, ORBVocabulary.h, Thirdparty/DBoW2) read in place;
    only <opencv2/...> is answered by a shim (no OpenCV in this image).  Where /root/reference is absent (the GPU box) there
    is nothing to check against."""
    import subprocess
    import pytest
    if not Path("/root/reference/include/Frame.h").exists():
        pytest.skip("the reference tree is not on this machine")
    r = subprocess.run(["make", "-C", str(ROOT / "tools" / "refcheck")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("refcheck:") == 3 and "nothing checked" not in r.stdout, r.stdout
