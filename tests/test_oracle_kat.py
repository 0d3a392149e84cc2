"""Known-answer tests that pin the ORACLE (oracle/orb_oracle.cpp).

The reference ships no tests or golden vectors, so these are the hand-derivable
known answers of SURVEY.md section 8c plus independent numpy restatements of the OpenCV primitives
(written from the spec in SURVEY.md Appendix A, not from the oracle's C++).
"""
import hashlib
import math
import os

import numpy as np
import pytest

import oracle_lib as O
from orb_slam2_amd import synth

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
        (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


# ---------------------------------------------------------------- (1) constructor constants
def test_scale_tables_and_quotas():
    t = O.OracleExtractor(2000, 1.2, 8, 20, 7).tables()
    exp = [1, 1.2000000477, 1.4400000572, 1.7280001640, 2.0736002922, 2.4883203506, 2.9859845638, 3.5831816196]
    assert np.array_equal(t["scale"], np.array(exp, np.float32))
    assert list(t["features_per_level"]) == [434, 362, 302, 251, 209, 175, 145, 122]
    assert list(t["umax"]) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert sum(2 * u + 1 for u in t["umax"][1:]) * 2 + 31 == 749
    assert list(O.OracleExtractor(1000, 1.2, 8, 20, 7).tables()["features_per_level"]) == [217, 181, 151, 126, 105, 87, 73, 60]
    assert list(O.OracleExtractor(1200, 1.2, 8, 20, 7).tables()["features_per_level"]) == [261, 217, 181, 151, 126, 105, 87, 72]


@pytest.mark.parametrize("wh,sizes", [
    ((1241, 376), [(1034, 313), (862, 261), (718, 218), (598, 181), (499, 151), (416, 126), (346, 105)]),
    ((640, 480), [(533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]),
    ((752, 480), [(627, 400), (522, 333), (435, 278), (363, 231), (302, 193), (252, 161), (210, 134)]),
])
def test_level_sizes(wh, sizes):
    e = O.OracleExtractor(500, 1.2, 8, 20, 7)
    e.extract(np.zeros((wh[1], wh[0]), np.uint8))
    got = [e.level(l).shape[::-1] for l in range(1, 8)]
    assert got == sizes


def test_cv_round_half_even():
    L = O.lib()
    assert [L.orc_cv_round(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]


# ---------------------------------------------------------------- (2) fastAtan2
def test_fast_atan2_axes():
    assert O.fast_atan2(0, 0) == 0.0
    assert O.fast_atan2(0, 1) == 0.0
    assert O.fast_atan2(1, 0) == 90.0
    assert O.fast_atan2(0, -1) == 180.0
    assert O.fast_atan2(-1, 0) == 270.0
    for ang in np.linspace(0.5, 359.5, 97):
        a = O.fast_atan2(math.sin(math.radians(ang)) * 1000, math.cos(math.radians(ang)) * 1000)
        assert abs(a - ang) < 0.3  # polynomial error bound of cv::fastAtan2
    assert abs(O.fast_atan2(5, 5) - 45) < 0.3 and abs(O.fast_atan2(-5, 5) - 315) < 0.3


def test_fast_atan2_numpy_float32():
    """Same polynomial evaluated with numpy float32 scalar ops (no FMA) must agree bit for bit."""
    f = np.float32
    k = f(180 / math.pi)
    p1, p3, p5, p7 = (f(0.9997878412794807) * k, f(-0.3258083974640975) * k, f(0.1555786518463281) * k,
                      f(-0.04432655554792128) * k)
    eps = f(2.220446049250313e-16)
    rng = np.random.default_rng(7)
    for _ in range(2000):
        y, x = f(rng.integers(-2800000, 2800000)), f(rng.integers(-2800000, 2800000))
        ax, ay = abs(x), abs(y)
        if ax >= ay:
            c = ay / (ax + eps)
            c2 = c * c
            a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
        else:
            c = ax / (ay + eps)
            c2 = c * c
            a = f(90) - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
        if x < 0:
            a = f(180) - a
        if y < 0:
            a = f(360) - a
        assert f(O.fast_atan2(y, x)) == a


# ---------------------------------------------------------------- (3) FAST
def _blank(v=100, n=15):
    return np.full((n, n), v, np.uint8)


def _arc(img, cx, cy, start, length, val):
    for k in range(start, start + length):
        dx, dy = RING[k % 16]
        img[cy + dy, cx + dx] = val


def np_fast_score(img):
    """Closed form of SURVEY 8a-3: s(p) = max over the 16 nine-arcs of max(min(v - ring), min(ring - v))."""
    h, w = img.shape
    im = img.astype(np.int32)
    S = np.zeros((h, w), np.int32)
    v = im[3:h - 3, 3:w - 3]
    ring = np.stack([im[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in RING])  # [16, h-6, w-6]
    d = v[None] - ring
    best = np.full(v.shape, -10**6, np.int32)
    for a in range(16):
        idx = [(a + i) % 16 for i in range(9)]
        best = np.maximum(best, np.maximum(d[idx].min(0), (-d[idx]).min(0)))
    S[3:h - 3, 3:w - 3] = best
    return S


def np_fast9(img, t):
    """FAST-9/16 + strict 3x3 NMS from the closed form; emits (x, y, s-1) in row-major order."""
    h, w = img.shape
    if h < 7 or w < 7:
        return np.zeros((0, 3), np.float32)
    S = np_fast_score(img)
    sc = np.where(S > t, S - 1, 0)
    sc[:3] = 0; sc[h - 3:] = 0; sc[:, :3] = 0; sc[:, w - 3:] = 0
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = sc[y, x]
            if s <= 0 or S[y, x] <= t:
                continue
            nb = sc[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if (s > nb).all():
                out.append((x, y, s))
    return np.array(out, np.float32).reshape(-1, 3)


def test_fast_bright_dot():
    img = _blank(100)
    img[7, 7] = 180  # all 16 ring pixels darker by 80 -> score 80, response 79
    r = O.fast9(img, 20)
    assert r.tolist() == [[7.0, 7.0, 79.0]]


def test_fast_9arc_fires_8arc_does_not():
    for start in range(16):
        img = _blank(100)
        _arc(img, 7, 7, start, 9, 160)
        r = O.fast9(img, 20)
        assert [7.0, 7.0, 59.0] in r.tolist(), (start, r)
        img = _blank(100)
        _arc(img, 7, 7, start, 8, 160)
        r = O.fast9(img, 20)
        assert not any(p[0] == 7 and p[1] == 7 for p in r.tolist()), (start, r)


def test_fast_threshold_edge():
    img = _blank(100)
    _arc(img, 7, 7, 3, 9, 121)  # ring - v = 21 > 20 -> corner, response 20
    assert [7.0, 7.0, 20.0] in O.fast9(img, 20).tolist()
    img = _blank(100)
    _arc(img, 7, 7, 3, 9, 120)  # == threshold: strict compare fails
    assert not any(p[0] == 7 and p[1] == 7 for p in O.fast9(img, 20).tolist())
    assert [7.0, 7.0, 19.0] in O.fast9(img, 7).tolist()


def test_fast_matches_closed_form_random():
    rng = np.random.default_rng(3)
    for trial in range(30):
        h, w = rng.integers(7, 48), rng.integers(7, 48)
        base = rng.integers(0, 256, (h, w)).astype(np.uint8)
        if trial % 3 == 0:  # smoother texture: plateaus and near-threshold contrasts
            base = (base // 32 * 32).astype(np.uint8)
        elif trial % 3 == 1:
            base = np.clip(100 + (base.astype(int) - 128) // 6, 0, 255).astype(np.uint8)
        for t in (7, 20):
            a, b = O.fast9(base, t), np_fast9(base, t)
            assert a.shape == b.shape and np.array_equal(a, b), (trial, t)


def test_fast_plateau_suppresses_itself():
    """Two adjacent equal-score corners: strict '>' NMS removes both (SURVEY 8a-3 note on 'empty' cells)."""
    img = synth.textured_image(64, 64, 99)
    r20 = O.fast9(img, 20)
    S = np_fast_score(img)
    n_corners = int((S[3:-3, 3:-3] > 20).sum())
    assert n_corners >= len(r20)


def test_fast_cells_fallback_and_border_nms():
    # image where one cell region has only low-contrast (10) corners -> that cell must use minTh=7
    img = np.full((120, 200), 100, np.uint8)
    img[30, 150] = 200                   # strong dot far right (a corner at th=20)
    img[40, 30] = 112                    # weak dot on the left (only visible at th=7)
    c = O.fast_cells(img, 20, 7)
    xs = c[:, 0] + 16
    assert (xs < 60).any() and (xs > 140).any()
    weak = c[xs < 60]
    assert (weak[:, 2] < 20).all()
    # with minTh == iniTh the weak blob disappears
    c2 = O.fast_cells(img, 20, 20)
    assert not ((c2[:, 0] + 16) < 60).any()


def np_fast_cells(img, ini_th, min_th):
    """Cell loop of src/ORBextractor.cc:771-829 on top of np_fast9 (independent restatement)."""
    h, w = img.shape
    minB, maxBX, maxBY = 16, w - 16, h - 16
    width, height = np.float32(maxBX - minB), np.float32(maxBY - minB)
    nCols, nRows = int(width / np.float32(30)), int(height / np.float32(30))
    wCell, hCell = int(math.ceil(width / nCols)), int(math.ceil(height / nRows))
    out = []
    for i in range(nRows):
        iniY = minB + i * hCell
        maxY = min(iniY + hCell + 6, maxBY)
        if iniY >= maxBY - 3:
            continue
        for j in range(nCols):
            iniX = minB + j * wCell
            maxX = min(iniX + wCell + 6, maxBX)
            if iniX >= maxBX - 6:
                continue
            win = img[iniY:maxY, iniX:maxX]
            r = np_fast9(win, ini_th)
            if len(r) == 0:
                r = np_fast9(win, min_th)
            for x, y, s in r:
                out.append((x + j * wCell, y + i * hCell, s))
    return np.array(out, np.float32).reshape(-1, 3)


def test_fast_cells_vs_numpy():
    img = synth.textured_image(190, 130, 5)
    a, b = O.fast_cells(img, 20, 7), np_fast_cells(img, 20, 7)
    assert len(a) > 50 and np.array_equal(a, b)


# ---------------------------------------------------------------- (4) resize
def np_resize(src, dw, dh):
    """cv::resize INTER_LINEAR 8UC1, OpenCV 2.4/3.2 generic path, numpy restatement."""
    sh, sw = src.shape

    def tables(ssize, dsize, clamp):
        scale = np.float64(1.0) / (np.float64(dsize) / np.float64(ssize))
        d = np.arange(dsize, dtype=np.float64)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = f - s.astype(np.float32)
        if clamp:
            f = np.where(s < 0, np.float32(0), f); s = np.maximum(s, 0)
            f = np.where(s >= ssize - 1, np.float32(0), f); s = np.minimum(s, ssize - 1)
        c0 = np.clip(np.rint((np.float32(1) - f) * np.float32(2048)), -32768, 32767).astype(np.int64)
        c1 = np.clip(np.rint(f * np.float32(2048)), -32768, 32767).astype(np.int64)
        return s, c0, c1

    sx, a0, a1 = tables(sw, dw, True)
    sy, b0, b1 = tables(sh, dh, False)
    S = src.astype(np.int64)
    sx1 = np.minimum(sx + 1, sw - 1)
    H = S[:, sx] * a0[None] + np.where((sx + 1 < sw)[None], S[:, sx1] * a1[None], 0)
    H = np.where((sx + 1 < sw)[None], H, S[:, sx] * 2048)
    r0 = np.clip(sy, 0, sh - 1)
    r1 = np.clip(sy + 1, 0, sh - 1)
    D = (((b0[:, None] * (H[r0] >> 4)) >> 16) + ((b1[:, None] * (H[r1] >> 4)) >> 16) + 2) >> 2
    return D.astype(np.uint8)


def test_resize_constant_and_numpy():
    const = np.full((100, 240), 137, np.uint8)
    assert (O.resize(const, 200, 83) == 137).all()
    img = synth.textured_image(1241, 376, 11)
    cur = img
    crcs = []
    for (dw, dh) in [(1034, 313), (862, 261), (718, 218), (598, 181), (499, 151), (416, 126), (346, 105)]:
        a, b = O.resize(cur, dw, dh), np_resize(cur, dw, dh)
        assert np.array_equal(a, b)
        cur = a
        crcs.append(hashlib.sha256(a.tobytes()).hexdigest()[:8])
    assert len(set(crcs)) == 7


def test_resize_tables_known():
    ofs, coef = O.resize_tables(1241, 1034)
    # first output pixel: fx = 0.5*1241/1034 - 0.5 = 0.10009..., a1 = round(0.1000967*2048) = 205
    assert ofs[0] == 0 and coef[0].tolist() == [1843, 205]
    assert (coef.sum(1) >= 2047).all() and (coef.sum(1) <= 2049).all()
    assert ofs.max() <= 1240 and (np.diff(ofs) >= 1).all()


def test_resize_horizontal_ramp_monotone():
    ramp = np.tile(np.arange(240, dtype=np.uint8), (50, 1))
    out = O.resize(ramp, 200, 42)
    assert (np.diff(out.astype(int), axis=1) >= 0).all()
    assert np.abs(out.astype(int) - out[0].astype(int)).max() <= 1   # per-term truncation in the vertical pass


# ---------------------------------------------------------------- (5) blur
def np_gauss7(src):
    k = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
    h, w = src.shape

    def refl(p, n):
        p = np.where(p < 0, -p, p)
        return np.where(p >= n, 2 * (n - 1) - p, p)

    S = src.astype(np.int64)
    R = sum(k[i] * S[:, refl(np.arange(w) + i - 3, w)] for i in range(7))
    D = sum(k[j] * R[refl(np.arange(h) + j - 3, h)] for j in range(7))
    return np.clip((D + 32768) >> 16, 0, 255).astype(np.uint8)


def test_gauss_kernel_ints():
    assert O.gauss7_kernel().tolist() == [18, 34, 49, 55, 49, 34, 18]


def test_gauss_impulse_and_saturation():
    k = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
    img = np.zeros((21, 21), np.uint8)
    img[10, 10] = 255
    out = O.gauss7(img)
    exp = ((np.outer(k, k) * 255 + 32768) >> 16).astype(np.uint8)
    assert np.array_equal(out[7:14, 7:14], exp) and out.sum() == exp.sum()
    assert (O.gauss7(np.full((30, 40), 255, np.uint8)) == 255).all()   # 257*257*255 > 65535*255 -> saturates
    assert (O.gauss7(np.full((30, 40), 77, np.uint8)) == ((77 * 257 * 257 + 32768) >> 16)).all()


def test_gauss_vs_numpy_with_borders():
    for seed, (w, h) in enumerate([(346, 105), (64, 20), (9, 7), (533, 400)]):
        img = synth.textured_image(w, h, 40 + seed)
        assert np.array_equal(O.gauss7(img), np_gauss7(img))


# ---------------------------------------------------------------- (6) Hamming
def test_hamming_kat():
    z = np.zeros(32, np.uint8)
    o = np.full(32, 255, np.uint8)
    assert O.hamming(z, z) == 0 and O.hamming(z, o) == 256 and O.hamming(o, o) == 0
    for bit in (0, 7, 8, 100, 255):
        v = z.copy()
        v[bit >> 3] |= 1 << (bit & 7)
        assert O.hamming(z, v) == 1 and O.hamming(o, v) == 255
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (200, 32)).astype(np.uint8)
    b = rng.integers(0, 256, (200, 32)).astype(np.uint8)
    ref = np.unpackbits(a ^ b, axis=1).sum(1)
    assert [O.hamming(a[i], b[i]) for i in range(200)] == ref.tolist()


def test_knn2_update_rule():
    q = np.zeros((1, 32), np.uint8)
    r = np.zeros((4, 32), np.uint8)
    r[0, 0] = 0b111      # d=3
    r[1, 0] = 0b1        # d=1
    r[2, 0] = 0b10       # d=1 (tie: first minimum wins, second == best)
    r[3, 0] = 0b11       # d=2
    bd, bi, sd = O.knn2(q, r)
    assert (bd[0], bi[0], sd[0]) == (1, 1, 1)
    bd, bi, sd = O.knn2(q, r[:1])
    assert (bd[0], bi[0], sd[0]) == (3, 0, 256)


# ---------------------------------------------------------------- (7) quadtree
def test_octree_four_quadrants():
    pts = np.array([[10, 10, 5], [90, 10, 6], [10, 90, 7], [90, 90, 8]], np.float32)
    out = O.distribute_octree(pts, 0, 100, 0, 100, 4)
    # one root (100x100), one split; children pushed to the front in order n1..n4 -> list order n4,n3,n2,n1
    assert out.tolist() == [[90, 90, 8], [10, 90, 7], [90, 10, 6], [10, 10, 5]]


def test_octree_best_response_first_wins():
    pts = np.array([[10, 10, 5], [12, 12, 9], [14, 14, 9], [90, 90, 1]], np.float32)
    out = O.distribute_octree(pts, 0, 100, 0, 100, 2)
    assert out.tolist() == [[90, 90, 1], [12, 12, 9]]


def test_octree_count_bounds():
    img = synth.textured_image(1241, 376, 21)
    c = O.fast_cells(img)
    for N in (50, 434, 2000):
        out = O.distribute_octree(c, 16, 1241 - 16, 16, 376 - 16, N)
        assert N <= len(out) <= N + 3 or len(out) == len(c)
        keys = {(x, y) for x, y, _ in out.tolist()}
        assert len(keys) == len(out) and keys <= {(x, y) for x, y, _ in c.tolist()}


# ---------------------------------------------------------------- (8) orientation / descriptor
def test_ic_angle_gradient_images():
    um = O.OracleExtractor().tables()["umax"]
    xs = np.arange(64)
    right = np.tile((xs * 3).astype(np.uint8), (64, 1))       # brighter to the right -> centroid at +x -> 0 deg
    down = np.ascontiguousarray(right.T)                      # brighter downwards -> +y -> 90 deg
    assert O.ic_angle(right, 32, 32, um) == 0.0
    assert O.ic_angle(down, 32, 32, um) == 90.0
    assert O.ic_angle(np.ascontiguousarray(right[:, ::-1]), 32, 32, um) == 180.0
    assert O.ic_angle(np.ascontiguousarray(down[::-1]), 32, 32, um) == 270.0


def test_descriptor_angle0_direct_taps():
    """At angle 0, a=1,b=0: tap (px,py) reads center[py][px]; compare with a direct numpy evaluation."""
    pat = np.loadtxt((O.ROOT / "oracle" / "pattern31.inc").open(), delimiter=",", comments="//",
                     usecols=range(16), dtype=np.int64).reshape(256, 4)
    img = synth.textured_image(80, 80, 77)
    d = O.descriptor(img, 40, 40, 0.0)
    bits = [(int(img[40 + y0, 40 + x0]) < int(img[40 + y1, 40 + x1])) for x0, y0, x1, y1 in pat]
    exp = np.packbits(np.array(bits, np.uint8), bitorder="little")
    assert np.array_equal(d, exp)
    # 90 degrees: a = cos(pi/2 as float) ~ -4.4e-8, b = 1 -> row = px, col = -py (rounding absorbs the tiny a)
    d90 = O.descriptor(img, 40, 40, 90.0)
    bits = [(int(img[40 + x0, 40 - y0]) < int(img[40 + x1, 40 - y1])) for x0, y0, x1, y1 in pat]
    assert np.array_equal(d90, np.packbits(np.array(bits, np.uint8), bitorder="little"))


def test_pattern_table_pinned():
    a = (O.ROOT / "oracle" / "pattern31.inc").read_bytes()
    b = (O.ROOT / "orb_slam2_amd" / "csrc" / "pattern31.inc").read_bytes()
    assert a == b
    vals = np.loadtxt((O.ROOT / "oracle" / "pattern31.inc").open(), delimiter=",", comments="//", usecols=range(16),
                      dtype=np.int64).reshape(-1)
    assert len(vals) == 1024 and vals.min() >= -13 and vals.max() <= 13
    assert hashlib.sha256(vals.astype(np.int8).tobytes()).hexdigest() == \
        "2164181aea6ff9ac426ca512d5130d15e1f6e3cd47b1cbdd568bbe1e55d49023"
    assert vals[:8].tolist() == [8, -3, 9, 5, 4, 2, 7, -12]   # first two pairs as printed in the ORB paper's table


# ---------------------------------------------------------------- (9) end to end
def test_extract_end_to_end_invariants(kitti_frame):
    e = O.OracleExtractor(2000, 1.2, 8, 20, 7)
    kps, desc = e.extract(kitti_frame)
    assert 2000 <= len(kps) <= 2000 + 3 * 8
    assert desc.shape == (len(kps), 32)
    t = e.tables()
    assert (np.diff(kps["octave"]) >= 0).all()
    for l in range(8):
        m = kps["octave"] == l
        lw, lh = e.level(l).shape[::-1]
        x, y = kps["x"][m] / t["scale"][l], kps["y"][m] / t["scale"][l]
        assert (x > 18.9).all() and (x < lw - 19.9).all() and (y > 18.9).all() and (y < lh - 19.9).all()
        assert (kps["size"][m] == float(int(31 * t["scale"][l]))).all()
    assert ((kps["angle"] >= 0) & (kps["angle"] < 360)).all() and (kps["class_id"] == -1).all()
    # re-running is deterministic
    k2, d2 = e.extract(kitti_frame)
    assert np.array_equal(kps, k2) and np.array_equal(desc, d2)
    # empty image: outputs untouched / zero keypoints
    k0, d0 = O.OracleExtractor().extract(np.zeros((376, 1241), np.uint8))
    assert len(k0) == 0


def _stereo_full_numpy(ol, orr, kl, dl, kr, dr, bf, b):
    """Independent restatement of Frame::ComputeStereoMatches after the row-band search (src/Frame.cc:551-639)."""
    f32 = np.float32
    nl = len(kl)
    ur, dep = np.full(nl, -1, f32), np.full(nl, -1, f32)
    sf, isf = ol.tables()["scale"], ol.tables()["inv_scale"]
    maxD = f32(bf) / f32(b)
    bd, bi = O.stereo_rowband(kl, dl, kr, dr, sf, ol.level(0).shape[0], 0.0, float(maxD))
    LV = [ol.level(l).astype(np.int64) for l in range(ol.nlevels)]
    RV = [orr.level(l).astype(np.int64) for l in range(ol.nlevels)]
    acc = []
    rnd = lambda v: f32(np.floor(abs(float(v)) + 0.5) * (1 if v >= 0 else -1))   # C round(): halves away from zero
    for iL in range(nl):
        if not bd[iL] < 75:
            continue
        o = int(kl["octave"][iL])
        uL, uR0 = f32(kl["x"][iL]), f32(kr["x"][bi[iL]])
        suL, svL, suR0 = rnd(uL * isf[o]), rnd(f32(kl["y"][iL]) * isf[o]), rnd(uR0 * isf[o])
        PL, PR = LV[o], RV[o]
        if suR0 < 0 or suR0 + 11 >= PR.shape[1]:
            continue
        cy, cxl, cxr0 = int(svL), int(suL), int(suR0)
        IL = PL[cy - 5:cy + 6, cxl - 5:cxl + 6] - PL[cy, cxl]
        dists = []
        for inc in range(-5, 6):
            cx = cxr0 + inc
            IR = PR[cy - 5:cy + 6, cx - 5:cx + 6] - PR[cy, cx]
            dists.append(f32(np.abs(IL - IR).sum()))
        best = int(np.argmin(dists))               # first minimum: strict '<' in ascending incR
        if best == 0 or best == 10:
            continue
        d1, d2, d3 = dists[best - 1], dists[best], dists[best + 1]
        with np.errstate(divide="ignore", invalid="ignore"):
            delta = f32(d1 - d3) / f32(f32(2) * f32(f32(d1 + d3) - f32(f32(2) * d2)))
        if delta < -1 or delta > 1:
            continue
        buR = f32(sf[o] * f32(f32(suR0 + f32(best - 5)) + delta))
        disp = f32(uL - buR)
        if disp >= 0 and disp < maxD:
            if disp <= 0:
                disp = f32(0.01)
                buR = f32(float(uL) - 0.01)
            dep[iL] = f32(bf) / disp
            ur[iL] = buR
            acc.append((int(dists[best]), iL))
    if acc:
        acc.sort()
        th = f32(f32(1.5) * f32(1.4)) * f32(acc[len(acc) // 2][0])
        for d, i in acc:
            if not f32(d) < th:
                ur[i] = dep[i] = -1
    return ur, dep, len(acc)


def test_stereo_full_vs_numpy_and_geometry():
    W, H = 752, 480
    left, right = synth.stereo_pair(W, H, synth.BASE_SEED + 31)
    ol, orr = O.OracleExtractor(1000, 1.2, 8, 20, 7), O.OracleExtractor(1000, 1.2, 8, 20, 7)
    kl, dl = ol.extract(left)
    kr, dr = orr.extract(right)
    bf, b = 47.90639384423901, 0.110074            # EuRoC.yaml: Camera.bf, baseline
    ur, dep = O.stereo_full(ol, orr, kl, dl, kr, dr, bf, b)
    nur, ndep, accepted = _stereo_full_numpy(ol, orr, kl, dl, kr, dr, bf, b)
    assert np.array_equal(ur.view(np.uint32), nur.view(np.uint32))
    assert np.array_equal(dep.view(np.uint32), ndep.view(np.uint32))
    ok = dep > 0
    assert ok.sum() > 100 and accepted > ok.sum()       # matches exist, and the median gate removed some of them
    # geometry: depth = bf / disparity, right coordinate left of the left one, disparity inside the synthetic range [2, 60] +- 1
    disp = kl["x"][ok] - ur[ok]
    assert np.allclose(dep[ok], np.float32(bf) / disp, rtol=1e-6)
    assert (disp > 0).all() and (disp < 62 * 1.2 ** 7).all()
    assert np.all((ur == -1) == (dep == -1))


# ---- DBoW2 transform (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1259) ----------------------------------------
def _tiny_vocab():
    """root -> A (all 0x00), B (all 0xFF); A -> A0 (0x00), A1 (0x0F);  B -> B0 (0xFF), B1 (0xFF: duplicate), B2 (0xF0)."""
    desc = np.zeros((8, 32), np.uint8)
    desc[2], desc[5], desc[6] = 0xFF, 0xFF, 0xFF
    desc[4], desc[7] = 0x0F, 0xF0
    child_off = np.array([0, 2, 4, 7, 7, 7, 7, 7, 7], np.int32)
    child_idx = np.array([1, 2, 3, 4, 5, 6, 7], np.int32)
    word_id = np.array([-1, -1, -1, 0, 1, 2, 3, 4], np.int32)
    weight = np.array([0, 0, 0, 1.0, 2.0, 4.0, 8.0, 0.0])       # word 4 (node 7) is stopped
    return child_off, child_idx, desc, word_id, weight


def test_bow_descend_hand_tree():
    co, ci, nd, wi, w = _tiny_vocab()
    f = np.zeros((5, 32), np.uint8)
    f[1] = 0x0F          # nearer to A (128) than to B (128)?  equal: 128 vs 128 -> the first child (A) wins, then A1 exactly
    f[2] = 0xFF          # B, then B0 and B1 tie at 0: the first (node 5) wins
    f[3] = 0xF0          # tie at the root -> A; inside A: A0 = 128, A1 = 256 -> A0
    f[4] = 0xFE          # B (32 vs 224); B0/B1 at 32, B2 at 96 -> node 5
    leaf, nid = O.bow_descend(co, ci, nd, 2, f, 1)
    assert leaf.tolist() == [3, 4, 5, 3, 5]
    assert nid.tolist() == [1, 1, 2, 1, 2]                      # level L - levelsup = 1
    _, nid0 = O.bow_descend(co, ci, nd, 2, f, 0)
    assert nid0.tolist() == leaf.tolist()                       # levelsup 0: the leaf itself
    _, nid2 = O.bow_descend(co, ci, nd, 2, f, 2)
    assert nid2.tolist() == [0] * 5                             # level <= 0: the root (:1227)


def test_bow_transform_weights_and_feature_vector():
    co, ci, nd, wi, w = _tiny_vocab()
    f = np.zeros((6, 32), np.uint8)
    f[1], f[2], f[3], f[4], f[5] = 0x0F, 0xFF, 0xFF, 0xF0 | 0x00, 0xF1
    f[5] = np.uint8(0xF0)                                       # descends root tie -> A -> A0
    f[4, :] = 0xFF
    f[4, :20] = 0xF0                                            # B side, nearest B2 (stopped word)
    leaf, _ = O.bow_descend(co, ci, nd, 2, f, 1)
    assert leaf.tolist() == [3, 4, 5, 5, 7, 3]
    # TF_IDF + L1: word 0 twice (1+1), word 1 once (2), word 2 twice (4+4); word 4 stopped -> dropped from both vectors
    (ids, vals), (fn, ff) = O.bow_transform(co, ci, nd, wi, w, 2, 0, 1, f, 1)
    assert ids.tolist() == [0, 1, 2]
    assert np.array_equal(vals, np.array([2.0, 2.0, 8.0]) / 12.0)
    assert fn.tolist() == [1, 1, 1, 2, 2] and ff.tolist() == [0, 1, 5, 2, 3]
    # TF without normalisation divides by the number of words (:1160-1166); BINARY keeps the first weight
    (_, v_tf), _ = O.bow_transform(co, ci, nd, wi, w, 2, 1, 0, f, 1)
    assert np.array_equal(v_tf, np.array([2.0, 2.0, 8.0]) / 3.0)
    (_, v_bin), _ = O.bow_transform(co, ci, nd, wi, w, 2, 3, 0, f, 1)
    assert np.array_equal(v_bin, np.array([1.0, 2.0, 4.0]))
    (_, v_l2), _ = O.bow_transform(co, ci, nd, wi, w, 2, 0, 2, f, 1)
    assert np.allclose(v_l2, np.array([2.0, 2.0, 8.0]) / np.sqrt(72.0), rtol=1e-15)


def test_bow_descend_vs_numpy_random_tree():
    co, ci, nd, wi, w = synth.random_vocabulary(5, 4, 123, irregular=True)
    f = synth.random_descriptors(300, 77)
    f[:100] = nd[np.arange(100) * 3 % len(nd)]                  # some features equal to node descriptors
    leaf, nid = O.bow_descend(co, ci, nd, 4, f, 2)
    pop = np.array([bin(i).count("1") for i in range(256)], np.int32)
    for i in range(len(f)):
        node, level, nn = 0, 0, 0
        while co[node + 1] > co[node]:
            level += 1
            ch = ci[co[node]:co[node + 1]]
            d = pop[nd[ch] ^ f[i]].sum(axis=1)
            node = int(ch[int(np.argmin(d))])                   # argmin: the first minimum
            if level == 4 - 2:
                nn = node
        assert leaf[i] == node and nid[i] == nn


def test_distinctive_descriptor_hand_cases():
    """MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:273-302): least median, self distance in the row, first wins."""
    z = np.zeros((1, 32), np.uint8)
    def bits(k):
        d = np.zeros(32, np.uint8)
        for b in range(k):
            d[b >> 3] |= 1 << (b & 7)
        return d
    # three collinear descriptors 0, 10 and 30 bits set: rows sorted (0,10,30), (0,10,20), (0,20,30); rank (int)(0.5*2) = 1
    # -> medians 10, 10, 20: the first of the two minima (index 0) wins
    p0 = np.stack([bits(0), bits(10), bits(30)])
    # four descriptors: rank (int)(0.5*3) = 1 -> the second smallest of each row, i.e. the nearest neighbour's distance
    p1 = np.stack([bits(0), bits(100), bits(104), bits(200)])     # nearest distances: 100, 4, 4, 96 -> index 1
    p2 = np.stack([bits(7)])                                      # single observation -> itself
    desc = np.concatenate([p0, p1, p2])
    off = np.array([0, 3, 7, 8, 8], np.int32)                     # last point: no descriptors
    assert O.distinctive(desc, off).tolist() == [0, 1, 0, -1]
    # numpy restatement on random points
    rng = np.random.default_rng(3)
    sizes = rng.integers(1, 40, 200)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    desc = synth.random_descriptors(int(off[-1]), 11)
    desc[5] = desc[6]                                             # duplicates inside a point
    got = O.distinctive(desc, off)
    pop = np.array([bin(i).count("1") for i in range(256)], np.int32)
    for p in range(len(sizes)):
        d = desc[off[p]:off[p + 1]]
        D = pop[d[:, None, :] ^ d[None, :, :]].sum(axis=2)
        med = np.sort(D, axis=1)[:, int(0.5 * (len(d) - 1))]
        assert got[p] == int(np.argmin(med))


def test_frame_grid_features_in_area_vs_bruteforce():
    """Frame::GetFeaturesInArea (src/Frame.cc:327-380): the same SET as a brute-force window test, in grid order
    (cell column, cell row, keypoint index); level filter semantics (:349, 362-370)."""
    rng = np.random.default_rng(5)
    n = 1500
    kx = rng.uniform(0, 1241, n).astype(np.float32)
    ky = rng.uniform(0, 376, n).astype(np.float32)
    koct = rng.integers(0, 8, n).astype(np.int32)
    kx[:3], ky[:3] = [-5.0, 1300.0, 10.0], [10.0, 10.0, -3.0]        # outside the grid: never returned (PosInGrid :388)
    w_inv, h_inv = np.float32(64) / np.float32(1241), np.float32(48) / np.float32(376)
    g = O.OracleGrid(kx, ky, koct, np.zeros((n, 32), np.uint8), None, 0.0, 0.0, float(w_inv), float(h_inv))
    for x, y, r, lo, hi in [(600, 200, 40, -1, -1), (5, 5, 30, 0, 3), (1235, 370, 25, 2, -1), (300, 100, 15.5, 3, 3),
                            (-100, 50, 20, -1, -1), (700, 188, 400, 0, 7)]:
        got = g.features_in_area(x, y, r, lo, hi)
        posx = np.round((kx - 0) * w_inv).astype(int)              # round(): never .5 exactly for these random values
        posy = np.round((ky - 0) * h_inv).astype(int)
        ok = (posx >= 0) & (posx < 64) & (posy >= 0) & (posy < 48)
        ok &= (np.abs(kx - np.float32(x)) < r) & (np.abs(ky - np.float32(y)) < r)
        if lo > 0 or hi >= 0:
            ok &= koct >= lo
            if hi >= 0:
                ok &= koct <= hi
        # cells reached: floor/ceil of the window (:332-346)
        cx0, cx1 = max(0, int(np.floor((x - r) * w_inv))), min(63, int(np.ceil((x + r) * w_inv)))
        cy0, cy1 = max(0, int(np.floor((y - r) * h_inv))), min(47, int(np.ceil((y + r) * h_inv)))
        ok &= (posx >= cx0) & (posx <= cx1) & (posy >= cy0) & (posy <= cy1)
        want = sorted(np.nonzero(ok)[0], key=lambda i: (posx[i], posy[i], i))
        assert got.tolist() == [int(i) for i in want]


def test_search_area_best_second_rule():
    """src/ORBmatcher.cc:71-115: strict '<' updates in list order, blocked keypoints and the right-coordinate gate."""
    kx = np.array([100, 101, 102, 103, 104], np.float32)
    ky = np.full(5, 50, np.float32)
    koct = np.array([2, 2, 3, 2, 2], np.int32)
    d = np.zeros((5, 32), np.uint8)
    d[0, 0] = 0b00000111      # distance 3 to the zero query
    d[1, 0] = 0b00000001      # 1
    d[2, 0] = 0b00000001      # 1 (tie with index 1: the first stays best, this one becomes second)
    d[3, 0] = 0b00111111      # 6
    d[4, 0] = 0               # 0, but blocked / gated below
    ur = np.array([-1, 90, 90, 90, 40], np.float32)
    g = O.OracleGrid(kx, ky, koct, d, ur, 0.0, 0.0, 64 / 1241, 48 / 376)
    q = np.zeros(32, np.uint8)
    assert g.search(102, 50, 10, 1, 3, 0.0, -1.0, q).tolist() == [0, 4, 2, 1, 2, 5, 1]
    blocked = np.array([0, 0, 0, 0, 1], np.uint8)
    assert g.search(102, 50, 10, 1, 3, 0.0, -1.0, q, blocked).tolist() == [1, 1, 2, 1, 3, 5, 2]
    # gate: |proj_xr - uRight| > er_max drops index 4 (uRight 40) but keeps the others (uRight 90) and index 0 (uRight <= 0)
    assert g.search(102, 50, 10, 1, 3, 92.0, 5.0, q).tolist() == [1, 1, 2, 1, 3, 5, 2]
    assert g.search(102, 50, 10, 3, 3, 0.0, -1.0, q).tolist() == [1, 2, 3, 256, -1, 1, -1]
    assert g.search(500, 300, 10, -1, -1, 0.0, -1.0, q).tolist() == [256, -1, -1, 256, -1, 0, -1]


def test_search_area_result_depends_only_on_its_two_holders():
    """The replay rule of INTEGRATION.md: blocking a keypoint that holds neither the best nor the second place of a query
    leaves that query's result unchanged, so a result is stale only if best_idx or best_idx2 got associated later."""
    rng = np.random.default_rng(31)
    n = 400
    kx = rng.uniform(0, 300, n).astype(np.float32)
    ky = rng.uniform(0, 200, n).astype(np.float32)
    koct = rng.integers(0, 4, n).astype(np.int32)
    d = rng.integers(0, 4, (n, 32)).astype(np.uint8)          # few distinct distances: many ties
    g = O.OracleGrid(kx, ky, koct, d, None, 0.0, 0.0, 64 / 300, 48 / 200)
    changed_by_holder = 0
    for t in range(200):
        x, y, r = float(rng.uniform(0, 300)), float(rng.uniform(0, 200)), float(rng.uniform(10, 40))
        q = rng.integers(0, 4, 32).astype(np.uint8)
        base = g.search(x, y, r, -1, -1, 0.0, -1.0, q)
        cand = g.features_in_area(x, y, r, -1, -1)
        if len(cand) < 3:
            continue
        for idx in rng.choice(cand, min(6, len(cand)), replace=False):
            blocked = np.zeros(n, np.uint8)
            blocked[idx] = 1
            after = g.search(x, y, r, -1, -1, 0.0, -1.0, q, blocked)
            if idx not in (base[1], base[6]):
                assert after.tolist() == base.tolist()
            else:
                changed_by_holder += after.tolist() != base.tolist()
    assert changed_by_holder > 20


def _bow_brute(desc1, valid1, desc2, blocked2, off1, idx1, off2, idx2, max_dist, ratio):
    """numpy restatement of the SearchByBoW loops, written independently of the oracle's C++ (sorting instead of scanning)."""
    pop = np.array([bin(i).count("1") for i in range(256)], np.int32)
    taken = np.zeros(len(desc2), bool) if blocked2 is None else np.asarray(blocked2, bool).copy()
    match1 = np.full(len(desc1), -1, np.int32)
    for k in range(len(off1) - 1):
        for a in idx1[off1[k]:off1[k + 1]]:
            if valid1 is not None and not valid1[a]:
                continue
            cand = [int(b) for b in idx2[off2[k]:off2[k + 1]] if not taken[b]]
            if not cand:
                continue
            d = pop[desc1[a][None, :] ^ desc2[cand]].sum(1)
            order = np.argsort(d, kind="stable")                    # first minimum = first in list order
            best, second = int(d[order[0]]), (int(d[order[1]]) if len(cand) > 1 else 256)
            if best <= max_dist and np.float32(best) < np.float32(ratio) * np.float32(second):
                match1[a] = cand[order[0]]
                taken[cand[order[0]]] = True
    return match1


def test_search_by_bow_oracle_vs_numpy():
    rng = np.random.default_rng(41)
    n1, n2, nn = 600, 700, 40
    d2 = synth.random_descriptors(n2, 5)
    src = rng.integers(0, n2, n1)
    d1 = d2[src].copy()                                             # side 1 = noisy copies of side-2 features: real matches,
    for i in range(n1):                                             # several side-1 features competing for one side-2 feature
        for b in rng.integers(0, 256, int(rng.integers(0, 40))):
            d1[i, b >> 3] ^= np.uint8(1 << (b & 7))
    node2 = rng.integers(0, nn, n2)
    node1 = node2[src]
    node1[:50] = rng.integers(0, nn, 50)                            # some features in the wrong node
    def csr(node, n):
        order = [np.nonzero(node == k)[0] for k in range(nn)]
        for o in order:
            rng.shuffle(o)                                          # FeatureVector order is insertion order, not sorted
        return np.concatenate([[0], np.cumsum([len(o) for o in order])]).astype(np.int32), np.concatenate(order).astype(np.int32)
    off1, idx1 = csr(node1, n1)
    off2, idx2 = csr(node2, n2)
    valid1 = (rng.uniform(size=n1) < 0.8).astype(np.uint8)
    blocked2 = (rng.uniform(size=n2) < 0.1).astype(np.uint8)
    for v1, b2, md, ratio in ((valid1, None, 50, 0.7), (valid1, blocked2, 49, 0.75), (None, None, 100, 0.9)):
        got = O.search_by_bow(d1, v1, d2, b2, off1, idx1, off2, idx2, md, ratio)
        want = _bow_brute(d1, v1, d2, b2, off1, idx1, off2, idx2, md, ratio)
        assert np.array_equal(got, want)
        m = got[got >= 0]
        assert len(m) > 100 and len(np.unique(m)) == len(m)         # a side-2 feature is matched at most once


def test_three_maxima_and_rotation_filter():
    """ComputeThreeMaxima (src/ORBmatcher.cc:1601-1642): strict '>' updates, the 10 % rule; then the histogram filter."""
    assert O.three_maxima([0] * 30) == [-1, -1, -1]
    s = [0] * 30
    s[4], s[9], s[2] = 10, 7, 7                                     # equal sizes: the earlier bin stays ahead
    assert O.three_maxima(s) == [4, 2, 9]
    s = [0] * 30
    s[1], s[5], s[7] = 100, 9, 50                                   # third below 10 % of the first: dropped
    assert O.three_maxima(s) == [1, 7, -1]
    s = [0] * 30
    s[3], s[8] = 100, 9                                             # second below 10 %: second and third dropped
    assert O.three_maxima(s) == [3, -1, -1]
    # rotation filter: bin = round(rot * (1/30)), half away from zero.  Differences 1..4 and 14.9 -> bin 0 (5 matches),
    # 45..47 -> 1.5.. -> bin 2 (3), 100, 101 -> bin 3 (2), 200 -> bin 7 (1), 359 -> bin 12 (1): bins 0, 2, 3 survive.
    a2 = np.zeros(12, np.float32)
    a1 = np.array([1, 2, 3, 4, 45, 46, 47, 100, 101, 200, 359, 14.9], np.float32)
    m, n = O.rotation_filter(a1, a2, np.arange(12))
    assert n == 10 and m.tolist() == [0, 1, 2, 3, 4, 5, 6, 7, 8, -1, -1, 11]
    # negative differences wrap by +360: -10 -> 350 -> bin 12; unmatched entries stay out of the histogram
    m, n = O.rotation_filter(np.array([0, 0, 0], np.float32), np.array([10, 10, 0], np.float32), np.array([0, 1, -1]))
    assert n == 2 and m.tolist() == [0, 1, -1]


def test_sincos_census_sampled(tmp_path):
    """tools/sincos_census.c at stride 509 (2.2 M of the 1.136 G float angles; the committed tests/golden/sincos_census.json
    is the stride-1 run): the host mirror of the device's cos/sin routine equals libm for every angle evaluated, and the
    golden file's libm values are the ones this host's libm gives."""
    import ctypes
    import ctypes.util
    import json
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not found")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "sincos_census")
    subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-fopenmp", os.path.join(root, "tools", "sincos_census.c"), "-o", exe,
                    "-lm"], check=True)
    out = json.loads(subprocess.run([exe, "509"], check=True, capture_output=True, text=True).stdout)
    assert out["angles_evaluated"] == (1135869952 + 508) // 509
    assert out["glibc"]["angles_where_cos_or_sin_differs"] == 0
    assert out["taylor"]["angles_where_cos_or_sin_differs"] > 1000      # the census does see the routine it was written for

    gold = json.load(open(os.path.join(root, "tests", "golden", "sincos_census.json")))
    m = ctypes.CDLL(ctypes.util.find_library("m"))
    for f in (m.cosf, m.sinf):
        f.restype, f.argtypes = ctypes.c_float, [ctypes.c_float]
    factor_pi = np.float32(3.1415926535897932384626433832795 / np.float32(180.0))
    rows = np.array(gold["tap_moving_angles"] + gold["sample_of_differing_angles"], np.uint32)
    ang = (rows[:, 0].copy().view(np.float32) * factor_pi).astype(np.float32)
    want = np.array([(m.cosf(float(a)), m.sinf(float(a))) for a in ang], np.float32).view(np.uint32)
    assert np.array_equal(want, rows[:, 1:3])
