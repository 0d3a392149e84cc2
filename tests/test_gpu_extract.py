"""GPU parity of the extractor: liborbfe (HIP, through the C-ABI) against the CPU oracle, stage by stage and end to end.
Bar (BASELINE.json north_star): pyramid levels, FAST corners, kept keypoints, descriptors bit-exact; angles within
1e-3 degrees (in practice identical floats)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from orb_slam2_amd import capi, synth

pytestmark = pytest.mark.gpu

CONFIGS = {
    "kitti": dict(w=1241, h=376, nfeatures=2000),
    "tum": dict(w=640, h=480, nfeatures=1000),
    "euroc": dict(w=752, h=480, nfeatures=1200),
}


def _pair(cfg, seed):
    c = CONFIGS[cfg]
    img = synth.textured_image(c["w"], c["h"], synth.BASE_SEED + seed)
    g = capi.ORBextractor(c["nfeatures"], 1.2, 8, 20, 7)
    o = O.OracleExtractor(c["nfeatures"], 1.2, 8, 20, 7)
    return img, g, o


def test_tables_match_oracle():
    g = capi.ORBextractor(2000, 1.2, 8, 20, 7)
    o = O.OracleExtractor(2000, 1.2, 8, 20, 7)
    tg, to = g._tables(), o.tables()
    for k in to:
        assert np.array_equal(tg[k], to[k]), k


@pytest.mark.parametrize("mode", ["device", "host"])
@pytest.mark.parametrize("cfg", ["kitti", "tum", "euroc"])
def test_stagewise_parity(cfg, mode):
    img, g, o = _pair(cfg, 3)
    g.set_quadtree_mode(mode)
    kg, dg = g(img)
    ko, do = o.extract(img)
    for l in range(8):
        assert np.array_equal(g.debug_stage(l, 0), o.level(l)), f"pyramid level {l}"
    for l in range(8):
        cg, co = g.debug_stage(l, 2), o.candidates(l)
        assert cg.shape == co.shape and np.array_equal(cg, co), f"FAST candidates level {l}"
    for l in range(8):
        sg, so = g.debug_stage(l, 3), o.selected(l)
        assert sg.shape[0] == so.shape[0] and np.array_equal(sg[:, :3], so[:, :3]), f"kept keypoints level {l}"
    for l in range(8):
        bo = o.blurred(l)
        if bo is not None:
            assert np.array_equal(g.debug_stage(l, 1), bo), f"blurred level {l}"
    assert len(kg) == len(ko)
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        assert np.array_equal(kg[f], ko[f]), f
    assert np.abs(kg["angle"] - ko["angle"]).max() <= 1e-3
    assert np.array_equal(kg["angle"], ko["angle"])
    nbits = int(np.unpackbits(dg ^ do).sum())
    assert nbits == 0, f"{nbits} descriptor bits differ"


def test_quadtree_quota_sweep():
    """Small and large quotas drive the quadtree through its different stop rules (N < 4*nIni, breadth-only, size-ordered)."""
    img = synth.textured_image(1241, 376, synth.BASE_SEED + 30)
    for nf in (8, 40, 150, 500, 1000, 3000, 4500):
        g = capi.ORBextractor(nf, 1.2, 8, 20, 7)
        o = O.OracleExtractor(nf, 1.2, 8, 20, 7)
        kg, dg = g(img)
        ko, do = o.extract(img)
        assert np.array_equal(kg, ko) and np.array_equal(dg, do), nf
    g = capi.ORBextractor(6000, 1.2, 8, 20, 7)      # quota beyond the device node table: host quadtree takes over
    o = O.OracleExtractor(6000, 1.2, 8, 20, 7)
    kg, dg = g(img)
    ko, do = o.extract(img)
    assert np.array_equal(kg, ko) and np.array_equal(dg, do)


def test_end_to_end_many_frames():
    c = CONFIGS["kitti"]
    g = capi.ORBextractor(c["nfeatures"], 1.2, 8, 20, 7)
    o = O.OracleExtractor(c["nfeatures"], 1.2, 8, 20, 7)
    for seed in range(10, 16):
        img = synth.textured_image(c["w"], c["h"], synth.BASE_SEED + seed)
        kg, dg = g(img)
        ko, do = o.extract(img)
        assert np.array_equal(kg, ko) and np.array_equal(dg, do), seed


def test_strided_input_and_pyramid_border():
    img, g, o = _pair("tum", 4)
    big = np.zeros((480, 700), np.uint8)
    big[:, :640] = img
    view = big[:, :640]                      # row stride 700 != width, as a cv::Mat ROI
    kg, dg = g(view)
    ko, do = o.extract(img)
    assert np.array_equal(kg, ko) and np.array_equal(dg, do)
    lvl = g.pyramid_level(2, with_border=True)
    roi = o.level(2)
    assert np.array_equal(lvl[19:-19, 19:-19], roi)
    exp = np.pad(roi, 19, mode="reflect")    # numpy 'reflect' == BORDER_REFLECT_101
    assert np.array_equal(lvl, exp)


def test_degenerate_inputs():
    g = capi.ORBextractor(1000, 1.2, 8, 20, 7)
    k, d = g(np.zeros((0, 0), np.uint8))
    assert len(k) == 0 and d.shape == (0, 32)
    k, d = g(np.full((480, 640), 90, np.uint8))      # featureless: zero keypoints, descriptors released
    assert len(k) == 0
    with pytest.raises(capi.OrbfeError):
        g(np.zeros((100, 100), np.uint8))            # smallest level < one FAST cell: reference divides by zero


def test_batch_device_matches_single(tmp_path):
    torch = pytest.importorskip("torch")
    c = CONFIGS["kitti"]
    F = 5
    frames = synth.frame_sequence(F, c["w"], c["h"], first=20)
    stride = (c["w"] + 63) // 64 * 64
    host = np.zeros((F, c["h"], stride), np.uint8)
    host[:, :, :c["w"]] = frames
    dev = torch.from_numpy(host).cuda()
    cap = 2100
    kps = torch.zeros((F, cap, 7), dtype=torch.float32, device="cuda")
    desc = torch.zeros((F, cap, 32), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(F, dtype=torch.int32, device="cuda")
    g = capi.ORBextractor(c["nfeatures"], 1.2, 8, 20, 7)
    torch.cuda.synchronize()
    g.extract_batch_device(dev.data_ptr(), F, c["w"], c["h"], stride, stride * c["h"], kps.data_ptr(), desc.data_ptr(),
                           cnt.data_ptr(), cap, sync=True)
    cnt_h = cnt.cpu().numpy()
    kps_h = kps.cpu().numpy().view(np.uint8).reshape(F, cap, 28)
    desc_h = desc.cpu().numpy()
    o = O.OracleExtractor(c["nfeatures"], 1.2, 8, 20, 7)
    for f in range(F):
        ko, do = o.extract(frames[f])
        n = int(cnt_h[f])
        assert n == len(ko)
        kg = kps_h[f, :n].copy().view(capi.KP_DTYPE).reshape(-1)
        assert np.array_equal(kg, ko), f
        assert np.array_equal(desc_h[f, :n], do), f


def _stage_check(img, nfeatures, scale, nlevels, ini_th=20, min_th=7):
    g = capi.ORBextractor(nfeatures, scale, nlevels, ini_th, min_th)
    o = O.OracleExtractor(nfeatures, scale, nlevels, ini_th, min_th)
    kg, dg = g(img)
    ko, do = o.extract(img)
    for l in range(nlevels):
        assert np.array_equal(g.debug_stage(l, 0), o.level(l)), f"pyramid level {l}"
        cg, co = g.debug_stage(l, 2), o.candidates(l)
        assert cg.shape == co.shape and np.array_equal(cg, co), f"FAST candidates level {l}"
        bo = o.blurred(l)
        if bo is not None:
            assert np.array_equal(g.debug_stage(l, 1), bo), f"blurred level {l}"
    assert np.array_equal(kg, ko)
    assert np.array_equal(dg, do)
    return len(kg)


@pytest.mark.parametrize("w,h", [(320, 240), (401, 301), (512, 512), (1280, 720), (1920, 1080), (1024, 96), (1283, 377),
                                 (1216, 352), (250, 1100)])
def test_odd_sizes(w, h):
    """Widths that are / are not multiples of 4, 64 and of the strip widths (pitch == w, partial last strips, packed FAST
    remainders, tall narrow levels); every stage bit-exact."""
    nlevels = 8
    # the reference needs >= one 30-px cell on the smallest level and a landscape-or-square detection area (nIni >= 1)
    while nlevels > 1 and (min(w, h) / 1.2 ** (nlevels - 1) < 70):
        nlevels -= 1
    img = synth.textured_image(w, h, synth.BASE_SEED + w + h)
    if w < h:   # portrait: the reference's nIni = round(W/H) is 0 and it divides by it (:543-545); the library refuses
        with pytest.raises(capi.OrbfeError):
            capi.ORBextractor(1500, 1.2, nlevels, 20, 7)(img)
        return
    assert _stage_check(img, 1500, 1.2, nlevels) > 300


@pytest.mark.parametrize("scale,nlevels", [(1.1, 10), (1.25, 6), (1.5, 5), (2.0, 4), (1.2, 1), (1.2, 12), (1.33, 7)])
def test_scale_factors_and_level_counts(scale, nlevels):
    """The resize kernel's 8-byte source window and row-block choice depend on the scale factor (wide 12-byte variant from ~2 on)."""
    w, h = (1241, 376) if scale ** (nlevels - 1) < 4.5 else (2048, 1024)   # the smallest level must hold a FAST cell
    img = synth.textured_image(w, h, synth.BASE_SEED + int(scale * 100) + nlevels)
    assert _stage_check(img, 1200, scale, nlevels) > 300


@pytest.mark.parametrize("ini_th,min_th", [(20, 7), (40, 5), (7, 7), (100, 30), (5, 20), (20, 0), (20, 1), (0, 0), (1, 0), (2, 1)])
def test_fast_thresholds(ini_th, min_th):
    """Per-cell iniTh -> minTh fallback for different threshold pairs (incl. ini < min, which the reference also accepts).
    Thresholds 0 and 1: cv::FAST stores S - 1, so a corner with S = 1 at threshold 0 scores 0 and never passes its NMS."""
    img = synth.textured_image(752, 480, synth.BASE_SEED + 500 + ini_th)
    img[:, :200] = (img[:, :200].astype(np.int32) // 8 + 100).astype(np.uint8)      # low-contrast band: cells that fall back
    _stage_check(img, 1000, 1.2, 8, ini_th, min_th)


def _fuzz_image(rng, w, h, kind, seed):
    img = synth.textured_image(w, h, synth.BASE_SEED + seed)
    if kind == "low_contrast":            # most cells fall back to minTh, many find nothing at all
        img = (img.astype(np.int32) // int(rng.integers(6, 20)) + int(rng.integers(0, 200))).astype(np.uint8)
    elif kind == "flat_regions":          # constant blocks: empty cells, empty levels, plateaus of equal scores
        for _ in range(int(rng.integers(2, 8))):
            x0, y0 = int(rng.integers(0, w - 8)), int(rng.integers(0, h - 8))
            x1, y1 = int(rng.integers(x0 + 8, w + 1)), int(rng.integers(y0 + 8, h + 1))
            img[y0:y1, x0:x1] = int(rng.integers(0, 256))
    elif kind == "saturated":             # values at 0 / 255: u8 wrap and saturation paths of blur and resize
        img = np.where(img > 128, 255, 0).astype(np.uint8)
    elif kind == "noise":                 # a corner nearly everywhere: candidate lists at their largest
        img = rng.integers(0, 256, (h, w)).astype(np.uint8)
    return np.ascontiguousarray(img)


_FUZZ_FIRST = int(os.environ.get("ORBFE_FUZZ_FIRST", "0"))     # a soak continues where the last one stopped


@pytest.mark.parametrize("case", range(_FUZZ_FIRST, _FUZZ_FIRST + int(os.environ.get("ORBFE_FUZZ_CASES", "40"))))
def test_fuzz_configurations(case):
    """Seeded random configurations (size, quota, scale factor, level count, thresholds, image statistics), every stage
    bit-exact against the oracle.  Configurations the reference itself cannot run (a level smaller than one FAST cell,
    portrait detection area) are avoided by construction."""
    rng = np.random.default_rng(9000 + case)
    scale = float(rng.choice([1.1, 1.15, 1.2, 1.2, 1.25, 1.3, 1.41, 1.5, 1.7, 2.0]))
    nlevels = int(rng.integers(1, 11))
    w = int(rng.integers(224, 1500))
    h = int(rng.integers(224, min(w, 1100) + 1))                    # landscape or square: nIni = round(W/H) >= 1
    while nlevels > 1 and min(w, h) / scale ** (nlevels - 1) < 72:   # smallest level holds one 30-px cell plus borders
        nlevels -= 1
    nfeatures = int(rng.choice([30, 100, 500, 1000, 2000, 3500]))
    ini_th = int(rng.integers(8, 45))
    min_th = int(rng.integers(2, ini_th + 1)) if case % 5 else int(rng.integers(0, 2))   # every fifth case: minTh 0 or 1
    kind = str(rng.choice(["textured", "textured", "low_contrast", "flat_regions", "saturated", "noise"]))
    img = _fuzz_image(rng, w, h, kind, 700 + case)
    cfg = (w, h, nfeatures, scale, nlevels, ini_th, min_th, kind)
    try:
        _stage_check(img, nfeatures, scale, nlevels, ini_th, min_th)
    except AssertionError as e:
        raise AssertionError(f"config {cfg}: {e}") from e


@pytest.mark.parametrize("case", range(_FUZZ_FIRST, _FUZZ_FIRST + max(1, int(os.environ.get("ORBFE_FUZZ_CASES", "40")) // 5)))
def test_fuzz_queued_batches(case):
    """The batch entry point under the same random configurations: one to four asynchronous batches of one to six frames of
    mixed statistics (noise frames among ordinary ones: the overflow rerun of queued batches; `cap` sometimes below the
    keypoint count: truncated outputs, full counts), one synchronisation, every frame equal to the oracle."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(19000 + case)
    scale = float(rng.choice([1.1, 1.2, 1.2, 1.3, 1.5, 2.0]))
    nlevels = int(rng.integers(1, 9))
    w = int(rng.integers(224, 1300))
    h = int(rng.integers(224, min(w, 800) + 1))
    while nlevels > 1 and min(w, h) / scale ** (nlevels - 1) < 72:
        nlevels -= 1
    nfeatures = int(rng.choice([30, 300, 1000, 2000]))
    ini_th = int(rng.integers(8, 45))
    min_th = int(rng.integers(2, ini_th + 1))
    o = O.OracleExtractor(nfeatures, scale, nlevels, ini_th, min_th)
    g = capi.ORBextractor(nfeatures, scale, nlevels, ini_th, min_th)
    stride = (w + 63) // 64 * 64
    cap = int(nfeatures * float(rng.choice([0.5, 1.1, 1.1, 2.0]))) + 4 * nlevels + 8
    held = []
    for b in range(int(rng.integers(1, 5))):
        F = int(rng.integers(1, 7))
        host = np.zeros((F, h, stride), np.uint8)
        for f in range(F):
            kind = str(rng.choice(["textured", "textured", "textured", "low_contrast", "flat_regions", "saturated", "noise"]))
            host[f, :, :w] = _fuzz_image(rng, w, h, kind, 30000 + 64 * case + 8 * b + f)
        frames = torch.from_numpy(host).cuda()
        kps = torch.zeros((F, cap, 7), dtype=torch.float32, device="cuda")
        desc = torch.zeros((F, cap, 32), dtype=torch.uint8, device="cuda")
        cnt = torch.zeros((F,), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        g.extract_batch_device(frames.data_ptr(), F, w, h, stride, stride * h, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(),
                               cap, sync=False)
        held.append((host, frames, kps, desc, cnt))
    g.synchronize()
    cfg = (case, w, h, nfeatures, scale, nlevels, ini_th, min_th, cap)
    for b, (host, _frames, kps, desc, cnt) in enumerate(held):
        n = cnt.cpu().numpy()
        kh = kps.cpu().numpy().view(np.uint8).reshape(len(host), cap, 28)
        dh = desc.cpu().numpy()
        for f in range(len(host)):
            kr, dr = o.extract(np.ascontiguousarray(host[f, :, :w]))
            assert n[f] == len(kr), (cfg, b, f)
            m = min(len(kr), cap)
            assert np.array_equal(kh[f, :m].copy().view(capi.KP_DTYPE).reshape(-1), kr[:m]), (cfg, b, f)
            assert np.array_equal(dh[f, :m], dr[:m]), (cfg, b, f)


def test_dense_noise_switches_to_the_large_survivor_list():
    """Uniform noise at minTh = 2: a FAST strip yields more survivor records than the normal kernel variant keeps in LDS.
    The handle reruns the batch with the large-list variant -- transparently on the synchronous paths and in
    orbfe_synchronize() for a single asynchronous batch -- and the results stay bit-exact."""
    torch = pytest.importorskip("torch")
    w, h = 1491, 344
    img = np.random.default_rng(9176).integers(0, 256, (h, w)).astype(np.uint8)
    o = O.OracleExtractor(100, 2.0, 3, 35, 2)
    ko, do = o.extract(img)
    assert sum(len(o.candidates(l)) for l in range(3)) > 40000        # about one pixel in nine survives
    g = capi.ORBextractor(100, 2.0, 3, 35, 2)
    kg, dg = g(img)                                                   # host image path: retried inside the call
    assert np.array_equal(kg, ko) and np.array_equal(dg, do)
    stride = (w + 63) // 64 * 64
    host = np.zeros((3, h, stride), np.uint8)
    host[:, :, :w] = img
    host[1, :, :w] = synth.textured_image(w, h, synth.BASE_SEED + 3)  # an ordinary frame between two noise frames
    k1, d1 = o.extract(np.ascontiguousarray(host[1, :, :w]))
    frames = torch.from_numpy(host).cuda()
    cap = 200
    for sync in (True, False):
        g2 = capi.ORBextractor(100, 2.0, 3, 35, 2)                    # a fresh handle: starts with the normal variant
        kps = torch.zeros((3, cap, 7), dtype=torch.float32, device="cuda")
        desc = torch.zeros((3, cap, 32), dtype=torch.uint8, device="cuda")
        cnt = torch.zeros(3, dtype=torch.int32, device="cuda")
        g2.extract_batch_device(frames.data_ptr(), 3, w, h, stride, stride * h, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(),
                                cap, sync=sync)
        if not sync:
            g2.synchronize()                                          # reruns the one batch in flight
        n = cnt.cpu().numpy()
        kh = kps.cpu().numpy().view(np.uint8).reshape(3, cap, 28)
        dh = desc.cpu().numpy()
        for f, (kr, dr) in enumerate([(ko, do), (k1, d1), (ko, do)]):
            assert n[f] == len(kr)
            assert np.array_equal(kh[f, :n[f]].copy().view(capi.KP_DTYPE).reshape(-1), kr), (sync, f)
            assert np.array_equal(dh[f, :n[f]], dr), (sync, f)


def test_overflow_recovery_with_several_batches_queued():
    """Five asynchronous batches queued back to back (the offline-extraction pattern of bench.py), uniform noise in a frame of
    the SECOND: orbfe_synchronize runs all five again with the large-list kernel and every output equals the oracle's.  Then
    the detour ends by itself: after 16 further batches the handle is back on the normal kernel, and ordinary frames stay
    bit-exact on the way."""
    torch = pytest.importorskip("torch")
    w, h = 1491, 344
    noise = np.random.default_rng(9176).integers(0, 256, (h, w)).astype(np.uint8)
    o = O.OracleExtractor(100, 2.0, 3, 35, 2)
    stride = (w + 63) // 64 * 64
    nb, fpb, cap = 5, 2, 200
    host = np.zeros((nb, fpb, h, stride), np.uint8)
    for b in range(nb):
        for f in range(fpb):
            host[b, f, :, :w] = synth.textured_image(w, h, synth.BASE_SEED + 20 + 2 * b + f)
    host[1, 1, :, :w] = noise
    want = [[o.extract(np.ascontiguousarray(host[b, f, :, :w])) for f in range(fpb)] for b in range(nb)]
    frames = torch.from_numpy(host).cuda()
    g = capi.ORBextractor(100, 2.0, 3, 35, 2)
    assert g.fast_variant() == 1
    kps = torch.zeros((nb, fpb, cap, 7), dtype=torch.float32, device="cuda")
    desc = torch.zeros((nb, fpb, cap, 32), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros((nb, fpb), dtype=torch.int32, device="cuda")

    def enqueue(b):
        g.extract_batch_device(frames[b].data_ptr(), fpb, w, h, stride, stride * h, kps[b].data_ptr(), desc[b].data_ptr(),
                               cnt[b].data_ptr(), cap, sync=False)

    def check_batch(b):
        n = cnt[b].cpu().numpy()
        kh = kps[b].cpu().numpy().view(np.uint8).reshape(fpb, cap, 28)
        dh = desc[b].cpu().numpy()
        for f in range(fpb):
            kr, dr = want[b][f]
            assert n[f] == len(kr), (b, f)
            assert np.array_equal(kh[f, :n[f]].copy().view(capi.KP_DTYPE).reshape(-1), kr), (b, f)
            assert np.array_equal(dh[f, :n[f]], dr), (b, f)

    for b in range(nb):
        enqueue(b)
    g.synchronize()                      # status bit 2 -> all five batches again on the large-list kernel
    assert g.fast_variant() == 2
    for b in range(nb):
        check_batch(b)
    # the detour: 16 batches on the large list, then the normal kernel again
    for k in range(17):
        kps.zero_(); cnt.zero_()
        enqueue(0)
        g.synchronize()
        check_batch(0)
    assert g.fast_variant() == 1
    # noise again: one rerun, and the next detour lasts twice as long
    enqueue(0); enqueue(1); enqueue(2)
    g.synchronize()
    assert g.fast_variant() == 2
    for b in range(3):
        check_batch(b)


def test_queued_batches_of_changing_frame_count_and_image_size():
    """Asynchronous batches queued back to back whose frame count grows (the workspace is re-allocated with earlier batches
    still in flight) and whose image size changes (the level geometry and its device tables are rebuilt in between): after
    one orbfe_synchronize every batch equals the oracle, frame by frame."""
    torch = pytest.importorskip("torch")
    o = O.OracleExtractor(300, 1.2, 5, 20, 7)
    g = capi.ORBextractor(300, 1.2, 5, 20, 7)
    cap = 400
    shapes = [(2, 320, 240), (7, 320, 240), (3, 411, 233), (1, 320, 240), (12, 411, 233), (4, 640, 200)]
    held = []
    for b, (F, w, h) in enumerate(shapes):
        stride = (w + 63) // 64 * 64
        host = np.zeros((F, h, stride), np.uint8)
        for f in range(F):
            host[f, :, :w] = synth.textured_image(w, h, synth.BASE_SEED + 400 + 16 * b + f)
        frames = torch.from_numpy(host).cuda()
        kps = torch.zeros((F, cap, 7), dtype=torch.float32, device="cuda")
        desc = torch.zeros((F, cap, 32), dtype=torch.uint8, device="cuda")
        cnt = torch.zeros((F,), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        g.extract_batch_device(frames.data_ptr(), F, w, h, stride, stride * h, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(),
                               cap, sync=False)
        held.append((host, frames, kps, desc, cnt))
    g.synchronize()
    for (F, w, h), (host, _frames, kps, desc, cnt) in zip(shapes, held):
        n = cnt.cpu().numpy()
        kh = kps.cpu().numpy().view(np.uint8).reshape(F, cap, 28)
        dh = desc.cpu().numpy()
        for f in range(F):
            kr, dr = o.extract(np.ascontiguousarray(host[f, :, :w]))
            assert n[f] == len(kr) and len(kr) > 50, (F, w, h, f)
            assert np.array_equal(kh[f, :n[f]].copy().view(capi.KP_DTYPE).reshape(-1), kr), (F, w, h, f)
            assert np.array_equal(dh[f, :n[f]], dr), (F, w, h, f)


def test_record_of_queued_batches_is_bounded():
    """A caller that queues asynchronous batches and never synchronises: the library keeps a record of every queued call
    (for the overflow rerun) and synchronises by itself every 4096 calls, so the record cannot grow without bound.  4200
    one-frame batches into four rotating output slots; the last four equal the oracle."""
    torch = pytest.importorskip("torch")
    w, h, cap = 320, 240, 300
    o = O.OracleExtractor(200, 1.2, 4, 20, 7)
    g = capi.ORBextractor(200, 1.2, 4, 20, 7)
    stride = (w + 63) // 64 * 64
    host = np.zeros((4, h, stride), np.uint8)
    for f in range(4):
        host[f, :, :w] = synth.textured_image(w, h, synth.BASE_SEED + 900 + f)
    frames = torch.from_numpy(host).cuda()
    kps = torch.zeros((4, cap, 7), dtype=torch.float32, device="cuda")
    desc = torch.zeros((4, cap, 32), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros((4,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for k in range(4200):
        s = k & 3
        g.extract_batch_device(frames[s].data_ptr(), 1, w, h, stride, stride * h, kps[s].data_ptr(), desc[s].data_ptr(),
                               cnt[s:].data_ptr(), cap, sync=False)
    g.synchronize()
    n = cnt.cpu().numpy()
    kh = kps.cpu().numpy().view(np.uint8).reshape(4, cap, 28)
    dh = desc.cpu().numpy()
    for f in range(4):
        kr, dr = o.extract(np.ascontiguousarray(host[f, :, :w]))
        assert n[f] == len(kr) and len(kr) > 50
        assert np.array_equal(kh[f, :n[f]].copy().view(capi.KP_DTYPE).reshape(-1), kr)
        assert np.array_equal(dh[f, :n[f]], dr)


@pytest.mark.parametrize("w,h,scale,nlevels", [(1434, 605, 2.0, 4), (1245, 574, 2.0, 3), (1400, 1000, 3.0, 3), (1300, 900, 2.5, 3)])
def test_resize_ratios_outside_the_register_windows(w, h, scale, nlevels):
    """Level sizes are rounded, so a scale factor of 2 can give a level ratio slightly above 2 (1434 -> 717 -> 358); those
    and factors of 2.5 / 3 go through the gather variant of the resize kernel.  Every stage bit-exact."""
    img = synth.textured_image(w, h, synth.BASE_SEED + w)
    assert _stage_check(img, 800, scale, nlevels) > 100


def test_two_extractors_from_two_host_threads():
    """The reference runs the left and right extractor objects in two threads (src/Frame.cc:78-81).  Distinct handles are
    independent (own stream, own workspace, thread-local error string): concurrent calls give the single-threaded results."""
    import threading
    left, right = synth.stereo_pair(1241, 376, synth.BASE_SEED + 404)
    imgs = [left, right]
    ex = [capi.ORBextractor(2000, 1.2, 8, 20, 7) for _ in range(2)]
    ref = [O.OracleExtractor(2000, 1.2, 8, 20, 7).extract(im) for im in imgs]
    errors = []

    def work(i):
        try:
            for _ in range(25):
                k, d = ex[i](imgs[i])
                if not (np.array_equal(k, ref[i][0]) and np.array_equal(d, ref[i][1])):
                    errors.append((i, "mismatch"))
                    return
        except Exception as e:      # noqa: BLE001 - reported below
            errors.append((i, repr(e)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("pinned", [False, True])
def test_extract_sequence_host_to_host(pinned):
    """orbfe_extract_sequence: a sequence in host memory, chunks of 8 with two in flight and a ragged last chunk; every
    frame bit-exact against the oracle.  Second sequence (the configuration of the dense-noise test above): uniform-noise
    frames between ordinary ones make the FAST survivor list overflow, and the whole sequence is run again with the
    large-list kernel variant inside the call."""
    torch = pytest.importorskip("torch")
    for (w, h, n, cap, chunk, params, noise) in [(640, 480, 21, 1100, 8, (1000, 1.2, 8, 20, 7), ()),
                                                 (1491, 344, 5, 200, 2, (100, 2.0, 3, 35, 2), (1, 3))]:
        stride = (w + 63) // 64 * 64
        t = torch.zeros((n, h, stride), dtype=torch.uint8)
        kps = torch.zeros((n, cap, 7), dtype=torch.float32)
        desc = torch.zeros((n, cap, 32), dtype=torch.uint8)
        cnt = torch.zeros(n, dtype=torch.int32)
        if pinned:
            t, kps, desc, cnt = t.pin_memory(), kps.pin_memory(), desc.pin_memory(), cnt.pin_memory()
        host = t.numpy()
        for f in range(n):
            host[f, :, :w] = (np.random.default_rng(9176 + f).integers(0, 256, (h, w)).astype(np.uint8) if f in noise
                              else synth.textured_image(w, h, synth.BASE_SEED + 400 + f))
        o = O.OracleExtractor(*params)
        g = capi.ORBextractor(*params)
        g.extract_sequence_host(t.data_ptr(), n, w, h, stride, stride * h, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap, chunk=chunk)
        kh = kps.numpy().view(np.uint8).reshape(n, cap, 28)
        for f in range(n):
            kr, dr = o.extract(np.ascontiguousarray(host[f, :, :w]))
            if f in noise:
                assert sum(len(o.candidates(l)) for l in range(params[2])) > 40000
            assert int(cnt[f]) == len(kr), (w, f)
            assert np.array_equal(kh[f, :len(kr)].copy().view(capi.KP_DTYPE).reshape(-1), kr), (w, f)
            assert np.array_equal(desc.numpy()[f, :len(kr)], dr), (w, f)
    with pytest.raises(capi.OrbfeError):
        g.extract_sequence_host(t.data_ptr(), n, w, h, stride + 4, stride * h, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap)


def _libm_cos_sin(ang_f32):
    """(cosf, sinf) of this host's libm -- what the reference's computeOrbDescriptor calls (src/ORBextractor.cc:112-113)."""
    import ctypes
    import ctypes.util
    m = ctypes.CDLL(ctypes.util.find_library("m"))
    for f in (m.cosf, m.sinf):
        f.restype, f.argtypes = ctypes.c_float, [ctypes.c_float]
    return np.array([(m.cosf(float(a)), m.sinf(float(a))) for a in ang_f32], np.float32)


def test_rotation_known_answers():
    """The descriptor rotation (a, b) = (cosf, sinf)(angle * factorPI) on the device against libm, bit for bit:
    (1) every angle of tests/golden/sincos_census.json -- the 96 angles where round 1's faster routine moved a pattern tap
    and a sample of 1450 of the 1 484 894 angles where it produced a different float (tools/sincos_census.c, all
    1 135 869 952 floats in [0, 360) enumerated; the routine in use now differs for none of them on the host mirror);
    (2) edge angles and 200 000 random floats of [0, 360) against this host's libm directly."""
    import json
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sincos_census.json")))
    assert gold["stride"] == 1 and gold["angles_evaluated"] == gold["floats_in_0_360"] == 1135869952
    assert gold["glibc"] == {"angles_where_cos_or_sin_differs": 0, "cos_differs": 0, "sin_differs": 0}
    rows = np.array(gold["tap_moving_angles"] + gold["sample_of_differing_angles"], np.uint32)
    assert len(gold["tap_moving_angles"]) == gold["taylor"]["angles_where_a_tap_moves"] == 96
    g = capi.ORBextractor(1000, 1.2, 8, 20, 7)
    got = g.debug_rotation(rows[:, 0].copy().view(np.float32))
    assert np.array_equal(got.view(np.uint32), rows[:, 1:3]), "device rotation differs from the census' libm values"

    rng = np.random.default_rng(20240607)
    edge = np.array([0.0, 1e-30, 1e-6, 0.0139, 0.014, 44.999996, 45.0, 45.000004, 89.99999, 90.0, 90.00001, 135.0, 179.99998, 180.0,
                     180.00002, 225.0, 269.99997, 270.0, 270.00003, 315.0, 359.99997], np.float32)
    ang = np.concatenate([edge, rng.integers(0, 1135869952, 200000, dtype=np.uint32).view(np.float32),
                          rng.uniform(0, 360, 50000).astype(np.float32)])
    factor_pi = np.float32(3.1415926535897932384626433832795 / np.float32(180.0))
    want = _libm_cos_sin((ang * factor_pi).astype(np.float32))
    got = g.debug_rotation(ang)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_extract_sequence_4541_frames():
    """BASELINE configs[3] at its full length through orbfe_extract_sequence: 4541 KITTI-sized frames in host memory, eight
    distinct images repeated.  The first period is checked against the oracle frame by frame; every later frame must then
    reproduce the frame eight before it exactly (count, keypoint bytes, descriptor bytes) -- any cross-frame leak in the
    chunked, double-buffered pipeline (a ragged last chunk included: 4541 = 70 * 64 + 61) breaks the periodicity."""
    import hashlib
    torch = pytest.importorskip("torch")
    n, w, h, cap, period = 4541, 1241, 376, 2200, 8
    stride = 1280
    t = torch.zeros((n, h, stride), dtype=torch.uint8).pin_memory()
    host = t.numpy()
    base = [synth.textured_image(w, h, synth.BASE_SEED + 900 + f) for f in range(period)]
    for f in range(n):
        host[f, :, :w] = base[f % period]
    kps = torch.zeros((n, cap, 7), dtype=torch.float32).pin_memory()
    desc = torch.zeros((n, cap, 32), dtype=torch.uint8).pin_memory()
    cnt = torch.zeros(n, dtype=torch.int32).pin_memory()
    g = capi.ORBextractor(2000, 1.2, 8, 20, 7)
    g.extract_sequence_host(t.data_ptr(), n, w, h, stride, stride * h, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap, chunk=64)
    c = cnt.numpy()
    kb = kps.numpy().view(np.uint8).reshape(n, cap, 28)
    db = desc.numpy()
    o = O.OracleExtractor(2000, 1.2, 8, 20, 7)
    for f in range(period):
        kr, dr = o.extract(base[f])
        assert int(c[f]) == len(kr), f
        assert np.array_equal(kb[f, :len(kr)].copy().view(capi.KP_DTYPE).reshape(-1), kr), f
        assert np.array_equal(db[f, :len(kr)], dr), f
    assert 1500 < int(c[:period].min()) and int(c.max()) <= cap
    assert np.array_equal(c, np.tile(c[:period], n // period + 1)[:n]), "keypoint counts are not periodic"

    def sha(f):
        m = hashlib.sha256()
        m.update(kb[f, :c[f]].tobytes())
        m.update(db[f, :c[f]].tobytes())
        return m.digest()

    first = [sha(f) for f in range(period)]
    assert len(set(first)) == period
    bad = [f for f in range(period, n) if sha(f) != first[f % period]]
    assert not bad, f"{len(bad)} frames differ from their first occurrence, first: {bad[:5]}"
