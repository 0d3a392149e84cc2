"""GPU parity of the extractor: liborbfe (HIP, through the C-ABI) against the CPU oracle, stage by stage and end to end.
Bar (BASELINE.json north_star): pyramid levels, FAST corners, kept keypoints, descriptors bit-exact; angles within
1e-3 degrees (in practice identical floats)."""
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


@pytest.mark.parametrize("ini_th,min_th", [(20, 7), (40, 5), (7, 7), (100, 30), (5, 20)])
def test_fast_thresholds(ini_th, min_th):
    """Per-cell iniTh -> minTh fallback for different threshold pairs (incl. ini < min, which the reference also accepts)."""
    img = synth.textured_image(752, 480, synth.BASE_SEED + 500 + ini_th)
    img[:, :200] = (img[:, :200].astype(np.int32) // 8 + 100).astype(np.uint8)      # low-contrast band: cells that fall back
    _stage_check(img, 1000, 1.2, 8, ini_th, min_th)
