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
