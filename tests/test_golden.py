"""Committed golden fixtures (tests/golden/front_end.json, written by tests/golden/make_golden.py from the oracle).
CPU: the oracle still reproduces them.  GPU: liborbfe, through the C-ABI, produces the same digests with no oracle in
the loop -- pyramid and blurred levels, FAST candidates, keypoints, descriptors, stereo depths, kNN-2, BoW descent,
distinctive descriptors, windowed search and the SearchByBoW loops, all bit-exact (SHA-256 / CRC32 of the raw outputs)."""
import importlib.util
from pathlib import Path

import numpy as np
import pytest

import golden_cases as G


def _diff(got, want, path=""):
    """first differing leaf of two nested digests, for a readable failure"""
    if isinstance(want, dict):
        assert isinstance(got, dict) and sorted(got) == sorted(want), (path, sorted(got), sorted(want))
        for k in want:
            _diff(got[k], want[k], f"{path}/{k}")
    elif isinstance(want, list) and want and isinstance(want[0], (dict, list)):
        assert len(got) == len(want), path
        for i, (a, b) in enumerate(zip(got, want)):
            _diff(a, b, f"{path}[{i}]")
    else:
        assert got == want, (path, got, want)


def test_golden_file_is_complete():
    g = G.load()
    assert sorted(g) == ["bow_descend", "distinctive", "extract", "fuse", "knn2", "search_area", "search_by_bow", "stereo",
                         "triangulation"]
    assert sorted(g["extract"]) == sorted(G.EXTRACT_CASES)
    k = g["extract"]["kitti_1241x376_2000"]
    # SURVEY.md section 8a check values: level sizes of 1241x376 at scale 1.2
    assert [l["size"] for l in k["levels"]] == [[1241, 376], [1034, 313], [862, 261], [718, 218], [598, 181], [499, 151],
                                               [416, 126], [346, 105]]
    assert 2000 <= k["keypoints"] <= 2000 + 3 * 8 and sum(l["kept"] for l in k["levels"]) == k["keypoints"]
    assert g["stereo"]["with_depth"] > 200 and g["search_area"]["with_match"] > 300
    assert g["fuse"]["fused"] > 300 and g["triangulation"]["matches"] > g["triangulation"]["matches_only_stereo"] > 30


def test_oracle_reproduces_golden():
    spec = importlib.util.spec_from_file_location("make_golden", Path(__file__).resolve().parent / "golden" / "make_golden.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    _diff(mod.oracle_digests(), G.load())


@pytest.mark.gpu
def test_hip_path_reproduces_golden():
    torch = pytest.importorskip("torch")
    from orb_slam2_amd import capi
    want = G.load()

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).cuda()

    kitti = None
    for name in G.EXTRACT_CASES:
        img, c = G.extract_inputs(name)
        g = capi.ORBextractor(c["nfeatures"], 1.2, 8, 20, 7)
        k, d = g(img)
        sel = [g.debug_stage(l, 3) for l in range(8)]
        got = G.extract_digest(img, [g.debug_stage(l, 0) for l in range(8)],
                               [g.debug_stage(l, 1) if len(sel[l]) else None for l in range(8)],
                               [g.debug_stage(l, 2) for l in range(8)], sel, k, d)
        _diff(got, want["extract"][name], name)
        if name == "kitti_1241x376_2000":
            kitti = (k, d)
    m = capi.ORBmatcher()
    # stereo
    left, right = G.stereo_inputs()
    s = G.STEREO
    gl, gr = capi.ORBextractor(s["nfeatures"], 1.2, 8, 20, 7), capi.ORBextractor(s["nfeatures"], 1.2, 8, 20, 7)
    kl, dl = gl(left)
    kr, dr = gr(right)
    dkl, dkr = dev(kl.view(np.uint8).reshape(-1, 28)), dev(kr.view(np.uint8).reshape(-1, 28))
    ddl, ddr = dev(dl), dev(dr)
    ur = torch.zeros(len(kl), dtype=torch.float32, device="cuda")
    dep = torch.zeros(len(kl), dtype=torch.float32, device="cuda")
    m.stereo_match_device(gl, 0, gr, 0, dkl.data_ptr(), ddl.data_ptr(), len(kl), dkr.data_ptr(), ddr.data_ptr(), len(kr),
                          s["bf"], s["b"], ur.data_ptr(), dep.data_ptr())
    _diff(G.stereo_digest(kl, kr, ur.cpu().numpy(), dep.cpu().numpy()), want["stereo"], "stereo")
    # kNN-2
    q, db = G.knn_inputs()
    dq, ddb = dev(q), dev(db)
    bd, bi, sd = (torch.zeros(len(q), dtype=torch.int32, device="cuda") for _ in range(3))
    m.knn2_device(dq.data_ptr(), len(q), ddb.data_ptr(), len(db), bd.data_ptr(), bi.data_ptr(), sd.data_ptr(), 0)
    _diff(G.knn_digest(bd.cpu().numpy(), bi.cpu().numpy(), sd.cpu().numpy()), want["knn2"], "knn2")
    # BoW descent
    (co, ci, nd, wi, w), f = G.bow_inputs()
    voc = capi.Vocabulary(co, ci, nd, G.BOW["L"])
    df = dev(f)
    leaf = torch.zeros(len(f), dtype=torch.int32, device="cuda")
    nid = torch.zeros(len(f), dtype=torch.int32, device="cuda")
    m.bow_transform_device(voc, df.data_ptr(), len(f), G.BOW["levelsup"], leaf.data_ptr(), nid.data_ptr())
    _diff(G.bow_digest(leaf.cpu().numpy(), nid.cpu().numpy()), want["bow_descend"], "bow")
    # distinctive descriptors
    desc, off = G.distinctive_inputs()
    dd, do = dev(desc), dev(off)
    best = torch.full((len(off) - 1,), -7, dtype=torch.int32, device="cuda")
    m.distinctive_device(dd.data_ptr(), do.data_ptr(), len(off) - 1, int(np.diff(off).max()), best.data_ptr())
    _diff(G.distinctive_digest(best.cpu().numpy()), want["distinctive"], "distinctive")
    # windowed search on the frame grid
    grid, aq, qd, blocked = G.area_inputs(*kitti)
    fg = capi.FrameGrid(grid["kx"], grid["ky"], grid["koct"], kitti[1], grid["ur"], grid["min_x"], grid["min_y"], grid["w_inv"],
                        grid["h_inv"])
    qq = np.zeros(len(qd), capi.AREA_QUERY_DTYPE)
    for fld in ("x", "y", "r", "min_level", "max_level", "proj_xr", "er_max"):
        qq[fld] = aq[fld]
    res = fg.search(m, qq, qd, blocked)
    rows = np.stack([res[fld] for fld in ("best_dist", "best_idx", "best_level", "best_dist2", "best_level2", "n_candidates",
                                          "best_idx2")], axis=1)
    _diff(G.area_digest(rows), want["search_area"], "search_area")
    # SearchByBoW loops
    d1, v1, d2, b2, (o1, i1), (o2, i2) = G.bowmatch_inputs()
    got = m.search_by_bow(d1, v1, d2, b2, o1, i1, o2, i2, G.BOWMATCH["max_dist"], G.BOWMATCH["ratio"])
    _diff(G.bowmatch_digest(got), want["search_by_bow"], "search_by_bow")
    # Fuse selection (reprojection gate) and the vMatchedDistance limit on the window search
    grid, fq, fqd, inv_sigma2, limit = G.fuse_inputs(*kitti)
    fg = capi.FrameGrid(grid["kx"], grid["ky"], grid["koct"], kitti[1], grid["ur"], grid["min_x"], grid["min_y"], grid["w_inv"],
                        grid["h_inv"])
    qq = np.zeros(len(fqd), capi.AREA_QUERY_DTYPE)
    for fld in ("x", "y", "r", "min_level", "max_level", "proj_xr", "er_max"):
        qq[fld] = fq[fld]
    res = fg.search_ex(m, qq, fqd, inv_level_sigma2=inv_sigma2)
    best = np.where(res["best_dist"] <= 50, res["best_idx"], -1)
    res = fg.search_ex(m, qq, fqd, dist_limit=limit)
    rows = np.stack([res[fld] for fld in ("best_dist", "best_idx", "best_dist2", "best_idx2")], axis=1)
    _diff(G.fuse_digest(best, rows), want["fuse"], "fuse")
    # SearchForTriangulation loops
    t = G.triangulation_inputs(*kitti)
    tri = [m.search_for_triangulation(t["d1"], t["mp1"], t["ur1"], t["xy1"], t["d2"], t["mp2"], t["ur2"], t["xy2"], t["oct2"],
                                      t["csr1"][0], t["csr1"][1], t["csr2"][0], t["csr2"][1], t["F12"], t["ex"], t["ey"], t["sf"],
                                      t["sigma2"], only_stereo=bool(only)) for only in (0, 1)]
    _diff(G.triangulation_digest(*tri), want["triangulation"], "triangulation")
