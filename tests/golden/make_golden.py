"""Writes tests/golden/front_end.json from the CPU oracle (run in the build container: python tests/golden/make_golden.py).
The cases and digests are defined in tests/golden_cases.py; tests/test_golden.py checks that the oracle still reproduces
the committed file (CPU) and that the HIP path produces the same digests without the oracle in the loop (GPU)."""
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
sys.path.insert(0, str(HERE.parent.parent))

import golden_cases as G   # noqa: E402
import oracle_lib as O     # noqa: E402


def oracle_digests():
    out = {"extract": {}}
    kitti = None
    for name in G.EXTRACT_CASES:
        img, c = G.extract_inputs(name)
        o = O.OracleExtractor(c["nfeatures"], 1.2, 8, 20, 7)
        k, d = o.extract(img)
        out["extract"][name] = G.extract_digest(img, [o.level(l) for l in range(8)], [o.blurred(l) for l in range(8)],
                                                [o.candidates(l) for l in range(8)], [o.selected(l) for l in range(8)], k, d)
        if name == "kitti_1241x376_2000":
            kitti = (k, d)
    left, right = G.stereo_inputs()
    s = G.STEREO
    ol, orr = O.OracleExtractor(s["nfeatures"], 1.2, 8, 20, 7), O.OracleExtractor(s["nfeatures"], 1.2, 8, 20, 7)
    kl, dl = ol.extract(left)
    kr, dr = orr.extract(right)
    ur, dep = O.stereo_full(ol, orr, kl, dl, kr, dr, s["bf"], s["b"])
    out["stereo"] = G.stereo_digest(kl, kr, ur, dep)
    q, db = G.knn_inputs()
    out["knn2"] = G.knn_digest(*O.knn2(q, db))
    (co, ci, nd, wi, w), f = G.bow_inputs()
    out["bow_descend"] = G.bow_digest(*O.bow_descend(co, ci, nd, G.BOW["L"], f, G.BOW["levelsup"]))
    desc, off = G.distinctive_inputs()
    out["distinctive"] = G.distinctive_digest(O.distinctive(desc, off))
    grid, aq, qd, blocked = G.area_inputs(*kitti)
    og = O.OracleGrid(grid["kx"], grid["ky"], grid["koct"], kitti[1], grid["ur"], grid["min_x"], grid["min_y"], grid["w_inv"],
                      grid["h_inv"])
    rows = [og.search(float(aq["x"][i]), float(aq["y"][i]), float(aq["r"][i]), int(aq["min_level"][i]), int(aq["max_level"][i]),
                      float(aq["proj_xr"][i]), float(aq["er_max"][i]), qd[i], blocked) for i in range(len(qd))]
    out["search_area"] = G.area_digest(np.array(rows, np.int32))
    d1, v1, d2, b2, (o1, i1), (o2, i2) = G.bowmatch_inputs()
    out["search_by_bow"] = G.bowmatch_digest(O.search_by_bow(d1, v1, d2, b2, o1, i1, o2, i2, G.BOWMATCH["max_dist"], G.BOWMATCH["ratio"]))
    grid, fq, fqd, inv_sigma2, limit = G.fuse_inputs(*kitti)
    og = O.OracleGrid(grid["kx"], grid["ky"], grid["koct"], kitti[1], grid["ur"], grid["min_x"], grid["min_y"], grid["w_inv"],
                      grid["h_inv"])
    best, _ = O.fuse_select(og, inv_sigma2, 1, fq["x"], fq["y"], fq["proj_xr"], fq["r"], fq["max_level"], fqd)
    out["fuse"] = G.fuse_digest(best, limit_rows(og, fq, fqd, limit))
    t = G.triangulation_inputs(*kitti)
    zeros1, zeros2 = np.zeros(len(t["d1"]), np.float32), np.zeros(len(t["d2"]), np.float32)
    tri = [O.search_for_triangulation(t["d1"], t["mp1"], t["ur1"], t["xy1"], zeros1, t["d2"], t["mp2"], t["ur2"], t["xy2"], t["oct2"],
                                      zeros2, t["csr1"][0], t["csr1"][1], t["csr2"][0], t["csr2"][1], t["F12"], t["ex"], t["ey"],
                                      t["sf"], t["sigma2"], only, 0)[0] for only in (0, 1)]
    out["triangulation"] = G.triangulation_digest(*tri)
    return out


def limit_rows(og, q, qd, limit):
    """the candidate loop of SearchForInitialization (reference src/ORBmatcher.cc:436-457) per window, with `limit` as
    vMatchedDistance: best / second over the candidates whose limit exceeds their distance, strict '<' in list order"""
    rows = []
    for i in range(len(qd)):
        cand = og.features_in_area(float(q["x"][i]), float(q["y"][i]), float(q["r"][i]), int(q["min_level"][i]), int(q["max_level"][i]))
        bd, bi, bd2, bi2 = 256, -1, 256, -1
        for c in cand:
            d = O.hamming(qd[i], og.kdesc[c])
            if limit[c] <= d:
                continue
            if d < bd:
                bd2, bi2, bd, bi = bd, bi, d, int(c)
            elif d < bd2:
                bd2, bi2 = d, int(c)
        rows.append([bd, bi, bd2, bi2])
    return np.array(rows, np.int32)


if __name__ == "__main__":
    dig = oracle_digests()
    with open(G.GOLDEN, "w") as fh:
        json.dump(dig, fh, indent=1, sort_keys=True)
        fh.write("\n")
    print("wrote", G.GOLDEN, {k: (len(v) if isinstance(v, dict) else v) for k, v in dig.items()})
