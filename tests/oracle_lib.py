"""ctypes binding of oracle/liborb_oracle.so (TEST INFRASTRUCTURE: the checker, never the product path)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
_LIB = None

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

u8p = C.POINTER(C.c_uint8)
f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)
i16p = C.POINTER(C.c_int16)


def _ptr(a, t):
    return a.ctypes.data_as(t)


def lib():
    global _LIB
    if _LIB is None:
        so = Path(os.environ.get("ORB_ORACLE_LIB", ROOT / "oracle" / "liborb_oracle.so"))   # ORB_ORACLE_LIB: `make asan` build
        if not so.exists():
            subprocess.check_call(["make", "-C", str(ROOT / "oracle")])
        L = C.CDLL(str(so))
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_extract.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, u8p, i32p]
        L.orc_nlevels.argtypes = [C.c_void_p]
        L.orc_scale_tables.argtypes = [C.c_void_p, f32p, f32p, f32p, f32p]
        L.orc_features_per_level.argtypes = [C.c_void_p, i32p]
        L.orc_umax.argtypes = [C.c_void_p, i32p]
        L.orc_level_size.argtypes = [C.c_void_p, C.c_int, i32p, i32p]
        L.orc_level_ptr.restype = C.c_void_p
        L.orc_level_ptr.argtypes = [C.c_void_p, C.c_int]
        L.orc_blur_ptr.restype = C.c_void_p
        L.orc_blur_ptr.argtypes = [C.c_void_p, C.c_int]
        L.orc_level_candidates.argtypes = [C.c_void_p, C.c_int, f32p, C.c_int]
        L.orc_level_selected.argtypes = [C.c_void_p, C.c_int, f32p, C.c_int]
        L.orc_resize_linear_u8.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int, C.c_int, C.c_int]
        L.orc_resize_tables.argtypes = [C.c_int, C.c_int, i32p, i16p]
        L.orc_gauss7_u8.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int]
        L.orc_gauss7_kernel.argtypes = [i32p]
        L.orc_fast9.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_int]
        L.orc_fast_cells.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_int]
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_ic_angle.restype = C.c_float
        L.orc_ic_angle.argtypes = [u8p, C.c_int, C.c_int, C.c_int, i32p]
        L.orc_descriptor.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_float, u8p]
        L.orc_distribute_octree.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_int]
        L.orc_cv_round.argtypes = [C.c_double]
        L.orc_hamming.argtypes = [u8p, u8p]
        L.orc_knn2.argtypes = [u8p, C.c_int, u8p, C.c_int, i32p, i32p, i32p]
        L.orc_match_csr.argtypes = [u8p, C.c_int, u8p, i32p, i32p, i32p, i32p, i32p]
        L.orc_stereo_rowband.argtypes = [C.c_void_p, u8p, C.c_int, C.c_void_p, u8p, C.c_int, f32p, C.c_int, C.c_float,
                                         C.c_float, i32p, i32p]
        L.orc_stereo_full.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, u8p, C.c_int, C.c_void_p, u8p, C.c_int,
                                      C.c_float, C.c_float, f32p, f32p]
        L.orc_grid_create.restype = C.c_void_p
        L.orc_grid_create.argtypes = [f32p, f32p, i32p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
        L.orc_grid_destroy.argtypes = [C.c_void_p]
        L.orc_grid_destroy.restype = None
        L.orc_grid_set_query_origin.argtypes = [C.c_void_p, C.c_float, C.c_float]
        L.orc_grid_set_query_origin.restype = None
        L.orc_features_in_area.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, i32p]
        L.orc_search_area.argtypes = [C.c_void_p, u8p, f32p, u8p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float,
                                      C.c_float, u8p, i32p]
        L.orc_search_area.restype = None
        L.orc_search_by_bow.argtypes = [u8p, u8p, C.c_int, u8p, u8p, C.c_int, i32p, i32p, i32p, i32p, C.c_int, C.c_int, C.c_float, i32p]
        L.orc_search_by_bow.restype = None
        L.orc_three_maxima.argtypes = [i32p, C.c_int, i32p]
        L.orc_three_maxima.restype = None
        L.orc_rotation_filter.argtypes = [f32p, f32p, i32p, C.c_int]
        L.orc_rotation_filter.restype = C.c_int
        vp_, ci, cf = C.c_void_p, C.c_int, C.c_float
        L.orc_search_by_projection_sim3.argtypes = [vp_, u8p, ci, f32p, f32p, f32p, i32p, u8p, u8p, i32p]
        L.orc_search_by_projection_reloc.argtypes = [vp_, u8p, f32p, ci, f32p, f32p, f32p, i32p, f32p, u8p, ci, ci, u8p, i32p]
        L.orc_fuse_select.argtypes = [vp_, u8p, f32p, f32p, ci, ci, f32p, f32p, f32p, f32p, i32p, u8p, i32p]
        L.orc_search_by_sim3.argtypes = [vp_, u8p, vp_, u8p, ci, i32p, f32p, f32p, f32p, i32p, u8p, ci, i32p, f32p, f32p, f32p, i32p,
                                         u8p, i32p]
        L.orc_search_for_initialization.argtypes = [vp_, u8p, f32p, ci, i32p, f32p, u8p, f32p, ci, cf, ci, i32p]
        L.orc_search_for_triangulation.argtypes = [u8p, u8p, f32p, f32p, f32p, ci, u8p, u8p, f32p, f32p, i32p, f32p, ci, i32p, i32p,
                                                   i32p, i32p, ci, f32p, cf, cf, f32p, f32p, ci, ci, i32p]
        L.orc_distinctive.argtypes = [u8p, i32p, C.c_int, i32p]
        L.orc_distinctive.restype = None
        L.orc_bow_descend.argtypes = [i32p, i32p, u8p, C.c_int, u8p, C.c_int, C.c_int, i32p, i32p]
        L.orc_bow_descend.restype = None
        L.orc_bow_transform.argtypes = [i32p, i32p, u8p, i32p, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, u8p, C.c_int,
                                        C.c_int, i32p, C.POINTER(C.c_double), C.POINTER(C.c_int32), i32p, i32p,
                                        C.POINTER(C.c_int32)]
        L.orc_bow_transform.restype = None
        L.orc_time_extract.restype = C.c_double
        L.orc_time_extract.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, u8p, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_long, C.c_int, C.c_int, C.POINTER(C.c_long)]
        L.orc_time_knn2.restype = C.c_double
        L.orc_time_knn2.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int]
        _LIB = L
    return _LIB


class OracleExtractor:
    """Mirror of ORB_SLAM2::ORBextractor on the oracle."""

    def __init__(self, nfeatures=2000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.L = lib()
        self.nfeatures, self.nlevels = nfeatures, nlevels
        self.h = self.L.orc_create(nfeatures, scale_factor, nlevels, ini_th, min_th)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def tables(self):
        n = self.nlevels
        out = [np.zeros(n, np.float32) for _ in range(4)]
        self.L.orc_scale_tables(self.h, *[_ptr(o, f32p) for o in out])
        fpl = np.zeros(n, np.int32)
        self.L.orc_features_per_level(self.h, _ptr(fpl, i32p))
        um = np.zeros(16, np.int32)
        self.L.orc_umax(self.h, _ptr(um, i32p))
        return dict(scale=out[0], inv_scale=out[1], sigma2=out[2], inv_sigma2=out[3], features_per_level=fpl, umax=um)

    def extract(self, img: np.ndarray):
        assert img.dtype == np.uint8 and img.ndim == 2
        img = np.ascontiguousarray(img)
        cap = self.nfeatures * 2 + 256
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int32(0)
        rc = self.L.orc_extract(self.h, _ptr(img, u8p), img.shape[1], img.shape[0], img.strides[0],
                                kps.ctypes.data_as(C.c_void_p), cap, _ptr(desc, u8p), C.byref(n))
        assert rc == 0, rc
        return kps[:n.value].copy(), desc[:n.value].copy()

    def level(self, lvl):
        w, h = C.c_int32(), C.c_int32()
        assert self.L.orc_level_size(self.h, lvl, C.byref(w), C.byref(h)) == 0
        p = self.L.orc_level_ptr(self.h, lvl)
        return np.ctypeslib.as_array(C.cast(p, u8p), shape=(h.value, w.value)).copy()

    def blurred(self, lvl):
        w, h = C.c_int32(), C.c_int32()
        assert self.L.orc_level_size(self.h, lvl, C.byref(w), C.byref(h)) == 0
        p = self.L.orc_blur_ptr(self.h, lvl)
        if not p:
            return None
        return np.ctypeslib.as_array(C.cast(p, u8p), shape=(h.value, w.value)).copy()

    def candidates(self, lvl):
        n = self.L.orc_level_candidates(self.h, lvl, None, 0)
        out = np.zeros((max(n, 1), 3), np.float32)
        self.L.orc_level_candidates(self.h, lvl, _ptr(out, f32p), n)
        return out[:n]

    def selected(self, lvl):
        n = self.L.orc_level_selected(self.h, lvl, None, 0)
        out = np.zeros((max(n, 1), 4), np.float32)
        self.L.orc_level_selected(self.h, lvl, _ptr(out, f32p), n)
        return out[:n]


def resize(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    src = np.ascontiguousarray(src)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_ptr(src, u8p), src.shape[1], src.shape[0], src.strides[0], _ptr(dst, u8p), dw, dh, dw)
    return dst


def resize_tables(ssize: int, dsize: int):
    ofs = np.zeros(dsize, np.int32)
    coef = np.zeros((dsize, 2), np.int16)
    lib().orc_resize_tables(ssize, dsize, _ptr(ofs, i32p), _ptr(coef, i16p))
    return ofs, coef


def gauss7(src: np.ndarray) -> np.ndarray:
    src = np.ascontiguousarray(src)
    dst = np.zeros_like(src)
    lib().orc_gauss7_u8(_ptr(src, u8p), src.shape[1], src.shape[0], src.strides[0], _ptr(dst, u8p), dst.strides[0])
    return dst


def gauss7_kernel():
    k = np.zeros(7, np.int32)
    lib().orc_gauss7_kernel(_ptr(k, i32p))
    return k


def fast9(win: np.ndarray, threshold: int) -> np.ndarray:
    win = np.ascontiguousarray(win)
    cap = win.size
    out = np.zeros((max(cap, 1), 3), np.float32)
    n = lib().orc_fast9(_ptr(win, u8p), win.shape[1], win.shape[0], win.strides[0], threshold, _ptr(out, f32p), cap)
    return out[:n]


def fast_cells(img: np.ndarray, ini_th=20, min_th=7) -> np.ndarray:
    img = np.ascontiguousarray(img)
    cap = img.size // 4 + 16
    out = np.zeros((cap, 3), np.float32)
    n = lib().orc_fast_cells(_ptr(img, u8p), img.shape[1], img.shape[0], img.strides[0], ini_th, min_th,
                             _ptr(out, f32p), cap)
    assert n <= cap
    return out[:n]


def fast_atan2(y, x) -> float:
    return float(lib().orc_fast_atan2(float(y), float(x)))


def ic_angle(img: np.ndarray, cx: int, cy: int, umax: np.ndarray) -> float:
    img = np.ascontiguousarray(img)
    umax = np.ascontiguousarray(umax, np.int32)
    return float(lib().orc_ic_angle(_ptr(img, u8p), img.strides[0], cx, cy, _ptr(umax, i32p)))


def descriptor(blurred: np.ndarray, cx: int, cy: int, angle_deg: float) -> np.ndarray:
    blurred = np.ascontiguousarray(blurred)
    d = np.zeros(32, np.uint8)
    lib().orc_descriptor(_ptr(blurred, u8p), blurred.strides[0], cx, cy, float(angle_deg), _ptr(d, u8p))
    return d


def distribute_octree(xyr: np.ndarray, minX, maxX, minY, maxY, N) -> np.ndarray:
    xyr = np.ascontiguousarray(xyr, np.float32).reshape(-1, 3)
    cap = len(xyr) + 8
    out = np.zeros((cap, 3), np.float32)
    n = lib().orc_distribute_octree(_ptr(xyr, f32p), len(xyr), minX, maxX, minY, maxY, N, _ptr(out, f32p), cap)
    return out[:n]


def hamming(a: np.ndarray, b: np.ndarray) -> int:
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return int(lib().orc_hamming(_ptr(a, u8p), _ptr(b, u8p)))


def knn2(q: np.ndarray, r: np.ndarray):
    q = np.ascontiguousarray(q, np.uint8)
    r = np.ascontiguousarray(r, np.uint8)
    nq, nr = len(q), len(r)
    bd, bi, sd = (np.zeros(nq, np.int32) for _ in range(3))
    lib().orc_knn2(_ptr(q, u8p), nq, _ptr(r, u8p), nr, _ptr(bd, i32p), _ptr(bi, i32p), _ptr(sd, i32p))
    return bd, bi, sd


def match_csr(q: np.ndarray, r: np.ndarray, off: np.ndarray, idx: np.ndarray):
    q = np.ascontiguousarray(q, np.uint8)
    r = np.ascontiguousarray(r, np.uint8)
    off = np.ascontiguousarray(off, np.int32)
    idx = np.ascontiguousarray(idx, np.int32)
    nq = len(q)
    bd, bi, sd = (np.zeros(nq, np.int32) for _ in range(3))
    lib().orc_match_csr(_ptr(q, u8p), nq, _ptr(r, u8p), _ptr(off, i32p), _ptr(idx, i32p), _ptr(bd, i32p), _ptr(bi, i32p),
                        _ptr(sd, i32p))
    return bd, bi, sd


def stereo_rowband(kl, dl, kr, dr, scale_factors, nrows, min_d, max_d):
    kl = np.ascontiguousarray(kl)
    kr = np.ascontiguousarray(kr)
    dl = np.ascontiguousarray(dl, np.uint8)
    dr = np.ascontiguousarray(dr, np.uint8)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    bd, bi = np.zeros(len(kl), np.int32), np.zeros(len(kl), np.int32)
    lib().orc_stereo_rowband(kl.ctypes.data_as(C.c_void_p), _ptr(dl, u8p), len(kl), kr.ctypes.data_as(C.c_void_p),
                             _ptr(dr, u8p), len(kr), _ptr(sf, f32p), nrows, min_d, max_d, _ptr(bd, i32p), _ptr(bi, i32p))
    return bd, bi


def stereo_full(el: OracleExtractor, er: OracleExtractor, kl, dl, kr, dr, bf, b):
    kl = np.ascontiguousarray(kl)
    kr = np.ascontiguousarray(kr)
    dl = np.ascontiguousarray(dl, np.uint8)
    dr = np.ascontiguousarray(dr, np.uint8)
    ur, dep = np.zeros(len(kl), np.float32), np.zeros(len(kl), np.float32)
    lib().orc_stereo_full(el.h, er.h, kl.ctypes.data_as(C.c_void_p), _ptr(dl, u8p), len(kl),
                          kr.ctypes.data_as(C.c_void_p), _ptr(dr, u8p), len(kr), bf, b, _ptr(ur, f32p), _ptr(dep, f32p))
    return ur, dep


def bow_descend(child_off, child_idx, node_desc, L, desc, levelsup):
    co, ci = np.ascontiguousarray(child_off, np.int32), np.ascontiguousarray(child_idx, np.int32)
    nd, d = np.ascontiguousarray(node_desc, np.uint8), np.ascontiguousarray(desc, np.uint8)
    n = len(d)
    leaf, nid = np.zeros(n, np.int32), np.zeros(n, np.int32)
    lib().orc_bow_descend(_ptr(co, i32p), _ptr(ci, i32p), _ptr(nd, u8p), L, _ptr(d, u8p), n, levelsup, _ptr(leaf, i32p),
                          _ptr(nid, i32p))
    return leaf, nid


def bow_transform(child_off, child_idx, node_desc, word_id, weight, L, weighting, norm, desc, levelsup):
    """-> (bow_ids, bow_vals), (fv_node, fv_feat): BowVector and flattened FeatureVector in map order."""
    co, ci = np.ascontiguousarray(child_off, np.int32), np.ascontiguousarray(child_idx, np.int32)
    nd, d = np.ascontiguousarray(node_desc, np.uint8), np.ascontiguousarray(desc, np.uint8)
    wi, w = np.ascontiguousarray(word_id, np.int32), np.ascontiguousarray(weight, np.float64)
    n = len(d)
    bi, bv = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.float64)
    fn, ff = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
    nb, nf = C.c_int32(0), C.c_int32(0)
    lib().orc_bow_transform(_ptr(co, i32p), _ptr(ci, i32p), _ptr(nd, u8p), _ptr(wi, i32p),
                            w.ctypes.data_as(C.POINTER(C.c_double)), L, weighting, norm, _ptr(d, u8p), n, levelsup,
                            _ptr(bi, i32p), bv.ctypes.data_as(C.POINTER(C.c_double)), C.byref(nb), _ptr(fn, i32p),
                            _ptr(ff, i32p), C.byref(nf))
    return (bi[:nb.value].copy(), bv[:nb.value].copy()), (fn[:nf.value].copy(), ff[:nf.value].copy())


def distinctive(desc, off):
    d = np.ascontiguousarray(desc, np.uint8)
    o = np.ascontiguousarray(off, np.int32)
    best = np.zeros(len(o) - 1, np.int32)
    lib().orc_distinctive(_ptr(d, u8p), _ptr(o, i32p), len(o) - 1, _ptr(best, i32p))
    return best


class OracleGrid:
    """Frame::AssignFeaturesToGrid + GetFeaturesInArea + the SearchByProjection candidate loop on the oracle."""

    def __init__(self, kx, ky, koct, kdesc, uright, min_x, min_y, width_inv, height_inv):
        self.kx, self.ky = np.ascontiguousarray(kx, np.float32), np.ascontiguousarray(ky, np.float32)
        self.koct, self.kdesc = np.ascontiguousarray(koct, np.int32), np.ascontiguousarray(kdesc, np.uint8)
        self.uright = None if uright is None else np.ascontiguousarray(uright, np.float32)
        self.h = lib().orc_grid_create(_ptr(self.kx, f32p), _ptr(self.ky, f32p), _ptr(self.koct, i32p), len(self.kx), min_x, min_y,
                                       width_inv, height_inv)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_grid_destroy(self.h)
            self.h = None

    def set_query_origin(self, min_x, min_y):
        """windows are computed from this origin (a KeyFrame's int mnMinX / mnMinY), cells keep the Frame's"""
        lib().orc_grid_set_query_origin(self.h, float(min_x), float(min_y))
        return self

    def features_in_area(self, x, y, r, min_level, max_level):
        out = np.zeros(len(self.kx) + 1, np.int32)
        n = lib().orc_features_in_area(self.h, x, y, r, min_level, max_level, _ptr(out, i32p))
        return out[:n].copy()

    def search(self, x, y, r, min_level, max_level, proj_xr, er_max, qdesc, blocked=None):
        res = np.zeros(7, np.int32)
        q = np.ascontiguousarray(qdesc, np.uint8)
        b = None if blocked is None else np.ascontiguousarray(blocked, np.uint8)
        lib().orc_search_area(self.h, _ptr(self.kdesc, u8p), None if self.uright is None else _ptr(self.uright, f32p),
                              None if b is None else _ptr(b, u8p), x, y, r, min_level, max_level, proj_xr, er_max, _ptr(q, u8p),
                              _ptr(res, i32p))
        return res


def _c(a, dt):
    return np.ascontiguousarray(a, dt)


def search_by_projection_sim3(g: "OracleGrid", u, v, radius, level, qdesc, matched):
    """SearchByProjection(KeyFrame*, Scw, ...) (src/ORBmatcher.cc:359-399); returns (match_idx, matched_after, nmatches)."""
    u, v, radius, level, qd = _c(u, np.float32), _c(v, np.float32), _c(radius, np.float32), _c(level, np.int32), _c(qdesc, np.uint8)
    m = _c(matched, np.uint8).copy()
    out = np.zeros(max(len(u), 1), np.int32)
    n = lib().orc_search_by_projection_sim3(g.h, _ptr(g.kdesc, u8p), len(u), _ptr(u, f32p), _ptr(v, f32p), _ptr(radius, f32p),
                                            _ptr(level, i32p), _ptr(qd, u8p), _ptr(m, u8p), _ptr(out, i32p))
    return out[:len(u)], m, n


def search_by_projection_reloc(g: "OracleGrid", kangle, u, v, radius, level, qangle, qdesc, orb_dist, check_ori, occupied):
    """SearchByProjection(CurrentFrame, KeyFrame*, sAlreadyFound, th, ORBdist) (:1525-1598); returns (assigned, occupied, n)."""
    ka, u, v, radius = _c(kangle, np.float32), _c(u, np.float32), _c(v, np.float32), _c(radius, np.float32)
    level, qa, qd = _c(level, np.int32), _c(qangle, np.float32), _c(qdesc, np.uint8)
    occ = _c(occupied, np.uint8).copy()
    out = np.zeros(max(len(g.kx), 1), np.int32)
    n = lib().orc_search_by_projection_reloc(g.h, _ptr(g.kdesc, u8p), _ptr(ka, f32p), len(u), _ptr(u, f32p), _ptr(v, f32p),
                                             _ptr(radius, f32p), _ptr(level, i32p), _ptr(qa, f32p), _ptr(qd, u8p), int(orb_dist),
                                             int(check_ori), _ptr(occ, u8p), _ptr(out, i32p))
    return out[:len(g.kx)], occ, n


def fuse_select(g: "OracleGrid", inv_level_sigma2, use_chi2, u, v, ur, radius, level, qdesc):
    """selection part of both Fuse overloads (:889-952 / :1048-1082); returns (best_idx, nFused)."""
    s2, u, v, ur = _c(inv_level_sigma2, np.float32), _c(u, np.float32), _c(v, np.float32), _c(ur, np.float32)
    radius, level, qd = _c(radius, np.float32), _c(level, np.int32), _c(qdesc, np.uint8)
    out = np.zeros(max(len(u), 1), np.int32)
    n = lib().orc_fuse_select(g.h, _ptr(g.kdesc, u8p), None if g.uright is None else _ptr(g.uright, f32p), _ptr(s2, f32p),
                              int(use_chi2), len(u), _ptr(u, f32p), _ptr(v, f32p), _ptr(ur, f32p), _ptr(radius, f32p),
                              _ptr(level, i32p), _ptr(qd, u8p), _ptr(out, i32p))
    return out[:len(u)], n


def search_by_sim3(g1: "OracleGrid", g2: "OracleGrid", q1, q2):
    """SearchBySim3 (:1188-1325); q = (index, u, v, radius, level, qdesc) per direction; returns (match12, nFound)."""
    def prep(q):
        return (_c(q[0], np.int32), _c(q[1], np.float32), _c(q[2], np.float32), _c(q[3], np.float32), _c(q[4], np.int32),
                _c(q[5], np.uint8))
    a, b = prep(q1), prep(q2)
    out = np.zeros(max(len(g1.kx), 1), np.int32)
    n = lib().orc_search_by_sim3(g1.h, _ptr(g1.kdesc, u8p), g2.h, _ptr(g2.kdesc, u8p), len(a[0]), _ptr(a[0], i32p), _ptr(a[1], f32p),
                                 _ptr(a[2], f32p), _ptr(a[3], f32p), _ptr(a[4], i32p), _ptr(a[5], u8p), len(b[0]), _ptr(b[0], i32p),
                                 _ptr(b[1], f32p), _ptr(b[2], f32p), _ptr(b[3], f32p), _ptr(b[4], i32p), _ptr(b[5], u8p),
                                 _ptr(out, i32p))
    return out[:len(g1.kx)], n


def search_for_initialization(g2: "OracleGrid", kangle2, oct1, angle1, desc1, prev_xy, window_size, nn_ratio, check_ori):
    """SearchForInitialization (:405-520); returns (vnMatches12, prev_xy_after, nmatches)."""
    ka, o1, a1, d1 = _c(kangle2, np.float32), _c(oct1, np.int32), _c(angle1, np.float32), _c(desc1, np.uint8)
    pxy = _c(prev_xy, np.float32).copy()
    out = np.zeros(max(len(o1), 1), np.int32)
    n = lib().orc_search_for_initialization(g2.h, _ptr(g2.kdesc, u8p), _ptr(ka, f32p), len(o1), _ptr(o1, i32p), _ptr(a1, f32p),
                                            _ptr(d1, u8p), _ptr(pxy, f32p), int(window_size), float(nn_ratio), int(check_ori),
                                            _ptr(out, i32p))
    return out[:len(o1)], pxy, n


def search_for_triangulation(desc1, has_mp1, uright1, xy1, angle1, desc2, has_mp2, uright2, xy2, oct2, angle2, off1, idx1, off2,
                             idx2, F12, ex, ey, scale_factors, level_sigma2, only_stereo, check_ori):
    """SearchForTriangulation (:657-823); returns (vMatches12, nmatches)."""
    d1, d2, h1, h2 = _c(desc1, np.uint8), _c(desc2, np.uint8), _c(has_mp1, np.uint8), _c(has_mp2, np.uint8)
    u1, u2, x1, x2 = _c(uright1, np.float32), _c(uright2, np.float32), _c(xy1, np.float32), _c(xy2, np.float32)
    a1, a2, o2 = _c(angle1, np.float32), _c(angle2, np.float32), _c(oct2, np.int32)
    f1, i1, f2, i2 = (_c(a, np.int32) for a in (off1, idx1, off2, idx2))
    F, sf, s2 = _c(F12, np.float32).reshape(9), _c(scale_factors, np.float32), _c(level_sigma2, np.float32)
    out = np.zeros(max(len(d1), 1), np.int32)
    n = lib().orc_search_for_triangulation(_ptr(d1, u8p), _ptr(h1, u8p), _ptr(u1, f32p), _ptr(x1, f32p), _ptr(a1, f32p), len(d1),
                                           _ptr(d2, u8p), _ptr(h2, u8p), _ptr(u2, f32p), _ptr(x2, f32p), _ptr(o2, i32p),
                                           _ptr(a2, f32p), len(d2), _ptr(f1, i32p), _ptr(i1, i32p), _ptr(f2, i32p), _ptr(i2, i32p),
                                           len(f1) - 1, _ptr(F, f32p), float(ex), float(ey), _ptr(sf, f32p), _ptr(s2, f32p),
                                           int(only_stereo), int(check_ori), _ptr(out, i32p))
    return out[:len(d1)], n


def search_by_bow(desc1, valid1, desc2, blocked2, off1, idx1, off2, idx2, max_dist, nn_ratio):
    """SearchByBoW loops (src/ORBmatcher.cc:180-264 / :550-632) over the shared nodes given as two CSR lists."""
    d1, d2 = np.ascontiguousarray(desc1, np.uint8), np.ascontiguousarray(desc2, np.uint8)
    v1 = None if valid1 is None else np.ascontiguousarray(valid1, np.uint8)
    b2 = None if blocked2 is None else np.ascontiguousarray(blocked2, np.uint8)
    o1, i1, o2, i2 = (np.ascontiguousarray(a, np.int32) for a in (off1, idx1, off2, idx2))
    match1 = np.zeros(max(len(d1), 1), np.int32)
    lib().orc_search_by_bow(_ptr(d1, u8p), None if v1 is None else _ptr(v1, u8p), len(d1), _ptr(d2, u8p),
                            None if b2 is None else _ptr(b2, u8p), len(d2), _ptr(o1, i32p), _ptr(i1, i32p), _ptr(o2, i32p),
                            _ptr(i2, i32p), len(o1) - 1, max_dist, nn_ratio, _ptr(match1, i32p))
    return match1[:len(d1)]


def three_maxima(sizes):
    s = np.ascontiguousarray(sizes, np.int32)
    ind = np.zeros(3, np.int32)
    lib().orc_three_maxima(_ptr(s, i32p), len(s), _ptr(ind, i32p))
    return ind.tolist()


def rotation_filter(angle1, angle2, match1):
    a1, a2 = np.ascontiguousarray(angle1, np.float32), np.ascontiguousarray(angle2, np.float32)
    m = np.ascontiguousarray(match1, np.int32).copy()
    n = lib().orc_rotation_filter(_ptr(a1, f32p), _ptr(a2, f32p), _ptr(m, i32p), len(m))
    return m, n


def time_extract(frames: np.ndarray, nfeatures, scale_factor, nlevels, ini_th, min_th, threads, reps=1):
    frames = np.ascontiguousarray(frames)
    n, h, w = frames.shape
    tot = C.c_long(0)
    t = lib().orc_time_extract(nfeatures, scale_factor, nlevels, ini_th, min_th, _ptr(frames, u8p), n, w, h, w, w * h,
                               threads, reps, C.byref(tot))
    return t, tot.value


def time_knn2(q, r, threads, swar=False):
    q = np.ascontiguousarray(q, np.uint8)
    r = np.ascontiguousarray(r, np.uint8)
    return lib().orc_time_knn2(_ptr(q, u8p), len(q), _ptr(r, u8p), len(r), threads, 1 if swar else 0)
