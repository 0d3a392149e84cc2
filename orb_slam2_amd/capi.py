"""ctypes binding of liborbfe.so (the C-ABI declared in include/orbfe.h).

This is plumbing for tests and bench.py: the product is the shared library.  There is NO CPU fallback -- if the
library is missing, or no GPU is visible when a compute entry point is called, the call raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

PKG = Path(__file__).resolve().parent
import os as _os

LIB_PATH = Path(_os.environ.get("ORBFE_LIB", PKG / "liborbfe.so"))   # ORBFE_LIB: A/B builds during kernel work

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

ORBFE_OK, ERR_ARG, ERR_DEVICE, ERR_CAPACITY, ERR_UNSUPPORTED = 0, -1, -2, -3, -4

# every symbol include/orbfe.h declares (tests check that the library exports exactly these)
EXPORTS = [
    "orbfe_last_error", "orbfe_device_count", "orbfe_extractor_create", "orbfe_extractor_destroy", "orbfe_get_tables",
    "orbfe_extract", "orbfe_extract_batch", "orbfe_synchronize", "orbfe_pyramid_level", "orbfe_debug_stage",
    "orbfe_hamming", "orbfe_matcher_create", "orbfe_matcher_destroy", "orbfe_match_knn2", "orbfe_match_csr",
    "orbfe_stereo_rowband", "orbfe_matcher_synchronize", "orbfe_matcher_stream", "orbfe_extractor_stream",
    "orbfe_merge_knn2", "orbfe_set_profiling", "orbfe_get_profile", "orbfe_set_quadtree_mode", "orbfe_stereo_match",
    "orbfe_vocab_create", "orbfe_vocab_destroy", "orbfe_bow_transform", "orbfe_distinctive_descriptors",
    "orbfe_frame_grid_create", "orbfe_frame_grid_destroy", "orbfe_search_area", "orbfe_search_by_bow",
    "orbfe_search_area_ex", "orbfe_search_for_triangulation", "orbfe_extract_sequence", "orbfe_match_knn2_noalloc",
    "orbfe_knn2_sharded_unique_id", "orbfe_knn2_sharded_create", "orbfe_knn2_sharded_run", "orbfe_knn2_sharded_destroy",
    "orbfe_frame_grid_update", "orbfe_pyramid_mirror", "orbfe_bow_transform_host", "orbfe_stereo_rowband_host", "orbfe_stereo_match_host",
    "orbfe_debug_rotation", "orbfe_knn2_sharded_available", "orbfe_debug_fast_variant",
]

STAGES = ("pyramid", "fast", "blur", "describe", "quadtree")


class OrbfeParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32)]


class OrbfeProfile(C.Structure):
    _fields_ = [("stage_ms", C.c_double * 5), ("stage_launches", C.c_int64 * 5), ("frames", C.c_int64),
                ("candidates", C.c_int64), ("keypoints", C.c_int64)]


class OrbfeAreaOptions(C.Structure):
    _fields_ = [("blocked", C.c_void_p), ("dist_limit", C.c_void_p), ("inv_level_sigma2", C.c_void_p), ("nlevels", C.c_int32),
                ("has_query_origin", C.c_int32), ("chi2_mono", C.c_double), ("chi2_stereo", C.c_double),
                ("query_min_x", C.c_float), ("query_min_y", C.c_float)]


class OrbfeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"orbfe error {code}: {msg}")
        self.code = code


_lib = None
vp, u8p, f32p, i32p = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_int32)


def lib():
    """Load liborbfe.so (built in-tree by __graft_entry__.build() / make -C orb_slam2_amd/csrc)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C orb_slam2_amd/csrc` "
                          "(the HIP library is the product; there is no fallback)")
    # One HIP runtime per process: PyTorch bundles its own libamdhip64.so.7.  Loading torch first makes the dynamic
    # linker bind liborbfe's libamdhip64.so.7 dependency to that copy; the other order leaves two runtimes in the
    # process and torch then reports "No HIP GPUs are available".
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(str(LIB_PATH))
    L.orbfe_last_error.restype = C.c_char_p
    L.orbfe_extractor_create.argtypes = [C.POINTER(OrbfeParams), C.c_int, C.POINTER(vp)]
    L.orbfe_extractor_destroy.argtypes = [vp]
    L.orbfe_get_tables.argtypes = [vp, f32p, f32p, f32p, f32p, i32p, i32p]
    L.orbfe_extract.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, i32p]
    L.orbfe_extract_batch.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, vp, vp, vp, C.c_int, C.c_int]
    L.orbfe_synchronize.argtypes = [vp]
    L.orbfe_extract_sequence.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, vp, vp, vp, C.c_int, C.c_int]
    L.orbfe_pyramid_level.argtypes = [vp, C.c_int, C.c_int, vp, C.c_size_t, i32p, i32p]
    L.orbfe_debug_stage.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.orbfe_debug_rotation.argtypes = [vp, vp, C.c_int, vp]
    L.orbfe_debug_fast_variant.argtypes = [vp]
    L.orbfe_hamming.argtypes = [vp, vp]
    L.orbfe_matcher_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.orbfe_matcher_destroy.argtypes = [vp]
    L.orbfe_match_knn2.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int32, vp, vp, vp, C.c_int]
    L.orbfe_match_knn2_noalloc.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int32, vp, vp, vp, C.c_int]
    L.orbfe_knn2_sharded_available.argtypes = []
    L.orbfe_knn2_sharded_unique_id.argtypes = [vp]
    L.orbfe_knn2_sharded_create.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.orbfe_knn2_sharded_run.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int32, vp, vp, vp, C.c_int]
    L.orbfe_knn2_sharded_destroy.argtypes = [vp]
    L.orbfe_match_csr.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int]
    L.orbfe_stereo_rowband.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int, f32p, C.c_int, C.c_int, C.c_float, C.c_float,
                                       vp, vp, C.c_int]
    L.orbfe_stereo_match.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, vp, vp, C.c_int, C.c_float, C.c_float, vp, vp,
                                     C.c_int]
    L.orbfe_vocab_create.argtypes = [C.c_int, C.c_int, vp, vp, vp, C.c_int, C.POINTER(vp)]
    L.orbfe_vocab_destroy.argtypes = [vp]
    L.orbfe_bow_transform.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, C.c_int]
    L.orbfe_distinctive_descriptors.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, C.c_int]
    L.orbfe_frame_grid_create.argtypes = [C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                          C.POINTER(vp)]
    L.orbfe_frame_grid_destroy.argtypes = [vp]
    L.orbfe_frame_grid_update.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
    L.orbfe_search_area.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp]
    L.orbfe_search_by_bow.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_float, vp]
    L.orbfe_search_area_ex.argtypes = [vp, vp, vp, vp, C.c_int, C.POINTER(OrbfeAreaOptions), vp]
    L.orbfe_search_for_triangulation.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int,
                                                 vp, C.c_float, C.c_float, vp, vp, C.c_int, C.c_int, vp]
    L.orbfe_matcher_synchronize.argtypes = [vp]
    L.orbfe_matcher_stream.restype = vp
    L.orbfe_matcher_stream.argtypes = [vp]
    L.orbfe_extractor_stream.restype = vp
    L.orbfe_extractor_stream.argtypes = [vp]
    L.orbfe_merge_knn2.argtypes = [i32p, i32p, i32p, C.c_int, C.c_int, i32p, i32p, i32p]
    L.orbfe_set_profiling.argtypes = [vp, C.c_int]
    L.orbfe_get_profile.argtypes = [vp, C.POINTER(OrbfeProfile)]
    L.orbfe_set_quadtree_mode.argtypes = [vp, C.c_int]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise OrbfeError(rc, lib().orbfe_last_error().decode())


def _check_csr(off, idx, what):
    """The C-ABI takes node_off[n_nodes] as the length of node_idx (it sees only pointers): a binding that knows the array
    lengths must refuse an offset array that promises more entries than the index array holds."""
    if len(off) < 1 or int(off[-1]) != len(idx):
        raise OrbfeError(-1, f"{what}: node offsets end at {int(off[-1]) if len(off) else None}, index array holds {len(idx)}")


def _np_ptr(a):
    return a.ctypes.data_as(vp)


class ORBextractor:
    """Mirror of ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:45-111) over the C-ABI."""

    def __init__(self, nfeatures=2000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, device=0):
        self._L = lib()
        self._h = vp()
        self.nfeatures, self.nlevels = int(nfeatures), int(nlevels)
        self._scaleFactor = float(np.float32(scaleFactor))
        p = OrbfeParams(int(nfeatures), float(scaleFactor), int(nlevels), int(iniThFAST), int(minThFAST))
        check(self._L.orbfe_extractor_create(C.byref(p), int(device), C.byref(self._h)))

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and h.value and getattr(self, "_L", None) is not None:
            self._L.orbfe_extractor_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:   # interpreter shutdown: module globals may already be gone
            pass

    # ---- getters (include/ORBextractor.h:63-83)
    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return self._scaleFactor

    def _tables(self):
        n = self.nlevels
        s, i, s2, i2 = (np.zeros(n, np.float32) for _ in range(4))
        fpl, um = np.zeros(n, np.int32), np.zeros(16, np.int32)
        check(self._L.orbfe_get_tables(self._h, s.ctypes.data_as(f32p), i.ctypes.data_as(f32p), s2.ctypes.data_as(f32p),
                                       i2.ctypes.data_as(f32p), fpl.ctypes.data_as(i32p), um.ctypes.data_as(i32p)))
        return dict(scale=s, inv_scale=i, sigma2=s2, inv_sigma2=i2, features_per_level=fpl, umax=um)

    def GetScaleFactors(self):
        return self._tables()["scale"]

    def GetInverseScaleFactors(self):
        return self._tables()["inv_scale"]

    def GetScaleSigmaSquares(self):
        return self._tables()["sigma2"]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()["inv_sigma2"]

    # ---- operator() (include/ORBextractor.h:59-61); the mask is ignored as in the reference (:58)
    def __call__(self, image: np.ndarray, mask=None):
        if image is None or image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2, "CV_8UC1 image expected (reference asserts the same)"
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        cap = self.nfeatures + 4 * self.nlevels + 64
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int32(0)
        rc = self._L.orbfe_extract(self._h, _np_ptr(image), image.shape[1], image.shape[0], image.strides[0], _np_ptr(kps),
                                   cap, _np_ptr(desc), C.byref(n))
        if rc == ERR_CAPACITY and n.value > cap:
            cap = n.value
            kps = np.zeros(cap, KP_DTYPE)
            desc = np.zeros((cap, 32), np.uint8)
            rc = self._L.orbfe_extract(self._h, _np_ptr(image), image.shape[1], image.shape[0], image.strides[0],
                                       _np_ptr(kps), cap, _np_ptr(desc), C.byref(n))
        check(rc)
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_batch_device(self, dev_imgs_ptr: int, n_frames, w, h, stride, frame_stride, dev_kps_ptr, dev_desc_ptr,
                             dev_counts_ptr, cap, sync=True):
        """Batched extraction on frames resident in HBM; pointers are raw device addresses (e.g. tensor.data_ptr())."""
        check(self._L.orbfe_extract_batch(self._h, vp(dev_imgs_ptr), n_frames, w, h, stride, frame_stride, vp(dev_kps_ptr),
                                          vp(dev_desc_ptr), vp(dev_counts_ptr), cap, 1 if sync else 0))

    def extract_sequence_host(self, imgs_ptr: int, n_frames, w, h, stride, frame_stride, kps_ptr, desc_ptr, counts_ptr, cap, chunk=0):
        """orbfe_extract_sequence: host frames in, host keypoints / descriptors / counts out (pointers to host memory)."""
        check(self._L.orbfe_extract_sequence(self._h, vp(imgs_ptr), n_frames, w, h, stride, frame_stride, vp(kps_ptr), vp(desc_ptr),
                                             vp(counts_ptr), cap, chunk))

    def synchronize(self):
        check(self._L.orbfe_synchronize(self._h))

    def set_quadtree_mode(self, mode: str):
        """'device' (default) or 'host' -- where DistributeOctTree runs; identical results."""
        check(self._L.orbfe_set_quadtree_mode(self._h, {"device": 0, "host": 1}[mode]))

    def set_profiling(self, enable):
        """False/0 off, True/1 every stage, 2 the FAST kernel only (least disturbance)."""
        check(self._L.orbfe_set_profiling(self._h, int(enable)))

    def profile(self) -> dict:
        p = OrbfeProfile()
        check(self._L.orbfe_get_profile(self._h, C.byref(p)))
        return dict(stage_ms={n: p.stage_ms[i] for i, n in enumerate(STAGES)},
                    stage_launches={n: p.stage_launches[i] for i, n in enumerate(STAGES)},
                    frames=p.frames, candidates=p.candidates, keypoints=p.keypoints)

    def stream(self) -> int:
        return int(self._L.orbfe_extractor_stream(self._h) or 0)

    def fast_variant(self) -> int:
        """orbfe_debug_fast_variant: 1 dense, 2 dense with the large survivor list (after an overflow)."""
        return int(self._L.orbfe_debug_fast_variant(self._h))

    # ---- mvImagePyramid (include/ORBextractor.h:85)
    def pyramid_level(self, level, frame=0, with_border=False):
        w, h = C.c_int32(), C.c_int32()
        check(self._L.orbfe_pyramid_level(self._h, frame, level, None, 0, C.byref(w), C.byref(h)))
        buf = np.zeros((h.value + 38, w.value + 38), np.uint8)
        check(self._L.orbfe_pyramid_level(self._h, frame, level, _np_ptr(buf), buf.size, C.byref(w), C.byref(h)))
        return buf if with_border else buf[19:-19, 19:-19]

    @property
    def mvImagePyramid(self):
        return [self.pyramid_level(l) for l in range(self.nlevels)]

    def debug_stage(self, level, what, frame=0):
        nbytes = C.c_size_t(0)
        check(self._L.orbfe_debug_stage(self._h, frame, level, what, None, 0, C.byref(nbytes)))
        raw = np.zeros(max(nbytes.value, 1), np.uint8)
        check(self._L.orbfe_debug_stage(self._h, frame, level, what, _np_ptr(raw), nbytes.value, C.byref(nbytes)))
        raw = raw[:nbytes.value]
        if what in (0, 1):
            w, h = C.c_int32(), C.c_int32()
            check(self._L.orbfe_pyramid_level(self._h, frame, level, None, 0, C.byref(w), C.byref(h)))
            return raw.reshape(h.value, w.value)
        return raw.view(np.float32).reshape(-1, 3 if what == 2 else 4)

    def debug_rotation(self, angles_deg):
        """(cos, sin) pairs the descriptor kernel derives from float keypoint angles in degrees."""
        a = np.ascontiguousarray(angles_deg, np.float32)
        out = np.zeros((a.size, 2), np.float32)
        check(self._L.orbfe_debug_rotation(self._h, _np_ptr(a), a.size, _np_ptr(out)))
        return out


class ORBmatcher:
    """Mirror of the Hamming part of ORB_SLAM2::ORBmatcher (reference include/ORBmatcher.h:41-106) over the C-ABI.

    Device-pointer methods take raw device addresses (torch ``tensor.data_ptr()``)."""

    TH_LOW, TH_HIGH, HISTO_LENGTH = 50, 100, 30   # src/ORBmatcher.cc:37-39

    def __init__(self, nnratio=0.6, checkOri=True, device=0):
        self._L = lib()
        self.mfNNratio, self.mbCheckOrientation = float(nnratio), bool(checkOri)
        self._h = vp()
        check(self._L.orbfe_matcher_create(int(device), C.byref(self._h)))

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and h.value and getattr(self, "_L", None) is not None:
            self._L.orbfe_matcher_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def DescriptorDistance(a: np.ndarray, b: np.ndarray) -> int:
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        assert a.size == 32 and b.size == 32
        return int(lib().orbfe_hamming(_np_ptr(a), _np_ptr(b)))

    def knn2_device(self, q_ptr, nq, r_ptr, nr, bd_ptr, bi_ptr, sd_ptr, idx_base=0, sync=True):
        check(self._L.orbfe_match_knn2(self._h, vp(q_ptr), nq, vp(r_ptr), nr, idx_base, vp(bd_ptr), vp(bi_ptr), vp(sd_ptr),
                                       1 if sync else 0))

    def knn2_noalloc_device(self, q_ptr, nq, r_ptr, nr, bd_ptr, bi_ptr, sd_ptr, idx_base=0, sync=True):
        """orbfe_match_knn2_noalloc: the same search with its scratch in the library's static device array (no hipMalloc)."""
        check(self._L.orbfe_match_knn2_noalloc(self._h, vp(q_ptr), nq, vp(r_ptr), nr, idx_base, vp(bd_ptr), vp(bi_ptr), vp(sd_ptr),
                                               1 if sync else 0))

    def csr_device(self, q_ptr, nq, r_ptr, nr, off_ptr, idx_ptr, bd_ptr, bi_ptr, sd_ptr, sync=True):
        check(self._L.orbfe_match_csr(self._h, vp(q_ptr), nq, vp(r_ptr), nr, vp(off_ptr), vp(idx_ptr), vp(bd_ptr), vp(bi_ptr),
                                      vp(sd_ptr), 1 if sync else 0))

    def stereo_rowband_device(self, kl_ptr, dl_ptr, nl, kr_ptr, dr_ptr, nr, scale_factors, nrows, min_d, max_d, bd_ptr, bi_ptr,
                              sync=True):
        sf = np.ascontiguousarray(scale_factors, np.float32)
        check(self._L.orbfe_stereo_rowband(self._h, vp(kl_ptr), vp(dl_ptr), nl, vp(kr_ptr), vp(dr_ptr), nr,
                                           sf.ctypes.data_as(f32p), len(sf), nrows, min_d, max_d, vp(bd_ptr), vp(bi_ptr),
                                           1 if sync else 0))

    def stereo_match_device(self, left: "ORBextractor", left_frame, right: "ORBextractor", right_frame, kl_ptr, dl_ptr, nl,
                            kr_ptr, dr_ptr, nr, mbf, mb, uright_ptr, depth_ptr, sync=True):
        """Frame::ComputeStereoMatches (src/Frame.cc:466-640) on the device; pyramids from the extractors' last call."""
        check(self._L.orbfe_stereo_match(self._h, left._h, left_frame, right._h, right_frame, vp(kl_ptr), vp(dl_ptr), nl,
                                         vp(kr_ptr), vp(dr_ptr), nr, mbf, mb, vp(uright_ptr), vp(depth_ptr), 1 if sync else 0))

    def bow_transform_device(self, vocab: "Vocabulary", desc_ptr, n, levelsup, leaf_ptr, nid_ptr, sync=True):
        check(self._L.orbfe_bow_transform(self._h, vocab._h, vp(desc_ptr), n, levelsup, vp(leaf_ptr), vp(nid_ptr),
                                          1 if sync else 0))

    def distinctive_device(self, desc_ptr, off_ptr, n_points, max_per_point, best_ptr, sync=True):
        check(self._L.orbfe_distinctive_descriptors(self._h, vp(desc_ptr), vp(off_ptr), n_points, max_per_point, vp(best_ptr),
                                                    1 if sync else 0))

    def search_by_bow(self, desc1, valid1, desc2, blocked2, off1, idx1, off2, idx2, max_dist=None, nn_ratio=None):
        """SearchByBoW loops of src/ORBmatcher.cc:180-264 / :550-632 over shared nodes (host arrays in, match1 out)."""
        d1, d2 = np.ascontiguousarray(desc1, np.uint8), np.ascontiguousarray(desc2, np.uint8)
        v1 = None if valid1 is None else np.ascontiguousarray(valid1, np.uint8)
        b2 = None if blocked2 is None else np.ascontiguousarray(blocked2, np.uint8)
        o1, i1, o2, i2 = (np.ascontiguousarray(a, np.int32) for a in (off1, idx1, off2, idx2))
        _check_csr(o1, i1, "side 1")
        _check_csr(o2, i2, "side 2")
        match1 = np.zeros(max(len(d1), 1), np.int32)
        check(self._L.orbfe_search_by_bow(self._h, _np_ptr(d1), None if v1 is None else _np_ptr(v1), len(d1), _np_ptr(d2),
                                          None if b2 is None else _np_ptr(b2), len(d2), _np_ptr(o1), _np_ptr(i1), _np_ptr(o2),
                                          _np_ptr(i2), len(o1) - 1, self.TH_LOW if max_dist is None else int(max_dist),
                                          self.mfNNratio if nn_ratio is None else float(nn_ratio), _np_ptr(match1)))
        return match1[:len(d1)]

    def search_for_triangulation(self, desc1, has_mp1, uright1, xy1, desc2, has_mp2, uright2, xy2, oct2, off1, idx1, off2, idx2,
                                 F12, ex, ey, scale_factors, level_sigma2, only_stereo=False):
        """SearchForTriangulation loops of src/ORBmatcher.cc:691-789 over shared nodes (host arrays in, vMatches12 out)."""
        d1, d2 = np.ascontiguousarray(desc1, np.uint8), np.ascontiguousarray(desc2, np.uint8)
        h1, h2 = np.ascontiguousarray(has_mp1, np.uint8), np.ascontiguousarray(has_mp2, np.uint8)
        u1, u2 = np.ascontiguousarray(uright1, np.float32), np.ascontiguousarray(uright2, np.float32)
        x1, x2 = np.ascontiguousarray(xy1, np.float32), np.ascontiguousarray(xy2, np.float32)
        o2 = np.ascontiguousarray(oct2, np.int32)
        f1, i1, f2, i2 = (np.ascontiguousarray(a, np.int32) for a in (off1, idx1, off2, idx2))
        F = np.ascontiguousarray(F12, np.float32).reshape(9)
        sf, s2 = np.ascontiguousarray(scale_factors, np.float32), np.ascontiguousarray(level_sigma2, np.float32)
        _check_csr(f1, i1, "side 1")
        _check_csr(f2, i2, "side 2")
        match1 = np.zeros(max(len(d1), 1), np.int32)
        check(self._L.orbfe_search_for_triangulation(self._h, _np_ptr(d1), _np_ptr(h1), _np_ptr(u1), _np_ptr(x1), len(d1), _np_ptr(d2),
                                                     _np_ptr(h2), _np_ptr(u2), _np_ptr(x2), _np_ptr(o2), len(d2), _np_ptr(f1),
                                                     _np_ptr(i1), _np_ptr(f2), _np_ptr(i2), len(f1) - 1, _np_ptr(F), float(ex),
                                                     float(ey), _np_ptr(sf), _np_ptr(s2), len(sf), 1 if only_stereo else 0,
                                                     _np_ptr(match1)))
        return match1[:len(d1)]

    def synchronize(self):
        check(self._L.orbfe_matcher_synchronize(self._h))

    def stream(self) -> int:
        return int(self._L.orbfe_matcher_stream(self._h) or 0)


def sharded_available() -> bool:
    """orbfe_knn2_sharded_available: True iff librccl could be bound.  Touches no GPU and is not collective; the loader's
    message is in orbfe_last_error() when it returns False."""
    return lib().orbfe_knn2_sharded_available() == 0


def sharded_unique_id() -> bytes:
    """orbfe_knn2_sharded_unique_id: rank 0 makes the 128-byte RCCL id; ship it to the other ranks before ShardedKnn2()."""
    buf = (C.c_uint8 * 128)()
    check(lib().orbfe_knn2_sharded_unique_id(buf))
    return bytes(buf)


class ShardedKnn2:
    """orbfe_knn2_sharded_*: kNN-2 against a database sharded over the ranks, RCCL broadcast + all-gather inside liborbfe."""

    def __init__(self, matcher: "ORBmatcher", unique_id: bytes, rank: int, world: int, max_queries: int):
        self._L, self._m = lib(), matcher       # the matcher (its stream) must outlive this object
        self._h = vp()
        idb = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        check(self._L.orbfe_knn2_sharded_create(matcher._h, idb, rank, world, max_queries, C.byref(self._h)))

    def run(self, q_ptr, nq, shard_ptr, n_shard, idx_base, bd_ptr, bi_ptr, sd_ptr, sync=True):
        check(self._L.orbfe_knn2_sharded_run(self._h, vp(q_ptr), nq, vp(shard_ptr), n_shard, idx_base, vp(bd_ptr), vp(bi_ptr),
                                             vp(sd_ptr), 1 if sync else 0))

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and h.value:
            self._L.orbfe_knn2_sharded_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


AREA_QUERY_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("r", "<f4"), ("min_level", "<i4"), ("max_level", "<i4"),
                             ("proj_xr", "<f4"), ("er_max", "<f4"), ("pad", "<i4")])
AREA_RESULT_DTYPE = np.dtype([("best_dist", "<i4"), ("best_idx", "<i4"), ("best_level", "<i4"), ("best_dist2", "<i4"),
                              ("best_level2", "<i4"), ("n_candidates", "<i4"), ("best_idx2", "<i4"), ("pad1", "<i4")])


class FrameGrid:
    """Device copy of a Frame's 64 x 48 keypoint grid (include/orbfe.h: orbfe_frame_grid_create) + windowed search."""

    def __init__(self, kx, ky, koct, kdesc, uright, min_x, min_y, width_inv, height_inv, device=0):
        self._L = lib()
        self._h = vp()
        kx, ky = np.ascontiguousarray(kx, np.float32), np.ascontiguousarray(ky, np.float32)
        koct, kdesc = np.ascontiguousarray(koct, np.int32), np.ascontiguousarray(kdesc, np.uint8)
        ur = None if uright is None else np.ascontiguousarray(uright, np.float32)
        self.n = len(kx)
        check(self._L.orbfe_frame_grid_create(device, _np_ptr(kx), _np_ptr(ky), _np_ptr(koct), _np_ptr(kdesc),
                                              None if ur is None else _np_ptr(ur), self.n, min_x, min_y, width_inv, height_inv,
                                              C.byref(self._h)))

    def search(self, matcher: "ORBmatcher", queries: np.ndarray, qdesc: np.ndarray, blocked=None) -> np.ndarray:
        q = np.ascontiguousarray(queries, AREA_QUERY_DTYPE)
        d = np.ascontiguousarray(qdesc, np.uint8)
        b = None if blocked is None else np.ascontiguousarray(blocked, np.uint8)
        res = np.zeros(len(q), AREA_RESULT_DTYPE)
        check(self._L.orbfe_search_area(matcher._h, self._h, _np_ptr(q), _np_ptr(d), len(q), None if b is None else _np_ptr(b),
                                        _np_ptr(res)))
        return res

    def update(self, matcher: "ORBmatcher", kx, ky, koct, kdesc, uright, min_x, min_y, width_inv, height_inv):
        """orbfe_frame_grid_update: refill this grid with another frame's keypoints (device block reused)."""
        kx, ky = np.ascontiguousarray(kx, np.float32), np.ascontiguousarray(ky, np.float32)
        koct, kdesc = np.ascontiguousarray(koct, np.int32), np.ascontiguousarray(kdesc, np.uint8)
        ur = None if uright is None else np.ascontiguousarray(uright, np.float32)
        self.n = len(kx)
        check(self._L.orbfe_frame_grid_update(matcher._h, self._h, _np_ptr(kx), _np_ptr(ky), _np_ptr(koct), _np_ptr(kdesc),
                                              None if ur is None else _np_ptr(ur), self.n, min_x, min_y, width_inv, height_inv))

    def search_ex(self, matcher: "ORBmatcher", queries: np.ndarray, qdesc: np.ndarray, blocked=None, dist_limit=None,
                  inv_level_sigma2=None, chi2_mono=5.99, chi2_stereo=7.8, query_origin=None) -> np.ndarray:
        """orbfe_search_area_ex: the window search with the skip mask, the vMatchedDistance limit and the Fuse gate."""
        q = np.ascontiguousarray(queries, AREA_QUERY_DTYPE)
        d = np.ascontiguousarray(qdesc, np.uint8)
        b = None if blocked is None else np.ascontiguousarray(blocked, np.uint8)
        lim = None if dist_limit is None else np.ascontiguousarray(dist_limit, np.int32)
        s2 = None if inv_level_sigma2 is None else np.ascontiguousarray(inv_level_sigma2, np.float32)
        opt = OrbfeAreaOptions(None if b is None else b.ctypes.data, None if lim is None else lim.ctypes.data,
                               None if s2 is None else s2.ctypes.data, 0 if s2 is None else len(s2),
                               0 if query_origin is None else 1, chi2_mono, chi2_stereo,
                               0.0 if query_origin is None else float(query_origin[0]),
                               0.0 if query_origin is None else float(query_origin[1]))
        res = np.zeros(len(q), AREA_RESULT_DTYPE)
        check(self._L.orbfe_search_area_ex(matcher._h, self._h, _np_ptr(q), _np_ptr(d), len(q), C.byref(opt), _np_ptr(res)))
        return res

    def close(self):
        if getattr(self, "_h", None):
            self._L.orbfe_frame_grid_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Vocabulary:
    """Device copy of a DBoW2 vocabulary tree (flat arrays, see include/orbfe.h: orbfe_vocab_create)."""

    def __init__(self, child_off, child_idx, node_desc, depth_L, device=0):
        self._L = lib()
        self._h = vp()
        co = np.ascontiguousarray(child_off, np.int32)
        ci = np.ascontiguousarray(child_idx, np.int32)
        nd = np.ascontiguousarray(node_desc, np.uint8)
        check(self._L.orbfe_vocab_create(device, len(co) - 1, _np_ptr(co), _np_ptr(ci) if len(ci) else None, _np_ptr(nd),
                                         depth_L, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            self._L.orbfe_vocab_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def merge_knn2(best_dist, best_idx, second_dist):
    """Fold per-shard results [S, nq] into the unsharded answer (include/orbfe.h: orbfe_merge_knn2)."""
    bd = np.ascontiguousarray(best_dist, np.int32)
    bi = np.ascontiguousarray(best_idx, np.int32)
    sd = np.ascontiguousarray(second_dist, np.int32)
    S, nq = bd.shape
    obd, obi, osd = (np.zeros(nq, np.int32) for _ in range(3))
    check(lib().orbfe_merge_knn2(bd.ctypes.data_as(i32p), bi.ctypes.data_as(i32p), sd.ctypes.data_as(i32p), S, nq,
                                 obd.ctypes.data_as(i32p), obi.ctypes.data_as(i32p), osd.ctypes.data_as(i32p)))
    return obd, obi, osd
