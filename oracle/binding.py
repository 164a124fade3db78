"""ctypes binding of oracle/libmpe_oracle.so (CPU oracle — test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmpe_oracle.so")


def build(force=False):
    """Compile the oracle with plain g++ (oracle/Makefile)."""
    src = [os.path.join(_HERE, f) for f in ("mpe_oracle.cpp", "mpe_oracle.h", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _LIB_PATH


class OrcParams(C.Structure):
    _fields_ = [
        ("threshold_value", C.c_int),
        ("gaussian_sigma", C.c_double),
        ("min_blob_area", C.c_double),
        ("max_blob_area", C.c_double),
        ("max_width_height_distortion", C.c_double),
        ("max_circular_distortion", C.c_double),
        ("back_projection_pixel_tolerance", C.c_double),
        ("nearest_neighbour_pixel_tolerance", C.c_double),
        ("certainty_threshold", C.c_double),
        ("valid_correspondence_threshold", C.c_double),
        ("roi_border_thickness", C.c_uint),
        ("histogram_threshold", C.c_uint),
    ]


class OrcResult(C.Structure):
    _fields_ = [
        ("T", C.c_double * 16),
        ("cov", C.c_double * 36),
        ("status", C.c_int),
        ("n_det", C.c_int),
        ("n_corr", C.c_int),
        ("gn_iterations", C.c_int),
    ]


ORC_RESULT_DTYPE = np.dtype([("T", "f8", (16,)), ("cov", "f8", (36,)), ("status", "i4"),
                             ("n_det", "i4"), ("n_corr", "i4"), ("gn_iterations", "i4")])
assert ORC_RESULT_DTYPE.itemsize == C.sizeof(OrcResult)

# demo.launch:12-22 parameter set (the BASELINE configs' parameters)
DEMO_PARAMS = dict(threshold_value=140, gaussian_sigma=0.6, min_blob_area=10.0, max_blob_area=200.0,
                   max_width_height_distortion=0.5, max_circular_distortion=0.5,
                   back_projection_pixel_tolerance=5.0, nearest_neighbour_pixel_tolerance=7.0,
                   certainty_threshold=0.75, valid_correspondence_threshold=0.7,
                   roi_border_thickness=20, histogram_threshold=0)


def make_params(**kw):
    d = dict(DEMO_PARAMS)
    d.update(kw)
    return OrcParams(**d)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_num_combinations.restype = C.c_uint
        _lib.orc_factorial.restype = C.c_uint
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def num_combinations(n, k):
    return int(lib().orc_num_combinations(C.c_uint(n), C.c_uint(k)))


def combinations3(n):
    rows = lib().orc_combinations3(C.c_uint(n), None)
    out = np.zeros((rows, 3), np.uint32)
    lib().orc_combinations3(C.c_uint(n), _p(out, C.c_uint))
    return out


def permutations3(n):
    rows = lib().orc_permutations3(C.c_uint(n), None)
    out = np.zeros((rows, 3), np.uint32)
    lib().orc_permutations3(C.c_uint(n), _p(out, C.c_uint))
    return out


def solve_quartic(factors):
    f = _f64(factors)
    r = np.zeros(4)
    lib().orc_solve_quartic(_p(f, C.c_double), _p(r, C.c_double))
    return r


def p3p(fv, wp):
    """fv, wp: (3,3) arrays whose ROWS are the bearings / world points. -> (rc, sol[4,3,4])"""
    f = _f64(fv)
    w = _f64(wp)
    s = np.zeros((4, 3, 4))
    rc = lib().orc_p3p(_p(f, C.c_double), _p(w, C.c_double), _p(s, C.c_double))
    return rc, s


def gaussian_kernel_q8(sigma):
    t = np.zeros(64, np.int32)
    n = lib().orc_gaussian_kernel_q8(C.c_double(sigma), _p(t, C.c_int), 64)
    if n < 0:
        raise ValueError("bad sigma")
    return t[:n].copy()


def blur_mask(img, thr, sigma, roi=None):
    img = np.ascontiguousarray(img, np.uint8)
    rows, cols = img.shape
    rx, ry, rw, rh = roi if roi is not None else (0, 0, cols, rows)
    b = np.zeros((rh, rw), np.uint8)
    m = np.zeros((rh, rw), np.uint8)
    rc = lib().orc_blur_mask(_p(img, C.c_uint8), rows, cols, C.c_size_t(img.strides[0]), rx, ry, rw, rh,
                             int(thr), C.c_double(sigma), _p(b, C.c_uint8), _p(m, C.c_uint8))
    if rc != 0:
        raise ValueError("orc_blur_mask rc=%d" % rc)
    return b, m


def external_contours(mask):
    mask = np.ascontiguousarray(mask, np.uint8)
    h, w = mask.shape
    cap = max(16, 4 * h * w + 16)
    pts = np.zeros((cap, 2), np.int32)
    counts = np.zeros(max(16, h * w + 1), np.int32)
    n = lib().orc_external_contours(_p(mask, C.c_uint8), h, w, _p(pts, C.c_int), cap, _p(counts, C.c_int),
                                    len(counts))
    if n < 0:
        raise ValueError("capacity")
    out, off = [], 0
    for i in range(n):
        out.append(pts[off:off + counts[i]].copy())
        off += counts[i]
    return out


def find_leds(img, params, K, D, roi=None, cap=4096):
    img = np.ascontiguousarray(img, np.uint8)
    rows, cols = img.shape
    rx, ry, rw, rh = roi if roi is not None else (0, 0, cols, rows)
    K = _f64(K).reshape(9)
    D = _f64(D).reshape(-1)
    und = np.zeros((cap, 2))
    dst = np.zeros((cap, 2), np.float32)
    n = C.c_int(0)
    rc = lib().orc_find_leds(_p(img, C.c_uint8), rows, cols, C.c_size_t(img.strides[0]), rx, ry, rw, rh,
                             C.byref(params), _p(K, C.c_double), _p(D, C.c_double), len(D),
                             _p(und, C.c_double), _p(dst, C.c_float), cap, C.byref(n))
    if rc != 0:
        raise ValueError("orc_find_leds rc=%d" % rc)
    return und[:n.value].copy(), dst[:n.value].copy()


def distort_points(xy, K, D):
    s = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    d = np.zeros_like(s)
    K = _f64(K).reshape(9)
    D = _f64(D).reshape(-1)
    lib().orc_distort_points(_p(s, C.c_float), _p(d, C.c_float), len(s), _p(K, C.c_double),
                             _p(D, C.c_double), len(D))
    return d


def determine_roi(px, rows, cols, border, K, D):
    px = _f64(px).reshape(-1, 2)
    K = _f64(K).reshape(9)
    D = _f64(D).reshape(-1)
    roi = np.zeros(4, np.int32)
    lib().orc_determine_roi(_p(px, C.c_double), len(px), int(rows), int(cols), int(border), _p(K, C.c_double),
                            _p(D, C.c_double), len(D), _p(roi, C.c_int))
    return tuple(int(v) for v in roi)


def undistort_points(xy, K, D):
    s = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    d = np.zeros_like(s)
    K = _f64(K).reshape(9)
    D = _f64(D).reshape(-1)
    lib().orc_undistort_points(_p(s, C.c_float), _p(d, C.c_float), len(s), _p(K, C.c_double),
                               _p(D, C.c_double), len(D))
    return d


def image_vectors(det, K):
    det = _f64(det).reshape(-1, 2)
    K = _f64(K).reshape(9)
    v = np.zeros((len(det), 3))
    lib().orc_image_vectors(_p(det, C.c_double), len(det), _p(K, C.c_double), _p(v, C.c_double))
    return v


def project2d(p4, T, K):
    p4 = _f64(p4).reshape(4)
    T = _f64(T).reshape(16)
    K = _f64(K).reshape(9)
    o = np.zeros(2)
    lib().orc_project2d(_p(p4, C.c_double), _p(T, C.c_double), _p(K, C.c_double), _p(o, C.c_double))
    return o


def exponential_map(twist):
    t = _f64(twist).reshape(6)
    T = np.zeros(16)
    lib().orc_exponential_map(_p(t, C.c_double), _p(T, C.c_double))
    return T.reshape(4, 4)


def jacobian(T, p4, fx, fy):
    T = _f64(T).reshape(16)
    p4 = _f64(p4).reshape(4)
    J = np.zeros(12)
    lib().orc_jacobian(_p(T, C.c_double), _p(p4, C.c_double), C.c_double(fx), C.c_double(fy),
                       _p(J, C.c_double))
    return J.reshape(2, 6)


def compute_transformation(obj, rep):
    a = _f64(obj).reshape(-1, 3)
    b = _f64(rep).reshape(-1, 3)
    T = np.zeros(16)
    lib().orc_compute_transformation(_p(a, C.c_double), _p(b, C.c_double), len(a), _p(T, C.c_double))
    return T.reshape(4, 4)


def vote_histogram(det, markers, K, tol):
    det = _f64(det).reshape(-1, 2)
    markers = _f64(markers).reshape(-1, 3)
    K = _f64(K).reshape(9)
    h = np.zeros((len(det), len(markers)), np.uint32)
    lib().orc_vote_histogram(_p(det, C.c_double), len(det), _p(markers, C.c_double), len(markers),
                             _p(K, C.c_double), C.c_double(tol), _p(h, C.c_uint32))
    return h


ENCODINGS = {"mono8": 0, "bgr8": 1, "rgb8": 2, "bgra8": 3, "rgba8": 4, "mono16": 5}


def convert_to_mono8(src, encoding, big_endian=False):
    """cv_bridge::toCvCopy(msg, MONO8) for one image: src (rows, cols[, channels]) uint8, or (rows, cols) uint16 for
    mono16 (given in the byte order the message declares: pass big_endian accordingly)."""
    enc = ENCODINGS[encoding]
    src = np.ascontiguousarray(src)
    rows, cols = src.shape[:2]
    raw = src.view(np.uint8).reshape(rows, -1)
    dst = np.zeros((rows, cols), np.uint8)
    rc = lib().orc_convert_to_mono8(raw.ctypes.data_as(C.c_void_p), enc, int(bool(big_endian)), rows, cols,
                                    C.c_size_t(raw.strides[0]), dst.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError("orc_convert_to_mono8 failed")
    return dst


def vote_items(det, markers, K, tol, lo, hi):
    """Votes of the hypotheses [lo, hi) of initialise()'s loop nest only (forensics)."""
    det = _f64(det).reshape(-1, 2)
    markers = _f64(markers).reshape(-1, 3)
    K = _f64(K).reshape(9)
    h = np.zeros((len(det), len(markers)), np.uint32)
    lib().orc_vote_items(_p(det, C.c_double), len(det), _p(markers, C.c_double), len(markers), _p(K, C.c_double),
                         C.c_double(tol), C.c_longlong(lo), C.c_longlong(hi), _p(h, C.c_uint32))
    return h


def correspondences_from_histogram(hist, threshold):
    h = np.ascontiguousarray(hist, np.uint32).copy()
    n_det, n_m = h.shape
    corr = np.zeros((n_m, 2), np.uint32)
    n = lib().orc_correspondences_from_histogram(_p(h, C.c_uint32), n_det, n_m, C.c_uint(threshold),
                                                 _p(corr, C.c_uint32))
    return corr[:n].copy()


def check_correspondences(det, markers, K, params, corr):
    det = _f64(det).reshape(-1, 2)
    markers = _f64(markers).reshape(-1, 3)
    K = _f64(K).reshape(9)
    corr = np.ascontiguousarray(corr, np.uint32).reshape(-1, 2)
    T = np.zeros(16)
    ok = lib().orc_check_correspondences(_p(det, C.c_double), len(det), _p(markers, C.c_double),
                                         len(markers), _p(K, C.c_double), C.byref(params),
                                         _p(corr, C.c_uint32), len(corr), _p(T, C.c_double))
    return ok, T.reshape(4, 4)


def optimise_pose(det, markers, K, corr, T0):
    det = _f64(det).reshape(-1, 2)
    markers = _f64(markers).reshape(-1, 3)
    K = _f64(K).reshape(9)
    corr = np.ascontiguousarray(corr, np.uint32).reshape(-1, 2)
    T = _f64(T0).reshape(16).copy()
    cov = np.zeros(36)
    it = lib().orc_optimise_pose(_p(det, C.c_double), _p(markers, C.c_double), _p(K, C.c_double),
                                 _p(corr, C.c_uint32), len(corr), _p(T, C.c_double), _p(cov, C.c_double))
    return T.reshape(4, 4), cov.reshape(6, 6), it


def solve_bruteforce(det, markers, K, params):
    det = _f64(det).reshape(-1, 2)
    markers = _f64(markers).reshape(-1, 3)
    K = _f64(K).reshape(9)
    res = OrcResult()
    hist = np.zeros((max(len(det), 1), len(markers)), np.uint32)
    corr = np.zeros((len(markers), 2), np.uint32)
    lib().orc_solve_bruteforce(_p(det, C.c_double), len(det), _p(markers, C.c_double), len(markers),
                               _p(K, C.c_double), C.byref(params), C.byref(res), _p(hist, C.c_uint32),
                               _p(corr, C.c_uint32))
    return dict(status=res.status, T=np.array(res.T).reshape(4, 4), cov=np.array(res.cov).reshape(6, 6),
                n_det=res.n_det, n_corr=res.n_corr, gn_iterations=res.gn_iterations,
                hist=hist[:len(det)].copy(), corr=corr[:res.n_corr].copy())


def estimate_batch(frames, markers, K, D, params, n_threads=1):
    """frames: (n, rows, cols) uint8, C-contiguous.  Returns a structured array (ORC_RESULT_DTYPE)."""
    frames = np.ascontiguousarray(frames, np.uint8)
    n, rows, cols = frames.shape
    markers = _f64(markers).reshape(-1, 3)
    K = _f64(K).reshape(9)
    D = _f64(D).reshape(-1)
    out = np.zeros(n, ORC_RESULT_DTYPE)
    rc = lib().orc_estimate_batch(_p(frames, C.c_uint8), n, rows, cols, C.c_size_t(cols),
                                  C.c_size_t(rows * cols), _p(markers, C.c_double), len(markers),
                                  _p(K, C.c_double), _p(D, C.c_double), len(D), C.byref(params),
                                  C.cast(out.ctypes.data, C.POINTER(OrcResult)), int(n_threads))
    if rc < 0:
        raise ValueError("orc_estimate_batch rc=%d" % rc)
    return out


def logarithm_map(T):
    T = _f64(T).reshape(16)
    xi = np.zeros(6)
    lib().orc_logarithm_map(_p(T, C.c_double), _p(xi, C.c_double))
    return xi


class Tracker:
    """Stateful estimator = the reference's PoseEstimator object driven frame after frame
    (uninitialised branch, then ROI tracking with fallback to brute force)."""

    def __init__(self, markers, K, D, params):
        self._lib = lib()
        self._lib.orc_tracker_create.restype = C.c_void_p
        self._lib.orc_tracker_estimate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_double,
                                                   C.POINTER(OrcResult), C.POINTER(C.c_int)]
        self._lib.orc_tracker_destroy.argtypes = [C.c_void_p]
        markers = _f64(markers).reshape(-1, 3)
        K = _f64(K).reshape(9)
        D = _f64(D).reshape(-1)
        self._t = C.c_void_p(self._lib.orc_tracker_create(_p(markers, C.c_double), len(markers), _p(K, C.c_double),
                                                          _p(D, C.c_double), len(D), C.byref(params)))

    def estimate(self, img, time):
        img = np.ascontiguousarray(img, np.uint8)
        res = OrcResult()
        info = (C.c_int * 8)()
        rc = self._lib.orc_tracker_estimate(self._t, img.ctypes.data, img.shape[0], img.shape[1], img.strides[0],
                                            float(time), C.byref(res), info)
        if rc < 0:
            raise ValueError("orc_tracker_estimate rc=%d" % rc)
        return dict(updated=bool(rc), T=np.array(res.T).reshape(4, 4), cov=np.array(res.cov).reshape(6, 6),
                    roi=tuple(info[0:4]), it_since_initialized=info[4], n_det=info[5], n_corr=info[6],
                    used_bruteforce=bool(info[7]))

    def close(self):
        if self._t:
            self._lib.orc_tracker_destroy(self._t)
            self._t = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def vote_batch(det, n_det, markers, K, tol, n_threads=1):
    """det: (n, max_det, 2) float64, n_det: (n,) int32 -> (n, max_det, n_markers) uint32."""
    det = _f64(det)
    n, max_det, _ = det.shape
    nd = np.ascontiguousarray(n_det, np.int32)
    markers = _f64(markers).reshape(-1, 3)
    K = _f64(K).reshape(9)
    h = np.zeros((n, max_det, len(markers)), np.uint32)
    lib().orc_vote_batch(_p(det, C.c_double), _p(nd, C.c_int), n, max_det, _p(markers, C.c_double), len(markers),
                         _p(K, C.c_double), C.c_double(tol), _p(h, C.c_uint32), int(n_threads))
    return h
