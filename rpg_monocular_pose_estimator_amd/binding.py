"""ctypes binding of libmpe_hip.so (include/mpe.h)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_LIB = os.environ.get("MPE_LIB") or os.path.join(_HERE, "libmpe_hip.so")  # MPE_LIB: kernel-variant experiments
_HEADER = os.path.join(os.path.dirname(_HERE), "include", "mpe.h")

MAX_MARKERS = 16
MAX_DETECTIONS = int(os.environ.get("MPE_MAX_DET", "64"))  # (MPE_MAX_DET: A/B runs against a library built with another capacity)


class MpeError(RuntimeError):
    pass


class MpeParams(C.Structure):
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


class MpeResult(C.Structure):
    _fields_ = [("T", C.c_double * 16), ("cov", C.c_double * 36), ("status", C.c_int), ("n_det", C.c_int),
                ("n_corr", C.c_int), ("gn_iterations", C.c_int)]


class MpeDetections(C.Structure):
    _fields_ = [("n", C.c_int), ("status", C.c_int), ("undist_xy", C.c_double * (2 * MAX_DETECTIONS)),
                ("dist_xy", C.c_float * (2 * MAX_DETECTIONS))]


RESULT_DTYPE = np.dtype([("T", "f8", (16,)), ("cov", "f8", (36,)), ("status", "i4"), ("n_det", "i4"),
                         ("n_corr", "i4"), ("gn_iterations", "i4")])
DETECTIONS_DTYPE = np.dtype([("n", "i4"), ("status", "i4"), ("undist_xy", "f8", (2 * MAX_DETECTIONS,)),
                             ("dist_xy", "f4", (2 * MAX_DETECTIONS,))])
assert RESULT_DTYPE.itemsize == C.sizeof(MpeResult)
assert DETECTIONS_DTYPE.itemsize == C.sizeof(MpeDetections)


def library_path():
    return _LIB


def build_library(force=False):
    """Compile the HIP library for gfx950 with hipcc (csrc/Makefile).  Works without a GPU."""
    cmd = ["make", "-s", "-j4", "-C", _CSRC] + (["-B"] if force else [])
    subprocess.check_call(cmd)
    if not os.path.exists(_LIB):
        raise MpeError("build did not produce %s" % _LIB)
    return _LIB


# the kernel sources, in the order in which they read as one text (the former single file mpe_kernels.hip)
DEVICE_SOURCES = ("mpe_kernels_common.h", "mpe_k1.hip", "mpe_k2_head.h", "mpe_k2.hip", "mpe_k3.hip")


def device_source():
    """The kernel sources as ONE text, file prologues / epilogues (`//@file-prologue` .. `-end`, `//@file-epilogue` ..
    `-end`) dropped: what the CPU tier cuts its host builds of the device code from."""
    out = []
    for name in DEVICE_SOURCES:
        with open(os.path.join(_CSRC, name)) as fh:
            txt = fh.read()
        txt = re.sub(r"//@file-prologue\n.*?//@file-prologue-end\n", "", txt, flags=re.S)
        txt = re.sub(r"//@file-epilogue\n.*?//@file-epilogue-end\n", "", txt, flags=re.S)
        # the blob extraction's device functions live in a header both mpe_k1.hip and mpe_k3.hip include (round 6):
        # its marked body stands where mpe_k1.hip includes it, the second include (mpe_k3.hip) is dropped
        if "//@include-k1b-dev" in txt:
            with open(os.path.join(_CSRC, "mpe_k1b_dev.h")) as fh:
                hdr = fh.read()
            body = hdr[hdr.index("//@k1b-dev-begin\n") + len("//@k1b-dev-begin\n"):hdr.index("//@k1b-dev-end\n")]
            txt = re.sub(r"//@include-k1b-dev.*?//@include-k1b-dev-end\n", lambda m: body, txt, count=1, flags=re.S)
        out.append(txt)
    return "".join(out)


def source_fingerprint():
    """sha256 (16 hex digits) over the kernel / ABI sources of libmpe_hip.so: ties committed profiler passes
    (profiles/round3_pmc.json) to the code a bench run times."""
    import hashlib
    hsh = hashlib.sha256()
    for name in DEVICE_SOURCES + ("mpe_k1b_dev.h", "mpe_ddmath.h", "mpe_p3p.h", "mpe_internal.h", "mpe_host.h", "mpe_schedule.cpp",
                                  "mpe_options.cpp", "mpe_track_abi.cpp", "mpe_abi.cpp"):
        with open(os.path.join(_CSRC, name), "rb") as fh:
            hsh.update(fh.read())
    return hsh.hexdigest()[:16]


def exported_symbols():
    """Function names declared in include/mpe.h (used by the symbol-export test)."""
    txt = open(_HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mpe_[a-z0-9_]+)\s*\(", txt)))


_lib = None


def load_library():
    """dlopen libmpe_hip.so.  Raises MpeError when it has not been built — no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise MpeError("%s is missing: run __graft_entry__.build() / make -C %s (there is no CPU fallback)"
                       % (_LIB, _CSRC))
    try:
        import torch  # noqa: F401  (so that one HIP runtime — the one torch loaded — serves both)
    except Exception:
        pass
    lib = C.CDLL(_LIB)
    lib.mpe_version.restype = C.c_char_p
    lib.mpe_last_error.restype = C.c_char_p
    lib.mpe_last_error.argtypes = [C.c_void_p]
    lib.mpe_get_stream.restype = C.c_void_p
    lib.mpe_get_stream.argtypes = [C.c_void_p]
    lib.mpe_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    lib.mpe_destroy.argtypes = [C.c_void_p]
    lib.mpe_destroy.restype = None
    lib.mpe_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.mpe_synchronize.argtypes = [C.c_void_p]
    lib.mpe_set_profiling.argtypes = [C.c_void_p, C.c_int]
    lib.mpe_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.mpe_last_launch_shape.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.mpe_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    lib.mpe_default_params.argtypes = [C.POINTER(MpeParams)]
    lib.mpe_default_params.restype = None
    dp = C.POINTER(C.c_double)
    lib.mpe_find_leds.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.POINTER(MpeParams), dp, dp, C.c_int, dp, C.POINTER(C.c_float), C.c_int,
                                  C.POINTER(C.c_int)]
    lib.mpe_solve_bruteforce.argtypes = [C.c_void_p, dp, C.c_int, dp, C.c_int, dp, C.POINTER(MpeParams),
                                         C.POINTER(MpeResult), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.mpe_estimate_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t,
                                       C.c_int, dp, C.c_int, dp, dp, C.c_int, C.POINTER(MpeParams), C.c_void_p]
    lib.mpe_estimate_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, dp, C.c_int, dp, dp,
                                              C.c_int, C.POINTER(MpeParams), C.c_void_p]
    lib.mpe_estimate_batch_device_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, dp, C.c_int, dp,
                                                     dp, C.c_int, C.POINTER(MpeParams), C.c_void_p, C.c_void_p, C.c_int]
    lib.mpe_estimate_batch_device_collect.argtypes = [C.c_void_p, C.c_void_p]
    lib.mpe_stream_next_ready.argtypes = [C.c_void_p, C.c_void_p]
    lib.mpe_stream_drop_prefetch.argtypes = [C.c_void_p]
    lib.mpe_track_step_batch_collect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mpe_track_step_batch_cancel.argtypes = [C.c_void_p]
    lib.mpe_estimate_batch_multi_device_gather.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, dp,
                                                           C.c_int, dp, dp, C.c_int, C.POINTER(MpeParams), C.c_void_p,
                                                           C.POINTER(C.c_int)]
    lib.mpe_convert_to_mono8.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_size_t, C.c_size_t, C.c_void_p, C.c_int]
    lib.mpe_detect_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_int,
                                     dp, dp, C.c_int, C.POINTER(MpeParams), C.c_void_p]
    lib.mpe_vote_batch.argtypes = [C.c_void_p, dp, C.POINTER(C.c_int), C.c_int, dp, C.c_int, dp, C.c_double,
                                   C.POINTER(C.c_uint32)]
    lib.mpe_vote_items.argtypes = [C.c_void_p, dp, C.POINTER(C.c_int), C.c_int, dp, C.c_int, dp, C.c_double,
                                   C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint32)]
    lib.mpe_check_and_refine.argtypes = [C.c_void_p, dp, C.c_int, dp, C.c_int, dp, C.POINTER(MpeParams),
                                         C.POINTER(C.c_uint32), C.c_int, C.POINTER(MpeResult)]
    lib.mpe_tracker_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.mpe_tracker_destroy.argtypes = [C.c_void_p]
    lib.mpe_tracker_destroy.restype = None
    lib.mpe_tracker_set_markers.argtypes = [C.c_void_p, dp, C.c_int]
    lib.mpe_tracker_set_camera.argtypes = [C.c_void_p, dp, dp, C.c_int]
    lib.mpe_tracker_set_params.argtypes = [C.c_void_p, C.POINTER(MpeParams)]
    lib.mpe_tracker_reset.argtypes = [C.c_void_p]
    lib.mpe_tracker_estimate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_double,
                                         C.POINTER(MpeResult), C.POINTER(C.c_int)]
    lib.mpe_tracker_run_sequence.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t,
                                             C.c_size_t, dp, C.c_void_p, C.c_void_p]
    lib.mpe_alloc_pinned.restype = C.c_void_p
    lib.mpe_alloc_pinned.argtypes = [C.c_size_t]
    lib.mpe_free_pinned.restype = None
    lib.mpe_free_pinned.argtypes = [C.c_void_p]
    hp = C.POINTER(C.c_void_p)
    lib.mpe_tracker_estimate_batch.argtypes = [hp, C.c_int, hp, C.c_int, C.c_int, C.c_size_t, dp, C.c_void_p, C.c_void_p,
                                               C.c_void_p]
    lib.mpe_tracker_run_sequences_batch.argtypes = [hp, C.c_int, hp, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t,
                                                    dp, C.c_void_p, C.c_void_p]
    lib.mpe_tracker_run_sequences_batch_threads.argtypes = [hp, C.c_int, hp, C.c_int, C.c_int, C.c_int, C.c_size_t,
                                                            C.c_size_t, dp, C.c_void_p, C.c_void_p, C.c_int]
    lib.mpe_shard_bounds.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.mpe_shard_bounds.restype = None
    lib.mpe_estimate_batch_multi.argtypes = [hp, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t,
                                             dp, C.c_int, dp, dp, C.c_int, C.POINTER(MpeParams), C.c_void_p]
    lib.mpe_estimate_batch_multi_device.argtypes = [hp, C.c_int, hp, C.POINTER(C.c_int), C.c_int, C.c_int, dp, C.c_int,
                                                    dp, dp, C.c_int, C.POINTER(MpeParams), C.c_void_p]
    _lib = lib
    return lib


def tracker_estimate_batch(trackers, imgs, times):
    """mpe_tracker_estimate_batch: frame k of N trackers (same handle / camera / markers / parameters) in lock step,
    one device submission per step in steady state.  imgs: list of (rows, cols) uint8 arrays; times: N floats.
    -> (records [RESULT_DTYPE] (N), info (N,8) int32, updated (N) bool)."""
    lib = load_library()
    n = len(trackers)
    imgs = [np.ascontiguousarray(im, np.uint8) for im in imgs]
    rows, cols = imgs[0].shape
    stride = imgs[0].strides[0]
    assert all(im.shape == (rows, cols) and im.strides[0] == stride for im in imgs)
    ts = (C.c_void_p * n)(*[t._t for t in trackers])
    ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
    times = _f64(times).reshape(-1)
    rec = np.zeros(n, RESULT_DTYPE)
    info = np.zeros((n, 8), np.int32)
    upd = np.zeros(n, np.int32)
    rc = lib.mpe_tracker_estimate_batch(ts, n, ptrs, rows, cols, stride, _dp(times), rec.ctypes.data, info.ctypes.data,
                                        upd.ctypes.data)
    if rc < 0:
        raise MpeError("mpe_tracker_estimate_batch failed (%d): %s"
                       % (rc, lib.mpe_last_error(trackers[0]._handle._h).decode()))
    return rec, info, upd.astype(bool)


def tracker_run_sequences_batch(trackers, frames, times, threads=1):
    """mpe_tracker_run_sequences_batch[_threads]: the lock-step loop in C.  frames: list of (n,rows,cols) uint8 arrays
    (one sequence per tracker), times: n floats; threads > 1: the handle groups on that many host threads.
    -> (records (N,n), info (N,n,8))."""
    lib = load_library()
    N = len(trackers)
    frames = [np.ascontiguousarray(f, np.uint8) for f in frames]
    n, rows, cols = frames[0].shape
    assert all(f.shape == (n, rows, cols) for f in frames)
    ts = (C.c_void_p * N)(*[t._t for t in trackers])
    ptrs = (C.c_void_p * N)(*[f.ctypes.data for f in frames])
    times = _f64(times).reshape(-1)
    rec = np.zeros((N, n), RESULT_DTYPE)
    info = np.zeros((N, n, 8), np.int32)
    rc = lib.mpe_tracker_run_sequences_batch_threads(ts, N, ptrs, n, rows, cols, frames[0].strides[1],
                                                     frames[0].strides[0], _dp(times), rec.ctypes.data,
                                                     info.ctypes.data, int(threads))
    if rc < 0:
        raise MpeError("mpe_tracker_run_sequences_batch failed (%d): %s"
                       % (rc, lib.mpe_last_error(trackers[0]._handle._h).decode()))
    return rec, info


class PinnedFrames:
    """(n, rows, cols) uint8 frame buffer in page-locked host memory (mpe_alloc_pinned): `.array` is a numpy view."""

    def __init__(self, n, rows, cols):
        self._lib = load_library()
        self._p = self._lib.mpe_alloc_pinned(n * rows * cols)
        if not self._p:
            raise MpeError("mpe_alloc_pinned(%d) failed" % (n * rows * cols))
        self.array = np.ctypeslib.as_array((C.c_uint8 * (n * rows * cols)).from_address(self._p)).reshape(n, rows, cols)

    def close(self):
        if getattr(self, "_p", None):
            self.array = None
            self._lib.mpe_free_pinned(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_bounds(n_frames, shard, n_shards):
    """mpe_shard_bounds: contiguous chunk [lo, hi) of shard `shard` (host arithmetic, no device)."""
    lo, hi = C.c_int(), C.c_int()
    load_library().mpe_shard_bounds(int(n_frames), int(shard), int(n_shards), C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def estimate_batch_multi(handles, frames, markers, K, D, params):
    """mpe_estimate_batch_multi: one host process, the batch sharded over several handles (= GPUs).
    frames: numpy (n,rows,cols) uint8 on the host, or a LIST of torch uint8 CUDA tensors, one per handle, each
    resident on its handle's device (mpe_estimate_batch_multi_device).  -> numpy records in frame order."""
    lib = load_library()
    markers = _f64(markers).reshape(-1, 3)
    K = _f64(K).reshape(9)
    D = _f64(D).reshape(-1)
    hs = (C.c_void_p * len(handles))(*[h._h for h in handles])
    if isinstance(frames, (list, tuple)):
        assert len(frames) == len(handles)
        rows, cols = frames[0].shape[1:]
        ptrs = (C.c_void_p * len(frames))(*[f.data_ptr() for f in frames])
        counts = (C.c_int * len(frames))(*[f.shape[0] for f in frames])
        out = np.zeros(sum(f.shape[0] for f in frames), RESULT_DTYPE)
        rc = lib.mpe_estimate_batch_multi_device(hs, len(handles), ptrs, counts, rows, cols, _dp(markers), len(markers),
                                                 _dp(K), _dp(D), len(D), C.byref(params), C.c_void_p(out.ctypes.data))
    else:
        frames = np.ascontiguousarray(frames, np.uint8)
        n, rows, cols = frames.shape
        out = np.zeros(n, RESULT_DTYPE)
        rc = lib.mpe_estimate_batch_multi(hs, len(handles), C.c_void_p(frames.ctypes.data), n, rows, cols,
                                          frames.strides[1], frames.strides[0], _dp(markers), len(markers), _dp(K),
                                          _dp(D), len(D), C.byref(params), C.c_void_p(out.ctypes.data))
    if rc != 0:
        msgs = [lib.mpe_last_error(h._h).decode() for h in handles]
        raise MpeError("mpe_estimate_batch_multi failed (%d): %s" % (rc, "; ".join(m for m in msgs if m)))
    return out


def estimate_batch_multi_device_gather(handles, frames, markers, K, D, params):
    """mpe_estimate_batch_multi_device_gather: device-resident shards (a LIST of torch uint8 CUDA tensors, one per
    handle, each on its handle's device) -> ONE torch uint8 tensor of records on handles[0]'s device, gathered GPU to
    GPU (RCCL over xGMI when the handles sit on distinct devices).  -> (records as numpy, used_rccl)."""
    import torch
    lib = load_library()
    markers = _f64(markers).reshape(-1, 3)
    K = _f64(K).reshape(9)
    D = _f64(D).reshape(-1)
    hs = (C.c_void_p * len(handles))(*[h._h for h in handles])
    rows, cols = frames[0].shape[1:]
    ptrs = (C.c_void_p * len(frames))(*[f.data_ptr() for f in frames])
    counts = (C.c_int * len(frames))(*[f.shape[0] for f in frames])
    n = sum(f.shape[0] for f in frames)
    out = torch.zeros(n * RESULT_DTYPE.itemsize, dtype=torch.uint8, device=frames[0].device)
    torch.cuda.synchronize()
    used = C.c_int(0)
    rc = lib.mpe_estimate_batch_multi_device_gather(hs, len(handles), ptrs, counts, rows, cols, _dp(markers), len(markers),
                                                    _dp(K), _dp(D), len(D), C.byref(params), C.c_void_p(out.data_ptr()),
                                                    C.byref(used))
    if rc != 0:
        msgs = [lib.mpe_last_error(h._h).decode() for h in handles]
        raise MpeError("mpe_estimate_batch_multi_device_gather failed (%d): %s" % (rc, "; ".join(m for m in msgs if m)))
    return np.frombuffer(out.cpu().numpy().tobytes(), RESULT_DTYPE), bool(used.value)


def determine_roi(px, rows, cols, border, K, D):
    """LEDDetector::determineROI (host arithmetic inside libmpe_hip.so; needs no device)."""
    lib = load_library()
    px = _f64(px).reshape(-1, 2)
    K = _f64(K).reshape(9)
    D = _f64(D).reshape(-1)
    roi = (C.c_int * 4)()
    rc = lib.mpe_determine_roi(_dp(px), len(px), int(rows), int(cols), int(border), _dp(K), _dp(D), len(D), roi)
    if rc != 0:
        raise MpeError("mpe_determine_roi failed (%d)" % rc)
    return tuple(roi)


def distort_points(xy, K, D):
    """LEDDetector::distortPoints, float32 in / out (host arithmetic)."""
    lib = load_library()
    s = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    d = np.zeros_like(s)
    K = _f64(K).reshape(9)
    D = _f64(D).reshape(-1)
    rc = lib.mpe_distort_points(s.ctypes.data_as(C.POINTER(C.c_float)), d.ctypes.data_as(C.POINTER(C.c_float)), len(s),
                                _dp(K), _dp(D), len(D))
    if rc != 0:
        raise MpeError("mpe_distort_points failed (%d)" % rc)
    return d


def exponential_map(twist):
    T = np.zeros(16)
    load_library().mpe_exponential_map(_dp(_f64(twist).reshape(6)), _dp(T))
    return T.reshape(4, 4)


def logarithm_map(T):
    xi = np.zeros(6)
    load_library().mpe_logarithm_map(_dp(_f64(T).reshape(16)), _dp(xi))
    return xi


def predict_pose(current, previous, t_current, t_previous, t_predict):
    """predictPose (pose_estimator.cpp:232-244), host arithmetic."""
    out = np.zeros(16)
    lib = load_library()
    lib.mpe_predict_pose.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double, C.c_double,
                                     C.POINTER(C.c_double)]
    lib.mpe_predict_pose(_dp(_f64(current).reshape(16)), _dp(_f64(previous).reshape(16)), t_current, t_previous,
                         t_predict, _dp(out))
    return out.reshape(4, 4)


def project_points(T, markers, K):
    markers = _f64(markers).reshape(-1, 3)
    px = np.zeros((len(markers), 2))
    load_library().mpe_project_points(_dp(_f64(T).reshape(16)), _dp(markers), len(markers), _dp(_f64(K).reshape(9)),
                                      _dp(px))
    return px


def find_correspondences(pred_px, det, tol):
    pred = _f64(pred_px).reshape(-1, 2)
    det = _f64(det).reshape(-1, 2)
    corr = np.zeros((len(pred) + 1, 2), np.uint32)
    lib = load_library()
    lib.mpe_find_correspondences.argtypes = [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int, C.c_double,
                                             C.c_void_p]
    n = lib.mpe_find_correspondences(_dp(pred), len(pred), _dp(det), len(det), float(tol), C.c_void_p(corr.ctypes.data))
    if n < 0:
        raise MpeError("mpe_find_correspondences failed (%d)" % n)
    return corr[:n].copy()


ENCODINGS = {"mono8": 0, "bgr8": 1, "rgb8": 2, "bgra8": 3, "rgba8": 4, "mono16": 5}  # MPE_ENC_*


def demo_params(**kw):
    """Parameter set of launch/demo.launch:12-22 (overridable by keyword)."""
    p = MpeParams()
    load_library().mpe_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class Handle:
    """One GPU + one HIP stream (mpe_handle)."""

    def __init__(self, device=-1):
        self._lib = load_library()
        self._h = C.c_void_p()
        rc = self._lib.mpe_create(C.byref(self._h), int(device))
        if rc != 0:
            self._h = None
            raise MpeError("mpe_create failed (%d): no usable HIP device — there is no CPU fallback" % rc)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mpe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise MpeError("%s failed (%d): %s" % (what, rc, self._lib.mpe_last_error(self._h).decode()))

    # ---- stream / options -------------------------------------------------------------------
    def set_stream(self, stream_ptr):
        self._check(self._lib.mpe_set_stream(self._h, C.c_void_p(stream_ptr or 0)), "mpe_set_stream")

    def synchronize(self):
        self._check(self._lib.mpe_synchronize(self._h), "mpe_synchronize")

    def get_option(self, name):
        v = C.c_int(0)
        self._check(self._lib.mpe_get_option(self._h, name.encode(), C.byref(v)), "mpe_get_option")
        return v.value

    def set_option(self, name, value):
        self._check(self._lib.mpe_set_option(self._h, name.encode(), int(value)), "mpe_set_option")

    def set_profiling(self, on):
        self._check(self._lib.mpe_set_profiling(self._h, 1 if on else 0), "mpe_set_profiling")

    def last_kernel_ms_sub(self, sub_batch):
        ms = (C.c_float * 4)()
        self._check(self._lib.mpe_last_kernel_ms_sub(self._h, int(sub_batch), ms), "mpe_last_kernel_ms_sub")
        return dict(scan=ms[0], blobs=ms[1], vote=ms[2], tail=ms[3])

    def last_kernel_ms(self):
        ms = (C.c_float * 5)()
        self._check(self._lib.mpe_last_kernel_ms(self._h, ms), "mpe_last_kernel_ms")
        nl, fpl = C.c_int(0), C.c_int(0)
        self._check(self._lib.mpe_last_launch_shape(self._h, C.byref(nl), C.byref(fpl)), "mpe_last_launch_shape")
        return dict(scan=ms[0], blobs=ms[1], vote=ms[2], tail=ms[3], total=ms[4], launches=nl.value,
                    frames_per_launch=fpl.value)

    # ---- LEDDetector::findLeds ----------------------------------------------------------------
    def find_leds(self, img, params, K, D, roi=None, cap=MAX_DETECTIONS):
        img = np.ascontiguousarray(img, np.uint8)
        rows, cols = img.shape
        rx, ry, rw, rh = roi if roi is not None else (0, 0, cols, rows)
        K = _f64(K).reshape(9)
        D = _f64(D).reshape(-1)
        und = np.zeros((cap, 2))
        dst = np.zeros((cap, 2), np.float32)
        n = C.c_int(0)
        rc = self._lib.mpe_find_leds(self._h, img.ctypes.data, rows, cols, img.strides[0], rx, ry, rw, rh,
                                     C.byref(params), _dp(K), _dp(D), len(D), _dp(und),
                                     dst.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(n))
        self._check(rc, "mpe_find_leds")
        return und[:n.value].copy(), dst[:n.value].copy()

    # ---- setImagePoints + initialise + optimiseAndUpdatePose ---------------------------------
    def solve_bruteforce(self, det, markers, K, params):
        det = _f64(det).reshape(-1, 2)
        markers = _f64(markers).reshape(-1, 3)
        K = _f64(K).reshape(9)
        res = MpeResult()
        hist = np.zeros((max(len(det), 1), len(markers)), np.uint32)
        corr = np.zeros((len(markers), 2), np.uint32)
        rc = self._lib.mpe_solve_bruteforce(self._h, _dp(det), len(det), _dp(markers), len(markers), _dp(K),
                                            C.byref(params), C.byref(res),
                                            hist.ctypes.data_as(C.POINTER(C.c_uint32)),
                                            corr.ctypes.data_as(C.POINTER(C.c_uint32)))
        self._check(rc, "mpe_solve_bruteforce")
        return dict(status=res.status, T=np.array(res.T).reshape(4, 4), cov=np.array(res.cov).reshape(6, 6),
                    n_det=res.n_det, n_corr=res.n_corr, gn_iterations=res.gn_iterations,
                    hist=hist[:len(det)].copy(), corr=corr[:res.n_corr].copy())

    # ---- setCorrespondences + checkCorrespondences + optimiseAndUpdatePose ----------------------
    def check_and_refine(self, det, markers, K, params, corr):
        det = _f64(det).reshape(-1, 2)
        markers = _f64(markers).reshape(-1, 3)
        K = _f64(K).reshape(9)
        corr = np.ascontiguousarray(corr, np.uint32).reshape(-1, 2)
        res = MpeResult()
        rc = self._lib.mpe_check_and_refine(self._h, _dp(det), len(det), _dp(markers), len(markers), _dp(K),
                                            C.byref(params), corr.ctypes.data_as(C.POINTER(C.c_uint32)), len(corr),
                                            C.byref(res))
        self._check(rc, "mpe_check_and_refine")
        return dict(status=res.status, T=np.array(res.T).reshape(4, 4), cov=np.array(res.cov).reshape(6, 6),
                    n_corr=res.n_corr, gn_iterations=res.gn_iterations)

    def check_correspondences(self, det, markers, K, params, corr):
        """checkCorrespondences alone -> (ok, unrefined T)."""
        det = _f64(det).reshape(-1, 2)
        markers = _f64(markers).reshape(-1, 3)
        K = _f64(K).reshape(9)
        corr = np.ascontiguousarray(corr, np.uint32).reshape(-1, 2)
        res = MpeResult()
        rc = self._lib.mpe_check_correspondences(self._h, _dp(det), len(det), _dp(markers), len(markers), _dp(K),
                                                 C.byref(params), corr.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                 len(corr), C.byref(res))
        self._check(rc, "mpe_check_correspondences")
        return res.status == 0, np.array(res.T).reshape(4, 4)

    def optimise_pose(self, det, markers, K, params, corr, T_init):
        """optimisePose alone from T_init -> dict(status, T, cov, gn_iterations)."""
        det = _f64(det).reshape(-1, 2)
        markers = _f64(markers).reshape(-1, 3)
        K = _f64(K).reshape(9)
        T0 = _f64(T_init).reshape(16)
        corr = np.ascontiguousarray(corr, np.uint32).reshape(-1, 2)
        res = MpeResult()
        rc = self._lib.mpe_optimise_pose(self._h, _dp(det), len(det), _dp(markers), len(markers), _dp(K),
                                         C.byref(params), corr.ctypes.data_as(C.POINTER(C.c_uint32)), len(corr),
                                         _dp(T0), C.byref(res))
        self._check(rc, "mpe_optimise_pose")
        return dict(status=res.status, T=np.array(res.T).reshape(4, 4), cov=np.array(res.cov).reshape(6, 6),
                    gn_iterations=res.gn_iterations)

    # ---- one tracked frame: findLeds(ROI) + findCorrespondences + checkCorrespondences + optimisePose ----
    def track_step(self, img, roi, params, K, D, markers, predicted_px):
        img = np.ascontiguousarray(img, np.uint8)
        markers = _f64(markers).reshape(-1, 3)
        K = _f64(K).reshape(9)
        D = _f64(D).reshape(-1)
        pred = _f64(predicted_px).reshape(-1, 2)
        assert len(pred) == len(markers)
        det = MpeDetections()
        corr = np.zeros((MAX_MARKERS, 2), np.uint32)
        res = MpeResult()
        rc = self._lib.mpe_track_step(self._h, C.c_void_p(img.ctypes.data), img.shape[0], img.shape[1],
                                      C.c_size_t(img.strides[0]), int(roi[0]), int(roi[1]), int(roi[2]), int(roi[3]),
                                      C.byref(params), _dp(K), _dp(D), len(D), _dp(markers), len(markers), _dp(pred),
                                      C.byref(det), C.c_void_p(corr.ctypes.data), C.byref(res))
        self._check(rc, "mpe_track_step")
        n = max(det.n, 0)
        return dict(status=res.status, T=np.array(res.T).reshape(4, 4), cov=np.array(res.cov).reshape(6, 6),
                    n_corr=res.n_corr, corr=corr[:max(res.n_corr, 0)].copy(), det_status=det.status,
                    undist=np.array(det.undist_xy[:2 * n]).reshape(-1, 2), gn_iterations=res.gn_iterations)

    # ---- static primitives, batched ----------------------------------------------------------------
    def p3p_batch(self, fv, wp):
        """fv, wp: (n,3,3), ROWS = bearings / world points.  -> (status (n,), solutions (n,4,3,4) [R|C])."""
        fv = _f64(fv).reshape(-1, 9)
        wp = _f64(wp).reshape(-1, 9)
        n = len(fv)
        sol = np.zeros((n, 4, 3, 4))
        st = np.zeros(n, np.int32)
        self._check(self._lib.mpe_p3p_batch(self._h, _dp(fv), _dp(wp), n, _dp(sol), C.c_void_p(st.ctypes.data)),
                    "mpe_p3p_batch")
        return st, sol

    def solve_quartic_batch(self, factors, variant=0):
        f = _f64(factors).reshape(-1, 5)
        r = np.zeros((len(f), 4))
        self._check(self._lib.mpe_solve_quartic_batch(self._h, _dp(f), len(f), int(variant), _dp(r)),
                    "mpe_solve_quartic_batch")
        return r

    # ---- estimateBodyPose on a fresh estimator per frame -------------------------------------
    def estimate_batch(self, frames, markers, K, D, params):
        """frames: numpy (n,rows,cols) uint8 on the host, or a torch uint8 CUDA tensor (n,rows,cols)."""
        markers = _f64(markers).reshape(-1, 3)
        K = _f64(K).reshape(9)
        D = _f64(D).reshape(-1)
        if _is_torch(frames):
            assert frames.is_cuda and frames.is_contiguous() and frames.dtype.is_floating_point is False
            n, rows, cols = frames.shape
            ptr, on_dev, stride, fstride = frames.data_ptr(), 1, cols, rows * cols
        else:
            frames = np.ascontiguousarray(frames, np.uint8)
            n, rows, cols = frames.shape
            ptr, on_dev, stride, fstride = frames.ctypes.data, 0, frames.strides[1], frames.strides[0]
        out = np.zeros(n, RESULT_DTYPE)
        rc = self._lib.mpe_estimate_batch(self._h, C.c_void_p(ptr), n, rows, cols, stride, fstride, on_dev,
                                          _dp(markers), len(markers), _dp(K), _dp(D), len(D), C.byref(params),
                                          C.c_void_p(out.ctypes.data))
        self._check(rc, "mpe_estimate_batch")
        return out

    def estimate_batch_device(self, d_frames_ptr, n, rows, cols, markers, K, D, params, d_results_ptr):
        """Fully asynchronous: device frames in, device results out, kernels only."""
        markers = _f64(markers).reshape(-1, 3)
        K = _f64(K).reshape(9)
        D = _f64(D).reshape(-1)
        rc = self._lib.mpe_estimate_batch_device(self._h, C.c_void_p(d_frames_ptr), n, rows, cols, _dp(markers),
                                                 len(markers), _dp(K), _dp(D), len(D), C.byref(params),
                                                 C.c_void_p(d_results_ptr))
        self._check(rc, "mpe_estimate_batch_device")

    def convert_to_mono8(self, src, encoding, big_endian=False):
        """mpe_convert_to_mono8 for a batch: src (n, rows, cols[, channels]) uint8 / (n, rows, cols) uint16 numpy array
        (host) or torch CUDA tensor; encoding "mono8" | "bgr8" | "rgb8" | "bgra8" | "rgba8" | "mono16".
        -> (n, rows, cols) uint8, numpy for a host source, a torch CUDA tensor for a device source."""
        enc = ENCODINGS[encoding]
        if _is_torch(src):
            import torch
            src = src.contiguous()
            n, rows, cols = src.shape[:3]
            bpp = src.element_size() * (src.shape[3] if src.dim() == 4 else 1)
            dst = torch.empty((n, rows, cols), dtype=torch.uint8, device=src.device)
            rc = self._lib.mpe_convert_to_mono8(self._h, C.c_void_p(src.data_ptr()), 1, enc, int(bool(big_endian)), n, rows,
                                                cols, C.c_size_t(cols * bpp), C.c_size_t(rows * cols * bpp),
                                                C.c_void_p(dst.data_ptr()), 1)
            self._check(rc, "mpe_convert_to_mono8")
            self.synchronize()
            return dst
        src = np.ascontiguousarray(src)
        n, rows, cols = src.shape[:3]
        raw = src.view(np.uint8).reshape(n, rows, -1)
        dst = np.zeros((n, rows, cols), np.uint8)
        rc = self._lib.mpe_convert_to_mono8(self._h, C.c_void_p(raw.ctypes.data), 0, enc, int(bool(big_endian)), n, rows, cols,
                                            C.c_size_t(raw.strides[1]), C.c_size_t(raw.strides[0]),
                                            C.c_void_p(dst.ctypes.data), 0)
        self._check(rc, "mpe_convert_to_mono8")
        return dst

    def estimate_batch_device_submit(self, d_frames_ptr, n, rows, cols, markers, K, D, params, d_results_ptr,
                                     d_next_frames_ptr=0, n_next=0):
        """Streaming variant: enqueue only, no join of the internal side streams; `d_next_frames_ptr` announces the
        frames of the next submission (its first sub-batch is then scanned by this one's last voting launch)."""
        markers = _f64(markers).reshape(-1, 3)
        K = _f64(K).reshape(9)
        D = _f64(D).reshape(-1)
        rc = self._lib.mpe_estimate_batch_device_submit(self._h, C.c_void_p(d_frames_ptr), n, rows, cols, _dp(markers),
                                                        len(markers), _dp(K), _dp(D), len(D), C.byref(params),
                                                        C.c_void_p(d_results_ptr), C.c_void_p(d_next_frames_ptr or None),
                                                        int(n_next))
        self._check(rc, "mpe_estimate_batch_device_submit")

    def stream_next_ready(self, event_ptr):
        """The frames the NEXT _submit announces are final once this hipEvent_t has completed (one-shot)."""
        self._check(self._lib.mpe_stream_next_ready(self._h, C.c_void_p(event_ptr or None)), "mpe_stream_next_ready")

    def stream_drop_prefetch(self):
        """Forget what the last submission scanned ahead: the announced buffer has been rewritten since."""
        self._check(self._lib.mpe_stream_drop_prefetch(self._h), "mpe_stream_drop_prefetch")

    def estimate_batch_device_collect(self, stream_ptr=0):
        """Make `stream_ptr` (0 = the handle's stream) wait for the records of the oldest un-collected submission."""
        self._check(self._lib.mpe_estimate_batch_device_collect(self._h, C.c_void_p(stream_ptr or None)),
                    "mpe_estimate_batch_device_collect")

    # ---- stage level ---------------------------------------------------------------------------
    def detect_batch(self, frames, K, D, params):
        K = _f64(K).reshape(9)
        D = _f64(D).reshape(-1)
        if _is_torch(frames):
            n, rows, cols = frames.shape
            ptr, on_dev, stride, fstride = frames.data_ptr(), 1, cols, rows * cols
        else:
            frames = np.ascontiguousarray(frames, np.uint8)
            n, rows, cols = frames.shape
            ptr, on_dev, stride, fstride = frames.ctypes.data, 0, frames.strides[1], frames.strides[0]
        out = np.zeros(n, DETECTIONS_DTYPE)
        rc = self._lib.mpe_detect_batch(self._h, C.c_void_p(ptr), n, rows, cols, stride, fstride, on_dev, _dp(K),
                                        _dp(D), len(D), C.byref(params), C.c_void_p(out.ctypes.data))
        self._check(rc, "mpe_detect_batch")
        return out

    def vote_batch(self, dets, markers, K, tol):
        """dets: list of (n_i,2) arrays.  -> list of (n_i, n_markers) uint32 histograms."""
        markers = _f64(markers).reshape(-1, 3)
        K = _f64(K).reshape(9)
        n = len(dets)
        buf = np.zeros((n, MAX_DETECTIONS, 2))
        nd = np.zeros(n, np.int32)
        for i, d in enumerate(dets):
            d = _f64(d).reshape(-1, 2)
            nd[i] = len(d)
            buf[i, :len(d)] = d
        hist = np.zeros((n, MAX_DETECTIONS, MAX_MARKERS), np.uint32)
        rc = self._lib.mpe_vote_batch(self._h, _dp(buf), nd.ctypes.data_as(C.POINTER(C.c_int)), n, _dp(markers),
                                      len(markers), _dp(K), float(tol), hist.ctypes.data_as(C.POINTER(C.c_uint32)))
        self._check(rc, "mpe_vote_batch")
        return [hist[i, :nd[i], :len(markers)].copy() for i in range(n)]


    def vote_items(self, det, markers, K, tol, lo, hi):
        """Forensics: the detection set `det` voted len(lo) times, copy i with the hypotheses [lo[i], hi[i]) only
        (flattened index = triple index * P(n_markers,3) + permutation index).  -> (len(lo), n_det, n_markers)."""
        markers = _f64(markers).reshape(-1, 3)
        K = _f64(K).reshape(9)
        det = _f64(det).reshape(-1, 2)
        lo = np.ascontiguousarray(lo, np.int32)
        hi = np.ascontiguousarray(hi, np.int32)
        n = len(lo)
        buf = np.zeros((n, MAX_DETECTIONS, 2))
        buf[:, :len(det)] = det
        nd = np.full(n, len(det), np.int32)
        hist = np.zeros((n, MAX_DETECTIONS, MAX_MARKERS), np.uint32)
        ip = C.POINTER(C.c_int)
        rc = self._lib.mpe_vote_items(self._h, _dp(buf), nd.ctypes.data_as(ip), n, _dp(markers), len(markers), _dp(K),
                                      float(tol), lo.ctypes.data_as(ip), hi.ctypes.data_as(ip),
                                      hist.ctypes.data_as(C.POINTER(C.c_uint32)))
        self._check(rc, "mpe_vote_items")
        return hist[:, :len(det), :len(markers)].copy()


class Tracker:
    """mpe_tracker: one stateful PoseEstimator object (uninitialised branch + tracking path)."""

    def __init__(self, handle, markers, K, D, params):
        self._handle = handle
        self._lib = load_library()
        self._t = C.c_void_p()
        rc = self._lib.mpe_tracker_create(handle._h, C.byref(self._t))
        if rc != 0:
            raise MpeError("mpe_tracker_create failed (%d)" % rc)
        markers = _f64(markers).reshape(-1, 3)
        K = _f64(K).reshape(9)
        D = _f64(D).reshape(-1)
        handle._check(self._lib.mpe_tracker_set_markers(self._t, _dp(markers), len(markers)), "mpe_tracker_set_markers")
        handle._check(self._lib.mpe_tracker_set_camera(self._t, _dp(K), _dp(D), len(D)), "mpe_tracker_set_camera")
        self.set_params(params)

    def set_params(self, params):
        self._handle._check(self._lib.mpe_tracker_set_params(self._t, C.byref(params)), "mpe_tracker_set_params")

    def reset(self):
        self._lib.mpe_tracker_reset(self._t)

    def estimate(self, img, time):
        img = np.ascontiguousarray(img, np.uint8)
        res = MpeResult()
        info = (C.c_int * 8)()
        rc = self._lib.mpe_tracker_estimate(self._t, img.ctypes.data, img.shape[0], img.shape[1], img.strides[0],
                                            float(time), C.byref(res), info)
        if rc < 0:
            raise MpeError("mpe_tracker_estimate failed (%d): %s" % (rc, self._lib.mpe_last_error(self._handle._h).decode()))
        return dict(updated=bool(rc), T=np.array(res.T).reshape(4, 4), cov=np.array(res.cov).reshape(6, 6),
                    roi=tuple(info[0:4]), it_since_initialized=info[4], n_det=info[5], n_corr=info[6],
                    used_bruteforce=bool(info[7]))

    def run_sequence(self, frames, times):
        """The image-callback loop in C over a recorded sequence (frames (n,rows,cols) uint8, C-contiguous).
        -> (records [RESULT_DTYPE], info (n,8) int32).  ctypes drops the GIL for the duration."""
        frames = np.ascontiguousarray(frames, np.uint8)
        times = _f64(times).reshape(-1)
        n = frames.shape[0]
        rec = np.zeros(n, RESULT_DTYPE)
        info = np.zeros((n, 8), np.int32)
        rc = self._lib.mpe_tracker_run_sequence(self._t, frames.ctypes.data, n, frames.shape[1], frames.shape[2],
                                                frames.strides[1], frames.strides[0], _dp(times), rec.ctypes.data,
                                                info.ctypes.data)
        if rc < 0:
            raise MpeError("mpe_tracker_run_sequence failed (%d): %s"
                           % (rc, self._lib.mpe_last_error(self._handle._h).decode()))
        return rec, info

    def close(self):
        if getattr(self, "_t", None):
            self._lib.mpe_tracker_destroy(self._t)
            self._t = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
