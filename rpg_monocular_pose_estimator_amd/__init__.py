"""MI355X (gfx950) compute back-end for the per-frame hot path of rpg_monocular_pose_estimator.

The product is the HIP library ``libmpe_hip.so`` (C ABI: ``include/mpe.h``).  This package is the
thin host-side binding used by tests / bench: ctypes over the C ABI, numpy or torch buffers in, numpy
out.  There is NO CPU fallback: if the library or a HIP device is missing every call raises.
"""
from .binding import (MpeError, MpeParams, MpeResult, MpeDetections, RESULT_DTYPE, DETECTIONS_DTYPE,  # noqa: F401
                      MAX_DETECTIONS, MAX_MARKERS, Handle, build_library, library_path, load_library,
                      demo_params, exported_symbols, source_fingerprint, device_source, DEVICE_SOURCES, Tracker, determine_roi, distort_points, exponential_map, logarithm_map,
                      predict_pose, project_points, find_correspondences, shard_bounds, estimate_batch_multi,
                      estimate_batch_multi_device_gather, ENCODINGS,
                      tracker_estimate_batch, tracker_run_sequences_batch, PinnedFrames)
from .pose_estimator import PoseEstimator  # noqa: F401

__all__ = ["MpeError", "MpeParams", "MpeResult", "MpeDetections", "RESULT_DTYPE", "DETECTIONS_DTYPE",
           "MAX_DETECTIONS", "MAX_MARKERS", "Handle", "build_library", "library_path", "load_library",
           "demo_params", "exported_symbols", "source_fingerprint", "device_source", "DEVICE_SOURCES", "PoseEstimator", "Tracker", "determine_roi", "distort_points",
           "exponential_map", "logarithm_map", "predict_pose", "project_points", "find_correspondences",
           "shard_bounds", "estimate_batch_multi", "estimate_batch_multi_device_gather", "ENCODINGS", "tracker_estimate_batch", "tracker_run_sequences_batch", "PinnedFrames"]
