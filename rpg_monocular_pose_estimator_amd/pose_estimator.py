"""Host-side mirror of the reference's ``monocular_pose_estimator::PoseEstimator`` surface
(monocular_pose_estimator_lib/include/monocular_pose_estimator_lib/pose_estimator.h:52-803) on top
of the HIP C ABI, so parity tests read like code written against the reference class.

estimateBodyPose runs the reference's whole state machine (pose_estimator.cpp:62-147): brute-force
initialisation while not initialised, then prediction + ROI detection + nearest-neighbour
correspondences with fallback to brute force (mpe_tracker_* in include/mpe.h).
``bruteforce_every_frame=True`` resets the state before every call (the BASELINE configs' mode).
"""
import numpy as np

from .binding import Handle, Tracker, demo_params


class PoseEstimator:
    def __init__(self, handle=None, bruteforce_every_frame=False):
        self._bf_every_frame = bruteforce_every_frame
        self._tracker = None
        # constructor defaults, pose_estimator.cpp:34-42
        self._h = handle if handle is not None else Handle()
        self._p = demo_params()
        self._p.back_projection_pixel_tolerance = 3
        self._p.nearest_neighbour_pixel_tolerance = 5
        self._p.certainty_threshold = 0.75
        self._p.valid_correspondence_threshold = 0.7
        self._p.histogram_threshold = 0
        self.camera_matrix_K_ = None           # 3x3 (pose_estimator.h:82)
        self.camera_distortion_coeffs_ = []    # (pose_estimator.h:83)
        self._markers = np.zeros((0, 3))
        self._pose = np.eye(4)
        self._cov = np.zeros((6, 6))
        self._image_points = np.zeros((0, 2))
        self._correspondences = np.zeros((0, 2), np.uint32)
        self._distorted = np.zeros((0, 2), np.float32)
        self._time = 0.0

    # public tuning fields of the reference (pose_estimator.h:85-91) as properties on the params
    def _prop(name):  # noqa: N805
        return property(lambda s: getattr(s._p, name), lambda s, v: setattr(s._p, name, v))

    detection_threshold_value_ = _prop("threshold_value")
    gaussian_sigma_ = _prop("gaussian_sigma")
    min_blob_area_ = _prop("min_blob_area")
    max_blob_area_ = _prop("max_blob_area")
    max_width_height_distortion_ = _prop("max_width_height_distortion")
    max_circular_distortion_ = _prop("max_circular_distortion")
    roi_border_thickness_ = _prop("roi_border_thickness")
    del _prop

    def setMarkerPositions(self, positions):  # pose_estimator.cpp:50-55
        m = np.asarray(positions, np.float64)
        self._markers = m[:, :3].copy()
        self._tracker = None
        self._p.histogram_threshold = 0  # -> numCombinations(n,3) inside the library

    def getMarkerPositions(self):
        return np.hstack([self._markers, np.ones((len(self._markers), 1))])

    def setBackProjectionPixelTolerance(self, v):
        self._p.back_projection_pixel_tolerance = v

    def getBackProjectionPixelTolerance(self):
        return self._p.back_projection_pixel_tolerance

    def setNearestNeighbourPixelTolerance(self, v):
        self._p.nearest_neighbour_pixel_tolerance = v

    def getNearestNeighbourPixelTolerance(self):
        return self._p.nearest_neighbour_pixel_tolerance

    def setCertaintyThreshold(self, v):
        self._p.certainty_threshold = v

    def getCertaintyThreshold(self):
        return self._p.certainty_threshold

    def setValidCorrespondenceThreshold(self, v):
        self._p.valid_correspondence_threshold = v

    def getValidCorrespondenceThreshold(self):
        return self._p.valid_correspondence_threshold

    def setHistogramThreshold(self, v):
        self._p.histogram_threshold = int(v)

    def getHistogramThreshold(self):
        return int(self._p.histogram_threshold)

    def getPredictedPose(self):
        return self._pose.copy()

    def getPoseCovariance(self):
        return self._cov.copy()

    def getImagePoints(self):
        return self._image_points.copy()

    def getCorrespondences(self):
        return self._correspondences.copy()

    def getPredictedTime(self):
        return self._time

    def estimateBodyPose(self, image, time_to_predict):
        """pose_estimator.cpp:62-147: True iff the pose was updated."""
        K = np.asarray(self.camera_matrix_K_, np.float64)
        D = np.asarray(self.camera_distortion_coeffs_, np.float64)
        if self._tracker is None:
            self._tracker = Tracker(self._h, self._markers, K, D, self._p)
        self._tracker.set_params(self._p)
        if self._bf_every_frame:
            self._tracker.reset()
        self._time = float(time_to_predict)
        r = self._tracker.estimate(image, time_to_predict)
        self._last = r
        if r["updated"]:
            self._pose, self._cov = r["T"], r["cov"]
        return r["updated"]
