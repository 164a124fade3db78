// message_conversions.h — the ROS-independent half of the node glue (tested without ROS):
//   * pose + covariance -> the fields of geometry_msgs/PoseWithCovarianceStamped exactly as the reference
//     node fills them (monocular_pose_estimator/src/monocular_pose_estimator.cpp:170-187): position =
//     T(0:3,3), orientation = Eigen::Quaterniond(R) (Eigen's matrix -> quaternion rule restated below),
//     covariance = the 6x6 in row-major order, twist order (upsilon, omega);
//   * sensor_msgs/CameraInfo K / D -> the estimator's calibration members (:110-120);
//   * the eleven dynamic-reconfigure values -> the estimator's tuning members (:222-238).
#ifndef MPE_COMPAT_ROS_MESSAGE_CONVERSIONS_H_
#define MPE_COMPAT_ROS_MESSAGE_CONVERSIONS_H_

#include "../monocular_pose_estimator_lib/facade_namespace.h"
#include <cmath>
#include <string>
#include <vector>

#include "monocular_pose_estimator_lib/pose_estimator.h"

MPE_FACADE_BEGIN

struct PoseMessageFields {
  double position[3];
  double orientation[4];  //!< x, y, z, w
  double covariance[36];  //!< row-major 6x6
};

//! Eigen::Quaterniond(Matrix3d): trace > 0 -> w from the trace; otherwise the largest diagonal element picks
//! the component that is computed from a square root, the other three follow from the off-diagonal sums.
inline void rotationToQuaternion(const Matrix4d& T, double q_xyzw[4]) {
  const double t = T(0, 0) + T(1, 1) + T(2, 2);
  if (t > 0.0) {
    double s = std::sqrt(t + 1.0);
    q_xyzw[3] = 0.5 * s;
    s = 0.5 / s;
    q_xyzw[0] = (T(2, 1) - T(1, 2)) * s;
    q_xyzw[1] = (T(0, 2) - T(2, 0)) * s;
    q_xyzw[2] = (T(1, 0) - T(0, 1)) * s;
    return;
  }
  int i = 0;
  if (T(1, 1) > T(0, 0)) i = 1;
  if (T(2, 2) > T(i, i)) i = 2;
  const int j = (i + 1) % 3, k = (j + 1) % 3;
  double s = std::sqrt(T(i, i) - T(j, j) - T(k, k) + 1.0);
  q_xyzw[i] = 0.5 * s;
  s = 0.5 / s;
  q_xyzw[3] = (T(k, j) - T(j, k)) * s;
  q_xyzw[j] = (T(j, i) + T(i, j)) * s;
  q_xyzw[k] = (T(k, i) + T(i, k)) * s;
}

inline PoseMessageFields poseToMessageFields(const Matrix4d& transform, const Matrix6d& cov) {
  PoseMessageFields m;
  for (int i = 0; i < 3; ++i) m.position[i] = transform(i, 3);
  rotationToQuaternion(transform, m.orientation);
  for (unsigned i = 0; i < 6; ++i)
    for (unsigned j = 0; j < 6; ++j) m.covariance[j + 6 * i] = cov(i, j);
  return m;
}

//! CameraInfo.K (row-major 9) and CameraInfo.D -> camera_matrix_K_ / camera_distortion_coeffs_
inline void applyCameraInfo(PoseEstimator& pe, const double K9[9], const std::vector<double>& D) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) pe.camera_matrix_K_(r, c) = K9[3 * r + c];
  pe.camera_distortion_coeffs_ = D;
}

//! The dynamic-reconfigure contract (cfg/MonocularPoseEstimator.cfg: names, defaults, ranges)
// sensor_msgs/Image.encoding -> MPE_ENC_* for the encodings the back-end decodes itself (the node's
// cv_bridge::toCvCopy(image_msg, MONO8), monocular_pose_estimator.cpp:147); -1: leave it to cv_bridge (Bayer etc.)
inline int mpeEncodingFromString(const std::string& encoding) {
  if (encoding == "mono8" || encoding == "8UC1") return MPE_ENC_MONO8;
  if (encoding == "bgr8") return MPE_ENC_BGR8;
  if (encoding == "rgb8") return MPE_ENC_RGB8;
  if (encoding == "bgra8") return MPE_ENC_BGRA8;
  if (encoding == "rgba8") return MPE_ENC_RGBA8;
  if (encoding == "mono16") return MPE_ENC_MONO16;   // (16UC1 carries no [0, 65535] convention: cv_bridge refuses it too)
  return -1;
}

//! How onImage obtains the MONO8 pixels of a frame (monocular_pose_estimator.cpp:147).  mono8 is used in place and
//! mono16 is decoded by the back-end in every build; the colour encodings are decoded by the back-end only when
//! the build states that the linked OpenCV uses the 14-bit gray weights mpe_convert_to_mono8 restates
//! (`gray_14bit`), otherwise they go through cv_bridge like everything the back-end does not know (Bayer ...).
//! `enc` receives the MPE_ENC_* code for FRAME_BACKEND_DECODE and -1 otherwise: MPE_ENC_MONO8 is 0, so "no code"
//! must never be spelled 0 (`enc = 0` sent interleaved colour bytes down the mono8 branch).
enum FramePath { FRAME_IN_PLACE = 0, FRAME_BACKEND_DECODE = 1, FRAME_CV_BRIDGE = 2 };
inline FramePath framePathForEncoding(const std::string& encoding, bool gray_14bit, int* enc) {
  const int e = mpeEncodingFromString(encoding);
  *enc = -1;
  if (e == MPE_ENC_MONO8) return FRAME_IN_PLACE;
  if (e == MPE_ENC_MONO16 || (gray_14bit && e > 0)) {
    *enc = e;
    return FRAME_BACKEND_DECODE;
  }
  return FRAME_CV_BRIDGE;
}

struct ReconfigureValues {
  int threshold_value;                       // 180  [0, 255]
  double gaussian_sigma;                     // 0.6  [0, 6]
  double min_blob_area;                      // 10   [0, 100]
  double max_blob_area;                      // 200  [0, 1000]
  double max_width_height_distortion;        // 0.5  [0, 1]
  double max_circular_distortion;            // 0.5  [0, 1]
  double back_projection_pixel_tolerance;    // 5    [0, 10]
  double nearest_neighbour_pixel_tolerance;  // 5    [0, 10]
  double certainty_threshold;                // 0.75 [0, 1]
  double valid_correspondence_threshold;     // 0.7  [0, 1]
  int roi_border_thickness;                  // 10   [0, 200]
  ReconfigureValues()
      : threshold_value(180), gaussian_sigma(0.6), min_blob_area(10), max_blob_area(200),
        max_width_height_distortion(0.5), max_circular_distortion(0.5), back_projection_pixel_tolerance(5),
        nearest_neighbour_pixel_tolerance(5), certainty_threshold(0.75), valid_correspondence_threshold(0.7),
        roi_border_thickness(10) {}
};

inline void applyReconfigure(PoseEstimator& pe, const ReconfigureValues& v) {
  pe.detection_threshold_value_ = v.threshold_value;
  pe.gaussian_sigma_ = v.gaussian_sigma;
  pe.min_blob_area_ = v.min_blob_area;
  pe.max_blob_area_ = v.max_blob_area;
  pe.max_width_height_distortion_ = v.max_width_height_distortion;
  pe.max_circular_distortion_ = v.max_circular_distortion;
  pe.roi_border_thickness_ = (unsigned)v.roi_border_thickness;
  pe.setBackProjectionPixelTolerance(v.back_projection_pixel_tolerance);
  pe.setNearestNeighbourPixelTolerance(v.nearest_neighbour_pixel_tolerance);
  pe.setCertaintyThreshold(v.certainty_threshold);
  pe.setValidCorrespondenceThreshold(v.valid_correspondence_threshold);
}

MPE_FACADE_END  // namespace monocular_pose_estimator
#endif
