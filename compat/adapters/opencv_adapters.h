// opencv_adapters.h — cv::Mat <-> the facade's image views and camera matrix.  The reference passes the frame as
// cv::Mat CV_8UC1 by value (pose_estimator.h:366), keeps K as a 3x3 CV_64F cv::Mat read with .at<double>
// (pose_estimator.h:82, pose_estimator.cpp:259) and draws into a CV_8UC3 cv::Mat (pose_estimator.h:341).
// Header-only; active where <opencv2/core.hpp> exists.  No pixel is copied: the views alias the cv::Mat buffers.
#ifndef MPE_COMPAT_OPENCV_ADAPTERS_H_
#define MPE_COMPAT_OPENCV_ADAPTERS_H_

#if defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#define MPE_COMPAT_HAVE_OPENCV 1
#endif
#endif

#ifdef MPE_COMPAT_HAVE_OPENCV
#include <opencv2/core.hpp>

#include <stdexcept>

#include "../monocular_pose_estimator_lib/datatypes.h"
#include "../monocular_pose_estimator_lib/visualization.h"

namespace monocular_pose_estimator {
namespace adapters {

inline hip::ImageView viewOf(const cv::Mat& image) {
  if (image.type() != CV_8UC1) throw std::invalid_argument("estimateBodyPose expects a CV_8UC1 (MONO8) image");
  return hip::ImageView(image.data, image.rows, image.cols, (size_t)image.step);
}
inline hip::ColorImageView colorViewOf(cv::Mat& image) {
  if (image.type() != CV_8UC3) throw std::invalid_argument("augmentImage expects a CV_8UC3 image");
  return hip::ColorImageView(image.data, image.rows, image.cols, (size_t)image.step);
}
inline hip::Matrix3d cameraMatrixFrom(const cv::Mat& K) {
  if (K.rows != 3 || K.cols != 3 || K.type() != CV_64F) throw std::invalid_argument("camera_matrix_K_ must be 3x3 CV_64F");
  hip::Matrix3d m;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) m(r, c) = K.at<double>(r, c);
  return m;
}
inline cv::Mat cameraMatrixTo(const hip::Matrix3d& m) {
  cv::Mat K(3, 3, CV_64F);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) K.at<double>(r, c) = m(r, c);
  return K;
}
inline cv::Rect rectTo(const hip::Rect& r) { return cv::Rect(r.x, r.y, r.width, r.height); }

}  // namespace adapters
}  // namespace monocular_pose_estimator
#endif  // MPE_COMPAT_HAVE_OPENCV
#endif
