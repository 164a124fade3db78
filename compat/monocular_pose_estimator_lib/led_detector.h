// led_detector.h — LEDDetector (reference: lib/include/monocular_pose_estimator_lib/led_detector.h:84-138,
// src/led_detector.cpp).  Same static interface on the facade's vocabulary types (ImageView for cv::Mat,
// Rect / Size / Point2f for their cv:: namesakes); findLeds runs on the MI355X (mpe_find_leds),
// determineROI / distortPoints are a dozen host flops inside the same library.
#ifndef MPE_COMPAT_LED_DETECTOR_H_
#define MPE_COMPAT_LED_DETECTOR_H_

#include "facade_namespace.h"
#include <vector>

#include "datatypes.h"

MPE_FACADE_BEGIN

//! the last back-end failure of a static primitive (P3P::*, LEDDetector::*), "" if none; the reference's primitives
//! have no error channel, these keep its return values and say why here
const char* mpe_facade_last_error();

class LEDDetector {
 public:
  //! led_detector.cpp:35-112.  pixel_positions is only rewritten when at least one LED was found (as in
  //! the reference); distorted_detection_centers always.  Never throws (the reference does not): on a HIP / usage
  //! error, or when more than MPE_MAX_DETECTIONS (64) blobs pass the filter, NO detections are reported, the reason is
  //! written to stderr once and kept for mpe_facade_last_error().
  static void findLeds(const ImageView& image, Rect ROI, const int& threshold_value, const double& gaussian_sigma,
                       const double& min_blob_area, const double& max_blob_area,
                       const double& max_width_height_distortion, const double& max_circular_distortion,
                       List2DPoints& pixel_positions, std::vector<Point2f>& distorted_detection_centers,
                       const Matrix3d& camera_matrix_K, const std::vector<double>& camera_distortion_coeffs);
  //! led_detector.cpp:114-179
  static Rect determineROI(List2DPoints pixel_positions, Size image_size, const int border_size,
                           const Matrix3d& camera_matrix_K, const std::vector<double>& camera_distortion_coeffs);
  //! led_detector.cpp:181-224
  static void distortPoints(const std::vector<Point2f>& src, std::vector<Point2f>& dst, const Matrix3d& camera_matrix_K,
                            const std::vector<double>& distortion_matrix);
};

MPE_FACADE_END  // namespace monocular_pose_estimator
#endif
