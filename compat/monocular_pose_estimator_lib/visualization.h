// visualization.h — Visualization (reference: lib/include/monocular_pose_estimator_lib/visualization.h:61-68,
// src/visualization.cpp:37-99): the debug overlay — body axes of the estimated pose (x red, y green, z blue,
// 7.5 cm long, projected WITH lens distortion), a ring of radius 10 around every detected LED, the region of
// interest as a blue rectangle; everything 2 px thick.  Host code with its own small rasteriser (Bresenham
// segments, midpoint circles, square 2x2 brush): same primitives, colours and positions as the reference, but
// NOT pixel-identical to OpenCV's line / circle drawing (which this toolchain does not have).
#ifndef MPE_COMPAT_VISUALIZATION_H_
#define MPE_COMPAT_VISUALIZATION_H_

#include "facade_namespace.h"
#include <vector>

#include "datatypes.h"

MPE_FACADE_BEGIN

//! Writable interleaved 3-channel 8-bit image in B,G,R memory order (the node publishes "bgr8").
struct ColorImageView {
  uint8_t* data;
  int rows, cols;
  size_t step;  //!< bytes per row (>= 3 * cols)
  ColorImageView() : data(0), rows(0), cols(0), step(0) {}
  ColorImageView(uint8_t* d, int r, int c, size_t s) : data(d), rows(r), cols(c), step(s) {}
};

struct Point3f {
  float x, y, z;
};

class Visualization {
 public:
  //! points_to_project: origin and the three axis tips in the CAMERA frame (visualization.cpp:37-57)
  static void projectOrientationVectorsOnImage(ColorImageView& image, const std::vector<Point3f>& points_to_project,
                                               const Matrix3d& camera_matrix_K,
                                               const std::vector<double>& camera_distortion_coeffs);
  //! visualization.cpp:59-99
  static void createVisualizationImage(ColorImageView& image, const Matrix4d& transform, const Matrix3d& camera_matrix_K,
                                       const std::vector<double>& camera_distortion_coeffs, Rect region_of_interest,
                                       const std::vector<Point2f>& distorted_detection_centers);
  //! cv::cvtColor(GRAY2RGB) of the node's image callback: replicate a mono8 frame into the three channels
  static void grayToColor(const ImageView& gray, ColorImageView& color);
};

MPE_FACADE_END  // namespace monocular_pose_estimator
#endif
