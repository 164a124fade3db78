// datatypes.h — interface vocabulary of the facade.
//
// Mirrors the names of the reference's lib/include/monocular_pose_estimator_lib/datatypes.h:38-52
// (List2DPoints, List4DPoints, VectorXuPairs, Matrix6d ...).  The reference builds them on Eigen,
// which is not available in this toolchain, so they are plain fixed-size aggregates here (row-major
// storage, operator()(r,c) access like Eigen).  When Eigen IS available, eigen_adapters.h converts.
#ifndef MPE_COMPAT_DATATYPES_H_
#define MPE_COMPAT_DATATYPES_H_

#include <array>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace monocular_pose_estimator {

template <int R, int C>
struct Matrix {
  double m[R * C];
  Matrix() : m{} {}
  double& operator()(int r, int c) { return m[r * C + c]; }
  const double& operator()(int r, int c) const { return m[r * C + c]; }
  double& operator()(int i) { return m[i]; }
  const double& operator()(int i) const { return m[i]; }
  static Matrix Identity() {
    Matrix I;
    for (int i = 0; i < (R < C ? R : C); ++i) I(i, i) = 1.0;
    return I;
  }
  const double* data() const { return m; }
  double* data() { return m; }
};

typedef Matrix<2, 1> Vector2d;
typedef Matrix<3, 1> Vector3d;
typedef Matrix<4, 1> Vector4d;
typedef Matrix<3, 3> Matrix3d;
typedef Matrix<4, 4> Matrix4d;
typedef Matrix<6, 6> Matrix6d;   //!< 6x6 covariance, twist order (upsilon, omega)
typedef Matrix<6, 1> Vector6d;
typedef std::vector<Vector2d> List2DPoints;  //!< detections, undistorted pixels
typedef std::vector<Vector4d> List4DPoints;  //!< marker positions, homogeneous
typedef std::vector<std::array<unsigned, 2> > VectorXuPairs;  //!< rows (marker, detection), 1-based

//! What the reference passes as cv::Mat (CV_8UC1): a view of a mono8 frame, never written.
struct ImageView {
  const uint8_t* data;
  int rows, cols;
  size_t step;  //!< bytes per row
  ImageView() : data(0), rows(0), cols(0), step(0) {}
  ImageView(const uint8_t* d, int r, int c, size_t s) : data(d), rows(r), cols(c), step(s) {}
};

//! cv::Point2f stand-in for the distorted detection centres (overlay only)
struct Point2f {
  float x, y;
};

}  // namespace monocular_pose_estimator
#endif
