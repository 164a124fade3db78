// datatypes.h — interface vocabulary of the facade.
//
// Mirrors the names of the reference's lib/include/monocular_pose_estimator_lib/datatypes.h:38-52
// (List2DPoints, List4DPoints, VectorXuPairs, Matrix6d ...).  The reference builds them on Eigen,
// which is not available in this toolchain, so they are plain fixed-size aggregates here (row-major
// storage, operator()(r,c) access like Eigen).  When Eigen IS available, compat/adapters/eigen_adapters.h converts
// (and compat/adapters/reference_surface.h offers the reference's Eigen / cv::Mat class surface itself).
#ifndef MPE_COMPAT_DATATYPES_H_
#define MPE_COMPAT_DATATYPES_H_

#include "facade_namespace.h"
#include <array>
#include <cstddef>
#include <cstdint>
#include <vector>

MPE_FACADE_BEGIN

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

typedef Matrix<3, 4> Matrix3x4d;
typedef Matrix<5, 1> Vector5d;
typedef std::vector<unsigned> RowXu;  //!< dynamic row of unsigned

//! Dynamic matrix of unsigned (the reference's MatrixXYu), row-major, (r,c) access like Eigen
class MatrixXYu {
 public:
  MatrixXYu() : rows_(0), cols_(0) {}
  MatrixXYu(size_t r, size_t c) : rows_(r), cols_(c), v_(r * c, 0u) {}
  unsigned& operator()(size_t r, size_t c) { return v_[r * cols_ + c]; }
  unsigned operator()(size_t r, size_t c) const { return v_[r * cols_ + c]; }
  size_t rows() const { return rows_; }
  size_t cols() const { return cols_; }
  const unsigned* data() const { return v_.data(); }

 private:
  size_t rows_, cols_;
  std::vector<unsigned> v_;
};

//! cv::Rect / cv::Size stand-ins
struct Rect {
  int x, y, width, height;
  Rect() : x(0), y(0), width(0), height(0) {}
  Rect(int x_, int y_, int w_, int h_) : x(x_), y(y_), width(w_), height(h_) {}
};
struct Size {
  int width, height;
  Size() : width(0), height(0) {}
  Size(int w_, int h_) : width(w_), height(h_) {}
};

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

MPE_FACADE_END  // namespace monocular_pose_estimator
#endif
