// eigen_adapters.h — conversions between the facade's plain aggregates (monocular_pose_estimator_lib/datatypes.h)
// and the Eigen types the reference uses for the same quantities
// (lib/include/monocular_pose_estimator_lib/datatypes.h:38-52).  Header-only; active where <Eigen/Dense> exists.
#ifndef MPE_COMPAT_EIGEN_ADAPTERS_H_
#define MPE_COMPAT_EIGEN_ADAPTERS_H_

#if defined(__has_include)
#if __has_include(<Eigen/Dense>)
#define MPE_COMPAT_HAVE_EIGEN 1
#endif
#endif

#ifdef MPE_COMPAT_HAVE_EIGEN
#include <Eigen/Dense>

#include "../monocular_pose_estimator_lib/datatypes.h"

namespace monocular_pose_estimator {
namespace adapters {

// fixed-size matrices: facade Matrix<R,C> (row-major array) <-> Eigen::Matrix<double,R,C>
template <int R, int C>
inline Eigen::Matrix<double, R, C> toEigen(const hip::Matrix<R, C>& m) {
  Eigen::Matrix<double, R, C> e;
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c) e(r, c) = m(r, c);
  return e;
}
template <int R, int C>
inline hip::Matrix<R, C> fromEigen(const Eigen::Matrix<double, R, C>& e) {
  hip::Matrix<R, C> m;
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c) m(r, c) = e(r, c);
  return m;
}

// List2DPoints / List4DPoints: std::vector<VectorNd> <-> Eigen::Matrix<Eigen::VectorNd, Dynamic, 1>
typedef Eigen::Matrix<Eigen::Vector2d, Eigen::Dynamic, 1> EigenList2DPoints;
typedef Eigen::Matrix<Eigen::Vector4d, Eigen::Dynamic, 1> EigenList4DPoints;
typedef Eigen::Matrix<unsigned, Eigen::Dynamic, 2> EigenVectorXuPairs;

inline EigenList2DPoints toEigen(const hip::List2DPoints& v) {
  EigenList2DPoints e(v.size());
  for (size_t i = 0; i < v.size(); ++i) e(i) = Eigen::Vector2d(v[i](0), v[i](1));
  return e;
}
inline hip::List2DPoints fromEigen(const EigenList2DPoints& e) {
  hip::List2DPoints v((size_t)e.size());
  for (size_t i = 0; i < v.size(); ++i) {
    v[i](0) = e(i)(0);
    v[i](1) = e(i)(1);
  }
  return v;
}
inline EigenList4DPoints toEigen(const hip::List4DPoints& v) {
  EigenList4DPoints e(v.size());
  for (size_t i = 0; i < v.size(); ++i) e(i) = Eigen::Vector4d(v[i](0), v[i](1), v[i](2), v[i](3));
  return e;
}
inline hip::List4DPoints fromEigen(const EigenList4DPoints& e) {
  hip::List4DPoints v((size_t)e.size());
  for (size_t i = 0; i < v.size(); ++i)
    for (int k = 0; k < 4; ++k) v[i](k) = e(i)(k);
  return v;
}
inline EigenVectorXuPairs toEigen(const hip::VectorXuPairs& v) {
  EigenVectorXuPairs e(v.size(), 2);
  for (size_t i = 0; i < v.size(); ++i) {
    e(i, 0) = v[i][0];
    e(i, 1) = v[i][1];
  }
  return e;
}
inline hip::VectorXuPairs fromEigen(const EigenVectorXuPairs& e) {
  hip::VectorXuPairs v((size_t)e.rows());
  for (size_t i = 0; i < v.size(); ++i) {
    v[i][0] = e(i, 0);
    v[i][1] = e(i, 1);
  }
  return v;
}

}  // namespace adapters
}  // namespace monocular_pose_estimator
#endif  // MPE_COMPAT_HAVE_EIGEN
#endif
