// p3p.h — P3P (reference: lib/include/monocular_pose_estimator_lib/p3p.h:110-127, src/p3p.cpp).
// Same static interface; the arithmetic runs on the MI355X through mpe_p3p_batch /
// mpe_solve_quartic_batch (the device functions the voting and validation kernels inline).
#ifndef MPE_COMPAT_P3P_H_
#define MPE_COMPAT_P3P_H_

#include "facade_namespace.h"
#include <array>

#include "datatypes.h"

MPE_FACADE_BEGIN

typedef std::array<Matrix3x4d, 4> P3PSolutions;  //!< the reference's Matrix<Matrix<double,3,4>,4,1>

class P3P {
 public:
  //! Columns of feature_vectors / world_points are the three unit bearings / world points.  Returns 0 and
  //! fills the four [R|C] solutions (NaN entries where the reference produces NaN), or -1 if the world
  //! points are collinear (solutions untouched).  Never throws (the reference does not): -1 also when the back-end fails (mpe_facade_last_error()).
  static int computePoses(const Matrix3d& feature_vectors, const Matrix3d& world_points, P3PSolutions& solutions);
  //! Real parts of the four Ferrari roots of factors(0) x^4 + ... + factors(4); always returns 0.
  static int solveQuartic(const Vector5d& factors, Vector4d& real_roots);
};

MPE_FACADE_END  // namespace monocular_pose_estimator
#endif
