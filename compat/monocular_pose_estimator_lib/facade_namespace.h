// facade_namespace.h — where the facade's names live.
//
// Default: everything is directly visible as monocular_pose_estimator::PoseEstimator, ::List4DPoints ... (the
// `hip` level is an inline namespace).  With -DMPE_REFERENCE_SURFACE (needs Eigen and OpenCV, see
// compat/adapters/reference_surface.h) the same classes stay in monocular_pose_estimator::hip, and the names of the
// reference — Eigen-based datatypes, a PoseEstimator whose public surface is literally the reference's
// (cv::Mat camera_matrix_K_, estimateBodyPose(cv::Mat, double), Eigen::Matrix4d getPredictedPose() ...) — take
// their place, so that the reference's MPENode compiles unchanged against this back-end.  The mangled names of the
// compiled facade library contain `hip` either way: ONE libmonocular_pose_estimator_compat.so serves both.
#ifndef MPE_COMPAT_FACADE_NAMESPACE_H_
#define MPE_COMPAT_FACADE_NAMESPACE_H_
#if defined(MPE_REFERENCE_SURFACE)
#define MPE_FACADE_BEGIN namespace monocular_pose_estimator { namespace hip {
#else
#define MPE_FACADE_BEGIN namespace monocular_pose_estimator { inline namespace hip {
#endif
#define MPE_FACADE_END } }
#endif
