// reference_surface.h — monocular_pose_estimator::PoseEstimator with LITERALLY the public surface of the reference
// class (lib/include/monocular_pose_estimator_lib/pose_estimator.h:82-91, 339-801): cv::Mat camera_matrix_K_,
// bool estimateBodyPose(cv::Mat, double), Eigen::Matrix4d getPredictedPose(), Matrix6d getPoseCovariance(),
// void setMarkerPositions(List4DPoints) with the Eigen-based datatypes of datatypes.h:38-52 — so that the
// reference's only caller, MPENode (monocular_pose_estimator/src/monocular_pose_estimator.cpp), compiles UNCHANGED
// against this back-end: build it with -DMPE_REFERENCE_SURFACE -I<repo>/compat -I<repo>/include and link
// libmonocular_pose_estimator_compat.so + libmpe_hip.so.  Included at the end of
// monocular_pose_estimator_lib/pose_estimator.h when MPE_REFERENCE_SURFACE is defined; needs Eigen and OpenCV.
// Every method converts its arguments and delegates to hip::PoseEstimator (the plain-array facade).
#ifndef MPE_COMPAT_REFERENCE_SURFACE_H_
#define MPE_COMPAT_REFERENCE_SURFACE_H_

#include "eigen_adapters.h"
#include "opencv_adapters.h"

#if !defined(MPE_COMPAT_HAVE_EIGEN) || !defined(MPE_COMPAT_HAVE_OPENCV)
#error "MPE_REFERENCE_SURFACE needs <Eigen/Dense> and <opencv2/core.hpp>"
#endif

namespace monocular_pose_estimator {

// the reference's vocabulary (datatypes.h:38-52)
typedef Eigen::Matrix<double, 6, 6> Matrix6d;
typedef Eigen::Matrix<double, 2, 6> Matrix2x6d;
typedef Eigen::Matrix<double, 3, 4> Matrix3x4d;
typedef Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic> MatrixXYd;
typedef Eigen::Matrix<unsigned, Eigen::Dynamic, Eigen::Dynamic> MatrixXYu;
typedef Eigen::Matrix<double, 6, 1> Vector6d;
typedef Eigen::Matrix<unsigned, 3, 1> Vector3u;
typedef Eigen::Matrix<unsigned, 4, 1> Vector4u;
typedef Eigen::Matrix<unsigned, Eigen::Dynamic, 1> VectorXu;
typedef Eigen::Matrix<unsigned, Eigen::Dynamic, 2> VectorXuPairs;
typedef Eigen::Matrix<double, 1, Eigen::Dynamic> RowXd;
typedef Eigen::Matrix<unsigned, 1, Eigen::Dynamic> RowXu;
typedef Eigen::Matrix<Eigen::Vector2d, Eigen::Dynamic, 1> List2DPoints;
typedef Eigen::Matrix<Eigen::Vector3d, Eigen::Dynamic, 1> List3DPoints;
typedef Eigen::Matrix<Eigen::Vector4d, Eigen::Dynamic, 1> List4DPoints;

class PoseEstimator {
 public:
  // public data members, as the reference (pose_estimator.h:82-91); read by every call below
  cv::Mat camera_matrix_K_;
  std::vector<double> camera_distortion_coeffs_;
  int detection_threshold_value_;
  double gaussian_sigma_;
  double min_blob_area_;
  double max_blob_area_;
  double max_width_height_distortion_;
  double max_circular_distortion_;
  unsigned roi_border_thickness_;

  PoseEstimator()
      : detection_threshold_value_(0), gaussian_sigma_(0), min_blob_area_(0), max_blob_area_(0),
        max_width_height_distortion_(0), max_circular_distortion_(0), roi_border_thickness_(0) {}

  void augmentImage(cv::Mat& image) {
    sync();
    hip::ColorImageView v = adapters::colorViewOf(image);
    impl_.augmentImage(v);
  }
  void setMarkerPositions(List4DPoints positions_of_markers_on_object) {
    impl_.setMarkerPositions(adapters::fromEigen(positions_of_markers_on_object));
  }
  List4DPoints getMarkerPositions() { return adapters::toEigen(impl_.getMarkerPositions()); }
  bool estimateBodyPose(cv::Mat image, double time_to_predict) {
    sync();
    return impl_.estimateBodyPose(adapters::viewOf(image), time_to_predict);
  }
  void setPredictedTime(double time) { impl_.setPredictedTime(time); }
  double getPredictedTime() { return impl_.getPredictedTime(); }
  void setPredictedPose(const Eigen::Matrix4d& pose, double time) { impl_.setPredictedPose(adapters::fromEigen<4, 4>(pose), time); }
  Eigen::Matrix4d getPredictedPose() { return adapters::toEigen(impl_.getPredictedPose()); }
  Matrix6d getPoseCovariance() { return adapters::toEigen(impl_.getPoseCovariance()); }
  void setImagePoints(List2DPoints points) { impl_.setImagePoints(adapters::fromEigen(points)); }
  List2DPoints getImagePoints() { return adapters::toEigen(impl_.getImagePoints()); }
  void setPredictedPixels(List2DPoints points) { impl_.setPredictedPixels(adapters::fromEigen(points)); }
  List2DPoints getPredictedPixelPositions() { return adapters::toEigen(impl_.getPredictedPixelPositions()); }
  void setCorrespondences(VectorXuPairs corrs) { impl_.setCorrespondences(adapters::fromEigen(corrs)); }
  VectorXuPairs getCorrespondences() { return adapters::toEigen(impl_.getCorrespondences()); }
  void setBackProjectionPixelTolerance(double tolerance) { impl_.setBackProjectionPixelTolerance(tolerance); }
  double getBackProjectionPixelTolerance() { return impl_.getBackProjectionPixelTolerance(); }
  void setNearestNeighbourPixelTolerance(double tolerance) { impl_.setNearestNeighbourPixelTolerance(tolerance); }
  double getNearestNeighbourPixelTolerance() { return impl_.getNearestNeighbourPixelTolerance(); }
  void setCertaintyThreshold(double threshold) { impl_.setCertaintyThreshold(threshold); }
  double getCertaintyThreshold() { return impl_.getCertaintyThreshold(); }
  void setValidCorrespondenceThreshold(double threshold) { impl_.setValidCorrespondenceThreshold(threshold); }
  double getValidCorrespondenceThreshold() { return impl_.getValidCorrespondenceThreshold(); }
  void setHistogramThreshold(unsigned threshold) { impl_.setHistogramThreshold(threshold); }
  unsigned getHistogramThreshold() { return impl_.getHistogramThreshold(); }
  // the step methods (public in the reference, uncalled from outside the class)
  void predictPose(double time_to_predict) { impl_.predictPose(time_to_predict); }
  void predictMarkerPositionsInImage() {
    sync();
    impl_.predictMarkerPositionsInImage();
  }
  void findCorrespondences() { impl_.findCorrespondences(); }
  unsigned checkCorrespondences() {
    sync();
    return impl_.checkCorrespondences();
  }
  unsigned initialise() {
    sync();
    return impl_.initialise();
  }
  void optimisePose() {
    sync();
    impl_.optimisePose();
  }
  void updatePose() { impl_.updatePose(); }
  void optimiseAndUpdatePose(double& time_to_predict) {
    sync();
    impl_.optimiseAndUpdatePose(time_to_predict);
  }
  void predictWithROI(double& time_to_predict, const cv::Mat& image) {
    sync();
    impl_.predictWithROI(time_to_predict, adapters::viewOf(image));
  }
  void findCorrespondencesAndPredictPose(double& time_to_predict) {
    sync();
    impl_.findCorrespondencesAndPredictPose(time_to_predict);
  }

  hip::PoseEstimator& backend() { return impl_; }  //!< the plain-array facade underneath (batched extension etc.)

 private:
  void sync() {  // the public members are plain data in the reference: copy them down before every call
    if (!camera_matrix_K_.empty()) impl_.camera_matrix_K_ = adapters::cameraMatrixFrom(camera_matrix_K_);
    impl_.camera_distortion_coeffs_ = camera_distortion_coeffs_;
    impl_.detection_threshold_value_ = detection_threshold_value_;
    impl_.gaussian_sigma_ = gaussian_sigma_;
    impl_.min_blob_area_ = min_blob_area_;
    impl_.max_blob_area_ = max_blob_area_;
    impl_.max_width_height_distortion_ = max_width_height_distortion_;
    impl_.max_circular_distortion_ = max_circular_distortion_;
    impl_.roi_border_thickness_ = roi_border_thickness_;
  }
  hip::PoseEstimator impl_;
};

}  // namespace monocular_pose_estimator
#endif
