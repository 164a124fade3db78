// pose_estimator.h — source-compatible facade of monocular_pose_estimator::PoseEstimator
// (reference: lib/include/monocular_pose_estimator_lib/pose_estimator.h:52-803) whose compute
// back-end is the MI355X library libmpe_hip.so (include/mpe.h).
//
// Same class name, namespace, public data members and method names as the reference, so that the
// only caller (MPENode, monocular_pose_estimator/src/monocular_pose_estimator.cpp) keeps compiling
// after the type substitutions listed in INTEGRATION.md (cv::Mat -> ImageView, Eigen types -> datatypes.h) — or
// WITHOUT any edit where Eigen and OpenCV exist: -DMPE_REFERENCE_SURFACE puts a PoseEstimator with the reference's
// literal cv::Mat / Eigen surface on top of this class (compat/adapters/reference_surface.h).
//
// estimateBodyPose runs the reference's whole state machine (pose_estimator.cpp:62-147) through the
// stateful mpe_tracker_* ABI: brute-force initialisation while not initialised, then constant-
// velocity prediction, ROI detection with whole-image retry, nearest-neighbour correspondences and
// fallback to brute force.  setBruteForceEveryFrame(true) resets the state before every call (the
// BASELINE configs' mode; the reference itself has no reset).
#ifndef MPE_COMPAT_POSE_ESTIMATOR_H_
#define MPE_COMPAT_POSE_ESTIMATOR_H_

#include "facade_namespace.h"
#include <string>
#include <vector>

#include "datatypes.h"
#include "led_detector.h"
#include "mpe.h"
#include "visualization.h"

MPE_FACADE_BEGIN

class PoseEstimator {
 public:
  // public members of the reference (pose_estimator.h:82-91)
  Matrix3d camera_matrix_K_;
  std::vector<double> camera_distortion_coeffs_;
  int detection_threshold_value_;
  double gaussian_sigma_;
  double min_blob_area_;
  double max_blob_area_;
  double max_width_height_distortion_;
  double max_circular_distortion_;
  unsigned roi_border_thickness_;

  PoseEstimator();  //!< tolerances 3 / 5 / 0.75 / 0.7 as pose_estimator.cpp:34-42
  ~PoseEstimator();
  // The object owns a device handle and the library-side state machine: not copyable (the reference
  // class is plain state and its only user, MPENode, holds one instance by value and never copies it).
  PoseEstimator(const PoseEstimator&) = delete;
  PoseEstimator& operator=(const PoseEstimator&) = delete;

  void setMarkerPositions(const List4DPoints& positions_of_markers_on_object);  // pose_estimator.cpp:50-55
  List4DPoints getMarkerPositions();
  //! pose_estimator.cpp:62-96; throws std::runtime_error on a HIP / usage error (the reference's
  //! OpenCV calls throw cv::Exception in the same situations), returns pose_updated_ otherwise
  bool estimateBodyPose(const ImageView& image, double time_to_predict);
  void setBruteForceEveryFrame(bool on) { bruteforce_every_frame_ = on; }  //!< extension, see header comment

  void setPredictedTime(double time) {
    predicted_time_ = time;
    pushState();
  }
  double getPredictedTime() { return predicted_time_; }
  void setPredictedPose(const Matrix4d& pose, double time) {
    predicted_pose_ = pose;
    predicted_time_ = time;
    pushState();
  }
  Matrix4d getPredictedPose() { return predicted_pose_; }
  Matrix6d getPoseCovariance() { return pose_covariance_; }
  List2DPoints getImagePoints() { return image_points_; }
  VectorXuPairs getCorrespondences() { return correspondences_; }
  const std::vector<Point2f>& getDistortedDetectionCenters() const { return distorted_detection_centers_; }

  void setBackProjectionPixelTolerance(double t) { params_.back_projection_pixel_tolerance = t; }
  double getBackProjectionPixelTolerance() { return params_.back_projection_pixel_tolerance; }
  void setNearestNeighbourPixelTolerance(double t) { params_.nearest_neighbour_pixel_tolerance = t; }
  double getNearestNeighbourPixelTolerance() { return params_.nearest_neighbour_pixel_tolerance; }
  void setCertaintyThreshold(double t) { params_.certainty_threshold = t; }
  double getCertaintyThreshold() { return params_.certainty_threshold; }
  void setValidCorrespondenceThreshold(double t) { params_.valid_correspondence_threshold = t; }
  double getValidCorrespondenceThreshold() { return params_.valid_correspondence_threshold; }
  void setHistogramThreshold(unsigned t) { params_.histogram_threshold = t; }
  unsigned getHistogramThreshold();

  // ---- the step methods of the reference class (public there, uncalled from outside the class;
  //      pose_estimator.h:425-801).  They operate on this object's state, which estimateBodyPose shares
  //      with the library-side state machine, so the two styles can be mixed.
  void setImagePoints(const List2DPoints& points) { image_points_ = points; }
  void setPredictedPixels(const List2DPoints& points) { predicted_pixel_positions_ = points; }
  List2DPoints getPredictedPixelPositions() { return predicted_pixel_positions_; }
  void setCorrespondences(const VectorXuPairs& corrs) { correspondences_ = corrs; }
  void predictPose(double time_to_predict);        //!< pose_estimator.cpp:232-244
  void predictMarkerPositionsInImage();            //!< :270-276
  void findCorrespondences();                      //!< :372-392
  unsigned checkCorrespondences();                 //!< :394-542, 1 = valid (predicted pose = unrefined pose)
  unsigned initialise();                           //!< :544-721, 1 = pose found (not yet refined)
  void optimisePose();                             //!< :733-792, refines the predicted pose, sets the covariance
  void updatePose();                               //!< :794-800
  void optimiseAndUpdatePose(double& time_to_predict);                       //!< :802-812
  void predictWithROI(double& time_to_predict, const ImageView& image);      //!< :814-829
  void findCorrespondencesAndPredictPose(double& time_to_predict);           //!< :831-848
  Rect getRegionOfInterest() const { return region_of_interest_; }
  //! Visualization::createVisualizationImage on an interleaved 3-channel image (pose_estimator.cpp:44-48)
  void augmentImage(ColorImageView& image);

  //! Batched extension: estimateBodyPose on a fresh estimator for each of n packed frames
  //! (host or device memory); results[i].status == 0 <=> estimateBodyPose returned true.
  void estimateBodyPoseBatch(const uint8_t* frames, int n_frames, int rows, int cols, bool frames_on_device,
                             mpe_result* results);
  //! The node's cv_bridge::toCvCopy(image_msg, MONO8) (monocular_pose_estimator.cpp:147) on the device for the
  //! encodings MPE_ENC_* (bgr8, rgb8, bgra8, rgba8, mono16, mono8): `data` / `step` as in sensor_msgs/Image,
  //! mono8_out = rows x cols bytes (packed).  Throws on an unsupported encoding.
  void decodeToMono8(const void* data, int mpe_encoding, bool is_bigendian, int rows, int cols, size_t step,
                     uint8_t* mono8_out);

 private:
  void syncParams();
  void pushState();  //!< this object's poses / times / counters -> library-side state machine
  void pullState();  //!< and back
  std::vector<double> flatImagePoints() const;
  std::vector<uint32_t> flatCorrespondences() const;
  mpe_handle* handle_;
  mpe_tracker* tracker_;
  bool bruteforce_every_frame_;
  mpe_params params_;
  std::vector<double> markers_xyz_;
  Matrix4d current_pose_, previous_pose_, predicted_pose_;
  Matrix6d pose_covariance_;
  double current_time_, previous_time_, predicted_time_;
  unsigned it_since_initialized_;
  Rect region_of_interest_;
  List2DPoints image_points_;
  List2DPoints predicted_pixel_positions_;
  VectorXuPairs correspondences_;
  std::vector<Point2f> distorted_detection_centers_;
  bool pose_updated_;
};

MPE_FACADE_END  // namespace monocular_pose_estimator

#ifdef MPE_REFERENCE_SURFACE
// the reference's literal class surface (cv::Mat / Eigen types) on top of the facade above
#include "../adapters/reference_surface.h"
#endif
#endif
