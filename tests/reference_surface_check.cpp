// Drives monocular_pose_estimator::PoseEstimator through the reference's LITERAL class surface
// (-DMPE_REFERENCE_SURFACE: cv::Mat / Eigen types, compat/adapters/reference_surface.h) the way the reference's
// MPENode does (monocular_pose_estimator.cpp:63-84 markers, :110-120 camera, :159-190 estimate + pose + covariance,
// :204 overlay, :222-233 parameters), and checks it against the plain-array facade underneath on the same frames.
//   usage: reference_surface_check frames.raw n rows cols        (exit 0 + "surface ok" on success)
// Compiled against tests/mock_deps (container-only stand-ins of the Eigen / OpenCV spellings the adapters use: this
// image has neither library); with the real libraries the same file compiles unchanged.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "monocular_pose_estimator_lib/pose_estimator.h"

using namespace monocular_pose_estimator;

int main(int argc, char** argv) {
  if (argc != 5) return 2;
  const int n = atoi(argv[2]), rows = atoi(argv[3]), cols = atoi(argv[4]);
  std::vector<unsigned char> raw((size_t)n * rows * cols);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(raw.data(), 1, raw.size(), f) != raw.size()) return 3;
  fclose(f);
  const double M[5][3] = {{0.0714197, 0.0800214, 0.0622611}, {0.0400755, -0.0912328, 0.0317064},
                          {-0.0647293, -0.0879977, 0.0830852}, {-0.0558663, -0.0165446, 0.053473}, {0.012, 0.031, 0.121}};
  PoseEstimator trackable_object_;  // the literal surface
  hip::PoseEstimator plain;         // the facade it wraps, driven directly
  // MPENode constructor: marker positions as List4DPoints of homogeneous Eigen vectors
  List4DPoints positions_of_markers_on_object(5);
  hip::List4DPoints plain_markers(5);
  for (int i = 0; i < 5; ++i) {
    positions_of_markers_on_object(i) = Eigen::Vector4d(M[i][0], M[i][1], M[i][2], 1.0);
    for (int k = 0; k < 3; ++k) plain_markers[i](k) = M[i][k];
    plain_markers[i](3) = 1.0;
  }
  trackable_object_.setMarkerPositions(positions_of_markers_on_object);
  plain.setMarkerPositions(plain_markers);
  // cameraInfoCallback: K as 3x3 CV_64F cv::Mat written with at<double>
  const double Kv[9] = {615.652408400557, 0, 362.655454167686, 0, 616.760184718123, 256.67210750994, 0, 0, 1};
  const double Dv[5] = {-0.358561237166698, 0.149312912580924, 0.000484551782515636, -0.000200189442379448, 0};
  trackable_object_.camera_matrix_K_ = cv::Mat(3, 3, CV_64F);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      trackable_object_.camera_matrix_K_.at<double>(r, c) = Kv[3 * r + c];
      plain.camera_matrix_K_(r, c) = Kv[3 * r + c];
    }
  trackable_object_.camera_distortion_coeffs_.assign(Dv, Dv + 5);
  plain.camera_distortion_coeffs_.assign(Dv, Dv + 5);
  // dynamicParametersCallback
  trackable_object_.detection_threshold_value_ = plain.detection_threshold_value_ = 140;
  trackable_object_.gaussian_sigma_ = plain.gaussian_sigma_ = 0.6;
  trackable_object_.min_blob_area_ = plain.min_blob_area_ = 10;
  trackable_object_.max_blob_area_ = plain.max_blob_area_ = 200;
  trackable_object_.max_width_height_distortion_ = plain.max_width_height_distortion_ = 0.5;
  trackable_object_.max_circular_distortion_ = plain.max_circular_distortion_ = 0.5;
  trackable_object_.roi_border_thickness_ = plain.roi_border_thickness_ = 20;
  trackable_object_.setBackProjectionPixelTolerance(5);
  plain.setBackProjectionPixelTolerance(5);
  trackable_object_.setNearestNeighbourPixelTolerance(7);
  plain.setNearestNeighbourPixelTolerance(7);
  trackable_object_.setCertaintyThreshold(0.75);
  plain.setCertaintyThreshold(0.75);
  trackable_object_.setValidCorrespondenceThreshold(0.7);
  plain.setValidCorrespondenceThreshold(0.7);
  int poses = 0;
  for (int k = 0; k < n; ++k) {
    cv::Mat image(rows, cols, CV_8UC1);  // imageCallback: cv_bridge::toCvCopy(..., MONO8)->image
    std::memcpy(image.data, raw.data() + (size_t)k * rows * cols, (size_t)rows * cols);
    const double time_to_predict = 0.02 * k;
    const bool found_body_pose = trackable_object_.estimateBodyPose(image, time_to_predict);
    const bool plain_found = plain.estimateBodyPose(hip::ImageView(image.data, rows, cols, (size_t)cols), time_to_predict);
    if (found_body_pose != plain_found) return 10;
    if (found_body_pose) {
      ++poses;
      Eigen::Matrix4d transform = trackable_object_.getPredictedPose();
      Matrix6d cov = trackable_object_.getPoseCovariance();
      const hip::Matrix4d pt = plain.getPredictedPose();
      const hip::Matrix6d pc = plain.getPoseCovariance();
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
          if (transform(i, j) != pt(i, j)) return 11;
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j)
          if (cov(i, j) != pc(i, j)) return 12;
      if (std::fabs(transform(3, 3) - 1.0) > 0) return 13;
    }
    List2DPoints pts = trackable_object_.getImagePoints();
    if ((size_t)pts.size() != plain.getImagePoints().size()) return 14;
    VectorXuPairs corr = trackable_object_.getCorrespondences();
    if ((size_t)corr.rows() != plain.getCorrespondences().size()) return 15;
  }
  cv::Mat visualized_image(rows, cols, CV_8UC3);  // overlay: augmentImage(cv::Mat&)
  trackable_object_.augmentImage(visualized_image);
  long painted = 0;
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < 3 * cols; ++x) painted += visualized_image.data[(size_t)y * visualized_image.step + x] != 0;
  if (poses < n / 2 || painted == 0) return 16;
  printf("surface ok: %d frames, %d poses through estimateBodyPose(cv::Mat, double), overlay painted %ld bytes\n", n, poses,
         painted);
  return 0;
}
