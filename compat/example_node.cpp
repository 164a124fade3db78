// example_node.cpp — what MPENode::imageCallback does with the estimator
// (monocular_pose_estimator/src/monocular_pose_estimator.cpp:84,110-120,159-190,222-233), written
// against the facade, without ROS: renders nothing, reads a raw mono8 frame file.
//
//   example_node <frame.raw> <rows> <cols>      (markers = demo_marker_positions.yaml + 1, K/D of the README)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "monocular_pose_estimator_lib/pose_estimator.h"

using namespace monocular_pose_estimator;

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s frame.raw rows cols\n", argv[0]);
    return 2;
  }
  const int rows = std::atoi(argv[2]), cols = std::atoi(argv[3]);
  std::vector<uint8_t> img((size_t)rows * cols);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(img.data(), 1, img.size(), f) != img.size()) {
    std::fprintf(stderr, "cannot read %s\n", argv[1]);
    return 2;
  }
  std::fclose(f);

  PoseEstimator trackable_object_;
  // MPENode::MPENode: marker_positions -> setMarkerPositions (ROS.cpp:63-84)
  const double M[5][3] = {{0.0714197, 0.0800214, 0.0622611},  {0.0400755, -0.0912328, 0.0317064},
                          {-0.0647293, -0.0879977, 0.0830852}, {-0.0558663, -0.0165446, 0.053473},
                          {0.0120, 0.0310, 0.1210}};
  List4DPoints markers(5);
  for (int i = 0; i < 5; ++i) {
    for (int k = 0; k < 3; ++k) markers[i](k) = M[i][k];
    markers[i](3) = 1;
  }
  trackable_object_.setMarkerPositions(markers);
  // cameraInfoCallback (ROS.cpp:103-126)
  trackable_object_.camera_matrix_K_(0, 0) = 615.652408400557;
  trackable_object_.camera_matrix_K_(0, 2) = 362.655454167686;
  trackable_object_.camera_matrix_K_(1, 1) = 616.760184718123;
  trackable_object_.camera_matrix_K_(1, 2) = 256.67210750994;
  trackable_object_.camera_matrix_K_(2, 2) = 1.0;
  trackable_object_.camera_distortion_coeffs_ = {-0.358561237166698, 0.149312912580924, 0.000484551782515636,
                                                 -0.000200189442379448, 0.0};
  // dynamicParametersCallback with the demo.launch values (ROS.cpp:220-236)
  trackable_object_.detection_threshold_value_ = 140;
  trackable_object_.gaussian_sigma_ = 0.6;
  trackable_object_.min_blob_area_ = 10;
  trackable_object_.max_blob_area_ = 200;
  trackable_object_.max_width_height_distortion_ = 0.5;
  trackable_object_.max_circular_distortion_ = 0.5;
  trackable_object_.roi_border_thickness_ = 20;
  trackable_object_.setBackProjectionPixelTolerance(5);
  trackable_object_.setNearestNeighbourPixelTolerance(7);
  trackable_object_.setCertaintyThreshold(0.75);
  trackable_object_.setValidCorrespondenceThreshold(0.7);

  trackable_object_.setBruteForceEveryFrame(true);
  // imageCallback (ROS.cpp:159-190)
  const bool found_body_pose = trackable_object_.estimateBodyPose(ImageView(img.data(), rows, cols, cols), 0.0);
  if (!found_body_pose) {
    std::printf("Unable to resolve a pose.\n");
    return 1;
  }
  Matrix6d cov = trackable_object_.getPoseCovariance();
  Matrix4d T = trackable_object_.getPredictedPose();
  std::printf("pose");
  for (int i = 0; i < 16; ++i) std::printf(" %.17g", T(i));
  std::printf("\ncov_diag");
  for (int i = 0; i < 6; ++i) std::printf(" %.6g", cov(i, i));
  std::printf("\n");
  return 0;
}
