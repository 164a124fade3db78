// replay_main.cpp — ROS-free counterpart of `roslaunch monocular_pose_estimator demo.launch`
// (BASELINE config C1): replays a frame sequence through the PoseEstimator facade — the whole
// estimateBodyPose state machine incl. tracking — and prints one line per frame, like the
// estimated_pose topic of MPENode (monocular_pose_estimator/src/monocular_pose_estimator.cpp:159-190).
//
//   replay --markers <marker_positions.yaml> --frames <file.raw> --rows R --cols C [--n N] [--dt seconds]
//          [--threshold 140] [--sigma 0.6] [--bp-tol 5] [--nn-tol 7] [--roi-border 20] [--bruteforce]
//          [--K fx fy cx cy] [--D k1 k2 p1 p2 k3]
//
// <file.raw>: N frames of rows*cols bytes back to back (mono8).  The YAML reader understands the
// reference's marker file format (marker_positions: - x: .. y: .. z: ..,
// monocular_pose_estimator/marker_positions/demo_marker_positions.yaml).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "marker_yaml.h"
#include "monocular_pose_estimator_lib/pose_estimator.h"

using namespace monocular_pose_estimator;

int main(int argc, char** argv) {
  const char *markers = 0, *frames = 0;
  int rows = 480, cols = 752, n = -1;
  double dt = 0.05;
  bool bruteforce = false;
  PoseEstimator pe;
  // demo.launch:12-22
  pe.detection_threshold_value_ = 140;
  pe.gaussian_sigma_ = 0.6;
  pe.min_blob_area_ = 10;
  pe.max_blob_area_ = 200;
  pe.max_width_height_distortion_ = 0.5;
  pe.max_circular_distortion_ = 0.5;
  pe.roi_border_thickness_ = 20;
  pe.setBackProjectionPixelTolerance(5);
  pe.setNearestNeighbourPixelTolerance(7);
  pe.setCertaintyThreshold(0.75);
  pe.setValidCorrespondenceThreshold(0.7);
  // README.md:165-166 camera
  double K[4] = {615.652408400557, 616.760184718123, 362.655454167686, 256.67210750994};
  std::vector<double> D = {-0.358561237166698, 0.149312912580924, 0.000484551782515636, -0.000200189442379448, 0.0};
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&](int k) { return (i + k < argc) ? argv[i + k] : "0"; };
    if (a == "--markers") markers = argv[++i];
    else if (a == "--frames") frames = argv[++i];
    else if (a == "--rows") rows = std::atoi(argv[++i]);
    else if (a == "--cols") cols = std::atoi(argv[++i]);
    else if (a == "--n") n = std::atoi(argv[++i]);
    else if (a == "--dt") dt = std::atof(argv[++i]);
    else if (a == "--threshold") pe.detection_threshold_value_ = std::atoi(argv[++i]);
    else if (a == "--sigma") pe.gaussian_sigma_ = std::atof(argv[++i]);
    else if (a == "--bp-tol") pe.setBackProjectionPixelTolerance(std::atof(argv[++i]));
    else if (a == "--nn-tol") pe.setNearestNeighbourPixelTolerance(std::atof(argv[++i]));
    else if (a == "--roi-border") pe.roi_border_thickness_ = (unsigned)std::atoi(argv[++i]);
    else if (a == "--bruteforce") bruteforce = true;
    else if (a == "--K") { for (int k = 0; k < 4; ++k) K[k] = std::atof(next(k + 1)); i += 4; }
    else if (a == "--D") { for (int k = 0; k < 5; ++k) D[k] = std::atof(next(k + 1)); i += 5; }
    else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
  }
  if (!markers || !frames) {
    std::fprintf(stderr, "usage: replay --markers file.yaml --frames file.raw --rows R --cols C [options]\n");
    return 2;
  }
  List4DPoints m;
  if (!read_markers(markers, m)) {
    std::fprintf(stderr, "no marker_positions in %s\n", markers);  // ROS.cpp:64-70 shuts down in this case
    return 2;
  }
  pe.setMarkerPositions(m);
  pe.camera_matrix_K_(0, 0) = K[0];
  pe.camera_matrix_K_(1, 1) = K[1];
  pe.camera_matrix_K_(0, 2) = K[2];
  pe.camera_matrix_K_(1, 2) = K[3];
  pe.camera_matrix_K_(2, 2) = 1.0;
  pe.camera_distortion_coeffs_ = D;
  pe.setBruteForceEveryFrame(bruteforce);

  FILE* f = std::fopen(frames, "rb");
  if (!f) {
    std::fprintf(stderr, "cannot open %s\n", frames);
    return 2;
  }
  std::vector<uint8_t> img((size_t)rows * cols);
  int k = 0, found = 0;
  while ((n < 0 || k < n) && std::fread(img.data(), 1, img.size(), f) == img.size()) {
    const double t = k * dt;
    const bool ok = pe.estimateBodyPose(ImageView(img.data(), rows, cols, cols), t);
    if (ok) {
      ++found;
      const Matrix4d T = pe.getPredictedPose();
      std::printf("%d %.6f pose", k, t);
      for (int i = 0; i < 16; ++i) std::printf(" %.17g", T(i));
      std::printf("\n");
    } else {
      std::printf("%d %.6f none\n", k, t);  // "Unable to resolve a pose." (ROS.cpp:194)
    }
    ++k;
  }
  std::fclose(f);
  std::fprintf(stderr, "%d frames, %d poses\n", k, found);
  return 0;
}
