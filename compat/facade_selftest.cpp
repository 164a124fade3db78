// facade_selftest.cpp — exercises the parts of the facade that the replay CLI does not reach.
//
//   facade_selftest combos N K
//       prints Combinations::combinationsNoReplacement(N,K), a line "--", then
//       Combinations::permutationsNoReplacement(N,K) (host only, no GPU needed)
//   facade_selftest message < 16+36 doubles per line
//       pose (row-major 4x4) + covariance (row-major 6x6) -> "px py pz qx qy qz qw c0 .. c35" per line, the
//       PoseWithCovarianceStamped fields as compat/ros/message_conversions.h packs them (host only)
//   facade_selftest framepath < encoding names
//       which of the node's three frame paths (in place / back-end decode / cv_bridge) each encoding takes, in the
//       default build and in the MPE_OPENCV_GRAY_14BIT build (host only)
//   facade_selftest steps --markers <yaml> --frames <file.raw> --rows R --cols C [--dt s] [--overlay-out file.bgr]
//       object A: estimateBodyPose per frame.  object B: the same state machine written out with the
//       class's public step methods exactly as pose_estimator.cpp:62-147 strings them together
//       (LEDDetector::findLeds, setImagePoints, initialise, optimiseAndUpdatePose, predictWithROI,
//       findCorrespondencesAndPredictPose).  Both must report the same poses.  Then the static P3P
//       primitives and the overlay are exercised on the last frame.  Prints "selftest ok".
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "marker_yaml.h"
#include "ros/message_conversions.h"
#include "monocular_pose_estimator_lib/combinations.h"
#include "monocular_pose_estimator_lib/p3p.h"
#include "monocular_pose_estimator_lib/pose_estimator.h"

using namespace monocular_pose_estimator;

static void print_matrix(const MatrixXYu& m) {
  for (size_t r = 0; r < m.rows(); ++r) {
    for (size_t c = 0; c < m.cols(); ++c) std::printf(c ? " %u" : "%u", m(r, c));
    std::printf("\n");
  }
}

static void configure(PoseEstimator& pe) {
  pe.camera_matrix_K_(0, 0) = 307.8119;  // README camera
  pe.camera_matrix_K_(0, 2) = 371.6954;
  pe.camera_matrix_K_(1, 1) = 307.5514;
  pe.camera_matrix_K_(1, 2) = 243.5497;
  pe.camera_matrix_K_(2, 2) = 1.0;
  const double D[5] = {-0.2819, 0.0675, 0.0004, -0.0003, -0.0063};
  pe.camera_distortion_coeffs_.assign(D, D + 5);
  pe.detection_threshold_value_ = 140;  // demo.launch
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
}

// estimateBodyPose written with the public step methods (pose_estimator.cpp:62-147)
struct StepDriver {
  PoseEstimator pe;
  unsigned it_since_initialized;
  std::vector<Point2f> centres;
  StepDriver() : it_since_initialized(0) {}
  bool estimate(const ImageView& image, double time_to_predict) {
    bool pose_updated = false;
    List2DPoints detected;
    if (it_since_initialized < 1) {
      pe.setPredictedTime(time_to_predict);
      LEDDetector::findLeds(image, Rect(0, 0, image.cols, image.rows), pe.detection_threshold_value_,
                            pe.gaussian_sigma_, pe.min_blob_area_, pe.max_blob_area_, pe.max_width_height_distortion_,
                            pe.max_circular_distortion_, detected, centres, pe.camera_matrix_K_,
                            pe.camera_distortion_coeffs_);
      if (detected.size() >= 4) {
        pe.setImagePoints(detected);
        if (pe.initialise() == 1) {
          pe.optimiseAndUpdatePose(time_to_predict);
          pose_updated = true;
        }
      }
    } else {
      pe.predictWithROI(time_to_predict, image);
      bool repeat_check = true;
      unsigned num_loops = 0;
      Rect roi = pe.getRegionOfInterest();
      do {
        num_loops++;
        LEDDetector::findLeds(image, roi, pe.detection_threshold_value_, pe.gaussian_sigma_, pe.min_blob_area_,
                              pe.max_blob_area_, pe.max_width_height_distortion_, pe.max_circular_distortion_, detected,
                              centres, pe.camera_matrix_K_, pe.camera_distortion_coeffs_);
        if (detected.size() >= 4) {
          pe.setImagePoints(detected);
          const Matrix4d before = pe.getPredictedPose();
          const double t_before = pe.getPredictedTime();
          (void)t_before;
          pe.findCorrespondencesAndPredictPose(time_to_predict);
          // pose_updated_ is private in the reference too: the driver detects an update by the state change
          pose_updated = std::memcmp(before.data(), pe.getPredictedPose().data(), sizeof(double) * 16) != 0;
          repeat_check = false;
        } else if (num_loops < 2) {
          roi = Rect(0, 0, image.cols, image.rows);
        } else {
          repeat_check = false;
        }
      } while (repeat_check);
    }
    if (pose_updated && it_since_initialized < 2) it_since_initialized++;
    return pose_updated;
  }
};

int main(int argc, char** argv) {
  if (argc >= 4 && !std::strcmp(argv[1], "combos")) {
    const unsigned N = (unsigned)std::atoi(argv[2]), K = (unsigned)std::atoi(argv[3]);
    print_matrix(Combinations::combinationsNoReplacement(N, K));
    std::printf("--\n");
    print_matrix(Combinations::permutationsNoReplacement(N, K));
    std::printf("--\n%u %u %u\n", Combinations::numCombinations(N, K), Combinations::numPermutations(N, K),
                Combinations::factorial((int)N));
    return 0;
  }
  if (argc >= 2 && !std::strcmp(argv[1], "message")) {
    for (;;) {
      Matrix4d T;
      Matrix6d cov;
      for (int i = 0; i < 16; ++i)
        if (std::scanf("%lf", &T(i)) != 1) return 0;
      for (int i = 0; i < 36; ++i)
        if (std::scanf("%lf", &cov(i)) != 1) return 1;
      const PoseMessageFields m = poseToMessageFields(T, cov);
      std::printf("%.17g %.17g %.17g %.17g %.17g %.17g %.17g", m.position[0], m.position[1], m.position[2],
                  m.orientation[0], m.orientation[1], m.orientation[2], m.orientation[3]);
      for (int i = 0; i < 36; ++i) std::printf(" %.17g", m.covariance[i]);
      std::printf("\n");
    }
  }
  if (argc >= 2 && !std::strcmp(argv[1], "nothrow")) {
    // The reference's static primitives have no error channel and never throw (p3p.h:105-108): neither do these — on a
    // box WITHOUT a HIP device (the CPU tier) every one of them comes back with the reference's own failure value and
    // says why through mpe_facade_last_error(); with a device the same calls succeed.
    try {
      Matrix3d fv, wp;
      for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) {
          fv(k, i) = (k == 2) ? 1.0 : 0.1 * (i + 1) * (k + 1);
          wp(k, i) = (k == i) ? 0.2 : 0.01 * (k + 1);
        }
      P3PSolutions sol;
      const int rc = P3P::computePoses(fv, wp, sol);
      Vector5d f;
      for (int i = 0; i < 5; ++i) f(i) = 1.0 + i;
      Vector4d roots;
      const int rq = P3P::solveQuartic(f, roots);
      std::vector<uint8_t> img(64 * 48, 0);
      ImageView view;
      view.data = img.data();
      view.rows = 48;
      view.cols = 64;
      view.step = 64;
      List2DPoints px;
      std::vector<Point2f> centers(3);
      Matrix3d K;
      for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) K(i, k) = (i == k) ? (i < 2 ? 300.0 : 1.0) : 0.0;
      K(0, 2) = 32;
      K(1, 2) = 24;
      LEDDetector::findLeds(view, Rect(0, 0, 64, 48), 140, 0.6, 10, 200, 0.5, 0.5, px, centers, K, std::vector<double>());
      std::printf("nothrow computePoses %d solveQuartic %d centers %d last_error [%s]\n", rc, rq, (int)centers.size(),
                  mpe_facade_last_error());
      return 0;
    } catch (...) {
      std::printf("nothrow THREW\n");
      return 4;
    }
  }
  if (argc >= 2 && !std::strcmp(argv[1], "framepath")) {  // "<encoding> <path, default build> <enc> <path, 14-bit build> <enc>" per line
    char name[64];
    while (std::scanf("%63s", name) == 1) {
      int e0 = 99, e1 = 99;
      const FramePath p0 = framePathForEncoding(name, false, &e0);
      const FramePath p1 = framePathForEncoding(name, true, &e1);
      std::printf("%s %d %d %d %d\n", name, (int)p0, e0, (int)p1, e1);
    }
    return 0;
  }
  if (argc < 2 || std::strcmp(argv[1], "steps")) {
    std::fprintf(stderr, "usage: facade_selftest combos N K | steps --markers y --frames f --rows R --cols C\n");
    return 2;
  }
  const char *markers = 0, *frames = 0, *overlay_out = 0;
  int rows = 480, cols = 752;
  double dt = 0.02;
  for (int i = 2; i + 1 < argc; i += 2) {
    if (!std::strcmp(argv[i], "--markers")) markers = argv[i + 1];
    else if (!std::strcmp(argv[i], "--frames")) frames = argv[i + 1];
    else if (!std::strcmp(argv[i], "--rows")) rows = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--cols")) cols = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--dt")) dt = std::atof(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--overlay-out")) overlay_out = argv[i + 1];
  }
  List4DPoints pts;
  if (!markers || !frames || !read_markers(markers, pts)) {
    std::fprintf(stderr, "cannot read markers / frames\n");
    return 2;
  }
  std::FILE* f = std::fopen(frames, "rb");
  if (!f) return 2;
  std::vector<uint8_t> buf((size_t)rows * cols);
  try {
    PoseEstimator a;
    StepDriver b;
    configure(a);
    configure(b.pe);
    a.setMarkerPositions(pts);
    b.pe.setMarkerPositions(pts);
    int k = 0, n_pose = 0, n_tracked = 0;
    double worst = 0;
    for (; std::fread(buf.data(), 1, buf.size(), f) == buf.size(); ++k) {
      const ImageView img(buf.data(), rows, cols, (size_t)cols);
      const bool ua = a.estimateBodyPose(img, k * dt);
      const bool ub = b.estimate(img, k * dt);
      if (ua != ub) {
        std::printf("frame %d: estimateBodyPose %d, step methods %d\n", k, (int)ua, (int)ub);
        return 1;
      }
      if (!ua) continue;
      ++n_pose;
      if (k > 2) ++n_tracked;
      const Matrix4d Ta = a.getPredictedPose(), Tb = b.pe.getPredictedPose();
      const Matrix6d Ca = a.getPoseCovariance(), Cb = b.pe.getPoseCovariance();
      for (int i = 0; i < 16; ++i) worst = std::fmax(worst, std::fabs(Ta(i) - Tb(i)));
      for (int i = 0; i < 36; ++i) worst = std::fmax(worst, std::fabs(Ca(i) - Cb(i)));
      if (a.getCorrespondences().size() != b.pe.getCorrespondences().size()) {
        std::printf("frame %d: correspondences differ\n", k);
        return 1;
      }
    }
    std::fclose(f);
    std::printf("frames %d poses %d tracked %d worst |A - B| %.3e\n", k, n_pose, n_tracked, worst);
    if (n_pose < k / 2 || worst > 1e-9) return 1;

    // static P3P primitives: a bearing triple generated from the last pose must give it back
    const Matrix4d T = a.getPredictedPose();
    Matrix3d fv, wp;
    for (int i = 0; i < 3; ++i) {
      double pc[3];
      for (int r = 0; r < 3; ++r) pc[r] = T(r, 0) * pts[i](0) + T(r, 1) * pts[i](1) + T(r, 2) * pts[i](2) + T(r, 3);
      const double nrm = std::sqrt(pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2]);
      for (int r = 0; r < 3; ++r) {
        fv(r, i) = pc[r] / nrm;
        wp(r, i) = pts[i](r);
      }
    }
    P3PSolutions sol;
    if (P3P::computePoses(fv, wp, sol) != 0) return 1;
    double best = 1e9;  // [R|C] is the inverse of T: C = -R_T^T t
    for (int s = 0; s < 4; ++s) {
      double e = 0;
      for (int r = 0; r < 3; ++r) {
        const double C = -(T(0, r) * T(0, 3) + T(1, r) * T(1, 3) + T(2, r) * T(2, 3));
        e = std::fmax(e, std::fabs(sol[s](r, 3) - C));
      }
      if (e == e) best = std::fmin(best, e);
    }
    Vector5d fac;
    const double c5[5] = {1, -10, 35, -50, 24};  // (x-1)(x-2)(x-3)(x-4)
    for (int i = 0; i < 5; ++i) fac(i) = c5[i];
    Vector4d roots;
    P3P::solveQuartic(fac, roots);
    double rsum = roots(0) + roots(1) + roots(2) + roots(3);
    std::printf("p3p camera-centre error %.3e, quartic root sum %.6f\n", best, rsum);
    if (best > 1e-6 || std::fabs(rsum - 10.0) > 1e-9) return 1;

    // overlay on the last frame
    std::vector<uint8_t> rgb((size_t)rows * cols * 3);
    ColorImageView color(rgb.data(), rows, cols, (size_t)cols * 3);
    Visualization::grayToColor(ImageView(buf.data(), rows, cols, (size_t)cols), color);
    a.augmentImage(color);
    int red = 0, blue = 0, green = 0;
    for (size_t i = 0; i < rgb.size(); i += 3) {
      red += rgb[i] == 0 && rgb[i + 1] == 0 && rgb[i + 2] == 255;
      green += rgb[i] == 0 && rgb[i + 1] == 255 && rgb[i + 2] == 0;
      blue += rgb[i] == 255 && rgb[i + 1] == 0 && rgb[i + 2] == 0;
    }
    const std::vector<Point2f>& c = a.getDistortedDetectionCenters();
    std::printf("overlay: %d red %d green %d blue pixels, %d detection rings\n", red, green, blue, (int)c.size());
    if (overlay_out) {  // the painted image + what it was painted from, for the independent geometric check
      FILE* fo = std::fopen(overlay_out, "wb");
      if (!fo || std::fwrite(rgb.data(), 1, rgb.size(), fo) != rgb.size()) return 4;
      std::fclose(fo);
      const Matrix4d Tp = a.getPredictedPose();
      const Rect r = a.getRegionOfInterest();
      std::printf("overlay_state");
      for (int i = 0; i < 16; ++i) std::printf(" %.17g", Tp(i));
      std::printf(" %d %d %d %d %d", r.x, r.y, r.width, r.height, (int)c.size());
      for (size_t i = 0; i < c.size(); ++i) std::printf(" %.9g %.9g", (double)c[i].x, (double)c[i].y);
      std::printf("\n");
    }
    if (c.size() < 4 || red < (int)c.size() * 60 || blue < 100 || green < 4) return 1;
    const int cx = (int)std::lrint(c[0].x) + 10, cy = (int)std::lrint(c[0].y);  // a ring pixel right of LED 0
    const uint8_t* px = rgb.data() + ((size_t)cy * cols + cx) * 3;
    if (!(px[0] == 0 && px[2] == 255)) return 1;
    std::printf("selftest ok\n");
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 3;
  }
}
