// primitives.cpp — the static classes P3P and LEDDetector on top of the C ABI.  The reference's static
// functions carry no state; here they share one lazily created library handle per process.
// Error behaviour: the reference's static primitives never throw (p3p.h:105-108:
// computePoses returns -1 as its only failure; findLeds / determineROI / distortPoints return void or a value), so
// neither do these: a back-end failure — no HIP device, a usage error — comes back as the reference's own failure value
// (computePoses / solveQuartic: -1 with the outputs untouched / NaN; findLeds: no detections; determineROI: the whole
// image; distortPoints: the input points), is written to stderr once per kind, and can be read with
// mpe_facade_last_error().
#include "facade_namespace.h"
#include <cmath>
#include <cstdio>
#include <mutex>
#include <string>

#include "led_detector.h"
#include "mpe.h"
#include "p3p.h"

MPE_FACADE_BEGIN

namespace {
std::mutex g_lock;  // one handle = one caller at a time (include/mpe.h)
std::string g_last_error;
mpe_handle* shared_handle() {  // 0: no HIP device (there is no CPU fallback)
  static mpe_handle* h = 0;
  static bool tried = false;
  if (!h && !tried) {
    tried = true;
    if (mpe_create(&h, -1) != MPE_OK) h = 0;
  }
  return h;
}
// true = the call succeeded; else the message is kept (and printed the first time this `what` fails)
bool ok(mpe_handle* h, int rc, const char* what) {
  if (h && rc == MPE_OK) return true;
  g_last_error = std::string(what) + ": " + (h ? mpe_last_error(h) : "mpe_create failed: no HIP device (there is no CPU fallback)");
  static std::string printed;
  if (printed.find(what) == std::string::npos) {
    printed += what;
    printed += ';';
    std::fprintf(stderr, "monocular_pose_estimator (mpe facade): %s\n", g_last_error.c_str());
  }
  return false;
}
}  // namespace

const char* mpe_facade_last_error() {
  std::lock_guard<std::mutex> guard(g_lock);
  static std::string copy;
  copy = g_last_error;
  return copy.c_str();
}

int P3P::computePoses(const Matrix3d& feature_vectors, const Matrix3d& world_points, P3PSolutions& solutions) {
  std::lock_guard<std::mutex> guard(g_lock);
  mpe_handle* h = shared_handle();
  double fv[9], wp[9], sol[48];
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) {
      fv[3 * i + k] = feature_vectors(k, i);  // column i -> point i
      wp[3 * i + k] = world_points(k, i);
    }
  for (int s = 0; s < 4; ++s)
    for (int i = 0; i < 12; ++i) sol[12 * s + i] = solutions[s](i);
  int status = 0;
  if (!ok(h, h ? mpe_p3p_batch(h, fv, wp, 1, sol, &status) : MPE_ERR_NO_DEVICE, "mpe_p3p_batch")) return -1;
  if (status != 0) return -1;
  for (int s = 0; s < 4; ++s)
    for (int i = 0; i < 12; ++i) solutions[s](i) = sol[12 * s + i];
  return 0;
}

int P3P::solveQuartic(const Vector5d& factors, Vector4d& real_roots) {
  std::lock_guard<std::mutex> guard(g_lock);
  mpe_handle* h = shared_handle();
  if (!ok(h, h ? mpe_solve_quartic_batch(h, factors.data(), 1, 0, real_roots.data()) : MPE_ERR_NO_DEVICE,
          "mpe_solve_quartic_batch")) {
    for (int i = 0; i < 4; ++i) real_roots(i) = std::nan("");
    return -1;  // (the reference always returns 0: it has no way to fail)
  }
  return 0;
}

void LEDDetector::findLeds(const ImageView& image, Rect ROI, const int& threshold_value, const double& gaussian_sigma,
                           const double& min_blob_area, const double& max_blob_area,
                           const double& max_width_height_distortion, const double& max_circular_distortion,
                           List2DPoints& pixel_positions, std::vector<Point2f>& distorted_detection_centers,
                           const Matrix3d& camera_matrix_K, const std::vector<double>& camera_distortion_coeffs) {
  std::lock_guard<std::mutex> guard(g_lock);
  mpe_handle* h = shared_handle();
  mpe_params p;
  mpe_default_params(&p);
  p.threshold_value = threshold_value;
  p.gaussian_sigma = gaussian_sigma;
  p.min_blob_area = min_blob_area;
  p.max_blob_area = max_blob_area;
  p.max_width_height_distortion = max_width_height_distortion;
  p.max_circular_distortion = max_circular_distortion;
  double und[2 * MPE_MAX_DETECTIONS];
  float dist[2 * MPE_MAX_DETECTIONS];
  int n = 0;
  if (!ok(h,
          h ? mpe_find_leds(h, image.data, image.rows, image.cols, image.step, ROI.x, ROI.y, ROI.width, ROI.height, &p,
                            camera_matrix_K.data(), camera_distortion_coeffs.empty() ? 0 : camera_distortion_coeffs.data(),
                            (int)camera_distortion_coeffs.size(), und, dist, MPE_MAX_DETECTIONS, &n)
            : MPE_ERR_NO_DEVICE,
          "mpe_find_leds"))
    n = 0;  // (no detections: what the reference's caller sees of a frame without LEDs)
  distorted_detection_centers.resize(n);
  for (int i = 0; i < n; ++i) {
    distorted_detection_centers[i].x = dist[2 * i];
    distorted_detection_centers[i].y = dist[2 * i + 1];
  }
  if (n > 0) {
    pixel_positions.resize(n);
    for (int i = 0; i < n; ++i) {
      pixel_positions[i](0) = und[2 * i];
      pixel_positions[i](1) = und[2 * i + 1];
    }
  }
}

Rect LEDDetector::determineROI(List2DPoints pixel_positions, Size image_size, const int border_size,
                               const Matrix3d& camera_matrix_K, const std::vector<double>& camera_distortion_coeffs) {
  std::vector<double> px(2 * pixel_positions.size());
  for (size_t i = 0; i < pixel_positions.size(); ++i) {
    px[2 * i] = pixel_positions[i](0);
    px[2 * i + 1] = pixel_positions[i](1);
  }
  int r[4] = {0, 0, image_size.width, image_size.height};
  const int rc = mpe_determine_roi(px.data(), (int)pixel_positions.size(), image_size.height, image_size.width,
                                   border_size, camera_matrix_K.data(),
                                   camera_distortion_coeffs.empty() ? 0 : camera_distortion_coeffs.data(),
                                   (int)camera_distortion_coeffs.size(), r);
  if (rc != MPE_OK) {  // (the whole image: the region the reference searches when it has no prediction)
    std::lock_guard<std::mutex> guard(g_lock);
    ok(0, rc, "mpe_determine_roi");
    return Rect(0, 0, image_size.width, image_size.height);
  }
  return Rect(r[0], r[1], r[2], r[3]);
}

void LEDDetector::distortPoints(const std::vector<Point2f>& src, std::vector<Point2f>& dst,
                                const Matrix3d& camera_matrix_K, const std::vector<double>& distortion_matrix) {
  dst.resize(src.size());
  if (src.empty()) return;
  static_assert(sizeof(Point2f) == 2 * sizeof(float), "Point2f must be two packed floats");
  const int rc = mpe_distort_points(&src[0].x, &dst[0].x, (int)src.size(), camera_matrix_K.data(),
                                    distortion_matrix.empty() ? 0 : distortion_matrix.data(),
                                    (int)distortion_matrix.size());
  if (rc != MPE_OK) {
    std::lock_guard<std::mutex> guard(g_lock);
    ok(0, rc, "mpe_distort_points");
    dst = src;
  }
}

MPE_FACADE_END  // namespace monocular_pose_estimator
