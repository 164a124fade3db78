// primitives.cpp — the static classes P3P and LEDDetector on top of the C ABI.  The reference's static
// functions carry no state; here they share one lazily created library handle per process.
#include "facade_namespace.h"
#include <mutex>
#include <stdexcept>
#include <string>

#include "led_detector.h"
#include "mpe.h"
#include "p3p.h"

MPE_FACADE_BEGIN

namespace {
std::mutex g_lock;  // one handle = one caller at a time (include/mpe.h)
mpe_handle* shared_handle() {
  static mpe_handle* h = 0;
  if (!h && mpe_create(&h, -1) != MPE_OK) {
    h = 0;
    throw std::runtime_error("mpe_create failed: no HIP device (there is no CPU fallback)");
  }
  return h;
}
void check(mpe_handle* h, int rc, const char* what) {
  if (rc != MPE_OK) throw std::runtime_error(std::string(what) + ": " + mpe_last_error(h));
}
}  // namespace

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
  check(h, mpe_p3p_batch(h, fv, wp, 1, sol, &status), "mpe_p3p_batch");
  if (status != 0) return -1;
  for (int s = 0; s < 4; ++s)
    for (int i = 0; i < 12; ++i) solutions[s](i) = sol[12 * s + i];
  return 0;
}

int P3P::solveQuartic(const Vector5d& factors, Vector4d& real_roots) {
  std::lock_guard<std::mutex> guard(g_lock);
  mpe_handle* h = shared_handle();
  check(h, mpe_solve_quartic_batch(h, factors.data(), 1, 0, real_roots.data()), "mpe_solve_quartic_batch");
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
  check(h,
        mpe_find_leds(h, image.data, image.rows, image.cols, image.step, ROI.x, ROI.y, ROI.width, ROI.height, &p,
                      camera_matrix_K.data(), camera_distortion_coeffs.empty() ? 0 : camera_distortion_coeffs.data(),
                      (int)camera_distortion_coeffs.size(), und, dist, MPE_MAX_DETECTIONS, &n),
        "mpe_find_leds");
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
  if (rc != MPE_OK) throw std::runtime_error("mpe_determine_roi: bad argument");
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
  if (rc != MPE_OK) throw std::runtime_error("mpe_distort_points: bad argument");
}

MPE_FACADE_END  // namespace monocular_pose_estimator
