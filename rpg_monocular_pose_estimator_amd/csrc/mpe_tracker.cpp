// mpe_tracker.cpp — the stateful PoseEstimator state machine (tracking path) on top of the HIP
// stages.  Host side only: a few 4x4 / 6-vector operations per frame; all image and pose compute
// runs in the kernels through the public C ABI (mpe_find_leds with a ROI, mpe_check_and_refine,
// mpe_solve_bruteforce).
//
// Reference: PoseEstimator::estimateBodyPose (pose_estimator.cpp:62-147), predictPose (:232-244),
// predictMarkerPositionsInImage (:270-276), findCorrespondences (:372-392), updatePose /
// optimiseAndUpdatePose / predictWithROI / findCorrespondencesAndPredictPose (:794-848),
// logarithmMap / exponentialMap (:962-1064), LEDDetector::determineROI / distortPoints
// (led_detector.cpp:114-224).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../../include/mpe.h"

namespace {

struct Mat4 {
  double a[4][4];
};

Mat4 eye4() {
  Mat4 m;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) m.a[i][j] = i == j ? 1.0 : 0.0;
  return m;
}

Mat4 matmul(const Mat4& x, const Mat4& y) {
  Mat4 r;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += x.a[i][k] * y.a[k][j];
      r.a[i][j] = s;
    }
  return r;
}

// general inverse (Gauss-Jordan, partial pivoting) — Eigen's Matrix4d::inverse() in the reference
Mat4 inverse(const Mat4& m) {
  double w[4][8];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      w[i][j] = m.a[i][j];
      w[i][4 + j] = i == j ? 1.0 : 0.0;
    }
  for (int c = 0; c < 4; ++c) {
    int piv = c;
    for (int r = c + 1; r < 4; ++r)
      if (std::fabs(w[r][c]) > std::fabs(w[piv][c])) piv = r;
    if (piv != c)
      for (int j = 0; j < 8; ++j) std::swap(w[c][j], w[piv][j]);
    const double d = w[c][c];
    for (int j = 0; j < 8; ++j) w[c][j] /= d;
    for (int r = 0; r < 4; ++r) {
      if (r == c) continue;
      const double f = w[r][c];
      for (int j = 0; j < 8; ++j) w[r][j] -= f * w[c][j];
    }
  }
  Mat4 out;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) out.a[i][j] = w[i][4 + j];
  return out;
}

// pose_estimator.cpp:962-994
Mat4 exp_se3(const double tw[6]) {
  const double wx = tw[3], wy = tw[4], wz = tw[5];
  const double theta = std::sqrt(wx * wx + wy * wy + wz * wz), th2 = theta * theta;
  const double O[3][3] = {{0, -wz, wy}, {wz, 0, -wx}, {-wy, wx, 0}};
  double O2[3][3], R[3][3], V[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) O2[i][j] = O[i][0] * O[0][j] + O[i][1] * O[1][j] + O[i][2] * O[2][j];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const double I = i == j ? 1.0 : 0.0;
      if (theta == 0) {
        R[i][j] = V[i][j] = I;
      } else {
        R[i][j] = I + O[i][j] / theta * std::sin(theta) + O2[i][j] / th2 * (1 - std::cos(theta));
        V[i][j] = I + (1 - std::cos(theta)) / th2 * O[i][j] + (theta - std::sin(theta)) / (th2 * theta) * O2[i][j];
      }
    }
  Mat4 T = eye4();
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T.a[i][j] = R[i][j];
    T.a[i][3] = V[i][0] * tw[0] + V[i][1] * tw[1] + V[i][2] * tw[2];
  }
  return T;
}

// pose_estimator.cpp:996-1064
void log_se3(const Mat4& T, double xi[6]) {
  double what[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  double dev = 0, nr = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const double e = T.a[i][j] - (i == j ? 1.0 : 0.0);
      dev += e * e;
      nr += T.a[i][j] * T.a[i][j];
    }
  // Eigen isApprox(Identity, 1e-10): squared distance <= prec^2 * min(squared norms)
  if (!(dev <= 1e-20 * std::min(nr, 3.0))) {
    double c = (T.a[0][0] + T.a[1][1] + T.a[2][2] - 1) / 2;
    c = c > 1 ? 1 : (c < -1 ? -1 : c);
    const double phi = std::acos(c);
    if (phi != 0)
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) what[i][j] = (T.a[i][j] - T.a[j][i]) / (2 * std::sin(phi)) * phi;
  }
  const double w[3] = {what[2][1], what[0][2], what[1][0]};
  const double wn = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const double t[3] = {T.a[0][3], T.a[1][3], T.a[2][3]};
  double Ai[3][3];
  if (t[0] == 0 && t[1] == 0 && t[2] == 0) {  // isApproxToConstant(0, 1e-10) holds only for exact zeros
    std::memset(Ai, 0, sizeof(Ai));
  } else if (wn == 0 || std::sin(wn) == 0) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Ai[i][j] = i == j ? 1.0 : 0.0;
  } else {
    const double k = (2 * std::sin(wn) - wn * (1 + std::cos(wn))) / (2 * wn * wn * std::sin(wn));
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        const double sq = what[i][0] * what[0][j] + what[i][1] * what[1][j] + what[i][2] * what[2][j];
        Ai[i][j] = (i == j ? 1.0 : 0.0) - what[i][j] / 2 + k * sq;
      }
  }
  for (int i = 0; i < 3; ++i) {
    xi[i] = Ai[i][0] * t[0] + Ai[i][1] * t[1] + Ai[i][2] * t[2];
    xi[3 + i] = w[i];
  }
}

// predictPose, pose_estimator.cpp:232-244
Mat4 predict_pose(const Mat4& current, const Mat4& previous, double t_current, double t_previous, double t_predicted) {
  double delta[6], dh[6];
  log_se3(matmul(inverse(previous), current), delta);
  for (int i = 0; i < 6; ++i) dh[i] = delta[i] / (t_current - t_previous) * (t_predicted - t_current);
  return matmul(current, exp_se3(dh));
}

// project2d, pose_estimator.cpp:251-268: (K [I|0] * T) * point, evaluated left to right like the reference
void project_point(const double K[9], const Mat4& T, const double* m, double& u, double& v) {
  double t[3];
  for (int i = 0; i < 3; ++i) {
    double ct[4];
    for (int j = 0; j < 4; ++j) {
      double s = K[3 * i] * T.a[0][j];
      s += K[3 * i + 1] * T.a[1][j];
      s += K[3 * i + 2] * T.a[2][j];
      s += 0.0 * T.a[3][j];
      ct[j] = s;
    }
    double s = ct[0] * m[0];
    s += ct[1] * m[1];
    s += ct[2] * m[2];
    s += ct[3] * 1.0;
    t[i] = s;
  }
  u = t[0] / t[2];
  v = t[1] / t[2];
}

// findCorrespondences, pose_estimator.cpp:372-392: nearest detection of every predicted marker pixel
// (first minimum wins), kept if within the tolerance.  corr: rows (marker, detection), 1-based.
int find_correspondences(const double* pred_px, int n_markers, const double* det_xy, int n_det, double tol,
                         uint32_t* corr) {
  int nc = 0;
  for (int i = 0; i < n_markers; ++i) {
    double best = std::numeric_limits<double>::infinity();
    unsigned bj = 0;
    for (int j = 0; j < n_det; ++j) {
      const double du = pred_px[2 * i] - det_xy[2 * j], dv = pred_px[2 * i + 1] - det_xy[2 * j + 1];
      const double d2 = du * du + dv * dv;
      if (d2 < best) {
        best = d2;
        bj = (unsigned)j + 1;
      }
    }
    if (std::sqrt(best) <= tol) {
      corr[2 * nc] = (unsigned)i + 1;
      corr[2 * nc + 1] = bj;
      ++nc;
    }
  }
  return nc;
}

}  // namespace

struct mpe_tracker {
  mpe_handle* h = nullptr;
  mpe_params p;
  double K[9];
  std::vector<double> D;
  std::vector<double> markers;  // n x 3
  Mat4 current, previous, predicted;
  double cov[36];
  double t_current = 0, t_previous = 0, t_predicted = 0;
  unsigned it_since_initialized = 0;
  int roi[4] = {0, 0, 0, 0};
  bool pose_updated = false;
  std::vector<double> predicted_px;  // n_markers x 2
  std::vector<double> det;           // detected_led_positions of the current call
  std::vector<float> det_dist;       // distorted_detection_centers_ (always rewritten by a detection)
  std::vector<uint32_t> corr;        // rows (marker, detection)
  int n_corr = 0, gn_iterations = 0;
  bool used_bruteforce = false;
  // result of the speculative nearest-neighbour + refine pass that mpe_track_step ran together with
  // the detection; only meaningful while fused_valid (>= 4 fresh detections)
  bool fused_valid = false;
  mpe_result fused_res;
  uint32_t fused_corr[2 * MPE_MAX_MARKERS];
};

namespace {

int n_markers(const mpe_tracker* t) { return (int)(t->markers.size() / 3); }

void project(const mpe_tracker* t, const Mat4& T, const double* m, double& u, double& v) {
  project_point(t->K, T, m, u, v);
}

// LEDDetector::distortPoints for one point, float in / float out (led_detector.cpp:181-224)
void distort_point(const double K[9], const double* D, int nD, float sx, float sy, float& ox, float& oy) {
  const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  const double k1 = nD > 0 ? D[0] : 0, k2 = nD > 1 ? D[1] : 0, p1 = nD > 2 ? D[2] : 0, p2 = nD > 3 ? D[3] : 0,
               k3 = nD > 4 ? D[4] : 0;
  const double x = ((double)sx - cx) / fx, y = ((double)sy - cy) / fy;
  const double r2 = x * x + y * y;
  double xc = x * (1. + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2);
  double yc = y * (1. + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2);
  xc = xc + (2. * p1 * x * y + p2 * (r2 + 2. * x * x));
  yc = yc + (p1 * (r2 + 2. * y * y) + 2. * p2 * x * y);
  ox = (float)(xc * fx + cx);
  oy = (float)(yc * fy + cy);
}

// LEDDetector::determineROI (led_detector.cpp:114-179)
void determine_roi_impl(const double* px, int n, int rows, int cols, int border, const double K[9], const double* D,
                        int nD, int roi[4]) {
  double x_min = std::numeric_limits<double>::infinity(), x_max = 0, y_min = x_min, y_max = 0;
  for (int i = 0; i < n; ++i) {
    const double u = px[2 * i], v = px[2 * i + 1];
    if (u < x_min) x_min = u;
    if (u > x_max) x_max = u;
    if (v < y_min) y_min = v;
    if (v > y_max) y_max = v;
  }
  float ax, ay, bx, by;
  distort_point(K, D, nD, (float)x_min, (float)y_min, ax, ay);
  distort_point(K, D, nD, (float)x_max, (float)y_max, bx, by);
  const double x0 = std::max(0.0, std::min((double)cols, (double)ax - border));
  const double x1 = std::max(0.0, std::min((double)cols, (double)bx + border));
  const double y0 = std::max(0.0, std::min((double)rows, (double)ay - border));
  const double y1 = std::max(0.0, std::min((double)rows, (double)by + border));
  if (x1 - x0 < 1 || y1 - y0 < 1) {
    roi[0] = 0;
    roi[1] = 0;
    roi[2] = cols;
    roi[3] = rows;
  } else {
    roi[0] = (int)x0;
    roi[1] = (int)y0;
    roi[2] = (int)(x1 - x0);
    roi[3] = (int)(y1 - y0);
  }
}

void determine_roi(mpe_tracker* t, int rows, int cols) {
  determine_roi_impl(t->predicted_px.data(), n_markers(t), rows, cols, (int)t->p.roi_border_thickness, t->K,
                     t->D.empty() ? nullptr : t->D.data(), (int)t->D.size(), t->roi);
}

int detect(mpe_tracker* t, const uint8_t* img, int rows, int cols, size_t stride) {
  double und[2 * MPE_MAX_DETECTIONS];
  float dist[2 * MPE_MAX_DETECTIONS];
  int n = 0;
  int rc = mpe_find_leds(t->h, img, rows, cols, stride, t->roi[0], t->roi[1], t->roi[2], t->roi[3], &t->p, t->K,
                         t->D.empty() ? nullptr : t->D.data(), (int)t->D.size(), und, dist, MPE_MAX_DETECTIONS, &n);
  if (rc != MPE_OK) return rc;
  t->det_dist.assign(dist, dist + 2 * n);
  if (n > 0) t->det.assign(und, und + 2 * n);  // pixel_positions is only rewritten when something was found
  return MPE_OK;
}

// Tracking branch: detection in the ROI and, in the same device submission, the nearest-neighbour
// correspondences + validation + refinement the reference would run next if >= 4 LEDs are found.
int detect_and_try(mpe_tracker* t, const uint8_t* img, int rows, int cols, size_t stride) {
  mpe_detections d;
  t->fused_valid = false;
  int rc = mpe_track_step(t->h, img, rows, cols, stride, t->roi[0], t->roi[1], t->roi[2], t->roi[3], &t->p, t->K,
                          t->D.empty() ? nullptr : t->D.data(), (int)t->D.size(), t->markers.data(), n_markers(t),
                          t->predicted_px.data(), &d, t->fused_corr, &t->fused_res);
  if (rc != MPE_OK) return rc;
  if (d.status != 0) return d.status;
  t->det_dist.assign(d.dist_xy, d.dist_xy + 2 * d.n);
  if (d.n > 0) t->det.assign(d.undist_xy, d.undist_xy + 2 * d.n);  // as in detect(): kept when nothing was found
  t->fused_valid = d.n >= 4;
  return MPE_OK;
}

void take_result(mpe_tracker* t, const mpe_result& r) {
  std::memcpy(t->predicted.a, r.T, sizeof(r.T));
  std::memcpy(t->cov, r.cov, sizeof(r.cov));
  t->gn_iterations = r.gn_iterations;
  // optimiseAndUpdatePose (pose_estimator.cpp:802-812) + updatePose (:794-800)
  if (t->it_since_initialized < 2) t->it_since_initialized++;
  t->previous = t->current;
  t->current = t->predicted;
  t->t_previous = t->t_current;
  t->t_current = t->t_predicted;
  t->pose_updated = true;
}

int bruteforce(mpe_tracker* t) {  // initialise() + optimiseAndUpdatePose()
  mpe_result r;
  std::vector<uint32_t> c(2 * MPE_MAX_MARKERS, 0);
  const int nd = (int)t->det.size() / 2, nm = n_markers(t);
  std::vector<uint32_t> hist((size_t)std::max(1, nd * nm), 0);
  t->used_bruteforce = true;
  int rc = mpe_solve_bruteforce(t->h, t->det.data(), nd, t->markers.data(), nm, t->K, &t->p, &r, hist.data(), c.data());
  if (rc != MPE_OK) return rc;
  if (r.status < 0) return r.status;
  // initialise() only assigns correspondences_ when the histogram holds a vote (pose_estimator.cpp:704-719);
  // with an all-zero histogram the member keeps its previous rows
  bool any_vote = false;
  for (uint32_t v : hist) any_vote |= (v != 0);
  if (any_vote) {
    t->n_corr = r.n_corr;
    t->corr.assign(c.begin(), c.begin() + 2 * r.n_corr);
  }
  if (r.status == MPE_FRAME_POSE) take_result(t, r);
  return MPE_OK;
}

// findCorrespondencesAndPredictPose (pose_estimator.cpp:831-848)
int track(mpe_tracker* t) {
  if (t->fused_valid) {  // already done on the device together with the detection
    t->fused_valid = false;
    const mpe_result& r = t->fused_res;
    if (r.status < 0) return r.status;
    t->n_corr = r.n_corr;
    t->corr.assign(t->fused_corr, t->fused_corr + 2 * r.n_corr);
    if (r.status == MPE_FRAME_POSE) {
      take_result(t, r);
      return MPE_OK;
    }
    return bruteforce(t);
  }
  const int nm = n_markers(t), nd = (int)t->det.size() / 2;
  t->corr.assign(2 * (size_t)nm, 0u);
  const int nc = find_correspondences(t->predicted_px.data(), nm, t->det.data(), nd,
                                      t->p.nearest_neighbour_pixel_tolerance, t->corr.data());
  t->corr.resize(2 * (size_t)nc);
  t->n_corr = (int)t->corr.size() / 2;
  mpe_result r;
  int rc = mpe_check_and_refine(t->h, t->det.data(), nd, t->markers.data(), nm, t->K, &t->p, t->corr.data(), t->n_corr,
                                &r);
  if (rc != MPE_OK) return rc;
  if (r.status < 0) return r.status;
  if (r.status == MPE_FRAME_POSE) {
    take_result(t, r);
    return MPE_OK;
  }
  return bruteforce(t);  // reinitialise if the correspondences were not valid
}

}  // namespace

extern "C" {

int mpe_determine_roi(const double* pixel_positions, int n_points, int rows, int cols, int border_size, const double K[9],
                      const double* D, int nD, int roi_xywh[4]) {
  if (!pixel_positions || n_points <= 0 || !K || !roi_xywh || rows <= 0 || cols <= 0 || nD < 0 || (nD > 0 && !D))
    return MPE_ERR_ARG;
  determine_roi_impl(pixel_positions, n_points, rows, cols, border_size, K, D, nD, roi_xywh);
  return MPE_OK;
}

int mpe_distort_points(const float* src_xy, float* dst_xy, int n, const double K[9], const double* D, int nD) {
  if (n < 0 || (n > 0 && (!src_xy || !dst_xy)) || !K || nD < 0 || (nD > 0 && !D)) return MPE_ERR_ARG;
  for (int i = 0; i < n; ++i) distort_point(K, D, nD, src_xy[2 * i], src_xy[2 * i + 1], dst_xy[2 * i], dst_xy[2 * i + 1]);
  return MPE_OK;
}

int mpe_exponential_map(const double twist[6], double T[16]) {
  if (!twist || !T) return MPE_ERR_ARG;
  const Mat4 m = exp_se3(twist);
  std::memcpy(T, m.a, sizeof(m.a));
  return MPE_OK;
}

int mpe_logarithm_map(const double T[16], double twist[6]) {
  if (!twist || !T) return MPE_ERR_ARG;
  Mat4 m;
  std::memcpy(m.a, T, sizeof(m.a));
  log_se3(m, twist);
  return MPE_OK;
}

int mpe_predict_pose(const double current_pose[16], const double previous_pose[16], double current_time,
                     double previous_time, double time_to_predict, double predicted_pose[16]) {
  if (!current_pose || !previous_pose || !predicted_pose) return MPE_ERR_ARG;
  Mat4 c, pr;
  std::memcpy(c.a, current_pose, sizeof(c.a));
  std::memcpy(pr.a, previous_pose, sizeof(pr.a));
  const Mat4 out = predict_pose(c, pr, current_time, previous_time, time_to_predict);
  std::memcpy(predicted_pose, out.a, sizeof(out.a));
  return MPE_OK;
}

int mpe_project_points(const double T[16], const double* markers_xyz, int n, const double K[9], double* px) {
  if (!T || !K || n < 0 || (n > 0 && (!markers_xyz || !px))) return MPE_ERR_ARG;
  Mat4 m;
  std::memcpy(m.a, T, sizeof(m.a));
  for (int i = 0; i < n; ++i) project_point(K, m, markers_xyz + 3 * i, px[2 * i], px[2 * i + 1]);
  return MPE_OK;
}

int mpe_find_correspondences(const double* predicted_px, int n_markers, const double* det_xy, int n_det,
                             double nearest_neighbour_pixel_tolerance, uint32_t* corr) {
  if (n_markers < 0 || n_det < 0 || (n_markers > 0 && (!predicted_px || !corr)) || (n_det > 0 && !det_xy))
    return MPE_ERR_ARG;
  return find_correspondences(predicted_px, n_markers, det_xy, n_det, nearest_neighbour_pixel_tolerance, corr);
}

int mpe_tracker_get_state(const mpe_tracker* t, mpe_tracker_state* st) {
  if (!t || !st) return MPE_ERR_ARG;
  std::memcpy(st->current_pose, t->current.a, sizeof(st->current_pose));
  std::memcpy(st->previous_pose, t->previous.a, sizeof(st->previous_pose));
  std::memcpy(st->predicted_pose, t->predicted.a, sizeof(st->predicted_pose));
  std::memcpy(st->pose_covariance, t->cov, sizeof(st->pose_covariance));
  st->current_time = t->t_current;
  st->previous_time = t->t_previous;
  st->predicted_time = t->t_predicted;
  st->it_since_initialized = t->it_since_initialized;
  for (int i = 0; i < 4; ++i) st->roi[i] = t->roi[i];
  return MPE_OK;
}

int mpe_tracker_set_state(mpe_tracker* t, const mpe_tracker_state* st) {
  if (!t || !st) return MPE_ERR_ARG;
  std::memcpy(t->current.a, st->current_pose, sizeof(st->current_pose));
  std::memcpy(t->previous.a, st->previous_pose, sizeof(st->previous_pose));
  std::memcpy(t->predicted.a, st->predicted_pose, sizeof(st->predicted_pose));
  std::memcpy(t->cov, st->pose_covariance, sizeof(st->pose_covariance));
  t->t_current = st->current_time;
  t->t_previous = st->previous_time;
  t->t_predicted = st->predicted_time;
  t->it_since_initialized = st->it_since_initialized;
  for (int i = 0; i < 4; ++i) t->roi[i] = st->roi[i];
  return MPE_OK;
}

int mpe_tracker_create(mpe_handle* h, mpe_tracker** out) {
  if (!h || !out) return MPE_ERR_ARG;
  mpe_tracker* t = new mpe_tracker();
  t->h = h;
  mpe_default_params(&t->p);
  t->p.back_projection_pixel_tolerance = 3;    // constructor defaults, pose_estimator.cpp:34-42
  t->p.nearest_neighbour_pixel_tolerance = 5;
  t->p.certainty_threshold = 0.75;
  t->p.valid_correspondence_threshold = 0.7;
  std::memset(t->K, 0, sizeof(t->K));
  t->current = t->previous = t->predicted = eye4();
  std::memset(t->cov, 0, sizeof(t->cov));
  *out = t;
  return MPE_OK;
}

void mpe_tracker_destroy(mpe_tracker* t) { delete t; }

int mpe_tracker_set_markers(mpe_tracker* t, const double* xyz, int n) {
  if (!t || (!xyz && n > 0) || n < 0 || n > MPE_MAX_MARKERS) return MPE_ERR_ARG;
  t->markers.assign(xyz, xyz + 3 * n);
  t->predicted_px.assign(2 * n, 0.0);
  t->p.histogram_threshold = 0;  // numCombinations(n,3), pose_estimator.cpp:54
  return MPE_OK;
}

int mpe_tracker_set_camera(mpe_tracker* t, const double K[9], const double* D, int nD) {
  if (!t || !K || nD < 0 || (nD > 0 && !D)) return MPE_ERR_ARG;
  std::memcpy(t->K, K, sizeof(t->K));
  if (nD > 0)
    t->D.assign(D, D + nD);
  else
    t->D.clear();
  return MPE_OK;
}

int mpe_tracker_set_params(mpe_tracker* t, const mpe_params* p) {
  if (!t || !p) return MPE_ERR_ARG;
  t->p = *p;
  return MPE_OK;
}

int mpe_tracker_reset(mpe_tracker* t) {
  if (!t) return MPE_ERR_ARG;
  t->it_since_initialized = 0;
  if (t->h) (void)mpe_track_step_batch_cancel(t->h);  // (a submission abandoned by a failed batch call)
  return MPE_OK;
}

int mpe_tracker_get_correspondences(mpe_tracker* t, uint32_t* corr, int cap_rows) {  // getCorrespondences()
  if (!t || (!corr && cap_rows > 0)) return MPE_ERR_ARG;
  const int n = std::min<int>((int)t->corr.size() / 2, cap_rows);
  for (int i = 0; i < 2 * n; ++i) corr[i] = t->corr[i];
  return (int)t->corr.size() / 2;
}

int mpe_tracker_get_image_points(mpe_tracker* t, double* xy, int cap_points) {  // getImagePoints()
  if (!t || (!xy && cap_points > 0)) return MPE_ERR_ARG;
  const int n = std::min<int>((int)t->det.size() / 2, cap_points);
  for (int i = 0; i < 2 * n; ++i) xy[i] = t->det[i];
  return (int)t->det.size() / 2;
}

int mpe_tracker_get_distorted_centers(mpe_tracker* t, float* xy, int cap_points) {  // distorted_detection_centers_
  if (!t || (!xy && cap_points > 0)) return MPE_ERR_ARG;
  const int n = std::min<int>((int)t->det_dist.size() / 2, cap_points);
  for (int i = 0; i < 2 * n; ++i) xy[i] = t->det_dist[i];
  return (int)t->det_dist.size() / 2;
}

int mpe_tracker_estimate(mpe_tracker* t, const uint8_t* img, int rows, int cols, size_t stride_bytes, double time,
                         mpe_result* out, int info[8]) {
  if (!t || !img) return MPE_ERR_ARG;
  t->pose_updated = false;
  t->used_bruteforce = false;
  t->det.clear();
  // correspondences_ is a member of the reference object: it keeps its last value when a frame has
  // too few detections (getCorrespondences() then returns the stale rows) — same here
  int rc = MPE_OK;
  if (t->it_since_initialized < 1) {  // pose_estimator.cpp:68-96
    t->t_predicted = time;
    t->roi[0] = t->roi[1] = 0;
    t->roi[2] = cols;
    t->roi[3] = rows;
    if ((rc = detect(t, img, rows, cols, stride_bytes)) != MPE_OK) return rc;
    if (t->det.size() / 2 >= 4)
      if ((rc = bruteforce(t)) != MPE_OK) return rc;
  } else {  // pose_estimator.cpp:98-144
    // predictWithROI (:814-829)
    if (t->it_since_initialized >= 2) {
      // predictPose (:232-244)
      t->t_predicted = time;
      t->predicted = predict_pose(t->current, t->previous, t->t_current, t->t_previous, t->t_predicted);
    } else {
      t->t_predicted = time;
    }
    for (int i = 0; i < n_markers(t); ++i)
      project(t, t->predicted, &t->markers[3 * i], t->predicted_px[2 * i], t->predicted_px[2 * i + 1]);
    determine_roi(t, rows, cols);
    if ((rc = detect_and_try(t, img, rows, cols, stride_bytes)) != MPE_OK) return rc;
    bool repeat_check = true;
    unsigned num_loops = 0;
    do {
      num_loops++;
      if (t->det.size() / 2 >= 4) {
        if ((rc = track(t)) != MPE_OK) return rc;
        repeat_check = false;
      } else if (num_loops < 2) {  // too few LEDs in the ROI: search the whole image once
        t->roi[0] = t->roi[1] = 0;
        t->roi[2] = cols;
        t->roi[3] = rows;
        if ((rc = detect_and_try(t, img, rows, cols, stride_bytes)) != MPE_OK) return rc;
      } else {
        repeat_check = false;
      }
    } while (repeat_check);
  }
  if (out) {
    std::memcpy(out->T, t->predicted.a, sizeof(out->T));
    std::memcpy(out->cov, t->cov, sizeof(out->cov));
    out->status = t->pose_updated ? MPE_FRAME_POSE : MPE_FRAME_NO_POSE;
    out->n_det = (int)t->det.size() / 2;
    out->n_corr = t->n_corr;
    out->gn_iterations = t->gn_iterations;
  }
  if (info) {
    for (int i = 0; i < 4; ++i) info[i] = t->roi[i];
    info[4] = (int)t->it_since_initialized;
    info[5] = (int)t->det.size() / 2;
    info[6] = t->n_corr;
    info[7] = t->used_bruteforce ? 1 : 0;
  }
  return t->pose_updated ? 1 : 0;
}

// ---- N trackers in lock step -----------------------------------------------------------------------
// The state machine of mpe_tracker_estimate per stream, with the device steps of all streams that are at the same
// point batched into one submission: DETECT = findLeds in the stream's ROI (+ nearest-neighbour correspondences,
// validation and refinement when tracking), BRUTE = initialise() + optimisePose on the stream's detections.
// A frame is processed in three parts so that a caller can overlap the host work of one group of streams with the
// device work of another: begin (per-stream prediction / ROI), the first DETECT submission (asynchronous), and
// finish (collect it, then whatever else the streams need — second size class, whole-image retries,
// re-initialisations — synchronously, until every stream is done).
}  // extern "C"

namespace {
enum BatchOp { OP_DETECT, OP_BRUTE, OP_DONE };
struct BatchLane {
  BatchOp op = OP_DONE;
  bool tracking = false;  // false: uninitialised branch (detection only, then brute force)
  unsigned num_loops = 0;
  int error = 0;          // per-stream capacity status (< 0) that ended this stream's frame
};

bool same_setup(const mpe_tracker* a, const mpe_tracker* b) {
  return a->h == b->h && a->markers == b->markers && a->D == b->D && !std::memcmp(a->K, b->K, sizeof(a->K)) &&
         !std::memcmp(&a->p, &b->p, sizeof(mpe_params));
}

struct BatchCtx {
  mpe_tracker* const* ts = nullptr;
  int n = 0;
  const uint8_t* const* imgs = nullptr;
  int rows = 0, cols = 0;
  size_t stride = 0;
  mpe_handle* h = nullptr;
  int nm = 0, nD = 0;
  const double* Dp = nullptr;
  std::vector<BatchLane> L;
  std::vector<int> pend;  // lanes of the submitted, not yet collected DETECT batch
  std::vector<mpe_track_item> items;
  std::vector<mpe_detections> dets;
  std::vector<uint32_t> corr;
  std::vector<mpe_result> res;

  int validate(mpe_tracker* const* ts_, int n_) {
    ts = ts_;
    n = n_;
    for (int i = 0; i < n; ++i) {
      if (!ts[i]) return MPE_ERR_ARG;
      if (!same_setup(ts[0], ts[i])) return MPE_ERR_ARG;  // one handle, camera model, marker set, parameter set
      for (int j = 0; j < i; ++j)
        if (ts[j] == ts[i]) return MPE_ERR_ARG;
    }
    h = ts[0]->h;
    nm = n_markers(ts[0]);
    Dp = ts[0]->D.empty() ? nullptr : ts[0]->D.data();
    nD = (int)ts[0]->D.size();
    L.assign((size_t)n, BatchLane());
    return MPE_OK;
  }

  // begin of frame: pose_estimator.cpp:62-72 / 98-103 (predictWithROI)
  void begin(const uint8_t* const* imgs_, int rows_, int cols_, size_t stride_, const double* times) {
    imgs = imgs_;
    rows = rows_;
    cols = cols_;
    stride = stride_;
    for (int i = 0; i < n; ++i) {
      mpe_tracker* t = ts[i];
      L[(size_t)i] = BatchLane();
      t->pose_updated = false;
      t->used_bruteforce = false;
      t->det.clear();
      t->fused_valid = false;
      t->t_predicted = times[i];
      if (t->it_since_initialized < 1) {
        t->roi[0] = t->roi[1] = 0;
        t->roi[2] = cols;
        t->roi[3] = rows;
        L[(size_t)i].tracking = false;
      } else {
        if (t->it_since_initialized >= 2)
          t->predicted = predict_pose(t->current, t->previous, t->t_current, t->t_previous, t->t_predicted);
        for (int k = 0; k < nm; ++k)
          project(t, t->predicted, &t->markers[3 * k], t->predicted_px[2 * k], t->predicted_px[2 * k + 1]);
        determine_roi(t, rows, cols);
        L[(size_t)i].tracking = true;
      }
      L[(size_t)i].op = OP_DETECT;
    }
  }

  bool is_big(int i) const { return (long long)ts[i]->roi[2] * ts[i]->roi[3] * 4 > (long long)rows * cols; }

  // submit the DETECT lanes of one size class (two classes, so that a few whole-image retries do not inflate every
  // ROI slot of the batch); returns the number of lanes submitted or < 0
  int submit_detect(int big) {
    items.clear();
    pend.clear();
    for (int i = 0; i < n; ++i) {
      if (L[(size_t)i].op != OP_DETECT || (int)is_big(i) != big) continue;
      const mpe_tracker* t = ts[i];
      mpe_track_item it;
      it.img = imgs[i];
      it.roi_x = t->roi[0];
      it.roi_y = t->roi[1];
      it.roi_w = t->roi[2];
      it.roi_h = t->roi[3];
      it.predicted_px = L[(size_t)i].tracking ? t->predicted_px.data() : nullptr;
      items.push_back(it);
      pend.push_back(i);
    }
    if (items.empty()) return 0;
    const int rc = mpe_track_step_batch_submit(h, items.data(), (int)items.size(), rows, cols, stride, &ts[0]->p, ts[0]->K,
                                               Dp, nD, ts[0]->markers.data(), nm);
    if (rc != MPE_OK) {
      pend.clear();
      return rc;
    }
    return (int)items.size();
  }

  // collect the submitted DETECT batch and advance its lanes
  int collect_detect() {
    if (pend.empty()) return MPE_OK;
    const int m = (int)pend.size();
    dets.resize((size_t)m);
    corr.resize((size_t)m * 2 * MPE_MAX_MARKERS);
    res.resize((size_t)m);
    const int rc = mpe_track_step_batch_collect(h, dets.data(), corr.data(), res.data());
    if (rc != MPE_OK) return rc;
    for (int k = 0; k < m; ++k) {
      const int i = pend[(size_t)k];
      mpe_tracker* t = ts[i];
      BatchLane& ln = L[(size_t)i];
      const mpe_detections& d = dets[(size_t)k];
      if (d.status != 0) {  // device capacity exceeded on this stream's frame
        ln.error = d.status;
        ln.op = OP_DONE;
        continue;
      }
      t->det_dist.assign(d.dist_xy, d.dist_xy + 2 * d.n);
      if (d.n > 0) t->det.assign(d.undist_xy, d.undist_xy + 2 * d.n);  // kept when nothing was found
      if (!ln.tracking) {  // pose_estimator.cpp:80-91
        ln.op = (t->det.size() / 2 >= 4) ? OP_BRUTE : OP_DONE;
        continue;
      }
      ln.num_loops++;  // pose_estimator.cpp:105-144
      if (t->det.size() / 2 >= 4) {
        // (>= 4 detections can only come from this very detection: the device already ran findCorrespondences +
        //  checkCorrespondences + optimisePose on them)
        const mpe_result& r = res[(size_t)k];
        if (r.status < 0) {
          ln.error = r.status;
          ln.op = OP_DONE;
          continue;
        }
        t->n_corr = r.n_corr;
        t->corr.assign(corr.begin() + (size_t)k * 2 * MPE_MAX_MARKERS,
                       corr.begin() + (size_t)k * 2 * MPE_MAX_MARKERS + 2 * r.n_corr);
        if (r.status == MPE_FRAME_POSE) {
          take_result(t, r);
          ln.op = OP_DONE;
        } else {
          ln.op = OP_BRUTE;  // reinitialise if the correspondences were not valid
        }
      } else if (ln.num_loops < 2) {  // too few LEDs in the ROI: search the whole image once
        t->roi[0] = t->roi[1] = 0;
        t->roi[2] = cols;
        t->roi[3] = rows;
        ln.op = OP_DETECT;
      } else {
        ln.op = OP_DONE;
      }
    }
    pend.clear();
    return MPE_OK;
  }

  // brute-force (re-)initialisations requested so far, one submission
  int brute() {
    std::vector<int> idx;
    for (int i = 0; i < n; ++i)
      if (L[(size_t)i].op == OP_BRUTE) idx.push_back(i);
    if (idx.empty()) return MPE_OK;
    const int m = (int)idx.size();
    std::vector<double> det_xy((size_t)m * 2 * MPE_MAX_DETECTIONS, 0.0);
    std::vector<int> nd((size_t)m);
    for (int k = 0; k < m; ++k) {
      const mpe_tracker* t = ts[idx[(size_t)k]];
      nd[(size_t)k] = (int)t->det.size() / 2;
      std::memcpy(&det_xy[(size_t)k * 2 * MPE_MAX_DETECTIONS], t->det.data(), t->det.size() * sizeof(double));
    }
    res.resize((size_t)m);
    std::vector<uint32_t> hist((size_t)m * MPE_MAX_DETECTIONS * MPE_MAX_MARKERS), bc((size_t)m * 2 * MPE_MAX_MARKERS);
    const int rc = mpe_solve_bruteforce_batch(h, det_xy.data(), nd.data(), m, ts[0]->markers.data(), nm, ts[0]->K, &ts[0]->p,
                                              res.data(), hist.data(), bc.data());
    if (rc != MPE_OK) return rc;
    for (int k = 0; k < m; ++k) {
      const int i = idx[(size_t)k];
      mpe_tracker* t = ts[i];
      const mpe_result& r = res[(size_t)k];
      t->used_bruteforce = true;
      L[(size_t)i].op = OP_DONE;
      if (r.status < 0) {
        L[(size_t)i].error = r.status;
        continue;
      }
      bool any_vote = false;  // initialise() keeps correspondences_ when the histogram is empty (:704-719)
      for (size_t q = 0; q < (size_t)MPE_MAX_DETECTIONS * MPE_MAX_MARKERS; ++q)
        any_vote |= hist[(size_t)k * MPE_MAX_DETECTIONS * MPE_MAX_MARKERS + q] != 0;
      if (any_vote) {
        t->n_corr = r.n_corr;
        t->corr.assign(bc.begin() + (size_t)k * 2 * MPE_MAX_MARKERS, bc.begin() + (size_t)k * 2 * MPE_MAX_MARKERS + 2 * r.n_corr);
      }
      if (r.status == MPE_FRAME_POSE) take_result(t, r);
    }
    return MPE_OK;
  }

  // error path: abandon whatever this group has in flight, so that its handle accepts the next submission
  void cancel() {
    pend.clear();
    if (h) (void)mpe_track_step_batch_cancel(h);
  }

  // first asynchronous submission of a frame: the size class that has lanes (the small one first)
  int submit_first() {
    const int rc = submit_detect(0);
    if (rc != 0) return rc < 0 ? rc : MPE_OK;
    const int rc2 = submit_detect(1);
    return rc2 < 0 ? rc2 : MPE_OK;
  }

  // everything else of the frame, synchronously, until every stream is done
  int finish() {
    int rc = collect_detect();
    if (rc != MPE_OK) return rc;
    for (;;) {
      bool any_detect = false, any_brute = false;
      for (int i = 0; i < n; ++i) {
        any_detect |= L[(size_t)i].op == OP_DETECT;
        any_brute |= L[(size_t)i].op == OP_BRUTE;
      }
      if (!any_detect && !any_brute) break;
      if (any_detect)
        for (int big = 0; big < 2; ++big) {
          const int m = submit_detect(big);
          if (m < 0) return m;
          if (m > 0 && (rc = collect_detect()) != MPE_OK) return rc;
        }
      if ((rc = brute()) != MPE_OK) return rc;
    }
    return MPE_OK;
  }

  int outputs(mpe_result* out, size_t out_stride, int* info, size_t info_stride, int* updated) const {
    int n_updated = 0;
    for (int i = 0; i < n; ++i) {
      const mpe_tracker* t = ts[i];
      const int err = L[(size_t)i].error;
      if (out) {
        mpe_result& o = out[(size_t)i * out_stride];
        if (err) {  // a device capacity was exceeded on this stream's frame: as mpe_tracker_run_sequence reports it —
          std::memset(&o, 0, sizeof(o));  // a zeroed record that carries the code
          o.status = err;
        } else {
          std::memcpy(o.T, t->predicted.a, sizeof(o.T));
          std::memcpy(o.cov, t->cov, sizeof(o.cov));
          o.status = t->pose_updated ? MPE_FRAME_POSE : MPE_FRAME_NO_POSE;
          o.n_det = (int)t->det.size() / 2;
          o.n_corr = t->n_corr;
          o.gn_iterations = t->gn_iterations;
        }
      }
      if (info) {
        int* q = info + (size_t)i * info_stride;
        if (err) {
          std::memset(q, 0, 8 * sizeof(int));
        } else {
          for (int k = 0; k < 4; ++k) q[k] = t->roi[k];
          q[4] = (int)t->it_since_initialized;
          q[5] = (int)t->det.size() / 2;
          q[6] = t->n_corr;
          q[7] = t->used_bruteforce ? 1 : 0;
        }
      }
      if (updated) updated[i] = t->pose_updated ? 1 : 0;
      n_updated += t->pose_updated ? 1 : 0;
    }
    return n_updated;
  }
};
}  // namespace

extern "C" {

int mpe_tracker_estimate_batch(mpe_tracker* const* ts, int n, const uint8_t* const* imgs, int rows, int cols,
                               size_t stride_bytes, const double* times, mpe_result* out, int* info, int* updated) {
  if (!ts || n < 0 || !imgs || !times) return MPE_ERR_ARG;
  if (n == 0) return 0;
  for (int i = 0; i < n; ++i)
    if (!imgs[i]) return MPE_ERR_ARG;
  BatchCtx c;
  int rc = c.validate(ts, n);
  if (rc != MPE_OK) return rc;
  c.begin(imgs, rows, cols, stride_bytes, times);
  if ((rc = c.submit_first()) != MPE_OK || (rc = c.finish()) != MPE_OK) {
    c.cancel();  // (a submission may still be in flight: leave the handle usable)
    return rc;
  }
  return c.outputs(out, 1, info, 8, updated);
}

// The lock-step loops of N streams over recorded sequences.  Trackers that live on DIFFERENT handles form groups
// (one group per handle, each with the same camera / marker / parameter set inside the group); the groups are
// pipelined against each other: while the device works on step k of one group, the host collects, advances and
// packs another — the host work of a time step (~3 us per stream) hides behind the device latency (~0.2 ms).
int mpe_tracker_run_sequences_batch_threads(mpe_tracker* const* ts, int n, const uint8_t* const* frames, int n_frames,
                                            int rows, int cols, size_t stride_bytes, size_t frame_stride_bytes,
                                            const double* times, mpe_result* out, int* info, int n_threads) {
  if (!ts || n < 0 || !frames || !times || n_frames < 0 || n_threads < 1) return MPE_ERR_ARG;
  if (n == 0 || n_frames == 0) return 0;
  for (int i = 0; i < n; ++i)
    if (!ts[i] || !frames[i]) return MPE_ERR_ARG;
  // groups by handle, in order of first appearance
  std::vector<mpe_handle*> hs;
  std::vector<std::vector<int> > members;
  for (int i = 0; i < n; ++i) {
    size_t g = 0;
    while (g < hs.size() && hs[g] != ts[i]->h) ++g;
    if (g == hs.size()) {
      hs.push_back(ts[i]->h);
      members.push_back(std::vector<int>());
    }
    members[g].push_back(i);
  }
  const size_t G = hs.size();
  std::vector<BatchCtx> ctx(G);
  std::vector<std::vector<mpe_tracker*> > gts(G);
  std::vector<std::vector<const uint8_t*> > gimgs(G);
  std::vector<std::vector<double> > gtimes(G);
  for (size_t g = 0; g < G; ++g) {
    for (int i : members[g]) gts[g].push_back(ts[i]);
    gimgs[g].resize(members[g].size());
    gtimes[g].resize(members[g].size());
    const int rc = ctx[g].validate(gts[g].data(), (int)gts[g].size());
    if (rc != MPE_OK) return rc;
  }
  // The groups gs[0..) on the calling thread, pipelined against each other: while the device works on step k of one
  // group, the host collects, advances and packs another.  Groups share nothing (own handle, own trackers, own rows
  // of out / info), so disjoint sets of groups can also run on different host threads.
  auto run_groups_impl = [&](const std::vector<size_t>& gs, long long& updated) -> int {
    std::vector<mpe_result> step_out((size_t)n);
    std::vector<int> step_info((size_t)n * 8);
    auto finish_group = [&](size_t g, int f) -> int {  // complete step f of group g and store its records
      int rc = ctx[g].finish();
      if (rc != MPE_OK) return rc;
      const int m = (int)members[g].size();
      updated += ctx[g].outputs(step_out.data(), 1, step_info.data(), 8, nullptr);
      for (int k = 0; k < m; ++k) {
        const int i = members[g][(size_t)k];
        if (out) out[(size_t)i * n_frames + f] = step_out[(size_t)k];
        if (info) std::memcpy(info + ((size_t)i * n_frames + f) * 8, &step_info[(size_t)k * 8], 8 * sizeof(int));
      }
      return MPE_OK;
    };
    for (int f = 0; f < n_frames; ++f)
      for (size_t g : gs) {
        if (f > 0) {
          const int rc = finish_group(g, f - 1);
          if (rc != MPE_OK) return rc;
        }
        for (size_t k = 0; k < members[g].size(); ++k) {
          gimgs[g][k] = frames[members[g][k]] + (size_t)f * frame_stride_bytes;
          gtimes[g][k] = times[f];
        }
        ctx[g].begin(gimgs[g].data(), rows, cols, stride_bytes, gtimes[g].data());
        const int rc = ctx[g].submit_first();
        if (rc != MPE_OK) return rc;
      }
    for (size_t g : gs) {
      const int rc = finish_group(g, n_frames - 1);
      if (rc != MPE_OK) return rc;
    }
    return MPE_OK;
  };
  // on any error every group of the set is drained: the other groups' submissions would otherwise stay un-collected
  // and their handles would refuse every later mpe_track_step / _submit
  auto run_groups = [&](const std::vector<size_t>& gs, long long& updated) -> int {
    const int rc = run_groups_impl(gs, updated);
    if (rc != MPE_OK)
      for (size_t g : gs) ctx[g].cancel();
    return rc;
  };
  const size_t T = std::min<size_t>((size_t)n_threads, G);
  std::vector<std::vector<size_t> > sets(T);
  for (size_t g = 0; g < G; ++g) sets[g % T].push_back(g);
  std::vector<long long> upd(T, 0);
  std::vector<int> rcs(T, MPE_OK);
  if (T <= 1) {
    rcs[0] = run_groups(sets[0], upd[0]);
  } else {
    std::vector<std::thread> th;
    for (size_t t = 0; t < T; ++t) th.emplace_back([&, t]() { rcs[t] = run_groups(sets[t], upd[t]); });
    for (auto& x : th) x.join();
  }
  long long updated = 0;
  for (size_t t = 0; t < T; ++t) {
    if (rcs[t] != MPE_OK) return rcs[t];
    updated += upd[t];
  }
  return (int)std::min<long long>(updated, 0x7fffffff);
}

int mpe_tracker_run_sequences_batch(mpe_tracker* const* ts, int n, const uint8_t* const* frames, int n_frames, int rows,
                                    int cols, size_t stride_bytes, size_t frame_stride_bytes, const double* times,
                                    mpe_result* out, int* info) {
  return mpe_tracker_run_sequences_batch_threads(ts, n, frames, n_frames, rows, cols, stride_bytes, frame_stride_bytes,
                                                 times, out, info, 1);
}

int mpe_tracker_run_sequence(mpe_tracker* t, const uint8_t* frames, int n_frames, int rows, int cols,
                             size_t stride_bytes, size_t frame_stride_bytes, const double* times, mpe_result* out,
                             int* info) {
  if (!t || !frames || !times || n_frames < 0) return MPE_ERR_ARG;
  int updated = 0;
  for (int f = 0; f < n_frames; ++f) {
    const int rc = mpe_tracker_estimate(t, frames + (size_t)f * frame_stride_bytes, rows, cols, stride_bytes, times[f],
                                        out ? out + f : nullptr, info ? info + 8 * f : nullptr);
    if (rc <= MPE_FRAME_TOO_MANY_DETECTIONS) {  // this frame exceeded a device capacity: recorded, the sequence goes on
      if (out) {
        std::memset(&out[f], 0, sizeof(mpe_result));
        out[f].status = rc;
      }
      if (info) std::memset(info + 8 * f, 0, 8 * sizeof(int));
      continue;
    }
    if (rc < 0) return rc;  // usage / HIP error
    updated += rc;
  }
  return updated;
}

}  // extern "C"
