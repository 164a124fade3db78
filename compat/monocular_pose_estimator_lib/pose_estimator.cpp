// pose_estimator.cpp — facade implementation: parameter marshalling + calls into the C ABI.
#include "facade_namespace.h"
#include "pose_estimator.h"

#include <stdexcept>

MPE_FACADE_BEGIN

static void check(mpe_handle* h, int rc, const char* what) {
  if (rc != MPE_OK) throw std::runtime_error(std::string(what) + ": " + (h ? mpe_last_error(h) : "no handle"));
}

PoseEstimator::PoseEstimator()
    : detection_threshold_value_(0), gaussian_sigma_(0), min_blob_area_(0), max_blob_area_(0),
      max_width_height_distortion_(0), max_circular_distortion_(0), roi_border_thickness_(0), handle_(0),
      tracker_(0), bruteforce_every_frame_(false), current_time_(0), previous_time_(0), predicted_time_(0),
      it_since_initialized_(0), pose_updated_(false) {
  mpe_default_params(&params_);
  params_.back_projection_pixel_tolerance = 3;    // pose_estimator.cpp:36
  params_.nearest_neighbour_pixel_tolerance = 5;  // :37
  params_.certainty_threshold = 0.75;             // :38
  params_.valid_correspondence_threshold = 0.7;   // :39
  params_.histogram_threshold = 0;
  current_pose_ = previous_pose_ = predicted_pose_ = Matrix4d::Identity();
  int rc = mpe_create(&handle_, -1);
  if (rc != MPE_OK) throw std::runtime_error("mpe_create failed: no HIP device (there is no CPU fallback)");
  check(handle_, mpe_tracker_create(handle_, &tracker_), "mpe_tracker_create");
}

PoseEstimator::~PoseEstimator() {
  mpe_tracker_destroy(tracker_);
  mpe_destroy(handle_);
}

void PoseEstimator::syncParams() {
  params_.threshold_value = detection_threshold_value_;
  params_.gaussian_sigma = gaussian_sigma_;
  params_.min_blob_area = min_blob_area_;
  params_.max_blob_area = max_blob_area_;
  params_.max_width_height_distortion = max_width_height_distortion_;
  params_.max_circular_distortion = max_circular_distortion_;
  params_.roi_border_thickness = roi_border_thickness_;
}

void PoseEstimator::setMarkerPositions(const List4DPoints& p) {
  markers_xyz_.resize(3 * p.size());
  for (size_t i = 0; i < p.size(); ++i)
    for (int k = 0; k < 3; ++k) markers_xyz_[3 * i + k] = p[i](k);
  params_.histogram_threshold = 0;  // numCombinations(n, 3) is applied inside the library
  check(handle_, mpe_tracker_set_markers(tracker_, markers_xyz_.data(), (int)p.size()), "mpe_tracker_set_markers");
}

List4DPoints PoseEstimator::getMarkerPositions() {
  List4DPoints out(markers_xyz_.size() / 3);
  for (size_t i = 0; i < out.size(); ++i) {
    for (int k = 0; k < 3; ++k) out[i](k) = markers_xyz_[3 * i + k];
    out[i](3) = 1.0;
  }
  return out;
}

unsigned PoseEstimator::getHistogramThreshold() {
  if (params_.histogram_threshold) return params_.histogram_threshold;
  unsigned n = (unsigned)(markers_xyz_.size() / 3), f = 1, f3 = 6, fn3 = 1;  // 32-bit factorials (combinations.cpp:34-45)
  for (unsigned i = 2; i <= n; ++i) f *= i;
  for (unsigned i = 2; i + 3 <= n; ++i) fn3 *= i;
  const unsigned den = f3 * fn3;
  return den != 0u ? f / den : 0u;
}

void PoseEstimator::pushState() {
  mpe_tracker_state st;
  for (int i = 0; i < 16; ++i) {
    st.current_pose[i] = current_pose_(i);
    st.previous_pose[i] = previous_pose_(i);
    st.predicted_pose[i] = predicted_pose_(i);
  }
  for (int i = 0; i < 36; ++i) st.pose_covariance[i] = pose_covariance_(i);
  st.current_time = current_time_;
  st.previous_time = previous_time_;
  st.predicted_time = predicted_time_;
  st.it_since_initialized = it_since_initialized_;
  st.roi[0] = region_of_interest_.x;
  st.roi[1] = region_of_interest_.y;
  st.roi[2] = region_of_interest_.width;
  st.roi[3] = region_of_interest_.height;
  check(handle_, mpe_tracker_set_state(tracker_, &st), "mpe_tracker_set_state");
}

void PoseEstimator::pullState() {
  mpe_tracker_state st;
  check(handle_, mpe_tracker_get_state(tracker_, &st), "mpe_tracker_get_state");
  for (int i = 0; i < 16; ++i) {
    current_pose_(i) = st.current_pose[i];
    previous_pose_(i) = st.previous_pose[i];
    predicted_pose_(i) = st.predicted_pose[i];
  }
  for (int i = 0; i < 36; ++i) pose_covariance_(i) = st.pose_covariance[i];
  current_time_ = st.current_time;
  previous_time_ = st.previous_time;
  predicted_time_ = st.predicted_time;
  it_since_initialized_ = st.it_since_initialized;
  region_of_interest_ = Rect(st.roi[0], st.roi[1], st.roi[2], st.roi[3]);
}

std::vector<double> PoseEstimator::flatImagePoints() const {
  std::vector<double> det(2 * image_points_.size());
  for (size_t i = 0; i < image_points_.size(); ++i) {
    det[2 * i] = image_points_[i](0);
    det[2 * i + 1] = image_points_[i](1);
  }
  return det;
}

std::vector<uint32_t> PoseEstimator::flatCorrespondences() const {
  std::vector<uint32_t> c(2 * correspondences_.size());
  for (size_t i = 0; i < correspondences_.size(); ++i) {
    c[2 * i] = correspondences_[i][0];
    c[2 * i + 1] = correspondences_[i][1];
  }
  return c;
}

void PoseEstimator::predictPose(double time_to_predict) {
  predicted_time_ = time_to_predict;
  check(handle_,
        mpe_predict_pose(current_pose_.data(), previous_pose_.data(), current_time_, previous_time_, predicted_time_,
                         predicted_pose_.data()),
        "mpe_predict_pose");
  pushState();
}

void PoseEstimator::predictMarkerPositionsInImage() {
  const int n = (int)(markers_xyz_.size() / 3);
  std::vector<double> px(2 * (size_t)n);
  check(handle_, mpe_project_points(predicted_pose_.data(), markers_xyz_.data(), n, camera_matrix_K_.data(), px.data()),
        "mpe_project_points");
  predicted_pixel_positions_.resize(n);
  for (int i = 0; i < n; ++i) {
    predicted_pixel_positions_[i](0) = px[2 * i];
    predicted_pixel_positions_[i](1) = px[2 * i + 1];
  }
}

void PoseEstimator::findCorrespondences() {
  std::vector<double> pred(2 * predicted_pixel_positions_.size());
  for (size_t i = 0; i < predicted_pixel_positions_.size(); ++i) {
    pred[2 * i] = predicted_pixel_positions_[i](0);
    pred[2 * i + 1] = predicted_pixel_positions_[i](1);
  }
  const std::vector<double> det = flatImagePoints();
  std::vector<uint32_t> corr(2 * predicted_pixel_positions_.size() + 2);
  const int nc = mpe_find_correspondences(pred.data(), (int)predicted_pixel_positions_.size(), det.data(),
                                          (int)image_points_.size(), params_.nearest_neighbour_pixel_tolerance,
                                          corr.data());
  if (nc < 0) throw std::runtime_error("mpe_find_correspondences: bad argument");
  correspondences_.clear();
  for (int i = 0; i < nc; ++i) correspondences_.push_back({{corr[2 * i], corr[2 * i + 1]}});
}

unsigned PoseEstimator::checkCorrespondences() {
  syncParams();
  const std::vector<double> det = flatImagePoints();
  const std::vector<uint32_t> corr = flatCorrespondences();
  mpe_result res;
  check(handle_,
        mpe_check_correspondences(handle_, det.data(), (int)image_points_.size(), markers_xyz_.data(),
                                  (int)(markers_xyz_.size() / 3), camera_matrix_K_.data(), &params_, corr.data(),
                                  (int)correspondences_.size(), &res),
        "mpe_check_correspondences");
  if (res.status != MPE_FRAME_POSE) return 0;
  for (int i = 0; i < 16; ++i) predicted_pose_(i) = res.T[i];
  pushState();
  return 1;
}

unsigned PoseEstimator::initialise() {
  syncParams();
  const std::vector<double> det = flatImagePoints();
  const int n_m = (int)(markers_xyz_.size() / 3);
  mpe_result res;
  std::vector<uint32_t> corr(2 * (n_m > 0 ? n_m : 1));
  std::vector<uint32_t> hist(image_points_.size() * (size_t)(n_m > 0 ? n_m : 1) + 1, 0u);
  check(handle_, mpe_initialise(handle_, det.data(), (int)image_points_.size(), markers_xyz_.data(), n_m,
                                camera_matrix_K_.data(), &params_, &res, hist.data(), corr.data()),
        "mpe_initialise");
  if (res.status < 0) throw std::runtime_error("frame exceeded a device capacity");
  bool any_vote = false;
  for (size_t i = 0; i < hist.size(); ++i) any_vote = any_vote || hist[i] != 0;
  if (any_vote) {  // pose_estimator.cpp:704-706: correspondences_ is only rewritten when the histogram has votes
    correspondences_.clear();
    for (int i = 0; i < res.n_corr; ++i) correspondences_.push_back({{corr[2 * i], corr[2 * i + 1]}});
  }
  if (res.status != MPE_FRAME_POSE) return 0;
  for (int i = 0; i < 16; ++i) predicted_pose_(i) = res.T[i];
  pushState();
  return 1;
}

void PoseEstimator::optimisePose() {
  syncParams();
  const std::vector<double> det = flatImagePoints();
  const std::vector<uint32_t> corr = flatCorrespondences();
  mpe_result res;
  check(handle_,
        mpe_optimise_pose(handle_, det.data(), (int)image_points_.size(), markers_xyz_.data(),
                          (int)(markers_xyz_.size() / 3), camera_matrix_K_.data(), &params_, corr.data(),
                          (int)correspondences_.size(), predicted_pose_.data(), &res),
        "mpe_optimise_pose");
  if (res.status != MPE_FRAME_POSE) return;  // fewer than 3 correspondences: nothing to refine
  for (int i = 0; i < 16; ++i) predicted_pose_(i) = res.T[i];
  for (int i = 0; i < 36; ++i) pose_covariance_(i) = res.cov[i];
  pushState();
}

void PoseEstimator::updatePose() {
  previous_pose_ = current_pose_;
  current_pose_ = predicted_pose_;
  previous_time_ = current_time_;
  current_time_ = predicted_time_;
  pushState();
}

void PoseEstimator::optimiseAndUpdatePose(double& /*time_to_predict*/) {
  optimisePose();
  if (it_since_initialized_ < 2) it_since_initialized_++;
  updatePose();
  pose_updated_ = true;
}

void PoseEstimator::predictWithROI(double& time_to_predict, const ImageView& image) {
  if (it_since_initialized_ >= 2)
    predictPose(time_to_predict);
  else
    setPredictedTime(time_to_predict);
  predictMarkerPositionsInImage();
  region_of_interest_ = LEDDetector::determineROI(getPredictedPixelPositions(), Size(image.cols, image.rows),
                                                  (int)roi_border_thickness_, camera_matrix_K_,
                                                  camera_distortion_coeffs_);
  pushState();
}

void PoseEstimator::findCorrespondencesAndPredictPose(double& time_to_predict) {
  findCorrespondences();
  if (checkCorrespondences() == 1) {
    optimiseAndUpdatePose(time_to_predict);
  } else if (initialise() == 1) {
    optimiseAndUpdatePose(time_to_predict);
  }
}

void PoseEstimator::augmentImage(ColorImageView& image) {
  Visualization::createVisualizationImage(image, predicted_pose_, camera_matrix_K_, camera_distortion_coeffs_,
                                          region_of_interest_, distorted_detection_centers_);
}

bool PoseEstimator::estimateBodyPose(const ImageView& image, double time_to_predict) {
  syncParams();
  const double* D = camera_distortion_coeffs_.empty() ? 0 : camera_distortion_coeffs_.data();
  check(handle_, mpe_tracker_set_params(tracker_, &params_), "mpe_tracker_set_params");
  check(handle_, mpe_tracker_set_camera(tracker_, camera_matrix_K_.data(), D, (int)camera_distortion_coeffs_.size()),
        "mpe_tracker_set_camera");
  if (bruteforce_every_frame_) mpe_tracker_reset(tracker_);
  predicted_time_ = time_to_predict;
  mpe_result r;
  const int rc = mpe_tracker_estimate(tracker_, image.data, image.rows, image.cols, image.step, time_to_predict, &r, 0);
  if (rc < 0) throw std::runtime_error(std::string("mpe_tracker_estimate: ") + mpe_last_error(handle_));
  pose_updated_ = rc == 1;
  double xy[2 * MPE_MAX_DETECTIONS];
  const int nd = mpe_tracker_get_image_points(tracker_, xy, MPE_MAX_DETECTIONS);
  image_points_.resize(nd > 0 ? nd : 0);
  for (int i = 0; i < nd; ++i) {
    image_points_[i](0) = xy[2 * i];
    image_points_[i](1) = xy[2 * i + 1];
  }
  uint32_t corr[2 * MPE_MAX_MARKERS];
  const int nc = mpe_tracker_get_correspondences(tracker_, corr, MPE_MAX_MARKERS);
  correspondences_.clear();
  for (int i = 0; i < nc; ++i) correspondences_.push_back({{corr[2 * i], corr[2 * i + 1]}});
  float dxy[2 * MPE_MAX_DETECTIONS];
  const int ndist = mpe_tracker_get_distorted_centers(tracker_, dxy, MPE_MAX_DETECTIONS);
  distorted_detection_centers_.resize(ndist > 0 ? ndist : 0);
  for (int i = 0; i < ndist; ++i) {
    distorted_detection_centers_[i].x = dxy[2 * i];
    distorted_detection_centers_[i].y = dxy[2 * i + 1];
  }
  pullState();  // poses, times, it_since_initialized_, region_of_interest_ of the library-side state machine
  predicted_pixel_positions_.clear();
  return pose_updated_;
}

void PoseEstimator::estimateBodyPoseBatch(const uint8_t* frames, int n_frames, int rows, int cols,
                                          bool frames_on_device, mpe_result* results) {
  syncParams();
  const double* D = camera_distortion_coeffs_.empty() ? 0 : camera_distortion_coeffs_.data();
  check(handle_, mpe_estimate_batch(handle_, frames, n_frames, rows, cols, (size_t)cols, (size_t)rows * cols,
                                    frames_on_device ? 1 : 0, markers_xyz_.data(), (int)(markers_xyz_.size() / 3),
                                    camera_matrix_K_.data(), D, (int)camera_distortion_coeffs_.size(), &params_,
                                    results),
        "mpe_estimate_batch");
}

void PoseEstimator::decodeToMono8(const void* data, int mpe_encoding, bool is_bigendian, int rows, int cols, size_t step,
                                  uint8_t* mono8_out) {
  check(handle_, mpe_convert_to_mono8(handle_, data, 0, mpe_encoding, is_bigendian ? 1 : 0, 1, rows, cols, step,
                                      step * (size_t)rows, mono8_out, 0),
        "mpe_convert_to_mono8");
}

MPE_FACADE_END  // namespace monocular_pose_estimator
