#pragma once
// what dynamic_reconfigure generates from cfg/MonocularPoseEstimator.cfg: one member per parameter
namespace monocular_pose_estimator {
struct MonocularPoseEstimatorConfig {
  int threshold_value;
  double gaussian_sigma, min_blob_area, max_blob_area, max_width_height_distortion, max_circular_distortion;
  double back_projection_pixel_tolerance, nearest_neighbour_pixel_tolerance, certainty_threshold,
      valid_correspondence_threshold;
  int roi_border_thickness;
};
}  // namespace monocular_pose_estimator
