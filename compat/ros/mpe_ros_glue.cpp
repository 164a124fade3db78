// mpe_ros_glue.cpp — ROS 1 node AND nodelet for the MI355X back-end, preserving the ROS surface of the
// reference package `monocular_pose_estimator`:
//   executable  monocular_pose_estimator            (built without MPE_BUILD_NODELET)
//   nodelet     monocular_pose_estimator/MPENodelet (built with    MPE_BUILD_NODELET)
//   subscribes  /camera/image_raw (sensor_msgs/Image, converted to MONO8), /camera/camera_info; queue 1
//   publishes   estimated_pose (geometry_msgs/PoseWithCovarianceStamped), image_with_detections (bgr8)
//   parameters  ~marker_positions (list of {x, y, z}), the 11 dynamic-reconfigure values of
//               cfg/MonocularPoseEstimator.cfg
//
// NOT BUILT IN THIS REPOSITORY'S ENVIRONMENT: ROS (roscpp, cv_bridge, image_transport, nodelet,
// dynamic_reconfigure) is not installed here, so this file is source only — but it does see a compiler: the CPU test
// tier runs `g++ -fsyntax-only` over it, as node and as nodelet, against declaration-only stand-ins for the ROS headers
// (tests/mock_deps/ros_stubs, tests/test_abi_cpu.py::test_ros_glue_syntax).  Everything it does beyond
// calling ROS — message field packing, calibration and parameter transfer — lives in
// message_conversions.h and IS built and tested (tests/test_abi_cpu.py::test_ros_message_conversions);
// the per-frame logic is the PoseEstimator facade (compat/monocular_pose_estimator_lib), tested on the GPU.
// Build files: CMakeLists.txt / package.xml next to this file.
#include <cv_bridge/cv_bridge.h>
#include <dynamic_reconfigure/server.h>
#include <geometry_msgs/PoseWithCovarianceStamped.h>
#include <image_transport/image_transport.h>
#include <ros/ros.h>
#include <sensor_msgs/CameraInfo.h>
#include <sensor_msgs/Image.h>
#include <sensor_msgs/image_encodings.h>

#include <memory>
#include <vector>

#include <monocular_pose_estimator/MonocularPoseEstimatorConfig.h>

#include "message_conversions.h"
#include "monocular_pose_estimator_lib/pose_estimator.h"

namespace monocular_pose_estimator {

class MPENode {
 public:
  MPENode(const ros::NodeHandle& nh = ros::NodeHandle(), const ros::NodeHandle& nh_private = ros::NodeHandle("~"))
      : nh_(nh), nh_private_(nh_private), transport_(nh_), calibrated_(false) {
    // the reference's order (monocular_pose_estimator.cpp:39-86): the dynamic-reconfigure server first ("before
    // reading parameter server values": its first callback installs the cfg defaults / launch-file values), then the
    // subscribers, the publishers, and the marker positions last
    reconfigure_.setCallback([this](MonocularPoseEstimatorConfig& cfg, uint32_t) { onReconfigure(cfg); });
    image_sub_ = nh_.subscribe("/camera/image_raw", 1, &MPENode::onImage, this);
    info_sub_ = nh_.subscribe("/camera/camera_info", 1, &MPENode::onCameraInfo, this);
    pose_pub_ = nh_.advertise<geometry_msgs::PoseWithCovarianceStamped>("estimated_pose", 1);
    overlay_pub_ = transport_.advertise("image_with_detections", 1);
    loadMarkers();
  }

 private:
  void loadMarkers() {
    XmlRpc::XmlRpcValue list;
    if (!nh_private_.getParam("marker_positions", list) || list.getType() != XmlRpc::XmlRpcValue::TypeArray) {
      ROS_ERROR("%s: parameter 'marker_positions' missing or malformed (load the marker YAML in the launch file)",
                ros::this_node::getName().c_str());
      ros::shutdown();
      return;
    }
    List4DPoints markers(list.size());
    for (int i = 0; i < list.size(); ++i) {
      markers[i](0) = static_cast<double>(list[i]["x"]);
      markers[i](1) = static_cast<double>(list[i]["y"]);
      markers[i](2) = static_cast<double>(list[i]["z"]);
      markers[i](3) = 1.0;
    }
    estimator_.setMarkerPositions(markers);
    ROS_INFO("%d markers on the object", (int)markers.size());
  }

  void onCameraInfo(const sensor_msgs::CameraInfo::ConstPtr& msg) {
    if (calibrated_) return;  // the first message wins, like the reference
    applyCameraInfo(estimator_, msg->K.data(), msg->D);
    calibrated_ = true;
    ROS_INFO("camera calibration received");
  }

  void onReconfigure(MonocularPoseEstimatorConfig& cfg) {
    ReconfigureValues v;
    v.threshold_value = cfg.threshold_value;
    v.gaussian_sigma = cfg.gaussian_sigma;
    v.min_blob_area = cfg.min_blob_area;
    v.max_blob_area = cfg.max_blob_area;
    v.max_width_height_distortion = cfg.max_width_height_distortion;
    v.max_circular_distortion = cfg.max_circular_distortion;
    v.back_projection_pixel_tolerance = cfg.back_projection_pixel_tolerance;
    v.nearest_neighbour_pixel_tolerance = cfg.nearest_neighbour_pixel_tolerance;
    v.certainty_threshold = cfg.certainty_threshold;
    v.valid_correspondence_threshold = cfg.valid_correspondence_threshold;
    v.roi_border_thickness = cfg.roi_border_thickness;
    applyReconfigure(estimator_, v);
    ROS_INFO("parameters changed");
  }

  void onImage(const sensor_msgs::Image::ConstPtr& msg) {
    if (!calibrated_) {
      ROS_WARN("no camera info yet");
      return;
    }
    // monocular_pose_estimator.cpp:147 converts every frame to MONO8 through cv_bridge.  mono8 is used in place and
    // mono16 is decoded by the back-end (mpe_convert_to_mono8: convertTo's single-precision rule, the same in every
    // OpenCV).  The colour encodings go through cv_bridge like everything else (Bayer patterns, ...) UNLESS the build
    // says that the linked OpenCV converts 8-bit RGB to gray with the 14-bit weights (OpenCV <= 3.4.1: 1868 / 9617 /
    // 4899, >> 14 — the rule mpe_convert_to_mono8 restates, for the OpenCV 3.3 of the reference's platform); OpenCV
    // >= 3.4.2 uses 15-bit weights and can differ by one gray level, which near threshold_value flips LED pixels.
    cv_bridge::CvImageConstPtr mono;
    int enc = -1;
#ifdef MPE_OPENCV_GRAY_14BIT
    const FramePath path = framePathForEncoding(msg->encoding, true, &enc);
#else
    const FramePath path = framePathForEncoding(msg->encoding, false, &enc);
#endif
    const uint8_t* pixels = nullptr;
    size_t step = 0;
    if (path == FRAME_IN_PLACE) {
      pixels = msg->data.data();
      step = msg->step;
    } else if (path == FRAME_BACKEND_DECODE) {
      decoded_.resize((size_t)msg->height * msg->width);
      try {
        estimator_.decodeToMono8(msg->data.data(), enc, msg->is_bigendian != 0, (int)msg->height, (int)msg->width, msg->step,
                                 decoded_.data());
      } catch (const std::exception& e) {
        ROS_ERROR("decodeToMono8: %s", e.what());
        return;
      }
      pixels = decoded_.data();
      step = msg->width;
    } else {
      try {
        mono = cv_bridge::toCvShare(msg, sensor_msgs::image_encodings::MONO8);  // the back-end never writes the frame
      } catch (const cv_bridge::Exception& e) {
        ROS_ERROR("cv_bridge: %s", e.what());
        return;
      }
      pixels = mono->image.data;
      step = mono->image.step;
    }
    const ImageView view(pixels, (int)msg->height, (int)msg->width, step);
    bool found = false;
    try {
      found = estimator_.estimateBodyPose(view, msg->header.stamp.toSec());
    } catch (const std::exception& e) {  // HIP / capacity errors surface here instead of cv::Exception
      ROS_ERROR("estimateBodyPose: %s", e.what());
      return;
    }
    if (found) {
      const PoseMessageFields f = poseToMessageFields(estimator_.getPredictedPose(), estimator_.getPoseCovariance());
      geometry_msgs::PoseWithCovarianceStamped out;
      out.header.stamp = msg->header.stamp;
      out.pose.pose.position.x = f.position[0];
      out.pose.pose.position.y = f.position[1];
      out.pose.pose.position.z = f.position[2];
      out.pose.pose.orientation.x = f.orientation[0];
      out.pose.pose.orientation.y = f.orientation[1];
      out.pose.pose.orientation.z = f.orientation[2];
      out.pose.pose.orientation.w = f.orientation[3];
      for (int i = 0; i < 36; ++i) out.pose.covariance[i] = f.covariance[i];
      pose_pub_.publish(out);
    } else {
      ROS_WARN("unable to resolve a pose");
    }
    if (overlay_pub_.getNumSubscribers() > 0) {
      cv_bridge::CvImage overlay(msg->header, sensor_msgs::image_encodings::BGR8,
                                 cv::Mat((int)msg->height, (int)msg->width, CV_8UC3));
      ColorImageView colour(overlay.image.data, (int)msg->height, (int)msg->width, overlay.image.step);
      Visualization::grayToColor(view, colour);
      if (found) estimator_.augmentImage(colour);
      overlay_pub_.publish(overlay.toImageMsg());
    }
  }

  ros::NodeHandle nh_, nh_private_;
  image_transport::ImageTransport transport_;
  image_transport::Publisher overlay_pub_;
  ros::Publisher pose_pub_;
  ros::Subscriber image_sub_, info_sub_;
  dynamic_reconfigure::Server<MonocularPoseEstimatorConfig> reconfigure_;
  PoseEstimator estimator_;
  std::vector<uint8_t> decoded_;  // mono8 frame decoded from a colour / 16-bit message
  bool calibrated_;
};

}  // namespace monocular_pose_estimator

#ifdef MPE_BUILD_NODELET
#include <nodelet/nodelet.h>
#include <pluginlib/class_list_macros.h>

namespace monocular_pose_estimator {
class MPENodelet : public nodelet::Nodelet {
 public:
  void onInit() override {
    node_.reset(new MPENode(getNodeHandle(), getPrivateNodeHandle()));
    NODELET_INFO_STREAM("initialised nodelet " << getName());
  }

 private:
  std::unique_ptr<MPENode> node_;
};
}  // namespace monocular_pose_estimator
PLUGINLIB_EXPORT_CLASS(monocular_pose_estimator::MPENodelet, nodelet::Nodelet)
#else
int main(int argc, char** argv) {
  ros::init(argc, argv, "monocular_pose_tracker");
  monocular_pose_estimator::MPENode node;
  ros::spin();  // single-threaded spinner: one stateful estimator, frames dropped (queue 1), never queued
  return 0;
}
#endif
