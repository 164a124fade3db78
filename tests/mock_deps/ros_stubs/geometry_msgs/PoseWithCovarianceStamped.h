#pragma once
#include <array>
#include "std_msgs/Header.h"
namespace geometry_msgs {
struct Point { double x, y, z; };
struct Quaternion { double x, y, z, w; };
struct Pose { Point position; Quaternion orientation; };
struct PoseWithCovariance { Pose pose; std::array<double, 36> covariance; };
struct PoseWithCovarianceStamped { std_msgs::Header header; PoseWithCovariance pose; };
}  // namespace geometry_msgs
