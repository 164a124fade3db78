#pragma once
#include <array>
#include <vector>
#include <boost_shared_ptr_stub.h>
#include "std_msgs/Header.h"
namespace sensor_msgs {
struct CameraInfo {
  std_msgs::Header header;
  std::vector<double> D;
  std::array<double, 9> K;
  typedef boost::shared_ptr<CameraInfo const> ConstPtr;
};
}  // namespace sensor_msgs
