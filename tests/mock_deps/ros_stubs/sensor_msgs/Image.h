#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <boost_shared_ptr_stub.h>
#include "std_msgs/Header.h"
namespace sensor_msgs {
struct Image {
  std_msgs::Header header;
  uint32_t height, width;
  std::string encoding;
  uint8_t is_bigendian;
  uint32_t step;
  std::vector<uint8_t> data;
  typedef boost::shared_ptr<Image const> ConstPtr;
  typedef boost::shared_ptr<Image> Ptr;
};
typedef boost::shared_ptr<Image> ImagePtr;
}  // namespace sensor_msgs
