#pragma once
#include <string>
#include <ros/ros.h>
#include <sensor_msgs/Image.h>
namespace image_transport {
class Publisher {
 public:
  unsigned getNumSubscribers() const;
  void publish(const sensor_msgs::ImagePtr& msg) const;
};
class ImageTransport {
 public:
  explicit ImageTransport(const ros::NodeHandle& nh);
  Publisher advertise(const std::string& topic, unsigned queue);
};
}  // namespace image_transport
