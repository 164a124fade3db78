#pragma once
#include <sstream>
#include <string>
#include <ros/ros.h>
namespace nodelet {
class Nodelet {
 public:
  virtual ~Nodelet();
  virtual void onInit() = 0;
 protected:
  ros::NodeHandle& getNodeHandle() const;
  ros::NodeHandle& getPrivateNodeHandle() const;
  const std::string& getName() const;
};
}  // namespace nodelet
#define NODELET_INFO_STREAM(args) do { std::ostringstream nodelet_ss__; nodelet_ss__ << args; } while (0)
