#pragma once
#include <stdexcept>
#include <string>
#include <opencv2/core.hpp>
#include <sensor_msgs/Image.h>
namespace cv_bridge {
class Exception : public std::runtime_error {
 public:
  explicit Exception(const std::string& s) : std::runtime_error(s) {}
};
class CvImage {
 public:
  std_msgs::Header header;
  std::string encoding;
  cv::Mat image;
  CvImage();
  CvImage(const std_msgs::Header& h, const std::string& enc, const cv::Mat& img = cv::Mat());
  sensor_msgs::ImagePtr toImageMsg() const;
};
typedef boost::shared_ptr<CvImage> CvImagePtr;
typedef boost::shared_ptr<CvImage const> CvImageConstPtr;
CvImageConstPtr toCvShare(const sensor_msgs::Image::ConstPtr& source, const std::string& encoding = std::string());
CvImagePtr toCvCopy(const sensor_msgs::Image::ConstPtr& source, const std::string& encoding = std::string());
}  // namespace cv_bridge
