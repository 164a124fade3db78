// tests/mock_deps/opencv2/core.hpp — NOT OpenCV.  The few members of cv::Mat / cv::Rect that compat/adapters use
// (rows, cols, step, data, type(), empty(), at<T>(), the (rows, cols, type) constructor), with real storage, so that
// the adapter code can be compiled and exercised here.  Test scaffolding for this repository's own adapters only
// (see tests/mock_deps/Eigen/Dense).
#ifndef MPE_TESTS_MOCK_OPENCV_CORE_
#define MPE_TESTS_MOCK_OPENCV_CORE_
#include <cstddef>
#include <memory>
#include <vector>
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_64F 6
namespace cv {
class Mat {
 public:
  int rows, cols;
  unsigned char* data;
  size_t step;
  Mat() : rows(0), cols(0), data(0), step(0), type_(0) {}
  Mat(int r, int c, int type) : rows(r), cols(c), type_(type) {
    const size_t es = type == CV_64F ? 8 : (type == CV_8UC3 ? 3 : 1);
    step = es * (size_t)c;
    buf_ = std::make_shared<std::vector<unsigned char> >(step * (size_t)r, (unsigned char)0);
    data = buf_->data();
  }
  int type() const { return type_; }
  bool empty() const { return data == 0 || rows == 0 || cols == 0; }
  template <class T>
  T& at(int r, int c) { return *reinterpret_cast<T*>(data + (size_t)r * step + (size_t)c * sizeof(T)); }
  template <class T>
  const T& at(int r, int c) const { return *reinterpret_cast<const T*>(data + (size_t)r * step + (size_t)c * sizeof(T)); }

 private:
  int type_;
  std::shared_ptr<std::vector<unsigned char> > buf_;  // header copies share the pixels, like cv::Mat
};
struct Rect {
  int x, y, width, height;
  Rect() : x(0), y(0), width(0), height(0) {}
  Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {}
};
}  // namespace cv
#endif
