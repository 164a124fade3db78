// visualization.cpp — overlay rasteriser (see visualization.h).
#include "facade_namespace.h"
#include "visualization.h"

#include <cmath>
#include <cstdlib>
#include <stdexcept>

#include "mpe.h"

MPE_FACADE_BEGIN

namespace {

struct Bgr {
  uint8_t b, g, r;
};
const Bgr kRed = {0, 0, 255}, kGreen = {0, 255, 0}, kBlue = {255, 0, 0};

// 2x2 brush with its top-left corner at (x, y), clipped to the image
void stamp(ColorImageView& im, int x, int y, Bgr c) {
  for (int dy = 0; dy < 2; ++dy)
    for (int dx = 0; dx < 2; ++dx) {
      const int px = x + dx, py = y + dy;
      if (px < 0 || py < 0 || px >= im.cols || py >= im.rows) continue;
      uint8_t* p = im.data + (size_t)py * im.step + 3 * (size_t)px;
      p[0] = c.b;
      p[1] = c.g;
      p[2] = c.r;
    }
}

void segment(ColorImageView& im, int x0, int y0, int x1, int y1, Bgr c) {
  // clamp far-away end points so that a wild projection cannot loop for ages
  const int lim = 1 << 15;
  if (std::abs(x0) > lim || std::abs(y0) > lim || std::abs(x1) > lim || std::abs(y1) > lim) return;
  const int dx = std::abs(x1 - x0), dy = -std::abs(y1 - y0);
  const int sx = x0 < x1 ? 1 : -1, sy = y0 < y1 ? 1 : -1;
  int err = dx + dy;
  for (;;) {
    stamp(im, x0, y0, c);
    if (x0 == x1 && y0 == y1) break;
    const int e2 = 2 * err;
    if (e2 >= dy) {
      err += dy;
      x0 += sx;
    }
    if (e2 <= dx) {
      err += dx;
      y0 += sy;
    }
  }
}

void ring(ColorImageView& im, int cx, int cy, int radius, Bgr c) {
  int x = radius, y = 0, err = 1 - radius;
  while (x >= y) {
    const int ox[8] = {x, y, -y, -x, -x, -y, y, x}, oy[8] = {y, x, x, y, -y, -x, -x, -y};
    for (int k = 0; k < 8; ++k) stamp(im, cx + ox[k], cy + oy[k], c);
    ++y;
    if (err < 0) {
      err += 2 * y + 1;
    } else {
      --x;
      err += 2 * (y - x) + 1;
    }
  }
}

int round_px(float v) { return (int)std::lrint((double)v); }

}  // namespace

void Visualization::projectOrientationVectorsOnImage(ColorImageView& image, const std::vector<Point3f>& pts,
                                                     const Matrix3d& K, const std::vector<double>& D) {
  if (pts.size() < 4) throw std::runtime_error("projectOrientationVectorsOnImage needs 4 points");
  // cv::projectPoints with zero rvec / tvec: pinhole pixel, then the plumb-bob model (= distortPoints)
  float ideal[8], dist[8];
  for (int i = 0; i < 4; ++i) {
    ideal[2 * i] = (float)(K(0, 0) * ((double)pts[i].x / (double)pts[i].z) + K(0, 2));
    ideal[2 * i + 1] = (float)(K(1, 1) * ((double)pts[i].y / (double)pts[i].z) + K(1, 2));
  }
  if (mpe_distort_points(ideal, dist, 4, K.data(), D.empty() ? 0 : D.data(), (int)D.size()) != MPE_OK)
    throw std::runtime_error("mpe_distort_points: bad argument");
  const Bgr colour[3] = {kRed, kGreen, kBlue};
  for (int a = 0; a < 3; ++a)
    segment(image, round_px(dist[0]), round_px(dist[1]), round_px(dist[2 * (a + 1)]), round_px(dist[2 * (a + 1) + 1]),
            colour[a]);
}

void Visualization::createVisualizationImage(ColorImageView& image, const Matrix4d& T, const Matrix3d& K,
                                             const std::vector<double>& D, Rect roi,
                                             const std::vector<Point2f>& centres) {
  const double len = 0.075;  // visualization.cpp:64
  const double tip[4][3] = {{0, 0, 0}, {len, 0, 0}, {0, len, 0}, {0, 0, len}};
  std::vector<Point3f> pts(4);
  for (int i = 0; i < 4; ++i) {
    pts[i].x = (float)(T(0, 0) * tip[i][0] + T(0, 1) * tip[i][1] + T(0, 2) * tip[i][2] + T(0, 3));
    pts[i].y = (float)(T(1, 0) * tip[i][0] + T(1, 1) * tip[i][1] + T(1, 2) * tip[i][2] + T(1, 3));
    pts[i].z = (float)(T(2, 0) * tip[i][0] + T(2, 1) * tip[i][1] + T(2, 2) * tip[i][2] + T(2, 3));
  }
  projectOrientationVectorsOnImage(image, pts, K, D);
  for (size_t i = 0; i < centres.size(); ++i) ring(image, round_px(centres[i].x), round_px(centres[i].y), 10, kRed);
  const int x0 = roi.x, y0 = roi.y, x1 = roi.x + roi.width - 1, y1 = roi.y + roi.height - 1;
  if (roi.width > 0 && roi.height > 0) {
    segment(image, x0, y0, x1, y0, kBlue);
    segment(image, x1, y0, x1, y1, kBlue);
    segment(image, x1, y1, x0, y1, kBlue);
    segment(image, x0, y1, x0, y0, kBlue);
  }
}

void Visualization::grayToColor(const ImageView& gray, ColorImageView& color) {
  if (gray.rows != color.rows || gray.cols != color.cols) throw std::runtime_error("grayToColor: size mismatch");
  for (int y = 0; y < gray.rows; ++y) {
    const uint8_t* s = gray.data + (size_t)y * gray.step;
    uint8_t* d = color.data + (size_t)y * color.step;
    for (int x = 0; x < gray.cols; ++x) d[3 * x] = d[3 * x + 1] = d[3 * x + 2] = s[x];
  }
}

MPE_FACADE_END  // namespace monocular_pose_estimator
