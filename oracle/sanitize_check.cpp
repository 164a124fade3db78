// sanitize_check.cpp — runs the oracle's whole path under AddressSanitizer + UBSan on a few
// procedurally generated frames (test infrastructure; built and run by tests/test_oracle_sanitize.py).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mpe_oracle.h"

static unsigned rng_state = 12345u;
static unsigned rnd() {
  rng_state = rng_state * 1664525u + 1013904223u;
  return rng_state >> 8;
}

int main() {
  const int rows = 240, cols = 376;
  const double K[9] = {307.8, 0, 181.3, 0, 308.4, 128.3, 0, 0, 1};
  const double D[5] = {-0.3585, 0.1493, 0.00048, -0.0002, 0.0};
  const double M[15] = {0.0714197, 0.0800214, 0.0622611, 0.0400755,  -0.0912328, 0.0317064, -0.0647293, -0.0879977,
                        0.0830852, -0.0558663, -0.0165446, 0.053473, 0.0120,     0.0310,    0.1210};
  orc_params p = {140, 0.6, 10, 200, 0.5, 0.5, 5, 7, 0.75, 0.7, 20, 0};
  int poses = 0;
  orc_tracker* tr = orc_tracker_create(M, 5, K, D, 5, &p);
  for (int f = 0; f < 6; ++f) {
    std::vector<uint8_t> img((size_t)rows * cols);
    for (auto& v : img) v = (uint8_t)(rnd() % 31);
    // five spots from a fixed pose, drifting a little per frame; one spot touches the border in frame 5
    const double tx = 0.02 + 0.003 * f, ty = -0.01, tz = 0.9;
    for (int m = 0; m < 5; ++m) {
      double X = M[3 * m] + tx, Y = M[3 * m + 1] + ty, Z = M[3 * m + 2] + tz;
      double u = K[0] * X / Z + K[2], v = K[4] * Y / Z + K[5];
      if (f == 5 && m == 0) u = 1.0;
      for (int y = (int)v - 6; y <= (int)v + 6; ++y)
        for (int x = (int)u - 6; x <= (int)u + 6; ++x) {
          if (x < 0 || y < 0 || x >= cols || y >= rows) continue;
          double g = 400.0 * std::exp(-((x - u) * (x - u) + (y - v) * (y - v)) / 4.5);
          int val = img[(size_t)y * cols + x] + (int)(g + 0.5);
          img[(size_t)y * cols + x] = (uint8_t)(val > 255 ? 255 : val);
        }
    }
    orc_result r;
    int info[8];
    int rc = orc_tracker_estimate(tr, img.data(), rows, cols, cols, 0.05 * f, &r, info);
    if (rc < 0) return 2;
    poses += rc;
    orc_result r2;
    if (orc_estimate_frame(img.data(), rows, cols, cols, M, 5, K, D, 5, &p, &r2) < 0) return 3;
  }
  orc_tracker_destroy(tr);
  // an all-bright and an empty frame
  std::vector<uint8_t> full((size_t)rows * cols, 255), empty((size_t)rows * cols, 0);
  orc_result r;
  orc_estimate_frame(full.data(), rows, cols, cols, M, 5, K, D, 5, &p, &r);
  orc_estimate_frame(empty.data(), rows, cols, cols, M, 5, K, D, 5, &p, &r);
  std::printf("sanitize_check ok, poses %d\n", poses);
  return poses >= 3 ? 0 : 4;
}
