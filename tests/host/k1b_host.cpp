// Host build of the DEVICE blob-extraction source (test scaffolding, CPU tier).
//
// tests/test_k1b_host.py cuts the region `struct BlobRec { ... }` .. `// final stage: kept blobs` out of
// rpg_monocular_pose_estimator_amd/csrc/mpe_kernels.hip (threshold helpers, fixed-point blur, Suzuki border
// following, polygon sums, shape filter, undistortion — everything of K1b that is not wave plumbing) and
// `struct DetectParams` out of mpe_internal.h into k1b_extract.inc, and compiles this file with g++.  The shims
// below give the HIP spellings a one-lane meaning.  The flow of host_find_leds is that of the device's
// whole-frame tier (k1b_general): threshold -> blurred-mask bitmap -> raster scan with border following ->
// kept blobs in descending raster order of their start pixel -> undistortion.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
using std::max;
using std::min;
#define __device__
#define __forceinline__ inline
#define __noinline__
#define __restrict__
#define MPE_MAX_KSIZE 9
typedef unsigned long long u64;
struct uint4 {
  unsigned x, y, z, w;
};
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline void atomicOr(u64* p, u64 v) { *p |= v; }
static inline void atomicOr(unsigned* p, unsigned v) { *p |= v; }
static inline int __mul24(int a, int b) { return a * b; }
static inline u64 __builtin_amdgcn_ballot_w64(bool b) { return b ? 1 : 0; }
static inline unsigned __builtin_bitreverse32(unsigned v) {
  unsigned r = 0;
  for (int i = 0; i < 32; ++i)
    if ((v >> i) & 1) r |= 1u << (31 - i);
  return r;
}
static inline unsigned __builtin_amdgcn_alignbyte(unsigned hi, unsigned lo, unsigned sh) {
  return (unsigned)(((((u64)hi) << 32) | lo) >> (8 * (sh & 3)));
}
static inline unsigned __builtin_amdgcn_udot4(unsigned a, unsigned b, unsigned c, bool) {
  for (int i = 0; i < 4; ++i) c += ((a >> (8 * i)) & 0xFFu) * ((b >> (8 * i)) & 0xFFu);
  return c;
}
#define __host__
#include "k1a_extract.inc"  // ThrTest, make_thr_test, gt_word, maybe_gt16, any_gt16 (`struct ThrTest {` .. `#ifndef K1A_UNROLL`)
#include "k1b_extract.inc"

extern "C" int host_find_leds(const uint8_t* img, int rows, int cols, int thr, const int* taps, int ksize,
                              const double* shape /* min_area max_area max_wh max_circ */, const double* K,
                              const double* D, int nD, int roi_x, int roi_y, float* dist_xy, double* undist_xy,
                              int cap) {
  DetectParams dp;
  std::memset(&dp, 0, sizeof(dp));
  dp.thr = thr;
  dp.ksize = ksize;
  for (int i = 0; i < ksize; ++i) dp.taps[i] = taps[i];
  pack_taps(dp);
  dp.min_area = shape[0];
  dp.max_area = shape[1];
  dp.max_wh = shape[2];
  dp.max_circ = shape[3];
  for (int i = 0; i < 9; ++i) dp.K[i] = K[i];
  dp.ifx = 1. / K[0];
  dp.ify = 1. / K[4];
  for (int i = 0; i < 8; ++i) dp.k[i] = i < nD ? D[i] : 0.0;
  dp.undist_iters = nD > 0 ? 5 : 0;
  dp.roi_x = roi_x;
  dp.roi_y = roi_y;
  // thresholded pixels, rows of 16-byte segments (THRESH_TOZERO four bytes at a time as the device does)
  const int nseg = (cols + 15) / 16, PW = 16 * nseg;
  std::vector<uint8_t> pix((size_t)rows * PW, 0);
  for (int y = 0; y < rows; ++y) std::memcpy(&pix[(size_t)y * PW], img + (size_t)y * cols, cols);
  const unsigned add = (unsigned)(255 - thr) * 0x00010001u;
  for (size_t i = 0; i < pix.size(); i += 4) {
    unsigned w;
    std::memcpy(&w, &pix[i], 4);
    w = tozero4(w, add);
    std::memcpy(&pix[i], &w, 4);
  }
  const int wb = (cols + 2 + 63) / 64;
  std::vector<u64> nz((size_t)(rows + 2) * wb + 1, 0), pm(nz.size(), 0), ng(nz.size(), 0);
  const PixWin pw = {pix.data(), 0, rows, 0, PW};
  for (int y = 0; y < rows; ++y)
    for (int c = 0; c < nseg; ++c) blur_to_bitmap(pw, rows, cols, dp, dp.taps, y, c, nz.data() + (size_t)(y + 1) * wb, 0);
  struct Kept {
    float x, y;
    unsigned key;
  };
  std::vector<Kept> kept;
  int over = 0;
  scan_window(nz.data(), pm.data(), ng.data(), wb, rows, 0, 0, dp, roi_x, roi_y, &over,
              [&](float mcx, float mcy, unsigned key) { kept.push_back({mcx, mcy, key}); });
  if (over) return -1;
  std::sort(kept.begin(), kept.end(), [](const Kept& a, const Kept& b) { return a.key > b.key; });
  const int n = (int)kept.size();
  for (int i = 0; i < n && i < cap; ++i) {
    float ux, uy;
    undistort_point(kept[i].x, kept[i].y, dp, ux, uy);
    dist_xy[2 * i] = kept[i].x;
    dist_xy[2 * i + 1] = kept[i].y;
    undist_xy[2 * i] = (double)ux;
    undist_xy[2 * i + 1] = (double)uy;
  }
  return n;
}

// The image scan's per-segment tests (k1a_scan and the voting kernel's scan rider): bit 0 = any_gt16, bit 1 = maybe_gt16
extern "C" int host_scan_tests(const uint8_t* seg16, int thr) {
  uint4 v;
  std::memcpy(&v, seg16, 16);
  const ThrTest q = make_thr_test(thr);
  // bits 2 / 3: the same through the compile-time forms the kernels branch to (k1a_scan, the rider's first level)
  const unsigned a_c = q.sel ? any_gt16_c<true>(v, q.kk) : any_gt16_c<false>(v, q.kk);
  const unsigned m_c = q.sel ? maybe_gt16_c<true>(v, q.kk) : maybe_gt16_c<false>(v, q.kk);
  return (any_gt16(v, q) ? 1 : 0) | (maybe_gt16(v, q) ? 2 : 0) | (a_c ? 4 : 0) | (m_c ? 8 : 0);
}
