//@file-prologue
// mpe_k2_head.h — index arithmetic and bearings shared by the voting kernels (mpe_k2.hip) and the tail (mpe_k3.hip)
#ifndef MPE_K2_HEAD_H_
#define MPE_K2_HEAD_H_
#include "mpe_kernels_common.h"
namespace mpe {
//@file-prologue-end
// =============================================================================================
// K2 — brute-force correspondence voting (pose_estimator.cpp:544-702)
// =============================================================================================
// lexicographic unranking of the idx-th 3-combination of {0..n-1}
__device__ __forceinline__ void unrank_combo3(int idx, int n, int& a, int& b, int& c) {
  a = 0;
  for (;;) {
    const int cnt = (n - 1 - a) * (n - 2 - a) / 2;  // combos starting with a
    if (idx < cnt) break;
    idx -= cnt;
    ++a;
  }
  b = a + 1;
  for (;;) {
    const int cnt = n - 1 - b;
    if (idx < cnt) break;
    idx -= cnt;
    ++b;
  }
  c = b + 1 + idx;
}

__device__ __forceinline__ V3 bearing(double u, double v, double fx, double fy, double cx, double cy) {
  V3 s = {(u - cx) / fx, (v - cy) / fy, 1.0};  // pose_estimator.cpp:288-301
  return vdiv(s, norm(s));
}

__device__ __forceinline__ double pick_root(const P3PCtx& c, int k) {
  return k == 0 ? c.root[0] : (k == 1 ? c.root[1] : (k == 2 ? c.root[2] : c.root[3]));
}
//@file-epilogue
}  // namespace mpe
#endif
//@file-epilogue-end
