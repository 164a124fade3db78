// Host build of the DEVICE geometry source (test scaffolding, CPU tier): rpg_monocular_pose_estimator_amd/csrc/mpe_p3p.h
// as it is, and the tail helpers of mpe_k3.hip (`struct T34` .. `#define K3_GROUP`: projection, Hestenes-Jacobi
// Kabsch rotation, LDL^T of the normal equations, exponential-map update) cut out at test time into k3_extract.inc.
#include <cstddef>
#include <cstring>
#include "mpe_p3p.h"
namespace mpe {
#include "k3_extract.inc"
#include "k2_extract.inc"  // unrank_combo3, bearing, pick_root, perm_from_index
}
using namespace mpe;

extern "C" void host_quartic(const double* factors, int n, int variant, double* roots) {
  for (int i = 0; i < n; ++i) {
    const double* a = factors + (size_t)i * 5;
    double r[4];
    if (variant == 1)
      solve_quartic_lit2(a[0], a[1], a[2], a[3], a[4], r);
    else if (variant == 2)  // libstdc++'s pow(complex, double) restated (mpe_ddmath.h; "vote_arith" = 3)
      solve_quartic<true>(a[0], a[1], a[2], a[3], a[4], r);
    else
      solve_quartic(a[0], a[1], a[2], a[3], a[4], r);
    for (int k = 0; k < 4; ++k) roots[(size_t)i * 4 + k] = r[k];
  }
}

// as k_p3p_batch of mpe_k3.hip
extern "C" void host_p3p(const double* fv, const double* wp, int n, double* sol, int* status) {
  for (int i = 0; i < n; ++i) {
    const double* f = fv + (size_t)i * 9;
    const double* w = wp + (size_t)i * 9;
    const V3 f0 = {f[0], f[1], f[2]}, f1 = {f[3], f[4], f[5]}, f2 = {f[6], f[7], f[8]};
    const V3 w0 = {w[0], w[1], w[2]}, w1 = {w[3], w[4], w[5]}, w2 = {w[6], w[7], w[8]};
    P3PCtx c;
    if (!p3p_prepare(f0, f1, f2, w0, w1, w2, c)) {
      status[i] = -1;
      continue;
    }
    double* o = sol + (size_t)i * 48;
    for (int k = 0; k < 4; ++k) {
      M3 R;
      V3 C;
      p3p_solution(c, c.root[k], R, C);
      double* q = o + 12 * k;
      q[0] = R.r0.x; q[1] = R.r0.y; q[2] = R.r0.z; q[3] = C.x;
      q[4] = R.r1.x; q[5] = R.r1.y; q[6] = R.r1.z; q[7] = C.y;
      q[8] = R.r2.x; q[9] = R.r2.y; q[10] = R.r2.z; q[11] = C.z;
    }
    status[i] = 0;
  }
}

extern "C" void host_kabsch(const double* H, double* R) {
  double h[3][3], r[3][3];
  std::memcpy(h, H, sizeof(h));
  kabsch_rotation(h, r);
  std::memcpy(R, r, sizeof(r));
}

extern "C" void host_apply_exp(const double* twist, double* T34_rows) {
  T34 T;
  std::memcpy(T.m, T34_rows, sizeof(T.m));
  apply_exp(twist, T);
  std::memcpy(T34_rows, T.m, sizeof(T.m));
}

extern "C" void host_ldl_solve(const double* A, const double* b, double* x) {
  double a[6][6];
  std::memcpy(a, A, sizeof(a));
  LDL6 F;
  std::memset(&F, 0, sizeof(F));
  ldl6_factor(a, F);
  ldl6_solve(F, b, x);
}

// the index arithmetic of the voting kernel: idx-th detection triple, pj-th marker permutation (0-based)
extern "C" void host_unrank(int n, int n_combos, int n_perms, int* combos, int* perms) {
  for (int i = 0; i < n_combos; ++i) unrank_combo3(i, n, combos[3 * i], combos[3 * i + 1], combos[3 * i + 2]);
  for (int i = 0; i < n_perms; ++i) perm_from_index(i, n, perms[3 * i], perms[3 * i + 1], perms[3 * i + 2]);
}

extern "C" void host_bearing(double u, double v, const double* k4, double* out) {
  const V3 b = bearing(u, v, k4[0], k4[1], k4[2], k4[3]);
  out[0] = b.x;
  out[1] = b.y;
  out[2] = b.z;
}

// optimisePose as k3b_refine runs it: rows = n_c x (marker xyz, detection uv)
extern "C" int host_gauss_newton(const double* rows, int n_c, const double* k4, double* T34_rows, double* cov) {
  T34 T;
  std::memcpy(T.m, T34_rows, sizeof(T.m));
  const int it = k3_gauss_newton(n_c, [&](int j, int k) -> double { return rows[5 * j + k]; }, k4[0], k4[1], k4[2],
                                 k4[3], T, cov);
  std::memcpy(T34_rows, T.m, sizeof(T.m));
  return it;
}
