// Host build of rpg_monocular_pose_estimator_amd/csrc/mpe_ddmath.h (test scaffolding, CPU tier): the device source of
// the libstdc++ / glibc restatement, against THIS image's libm / libstdc++ — the arithmetic the CPU oracle runs on.
#include <complex>
#include <cstddef>
#include "mpe_ddmath.h"
using namespace mpe::ddm;

// which: 0 log, 1 log1p, 2 exp, 3 sin, 4 cos, 5 hypot(x, y), 6 atan2(y = a, x = b), 7 pow(a, b), 8 log1p restated literally
// out_mine / out_libm: n doubles each
extern "C" void ddm_compare(int which, const double* a, const double* b, int n, double* mine, double* libm) {
  for (int i = 0; i < n; ++i) {
    double s, c;
    switch (which) {
      case 0: mine[i] = log_cr(a[i]); libm[i] = std::log(a[i]); break;
      case 1: mine[i] = log1p_cr(a[i]); libm[i] = std::log1p(a[i]); break;
      case 2: mine[i] = exp_cr(a[i]); libm[i] = std::exp(a[i]); break;
      case 3: sincos_cr(a[i], s, c); mine[i] = s; libm[i] = std::sin(a[i]); break;
      case 4: sincos_cr(a[i], s, c); mine[i] = c; libm[i] = std::cos(a[i]); break;
      case 5: mine[i] = hypot_g(a[i], b[i]); libm[i] = std::hypot(a[i], b[i]); break;
      case 6: mine[i] = atan2_cr(a[i], b[i]); libm[i] = std::atan2(a[i], b[i]); break;
      case 8: mine[i] = log1p_g(a[i]); libm[i] = std::log1p(a[i]); break;
      default: mine[i] = pow_cr(a[i], b[i]); libm[i] = std::pow(a[i], b[i]); break;
    }
  }
}
// clog and pow(complex, double): mine / lib hold (re, im) pairs
extern "C" void ddm_clog(const double* re, const double* im, int n, double* mine, double* lib) {
  for (int i = 0; i < n; ++i) {
    clog_g(re[i], im[i], mine[2 * i], mine[2 * i + 1]);
    const std::complex<double> l = std::log(std::complex<double>(re[i], im[i]));
    lib[2 * i] = l.real();
    lib[2 * i + 1] = l.imag();
  }
}
extern "C" void ddm_cpow(const double* re, const double* im, double y, int n, double* mine, double* lib) {
  // (y reaches std::pow through a volatile, as a run-time value: the library call the oracle's TU makes)
  volatile double yv = y;
  for (int i = 0; i < n; ++i) {
    cpow_g(re[i], im[i], y, mine[2 * i], mine[2 * i + 1]);
    const double yy = yv;
    const std::complex<double> p = std::pow(std::complex<double>(re[i], im[i]), yy);
    lib[2 * i] = p.real();
    lib[2 * i + 1] = p.imag();
  }
}
