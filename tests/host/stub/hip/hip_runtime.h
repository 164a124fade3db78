// Stand-in for <hip/hip_runtime.h> when the DEVICE geometry headers are compiled for the host by the CPU-tier tests
// (tests/test_geometry_host.py): the one-lane meaning of the HIP spellings they use.  Test scaffolding only.
#pragma once
#include <cmath>
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__
// hardware reciprocal / reciprocal square root estimates: the Newton-Raphson code behind them converges to the same
// value from any start that is good to a few bits
#define __builtin_amdgcn_rcp(x) (1.0 / (x))
#define __builtin_amdgcn_rsq(x) (1.0 / std::sqrt(x))
// fast single-precision sine / cosine of the cube-root seed (refined by Newton steps in double afterwards)
// (glibc declares __sinf / __cosf itself but does not export them: route the names to the public functions)
#define __sinf(x) sinf(x)
#define __cosf(x) cosf(x)
// single-precision hardware units of the cube-root seed (v_exp_f32 / v_log_f32 are base 2, v_sin_f32 / v_cos_f32
// take revolutions)
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_logf(x) log2f(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_sinf(x) sinf(6.28318530717958647692f * (x))
#define __builtin_amdgcn_cosf(x) cosf(6.28318530717958647692f * (x))
// high word of a double (biased exponent extraction in mpe_p3p.h) and the integer min it uses
#include <cstring>
static inline int __double2hiint(double x) {
  unsigned long long u;
  std::memcpy(&u, &x, 8);
  return (int)(u >> 32);
}
#include <algorithm>
using std::min;
