// mpe_ddmath.h — std::pow(std::complex<double>, double) as libstdc++ / glibc evaluate it, for the strict voting item.
//
// Why this exists.  The reference's Ferrari solver calls
// std::pow(Q, 2.0), std::pow(P, 3.0), std::pow(R, 1/3.) and std::pow(Q, 1/3.) on std::complex<double>
// (p3p.cpp:262,264,268).  libstdc++ evaluates pow(complex z, double y) as
//     z real and positive:  pow(z.re, y)                       (glibc's real pow)
//     otherwise:            t = clog(z);  polar(exp(y * t.re), y * t.im)     (<complex>:1028-1039 of GCC 11)
// i.e. |z|^y goes through exp(y log|z|) — off the exact power by up to |y log|z|| ulp — and log|z| through glibc
// clog's five branches (log1p of (|x| - 1)(|x| + 1) near the unit circle, __x2y2m1, log(hypot) elsewhere).  Where
// Ferrari's discriminant Q^2/4 + P^3/27 cancels, those ulps pick the branch of the cube root, and with it four roots
// and a handful of votes.  The strict device arithmetic of rounds 1 - 5 used exact products and cbrt(hypot): more
// accurate, and on 1 / 65 536 C2 and 22 / 14 336 C3 frames a different histogram than the CPU oracle's.
//
// What is restated here, and how it is pinned:
//   * the STRUCTURE — libstdc++'s pow / polar and glibc 2.35's clog branch by branch, its __x2y2m1 and hypot kernel
//     operation by operation (they are sequences of IEEE additions, multiplications, one division, one square root:
//     bit-exact by construction);
//   * the transcendental PRIMITIVES — log, log1p, exp, pow, atan2, sin, cos — as correctly rounded functions:
//     evaluated in double-double (~2^-100) and rounded once.  glibc's own versions are within 0.51 - 0.55 ulp, so the
//     two agree except where glibc itself misrounds (tests/test_ddmath_host.py measures the rate against this image's
//     libm on millions of arguments: the CPU tier compiles this header for the host).
// Everything is __device__ code without tables in memory beyond a few constants; it runs for the 0.2 - 0.3 % of the
// hypotheses the fast kernels hand to the strict arithmetic (option "vote_arith" = 3) and for the strict kernel.
#pragma once
#include <hip/hip_runtime.h>

namespace mpe {
namespace ddm {

#define MPE_DDM __device__ __forceinline__

struct dd {
  double hi, lo;
};

// ---- error-free transformations and double-double arithmetic (Dekker / Knuth; FMA for the products) ------------------
MPE_DDM dd two_sum(double a, double b) {
  const double s = a + b, bb = s - a;
  return {s, (a - (s - bb)) + (b - bb)};
}
MPE_DDM dd quick_two_sum(double a, double b) {  // |a| >= |b|
  const double s = a + b;
  return {s, b - (s - a)};
}
MPE_DDM dd two_prod(double a, double b) {
  const double p = a * b;
  return {p, __builtin_fma(a, b, -p)};
}
MPE_DDM dd neg(dd a) { return {-a.hi, -a.lo}; }
MPE_DDM dd add(dd a, dd b) {
  dd s = two_sum(a.hi, b.hi);
  const dd t = two_sum(a.lo, b.lo);
  s.lo += t.hi;
  s = quick_two_sum(s.hi, s.lo);
  s.lo += t.lo;
  return quick_two_sum(s.hi, s.lo);
}
MPE_DDM dd add(dd a, double b) {
  dd s = two_sum(a.hi, b);
  s.lo += a.lo;
  return quick_two_sum(s.hi, s.lo);
}
MPE_DDM dd sub(dd a, dd b) { return add(a, neg(b)); }
MPE_DDM dd mul(dd a, dd b) {
  dd p = two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return quick_two_sum(p.hi, p.lo);
}
MPE_DDM dd mul(dd a, double b) {
  dd p = two_prod(a.hi, b);
  p.lo += a.lo * b;
  return quick_two_sum(p.hi, p.lo);
}
MPE_DDM dd div(dd a, dd b) {
  const double q1 = a.hi / b.hi;
  dd r = sub(a, mul(b, q1));
  const double q2 = r.hi / b.hi;
  r = sub(r, mul(b, q2));
  const double q3 = r.hi / b.hi;
  return add(quick_two_sum(q1, q2), q3);
}
MPE_DDM dd sqrt_dd(dd a) {  // a > 0
  const double x = sqrt(a.hi);
  dd r = sub(a, two_prod(x, x));
  const dd y = quick_two_sum(x, r.hi / (2.0 * x));
  r = sub(a, mul(y, y));
  return add(y, r.hi / (2.0 * x));
}
MPE_DDM dd scale2(dd a, int e) { return {ldexp(a.hi, e), ldexp(a.lo, e)}; }  // (exact away from the subnormals)
MPE_DDM double to_double(dd a) { return a.hi + a.lo; }

// ---- constants (tools/gen_dd_constants.py: exact rational series, rounded to nearest part by part) -------------------
// ln 2: 42 bits (k * hi is exact for |k| < 2^11), then two doubles
static __device__ const double kLn2[3] = {0x1.62e42fefa3800p-1, 0x1.ef35793c76730p-45, 0x1.f97b57a079a19p-103};
static __device__ const double kInvLn2 = 0x1.71547652b82fep+0;
static __device__ const double kPi[3] = {0x1.921fb54442d18p+1, 0x1.1a62633145c07p-53, -0x1.f1976b7ed8fbcp-109};
static __device__ const double kPi2[3] = {0x1.921fb54442d18p+0, 0x1.1a62633145c07p-54, -0x1.f1976b7ed8fbcp-110};
// 1 / (2k + 1), k = 0 .. 25, as (hi, lo) pairs
static __device__ const double kOddRec[52] = {
    0x1.0000000000000p+0, 0x0.0p+0, 0x1.5555555555555p-2, 0x1.5555555555555p-56, 0x1.999999999999ap-3,
    -0x1.999999999999ap-57, 0x1.2492492492492p-3, 0x1.2492492492492p-57, 0x1.c71c71c71c71cp-4, 0x1.c71c71c71c71cp-58,
    0x1.745d1745d1746p-4, -0x1.745d1745d1746p-59, 0x1.3b13b13b13b14p-4, -0x1.3b13b13b13b14p-58, 0x1.1111111111111p-4,
    0x1.1111111111111p-60, 0x1.e1e1e1e1e1e1ep-5, 0x1.e1e1e1e1e1e1ep-61, 0x1.af286bca1af28p-5, 0x1.af286bca1af28p-59,
    0x1.8618618618618p-5, 0x1.8618618618618p-59, 0x1.642c8590b2164p-5, 0x1.642c8590b2164p-60, 0x1.47ae147ae147bp-5,
    -0x1.eb851eb851eb8p-61, 0x1.2f684bda12f68p-5, 0x1.2f684bda12f68p-59, 0x1.1a7b9611a7b96p-5, 0x1.1a7b9611a7b96p-61,
    0x1.0842108421084p-5, 0x1.0842108421084p-60, 0x1.f07c1f07c1f08p-6, -0x1.f07c1f07c1f08p-61, 0x1.d41d41d41d41dp-6,
    0x1.0750750750750p-60, 0x1.bacf914c1bad0p-6, -0x1.bacf914c1bad0p-60, 0x1.a41a41a41a41ap-6, 0x1.0690690690690p-60,
    0x1.8f9c18f9c18fap-6, -0x1.f3831f3831f38p-61, 0x1.7d05f417d05f4p-6, 0x1.7d05f417d05f4p-62, 0x1.6c16c16c16c17p-6,
    -0x1.f49f49f49f49fp-61, 0x1.5c9882b931057p-6, 0x1.310572620ae4cp-61, 0x1.4e5e0a72f0539p-6, 0x1.e0a72f0539783p-60,
    0x1.4141414141414p-6, 0x1.4141414141414p-62};
// 1 / n!, n = 0 .. 23, as (hi, lo) pairs
static __device__ const double kFacRec[48] = {
    0x1.0000000000000p+0, 0x0.0p+0, 0x1.0000000000000p+0, 0x0.0p+0, 0x1.0000000000000p-1, 0x0.0p+0,
    0x1.5555555555555p-3, 0x1.5555555555555p-57, 0x1.5555555555555p-5, 0x1.5555555555555p-59, 0x1.1111111111111p-7,
    0x1.1111111111111p-63, 0x1.6c16c16c16c17p-10, -0x1.f49f49f49f49fp-65, 0x1.a01a01a01a01ap-13, 0x1.a01a01a01a01ap-73,
    0x1.a01a01a01a01ap-16, 0x1.a01a01a01a01ap-76, 0x1.71de3a556c734p-19, -0x1.c154f8ddc6c00p-73, 0x1.27e4fb7789f5cp-22,
    0x1.cbbc05b4fa99ap-76, 0x1.ae64567f544e4p-26, -0x1.c062e06d1f209p-80, 0x1.1eed8eff8d898p-29, -0x1.2aec959e14c06p-83,
    0x1.6124613a86d09p-33, 0x1.f28e0cc748ebep-87, 0x1.93974a8c07c9dp-37, 0x1.05d6f8a2efd1fp-92, 0x1.ae7f3e733b81fp-41,
    0x1.1d8656b0ee8cbp-97, 0x1.ae7f3e733b81fp-45, 0x1.1d8656b0ee8cbp-101, 0x1.952c77030ad4ap-49, 0x1.ac981465ddc6cp-103,
    0x1.6827863b97d97p-53, 0x1.eec01221a8b0bp-107, 0x1.2f49b46814157p-57, 0x1.2650f61dbdcb4p-112, 0x1.e542ba4020225p-62,
    0x1.ea72b4afe3c2fp-120, 0x1.71b8ef6dcf572p-66, -0x1.d043ae40c4647p-120, 0x1.0ce396db7f853p-70, -0x1.aebcdbd20331cp-124,
    0x1.761b41316381ap-75, -0x1.3423c7d91404fp-130};
MPE_DDM dd odd_rec(int k) { return {kOddRec[2 * k], kOddRec[2 * k + 1]}; }
MPE_DDM dd fac_rec(int n) { return {kFacRec[2 * n], kFacRec[2 * n + 1]}; }

// ---- log, exp in double-double -----------------------------------------------------------------------------------------
// log(a), a > 0 finite and normal: a = 2^e m, m in [sqrt(1/2), sqrt(2)); log m = 2 atanh(s), s = (m - 1) / (m + 1),
// |s| <= 0.1716: 25 terms of the odd series reach 2^-127
MPE_DDM dd log_dd(dd a) {
  int e = ilogb(a.hi);
  dd m = scale2(a, -e);
  if (m.hi > 1.4142135623730951) {
    m = scale2(m, -1);
    ++e;
  }
  const dd s = div(add(m, -1.0), add(m, 1.0));
  const dd s2 = mul(s, s);
  dd acc = odd_rec(24);
#pragma unroll 1
  for (int k = 23; k >= 0; --k) acc = add(mul(acc, s2), odd_rec(k));
  const dd r = scale2(mul(s, acc), 1);
  const double ed = (double)e;
  // e ln 2 = e hi (exact) + e (mid, lo)
  return add(dd{ed * kLn2[0], 0.0}, add(mul(dd{kLn2[1], kLn2[2]}, ed), r));
}
// exp(x) for |x| < 745: x = k ln 2 + r, exp(r / 64) by 14 terms of the Taylor series (|r / 64| <= 0.0055), six squarings
MPE_DDM dd exp_dd(dd x) {
  const double k = rint(x.hi * kInvLn2);
  dd r = add(two_sum(x.hi, -k * kLn2[0]), x.lo);
  r = sub(r, two_prod(k, kLn2[1]));
  r = add(r, -k * kLn2[2]);
  r = scale2(r, -6);
  dd acc = fac_rec(14);
#pragma unroll 1
  for (int n = 13; n >= 0; --n) acc = add(mul(acc, r), fac_rec(n));
#pragma unroll 1
  for (int i = 0; i < 6; ++i) acc = mul(acc, acc);
  return scale2(acc, (int)k);
}

// ---- correctly rounded real functions (rounded once from ~2^-100) -----------------------------------------------------
MPE_DDM double log_cr(double x) {
  if (!(x > 0.0) || !(x < INFINITY) || x < 2.2250738585072014e-308) return log(x);  // (zero, negative, NaN, inf, subnormal)
  if (x == 1.0) return 0.0;
  return to_double(log_dd(dd{x, 0.0}));
}
// (a correctly rounded log1p for reference; clog below uses glibc's own, log1p_g: that one is NOT within 0.55 ulp —
//  the correctly rounded value differs from it on 8.6 % of the arguments in (-0.75, 3))
MPE_DDM double log1p_cr(double x) {
  if (!(x > -1.0) || !(x < INFINITY)) return log1p(x);
  if (x == 0.0) return x;
  if (fabs(x) < 0x1p-900) return x;  // (log1p(x) = x (1 - x / 2 + ..) rounds to x; keeps log_dd away from subnormal parts)
  return to_double(log_dd(two_sum(1.0, x)));
}
MPE_DDM double exp_cr(double x) {
  if (!(x == x)) return x;
  if (x > 709.7) return exp(x);   // (overflow edge: the library's answer)
  if (x < -708.0) return exp(x);  // (subnormal results: likewise)
  return to_double(exp_dd(dd{x, 0.0}));
}
// pow(x, y) for x > 0 finite: exp(y log x) with the logarithm and the product in double-double
MPE_DDM double pow_cr(double x, double y) {
  if (!(x > 0.0) || !(x < INFINITY) || x < 2.2250738585072014e-308 || !(y == y)) return pow(x, y);
  if (x == 1.0) return 1.0;
  const dd p = mul(log_dd(dd{x, 0.0}), y);
  if (!(fabs(p.hi) < 700.0)) return pow(x, y);
  return to_double(exp_dd(p));
}
// sin and cos of |theta| <= pi (the quartic only asks for |theta| <= pi / 3): theta / 16 by the Taylor series (12 terms:
// (pi / 16)^24 / 24! < 2^-135), four angle doublings.  Relative accuracy ~2^-98 for the sine everywhere and for the
// cosine away from pi / 2.
MPE_DDM void sincos_cr(double theta, double& s_out, double& c_out) {
  if (!(fabs(theta) <= 3.1415926535897936)) {
    sincos(theta, &s_out, &c_out);
    return;
  }
  const dd a = {ldexp(theta, -4), 0.0};
  const dd a2 = mul(a, a);
  dd sa = fac_rec(23), ca = fac_rec(22);
#pragma unroll 1
  for (int k = 10; k >= 0; --k) {
    sa = sub(fac_rec(2 * k + 1), mul(sa, a2));
    ca = sub(fac_rec(2 * k), mul(ca, a2));
  }
  dd s = mul(sa, a), c = ca;
#pragma unroll 1
  for (int i = 0; i < 4; ++i) {
    const dd s2 = scale2(mul(s, c), 1);
    c = add(scale2(mul(s, s), 1), -1.0);
    c = neg(c);
    s = s2;
  }
  s_out = to_double(s);
  c_out = to_double(c);
}
// atan of t in [0, 1]: three half-angle steps t <- t / (1 + sqrt(1 + t^2)) (to tan(pi / 32) = 0.0985), 20 terms
MPE_DDM dd atan_dd(dd t) {
#pragma unroll 1
  for (int i = 0; i < 3; ++i) t = div(t, add(sqrt_dd(add(mul(t, t), 1.0)), 1.0));
  const dd u2 = mul(t, t);
  dd acc = odd_rec(19);
#pragma unroll 1
  for (int k = 18; k >= 0; --k) acc = sub(odd_rec(k), mul(acc, u2));
  return scale2(mul(t, acc), 3);
}
MPE_DDM double atan2_cr(double y, double x) {
  if (!(x == x) || !(y == y)) return x + y;
  const double ay = fabs(y), ax = fabs(x);
  if (ay == 0.0) return __builtin_signbit(x) ? copysign(kPi[0], y) : y;  // atan2(+-0, -x) = +-pi, atan2(+-0, +x) = +-0
  if (ax == 0.0) return copysign(kPi2[0], y);
  if (!(ax < INFINITY) || !(ay < INFINITY) || ax < 0x1p-500 || ay < 0x1p-500 || ax > 0x1p500 || ay > 0x1p500)
    return atan2(y, x);  // (infinities, and magnitudes whose quotient leaves the normal range: the library's answer)
  const bool inv = ay > ax;
  const dd t = inv ? div(dd{ax, 0.0}, dd{ay, 0.0}) : div(dd{ay, 0.0}, dd{ax, 0.0});
  dd a = atan_dd(t);
  if (inv) a = add(add(neg(a), dd{kPi2[0], kPi2[1]}), kPi2[2]);
  if (x < 0.0) a = add(add(neg(a), dd{kPi[0], kPi[1]}), kPi[2]);
  return copysign(to_double(a), y);
}

// high word of a double / a double with its high word replaced (the fdlibm idiom of log1p below)
MPE_DDM int hi_word(double x) { return (int)((unsigned long long)__builtin_bit_cast(long long, x) >> 32); }
MPE_DDM double with_hi_word(double x, int hi) {
  const unsigned long long u = ((unsigned long long)__builtin_bit_cast(long long, x) & 0xffffffffull) |
                               ((unsigned long long)(unsigned)hi << 32);
  return __builtin_bit_cast(double, (long long)u);
}
// ---- glibc 2.35, operation by operation ---------------------------------------------------------------------------------
// hypot (sysdeps/ieee754/dbl-64/e_hypot.c, the kernel without a fast FMA: sqrt and one correction step)
MPE_DDM double hypot_kernel_g(double ax, double ay) {
  double h = sqrt(ax * ax + ay * ay);
  double t1, t2;
  if (h <= 2.0 * ay) {
    const double delta = h - ay;
    t1 = ax * (2.0 * delta - ax);
    t2 = (delta - 2.0 * (ax - ay)) * delta;
  } else {
    const double delta = h - ax;
    t1 = 2.0 * delta * (ax - 2.0 * ay);
    t2 = (4.0 * delta - ay) * ay + delta * delta;
  }
  h -= (t1 + t2) / (2.0 * h);
  return h;
}
MPE_DDM double hypot_g(double x, double y) {
  if (!(fabs(x) < INFINITY) || !(fabs(y) < INFINITY)) return hypot(x, y);
  x = fabs(x);
  y = fabs(y);
  const double ax = x < y ? y : x, ay = x < y ? x : y;
  if (ax > 0x1p+511) {
    if (ay <= ax * 0x1p-54) return ax + ay;
    return hypot_kernel_g(ax * 0x1p-600, ay * 0x1p-600) / 0x1p-600;
  }
  if (ay < 0x1p-459) {
    if (ax >= ay / 0x1p-54) return ax + ay;
    return hypot_kernel_g(ax / 0x1p-600, ay / 0x1p-600) * 0x1p-600;
  }
  if (ax >= ay / 0x1p-54) return ax + ay;
  return hypot_kernel_g(ax, ay);
}
// __x2y2m1 (sysdeps/ieee754/dbl-64/x2y2m1.c): x^2 + y^2 - 1 from the two exact products and -1, the five terms
// sorted by magnitude and added with the rounding errors carried along
MPE_DDM void sort_abs(double* v, int n) {  // ascending |v|, insertion sort (n <= 5)
  for (int i = 1; i < n; ++i) {
    const double x = v[i];
    int j = i - 1;
    while (j >= 0 && fabs(v[j]) > fabs(x)) {
      v[j + 1] = v[j];
      --j;
    }
    v[j + 1] = x;
  }
}
MPE_DDM double x2y2m1_g(double x, double y) {
  double v[5];
  const dd px = two_prod(x, x), py = two_prod(y, y);
  v[0] = px.lo;
  v[1] = px.hi;
  v[2] = py.lo;
  v[3] = py.hi;
  v[4] = -1.0;
  sort_abs(v, 5);
#pragma unroll
  for (int i = 0; i <= 3; ++i) {
    const double hi = v[i + 1] + v[i];
    const double lo = (v[i + 1] - hi) + v[i];
    v[i + 1] = hi;
    v[i] = lo;
    sort_abs(v + i + 1, 4 - i);
  }
  return v[4] + v[3] + v[2] + v[1] + v[0];
}
// log1p (sysdeps/ieee754/dbl-64/s_log1p.c: fdlibm's algorithm, the polynomial in glibc's four-way split) — errors
// up to ~0.9 ulp, so it is restated literally: every operation is an IEEE addition, multiplication or division
MPE_DDM double log1p_g(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
               Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
               Lp7 = 1.479819860511658591e-01;
  double hfsq, f = 0.0, c = 0.0, s, z, R, u;
  int k = 1, hu = 0;
  const int hx = hi_word(x), ax = hx & 0x7fffffff;
  if (hx < 0x3FDA827A) {  // x < 0.41422
    if (ax >= 0x3ff00000) {  // x <= -1.0
      if (x == -1.0) return -INFINITY;
      return (x - x) / (x - x);
    }
    if (ax < 0x3e200000) {  // |x| < 2^-29
      if (ax < 0x3c900000) return x;  // |x| < 2^-54
      return x - x * x * 0.5;
    }
    if (hx > 0 || hx <= (int)0xbfd2bec3) {  // -0.2929 < x < 0.41422
      k = 0;
      f = x;
      hu = 1;
    }
  } else if (hx >= 0x7ff00000) {
    return x + x;
  }
  if (k != 0) {
    if (hx < 0x43400000) {
      u = 1.0 + x;
      hu = hi_word(u);
      k = (hu >> 20) - 1023;
      c = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0);  // correction term
      c /= u;
    } else {
      u = x;
      hu = hi_word(u);
      k = (hu >> 20) - 1023;
      c = 0;
    }
    hu &= 0x000fffff;
    if (hu < 0x6a09e) {
      u = with_hi_word(u, hu | 0x3ff00000);  // normalize u
    } else {
      k += 1;
      u = with_hi_word(u, hu | 0x3fe00000);  // normalize u / 2
      hu = (0x00100000 - hu) >> 2;
    }
    f = u - 1.0;
  }
  hfsq = 0.5 * f * f;
  if (hu == 0) {  // |f| < 2^-20
    if (f == 0.0) {
      if (k == 0) return 0.0;
      c += k * ln2_lo;
      return k * ln2_hi + c;
    }
    R = hfsq * (1.0 - 0.66666666666666666 * f);
    if (k == 0) return f - R;
    return k * ln2_hi - ((R - (k * ln2_lo + c)) - f);
  }
  s = f / (2.0 + f);
  z = s * s;
  const double R1 = z * Lp1, z2 = z * z;
  const double R2 = Lp2 + z * Lp3, z4 = z2 * z2;
  const double R3 = Lp4 + z * Lp5, z6 = z4 * z2;
  const double R4 = Lp6 + z * Lp7;
  R = R1 + z2 * R2 + z4 * R3 + z6 * R4;
  if (k == 0) return f - (hfsq - s * (hfsq + R));
  return k * ln2_hi - ((hfsq - (s * (hfsq + R) + (k * ln2_lo + c))) - f);
}
// clog (math/s_clog_template.c): real part by the branch the magnitudes select, imaginary part atan2
MPE_DDM void clog_g(double re, double im, double& lr, double& li) {
  if (re == 0.0 && im == 0.0) {
    li = copysign(__builtin_signbit(re) ? kPi[0] : 0.0, im);
    lr = -1.0 / fabs(re);
    return;
  }
  if (!(re == re) || !(im == im)) {  // (NaN operands: NaN out, as every caller here treats them)
    lr = li = re + im;
    return;
  }
  double absx = fabs(re), absy = fabs(im);
  if (absx < absy) {
    const double t = absx;
    absx = absy;
    absy = t;
  }
  int scale = 0;
  if (absx > 0x1.fffffffffffffp+1022) {  // DBL_MAX / 2
    scale = -1;
    absx = ldexp(absx, scale);
    absy = absy >= 2.2250738585072014e-308 * 2 ? ldexp(absy, scale) : 0.0;
  } else if (absx < 2.2250738585072014e-308 && absy < 2.2250738585072014e-308) {
    scale = 53;
    absx = ldexp(absx, scale);
    absy = ldexp(absy, scale);
  }
  const double eps = 0x1p-52;
  if (absx == 1.0 && scale == 0) {
    lr = log1p_g(absy * absy) / 2.0;
  } else if (absx > 1.0 && absx < 2.0 && absy < 1.0 && scale == 0) {
    double d2m1 = (absx - 1.0) * (absx + 1.0);
    if (absy >= eps) d2m1 += absy * absy;
    lr = log1p_g(d2m1) / 2.0;
  } else if (absx < 1.0 && absx >= 0.5 && absy < eps / 2.0 && scale == 0) {
    const double d2m1 = (absx - 1.0) * (absx + 1.0);
    lr = log1p_g(d2m1) / 2.0;
  } else if (absx < 1.0 && absx >= 0.5 && scale == 0 && absx * absx + absy * absy >= 0.5) {
    lr = log1p_g(x2y2m1_g(absx, absy)) / 2.0;
  } else {
    const double d = hypot_g(absx, absy);
    lr = log_cr(d) - scale * 0x1.62e42fefa39efp-1;  // M_LN2
  }
  li = atan2_cr(im, re);
}
// std::pow(std::complex<double>(re, im), y)  (libstdc++ <complex>: real pow for a positive real, else polar form)
MPE_DDM void cpow_g(double re, double im, double y, double& out_re, double& out_im) {
  if (im == 0.0 && re > 0.0) {
    // (y = 2: the correctly rounded square IS the product; y = 3: the exact square times re, rounded once)
    out_re = y == 2.0 ? re * re : y == 3.0 ? to_double(mul(two_prod(re, re), re)) : pow_cr(re, y);
    out_im = 0.0;
    return;
  }
  double lr, li;
  clog_g(re, im, lr, li);
  const double rho = exp_cr(y * lr);
  const double theta = y * li;
  double s, c;
  // a negative real to the powers 2 and 3: theta = 2 pi_d, 3 pi_d exactly; sin and cos of those doubles, correctly
  // rounded (the dust that gives the discriminant its branch-selecting imaginary part)
  if (theta == 2.0 * kPi[0]) {
    s = -2.4492935982947064e-16;
    c = 1.0;
  } else if (theta == 3.0 * kPi[0]) {
    s = 3.6739403974420594e-16;
    c = -1.0;
  } else {
    sincos_cr(theta, s, c);
  }
  out_re = rho * c;
  out_im = rho * s;
}

#undef MPE_DDM
}  // namespace ddm
}  // namespace mpe
