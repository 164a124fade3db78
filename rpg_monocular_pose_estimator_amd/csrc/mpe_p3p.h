// mpe_p3p.h — FP64 device geometry for gfx950: small vectors, complex Ferrari quartic, Kneip P3P.
//
// Written for the CDNA4 VALU (wave64, FP64 FMA at 16 lanes/clk/SIMD); everything is inlined into
// the calling kernel, state lives in VGPRs, no scratch arrays with dynamic indices.
// Behaviour follows the reference P3P (monocular_pose_estimator_lib/src/p3p.cpp:65-286):
// principal-branch complex sqrt / pow, REAL PARTS of the complex roots are consumed, NaNs flow
// through and are filtered by the caller (pose_estimator.cpp:653).
#pragma once
#include <hip/hip_runtime.h>
#include "mpe_ddmath.h"

namespace mpe {

struct V3 {
  double x, y, z;
};
__device__ __forceinline__ V3 operator-(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator+(const V3& a, const V3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 vdiv(const V3& a, double s) { return {a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ V3 cross(const V3& a, const V3& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ double dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ double norm(const V3& a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }

// 3x3 stored as three row vectors
struct M3 {
  V3 r0, r1, r2;
};
__device__ __forceinline__ V3 mul(const M3& A, const V3& v) { return {dot(A.r0, v), dot(A.r1, v), dot(A.r2, v)}; }
// A^T * v
__device__ __forceinline__ V3 mulT(const M3& A, const V3& v) {
  return {A.r0.x * v.x + A.r1.x * v.y + A.r2.x * v.z, A.r0.y * v.x + A.r1.y * v.y + A.r2.y * v.z,
          A.r0.z * v.x + A.r1.z * v.y + A.r2.z * v.z};
}
__device__ __forceinline__ M3 mul(const M3& A, const M3& B) {
  M3 C;
  C.r0 = {A.r0.x * B.r0.x + A.r0.y * B.r1.x + A.r0.z * B.r2.x, A.r0.x * B.r0.y + A.r0.y * B.r1.y + A.r0.z * B.r2.y,
          A.r0.x * B.r0.z + A.r0.y * B.r1.z + A.r0.z * B.r2.z};
  C.r1 = {A.r1.x * B.r0.x + A.r1.y * B.r1.x + A.r1.z * B.r2.x, A.r1.x * B.r0.y + A.r1.y * B.r1.y + A.r1.z * B.r2.y,
          A.r1.x * B.r0.z + A.r1.y * B.r1.z + A.r1.z * B.r2.z};
  C.r2 = {A.r2.x * B.r0.x + A.r2.y * B.r1.x + A.r2.z * B.r2.x, A.r2.x * B.r0.y + A.r2.y * B.r1.y + A.r2.z * B.r2.y,
          A.r2.x * B.r0.z + A.r2.y * B.r1.z + A.r2.z * B.r2.z};
  return C;
}
__device__ __forceinline__ M3 transpose(const M3& A) {
  return {{A.r0.x, A.r1.x, A.r2.x}, {A.r0.y, A.r1.y, A.r2.y}, {A.r0.z, A.r1.z, A.r2.z}};
}

// ---- complex double (principal branches, as libstdc++ <complex> on top of C99 csqrt/clog) ----
struct C2 {
  double re, im;
};
__device__ __forceinline__ C2 cadd(C2 a, C2 b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ C2 csub(C2 a, C2 b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ C2 cscale(C2 a, double s) { return {a.re * s, a.im * s}; }
// (a + ib) / (c + id), Smith's algorithm (what libgcc's __divdc3 evaluates for finite operands)
__device__ __forceinline__ C2 cdiv(C2 n, C2 d) {
  double a = n.re, b = n.im, c = d.re, e = d.im;
  if (fabs(c) < fabs(e)) {
    double ratio = c / e, denom = c * ratio + e;
    return {(a * ratio + b) / denom, (b * ratio - a) / denom};
  }
  double ratio = e / c, denom = e * ratio + c;
  return {(b * ratio + a) / denom, (b - a * ratio) / denom};
}
// principal square root (glibc csqrt for operands in the normal range).  GLIBC: |z| by glibc's own hypot kernel
// (mpe_ddmath.h) instead of the device library's
template <bool GLIBC = false>
__device__ __forceinline__ C2 csqrt_(C2 z) {
  if (z.im == 0.0) {
    if (z.re < 0.0) return {0.0, copysign(sqrt(-z.re), z.im)};
    return {fabs(sqrt(z.re)), z.im};
  }
  double d = GLIBC ? ddm::hypot_g(z.re, z.im) : hypot(z.re, z.im);
  double r, s;
  if (z.re > 0.0) {
    r = sqrt(0.5 * (d + z.re));
    s = 0.5 * (z.im / r);
  } else {
    s = sqrt(0.5 * (d - z.re));
    r = fabs(0.5 * (z.im / s));
  }
  return {r, copysign(s, z.im)};
}
// std::pow(complex z, double y) for y = 1/3: real pow when z is a positive real, else
// polar(exp(y log|z|), y arg z).  |z|^(1/3) is evaluated with cbrt (same value to ~1 ulp).
__device__ __forceinline__ C2 cpow_third(C2 z) {
  const double third = 1.0 / 3.0;
  if (z.im == 0.0 && z.re > 0.0) return {cbrt(z.re), 0.0};
  double rho = cbrt(hypot(z.re, z.im));
  double phi = third * atan2(z.im, z.re);
  double s, c;
  sincos(phi, &s, &c);
  return {rho * c, rho * s};
}
// std::pow(complex(q,0), 2.0) and std::pow(complex(p,0), 3.0): for a negative real the library
// goes through polar(|x|^y, y*pi); sin(2*pi_d) and sin(3*pi_d) are not zero in double, which gives
// the discriminant its (branch-selecting) imaginary dust.  Replicated with the constants.
__device__ __forceinline__ C2 cpow2_real(double q) {
  double m = q * q;
  if (q < 0.0) return {m, m * -2.4492935982947064e-16};
  return {m, 0.0};
}
__device__ __forceinline__ C2 cpow3_real(double p) {
  double m = p * p * p;
  if (p < 0.0) return {m, -m * 3.6739403974420594e-16};
  return {m, 0.0};
}

// ... and the same three powers as libstdc++ / glibc evaluate them (mpe_ddmath.h: exp(y log|z|) through clog's branches,
// correctly rounded primitives) — option "vote_arith" = 3
__device__ __forceinline__ C2 cpow_glibc(C2 z, double y) {
  C2 r;
  ddm::cpow_g(z.re, z.im, y, r.re, r.im);
  return r;
}

// Ferrari, real parts of the four (possibly complex) roots.  p3p.cpp:238-286
// GLIBC = false: exact products and cbrt(hypot) for the three complex powers (rounds 1 - 5); true: libstdc++'s
// pow(complex, double) restated (the CPU reference's own digits in Ferrari's unstable corner)
template <bool GLIBC = false>
__device__ __forceinline__ void solve_quartic(double A, double B, double C, double D, double E, double rr[4]) {
  double A_pw2 = A * A, B_pw2 = B * B;
  double A_pw3 = A_pw2 * A, B_pw3 = B_pw2 * B;
  double A_pw4 = A_pw3 * A, B_pw4 = B_pw3 * B;
  double alpha = -3 * B_pw2 / (8 * A_pw2) + C / A;
  double beta = B_pw3 / (8 * A_pw3) - B * C / (2 * A_pw2) + D / A;
  double gamma = -3 * B_pw4 / (256 * A_pw4) + B_pw2 * C / (16 * A_pw3) - B * D / (4 * A_pw2) + E / A;
  double alpha_pw2 = alpha * alpha, alpha_pw3 = alpha_pw2 * alpha;

  double Pr = -alpha_pw2 / 12 - gamma;
  double Qr = -alpha_pw3 / 108 + alpha * gamma / 3 - (beta * beta) / 8;
  C2 q2 = GLIBC ? cpow_glibc(C2{Qr, 0.0}, 2.0) : cpow2_real(Qr), p3 = GLIBC ? cpow_glibc(C2{Pr, 0.0}, 3.0) : cpow3_real(Pr);
  C2 disc = {q2.re / 4.0 + p3.re / 27.0, q2.im / 4.0 + p3.im / 27.0};
  C2 sq = csqrt_<GLIBC>(disc);
  C2 R = {-Qr / 2.0 + sq.re, sq.im};  // -Q/2 has imaginary part -0/2 = -0
  C2 U = GLIBC ? cpow_glibc(R, 1.0 / 3.0) : cpow_third(R);
  C2 y;
  if (U.re == 0.0) {
    C2 qc = GLIBC ? cpow_glibc(C2{Qr, 0.0}, 1.0 / 3.0) : cpow_third(C2{Qr, 0.0});
    y = {-5.0 * alpha / 6.0 - qc.re, -qc.im};
  } else {
    C2 t = cdiv(C2{Pr, 0.0}, cscale(U, 3.0));
    y = {-5.0 * alpha / 6.0 - t.re + U.re, -t.im + U.im};
  }
  C2 w = csqrt_<GLIBC>(C2{alpha + 2.0 * y.re, 2.0 * y.im});
  C2 bw = cdiv(C2{2.0 * beta, 0.0}, w);
  C2 base = {3.0 * alpha + 2.0 * y.re, 2.0 * y.im};
  C2 s1 = csqrt_<GLIBC>(C2{-(base.re + bw.re), -(base.im + bw.im)});
  C2 s2 = csqrt_<GLIBC>(C2{-(base.re - bw.re), -(base.im - bw.im)});
  double off = -B / (4.0 * A);
  rr[0] = off + 0.5 * (w.re + s1.re);
  rr[1] = off + 0.5 * (w.re - s1.re);
  rr[2] = off + 0.5 * (-w.re + s2.re);
  rr[3] = off + 0.5 * (-w.re - s2.re);
}

// single-precision hardware units used by the cube-root seed (the host-tier build maps the builtins to libm)
__device__ __forceinline__ float p3p_exp2f(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32
__device__ __forceinline__ float p3p_log2f(float x) { return __builtin_amdgcn_logf(x); }    // v_log_f32 (base 2)
__device__ __forceinline__ float p3p_rcpf(float x) { return __builtin_amdgcn_rcpf(x); }     // v_rcp_f32
__device__ __forceinline__ float p3p_sin_rev(float x) { return __builtin_amdgcn_sinf(x); }  // sin(2 pi x)
__device__ __forceinline__ float p3p_cos_rev(float x) { return __builtin_amdgcn_cosf(x); }  // cos(2 pi x)

// ---- fast arithmetic of the voting kernel (option "vote_arith" = 1, the default) -----------------
// Newton-Raphson reciprocal / division / square root on top of v_rcp_f64 / v_rsq_f64, without the
// range scaling and fix-up of the IEEE expansions (12 / 22 VALU ops each): ~7 / 9 ops, <= 1 ulp for
// normal-range operands; NaN in -> NaN out, negative radicand -> NaN (the voting kernel relies on
// that to drop |root| > 1), sqrt(0) = 0.
__device__ __forceinline__ double rcp_nr(double b) {
  double x = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, x, 1.0);
  x = __builtin_fma(x, e, x);
  e = __builtin_fma(-b, x, 1.0);
  x = __builtin_fma(x, e, x);
  return x;
}
__device__ __forceinline__ double div_nr(double a, double b) {
  const double x = rcp_nr(b);
  const double q = a * x;
  const double r = __builtin_fma(-b, q, a);
  return __builtin_fma(r, x, q);
}
// sqrt(a) and, as a by-product of the same Newton-Raphson sequence, hh = 1 / (2 sqrt(a)) (relative error ~1e-16):
// a ready reciprocal for a division by the root that follows (a == 0: the root is 0, hh is not usable)
__device__ __forceinline__ double sqrt_nr_h(double a, double& hh) {
  const double y = __builtin_amdgcn_rsq(a);
  double g = a * y, h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  const double d = __builtin_fma(-g, g, a);
  g = __builtin_fma(d, h, g);
  hh = h;
  return a == 0.0 ? 0.0 : g;
}
__device__ __forceinline__ double sqrt_nr(double a) {
  double hh;
  return sqrt_nr_h(a, hh);
}
// 1/sqrt(a)
__device__ __forceinline__ double rsqrt_nr(double a) {
  const double y = __builtin_amdgcn_rsq(a);
  double g = a * y, h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  r = __builtin_fma(-h, g, 0.5);
  h = __builtin_fma(h, r, h);
  return 2.0 * h;
}
// ---- voting-kernel quartic, literal operation order -----------------------------------------------
// Same sequence of operations as solve_quartic (= p3p.cpp:238-286 statement by statement), with the
// divisions / square roots in their Newton-Raphson forms (correctly rounded in all but rare cases) and
// |z| from a compensated x^2 + y^2.  Keeping the ORDER matters more than the cost of the operations:
// in the unstable corner of Ferrari's method (alpha + 2y ~ 0) every rounding is amplified ~1e14 times,
// and a reformulated variant (monic coefficients, one reciprocal) disagreed with the CPU path's votes 4x
// more often.
__device__ __forceinline__ double hypot_acc(double x, double y) {
  const double s = x * x, t = y * y;
  const double es = __builtin_fma(x, x, -s), et = __builtin_fma(y, y, -t);
  const double sum = s + t;
  const double bb = sum - s;
  const double err = ((s - (sum - bb)) + (t - bb)) + (es + et);  // two-sum residue + product residues
  double h;
  const double r = sqrt_nr_h(sum, h);
  return r + err * h;  // first-order correction of sqrt(sum + err): err / (2 r), the reciprocal from the root itself
}
// ---- cheaper forms of the same operations ------------------------------------------------------------
// x / c for a compile-time constant c: reciprocal constant + one residual correction (3 ops, <= 1 ulp)
__device__ __forceinline__ double div_const(double a, double c, double rc) {
  const double q = a * rc;
  const double r = __builtin_fma(-c, q, a);
  return __builtin_fma(r, rc, q);
}
// a / den with a ready approximation rden of 1/den (relative error ~1e-16): residual-corrected product
__device__ __forceinline__ double div_with_rcp(double a, double den, double rden) {
  const double q = a * rden;
  const double r = __builtin_fma(-den, q, a);
  return __builtin_fma(r, rden, q);
}
// (a + ib) / (c + id), Smith's algorithm as cdiv with the two cases folded into operand selects (lanes of a
// wave take both cases, a branch would run both sides): the same operations on the same operands, and the two
// quotients share one reciprocal of denom.
__device__ __forceinline__ C2 cdiv_lit2(C2 n, C2 d) {
  const bool swap = fabs(d.re) < fabs(d.im);
  const double big = swap ? d.im : d.re, small = swap ? d.re : d.im;
  const double p = swap ? n.re : n.im, q = swap ? n.im : n.re;
  const double ratio = div_nr(small, big), denom = small * ratio + big, x = rcp_nr(denom);
  // swap: (a ratio + b, b ratio - a);  else: (b ratio + a, b - a ratio) = (p ratio + q, -(q ratio - p))
  const double t = q * ratio - p;
  return {div_with_rcp(p * ratio + q, denom, x), div_with_rcp(swap ? t : -t, denom, x)};
}
// csqrt_ with its two half-planes folded the same way: t = sqrt((|z| + |re|) / 2), u = im / (2 t);
// re > 0: (t, u), else (|u|, copysign(t, im)).  The real-axis case of csqrt_ (im == 0: sqrt(|re|) on the real or the
// imaginary axis, im keeps its signed zero) goes through the SAME instructions — its radicand |re| is selected in
// front of the one square root, its u is im itself — because in a wave of 64 hypotheses some lane nearly always has a
// real operand (a real discriminant, a real resolvent root) and a branch would make the whole wave walk both sides.
// The division by t takes its reciprocal from the square root's own Newton sequence (2h = 1/t).
__device__ __forceinline__ C2 csqrt_lit2(C2 z) {
  const bool real = z.im == 0.0;
  const double are = fabs(z.re);
  const double d = hypot_acc(z.re, z.im);
  double h;
  const double t = sqrt_nr_h(real ? are : 0.5 * (d + are), h);
  double u = 0.5 * div_with_rcp(z.im, t, 2.0 * h);
  u = real ? z.im : u;  // (+-0; also when t == 0)
  const bool right = !(z.re <= 0.0);  // re > 0, or NaN (which then stays in the real part, as in csqrt_)
  return {right ? t : fabs(u), right ? copysign(u, z.im) : copysign(t, z.im)};
}
// Principal complex cube root (= std::pow(z, 1/3.) of p3p.cpp:262,266 through its polar form) without the
// double-precision atan2 / sincos / cbrt (~330 VALU ops): a single-precision polar seed (relative error < 5e-6)
// refined by two Newton steps w <- (2w + z / w^2) / 3 in double (quadratic: 5e-6 -> 2.5e-11 -> rounding level).
// z is first scaled by a power of 8 into [1/8, 8) so that the float seed cannot overflow or flush.
// The seed costs ~25 instructions: |z|^(1/3) = exp2(log2(|z|^2) / 6) on the hardware log / exp units, the angle by
// a 5-term odd polynomial for atan on [0, 1] (|error| <= 1e-5 rad, Abramowitz & Stegun 4.4.47) with the usual octant
// fix-ups, sine / cosine on the hardware units (argument in revolutions).  A positive real z (std::pow's real branch:
// cbrt) takes the same path — seed angle 0, imaginary part 0 throughout — instead of a branch that a wave of 64
// hypotheses would nearly always have to walk as well; its imaginary part is returned as +0.
__device__ __forceinline__ C2 cpow_third_newton(C2 z) {
  const bool posreal = z.im == 0.0 && z.re > 0.0;
  const double m = fmax(fabs(z.re), fabs(z.im));
  if (m == 0.0) return {0.0, 0.0};  // pow(0, 1/3): rho = 0
  const int e = ilogb(m);                       // NaN / inf propagate through the arithmetic below
  const int k = (e >= 0 ? e : e - 2) / 3;       // floor(e / 3)
  const double a = ldexp(z.re, -3 * k), b = ldexp(z.im, -3 * k);
  const float af = (float)a, bf = (float)b;
  const float rho = p3p_exp2f(p3p_log2f(af * af + bf * bf) * (1.0f / 6.0f));
  const float ax = fabsf(af), ay = fabsf(bf);
  const float tq = fminf(ax, ay) * p3p_rcpf(fmaxf(ax, ay));
  const float t2 = tq * tq;
  float phi = fmaf(fmaf(fmaf(fmaf(0.0208351f, t2, -0.0851330f), t2, 0.1801410f), t2, -0.3302995f), t2, 0.9998660f) * tq;
  phi = ay > ax ? 1.57079632679f - phi : phi;
  phi = af < 0.0f ? 3.14159265359f - phi : phi;
  phi = copysignf(phi, bf);
  const float rev = phi * (float)(1.0 / (6.0 * 3.14159265358979323846));  // phi / 3 in revolutions
  double wr = (double)(rho * p3p_cos_rev(rev)), wi = (double)(rho * p3p_sin_rev(rev));
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const double sr = __builtin_fma(wr, wr, -(wi * wi)), si = (wr + wr) * wi;  // w^2
    const double n2 = __builtin_fma(sr, sr, si * si);
    double in = __builtin_amdgcn_rcp(n2);
    in = __builtin_fma(in, __builtin_fma(-n2, in, 1.0), in);
    if (it == 1) in = __builtin_fma(in, __builtin_fma(-n2, in, 1.0), in);  // (the first step only needs ~1e-11)
    const double tr = __builtin_fma(a, sr, b * si) * in, ti = __builtin_fma(b, sr, -(a * si)) * in;  // z / w^2
    wr = __builtin_fma(wr, 2.0 / 3.0, tr * (1.0 / 3.0));
    wi = __builtin_fma(wi, 2.0 / 3.0, ti * (1.0 / 3.0));
  }
  return {ldexp(wr, k), posreal ? 0.0 : ldexp(wi, k)};
}
struct NoService {
  __device__ __forceinline__ void operator()() const {}
};
// Cancellation below which a subtraction of Ferrari's method makes a hypothesis `suspect`: the result keeps less than
// 2^MPE_FERRARI_SUSPECT_EXP of its operands.  Measured on the quartics of 1.2 M C2 hypotheses (host build of this header,
// fast against strict arithmetic): the real parts of the roots differ by at most ~2e-16 / c where c is the smallest of
// the five ratios solve_quartic_lit2 looks at — and small ratios are COMMON (c < 1e-6 for 1.9 % of the hypotheses,
// < 1e-8 for 0.25 %, < 1e-10 for 0.04 %: the P3P quartic of a wrong correspondence often has nearly coinciding roots).
// With 2^-30 = 9.3e-10 (0.06 % of the hypotheses) the two arithmetics agree to ~2.4e-7 in every root that is not
// reported, i.e. to ~4e-4 px in a back-projection as far as cos(theta) itself goes; the amplification by 1 / sin(theta)
// and 1 / |(cn, cd)| is bounded root by root in the voting kernel from the exponent c returned here (K2_SUS_ROOT_BASE).
// The voting kernel hands reported hypotheses to the strict functions (k2_vote_fixup) instead of voting on them itself.
// The ratios are compared by their binary EXPONENTS only (integer arithmetic on the high words, the smallest difference
// kept in one vector register), each good to a factor of two.  Written with double-precision compares the levels cost
// pairs of scalar registers across the whole quartic plus a pair per threshold literal, and the scan-carrying voting
// kernel, which has no scalar register to spare, spilled hundreds of them to vector lanes.
#ifndef MPE_FERRARI_SUSPECT_EXP
#define MPE_FERRARI_SUSPECT_EXP (-30)
#endif
// biased binary exponent of |x| (0 for zero / subnormal, 2047 for inf / NaN)
__device__ __forceinline__ int p3p_expo(double x) { return (int)((__double2hiint(x) >> 20) & 0x7FF); }
__device__ __forceinline__ double cabs1(C2 z) { return fabs(z.re) + fabs(z.im); }
// `service` is called at two points inside (after the cube root, after w): the voting kernel's scan rider uses
// them to retire / start LDS-DMA rounds behind the arithmetic; a no-op everywhere else.
// `cancel_exp` (out): the binary exponent of the worst cancellation among the subtractions of p3p.cpp:253-283 (compare
// with MPE_FERRARI_SUSPECT_EXP) — the
// discriminant Q^2/4 + P^3/27, R = -Q/2 + sqrt(disc), w^2 = alpha + 2y (y with the magnitudes of ITS operands, so a
// cancellation inside y counts), the two outer radicands -(3 alpha + 2y +- 2 beta / w): the quantities
// tests/forensics.py classifies mismatching frames by.  NaN operands never compare true (such roots vote nowhere).
template <class Service>
__device__ __forceinline__ void solve_quartic_lit2(double A, double B, double C, double D, double E, double rr[4],
                                                   Service service, int& cancel_exp) {
  int minexp = 4096;  // smallest exponent(|result|) - exponent(sum of |operands|) seen (a NaN result: +, never reported)
  auto check = [&](const double result, const double operands) {
    minexp = min(minexp, p3p_expo(result) - p3p_expo(operands));
  };
  const double A_pw2 = A * A, B_pw2 = B * B;
  const double A_pw3 = A_pw2 * A, B_pw3 = B_pw2 * B;
  const double A_pw4 = A_pw3 * A, B_pw4 = B_pw3 * B;
  // one reciprocal of A serves all ten divisions by A, 2A^2 .. 256A^4 (the factors are powers of two: exact)
  const double r1 = rcp_nr(A), r2 = r1 * r1, r3 = r2 * r1, r4 = r2 * r2;
  const double alpha = div_with_rcp(-3 * B_pw2, 8 * A_pw2, 0.125 * r2) + div_with_rcp(C, A, r1);
  const double beta = div_with_rcp(B_pw3, 8 * A_pw3, 0.125 * r3) - div_with_rcp(B * C, 2 * A_pw2, 0.5 * r2) +
                      div_with_rcp(D, A, r1);
  const double gamma = div_with_rcp(-3 * B_pw4, 256 * A_pw4, 0.00390625 * r4) +
                       div_with_rcp(B_pw2 * C, 16 * A_pw3, 0.0625 * r3) - div_with_rcp(B * D, 4 * A_pw2, 0.25 * r2) +
                       div_with_rcp(E, A, r1);
  const double alpha_pw2 = alpha * alpha, alpha_pw3 = alpha_pw2 * alpha;
  const double Pr = div_const(-alpha_pw2, 12.0, 1.0 / 12.0) - gamma;
  const double Qr = div_const(-alpha_pw3, 108.0, 1.0 / 108.0) + div_const(alpha * gamma, 3.0, 1.0 / 3.0) -
                    (beta * beta) * 0.125;
  const C2 q2 = cpow2_real(Qr), p3 = cpow3_real(Pr);
  const C2 disc = {q2.re * 0.25 + div_const(p3.re, 27.0, 1.0 / 27.0), q2.im * 0.25 + div_const(p3.im, 27.0, 1.0 / 27.0)};
  check(fabs(disc.re), q2.re * 0.25 + fabs(p3.re) * (1.0 / 27.0));
  const C2 sq = csqrt_lit2(disc);
  const C2 R = {-Qr * 0.5 + sq.re, sq.im};
  check(cabs1(R), fabs(Qr) * 0.5 + cabs1(sq));
  const C2 U = cpow_third_newton(R);
  service();
  C2 y;
  double ysc;  // sum of the magnitudes of y's operands
  const double a56 = div_const(-5.0 * alpha, 6.0, 1.0 / 6.0);
  if (U.re == 0.0) {
    const C2 qc = cpow_third_newton(C2{Qr, 0.0});
    y = {a56 - qc.re, -qc.im};
    ysc = fabs(a56) + cabs1(qc);
  } else {
    const C2 t = cdiv_lit2(C2{Pr, 0.0}, cscale(U, 3.0));
    y = {a56 - t.re + U.re, -t.im + U.im};
    ysc = fabs(a56) + cabs1(t) + cabs1(U);
  }
  const C2 w2 = {alpha + 2.0 * y.re, 2.0 * y.im};
  check(cabs1(w2), fabs(alpha) + 2.0 * ysc);
  const C2 w = csqrt_lit2(w2);
  const C2 bw = cdiv_lit2(C2{2.0 * beta, 0.0}, w);
  service();
  const C2 base = {3.0 * alpha + 2.0 * y.re, 2.0 * y.im};
  const C2 rad1 = {-(base.re + bw.re), -(base.im + bw.im)}, rad2 = {-(base.re - bw.re), -(base.im - bw.im)};
  const double rsc = 3.0 * fabs(alpha) + 2.0 * ysc + cabs1(bw);
  check(cabs1(rad1), rsc);
  check(cabs1(rad2), rsc);
  cancel_exp = minexp;
  const C2 s1 = csqrt_lit2(rad1);
  const C2 s2 = csqrt_lit2(rad2);
  const double off = div_with_rcp(-B, 4.0 * A, 0.25 * r1);
  rr[0] = off + 0.5 * (w.re + s1.re);
  rr[1] = off + 0.5 * (w.re - s1.re);
  rr[2] = off + 0.5 * (-w.re + s2.re);
  rr[3] = off + 0.5 * (-w.re - s2.re);
}
template <class Service = NoService>
__device__ __forceinline__ void solve_quartic_lit2(double A, double B, double C, double D, double E, double rr[4],
                                                   Service service = Service()) {
  int cancel_exp;
  solve_quartic_lit2(A, B, C, D, E, rr, service, cancel_exp);
}

// Everything of computePoses that does not depend on the root index.  p3p.cpp:65-190
struct P3PCtx {
  M3 T, N;
  V3 P1;
  double f_1, f_2, p_1, p_2, d_12, b;
  double root[4];
};

// returns false iff the world points are exactly collinear (p3p.cpp:77-80)
// the glibc-faithful quartic as ONE out-of-line function: it is large (double-double log / exp / atan2 / sincos) and
// runs for the few hypotheses of the strict item only
static __device__ __noinline__ void solve_quartic_glibc(double A, double B, double C, double D, double E, double rr[4]) {
  solve_quartic<true>(A, B, C, D, E, rr);
}
// glibc_pow: the three complex powers of the quartic as libstdc++ / glibc evaluate them ("vote_arith" 3 / 4)
__device__ __forceinline__ bool p3p_prepare(const V3& fa, const V3& fb, const V3& fc, const V3& wa, const V3& wb,
                                            const V3& wc, P3PCtx& c, const bool glibc_pow = false) {
  V3 P1 = wa, P2 = wb, P3 = wc;
  if (norm(cross(P2 - P1, P3 - P1)) == 0.0) return false;
  V3 f1 = fa, f2 = fb;
  V3 e1 = f1;
  V3 e3 = cross(f1, f2);
  e3 = vdiv(e3, norm(e3));
  V3 e2 = cross(e3, e1);
  M3 T = {e1, e2, e3};
  V3 f3 = mul(T, fc);
  if (f3.z > 0.0) {  // p3p.cpp:100-121: swap the first two correspondences
    f1 = fb;
    f2 = fa;
    e1 = f1;
    e3 = cross(f1, f2);
    e3 = vdiv(e3, norm(e3));
    e2 = cross(e3, e1);
    T = {e1, e2, e3};
    f3 = mul(T, fc);
    P1 = wb;
    P2 = wa;
  }
  V3 n1 = P2 - P1;
  n1 = vdiv(n1, norm(n1));
  V3 n3 = cross(n1, P3 - P1);
  n3 = vdiv(n3, norm(n3));
  V3 n2 = cross(n3, n1);
  M3 N = {n1, n2, n3};
  V3 P3n = mul(N, P3 - P1);
  double d_12 = norm(P2 - P1);
  double f_1 = f3.x / f3.z, f_2 = f3.y / f3.z;
  double p_1 = P3n.x, p_2 = P3n.y;
  double cos_beta = dot(f1, f2);
  double b = 1 / (1 - cos_beta * cos_beta) - 1;
  b = (cos_beta < 0) ? -sqrt(b) : sqrt(b);

  double f_1_pw2 = f_1 * f_1, f_2_pw2 = f_2 * f_2;
  double p_1_pw2 = p_1 * p_1, p_1_pw3 = p_1_pw2 * p_1, p_1_pw4 = p_1_pw3 * p_1;
  double p_2_pw2 = p_2 * p_2, p_2_pw3 = p_2_pw2 * p_2, p_2_pw4 = p_2_pw3 * p_2;
  double d_12_pw2 = d_12 * d_12, b_pw2 = b * b;

  double F0 = -f_2_pw2 * p_2_pw4 - p_2_pw4 * f_1_pw2 - p_2_pw4;
  double F1 = 2 * p_2_pw3 * d_12 * b + 2 * f_2_pw2 * p_2_pw3 * d_12 * b - 2 * f_2 * p_2_pw3 * f_1 * d_12;
  double F2 = -f_2_pw2 * p_2_pw2 * p_1_pw2 - f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2 - f_2_pw2 * p_2_pw2 * d_12_pw2 +
              f_2_pw2 * p_2_pw4 + p_2_pw4 * f_1_pw2 + 2 * p_1 * p_2_pw2 * d_12 +
              2 * f_1 * f_2 * p_1 * p_2_pw2 * d_12 * b - p_2_pw2 * p_1_pw2 * f_1_pw2 +
              2 * p_1 * p_2_pw2 * f_2_pw2 * d_12 - p_2_pw2 * d_12_pw2 * b_pw2 - 2 * p_1_pw2 * p_2_pw2;
  double F3 = 2 * p_1_pw2 * p_2 * d_12 * b + 2 * f_2 * p_2_pw3 * f_1 * d_12 - 2 * f_2_pw2 * p_2_pw3 * d_12 * b -
              2 * p_1 * p_2 * d_12_pw2 * b;
  double F4 = -2 * f_2 * p_2_pw2 * f_1 * p_1 * d_12 * b + f_2_pw2 * p_2_pw2 * d_12_pw2 + 2 * p_1_pw3 * d_12 -
              p_1_pw2 * d_12_pw2 + f_2_pw2 * p_2_pw2 * p_1_pw2 - p_1_pw4 - 2 * f_2_pw2 * p_2_pw2 * p_1 * d_12 +
              p_2_pw2 * f_1_pw2 * p_1_pw2 + f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2;
  if (glibc_pow)
    solve_quartic_glibc(F0, F1, F2, F3, F4, c.root);
  else
    solve_quartic(F0, F1, F2, F3, F4, c.root);
  c.T = T;
  c.N = N;
  c.P1 = P1;
  c.f_1 = f_1;
  c.f_2 = f_2;
  c.p_1 = p_1;
  c.p_2 = p_2;
  c.d_12 = d_12;
  c.b = b;
  return true;
}

// Back-substitution of one root: R (camera -> marker frame) and C (camera centre in the marker
// frame).  p3p.cpp:193-233
__device__ __forceinline__ void p3p_solution(const P3PCtx& c, double root, M3& R, V3& C) {
  double cot_alpha = (-c.f_1 * c.p_1 / c.f_2 - root * c.p_2 + c.d_12 * c.b) /
                     (-c.f_1 * root * c.p_2 / c.f_2 + c.p_1 - c.d_12);
  double cos_theta = root;
  double sin_theta = sqrt(1 - root * root);
  double sin_alpha = sqrt(1 / (cot_alpha * cot_alpha + 1));
  double cos_alpha = sqrt(1 - sin_alpha * sin_alpha);
  if (cot_alpha < 0) cos_alpha = -cos_alpha;
  double k = sin_alpha * c.b + cos_alpha;
  V3 Cn = {c.d_12 * cos_alpha * k, cos_theta * c.d_12 * sin_alpha * k, sin_theta * c.d_12 * sin_alpha * k};
  C = c.P1 + mulT(c.N, Cn);
  // Rt = transpose of the matrix written at p3p.cpp:215-224
  M3 Rt = {{-cos_alpha, sin_alpha, 0.0},
           {-sin_alpha * cos_theta, -cos_alpha * cos_theta, -sin_theta},
           {-sin_alpha * sin_theta, -cos_alpha * sin_theta, cos_theta}};
  R = mul(mul(transpose(c.N), Rt), c.T);
}

// isFinite([R C; 0 0 0 1]) — pose_estimator.cpp:856-860
__device__ __forceinline__ bool rc_finite(const M3& R, const V3& C) {
  double z = (R.r0.x - R.r0.x) + (R.r0.y - R.r0.y) + (R.r0.z - R.r0.z) + (R.r1.x - R.r1.x) + (R.r1.y - R.r1.y) +
             (R.r1.z - R.r1.z) + (R.r2.x - R.r2.x) + (R.r2.y - R.r2.y) + (R.r2.z - R.r2.z) + (C.x - C.x) +
             (C.y - C.y) + (C.z - C.z);
  return z == 0.0;
}

// Pinhole projection matrix of the INVERSE of H = [R C; 0 1]:  M = K [I|0] H^-1  (3x4).
// The reference calls the general 4x4 inverse on this rigid transform (pose_estimator.cpp:660);
// the rigid closed form [R^T | -R^T C] is the same matrix up to rounding of an orthonormal R.
struct Proj {
  V3 r0, r1, r2;    // rows of the 3x3 part
  double t0, t1, t2;  // fourth column
};
__device__ __forceinline__ Proj make_projection(const M3& R, const V3& C, double fx, double fy, double cx, double cy) {
  M3 Ri = transpose(R);
  V3 ti = mul(Ri, C);
  ti = {-ti.x, -ti.y, -ti.z};
  Proj P;
  P.r0 = {fx * Ri.r0.x + cx * Ri.r2.x, fx * Ri.r0.y + cx * Ri.r2.y, fx * Ri.r0.z + cx * Ri.r2.z};
  P.r1 = {fy * Ri.r1.x + cy * Ri.r2.x, fy * Ri.r1.y + cy * Ri.r2.y, fy * Ri.r1.z + cy * Ri.r2.z};
  P.r2 = Ri.r2;
  P.t0 = fx * ti.x + cx * ti.z;
  P.t1 = fy * ti.y + cy * ti.z;
  P.t2 = ti.z;
  return P;
}
__device__ __forceinline__ void project(const Proj& P, const V3& m, double& u, double& v) {
  double x = dot(P.r0, m) + P.t0, y = dot(P.r1, m) + P.t1, z = dot(P.r2, m) + P.t2;
  u = x / z;
  v = y / z;
}

}  // namespace mpe
