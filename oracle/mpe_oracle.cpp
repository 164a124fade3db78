/*
 * mpe_oracle.cpp — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See mpe_oracle.h.
 *
 * A from-scratch, dependency-free (no Eigen / OpenCV / ROS) restatement of the reference's
 * per-frame hot path.  Every function cites the reference file:line it follows
 * (aliases: PE = monocular_pose_estimator_lib/src/pose_estimator.cpp, LED = .../led_detector.cpp,
 * P3P = .../p3p.cpp, COMB = .../combinations.cpp).  OpenCV / Eigen semantics the reference
 * relies on are restated from their documented behaviour — PARITY
 * UNPINNED: no reference test or binary exists to pin them.
 */
#include "mpe_oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <complex>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------
// tiny fixed-size linear algebra (row-major)
// ---------------------------------------------------------------------------------------------
struct V3 {
  double x, y, z;
};
inline V3 operator-(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator+(const V3& a, const V3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator/(const V3& a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 cross(const V3& a, const V3& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double norm(const V3& a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }

struct M3 {
  double m[3][3];
};
inline V3 mul(const M3& A, const V3& v) {
  return {A.m[0][0] * v.x + A.m[0][1] * v.y + A.m[0][2] * v.z,
          A.m[1][0] * v.x + A.m[1][1] * v.y + A.m[1][2] * v.z,
          A.m[2][0] * v.x + A.m[2][1] * v.y + A.m[2][2] * v.z};
}
inline M3 mul(const M3& A, const M3& B) {
  M3 C;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = A.m[i][0] * B.m[0][j];
      s += A.m[i][1] * B.m[1][j];
      s += A.m[i][2] * B.m[2][j];
      C.m[i][j] = s;
    }
  return C;
}
inline M3 transpose(const M3& A) {
  M3 T;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T.m[i][j] = A.m[j][i];
  return T;
}
inline M3 rows(const V3& a, const V3& b, const V3& c) {
  return {{{a.x, a.y, a.z}, {b.x, b.y, b.z}, {c.x, c.y, c.z}}};
}

struct M4 {
  double m[4][4];
};
inline M4 identity4() {
  M4 I;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) I.m[i][j] = (i == j) ? 1.0 : 0.0;
  return I;
}
inline M4 mul(const M4& A, const M4& B) {
  M4 C;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = A.m[i][0] * B.m[0][j];
      for (int k = 1; k < 4; ++k) s += A.m[i][k] * B.m[k][j];
      C.m[i][j] = s;
    }
  return C;
}

// General 4x4 inverse (adjugate / determinant), standing in for Eigen's Matrix4d::inverse()
// used on rigid transforms at PE.cpp:486,516,660.
M4 inverse4(const M4& A) {
  const double(*a)[4] = A.m;
  double s0 = a[0][0] * a[1][1] - a[1][0] * a[0][1];
  double s1 = a[0][0] * a[1][2] - a[1][0] * a[0][2];
  double s2 = a[0][0] * a[1][3] - a[1][0] * a[0][3];
  double s3 = a[0][1] * a[1][2] - a[1][1] * a[0][2];
  double s4 = a[0][1] * a[1][3] - a[1][1] * a[0][3];
  double s5 = a[0][2] * a[1][3] - a[1][2] * a[0][3];
  double c5 = a[2][2] * a[3][3] - a[3][2] * a[2][3];
  double c4 = a[2][1] * a[3][3] - a[3][1] * a[2][3];
  double c3 = a[2][1] * a[3][2] - a[3][1] * a[2][2];
  double c2 = a[2][0] * a[3][3] - a[3][0] * a[2][3];
  double c1 = a[2][0] * a[3][2] - a[3][0] * a[2][2];
  double c0 = a[2][0] * a[3][1] - a[3][0] * a[2][1];
  double det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
  double id = 1.0 / det;
  M4 B;
  B.m[0][0] = (a[1][1] * c5 - a[1][2] * c4 + a[1][3] * c3) * id;
  B.m[0][1] = (-a[0][1] * c5 + a[0][2] * c4 - a[0][3] * c3) * id;
  B.m[0][2] = (a[3][1] * s5 - a[3][2] * s4 + a[3][3] * s3) * id;
  B.m[0][3] = (-a[2][1] * s5 + a[2][2] * s4 - a[2][3] * s3) * id;
  B.m[1][0] = (-a[1][0] * c5 + a[1][2] * c2 - a[1][3] * c1) * id;
  B.m[1][1] = (a[0][0] * c5 - a[0][2] * c2 + a[0][3] * c1) * id;
  B.m[1][2] = (-a[3][0] * s5 + a[3][2] * s2 - a[3][3] * s1) * id;
  B.m[1][3] = (a[2][0] * s5 - a[2][2] * s2 + a[2][3] * s1) * id;
  B.m[2][0] = (a[1][0] * c4 - a[1][1] * c2 + a[1][3] * c0) * id;
  B.m[2][1] = (-a[0][0] * c4 + a[0][1] * c2 - a[0][3] * c0) * id;
  B.m[2][2] = (a[3][0] * s4 - a[3][1] * s2 + a[3][3] * s0) * id;
  B.m[2][3] = (-a[2][0] * s4 + a[2][1] * s2 - a[2][3] * s0) * id;
  B.m[3][0] = (-a[1][0] * c3 + a[1][1] * c1 - a[1][2] * c0) * id;
  B.m[3][1] = (a[0][0] * c3 - a[0][1] * c1 + a[0][2] * c0) * id;
  B.m[3][2] = (-a[3][0] * s3 + a[3][1] * s1 - a[3][2] * s0) * id;
  B.m[3][3] = (a[2][0] * s3 - a[2][1] * s1 + a[2][2] * s0) * id;
  return B;
}

// ((x - x) == (x - x)).all()  — PE.cpp:856-860
bool is_finite4(const M4& A) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double d = A.m[i][j] - A.m[i][j];
      if (!(d == d)) return false;
    }
  return true;
}

// 6x6 symmetric solve with diagonally pivoted LDL^T (stands in for Eigen A.ldlt().solve(b),
// PE.cpp:778).
void ldlt_solve6(const double Ain[36], const double bin[6], double x[6]) {
  double A[6][6];
  int perm[6];
  for (int i = 0; i < 6; ++i) {
    perm[i] = i;
    for (int j = 0; j < 6; ++j) A[i][j] = Ain[i * 6 + j];
  }
  double L[6][6] = {{0}};
  double Dg[6];
  for (int k = 0; k < 6; ++k) {
    // pivot: largest |diagonal| of the trailing block
    int piv = k;
    double best = std::fabs(A[k][k]);
    for (int i = k + 1; i < 6; ++i)
      if (std::fabs(A[i][i]) > best) {
        best = std::fabs(A[i][i]);
        piv = i;
      }
    if (piv != k) {
      for (int j = 0; j < 6; ++j) std::swap(A[k][j], A[piv][j]);
      for (int i = 0; i < 6; ++i) std::swap(A[i][k], A[i][piv]);
      for (int j = 0; j < k; ++j) std::swap(L[k][j], L[piv][j]);
      std::swap(perm[k], perm[piv]);
    }
    Dg[k] = A[k][k];
    L[k][k] = 1.0;
    for (int i = k + 1; i < 6; ++i) L[i][k] = A[i][k] / Dg[k];
    for (int i = k + 1; i < 6; ++i)
      for (int j = k + 1; j < 6; ++j) A[i][j] -= L[i][k] * Dg[k] * L[j][k];
  }
  double y[6], z[6];
  for (int i = 0; i < 6; ++i) {
    double s = bin[perm[i]];
    for (int j = 0; j < i; ++j) s -= L[i][j] * y[j];
    y[i] = s;
  }
  for (int i = 0; i < 6; ++i) y[i] /= Dg[i];
  for (int i = 5; i >= 0; --i) {
    double s = y[i];
    for (int j = i + 1; j < 6; ++j) s -= L[j][i] * z[j];
    z[i] = s;
  }
  for (int i = 0; i < 6; ++i) x[perm[i]] = z[i];
}

// 6x6 general inverse, Gauss-Jordan with partial pivoting (stands in for A.inverse(), PE.cpp:790)
void inverse6(const double Ain[36], double out[36]) {
  double A[6][12];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      A[i][j] = Ain[i * 6 + j];
      A[i][6 + j] = (i == j) ? 1.0 : 0.0;
    }
  for (int k = 0; k < 6; ++k) {
    int piv = k;
    for (int i = k + 1; i < 6; ++i)
      if (std::fabs(A[i][k]) > std::fabs(A[piv][k])) piv = i;
    if (piv != k)
      for (int j = 0; j < 12; ++j) std::swap(A[k][j], A[piv][j]);
    double d = A[k][k];
    for (int j = 0; j < 12; ++j) A[k][j] /= d;
    for (int i = 0; i < 6; ++i) {
      if (i == k) continue;
      double f = A[i][k];
      if (f == 0.0) continue;
      for (int j = 0; j < 12; ++j) A[i][j] -= f * A[k][j];
    }
  }
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) out[i * 6 + j] = A[i][6 + j];
}

// One-sided (Hestenes) Jacobi SVD of a 3x3: H = U diag(s) V^T.  Stands in for
// Eigen::JacobiSVD(ThinU|ThinV) at PE.cpp:916-920; only V*U^T (the orthogonal polar factor,
// unique for full-rank H) is consumed.
// Rank-deficient H (coplanar markers: H = A B^T with a rank-2 A): the third singular value is a
// rounding residue (~1e-17 relative) and Eigen's JacobiSVD gives U.col(2) the SIGN of that residue
// (JacobiSVD.h "m_matrixU.col(i) *= m_workMatrix(i,i) / a"), so whether the reference returns the
// proper rotation or its mirror image through the marker plane is decided by the last bit of H —
// two builds of the reference do not agree with each other there.  The restatement (and the HIP
// tail kernel, same algorithm) resolves every sigma_3 <= 1e-12 sigma_1 to the completion
// u_3 = u_1 x u_2, i.e. det(U) = +1: one of the reference's two outcomes, deterministically.
void svd3(const M3& H, M3& U, double s[3], M3& V) {
  double G[3][3], Vm[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      G[i][j] = H.m[i][j];
      Vm[i][j] = (i == j) ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 60; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < 3; ++i) {
          alpha += G[i][p] * G[i][p];
          beta += G[i][q] * G[i][q];
          gamma += G[i][p] * G[i][q];
        }
        if (gamma == 0.0 || std::fabs(gamma) <= 1e-300 + 2.3e-16 * std::sqrt(alpha * beta)) continue;
        rotated = true;
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
        for (int i = 0; i < 3; ++i) {
          double gp = G[i][p], gq = G[i][q];
          G[i][p] = c * gp - sn * gq;
          G[i][q] = sn * gp + c * gq;
          double vp = Vm[i][p], vq = Vm[i][q];
          Vm[i][p] = c * vp - sn * vq;
          Vm[i][q] = sn * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  int zero_col = -1, nzero = 0;
  for (int j = 0; j < 3; ++j) {
    double n2 = G[0][j] * G[0][j] + G[1][j] * G[1][j] + G[2][j] * G[2][j];
    s[j] = std::sqrt(n2);
  }
  double smax = std::max(s[0], std::max(s[1], s[2]));
  for (int j = 0; j < 3; ++j) {
    if (s[j] > smax * 1e-12 && s[j] > 0) {
      for (int i = 0; i < 3; ++i) U.m[i][j] = G[i][j] / s[j];
    } else {
      zero_col = j;
      ++nzero;
    }
  }
  if (nzero == 1) {  // rank 2: complete U with the cross product of the other two columns
    int a = (zero_col + 1) % 3, b = (zero_col + 2) % 3;
    V3 ua = {U.m[0][a], U.m[1][a], U.m[2][a]}, ub = {U.m[0][b], U.m[1][b], U.m[2][b]};
    V3 uc = cross(ua, ub);
    U.m[0][zero_col] = uc.x;
    U.m[1][zero_col] = uc.y;
    U.m[2][zero_col] = uc.z;
  } else if (nzero > 1) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) U.m[i][j] = (i == j) ? 1.0 : 0.0;
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) V.m[i][j] = Vm[i][j];
}

// ---------------------------------------------------------------------------------------------
// Combinations — COMB.cpp
// ---------------------------------------------------------------------------------------------
unsigned factorial_u32(int N) {  // COMB.cpp:34-40 (unsigned 32-bit wrap-around kept, quirk A.6.9)
  if (N == 1 || N == 0) return 1u;
  return factorial_u32(N - 1) * (unsigned)N;
}

// ---------------------------------------------------------------------------------------------
// P3P — P3P.cpp
// ---------------------------------------------------------------------------------------------
typedef std::complex<double> cplx;

int solve_quartic(const double f[5], double rr[4]) {  // P3P.cpp:238-286
  double A = f[0], B = f[1], C = f[2], D = f[3], E = f[4];
  double A_pw2 = A * A, B_pw2 = B * B;
  double A_pw3 = A_pw2 * A, B_pw3 = B_pw2 * B;
  double A_pw4 = A_pw3 * A, B_pw4 = B_pw3 * B;
  double alpha = -3 * B_pw2 / (8 * A_pw2) + C / A;
  double beta = B_pw3 / (8 * A_pw3) - B * C / (2 * A_pw2) + D / A;
  double gamma = -3 * B_pw4 / (256 * A_pw4) + B_pw2 * C / (16 * A_pw3) - B * D / (4 * A_pw2) + E / A;
  double alpha_pw2 = alpha * alpha, alpha_pw3 = alpha_pw2 * alpha;

  cplx P(-alpha_pw2 / 12 - gamma, 0);
  cplx Q(-alpha_pw3 / 108 + alpha * gamma / 3 - std::pow(beta, 2) / 8, 0);
  cplx R = -Q / 2.0 + std::sqrt(std::pow(Q, 2.0) / 4.0 + std::pow(P, 3.0) / 27.0);
  cplx U = std::pow(R, (1.0 / 3.0));
  cplx y;
  if (U.real() == 0)
    y = -5.0 * alpha / 6.0 - std::pow(Q, (1.0 / 3.0));
  else
    y = -5.0 * alpha / 6.0 - P / (3.0 * U) + U;
  cplx w = std::sqrt(alpha + 2.0 * y);
  cplx temp;
  temp = -B / (4.0 * A) + 0.5 * (w + std::sqrt(-(3.0 * alpha + 2.0 * y + 2.0 * beta / w)));
  rr[0] = temp.real();
  temp = -B / (4.0 * A) + 0.5 * (w - std::sqrt(-(3.0 * alpha + 2.0 * y + 2.0 * beta / w)));
  rr[1] = temp.real();
  temp = -B / (4.0 * A) + 0.5 * (-w + std::sqrt(-(3.0 * alpha + 2.0 * y - 2.0 * beta / w)));
  rr[2] = temp.real();
  temp = -B / (4.0 * A) + 0.5 * (-w - std::sqrt(-(3.0 * alpha + 2.0 * y - 2.0 * beta / w)));
  rr[3] = temp.real();
  return 0;
}

// fv[i], wp[i]: i-th bearing / world point.  sol[s] = 3x4 [R|C].   P3P.cpp:65-236
int p3p_compute(const V3 fv[3], const V3 wp[3], double sol[4][3][4]) {
  V3 P1 = wp[0], P2 = wp[1], P3 = wp[2];
  V3 temp1 = P2 - P1, temp2 = P3 - P1;
  if (norm(cross(temp1, temp2)) == 0) return -1;  // P3P.cpp:77-80

  V3 f1 = fv[0], f2 = fv[1], f3 = fv[2];
  V3 e1 = f1;
  V3 e3 = cross(f1, f2);
  e3 = e3 / norm(e3);
  V3 e2 = cross(e3, e1);
  M3 T = rows(e1, e2, e3);
  f3 = mul(T, f3);

  if (f3.z > 0) {  // P3P.cpp:100-121
    f1 = fv[1];
    f2 = fv[0];
    f3 = fv[2];
    e1 = f1;
    e3 = cross(f1, f2);
    e3 = e3 / norm(e3);
    e2 = cross(e3, e1);
    T = rows(e1, e2, e3);
    f3 = mul(T, f3);
    P1 = wp[1];
    P2 = wp[0];
    P3 = wp[2];
  }

  V3 n1 = P2 - P1;
  n1 = n1 / norm(n1);
  V3 n3 = cross(n1, P3 - P1);
  n3 = n3 / norm(n3);
  V3 n2 = cross(n3, n1);
  M3 N = rows(n1, n2, n3);

  P3 = mul(N, P3 - P1);
  double d_12 = norm(P2 - P1);
  double f_1 = f3.x / f3.z;
  double f_2 = f3.y / f3.z;
  double p_1 = P3.x;
  double p_2 = P3.y;

  double cos_beta = dot(f1, f2);
  double b = 1 / (1 - std::pow(cos_beta, 2)) - 1;
  if (cos_beta < 0)
    b = -std::sqrt(b);
  else
    b = std::sqrt(b);

  double f_1_pw2 = std::pow(f_1, 2);
  double f_2_pw2 = std::pow(f_2, 2);
  double p_1_pw2 = std::pow(p_1, 2);
  double p_1_pw3 = p_1_pw2 * p_1;
  double p_1_pw4 = p_1_pw3 * p_1;
  double p_2_pw2 = std::pow(p_2, 2);
  double p_2_pw3 = p_2_pw2 * p_2;
  double p_2_pw4 = p_2_pw3 * p_2;
  double d_12_pw2 = std::pow(d_12, 2);
  double b_pw2 = std::pow(b, 2);

  double factors[5];  // P3P.cpp:171-185
  factors[0] = -f_2_pw2 * p_2_pw4 - p_2_pw4 * f_1_pw2 - p_2_pw4;
  factors[1] = 2 * p_2_pw3 * d_12 * b + 2 * f_2_pw2 * p_2_pw3 * d_12 * b - 2 * f_2 * p_2_pw3 * f_1 * d_12;
  factors[2] = -f_2_pw2 * p_2_pw2 * p_1_pw2 - f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2 -
               f_2_pw2 * p_2_pw2 * d_12_pw2 + f_2_pw2 * p_2_pw4 + p_2_pw4 * f_1_pw2 +
               2 * p_1 * p_2_pw2 * d_12 + 2 * f_1 * f_2 * p_1 * p_2_pw2 * d_12 * b -
               p_2_pw2 * p_1_pw2 * f_1_pw2 + 2 * p_1 * p_2_pw2 * f_2_pw2 * d_12 -
               p_2_pw2 * d_12_pw2 * b_pw2 - 2 * p_1_pw2 * p_2_pw2;
  factors[3] = 2 * p_1_pw2 * p_2 * d_12 * b + 2 * f_2 * p_2_pw3 * f_1 * d_12 -
               2 * f_2_pw2 * p_2_pw3 * d_12 * b - 2 * p_1 * p_2 * d_12_pw2 * b;
  factors[4] = -2 * f_2 * p_2_pw2 * f_1 * p_1 * d_12 * b + f_2_pw2 * p_2_pw2 * d_12_pw2 +
               2 * p_1_pw3 * d_12 - p_1_pw2 * d_12_pw2 + f_2_pw2 * p_2_pw2 * p_1_pw2 - p_1_pw4 -
               2 * f_2_pw2 * p_2_pw2 * p_1 * d_12 + p_2_pw2 * f_1_pw2 * p_1_pw2 +
               f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2;

  double realRoots[4];
  solve_quartic(factors, realRoots);

  M3 Nt = transpose(N);
  for (int i = 0; i < 4; ++i) {  // P3P.cpp:193-233
    double cot_alpha = (-f_1 * p_1 / f_2 - realRoots[i] * p_2 + d_12 * b) /
                       (-f_1 * realRoots[i] * p_2 / f_2 + p_1 - d_12);
    double cos_theta = realRoots[i];
    double sin_theta = std::sqrt(1 - std::pow((double)realRoots[i], 2));
    double sin_alpha = std::sqrt(1 / (std::pow(cot_alpha, 2) + 1));
    double cos_alpha = std::sqrt(1 - std::pow(sin_alpha, 2));
    if (cot_alpha < 0) cos_alpha = -cos_alpha;

    V3 C;
    C.x = d_12 * cos_alpha * (sin_alpha * b + cos_alpha);
    C.y = cos_theta * d_12 * sin_alpha * (sin_alpha * b + cos_alpha);
    C.z = sin_theta * d_12 * sin_alpha * (sin_alpha * b + cos_alpha);
    C = P1 + mul(Nt, C);

    M3 R;
    R.m[0][0] = -cos_alpha;
    R.m[0][1] = -sin_alpha * cos_theta;
    R.m[0][2] = -sin_alpha * sin_theta;
    R.m[1][0] = sin_alpha;
    R.m[1][1] = -cos_alpha * cos_theta;
    R.m[1][2] = -cos_alpha * sin_theta;
    R.m[2][0] = 0;
    R.m[2][1] = -sin_theta;
    R.m[2][2] = cos_theta;
    R = mul(mul(Nt, transpose(R)), T);

    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) sol[i][r][c] = R.m[r][c];
    }
    sol[i][0][3] = C.x;
    sol[i][1][3] = C.y;
    sol[i][2][3] = C.z;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// LED detector — LED.cpp:35-112 with the OpenCV semantics of SURVEY.md §A.1
// ---------------------------------------------------------------------------------------------
inline int cv_round(double v) { return (int)std::lrint(v); }  // round-half-to-even

inline int reflect101(int p, int len) {  // cv::borderInterpolate(BORDER_REFLECT_101)
  if ((unsigned)p < (unsigned)len) return p;
  if (len == 1) return 0;
  do {
    if (p < 0)
      p = -p;
    else
      p = 2 * (len - 1) - p;
  } while ((unsigned)p >= (unsigned)len);
  return p;
}

// cv::getGaussianKernel(n, sigma, CV_32F) followed by convertTo(CV_32S, 256) as done by
// createSeparableLinearFilter for an 8U->8U smooth symmetric kernel.
int gaussian_kernel_q8(double sigma, std::vector<int>& taps) {
  if (!(sigma > 0)) return -1;  // OpenCV asserts ksize > 0 when sigma <= 0 and ksize == 0
  int n = cv_round(sigma * 3 * 2 + 1) | 1;
  std::vector<float> cf(n);
  double scale2x = -0.5 / (sigma * sigma);
  double sum = 0;
  for (int i = 0; i < n; ++i) {
    double x = i - (n - 1) * 0.5;
    double t = std::exp(scale2x * x * x);
    cf[i] = (float)t;
    sum += cf[i];
  }
  sum = 1. / sum;
  taps.resize(n);
  for (int i = 0; i < n; ++i) {
    cf[i] = (float)(cf[i] * sum);
    taps[i] = cv_round((double)cf[i] * 256.0);
  }
  return n;
}

// threshold(TOZERO) + GaussianBlur(ksize=0, sigma) on the ROI as a stand-alone matrix.
// blurred: roi_h x roi_w.   LED.cpp:44-51
int blur_roi(const uint8_t* img, size_t stride, int rx, int ry, int rw, int rh, int thr,
             double sigma, std::vector<uint8_t>& blurred) {
  std::vector<int> taps;
  int n = gaussian_kernel_q8(sigma, taps);
  if (n < 0) return -1;
  const int r = n / 2;
  blurred.assign((size_t)rw * rh, 0);
  if (rw <= 0 || rh <= 0) return 0;
  // thresholded ROI, padded by r with REFLECT_101 (the ROI clone is a stand-alone matrix)
  const int pw = rw + 2 * r, ph = rh + 2 * r;
  // scratch buffers are thread_local and only grow: a frame-parallel run (orc_estimate_batch) then
  // does not mmap/munmap megabytes per frame, which would serialise the threads in the kernel
  static thread_local std::vector<uint8_t> pad;
  static thread_local std::vector<int> rowf, acc;
  if (pad.size() < (size_t)pw * ph) pad.resize((size_t)pw * ph);
  for (int y = 0; y < ph; ++y) {
    int sy = reflect101(y - r, rh);
    const uint8_t* src = img + (size_t)(ry + sy) * stride + rx;
    uint8_t* dst = &pad[(size_t)y * pw];
    for (int x = 0; x < rw; ++x) {
      uint8_t v = src[x];
      dst[x + r] = (v > thr) ? v : 0;  // THRESH_TOZERO: strictly greater
    }
    for (int x = 0; x < r; ++x) {
      dst[x] = dst[r + reflect101(x - r, rw)];
      dst[r + rw + x] = dst[r + reflect101(rw + x, rw)];
    }
  }
  if (n == 1) {  // GaussianBlur copies when the kernel is 1x1
    for (int y = 0; y < rh; ++y)
      std::memcpy(&blurred[(size_t)y * rw], &pad[(size_t)(y + r) * pw + r], rw);
    return 0;
  }
  if (rowf.size() < (size_t)rw * ph) rowf.resize((size_t)rw * ph);
  for (int y = 0; y < ph; ++y) {
    const uint8_t* s = &pad[(size_t)y * pw];
    int* d = &rowf[(size_t)y * rw];
    for (int x = 0; x < rw; ++x) d[x] = 0;
    for (int j = 0; j < n; ++j) {
      const int k = taps[j];
      const uint8_t* sj = s + j;
      for (int x = 0; x < rw; ++x) d[x] += k * sj[x];
    }
  }
  if (acc.size() < (size_t)rw) acc.resize(rw);
  for (int y = 0; y < rh; ++y) {
    for (int x = 0; x < rw; ++x) acc[x] = 0;
    for (int i = 0; i < n; ++i) {
      const int k = taps[i];
      const int* s = &rowf[(size_t)(y + i) * rw];
      for (int x = 0; x < rw; ++x) acc[x] += k * s[x];
    }
    uint8_t* d = &blurred[(size_t)y * rw];
    for (int x = 0; x < rw; ++x) {
      int v = (acc[x] + (1 << 15)) >> 16;  // FixedPtCastEx<int, uchar>(16)
      d[x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  }
  return 0;
}

struct Pt {
  int x, y;
};

// icvFetchContour (outer border, CHAIN_APPROX_NONE): Suzuki-Abe border following with OpenCV's
// marks: +2 for visited border pixels, -126 (2|-128) when the east neighbour was examined and 0.
void fetch_outer_border(signed char* i0, int step, Pt pt, std::vector<Pt>& c) {
  const signed char nbd = 2;
  const int deltas[16] = {1, -step + 1, -step, -step - 1, -1, step - 1, step, step + 1,
                          1, -step + 1, -step, -step - 1, -1, step - 1, step, step + 1};
  static const int dx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
  static const int dy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
  signed char *i1, *i3, *i4 = nullptr;
  int s, s_end;
  s_end = s = 4;
  do {
    s = (s - 1) & 7;
    i1 = i0 + deltas[s];
  } while (*i1 == 0 && s != s_end);
  if (s == s_end) {  // single pixel
    *i0 = (signed char)(nbd | -128);
    c.push_back(pt);
    return;
  }
  i3 = i0;
  for (;;) {
    s_end = s;
    while (s < 15) {
      i4 = i3 + deltas[++s];
      if (*i4 != 0) break;
    }
    s &= 7;
    if ((unsigned)(s - 1) < (unsigned)s_end)
      *i3 = (signed char)(nbd | -128);
    else if (*i3 == 1)
      *i3 = nbd;
    c.push_back(pt);
    pt.x += dx[s];
    pt.y += dy[s];
    if (i4 == i0 && i3 == i1) break;
    i3 = i4;
    s = (s + 4) & 7;
  }
}

// cv::findContours(RETR_EXTERNAL, CHAIN_APPROX_NONE) on a mask (nonzero = foreground), image
// zero-padded by one pixel; raster scan of cvFindNextContour with mode 0.  Result order =
// OpenCV's: newest contour first (each new contour is linked at the head of the list).
void external_contours(const uint8_t* mask, int h, int w, std::vector<std::vector<Pt>>& out) {
  out.clear();
  if (h <= 0 || w <= 0) return;
  const int W = w + 2, H = h + 2;
  static thread_local std::vector<signed char> im;
  im.assign((size_t)W * H, 0);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) im[(size_t)(y + 1) * W + x + 1] = mask[(size_t)y * w + x] ? 1 : 0;
  std::vector<std::vector<Pt>> found;
  for (int y = 1; y < H - 1; ++y) {
    signed char* img = &im[(size_t)y * W];
    int lnbd_x = 0;
    int prev = 0;
    for (int x = 1; x < W - 1; ++x) {
      int p = img[x];
      if (p == prev) continue;
      bool start = false;
      if (!(prev == 0 && p == 1)) {
        if (p != 0 || prev < 1) goto resume_scan;
        if (prev & -2) lnbd_x = x - 1;
        goto resume_scan;  // hole border: skipped in RETR_EXTERNAL
      }
      if (img[lnbd_x] > 0) goto resume_scan;  // inside an already traced outer border
      start = true;
    resume_scan:
      if (start) {
        found.emplace_back();
        fetch_outer_border(img + x, W, Pt{x - 1, y - 1}, found.back());
        p = img[x];  // now marked
      }
      prev = p;
      if (prev & -2) lnbd_x = x;
    }
  }
  out.assign(found.rbegin(), found.rend());
}

void distort_points(const float* src, float* dst, int n, const double K[9], const double* D,
                    int nD) {  // LED.cpp:181-224
  double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  double k1 = nD > 0 ? D[0] : 0, k2 = nD > 1 ? D[1] : 0, p1 = nD > 2 ? D[2] : 0,
         p2 = nD > 3 ? D[3] : 0, k3 = nD > 4 ? D[4] : 0;
  for (int i = 0; i < n; ++i) {
    double px = src[2 * i], py = src[2 * i + 1];
    double x = (px - cx) / fx;
    double y = (py - cy) / fy;
    double r2 = x * x + y * y;
    double xc = x * (1. + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2);
    double yc = y * (1. + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2);
    xc = xc + (2. * p1 * x * y + p2 * (r2 + 2. * x * x));
    yc = yc + (p1 * (r2 + 2. * y * y) + 2. * p2 * x * y);
    xc = xc * fx + cx;
    yc = yc * fy + cy;
    dst[2 * i] = (float)xc;
    dst[2 * i + 1] = (float)yc;
  }
}

// cv::undistortPoints(src, dst, K, D, noArray(), P = K): 5 fixed-point iterations, float I/O.
int undistort_points(const float* src, float* dst, int n, const double K[9], const double* D,
                     int nD) {  // LED.cpp:97-98
  double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  double ifx = 1. / fx, ify = 1. / fy;
  double k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < nD && i < 8; ++i) k[i] = D[i];
  const int iters = nD > 0 ? 5 : 0;
  for (int i = 0; i < n; ++i) {
    double x = src[2 * i], y = src[2 * i + 1];
    double x0 = x = (x - cx) * ifx;
    double y0 = y = (y - cy) * ify;
    for (int j = 0; j < iters; ++j) {
      double r2 = x * x + y * y;
      double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) /
                      (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
      double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
      double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
    // new camera matrix P = K (R = I): [xx yy ww]^T = K * [x y 1]^T
    double xx = K[0] * x + K[1] * y + K[2];
    double yy = K[3] * x + K[4] * y + K[5];
    double ww = 1. / (K[6] * x + K[7] * y + K[8]);
    dst[2 * i] = (float)(xx * ww);
    dst[2 * i + 1] = (float)(yy * ww);
  }
  return 0;
}

int find_leds(const uint8_t* img, int rows, int cols, size_t stride, int rx, int ry, int rw,
              int rh, const orc_params& p, const double K[9], const double* D, int nD,
              std::vector<double>& undist, std::vector<float>& dist) {
  undist.clear();
  dist.clear();
  if (rx < 0 || ry < 0 || rw < 0 || rh < 0 || rx + rw > cols || ry + rh > rows) return -1;
  static thread_local std::vector<uint8_t> g;
  if (blur_roi(img, stride, rx, ry, rw, rh, p.threshold_value, p.gaussian_sigma, g) != 0) return -2;
  std::vector<std::vector<Pt>> contours;
  external_contours(g.data(), rh, rw, contours);  // LED.cpp:56-57

  for (size_t i = 0; i < contours.size(); ++i) {  // LED.cpp:65-86
    const std::vector<Pt>& c = contours[i];
    const int n = (int)c.size();
    // cv::contourArea
    double a00 = 0;
    {
      float px = (float)c[n - 1].x, py = (float)c[n - 1].y;
      for (int j = 0; j < n; ++j) {
        float qx = (float)c[j].x, qy = (float)c[j].y;
        a00 += (double)px * qy - (double)py * qx;
        px = qx;
        py = qy;
      }
    }
    double area = std::fabs(a00 * 0.5);
    // cv::boundingRect
    int xmin = c[0].x, xmax = c[0].x, ymin = c[0].y, ymax = c[0].y;
    for (int j = 1; j < n; ++j) {
      xmin = std::min(xmin, c[j].x);
      xmax = std::max(xmax, c[j].x);
      ymin = std::min(ymin, c[j].y);
      ymax = std::max(ymax, c[j].y);
    }
    const int width = xmax - xmin + 1, height = ymax - ymin + 1;
    // cv::moments(contour) — Green's theorem on the polygon (contourMoments<int,double>)
    double m00 = 0, m10 = 0, m01 = 0;
    {
      double s00 = 0, s10 = 0, s01 = 0;
      double xi_1 = c[n - 1].x, yi_1 = c[n - 1].y;
      for (int j = 0; j < n; ++j) {
        double xi = c[j].x, yi = c[j].y;
        double dxy = xi_1 * yi - xi * yi_1;
        double xii_1 = xi_1 + xi, yii_1 = yi_1 + yi;
        s00 += dxy;
        s10 += dxy * xii_1;
        s01 += dxy * yii_1;
        xi_1 = xi;
        yi_1 = yi;
      }
      if (std::fabs(s00) > FLT_EPSILON) {
        double db1_2, db1_6;
        if (s00 > 0) {
          db1_2 = 0.5;
          db1_6 = 0.16666666666666666666666666666667;
        } else {
          db1_2 = -0.5;
          db1_6 = -0.16666666666666666666666666666667;
        }
        m00 = s00 * db1_2;
        m10 = s10 * db1_6;
        m01 = s01 * db1_6;
      }
    }
    // mc = Point2f(m10/m00, m01/m00) + Point2f(ROI.x, ROI.y)   (float arithmetic)  LED.cpp:73-74
    float mcx = (float)(m10 / m00) + (float)rx;
    float mcy = (float)(m01 / m00) + (float)ry;

    if (area >= p.min_blob_area && area <= p.max_blob_area &&
        std::abs(1 - std::min((double)width / (double)height, (double)height / (double)width)) <=
            p.max_width_height_distortion &&
        std::abs(1 - (area / (M_PI * std::pow(width / 2, 2)))) <= p.max_circular_distortion &&
        std::abs(1 - (area / (M_PI * std::pow(height / 2, 2)))) <= p.max_circular_distortion) {
      dist.push_back(mcx);
      dist.push_back(mcy);
    }
  }
  const int n = (int)dist.size() / 2;
  if (n > 0) {
    std::vector<float> und(2 * n);
    undistort_points(dist.data(), und.data(), n, K, D, nD);
    undist.resize(2 * n);
    for (int j = 0; j < 2 * n; ++j) undist[j] = (double)und[j];  // LED.cpp:104-110
  }
  return n;
}

// ---------------------------------------------------------------------------------------------
// Pose estimator — PE.cpp (fresh estimator, uninitialised branch)
// ---------------------------------------------------------------------------------------------
struct V2 {
  double x, y;
};
struct V4 {
  double v[4];
};

inline double square_dist(const V2& a, const V2& b) {  // PE.cpp:850-854
  double dx = a.x - b.x, dy = a.y - b.y;
  return dx * dx + dy * dy;
}

class Estimator {
 public:
  // tuning (PE.h:68-72) and camera
  double back_projection_pixel_tolerance_ = 3;
  double nearest_neighbour_pixel_tolerance_ = 5;
  double certainty_threshold_ = 0.75;
  double valid_correspondence_threshold_ = 0.7;
  unsigned histogram_threshold_ = 0;
  double K_[3][3];

  std::vector<V4> object_points_;
  std::vector<V2> image_points_;
  std::vector<V3> image_vectors_;
  std::vector<unsigned> corr_;  // rows of (marker, detection), 1-based
  M4 predicted_pose_;
  double pose_covariance_[36];
  int gn_iterations_ = 0;

  Estimator() {
    predicted_pose_ = identity4();
    std::memset(pose_covariance_, 0, sizeof(pose_covariance_));
  }
  void setCamera(const double K[9]) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) K_[i][j] = K[i * 3 + j];
  }
  void setMarkerPositions(const double* m, int n) {  // PE.cpp:50-55
    object_points_.resize(n);
    for (int i = 0; i < n; ++i) object_points_[i] = {{m[3 * i], m[3 * i + 1], m[3 * i + 2], 1.0}};
    histogram_threshold_ = orc_num_combinations((unsigned)n, 3);
  }
  void setImagePoints(const double* d, int n) {  // PE.cpp:166-170
    image_points_.resize(n);
    for (int i = 0; i < n; ++i) image_points_[i] = {d[2 * i], d[2 * i + 1]};
    calculateImageVectors();
  }
  void calculateImageVectors() {  // PE.cpp:288-301
    image_vectors_.resize(image_points_.size());
    for (size_t i = 0; i < image_points_.size(); ++i) {
      V3 s;
      s.x = (image_points_[i].x - K_[0][2]) / K_[0][0];
      s.y = (image_points_[i].y - K_[1][2]) / K_[1][1];
      s.z = 1;
      image_vectors_[i] = s / norm(s);
    }
  }
  V2 project2d(const V4& point, const M4& transform) const {  // PE.cpp:251-268
    double cam[3][4];
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) cam[i][j] = K_[i][j];
      cam[i][3] = 0.0;
    }
    // temp = (camera_matrix * transform) * point, evaluated left to right
    double CT[3][4];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 4; ++j) {
        double s = cam[i][0] * transform.m[0][j];
        for (int k = 1; k < 4; ++k) s += cam[i][k] * transform.m[k][j];
        CT[i][j] = s;
      }
    double t[3];
    for (int i = 0; i < 3; ++i) {
      double s = CT[i][0] * point.v[0];
      for (int k = 1; k < 4; ++k) s += CT[i][k] * point.v[k];
      t[i] = s;
    }
    return {t[0] / t[2], t[1] / t[2]};
  }

  // PE.cpp:862-906
  void calculateMinDistancesAndPairs(const std::vector<V2>& a, const std::vector<V2>& b,
                                     std::vector<unsigned>& pairs1, std::vector<double>& mind) const {
    const size_t na = a.size(), nb = b.size();
    pairs1.assign(na, 0u);
    mind.assign(na, 0.0);
    for (size_t i = 0; i < na; ++i) {
      double best = INFINITY;
      for (size_t j = 0; j < nb; ++j) {
        double d2 = square_dist(a[i], b[j]);
        if (d2 < best) {
          best = d2;
          pairs1[i] = (unsigned)j + 1;
        }
      }
      mind[i] = std::sqrt(best);
    }
  }

  // PE.cpp:303-342
  double calculateSquaredReprojectionErrorAndCertainty(const std::vector<V2>& image_pts,
                                                       const std::vector<V2>& object_pts,
                                                       double& certainty) const {
    double squared_error = 0;
    unsigned num_correspondences = 0;
    const size_t R = image_pts.size(), C = object_pts.size();
    std::vector<double> dist(R * C);
    for (size_t i = 0; i < R; ++i)
      for (size_t j = 0; j < C; ++j) dist[i * C + j] = std::sqrt(square_dist(image_pts[i], object_pts[j]));
    for (size_t it = 1; it <= std::min(R, C); ++it) {
      // Eigen minCoeff(&r,&c): column-major visit, first strict minimum wins
      double mv = 0;
      size_t ri = 0, ci = 0;
      bool first = true;
      for (size_t j = 0; j < C; ++j)
        for (size_t i = 0; i < R; ++i) {
          double v = dist[i * C + j];
          if (first || v < mv) {
            mv = v;
            ri = i;
            ci = j;
            first = false;
          }
        }
      if (mv <= back_projection_pixel_tolerance_) {
        squared_error += std::pow((double)dist[ri * C + ci], 2);
        num_correspondences++;
        for (size_t j = 0; j < C; ++j) dist[ri * C + j] = INFINITY;
        for (size_t i = 0; i < R; ++i) dist[i * C + ci] = INFINITY;
      } else
        break;
    }
    certainty = (double)num_correspondences / C;
    return squared_error;
  }

  // PE.cpp:344-370.  hist: n_det x n_markers row-major, consumed.
  void correspondencesFromHistogram(std::vector<unsigned>& hist, int n_det, int n_m) {
    corr_.clear();
    for (int j = 0; j < n_m; ++j) {
      unsigned mv = 0;
      int ri = 0, ci = 0;
      bool first = true;
      for (int c = 0; c < n_m; ++c)
        for (int r = 0; r < n_det; ++r) {
          unsigned v = hist[(size_t)r * n_m + c];
          if (first || v > mv) {
            mv = v;
            ri = r;
            ci = c;
            first = false;
          }
        }
      if (mv < histogram_threshold_) break;
      corr_.push_back((unsigned)ci + 1);
      corr_.push_back((unsigned)ri + 1);
      for (int r = 0; r < n_det; ++r) hist[(size_t)r * n_m + ci] = 0;
    }
  }

  // PE.cpp:908-930.  object / reprojected: n points each.
  static M4 computeTransformation(const std::vector<V3>& obj, const std::vector<V3>& rep) {
    const size_t n = obj.size();
    V3 mo = {0, 0, 0}, mr = {0, 0, 0};
    for (size_t i = 0; i < n; ++i) {
      mo = mo + obj[i];
      mr = mr + rep[i];
    }
    mo = mo / (double)n;
    mr = mr / (double)n;
    M3 H;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) H.m[i][j] = 0;
    for (size_t k = 0; k < n; ++k) {
      V3 a = obj[k] - mo, b = rep[k] - mr;
      const double av[3] = {a.x, a.y, a.z}, bv[3] = {b.x, b.y, b.z};
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) H.m[i][j] += av[i] * bv[j];
    }
    M3 U, V;
    double s[3];
    svd3(H, U, s, V);
    M3 R = mul(V, transpose(U));  // no reflection guard (quirk A.6.8)
    V3 Rm = mul(R, mo);
    V3 t = mr - Rm;
    M4 T = identity4();
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) T.m[i][j] = R.m[i][j];
    T.m[0][3] = t.x;
    T.m[1][3] = t.y;
    T.m[2][3] = t.z;
    return T;
  }

  unsigned checkCorrespondences() {  // PE.cpp:394-542
    const int n_c = (int)corr_.size() / 2;
    const int n_m = (int)object_points_.size();
    if (n_c < 4) return 0;
    std::vector<V4> mean(n_m, V4{{0, 0, 0, 0}});
    std::vector<unsigned> combos((size_t)orc_combinations3(n_c, nullptr) * 3);
    const unsigned N = (unsigned)orc_combinations3(n_c, combos.data());
    unsigned num_valid = 0;
    const unsigned total_unused = n_c - 3;
    for (unsigned i = 0; i < N; ++i) {
      V3 fv[3], wp[3];
      for (int k = 0; k < 3; ++k) {
        const unsigned row = combos[i * 3 + k] - 1;
        const V4& mp = object_points_[corr_[row * 2 + 0] - 1];
        wp[k] = {mp.v[0], mp.v[1], mp.v[2]};
        fv[k] = image_vectors_[corr_[row * 2 + 1] - 1];
      }
      std::vector<V2> unused_im(total_unused);
      std::vector<V4> unused_obj(total_unused);
      unsigned nu = 0;
      for (int l = 0; l < n_c; ++l) {
        bool used = false;
        for (int n = 0; n < 3; ++n)
          if ((int)combos[i * 3 + n] - 1 == l) used = true;
        if (!used) {
          unused_obj[nu] = object_points_[corr_[l * 2 + 0] - 1];
          unused_im[nu] = image_points_[corr_[l * 2 + 1] - 1];
          nu++;
        }
        if (nu == total_unused) break;
      }
      double sol[4][3][4];
      if (p3p_compute(fv, wp, sol) != 0) continue;
      double min_sq = INFINITY;
      unsigned best = 0;
      bool found = false;
      for (unsigned j = 0; j < 4; ++j) {
        M4 H = identity4();
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 4; ++c) H.m[r][c] = sol[j][r][c];
        if (!is_finite4(H)) continue;
        std::vector<V2> back(total_unused);
        for (unsigned ii = 0; ii < total_unused; ++ii) back[ii] = project2d(unused_obj[ii], inverse4(H));
        double certainty;
        double sq = calculateSquaredReprojectionErrorAndCertainty(unused_im, back, certainty);
        if (certainty >= certainty_threshold_) {
          found = true;
          if (sq < min_sq) {
            min_sq = sq;
            best = j;
          }
        }
      }
      if (found) {
        num_valid++;
        M4 H = identity4();
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 4; ++c) H.m[r][c] = sol[best][r][c];
        M4 Hi = inverse4(H);
        for (int jj = 0; jj < n_m; ++jj) {
          for (int r = 0; r < 4; ++r) {
            double s = Hi.m[r][0] * object_points_[jj].v[0];
            for (int k = 1; k < 4; ++k) s += Hi.m[r][k] * object_points_[jj].v[k];
            mean[jj].v[r] = mean[jj].v[r] + s;
          }
        }
      }
    }
    if ((double)num_valid / N >= valid_correspondence_threshold_) {
      std::vector<V3> obj(n_m), rep(n_m);
      for (int kk = 0; kk < n_m; ++kk) {
        rep[kk] = {mean[kk].v[0] / num_valid, mean[kk].v[1] / num_valid, mean[kk].v[2] / num_valid};
        obj[kk] = {object_points_[kk].v[0], object_points_[kk].v[1], object_points_[kk].v[2]};
      }
      predicted_pose_ = computeTransformation(obj, rep);
      return 1;
    }
    return 0;
  }

  // the voting part of initialise(), PE.cpp:544-702
  // (item_lo, item_hi: forensics only — the hypotheses [lo, hi) of the loop nest, flattened as triple * n_perms +
  //  permutation; the default is the whole nest)
  void voteHistogram(std::vector<unsigned>& hist, long long item_lo = 0, long long item_hi = -1) const {
    const int n_d = (int)image_points_.size(), n_m = (int)object_points_.size();
    hist.assign((size_t)n_d * n_m, 0u);
    std::vector<unsigned> combos((size_t)orc_combinations3(n_d, nullptr) * 3);
    const unsigned n_combos = (unsigned)orc_combinations3(n_d, combos.data());
    std::vector<unsigned> perms((size_t)orc_permutations3(n_m, nullptr) * 3);
    const unsigned n_perms = (unsigned)orc_permutations3(n_m, perms.data());
    const unsigned total_unused_im = n_d - 3;
    const unsigned total_unused_obj = n_m - 3;
    std::vector<V2> unused_im(total_unused_im), back(total_unused_obj);
    std::vector<unsigned> unused_im_idx(total_unused_im), unused_obj_idx(total_unused_obj);
    std::vector<V4> unused_obj(total_unused_obj);
    std::vector<unsigned> pairs1;
    std::vector<double> mind;
    for (unsigned i = 0; i < n_combos; ++i) {
      V3 fv[3];
      for (int k = 0; k < 3; ++k) fv[k] = image_vectors_[combos[i * 3 + k] - 1];
      unsigned nu = 0;
      for (int kk = 0; kk < n_d && nu < total_unused_im; ++kk) {
        bool used = false;
        for (int ii = 0; ii < 3; ++ii)
          if ((int)combos[i * 3 + ii] - 1 == kk) used = true;
        if (!used) {
          unused_im[nu] = image_points_[kk];
          unused_im_idx[nu] = kk;
          nu++;
        }
      }
      for (unsigned j = 0; j < n_perms; ++j) {
        if (item_hi >= 0) {
          const long long g = (long long)i * n_perms + j;
          if (g < item_lo || g >= item_hi) continue;
        }
        V3 wp[3];
        for (int k = 0; k < 3; ++k) {
          const V4& mp = object_points_[perms[j * 3 + k] - 1];
          wp[k] = {mp.v[0], mp.v[1], mp.v[2]};
        }
        double sol[4][3][4];
        if (p3p_compute(fv, wp, sol) != 0) continue;
        unsigned no = 0;
        for (int ll = 0; ll < n_m && no < total_unused_obj; ++ll) {
          bool used = false;
          for (int jj = 0; jj < 3; ++jj)
            if ((int)perms[j * 3 + jj] - 1 == ll) used = true;
          if (!used) {
            unused_obj[no] = object_points_[ll];
            unused_obj_idx[no] = ll;
            no++;
          }
        }
        for (unsigned k = 0; k < 4; ++k) {
          M4 H = identity4();
          for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) H.m[r][c] = sol[k][r][c];
          if (!is_finite4(H)) continue;
          for (unsigned m = 0; m < total_unused_obj; ++m) back[m] = project2d(unused_obj[m], inverse4(H));
          calculateMinDistancesAndPairs(unused_im, back, pairs1, mind);
          unsigned count = 0;
          for (size_t ll = 0; ll < mind.size(); ++ll)
            if (mind[ll] < back_projection_pixel_tolerance_) count++;
          if (count > 0) {
            for (int mm = 0; mm < 3; ++mm) {
              unsigned im_idx = combos[i * 3 + mm] - 1, obj_idx = perms[j * 3 + mm] - 1;
              hist[(size_t)im_idx * n_m + obj_idx] += 1;
            }
            for (size_t nn = 0; nn < mind.size(); ++nn)
              if (mind[nn] < back_projection_pixel_tolerance_) {
                // pairs(nn,0) = nn+1 ; pairs(nn,1) = nearest back-projection (1-based)
                unsigned im_idx = unused_im_idx[nn];
                unsigned obj_idx = unused_obj_idx[pairs1[nn] - 1];
                hist[(size_t)im_idx * n_m + obj_idx] += 1;
              }
          }
        }
      }
    }
  }

  unsigned initialise(std::vector<unsigned>* hist_out) {  // PE.cpp:544-721
    std::vector<unsigned> hist;
    voteHistogram(hist);
    if (hist_out) *hist_out = hist;
    bool all_zero = true;
    for (unsigned v : hist)
      if (v != 0) all_zero = false;
    if (all_zero) return 0;
    correspondencesFromHistogram(hist, (int)image_points_.size(), (int)object_points_.size());
    return checkCorrespondences() == 1 ? 1 : 0;
  }

  static M3 skew(const V3& w) {  // PE.cpp:1066-1071
    return {{{0, -w.z, w.y}, {w.z, 0, -w.x}, {-w.y, w.x, 0}}};
  }
  static M4 exponentialMap(const double twist[6]) {  // PE.cpp:962-994
    V3 upsilon = {twist[0], twist[1], twist[2]};
    V3 omega = {twist[3], twist[4], twist[5]};
    double theta = norm(omega);
    double theta_squared = theta * theta;
    M3 Omega = skew(omega);
    M3 Omega_squared = mul(Omega, Omega);
    M3 rotation, V;
    if (theta == 0) {
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) rotation.m[i][j] = V.m[i][j] = (i == j) ? 1.0 : 0.0;
    } else {
      const double st = std::sin(theta), ct = std::cos(theta);
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          double I = (i == j) ? 1.0 : 0.0;
          rotation.m[i][j] = I + Omega.m[i][j] / theta * st + Omega_squared.m[i][j] / theta_squared * (1 - ct);
          V.m[i][j] = (I + (1 - ct) / (theta_squared)*Omega.m[i][j] +
                       (theta - st) / (theta_squared * theta) * Omega_squared.m[i][j]);
        }
    }
    M4 T = identity4();
    V3 t = mul(V, upsilon);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) T.m[i][j] = rotation.m[i][j];
    T.m[0][3] = t.x;
    T.m[1][3] = t.y;
    T.m[2][3] = t.z;
    return T;
  }
  static void computeJacobian(const M4& T, const V4& wp, double fx, double fy, double J[2][6]) {  // PE.cpp:932-960
    double pc[4];
    for (int i = 0; i < 4; ++i) {
      double s = T.m[i][0] * wp.v[0];
      for (int k = 1; k < 4; ++k) s += T.m[i][k] * wp.v[k];
      pc[i] = s;
    }
    double x = pc[0], y = pc[1], z = pc[2], z_2 = z * z;
    J[0][0] = 1 / z * fx;
    J[0][1] = 0;
    J[0][2] = -x / z_2 * fx;
    J[0][3] = -x * y / z_2 * fx;
    J[0][4] = (1 + (x * x / z_2)) * fx;
    J[0][5] = -y / z * fx;
    J[1][0] = 0;
    J[1][1] = 1 / z * fy;
    J[1][2] = -y / z_2 * fy;
    J[1][3] = -(1 + y * y / z_2) * fy;
    J[1][4] = x * y / z_2 * fy;
    J[1][5] = x / z * fy;
  }
  static double norm_max(const double v[6]) {  // PE.cpp:1073-1085
    double mx = -1;
    for (int i = 0; i < 6; ++i) {
      double a = std::abs(v[i]);
      if (a > mx) mx = a;
    }
    return mx;
  }

  void optimisePose() {  // PE.cpp:733-792
    const double converged = 1e-13;
    const unsigned max_itr = 500;
    double A[36], b[6], dT[6];
    const double fx = K_[0][0], fy = K_[1][1];
    const int n_c = (int)corr_.size() / 2;
    gn_iterations_ = 0;
    for (unsigned i = 0; i < max_itr; ++i) {
      std::memset(A, 0, sizeof(A));
      std::memset(b, 0, sizeof(b));
      for (int j = 0; j < n_c; ++j) {
        if (corr_[j * 2 + 1] == 0) continue;
        const V4& op = object_points_[corr_[j * 2 + 0] - 1];
        V2 pim = project2d(op, predicted_pose_);
        const V2& ip = image_points_[corr_[j * 2 + 1] - 1];
        double e[2] = {ip.x - pim.x, ip.y - pim.y};
        double J[2][6];
        computeJacobian(predicted_pose_, op, fx, fy, J);
        // A += J^T * R^-1 * J ; b += J^T * R^-1 * e   with R = I (PE.cpp:774-775)
        for (int r = 0; r < 6; ++r) {
          for (int c = 0; c < 6; ++c) A[r * 6 + c] += J[0][r] * J[0][c] + J[1][r] * J[1][c];
          b[r] += J[0][r] * e[0] + J[1][r] * e[1];
        }
      }
      ldlt_solve6(A, b, dT);
      predicted_pose_ = mul(exponentialMap(dT), predicted_pose_);
      gn_iterations_ = (int)i + 1;
      if (norm_max(dT) <= converged) break;
    }
    inverse6(A, pose_covariance_);
  }
};

// ---------------------------------------------------------------------------------------------
// Stateful estimator: the full estimateBodyPose state machine incl. the tracking path
// (PE.cpp:62-147, 232-244, 270-276, 372-392, 794-848, 996-1064; LED.cpp:114-179)
// ---------------------------------------------------------------------------------------------
struct RectI {
  int x, y, width, height;
};

class Tracker {
 public:
  Estimator e;
  orc_params p;
  double Kc[9];
  std::vector<double> D;
  M4 current_pose_, previous_pose_;
  double current_time_ = 0, previous_time_ = 0, predicted_time_ = 0;
  unsigned it_since_initialized_ = 0;  // PE.cpp:41
  RectI roi_ = {0, 0, 0, 0};
  bool pose_updated_ = false;
  std::vector<V2> predicted_pixel_positions_;
  std::vector<double> det_;    // detected_led_positions of the current call
  std::vector<float> dist_;
  int last_used_bruteforce_ = 0;

  Tracker() {
    current_pose_ = identity4();
    previous_pose_ = identity4();
  }

  // PE.cpp:996-1064
  static void logarithmMap(const M4& trans, double xi[6]) {
    double R[3][3], t[3];
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) R[i][j] = trans.m[i][j];
      t[i] = trans.m[i][3];
    }
    double w_hat[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    double phi = 0;
    // R.isApprox(Identity, 1e-10): ||R - I||_F^2 <= prec^2 * min(||R||_F^2, ||I||_F^2)
    double dn = 0, rn = 0;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double dI = R[i][j] - (i == j ? 1.0 : 0.0);
        dn += dI * dI;
        rn += R[i][j] * R[i][j];
      }
    const bool is_identity = dn <= 1e-10 * 1e-10 * std::min(rn, 3.0);
    if (!is_identity) {
      double temp = (R[0][0] + R[1][1] + R[2][2] - 1) / 2;
      if (temp > 1)
        temp = 1;
      else if (temp < -1)
        temp = -1;
      phi = std::acos(temp);
      if (phi != 0) {
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) w_hat[i][j] = (R[i][j] - R[j][i]) / (2 * std::sin(phi)) * phi;
      }
    }
    double w[3] = {w_hat[2][1], w_hat[0][2], w_hat[1][0]};
    double w_norm = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    double A_inv[3][3];
    // t.isApproxToConstant(0, 1e-10): |t_i - 0| <= min(|t_i|, 0) * prec  <=>  t == 0 exactly
    const bool t_zero = (t[0] == 0 && t[1] == 0 && t[2] == 0);
    if (t_zero) {
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A_inv[i][j] = 0;
    } else if (w_norm == 0 || std::sin(w_norm) == 0) {
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A_inv[i][j] = (i == j) ? 1.0 : 0.0;
    } else {
      double w2[3][3];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          double s = w_hat[i][0] * w_hat[0][j];
          s += w_hat[i][1] * w_hat[1][j];
          s += w_hat[i][2] * w_hat[2][j];
          w2[i][j] = s;
        }
      const double c = (2 * std::sin(w_norm) - w_norm * (1 + std::cos(w_norm))) / (2 * w_norm * w_norm * std::sin(w_norm));
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A_inv[i][j] = ((i == j) ? 1.0 : 0.0) - w_hat[i][j] / 2 + c * w2[i][j];
    }
    for (int i = 0; i < 3; ++i) {
      double s = A_inv[i][0] * t[0];
      s += A_inv[i][1] * t[1];
      s += A_inv[i][2] * t[2];
      xi[i] = s;
      xi[3 + i] = w[i];
    }
  }

  void predictPose(double time_to_predict) {  // PE.cpp:232-244
    predicted_time_ = time_to_predict;
    double delta[6], delta_hat[6];
    logarithmMap(mul(inverse4(previous_pose_), current_pose_), delta);
    for (int i = 0; i < 6; ++i) delta_hat[i] = delta[i] / (current_time_ - previous_time_) * (predicted_time_ - current_time_);
    e.predicted_pose_ = mul(current_pose_, Estimator::exponentialMap(delta_hat));
  }

  void predictMarkerPositionsInImage() {  // PE.cpp:270-276
    predicted_pixel_positions_.resize(e.object_points_.size());
    for (size_t i = 0; i < e.object_points_.size(); ++i)
      predicted_pixel_positions_[i] = e.project2d(e.object_points_[i], e.predicted_pose_);
  }

  RectI determineROI(int rows, int cols) const {  // LED.cpp:114-179
    std::vector<double> px;
    for (const V2& q : predicted_pixel_positions_) {
      px.push_back(q.x);
      px.push_back(q.y);
    }
    int r[4];
    orc_determine_roi(px.data(), (int)predicted_pixel_positions_.size(), rows, cols, (int)p.roi_border_thickness, Kc,
                      D.data(), (int)D.size(), r);
    return RectI{r[0], r[1], r[2], r[3]};
  }

  void findCorrespondences() {  // PE.cpp:372-392
    std::vector<unsigned> pairs1;
    std::vector<double> mind;
    e.calculateMinDistancesAndPairs(predicted_pixel_positions_, e.image_points_, pairs1, mind);
    e.corr_.clear();
    for (size_t i = 0; i < pairs1.size(); ++i)
      if (mind[i] <= p.nearest_neighbour_pixel_tolerance) {
        e.corr_.push_back((unsigned)i + 1);
        e.corr_.push_back(pairs1[i]);
      }
  }

  void updatePose() {  // PE.cpp:794-800
    previous_pose_ = current_pose_;
    current_pose_ = e.predicted_pose_;
    previous_time_ = current_time_;
    current_time_ = predicted_time_;
  }
  void optimiseAndUpdatePose() {  // PE.cpp:802-812
    e.optimisePose();
    if (it_since_initialized_ < 2) it_since_initialized_++;
    updatePose();
    pose_updated_ = true;
  }
  void findCorrespondencesAndPredictPose() {  // PE.cpp:831-848
    findCorrespondences();
    last_used_bruteforce_ = 0;
    if (e.checkCorrespondences() == 1) {
      optimiseAndUpdatePose();
    } else {
      last_used_bruteforce_ = 1;
      if (e.initialise(nullptr) == 1) optimiseAndUpdatePose();
    }
  }

  int detect(const uint8_t* img, int rows, int cols, size_t stride) {
    std::vector<double> u;
    std::vector<float> d;
    int n = find_leds(img, rows, cols, stride, roi_.x, roi_.y, roi_.width, roi_.height, p, Kc, D.data(), (int)D.size(), u, d);
    if (n < 0) return n;
    dist_ = d;                // distorted_detection_centers = distorted_points (LED.cpp:89)
    if (n > 0) det_ = u;      // pixel_positions is only resized/filled when numPoints > 0 (LED.cpp:91-111)
    return n;
  }

  // PE.cpp:62-147
  int estimateBodyPose(const uint8_t* img, int rows, int cols, size_t stride, double time_to_predict) {
    pose_updated_ = false;
    last_used_bruteforce_ = 0;
    det_.clear();
    if (it_since_initialized_ < 1) {
      predicted_time_ = time_to_predict;
      roi_ = {0, 0, cols, rows};
      if (detect(img, rows, cols, stride) < 0) return -1;
      if (det_.size() / 2 >= 4) {
        e.setImagePoints(det_.data(), (int)det_.size() / 2);
        last_used_bruteforce_ = 1;
        if (e.initialise(nullptr) == 1) optimiseAndUpdatePose();
      }
    } else {
      // predictWithROI, PE.cpp:814-829
      if (it_since_initialized_ >= 2)
        predictPose(time_to_predict);
      else
        predicted_time_ = time_to_predict;
      predictMarkerPositionsInImage();
      roi_ = determineROI(rows, cols);
      if (detect(img, rows, cols, stride) < 0) return -1;
      bool repeat_check = true;
      unsigned num_loops = 0;
      do {
        num_loops++;
        if (det_.size() / 2 >= 4) {
          e.setImagePoints(det_.data(), (int)det_.size() / 2);
          findCorrespondencesAndPredictPose();
          repeat_check = false;
        } else {
          if (num_loops < 2) {
            roi_ = {0, 0, cols, rows};
            if (detect(img, rows, cols, stride) < 0) return -1;
          } else {
            repeat_check = false;
          }
        }
      } while (repeat_check);
    }
    return pose_updated_ ? 1 : 0;
  }
};

void fill_params(Estimator& e, const orc_params* p, int n_markers) {
  e.back_projection_pixel_tolerance_ = p->back_projection_pixel_tolerance;
  e.nearest_neighbour_pixel_tolerance_ = p->nearest_neighbour_pixel_tolerance;
  e.certainty_threshold_ = p->certainty_threshold;
  e.valid_correspondence_threshold_ = p->valid_correspondence_threshold;
  if (p->histogram_threshold != 0) e.histogram_threshold_ = p->histogram_threshold;
  (void)n_markers;
}

void write_result(const Estimator& e, bool ok, int n_det, orc_result* out) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) out->T[i * 4 + j] = e.predicted_pose_.m[i][j];
  std::memcpy(out->cov, e.pose_covariance_, sizeof(out->cov));
  out->status = ok ? 0 : 1;
  out->n_det = n_det;
  out->n_corr = (int)e.corr_.size() / 2;
  out->gn_iterations = e.gn_iterations_;
}

int solve_bruteforce(const double* det, int n_det, const double* markers, int n_markers,
                     const double K[9], const orc_params* p, orc_result* out, uint32_t* hist,
                     uint32_t* corr) {
  Estimator e;
  e.setCamera(K);
  e.setMarkerPositions(markers, n_markers);
  fill_params(e, p, n_markers);
  bool ok = false;
  if (n_det >= 4) {  // min_num_leds_detected_, PE.h:78 / PE.cpp:80
    e.setImagePoints(det, n_det);
    std::vector<unsigned> h;
    if (e.initialise(&h) == 1) {
      e.optimisePose();  // optimiseAndUpdatePose, PE.cpp:89,802-812
      ok = true;
    }
    if (hist) std::memcpy(hist, h.data(), h.size() * sizeof(unsigned));
    if (corr) {
      std::memset(corr, 0, sizeof(uint32_t) * 2 * n_markers);
      std::memcpy(corr, e.corr_.data(), e.corr_.size() * sizeof(unsigned));
    }
  } else {
    if (hist) std::memset(hist, 0, sizeof(uint32_t) * (size_t)n_det * n_markers);
    if (corr) std::memset(corr, 0, sizeof(uint32_t) * 2 * n_markers);
  }
  write_result(e, ok, n_det, out);
  return ok ? 0 : 1;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

unsigned orc_factorial(int N) { return factorial_u32(N); }

unsigned orc_num_combinations(unsigned N, unsigned K) {  // COMB.cpp:42-45
  return factorial_u32((int)N) / (factorial_u32((int)K) * factorial_u32((int)(N - K)));
}

int orc_combinations3(unsigned N, unsigned* out) {  // COMB.cpp:52-129: lexicographic, 1-based
  if (N < 3) return 0;
  int rows = 0;
  for (unsigned a = 1; a <= N; ++a)
    for (unsigned b = a + 1; b <= N; ++b)
      for (unsigned c = b + 1; c <= N; ++c) {
        if (out) {
          out[rows * 3 + 0] = a;
          out[rows * 3 + 1] = b;
          out[rows * 3 + 2] = c;
        }
        ++rows;
      }
  return rows;
}

int orc_permutations3(unsigned N, unsigned* out) {
  // COMB.cpp:131-244: for each lexicographic combination (a<b<c) the block
  // [c b a],[c a b],[b c a],[b a c],[a b c],[a c b]  (= permutations(3) applied as indices)
  if (N < 3) return 0;
  static const int P3[6][3] = {{2, 1, 0}, {2, 0, 1}, {1, 2, 0}, {1, 0, 2}, {0, 1, 2}, {0, 2, 1}};
  int rows = 0;
  for (unsigned a = 1; a <= N; ++a)
    for (unsigned b = a + 1; b <= N; ++b)
      for (unsigned c = b + 1; c <= N; ++c) {
        const unsigned wv[3] = {a, b, c};
        for (int r = 0; r < 6; ++r) {
          if (out)
            for (int k = 0; k < 3; ++k) out[rows * 3 + k] = wv[P3[r][k]];
          ++rows;
        }
      }
  return rows;
}

int orc_solve_quartic(const double factors[5], double real_roots[4]) {
  return solve_quartic(factors, real_roots);
}

// LED.cpp:114-179 — bounding box of the predicted pixels, corners distorted, border added, clipped
void orc_determine_roi(const double* px, int n, int rows, int cols, int border_size, const double K[9],
                       const double* D, int nD, int roi[4]) {
  double x_min = INFINITY, x_max = 0, y_min = INFINITY, y_max = 0;
  for (int i = 0; i < n; ++i) {
    if (px[2 * i] < x_min) x_min = px[2 * i];
    if (px[2 * i] > x_max) x_max = px[2 * i];
    if (px[2 * i + 1] < y_min) y_min = px[2 * i + 1];
    if (px[2 * i + 1] > y_max) y_max = px[2 * i + 1];
  }
  float und[4] = {(float)x_min, (float)y_min, (float)x_max, (float)y_max};  // cv::Point2f
  float dst[4];
  distort_points(und, dst, 2, K, D, nD);
  const double x_min_dist = dst[0], y_min_dist = dst[1], x_max_dist = dst[2], y_max_dist = dst[3];
  const double x0 = std::max(0.0, std::min((double)cols, x_min_dist - border_size));
  const double x1 = std::max(0.0, std::min((double)cols, x_max_dist + border_size));
  const double y0 = std::max(0.0, std::min((double)rows, y_min_dist - border_size));
  const double y1 = std::max(0.0, std::min((double)rows, y_max_dist + border_size));
  if (x1 - x0 < 1 || y1 - y0 < 1) {
    roi[0] = 0;
    roi[1] = 0;
    roi[2] = cols;
    roi[3] = rows;
  } else {
    roi[0] = (int)x0;
    roi[1] = (int)y0;
    roi[2] = (int)(x1 - x0);
    roi[3] = (int)(y1 - y0);
  }
}

int orc_p3p(const double fv[9], const double wp[9], double sol[48]) {
  V3 f[3], w[3];
  for (int i = 0; i < 3; ++i) {
    f[i] = {fv[i * 3], fv[i * 3 + 1], fv[i * 3 + 2]};
    w[i] = {wp[i * 3], wp[i * 3 + 1], wp[i * 3 + 2]};
  }
  double s[4][3][4];
  int rc = p3p_compute(f, w, s);
  if (rc == 0) std::memcpy(sol, s, sizeof(s));
  return rc;
}

int orc_gaussian_kernel_q8(double sigma, int* taps, int cap) {
  std::vector<int> t;
  int n = gaussian_kernel_q8(sigma, t);
  if (n < 0) return n;
  if (n > cap) return -2;
  for (int i = 0; i < n; ++i) taps[i] = t[i];
  return n;
}

int orc_blur_mask(const uint8_t* img, int rows, int cols, size_t stride, int rx, int ry, int rw,
                  int rh, int threshold_value, double sigma, uint8_t* blurred, uint8_t* mask) {
  if (rx < 0 || ry < 0 || rw < 0 || rh < 0 || rx + rw > cols || ry + rh > rows) return -1;
  std::vector<uint8_t> g;
  if (blur_roi(img, stride, rx, ry, rw, rh, threshold_value, sigma, g) != 0) return -2;
  for (size_t i = 0; i < g.size(); ++i) {
    if (blurred) blurred[i] = g[i];
    if (mask) mask[i] = g[i] ? 1 : 0;
  }
  return 0;
}

int orc_external_contours(const uint8_t* mask, int h, int w, int* pts, int pts_cap, int* counts,
                          int counts_cap) {
  std::vector<std::vector<Pt>> cs;
  external_contours(mask, h, w, cs);
  if ((int)cs.size() > counts_cap) return -1;
  int off = 0;
  for (size_t i = 0; i < cs.size(); ++i) {
    if (off + (int)cs[i].size() > pts_cap) return -2;
    counts[i] = (int)cs[i].size();
    for (const Pt& p : cs[i]) {
      pts[2 * off] = p.x;
      pts[2 * off + 1] = p.y;
      ++off;
    }
  }
  return (int)cs.size();
}

int orc_find_leds(const uint8_t* img, int rows, int cols, size_t stride, int rx, int ry, int rw,
                  int rh, const orc_params* p, const double K[9], const double* D, int nD,
                  double* undist_xy, float* dist_xy, int cap, int* n_out) {
  std::vector<double> u;
  std::vector<float> d;
  int n = find_leds(img, rows, cols, stride, rx, ry, rw, rh, *p, K, D, nD, u, d);
  if (n < 0) return n;
  if (n_out) *n_out = n;
  if (n > cap) return -3;
  for (int i = 0; i < 2 * n; ++i) {
    if (undist_xy) undist_xy[i] = u[i];
    if (dist_xy) dist_xy[i] = d[i];
  }
  return 0;
}

void orc_distort_points(const float* src_xy, float* dst_xy, int n, const double K[9],
                        const double* D, int nD) {
  distort_points(src_xy, dst_xy, n, K, D, nD);
}
int orc_undistort_points(const float* src_xy, float* dst_xy, int n, const double K[9],
                         const double* D, int nD) {
  return undistort_points(src_xy, dst_xy, n, K, D, nD);
}

void orc_image_vectors(const double* det, int n_det, const double K[9], double* vec3) {
  Estimator e;
  e.setCamera(K);
  e.setImagePoints(det, n_det);
  for (int i = 0; i < n_det; ++i) {
    vec3[3 * i] = e.image_vectors_[i].x;
    vec3[3 * i + 1] = e.image_vectors_[i].y;
    vec3[3 * i + 2] = e.image_vectors_[i].z;
  }
}

void orc_project2d(const double p4[4], const double T[16], const double K[9], double out[2]) {
  Estimator e;
  e.setCamera(K);
  M4 M;
  std::memcpy(M.m, T, sizeof(M.m));
  V4 p = {{p4[0], p4[1], p4[2], p4[3]}};
  V2 r = e.project2d(p, M);
  out[0] = r.x;
  out[1] = r.y;
}

void orc_exponential_map(const double twist[6], double T[16]) {
  M4 M = Estimator::exponentialMap(twist);
  std::memcpy(T, M.m, sizeof(M.m));
}

void orc_jacobian(const double T[16], const double p4[4], double fx, double fy, double J[12]) {
  M4 M;
  std::memcpy(M.m, T, sizeof(M.m));
  V4 p = {{p4[0], p4[1], p4[2], p4[3]}};
  double Jm[2][6];
  Estimator::computeJacobian(M, p, fx, fy, Jm);
  std::memcpy(J, Jm, sizeof(Jm));
}

void orc_compute_transformation(const double* obj, const double* rep, int n, double T[16]) {
  std::vector<V3> a(n), b(n);
  for (int i = 0; i < n; ++i) {
    a[i] = {obj[3 * i], obj[3 * i + 1], obj[3 * i + 2]};
    b[i] = {rep[3 * i], rep[3 * i + 1], rep[3 * i + 2]};
  }
  M4 M = Estimator::computeTransformation(a, b);
  std::memcpy(T, M.m, sizeof(M.m));
}

// ---- frame decode: cv_bridge::toCvCopy(image_msg, MONO8), monocular_pose_estimator.cpp:147 -------------------
// What cv_bridge does for the encodings a camera driver publishes (cv_bridge/src/cv_bridge.cpp, toCvCopyImpl):
//   bgr8 / rgb8 / bgra8 / rgba8 -> mono8: cv::cvtColor(COLOR_{BGR,RGB,BGRA,RGBA}2GRAY), for CV_8U the integer form
//       Y = (B * B2Y + G * G2Y + R * R2Y + (1 << (yuv_shift - 1))) >> yuv_shift, B2Y 1868, G2Y 9617, R2Y 4899, yuv_shift 14
//   mono16 -> mono8: a byte swap when Image.is_bigendian differs from the host, then
//       Mat::convertTo(CV_8U, 255. / 65535.): saturate_cast<uchar>(src * (float)alpha) = round-half-even, clamped
//   mono8: a copy.
// encoding: 0 mono8, 1 bgr8, 2 rgb8, 3 bgra8, 4 rgba8, 5 mono16 (the MPE_ENC_* values).  dst: packed rows x cols.
int orc_convert_to_mono8(const uint8_t* src, int encoding, int big_endian, int rows, int cols, size_t src_stride,
                         uint8_t* dst) {
  if (!src || !dst || rows <= 0 || cols <= 0 || encoding < 0 || encoding > 5) return -1;
  const float alpha = (float)(255. / 65535.);
  for (int y = 0; y < rows; ++y) {
    const uint8_t* s = src + (size_t)y * src_stride;
    uint8_t* d = dst + (size_t)y * cols;
    for (int x = 0; x < cols; ++x) {
      if (encoding == 0) {
        d[x] = s[x];
      } else if (encoding == 5) {
        const unsigned v = big_endian ? ((unsigned)s[2 * x] << 8 | s[2 * x + 1]) : ((unsigned)s[2 * x + 1] << 8 | s[2 * x]);
        const float r = std::nearbyintf((float)v * alpha);  // cvRound: round half to even (default rounding mode)
        d[x] = (uint8_t)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
      } else {
        const int bpp = (encoding == 3 || encoding == 4) ? 4 : 3;
        const bool rgb = encoding == 2 || encoding == 4;
        const unsigned c0 = s[bpp * x], c1 = s[bpp * x + 1], c2 = s[bpp * x + 2];
        const unsigned b = rgb ? c2 : c0, r = rgb ? c0 : c2;
        d[x] = (uint8_t)((b * 1868u + c1 * 9617u + r * 4899u + (1u << 13)) >> 14);
      }
    }
  }
  return 0;
}

// forensics (tests/forensics.py): the votes of the hypotheses [item_lo, item_hi) of initialise()'s loop nest only
int orc_vote_items(const double* det, int n_det, const double* markers, int n_markers, const double K[9], double tol,
                   long long item_lo, long long item_hi, uint32_t* hist) {
  Estimator e;
  e.setCamera(K);
  e.setMarkerPositions(markers, n_markers);
  e.back_projection_pixel_tolerance_ = tol;
  e.setImagePoints(det, n_det);
  std::vector<unsigned> h;
  e.voteHistogram(h, item_lo, item_hi < 0 ? 0 : item_hi);
  std::memcpy(hist, h.data(), h.size() * sizeof(unsigned));
  return 0;
}

int orc_vote_histogram(const double* det, int n_det, const double* markers, int n_markers,
                       const double K[9], double tol, uint32_t* hist) {
  Estimator e;
  e.setCamera(K);
  e.setMarkerPositions(markers, n_markers);
  e.back_projection_pixel_tolerance_ = tol;
  e.setImagePoints(det, n_det);
  std::vector<unsigned> h;
  e.voteHistogram(h);
  std::memcpy(hist, h.data(), h.size() * sizeof(unsigned));
  return 0;
}

int orc_correspondences_from_histogram(uint32_t* hist, int n_det, int n_markers,
                                       unsigned histogram_threshold, uint32_t* corr) {
  Estimator e;
  e.histogram_threshold_ = histogram_threshold;
  std::vector<unsigned> h(hist, hist + (size_t)n_det * n_markers);
  e.correspondencesFromHistogram(h, n_det, n_markers);
  std::memcpy(hist, h.data(), h.size() * sizeof(unsigned));
  std::memcpy(corr, e.corr_.data(), e.corr_.size() * sizeof(unsigned));
  return (int)e.corr_.size() / 2;
}

int orc_check_correspondences(const double* det, int n_det, const double* markers, int n_markers,
                              const double K[9], const orc_params* p, const uint32_t* corr,
                              int n_corr, double T[16]) {
  Estimator e;
  e.setCamera(K);
  e.setMarkerPositions(markers, n_markers);
  fill_params(e, p, n_markers);
  e.setImagePoints(det, n_det);
  e.corr_.assign(corr, corr + 2 * n_corr);
  unsigned ok = e.checkCorrespondences();
  if (ok) std::memcpy(T, e.predicted_pose_.m, sizeof(e.predicted_pose_.m));
  return (int)ok;
}

int orc_optimise_pose(const double* det, const double* markers, const double K[9],
                      const uint32_t* corr, int n_corr, double T[16], double cov[36]) {
  Estimator e;
  e.setCamera(K);
  int max_m = 0, max_d = 0;
  for (int i = 0; i < n_corr; ++i) {
    max_m = std::max<int>(max_m, corr[2 * i]);
    max_d = std::max<int>(max_d, corr[2 * i + 1]);
  }
  e.setMarkerPositions(markers, max_m);
  e.setImagePoints(det, max_d);
  e.corr_.assign(corr, corr + 2 * n_corr);
  std::memcpy(e.predicted_pose_.m, T, sizeof(e.predicted_pose_.m));
  e.optimisePose();
  std::memcpy(T, e.predicted_pose_.m, sizeof(e.predicted_pose_.m));
  std::memcpy(cov, e.pose_covariance_, sizeof(e.pose_covariance_));
  return e.gn_iterations_;
}

int orc_solve_bruteforce(const double* det, int n_det, const double* markers, int n_markers,
                         const double K[9], const orc_params* p, orc_result* out, uint32_t* hist,
                         uint32_t* corr) {
  return solve_bruteforce(det, n_det, markers, n_markers, K, p, out, hist, corr);
}

int orc_estimate_frame(const uint8_t* img, int rows, int cols, size_t stride,
                       const double* markers, int n_markers, const double K[9], const double* D,
                       int nD, const orc_params* p, orc_result* out) {
  // estimateBodyPose, uninitialised branch: ROI = whole image (PE.cpp:68-78)
  std::vector<double> u;
  std::vector<float> d;
  int n = find_leds(img, rows, cols, stride, 0, 0, cols, rows, *p, K, D, nD, u, d);
  if (n < 0) return n;
  return solve_bruteforce(u.data(), n, markers, n_markers, K, p, out, nullptr, nullptr);
}

int orc_estimate_batch(const uint8_t* frames, int n_frames, int rows, int cols, size_t stride,
                       size_t frame_stride, const double* markers, int n_markers,
                       const double K[9], const double* D, int nD, const orc_params* p,
                       orc_result* out, int n_threads) {
  if (n_threads < 1) n_threads = 1;
  std::vector<std::thread> th;
  std::vector<int> rc(n_threads, 0);
  for (int t = 0; t < n_threads; ++t) {
    th.emplace_back([=, &rc]() {
      for (int f = t; f < n_frames; f += n_threads) {
        int r = orc_estimate_frame(frames + (size_t)f * frame_stride, rows, cols, stride, markers,
                                   n_markers, K, D, nD, p, &out[f]);
        if (r < 0) rc[t] = r;
      }
    });
  }
  for (auto& x : th) x.join();
  for (int r : rc)
    if (r < 0) return r;
  return 0;
}


// ---- stateful estimator (tracking path) ----
struct orc_tracker {
  Tracker t;
};

orc_tracker* orc_tracker_create(const double* markers, int n_markers, const double K[9], const double* D, int nD,
                                const orc_params* p) {
  orc_tracker* tr = new orc_tracker();
  tr->t.p = *p;
  for (int i = 0; i < 9; ++i) tr->t.Kc[i] = K[i];
  tr->t.D.assign(D, D + nD);
  tr->t.e.setCamera(K);
  tr->t.e.setMarkerPositions(markers, n_markers);
  fill_params(tr->t.e, p, n_markers);
  return tr;
}
void orc_tracker_destroy(orc_tracker* tr) { delete tr; }

/* returns 1 (pose updated), 0 (not), <0 error.  info[8] = {roi x,y,w,h, it_since_initialized, n_det, n_corr,
 * used_bruteforce} */
int orc_tracker_estimate(orc_tracker* tr, const uint8_t* img, int rows, int cols, size_t stride, double time,
                         orc_result* out, int* info) {
  Tracker& t = tr->t;
  int rc = t.estimateBodyPose(img, rows, cols, stride, time);
  if (rc < 0) return rc;
  if (out) write_result(t.e, rc == 1, (int)t.det_.size() / 2, out);
  if (info) {
    info[0] = t.roi_.x;
    info[1] = t.roi_.y;
    info[2] = t.roi_.width;
    info[3] = t.roi_.height;
    info[4] = (int)t.it_since_initialized_;
    info[5] = (int)t.det_.size() / 2;
    info[6] = (int)t.e.corr_.size() / 2;
    info[7] = t.last_used_bruteforce_;
  }
  return rc;
}

void orc_logarithm_map(const double T[16], double xi[6]) {
  M4 M;
  std::memcpy(M.m, T, sizeof(M.m));
  Tracker::logarithmMap(M, xi);
}


/* voting histograms for a batch of detection sets (det: n x max_det x 2, hist: n x max_det x n_markers),
 * frame-parallel */
int orc_vote_batch(const double* det, const int* n_det, int n, int max_det, const double* markers, int n_markers,
                   const double K[9], double tol, uint32_t* hist, int n_threads) {
  if (n_threads < 1) n_threads = 1;
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; ++t)
    th.emplace_back([=]() {
      for (int f = t; f < n; f += n_threads) {
        uint32_t* h = hist + (size_t)f * max_det * n_markers;
        std::memset(h, 0, sizeof(uint32_t) * (size_t)max_det * n_markers);
        if (n_det[f] >= 4) orc_vote_histogram(det + (size_t)f * max_det * 2, n_det[f], markers, n_markers, K, tol, h);
      }
    });
  for (auto& x : th) x.join();
  return 0;
}

}  // extern "C"
