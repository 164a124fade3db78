//@file-prologue
// mpe_k3.hip — K3 validate + refine, primitive batches, frame decode, spin (see mpe_kernels_common.h for the map of the kernel sources)
#include "mpe_kernels_common.h"
#include "mpe_k2_head.h"
#include "mpe_k1b_dev.h"  // (the blob extraction's device functions: k_track_frame below runs a whole tracked frame)
namespace mpe {
//@file-prologue-end
// =============================================================================================
// K3 — per-frame tail: correspondences from the histogram, validation, Kabsch, Gauss-Newton
// =============================================================================================
struct T34 {  // rigid transform rows [R | t]
  double m[3][4];
};

__device__ __forceinline__ void project_T(const T34& T, const double* mk, double fx, double fy, double cx, double cy,
                                          double& u, double& v, double& X, double& Y, double& Z) {
  X = T.m[0][0] * mk[0] + T.m[0][1] * mk[1] + T.m[0][2] * mk[2] + T.m[0][3];
  Y = T.m[1][0] * mk[0] + T.m[1][1] * mk[1] + T.m[1][2] * mk[2] + T.m[1][3];
  Z = T.m[2][0] * mk[0] + T.m[2][1] * mk[1] + T.m[2][2] * mk[2] + T.m[2][3];
  u = (fx * X + cx * Z) / Z;
  v = (fy * Y + cy * Z) / Z;
}

// R = V U^T of H = U S V^T (computeTransformation, pose_estimator.cpp:916-922: JacobiSVD, no reflection
// guard) by a one-sided (Hestenes) Jacobi SVD: the columns of G = H V are rotated pairwise until they are
// orthogonal, U = G with normalised columns.  Same sweeps, thresholds and operation order as the CPU
// oracle's svd3.  A rank-deficient H (coplanar markers) leaves one column of G at rounding level; the
// reference's JacobiSVD gives that column of U the sign of its rounding residue (a coin flip between the
// rotation and its mirror image through the marker plane) — here, as in the oracle, every sigma <= 1e-12
// sigma_max is completed with the cross product of the other two columns (det U = +1).
__device__ void kabsch_rotation(const double Hm[3][3], double R[3][3]) {
  double G[3][3], V[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      G[i][j] = Hm[i][j];
      V[i][j] = (i == j) ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 60; ++sweep) {
    bool rotated = false;
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;  // (0,1) (0,2) (1,2)
      double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        alpha += G[i][p] * G[i][p];
        beta += G[i][q] * G[i][q];
        gamma += G[i][p] * G[i][q];
      }
      if (gamma == 0.0 || fabs(gamma) <= 1e-300 + 2.3e-16 * sqrt(alpha * beta)) continue;
      rotated = true;
      const double zeta = (beta - alpha) / (2.0 * gamma);
      const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
      const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double gp = G[i][p], gq = G[i][q];
        G[i][p] = c * gp - sn * gq;
        G[i][q] = sn * gp + c * gq;
        const double vp = V[i][p], vq = V[i][q];
        V[i][p] = c * vp - sn * vq;
        V[i][q] = sn * vp + c * vq;
      }
    }
    if (!rotated) break;
  }
  double sv[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) sv[j] = sqrt(G[0][j] * G[0][j] + G[1][j] * G[1][j] + G[2][j] * G[2][j]);
  const double smax = fmax(sv[0], fmax(sv[1], sv[2]));
  double U[3][3];
  int zero_col = -1, nzero = 0;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (sv[j] > smax * 1e-12 && sv[j] > 0) {
#pragma unroll
      for (int i = 0; i < 3; ++i) U[i][j] = G[i][j] / sv[j];
    } else {
#pragma unroll
      for (int i = 0; i < 3; ++i) U[i][j] = 0.0;
      zero_col = j;
      ++nzero;
    }
  }
  if (nzero == 1) {  // rank 2: u_zero = u_a x u_b, (zero, a, b) cyclic
#pragma unroll
    for (int z = 0; z < 3; ++z)
      if (z == zero_col) {
        const int a = (z + 1) % 3, b = (z + 2) % 3;
        U[0][z] = U[1][a] * U[2][b] - U[2][a] * U[1][b];
        U[1][z] = U[2][a] * U[0][b] - U[0][a] * U[2][b];
        U[2][z] = U[0][a] * U[1][b] - U[1][a] * U[0][b];
      }
  } else if (nzero > 1) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) U[i][j] = (i == j) ? 1.0 : 0.0;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double acc = V[i][0] * U[j][0];  // (V U^T)(i,j) = sum_k V(i,k) U(j,k), k ascending like the oracle's mul()
      acc += V[i][1] * U[j][1];
      acc += V[i][2] * U[j][2];
      R[i][j] = acc;
    }
}

// unpivoted LDL^T of a symmetric positive definite 6x6 (normal equations of GN)
struct LDL6 {
  double L[6][6];
  double D[6];
};
__device__ __forceinline__ void ldl6_factor(const double A[6][6], LDL6& F) {
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double d = A[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= F.L[j][k] * F.L[j][k] * F.D[k];
    F.D[j] = d;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double s = A[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= F.L[i][k] * F.L[j][k] * F.D[k];
      F.L[i][j] = s / d;
    }
  }
}
__device__ __forceinline__ void ldl6_solve(const LDL6& F, const double b[6], double x[6]) {
  double y[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= F.L[i][k] * y[k];
    y[i] = s;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) y[i] /= F.D[i];
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double s = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) s -= F.L[k][i] * x[k];
    x[i] = s;
  }
}

// exponentialMap(dT) * T   (pose_estimator.cpp:781, 962-994)
__device__ __forceinline__ void apply_exp(const double tw[6], T34& T) {
  const double ux = tw[0], uy = tw[1], uz = tw[2], wx = tw[3], wy = tw[4], wz = tw[5];
  const double theta = sqrt(wx * wx + wy * wy + wz * wz);
  const double th2 = theta * theta;
  double Rm[3][3], Vm[3][3];
  const double O[3][3] = {{0, -wz, wy}, {wz, 0, -wx}, {-wy, wx, 0}};
  double O2[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) O2[i][j] = O[i][0] * O[0][j] + O[i][1] * O[1][j] + O[i][2] * O[2][j];
  if (theta == 0) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Rm[i][j] = Vm[i][j] = (i == j) ? 1.0 : 0.0;
  } else {
    double st, ct;
    sincos(theta, &st, &ct);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        const double I = (i == j) ? 1.0 : 0.0;
        Rm[i][j] = I + O[i][j] / theta * st + O2[i][j] / th2 * (1 - ct);
        Vm[i][j] = I + (1 - ct) / th2 * O[i][j] + (theta - st) / (th2 * theta) * O2[i][j];
      }
  }
  const double t0 = Vm[0][0] * ux + Vm[0][1] * uy + Vm[0][2] * uz;
  const double t1 = Vm[1][0] * ux + Vm[1][1] * uy + Vm[1][2] * uz;
  const double t2 = Vm[2][0] * ux + Vm[2][1] * uy + Vm[2][2] * uz;
  const double tv[3] = {t0, t1, t2};
  T34 N;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 4; ++j) {
      double s = Rm[i][0] * T.m[0][j] + Rm[i][1] * T.m[1][j] + Rm[i][2] * T.m[2][j];
      if (j == 3) s += tv[i];
      N.m[i][j] = s;
    }
  }
  T = N;
}

// optimisePose (pose_estimator.cpp:733-792): Gauss-Newton on SE(3) over the n_c correspondence rows
// row(j, 0..2) = marker xyz, row(j, 3..4) = detection uv; T is updated in place, cov = A^-1 of the last iteration
// (row-major 6x6), returns the number of iterations.
template <class Row>
__device__ __forceinline__ int k3_gauss_newton(int n_c, Row row, double fx, double fy, double cx, double cy, T34& T,
                                               double* __restrict__ cov) {
  double A[6][6];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) A[r][c] = 0;
  int iters = 0;
  for (int it = 0; it < 500; ++it) {
    double b[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c) A[r][c] = 0;
    for (int j = 0; j < n_c; ++j) {
      const double mk[3] = {row(j, 0), row(j, 1), row(j, 2)};
      double u, v, x, y, z;
      project_T(T, mk, fx, fy, cx, cy, u, v, x, y, z);
      const double e0 = row(j, 3) - u, e1 = row(j, 4) - v;
      const double z_2 = z * z;
      // computeJacobian, pose_estimator.cpp:945-957
      double J0[6] = {0, 0, 0, 0, 0, 0}, J1[6] = {0, 0, 0, 0, 0, 0};
      J0[0] = 1 / z * fx;
      J0[2] = -x / z_2 * fx;
      J0[3] = -x * y / z_2 * fx;
      J0[4] = (1 + (x * x / z_2)) * fx;
      J0[5] = -y / z * fx;
      J1[1] = 1 / z * fy;
      J1[2] = -y / z_2 * fy;
      J1[3] = -(1 + y * y / z_2) * fy;
      J1[4] = x * y / z_2 * fy;
      J1[5] = x / z * fy;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
#pragma unroll
        for (int c = r; c < 6; ++c) A[r][c] += J0[r] * J0[c] + J1[r] * J1[c];  // A += J^T J (upper triangle)
        b[r] += J0[r] * e0 + J1[r] * e1;                                       // b += J^T e
      }
    }
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < r; ++c) A[r][c] = A[c][r];
    LDL6 F;
    ldl6_factor(A, F);
    double dT[6];
    ldl6_solve(F, b, dT);
    apply_exp(dT, T);
    iters = it + 1;
    double mx = -1;  // norm_max, pose_estimator.cpp:1073-1085
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double av = fabs(dT[r]);
      if (av > mx) mx = av;
    }
    if (mx <= 1e-13) break;
  }
  // pose_covariance_ = A.inverse() with the A of the last iteration (pose_estimator.cpp:790)
  LDL6 F;
  ldl6_factor(A, F);
#pragma unroll 1
  for (int c = 0; c < 6; ++c) {
    double e[6], x[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) e[r] = (r == c) ? 1.0 : 0.0;
    ldl6_solve(F, e, x);
#pragma unroll
    for (int r = 0; r < 6; ++r) cov[r * 6 + c] = x[r];
  }
  return iters;
}

#define K3_GROUP 16                 // lanes cooperating on one frame in the validation kernel
#define K3_FRAMES_PER_BLOCK 4       // one wave = 4 frames
#define K3B_THREADS 64              // refinement kernel: one lane per frame

// What the validation kernel hands to the refinement kernel, one record per frame (global memory).
struct TailMid {
  int n_c;             // rows of correspondences_
  int active;          // 1: computeTransformation / optimisePose run for this frame
  unsigned num_valid;  // P3P triples with a valid solution (pose_estimator.cpp:506)
  unsigned pad;
  unsigned char cm[MPE_MAX_MARKERS], cd[MPE_MAX_MARKERS];  // rows (marker, detection), 1-based
  double mean[3 * MPE_MAX_MARKERS];  // sum over the valid triples of inverse(H_best) * marker (not yet divided)
};
size_t k3_mid_bytes(int n_frames) { return (size_t)(n_frames > 0 ? n_frames : 1) * sizeof(TailMid); }

// ---------------------------------------------------------------------------------------------
// K3a  k3a_validate: 16 lanes per frame (4 frames per wave).  Lane 0 of a group builds the correspondences
// (histogram peeling, given rows, or nearest neighbour), then the C(n_c,3) P3P validations of
// checkCorrespondences run 16 at a time, one per lane, and are summed in combination order.
// MODE 0 / 1: as described.  MODE 2 (optimisePose alone): only the rows are parsed, no validation.
// ---------------------------------------------------------------------------------------------
// (the body is a device function of one 64-thread block `blk`: k3a_validate is that block as a kernel, k_track_frame
//  runs it behind the blob extraction of a tracked frame inside one launch)
template <int MODE>
__device__ __forceinline__ void k3a_body(const mpe_detections* __restrict__ dets, const uint32_t* __restrict__ hist,
                                         int n_frames, const SolveParams& sp, mpe_result* __restrict__ results,
                                         uint32_t* __restrict__ corr_out, const uint32_t* __restrict__ corr_in,
                                         const double* __restrict__ nn_pred, double nn_tol, TailMid* __restrict__ mid,
                                         const int blk) {
  // dynamic LDS, sized for the actual marker count: per-lane contributions [4][16][3 n_m] and
  // back-projections [2 (rows - 3)][64]
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  const int nm3 = 3 * sp.n_markers;
  double* s_part_ = reinterpret_cast<double*>(smem3);
  double* s_q_ = s_part_ + K3_FRAMES_PER_BLOCK * K3_GROUP * nm3;
#define s_part(g_, l_, i_) s_part_[((g_)*K3_GROUP + (l_)) * nm3 + (i_)]
#define s_q(j_, t_) s_q_[(j_)*64 + (t_)]
  __shared__ double s_mean[K3_FRAMES_PER_BLOCK][MPE_MAX_MARKERS * 3];
  __shared__ double s_det[K3_FRAMES_PER_BLOCK][MPE_MAX_DETECTIONS][2];
  __shared__ double s_pred[K3_FRAMES_PER_BLOCK][MPE_MAX_MARKERS][2];
  __shared__ unsigned s_colmax[K3_FRAMES_PER_BLOCK][MPE_MAX_MARKERS];
  __shared__ unsigned char s_colrow[K3_FRAMES_PER_BLOCK][MPE_MAX_MARKERS];
  __shared__ double s_mk[MPE_MAX_MARKERS][3];
  __shared__ unsigned s_valid[K3_FRAMES_PER_BLOCK][K3_GROUP];
  __shared__ unsigned char s_cm[K3_FRAMES_PER_BLOCK][MPE_MAX_MARKERS], s_cd[K3_FRAMES_PER_BLOCK][MPE_MAX_MARKERS];
  __shared__ int s_nc[K3_FRAMES_PER_BLOCK];

  const int tid = threadIdx.x;
  const int grp = tid >> 4, l = tid & 15;
  const int f = blk * K3_FRAMES_PER_BLOCK + grp;
  const bool live = f < n_frames;
  const mpe_detections* d = dets + (live ? f : 0);
  mpe_result* res = results + (live ? f : 0);
  const int n_d = live ? d->n : 0, n_m = sp.n_markers;
  const int dstatus = live ? d->status : 0;
  const uint32_t* H = hist + (size_t)(live ? f : 0) * MPE_HIST_STRIDE;
  const double fx = sp.fx, fy = sp.fy, cx = sp.cx, cy = sp.cy;

  // stage markers and detections in LDS; default output = identity pose, zero covariance
  if (tid < n_m) {
    s_mk[tid][0] = sp.markers[3 * tid];
    s_mk[tid][1] = sp.markers[3 * tid + 1];
    s_mk[tid][2] = sp.markers[3 * tid + 2];
  }
  for (int i = l; i < n_d; i += K3_GROUP) {
    s_det[grp][i][0] = d->undist_xy[2 * i];
    s_det[grp][i][1] = d->undist_xy[2 * i + 1];
  }
  if (nn_pred && live)  // (tracking path: lane 0's nearest-neighbour search below reads LDS, not a chain of global loads)
    for (int i = l; i < n_m; i += K3_GROUP) {
      s_pred[grp][i][0] = nn_pred[(size_t)f * 2 * MPE_MAX_MARKERS + 2 * i];
      s_pred[grp][i][1] = nn_pred[(size_t)f * 2 * MPE_MAX_MARKERS + 2 * i + 1];
    }
  if (live && !corr_in && !nn_pred && MODE != 2)  // (the histogram path: see lane 0 below)
    for (int c = l; c < n_m; c += K3_GROUP) {
      unsigned mv = 0, mr = 0;
      for (int r = 0; r < n_d; ++r) {
        const unsigned v = H[r * MPE_MAX_MARKERS + c];
        if (v > mv) {  // (first row of the column's maximum; an all-zero column keeps row 0)
          mv = v;
          mr = (unsigned)r;
        }
      }
      s_colmax[grp][c] = mv;
      s_colrow[grp][c] = (unsigned char)mr;
    }
  wave_sync();  // (the block is one wave)
  if (live && MODE != 2) {  // (MODE 2: results[f].T holds the start pose for the refinement kernel)
    for (int i = l; i < 16; i += K3_GROUP) res->T[i] = (i % 5 == 0) ? 1.0 : 0.0;
  }
  if (live) {
    for (int i = l; i < 36; i += K3_GROUP) res->cov[i] = 0.0;
    if (corr_out)
      for (int i = l; i < 2 * MPE_MAX_MARKERS; i += K3_GROUP) corr_out[(size_t)f * 2 * MPE_MAX_MARKERS + i] = 0;
  }

  // ---- lane 0 of the group: initialise()'s all-zero test (pose_estimator.cpp:704) and
  //      correspondencesFromHistogram (pose_estimator.cpp:344-370)
  if (l == 0) {
    int n_c = 0;
    bool go0 = live && dstatus == 0 && n_d >= 4 && n_m >= 4;
    if (MODE == 2) go0 = live && dstatus == 0 && n_d >= 1 && n_m >= 1;
    if (go0 && corr_in) {
      // tracking path: correspondences come from findCorrespondences (pose_estimator.cpp:372-392),
      // rows (marker, detection) terminated by a 0 marker; checkCorrespondences starts from them
      const uint32_t* ci = corr_in + (size_t)f * 2 * MPE_MAX_MARKERS;
      while (n_c < MPE_MAX_MARKERS && ci[2 * n_c] != 0) {
        s_cm[grp][n_c] = (unsigned char)ci[2 * n_c];
        s_cd[grp][n_c] = (unsigned char)ci[2 * n_c + 1];
        ++n_c;
      }
      go0 = false;
    }
    if (go0 && nn_pred) {
      // tracking path, correspondences found here: findCorrespondences (pose_estimator.cpp:372-392) —
      // nearest detection of every predicted marker pixel (first minimum wins), kept if within
      // nearest_neighbour_pixel_tolerance_
      for (int i = 0; i < n_m; ++i) {
        double best = __builtin_huge_val();
        int bj = 0;
        const double pu = s_pred[grp][i][0], pv = s_pred[grp][i][1];
        for (int j = 0; j < n_d; ++j) {
          const double du = pu - s_det[grp][j][0], dv = pv - s_det[grp][j][1];
          const double d2 = du * du + dv * dv;
          if (d2 < best) {
            best = d2;
            bj = j + 1;
          }
        }
        if (sqrt(best) <= nn_tol) {
          s_cm[grp][n_c] = (unsigned char)(i + 1);
          s_cd[grp][n_c] = (unsigned char)bj;
          ++n_c;
        }
      }
      go0 = false;
    }
    if (go0) {
      // The reference scans the whole histogram n_m times, column-major, for the first position of the maximum and
      // then zeroes that COLUMN (pose_estimator.cpp:349-368).  Only columns are ever removed, so a column's maximum
      // and the first row that reaches it never change: the group's lanes found them above (one column per lane, n_d
      // independent loads each instead of n_m * n_m * n_d dependent ones here), and a round is the first column, in
      // ascending order, with the largest value still standing.  A removed column stands at 0 with row 0, as in
      // the reference's scan, which matters when hist_thr is 0.
      bool any = false;
      for (int c = 0; c < n_m; ++c) any |= (s_colmax[grp][c] != 0);
      go0 = any;
    }
    if (go0) {
      unsigned removed = 0;  // zeroed columns
      for (int j = 0; j < n_m; ++j) {
        unsigned mv = 0;
        int ri = 0, ci = 0;
        bool first = true;
        for (int c = 0; c < n_m; ++c) {
          const bool gone = (removed >> c) & 1;
          const unsigned v = gone ? 0u : s_colmax[grp][c];
          if (first || v > mv) {
            mv = v;
            ri = gone ? 0 : (int)s_colrow[grp][c];
            ci = c;
            first = false;
          }
        }
        if (mv < sp.hist_thr) break;
        s_cm[grp][n_c] = (unsigned char)(ci + 1);
        s_cd[grp][n_c] = (unsigned char)(ri + 1);
        ++n_c;
        removed |= 1u << ci;
      }
    }
    s_nc[grp] = n_c;
    if (live) {
      res->n_det = n_d;
      res->n_corr = n_c;
      res->gn_iterations = 0;
      res->status = (dstatus != 0) ? dstatus : MPE_FRAME_NO_POSE;
      if (corr_out)
        for (int i = 0; i < n_c; ++i) {
          corr_out[(size_t)f * 2 * MPE_MAX_MARKERS + 2 * i] = s_cm[grp][i];
          corr_out[(size_t)f * 2 * MPE_MAX_MARKERS + 2 * i + 1] = s_cd[grp][i];
        }
    }
  }
  __syncthreads();
  const int n_c = s_nc[grp];
  const bool go = (MODE != 2) && n_c >= 4;

  // ---- checkCorrespondences (pose_estimator.cpp:394-542): the C(n_c,3) P3P validations run 16 at a time,
  //      one per lane of the group; after every round the lanes' inverse(H_best) * markers are added to the
  //      running sums IN COMBINATION ORDER (lane 0 first), i.e. in the reference's summation order for any n_c
  const int nu = n_c - 3;
  const int N = go ? n_c * (n_c - 1) * (n_c - 2) / 6 : 0;
  for (int v = l; v < nm3; v += K3_GROUP) s_mean[grp][v] = 0.0;
  unsigned num_valid = 0;
  for (int r0 = 0; __any(r0 < N); r0 += K3_GROUP) {  // wave-uniform trip count (the barriers below)
    const int ci = r0 + l;
    bool contributes = false;
    do {
      if (ci >= N) break;
      int a, b, c;
      unrank_combo3(ci, n_c, a, b, c);
      V3 fv[3], wp[3];
      {
        const int rows3[3] = {a, b, c};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int mi = s_cm[grp][rows3[k]] - 1, di = s_cd[grp][rows3[k]] - 1;
          wp[k] = {s_mk[mi][0], s_mk[mi][1], s_mk[mi][2]};
          fv[k] = bearing(s_det[grp][di][0], s_det[grp][di][1], fx, fy, cx, cy);
        }
      }
      P3PCtx ctx;
      if (!p3p_prepare(fv[0], fv[1], fv[2], wp[0], wp[1], wp[2], ctx)) break;
      double min_sq = INFINITY;
      int best = -1;
      bool found = false;
#pragma unroll 1
      for (int k = 0; k < 4; ++k) {
        M3 R;
        V3 C;
        p3p_solution(ctx, pick_root(ctx, k), R, C);
        if (!rc_finite(R, C)) continue;
        const Proj P = make_projection(R, C, fx, fy, cx, cy);
        // back-project the unused correspondences' markers, ascending row index
        for (int q = 0; q < nu; ++q) {
          int row = q;
          row += (row >= a);
          row += (row >= b);
          row += (row >= c);
          const int mi = s_cm[grp][row] - 1;
          double u, v;
          project(P, V3{s_mk[mi][0], s_mk[mi][1], s_mk[mi][2]}, u, v);
          s_q(2 * q, tid) = u;
          s_q(2 * q + 1, tid) = v;
        }
        // calculateSquaredReprojectionErrorAndCertainty (pose_estimator.cpp:303-342): greedy
        // global-minimum matching, column-major first minimum, rows = image points
        unsigned rowdone = 0, coldone = 0;
        double sq = 0;
        unsigned ncorr = 0;
        for (int it = 0; it < nu; ++it) {
          double mv = 0;
          int ri = 0, cj0 = 0;
          bool first = true;
          for (int cj = 0; cj < nu; ++cj) {
            const double bu = s_q(2 * cj, tid), bv = s_q(2 * cj + 1, tid);
            for (int rr = 0; rr < nu; ++rr) {
              double v;
              if (((rowdone >> rr) & 1) || ((coldone >> cj) & 1))
                v = INFINITY;
              else {
                int row = rr;
                row += (row >= a);
                row += (row >= b);
                row += (row >= c);
                const int di = s_cd[grp][row] - 1;
                const double du = s_det[grp][di][0] - bu, dv = s_det[grp][di][1] - bv;
                v = sqrt(du * du + dv * dv);
              }
              if (first || v < mv) {
                mv = v;
                ri = rr;
                cj0 = cj;
                first = false;
              }
            }
          }
          if (mv <= sp.back_tol) {
            sq += mv * mv;
            ++ncorr;
            rowdone |= 1u << ri;
            coldone |= 1u << cj0;
          } else
            break;
        }
        const double certainty = (double)ncorr / (double)nu;
        if (certainty >= sp.certainty_thr) {  // pose_estimator.cpp:494-502
          found = true;
          if (sq < min_sq) {
            min_sq = sq;
            best = k;
          }
        }
      }
      if (!found) break;
      if (best < 0) best = 0;  // unreachable: sq is always finite
      M3 R;
      V3 C;
      p3p_solution(ctx, pick_root(ctx, best), R, C);
      // inverse(H) * marker for ALL markers (pose_estimator.cpp:513-517)
      for (int jj = 0; jj < n_m; ++jj) {
        const V3 mk = {s_mk[jj][0] - C.x, s_mk[jj][1] - C.y, s_mk[jj][2] - C.z};
        const V3 pc = mulT(R, mk);  // R^T (m - C)
        s_part(grp, l, 3 * jj) = pc.x;
        s_part(grp, l, 3 * jj + 1) = pc.y;
        s_part(grp, l, 3 * jj + 2) = pc.z;
      }
      contributes = true;
    } while (false);
    s_valid[grp][l] = contributes ? 1u : 0u;
    __syncthreads();
    for (int v = l; v < nm3; v += K3_GROUP) {
      double sacc = s_mean[grp][v];
      for (int q = 0; q < K3_GROUP; ++q)
        if (s_valid[grp][q]) sacc += s_part(grp, q, v);
      s_mean[grp][v] = sacc;
    }
    for (int q = 0; q < K3_GROUP; ++q) num_valid += s_valid[grp][q];
    __syncthreads();
  }
  bool active = go && ((double)num_valid / (double)N >= sp.valid_corr_thr);
  if (MODE == 2) active = n_c >= 3;  // fewer rows leave the 6x6 normal equations singular
  if (live) {
    TailMid* m = mid + f;
    if (l == 0) {
      m->n_c = n_c;
      m->active = active ? 1 : 0;
      m->num_valid = num_valid;
      m->pad = 0;
    }
    if (l < n_c) {
      m->cm[l] = s_cm[grp][l];
      m->cd[l] = s_cd[grp][l];
    }
    if (active)
      for (int v = l; v < nm3; v += K3_GROUP) m->mean[v] = s_mean[grp][v];
  }
#undef s_part
#undef s_q
}
template <int MODE>
__global__ __launch_bounds__(64) void k3a_validate(const mpe_detections* __restrict__ dets,
                                                   const uint32_t* __restrict__ hist, int n_frames, SolveParams sp,
                                                   mpe_result* __restrict__ results, uint32_t* __restrict__ corr_out,
                                                   const uint32_t* __restrict__ corr_in,
                                                   const double* __restrict__ nn_pred, double nn_tol,
                                                   TailMid* __restrict__ mid) {
  k3a_body<MODE>(dets, hist, n_frames, sp, results, corr_out, corr_in, nn_pred, nn_tol, mid, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// K3b  k3b_refine: ONE LANE PER FRAME.  computeTransformation (Kabsch, pose_estimator.cpp:908-930), then
// optimisePose (pose_estimator.cpp:733-792): Gauss-Newton on SE(3), the normal equations accumulated over
// the correspondences one after the other in row order — the reference's summation order —, unpivoted LDL^T,
// exponentialMap update, covariance = inverse of the last iteration's A (pose_estimator.cpp:790).
// Everything is lane-local: no shuffles, no barriers inside the iteration; lanes of frames without a pose
// (or whose iteration has converged) idle.  A 16 384-frame sub-batch is 256 waves of ~8 k instructions.
// MODE 0: Kabsch + GN.  MODE 1: Kabsch only (checkCorrespondences alone).  MODE 2: GN from results[f].T.
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(K3B_THREADS) void k3b_refine(const mpe_detections* __restrict__ dets, int n_frames,
                                                          SolveParams sp, mpe_result* __restrict__ results,
                                                          const TailMid* __restrict__ mid, int row_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3b[];
  // per-lane rows of (marker xyz, detection uv), [row][field][lane]: conflict-free, dynamic row index
  double* s_row = reinterpret_cast<double*>(smem3b);
#define ROW(r_, k_) s_row[((r_)*5 + (k_)) * K3B_THREADS + threadIdx.x]
  const int f = blockIdx.x * K3B_THREADS + threadIdx.x;
  if (f >= n_frames) return;
  const TailMid* m = mid + f;
  if (!m->active) return;
  const mpe_detections* d = dets + f;
  mpe_result* res = results + f;
  const int n_m = sp.n_markers;
  const int n_c = min(m->n_c, row_cap);
  const double fx = sp.fx, fy = sp.fy, cx = sp.cx, cy = sp.cy;
  for (int j = 0; j < n_c; ++j) {
    const int mi = m->cm[j] - 1, di = m->cd[j] - 1;
    ROW(j, 0) = sp.markers[3 * mi];
    ROW(j, 1) = sp.markers[3 * mi + 1];
    ROW(j, 2) = sp.markers[3 * mi + 2];
    ROW(j, 3) = d->undist_xy[2 * di];
    ROW(j, 4) = d->undist_xy[2 * di + 1];
  }
  T34 T;
  if (MODE == 2) {
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) T.m[r][c] = res->T[r * 4 + c];
  } else {
    // ---- computeTransformation (pose_estimator.cpp:908-930)
    const double nv = (double)m->num_valid;
    double mo[3] = {0, 0, 0}, mr[3] = {0, 0, 0};
    for (int i = 0; i < n_m; ++i)
      for (int k = 0; k < 3; ++k) {
        mo[k] += sp.markers[3 * i + k];
        mr[k] += m->mean[3 * i + k] / nv;
      }
    for (int k = 0; k < 3; ++k) {
      mo[k] /= (double)n_m;
      mr[k] /= (double)n_m;
    }
    double Hm[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int i = 0; i < n_m; ++i) {
      const double a[3] = {sp.markers[3 * i] - mo[0], sp.markers[3 * i + 1] - mo[1], sp.markers[3 * i + 2] - mo[2]};
      const double b[3] = {m->mean[3 * i] / nv - mr[0], m->mean[3 * i + 1] / nv - mr[1], m->mean[3 * i + 2] / nv - mr[2]};
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) Hm[r][c] += a[r] * b[c];
    }
    double X[3][3];
    kabsch_rotation(Hm, X);  // R = V U^T
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) T.m[r][c] = X[r][c];
      T.m[r][3] = mr[r] - (X[r][0] * mo[0] + X[r][1] * mo[1] + X[r][2] * mo[2]);
    }
  }

  // ---- optimisePose (pose_estimator.cpp:733-792)
  int iters = 0;
  if (MODE != 1)
    iters = k3_gauss_newton(n_c, [&](int j, int k) -> double { return ROW(j, k); }, fx, fy, cx, cy, T, res->cov);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) res->T[r * 4 + c] = T.m[r][c];
  res->T[12] = 0.0;
  res->T[13] = 0.0;
  res->T[14] = 0.0;
  res->T[15] = 1.0;
  res->gn_iterations = iters;
  res->status = MPE_FRAME_POSE;
#undef ROW
}

// ---------------------------------------------------------------------------------------------
// K3b for SMALL launches (tracked frames, a handful of streams in lock step): the same computeTransformation +
// optimisePose with 16 LANES PER FRAME.  One lane per frame (k3b_refine) is the right shape for 16 384 frames, but a
// single tracked frame then waits for ~18 000 dependent FP64 instructions of one lane (54 us).  Here a Gauss-Newton
// iteration is spread over the group: lane j computes the Jacobian rows of correspondence j, 27 lanes-slots sum the
// 21 + 6 entries of A = sum J^T J and b = sum J^T e over the correspondences IN ROW ORDER (the reference's summation
// order, as in the one-lane kernel), the LDL^T factorisation runs column by column with the five divisions of a column
// on five lanes, the 3x3 matrices of the exponential map one entry per lane.  Every scalar is computed by the same
// sequence of operations as in k3b_refine, so the two kernels return bit-identical poses, covariances and iteration
// counts (tested).  Groups of a wave converge at different iterations; a finished group idles through the others'
// synchronisation points.
// ---------------------------------------------------------------------------------------------
#define K3G_LANES 16
#define K3G_FRAMES 4
struct GnGroupLds {
  double J[MPE_MAX_MARKERS][14];  // per correspondence row: J0[6], J1[6], e0, e1
  double Ab[27];                  // upper triangle of A (21, row-major) then b (6)
  double L[6][6];
  double D[6];
  double O[9], O2[9], Rm[9], Vm[9];
  double T[12];
};
// index of A[r][c], r <= c, in the packed upper triangle
__device__ __forceinline__ int k3g_tri(int r, int c) { return r * 6 - (r * (r - 1)) / 2 + (c - r); }

template <int MODE>
__device__ __forceinline__ void k3b_group_body(const mpe_detections* __restrict__ dets, int n_frames, const SolveParams& sp,
                                               mpe_result* __restrict__ results, const TailMid* __restrict__ mid,
                                               int row_cap, const int blk) {
  __shared__ GnGroupLds s_g[K3G_FRAMES];
  const int tid = threadIdx.x;
  const int grp = tid >> 4, l = tid & 15;
  GnGroupLds& G = s_g[grp];
  const int f = blk * K3G_FRAMES + grp;
  const bool in_range = f < n_frames;
  const TailMid* m = mid + (in_range ? f : 0);
  const bool live = in_range && m->active;
  const mpe_detections* d = dets + (in_range ? f : 0);
  mpe_result* res = results + (in_range ? f : 0);
  const int n_m = sp.n_markers;
  const int n_c = live ? min(m->n_c, row_cap) : 0;
  const double fx = sp.fx, fy = sp.fy, cx = sp.cx, cy = sp.cy;
  // this lane's correspondence row
  double mk[3] = {0, 0, 0}, du = 0, dv = 0;
  if (l < n_c) {
    const int mi = m->cm[l] - 1, di = m->cd[l] - 1;
    mk[0] = sp.markers[3 * mi];
    mk[1] = sp.markers[3 * mi + 1];
    mk[2] = sp.markers[3 * mi + 2];
    du = d->undist_xy[2 * di];
    dv = d->undist_xy[2 * di + 1];
  }
  // ---- start pose: computeTransformation (pose_estimator.cpp:908-930) by lane 0, or the given pose (MODE 2)
  if (live && l == 0) {
    T34 T0;
    if (MODE == 2) {
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) T0.m[r][c] = res->T[r * 4 + c];
    } else {
      const double nv = (double)m->num_valid;
      double mo[3] = {0, 0, 0}, mr[3] = {0, 0, 0};
      for (int i = 0; i < n_m; ++i)
        for (int k = 0; k < 3; ++k) {
          mo[k] += sp.markers[3 * i + k];
          mr[k] += m->mean[3 * i + k] / nv;
        }
      for (int k = 0; k < 3; ++k) {
        mo[k] /= (double)n_m;
        mr[k] /= (double)n_m;
      }
      double Hm[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      for (int i = 0; i < n_m; ++i) {
        const double a[3] = {sp.markers[3 * i] - mo[0], sp.markers[3 * i + 1] - mo[1], sp.markers[3 * i + 2] - mo[2]};
        const double b[3] = {m->mean[3 * i] / nv - mr[0], m->mean[3 * i + 1] / nv - mr[1], m->mean[3 * i + 2] / nv - mr[2]};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) Hm[r][c] += a[r] * b[c];
      }
      double X[3][3];
      kabsch_rotation(Hm, X);  // R = V U^T
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) T0.m[r][c] = X[r][c];
        T0.m[r][3] = mr[r] - (X[r][0] * mo[0] + X[r][1] * mo[1] + X[r][2] * mo[2]);
      }
    }
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) G.T[r * 4 + c] = T0.m[r][c];
  }
  wave_sync();
  T34 T;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) T.m[r][c] = live ? G.T[r * 4 + c] : ((r == c) ? 1.0 : 0.0);

  // ---- optimisePose (pose_estimator.cpp:733-792)
  bool done = !live || MODE == 1;
  int iters = 0;
  for (int it = 0; it < 500; ++it) {
    if (__builtin_amdgcn_ballot_w64(!done) == 0) break;  // every group of the wave has converged
    // (a) Jacobian rows, computeJacobian (pose_estimator.cpp:945-957): lane j <- correspondence j
    if (!done && l < n_c) {
      double u, v, x, y, z;
      project_T(T, mk, fx, fy, cx, cy, u, v, x, y, z);
      const double e0 = du - u, e1 = dv - v;
      const double z_2 = z * z;
      double* Jr = G.J[l];
      Jr[0] = 1 / z * fx;
      Jr[1] = 0;
      Jr[2] = -x / z_2 * fx;
      Jr[3] = -x * y / z_2 * fx;
      Jr[4] = (1 + (x * x / z_2)) * fx;
      Jr[5] = -y / z * fx;
      Jr[6] = 0;
      Jr[7] = 1 / z * fy;
      Jr[8] = -y / z_2 * fy;
      Jr[9] = -(1 + y * y / z_2) * fy;
      Jr[10] = x * y / z_2 * fy;
      Jr[11] = x / z * fy;
      Jr[12] = e0;
      Jr[13] = e1;
    }
    wave_sync();
    // (b) A = sum J^T J (upper triangle), b = sum J^T e, each entry summed over the rows in row order
    if (!done) {
      for (int q = l; q < 27; q += K3G_LANES) {
        double acc = 0;
        if (q < 21) {
          int r = 0, base = 0;
          while (q >= base + (6 - r)) {
            base += 6 - r;
            ++r;
          }
          const int c = r + (q - base);
          for (int j = 0; j < n_c; ++j) acc += G.J[j][r] * G.J[j][c] + G.J[j][6 + r] * G.J[j][6 + c];
        } else {
          const int r = q - 21;
          for (int j = 0; j < n_c; ++j) acc += G.J[j][r] * G.J[j][12] + G.J[j][6 + r] * G.J[j][13];
        }
        G.Ab[q] = acc;
      }
    }
    wave_sync();
    // (c) unpivoted LDL^T (ldl6_factor), column by column: lane j the pivot, lanes j+1..5 the column's divisions
    double Lrow[6] = {0, 0, 0, 0, 0, 0};  // row l of L (lanes 0..5)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if (!done && l == j) {
        double dd = G.Ab[k3g_tri(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) dd -= Lrow[k] * Lrow[k] * G.D[k];
        G.D[j] = dd;
      }
      wave_sync();
      if (!done && l > j && l < 6) {
        double sacc = G.Ab[k3g_tri(j, l)];  // A[l][j] = A[j][l]
#pragma unroll
        for (int k = 0; k < j; ++k) sacc -= Lrow[k] * G.L[j][k] * G.D[k];
        Lrow[j] = sacc / G.D[j];
        G.L[l][j] = Lrow[j];
      }
      wave_sync();
    }
    // (d) ldl6_solve, every lane for itself (a chain of 36 dependent operations: nothing to spread)
    double dT[6] = {0, 0, 0, 0, 0, 0};
    if (!done) {
      double yv[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        double sacc = G.Ab[21 + i];
#pragma unroll
        for (int k = 0; k < i; ++k) sacc -= G.L[i][k] * yv[k];
        yv[i] = sacc;
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) yv[i] /= G.D[i];
#pragma unroll
      for (int i = 5; i >= 0; --i) {
        double sacc = yv[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) sacc -= G.L[k][i] * dT[k];
        dT[i] = sacc;
      }
    }
    // (e) exponentialMap(dT) * T (apply_exp), the 3x3 matrices one entry per lane
    const double ux = dT[0], uy = dT[1], uz = dT[2], wx = dT[3], wy = dT[4], wz = dT[5];
    const double theta = sqrt(wx * wx + wy * wy + wz * wz);
    const double th2 = theta * theta;
    if (!done && l < 9) {
      const double Ov[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
      double o = Ov[0];
#pragma unroll
      for (int q = 1; q < 9; ++q) o = (l == q) ? Ov[q] : o;
      G.O[l] = o;
    }
    wave_sync();
    if (!done && l < 9) {
      const int i = l / 3, j = l - 3 * i;
      const double o2 = G.O[3 * i] * G.O[j] + G.O[3 * i + 1] * G.O[3 + j] + G.O[3 * i + 2] * G.O[6 + j];
      const double o = G.O[l];
      const double I = (i == j) ? 1.0 : 0.0;
      double rm = I, vm = I;
      if (theta != 0) {
        double st, ct;
        sincos(theta, &st, &ct);
        rm = I + o / theta * st + o2 / th2 * (1 - ct);
        vm = I + (1 - ct) / th2 * o + (theta - st) / (th2 * theta) * o2;
      }
      G.Rm[l] = rm;
      G.Vm[l] = vm;
    }
    wave_sync();
    double nvv = 0;
    if (!done && l < 12) {
      const int i = l >> 2, j = l & 3;
      double sacc = G.Rm[3 * i] * G.T[j] + G.Rm[3 * i + 1] * G.T[4 + j] + G.Rm[3 * i + 2] * G.T[8 + j];
      if (j == 3) sacc += G.Vm[3 * i] * ux + G.Vm[3 * i + 1] * uy + G.Vm[3 * i + 2] * uz;
      nvv = sacc;
    }
    wave_sync();  // (every read of the old pose is done)
    if (!done && l < 12) G.T[l] = nvv;
    wave_sync();
    if (!done) {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) T.m[r][c] = G.T[r * 4 + c];
      iters = it + 1;
      double mx = -1;  // norm_max, pose_estimator.cpp:1073-1085
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const double av = fabs(dT[r]);
        if (av > mx) mx = av;
      }
      if (mx <= 1e-13) done = true;
    }
  }
  if (!live) return;
  // pose_covariance_ = A.inverse() with the A of the last iteration (pose_estimator.cpp:790): its factors are still in
  // G.L / G.D; column c of the inverse on lane c
  if (MODE != 1 && l < 6) {
    double yv[6], xv[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      double sacc = (i == l) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < i; ++k) sacc -= G.L[i][k] * yv[k];
      yv[i] = sacc;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) yv[i] /= G.D[i];
#pragma unroll
    for (int i = 5; i >= 0; --i) {
      double sacc = yv[i];
#pragma unroll
      for (int k = i + 1; k < 6; ++k) sacc -= G.L[k][i] * xv[k];
      xv[i] = sacc;
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) res->cov[r * 6 + l] = xv[r];
  }
  if (l < 12) res->T[l] = G.T[l];  // (the group's latest pose)
  if (l == 0) {
    res->T[12] = 0.0;
    res->T[13] = 0.0;
    res->T[14] = 0.0;
    res->T[15] = 1.0;
    res->gn_iterations = iters;
    res->status = MPE_FRAME_POSE;
  }
}
template <int MODE>
__global__ __launch_bounds__(64) void k3b_refine_group(const mpe_detections* __restrict__ dets, int n_frames, SolveParams sp,
                                                       mpe_result* __restrict__ results, const TailMid* __restrict__ mid,
                                                       int row_cap) {
  k3b_group_body<MODE>(dets, n_frames, sp, results, mid, row_cap, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// k_track_frame — TRACKED frames as ONE launch, one block of one wave per frame (round 6; pose_estimator.cpp:98-147 on
// an initialised estimator): image scan of the frame's ROI slot, blob extraction (small tier), nearest-neighbour
// correspondences + validation, Kabsch + Gauss-Newton — the bodies of k1a_scan / k1b_blobs<K1bSmall> / k3a_validate<0>
// / k3b_refine_group<0>, one after the other, with block barriers where a launch boundary used to be, and the finished
// records stored to the caller's pinned host memory by the kernel itself.  One frame (mpe_track_step) or the N streams
// of a lock-step time step (mpe_track_step_batch): a block never waits for another stream's slowest stage.  The
// arithmetic and the records are the chain's, bit for bit (tests: test_track_step_matches_oracle_pieces, the tracker
// suites).  A frame the small blob tier cannot hold comes back with det.status = MPE_FRAME_TOO_MANY_ROWS and the
// caller repeats the submission through the chain of kernels.
// ---------------------------------------------------------------------------------------------
struct TrackFrames {  // (kernel argument; frame b of the launch = block b)
  const uint8_t* pix;        // ROI slots: frame b at pix + b * slot_bytes, g.rows x g.pitch
  size_t slot_bytes;
  const double* pred;        // predicted pixels, 2 * MPE_MAX_MARKERS doubles per frame
  const FrameWin* wins;      // per-frame windows inside the slots, or nullptr (every frame fills its slot)
  u64* flags;                // flag words, flag_words per frame (a region of its own per block)
  size_t flag_words;
  uint32_t* hist;            // MPE_HIST_STRIDE words per frame (unused by the nearest-neighbour mode, addressed all the same)
  TailMid* mid;
  mpe_detections* dets;      // device records: three arrays of n
  uint32_t* corr;
  mpe_result* res;
  mpe_detections* h_dets;    // the same three arrays in pinned host memory, or nullptr (the caller copies)
  uint32_t* h_corr;
  mpe_result* h_res;
  unsigned long long* clk;   // CLOCKS: 5 stamps of block 0
};
__device__ __forceinline__ void track_copy_words(void* dst, const void* src, unsigned bytes, int lane) {
  const unsigned long long* s = static_cast<const unsigned long long*>(src);
  unsigned long long* d = static_cast<unsigned long long*>(dst);
  for (unsigned i = lane; i < bytes / 8; i += 64) d[i] = s[i];
}
// CLOCKS (option "track_phase_clocks"): the shader clock at the five phase boundaries -> clk[0 .. 4] (s_memtime)
#define K_TRACK_THREADS 256  // wave 0 runs the frame; waves 1 .. 3 take their share of the blur's items (k1b_wave<C, true>)
template <bool CLOCKS>
__global__ __launch_bounds__(K_TRACK_THREADS) void k_track_frame(TrackFrames a, FrameGeom g, DetectParams dp, SolveParams sp,
                                                                 ThrTest thr, double nn_tol, int row_cap) {
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  mpe_detections* det = a.dets + b;
  if (threadIdx.x >= 64) {  // a helper wave: the barrier behind the image pass, then the blob extraction's blur
    __syncthreads();
    k1b_wave<K1bSmall, true>(0, true, a.pix + (size_t)b * a.slot_bytes, a.flags + (size_t)b * a.flag_words, g, dp, det, nullptr,
                             a.wins ? a.wins + b : nullptr);
    return;
  }
  uint32_t* corr = a.corr + (size_t)b * 2 * MPE_MAX_MARKERS;
  mpe_result* res = a.res + b;
  u64* flags = a.flags + (size_t)b * a.flag_words;
  // the record goes to the caller's pinned host memory from here instead of through a copy command behind the kernel
  static_assert(sizeof(mpe_detections) % 8 == 0 && sizeof(mpe_result) % 8 == 0, "records copied in 8-byte words");
  auto deliver = [&]() {
    if (!a.h_dets) return;
    __syncthreads();  // (every lane's stores to the device records are visible to the block)
    track_copy_words(a.h_dets + b, det, sizeof(mpe_detections), lane);
    track_copy_words(a.h_corr + (size_t)b * 2 * MPE_MAX_MARKERS, corr, 2 * MPE_MAX_MARKERS * sizeof(uint32_t), lane);
    track_copy_words(a.h_res + b, res, sizeof(mpe_result), lane);
  };
  auto stamp = [&](int i) {
    if constexpr (CLOCKS) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if (lane == 0 && b == 0) a.clk[i] = t;
    }
  };
  stamp(0);
  const uint8_t* roi = a.pix + (size_t)b * a.slot_bytes;
  // ---- the image pass over the slot: one flag bit per 16-byte segment (k1a_scan's test)
  {
    const size_t n_seg = ((size_t)g.rows * g.pitch) / 16;
    const uint4* px = reinterpret_cast<const uint4*>(roi);
    // (uniform trip counts: the ballots need every lane; eight independent loads in flight per lane — one after the
    //  other the 15 trips of a 120 x 120 ROI each waited for their own miss: 4.2 us of a 100 us frame)
    for (size_t s0 = 0; s0 < n_seg; s0 += 64 * 8) {
      uint4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const size_t s = s0 + 64 * k + lane;
        v[k] = make_uint4(0, 0, 0, 0);
        if (s < n_seg) v[k] = px[s];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const u64 bb = __ballot(any_gt16(v[k], thr) != 0);
        if (lane == 0 && s0 + 64 * k < n_seg) flags[(s0 >> 6) + k] = bb;
      }
    }
  }
  __syncthreads();  // (the flag words are read back by other lanes: workgroup-scope release / acquire)
  stamp(1);
  // ---- blob extraction, small tier: the block's slot is "frame 0" of its own flag region
  k1b_wave<K1bSmall, true>(0, true, roi, flags, g, dp, det, nullptr, a.wins ? a.wins + b : nullptr);
  __syncthreads();
  stamp(2);
  if (det->status == MPE_FRAME_TOO_MANY_ROWS) {  // (uniform: written before the barrier)
    deliver();
    return;
  }
  // ---- correspondences by nearest neighbour to the predicted pixels, validation; Kabsch + Gauss-Newton
  k3a_body<0>(det, a.hist + (size_t)b * MPE_HIST_STRIDE, 1, sp, res, corr, nullptr, a.pred + (size_t)b * 2 * MPE_MAX_MARKERS,
              nn_tol, a.mid + b, 0);
  __syncthreads();
  stamp(3);
  k3b_group_body<0>(det, 1, sp, res, a.mid + b, row_cap, 0);
  stamp(4);
  deliver();
}
size_t track_flag_words(const FrameGeom& g) { return ((size_t)g.rows * g.pitch / 16 + 63) / 64 + 2; }
hipError_t launch_track_frames(const TrackFramesArgs& t, int n_frames, const FrameGeom& g, const DetectParams& dp,
                               const SolveParams& sp, double nn_tol, hipStream_t s) {
  if (n_frames <= 0) return hipSuccess;
  const int rows = sp.n_markers > 3 ? sp.n_markers : 4;
  const size_t lds_a = ((size_t)K3_FRAMES_PER_BLOCK * K3_GROUP * 3 * sp.n_markers + (size_t)2 * (rows - 3) * 64) * sizeof(double);
  TrackFrames a;
  a.pix = t.pix;
  a.slot_bytes = t.slot_bytes;
  a.pred = t.pred;
  a.wins = static_cast<const FrameWin*>(t.wins);
  a.flags = reinterpret_cast<u64*>(t.flags);
  a.flag_words = track_flag_words(g);
  a.hist = t.hist;
  a.mid = static_cast<TailMid*>(t.mid);
  a.dets = t.dets;
  a.corr = t.corr;
  a.res = t.res;
  a.h_dets = t.h_dets;
  a.h_corr = t.h_corr;
  a.h_res = t.h_res;
  a.clk = t.phase_clocks;
  if (t.phase_clocks)
    hipLaunchKernelGGL(k_track_frame<true>, dim3((unsigned)n_frames), dim3(K_TRACK_THREADS), lds_a, s, a, g, dp, sp, make_thr_test(dp.thr),
                       nn_tol, rows);
  else
    hipLaunchKernelGGL(k_track_frame<false>, dim3((unsigned)n_frames), dim3(K_TRACK_THREADS), lds_a, s, a, g, dp, sp, make_thr_test(dp.thr),
                       nn_tol, rows);
  return hipGetLastError();
}

hipError_t launch_k3_tail(const mpe_detections* dets, const uint32_t* hist, int n_frames, const SolveParams& sp,
                          mpe_result* results, uint32_t* corr_out, const uint32_t* corr_in, const double* nn_pred,
                          double nn_tol, void* mid_buf, hipStream_t s, int mode) {
  if (n_frames <= 0) return hipSuccess;
  const int refine_variant = sp.refine_variant;
  TailMid* mid = static_cast<TailMid*>(mid_buf);
  // explicit correspondences (corr_in) may hold up to MPE_MAX_MARKERS rows, also more than n_markers (repeated
  // markers are defined input for checkCorrespondences): size the row buffers for the row capacity
  const int rows = corr_in ? MPE_MAX_MARKERS : (sp.n_markers > 3 ? sp.n_markers : 4);
  const int nu = rows - 3;
  const size_t lds_a = ((size_t)K3_FRAMES_PER_BLOCK * K3_GROUP * 3 * sp.n_markers + (size_t)2 * nu * 64) * sizeof(double);
  const size_t lds_b = (size_t)rows * 5 * K3B_THREADS * sizeof(double);
  const dim3 grid_a((n_frames + K3_FRAMES_PER_BLOCK - 1) / K3_FRAMES_PER_BLOCK);
  const dim3 grid_b((n_frames + K3B_THREADS - 1) / K3B_THREADS);
  const dim3 grid_g((n_frames + K3G_FRAMES - 1) / K3G_FRAMES);
  // refinement: 16 lanes per frame while the launch is too small to fill the chip with one lane per frame (a tracked
  // frame, a few hundred streams in lock step), else one lane per frame; bit-identical results (refine_variant forces
  // one of them: 1 = lane, 2 = group)
  const bool group = refine_variant == 2 || (refine_variant == 0 && n_frames <= 2048);
#define K3_LAUNCH(M_)                                                                                              \
  do {                                                                                                             \
    hipLaunchKernelGGL(k3a_validate<M_>, grid_a, dim3(64), lds_a, s, dets, hist, n_frames, sp, results, corr_out,  \
                       corr_in, nn_pred, nn_tol, mid);                                                             \
    if (group)                                                                                                     \
      hipLaunchKernelGGL(k3b_refine_group<M_>, grid_g, dim3(64), 0, s, dets, n_frames, sp, results,                \
                         (const TailMid*)mid, rows);                                                               \
    else                                                                                                           \
      hipLaunchKernelGGL(k3b_refine<M_>, grid_b, dim3(K3B_THREADS), lds_b, s, dets, n_frames, sp, results,         \
                         (const TailMid*)mid, rows);                                                               \
  } while (0)
  if (mode == 1)
    K3_LAUNCH(1);
  else if (mode == 2)
    K3_LAUNCH(2);
  else
    K3_LAUNCH(0);
#undef K3_LAUNCH
  return hipGetLastError();
}

// =============================================================================================
// Primitive batches — P3P::computePoses / P3P::solveQuartic (p3p.h:110-127) for n independent
// problems, one lane each.  Not on the batch path (K2 / K3 inline the same device functions); they
// exist for callers of the static primitives and for stage-level parity tests of mpe_p3p.h.
// =============================================================================================
__global__ __launch_bounds__(64) void k_p3p_batch(const double* __restrict__ fv, const double* __restrict__ wp, int n,
                                                 double* __restrict__ sol, int* __restrict__ status) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* f = fv + (size_t)i * 9;
  const double* w = wp + (size_t)i * 9;
  const V3 f0 = {f[0], f[1], f[2]}, f1 = {f[3], f[4], f[5]}, f2 = {f[6], f[7], f[8]};
  const V3 w0 = {w[0], w[1], w[2]}, w1 = {w[3], w[4], w[5]}, w2 = {w[6], w[7], w[8]};
  P3PCtx c;
  if (!p3p_prepare(f0, f1, f2, w0, w1, w2, c)) {
    status[i] = -1;  // collinear world points, p3p.cpp:77-80; solutions left untouched like the reference
    return;
  }
  double* o = sol + (size_t)i * 48;
  for (int k = 0; k < 4; ++k) {
    M3 R;
    V3 C;
    p3p_solution(c, pick_root(c, k), R, C);
    double* q = o + 12 * k;
    q[0] = R.r0.x; q[1] = R.r0.y; q[2] = R.r0.z; q[3] = C.x;
    q[4] = R.r1.x; q[5] = R.r1.y; q[6] = R.r1.z; q[7] = C.y;
    q[8] = R.r2.x; q[9] = R.r2.y; q[10] = R.r2.z; q[11] = C.z;
  }
  status[i] = 0;
}

__global__ __launch_bounds__(64) void k_quartic_batch(const double* __restrict__ factors, int n, int variant,
                                                     double* __restrict__ roots) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* a = factors + (size_t)i * 5;
  double r[4];
  if (variant == 1)
    solve_quartic_lit2(a[0], a[1], a[2], a[3], a[4], r);  // the voting kernel's variant
  else
    solve_quartic(a[0], a[1], a[2], a[3], a[4], r);  // IEEE operators (validation kernel)
  for (int k = 0; k < 4; ++k) roots[(size_t)i * 4 + k] = r[k];
}

// =============================================================================================
// frame decode: sensor_msgs/Image payloads -> mono8 (what cv_bridge::toCvCopy(msg, MONO8) does for the node,
// monocular_pose_estimator.cpp:147).  HBM bound, one pass: 4 output pixels per lane and step.
//   bgr8 / rgb8 / bgra8 / rgba8: cv::cvtColor(..., COLOR_*2GRAY) for CV_8U — integer, 14 fractional bits,
//       Y = (B * 1868 + G * 9617 + R * 4899 + 2^13) >> 14   (OpenCV 2.4, 3.0 .. 3.4.1; from 3.4.2 on: 15 bits, see mpe.h)
//   mono16 (host byte order after cv_bridge's endianness fix): Mat::convertTo(CV_8U, 255. / 65535.) —
//       saturate_cast<uchar>((float)v * (float)(255. / 65535.)), i.e. round-half-even of the single-precision product
// =============================================================================================
__device__ __forceinline__ unsigned gray_px(unsigned c0, unsigned c1, unsigned c2, bool rgb) {
  const unsigned b = rgb ? c2 : c0, r = rgb ? c0 : c2;
  return (b * 1868u + c1 * 9617u + r * 4899u + (1u << 13)) >> 14;
}
__global__ __launch_bounds__(256) void k_to_mono8(const uint8_t* __restrict__ src, size_t src_stride, size_t src_frame_stride,
                                                  int encoding, int big_endian, int rows, int cols, long long n_rows_total,
                                                  uint8_t* __restrict__ dst) {
  const int quads = (cols + 3) >> 2;  // 4 output pixels per work item
  const long long total = n_rows_total * quads;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / quads;
    const int x0 = (int)(i - row * quads) * 4;
    const long long f = row / rows;
    const int y = (int)(row - f * rows);
    const uint8_t* s = src + (size_t)f * src_frame_stride + (size_t)y * src_stride;
    uint8_t* d = dst + ((size_t)f * rows + y) * cols + x0;
    const int n = min(4, cols - x0);
    unsigned out[4] = {0, 0, 0, 0};
    if (encoding == MPE_ENC_MONO16) {
      for (int k = 0; k < n; ++k) {
        const uint8_t* p = s + 2 * (size_t)(x0 + k);
        const unsigned v = big_endian ? ((unsigned)p[0] << 8 | p[1]) : ((unsigned)p[1] << 8 | p[0]);
        float r = rintf((float)v * (float)(255.0 / 65535.0));
        r = fminf(fmaxf(r, 0.f), 255.f);
        out[k] = (unsigned)r;
      }
    } else if (encoding == MPE_ENC_MONO8) {
      for (int k = 0; k < n; ++k) out[k] = s[x0 + k];
    } else {
      const int bpp = (encoding == MPE_ENC_BGRA8 || encoding == MPE_ENC_RGBA8) ? 4 : 3;
      const bool rgb = encoding == MPE_ENC_RGB8 || encoding == MPE_ENC_RGBA8;
      const uint8_t* p = s + (size_t)bpp * x0;
      if (n == 4 && ((reinterpret_cast<uintptr_t>(p) & 3) == 0)) {  // three or four aligned 32-bit loads
        const unsigned* w = reinterpret_cast<const unsigned*>(p);
        if (bpp == 3) {
          const unsigned w0 = w[0], w1 = w[1], w2 = w[2];
          out[0] = gray_px(w0 & 0xFF, (w0 >> 8) & 0xFF, (w0 >> 16) & 0xFF, rgb);
          out[1] = gray_px(w0 >> 24, w1 & 0xFF, (w1 >> 8) & 0xFF, rgb);
          out[2] = gray_px((w1 >> 16) & 0xFF, w1 >> 24, w2 & 0xFF, rgb);
          out[3] = gray_px((w2 >> 8) & 0xFF, (w2 >> 16) & 0xFF, w2 >> 24, rgb);
        } else {
          for (int k = 0; k < 4; ++k) out[k] = gray_px(w[k] & 0xFF, (w[k] >> 8) & 0xFF, (w[k] >> 16) & 0xFF, rgb);
        }
      } else {
        for (int k = 0; k < n; ++k) out[k] = gray_px(p[bpp * k], p[bpp * k + 1], p[bpp * k + 2], rgb);
      }
    }
    if (n == 4 && ((reinterpret_cast<uintptr_t>(d) & 3) == 0)) {
      *reinterpret_cast<unsigned*>(d) = out[0] | (out[1] << 8) | (out[2] << 16) | (out[3] << 24);
    } else {
      for (int k = 0; k < n; ++k) d[k] = (uint8_t)out[k];
    }
  }
}

hipError_t launch_to_mono8(const uint8_t* src, size_t src_stride, size_t src_frame_stride, int encoding, int big_endian,
                           int n_frames, int rows, int cols, uint8_t* dst, hipStream_t s) {
  if (n_frames <= 0 || rows <= 0 || cols <= 0) return hipSuccess;
  const long long n_rows = (long long)n_frames * rows;
  const long long items = n_rows * ((cols + 3) / 4);
  long long blocks = (items + 255) / 256;
  const long long most = (long long)device_cu_count() * 64;  // grid-stride beyond 64 blocks per CU
  if (blocks > most) blocks = most;
  hipLaunchKernelGGL(k_to_mono8, dim3((unsigned)blocks), dim3(256), 0, s, src, src_stride, src_frame_stride, encoding,
                     big_endian, rows, cols, n_rows, dst);
  return hipGetLastError();
}

// one wave that keeps a CU slot busy for `ticks` of the constant-rate counter (100 MHz): used once per
// handle to find two side streams that really run concurrently (see pick_concurrent_streams)
__global__ void k_spin(unsigned long long ticks, unsigned long long* sink) {
  const unsigned long long t0 = wall_clock64();
  unsigned long long t = t0;
  while (t - t0 < ticks) {
    __builtin_amdgcn_s_sleep(32);
    t = wall_clock64();
  }
  if (sink && threadIdx.x == 0) *sink = t - t0;
}

hipError_t launch_spin(unsigned long long ticks, hipStream_t s) {
  hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, ticks, (unsigned long long*)nullptr);
  return hipGetLastError();
}

hipError_t launch_p3p_batch(const double* fv, const double* wp, int n, double* sol, int* status, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_p3p_batch, dim3((n + 63) / 64), dim3(64), 0, s, fv, wp, n, sol, status);
  return hipGetLastError();
}

hipError_t launch_quartic_batch(const double* factors, int n, int variant, double* roots, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_quartic_batch, dim3((n + 63) / 64), dim3(64), 0, s, factors, n, variant, roots);
  return hipGetLastError();
}

//@file-epilogue
}  // namespace mpe
//@file-epilogue-end
