/*
 * mpe_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * Dependency-free CPU restatement of the per-frame hot path of
 * uzh-rpg/rpg_monocular_pose_estimator (LEDDetector::findLeds -> PoseEstimator::initialise ->
 * checkCorrespondences -> optimisePose).  It exists ONLY to check the HIP path:
 *   - tests/            (parity checker)
 *   - __graft_entry__.smoke()
 *   - bench.py's cpu_baseline leg
 * Nothing under rpg_monocular_pose_estimator_amd/ (the product) may include, link or call it.
 *
 * PARITY UNPINNED: the reference ships no tests / golden vectors and cannot be built here
 * (needs Eigen, OpenCV, ROS — none installed, no network), so this restatement is pinned only
 * by the build's own known-answer tests (tests/test_oracle_*.py) and an independent
 * numpy/mpmath/scipy witness (tests/witness.py).  See DESIGN.md §Oracle.
 *
 * Conventions (plain C ABI so ctypes can bind it):
 *   K         : 9 doubles, row-major 3x3 camera matrix
 *   D         : nD doubles (k1,k2,p1,p2,k3), nD may be 0, 4 or >=5 (reference LED.cpp:190-194)
 *   markers   : n_markers x 3 doubles (x,y,z), metres, marker frame
 *   det       : n_det x 2 doubles, undistorted pixel coordinates
 *   T         : 16 doubles, row-major 4x4 (T_camera_object)
 *   cov       : 36 doubles, row-major 6x6, twist order (upsilon, omega)
 *   hist      : n_det x n_markers uint32, row-major (row = detection, col = marker)
 *   corr      : rows of (marker, detection), 1-based (reference PE.cpp:361-362)
 */
#ifndef MPE_ORACLE_H_
#define MPE_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_params {
  int threshold_value;                 /* PE.h:85  detection_threshold_value_ */
  double gaussian_sigma;               /* PE.h:86 */
  double min_blob_area;                /* PE.h:87 */
  double max_blob_area;                /* PE.h:88 */
  double max_width_height_distortion;  /* PE.h:89 */
  double max_circular_distortion;      /* PE.h:90 */
  double back_projection_pixel_tolerance;   /* PE.h:68 */
  double nearest_neighbour_pixel_tolerance; /* PE.h:69 */
  double certainty_threshold;               /* PE.h:70 */
  double valid_correspondence_threshold;    /* PE.h:71 */
  unsigned roi_border_thickness;            /* PE.h:91 */
  unsigned histogram_threshold;             /* 0 -> numCombinations(n_markers,3), PE.cpp:54 */
} orc_params;

typedef struct orc_result {
  double T[16];
  double cov[36];
  int status;  /* 0 = pose found (estimateBodyPose true), 1 = no pose */
  int n_det;
  int n_corr;
  int gn_iterations;
} orc_result;

/* ---- Combinations (COMB.cpp) ---- */
unsigned orc_factorial(int N);                         /* COMB.cpp:34-40, 32-bit wrap */
unsigned orc_num_combinations(unsigned N, unsigned K); /* COMB.cpp:42-45 */
/* rows x 3, 1-based; returns number of rows.  out may be NULL to query the count. */
int orc_combinations3(unsigned N, unsigned* out);      /* COMB.cpp:52-129 (K=3) */
int orc_permutations3(unsigned N, unsigned* out);      /* COMB.cpp:131-203 (K=3) */

/* ---- P3P (P3P.cpp) ---- */
int orc_solve_quartic(const double factors[5], double real_roots[4]); /* P3P.cpp:238-286 */
/* fv[i*3+k]: k-th component of i-th unit bearing; wp[i*3+k] likewise for world points.
 * sol[s*12 + r*4 + c]: s-th solution, 3x4 row-major [R|C].  returns 0 or -1 (collinear). */
int orc_p3p(const double fv[9], const double wp[9], double sol[48]);  /* P3P.cpp:65-236 */

/* ---- LED detection (LED.cpp:35-112) ---- */
/* Stage outputs for stage-boundary parity tests; any pointer may be NULL.
 * blurred/mask are roi_h x roi_w bytes. */
int orc_blur_mask(const uint8_t* img, int rows, int cols, size_t stride, int roi_x, int roi_y,
                  int roi_w, int roi_h, int threshold_value, double sigma, uint8_t* blurred,
                  uint8_t* mask);
/* Quantised kernel (8 fractional bits); returns ksize, or <0 on error. taps gets ksize ints. */
int orc_gaussian_kernel_q8(double sigma, int* taps, int cap);
/* External contours of a binary mask (nonzero = fg) in OpenCV findContours(RETR_EXTERNAL,
 * CHAIN_APPROX_NONE) order.  pts gets (x,y) int pairs back to back, counts[i] the length of
 * contour i.  Returns number of contours or <0 when capacity is exceeded. */
int orc_external_contours(const uint8_t* mask, int h, int w, int* pts, int pts_cap, int* counts,
                          int counts_cap);
int orc_find_leds(const uint8_t* img, int rows, int cols, size_t stride, int roi_x, int roi_y,
                  int roi_w, int roi_h, const orc_params* p, const double K[9], const double* D,
                  int nD, double* undist_xy, float* dist_xy, int cap, int* n_out);
/* LED.cpp:181-224 (used by the synthetic renderer to place spots, and by determineROI) */
void orc_distort_points(const float* src_xy, float* dst_xy, int n, const double K[9],
                        const double* D, int nD);
int orc_undistort_points(const float* src_xy, float* dst_xy, int n, const double K[9],
                         const double* D, int nD);

/* LED.cpp:114-179; roi = x, y, width, height */
void orc_determine_roi(const double* px, int n, int rows, int cols, int border_size, const double K[9],
                       const double* D, int nD, int roi[4]);

/* ---- pose (PE.cpp) ---- */
void orc_image_vectors(const double* det, int n_det, const double K[9], double* vec3); /* PE.cpp:288-301 */
void orc_project2d(const double p4[4], const double T[16], const double K[9], double out[2]); /* PE.cpp:251-268 */
void orc_exponential_map(const double twist[6], double T[16]);   /* PE.cpp:962-994 */
void orc_jacobian(const double T[16], const double p4[4], double fx, double fy, double J[12]); /* PE.cpp:932-960 */
void orc_compute_transformation(const double* obj, const double* rep, int n, double T[16]); /* PE.cpp:908-930; 3 x n column sets given as n x 3 */
/* voting only: PE.cpp:544-702.  returns 0 */
int orc_vote_histogram(const double* det, int n_det, const double* markers, int n_markers,
                       const double K[9], double back_projection_pixel_tolerance, uint32_t* hist);
/* cv_bridge::toCvCopy(image_msg, MONO8) (monocular_pose_estimator.cpp:147) for encoding 0 mono8, 1 bgr8, 2 rgb8,
 * 3 bgra8, 4 rgba8, 5 mono16 (big_endian: Image.is_bigendian); dst packed rows x cols.  returns 0 */
int orc_convert_to_mono8(const uint8_t* src, int encoding, int big_endian, int rows, int cols, size_t src_stride,
                         uint8_t* dst);
/* forensics: the votes of the hypotheses [item_lo, item_hi) of that loop nest only (flattened index = detection-triple
 * index * P(n_markers,3) + marker-permutation index, COMB.cpp table order) */
int orc_vote_items(const double* det, int n_det, const double* markers, int n_markers, const double K[9],
                   double back_projection_pixel_tolerance, long long item_lo, long long item_hi, uint32_t* hist);
/* PE.cpp:344-370 — consumes (zeroes columns of) hist; returns number of rows written to corr */
int orc_correspondences_from_histogram(uint32_t* hist, int n_det, int n_markers,
                                       unsigned histogram_threshold, uint32_t* corr);
/* PE.cpp:394-542: returns 1 and writes T when the correspondences validate, else 0 */
int orc_check_correspondences(const double* det, int n_det, const double* markers, int n_markers,
                              const double K[9], const orc_params* p, const uint32_t* corr,
                              int n_corr, double T[16]);
/* PE.cpp:733-792: refines T in place, writes cov; returns iterations used */
int orc_optimise_pose(const double* det, const double* markers, const double K[9],
                      const uint32_t* corr, int n_corr, double T[16], double cov[36]);
/* setImagePoints + initialise + optimiseAndUpdatePose on a fresh estimator (PE.cpp:80-91) */
int orc_solve_bruteforce(const double* det, int n_det, const double* markers, int n_markers,
                         const double K[9], const orc_params* p, orc_result* out, uint32_t* hist,
                         uint32_t* corr);
/* estimateBodyPose on a FRESH estimator, uninitialised branch (PE.cpp:62-96) */
int orc_estimate_frame(const uint8_t* img, int rows, int cols, size_t stride,
                       const double* markers, int n_markers, const double K[9], const double* D,
                       int nD, const orc_params* p, orc_result* out);
/* Same for a batch of contiguous frames using n_threads std::threads (one estimator each);
 * frame f starts at frames + f*frame_stride. */
int orc_estimate_batch(const uint8_t* frames, int n_frames, int rows, int cols, size_t stride,
                       size_t frame_stride, const double* markers, int n_markers,
                       const double K[9], const double* D, int nD, const orc_params* p,
                       orc_result* out, int n_threads);

/* ---- stateful estimator: the whole estimateBodyPose state machine incl. the tracking path
 * (PE.cpp:62-147, 232-244, 372-392, 794-848, 996-1064; LED.cpp:114-179) ---- */
typedef struct orc_tracker orc_tracker;
orc_tracker* orc_tracker_create(const double* markers, int n_markers, const double K[9], const double* D,
                                int nD, const orc_params* p);
void orc_tracker_destroy(orc_tracker* tr);
/* returns 1 pose updated / 0 not / <0 error; info[8] = roi x,y,w,h, it_since_initialized, n_det,
 * n_corr, used_bruteforce */
int orc_tracker_estimate(orc_tracker* tr, const uint8_t* img, int rows, int cols, size_t stride,
                         double time, orc_result* out, int* info);
void orc_logarithm_map(const double T[16], double xi[6]); /* PE.cpp:996-1064 */
/* frame-parallel orc_vote_histogram: det n x max_det x 2, hist n x max_det x n_markers */
int orc_vote_batch(const double* det, const int* n_det, int n, int max_det, const double* markers,
                   int n_markers, const double K[9], double tol, uint32_t* hist, int n_threads);

#ifdef __cplusplus
}
#endif
#endif /* MPE_ORACLE_H_ */
