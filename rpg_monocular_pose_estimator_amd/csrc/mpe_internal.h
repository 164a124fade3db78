// mpe_internal.h — structures shared by the host side (mpe_host.h, mpe_schedule / mpe_options / mpe_track_abi / mpe_abi .cpp) and the gfx950 kernels
// (mpe_k1.hip, mpe_k2.hip, mpe_k3.hip).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mpe.h"

namespace mpe {

// ---- image geometry in HBM -------------------------------------------------------------------
// A frame is rows x pitch bytes, pitch % 16 == 0, pixels at x >= cols are zero (never bright).
// The image pass sees the whole batch as one flat array of 16-byte "segments"; flag bit G tells
// whether segment G holds a pixel above the threshold.
struct FrameGeom {
  int rows, cols, pitch;
  int segs_per_row;    // pitch / 16
  int segs_per_frame;  // rows * segs_per_row
  int wb;              // u64 words per bitmap row (bit index = x + 1, plus one pad word)
  int rw;              // u64 words of the row-activity bitset
  int tw;              // u64 words per row of the todo (segment) bitset
  int slot_cap;        // bitmap rows that fit in LDS
};

struct DetectParams {
  int thr;    // THRESH_TOZERO threshold (strict >), clamped to [-1, 255]
  int ksize;  // Gaussian taps (odd, <= MPE_MAX_KSIZE); 1 = identity
  int taps[MPE_MAX_KSIZE];
  double min_area, max_area, max_wh, max_circ;
  double K[9];
  double ifx, ify;
  double k[8];  // distortion coefficients k1 k2 p1 p2 k3 k4 k5 k6
  int undist_iters;
  int roi_x, roi_y;  // added to the centroid (float add), LED.cpp:74
  // the taps as bytes for the packed dot product of the blur: [0] = taps 0..3, [1] = tap 4 in byte 0; taps_u8 = 0 when
  // a tap does not fit a byte (a centre tap of 256 at sigma < 0.3: the byte-wise blur code then does the work)
  unsigned taps_packed[2];
  int taps_u8;
};
inline void pack_taps(DetectParams& dp) {
  dp.taps_packed[0] = dp.taps_packed[1] = 0;
  dp.taps_u8 = 1;
  for (int i = 0; i < dp.ksize && i < MPE_MAX_KSIZE; ++i) {
    if (dp.taps[i] < 0 || dp.taps[i] > 255) dp.taps_u8 = 0;
    if (i < 8) dp.taps_packed[i >> 2] |= ((unsigned)dp.taps[i] & 0xFFu) << (8 * (i & 3));
  }
}

struct SolveParams {
  int n_markers;
  double markers[MPE_MAX_MARKERS * 3];
  double fx, fy, cx, cy;
  double back_tol;        // back_projection_pixel_tolerance_
  double certainty_thr;   // certainty_threshold_
  double valid_corr_thr;  // valid_correspondence_threshold_
  unsigned hist_thr;      // histogram_threshold_
  int vote_arith;         // option "vote_arith": 1 fast voting arithmetic + strict re-evaluation of the hypotheses it
                          // cannot decide (default), 0 strict (IEEE, literal order), 2 fast alone (round-3 behaviour);
                          // round 6: 3 = 1 and 4 = 0 with the quartic's three complex powers evaluated as libstdc++ /
                          // glibc do (mpe_ddmath.h) — the CPU reference's own digits in Ferrari's unstable corner
  int refine_variant;     // option "refine_variant": 0 automatic, 1 one lane per frame, 2 sixteen lanes per frame
};

__host__ __device__ inline bool vote_arith_is_strict(int a) { return a == 0 || a == 4; }   // the strict kernel votes
__host__ __device__ inline bool vote_arith_screens(int a) { return a == 1 || a == 3; }     // fast kernel + suspect list
__host__ __device__ inline bool vote_arith_glibc_pow(int a) { return a == 3 || a == 4; }   // strict item: glibc's powers

// A frame's vote histogram: MPE_HIST_WORDS = MPE_MAX_DETECTIONS x MPE_MAX_MARKERS words (the shape the C ABI hands out),
// on the device MPE_HIST_STRIDE words apart — 33 x 128 B, not the 4 KB of the bare table: a power-of-two stride puts
// the few rows every frame of a launch writes (5 x 64 B at C2) onto the same HBM channels, and the voting launch that
// carries the scan paid for it (round 6: 1.54 -> 1.66 ms per launch when the table went from 2 KB to 4 KB).
#define MPE_HIST_WORDS (MPE_MAX_DETECTIONS * MPE_MAX_MARKERS)
#define MPE_HIST_STRIDE (MPE_HIST_WORDS + 32)

// Hypotheses a fast voting launch does not decide itself (mpe_k2.hip, k2_sus_push): a list in device memory that
// launch_k2_fixup works off with the strict arithmetic, behind the voting launch and in front of the tail.
#define MPE_FIX_CTL_WORDS 8
struct VoteFixup {
  unsigned* ctl;             // MPE_FIX_CTL_WORDS words: [0] entries appended (reset by the fix-up kernel), [1] appends
                             // that found the list full (cumulative), [2] blocks done (internal), [3] entries
                             // re-evaluated (cumulative), [4] value of [1] the last k2_vote_relost launch handled,
                             // [5] its blocks done (internal), [6] frames voted again by it (cumulative),
                             // [7] frames too wide for the fast kernels, left to that launch (cumulative)
  unsigned long long* list;  // cap entries of 2 words
  unsigned cap;
  unsigned screen;           // 1: the voting kernel screens its hypotheses (vote_arith 1); 0: it only sends what its
                             // per-wave vote queue cannot hold (vote_arith 2: the fast arithmetic decides the rest)
};

// launchers (mpe_k1.hip / mpe_k2.hip / mpe_k3.hip)
int device_cu_count();  // compute units of the current device (cached)
size_t k1b_scratch_bytes(const FrameGeom& g, int n_frames);  // general-tier slabs for launches of up to n_frames
void k1b_set_general_blocks(int cap);  // blocks (= scratch slabs) of the general blob tier at most; process-wide
int k1b_get_general_blocks();
hipError_t launch_k1a_scan(const uint8_t* frames, size_t n_bytes, unsigned long long* flags, int thr,
                           int dummy_lds_bytes, hipStream_t s, int blocks_per_cu = 0);
hipError_t launch_k1b_blobs(const uint8_t* frames, const unsigned long long* flags, int n_frames, const FrameGeom& g,
                            const DetectParams& dp, mpe_detections* dets, int* worklist, uint8_t* scratch,
                            size_t scratch_bytes, int blob_hint, hipStream_t s, const void* frame_windows = nullptr,
                            bool lists_zeroed = false, bool first_tier_only = false, bool general_lds = false);
// general_lds: frames are expected to reach the general tier (the caller saw them in the previous call's work-list): that
// tier then runs as k1b_general_lds — a block per CU, the frame's bitmaps in LDS — where the frame size allows it
// worklist: 2 * (n_frames + 1) ints (two device work-lists that chain the capacity tiers); lists_zeroed = the caller
// has zeroed it in stream order already (one memset for all the sub-batches of a call) and no memset is issued here.
// first_tier_only: launch the small-pool tier alone (blob_hint 1 .. 8); frames it cannot hold keep status
// MPE_FRAME_TOO_MANY_ROWS and the caller repeats them through the whole chain.
// frame_windows: optional device array of n_frames x {rows, cols, roi_x, roi_y} ints — every frame then is a window
// of that size in the top-left corner of its g.rows x g.pitch slot (zero beyond), as when LEDDetector::findLeds clones
// image(ROI) (led_detector.cpp:44): borders follow the window, centroids get its ROI origin added.
size_t k2_table_bytes(int n_markers);
hipError_t launch_k2_prep(const SolveParams& sp, double* tab, hipStream_t s);
// sp.vote_arith: 1 = the fast voting kernel (tables, Newton-Raphson division / square root, Newton cube root),
// 0 = the strict kernel (the validation kernel's P3P with IEEE operators; never carries a scan: *scanned_bytes = 0).
// scan_px != nullptr: the voting waves also scan scan_bytes of pixels (the next sub-batch) into scan_flags;
// *scanned_bytes = the prefix they cover (whole chunks), the caller scans the rest with launch_k1a_scan
hipError_t launch_k2_vote(mpe_detections* dets, int n_frames, const SolveParams& sp, const double* tab,
                          uint32_t* hist, int splits, int n_det_hint, hipStream_t s, const uint8_t* scan_px = nullptr,
                          size_t scan_bytes = 0, unsigned long long* scan_flags = nullptr, int scan_thr = 0,
                          size_t* scanned_bytes = nullptr, const int* item_range = nullptr,
                          const VoteFixup* fixup = nullptr);  // (required unless sp.vote_arith == 0)
// the strict re-evaluation of what that launch appended to `fixup` (sp.vote_arith == 1): same dets / hist pointers,
// on a stream ordered behind the voting launch; its votes must be in before the tail reads the histograms
// ... and, behind it, the strict re-vote of the frames that lost an entry to a full list (they come out unmarked)
hipError_t launch_k2_fixup(mpe_detections* dets, int n_frames, const SolveParams& sp, uint32_t* hist,
                           const VoteFixup& fixup, hipStream_t s, int relost_blocks = 32,
                           const int* item_range = nullptr);  // (item_range: as launch_k2_vote's, for the re-vote)
// splits < 0 (plain kernel): -splits blocks per frame that divide the marker PERMUTATIONS among themselves and keep
// their slice of the per-permutation table in LDS (k2_table_slices says when and into how many)
int k2_table_slices(int n_markers);
// item_range (device, 2 ints per frame, forensics only): frame f votes with hypotheses [lo, hi) only, see mpe_vote_items
// the tail = k3a_validate + k3b_refine; mid_buf: k3_mid_bytes(n_frames) of device memory handed from one to the other
size_t k3_mid_bytes(int n_frames);
hipError_t launch_k3_tail(const mpe_detections* dets, const uint32_t* hist, int n_frames, const SolveParams& sp,
                          mpe_result* results, uint32_t* corr_out, const uint32_t* corr_in, const double* nn_pred,
                          double nn_tol, void* mid_buf, hipStream_t s, int mode = 0);
// tracked frames as ONE launch, a block per frame (mpe_k3.hip k_track_frame): scan of the ROI slot, small blob tier,
// nearest-neighbour correspondences + validation, refinement; the records go to three device arrays and, when h_* are
// given (pinned host memory), are stored there by the kernel itself.  Small blob tier only: a frame it cannot hold has
// det.status MPE_FRAME_TOO_MANY_ROWS and the caller repeats the submission through launch_k1a_scan / launch_k1b_blobs /
// launch_k3_tail.
struct TrackFramesArgs {
  const uint8_t* pix;          // frame b at pix + b * slot_bytes (g.rows x g.pitch)
  size_t slot_bytes;
  const double* pred;          // 2 * MPE_MAX_MARKERS doubles per frame
  const void* wins;            // n x {rows, cols, roi_x, roi_y} ints, or nullptr
  unsigned long long* flags;   // n_frames * track_flag_words(g) words
  uint32_t* hist;              // n_frames * MPE_HIST_STRIDE words
  void* mid;                   // k3_mid_bytes(n_frames)
  mpe_detections* dets;        // device records
  uint32_t* corr;
  mpe_result* res;
  mpe_detections* h_dets;      // pinned host records (optional)
  uint32_t* h_corr;
  mpe_result* h_res;
  unsigned long long* phase_clocks;  // optional: 5 shader-clock stamps of frame 0 (start / scan / blobs / validation / refinement)
};
size_t track_flag_words(const FrameGeom& g);
hipError_t launch_track_frames(const TrackFramesArgs& t, int n_frames, const FrameGeom& g, const DetectParams& dp,
                               const SolveParams& sp, double nn_tol, hipStream_t s);
hipError_t launch_repack(const uint8_t* src, size_t src_stride, size_t src_frame_stride, int n_frames, int roi_x,
                         int roi_y, int roi_w, int roi_h, uint8_t* dst, int dst_pitch, hipStream_t s);

// sensor_msgs/Image payload (bgr8 / rgb8 / bgra8 / rgba8 / mono16 / mono8, MPE_ENC_*) -> packed mono8 frames
hipError_t launch_to_mono8(const uint8_t* src, size_t src_stride, size_t src_frame_stride, int encoding, int big_endian,
                           int n_frames, int rows, int cols, uint8_t* dst, hipStream_t s);
hipError_t launch_spin(unsigned long long ticks_100mhz, hipStream_t s);
hipError_t launch_p3p_batch(const double* fv, const double* wp, int n, double* sol, int* status, hipStream_t s);
hipError_t launch_quartic_batch(const double* factors, int n, int variant, double* roots, hipStream_t s);

}  // namespace mpe
