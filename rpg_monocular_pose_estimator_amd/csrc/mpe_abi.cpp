// mpe_abi.cpp — host side of libmpe_hip.so: the C ABI of include/mpe.h on top of the gfx950
// kernels.  Owns device workspaces, the HIP stream, parameter marshalling; launches
// K1a -> K1b -> K2 -> K3 per batch.  No torch, no CPU fallback: without a HIP device every entry
// point fails with MPE_ERR_NO_DEVICE / MPE_ERR_HIP.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
// RCCL is dlopen'ed at first use by the one optional entry that needs it (mpe_estimate_batch_multi_device_gather): its
// header is used when it is there, else the handful of declarations that entry touches are spelled out — the library
// builds, and everything else works, on a box without RCCL.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1 } ncclDataType_t;
#endif

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "mpe_internal.h"

using namespace mpe;

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) return e;
    cap = want;
    return hipSuccess;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

}  // namespace

struct mpe_handle {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;
  DevBuf frames, flags, dets, hist, results, corr, mtab, work, scratch, track, mid;
  // hypotheses the fast voting kernel leaves to the strict arithmetic (VoteFixup, mpe_internal.h): a control block of
  // kMaxSub x MPE_FIX_CTL_WORDS counters, then one list region per voting launch that can be in flight (sub-batch slot;
  // only as many regions as a call has needed so far: fix_slots)
  DevBuf fix;
  unsigned fix_cap = 0;                 // entries per slot of the current layout
  int fix_slots = 0;                    // list regions of the current layout
  unsigned fix_cap_limit = 0;           // option "vote_list_cap" (tests): entries per slot at most; 0 = no limit
  unsigned long long fix_relost_base = 0, fix_wide_base = 0;
  bool fix_pending[16] = {};            // slot: a voting launch has appended, its fix-up has not been launched yet
  // a pinned host mirror of the lists' control block, copied behind the first fix-up launch of every call and read —
  // stale by a call or two, which is all a heuristic needs — when the next call sizes its re-vote launches (relost_grid)
  unsigned* fix_ctl_host = nullptr;
  unsigned long long relost_prev_sum = 0;
  bool relost_hot = false;              // frames were marked for the strict re-vote since the reading before
  unsigned long long fix_items_base = 0, fix_overflow_base = 0;  // cumulative counters of layouts that were replaced
  void* mailbox = nullptr;  // pinned host memory for the single-frame tracking step (ROI in, record out)
  size_t mailbox_cap = 0;
  // host-side time of the tracked frame (option "track_profile" = 1 starts / resets): sums in ns
  int track_profile = 0;
  long long track_ns[3] = {0, 0, 0}, track_steps = 0;  // pack, enqueue, wait
  // what mpe_track_step_batch_collect needs to repeat a submission whose blobs overflowed the small tier
  struct PendingTrack {
    bool optimistic = false;
    bool fused = false;        // the submission ran as k_track_frame: its flag words are per block, not the scan's bitstream
    size_t slot_bytes = 0;
    FrameGeom g;
    DetectParams dp;
    SolveParams sp;
    double nn_tol = 0;
    size_t rec_bytes = 0;
    const uint8_t* d_pix = nullptr;
    const void* d_wins = nullptr;
    const double* d_pred = nullptr;
  } pending_track;
  int pending_track_n = 0;            // mpe_track_step_batch_submit without its _collect yet: streams in flight
  const uint8_t* pending_track_rec = nullptr;
  // How many detections the frames of a pipelined call are expected to carry: picks the voting-kernel variant (from 9
  // on: the scan-carrying kernel with an occupancy grid of the detections, mpe_k2.hip K2_CGRID) and sizes the suspect
  // lists.  Never a matter of correctness.  Option "detections_hint" (0 = automatic: the number of markers, or what the
  // last call whose records came back to the host saw, det_seen)
  int detections_hint = 0;
  int det_seen = 0;
  unsigned long long* track_clk = nullptr;   // option "track_phase_clocks": pinned, device-visible; 5 stamps per frame
  unsigned long long track_clk_sum[4] = {0, 0, 0, 0};
  long long track_clk_n = 0;
  int track_fused = 2;         // option "track_fused": a tracked frame's optimistic pass as one launch (k_track_frame)
  int lds_budget = 64 * 1024;  // K1b dynamic LDS per wave (bitmap rows)
  int vote_splits = 0;         // 0 = auto
  int vote_arith = 3;          // 3 (default since round 6) = fast voting arithmetic + strict re-evaluation of the hypotheses
                               //     it cannot decide, the strict item evaluating the quartic's three complex powers as
                               //     libstdc++ / glibc do (mpe_ddmath.h): the CPU reference's digits in Ferrari's corner;
                               // 1 = the same with exact products / cbrt(hypot) (default of rounds 4 - 5), 0 / 4 = the
                               //     strict kernel (IEEE operators, the validation kernel's P3P) with the powers of 1 / 3,
                               //     2 = the fast arithmetic alone (round-3 behaviour, A/B only)
  int assume_side_streams = 0; // option: take the side streams of schedules 4 / 6 as concurrent without the spin probe —
                               // for counter passes: the profiler serialises kernels, the probe then fails and the
                               // call would fall back to schedule 3, i.e. other launch shapes than the timed run's
  int force_rccl_gather = 0;   // option: mpe_estimate_batch_multi_device_gather sends EVERY shard's records (shard 0's
                               // too: a send to itself) through RCCL, also with one handle — the self-test of that leg
                               // on a 1-GPU box (dlopen, ncclCommInitAll, grouped send / recv)
  int refine_variant = 0;      // refinement kernel: 0 automatic (16 lanes per frame up to 2048 frames per launch, else one
                               // lane per frame), 1 / 2 force one of them; bit-identical results
  int k1a_dummy_lds = -1;      // tuning: dummy LDS per scan block in the two-stream schedule (-1 = automatic)
  int last_schedule = 0;       // schedule the last large batch actually ran with
  int pipeline_mode = -1;      // -1 automatic; 0 two-stream staggered pipeline, 3 fused single stream (scan rides in the voting
                               // kernel), 4 fused + validate / refine on a side stream, 6 = 4 + the scan split between a
                               // side k1a_scan and the rider (default)
  bool profiling = false;
  int pipeline = 16;  // a large call is cut into up to this many sub-batches (about 16384 frames each, never
                      // below 8192) that the schedules pipeline against each other; 1 = one chain of kernels
  static const int kMaxSub = 16;
  hipStream_t sub_stream[kMaxSub] = {};
  bool streams_probed = false;  // sub_stream[0] / [1] verified to execute concurrently
  int streams_concurrent = -1;  // result of the probe: 1 yes, 0 no pair found, -1 not probed
  hipEvent_t sub_done[kMaxSub] = {};
  hipEvent_t vote_done[kMaxSub] = {};
  hipEvent_t fork_ev = nullptr;
  hipStream_t copy_stream = nullptr;  // host-frame ingest: the H2D copy of chunk c + 1 runs beside the kernels of chunk c
  hipEvent_t copy_done[2] = {nullptr, nullptr};
  int ingest_chunk = 2048;            // frames per ingest chunk (option "ingest_chunk"; 0 = one blocking copy per call)
  hipStream_t scan_stream = nullptr;  // mode 6: part of the next-but-one sub-batch's scan beside blobs / tail
  hipEvent_t scanpart_done[kMaxSub] = {};
  // mode 6: resident blocks per CU of the side scan (4 waves each) and the share of a sub-batch it scans on the side
  // stream.  Round 4: ONE block, 28 % — three blocks (round 3) crowd the blob kernel (window 0.84 instead of 0.50 ms per
  // 32 768 frames) and, once the voting launch got shorter, did not even finish inside blob window + vote; one block
  // streams at ~1.5 TB/s beside the rider for the whole period (same-box sweeps: profiles/round4_sweep_side_scan.json)
  int side_scan_blocks = 1;
  // Stream priority of the two side streams (options "tail_priority" / "scan_priority": -1 lowest, 0 default level,
  // 1 highest, 2 = the default level through the priority entry point; applied when the streams are created).  Round 5:
  // NOT the default level.  The runtime multiplexes the streams of one priority level onto GPU_MAX_HW_QUEUES (4)
  // hardware queues; a caller with a work stream, a consumer stream for the records and torch's own streams already
  // fills them, and a side stream that shares a queue with the consumer's 113 MB D2H copy stalls behind it at every
  // submission boundary (window in front of the first voting launch 0.8 - 1.2 ms instead of 0.55; step 18.45 ->
  // 17.55 ms, profiles/round5_exp_side_priorities.json).  Streams of another level get queues of their own.  Both side
  // streams sit on the SAME non-default level: over three boxes (calls r5f, r5k, r5l, interleaved repetitions) the two
  // same-level settings average 17.4 ms per step, the two mixed ones 17.8; highest rather than lowest because a
  // sub-batch's tail then finishes in 1.7 instead of 2.1 ms (its thin kernels get their blocks dispatched in front of
  // the voting launch's 32 768 pending ones) and the tail chain must never become longer than the period.
  int tail_priority = 1;
  int scan_priority = 1;
  int scan_split_pct = 28;
  unsigned long long last_rider_bytes = 0;  // bytes one fused voting launch scanned in the last large call
  hipStream_t tail_stream = nullptr;  // fused schedule, mode 4: validate + refine of sub-batch s beside blobs(s + 1)
  hipEvent_t tail_done = nullptr;
  // ---- streaming submissions (mpe_estimate_batch_device_submit / _collect): up to two batches in flight
  hipEvent_t batch_done[2] = {nullptr, nullptr};  // records of submission q complete: batch_done[q & 1]
  unsigned submit_seq = 0, collect_seq = 0;       // submissions made / collected
  hipEvent_t tail_sub_done[kMaxSub] = {};         // tail(s) of the previous submission has read dets / hist of region s
  bool tail_sub_pending = false;
  int tail_last = 0;                              // index of the last event recorded there
  int tail_per = 0;                               // frames per region of the submission those events belong to
  // image scan of the NEXT submission's first sub-batch, carried by the last voting launch of this one
  struct Prefetch {
    bool valid = false;
    const uint8_t* frames = nullptr;
    int per = 0;              // frames of that sub-batch
    unsigned long long* flags_ptr = nullptr;  // where its flag words are (the producer's layout, not the consumer's)
    size_t frame_bytes = 0;
    int thr = 0;
    void* flags_base = nullptr;  // flags buffer the prefetched words live in (a re-allocation loses them)
    size_t fw_per = 0;
    bool side_part = false;      // part of it came from the side scan: wait for prefetch_side_done
  } prefetch;
  hipEvent_t prefetch_side_done = nullptr;
  hipEvent_t next_ready = nullptr;  // one-shot, consumed by the next _submit (mpe_stream_next_ready)
  bool done_recorded = false;  // run_pipeline has recorded batch_done[submit_seq & 1] itself (fused schedules)
  int last_nsub = 0, last_per = 0;  // work-list layout of the last pipelined batch (option "overflow_*")
  // the marker-permutation table in mtab is that of these markers, built in the order of this stream (a call with the
  // same rig on the same stream does not rebuild it: one 25-85 us single-wave kernel less in front of every batch)
  double mtab_markers[MPE_MAX_MARKERS * 3] = {};
  int mtab_n = 0;
  const void* mtab_ptr = nullptr;
  hipStream_t mtab_stream = nullptr;
  // option "vote_events" = N > 0: a pair of timing events around every voting launch that carries a scan, for the
  // launches of the last N pipelined calls (ring) — the duration of the dominant kernel INSIDE a timed region, with
  // nothing else recorded; read back as "vote_launch_ns_mean" / "vote_launches" (synchronises the stream)
  struct VotePair {
    hipEvent_t a = nullptr, b = nullptr;
    bool used = false;
  };
  std::vector<VotePair> vote_ev;  // N x kMaxSub
  int vote_ev_calls = 0;          // N
  long long vote_ev_seq = 0;      // pipelined calls seen since the option was set
  std::vector<std::pair<size_t, int>> blob_launches;  // its blob launches: work-list offset (ints), frames
  size_t work_ints = 0;
  // side streams of schedules 4 / 6 verified (spin probe) to execute beside the caller's stream
  int side_streams_ok = -1;           // 1 yes, 0 no concurrent set found (-> schedule 3), -1 not probed
  hipStream_t probed_for = nullptr;   // the caller's stream the verdict holds for
  bool probed_scan = false;           // ... including the scan stream
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  // per-sub-batch kernel brackets for the pipelined mode: [s][0..1] scan, [2..3] blobs (all tiers),
  // [4..5] vote, [6..7] tail
  hipEvent_t pev[kMaxSub][8] = {};
  int prof_launches = 0;       // sub-batches (= launches per kernel) of the last profiled call
  int prof_frames_per_launch = 0;
  bool prof_pipelined = false;
  bool have_ms = false;
  // chunked host ingest with profiling on: the kernel times of ALL chunks summed (mpe_last_kernel_ms), not the last one's
  bool ms_accum_valid = false;
  float ms_accum[5] = {0, 0, 0, 0, 0};
};

namespace {

int fail(mpe_handle* h, int code, const char* what, hipError_t e = hipSuccess) {
  if (h) {
    h->err = what;
    if (e != hipSuccess) {
      h->err += ": ";
      h->err += hipGetErrorString(e);
    }
  }
  return code;
}

#define HIP_TRY(h, call)                                         \
  do {                                                           \
    hipError_t e__ = (call);                                     \
    if (e__ != hipSuccess) return fail(h, MPE_ERR_HIP, #call, e__); \
  } while (0)

// Every entry point that re-uses the handle's device buffers on its stream: select the device and, if a streaming
// submission (mpe_estimate_batch_device_submit) still has validate / refine kernels on the internal tail stream, make
// the handle's stream wait for them first (they read the detection / histogram buffers).  The streaming entry itself
// orders those buffers region by region instead (run_pipeline).
int enter(mpe_handle* h) {
  hipError_t e = hipSetDevice(h->device);
  if (e != hipSuccess) return fail(h, MPE_ERR_HIP, "hipSetDevice", e);
  if (h->tail_sub_pending) {
    e = hipStreamWaitEvent(h->stream, h->tail_sub_done[h->tail_last], 0);
    if (e != hipSuccess) return fail(h, MPE_ERR_HIP, "hipStreamWaitEvent", e);
    h->tail_sub_pending = false;
  }
  return MPE_OK;
}
#define ENTER(h)                  \
  do {                            \
    const int rc__ = enter(h);    \
    if (rc__ != MPE_OK) return rc__; \
  } while (0)

unsigned factorial_u32(int n) {  // combinations.cpp:34-40: 32-bit wrap-around kept on purpose
  unsigned r = 1;
  for (int i = 2; i <= n; ++i) r *= (unsigned)i;
  return r;
}
unsigned num_combinations_u32(unsigned n, unsigned k) {  // combinations.cpp:42-45
  const unsigned den = factorial_u32((int)k) * factorial_u32((int)(n - k));
  return den ? factorial_u32((int)n) / den : 0u;
}

// cv::getGaussianKernel(n, sigma, CV_32F) quantised to 8 fractional bits, n = cvRound(6*sigma+1)|1
// (what GaussianBlur(ksize = 0) uses for CV_8U, led_detector.cpp:48-51)
int gaussian_taps(double sigma, int* taps) {
  if (!(sigma > 0)) return -1;
  const int n = (int)std::lrint(sigma * 3 * 2 + 1) | 1;
  if (n > MPE_MAX_KSIZE) return -1;
  float cf[MPE_MAX_KSIZE];
  const double scale2x = -0.5 / (sigma * sigma);
  double sum = 0;
  for (int i = 0; i < n; ++i) {
    const double x = i - (n - 1) * 0.5;
    cf[i] = (float)std::exp(scale2x * x * x);
    sum += cf[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < n; ++i) {
    cf[i] = (float)(cf[i] * sum);
    taps[i] = (int)std::lrint((double)cf[i] * 256.0);
  }
  return n;
}

int make_detect_params(const mpe_params* p, const double K[9], const double* D, int nD, int roi_x, int roi_y,
                       DetectParams& dp) {
  std::memset(&dp, 0, sizeof(dp));
  dp.thr = p->threshold_value < -1 ? -1 : (p->threshold_value > 255 ? 255 : p->threshold_value);
  dp.ksize = gaussian_taps(p->gaussian_sigma, dp.taps);
  if (dp.ksize < 0) return -1;
  pack_taps(dp);
  dp.min_area = p->min_blob_area;
  dp.max_area = p->max_blob_area;
  dp.max_wh = p->max_width_height_distortion;
  dp.max_circ = p->max_circular_distortion;
  for (int i = 0; i < 9; ++i) dp.K[i] = K[i];
  dp.ifx = 1. / K[0];
  dp.ify = 1. / K[4];
  for (int i = 0; i < 8; ++i) dp.k[i] = (D && i < nD) ? D[i] : 0.0;
  dp.undist_iters = (D && nD > 0) ? 5 : 0;
  dp.roi_x = roi_x;
  dp.roi_y = roi_y;
  return 0;
}

int make_solve_params(const mpe_handle* h, const mpe_params* p, const double* markers, int n_markers, const double K[9],
                      SolveParams& sp) {
  if (n_markers < 0 || n_markers > MPE_MAX_MARKERS) return -1;
  std::memset(&sp, 0, sizeof(sp));
  sp.n_markers = n_markers;
  for (int i = 0; i < 3 * n_markers; ++i) sp.markers[i] = markers[i];
  sp.fx = K[0];
  sp.fy = K[4];
  sp.cx = K[2];
  sp.cy = K[5];
  sp.back_tol = p->back_projection_pixel_tolerance;
  sp.certainty_thr = p->certainty_threshold;
  sp.valid_corr_thr = p->valid_correspondence_threshold;
  sp.hist_thr = p->histogram_threshold ? p->histogram_threshold : num_combinations_u32((unsigned)n_markers, 3);
  sp.vote_arith = h->vote_arith;
  sp.refine_variant = h->refine_variant;
  return 0;
}

int make_geom(const mpe_handle* h, int rows, int cols, FrameGeom& g) {
  if (rows <= 0 || cols <= 0 || rows > 4096 || cols > 4000) return -1;
  g.rows = rows;
  g.cols = cols;
  g.pitch = (cols + 15) & ~15;
  g.segs_per_row = g.pitch / 16;
  g.segs_per_frame = rows * g.segs_per_row;
  g.wb = (cols + 2 + 63) / 64 + 1;
  g.rw = (rows + 63) / 64;
  g.tw = (g.segs_per_row + 63) / 64;
  // bytes per bitmap row: 3 bitmaps + todo bits + its share of the word mask
  const size_t per_slot = (size_t)(3 * g.wb + g.tw) * 8 + (size_t)g.wb / 8 + 1;
  size_t fixed = 2 * (size_t)g.rw * 8 + 2 * (size_t)g.rw * 4 + 64;
  long cap = ((long)h->lds_budget - (long)fixed) / (long)per_slot;
  if (cap > rows + 2 + rows / 2) cap = rows + 2 + rows / 2;  // every row active, worst-case separators
  if (cap < 8) return -1;
  g.slot_cap = (int)cap;
  return 0;
}

size_t flag_words(size_t n_bytes) {  // K1a writes whole chunks of up to 8 words
  const size_t n_seg = n_bytes / 16;
  return ((n_seg + 511) / 512) * 8 + 8;
}

// Bring `n_frames` frames into the packed device layout.  Returns the device pointer to use.
int stage_frames(mpe_handle* h, const uint8_t* frames, int n_frames, int rows, int cols, size_t stride,
                 size_t frame_stride, int on_device, int roi_x, int roi_y, int roi_w, int roi_h, const FrameGeom& g,
                 const uint8_t** d_out) {
  const bool full = (roi_x == 0 && roi_y == 0 && roi_w == cols && roi_h == rows);
  const bool packed = full && stride == (size_t)g.pitch && frame_stride == (size_t)rows * g.pitch &&
                      (reinterpret_cast<uintptr_t>(frames) & 15) == 0 && g.pitch == cols;
  if (on_device && packed) {
    *d_out = frames;
    return MPE_OK;
  }
  const size_t bytes = (size_t)n_frames * g.rows * g.pitch;
  HIP_TRY(h, h->frames.reserve(bytes + 16));
  uint8_t* dst = static_cast<uint8_t*>(h->frames.p);
  if (on_device) {
    HIP_TRY(h, launch_repack(frames, stride, frame_stride, n_frames, roi_x, roi_y, roi_w, roi_h, dst, g.pitch,
                             h->stream));
  } else {
    if (g.pitch != roi_w) HIP_TRY(h, hipMemsetAsync(dst, 0, bytes, h->stream));
    if (full && stride == (size_t)cols && frame_stride == (size_t)rows * cols && g.pitch == cols) {
      HIP_TRY(h, hipMemcpyAsync(dst, frames, bytes, hipMemcpyHostToDevice, h->stream));
    } else {
      for (int f = 0; f < n_frames; ++f) {
        const uint8_t* src = frames + (size_t)f * frame_stride + (size_t)roi_y * stride + roi_x;
        HIP_TRY(h, hipMemcpy2DAsync(dst + (size_t)f * g.rows * g.pitch, g.pitch, src, stride, roi_w, roi_h,
                                    hipMemcpyHostToDevice, h->stream));
      }
    }
  }
  *d_out = dst;
  return MPE_OK;
}

int det_hint_for(const mpe_handle* h, int n_markers) {
  const int v = h->detections_hint > 0 ? h->detections_hint : h->det_seen;
  return std::min(MPE_FAST_VOTE_DETECTIONS, std::max(n_markers, v));
}

// > 0: blocks per frame, each block a share of the flattened (triple, permutation) items;  < 0: -(blocks per frame),
// each block a share of the marker PERMUTATIONS whose table slice it keeps in LDS (6 .. 10 markers, fast arithmetic)
int auto_splits(const mpe_handle* h, int n_frames, int n_markers) {
  if (h->vote_splits > 0) return h->vote_splits;
  if (!vote_arith_is_strict(h->vote_arith) && h->vote_splits == 0) {
    const int slices = k2_table_slices(n_markers);
    if (slices > 0) return -slices;
  }
  // few frames with a large hypothesis space: spread one frame over several workgroups
  if (n_frames >= 1024 || n_markers <= 5) return 1;
  int s = 2048 / std::max(1, n_frames);
  return std::max(1, std::min(s, 64));
}

// ---- strict re-evaluation of the fast voting kernel's suspects (VoteFixup) ----------------------------------------
constexpr size_t kFixCtlBytes = (size_t)mpe_handle::kMaxSub * MPE_FIX_CTL_WORDS * sizeof(unsigned);
constexpr size_t kFixEntryBytes = 2 * sizeof(unsigned long long);
// sum of the per-slot cumulative counters (synchronises the device); which = 1 list-full events, 3 entries
// re-evaluated, 6 frames voted again after a list-full event
int fix_counter_sum(mpe_handle* h, int which, unsigned long long& out) {
  out = which == 1 ? h->fix_overflow_base : which == 6 ? h->fix_relost_base : which == 7 ? h->fix_wide_base : h->fix_items_base;
  if (!h->fix.p) return MPE_OK;
  HIP_TRY(h, hipDeviceSynchronize());
  unsigned ctl[mpe_handle::kMaxSub * MPE_FIX_CTL_WORDS];
  HIP_TRY(h, hipMemcpy(ctl, h->fix.p, sizeof(ctl), hipMemcpyDeviceToHost));
  for (int s = 0; s < mpe_handle::kMaxSub; ++s) out += ctl[MPE_FIX_CTL_WORDS * s + which];
  return MPE_OK;
}
// The list of voting launch `slot` (sub-batch index; 0 for single launches) of a call that uses `n_slots` of them, sized
// for n_frames frames: ~0.6 % of the hypotheses go to the list, the region holds 1/32 of them
// (>= 64 per frame; small launches are sized for the capacity limit of 32 detections, whatever the caller expects),
// within 1 GB per slot.  A full list is not an error and costs no pose: the frames that lost an entry are voted again
// by the strict loop nest behind the fix-up kernel (k2_vote_relost); "vote_fixup_overflow" counts the events,
// "vote_relost_frames" the frames.  If the device cannot hold the layout the list shrinks (down to 4 096 entries)
// before the call fails.
int vote_fixup_for(mpe_handle* h, int slot, int n_slots, int n_frames, int n_markers, int n_det_hint, hipStream_t st,
                   VoteFixup& fx) {
  fx = VoteFixup{nullptr, nullptr, 0u, 0u};
  if (vote_arith_is_strict(h->vote_arith) || n_markers < 4 || slot < 0 || slot >= mpe_handle::kMaxSub) return MPE_OK;
  n_slots = std::min((int)mpe_handle::kMaxSub, std::max(n_slots, slot + 1));
  // (wider frames than MPE_FAST_VOTE_DETECTIONS append nothing: the strict loop nest votes them)
  const long long nd = n_frames <= 256 ? MPE_FAST_VOTE_DETECTIONS
                                       : std::min(MPE_FAST_VOTE_DETECTIONS, std::max(n_det_hint, n_markers) + 4);
  const long long items = nd * (nd - 1) * (nd - 2) / 6 * n_markers * (n_markers - 1) * (n_markers - 2);
  unsigned long long want = (unsigned long long)n_frames * (unsigned long long)std::max(64ll, items / 32);
  const unsigned long long most = (1ull << 30) / kFixEntryBytes;
  want = std::min(want, most);
  if (h->fix_cap_limit) want = std::min<unsigned long long>(want, h->fix_cap_limit);
  if (want > h->fix_cap || n_slots > h->fix_slots || !h->fix.p) {
    // a new layout: nothing may be in flight on the old one (hipFree inside reserve() waits for the device anyway)
    if (h->fix.p) {
      unsigned long long v = 0;
      int rc = fix_counter_sum(h, 1, v);
      if (rc) return rc;
      h->fix_overflow_base = v;
      rc = fix_counter_sum(h, 3, v);
      if (rc) return rc;
      h->fix_items_base = v;
      rc = fix_counter_sum(h, 6, v);
      if (rc) return rc;
      h->fix_relost_base = v;
      rc = fix_counter_sum(h, 7, v);
      if (rc) return rc;
      h->fix_wide_base = v;
    }
    HIP_TRY(h, hipDeviceSynchronize());
    const int slots = std::max(n_slots, h->fix_slots);  // (a layout only grows)
    unsigned long long cap = std::max<unsigned long long>(want, h->fix_cap);
    h->fix.release();
    h->fix_cap = 0;
    h->fix_slots = 0;
    hipError_t e = hipSuccess;
    for (;; cap /= 4) {
      e = h->fix.reserve(kFixCtlBytes + (size_t)slots * cap * kFixEntryBytes);
      if (e == hipSuccess || cap <= 4096) break;
      (void)hipGetLastError();  // (out of memory: a smaller list only means more frames voted twice)
    }
    HIP_TRY(h, e);
    h->fix_cap = (unsigned)cap;
    h->fix_slots = slots;
    HIP_TRY(h, hipMemsetAsync(h->fix.p, 0, kFixCtlBytes, st));
    HIP_TRY(h, hipStreamSynchronize(st));  // (other streams may be the first to touch it)
    for (auto& b : h->fix_pending) b = false;
  }
  fx.ctl = static_cast<unsigned*>(h->fix.p) + MPE_FIX_CTL_WORDS * slot;
  fx.list = reinterpret_cast<unsigned long long*>(static_cast<unsigned char*>(h->fix.p) + kFixCtlBytes) +
            (size_t)slot * h->fix_cap * 2;
  fx.cap = h->fix_cap;
  fx.screen = vote_arith_screens(h->vote_arith) ? 1u : 0u;
  if (h->fix_pending[slot]) {  // an earlier call failed between a voting launch and its fix-up: drop those entries
    HIP_TRY(h, hipMemsetAsync(fx.ctl, 0, sizeof(unsigned), st));
    HIP_TRY(h, hipMemsetAsync(fx.ctl + 2, 0, sizeof(unsigned), st));
    HIP_TRY(h, hipMemsetAsync(fx.ctl + 5, 0, sizeof(unsigned), st));
  }
  h->fix_pending[slot] = true;
  return MPE_OK;
}
// Blocks of the strict re-vote launch behind a fix-up (k2_vote_relost).  As a rule nothing is marked and the launch
// only has to leave quickly: 32 blocks where it sits in the tail chain of a sub-batch beside the next voting launch
// (256 no-op blocks took 0.39 ms there, round 5).  Once frames HAVE been marked — a list that overflowed, or frames
// with more than MPE_FAST_VOTE_DETECTIONS detections — 32 blocks are a cliff: the launch then takes
// the whole chip, until a call goes by without a mark.  Small calls (single frames, the tracker's initialisation, the
// stage-level entries) have nothing beside them and always get a grid that follows their frames.
int relost_grid(mpe_handle* h, int n_frames) {
  const int wide = 2 * device_cu_count();
  if (h->relost_hot) return wide;
  if (n_frames < 4096) return std::min(wide, std::max(32, 8 * n_frames));
  return 32;
}
hipError_t fixup_launch(mpe_handle* h, int slot, mpe_detections* dets, int n_frames, const SolveParams& sp, uint32_t* hist,
                        const VoteFixup& fx, hipStream_t st, const int* item_range = nullptr) {
  if (!fx.ctl) return hipSuccess;
  if (slot == 0 && h->fix_ctl_host) {  // once per call: what the mirror says, then the next reading on its way
    unsigned long long sum = 0;
    for (int s = 0; s < mpe_handle::kMaxSub; ++s)
      sum += (unsigned long long)h->fix_ctl_host[MPE_FIX_CTL_WORDS * s + 1] + h->fix_ctl_host[MPE_FIX_CTL_WORDS * s + 7];
    h->relost_hot = sum != h->relost_prev_sum;
    h->relost_prev_sum = sum;
  }
  const hipError_t e = launch_k2_fixup(dets, n_frames, sp, hist, fx, st, relost_grid(h, n_frames), item_range);
  if (e != hipSuccess) return e;
  h->fix_pending[slot] = false;
  if (slot == 0) {
    if (!h->fix_ctl_host) {
      const hipError_t ea = hipHostMalloc(reinterpret_cast<void**>(&h->fix_ctl_host), kFixCtlBytes, hipHostMallocDefault);
      if (ea != hipSuccess) return ea;
      std::memset(h->fix_ctl_host, 0, kFixCtlBytes);
    }
    return hipMemcpyAsync(h->fix_ctl_host, h->fix.p, kFixCtlBytes, hipMemcpyDeviceToHost, st);
  }
  return hipSuccess;
}

// dummy LDS per block of the stand-alone scan kernel: the handle's tuning override, else 40 KB when the scan is
// about to share the chip with the voting kernel of another sub-batch (two-stream schedule), else none
int scan_lds(const mpe_handle* h, bool co_resident) {
  if (h->k1a_dummy_lds >= 0) return h->k1a_dummy_lds;
  return co_resident ? 40000 : 0;
}

void rec(mpe_handle* h, int i) {
  if (h->profiling && h->ev[i]) (void)hipEventRecord(h->ev[i], h->stream);
}

// front half (image scan + blob extraction) and back half (voting + tail) of the per-batch chain
int run_front(mpe_handle* h, hipStream_t st, bool prof, int chain, int chain_frames, const uint8_t* d_frames,
              int n_frames, const FrameGeom& g, const DetectParams& dp, const SolveParams* sp,
              unsigned long long* d_flags, mpe_detections* d_dets) {
  const size_t bytes = (size_t)n_frames * g.rows * g.pitch;
  if (prof) rec(h, 0);
  HIP_TRY(h, launch_k1a_scan(d_frames, bytes, d_flags, dp.thr, scan_lds(h, false), st));
  if (prof) rec(h, 1);
  HIP_TRY(h, launch_k1b_blobs(d_frames, d_flags, n_frames, g, dp, d_dets,
                              static_cast<int*>(h->work.p) + (size_t)chain * 2 * (chain_frames + 1),
                              static_cast<uint8_t*>(h->scratch.p), h->scratch.cap, sp ? sp->n_markers : 0, st));
  if (prof) rec(h, 2);
  return MPE_OK;
}

int run_back(mpe_handle* h, hipStream_t st, bool prof, int n_frames, const SolveParams* sp, mpe_detections* d_dets,
             uint32_t* d_hist, mpe_result* d_results, uint32_t* d_corr) {
  if (sp) {
    HIP_TRY(h, hipMemsetAsync(d_hist, 0, (size_t)n_frames * MPE_HIST_STRIDE * sizeof(uint32_t), st));
    VoteFixup fx;
    { const int rc = vote_fixup_for(h, 0, 1, n_frames, sp->n_markers, det_hint_for(h, sp->n_markers), st, fx); if (rc) return rc; }
    HIP_TRY(h, launch_k2_vote(d_dets, n_frames, *sp, static_cast<const double*>(h->mtab.p), d_hist,
                              auto_splits(h, n_frames, sp->n_markers), det_hint_for(h, sp->n_markers), st, nullptr, 0, nullptr, 0,
                              nullptr, nullptr, &fx));
    HIP_TRY(h, fixup_launch(h, 0, d_dets, n_frames, *sp, d_hist, fx, st));
    if (prof) rec(h, 3);
    HIP_TRY(h, launch_k3_tail(d_dets, d_hist, n_frames, *sp, d_results, d_corr, nullptr, nullptr, 0.0, h->mid.p, st));
  } else if (prof) {
    rec(h, 3);
  }
  if (prof) rec(h, 4);
  return MPE_OK;
}

// The two-stream software pipeline only pays when its side streams sit on DIFFERENT hardware queues.  The
// runtime multiplexes streams onto a few queues (GPU_MAX_HW_QUEUES, default 4) in an order that depends on
// which other streams the process created (e.g. torch's stream pool), so two fresh streams can end up
// serialised.  Probe once per handle: a 1 ms spin kernel on each candidate — concurrent streams finish both
// in ~1 ms, serialised ones in ~2 ms — and keep the first pair that overlaps.
int pick_concurrent_streams(mpe_handle* h) {
  if (h->streams_probed) return MPE_OK;
  HIP_TRY(h, hipStreamSynchronize(h->stream));  // once per handle: time the probe on an idle device
  const int kCandidates = 8;
  hipStream_t cand[kCandidates] = {};
  for (int i = 0; i < kCandidates; ++i) HIP_TRY(h, hipStreamCreateWithFlags(&cand[i], hipStreamNonBlocking));
  const unsigned long long ticks = 100000;  // 1 ms at 100 MHz
  // device time from the first launch to the end of both spins, by HIP events: t0 is recorded on a, b waits for it
  // (so neither spin starts early), t1 on a after b's completion event has been joined into a
  hipEvent_t t0 = nullptr, t1 = nullptr, eb = nullptr;
  HIP_TRY(h, hipEventCreate(&t0));
  HIP_TRY(h, hipEventCreate(&t1));
  HIP_TRY(h, hipEventCreateWithFlags(&eb, hipEventDisableTiming));
  auto both_ms = [&](hipStream_t a, hipStream_t b, double& ms) -> hipError_t {
    hipError_t e = hipStreamSynchronize(a);
    if (e != hipSuccess) return e;
    if ((e = hipStreamSynchronize(b)) != hipSuccess) return e;
    if ((e = hipEventRecord(t0, a)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(b, t0, 0)) != hipSuccess) return e;
    if ((e = launch_spin(ticks, a)) != hipSuccess) return e;
    if ((e = launch_spin(ticks, b)) != hipSuccess) return e;
    if ((e = hipEventRecord(eb, b)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(a, eb, 0)) != hipSuccess) return e;
    if ((e = hipEventRecord(t1, a)) != hipSuccess) return e;
    if ((e = hipEventSynchronize(t1)) != hipSuccess) return e;
    float fms = 0.f;
    if ((e = hipEventElapsedTime(&fms, t0, t1)) != hipSuccess) return e;
    ms = fms;
    return hipSuccess;
  };
  double warm = 0;
  HIP_TRY(h, both_ms(cand[0], cand[0], warm));  // first launch of the kernel (code object load) is not timed
  int ia = -1, ib = -1;
  for (int i = 0; i < kCandidates && ia < 0; ++i)
    for (int j = i + 1; j < kCandidates; ++j) {
      double ms = 0;
      HIP_TRY(h, both_ms(cand[i], cand[j], ms));
      if (ms < 1.6) {
        ia = i;
        ib = j;
        break;
      }
    }
  (void)hipEventDestroy(t0);
  (void)hipEventDestroy(t1);
  (void)hipEventDestroy(eb);
  h->streams_concurrent = ia >= 0 ? 1 : 0;
  if (ia < 0) {
    ia = 0;
    ib = 1;
  }
  int ic = -1;
  for (int i = 0; i < kCandidates; ++i)
    if (i != ia && i != ib) {
      ic = i;
      break;
    }
  for (int i = 0; i < 3; ++i)
    if (h->sub_stream[i]) (void)hipStreamDestroy(h->sub_stream[i]);
  h->sub_stream[0] = cand[ia];
  h->sub_stream[1] = cand[ib];
  h->sub_stream[2] = cand[ic];
  for (int i = 0; i < kCandidates; ++i)
    if (i != ia && i != ib && i != ic) (void)hipStreamDestroy(cand[i]);
  h->streams_probed = true;
  return MPE_OK;
}

// how a large batch is cut into sub-batches (shared by a call and by the previous call that prefetches for it)
void sub_batch_shape(const mpe_handle* h, int n_frames, size_t frame_bytes, bool have_sp, int vote_arith, int& nsub,
                     int& per) {
  // sub-batches of about 16384 frames (measured sweet spot at 752x480: 8192 and 32768 are 3-5 % slower), never
  // below 8192 (tail effects then cost more than the overlap gains)
  // (with the streaming entry, whose calls have no un-overlapped ends, 32768 frames per sub-batch measured 3 % faster
  //  than 16384 at 752x480 — half as many kernel boundaries; 65536: 1 % slower again)
  nsub = frame_bytes <= (size_t)512 * 1024 ? n_frames / 32768 : 0;
  if (nsub < 2) nsub = n_frames / 16384;
  if (nsub < 2) nsub = n_frames / 8192;
  if (nsub > h->pipeline) nsub = h->pipeline;
  if (nsub > mpe_handle::kMaxSub) nsub = mpe_handle::kMaxSub;
  if (nsub < 1 || !have_sp) nsub = 1;
  if (have_sp && vote_arith_is_strict(vote_arith)) nsub = 1;  // strict voting arithmetic: one plain chain of kernels (no scan rider)
  // frames per sub-batch: multiple of 64 so every sub-batch starts on a 16-byte / flag-word boundary
  per = nsub > 1 ? (((n_frames + nsub - 1) / nsub + 63) & ~63) : n_frames;
}

// streaming: what the caller knows about the submission that follows this one
struct StreamHint {
  const uint8_t* next_frames = nullptr;  // device frames of the next submission (same geometry / parameters), or null
  int n_next = 0;
  bool no_join = false;  // do not join the side streams back into the caller's stream: completion = batch_done event
  hipEvent_t next_ready = nullptr;  // the announced frames are final once this event has completed (mpe_stream_next_ready)
};

// Device time (ms) for one 1 ms spin kernel on each of two streams started together: ~1 when they execute
// concurrently, ~2 when the runtime put them on one hardware queue.
hipError_t spin_pair_ms(hipStream_t a, hipStream_t b, double& ms) {
  const unsigned long long ticks = 100000;  // 1 ms at 100 MHz
  hipEvent_t t0 = nullptr, t1 = nullptr, eb = nullptr;
  hipError_t e = hipEventCreate(&t0);
  if (e == hipSuccess) e = hipEventCreate(&t1);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&eb, hipEventDisableTiming);
  if (e == hipSuccess) e = hipStreamSynchronize(a);
  if (e == hipSuccess) e = hipStreamSynchronize(b);
  if (e == hipSuccess) e = hipEventRecord(t0, a);
  if (e == hipSuccess) e = hipStreamWaitEvent(b, t0, 0);
  if (e == hipSuccess) e = launch_spin(ticks, a);
  if (e == hipSuccess) e = launch_spin(ticks, b);
  if (e == hipSuccess) e = hipEventRecord(eb, b);
  if (e == hipSuccess) e = hipStreamWaitEvent(a, eb, 0);
  if (e == hipSuccess) e = hipEventRecord(t1, a);
  if (e == hipSuccess) e = hipEventSynchronize(t1);
  float fms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&fms, t0, t1);
  ms = fms;
  if (t0) (void)hipEventDestroy(t0);
  if (t1) (void)hipEventDestroy(t1);
  if (eb) (void)hipEventDestroy(eb);
  return e;
}

// Schedules 4 / 6 put the validate / refine kernels (and, in 6, a share of the image scan) on internal side streams;
// that only pays when those streams execute BESIDE the caller's stream.  The runtime multiplexes streams onto a few
// hardware queues (GPU_MAX_HW_QUEUES) in an order that depends on the process's other streams, so the overlap can
// silently vanish (Schedules).  Verified here once per (handle, caller stream): every pair of {caller's
// stream, tail stream, scan stream} must run two 1 ms spin kernels in ~1 ms; a side stream that shares a queue is
// replaced (the rejected ones stay alive until the end so that the runtime hands out other queues).  No concurrent set
// after 8 replacements -> side_streams_ok = 0 and the caller falls back to the one-stream schedule 3.
// a side stream with the priority the handle asks for (hipStreamCreateWithPriority: lower number = higher priority)
hipError_t make_side_stream(hipStream_t* s, int want) {
  if (want == 0) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
  int least = 0, greatest = 0;
  hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
  if (e == hipSuccess)
    // (2: the priority entry point at the default level — separates "which hardware queue" from "which priority")
    e = hipStreamCreateWithPriority(s, hipStreamNonBlocking, want == 2 ? 0 : (want < 0 ? least : greatest));
  if (e != hipSuccess) {  // a runtime without priority levels: an ordinary stream
    (void)hipGetLastError();
    e = hipStreamCreateWithFlags(s, hipStreamNonBlocking);
  }
  return e;
}

int ensure_side_streams(mpe_handle* h, bool need_scan) {
  if (h->side_streams_ok >= 0 && h->probed_for == h->stream && (!need_scan || h->probed_scan)) return MPE_OK;
  if (h->assume_side_streams) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (!h->tail_stream) HIP_TRY(h, make_side_stream(&h->tail_stream, h->tail_priority));
    if (need_scan && !h->scan_stream) HIP_TRY(h, make_side_stream(&h->scan_stream, h->scan_priority));
    h->side_streams_ok = 1;
    h->streams_concurrent = -1;  // (not probed)
    h->probed_for = h->stream;
    h->probed_scan = need_scan;
    h->tail_sub_pending = false;
    return MPE_OK;
  }
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (!h->tail_stream) HIP_TRY(h, make_side_stream(&h->tail_stream, h->tail_priority));
  if (need_scan && !h->scan_stream) HIP_TRY(h, make_side_stream(&h->scan_stream, h->scan_priority));
  HIP_TRY(h, hipStreamSynchronize(h->tail_stream));
  if (h->scan_stream) HIP_TRY(h, hipStreamSynchronize(h->scan_stream));
  double ms = 0;
  HIP_TRY(h, spin_pair_ms(h->stream, h->stream, ms));  // first launch of the kernel (code object load) is not timed
  std::vector<hipStream_t> rejected;
  bool ok = false;
  for (int attempt = 0; attempt <= 8 && !ok; ++attempt) {
    bool tail_bad = false, scan_bad = false;
    HIP_TRY(h, spin_pair_ms(h->stream, h->tail_stream, ms));
    if (ms >= 1.6) tail_bad = true;
    if (!tail_bad && need_scan) {
      HIP_TRY(h, spin_pair_ms(h->stream, h->scan_stream, ms));
      if (ms >= 1.6) scan_bad = true;
      if (!scan_bad) {
        HIP_TRY(h, spin_pair_ms(h->tail_stream, h->scan_stream, ms));
        if (ms >= 1.6) scan_bad = true;
      }
    }
    if (!tail_bad && !scan_bad) {
      ok = true;
      break;
    }
    if (attempt == 8) break;
    hipStream_t fresh = nullptr;
    HIP_TRY(h, make_side_stream(&fresh, tail_bad ? h->tail_priority : h->scan_priority));
    if (tail_bad) {
      rejected.push_back(h->tail_stream);
      h->tail_stream = fresh;
    } else {
      rejected.push_back(h->scan_stream);
      h->scan_stream = fresh;
    }
  }
  for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
  h->side_streams_ok = ok ? 1 : 0;
  h->streams_concurrent = h->side_streams_ok;
  h->probed_for = h->stream;
  h->probed_scan = need_scan;
  h->tail_sub_pending = false;  // (everything was synchronised above)
  return MPE_OK;
}

// timing events around one voting launch (option "vote_events"); slot = sub-batch index of the current call
hipError_t vote_ev_begin(mpe_handle* h, int slot, hipStream_t st) {
  if (h->vote_ev_calls <= 0) return hipSuccess;
  mpe_handle::VotePair& p = h->vote_ev[(size_t)(h->vote_ev_seq % h->vote_ev_calls) * mpe_handle::kMaxSub + slot];
  p.used = false;
  if (!p.a) {
    hipError_t e = hipEventCreate(&p.a);
    if (e != hipSuccess) return e;
    e = hipEventCreate(&p.b);
    if (e != hipSuccess) return e;
  }
  return hipEventRecord(p.a, st);
}
hipError_t vote_ev_end(mpe_handle* h, int slot, hipStream_t st, bool carried_a_scan) {
  if (h->vote_ev_calls <= 0) return hipSuccess;
  mpe_handle::VotePair& p = h->vote_ev[(size_t)(h->vote_ev_seq % h->vote_ev_calls) * mpe_handle::kMaxSub + slot];
  const hipError_t e = hipEventRecord(p.b, st);
  p.used = e == hipSuccess && carried_a_scan;
  return e;
}

// the marker-permutation table of sp in h->mtab (rebuilt only when the rig, the buffer or the stream changed)
int prep_marker_table(mpe_handle* h, const SolveParams& sp) {
  HIP_TRY(h, h->mtab.reserve(k2_table_bytes(sp.n_markers)));
  if (h->mtab_ptr == h->mtab.p && h->mtab_n == sp.n_markers && h->mtab_stream == h->stream &&
      std::memcmp(h->mtab_markers, sp.markers, sizeof(double) * 3 * (size_t)sp.n_markers) == 0)
    return MPE_OK;
  h->mtab_ptr = nullptr;
  HIP_TRY(h, launch_k2_prep(sp, static_cast<double*>(h->mtab.p), h->stream));
  std::memcpy(h->mtab_markers, sp.markers, sizeof(double) * 3 * (size_t)sp.n_markers);
  h->mtab_n = sp.n_markers;
  h->mtab_ptr = h->mtab.p;
  h->mtab_stream = h->stream;
  return MPE_OK;
}


int run_pipeline(mpe_handle* h, const uint8_t* d_frames, int n_frames, const FrameGeom& g, const DetectParams& dp,
                 const SolveParams* sp, mpe_detections* d_dets, uint32_t* d_hist, mpe_result* d_results,
                 uint32_t* d_corr, const StreamHint* hint = nullptr) {
  h->done_recorded = false;
  h->ms_accum_valid = false;
  const size_t frame_bytes = (size_t)g.rows * g.pitch;
  const mpe_handle::Prefetch pf = h->prefetch;  // what the previous submission scanned for this one (if anything)
  h->prefetch.valid = false;
  if (sp) {
    const int rc = prep_marker_table(h, *sp);
    if (rc) return rc;
  }
  int nsub, per;
  sub_batch_shape(h, n_frames, frame_bytes, sp != nullptr, sp ? sp->vote_arith : 1, nsub, per);
  h->have_ms = false;
  // a streaming submission may still have validate / refine kernels on the tail stream that read the detection and
  // histogram buffers this call is about to overwrite: the fused schedules order themselves region by region, every
  // other path waits for all of them here
  auto drain_tails = [&]() -> int {
    if (h->tail_sub_pending) {
      HIP_TRY(h, hipStreamWaitEvent(h->stream, h->tail_sub_done[h->tail_last], 0));
      h->tail_sub_pending = false;
    }
    return MPE_OK;
  };
  HIP_TRY(h, h->scratch.reserve(k1b_scratch_bytes(g, nsub <= 1 ? n_frames : per)));
  if (sp) HIP_TRY(h, h->mid.reserve(k3_mid_bytes(n_frames)));
  if (nsub <= 1) {
    { const int rc = drain_tails(); if (rc) return rc; }
    HIP_TRY(h, h->flags.reserve(flag_words(frame_bytes * n_frames) * 8));
    HIP_TRY(h, h->work.reserve((size_t)2 * (n_frames + 1) * sizeof(int)));
    int rc = run_front(h, h->stream, h->profiling, 0, n_frames, d_frames, n_frames, g, dp, sp,
                       static_cast<unsigned long long*>(h->flags.p), d_dets);
    if (rc) return rc;
    rc = run_back(h, h->stream, h->profiling, n_frames, sp, d_dets, d_hist, d_results, d_corr);
    h->have_ms = (rc == MPE_OK) && h->profiling;
    h->prof_pipelined = false;
    h->prof_launches = 1;
    h->prof_frames_per_launch = n_frames;
    return rc;
  }
  const bool prof = h->profiling;
  if (prof)
    for (int s = 0; s < nsub; ++s)
      for (int k = 0; k < 8; ++k)
        if (!h->pev[s][k]) HIP_TRY(h, hipEventCreate(&h->pev[s][k]));
  // Software pipeline over nsub sub-batches on two streams: A runs scan + blobs, B voting + tail.
  const size_t fw_per = flag_words(frame_bytes * per);
  // (one region per sub-batch + one for the first sub-batch of the NEXT submission, see StreamHint)
  HIP_TRY(h, h->flags.reserve(fw_per * (nsub + 1) * 8));
  HIP_TRY(h, h->work.reserve((size_t)2 * (per + 1) * nsub * sizeof(int)));
  h->work_ints = (size_t)2 * (per + 1) * nsub;
  h->blob_launches.clear();
  // schedule = option "pipeline_mode": -1 (default) = automatic = 6 (fused voting + scan, validate / refine on a side
  // stream, the scan split between a side k1a_scan and the rider); 3 / 4 = its one-stream / no-split-scan variants;
  // 0 = the older two-stream software pipeline.  Schedules with side streams verify once
  // per caller stream that those streams really execute concurrently (ensure_side_streams / pick_concurrent_streams)
  // and fall back to the one-stream schedule 3 when the runtime cannot give them separate hardware queues.
  int schedule = h->pipeline_mode;
  // automatic: the fused schedule for every marker count.  For more than 5 markers the voting kernel cannot carry
  // the scan (its LDS table would not fit) and launch_k2_vote falls back to the plain kernel + a stand-alone scan —
  // the voting then takes > 95 % of a sub-batch anyway (C(n_d,3) P(n_m,3) P3P solves), so nothing is lost.
  if (schedule < 0) schedule = 6;
  if (schedule == 0) {
    const int rc = pick_concurrent_streams(h);
    if (rc) return rc;
    if (h->streams_concurrent == 0) schedule = 6;
  }
  h->last_schedule = schedule;
  if (schedule == 4 || schedule == 6) {
    // the side streams of these schedules only pay when they really execute beside the caller's stream: verify it
    // once per (handle, caller stream) with the spin probe; without a concurrent triple -> schedule 3 (one stream)
    const int rc = ensure_side_streams(h, schedule == 6 && h->scan_split_pct > 0);
    if (rc) return rc;
    if (h->side_streams_ok == 0) schedule = 3;
    h->last_schedule = schedule;
  }
  if (schedule == 3 || schedule == 4 || schedule == 6) {
    const bool side_tail = schedule != 3;
    // mode 6: the HBM stream is spread over the whole sub-batch period.  In modes 3 / 4 the voting kernel scans all
    // of the next sub-batch and is HBM bound (0.95 ms for 5.9 GB) with 40 % of its issue slots idle, while the blob /
    // tail window before it (0.4 ms) moves no image bytes.  Here a stand-alone k1a_scan on a side stream takes
    // scan_split_pct % of sub-batch s + 2 from the end of vote(s) to the start of blobs(s + 2) — the blob / tail window of
    // sub-batch s + 1 and its voting launch — as side_scan_blocks resident blocks per CU (48-VGPR waves, no LDS), and the
    // rider of vote(s + 1) scans only the rest.
    const bool split_scan = schedule == 6 && h->scan_split_pct > 0;
    if (split_scan) {
      for (auto& e : h->scanpart_done)
        if (!e) HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
      if (!h->fork_ev) HIP_TRY(h, hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming));
      if (!h->prefetch_side_done) HIP_TRY(h, hipEventCreateWithFlags(&h->prefetch_side_done, hipEventDisableTiming));
    }
    auto split_bytes = [&](size_t nbytes) -> size_t {
      return split_scan ? (nbytes * (size_t)h->scan_split_pct / 100) / 8192 * 8192 : 0;
    };
    h->last_rider_bytes = 0;
    h->last_nsub = nsub;
    h->last_per = per;
    if (side_tail) {
      if (!h->tail_done) HIP_TRY(h, hipEventCreateWithFlags(&h->tail_done, hipEventDisableTiming));
      if (!h->vote_done[0])
        for (int i = 0; i < mpe_handle::kMaxSub; ++i)
          HIP_TRY(h, hipEventCreateWithFlags(&h->vote_done[i], hipEventDisableTiming));
      for (auto& e : h->tail_sub_done)
        if (!e) HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // Fused schedule, ONE stream: the voting kernel of sub-batch s carries the image scan of sub-batch
    // s + 1 on its idle memory pipeline (ScanRider in mpe_k2.hip).
    //   scan(0) | blobs(0) vote(0)+scan(1) tail(0) | blobs(1) vote(1)+scan(2) tail(1) | ...
    // Streaming (StreamHint): the LAST voting launch carries the scan of the first sub-batch of the NEXT submission
    // (into the extra flag region behind the nsub regions of this one), whose stand-alone scan then disappears:
    //   ... vote(n-1)+scan(next 0) tail(n-1) || blobs(next 0) vote(next 0)+scan(next 1) ...
    hipStream_t st = h->stream;
    unsigned long long* flags_base = static_cast<unsigned long long*>(h->flags.p);
    // was sub-batch 0 of THIS call scanned by the previous submission?
    const bool prefetched = pf.valid && pf.frames == d_frames && pf.per == std::min(per, n_frames) &&
                            pf.frame_bytes == frame_bytes && pf.thr == dp.thr && pf.flags_base == h->flags.p &&
                            pf.fw_per == fw_per && pf.flags_ptr != nullptr;
    // the next submission's first sub-batch, if the caller announced it and it will run pipelined as well
    int next_per = 0;
    if (hint && hint->next_frames && hint->n_next > 0) {
      int nn, np;
      sub_batch_shape(h, hint->n_next, frame_bytes, true, sp->vote_arith, nn, np);
      if (nn > 1 && flag_words(frame_bytes * std::min(np, hint->n_next)) <= fw_per) next_per = std::min(np, hint->n_next);
    }
    // region index of a sub-batch's flag words: 0 .. nsub-1, nsub = the extra region (prefetch target / source)
    auto sub_ptrs = [&](int s, int& f0, int& nf, const uint8_t*& fr, unsigned long long*& fl) {
      if (s >= nsub) {  // the virtual sub-batch behind the last one = the next submission's first
        f0 = 0;
        nf = next_per;
        fr = hint->next_frames;
        fl = flags_base + fw_per * nsub;
        return;
      }
      f0 = s * per;
      nf = std::min(per, n_frames - f0);
      fr = d_frames + (size_t)f0 * frame_bytes;
      fl = (s == 0 && prefetched) ? pf.flags_ptr : flags_base + fw_per * s;
    };
    // number of real sub-batches (the last ones may be empty when n_frames is not a multiple of `per`)
    int n_real = 0;
    while (n_real < nsub && n_real * per < n_frames) ++n_real;
    const bool tail_was_pending = h->tail_sub_pending;  // (this call records the same events anew)
    const int tail_was_last = h->tail_last;
    // region s of this call covers the same frames as region s of the previous submission only if both cut their
    // batches alike; otherwise every region waits for the previous submission's LAST tail (the tail stream executes
    // them in order)
    const bool tail_same_shape = h->tail_per == per;
    auto has_sub = [&](int s) { return s < n_real || (s == n_real && next_per > 0); };
    int f0, nf;
    const uint8_t* fr;
    unsigned long long* fl;
    sub_ptrs(0, f0, nf, fr, fl);
    if (prof) HIP_TRY(h, hipEventRecord(h->pev[0][0], st));
    // the work-lists of all sub-batches with one memset (instead of one per sub-batch in front of its blob kernels)
    HIP_TRY(h, hipMemsetAsync(h->work.p, 0, (size_t)2 * (per + 1) * nsub * sizeof(int), st));
    if (prefetched) {
      if (pf.side_part) HIP_TRY(h, hipStreamWaitEvent(st, h->prefetch_side_done, 0));
    } else {
      HIP_TRY(h, launch_k1a_scan(fr, (size_t)nf * frame_bytes, fl, dp.thr, scan_lds(h, false), st));
    }
    if (prof) HIP_TRY(h, hipEventRecord(h->pev[0][1], st));
    // side scan of the first part of sub-batch k (k >= 1; k == n_real: the next submission's first sub-batch), gated
    // so that it runs in the blob / tail window that follows vote(k - 2) (k = 1: at the start of the call)
    bool prefetch_side = false;
    auto side_scan = [&](int k) -> int {
      if (!split_scan || !has_sub(k)) return MPE_OK;
      int q0, qn;
      const uint8_t* qfr;
      unsigned long long* qfl;
      sub_ptrs(k >= n_real ? nsub : k, q0, qn, qfr, qfl);
      const size_t P = split_bytes((size_t)qn * frame_bytes);
      if (k == 1) {
        HIP_TRY(h, hipEventRecord(h->fork_ev, st));
        HIP_TRY(h, hipStreamWaitEvent(h->scan_stream, h->fork_ev, 0));
      } else {
        HIP_TRY(h, hipStreamWaitEvent(h->scan_stream, h->vote_done[k - 2], 0));
      }
      if (k >= n_real && hint && hint->next_ready)  // the announced frames may still be uploading
        HIP_TRY(h, hipStreamWaitEvent(h->scan_stream, hint->next_ready, 0));
      if (P) HIP_TRY(h, launch_k1a_scan(qfr, P, qfl, dp.thr, 0, h->scan_stream, h->side_scan_blocks));
      if (k >= n_real) {
        HIP_TRY(h, hipEventRecord(h->prefetch_side_done, h->scan_stream));
        prefetch_side = true;
      } else {
        HIP_TRY(h, hipEventRecord(h->scanpart_done[k], h->scan_stream));
      }
      return MPE_OK;
    };
    { const int rc = side_scan(1); if (rc) return rc; }
    int used = 0;
    for (int s = 0; s < n_real; ++s) {
      sub_ptrs(s, f0, nf, fr, fl);
      used = s + 1;
      if (split_scan && s >= 1) HIP_TRY(h, hipStreamWaitEvent(st, h->scanpart_done[s], 0));
      // streaming: the tail of the PREVIOUS submission has read the detections / histograms of this region
      if (tail_was_pending)
        HIP_TRY(h, hipStreamWaitEvent(st, h->tail_sub_done[tail_same_shape ? std::min(s, tail_was_last) : tail_was_last], 0));
      if (prof) HIP_TRY(h, hipEventRecord(h->pev[s][2], st));
      HIP_TRY(h, launch_k1b_blobs(fr, fl, nf, g, dp, d_dets + f0,
                                  static_cast<int*>(h->work.p) + (size_t)s * 2 * (per + 1),
                                  static_cast<uint8_t*>(h->scratch.p), h->scratch.cap, sp->n_markers, st, nullptr, true));
      h->blob_launches.emplace_back((size_t)s * 2 * (per + 1), nf);
      if (prof) HIP_TRY(h, hipEventRecord(h->pev[s][3], st));
      uint32_t* hs = d_hist + (size_t)f0 * MPE_HIST_STRIDE;
      // (with one voting block per frame the kernel stores every histogram row the tail reads: no memset)
      if (vote_arith_is_strict(sp->vote_arith) || auto_splits(h, nf, sp->n_markers) != 1)
        HIP_TRY(h, hipMemsetAsync(hs, 0, (size_t)nf * MPE_HIST_STRIDE * sizeof(uint32_t), st));
      const uint8_t* nfr = nullptr;
      unsigned long long* nfl = nullptr;
      size_t nbytes = 0, scanned = 0;
      if (has_sub(s + 1)) {
        int nf0, nnf;
        sub_ptrs(s + 1 >= n_real ? nsub : s + 1, nf0, nnf, nfr, nfl);
        nbytes = (size_t)nnf * frame_bytes;
      }
      const size_t P = split_bytes(nbytes);  // (the first P bytes of sub-batch s + 1 come from the side scan)
      if (nbytes && s + 1 >= n_real && hint && hint->next_ready)  // this launch reads the NEXT submission's frames
        HIP_TRY(h, hipStreamWaitEvent(st, hint->next_ready, 0));
      VoteFixup fx;
      { const int rc = vote_fixup_for(h, s, nsub, per, sp->n_markers, det_hint_for(h, sp->n_markers), st, fx); if (rc) return rc; }
      if (prof) HIP_TRY(h, hipEventRecord(h->pev[s][4], st));
      HIP_TRY(h, vote_ev_begin(h, s, st));
      HIP_TRY(h, launch_k2_vote(d_dets + f0, nf, *sp, static_cast<const double*>(h->mtab.p), hs,
                                auto_splits(h, nf, sp->n_markers), det_hint_for(h, sp->n_markers), st, nbytes ? nfr + P : nullptr,
                                nbytes - P, nbytes ? nfl + P / 1024 : nullptr, dp.thr, &scanned, nullptr, &fx));
      HIP_TRY(h, vote_ev_end(h, s, st, scanned > 0));
      if (prof) HIP_TRY(h, hipEventRecord(h->pev[s][5], st));
      if (scanned > h->last_rider_bytes) h->last_rider_bytes = scanned;
      if (nbytes > 0) {  // what the riders left over: less than one chunk, or everything if they could not run
        const bool real_next = s + 1 < n_real;
        if (prof && real_next) HIP_TRY(h, hipEventRecord(h->pev[s + 1][0], st));
        if (nbytes - P > scanned)
          HIP_TRY(h, launch_k1a_scan(nfr + P + scanned, nbytes - P - scanned, nfl + (P + scanned) / 1024, dp.thr,
                                     scan_lds(h, false), st));
        if (prof && real_next) HIP_TRY(h, hipEventRecord(h->pev[s + 1][1], st));
      }
      // validate + refine of this sub-batch: on the caller's stream, or (mode 4) on a side stream so that its
      // thin, latency-bound kernels run beside the blob extraction of the next sub-batch
      hipStream_t tst = st;
      if (side_tail) {
        HIP_TRY(h, hipEventRecord(h->vote_done[s], st));
        HIP_TRY(h, hipStreamWaitEvent(h->tail_stream, h->vote_done[s], 0));
        tst = h->tail_stream;
        const int rc = side_scan(s + 2);  // runs beside blobs(s + 1) / tail(s)
        if (rc) return rc;
      }
      if (prof) HIP_TRY(h, hipEventRecord(h->pev[s][6], tst));
      // the strict verdicts on what vote(s) left undecided: in front of the tail, off the caller's stream with it
      HIP_TRY(h, fixup_launch(h, s, d_dets + f0, nf, *sp, hs, fx, tst));
      HIP_TRY(h, launch_k3_tail(d_dets + f0, hs, nf, *sp, d_results + f0,
                                d_corr ? d_corr + (size_t)f0 * 2 * MPE_MAX_MARKERS : nullptr, nullptr, nullptr, 0.0,
                                static_cast<uint8_t*>(h->mid.p) + k3_mid_bytes(1) * (size_t)f0, tst));
      if (prof) HIP_TRY(h, hipEventRecord(h->pev[s][7], tst));
      if (side_tail) {
        HIP_TRY(h, hipEventRecord(h->tail_sub_done[s], tst));
        h->tail_last = s;
      }
    }
    h->tail_sub_pending = side_tail;
    h->tail_per = per;
    if (next_per > 0) {  // sub-batch 0 of the next submission has been scanned into the extra region
      h->prefetch.valid = true;
      h->prefetch.flags_ptr = flags_base + fw_per * nsub;
      h->prefetch.frames = hint->next_frames;
      h->prefetch.per = next_per;
      h->prefetch.frame_bytes = frame_bytes;
      h->prefetch.thr = dp.thr;
      h->prefetch.flags_base = h->flags.p;
      h->prefetch.fw_per = fw_per;
      h->prefetch.side_part = prefetch_side;
    }
    // completion: everything of this submission is done when its last tail is (side_tail: on the tail stream, which
    // executes the tails in order; else on the caller's stream)
    if (!h->batch_done[0])
      for (auto& e : h->batch_done) HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t done = h->batch_done[h->submit_seq & 1];
    HIP_TRY(h, hipEventRecord(done, side_tail ? h->tail_stream : st));
    h->done_recorded = true;
    if (!(hint && hint->no_join) && side_tail) {  // join: the call behaves like one operation on the caller's stream
      HIP_TRY(h, hipStreamWaitEvent(st, done, 0));
      h->tail_sub_pending = false;  // (the next call's kernels are ordered behind every tail of this one anyway)
    }
    // (every side scan was waited for by the blob extraction of its sub-batch; a prefetch side scan by the next call)
    ++h->vote_ev_seq;
    if (prof) {
      h->prof_launches = used;
      h->have_ms = true;
      h->prof_pipelined = true;
      h->prof_frames_per_launch = per;
    }
    return MPE_OK;
  }
  {
    int rc = drain_tails();
    if (rc) return rc;
    rc = pick_concurrent_streams(h);
    if (rc) return rc;
  }
  if (!h->fork_ev) HIP_TRY(h, hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming));
  hipStream_t sa = h->sub_stream[0], sb = h->sub_stream[1];
  HIP_TRY(h, hipEventRecord(h->fork_ev, h->stream));
  HIP_TRY(h, hipStreamWaitEvent(sa, h->fork_ev, 0));
  HIP_TRY(h, hipStreamWaitEvent(sb, h->fork_ev, 0));
  // Staggered schedule: scan(i+1) runs beside vote(i) (HBM-bound beside FP64-bound), blobs(i+1)
  // beside tail(i) (two latency-bound kernels): blobs(i+1) is held back until vote(i) has finished.
  if (!h->vote_done[0])
    for (int i = 0; i < mpe_handle::kMaxSub; ++i)
      HIP_TRY(h, hipEventCreateWithFlags(&h->vote_done[i], hipEventDisableTiming));
  for (int s = 0; s < nsub; ++s) {
    const int f0 = s * per;
    if (f0 >= n_frames) break;
    const int nf = std::min(per, n_frames - f0);
    if (!h->sub_done[s]) HIP_TRY(h, hipEventCreateWithFlags(&h->sub_done[s], hipEventDisableTiming));
    const uint8_t* fr = d_frames + (size_t)f0 * frame_bytes;
    unsigned long long* fl = static_cast<unsigned long long*>(h->flags.p) + fw_per * s;
    if (prof) HIP_TRY(h, hipEventRecord(h->pev[s][0], sa));
    HIP_TRY(h, launch_k1a_scan(fr, (size_t)nf * frame_bytes, fl, dp.thr, scan_lds(h, true), sa));
    if (prof) HIP_TRY(h, hipEventRecord(h->pev[s][1], sa));
    hipStream_t sblob = sa;
    if (s > 0) HIP_TRY(h, hipStreamWaitEvent(sblob, h->vote_done[s - 1], 0));
    if (prof) HIP_TRY(h, hipEventRecord(h->pev[s][2], sblob));
    HIP_TRY(h, launch_k1b_blobs(fr, fl, nf, g, dp, d_dets + f0,
                                static_cast<int*>(h->work.p) + (size_t)s * 2 * (per + 1),
                                static_cast<uint8_t*>(h->scratch.p), h->scratch.cap, sp->n_markers, sblob));
    if (prof) HIP_TRY(h, hipEventRecord(h->pev[s][3], sblob));
    HIP_TRY(h, hipEventRecord(h->sub_done[s], sblob));
    HIP_TRY(h, hipStreamWaitEvent(sb, h->sub_done[s], 0));
    uint32_t* hs = d_hist + (size_t)f0 * MPE_HIST_STRIDE;
    HIP_TRY(h, hipMemsetAsync(hs, 0, (size_t)nf * MPE_HIST_STRIDE * sizeof(uint32_t), sb));
    VoteFixup fx;
    { const int rc = vote_fixup_for(h, s, nsub, per, sp->n_markers, det_hint_for(h, sp->n_markers), sb, fx); if (rc) return rc; }
    if (prof) HIP_TRY(h, hipEventRecord(h->pev[s][4], sb));
    HIP_TRY(h, launch_k2_vote(d_dets + f0, nf, *sp, static_cast<const double*>(h->mtab.p), hs,
                              auto_splits(h, nf, sp->n_markers), det_hint_for(h, sp->n_markers), sb, nullptr, 0, nullptr, 0, nullptr,
                              nullptr, &fx));
    HIP_TRY(h, fixup_launch(h, s, d_dets + f0, nf, *sp, hs, fx, sb));
    if (prof) HIP_TRY(h, hipEventRecord(h->pev[s][5], sb));
    HIP_TRY(h, hipEventRecord(h->vote_done[s], sb));
    hipStream_t stail = sb;
    if (prof) HIP_TRY(h, hipEventRecord(h->pev[s][6], stail));
    HIP_TRY(h, launch_k3_tail(d_dets + f0, hs, nf, *sp, d_results + f0,
                              d_corr ? d_corr + (size_t)f0 * 2 * MPE_MAX_MARKERS : nullptr, nullptr, nullptr, 0.0,
                              static_cast<uint8_t*>(h->mid.p) + k3_mid_bytes(1) * (size_t)f0, stail));
    if (prof) HIP_TRY(h, hipEventRecord(h->pev[s][7], stail));
    if (prof) h->prof_launches = s + 1;
  }
  HIP_TRY(h, hipEventRecord(h->fork_ev, sb));  // B waited for every front half
  HIP_TRY(h, hipStreamWaitEvent(h->stream, h->fork_ev, 0));
  if (prof) {
    h->have_ms = true;
    h->prof_pipelined = true;
    h->prof_frames_per_launch = per;
  }
  return MPE_OK;
}

}  // namespace

namespace {
template <class Fn>
int run_shards(mpe_handle* const* handles, int n_dev, Fn fn) {
  if (!handles || n_dev < 1) return MPE_ERR_ARG;
  for (int d = 0; d < n_dev; ++d) {
    if (!handles[d]) return MPE_ERR_ARG;
    for (int e = 0; e < d; ++e)
      if (handles[e] == handles[d]) return fail(handles[d], MPE_ERR_ARG, "the same handle was passed for two shards");
  }
  std::vector<int> rc((size_t)n_dev, MPE_OK);
  if (n_dev == 1) {
    rc[0] = fn(0);
  } else {
    std::vector<std::thread> th;
    th.reserve((size_t)n_dev);
    for (int d = 0; d < n_dev; ++d) th.emplace_back([&rc, &fn, d]() { rc[(size_t)d] = fn(d); });
    for (auto& t : th) t.join();
  }
  for (int d = 0; d < n_dev; ++d)
    if (rc[(size_t)d] != MPE_OK) return rc[(size_t)d];  // message: mpe_last_error(handles[d])
  return MPE_OK;
}
}  // namespace

// =============================================================================================
extern "C" {

const char* mpe_version(void) { return "mpe-hip 0.1 (gfx950)"; }

int mpe_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void mpe_default_params(mpe_params* p) {  // monocular_pose_estimator/launch/demo.launch:12-22
  p->threshold_value = 140;
  p->gaussian_sigma = 0.6;
  p->min_blob_area = 10;
  p->max_blob_area = 200;
  p->max_width_height_distortion = 0.5;
  p->max_circular_distortion = 0.5;
  p->back_projection_pixel_tolerance = 5;
  p->nearest_neighbour_pixel_tolerance = 7;
  p->certainty_threshold = 0.75;
  p->valid_correspondence_threshold = 0.7;
  p->roi_border_thickness = 20;
  p->histogram_threshold = 0;
}

int mpe_create(mpe_handle** out, int device) {
  if (!out) return MPE_ERR_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return MPE_ERR_NO_DEVICE;
  mpe_handle* h = new mpe_handle();
  if (device < 0) {
    if (hipGetDevice(&h->device) != hipSuccess) {
      delete h;
      return MPE_ERR_NO_DEVICE;
    }
  } else {
    if (device >= n || hipSetDevice(device) != hipSuccess) {
      delete h;
      return MPE_ERR_NO_DEVICE;
    }
    h->device = device;
  }
  if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete h;
    return MPE_ERR_HIP;
  }
  h->stream = h->own_stream;
  if (const char* e = std::getenv("MPE_TRACK_FUSED")) h->track_fused = std::max(0, std::min(2, std::atoi(e)));  // (A/B runs of scripts that take no options)
  *out = h;
  return MPE_OK;
}

void mpe_destroy(mpe_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  if (h->tail_stream) (void)hipStreamSynchronize(h->tail_stream);  // (an un-collected streaming submission)
  if (h->scan_stream) (void)hipStreamSynchronize(h->scan_stream);
  h->frames.release();
  h->flags.release();
  h->dets.release();
  h->hist.release();
  h->results.release();
  h->corr.release();
  h->mtab.release();
  h->work.release();
  h->scratch.release();
  h->track.release();
  h->mid.release();
  if (h->fix_ctl_host) (void)hipHostFree(h->fix_ctl_host);
  if (h->track_clk) (void)hipHostFree(h->track_clk);
  h->fix.release();
  if (h->mailbox) (void)hipHostFree(h->mailbox);
  for (auto& e : h->ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : h->sub_done)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : h->vote_done)
    if (e) (void)hipEventDestroy(e);
  for (auto& row : h->pev)
    for (auto& e : row)
      if (e) (void)hipEventDestroy(e);
  if (h->fork_ev) (void)hipEventDestroy(h->fork_ev);
  if (h->tail_done) (void)hipEventDestroy(h->tail_done);
  for (auto& e : h->batch_done)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : h->tail_sub_done)
    if (e) (void)hipEventDestroy(e);
  if (h->prefetch_side_done) (void)hipEventDestroy(h->prefetch_side_done);
  for (auto& p : h->vote_ev) {
    if (p.a) (void)hipEventDestroy(p.a);
    if (p.b) (void)hipEventDestroy(p.b);
  }

  if (h->tail_stream) (void)hipStreamDestroy(h->tail_stream);
  if (h->scan_stream) (void)hipStreamDestroy(h->scan_stream);
  for (auto& e : h->scanpart_done)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : h->copy_done)
    if (e) (void)hipEventDestroy(e);
  if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
  for (auto& st : h->sub_stream)
    if (st) (void)hipStreamDestroy(st);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
}

const char* mpe_last_error(const mpe_handle* h) { return h ? h->err.c_str() : "null handle"; }

int mpe_set_stream(mpe_handle* h, void* hip_stream) {
  if (!h) return MPE_ERR_ARG;
  h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
  return MPE_OK;
}
void* mpe_get_stream(mpe_handle* h) { return h ? h->stream : nullptr; }

int mpe_synchronize(mpe_handle* h) {
  if (!h) return MPE_ERR_ARG;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPE_OK;
}

int mpe_set_profiling(mpe_handle* h, int enable) {
  if (!h) return MPE_ERR_ARG;
  h->profiling = enable != 0;
  if (h->profiling)
    for (auto& e : h->ev)
      if (!e) HIP_TRY(h, hipEventCreate(&e));
  return MPE_OK;
}

namespace {
int last_kernel_ms_of_call(mpe_handle* h, float ms[5]);
}
int mpe_last_kernel_ms(mpe_handle* h, float ms[5]) {
  if (!h || !ms) return MPE_ERR_ARG;
  if (h->ms_accum_valid) {  // a chunked host ingest: sums over its chunks
    for (int i = 0; i < 5; ++i) ms[i] = h->ms_accum[i];
    return MPE_OK;
  }
  return last_kernel_ms_of_call(h, ms);
}
namespace {
int last_kernel_ms_of_call(mpe_handle* h, float ms[5]) {
  if (!h->have_ms) return fail(h, MPE_ERR_ARG, "profiling not enabled for the last batch");
  if (!h->prof_pipelined) {
    HIP_TRY(h, hipEventSynchronize(h->ev[4]));
    for (int i = 0; i < 4; ++i) HIP_TRY(h, hipEventElapsedTime(&ms[i], h->ev[i], h->ev[i + 1]));
    HIP_TRY(h, hipEventElapsedTime(&ms[4], h->ev[0], h->ev[4]));
    return MPE_OK;
  }
  // pipelined call: average duration PER LAUNCH of each kernel over the sub-batches; ms[4] = their sum
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (int i = 0; i < 5; ++i) ms[i] = 0.f;
  for (int s = 0; s < h->prof_launches; ++s)
    for (int k = 0; k < 4; ++k) {
      float t = 0.f;
      HIP_TRY(h, hipEventElapsedTime(&t, h->pev[s][2 * k], h->pev[s][2 * k + 1]));
      ms[k] += t / (float)h->prof_launches;
    }
  ms[4] = ms[0] + ms[1] + ms[2] + ms[3];
  return MPE_OK;
}
}  // namespace

/* launches per kernel and frames per launch of the last profiled batch (1 / n_frames when not pipelined) */
int mpe_last_kernel_ms_sub(mpe_handle* h, int sub_batch, float ms[4]) {
  if (!h || !ms) return MPE_ERR_ARG;
  if (!h->have_ms || !h->prof_pipelined || sub_batch < 0 || sub_batch >= h->prof_launches)
    return fail(h, MPE_ERR_ARG, "no per-sub-batch timing for the last batch");
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (int k = 0; k < 4; ++k) HIP_TRY(h, hipEventElapsedTime(&ms[k], h->pev[sub_batch][2 * k], h->pev[sub_batch][2 * k + 1]));
  return MPE_OK;
}

int mpe_last_launch_shape(mpe_handle* h, int* launches, int* frames_per_launch) {
  if (!h || !h->have_ms) return MPE_ERR_ARG;
  if (launches) *launches = h->prof_launches;
  if (frames_per_launch) *frames_per_launch = h->prof_frames_per_launch;
  return MPE_OK;
}

// tuning knobs (not part of the reference surface; used by bench / tests)
int mpe_get_option(mpe_handle* h, const char* name, int* value) {
  if (!h || !name || !value) return MPE_ERR_ARG;
  const std::string n(name);
  if (n == "pipeline") *value = h->pipeline;
  else if (n == "pipeline_mode") *value = h->pipeline_mode;
  else if (n == "lds_budget") *value = h->lds_budget;
  else if (n == "vote_splits") *value = h->vote_splits;
  else if (n == "vote_arith") *value = h->vote_arith;
  else if (n == "force_rccl_gather") *value = h->force_rccl_gather;
  else if (n == "assume_side_streams") *value = h->assume_side_streams;
  else if (n == "refine_variant") *value = h->refine_variant;
  else if (n == "ingest_chunk") *value = h->ingest_chunk;
  else if (n == "scan_split_pct") *value = h->scan_split_pct;
  else if (n == "side_scan_blocks") *value = h->side_scan_blocks;
  else if (n == "last_rider_kib") *value = (int)(h->last_rider_bytes >> 10);
  else if (n == "k1a_dummy_lds") *value = h->k1a_dummy_lds;
  else if (n == "streams_concurrent") *value = h->streams_concurrent;
  else if (n == "last_schedule") *value = h->last_schedule;
  else if (n == "vote_list_cap") *value = (int)h->fix_cap_limit;
  else if (n == "detections_hint") *value = h->detections_hint;
  else if (n == "track_fused") *value = h->track_fused;
  else if (n.rfind("track_phase_cycles_", 0) == 0) {  // mean shader-clock cycles of phase i = 0 .. 3 of the fused tracked frame
    const int i = std::atoi(n.c_str() + 19);
    if (i < 0 || i > 3) return fail(h, MPE_ERR_ARG, "phase out of range");
    *value = h->track_clk_n ? (int)(h->track_clk_sum[i] / (unsigned long long)h->track_clk_n) : 0;
  }
  else if (n == "detections_seen") *value = h->det_seen;
  else if (n == "k1b_general_blocks") *value = k1b_get_general_blocks();
  else if (n == "vote_fixup_items" || n == "vote_fixup_overflow" || n == "vote_relost_frames" || n == "vote_wide_frames") {
    // hypotheses (roots, detections) the fast voting kernel handed to the strict arithmetic since the handle was made,
    // how many it could not hand over because a list was full, and how many frames were therefore voted again by the
    // strict loop nest (k2_vote_relost); saturating at INT_MAX
    HIP_TRY(h, hipSetDevice(h->device));
    unsigned long long v = 0;
    // ("vote_wide_frames": frames with more than MPE_FAST_VOTE_DETECTIONS detections, voted by that loop nest alone)
    const int rc = fix_counter_sum(h, n == "vote_fixup_items" ? 3 : n == "vote_relost_frames" ? 6 : n == "vote_wide_frames" ? 7 : 1, v);
    if (rc) return rc;
    *value = v > 0x7fffffffull ? 0x7fffffff : (int)v;
  }
  else if (n.rfind("vote_launch_ns_slot_", 0) == 0 || n.rfind("vote_gap_ns_slot_", 0) == 0) {
    // per position within a pipelined call (sub-batch slot): mean duration of the scan-carrying voting launch, and mean
    // time from the end of the previous voting launch on the same stream (the previous call's last one for slot 0) to
    // its start — the blob window in front of it
    const bool gap = n[5] == 'g';
    const int slot = std::atoi(n.c_str() + (gap ? 17 : 20));
    if (slot < 0 || slot >= mpe_handle::kMaxSub) return fail(h, MPE_ERR_ARG, "slot out of range");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    double sum_ms = 0;
    long long cnt = 0;
    const long long calls = std::min<long long>(h->vote_ev_seq, h->vote_ev_calls);
    for (long long c = 0; c < calls; ++c) {
      mpe_handle::VotePair& p = h->vote_ev[(size_t)c * mpe_handle::kMaxSub + slot];
      if (!p.used) continue;
      float ms = 0;
      if (!gap) {
        HIP_TRY(h, hipEventElapsedTime(&ms, p.a, p.b));
      } else {
        // the launch in front: slot - 1 of the same call, or the last used slot of the call before (ring order)
        mpe_handle::VotePair* q = nullptr;
        if (slot > 0) {
          q = &h->vote_ev[(size_t)c * mpe_handle::kMaxSub + slot - 1];
        } else if (h->vote_ev_seq <= h->vote_ev_calls ? c > 0 : true) {
          const long long pc = (c + h->vote_ev_calls - 1) % h->vote_ev_calls;
          if (!(h->vote_ev_seq > h->vote_ev_calls && c == h->vote_ev_seq % h->vote_ev_calls))  // (the oldest call of the ring)
            for (int k = mpe_handle::kMaxSub - 1; k >= 0 && !q; --k)
              if (h->vote_ev[(size_t)pc * mpe_handle::kMaxSub + k].used) q = &h->vote_ev[(size_t)pc * mpe_handle::kMaxSub + k];
        }
        if (!q || !q->used) continue;
        HIP_TRY(h, hipEventElapsedTime(&ms, q->b, p.a));
      }
      sum_ms += ms;
      ++cnt;
    }
    *value = cnt ? (int)(sum_ms * 1e6 / (double)cnt + 0.5) : 0;
  }
  else if (n == "vote_launch_ns_mean" || n == "vote_launches") {
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    double sum_ms = 0;
    long long cnt = 0;
    for (auto& p : h->vote_ev)
      if (p.used) {
        float ms = 0;
        HIP_TRY(h, hipEventSynchronize(p.b));
        HIP_TRY(h, hipEventElapsedTime(&ms, p.a, p.b));
        sum_ms += ms;
        ++cnt;
      }
    *value = n == "vote_launches" ? (int)cnt : (cnt ? (int)(sum_ms * 1e6 / (double)cnt + 0.5) : 0);
  }
  else if (n == "track_steps") *value = (int)h->track_steps;
  else if (n == "track_ns_pack") *value = (int)(h->track_ns[0] / std::max(1LL, h->track_steps));
  else if (n == "track_ns_enqueue") *value = (int)(h->track_ns[1] / std::max(1LL, h->track_steps));
  else if (n == "track_ns_wait") *value = (int)(h->track_ns[2] / std::max(1LL, h->track_steps));
  else if (n.rfind("overflow_", 0) == 0) {
    // statistics of the last large batch (synchronises): frames the first blob tier handed on, in all
    // ("overflow_frames") or by the capacity that was exceeded ("overflow_why_1" .. 6: bright segments, bands,
    // islands, pixel pool, bitmap pool, blobs kept); "overflow_general": frames that went on to the general tier
    if (h->last_nsub <= 0 || !h->work.p || h->blob_launches.empty() || h->work_ints == 0)
      return fail(h, MPE_ERR_ARG, "no pipelined batch has run");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    std::vector<int> w(h->work_ints);
    HIP_TRY(h, hipMemcpy(w.data(), h->work.p, w.size() * sizeof(int), hipMemcpyDeviceToHost));
    const int why = n.rfind("overflow_why_", 0) == 0 ? std::atoi(n.c_str() + 13) : 0;
    long long cnt = 0;
    for (const auto& bl : h->blob_launches) {  // (offset of the launch's two lists, its frame count)
      const int* la = w.data() + bl.first;
      const int* lb = la + (bl.second + 1);
      if (n == "overflow_general") {
        cnt += lb[0];
      } else if (why == 0) {
        cnt += la[0];
      } else {
        for (int k = 0; k < la[0] && k < bl.second; ++k) cnt += ((la[1 + k] >> 24) & 0xFF) == why;
      }
    }
    *value = (int)std::min<long long>(cnt, 0x7fffffff);
  }
  else return fail(h, MPE_ERR_ARG, "unknown option");
  return MPE_OK;
}

int mpe_set_option(mpe_handle* h, const char* name, int value) {
  if (!h || !name) return MPE_ERR_ARG;
  if (!std::strcmp(name, "lds_budget")) {
    if (value < 8 * 1024 || value > 160 * 1024) return fail(h, MPE_ERR_ARG, "lds_budget out of range");
    h->lds_budget = value;
    return MPE_OK;
  }
  if (!std::strcmp(name, "k1a_dummy_lds")) {
    h->k1a_dummy_lds = value;
    return MPE_OK;
  }
  if (!std::strcmp(name, "pipeline_mode")) {
    if (value != -1 && value != 0 && value != 3 && value != 4 && value != 6)
      return fail(h, MPE_ERR_ARG, "pipeline_mode must be -1 (automatic), 0, 3, 4 or 6");
    h->pipeline_mode = value;
    h->prefetch.valid = false;  // (words scanned ahead by another schedule are not picked up)
    return MPE_OK;
  }
  if (!std::strcmp(name, "pipeline")) {
    if (value < 1 || value > mpe_handle::kMaxSub) return fail(h, MPE_ERR_ARG, "pipeline out of range");
    h->pipeline = value;
    return MPE_OK;
  }
  if (!std::strcmp(name, "track_profile")) {  // 1: start / reset the host-side timers of mpe_track_step ("track_ns_*")
    h->track_profile = value != 0;
    h->track_ns[0] = h->track_ns[1] = h->track_ns[2] = 0;
    h->track_steps = 0;
    return MPE_OK;
  }
  if (!std::strcmp(name, "vote_events")) {  // N > 0: time the scan-carrying voting launches of the last N pipelined calls
    if (value < 0 || value > 4096) return fail(h, MPE_ERR_ARG, "vote_events out of range (0..4096)");
    for (auto& p : h->vote_ev) {
      if (p.a) (void)hipEventDestroy(p.a);
      if (p.b) (void)hipEventDestroy(p.b);
    }
    h->vote_ev.assign((size_t)value * mpe_handle::kMaxSub, mpe_handle::VotePair());
    h->vote_ev_calls = value;
    h->vote_ev_seq = 0;
    return MPE_OK;
  }
  if (!std::strcmp(name, "vote_splits")) {
    h->vote_splits = value;
    return MPE_OK;
  }
  if (!std::strcmp(name, "side_scan_blocks")) {
    if (value < 1 || value > 32) return fail(h, MPE_ERR_ARG, "side_scan_blocks out of range (1..32)");
    h->side_scan_blocks = value;
    return MPE_OK;
  }
  if (!std::strcmp(name, "scan_split_pct")) {
    if (value < 0 || value > 90) return fail(h, MPE_ERR_ARG, "scan_split_pct out of range (0..90)");
    h->scan_split_pct = value;
    return MPE_OK;
  }
  if (!std::strcmp(name, "ingest_chunk")) {
    if (value < 0) return fail(h, MPE_ERR_ARG, "ingest_chunk must be >= 0");
    h->ingest_chunk = value;
    return MPE_OK;
  }
  if (!std::strcmp(name, "refine_variant")) {
    if (value < 0 || value > 2) return fail(h, MPE_ERR_ARG, "refine_variant must be 0 (automatic), 1 (lane) or 2 (group)");
    h->refine_variant = value;
    return MPE_OK;
  }
  if (!std::strcmp(name, "assume_side_streams")) {
    h->assume_side_streams = value ? 1 : 0;
    h->side_streams_ok = -1;
    return MPE_OK;
  }
  if (!std::strcmp(name, "force_rccl_gather")) {
    h->force_rccl_gather = value ? 1 : 0;
    return MPE_OK;
  }
  if (!std::strcmp(name, "tail_priority") || !std::strcmp(name, "scan_priority")) {  // experiments: side-stream priority
    if (value < -1 || value > 2) return fail(h, MPE_ERR_ARG, "priority must be -1 (lowest), 0 (default), 1 (highest) or 2 (default level through the priority entry point)");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    const bool tail = name[0] == 't';
    (tail ? h->tail_priority : h->scan_priority) = value;
    hipStream_t& st = tail ? h->tail_stream : h->scan_stream;
    if (st) {  // recreated with the new priority by the next pipelined call (which probes the set again)
      (void)hipStreamDestroy(st);
      st = nullptr;
    }
    h->side_streams_ok = -1;
    h->tail_sub_pending = false;
    return MPE_OK;
  }
  if (!std::strcmp(name, "k1b_general_blocks")) {  // tuning, process-wide: waves of the general blob tier in flight
    if (value < 32 || value > 8192) return fail(h, MPE_ERR_ARG, "k1b_general_blocks must be in [32, 8192]");
    k1b_set_general_blocks(value);
    return MPE_OK;
  }
  if (!std::strcmp(name, "track_phase_clocks")) {  // 1: time the phases of k_track_frame (scan / blobs / validate / refine)
    HIP_TRY(h, hipSetDevice(h->device));
    if (value && !h->track_clk)
      HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&h->track_clk), 8 * sizeof(unsigned long long), hipHostMallocDefault));
    if (!value && h->track_clk) {
      HIP_TRY(h, hipStreamSynchronize(h->stream));
      (void)hipHostFree(h->track_clk);
      h->track_clk = nullptr;
    }
    for (auto& v : h->track_clk_sum) v = 0;
    h->track_clk_n = 0;
    return MPE_OK;
  }
  if (!std::strcmp(name, "track_fused")) {  // A/B: 0 = the tracked frame as the chain of four kernels (rounds 3 - 5)
    // (2, the default: the kernel also stores the record to the caller's pinned memory itself; 1: fused kernel + copy)
    h->track_fused = value < 0 ? 0 : (value > 2 ? 2 : value);
    return MPE_OK;
  }
  if (!std::strcmp(name, "detections_hint")) {  // detections per frame the caller expects (0 = automatic); see det_hint_for
    if (value < 0 || value > MPE_MAX_DETECTIONS) return fail(h, MPE_ERR_ARG, "detections_hint must be in [0, MPE_MAX_DETECTIONS]");
    h->detections_hint = value;
    return MPE_OK;
  }
  if (!std::strcmp(name, "vote_list_cap")) {  // tests: a list this small overflows and exercises k2_vote_relost
    if (value < 0) return fail(h, MPE_ERR_ARG, "vote_list_cap must be >= 0");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    if (h->fix.p) {  // keep the cumulative counters of the layout that goes
      unsigned long long v = 0;
      int rc = fix_counter_sum(h, 1, v);
      if (rc) return rc;
      h->fix_overflow_base = v;
      rc = fix_counter_sum(h, 3, v);
      if (rc) return rc;
      h->fix_items_base = v;
      rc = fix_counter_sum(h, 6, v);
      if (rc) return rc;
      h->fix_relost_base = v;
      rc = fix_counter_sum(h, 7, v);
      if (rc) return rc;
      h->fix_wide_base = v;
    }
    h->fix.release();
    h->fix_cap = 0;
    h->fix_slots = 0;
    for (auto& b : h->fix_pending) b = false;
    h->fix_cap_limit = (unsigned)value;
    return MPE_OK;
  }
  if (!std::strcmp(name, "vote_arith")) {
    if (value < 0 || value > 4)
      return fail(h, MPE_ERR_ARG, "vote_arith must be 0 (strict), 1 (fast + strict re-evaluation of suspects), 2 (fast alone), "
                                  "3 (as 1) or 4 (as 0) with the quartic's complex powers as libstdc++ / glibc evaluate them");
    h->vote_arith = value;
    return MPE_OK;
  }
  return fail(h, MPE_ERR_ARG, "unknown option");
}

int mpe_detect_batch(mpe_handle* h, const uint8_t* frames, int n_frames, int rows, int cols, size_t stride_bytes,
                     size_t frame_stride_bytes, int frames_on_device, const double K[9], const double* D, int nD,
                     const mpe_params* p, mpe_detections* dets) {
  if (!h || !frames || !p || !K || !dets || n_frames < 0) return fail(h, MPE_ERR_ARG, "bad argument");
  if (n_frames == 0) return MPE_OK;
  ENTER(h);
  FrameGeom g;
  if (make_geom(h, rows, cols, g)) return fail(h, MPE_ERR_UNSUPPORTED, "frame size unsupported");
  DetectParams dp;
  if (make_detect_params(p, K, D, nD, 0, 0, dp)) return fail(h, MPE_ERR_ARG, "gaussian_sigma must be in (0, 6]");
  const uint8_t* d_frames = nullptr;
  int rc = stage_frames(h, frames, n_frames, rows, cols, stride_bytes, frame_stride_bytes, frames_on_device, 0, 0, cols,
                        rows, g, &d_frames);
  if (rc) return rc;
  HIP_TRY(h, h->dets.reserve((size_t)n_frames * sizeof(mpe_detections)));
  rc = run_pipeline(h, d_frames, n_frames, g, dp, nullptr, static_cast<mpe_detections*>(h->dets.p), nullptr, nullptr,
                    nullptr);
  if (rc) return rc;
  HIP_TRY(h, hipMemcpyAsync(dets, h->dets.p, (size_t)n_frames * sizeof(mpe_detections), hipMemcpyDeviceToHost,
                            h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPE_OK;
}

int mpe_find_leds(mpe_handle* h, const uint8_t* img, int rows, int cols, size_t stride_bytes, int roi_x, int roi_y,
                  int roi_w, int roi_h, const mpe_params* p, const double K[9], const double* D, int nD,
                  double* undist_xy, float* dist_xy, int cap, int* n_out) {
  if (!h || !img || !p || !K) return fail(h, MPE_ERR_ARG, "bad argument");
  if (roi_x < 0 || roi_y < 0 || roi_w <= 0 || roi_h <= 0 || roi_x + roi_w > cols || roi_y + roi_h > rows)
    return fail(h, MPE_ERR_ARG, "ROI outside the image");  // cv::Mat::operator()(Rect) asserts the same
  ENTER(h);
  FrameGeom g;
  if (make_geom(h, roi_h, roi_w, g)) return fail(h, MPE_ERR_UNSUPPORTED, "frame size unsupported");
  DetectParams dp;
  if (make_detect_params(p, K, D, nD, roi_x, roi_y, dp)) return fail(h, MPE_ERR_ARG, "gaussian_sigma must be in (0, 6]");
  const uint8_t* d_frames = nullptr;
  int rc = stage_frames(h, img, 1, rows, cols, stride_bytes, (size_t)rows * stride_bytes, 0, roi_x, roi_y, roi_w, roi_h,
                        g, &d_frames);
  if (rc) return rc;
  HIP_TRY(h, h->dets.reserve(sizeof(mpe_detections)));
  rc = run_pipeline(h, d_frames, 1, g, dp, nullptr, static_cast<mpe_detections*>(h->dets.p), nullptr, nullptr, nullptr);
  if (rc) return rc;
  mpe_detections d;
  HIP_TRY(h, hipMemcpyAsync(&d, h->dets.p, sizeof(d), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (n_out) *n_out = d.n;
  if (d.status != 0) {
    h->err = "frame exceeded a device capacity";
    return d.status;
  }
  for (int i = 0; i < d.n && i < cap; ++i) {
    if (undist_xy) {
      undist_xy[2 * i] = d.undist_xy[2 * i];
      undist_xy[2 * i + 1] = d.undist_xy[2 * i + 1];
    }
    if (dist_xy) {
      dist_xy[2 * i] = d.dist_xy[2 * i];
      dist_xy[2 * i + 1] = d.dist_xy[2 * i + 1];
    }
  }
  return MPE_OK;
}

namespace {
int vote_batch_impl(mpe_handle* h, const double* det_xy, const int* n_det, int n_frames, const double* markers_xyz,
                    int n_markers, const double K[9], double back_projection_pixel_tolerance, const int* item_lo,
                    const int* item_hi, uint32_t* hist);
}
int mpe_vote_batch(mpe_handle* h, const double* det_xy, const int* n_det, int n_frames, const double* markers_xyz,
                   int n_markers, const double K[9], double back_projection_pixel_tolerance, uint32_t* hist) {
  return vote_batch_impl(h, det_xy, n_det, n_frames, markers_xyz, n_markers, K, back_projection_pixel_tolerance, nullptr,
                         nullptr, hist);
}
int mpe_vote_items(mpe_handle* h, const double* det_xy, const int* n_det, int n_frames, const double* markers_xyz,
                   int n_markers, const double K[9], double back_projection_pixel_tolerance, const int* item_lo,
                   const int* item_hi, uint32_t* hist) {
  if (!item_lo || !item_hi) return fail(h, MPE_ERR_ARG, "bad argument");
  return vote_batch_impl(h, det_xy, n_det, n_frames, markers_xyz, n_markers, K, back_projection_pixel_tolerance, item_lo,
                         item_hi, hist);
}
namespace {
int vote_batch_impl(mpe_handle* h, const double* det_xy, const int* n_det, int n_frames, const double* markers_xyz,
                    int n_markers, const double K[9], double back_projection_pixel_tolerance, const int* item_lo,
                    const int* item_hi, uint32_t* hist) {
  if (!h || !det_xy || !n_det || !markers_xyz || !K || !hist || n_frames < 0) return fail(h, MPE_ERR_ARG, "bad argument");
  if (n_frames == 0) return MPE_OK;
  ENTER(h);
  mpe_params p;
  mpe_default_params(&p);
  p.back_projection_pixel_tolerance = back_projection_pixel_tolerance;
  SolveParams sp;
  if (make_solve_params(h, &p, markers_xyz, n_markers, K, sp)) return fail(h, MPE_ERR_ARG, "too many markers");
  std::vector<mpe_detections> hd(n_frames);
  for (int f = 0; f < n_frames; ++f) {
    std::memset(&hd[f], 0, sizeof(mpe_detections));
    if (n_det[f] < 0 || n_det[f] > MPE_MAX_DETECTIONS) return fail(h, MPE_ERR_ARG, "n_det out of range");
    hd[f].n = n_det[f];
    std::memcpy(hd[f].undist_xy, det_xy + (size_t)f * 2 * MPE_MAX_DETECTIONS, sizeof(double) * 2 * n_det[f]);
  }
  HIP_TRY(h, h->dets.reserve((size_t)n_frames * sizeof(mpe_detections)));
  HIP_TRY(h, h->hist.reserve((size_t)n_frames * MPE_HIST_STRIDE * sizeof(uint32_t)));
  HIP_TRY(h, hipMemcpyAsync(h->dets.p, hd.data(), (size_t)n_frames * sizeof(mpe_detections), hipMemcpyHostToDevice,
                            h->stream));
  HIP_TRY(h, hipMemsetAsync(h->hist.p, 0, (size_t)n_frames * MPE_HIST_STRIDE * sizeof(uint32_t), h->stream));
  { const int rc = prep_marker_table(h, sp); if (rc) return rc; }
  const int* d_range = nullptr;
  if (item_lo) {  // forensics: per-frame hypothesis ranges, interleaved {lo, hi}
    std::vector<int> rg((size_t)2 * n_frames);
    for (int f = 0; f < n_frames; ++f) {
      rg[(size_t)2 * f] = item_lo[f];
      rg[(size_t)2 * f + 1] = item_hi[f];
    }
    HIP_TRY(h, h->work.reserve(rg.size() * sizeof(int)));
    HIP_TRY(h, hipMemcpyAsync(h->work.p, rg.data(), rg.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));  // (rg goes out of scope)
    d_range = static_cast<const int*>(h->work.p);
  }
  VoteFixup fx;
  {
    int nd_max = n_markers;
    for (int f = 0; f < n_frames; ++f) nd_max = std::max(nd_max, n_det[f]);
    const int rc = vote_fixup_for(h, 0, 1, n_frames, n_markers, nd_max, h->stream, fx);
    if (rc) return rc;
  }
  HIP_TRY(h, launch_k2_vote(static_cast<mpe_detections*>(h->dets.p), n_frames, sp, static_cast<const double*>(h->mtab.p),
                            static_cast<uint32_t*>(h->hist.p), auto_splits(h, n_frames, n_markers), n_markers,
                            h->stream, nullptr, 0, nullptr, 0, nullptr, d_range, &fx));
  HIP_TRY(h, fixup_launch(h, 0, static_cast<mpe_detections*>(h->dets.p), n_frames, sp, static_cast<uint32_t*>(h->hist.p), fx,
                          h->stream, d_range));
  HIP_TRY(h, hipMemcpy2DAsync(hist, MPE_HIST_WORDS * sizeof(uint32_t), h->hist.p, MPE_HIST_STRIDE * sizeof(uint32_t),
                              MPE_HIST_WORDS * sizeof(uint32_t), (size_t)n_frames,
                            hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPE_OK;
}
}  // namespace

namespace {
int solve_bruteforce_impl(mpe_handle* h, const double* det_xy, int n_det, const double* markers_xyz, int n_markers,
                          const double K[9], const mpe_params* p, mpe_result* out, uint32_t* hist, uint32_t* corr,
                          int tail_mode) {
  if (!h || (!det_xy && n_det > 0) || !markers_xyz || !K || !p || !out || n_det < 0)
    return fail(h, MPE_ERR_ARG, "bad argument");
  if (n_det > MPE_MAX_DETECTIONS) return fail(h, MPE_ERR_UNSUPPORTED, "n_det > MPE_MAX_DETECTIONS");
  ENTER(h);
  SolveParams sp;
  if (make_solve_params(h, p, markers_xyz, n_markers, K, sp)) return fail(h, MPE_ERR_UNSUPPORTED, "n_markers > MPE_MAX_MARKERS");
  mpe_detections hd;
  std::memset(&hd, 0, sizeof(hd));
  hd.n = n_det;
  if (n_det) std::memcpy(hd.undist_xy, det_xy, sizeof(double) * 2 * n_det);
  HIP_TRY(h, h->dets.reserve(sizeof(mpe_detections)));
  HIP_TRY(h, h->hist.reserve(MPE_HIST_STRIDE * sizeof(uint32_t)));
  HIP_TRY(h, h->results.reserve(sizeof(mpe_result)));
  HIP_TRY(h, h->corr.reserve(2 * MPE_MAX_MARKERS * sizeof(uint32_t)));
  HIP_TRY(h, h->mid.reserve(k3_mid_bytes(1)));
  HIP_TRY(h, hipMemcpyAsync(h->dets.p, &hd, sizeof(hd), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemsetAsync(h->hist.p, 0, MPE_HIST_STRIDE * sizeof(uint32_t), h->stream));
  { const int rc = prep_marker_table(h, sp); if (rc) return rc; }
  VoteFixup fx;
  { const int rc = vote_fixup_for(h, 0, 1, 1, n_markers, n_det, h->stream, fx); if (rc) return rc; }
  HIP_TRY(h, launch_k2_vote(static_cast<mpe_detections*>(h->dets.p), 1, sp, static_cast<const double*>(h->mtab.p),
                            static_cast<uint32_t*>(h->hist.p), auto_splits(h, 1, n_markers), n_det, h->stream, nullptr,
                            0, nullptr, 0, nullptr, nullptr, &fx));
  HIP_TRY(h, fixup_launch(h, 0, static_cast<mpe_detections*>(h->dets.p), 1, sp, static_cast<uint32_t*>(h->hist.p), fx,
                          h->stream));
  HIP_TRY(h, launch_k3_tail(static_cast<mpe_detections*>(h->dets.p), static_cast<uint32_t*>(h->hist.p), 1, sp,
                            static_cast<mpe_result*>(h->results.p), static_cast<uint32_t*>(h->corr.p), nullptr,
                            nullptr, 0.0, h->mid.p, h->stream, tail_mode));
  uint32_t hh[MPE_HIST_STRIDE], hc[2 * MPE_MAX_MARKERS];
  HIP_TRY(h, hipMemcpyAsync(out, h->results.p, sizeof(mpe_result), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(hh, h->hist.p, sizeof(hh), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(hc, h->corr.p, sizeof(hc), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (hist)
    for (int r = 0; r < n_det; ++r)
      for (int c = 0; c < n_markers; ++c) hist[r * n_markers + c] = hh[r * MPE_MAX_MARKERS + c];
  if (corr) std::memcpy(corr, hc, sizeof(uint32_t) * 2 * n_markers);
  return MPE_OK;
}
}  // namespace

int mpe_solve_bruteforce(mpe_handle* h, const double* det_xy, int n_det, const double* markers_xyz, int n_markers,
                         const double K[9], const mpe_params* p, mpe_result* out, uint32_t* hist, uint32_t* corr) {
  return solve_bruteforce_impl(h, det_xy, n_det, markers_xyz, n_markers, K, p, out, hist, corr, 0);
}

int mpe_initialise(mpe_handle* h, const double* det_xy, int n_det, const double* markers_xyz, int n_markers,
                   const double K[9], const mpe_params* p, mpe_result* out, uint32_t* hist, uint32_t* corr) {
  return solve_bruteforce_impl(h, det_xy, n_det, markers_xyz, n_markers, K, p, out, hist, corr, 1);
}

namespace {
// one frame through the tail kernel from explicit correspondences: mode 0 check + refine, 1 check only,
// 2 refine only from T_init
int run_tail_single(mpe_handle* h, const double* det_xy, int n_det, const double* markers_xyz, int n_markers,
                    const double K[9], const mpe_params* p, const uint32_t* corr, int n_corr, int mode,
                    const double* T_init, mpe_result* out) {
  if (!h || (!det_xy && n_det > 0) || !markers_xyz || !K || !p || !out || n_det < 0 || n_corr < 0 || (!corr && n_corr > 0))
    return fail(h, MPE_ERR_ARG, "bad argument");
  if (n_det > MPE_MAX_DETECTIONS || n_corr > MPE_MAX_MARKERS) return fail(h, MPE_ERR_UNSUPPORTED, "too many points");
  ENTER(h);
  SolveParams sp;
  if (make_solve_params(h, p, markers_xyz, n_markers, K, sp)) return fail(h, MPE_ERR_UNSUPPORTED, "n_markers > MPE_MAX_MARKERS");
  for (int i = 0; i < n_corr; ++i)
    if (corr[2 * i] < 1 || corr[2 * i] > (uint32_t)n_markers || corr[2 * i + 1] < 1 || corr[2 * i + 1] > (uint32_t)n_det)
      return fail(h, MPE_ERR_ARG, "correspondence index out of range");
  mpe_detections hd;
  std::memset(&hd, 0, sizeof(hd));
  hd.n = n_det;
  if (n_det) std::memcpy(hd.undist_xy, det_xy, sizeof(double) * 2 * n_det);
  uint32_t hc[2 * MPE_MAX_MARKERS];
  std::memset(hc, 0, sizeof(hc));
  if (n_corr) std::memcpy(hc, corr, sizeof(uint32_t) * 2 * n_corr);
  HIP_TRY(h, h->dets.reserve(sizeof(mpe_detections)));
  HIP_TRY(h, h->hist.reserve(MPE_HIST_STRIDE * sizeof(uint32_t)));
  HIP_TRY(h, h->results.reserve(sizeof(mpe_result)));
  HIP_TRY(h, h->corr.reserve(2 * MPE_MAX_MARKERS * sizeof(uint32_t)));
  HIP_TRY(h, h->mid.reserve(k3_mid_bytes(1)));
  HIP_TRY(h, hipMemcpyAsync(h->dets.p, &hd, sizeof(hd), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->corr.p, hc, sizeof(hc), hipMemcpyHostToDevice, h->stream));
  if (mode == 2) {
    mpe_result seed;
    std::memset(&seed, 0, sizeof(seed));
    std::memcpy(seed.T, T_init, sizeof(seed.T));
    HIP_TRY(h, hipMemcpyAsync(h->results.p, &seed, sizeof(seed), hipMemcpyHostToDevice, h->stream));
  }
  HIP_TRY(h, launch_k3_tail(static_cast<mpe_detections*>(h->dets.p), static_cast<uint32_t*>(h->hist.p), 1, sp,
                            static_cast<mpe_result*>(h->results.p), nullptr, static_cast<uint32_t*>(h->corr.p),
                            nullptr, 0.0, h->mid.p, h->stream, mode));
  HIP_TRY(h, hipMemcpyAsync(out, h->results.p, sizeof(mpe_result), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPE_OK;
}
}  // namespace

int mpe_check_and_refine(mpe_handle* h, const double* det_xy, int n_det, const double* markers_xyz, int n_markers,
                         const double K[9], const mpe_params* p, const uint32_t* corr, int n_corr, mpe_result* out) {
  return run_tail_single(h, det_xy, n_det, markers_xyz, n_markers, K, p, corr, n_corr, 0, nullptr, out);
}

int mpe_check_correspondences(mpe_handle* h, const double* det_xy, int n_det, const double* markers_xyz, int n_markers,
                              const double K[9], const mpe_params* p, const uint32_t* corr, int n_corr,
                              mpe_result* out) {
  return run_tail_single(h, det_xy, n_det, markers_xyz, n_markers, K, p, corr, n_corr, 1, nullptr, out);
}

int mpe_optimise_pose(mpe_handle* h, const double* det_xy, int n_det, const double* markers_xyz, int n_markers,
                      const double K[9], const mpe_params* p, const uint32_t* corr, int n_corr, const double T_init[16],
                      mpe_result* out) {
  if (!T_init) return fail(h, MPE_ERR_ARG, "bad argument");
  return run_tail_single(h, det_xy, n_det, markers_xyz, n_markers, K, p, corr, n_corr, 2, T_init, out);
}

int mpe_p3p_batch(mpe_handle* h, const double* feature_vectors, const double* world_points, int n, double* solutions,
                  int* status) {
  if (!h || n < 0 || (n > 0 && (!feature_vectors || !world_points || !solutions || !status)))
    return fail(h, MPE_ERR_ARG, "bad argument");
  if (n == 0) return MPE_OK;
  ENTER(h);
  const size_t in = (size_t)n * 9 * sizeof(double), out = (size_t)n * 48 * sizeof(double);
  HIP_TRY(h, h->scratch.reserve(2 * in + out + (size_t)n * sizeof(int) + 64));
  uint8_t* base = static_cast<uint8_t*>(h->scratch.p);
  double* d_fv = reinterpret_cast<double*>(base);
  double* d_wp = reinterpret_cast<double*>(base + in);
  double* d_sol = reinterpret_cast<double*>(base + 2 * in);
  int* d_st = reinterpret_cast<int*>(base + 2 * in + out);
  HIP_TRY(h, hipMemcpyAsync(d_fv, feature_vectors, in, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(d_wp, world_points, in, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(d_sol, solutions, out, hipMemcpyHostToDevice, h->stream));  // collinear: kept as passed
  HIP_TRY(h, launch_p3p_batch(d_fv, d_wp, n, d_sol, d_st, h->stream));
  HIP_TRY(h, hipMemcpyAsync(solutions, d_sol, out, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(status, d_st, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPE_OK;
}

int mpe_solve_quartic_batch(mpe_handle* h, const double* factors, int n, int variant, double* real_roots) {
  if (!h || n < 0 || (n > 0 && (!factors || !real_roots)) || (variant != 0 && variant != 1))
    return fail(h, MPE_ERR_ARG, "bad argument");
  if (n == 0) return MPE_OK;
  ENTER(h);
  const size_t in = (size_t)n * 5 * sizeof(double), out = (size_t)n * 4 * sizeof(double);
  HIP_TRY(h, h->scratch.reserve(in + out + 64));
  double* d_f = static_cast<double*>(h->scratch.p);
  double* d_r = d_f + (size_t)n * 5;
  HIP_TRY(h, hipMemcpyAsync(d_f, factors, in, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, launch_quartic_batch(d_f, n, variant, d_r, h->stream));
  HIP_TRY(h, hipMemcpyAsync(real_roots, d_r, out, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPE_OK;
}

// Device-side record of one tracking step; fetched with ONE copy.
namespace {
struct TrackRecord {
  mpe_detections det;
  uint32_t corr[2 * MPE_MAX_MARKERS];
  mpe_result res;
};
const size_t kTrackHeader = 2 * MPE_MAX_MARKERS * sizeof(double);  // predicted pixels in front of the ROI
}  // namespace

int mpe_track_step(mpe_handle* h, const uint8_t* img, int rows, int cols, size_t stride_bytes, int roi_x, int roi_y,
                   int roi_w, int roi_h, const mpe_params* p, const double K[9], const double* D, int nD,
                   const double* markers_xyz, int n_markers, const double* predicted_px, mpe_detections* dets_out,
                   uint32_t* corr_out, mpe_result* out) {
  if (!h || !img || !p || !K || !markers_xyz || !predicted_px || !dets_out || !corr_out || !out)
    return fail(h, MPE_ERR_ARG, "bad argument");
  if (roi_x < 0 || roi_y < 0 || roi_w <= 0 || roi_h <= 0 || roi_x + roi_w > cols || roi_y + roi_h > rows)
    return fail(h, MPE_ERR_ARG, "ROI outside the image");
  if (h->pending_track_n) return fail(h, MPE_ERR_ARG, "a submitted batch has not been collected yet (shared staging memory)");
  ENTER(h);
  using clk = std::chrono::steady_clock;
  const clk::time_point t_in = h->track_profile ? clk::now() : clk::time_point();
  clk::time_point t_packed, t_queued;
  FrameGeom g;
  if (make_geom(h, roi_h, roi_w, g)) return fail(h, MPE_ERR_UNSUPPORTED, "frame size unsupported");
  DetectParams dp;
  if (make_detect_params(p, K, D, nD, roi_x, roi_y, dp)) return fail(h, MPE_ERR_ARG, "gaussian_sigma must be in (0, 6]");
  SolveParams sp;
  if (make_solve_params(h, p, markers_xyz, n_markers, K, sp)) return fail(h, MPE_ERR_UNSUPPORTED, "n_markers > MPE_MAX_MARKERS");
  const size_t roi_bytes = (size_t)g.rows * g.pitch;
  const size_t in_bytes = kTrackHeader + roi_bytes;
  const size_t need = in_bytes + sizeof(TrackRecord);
  if (need > h->mailbox_cap) {
    if (h->mailbox) (void)hipHostFree(h->mailbox);
    h->mailbox = nullptr;
    h->mailbox_cap = 0;
    const size_t want = std::max(need + need / 4, (size_t)1 << 16);
    HIP_TRY(h, hipHostMalloc(&h->mailbox, want, hipHostMallocDefault));
    h->mailbox_cap = want;

  }
  // pack [predicted pixels | ROI rows, zero padded to the pitch] into pinned memory -> one H2D copy
  uint8_t* mb = static_cast<uint8_t*>(h->mailbox);
  double* pred = reinterpret_cast<double*>(mb);
  for (int i = 0; i < 2 * MPE_MAX_MARKERS; ++i) pred[i] = i < 2 * n_markers ? predicted_px[i] : 0.0;
  for (int y = 0; y < roi_h; ++y) {
    uint8_t* dst = mb + kTrackHeader + (size_t)y * g.pitch;
    std::memcpy(dst, img + (size_t)(roi_y + y) * stride_bytes + roi_x, (size_t)roi_w);
    if (g.pitch > roi_w) std::memset(dst + roi_w, 0, (size_t)(g.pitch - roi_w));
  }
  TrackRecord* host_rec = reinterpret_cast<TrackRecord*>(mb + ((h->mailbox_cap - sizeof(TrackRecord)) & ~(size_t)63));
  HIP_TRY(h, h->frames.reserve(in_bytes + 16));
  HIP_TRY(h, h->flags.reserve(std::max(flag_words(roi_bytes), track_flag_words(g)) * 8));
  HIP_TRY(h, h->work.reserve(4 * sizeof(int)));
  HIP_TRY(h, h->scratch.reserve(k1b_scratch_bytes(g, 1)));
  HIP_TRY(h, h->hist.reserve(MPE_HIST_STRIDE * sizeof(uint32_t)));
  HIP_TRY(h, h->track.reserve(sizeof(TrackRecord)));
  HIP_TRY(h, h->mid.reserve(k3_mid_bytes(1)));
  // (Zero-copy I/O — the kernels reading the pinned mailbox over PCIe, a copy kernel writing the record back — was
  //  built and measured in round 3: the image scan then waits for PCIe reads (4 -> 46 us for 64 streams) and the step
  //  is no faster, 0.135 vs 0.136 ms for one stream.  The two copy commands stay.)
  uint8_t* d_in = static_cast<uint8_t*>(h->frames.p);
  TrackRecord* d_rec = static_cast<TrackRecord*>(h->track.p);
  h->have_ms = false;
  if (h->track_profile) t_packed = clk::now();
  HIP_TRY(h, hipMemcpyAsync(d_in, mb, in_bytes, hipMemcpyHostToDevice, h->stream));
  // the small blob tier alone first (a tracked ROI holds a handful of LEDs): three launches and a memset less per
  // frame; a frame that overflows it comes back with MPE_FRAME_TOO_MANY_ROWS and is repeated through the whole chain
  const bool optimistic = sp.n_markers >= 1 && sp.n_markers <= 8;
  // round 6: that optimistic pass is ONE launch — scan, blob extraction, correspondences + validation, refinement as
  // one kernel of one wave (k_track_frame): the three launch boundaries of the chain are gone (option "track_fused")
  const bool fused = optimistic && h->track_fused;
  if (!fused)
    HIP_TRY(h, launch_k1a_scan(d_in + kTrackHeader, roi_bytes, static_cast<unsigned long long*>(h->flags.p), dp.thr, 0,
                               h->stream));
  for (int pass = optimistic ? 0 : 1; pass < 2; ++pass) {
    if (pass == 0 && fused) {
      const bool deliver = h->track_fused >= 2;  // the kernel stores the record to the pinned mailbox itself
      TrackFramesArgs ta = {d_in + kTrackHeader, roi_bytes, reinterpret_cast<const double*>(d_in), nullptr,
                            static_cast<unsigned long long*>(h->flags.p), static_cast<uint32_t*>(h->hist.p), h->mid.p,
                            &d_rec->det, d_rec->corr, &d_rec->res, deliver ? &host_rec->det : nullptr,
                            deliver ? host_rec->corr : nullptr, deliver ? &host_rec->res : nullptr, h->track_clk};
      HIP_TRY(h, launch_track_frames(ta, 1, g, dp, sp, p->nearest_neighbour_pixel_tolerance, h->stream));
      if (h->track_fused < 2)
        HIP_TRY(h, hipMemcpyAsync(host_rec, d_rec, sizeof(TrackRecord), hipMemcpyDeviceToHost, h->stream));
      if (h->track_profile) t_queued = clk::now();
      HIP_TRY(h, hipStreamSynchronize(h->stream));
      if (h->track_clk && host_rec->det.status != MPE_FRAME_TOO_MANY_ROWS) {
        for (int i = 0; i < 4; ++i) h->track_clk_sum[i] += h->track_clk[i + 1] - h->track_clk[i];
        ++h->track_clk_n;
      }
      if (host_rec->det.status != MPE_FRAME_TOO_MANY_ROWS) break;
      continue;  // (rare: the whole chain, its own scan included — the fused kernel wrote the same flag words)
    }
    HIP_TRY(h, launch_k1b_blobs(d_in + kTrackHeader, static_cast<unsigned long long*>(h->flags.p), 1, g, dp, &d_rec->det,
                                static_cast<int*>(h->work.p), static_cast<uint8_t*>(h->scratch.p), h->scratch.cap, sp.n_markers, h->stream,
                                nullptr, false, pass == 0));
    HIP_TRY(h, launch_k3_tail(&d_rec->det, static_cast<uint32_t*>(h->hist.p), 1, sp, &d_rec->res, d_rec->corr, nullptr,
                              reinterpret_cast<const double*>(d_in), p->nearest_neighbour_pixel_tolerance, h->mid.p,
                              h->stream));
    HIP_TRY(h, hipMemcpyAsync(host_rec, d_rec, sizeof(TrackRecord), hipMemcpyDeviceToHost, h->stream));
    if (h->track_profile) t_queued = clk::now();
    // (polling hipStreamQuery instead of blocking in the runtime's wait measured 133-135 against 128-129 us per frame)
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (host_rec->det.status != MPE_FRAME_TOO_MANY_ROWS) break;
  }
  if (h->track_profile) {
    const clk::time_point t_done = clk::now();
    auto ns = [](clk::time_point a, clk::time_point b) { return (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count(); };
    h->track_ns[0] += ns(t_in, t_packed);
    h->track_ns[1] += ns(t_packed, t_queued);
    h->track_ns[2] += ns(t_queued, t_done);
    ++h->track_steps;
  }
  *dets_out = host_rec->det;
  std::memcpy(corr_out, host_rec->corr, sizeof(host_rec->corr));
  *out = host_rec->res;
  return MPE_OK;
}

namespace {
int estimate_device_impl(mpe_handle* h, const uint8_t* d_frames, int n_frames, int rows, int cols,
                         const double* markers_xyz, int n_markers, const double K[9], const double* D, int nD,
                         const mpe_params* p, mpe_result* d_results, const StreamHint* hint) {
  if (!h || !d_frames || !markers_xyz || !K || !p || !d_results || n_frames < 0)
    return fail(h, MPE_ERR_ARG, "bad argument");
  if ((cols & 15) || (reinterpret_cast<uintptr_t>(d_frames) & 15))
    return fail(h, MPE_ERR_UNSUPPORTED, "device frames must be packed, 16-byte aligned, cols % 16 == 0");
  HIP_TRY(h, hipSetDevice(h->device));
  FrameGeom g;
  if (make_geom(h, rows, cols, g)) return fail(h, MPE_ERR_UNSUPPORTED, "frame size unsupported");
  DetectParams dp;
  if (make_detect_params(p, K, D, nD, 0, 0, dp)) return fail(h, MPE_ERR_ARG, "gaussian_sigma must be in (0, 6]");
  SolveParams sp;
  if (make_solve_params(h, p, markers_xyz, n_markers, K, sp)) return fail(h, MPE_ERR_UNSUPPORTED, "n_markers > MPE_MAX_MARKERS");
  HIP_TRY(h, h->dets.reserve((size_t)n_frames * sizeof(mpe_detections)));
  HIP_TRY(h, h->hist.reserve((size_t)n_frames * MPE_HIST_STRIDE * sizeof(uint32_t)));
  const int rc = run_pipeline(h, d_frames, n_frames, g, dp, &sp, static_cast<mpe_detections*>(h->dets.p),
                              static_cast<uint32_t*>(h->hist.p), d_results, nullptr, hint);
  if (rc) return rc;
  if (!h->done_recorded) {  // schedules without side streams: the records are complete in the caller's stream order
    if (!h->batch_done[0])
      for (auto& e : h->batch_done) HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_TRY(h, hipEventRecord(h->batch_done[h->submit_seq & 1], h->stream));
  }
  return MPE_OK;
}
}  // namespace

int mpe_estimate_batch_device(mpe_handle* h, const uint8_t* d_frames, int n_frames, int rows, int cols,
                              const double* markers_xyz, int n_markers, const double K[9], const double* D, int nD,
                              const mpe_params* p, mpe_result* d_results) {
  if (h && h->submit_seq != h->collect_seq)
    return fail(h, MPE_ERR_ARG, "a submitted batch has not been collected yet (mpe_estimate_batch_device_collect)");
  if (n_frames == 0 && h && d_frames && d_results) return MPE_OK;
  const int rc = estimate_device_impl(h, d_frames, n_frames, rows, cols, markers_xyz, n_markers, K, D, nD, p, d_results,
                                      nullptr);  // (no hint: the side streams are joined back, one operation on the stream)
  return rc;
}

int mpe_estimate_batch_device_submit(mpe_handle* h, const uint8_t* d_frames, int n_frames, int rows, int cols,
                                     const double* markers_xyz, int n_markers, const double K[9], const double* D, int nD,
                                     const mpe_params* p, mpe_result* d_results, const uint8_t* d_next_frames,
                                     int n_next_frames) {
  if (!h || n_frames <= 0) return fail(h, MPE_ERR_ARG, "bad argument");
  if (h->submit_seq - h->collect_seq >= 2)
    return fail(h, MPE_ERR_ARG, "two submissions are in flight already: collect the older one first");
  StreamHint hint;
  hint.next_frames = d_next_frames;
  hint.n_next = d_next_frames ? n_next_frames : 0;
  hint.no_join = true;
  hint.next_ready = hint.next_frames ? h->next_ready : nullptr;
  h->next_ready = nullptr;  // one-shot
  const int rc = estimate_device_impl(h, d_frames, n_frames, rows, cols, markers_xyz, n_markers, K, D, nD, p, d_results,
                                      &hint);
  if (rc) return rc;
  ++h->submit_seq;
  return MPE_OK;
}

int mpe_stream_next_ready(mpe_handle* h, void* hip_event) {
  if (!h) return MPE_ERR_ARG;
  h->next_ready = static_cast<hipEvent_t>(hip_event);
  return MPE_OK;
}

int mpe_stream_drop_prefetch(mpe_handle* h) {
  if (!h) return MPE_ERR_ARG;
  h->prefetch.valid = false;  // the next _submit scans its first sub-batch itself
  return MPE_OK;
}

int mpe_estimate_batch_device_collect(mpe_handle* h, void* hip_stream) {
  if (!h) return MPE_ERR_ARG;
  if (h->submit_seq == h->collect_seq) return fail(h, MPE_ERR_ARG, "nothing has been submitted");
  HIP_TRY(h, hipSetDevice(h->device));
  hipStream_t consumer = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->stream;
  HIP_TRY(h, hipStreamWaitEvent(consumer, h->batch_done[h->collect_seq & 1], 0));
  ++h->collect_seq;
  return MPE_OK;
}

int mpe_estimate_batch(mpe_handle* h, const uint8_t* frames, int n_frames, int rows, int cols, size_t stride_bytes,
                       size_t frame_stride_bytes, int frames_on_device, const double* markers_xyz, int n_markers,
                       const double K[9], const double* D, int nD, const mpe_params* p, mpe_result* results) {
  if (!h || !frames || !markers_xyz || !K || !p || !results || n_frames < 0) return fail(h, MPE_ERR_ARG, "bad argument");
  if (n_frames == 0) return MPE_OK;
  ENTER(h);
  FrameGeom g;
  if (make_geom(h, rows, cols, g)) return fail(h, MPE_ERR_UNSUPPORTED, "frame size unsupported");
  DetectParams dp;
  if (make_detect_params(p, K, D, nD, 0, 0, dp)) return fail(h, MPE_ERR_ARG, "gaussian_sigma must be in (0, 6]");
  SolveParams sp;
  if (make_solve_params(h, p, markers_xyz, n_markers, K, sp)) return fail(h, MPE_ERR_UNSUPPORTED, "n_markers > MPE_MAX_MARKERS");
  HIP_TRY(h, h->dets.reserve((size_t)n_frames * sizeof(mpe_detections)));
  HIP_TRY(h, h->hist.reserve((size_t)n_frames * MPE_HIST_STRIDE * sizeof(uint32_t)));
  HIP_TRY(h, h->results.reserve((size_t)n_frames * sizeof(mpe_result)));
  mpe_detections* d_dets = static_cast<mpe_detections*>(h->dets.p);
  uint32_t* d_hist = static_cast<uint32_t*>(h->hist.p);
  mpe_result* d_res = static_cast<mpe_result*>(h->results.p);
  const bool host_packed = !frames_on_device && stride_bytes == (size_t)cols && frame_stride_bytes == (size_t)rows * cols &&
                           g.pitch == cols;
  if (host_packed && h->ingest_chunk > 0 && n_frames > h->ingest_chunk) {
    // Double-buffered ingest of HOST frames (sensor_msgs/Image payloads as monocular_pose_estimator.cpp:147 hands
    // them over): the batch is cut into chunks; the H2D copy of chunk c + 1 runs on a copy stream beside the kernels
    // of chunk c, each chunk in its own part of the device frame buffer.  With pinned host memory (hipHostMalloc /
    // hipHostRegister, or mpe_alloc_pinned) the copies are asynchronous DMA and the call is PCIe bound with the
    // compute hidden; with pageable memory the runtime stages the copy itself (same rate measured, host blocked).
    const size_t frame_bytes = (size_t)rows * cols;
    HIP_TRY(h, h->frames.reserve(frame_bytes * n_frames + 16));
    uint8_t* d_all = static_cast<uint8_t*>(h->frames.p);
    if (!h->copy_stream) HIP_TRY(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    for (auto& e : h->copy_done)
      if (!e) HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    // the copy stream starts after everything already queued on the caller's stream (the frame buffer may be in use)
    HIP_TRY(h, hipEventRecord(h->copy_done[0], h->stream));
    HIP_TRY(h, hipStreamWaitEvent(h->copy_stream, h->copy_done[0], 0));
    const int chunk = h->ingest_chunk;
    int ci = 0;
    float acc[5] = {0, 0, 0, 0, 0};
    // every early return below first waits for the copy stream: no DMA may still be reading the caller's (pinned)
    // buffer when the call hands it back
    auto bail = [&](int rc) -> int {
      (void)hipStreamSynchronize(h->copy_stream);
      (void)hipStreamSynchronize(h->stream);
      return rc;
    };
#define INGEST_TRY(call)                                                   \
  do {                                                                     \
    const hipError_t e__ = (call);                                         \
    if (e__ != hipSuccess) return bail(fail(h, MPE_ERR_HIP, #call, e__)); \
  } while (0)
    for (int f0 = 0; f0 < n_frames; f0 += chunk, ++ci) {
      const int nf = std::min(chunk, n_frames - f0);
      INGEST_TRY(hipMemcpyAsync(d_all + (size_t)f0 * frame_bytes, frames + (size_t)f0 * frame_bytes, frame_bytes * nf,
                                hipMemcpyHostToDevice, h->copy_stream));
      INGEST_TRY(hipEventRecord(h->copy_done[ci & 1], h->copy_stream));
      INGEST_TRY(hipStreamWaitEvent(h->stream, h->copy_done[ci & 1], 0));
      const int rc = run_pipeline(h, d_all + (size_t)f0 * frame_bytes, nf, g, dp, &sp, d_dets + f0,
                                  d_hist + (size_t)f0 * MPE_HIST_STRIDE, d_res + f0, nullptr);
      if (rc) return bail(rc);
      if (h->profiling && h->have_ms) {  // (profiling: the chunks' kernel times add up; this synchronises per chunk)
        float ms[5];
        const int rm = last_kernel_ms_of_call(h, ms);
        if (rm) return bail(rm);
        for (int i = 0; i < 5; ++i) acc[i] += ms[i];
      }
    }
#undef INGEST_TRY
    if (h->profiling && h->have_ms) {
      for (int i = 0; i < 5; ++i) h->ms_accum[i] = acc[i];
      h->ms_accum_valid = true;
    }
  } else {
    const uint8_t* d_frames = nullptr;
    int rc = stage_frames(h, frames, n_frames, rows, cols, stride_bytes, frame_stride_bytes, frames_on_device, 0, 0, cols,
                          rows, g, &d_frames);
    if (rc) return rc;
    rc = run_pipeline(h, d_frames, n_frames, g, dp, &sp, d_dets, d_hist, d_res, nullptr);
    if (rc) return rc;
  }
  HIP_TRY(h, hipMemcpyAsync(results, h->results.p, (size_t)n_frames * sizeof(mpe_result), hipMemcpyDeviceToHost,
                            h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  {  // what the next call can expect (det_hint_for): the detections of a typical frame of this one (the 90th percentile
     // of a sample: a few cluttered frames among clean ones do not switch the voting kernel)
    const int step = std::max(1, n_frames / 1024);
    std::vector<int> nd;
    for (int i = 0; i < n_frames; i += step) nd.push_back(results[i].n_det);
    std::nth_element(nd.begin(), nd.begin() + (nd.size() * 9) / 10, nd.end());
    h->det_seen = nd[(nd.size() * 9) / 10];
  }
  return MPE_OK;
}

int mpe_convert_to_mono8(mpe_handle* h, const void* src, int src_on_device, int encoding, int src_big_endian, int n_frames,
                         int rows, int cols, size_t src_stride_bytes, size_t src_frame_stride_bytes, uint8_t* dst,
                         int dst_on_device) {
  if (!h || !src || !dst || n_frames < 0 || rows <= 0 || cols <= 0) return fail(h, MPE_ERR_ARG, "bad argument");
  if (encoding < MPE_ENC_MONO8 || encoding > MPE_ENC_MONO16)
    return fail(h, MPE_ERR_UNSUPPORTED, "encoding not supported (mono8, bgr8, rgb8, bgra8, rgba8, mono16)");
  const size_t bpp = encoding == MPE_ENC_MONO8 ? 1 : (encoding == MPE_ENC_MONO16 ? 2 : ((encoding == MPE_ENC_BGRA8 || encoding == MPE_ENC_RGBA8) ? 4 : 3));
  if (src_stride_bytes < bpp * (size_t)cols || src_frame_stride_bytes < src_stride_bytes * (size_t)rows)
    return fail(h, MPE_ERR_ARG, "source strides smaller than the image");
  if (n_frames == 0) return MPE_OK;
  ENTER(h);
  const size_t in_bytes = src_frame_stride_bytes * (size_t)(n_frames - 1) + src_stride_bytes * (size_t)(rows - 1) + bpp * (size_t)cols;
  const size_t out_bytes = (size_t)n_frames * rows * cols;
  const uint8_t* d_src = static_cast<const uint8_t*>(src);
  uint8_t* d_dst = dst;
  // staging: the source behind the destination in the handle's frame buffer (only what is not on the device already)
  const size_t out_off = 0, in_off = dst_on_device ? 0 : ((out_bytes + 255) & ~(size_t)255);
  if (!src_on_device || !dst_on_device) HIP_TRY(h, h->frames.reserve(in_off + (src_on_device ? 0 : in_bytes) + 16));
  if (!dst_on_device) d_dst = static_cast<uint8_t*>(h->frames.p) + out_off;
  if (!src_on_device) {
    uint8_t* stage = static_cast<uint8_t*>(h->frames.p) + in_off;
    HIP_TRY(h, hipMemcpyAsync(stage, src, in_bytes, hipMemcpyHostToDevice, h->stream));
    d_src = stage;
  }
  HIP_TRY(h, launch_to_mono8(d_src, src_stride_bytes, src_frame_stride_bytes, encoding, src_big_endian ? 1 : 0, n_frames,
                             rows, cols, d_dst, h->stream));
  if (!dst_on_device) {
    HIP_TRY(h, hipMemcpyAsync(dst, d_dst, out_bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  } else if (!src_on_device) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));  // (the caller's host buffer has been read)
  }
  return MPE_OK;
}

void* mpe_alloc_pinned(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}
void mpe_free_pinned(void* p) {
  if (p) (void)hipHostFree(p);
}

// ---- lock-step batches: frame k of N independent camera streams in ONE device submission ---------------
// (BASELINE configs[4]: N streams' steps are independent of each other, pose_estimator.cpp:98-147 is sequential only
// within a stream.)  Every stream's ROI is cloned into one slot of a uniform slot array — zero beyond the ROI, the
// window size and origin in a per-slot table that the blob kernels read, so borders and centroid offsets are those
// of the stand-alone cv::Mat clone of led_detector.cpp:44 — then ONE k1a_scan + ONE blob extraction over the N
// slots and ONE validate / refine over the N detection sets (nearest-neighbour correspondences from the stream's
// predicted pixels) run, and one copy brings the N records back.
int mpe_track_step_batch_submit(mpe_handle* h, const mpe_track_item* items, int n, int rows, int cols,
                                size_t stride_bytes, const mpe_params* p, const double K[9], const double* D, int nD,
                                const double* markers_xyz, int n_markers) {
  if (!h || !items || n < 0 || !p || !K || !markers_xyz) return fail(h, MPE_ERR_ARG, "bad argument");
  if (h->pending_track_n) return fail(h, MPE_ERR_ARG, "a submitted batch has not been collected yet");
  if (n == 0) return MPE_OK;
  int rmax = 0, wmax = 0;
  for (int i = 0; i < n; ++i) {
    const mpe_track_item& it = items[i];
    if (!it.img || it.roi_x < 0 || it.roi_y < 0 || it.roi_w <= 0 || it.roi_h <= 0 || it.roi_x + it.roi_w > cols ||
        it.roi_y + it.roi_h > rows)
      return fail(h, MPE_ERR_ARG, "ROI outside the image");
    rmax = std::max(rmax, it.roi_h);
    wmax = std::max(wmax, it.roi_w);
  }
  ENTER(h);
  FrameGeom g;
  if (make_geom(h, rmax, wmax, g)) return fail(h, MPE_ERR_UNSUPPORTED, "frame size unsupported");
  DetectParams dp;
  if (make_detect_params(p, K, D, nD, 0, 0, dp)) return fail(h, MPE_ERR_ARG, "gaussian_sigma must be in (0, 6]");
  SolveParams sp;
  if (make_solve_params(h, p, markers_xyz, n_markers, K, sp)) return fail(h, MPE_ERR_UNSUPPORTED, "n_markers > MPE_MAX_MARKERS");
  const size_t slot = (size_t)g.rows * g.pitch;
  const size_t pred_bytes = (size_t)n * 2 * MPE_MAX_MARKERS * sizeof(double);
  const size_t win_bytes = ((size_t)n * 4 * sizeof(int) + 15) & ~(size_t)15;
  const size_t in_bytes = pred_bytes + win_bytes + (size_t)n * slot;
  const size_t rec_bytes = (size_t)n * (sizeof(mpe_detections) + 2 * MPE_MAX_MARKERS * sizeof(uint32_t) + sizeof(mpe_result));
  const size_t need = in_bytes + rec_bytes + 256;
  if (need > h->mailbox_cap) {
    if (h->mailbox) (void)hipHostFree(h->mailbox);
    h->mailbox = nullptr;
    h->mailbox_cap = 0;
    const size_t want = std::max(need + need / 4, (size_t)1 << 16);
    HIP_TRY(h, hipHostMalloc(&h->mailbox, want, hipHostMallocDefault));
    h->mailbox_cap = want;

  }
  uint8_t* mb = static_cast<uint8_t*>(h->mailbox);
  double* pred = reinterpret_cast<double*>(mb);
  int* wins = reinterpret_cast<int*>(mb + pred_bytes);
  uint8_t* pix = mb + pred_bytes + win_bytes;
  const double qnan = std::nan("");
  for (int i = 0; i < n; ++i) {
    const mpe_track_item& it = items[i];
    // no predicted pixels = detection only: NaN predictions are nearest to nothing, the tail then reports "no pose"
    for (int k = 0; k < 2 * MPE_MAX_MARKERS; ++k)
      pred[(size_t)i * 2 * MPE_MAX_MARKERS + k] = (it.predicted_px && k < 2 * n_markers) ? it.predicted_px[k] : (it.predicted_px ? 0.0 : qnan);
    wins[4 * i] = it.roi_h;
    wins[4 * i + 1] = it.roi_w;
    wins[4 * i + 2] = it.roi_x;
    wins[4 * i + 3] = it.roi_y;
    uint8_t* dst0 = pix + (size_t)i * slot;
    for (int y = 0; y < g.rows; ++y) {
      uint8_t* dst = dst0 + (size_t)y * g.pitch;
      if (y < it.roi_h) {
        std::memcpy(dst, it.img + (size_t)(it.roi_y + y) * stride_bytes + it.roi_x, (size_t)it.roi_w);
        if (g.pitch > it.roi_w) std::memset(dst + it.roi_w, 0, (size_t)(g.pitch - it.roi_w));
      } else {
        std::memset(dst, 0, (size_t)g.pitch);
      }
    }
  }
  uint8_t* host_rec = mb + ((in_bytes + 255) & ~(size_t)255);
  HIP_TRY(h, h->frames.reserve(in_bytes + 16));
  HIP_TRY(h, h->flags.reserve(std::max(flag_words((size_t)n * slot), (size_t)n * track_flag_words(g)) * 8));
  HIP_TRY(h, h->work.reserve((size_t)2 * (n + 1) * sizeof(int)));
  HIP_TRY(h, h->scratch.reserve(k1b_scratch_bytes(g, n)));
  HIP_TRY(h, h->hist.reserve((size_t)n * MPE_HIST_STRIDE * sizeof(uint32_t)));
  HIP_TRY(h, h->track.reserve(rec_bytes));
  HIP_TRY(h, h->mid.reserve(k3_mid_bytes(n)));
  uint8_t* d_in = static_cast<uint8_t*>(h->frames.p);
  const double* d_pred = reinterpret_cast<const double*>(d_in);
  const void* d_wins = d_in + pred_bytes;
  const uint8_t* d_pix = d_in + pred_bytes + win_bytes;
  mpe_detections* d_dets = static_cast<mpe_detections*>(h->track.p);
  uint32_t* d_corr = reinterpret_cast<uint32_t*>(d_dets + n);
  mpe_result* d_res = reinterpret_cast<mpe_result*>(d_corr + (size_t)n * 2 * MPE_MAX_MARKERS);
  h->have_ms = false;
  HIP_TRY(h, hipMemcpyAsync(d_in, mb, in_bytes, hipMemcpyHostToDevice, h->stream));
  // the small blob tier alone (see mpe_track_step): a slot that overflows it is seen by _collect, which then repeats
  // the blob extraction and the tail of the whole submission through the tier chain
  mpe_handle::PendingTrack& pt = h->pending_track;
  pt.optimistic = sp.n_markers >= 1 && sp.n_markers <= 8;
  pt.fused = pt.optimistic && h->track_fused;
  pt.g = g;
  pt.dp = dp;
  pt.sp = sp;
  pt.nn_tol = p->nearest_neighbour_pixel_tolerance;
  pt.rec_bytes = rec_bytes;
  pt.d_pix = d_pix;
  pt.d_wins = d_wins;
  pt.d_pred = d_pred;
  if (pt.fused) {
    // round 6: the time step of the n streams as ONE launch, a block per stream (k_track_frame), the records stored to
    // the pinned staging memory by the kernel (track_fused 2) — scan, small blob tier, tail and copy-out were five
    // commands, and every stage waited for the slowest stream of the one before
    const bool deliver = h->track_fused >= 2;
    mpe_detections* hd = reinterpret_cast<mpe_detections*>(host_rec);
    uint32_t* hc = reinterpret_cast<uint32_t*>(hd + n);
    mpe_result* hr = reinterpret_cast<mpe_result*>(hc + (size_t)n * 2 * MPE_MAX_MARKERS);
    TrackFramesArgs ta = {d_pix, slot, d_pred, d_wins, static_cast<unsigned long long*>(h->flags.p),
                          static_cast<uint32_t*>(h->hist.p), h->mid.p, d_dets, d_corr, d_res, deliver ? hd : nullptr,
                          deliver ? hc : nullptr, deliver ? hr : nullptr, nullptr};
    HIP_TRY(h, launch_track_frames(ta, n, g, dp, sp, pt.nn_tol, h->stream));
    if (!deliver) HIP_TRY(h, hipMemcpyAsync(host_rec, d_dets, rec_bytes, hipMemcpyDeviceToHost, h->stream));
  } else {
    HIP_TRY(h, launch_k1a_scan(d_pix, (size_t)n * slot, static_cast<unsigned long long*>(h->flags.p), dp.thr, 0, h->stream));
    HIP_TRY(h, launch_k1b_blobs(d_pix, static_cast<unsigned long long*>(h->flags.p), n, g, dp, d_dets,
                                static_cast<int*>(h->work.p), static_cast<uint8_t*>(h->scratch.p), h->scratch.cap, sp.n_markers, h->stream,
                                d_wins, false, pt.optimistic));
    HIP_TRY(h, launch_k3_tail(d_dets, static_cast<uint32_t*>(h->hist.p), n, sp, d_res, d_corr, nullptr, d_pred, pt.nn_tol,
                              h->mid.p, h->stream));
    HIP_TRY(h, hipMemcpyAsync(host_rec, d_dets, rec_bytes, hipMemcpyDeviceToHost, h->stream));
  }
  pt.slot_bytes = slot;
  h->pending_track_n = n;
  h->pending_track_rec = host_rec;
  return MPE_OK;
}

int mpe_track_step_batch_cancel(mpe_handle* h) {
  if (!h) return MPE_ERR_ARG;
  if (h->pending_track_n == 0) return MPE_OK;
  h->pending_track_n = 0;
  h->pending_track_rec = nullptr;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));  // the copy-out of the abandoned submission has left the staging memory
  return MPE_OK;
}

int mpe_track_step_batch_collect(mpe_handle* h, mpe_detections* dets_out, uint32_t* corr_out, mpe_result* out) {
  if (!h || !dets_out || !corr_out || !out) return fail(h, MPE_ERR_ARG, "bad argument");
  const int n = h->pending_track_n;
  if (n == 0) return fail(h, MPE_ERR_ARG, "no submitted batch to collect (did mpe_track_step_batch_submit fail?)");
  const uint8_t* host_rec = h->pending_track_rec;
  h->pending_track_n = 0;
  h->pending_track_rec = nullptr;
  ENTER(h);
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  const mpe_detections* hd = reinterpret_cast<const mpe_detections*>(host_rec);
  const mpe_handle::PendingTrack& pt = h->pending_track;
  if (pt.optimistic) {
    bool again = false;
    for (int i = 0; i < n && !again; ++i) again = hd[i].status == MPE_FRAME_TOO_MANY_ROWS;
    if (again) {  // (the inputs are still on the device: nothing has been submitted on this handle since)
      mpe_detections* d_dets = static_cast<mpe_detections*>(h->track.p);
      uint32_t* d_corr = reinterpret_cast<uint32_t*>(d_dets + n);
      mpe_result* d_res = reinterpret_cast<mpe_result*>(d_corr + (size_t)n * 2 * MPE_MAX_MARKERS);
      if (pt.fused)  // (the blob tiers read the image pass's flag bitstream over all slots)
        HIP_TRY(h, launch_k1a_scan(pt.d_pix, (size_t)n * pt.slot_bytes, static_cast<unsigned long long*>(h->flags.p),
                                   pt.dp.thr, 0, h->stream));
      HIP_TRY(h, launch_k1b_blobs(pt.d_pix, static_cast<unsigned long long*>(h->flags.p), n, pt.g, pt.dp, d_dets,
                                  static_cast<int*>(h->work.p), static_cast<uint8_t*>(h->scratch.p), h->scratch.cap, pt.sp.n_markers,
                                  h->stream, pt.d_wins));
      HIP_TRY(h, launch_k3_tail(d_dets, static_cast<uint32_t*>(h->hist.p), n, pt.sp, d_res, d_corr, nullptr, pt.d_pred,
                                pt.nn_tol, h->mid.p, h->stream));
      HIP_TRY(h, hipMemcpyAsync(const_cast<uint8_t*>(host_rec), d_dets, pt.rec_bytes, hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
  }
  const uint32_t* hc = reinterpret_cast<const uint32_t*>(hd + n);
  const mpe_result* hr = reinterpret_cast<const mpe_result*>(hc + (size_t)n * 2 * MPE_MAX_MARKERS);
  std::memcpy(dets_out, hd, (size_t)n * sizeof(mpe_detections));
  std::memcpy(corr_out, hc, (size_t)n * 2 * MPE_MAX_MARKERS * sizeof(uint32_t));
  std::memcpy(out, hr, (size_t)n * sizeof(mpe_result));
  return MPE_OK;
}

int mpe_track_step_batch(mpe_handle* h, const mpe_track_item* items, int n, int rows, int cols, size_t stride_bytes,
                         const mpe_params* p, const double K[9], const double* D, int nD, const double* markers_xyz,
                         int n_markers, mpe_detections* dets_out, uint32_t* corr_out, mpe_result* out) {
  if (!dets_out || !corr_out || !out) return fail(h, MPE_ERR_ARG, "bad argument");
  const int rc = mpe_track_step_batch_submit(h, items, n, rows, cols, stride_bytes, p, K, D, nD, markers_xyz, n_markers);
  if (rc != MPE_OK) return rc;
  if (n == 0) return MPE_OK;  // (nothing was submitted)
  return mpe_track_step_batch_collect(h, dets_out, corr_out, out);
}

// setImagePoints + initialise + optimiseAndUpdatePose for N detection sets in one submission (the brute-force
// re-initialisations of a lock-step batch): det_xy n x MPE_MAX_DETECTIONS x 2, n_det[i] valid rows each; hist
// (optional) n x MPE_MAX_DETECTIONS x MPE_MAX_MARKERS, corr (optional) n x 2*MPE_MAX_MARKERS.
int mpe_solve_bruteforce_batch(mpe_handle* h, const double* det_xy, const int* n_det, int n, const double* markers_xyz,
                               int n_markers, const double K[9], const mpe_params* p, mpe_result* out, uint32_t* hist,
                               uint32_t* corr) {
  if (!h || !det_xy || !n_det || n < 0 || !markers_xyz || !K || !p || !out) return fail(h, MPE_ERR_ARG, "bad argument");
  if (n == 0) return MPE_OK;
  ENTER(h);
  SolveParams sp;
  if (make_solve_params(h, p, markers_xyz, n_markers, K, sp)) return fail(h, MPE_ERR_UNSUPPORTED, "n_markers > MPE_MAX_MARKERS");
  std::vector<mpe_detections> hd((size_t)n);
  int nd_max = 0;
  for (int f = 0; f < n; ++f) {
    std::memset(&hd[f], 0, sizeof(mpe_detections));
    if (n_det[f] < 0 || n_det[f] > MPE_MAX_DETECTIONS) return fail(h, MPE_ERR_ARG, "n_det out of range");
    hd[f].n = n_det[f];
    nd_max = std::max(nd_max, n_det[f]);
    std::memcpy(hd[f].undist_xy, det_xy + (size_t)f * 2 * MPE_MAX_DETECTIONS, sizeof(double) * 2 * n_det[f]);
  }
  const size_t hist_bytes = (size_t)n * MPE_HIST_STRIDE * sizeof(uint32_t);
  const size_t corr_bytes = (size_t)n * 2 * MPE_MAX_MARKERS * sizeof(uint32_t);
  HIP_TRY(h, h->dets.reserve((size_t)n * sizeof(mpe_detections)));
  HIP_TRY(h, h->hist.reserve(hist_bytes));
  HIP_TRY(h, h->results.reserve((size_t)n * sizeof(mpe_result)));
  HIP_TRY(h, h->corr.reserve(corr_bytes));
  HIP_TRY(h, h->mid.reserve(k3_mid_bytes(n)));
  HIP_TRY(h, hipMemcpyAsync(h->dets.p, hd.data(), (size_t)n * sizeof(mpe_detections), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemsetAsync(h->hist.p, 0, hist_bytes, h->stream));
  { const int rc = prep_marker_table(h, sp); if (rc) return rc; }
  VoteFixup fx;
  { const int rc = vote_fixup_for(h, 0, 1, n, n_markers, nd_max, h->stream, fx); if (rc) return rc; }
  HIP_TRY(h, launch_k2_vote(static_cast<mpe_detections*>(h->dets.p), n, sp, static_cast<const double*>(h->mtab.p),
                            static_cast<uint32_t*>(h->hist.p), auto_splits(h, n, n_markers), nd_max, h->stream, nullptr,
                            0, nullptr, 0, nullptr, nullptr, &fx));
  HIP_TRY(h, fixup_launch(h, 0, static_cast<mpe_detections*>(h->dets.p), n, sp, static_cast<uint32_t*>(h->hist.p), fx,
                          h->stream));
  HIP_TRY(h, launch_k3_tail(static_cast<mpe_detections*>(h->dets.p), static_cast<uint32_t*>(h->hist.p), n, sp,
                            static_cast<mpe_result*>(h->results.p), static_cast<uint32_t*>(h->corr.p), nullptr, nullptr,
                            0.0, h->mid.p, h->stream));
  HIP_TRY(h, hipMemcpyAsync(out, h->results.p, (size_t)n * sizeof(mpe_result), hipMemcpyDeviceToHost, h->stream));
  if (hist)  // (device rows are MPE_HIST_STRIDE words apart, the caller's MPE_HIST_WORDS)
    HIP_TRY(h, hipMemcpy2DAsync(hist, MPE_HIST_WORDS * sizeof(uint32_t), h->hist.p, MPE_HIST_STRIDE * sizeof(uint32_t),
                                MPE_HIST_WORDS * sizeof(uint32_t), (size_t)n, hipMemcpyDeviceToHost, h->stream));
  if (corr) HIP_TRY(h, hipMemcpyAsync(corr, h->corr.p, corr_bytes, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPE_OK;
}

// ---- one host process, several GPUs -----------------------------------------------------------
// Frames are independent on the uninitialised branch (pose_estimator.cpp:68-91 reads no estimator state), so
// a batch shards into contiguous chunks, one per handle / device, with no exchange step: one host thread per
// shard drives that handle's ordinary single-device entry point and writes its slice of the one result array.
void mpe_shard_bounds(int n_frames, int shard, int n_shards, int* lo, int* hi) {
  if (n_shards < 1) n_shards = 1;
  if (n_frames < 0) n_frames = 0;
  const int base = n_frames / n_shards, rem = n_frames % n_shards;
  const int a = shard * base + std::min(shard, rem);
  if (lo) *lo = a;
  if (hi) *hi = a + base + (shard < rem ? 1 : 0);
}


int mpe_estimate_batch_multi(mpe_handle* const* handles, int n_dev, const uint8_t* frames, int n_frames, int rows,
                             int cols, size_t stride_bytes, size_t frame_stride_bytes, const double* markers_xyz,
                             int n_markers, const double K[9], const double* D, int nD, const mpe_params* p,
                             mpe_result* results) {
  if (!frames || !results || n_frames < 0) return MPE_ERR_ARG;
  return run_shards(handles, n_dev, [&](int d) -> int {
    int lo, hi;
    mpe_shard_bounds(n_frames, d, n_dev, &lo, &hi);
    if (hi <= lo) return MPE_OK;
    return mpe_estimate_batch(handles[d], frames + (size_t)lo * frame_stride_bytes, hi - lo, rows, cols, stride_bytes,
                              frame_stride_bytes, 0, markers_xyz, n_markers, K, D, nD, p, results + lo);
  });
}

int mpe_estimate_batch_multi_device(mpe_handle* const* handles, int n_dev, const uint8_t* const* d_frames,
                                    const int* n_frames, int rows, int cols, const double* markers_xyz, int n_markers,
                                    const double K[9], const double* D, int nD, const mpe_params* p,
                                    mpe_result* results) {
  if (!d_frames || !n_frames || !results) return MPE_ERR_ARG;
  std::vector<size_t> off((size_t)std::max(n_dev, 1) + 1, 0);
  for (int d = 0; d < n_dev; ++d) {
    if (n_frames[d] < 0 || (n_frames[d] > 0 && !d_frames[d])) return MPE_ERR_ARG;
    off[(size_t)d + 1] = off[(size_t)d] + (size_t)n_frames[d];
  }
  return run_shards(handles, n_dev, [&](int d) -> int {
    if (n_frames[d] == 0) return MPE_OK;
    return mpe_estimate_batch(handles[d], d_frames[d], n_frames[d], rows, cols, (size_t)cols, (size_t)rows * cols, 1,
                              markers_xyz, n_markers, K, D, nD, p, results + off[(size_t)d]);
  });
}

}  // extern "C"

// ---- several GPUs from one process, records gathered ON THE DEVICE over RCCL -------------------------------
// ("ncclCommInitAll, one host thread + stream per device"; the only exchange of the sharded path is the
// gather of the 432-byte pose records.)  RCCL is loaded at first use (dlopen of librccl.so: no link-time dependency,
// and no clash with the copy a host framework may have loaded); one communicator set per device list, kept for the
// life of the process.
namespace {
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
  bool load() {
    if (lib) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) {
      err = "librccl.so not found (dlopen)";
      return false;
    }
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
    Send = reinterpret_cast<decltype(Send)>(dlsym(lib, "ncclSend"));
    Recv = reinterpret_cast<decltype(Recv)>(dlsym(lib, "ncclRecv"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    if (!CommInitAll || !GroupStart || !GroupEnd || !Send || !Recv) {
      err = "librccl.so lacks ncclCommInitAll / ncclGroupStart / ncclSend / ncclRecv";
      lib = nullptr;
      return false;
    }
    return true;
  }
};
struct RcclComms {
  std::vector<int> devices;
  std::vector<ncclComm_t> comms;
};
std::mutex g_rccl_mutex;
RcclApi g_rccl;
std::vector<RcclComms> g_rccl_comms;
}  // namespace

extern "C" {

int mpe_estimate_batch_multi_device_gather(mpe_handle* const* handles, int n_dev, const uint8_t* const* d_frames,
                                           const int* n_frames, int rows, int cols, const double* markers_xyz,
                                           int n_markers, const double K[9], const double* D, int nD, const mpe_params* p,
                                           mpe_result* d_results_dev0, int* used_rccl) {
  if (used_rccl) *used_rccl = 0;
  if (!handles || n_dev < 1 || !d_frames || !n_frames || !d_results_dev0) return MPE_ERR_ARG;
  std::vector<size_t> off((size_t)n_dev + 1, 0);
  for (int d = 0; d < n_dev; ++d) {
    if (!handles[d] || n_frames[d] < 0 || (n_frames[d] > 0 && !d_frames[d])) return MPE_ERR_ARG;
    off[(size_t)d + 1] = off[(size_t)d] + (size_t)n_frames[d];
  }
  // distinct devices -> RCCL; handles that share a device (a 1-GPU box) -> plain device-to-device copies.
  // Option "force_rccl_gather" on handles[0]: RCCL for every shard including shard 0 (which then sends to itself) —
  // with ONE handle that exercises the whole leg (library load, communicator, grouped send / recv) on a 1-GPU box.
  bool distinct = true;
  for (int d = 0; d < n_dev; ++d)
    for (int e = 0; e < d; ++e) distinct = distinct && handles[d]->device != handles[e]->device;
  const bool self_send = handles[0]->force_rccl_gather != 0 && distinct;
  const bool via_rccl = distinct && (n_dev > 1 || self_send);
  const int first = self_send ? 0 : 1;  // first shard whose records travel
  // every shard computes into its own device: shard 0 straight into the result array (unless it is to travel as
  // well), the others into their handle's record buffer
  std::vector<mpe_result*> d_part((size_t)n_dev, nullptr);
  int rc = run_shards(handles, n_dev, [&](int d) -> int {
    mpe_handle* h = handles[d];
    if (n_frames[d] == 0) return MPE_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    if (d < first) {
      d_part[0] = d_results_dev0;
    } else {
      HIP_TRY(h, h->results.reserve((size_t)n_frames[d] * sizeof(mpe_result)));
      d_part[(size_t)d] = static_cast<mpe_result*>(h->results.p);
    }
    return mpe_estimate_batch_device(h, d_frames[d], n_frames[d], rows, cols, markers_xyz, n_markers, K, D, nD, p,
                                     d_part[(size_t)d]);
  });
  if (rc != MPE_OK) return rc;
  mpe_handle* h0 = handles[0];
  if (via_rccl) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (!g_rccl.load()) return fail(h0, MPE_ERR_UNSUPPORTED, g_rccl.err.c_str());
    std::vector<int> devs((size_t)n_dev);
    for (int d = 0; d < n_dev; ++d) devs[(size_t)d] = handles[d]->device;
    RcclComms* cs = nullptr;
    for (auto& c : g_rccl_comms)
      if (c.devices == devs) cs = &c;
    if (!cs) {
      RcclComms c;
      c.devices = devs;
      c.comms.resize((size_t)n_dev);
      const ncclResult_t r = g_rccl.CommInitAll(c.comms.data(), n_dev, devs.data());
      if (r != ncclSuccess) return fail(h0, MPE_ERR_HIP, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "ncclCommInitAll failed");
      g_rccl_comms.push_back(c);
      cs = &g_rccl_comms.back();
    }
    // one grouped exchange: rank d sends its records to rank 0 on its own stream (behind its kernels), rank 0 receives
    // them into their place of the result array on its stream — point-to-point over xGMI, 432 B per frame
    ncclResult_t r = g_rccl.GroupStart();
    for (int d = first; d < n_dev && r == ncclSuccess; ++d) {
      if (n_frames[d] == 0) continue;
      const size_t bytes = (size_t)n_frames[d] * sizeof(mpe_result);
      (void)hipSetDevice(handles[d]->device);
      r = g_rccl.Send(d_part[(size_t)d], bytes, ncclUint8, 0, cs->comms[(size_t)d], handles[d]->stream);
      if (r != ncclSuccess) break;
      (void)hipSetDevice(h0->device);
      r = g_rccl.Recv(d_results_dev0 + off[(size_t)d], bytes, ncclUint8, d, cs->comms[0], h0->stream);
    }
    const ncclResult_t r2 = g_rccl.GroupEnd();
    // the senders' buffers are free again once their streams are through — also when the exchange failed half way:
    // whatever was enqueued must not still be reading a handle's record buffer when the caller retries
    hipError_t sync_err = hipSuccess;
    mpe_handle* sync_h = nullptr;
    for (int d = 0; d < n_dev; ++d) {
      (void)hipSetDevice(handles[d]->device);
      const hipError_t e = hipStreamSynchronize(handles[d]->stream);
      if (e != hipSuccess && sync_err == hipSuccess) {
        sync_err = e;
        sync_h = handles[d];
      }
    }
    if (r != ncclSuccess || r2 != ncclSuccess)
      return fail(h0, MPE_ERR_HIP, g_rccl.GetErrorString ? g_rccl.GetErrorString(r != ncclSuccess ? r : r2) : "RCCL send / recv failed");
    if (sync_err != hipSuccess) return fail(sync_h, MPE_ERR_HIP, "hipStreamSynchronize after the RCCL gather", sync_err);
    if (used_rccl) *used_rccl = 1;
  } else {
    for (int d = 1; d < n_dev; ++d) {
      if (n_frames[d] == 0) continue;
      mpe_handle* h = handles[d];
      HIP_TRY(h, hipSetDevice(h->device));
      HIP_TRY(h, hipMemcpyAsync(d_results_dev0 + off[(size_t)d], d_part[(size_t)d], (size_t)n_frames[d] * sizeof(mpe_result),
                                hipMemcpyDeviceToDevice, h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
  }
  HIP_TRY(h0, hipSetDevice(h0->device));
  HIP_TRY(h0, hipStreamSynchronize(h0->stream));
  return MPE_OK;
}

}  // extern "C"
