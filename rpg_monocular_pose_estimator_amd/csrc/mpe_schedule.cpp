// mpe_schedule.cpp — host side of libmpe_hip.so, part 1 (see mpe_host.h): error / device entry helpers, parameter
// marshalling (mpe_params -> DetectParams / SolveParams / FrameGeom), frame staging, the suspect lists of the voting
// launches, the side streams and their concurrency probe, and run_pipeline — the schedules of one batch.
#include "mpe_host.h"

namespace mpe_host {

int fail(mpe_handle* h, int code, const char* what, hipError_t e) {
  if (h) {
    h->err = what;
    if (e != hipSuccess) {
      h->err += ": ";
      h->err += hipGetErrorString(e);
    }
  }
  return code;
}


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

// the general blob tier as k1b_general_lds?  (see mpe_handle::gen_seen_host)
bool general_lds_now(const mpe_handle* h) {
  if (h->general_lds >= 0) return h->general_lds != 0;
  return h->gen_seen_host && *h->gen_seen_host > 0;
}
// ... and the reading for the next call: list_b's count of a blob launch over n_frames frames, behind it on `st`
hipError_t general_seen_copy(mpe_handle* h, const int* worklist, int n_frames, hipStream_t st) {
  if (!h->gen_seen_host) {
    const hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&h->gen_seen_host), 64, hipHostMallocDefault);
    if (e != hipSuccess) return e;
    *h->gen_seen_host = 0;
  }
  return hipMemcpyAsync(h->gen_seen_host, worklist + (n_frames + 1), sizeof(int), hipMemcpyDeviceToHost, st);
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
                        const VoteFixup& fx, hipStream_t st, const int* item_range) {
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
  int* wl = static_cast<int*>(h->work.p) + (size_t)chain * 2 * (chain_frames + 1);
  HIP_TRY(h, launch_k1b_blobs(d_frames, d_flags, n_frames, g, dp, d_dets, wl, static_cast<uint8_t*>(h->scratch.p),
                              h->scratch.cap, sp ? sp->n_markers : 0, st, nullptr, false, false, general_lds_now(h)));
  if (chain == 0 && n_frames >= 64) HIP_TRY(h, general_seen_copy(h, wl, n_frames, st));
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
                 uint32_t* d_corr, const StreamHint* hint) {
  h->done_recorded = false;
  h->ms_accum_valid = false;
  const bool gen_lds = general_lds_now(h);  // (one reading per call: every sub-batch the same kernel)
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
                                  static_cast<uint8_t*>(h->scratch.p), h->scratch.cap, sp->n_markers, st, nullptr, true,
                                  false, gen_lds));
      if (s == 0) HIP_TRY(h, general_seen_copy(h, static_cast<int*>(h->work.p), nf, st));
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
                                static_cast<uint8_t*>(h->scratch.p), h->scratch.cap, sp->n_markers, sblob, nullptr, false,
                                false, gen_lds));
    if (s == 0) HIP_TRY(h, general_seen_copy(h, static_cast<int*>(h->work.p), nf, sblob));
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

}  // namespace mpe_host
