// mpe_track_abi.cpp — host side of libmpe_hip.so, part 3 (see mpe_host.h): one tracked frame (mpe_track_step) and the
// lock-step time step of N camera streams (mpe_track_step_batch[_submit / _collect / _cancel]); the per-stream state
// machine on top of them is mpe_tracker.cpp.
#include "mpe_host.h"

extern "C" {

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

}  // extern "C"
