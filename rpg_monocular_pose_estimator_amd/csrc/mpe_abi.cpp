// mpe_abi.cpp — host side of libmpe_hip.so, part 4 (see mpe_host.h): the entries of include/mpe.h that are not
// options (mpe_options.cpp) or tracked frames (mpe_track_abi.cpp) — detection, voting, brute force, validation /
// refinement, primitive batches, the batch entries (host, device-resident, streaming), frame decode, pinned memory,
// several GPUs from one process (threads; RCCL record gather).
#include "mpe_host.h"

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

extern "C" {

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
