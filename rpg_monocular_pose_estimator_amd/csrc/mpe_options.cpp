// mpe_options.cpp — host side of libmpe_hip.so, part 2 (see mpe_host.h): library / device introspection, the handle's
// life cycle, streams, profiling read-outs, mpe_get_option / mpe_set_option.
#include "mpe_host.h"

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
  if (h->gen_seen_host) (void)hipHostFree(h->gen_seen_host);
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

int mpe_last_kernel_ms(mpe_handle* h, float ms[5]) {
  if (!h || !ms) return MPE_ERR_ARG;
  if (h->ms_accum_valid) {  // a chunked host ingest: sums over its chunks
    for (int i = 0; i < 5; ++i) ms[i] = h->ms_accum[i];
    return MPE_OK;
  }
  return last_kernel_ms_of_call(h, ms);
}
}  // extern "C"
namespace mpe_host {
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
}  // namespace mpe_host
extern "C" {

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
  else if (n == "general_lds") *value = h->general_lds;
  else if (n == "general_seen") *value = h->gen_seen_host ? *h->gen_seen_host : 0;
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
  if (!std::strcmp(name, "general_lds")) {  // the general blob tier's kernel: -1 automatic, 0 slabs in global memory, 1 LDS-resident
    if (value < -1 || value > 1) return fail(h, MPE_ERR_ARG, "general_lds must be -1, 0 or 1");
    h->general_lds = value;
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

}  // extern "C"
