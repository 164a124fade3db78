// mpe_host.h — what the host-side translation units of libmpe_hip.so share (round 6: mpe_abi.cpp was one 2 800-line
// file; it is now mpe_schedule.cpp — parameter marshalling, workspaces, the schedules of a batch (run_pipeline) —,
// mpe_options.cpp — handle life cycle, streams, profiling read-outs, mpe_set_option / mpe_get_option —,
// mpe_track_abi.cpp — tracked frames and lock-step time steps — and mpe_abi.cpp — every other entry of include/mpe.h;
// same exported symbols).  No torch, no CPU fallback: without a HIP device every entry point fails with
// MPE_ERR_NO_DEVICE / MPE_ERR_HIP.
#pragma once
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
  // Do frames reach the general blob tier?  A pinned mirror of the first blob launch's hand-over count, copied behind it
  // once per call and read — a call late — when the next call picks that tier's kernel: k1b_general_lds (a CU's whole
  // LDS per block: not something to launch empty into every blob window) once frames have been seen there, the
  // slab kernel otherwise.  Option "general_lds": 0 never (DEFAULT: measured, the LDS-resident kernel is 7 % faster on
  // frames with one large blob and 2 x SLOWER on salt noise — mpe_k1.hip), -1 that automatic choice, 1 always.
  int* gen_seen_host = nullptr;
  int general_lds = 0;
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

// ---- shared helpers (defined in mpe_schedule.cpp unless noted) ------------------------------------------------------------
namespace mpe_host {
int fail(mpe_handle* h, int code, const char* what, hipError_t e = hipSuccess);
// Every entry point that re-uses the handle's device buffers on its stream: select the device and, if a streaming
// submission still has validate / refine kernels on the internal tail stream, make the handle's stream wait for them
int enter(mpe_handle* h);
unsigned num_combinations_u32(unsigned n, unsigned k);
int make_detect_params(const mpe_params* p, const double K[9], const double* D, int nD, int roi_x, int roi_y,
                       DetectParams& dp);
int make_solve_params(const mpe_handle* h, const mpe_params* p, const double* markers, int n_markers, const double K[9],
                      SolveParams& sp);
int make_geom(const mpe_handle* h, int rows, int cols, FrameGeom& g);
size_t flag_words(size_t n_bytes);
int stage_frames(mpe_handle* h, const uint8_t* frames, int n_frames, int rows, int cols, size_t stride,
                 size_t frame_stride, int on_device, int roi_x, int roi_y, int roi_w, int roi_h, const FrameGeom& g,
                 const uint8_t** d_out);
int det_hint_for(const mpe_handle* h, int n_markers);
int auto_splits(const mpe_handle* h, int n_frames, int n_markers);
constexpr size_t kFixCtlBytes = (size_t)mpe_handle::kMaxSub * MPE_FIX_CTL_WORDS * sizeof(unsigned);
constexpr size_t kFixEntryBytes = 2 * sizeof(unsigned long long);
int fix_counter_sum(mpe_handle* h, int which, unsigned long long& out);
int vote_fixup_for(mpe_handle* h, int slot, int n_slots, int n_frames, int n_markers, int n_det_hint, hipStream_t st,
                   VoteFixup& fx);
hipError_t fixup_launch(mpe_handle* h, int slot, mpe_detections* dets, int n_frames, const SolveParams& sp, uint32_t* hist,
                        const VoteFixup& fx, hipStream_t st, const int* item_range = nullptr);
int prep_marker_table(mpe_handle* h, const SolveParams& sp);
void sub_batch_shape(const mpe_handle* h, int n_frames, size_t frame_bytes, bool have_sp, int vote_arith, int& nsub,
                     int& per);
// streaming: what the caller knows about the submission that follows this one
struct StreamHint {
  const uint8_t* next_frames = nullptr;  // device frames of the next submission (same geometry / parameters), or null
  int n_next = 0;
  bool no_join = false;  // do not join the side streams back into the caller's stream: completion = batch_done event
  hipEvent_t next_ready = nullptr;  // the announced frames are final once this event has completed (mpe_stream_next_ready)
};
int run_pipeline(mpe_handle* h, const uint8_t* d_frames, int n_frames, const FrameGeom& g, const DetectParams& dp,
                 const SolveParams* sp, mpe_detections* d_dets, uint32_t* d_hist, mpe_result* d_results,
                 uint32_t* d_corr, const StreamHint* hint = nullptr);
int last_kernel_ms_of_call(mpe_handle* h, float ms[5]);  // (mpe_options.cpp)
}  // namespace mpe_host
using namespace mpe_host;

#define HIP_TRY(h, call)                                         \
  do {                                                           \
    hipError_t e__ = (call);                                     \
    if (e__ != hipSuccess) return fail(h, MPE_ERR_HIP, #call, e__); \
  } while (0)
#define ENTER(h)                  \
  do {                            \
    const int rc__ = enter(h);    \
    if (rc__ != MPE_OK) return rc__; \
  } while (0)
