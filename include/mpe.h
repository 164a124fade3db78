/*
 * mpe.h — C ABI of the MI355X (gfx950) compute back-end for the per-frame hot path of
 * uzh-rpg/rpg_monocular_pose_estimator.
 *
 * The reference has no C ABI for this path; its seams are C++.  Every entry
 * point below names the reference interface it replaces (paths relative to the reference root,
 * lib = monocular_pose_estimator_lib).  The library is libmpe_hip.so, built by
 * rpg_monocular_pose_estimator_amd/csrc/Makefile with hipcc --offload-arch=gfx950.
 *
 * Conventions
 *   - plain pointers and sizes only; caller owns every buffer; no exceptions cross the ABI.
 *   - one mpe_handle = one GPU + one HIP stream; a handle is used by one thread at a time
 *     (the reference's PoseEstimator is a single-threaded stateful object too, node.cpp:31).
 *   - return value: 0 = call executed, <0 = usage / HIP error (see mpe_last_error()).
 *     Per-frame outcome is in mpe_result.status: 0 = pose found (estimateBodyPose -> true),
 *     1 = no pose (-> false), <0 = frame exceeded a documented device capacity (never silent).
 *   - matrices are row-major doubles: K[9], T[16] (T_camera_object), cov[36] in twist order
 *     (upsilon, omega) exactly as PoseEstimator::getPoseCovariance() / ROS.cpp:178-187.
 *   - markers: n x 3 doubles (x,y,z) — List4DPoints without the homogeneous 1 (PE.h:351).
 *   - detections: n x 2 doubles, undistorted pixels — List2DPoints (DT.h:50).
 *   - correspondences: rows (marker, detection), 1-based, 0 = none — VectorXuPairs (DT.h:47).
 *   - hist: n_det x n_markers uint32 row-major (row = detection) — hist_corr, PE.cpp:556.
 */
#ifndef MPE_H_
#define MPE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPE_MAX_MARKERS 16    /* object_points_ capacity (reference: unbounded; 32-bit factorial
                                 already overflows at 13, COMB.cpp:34-45 — replicated) */
#define MPE_MAX_DETECTIONS 64 /* detections kept per frame; more -> status MPE_FRAME_TOO_MANY_DETECTIONS.  (The
                                 reference has no limit, led_detector.cpp:65-86 / pose_estimator.cpp:549-557; with 64
                                 detections and 5 markers initialise() already runs 2.5 M P3P solves for one frame.
                                 Frames with more than MPE_FAST_VOTE_DETECTIONS are voted by the strict loop nest.) */
#define MPE_FAST_VOTE_DETECTIONS 32 /* widest frame the fast voting kernels' 32-bit detection masks hold */
#define MPE_MAX_RAW_BLOBS 256 /* external contours per frame before the shape filter */
#define MPE_MAX_KSIZE 37      /* Gaussian kernel taps: sigma <= 6 (cfg:13) */

/* per-frame status codes (mpe_result.status, mpe_detections.status) */
#define MPE_FRAME_POSE 0
#define MPE_FRAME_NO_POSE 1
#define MPE_FRAME_TOO_MANY_DETECTIONS (-10) /* > MPE_MAX_DETECTIONS blobs passed the filter: the record then holds the
                                               first MPE_MAX_DETECTIONS of the reference's order (OpenCV's contour order)
                                               — unless more than 512 passed, when WHICH blobs it holds is unspecified
                                               (the general blob tier keeps 512, appended concurrently) */
#define MPE_FRAME_TOO_MANY_BLOBS (-11)      /* > MPE_MAX_RAW_BLOBS external contours */
#define MPE_FRAME_TOO_MANY_ROWS (-12)       /* bright rows exceed the LDS band capacity */
#define MPE_FRAME_VOTE_LIST_FULL (-13)      /* internal since round 5: hypotheses left to the strict arithmetic did not fit
                                               their list; such a frame is voted again by the strict loop nest before the
                                               tail and comes out with an ordinary status (never returned to a caller) */

/* call-level error codes */
#define MPE_OK 0
#define MPE_ERR_ARG (-1)
#define MPE_ERR_HIP (-2)
#define MPE_ERR_UNSUPPORTED (-3)
#define MPE_ERR_NO_DEVICE (-4)

typedef struct mpe_handle mpe_handle;

/* Mirrors the tuning members of PoseEstimator (PE.h:68-72, 85-91) and the dynamic-reconfigure
 * contract (monocular_pose_estimator/cfg/MonocularPoseEstimator.cfg:12-22). */
typedef struct mpe_params {
  int threshold_value;                      /* detection_threshold_value_ */
  double gaussian_sigma;                    /* gaussian_sigma_ */
  double min_blob_area;                     /* min_blob_area_ */
  double max_blob_area;                     /* max_blob_area_ */
  double max_width_height_distortion;       /* max_width_height_distortion_ */
  double max_circular_distortion;           /* max_circular_distortion_ */
  double back_projection_pixel_tolerance;   /* setBackProjectionPixelTolerance, PE.h:483 */
  double nearest_neighbour_pixel_tolerance; /* setNearestNeighbourPixelTolerance, PE.h:503 */
  double certainty_threshold;               /* setCertaintyThreshold, PE.h:523 */
  double valid_correspondence_threshold;    /* setValidCorrespondenceThreshold, PE.h:543 */
  unsigned roi_border_thickness;            /* roi_border_thickness_ (tracking path only) */
  unsigned histogram_threshold;             /* 0 -> numCombinations(n_markers,3) as PE.cpp:54 */
} mpe_params;

typedef struct mpe_result {
  double T[16];      /* getPredictedPose(), PE.h:406 */
  double cov[36];    /* getPoseCovariance(), PE.h:416 */
  int status;        /* MPE_FRAME_* */
  int n_det;         /* detections that passed the blob filter */
  int n_corr;        /* rows of correspondences_ after correspondencesFromHistogram */
  int gn_iterations; /* Gauss-Newton iterations used by optimisePose */
} mpe_result;

typedef struct mpe_detections {
  int n;      /* min(number of detections, MPE_MAX_DETECTIONS) */
  int status; /* 0 or MPE_FRAME_TOO_MANY_* */
  double undist_xy[2 * MPE_MAX_DETECTIONS]; /* pixel_positions of findLeds (LED.h:84-88); entries 2 n .. are unspecified */
  float dist_xy[2 * MPE_MAX_DETECTIONS];    /* distorted_detection_centers; entries 2 n .. are unspecified */
} mpe_detections;

/* fills *p with the parameter set of launch/demo.launch:12-22 */
void mpe_default_params(mpe_params* p);

/* device < 0: use the current HIP device.  Fails (MPE_ERR_NO_DEVICE) when no gfx950 GPU is
 * visible — there is no CPU fallback. */
int mpe_create(mpe_handle** out, int device);
void mpe_destroy(mpe_handle* h);
const char* mpe_last_error(const mpe_handle* h);
/* Use an existing hipStream_t (e.g. a torch.cuda.Stream); NULL restores the handle's own non-blocking
 * stream (it does NOT select the legacy default stream). */
int mpe_set_stream(mpe_handle* h, void* hip_stream);
void* mpe_get_stream(mpe_handle* h);
int mpe_synchronize(mpe_handle* h);

/* ≙ LEDDetector::findLeds (lib/include/.../led_detector.h:84-88, lib/src/led_detector.cpp:35-112)
 * on ONE host frame: threshold -> Gaussian blur -> external contours -> polygon area/moments ->
 * shape filter -> undistortPoints.  undist_xy / dist_xy hold 2*cap values. */
int mpe_find_leds(mpe_handle* h, const uint8_t* img, int rows, int cols, size_t stride_bytes,
                  int roi_x, int roi_y, int roi_w, int roi_h, const mpe_params* p,
                  const double K[9], const double* D, int nD, double* undist_xy, float* dist_xy,
                  int cap, int* n_out);

/* ≙ PoseEstimator::setImagePoints + initialise + optimiseAndUpdatePose on a fresh estimator
 * (lib/include/.../pose_estimator.h:425,749,773; lib/src/pose_estimator.cpp:80-91,544-721,
 * 733-812).  hist (n_det*n_markers) and corr (2*n_markers) are optional outputs. */
int mpe_solve_bruteforce(mpe_handle* h, const double* det_xy, int n_det, const double* markers_xyz,
                         int n_markers, const double K[9], const mpe_params* p, mpe_result* out,
                         uint32_t* hist, uint32_t* corr);
/* PoseEstimator::initialise() alone (pose_estimator.h:749, pose_estimator.cpp:544-721): voting,
 * correspondencesFromHistogram and checkCorrespondences WITHOUT the Gauss-Newton refinement — out->T is
 * the pose of computeTransformation, out->cov zero.  Same arguments as mpe_solve_bruteforce. */
int mpe_initialise(mpe_handle* h, const double* det_xy, int n_det, const double* markers_xyz, int n_markers,
                   const double K[9], const mpe_params* p, mpe_result* out, uint32_t* hist, uint32_t* corr);

/* ≙ PoseEstimator::estimateBodyPose (pose_estimator.h:366, pose_estimator.cpp:62-96) on a FRESH
 * estimator per frame (it_since_initialized_ == 0: whole-image detection + brute-force
 * initialisation) for a batch of independent frames — the measured entry point.
 * frames: n_frames images, frame f at frames + f*frame_stride_bytes, rows of stride_bytes.
 * frames_on_device != 0: `frames` is a device pointer (results is still a HOST pointer). */
int mpe_estimate_batch(mpe_handle* h, const uint8_t* frames, int n_frames, int rows, int cols,
                       size_t stride_bytes, size_t frame_stride_bytes, int frames_on_device,
                       const double* markers_xyz, int n_markers, const double K[9],
                       const double* D, int nD, const mpe_params* p, mpe_result* results);

/* ---- frame decode (SURVEY 8f "input side") -------------------------------------------------------------
 * ≙ cv_bridge::toCvCopy(image_msg, sensor_msgs::image_encodings::MONO8) (monocular_pose_estimator.cpp:147) for the
 * encodings a camera driver publishes: the payload of n_frames sensor_msgs/Image messages (rows x cols pixels, rows of
 * src_stride_bytes = Image.step, frame f at src + f * src_frame_stride_bytes) -> packed mono8 frames (rows x cols bytes
 * each, stride = cols), which is what every other entry point takes.  src / dst may each be a host or a device pointer.
 *   MPE_ENC_BGR8 / RGB8 / BGRA8 / RGBA8: cv::cvtColor(..., COLOR_{BGR,RGB,BGRA,RGBA}2GRAY) on CV_8U:
 *       Y = (B * 1868 + G * 9617 + R * 4899 + 2^13) >> 14   (integer, bit-exact)
 *       — the 14-bit weights of OpenCV 2.4 and 3.0 .. 3.4.1, the versions the reference's ROS Indigo / Kinetic ship.
 *       OpenCV >= 3.4.2 / 4.x use 15-bit weights (9798, 19235, 3735, >> 15) and can differ by one gray level: a caller
 *       linked against one of those keeps cv_bridge for colour frames (compat/ros/mpe_ros_glue.cpp does, unless built
 *       with -DMPE_OPENCV_GRAY_14BIT)
 *   MPE_ENC_MONO16: cv::Mat::convertTo(CV_8U, 255. / 65535.) = saturate_cast<uchar>((float)v * (float)(255. / 65535.)),
 *       after cv_bridge's byte swap when Image.is_bigendian differs from the host (src_big_endian)
 *   MPE_ENC_MONO8: a (strided) copy.
 * Bayer encodings (cv_bridge demosaics them through COLOR_Bayer*2GRAY) are not covered: MPE_ERR_UNSUPPORTED. */
#define MPE_ENC_MONO8 0
#define MPE_ENC_BGR8 1
#define MPE_ENC_RGB8 2
#define MPE_ENC_BGRA8 3
#define MPE_ENC_RGBA8 4
#define MPE_ENC_MONO16 5
int mpe_convert_to_mono8(mpe_handle* h, const void* src, int src_on_device, int encoding, int src_big_endian, int n_frames,
                         int rows, int cols, size_t src_stride_bytes, size_t src_frame_stride_bytes, uint8_t* dst,
                         int dst_on_device);

/* Page-locked host memory for frame buffers (what a camera driver / cv_bridge::toCvCopy target should write into,
 * monocular_pose_estimator.cpp:147): with HOST frames in pinned memory mpe_estimate_batch ingests a large batch
 * in chunks (option "ingest_chunk", default 2048 frames, 0 = one copy), the H2D copy of chunk c + 1 running
 * beside the kernels of chunk c — the call is then bound by the PCIe link alone.  NULL on failure. */
void* mpe_alloc_pinned(size_t bytes);
void mpe_free_pinned(void* p);

/* ---- the same entry point over SEVERAL GPUs of one node, from ONE host process ----
 * Frames are independent on this branch (pose_estimator.cpp:68-91 reads no estimator state when
 * it_since_initialized_ < 1), so the batch shards into contiguous chunks — shard d of n_dev gets frames
 * [lo, hi) as mpe_shard_bounds says (sizes differ by at most one frame) — with NO exchange step between the
 * devices; the only "gather" is that every shard's pose records land in the caller's ONE host array, in
 * frame order.  handles[d]: n_dev DISTINCT handles (normally created on n_dev different devices; several
 * handles on one device work too).  One host thread per shard drives that handle's mpe_estimate_batch.
 * Returns 0, or the first failing shard's error code (text: mpe_last_error of that shard's handle).
 * (One process per GPU + a pose gather over RCCL, as bench.py and rpg_monocular_pose_estimator_amd/parallel.py
 * do it, is the other way to use N GPUs; results are identical.) */
void mpe_shard_bounds(int n_frames, int shard, int n_shards, int* lo, int* hi);
/* host frames: frame f at frames + f*frame_stride_bytes, copied to the shard's device by its own thread */
int mpe_estimate_batch_multi(mpe_handle* const* handles, int n_dev, const uint8_t* frames, int n_frames,
                             int rows, int cols, size_t stride_bytes, size_t frame_stride_bytes,
                             const double* markers_xyz, int n_markers, const double K[9], const double* D,
                             int nD, const mpe_params* p, mpe_result* results);
/* device-resident shards: d_frames[d] = n_frames[d] packed frames (cols % 16 == 0, 16-byte aligned) in the
 * memory of handles[d]'s device; results = one HOST array of sum(n_frames) records, shard after shard */
int mpe_estimate_batch_multi_device(mpe_handle* const* handles, int n_dev, const uint8_t* const* d_frames,
                                    const int* n_frames, int rows, int cols, const double* markers_xyz,
                                    int n_markers, const double K[9], const double* D, int nD,
                                    const mpe_params* p, mpe_result* results);

/* The same with the gather ON THE DEVICES: every shard's pose records travel from its GPU into ONE device-resident
 * array on handles[0]'s device (sum(n_frames) records, shard after shard) — over RCCL (ncclCommInitAll once per device
 * list, one grouped ncclSend / ncclRecv exchange per call: point-to-point over xGMI, 432 bytes per frame, no host
 * copy) when the handles sit on distinct devices, by plain device-to-device copies when several handles share a
 * device.  *used_rccl (optional) tells which.  RCCL is loaded at first use (dlopen of librccl.so); MPE_ERR_UNSUPPORTED
 * if it cannot be found.  Blocking: the records are complete when the call returns. */
int mpe_estimate_batch_multi_device_gather(mpe_handle* const* handles, int n_dev, const uint8_t* const* d_frames,
                                           const int* n_frames, int rows, int cols, const double* markers_xyz,
                                           int n_markers, const double K[9], const double* D, int nD,
                                           const mpe_params* p, mpe_result* d_results_dev0, int* used_rccl);

/* Streaming entry, see mpe_estimate_batch_device_submit below: `hip_event` (a hipEvent_t, or NULL) marks the point at
 * which the frames the NEXT _submit call announces (d_next_frames) are final; one-shot, consumed by that _submit. */
int mpe_stream_next_ready(mpe_handle* h, void* hip_event);
/* ... and: forget what the last submission scanned ahead (the announced buffer has been rewritten since). */
int mpe_stream_drop_prefetch(mpe_handle* h);

/* Fully asynchronous variant for device-resident pipelines: frames AND results are device
 * pointers, nothing is copied, the call only enqueues kernels on the handle's stream.
 * frames must be 16-byte aligned with cols % 16 == 0, stride_bytes == cols and
 * frame_stride_bytes == rows*cols (the packed layout).
 * Ordering: the call behaves like ONE operation on the handle's stream — its kernels start after
 * everything enqueued on that stream before the call (large batches fork onto internal side
 * streams and join back) and d_results is complete for anything enqueued on it afterwards.  Work
 * on OTHER streams, including the legacy default stream 0, is not ordered with it: share a real
 * stream through mpe_set_stream (a NULL argument means the handle's own stream, NOT stream 0). */
int mpe_estimate_batch_device(mpe_handle* h, const uint8_t* d_frames, int n_frames, int rows,
                              int cols, const double* markers_xyz, int n_markers,
                              const double K[9], const double* D, int nD, const mpe_params* p,
                              mpe_result* d_results);

/* The same for a STREAM of device-resident batches (a camera pipeline that keeps frames coming): _submit only enqueues,
 * and it does not join the library's internal side streams back into the handle's stream — the records of the
 * submission are complete when the event behind _collect says so, not at a point of the handle's stream.  _collect
 * makes `hip_stream` (NULL = the handle's stream) wait for the records of the OLDEST un-collected submission; use the
 * stream that consumes them (a D2H copy, a pose gather) so that the next batch's kernels need not wait for it.
 * Up to two submissions may be in flight (submit k, submit k+1, collect k, ...); d_results must stay valid and
 * untouched until its submission has been collected and the consumer stream has passed that point.
 * d_next_frames / n_next_frames (optional): the frames of the NEXT submission (same geometry, camera, parameters).
 * The last voting launch of this submission then also scans the next one's first sub-batch, and the next _submit with
 * exactly those frames skips that scan — in steady state no kernel of a batch runs without an image scan beside it
 * (pose_estimator.cpp:62-96 has no counterpart: the reference handles one frame at a time).  A hint that does not come
 * true only costs the wasted scan.  mpe_estimate_batch_device = _submit without a hint + _collect on the handle's
 * stream; it refuses to run while a submission is un-collected.
 * CONTRACT for an announced buffer: its pixels are READ by this submission's last launches (on the handle's stream and
 * on an internal side stream), i.e. possibly long before the next _submit.  They must therefore (a) be final, in the
 * order of the handle's stream, when this _submit is called — or, if they are still being written on another stream
 * (the upload of batch k+1 while batch k runs), the caller passes the event that marks their completion through
 * mpe_stream_next_ready(h, event) right BEFORE this _submit: the reading launches then wait for it — and (b) stay
 * unmodified until the next _submit has been called.  A buffer that was rewritten in between (a double-buffered camera
 * pipeline that re-uses it) must be withdrawn with mpe_stream_drop_prefetch(h) before the next _submit, which then
 * scans it again; the library recognises a prefetched batch by POINTER and shape only and cannot see the rewrite. */
int mpe_estimate_batch_device_submit(mpe_handle* h, const uint8_t* d_frames, int n_frames, int rows, int cols,
                                     const double* markers_xyz, int n_markers, const double K[9], const double* D,
                                     int nD, const mpe_params* p, mpe_result* d_results,
                                     const uint8_t* d_next_frames, int n_next_frames);
int mpe_estimate_batch_device_collect(mpe_handle* h, void* hip_stream);

/* ≙ PoseEstimator::setCorrespondences + checkCorrespondences + optimiseAndUpdatePose
 * (pose_estimator.h:463,724,773; pose_estimator.cpp:394-542,733-812) — the tracking path's
 * validate-and-refine step for correspondences found by nearest neighbour.  corr: n_corr rows
 * (marker, detection), 1-based.  out->status: 0 pose, 1 correspondences rejected. */
int mpe_check_and_refine(mpe_handle* h, const double* det_xy, int n_det, const double* markers_xyz,
                         int n_markers, const double K[9], const mpe_params* p, const uint32_t* corr,
                         int n_corr, mpe_result* out);

/* The two halves of the call above as separate stage entry points (same kernel, other instantiations):
 * PoseEstimator::checkCorrespondences (pose_estimator.h:724, pose_estimator.cpp:394-542) — out->status 0
 * and out->T = the UNREFINED pose of computeTransformation when the correspondences validate, else 1;
 * PoseEstimator::optimisePose (pose_estimator.h:773, pose_estimator.cpp:733-792) — Gauss-Newton from
 * T_init on the given correspondences (no validation): out->T, out->cov, out->gn_iterations; status 1
 * when fewer than 3 correspondences were given (singular normal equations). */
int mpe_check_correspondences(mpe_handle* h, const double* det_xy, int n_det, const double* markers_xyz,
                              int n_markers, const double K[9], const mpe_params* p, const uint32_t* corr,
                              int n_corr, mpe_result* out);
int mpe_optimise_pose(mpe_handle* h, const double* det_xy, int n_det, const double* markers_xyz, int n_markers,
                      const double K[9], const mpe_params* p, const uint32_t* corr, int n_corr,
                      const double T_init[16], mpe_result* out);

/* ---- static primitives of the reference, batched (n independent problems, one GPU lane each) ----
 * P3P::computePoses (p3p.h:110, p3p.cpp:65-236): feature_vectors / world_points n x 9, point i of a
 * problem at [3i..3i+2] (= column i of the reference's 3x3 matrices); solutions n x 48 = 4 solutions
 * x 3x4 row-major [R|C]; status[i] 0, or -1 for collinear world points (solutions[i] untouched). */
int mpe_p3p_batch(mpe_handle* h, const double* feature_vectors, const double* world_points, int n,
                  double* solutions, int* status);
/* P3P::solveQuartic (p3p.h:127, p3p.cpp:238-286): factors n x 5 (highest power first), real_roots n x 4
 * (real parts of the four complex Ferrari roots, like the reference).  variant 0 = IEEE operators
 * (as the validation kernel uses it), 1 = the voting kernel's literal-order variant. */
int mpe_solve_quartic_batch(mpe_handle* h, const double* factors, int n, int variant, double* real_roots);

/* LEDDetector::determineROI (led_detector.h:105, led_detector.cpp:114-179): bounding box of the predicted
 * (undistorted) pixel positions, its two corners distorted, grown by border_size, clipped to the
 * image; the whole image when the box degenerates.  Host arithmetic (a dozen flops), no device. */
int mpe_determine_roi(const double* pixel_positions, int n_points, int rows, int cols, int border_size,
                      const double K[9], const double* D, int nD, int roi_xywh[4]);
/* LEDDetector::distortPoints (led_detector.h:135, led_detector.cpp:181-224), float in / float out. */
int mpe_distort_points(const float* src_xy, float* dst_xy, int n, const double K[9], const double* D, int nD);

/* One frame of the tracking branch as ONE device submission (pose_estimator.cpp:98-110 + 831-839):
 * LEDDetector::findLeds inside the ROI, then — when at least 4 LEDs were found — findCorrespondences
 * (nearest detection of every predicted marker pixel within nearest_neighbour_pixel_tolerance,
 * pose_estimator.cpp:372-392), checkCorrespondences and optimisePose.  img is a HOST image; the ROI is
 * packed into pinned memory and sent with the predicted pixels in one copy, the detections,
 * correspondences and pose come back in one copy.  predicted_px: n_markers x 2 (undistorted pixels).
 * dets_out: what findLeds found (always written); corr_out: 2*MPE_MAX_MARKERS uint32, rows
 * (marker, detection) 1-based, out->n_corr rows valid; out->status 0 = pose refined, 1 = fewer than
 * 4 LEDs or correspondences rejected (the caller then retries / re-initialises as the reference does),
 * <0 = a device capacity was exceeded.  mpe_tracker_estimate uses this for every tracked frame. */
int mpe_track_step(mpe_handle* h, const uint8_t* img, int rows, int cols, size_t stride_bytes, int roi_x,
                   int roi_y, int roi_w, int roi_h, const mpe_params* p, const double K[9], const double* D,
                   int nD, const double* markers_xyz, int n_markers, const double* predicted_px,
                   mpe_detections* dets_out, uint32_t* corr_out, mpe_result* out);

/* ---- lock-step batches over N independent camera streams (BASELINE configs[4]) ----------------------
 * mpe_track_step for frame k of N streams in ONE device submission: every stream's ROI goes into one slot of
 * a uniform slot array (zero beyond the ROI; the blob kernels read the window size / origin per slot, so image
 * borders and centroid offsets are those of the stand-alone clone of led_detector.cpp:44), one image scan, one
 * blob extraction and one validate / refine kernel pair run over the N slots, one copy returns N records.
 * All streams share rows / cols / stride, camera, marker set and parameters.  predicted_px == NULL for an
 * item = detection only (out[i].status = 1).  Group items of very different ROI size into separate calls:
 * the slot is as large as the largest ROI of the call.  dets_out n, corr_out n x 2*MPE_MAX_MARKERS, out n. */
typedef struct mpe_track_item {
  const uint8_t* img;          /* HOST image of this stream */
  int roi_x, roi_y, roi_w, roi_h;
  const double* predicted_px;  /* n_markers x 2 undistorted pixels, or NULL */
} mpe_track_item;
int mpe_track_step_batch(mpe_handle* h, const mpe_track_item* items, int n, int rows, int cols, size_t stride_bytes,
                         const mpe_params* p, const double K[9], const double* D, int nD,
                         const double* markers_xyz, int n_markers, mpe_detections* dets_out, uint32_t* corr_out,
                         mpe_result* out);
/* The same in two halves, for callers that overlap the host work of one group of streams with the device work of
 * another: _submit packs the ROIs and enqueues copy-in, kernels and copy-out on the handle's stream WITHOUT waiting;
 * _collect waits and hands out the records of that submission (one outstanding submission per handle). */
int mpe_track_step_batch_submit(mpe_handle* h, const mpe_track_item* items, int n, int rows, int cols,
                                size_t stride_bytes, const mpe_params* p, const double K[9], const double* D, int nD,
                                const double* markers_xyz, int n_markers);
int mpe_track_step_batch_collect(mpe_handle* h, mpe_detections* dets_out, uint32_t* corr_out, mpe_result* out);
/* _collect without a submission in flight is an error (MPE_ERR_ARG: a caller that missed a failed _submit must not read
 * records that were never written).  _cancel abandons the submission in flight, if any: it waits for the device and
 * frees the handle for the next _submit / mpe_track_step (used on error paths that drive several handles). */
int mpe_track_step_batch_cancel(mpe_handle* h);
/* mpe_solve_bruteforce for N detection sets in one submission (the re-initialisations of a lock-step batch):
 * det_xy n x MPE_MAX_DETECTIONS x 2 (n_det[i] valid rows); hist (optional) n x MPE_MAX_DETECTIONS x
 * MPE_MAX_MARKERS, corr (optional) n x 2*MPE_MAX_MARKERS. */
int mpe_solve_bruteforce_batch(mpe_handle* h, const double* det_xy, const int* n_det, int n,
                               const double* markers_xyz, int n_markers, const double K[9], const mpe_params* p,
                               mpe_result* out, uint32_t* hist, uint32_t* corr);

/* ---- stateful estimator: the whole PoseEstimator::estimateBodyPose state machine, i.e. the
 * uninitialised branch AND the tracking path (pose_estimator.cpp:62-147): pose prediction by the
 * constant-velocity model (predictPose :232-244), ROI from the predicted LED pixels
 * (LEDDetector::determineROI, led_detector.cpp:114-179), ROI detection with whole-image retry,
 * nearest-neighbour correspondences (findCorrespondences :372-392), validation + refinement,
 * fallback to brute-force initialisation (:831-848).  One tracker = one PoseEstimator object. */
typedef struct mpe_tracker mpe_tracker;
int mpe_tracker_create(mpe_handle* h, mpe_tracker** out);              /* PoseEstimator() */
void mpe_tracker_destroy(mpe_tracker* t);
int mpe_tracker_set_markers(mpe_tracker* t, const double* xyz, int n); /* setMarkerPositions */
int mpe_tracker_set_camera(mpe_tracker* t, const double K[9], const double* D, int nD);
int mpe_tracker_set_params(mpe_tracker* t, const mpe_params* p);       /* tuning members / setters */
int mpe_tracker_reset(mpe_tracker* t); /* back to "not initialised" (the reference has no reset) */
/* estimateBodyPose(image, time_to_predict): returns 1 pose updated, 0 not, <0 error.  out (optional)
 * gets getPredictedPose / getPoseCovariance; info (optional, 8 ints) = region_of_interest_
 * x,y,w,h, it_since_initialized_, detections, correspondences, 1 if brute force ran. */
int mpe_tracker_estimate(mpe_tracker* t, const uint8_t* img, int rows, int cols, size_t stride_bytes,
                         double time_to_predict, mpe_result* out, int info[8]);

/* estimateBodyPose for frame k of N trackers (N independent streams of the same camera model, marker set and
 * parameters, all created on the SAME handle) in lock step: the per-stream state machines run on the host as in
 * mpe_tracker_estimate, but every device step is ONE batched submission over the streams that need it
 * (mpe_track_step_batch for the ROI / whole-image detections with their validate + refine,
 * mpe_solve_bruteforce_batch for the re-initialisations): in steady state one submission per time step for all
 * N streams.  imgs[i], times[i]: stream i's frame and time stamp; out (optional) n records, info (optional)
 * n x 8 ints, updated (optional) n flags.  Results are identical to calling mpe_tracker_estimate per stream.
 * Returns the number of streams whose pose was updated, or <0 (usage / HIP error; a per-stream capacity overrun
 * is reported in out[i].status and that stream is left as its failed call leaves it). */
int mpe_tracker_estimate_batch(mpe_tracker* const* trackers, int n, const uint8_t* const* imgs, int rows, int cols,
                               size_t stride_bytes, const double* times, mpe_result* out, int* info, int* updated);

/* The image-callback loops of N streams in lock step over recorded sequences: frames[i] = stream i's sequence
 * (frame f at frames[i] + f*frame_stride_bytes), times[f] the common time stamps; out / info (optional):
 * n x n_frames records / n x n_frames x 8 ints, stream-major.  Returns the number of pose updates or <0.
 * The trackers may live on SEVERAL handles (same device or not): each handle's trackers form one lock-step group
 * (same camera / markers / parameters within a group), and the groups are pipelined against each other — while
 * the device works on step k of one group the host collects, advances and packs another — which hides the host
 * work of a time step behind the device latency.  Results do not depend on the grouping. */
int mpe_tracker_run_sequences_batch(mpe_tracker* const* trackers, int n, const uint8_t* const* frames, int n_frames,
                                    int rows, int cols, size_t stride_bytes, size_t frame_stride_bytes,
                                    const double* times, mpe_result* out, int* info);
/* The same with the groups spread over up to n_threads host threads (group g on thread g % n_threads; the groups of
 * one thread are pipelined against each other as above).  Groups share nothing — own handle and stream, own trackers,
 * own rows of out / info — so the records are those of the single-threaded call; with 64 streams in 4-8 groups the
 * host work of a time step no longer bounds the rate.  n_threads = 1: the call above. */
int mpe_tracker_run_sequences_batch_threads(mpe_tracker* const* trackers, int n, const uint8_t* const* frames,
                                            int n_frames, int rows, int cols, size_t stride_bytes,
                                            size_t frame_stride_bytes, const double* times, mpe_result* out, int* info,
                                            int n_threads);

/* The estimator's private state (pose_estimator.h:56-62, 74-79), for callers that drive the public
 * step methods of the class (predictPose, findCorrespondences, ... — see compat/) between calls of
 * mpe_tracker_estimate.  Poses row-major 4x4. */
typedef struct mpe_tracker_state {
  double current_pose[16], previous_pose[16], predicted_pose[16];
  double pose_covariance[36];
  double current_time, previous_time, predicted_time;
  unsigned it_since_initialized;
  int roi[4]; /* region_of_interest_ x, y, width, height */
} mpe_tracker_state;
int mpe_tracker_get_state(const mpe_tracker* t, mpe_tracker_state* st);
int mpe_tracker_set_state(mpe_tracker* t, const mpe_tracker_state* st);

/* Host arithmetic of the state machine, exported so that the facade and the tracker share ONE
 * implementation (no device involved): predictPose (pose_estimator.cpp:232-244: constant-velocity
 * extrapolation through logarithmMap / exponentialMap), exponentialMap (:962-994), logarithmMap
 * (:996-1064), project2d for a list of marker positions (:251-276; markers n x 3, px n x 2). */
int mpe_predict_pose(const double current_pose[16], const double previous_pose[16], double current_time,
                     double previous_time, double time_to_predict, double predicted_pose[16]);
int mpe_exponential_map(const double twist[6], double T[16]);
int mpe_logarithm_map(const double T[16], double twist[6]);
int mpe_project_points(const double T[16], const double* markers_xyz, int n, const double K[9], double* px);
/* findCorrespondences (pose_estimator.cpp:372-392): corr gets up to n_markers rows (marker, detection),
 * 1-based; returns the number of rows (>= 0) or MPE_ERR_ARG. */
int mpe_find_correspondences(const double* predicted_px, int n_markers, const double* det_xy, int n_det,
                             double nearest_neighbour_pixel_tolerance, uint32_t* corr);

/* The image callback loop (MPENode::imageCallback -> estimateBodyPose, monocular_pose_estimator.cpp:
 * 125-190) over a recorded sequence: frame f at frames + f*frame_stride_bytes with time stamp
 * times[f].  out (optional) n_frames records, info (optional) n_frames x 8 ints as above.  Returns
 * the number of frames whose pose was updated, or <0 on the first usage / HIP error.  A frame that exceeds
 * a device capacity (MPE_FRAME_TOO_MANY_*) does not end the sequence: its record is zeroed, out[f].status
 * holds the code, the estimator state is as that frame's failed call left it, and the replay continues. */
int mpe_tracker_run_sequence(mpe_tracker* t, const uint8_t* frames, int n_frames, int rows, int cols,
                             size_t stride_bytes, size_t frame_stride_bytes, const double* times,
                             mpe_result* out, int* info);

/* getCorrespondences() / getImagePoints() of the tracker object: copy up to cap rows / points,
 * return the number available (or <0). */
int mpe_tracker_get_correspondences(mpe_tracker* t, uint32_t* corr, int cap_rows);
int mpe_tracker_get_image_points(mpe_tracker* t, double* xy, int cap_points);
/* distorted_detection_centers_ of the last detection (what the overlay circles, pose_estimator.cpp:44-48) */
int mpe_tracker_get_distorted_centers(mpe_tracker* t, float* xy, int cap_points);

/* Stage-level batch entry points (used by the parity tests at every stage boundary). */
/* detection only (a1): dets is a HOST array of n_frames records */
int mpe_detect_batch(mpe_handle* h, const uint8_t* frames, int n_frames, int rows, int cols,
                     size_t stride_bytes, size_t frame_stride_bytes, int frames_on_device,
                     const double K[9], const double* D, int nD, const mpe_params* p,
                     mpe_detections* dets);
/* voting only (a4, PE.cpp:544-702): det_xy is n_frames x MPE_MAX_DETECTIONS x 2 (host), n_det[f]
 * valid rows each; hist is n_frames x MPE_MAX_DETECTIONS x MPE_MAX_MARKERS (host) */
int mpe_vote_batch(mpe_handle* h, const double* det_xy, const int* n_det, int n_frames,
                   const double* markers_xyz, int n_markers, const double K[9],
                   double back_projection_pixel_tolerance, uint32_t* hist);

/* Forensics for the parity soaks (tests/forensics.py): the voting of mpe_vote_batch restricted, per detection set, to
 * the hypotheses [item_lo[f], item_hi[f]) of initialise()'s loop nest — flattened index = detection-triple index x
 * P(n_markers,3) + marker-permutation index, both in the reference's table order (combinations.cpp:52-203,
 * pose_estimator.cpp:565-600).  With n copies of one detection set and ranges [i, i+1) it returns every hypothesis'
 * own votes, which is how a HIP-vs-oracle histogram difference is traced to the hypotheses that cast it.  Same
 * kernels and arithmetic as mpe_vote_batch (option "vote_arith" applies). */
int mpe_vote_items(mpe_handle* h, const double* det_xy, const int* n_det, int n_frames, const double* markers_xyz,
                   int n_markers, const double K[9], double back_projection_pixel_tolerance, const int* item_lo,
                   const int* item_hi, uint32_t* hist);

/* Time (ms) of the kernels of the last mpe_estimate_batch* call, measured with HIP events on the
 * handle's stream: [0] image scan, [1] blob extraction, [2] voting, [3] validate+refine, [4] total.
 * Only valid after mpe_set_profiling(h, 1); profiling adds event records to the stream. */
int mpe_set_profiling(mpe_handle* h, int enable);
int mpe_last_kernel_ms(mpe_handle* h, float ms[5]);
/* A large batch runs as several sub-batches (option "pipeline"): every kernel is then launched once
 * per sub-batch and mpe_last_kernel_ms reports the AVERAGE PER LAUNCH.  This returns the number of
 * launches per kernel and the frames each one processed for the last profiled call. */
/* Same for ONE sub-batch of a pipelined call (0 <= sub_batch < launches): scan, blobs, vote, tail.  With
 * pipeline_mode 3 ("fused") the scan of sub-batch s + 1 runs INSIDE the voting kernel of sub-batch s: the
 * vote time of every sub-batch but the last then includes that scan, and scan = the stand-alone scan
 * launches only (the whole first sub-batch, afterwards only remainders of less than one chunk). */
int mpe_last_kernel_ms_sub(mpe_handle* h, int sub_batch, float ms[4]);
int mpe_last_launch_shape(mpe_handle* h, int* launches, int* frames_per_launch);

/* Tuning knobs that are not part of the reference surface: "lds_budget" (bytes of LDS per frame
 * for the blob bitmaps, 8192..163840), "vote_splits" (workgroups per frame in the voting kernel,
 * 0 = auto), "pipeline" (a large call is cut into up to this many sub-batches of about 32768 frames — 16384 for frames larger than 512 KB —,
 * default 16, 1 = one chain of four kernels), "pipeline_mode" (how the sub-batches are scheduled:
 * -1 automatic (default) = 6;  0 = two-stream software
 * pipeline, the scan of sub-batch s+1 beside the voting of sub-batch s;  3 = fused, one stream: the
 * scan of sub-batch s+1 rides inside the voting kernel of sub-batch s;  4 = as 3, with the validate /
 * refine kernels of sub-batch s on an internal side stream, beside the blob extraction of sub-batch
 * s+1 (joined back before the call returns its place on the stream);  6 = as 4, and the image scan of a
 * sub-batch is split: "scan_split_pct" % of it (default 30) is taken by a stand-alone scan kernel on a second
 * side stream during the blob / tail window two sub-batches earlier ("side_scan_blocks" resident blocks per CU,
 * default 3), the voting kernel's rider scans the rest), "k1a_dummy_lds" (occupancy cap
 * of the stand-alone scan kernel in mode 0, per handle), "ingest_chunk" (frames per chunk of the double-buffered
 * host-frame ingest of mpe_estimate_batch, default 2048, 0 = one blocking copy per call), "refine_variant" (the refinement kernel: 0 automatic = 16 lanes per frame for launches of up to 2048 frames, else one lane per frame; 1 / 2 force one of them; bit-identical results), "vote_arith" (arithmetic of the voting kernel:
 * 3 (default since round 6) = fast — Newton-Raphson division / square root, Newton cube root, per-permutation
 * tables, [R|C]-free back-projection — with every hypothesis it cannot decide safely re-evaluated by the strict
 * functions, whose quartic evaluates std::pow(complex, double) of p3p.cpp:262,264,268 as libstdc++ / glibc do
 * (exp(y log|z|) through clog's branches; csrc/mpe_ddmath.h): the vote histograms of the CPU reference build also in
 * the unstable corner of its Ferrari solver; 1 = the same with exact products / cbrt(hypot)
 * for those powers (default of rounds 4 - 5); 4 / 0 = strict — the validation kernel's P3P functions with IEEE
 * operators in the reference's statement order for every hypothesis, powers as in 3 / 1; slower, never fused with
 * the scan; 2 = the fast arithmetic alone (A/B measurements).  3 and 4 produce identical histograms, as do 1 and 0.
 * Results are bit-identical in every pipeline mode (for a given vote_arith). */
int mpe_set_option(mpe_handle* h, const char* name, int value);
/* Measurement read-outs through the same pair of calls:
 * set "vote_events" = N > 0: a pair of timing events is recorded around every voting launch that carries a scan, for
 *   the launches of the last N pipelined calls — nothing else, so a timed region stays what it is; get
 *   "vote_launch_ns_mean" / "vote_launches" (synchronises the handle's stream); set 0 to release the events.
 * set "track_profile" = 1 starts / resets host-side timers inside mpe_track_step; get "track_ns_pack",
 *   "track_ns_enqueue", "track_ns_wait" (mean ns per step), "track_steps".
 * get "overflow_frames", "overflow_general", "overflow_why_1" .. "overflow_why_6": frames of the last pipelined batch
 *   that the first blob tier handed on, in all / to the general tier / by the capacity exceeded (bright segments,
 *   bands, islands, pixel pool, bitmap pool, blobs kept); synchronises.
 * get "vote_fixup_items" / "vote_fixup_overflow" / "vote_relost_frames": hypotheses the fast voting kernel handed to
 *   the strict arithmetic since the handle was made, appends that found their list full, and frames that were therefore
 *   voted again by the strict loop nest (a full list costs time, never a pose); synchronises.
 * Tuning / test knobs (round 5): "vote_list_cap" (entries per suspect list at most, 0 = no limit: a tiny list exercises
 *   the re-vote path), "k1b_general_blocks" (PROCESS-wide: blocks = scratch slabs of the general blob tier, 32 .. 8192,
 *   default 4096, within 1 GB of scratch, never more than the frames of a launch nor than the blocks the device holds
 *   at once — 4 096 on an MI355X: a block walks the work-list with the grid as its stride), "tail_priority" /
 *   "scan_priority" (stream priority of the library's two side streams, applied when they are created: -1 lowest,
 *   0 ordinary streams, 1 highest, 2 the default level but created through the priority entry point; default 1 / 1 —
 *   off the default level so that they get hardware queues of their own, DESIGN.md section 3, Schedules.  NOTE for
 *   callers: at 1 the side streams' kernels are dispatched ahead of work on the caller's own ordinary-priority
 *   streams of the same process while a pipelined call is in flight; set both to -1 to keep the queues and yield
 *   instead — same step time within noise, profiles/round5_exp_side_priorities.json).
 * "detections_hint" (round 6): detections per frame the caller expects in device-resident / streaming calls (0 =
 *   automatic: the number of markers, or what the last call whose records came back to the host saw — read-out
 *   "detections_seen").  From 9 on the <= 5-marker voting kernel prefilters its back-projections with an occupancy
 *   grid of the detections instead of a distance per detection; the suspect lists are sized from it.  Results do not
 *   depend on it.
 * "track_fused" (round 6): 2 (default) a tracked frame / lock-step time step is ONE launch (k_track_frame) that also
 *   stores the records to pinned host memory; 1 the same with a copy command for the records; 0 the chain of four
 *   kernels + two copies of rounds 3 - 5.  Bit-identical records.  "track_phase_clocks" = 1: that kernel stamps its
 *   phases; get "track_phase_cycles_0" .. "_3" (scan, blobs, validation, refinement: mean shader-clock cycles).
 * "general_lds" (round 6): kernel of the general blob tier — 0 (default) bitmaps in global-memory slabs, 16 frames per
 *   CU in flight, (band, column run, row piece) items of any number per frame for frames up to 3 966 pixels wide; 1 a
 *   block per CU with the frame's bitmaps in LDS (faster per frame, one frame per CU: 2 x slower on
 *   salt noise, 7 % faster on one large blob); -1 the LDS kernel once the previous call saw frames reach the tier
 *   (read-out "general_seen").  Bit-identical detections.
 * get "vote_wide_frames" (round 6): frames with more than MPE_FAST_VOTE_DETECTIONS detections, voted by the strict
 *   loop nest alone; synchronises. */
/* Read an option back.  Also "streams_concurrent": 1 once the library has verified (spin-kernel probe at
 * the first large batch) that its two pipeline side streams execute concurrently, 0 if no concurrent
 * pair was found (the runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues), -1 not probed yet;
 * "last_schedule": the pipeline_mode the last large batch actually ran with. */
int mpe_get_option(mpe_handle* h, const char* name, int* value);

/* library / device introspection */
int mpe_device_count(void);
const char* mpe_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MPE_H_ */
