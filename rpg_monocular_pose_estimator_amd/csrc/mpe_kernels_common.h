#ifndef MPE_KERNELS_COMMON_H_
#define MPE_KERNELS_COMMON_H_
// mpe_kernels_common.h — what the kernel translation units share.  The hand-written gfx950 (CDNA4, wave64) kernels of
// the per-frame hot path live in mpe_k1.hip (image scan, blob extraction), mpe_k2.hip (voting; its head mpe_k2_head.h is
// shared with the tail) and mpe_k3.hip (validate / refine, primitive batches, frame decode), one object file each
// behind the same libmpe_hip.so (round 5: an experiment on one kernel no longer rebuilds the other three).  Read in
// the order common -> k1 -> k2_head -> k2 -> k3 they are the former single file mpe_kernels.hip, which the CPU tier
// still treats as one text (binding.device_source: the marked file prologues / epilogues dropped).
//
//
//   K1a  k1a_scan   : the image pass.  Streams the uint8 batch from HBM once, 16 B per lane
//                     (1 KiB per wave-instruction), compares against the TOZERO threshold with
//                     SWAR byte arithmetic and emits ONE flag bit per 16-byte segment via a
//                     wave ballot (8 B written per 1 KiB read).  HBM-bandwidth bound.
//   K1b  k1b_blobs  : one wave per frame.  From the flag bits: activates the few image rows near
//                     bright pixels, computes the exact fixed-point Gaussian blur mask for them
//                     into LDS bitmaps, then reproduces OpenCV's findContours(RETR_EXTERNAL)
//                     raster scan + Suzuki border following on the bitmaps with polygon
//                     area/moments accumulated on the fly (int64, exact), shape filter, float32
//                     centroid and undistortPoints.  (reference: led_detector.cpp:35-112)
//   K2   k2_vote    : one workgroup per frame; every (detection triple, marker permutation) P3P
//                     problem is one work item; FP64 Kneip P3P + reprojection voting with LDS
//                     integer atomics.  (reference: pose_estimator.cpp:544-702)
//                     k2_vote<true> additionally carries the image scan of the NEXT sub-batch on
//                     its idle memory pipeline (ScanRider: global_load_lds LDS-DMA rounds served
//                     between pieces of P3P arithmetic) — the default schedule for <= 5 markers.
//   K3a  k3a_validate : 16 lanes per frame; histogram peeling, the C(n_c,3) P3P validations of
//                     checkCorrespondences summed in combination order.  (pose_estimator.cpp:344-370, 394-542)
//   K3b  k3b_refine : one lane per frame; Kabsch, Gauss-Newton refine + covariance.
//                     (pose_estimator.cpp:733-792, 908-994)
//
// FP64 everywhere on the geometry path (the reference is double; vote thresholds are knife
// edges), no MFMA (no dense contraction on this path), compiled with -ffp-contract=off.
#include <type_traits>

#include "mpe_internal.h"
#include "mpe_p3p.h"

namespace mpe {

typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Synchronisation among the 64 lanes of ONE wave that communicate through LDS (the front phases of a frame
// belong to one wave; a block barrier there would couple the data-dependent control flow of the block's waves).
// DS operations of a wave execute in program order, so all that is needed is that the compiler keeps that order.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// =============================================================================================
// K1a — image scan
// =============================================================================================
// "any of the 16 bytes > thr" with 3 VALU ops per 32-bit word.  For thr >= 128 a byte exceeds thr iff its
// top bit is set AND its low 7 bits exceed thr - 128; for thr < 128 iff the top bit is set OR the low 7 bits
// exceed thr.  "low 7 bits > n" is the classic SWAR carry test: (b & 0x7F) + (127 - n) sets bit 7 (no carry
// leaves the byte).  kk = (127 - n) * 0x01010101, sel = ~0 for the AND form, 0 for the OR form; the select
// t&w / t|w is one v_bitop3_b32 on gfx950.  thr = 255 -> AND form with kk = 0 (never), thr = -1 -> OR form with
// kk = 128 * 0x01010101 (always).
struct ThrTest {
  unsigned kk, sel;
};
__host__ __device__ inline ThrTest make_thr_test(int thr) {
  const int t = thr < -1 ? -1 : (thr > 255 ? 255 : thr);
  ThrTest r;
  if (t >= 128) {
    r.kk = (unsigned)(255 - t) * 0x01010101u;
    r.sel = 0xFFFFFFFFu;
  } else {
    r.kk = (unsigned)(127 - t) * 0x01010101u;
    r.sel = 0u;
  }
  return r;
}
__device__ __forceinline__ unsigned gt_word(unsigned w, unsigned kk, unsigned sel) {
  const unsigned t = (w & 0x7F7F7F7Fu) + kk;
  return (sel & (t & w)) | (~sel & (t | w));
}
// Cheap necessary condition: the bytewise OR of the four words is >= every byte, so if no byte of the OR
// exceeds thr none of the 16 does (no false negative; a hit is confirmed with any_gt16).
__device__ __forceinline__ unsigned maybe_gt16(const uint4& v, ThrTest q) {
  return gt_word(v.x | v.y | v.z | v.w, q.kk, q.sel) & 0x80808080u;
}
__device__ __forceinline__ unsigned any_gt16(const uint4& v, ThrTest q) {
  const unsigned r = gt_word(v.x, q.kk, q.sel) | gt_word(v.y, q.kk, q.sel) | gt_word(v.z, q.kk, q.sel) |
                     gt_word(v.w, q.kk, q.sel);
  return r & 0x80808080u;
}

// The threshold test with the AND / OR form fixed at compile time (3 VALU ops per word instead of the 5 of the
// run-time select): the kernels branch ONCE, wave-uniformly, on thr.sel.
template <bool HI>
__device__ __forceinline__ unsigned gt_word_c(unsigned w, unsigned kk) {
  const unsigned t = (w & 0x7F7F7F7Fu) + kk;
  return HI ? (t & w) : (t | w);
}
template <bool HI>
__device__ __forceinline__ unsigned maybe_gt16_c(const uint4& v, unsigned kk) {
  return gt_word_c<HI>(v.x | v.y | v.z | v.w, kk) & 0x80808080u;
}
template <bool HI>
__device__ __forceinline__ unsigned any_gt16_c(const uint4& v, unsigned kk) {
  return (gt_word_c<HI>(v.x, kk) | gt_word_c<HI>(v.y, kk) | gt_word_c<HI>(v.z, kk) | gt_word_c<HI>(v.w, kk)) & 0x80808080u;
}

//@file-epilogue
}  // namespace mpe
#endif
//@file-epilogue-end
