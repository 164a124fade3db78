// mpe_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the per-frame hot path.
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

#ifndef K1A_UNROLL
#define K1A_UNROLL 8  // 8 KiB per wave in flight: measured best on MI355X (6.5 TB/s)
#endif
#ifndef K1A_NT
#define K1A_NT 1
#endif
#ifndef K1A_BLOCKS_PER_CU
#define K1A_BLOCKS_PER_CU 32
#endif
// One chunk of 64 * K1A_UNROLL segments of a wave.  FULL: the whole chunk lies inside the data — no bounds checks, one
// base address per chunk, the loads differ by their immediate offsets only (round 3: the per-load 64-bit bounds check,
// zero fill and address arithmetic were 9 of the 18 VALU instructions per KiB; with the compile-time threshold form
// the full-chunk path is down to ~8, which matters because the scan shares the chip's VALU with the blob extraction).
template <bool HI, bool FULL>
__device__ __forceinline__ void k1a_chunk(const uint4* __restrict__ px, u64* __restrict__ flags, size_t n_seg, size_t c,
                                          int lane, unsigned kk) {
  const size_t base = c * (64 * K1A_UNROLL) + lane;
  uint4 v[K1A_UNROLL];
  const u32x4* p = reinterpret_cast<const u32x4*>(px) + base;
#pragma unroll
  for (int k = 0; k < K1A_UNROLL; ++k) {
    if (FULL || base + 64 * k < n_seg) {
#if K1A_NT
      const u32x4 t = __builtin_nontemporal_load(p + 64 * k);
#else
      const u32x4 t = p[64 * k];
#endif
      v[k] = make_uint4(t.x, t.y, t.z, t.w);
    } else {
      v[k] = make_uint4(0, 0, 0, 0);
    }
  }
  u64 b[K1A_UNROLL];
#pragma unroll
  for (int k = 0; k < K1A_UNROLL; ++k) {
    b[k] = __ballot(maybe_gt16_c<HI>(v[k], kk) != 0);
    if (b[k]) b[k] = __ballot(any_gt16_c<HI>(v[k], kk) != 0);  // wave-uniform, rare on dark frames
  }
  if (lane == 0) {
    ulonglong2* out = reinterpret_cast<ulonglong2*>(flags + c * K1A_UNROLL);
#pragma unroll
    for (int k = 0; k < K1A_UNROLL / 2; ++k) out[k] = make_ulonglong2(b[2 * k], b[2 * k + 1]);
  }
}
template <bool HI>
__device__ __forceinline__ void k1a_scan_body(const uint4* __restrict__ px, u64* __restrict__ flags, size_t n_seg, unsigned kk) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
  const size_t n_full = n_seg / (64 * K1A_UNROLL);
  const size_t n_chunks = (n_seg + 64 * K1A_UNROLL - 1) / (64 * K1A_UNROLL);
  for (size_t c = wave; c < n_chunks; c += n_waves) {
    if (c < n_full)
      k1a_chunk<HI, true>(px, flags, n_seg, c, lane, kk);
    else
      k1a_chunk<HI, false>(px, flags, n_seg, c, lane, kk);
  }
}
__global__ __launch_bounds__(256) void k1a_scan(const uint4* __restrict__ px, u64* __restrict__ flags, size_t n_seg,
                                                ThrTest thr) {
  if (thr.sel)  // (wave-uniform: a kernel argument)
    k1a_scan_body<true>(px, flags, n_seg, thr.kk);
  else
    k1a_scan_body<false>(px, flags, n_seg, thr.kk);
}

// dummy_lds > 0: the scan is about to run beside the FP64 voting kernel of another sub-batch (two-stream
// schedule).  It is then capped to 4 blocks = 4 waves per SIMD (via an otherwise unused dynamic LDS allocation,
// 40 KB per block by default) so that its 44-VGPR waves leave room for the voting waves; HBM throughput is
// unchanged at that occupancy (measured).  The value is a per-handle setting (option "k1a_dummy_lds").
// Compute units of the CURRENT device (hipDeviceProp_t::multiProcessorCount, cached per device): 256 on a whole
// MI355X, 32 on one partition in CPX mode — "one resident block per CU" must mean that on either.
int device_cu_count() {
  static int cached[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] > 0) return cached[dev];
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    n = 256;
  }
  cached[dev] = n;
  return n;
}

hipError_t launch_k1a_scan(const uint8_t* frames, size_t n_bytes, unsigned long long* flags, int thr,
                           int dummy_lds_bytes, hipStream_t s, int blocks_per_cu) {
  const size_t n_seg = n_bytes / 16;
  if (n_seg == 0) return hipSuccess;
  const ThrTest q = make_thr_test(thr);
  const size_t n_chunks = (n_seg + 64 * K1A_UNROLL - 1) / (64 * K1A_UNROLL);
  size_t blocks = (n_chunks + 3) / 4;  // 4 waves per block
  // CUs x blocks per CU, grid-stride beyond.  blocks_per_cu > 0: a deliberately small resident set (a side scan
  // that must leave the wave slots to the kernel it runs beside)
  const size_t max_blocks = (size_t)device_cu_count() * (size_t)(blocks_per_cu > 0 ? blocks_per_cu : K1A_BLOCKS_PER_CU);
  if (blocks > max_blocks) blocks = max_blocks;
  const size_t dummy_lds = dummy_lds_bytes > 0 ? (size_t)dummy_lds_bytes : 0;
  if (dummy_lds > 65536) {  // tuning experiments only: more than the default dynamic-LDS limit
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k1a_scan), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)dummy_lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k1a_scan, dim3((unsigned)blocks), dim3(256), dummy_lds, s,
                     reinterpret_cast<const uint4*>(frames), (u64*)flags, n_seg, q);
  return hipGetLastError();
}

// =============================================================================================
// repack — copy an ROI of strided frames into the packed layout (pitch % 16 == 0, zero padded).
// Used for host frames with odd strides / widths and for ROI detection (the reference clones the
// ROI into a stand-alone matrix, led_detector.cpp:44).
// =============================================================================================
__global__ void repack_kernel(const uint8_t* __restrict__ src, size_t src_stride, size_t src_frame_stride, int n_frames,
                              int roi_x, int roi_y, int roi_w, int roi_h, uint8_t* __restrict__ dst, int dst_pitch) {
  const int y = blockIdx.y;
  for (int f = blockIdx.z; f < n_frames; f += gridDim.z) {  // gridDim.z is limited to 65535: stride over the frames
    const uint8_t* s = src + (size_t)f * src_frame_stride + (size_t)(roi_y + y) * src_stride + roi_x;
    uint8_t* d = dst + ((size_t)f * roi_h + y) * dst_pitch;
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < dst_pitch; x += gridDim.x * blockDim.x)
      d[x] = (x < roi_w) ? s[x] : (uint8_t)0;
  }
}

hipError_t launch_repack(const uint8_t* src, size_t src_stride, size_t src_frame_stride, int n_frames, int roi_x,
                         int roi_y, int roi_w, int roi_h, uint8_t* dst, int dst_pitch, hipStream_t s) {
  if (n_frames <= 0 || roi_h <= 0) return hipSuccess;
  dim3 grid((dst_pitch + 255) / 256, roi_h, n_frames < 65535 ? n_frames : 65535);
  hipLaunchKernelGGL(repack_kernel, grid, dim3(256), 0, s, src, src_stride, src_frame_stride, n_frames, roi_x, roi_y,
                     roi_w, roi_h, dst, dst_pitch);
  return hipGetLastError();
}

// =============================================================================================
// K1b — blob extraction, one wave per frame, small LDS footprint (high occupancy)
//
// The flag bits give the bright 16-byte segments of the frame.  Rows within +-r of a bright
// segment form BANDS (maximal runs of such rows); inside a band the occupied segment columns
// (dilated by the blur reach) split into ISLANDS.  Blurred-mask components can neither cross an
// inactive row nor an empty column run, and an island cannot lie inside a hole of another island
// (disjoint bounding boxes), so OpenCV's raster scan decomposes exactly: every island is scanned
// on its own in a small LDS window (thresholded pixels + three bitmaps), and the blobs are put
// back into raster order of their start pixels at the end.
// =============================================================================================
struct BlobRec {
  long long a00, a10, a01;  // polygon sums: sum dxy, sum dxy*(x_{i-1}+x_i), sum dxy*(y_{i-1}+y_i)
  int xmin, xmax, ymin, ymax;
};
#ifndef K1B_ON_BLOBREC
#define K1B_ON_BLOBREC(rec, key)  // (the CPU tier records the raw contour sums here, from both contour phases)
#endif

__device__ __forceinline__ int reflect101(int p, int len) {  // cv::borderInterpolate(BORDER_REFLECT_101)
  if ((unsigned)p < (unsigned)len) return p;
  if (len == 1) return 0;
  do {
    p = (p < 0) ? -p : 2 * (len - 1) - p;
  } while ((unsigned)p >= (unsigned)len);
  return p;
}

__device__ __forceinline__ void lds_set_range(u64* bits, int lo, int hi) {  // inclusive, hi - lo < 64
  const int wl = lo >> 6, wh = hi >> 6;
  if (wl == wh) {
    const u64 m = (~0ull << (lo & 63)) & (~0ull >> (63 - (hi & 63)));
    atomicOr(&bits[wl], m);
  } else {
    atomicOr(&bits[wl], ~0ull << (lo & 63));
    atomicOr(&bits[wh], ~0ull >> (63 - (hi & 63)));
  }
}

// THRESH_TOZERO on four packed bytes: keep bytes > thr, zero the others (add = (255-thr)*0x10001)
__device__ __forceinline__ unsigned tozero4(unsigned w, unsigned add) {
  unsigned e = w & 0x00FF00FFu, o = (w >> 8) & 0x00FF00FFu;
  const unsigned me = (((e + add) >> 8) & 0x00010001u) * 0xFFu;
  const unsigned mo = (((o + add) >> 8) & 0x00010001u) * 0xFFu;
  return (e & me) | ((o & mo) << 8);
}

// Window of thresholded pixels of one island in LDS: rows ylo..ylo+H-1, segment columns
// pwc0..pwc0+nseg-1 (16 bytes each; columns outside the image hold zeros).
struct PixWin {
  const uint8_t* pix;
  int ylo, H, pwc0, PW;  // PW = bytes per window row
  unsigned add;          // 0: `pix` holds thresholded pixels (the LDS windows); else: raw frame bytes, THRESH_TOZERO
                         // applied on the fly with tozero4's constant (255 - thr) * 0x10001 (the general kernel)
};

// fixed-point Gaussian for the 16 outputs x0..x0+15 of image row y, reading the LDS window.
// Returns the bit mask of outputs whose blurred value is non-zero: (sum + 2^15) >> 16 != 0.
// Pixels outside the window count as zero here.  EDGE (segments within reach of the left / right image border):
// BORDER_REFLECT_101 mirrors the pixels next to the border instead; that only matters if one of the mirrored pixels
// — the `zone` bits over the taps' input positions j (pixel x0 - R + j) — is non-zero after the threshold, which
// is reported in edge_or so that the caller can redo the item with the byte-wise border code.  LED spots sit well
// inside the frame / the tracking ROI (20 px border), so the mirrored zone is almost always dark.
template <int KS, bool EDGE, bool RAW = false>
__device__ __forceinline__ unsigned blur_item_fast(const PixWin& w, int rows, int cols, int y, int c,
                                                   const DetectParams& dp, unsigned zone, unsigned& edge_or) {
  constexpr int R = KS / 2;
  // The horizontal pass as packed byte dot products: the four bytes from input position x .. x + 3 against taps 0..3
  // (v_dot4_u32_u8), for five taps a second one for tap 4.  Integer arithmetic throughout: the same sums as tap by
  // tap, in any order.  (Constant indices into the by-value kernel argument: scalar registers, no LDS traffic.)
  const unsigned TA = dp.taps_packed[0], TB = dp.taps_packed[1];
  unsigned acc[16];
#pragma unroll
  for (int x = 0; x < 16; ++x) acc[x] = 0;
  unsigned eor = 0;
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    const int yb = reflect101(y + i - R, rows) - w.ylo;
    if ((unsigned)yb >= (unsigned)w.H) continue;  // rows outside the band hold no bright pixel
    const int sc = c - 1 - w.pwc0, nsw = w.PW >> 4;  // window segment index of column c-1
    const uint4* p = reinterpret_cast<const uint4*>(w.pix + (size_t)yb * w.PW);
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    const uint4 q0 = ((unsigned)sc < (unsigned)nsw) ? p[sc] : z4;
    const uint4 q1 = ((unsigned)(sc + 1) < (unsigned)nsw) ? p[sc + 1] : z4;
    const uint4 q2 = ((unsigned)(sc + 2) < (unsigned)nsw) ? p[sc + 2] : z4;
    unsigned q[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
    if (RAW) {
#pragma unroll
      for (int k = 0; k < 12; ++k) q[k] = tozero4(q[k], w.add);
    }
    if (EDGE) {
#pragma unroll
      for (int j = 0; j < 16 + 2 * R; ++j) {
        const int k = 16 - R + j;
        eor |= ((zone >> j) & 1u) ? ((q[k >> 2] >> (8 * (k & 3))) & 0xFFu) : 0u;
      }
    }
    constexpr int NW = 16 + (KS > 4 ? 4 : 0);
    unsigned win[NW];  // win[x] = the bytes of input positions x .. x + 3 (position j = pixel x0 - R + j)
#pragma unroll
    for (int x = 0; x < NW; ++x) {
      const int k = 16 - R + x;
      win[x] = (k & 3) ? __builtin_amdgcn_alignbyte(q[(k >> 2) + 1], q[k >> 2], k & 3) : q[k >> 2];
    }
    const unsigned ky = (unsigned)dp.taps[i];
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      unsigned h = __builtin_amdgcn_udot4(win[x], TA, 0u, false);
      if (KS > 4) h = __builtin_amdgcn_udot4(win[x + 4], TB, h, false);
      acc[x] += ky * h;
    }
  }
  unsigned m = 0;
#pragma unroll
  for (int x = 0; x < 16; ++x)
    if (acc[x] >= (1u << 15)) m |= 1u << x;
  if (EDGE) {
    edge_or = eor;
    const int valid = cols - 16 * c;  // outputs at x >= cols do not exist
    if (valid < 16) m &= (1u << valid) - 1u;
  }
  return m;
}

// any kernel size / image border (BORDER_REFLECT_101 in x): byte-wise from the LDS window
__device__ __noinline__ unsigned blur_item_generic(const PixWin& w, int rows, int cols, int y, int c,
                                                   const int* __restrict__ taps, int ks) {
  const int r = ks / 2;
  const int x0 = 16 * c;
  unsigned m = 0;
  for (int x = 0; x < 16; ++x) {
    if (x0 + x >= cols) break;
    int acc = 0;
    for (int i = 0; i < ks; ++i) {
      const int yb = reflect101(y + i - r, rows) - w.ylo;
      if ((unsigned)yb >= (unsigned)w.H) continue;
      int h = 0;
      for (int j = 0; j < ks; ++j) {
        const int so = reflect101(x0 + x + j - r, cols) - 16 * w.pwc0;
        if ((unsigned)so < (unsigned)w.PW) {
          int v = (int)w.pix[(size_t)yb * w.PW + so];
          if (w.add && v <= 255 - (int)(w.add & 0xFFFFu)) v = 0;  // raw frame bytes: THRESH_TOZERO here
          h += taps[j] * v;
        }
      }
      acc += taps[i] * h;
    }
    if (acc >= (1 << 15)) m |= 1u << x;
  }
  return m;
}

// 8-neighbourhood occupancy codes, bit d = direction d non-zero; directions as OpenCV's chain codes:
// 0 E, 1 NE, 2 N, 3 NW, 4 W, 5 SW, 6 S, 7 SE (y grows downwards)
// (image coordinates are < 2^15 and steps are -1 / 0 / 1: the 24-bit multiplier is exact and full rate)
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }
// chain-code steps, packed 2 bits per direction (value + 1)
__device__ __forceinline__ int dir_dx(int s) { return (int)((0x901Au >> (2 * s)) & 3u) - 1; }
__device__ __forceinline__ int dir_dy(int s) { return (int)((0xA901u >> (2 * s)) & 3u) - 1; }

struct PolyAcc {
  long long a00, a10, a01;
  int fx, fy, lx, ly, n;
  int xmin, xmax, ymin, ymax;
  __device__ __forceinline__ void init() {
    a00 = a10 = a01 = 0;
    n = 0;
    fx = fy = lx = ly = 0;
    xmin = ymin = 0x7fffffff;
    xmax = ymax = -0x7fffffff;
  }
  __device__ __forceinline__ void edge(int ax, int ay, int bx, int by) {
    const long long dxy = (long long)ax * by - (long long)bx * ay;
    a00 += dxy;
    a10 += dxy * (ax + bx);
    a01 += dxy * (ay + by);
  }
  __device__ __forceinline__ void emit(int x, int y) {
    if (n == 0) {
      fx = x;
      fy = y;
    } else {
      edge(lx, ly, x, y);
    }
    lx = x;
    ly = y;
    ++n;
    xmin = min(xmin, x);
    xmax = max(xmax, x);
    ymin = min(ymin, y);
    ymax = max(ymax, y);
  }
  __device__ __forceinline__ void close() { edge(lx, ly, fx, fy); }
};

__device__ __forceinline__ void set_bit(u64* bm, int wb, int slot, int xb) {
  atomicOr(&bm[(size_t)slot * wb + (xb >> 6)], 1ull << (xb & 63));
}

// Suzuki-Abe outer-border following exactly as OpenCV's icvFetchContour (CHAIN_APPROX_NONE):
// visited pixels are marked "positive" (pm) or, when the east neighbour was examined and is 0,
// "negative" (ng, takes precedence).  (xoff, yoff) turn window coordinates into image
// coordinates.  Returns false if the step bound was hit.
// The bitmaps as 32-bit words (a row = 2 * wb of them): the three bits x-1, x, x+1 of a row come out of two
// consecutive words and one funnel shift, for any x >= 1 (the pools end in a pad word).
__device__ __forceinline__ unsigned bits3_at(const unsigned* row32, int i, int sh) {
  const u64 v = ((u64)row32[i + 1] << 32) | row32[i];
  return (unsigned)(v >> sh);  // (callers mask)
}
__device__ __forceinline__ unsigned neighbours_at(const unsigned* nz32, int rd, int stride, int xb) {  // rd = slot * stride
  const int i = rd + ((xb - 1) >> 5), sh = (xb - 1) & 31;
  const unsigned u3 = bits3_at(nz32, i - stride, sh), m3 = bits3_at(nz32, i, sh), d3 = bits3_at(nz32, i + stride, sh);
  const unsigned urev = (__builtin_bitreverse32(u3) >> 28) & 0xEu;  // NE, N, NW at bits 1, 2, 3
  return ((m3 >> 2) & 1u) | urev | ((m3 & 1u) << 4) | ((d3 & 7u) << 5);  // 0 E, 1 NE, 2 N, 3 NW, 4 W, 5 SW, 6 S, 7 SE
}
__device__ __forceinline__ bool trace_outer_border(const u64* nz, u64* pm, u64* ng, int wb, int slot0, int xb0, int xoff, int yoff,
                                   PolyAcc& acc) {
  acc.init();
  const unsigned* nz32 = reinterpret_cast<const unsigned*>(nz);
  unsigned* pm32 = reinterpret_cast<unsigned*>(pm);
  unsigned* ng32 = reinterpret_cast<unsigned*>(ng);
  const int stride = 2 * wb;
  int rd = slot0 * stride;  // 32-bit word offset of the current row, advanced by +-stride (no multiplication per step)
  unsigned nb = neighbours_at(nz32, rd, stride, xb0);
  int s = 4;
  const int s_end0 = 4;
  bool hit;
  do {
    s = (s - 1) & 7;
    hit = (nb >> s) & 1;
  } while (!hit && s != s_end0);
  if (s == s_end0) {  // single-pixel component
    set_bit(ng, wb, slot0, xb0);
    acc.emit(xb0 + xoff, slot0 + yoff);
    acc.close();
    return true;
  }
  // positions packed as slot << 16 | x (both < 2^15): one comparison each for "back at the start" and "about to repeat
  // the first step"
  const int pos0 = (slot0 << 16) | xb0;
  const int pos1 = pos0 + (dir_dy(s) << 16) + dir_dx(s);
  int pos = pos0;
  int X = xb0 + xoff, Y = slot0 + yoff;  // image coordinates of the current border pixel
  long long a00 = 0, a10 = 0, a01 = 0;
  int xmin = X, xmax = X, ymin = Y, ymax = Y;
  // Straight-line loop body (lanes of different blobs stay in lock step).  The polygon sums take the edge to the
  // NEXT border pixel every step: for b = a + (dx, dy), a_x b_y - b_x a_y = a_x dy - a_y dx; at the last step the
  // next pixel is the start pixel, i.e. that edge closes the polygon.
  bool done;
  int step = 0;
  do {  // (everything after the test of `done` is harmless on the last step: the next pixel is the start pixel)
    const int s_end = s;
    const unsigned m16 = nb | (nb << 8);
    const int k = __builtin_ctz(m16 >> (s + 1));
    const int sn = (s + 1 + k) & 7;
    const bool negative = (unsigned)(sn - 1) < (unsigned)s_end;
    const int xb = pos & 0xFFFF;
    atomicOr((negative ? ng32 : pm32) + rd + (xb >> 5), 1u << (xb & 31));
    const int dx = dir_dx(sn), dy = dir_dy(sn);
    const int npos = pos + (dy << 16) + dx;
    done = (npos == pos0) & (pos == pos1);
    const int dxy = mul24(X, dy) - mul24(Y, dx);
    a00 += dxy;
    a10 += (long long)dxy * (2 * X + dx);
    a01 += (long long)dxy * (2 * Y + dy);
    xmin = min(xmin, X);
    xmax = max(xmax, X);
    ymin = min(ymin, Y);
    ymax = max(ymax, Y);
    pos = npos;
    X += dx;
    Y += dy;
    rd += dy * stride;
    s = (sn + 4) & 7;
    nb = neighbours_at(nz32, rd, stride, xb + dx);
  } while (!done && ++step < (1 << 20));
  acc.a00 = a00;
  acc.a10 = a10;
  acc.a01 = a01;
  acc.xmin = xmin;
  acc.xmax = xmax;
  acc.ymin = ymin;
  acc.ymax = ymax;
  return done;
}

// cv::undistortPoints(src, dst, K, D, noArray(), K) for one float point  (led_detector.cpp:97-98)
__device__ __forceinline__ void undistort_point(float sx, float sy, const DetectParams& dp, float& ox, float& oy) {
  double x = sx, y = sy;
  const double cx = dp.K[2], cy = dp.K[5];
  double x0 = x = (x - cx) * dp.ifx;
  double y0 = y = (y - cy) * dp.ify;
  const double* k = dp.k;
  for (int j = 0; j < dp.undist_iters; ++j) {
    double r2 = x * x + y * y;
    double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
    double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
    double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
    x = (x0 - deltaX) * icdist;
    y = (y0 - deltaY) * icdist;
  }
  double xx = dp.K[0] * x + dp.K[1] * y + dp.K[2];
  double yy = dp.K[3] * x + dp.K[4] * y + dp.K[5];
  double ww = 1. / (dp.K[6] * x + dp.K[7] * y + dp.K[8]);
  ox = (float)(xx * ww);
  oy = (float)(yy * ww);
}

// led_detector.cpp:65-86 for one contour given its exact polygon sums and bounding box
__device__ __forceinline__ bool blob_filter(const BlobRec& b, const DetectParams& dp, int roi_x, int roi_y, float& mcx,
                                            float& mcy) {
  const double s00 = (double)b.a00, s10 = (double)b.a10, s01 = (double)b.a01;
  const double area = fabs(s00 * 0.5);  // cv::contourArea
  const int width = b.xmax - b.xmin + 1, height = b.ymax - b.ymin + 1;
  double m00 = 0, m10 = 0, m01 = 0;  // cv::moments(contour)
  if (fabs(s00) > 1.1920928955078125e-07 /* FLT_EPSILON */) {
    const double db1_2 = s00 > 0 ? 0.5 : -0.5;
    const double db1_6 = s00 > 0 ? 0.16666666666666666666666666666667 : -0.16666666666666666666666666666667;
    m00 = s00 * db1_2;
    m10 = s10 * db1_6;
    m01 = s01 * db1_6;
  }
  mcx = (float)(m10 / m00) + (float)roi_x;
  mcy = (float)(m01 / m00) + (float)roi_y;
  const double w = (double)width, h = (double)height;
  const double hw = (double)(width / 2), hh = (double)(height / 2);  // INTEGER halves (quirk A.6.2)
  const double pi = 3.1415926535897932384626433832795;
  return area >= dp.min_area && area <= dp.max_area && fabs(1 - fmin(w / h, h / w)) <= dp.max_wh &&
         fabs(1 - (area / (pi * (hw * hw)))) <= dp.max_circ && fabs(1 - (area / (pi * (hh * hh)))) <= dp.max_circ;
}

// ---- shared by the fast (LDS) and the general (global scratch) blob kernels -------------------

// raster scan of one window for external contours: OpenCV's cvFindNextContour in RETR_EXTERNAL
// mode.  Window rows are bitmap slots 1..H (slot 0 and H+1 are zero separators), bit index
// xb = x - xw0 + 1.  For every traced outer border the polygon sums go through the shape filter;
// blobs that pass are handed to emit(mcx, mcy, key) with key = raster position of the start pixel.
// Raster scan of one island window for outer-border start points (one LANE per island).  Written as a per-lane
// state machine (slot, w, done, last_sign) so that the wave alternates between two converged phases: every lane
// advances its scan to its next start point (cheap, divergent trip counts), then ALL lanes that found one follow
// their borders in the same loop.  With the border following nested inside the scan loops the lanes reached it in
// different iterations and the wave executed the traces one after the other (the sum of the perimeters instead
// of the longest one).
template <class Emit>
__device__ __forceinline__ void scan_window(u64* nz, u64* pm, u64* ng, int W, int H, int ylo, int xw0,
                                            const DetectParams& dp, int roi_x, int roi_y, int* over, Emit emit) {
  int slot = 1, w = 0;
  int last_sign = 0;  // sign of the nearest marked pixel to the left (lnbd), 0 = none yet
  u64 done = 0;
  bool fin = H < 1;
  for (;;) {
    // ---- find: the next unmarked 1 with a 0 on its left that is not inside an already traced outer border
    bool have = false;
    int xb = 0;
    while (!have && !fin) {
      // (empty words — most of a window — only move the cursor: done is 0 on arrival, no mark can sit on them)
      while (!fin && nz[slot * W + w] == 0) {
        if (++w == W) {
          w = 0;
          last_sign = 0;
          fin = ++slot > H;
        }
      }
      if (fin) break;
      const int ro = slot * W + w;
      const u64 nzw = nz[ro];
      const u64 pw_ = pm[ro], gw = ng[ro];
      const u64 leftnz = (nzw << 1) | (w ? (nz[ro - 1] >> 63) : 0);
      const u64 cand = nzw & ~(pw_ | gw) & ~leftnz & ~done;
      if (cand) {
        const int bb = __builtin_ctzll(cand);
        done |= (bb == 63) ? ~0ull : ((2ull << bb) - 1);
        const u64 below = (pw_ | gw) & ((1ull << bb) - 1);
        int sign = last_sign;
        if (below) {
          const int hb = 63 - __builtin_clzll(below);
          sign = ((gw >> hb) & 1) ? -1 : 1;
        }
        if (sign <= 0) {  // (sign > 0: inside an already traced outer border, not external)
          have = true;
          xb = w * 64 + bb;
        }
      } else {  // this word is finished
        const u64 mk = pw_ | gw;
        if (mk) {
          const int hb = 63 - __builtin_clzll(mk);
          last_sign = ((gw >> hb) & 1) ? -1 : 1;
        }
        done = 0;
        if (++w == W) {
          w = 0;
          last_sign = 0;
          fin = ++slot > H;
        }
      }
    }
    // (a wave-uniform exit test: the compiler must finish the find loop of every lane before the border following
    //  starts instead of merging the two loops into one, which would serialise the lanes again)
    if (__builtin_amdgcn_ballot_w64(have) == 0) break;  // every lane of this call has scanned its whole window
    // ---- follow: all lanes that hold a start point, in lock step
    if (have) {
      PolyAcc acc;
      if (!trace_outer_border(nz, pm, ng, W, slot, xb, xw0 - 1, ylo - 1, acc)) *over = 1;
      BlobRec br;
      br.a00 = acc.a00;
      br.a10 = acc.a10;
      br.a01 = acc.a01;
      br.xmin = acc.xmin;
      br.xmax = acc.xmax;
      br.ymin = acc.ymin;
      br.ymax = acc.ymax;
      float mcx, mcy;
      K1B_ON_BLOBREC(br, ((unsigned)(ylo + slot - 1) << 12) | (unsigned)(xb + xw0 - 1));
      if (blob_filter(br, dp, roi_x, roi_y, mcx, mcy)) emit(mcx, mcy, ((unsigned)(ylo + slot - 1) << 12) | (unsigned)(xb + xw0 - 1));
    }
  }
}

// ---- contour phase WITHOUT border following ------------------------------------------------------------------
// What the reference needs of an external contour is its polygon area, its first moments (cv::moments of the point
// list) and its bounding box.  For a component without holes the polygon OpenCV's border following visits (8-connected
// foreground, CHAIN_APPROX_NONE) is the boundary of a cell complex: one unit square for every 2 x 2 block of pixels that
// is full, one half-square triangle for every block with exactly three pixels (the trace cuts the concave corner
// diagonally), edges walked out and back for everything thinner.  By Green's theorem the contour sums are then sums over
// cells — a00 = sum 2 |cell|, a10 = sum 6 int x dA, a01 = sum 6 int y dA, integers, any order — so they come from bit
// operations on pairs of bitmap rows, all rows at once, instead of ~30 dependent steps per LED with one lane alive
// (that phase was 20 of the 35 us of a frame's wave).  Holes are detected, not assumed away: 4 x the Euler number of the
// complex is Q1 - Q3 - 2 QD over the same blocks (Gray's bit-quad count, 8-connectivity) and must be 4 for one component
// without a hole; anything else — a hole, hence possibly components nested inside it which RETR_EXTERNAL must not report,
// an island too large for the cap, more blobs than the island record holds, a flood that does not settle — sends the
// WHOLE island to scan_window, the literal Suzuki-Abe trace.  Components are separated by flooding from the
// raster-first remaining pixel (= the pixel the trace starts from: same key) with 3 x 3 dilations under the mask, the
// rows of all islands of the frame at once.  (Checked against the trace on random masks in the CPU tier,
// tests/test_k1b_host.py::test_cell_sums_equal_the_border_trace, and by every detection parity test on the GPU.)
__device__ __forceinline__ void wave_sync();  // (defined with the fast path's wave plumbing below)
#define K1B_CELL_BLOBS 4     // blobs an island may yield in this phase
#define K1B_CELL_ITEMS 160   // (row, word) items an island may have
#define K1B_CELL_ITERS 96    // flood rounds before giving up
struct CellIsl {
  int bm_off, W, H, ylo, xw0;  // the island's bitmap window (scan_window's arguments)
  int lo, hi;                  // first / last slot with a pixel
  int item_end;                // inclusive prefix sum of the islands' item counts (hi - lo + 3) * W
  int seed;                    // slot << 16 | xb of the raster-first remaining pixel, INT_MAX: none
  int a00, a10, a01, chi, xmin, xmax, ymin, ymax;
  int nblob, state;            // state: 0 in progress, 1 finished, 2 handed to the border trace
  int simple;                  // one word wide, every row a single run that touches the next row's: one component
  float bx[K1B_CELL_BLOBS], by[K1B_CELL_BLOBS];
  unsigned bkey[K1B_CELL_BLOBS];
};
__device__ __forceinline__ int bitpos_sum(u64 m) {  // sum of the positions of the set bits
  return __builtin_popcountll(m & 0xAAAAAAAAAAAAAAAAull) + 2 * __builtin_popcountll(m & 0xCCCCCCCCCCCCCCCCull) +
         4 * __builtin_popcountll(m & 0xF0F0F0F0F0F0F0F0ull) + 8 * __builtin_popcountll(m & 0xFF00FF00FF00FF00ull) +
         16 * __builtin_popcountll(m & 0xFFFF0000FFFF0000ull) + 32 * __builtin_popcountll(m & 0xFFFFFFFF00000000ull);
}
// All islands cs[0 .. nisl) of one frame; lane / nl: this lane and the number of lanes working together (64; 1 in the
// CPU tier).  cs[k].bm_off .. xw0 filled in by the caller.  Returns a bit mask of the islands left to scan_window
// (their pm / ng regions zeroed again); every other island's blobs have been emitted.
// Every lane owns up to K1B_CELL_LANE_ITEMS (row, word) items, located once and kept in registers (island, bitmap
// offset, the row's remaining mask and current component): a flood round then costs a lane two LDS reads per item (the
// rows above and below; six more for islands wider than one word) and one write when its row grew.
#ifndef K1B_CELL_LANE_ITEMS
#define K1B_CELL_LANE_ITEMS 3
#endif
#ifdef K1B_STOP_AFTER  // (experiment builds, see K1B_STOP_POINT: 41 .. 44 end the contour phase early)
#define K1B_CELL_STOP(PHASE) \
  if (K1B_STOP_AFTER == (PHASE)) return 0u;
#else
#define K1B_CELL_STOP(PHASE)
#endif
template <class Emit>
__device__ __forceinline__ unsigned cells_phase(const u64* nz, u64* pm, u64* ng, CellIsl* cs, int nisl, int lane, int nl,
                                                const DetectParams& dp, int roi_x, int roi_y, Emit emit) {
  const int kIntMax = 0x7fffffff;
  constexpr int NIT = K1B_CELL_LANE_ITEMS;
  // ---- occupied slot range per island
  for (int k = lane; k < nisl; k += nl) {
    cs[k].lo = kIntMax;
    cs[k].hi = -1;
    cs[k].nblob = 0;
    cs[k].state = 0;
    cs[k].a00 = cs[k].a10 = cs[k].a01 = cs[k].chi = 0;
    cs[k].xmin = cs[k].ymin = kIntMax;
    cs[k].xmax = cs[k].ymax = -1;
  }
  wave_sync();
  int rows_total = 0;
  for (int k = 0; k < nisl; ++k) rows_total += cs[k].H;
  for (int i = lane; i < rows_total; i += nl) {
    int k = 0, r = i;
    while (r >= cs[k].H) r -= cs[k++].H;
    const int slot = r + 1, W = cs[k].W;
    u64 any = 0;
    for (int w = 0; w < W; ++w) any |= nz[cs[k].bm_off + slot * W + w];
    if (any) {
      atomicMin(&cs[k].lo, slot);
      atomicMax(&cs[k].hi, slot);
    }
  }
  wave_sync();
  // item ranges: lane k sizes island k (its `seed` word holds the count for a moment), then sums the counts up to k
  for (int k = lane; k < nisl; k += nl) {
    int n = 0;
    if (cs[k].hi < cs[k].lo) {
      cs[k].state = 1;  // nothing in this island
    } else {
      n = (cs[k].hi - cs[k].lo + 3) * cs[k].W;
      if (n > K1B_CELL_ITEMS || cs[k].W > 15) {
        cs[k].state = 2;
        n = 0;
      }
    }
    cs[k].seed = n;
  }
  wave_sync();
  for (int k = lane; k < nisl; k += nl) {
    int acc = 0, tot = 0;
    for (int j = 0; j < nisl; ++j) {
      const int n = cs[j].seed;
      tot += n;
      if (j <= k) acc += n;
    }
    if (tot > NIT * nl) {  // more rows than the lanes hold: the whole frame to the trace (rare)
      if (cs[k].state == 0) cs[k].state = 2;
      acc = 0;
    }
    cs[k].item_end = acc;
  }
  wave_sync();
  const int T = cs[nisl - 1].item_end;
  // ---- this lane's items: island, word offset in the bitmaps, word index / words per row / slot (packed)
  int it_o[NIT], it_m[NIT];
#pragma unroll
  for (int t = 0; t < NIT; ++t) {
    const int i = lane + nl * t;
    it_o[t] = -1;
    it_m[t] = 0;
    if (i < T) {
      int k = 0;
      while (i >= cs[k].item_end) ++k;
      const int li = i - (k ? cs[k - 1].item_end : 0), W = cs[k].W;
      const int r = li / W, w = li - r * W, slot = cs[k].lo - 1 + r;
      it_o[t] = cs[k].bm_off + slot * W + w;
      it_m[t] = k | (w << 6) | (W << 10) | (slot << 14);
    }
  }
  auto isl_of = [](int m) { return m & 63; };
  auto w_of = [](int m) { return (m >> 6) & 15; };
  auto W_of = [](int m) { return (m >> 10) & 15; };
  auto slot_of = [](int m) { return m >> 14; };
  u64 rem[NIT], cur[NIT];  // the row's remaining pixels and those of the component being flooded
#pragma unroll
  for (int t = 0; t < NIT; ++t) {
    rem[t] = it_o[t] >= 0 ? nz[it_o[t]] : 0;
    cur[t] = 0;
    if (it_o[t] >= 0) pm[it_o[t]] = 0;
  }
  wave_sync();
  K1B_CELL_STOP(41)
  // ---- the usual LED needs no flood: an island one word wide whose occupied rows are contiguous, each a single run
  //      that touches (8-neighbourhood) the run of the next row, is ONE component, and no background pixel of it is
  //      enclosed (it escapes along its own row, on its side of the run) — all its pixels are the first component
  for (int k = lane; k < nisl; k += nl) cs[k].simple = (cs[k].W == 1 && cs[k].state == 0) ? 1 : 0;
  wave_sync();
#pragma unroll
  for (int t = 0; t < NIT; ++t) {
    if (it_o[t] < 0) continue;
    const int k = isl_of(it_m[t]), slot = slot_of(it_m[t]);
    if (!cs[k].simple || slot < cs[k].lo || slot > cs[k].hi) continue;
    const u64 r = rem[t];
    bool bad = r == 0;
    if (!bad) {
      const u64 x = r >> __builtin_ctzll(r);
      bad = (x & (x + 1)) != 0;
      if (!bad && slot < cs[k].hi) {
        const u64 dn = nz[it_o[t] + 1];  // (W == 1: the next row)
        bad = ((dn | (dn << 1) | (dn >> 1)) & r) == 0;
      }
    }
    if (bad) cs[k].simple = 0;
  }
  wave_sync();
  K1B_CELL_STOP(42)
  // the cells of the row pair (slot, slot + 1) of one item: a / an = this row's word and the next word of the row, c / cn =
  // the same of the row below (pixels of the component only)
  auto item_sums = [&](const int m, const u64 a, const u64 an, const u64 c, const u64 cn) {
    const int k = isl_of(m), w = w_of(m), slot = slot_of(m);
    if (slot > cs[k].hi) return;
    const u64 b = (a >> 1) | (an << 63), d = (c >> 1) | (cn << 63);
    if ((a | b | c | d) == 0) return;
    const u64 full = a & b & c & d;
    const u64 t1 = ~a & b & c & d, t2 = a & ~b & c & d, t3 = a & b & ~c & d, t4 = a & b & c & ~d;  // missing tl tr bl br
    const u64 tri = t1 | t2 | t3 | t4;
    const u64 one = (a ^ b ^ c ^ d) & ~tri;
    const u64 diag = (a & d & ~b & ~c) | (b & c & ~a & ~d);
    const int nf = __builtin_popcountll(full), nt = __builtin_popcountll(tri);
    const int Xb = cs[k].xw0 + 64 * w - 1, Yb = cs[k].ylo + slot - 1;  // image coordinates of bit 0 / of this row
    atomicAdd(&cs[k].a00, 2 * nf + nt);
    atomicAdd(&cs[k].a10, 6 * (nf * Xb + bitpos_sum(full)) + 3 * nf + 3 * (nt * Xb + bitpos_sum(tri)) +
                              2 * __builtin_popcountll(t1 | t3) + __builtin_popcountll(t2 | t4));
    atomicAdd(&cs[k].a01, (6 * Yb + 3) * nf + 3 * Yb * nt + 2 * __builtin_popcountll(t1 | t2) +
                              __builtin_popcountll(t3 | t4));
    atomicAdd(&cs[k].chi, __builtin_popcountll(one) - nt - 2 * __builtin_popcountll(diag));
    if (a) {
      atomicMin(&cs[k].xmin, 64 * w + __builtin_ctzll(a));
      atomicMax(&cs[k].xmax, 64 * w + 63 - __builtin_clzll(a));
      atomicMin(&cs[k].ymin, slot);
      atomicMax(&cs[k].ymax, slot);
    }
  };
  // island k's component: its record through the shape filter, or the island to the border trace
  auto island_record = [&](const int k, const bool settled) {
    if (cs[k].state != 0) return;
    if (!settled || cs[k].chi != 4) {
      cs[k].state = 2;  // a hole (or a flood that did not settle): the literal trace decides
    } else {
      BlobRec br;
      br.a00 = cs[k].a00;
      br.a10 = cs[k].a10;
      br.a01 = cs[k].a01;
      br.xmin = cs[k].xmin + cs[k].xw0 - 1;
      br.xmax = cs[k].xmax + cs[k].xw0 - 1;
      br.ymin = cs[k].ymin + cs[k].ylo - 1;
      br.ymax = cs[k].ymax + cs[k].ylo - 1;
      const int sd = cs[k].seed;
      const unsigned key = ((unsigned)(cs[k].ylo + (sd >> 16) - 1) << 12) | (unsigned)((sd & 0xFFFF) + cs[k].xw0 - 1);
      K1B_ON_BLOBREC(br, key);
      float mcx, mcy;
      if (blob_filter(br, dp, roi_x, roi_y, mcx, mcy)) {
        if (cs[k].nblob >= K1B_CELL_BLOBS) {
          cs[k].state = 2;  // more blobs than the island record holds
        } else {
          const int n = cs[k].nblob++;
          cs[k].bx[n] = mcx;
          cs[k].by[n] = mcy;
          cs[k].bkey[n] = key;
        }
      }
    }
    cs[k].a00 = cs[k].a10 = cs[k].a01 = cs[k].chi = 0;
    cs[k].xmin = cs[k].ymin = kIntMax;
    cs[k].xmax = cs[k].ymax = -1;
  };
  // ---- every island of the frame is such a one-component island (the usual frame): no seeds, no flood, no mark bitmap
  //      — the component's rows are the bitmap's rows, its start pixel the first pixel of its first row
  bool flood_needed = false;
  for (int k = lane; k < nisl; k += nl) flood_needed = flood_needed || (cs[k].state == 0 && !cs[k].simple);
  if (__builtin_amdgcn_ballot_w64(flood_needed) == 0) {
#pragma unroll
    for (int t = 0; t < NIT; ++t) {
      if (it_o[t] < 0) continue;
      const int k = isl_of(it_m[t]);
      if (cs[k].state != 0) continue;
      if (slot_of(it_m[t]) == cs[k].lo) cs[k].seed = (cs[k].lo << 16) | __builtin_ctzll(rem[t]);  // (W == 1, rem != 0)
      item_sums(it_m[t], rem[t], 0, nz[it_o[t] + 1], 0);
    }
    wave_sync();
    K1B_CELL_STOP(44)
    unsigned todo1 = 0;
    for (int k = lane; k < nisl; k += nl) {
      island_record(k, true);
      if (cs[k].state == 0) cs[k].state = 1;
      if (cs[k].state == 2)
        todo1 |= 1u << k;
      else
        for (int n = 0; n < cs[k].nblob; ++n) emit(cs[k].bx[n], cs[k].by[n], cs[k].bkey[n]);
    }
    // (the islands' states sit in different lanes: a wave-wide OR)
    unsigned todo_all = 0;
    for (int k = 0; k < nisl; ++k) todo_all |= (__builtin_amdgcn_ballot_w64(((todo1 >> k) & 1u) != 0) != 0) ? 1u << k : 0u;
    return todo_all;
  }
  for (int round = 0;; ++round) {
    // ---- seed: the raster-first remaining pixel of every island still in progress
    for (int k = lane; k < nisl; k += nl) cs[k].seed = kIntMax;
    wave_sync();
    bool on[NIT];  // the item's island is in progress
#pragma unroll
    for (int t = 0; t < NIT; ++t) {
      on[t] = it_o[t] >= 0 && cs[isl_of(it_m[t])].state == 0;
      if (on[t] && rem[t])
        atomicMin(&cs[isl_of(it_m[t])].seed, (slot_of(it_m[t]) << 16) | (64 * w_of(it_m[t]) + __builtin_ctzll(rem[t])));
    }
    wave_sync();
    bool active = false;
    for (int k = lane; k < nisl; k += nl) {
      if (cs[k].state == 0 && cs[k].seed == kIntMax) cs[k].state = 1;  // every component of the island is done
      if (cs[k].state == 0) {
        active = true;
        if (round >= 2 * K1B_CELL_BLOBS) cs[k].state = 2;  // more components than this phase cares to separate
      }
    }
    if (__builtin_amdgcn_ballot_w64(active) == 0) break;  // (uniform)
    wave_sync();
    bool flood = false;  // some island of this lane's items has to be flooded
#pragma unroll
    for (int t = 0; t < NIT; ++t) {
      on[t] = on[t] && cs[isl_of(it_m[t])].state == 0;
      cur[t] = 0;
      if (on[t]) {
        const int sd = cs[isl_of(it_m[t])].seed;
        if (round == 0 && cs[isl_of(it_m[t])].simple)
          cur[t] = rem[t];
        else if ((sd >> 16) == slot_of(it_m[t]) && ((sd & 0xFFFF) >> 6) == w_of(it_m[t]))
          cur[t] = 1ull << (sd & 63);
        pm[it_o[t]] = cur[t];
        if (!(round == 0 && cs[isl_of(it_m[t])].simple)) flood = true;
      }
    }
    wave_sync();
    // ---- flood: 3 x 3 dilation under the mask until nothing changes
    bool changed;
    int it = 0;
    if (__builtin_amdgcn_ballot_w64(flood) != 0) do {
      changed = false;
#pragma unroll
      for (int t = 0; t < NIT; ++t) {
        if (!on[t] || !rem[t]) continue;
        const int W = W_of(it_m[t]), w = w_of(it_m[t]), o = it_o[t];
        const u64 up = pm[o - W], dn = pm[o + W];
        u64 acc = cur[t] | (cur[t] << 1) | (cur[t] >> 1) | up | (up << 1) | (up >> 1) | dn | (dn << 1) | (dn >> 1);
        if (W > 1) {  // bits carried in from the neighbouring words of the three rows
          if (w > 0) acc |= (pm[o - 1] | pm[o - W - 1] | pm[o + W - 1]) >> 63;
          if (w + 1 < W) acc |= (pm[o + 1] | pm[o - W + 1] | pm[o + W + 1]) << 63;
        }
        const u64 nv = acc & rem[t];
        if (nv != cur[t]) {
          cur[t] = nv;
          pm[o] = nv;
          changed = true;
        }
      }
      wave_sync();
    } while (__builtin_amdgcn_ballot_w64(changed) != 0 && ++it < K1B_CELL_ITERS);
    const bool settled = it < K1B_CELL_ITERS;
    K1B_CELL_STOP(43)
    // ---- sums over the cells of the row pairs (slot, slot + 1)
#pragma unroll
    for (int t = 0; t < NIT; ++t) {
      if (!on[t]) continue;
      const int W = W_of(it_m[t]), w = w_of(it_m[t]), o = it_o[t];
      item_sums(it_m[t], cur[t], w + 1 < W ? pm[o + 1] : 0, pm[o + W], w + 1 < W ? pm[o + W + 1] : 0);
    }
    wave_sync();
    K1B_CELL_STOP(44)
    // ---- one lane per island: the blob record through the shape filter, or the island to the border trace
    for (int k = lane; k < nisl; k += nl) island_record(k, settled);
#pragma unroll
    for (int t = 0; t < NIT; ++t) {  // the component leaves the remaining set
      if (it_o[t] < 0) continue;
      rem[t] &= ~cur[t];
      cur[t] = 0;
      pm[it_o[t]] = 0;
    }
    wave_sync();
  }
  wave_sync();
  // ---- finished islands emit; the others get their mark bitmaps back clean for scan_window (ng was never touched)
  unsigned todo = 0;
  for (int k = 0; k < nisl; ++k)
    if (cs[k].state == 2) todo |= 1u << k;
  for (int k = lane; k < nisl; k += nl) {
    if (cs[k].state != 2)
      for (int n = 0; n < cs[k].nblob; ++n) emit(cs[k].bx[n], cs[k].by[n], cs[k].bkey[n]);
  }
  (void)ng;
  return todo;
}

// blurred-mask bits of the 16 outputs of segment column c in image row y -> OR into the bitmap
template <bool RAW = false>
__device__ __forceinline__ void blur_to_bitmap(const PixWin& pw, int rows, int cols, const DetectParams& dp,
                                               const int* taps, int y, int c, u64* nzrow, int xw0) {
  const int ksize = dp.ksize;
  const int r = ksize / 2;
  const int x0 = 16 * c;
  if (x0 >= cols) return;
  unsigned m = 0;
  const bool interior = (x0 - r >= 0) && (x0 + 15 + r < cols);
  unsigned edge_or = 0;
  if (interior && ksize == 5 && dp.taps_u8) {
    m = blur_item_fast<5, false, RAW>(pw, rows, cols, y, c, dp, 0u, edge_or);
  } else if (interior && ksize == 3 && dp.taps_u8) {
    m = blur_item_fast<3, false, RAW>(pw, rows, cols, y, c, dp, 0u, edge_or);
  } else if ((ksize == 5 || ksize == 3) && dp.taps_u8 && cols >= 2 * r + 2) {
    // border segment: mirrored input positions j (pixel x = x0 - r + j): left border x in [1, r], right border
    // x in [cols - 1 - r, cols - 2]
    unsigned zone = 0;
    for (int j = 0; j < 16 + 2 * r; ++j) {
      const int x = x0 - r + j;
      if ((x0 - r < 0 && x >= 1 && x <= r) || (x0 + 15 + r >= cols && x >= cols - 1 - r && x <= cols - 2)) zone |= 1u << j;
    }
    m = ksize == 5 ? blur_item_fast<5, true, RAW>(pw, rows, cols, y, c, dp, zone, edge_or)
                   : blur_item_fast<3, true, RAW>(pw, rows, cols, y, c, dp, zone, edge_or);
    if (edge_or) m = blur_item_generic(pw, rows, cols, y, c, taps, ksize);  // a bright pixel next to the border
  } else {
    m = blur_item_generic(pw, rows, cols, y, c, taps, ksize);
  }
  if (m) {
    int xb0 = x0 - xw0 + 1;
    if (xb0 < 0) {  // (island windows start r pixels left of the first bright segment: the outputs further left are 0)
      m >>= -xb0;
      xb0 = 0;
      if (!m) return;
    }
    const int wi = xb0 >> 6, shb = xb0 & 63;
    atomicOr(&nzrow[wi], (u64)m << shb);
    if (shb > 48) atomicOr(&nzrow[wi + 1], (u64)m >> (64 - shb));
  }
}

// final stage: kept blobs -> OpenCV's contour order (newest first = descending raster order of
// the start pixel), float32 centroid -> undistortPoints, write the detection record
__device__ __forceinline__ void write_detections(const float* kx, const float* ky, const unsigned* kkey, int nk_all,
                                                 int kept_cap, int over, const DetectParams& dp, mpe_detections* out,
                                                 int lane) {
  const int nk = min(nk_all, kept_cap);
  for (int i = lane; i < nk; i += 64) {
    const unsigned key = kkey[i];
    int pos = 0;
    for (int j = 0; j < nk; ++j) pos += (kkey[j] > key) ? 1 : 0;
    if (pos < MPE_MAX_DETECTIONS) {
      const float mcx = kx[i], mcy = ky[i];
      float ux, uy;
      undistort_point(mcx, mcy, dp, ux, uy);
      out->dist_xy[2 * pos] = mcx;
      out->dist_xy[2 * pos + 1] = mcy;
      out->undist_xy[2 * pos] = (double)ux;
      out->undist_xy[2 * pos + 1] = (double)uy;
    }
  }
  if (lane == 0) {
    out->n = min(nk_all, MPE_MAX_DETECTIONS);
    int st = 0;
    if (nk_all > MPE_MAX_DETECTIONS) st = MPE_FRAME_TOO_MANY_DETECTIONS;
    if (over) st = MPE_FRAME_TOO_MANY_ROWS;
    out->status = st;
  }
}

// =============================================================================================
// K1b fast path.  One wave per frame: the front phases (A-D: flag bits -> bands -> islands -> thresholded
// pixels -> blurred mask bitmaps) use all 64 lanes; in the contour phase (E) one lane owns one island, and the lanes
// follow their borders in lock step (scan_window).
// Two capacity tiers, tried in turn (device work-lists chain them): K1bSmall covers the 4-6 LED case in
// 9.6 KB per wave (16 waves per CU), K1bLarge ~16 blobs per frame.
// Measured on MI355X (16 384 C2 frames, kernel alone): 0.317 ms with the border following nested in the raster scan
// (the lanes then follow their borders one after the other: phase E was 58 % of the kernel) -> 0.26 ms.
// =============================================================================================
// Per-frame window inside a uniform frame slot (batched ROI detection: every stream's ROI is cloned into a slot of
// g.rows x g.pitch bytes, zero beyond its own rows x cols; borders — BORDER_REFLECT_101, clipping — follow the
// window, the centroid offset (led_detector.cpp:74) its ROI origin).  wins == nullptr: every frame fills its slot.
struct FrameWin {
  int rows, cols, roi_x, roi_y;
};
__device__ __forceinline__ FrameGeom window_geom(const FrameGeom& g, const FrameWin* wins, int f, const DetectParams& dp,
                                                 int& roi_x, int& roi_y) {
  FrameGeom gl = g;  // slot layout (pitch, segments, bitset words) stays; rows / cols become the window's
  roi_x = dp.roi_x;
  roi_y = dp.roi_y;
  if (wins) {
    const FrameWin w = wins[f];
    gl.rows = w.rows;
    gl.cols = w.cols;
    roi_x = w.roi_x;
    roi_y = w.roi_y;
  }
  return gl;
}

struct Island {
  short ylo, yhi;      // band rows
  short clo, chi;      // output segment columns
  short cfirst, clast; // bright segment columns (pixel window)
  int pix_off, bm_off; // offsets into the pools
  int stage_end, blur_end;  // inclusive prefix sums of the flattened work-item counts
};

// The island's bitmap window: the blurred mask can only be non-zero within r pixels of a bright segment, i.e. in
// x = [16 cfirst - r, 16 clast + 15 + r] (clipped to the output columns clo .. chi and the image): bit 0 of a row is pixel
// xw0 - 1 (one pixel of margin on both sides for the 3 x 3 neighbourhoods).  An LED that straddles a segment boundary
// then still fits ONE 64-bit word per row (two bright segments: 38 pixels), where the window of the dilated segment
// columns took two — half the bitmap words to clear and combine, and the contour phase's one-word shortcuts apply.
#ifdef K1B_WIDE_WINDOWS  // (experiment builds: the windows of rounds 1 - 3, the dilated segment columns)
__device__ __forceinline__ int isl_xw0(const Island& is, int) { return 16 * is.clo; }
__device__ __forceinline__ int isl_words(const Island& is, int cols, int) {
  return ((min(cols - 1, 16 * is.chi + 15) - 16 * is.clo + 1) + 2 + 63) / 64;
}
#else
__device__ __forceinline__ int isl_xw0(const Island& is, int r) { return max(16 * is.clo, 16 * is.cfirst - r); }
__device__ __forceinline__ int isl_words(const Island& is, int cols, int r) {
  const int xhi = min(min(cols - 1, 16 * is.chi + 15), 16 * is.clast + 15 + r);
  return ((xhi - isl_xw0(is, r) + 1) + 2 + 63) / 64;
}
#endif
// capacities: thresholded-pixel pool [bytes], bitmap pool [u64 words per bitmap], bright segments, bands,
// islands, blobs kept per frame; WAVES = frames (one wave each) per block, whose islands ONE wave traces together
#ifndef K1B_SMALL_WAVES
#define K1B_SMALL_WAVES 1
#endif
#ifndef K1B_SMALL_PIX
#define K1B_SMALL_PIX 4096
#endif
#ifndef K1B_SMALL_BM
#define K1B_SMALL_BM 208
#endif
struct K1bSmall {
  enum { PIX = K1B_SMALL_PIX, BM = K1B_SMALL_BM, SEG = 64, BAND = 8, ISL = 8, KEPT = 16, WAVES = K1B_SMALL_WAVES, MIN_WAVES = 4 };
};
struct K1bLarge {
  enum { PIX = 12288, BM = 704, SEG = 512, BAND = 32, ISL = 32, KEPT = 64, WAVES = 1, MIN_WAVES = 2 };
};

// Synchronisation among the 64 lanes of ONE wave that communicate through LDS (the front phases of a frame
// belong to one wave; a block barrier there would couple the data-dependent control flow of the block's waves).
// DS operations of a wave execute in program order, so all that is needed is that the compiler keeps that order.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <class C>
struct K1bWaveLds {  // front-phase storage of one wave
  enum { SCRATCH = 4 * C::SEG + 512 + 32 * C::BAND + 4 * C::BAND, POOL = C::PIX > SCRATCH ? C::PIX : SCRATCH };
  __attribute__((aligned(16))) uint8_t pool[POOL];
  int taps[MPE_MAX_KSIZE];  // (taking the address of the by-value kernel argument would make the
                            //  compiler copy all of it to scratch)
  int nseg, nband;
};
template <class C>
struct K1bFrameLds {  // what the contour phase needs of one frame
  u64 nz[C::BM + 1], pm[C::BM + 1], ng[C::BM + 1];
  Island isl[C::ISL];
  float kx[C::KEPT], ky[C::KEPT];
  unsigned kkey[C::KEPT];
  int nkept, over, nisl;
  int ready, cols, roi_x, roi_y;  // for the wave that traces the block's islands
};

// (experiment builds, profiles/build_k1b_stops.sh: -DK1B_STOP_AFTER=n ends a frame's work after phase n — 1 A, 2 B, 3 C,
//  4 D, 5 the contour phase — with an empty record, to time the phases on the GPU; never defined in the product build)
#ifdef K1B_STOP_AFTER
#define K1B_STOP_POINT(PHASE, REC) \
  if (K1B_STOP_AFTER <= (PHASE)) { \
    if (lane == 0) {               \
      (REC)->n = 0;                \
      (REC)->status = 0;           \
    }                              \
    return false;                  \
  }
#else
#define K1B_STOP_POINT(PHASE, REC)
#endif
// Front phases of frame f.  Returns true when the island bitmaps in S are ready for the contour phase,
// false when the frame is finished (no bright pixel) or was handed to the next tier's work-list.
template <class C>
__device__ __forceinline__ bool k1b_front(const int f, const uint8_t* __restrict__ frames, size_t slot_bytes,
                                          const u64* __restrict__ flags, const FrameGeom& g, const DetectParams& dp,
                                          mpe_detections* __restrict__ dets, int* __restrict__ worklist,
                                          K1bWaveLds<C>& W, K1bFrameLds<C>& S) {
  const int lane = threadIdx.x & 63;
  // front-phase scratch lives in the pixel pool (dead before the pool is filled in phase C)
  unsigned* s_seg = reinterpret_cast<unsigned*>(W.pool);  // y << 16 | segment column
  u64* s_rowact = reinterpret_cast<u64*>(W.pool + 4 * C::SEG);
  u64(*s_colocc)[4] = reinterpret_cast<u64(*)[4]>(W.pool + 4 * C::SEG + 512);
  short* s_bandlo = reinterpret_cast<short*>(W.pool + 4 * C::SEG + 512 + 32 * C::BAND);
  short* s_bandhi = s_bandlo + C::BAND;
  uint8_t* s_pix = W.pool;
  const int* s_taps = W.taps;
  int& s_nseg = W.nseg;
  int& s_nband = W.nband;
  u64 *s_nz = S.nz, *s_pm = S.pm, *s_ng = S.ng;
  Island* s_isl = S.isl;
  int& s_nkept = S.nkept;
  int& s_over = S.over;
  int& s_nisl = S.nisl;
  const uint8_t* frame = frames + (size_t)f * slot_bytes;  // (g = the frame's window geometry, see window_geom)
  mpe_detections* out = dets + f;
  const int r = dp.ksize / 2;
  const int dc = (r + 15) / 16;  // segment columns a bright segment can influence on each side
  const int spr = g.segs_per_row;
  const unsigned add = (unsigned)(255 - dp.thr) * 0x00010001u;

  s_rowact[lane] = 0;
  for (int i = lane; i < C::BAND * 4; i += 64) (&s_colocc[0][0])[i] = 0;
  if (lane == 0) {
    s_nseg = 0;
    s_nkept = 0;
    s_over = 0;
    s_nband = 0;
    s_nisl = 0;
  }
  wave_sync();

  // ---- A: bright segments of this frame -> LDS list; rows within +-r become active
  {
    const size_t G0 = (size_t)f * g.segs_per_frame;
    const int nwin = (g.segs_per_frame + 63) >> 6;
    const size_t w0 = G0 >> 6;
    const int sh = (int)(G0 & 63);
    for (int i0 = 0; i0 < nwin; i0 += 256) {  // four independent flag loads per lane in flight
      u64 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + 64 * k + lane;
        v[k] = 0;
        if (i < nwin) {
          const u64 a = flags[w0 + i], b = flags[w0 + i + 1];
          v[k] = sh ? ((a >> sh) | (b << (64 - sh))) : a;
          const int rem = g.segs_per_frame - i * 64;
          if (rem < 64) v[k] &= (1ull << rem) - 1;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        u64 vv = v[k];
        const int i = i0 + 64 * k + lane;
        while (vv) {
          const int s = i * 64 + __builtin_ctzll(vv);
          vv &= vv - 1;
          const int y0 = s / spr, c0 = s - y0 * spr;
          const int slot = atomicAdd(&s_nseg, 1);
          if (slot < C::SEG) s_seg[slot] = ((unsigned)y0 << 16) | (unsigned)c0;
          lds_set_range(s_rowact, max(0, y0 - r), min(g.rows - 1, y0 + r));
        }
      }
    }
  }
  wave_sync();
  const int nseg = s_nseg;
  if (nseg == 0) {
    if (lane == 0) {
      out->n = 0;
      out->status = 0;
    }
    return false;
  }
  K1B_STOP_POINT(1, out)
  bool fallback = nseg > C::SEG;
  int why = fallback ? 1 : 0;  // which capacity sent the frame on (kept in the top byte of its work-list entry: statistics)

  // ---- B1: bands = maximal runs of active rows.  Lane w owns word w of the row bitset: band starts
  //      / ends are bit tricks, their ranks a wave prefix sum (starts and ends pair up in order).
  {
    const u64 act = (lane < g.rw) ? s_rowact[lane] : 0;
    const u64 prevw = (lane > 0 && lane < g.rw) ? s_rowact[lane - 1] : 0;
    const u64 nextw = (lane + 1 < g.rw) ? s_rowact[lane + 1] : 0;
    u64 st = act & ~((act << 1) | (prevw >> 63));   // row active, row above not
    u64 en = act & ~((act >> 1) | (nextw << 63));   // row active, row below not
    const int cs = __builtin_popcountll(st), ce = __builtin_popcountll(en);
    int ps = cs, pe = ce;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int a = __shfl_up(ps, d), b = __shfl_up(pe, d);
      if (lane >= d) {
        ps += a;
        pe += b;
      }
    }
    int is_ = ps - cs, ie = pe - ce;  // exclusive ranks
    if (!fallback) {
      while (st) {
        const int b = __builtin_ctzll(st);
        st &= st - 1;
        if (is_ < C::BAND) s_bandlo[is_] = (short)(lane * 64 + b);
        ++is_;
      }
      while (en) {
        const int b = __builtin_ctzll(en);
        en &= en - 1;
        if (ie < C::BAND) s_bandhi[ie] = (short)(lane * 64 + b);
        ++ie;
      }
    }
    if (lane == 63) s_nband = ps;
  }
  wave_sync();
  const int nband = s_nband;
  fallback = fallback || nband > C::BAND;
  if (fallback && !why) why = 2;

  // ---- B2: segment-column occupancy per band
  if (!fallback) {
    for (int i = lane; i < nseg; i += 64) {
      const unsigned sg = s_seg[i];
      const int y = (int)(sg >> 16), c = (int)(sg & 0xFFFF);
      int b = 0;
      while (b < nband - 1 && y > s_bandhi[b]) ++b;
      atomicOr(&s_colocc[b][c >> 6], 1ull << (c & 63));
    }
  }
  wave_sync();

  // ---- B3: islands = runs of occupied columns (dilated by dc) inside a band; lane b owns band b
  if (!fallback && lane < nband) {
    const int ylo = s_bandlo[lane], yhi = s_bandhi[lane];
    for (int cstart = 0; cstart < spr;) {
      int cfirst = -1;
      for (int wi = cstart >> 6; wi < 4 && wi * 64 < spr; ++wi) {
        u64 w = s_colocc[lane][wi];
        if (wi == (cstart >> 6)) w &= ~0ull << (cstart & 63);
        if (w) {
          cfirst = wi * 64 + __builtin_ctzll(w);
          break;
        }
      }
      if (cfirst < 0) break;
      int clast = cfirst;
      for (;;) {  // extend while the dilated runs touch: gap <= 2*dc
        int nxt = -1;
        for (int c = clast + 1; c <= min(spr - 1, clast + 2 * dc + 1); ++c)
          if ((s_colocc[lane][c >> 6] >> (c & 63)) & 1) {
            nxt = c;
            break;
          }
        if (nxt < 0) break;
        clast = nxt;
      }
      cstart = clast + 2 * dc + 2;
      const int idx = atomicAdd(&s_nisl, 1);
      if (idx < C::ISL) {
        Island is;
        is.ylo = (short)ylo;
        is.yhi = (short)yhi;
        is.cfirst = (short)cfirst;
        is.clast = (short)clast;
        is.clo = (short)max(0, cfirst - dc);
        is.chi = (short)min(spr - 1, clast + dc);
        is.pix_off = is.bm_off = is.stage_end = is.blur_end = 0;
        s_isl[idx] = is;
      }
    }
  }
  wave_sync();
  const int nisl = s_nisl;
  fallback = fallback || nisl > C::ISL;
  if (fallback && !why) why = 3;

  // ---- B4: pool offsets and work-item prefix sums (lane i owns island i; nisl <= 32)
  if (!fallback) {
    int pixb = 0, bmw = 0, nst = 0, nbl = 0;
    if (lane < nisl) {
      const Island is = s_isl[lane];
      const int H = is.yhi - is.ylo + 1;
      const int W = isl_words(is, g.cols, r);
      const int nbs = is.clast - is.cfirst + 1;
      pixb = H * 16 * nbs;
      bmw = (H + 2) * W;
      nst = H * nbs;
      nbl = H * (is.chi - is.clo + 1);
    }
    int ip = pixb, ib = bmw, is_ = nst, il = nbl;  // inclusive scans
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
      const int a = __shfl_up(ip, d), b = __shfl_up(ib, d), c = __shfl_up(is_, d), e = __shfl_up(il, d);
      if (lane >= d) {
        ip += a;
        ib += b;
        is_ += c;
        il += e;
      }
    }
    if (lane < nisl) {
      s_isl[lane].pix_off = ip - pixb;
      s_isl[lane].bm_off = ib - bmw;
      s_isl[lane].stage_end = is_;
      s_isl[lane].blur_end = il;
    }
    const int tot_pix = __shfl(ip, nisl - 1), tot_bm = __shfl(ib, nisl - 1);
    fallback = tot_pix > C::PIX || tot_bm > C::BM;
    if (fallback) why = tot_pix > C::PIX ? 4 : 5;
  }
  if (fallback) {  // hand the frame to the general kernel
    if (lane == 0) {
      out->n = 0;
      out->status = MPE_FRAME_TOO_MANY_ROWS;  // overwritten by the general kernel
      if (worklist) {
        const int k = atomicAdd(&worklist[0], 1);
        worklist[1 + k] = f | (why << 24);
      }
    }
    return false;
  }
  wave_sync();
  K1B_STOP_POINT(2, out)

  // ---- C: clear the bitmaps, stage the thresholded pixels of every island (16-byte loads)
  {
    const int tot_bm = s_isl[nisl - 1].bm_off +
                       (s_isl[nisl - 1].yhi - s_isl[nisl - 1].ylo + 3) * isl_words(s_isl[nisl - 1], g.cols, r);
    for (int i = lane; i < tot_bm; i += 64) {
      s_nz[i] = 0;
      s_pm[i] = 0;
      s_ng[i] = 0;
    }
    const int tot_stage = s_isl[nisl - 1].stage_end;
    for (int i = lane; i < tot_stage; i += 64) {
      int k = 0;
      while (i >= s_isl[k].stage_end) ++k;
      const Island is = s_isl[k];
      const int li = i - (k ? s_isl[k - 1].stage_end : 0);
      const int nbs = is.clast - is.cfirst + 1;
      const int yb = li / nbs, sc = li - yb * nbs;
      uint4 v = *reinterpret_cast<const uint4*>(frame + (size_t)(is.ylo + yb) * g.pitch + 16 * (is.cfirst + sc));
      v.x = tozero4(v.x, add);
      v.y = tozero4(v.y, add);
      v.z = tozero4(v.z, add);
      v.w = tozero4(v.w, add);
      *reinterpret_cast<uint4*>(s_pix + is.pix_off + (size_t)yb * 16 * nbs + 16 * sc) = v;
    }
  }
  wave_sync();
  K1B_STOP_POINT(3, out)

  // ---- D: blurred mask of every island
  {
    const int tot_blur = s_isl[nisl - 1].blur_end;
    for (int i = lane; i < tot_blur; i += 64) {
      int k = 0;
      while (i >= s_isl[k].blur_end) ++k;
      const Island is = s_isl[k];
      const int li = i - (k ? s_isl[k - 1].blur_end : 0);
      const int ncols = is.chi - is.clo + 1;
      const int yb = li / ncols, c = is.clo + (li - yb * ncols);
      const int H = is.yhi - is.ylo + 1;
      const int W = isl_words(is, g.cols, r);
      const PixWin pw = {s_pix + is.pix_off, is.ylo, H, is.cfirst, 16 * (is.clast - is.cfirst + 1), 0u};
      blur_to_bitmap(pw, g.rows, g.cols, dp, s_taps, is.ylo + yb, c, s_nz + is.bm_off + (size_t)(yb + 1) * W,
                     isl_xw0(is, r));
    }
  }
  wave_sync();
  K1B_STOP_POINT(4, out)
  return true;
}

// hand a frame to the next tier
__device__ __forceinline__ void k1b_hand_over(int f, mpe_detections* __restrict__ dets, int* __restrict__ worklist) {
  dets[f].n = 0;
  dets[f].status = MPE_FRAME_TOO_MANY_ROWS;  // overwritten by the next tier
  if (worklist) {
    const int k = atomicAdd(&worklist[0], 1);
    worklist[1 + k] = f | (6 << 24);  // (more blobs kept than the tier records)
  }
}

// One block = C::WAVES frames.  Every wave runs the front phases of its own frame; the contour phase, in which one
// LANE owns one island, is run by wave 0 over the islands of ALL the block's frames.  Measured on MI355X (16 384 C2
// frames, 5 islands per frame): WAVES = 1 / 2 / 4 / 8 take 0.260 / 0.256 / 0.258 / 0.282 ms alone and 20.73 / 20.92 /
// 20.99 / 21.97 ms per 262 144-frame step inside the pipeline (the other waves of a block wait at the barrier while
// wave 0 follows the borders), so one frame per block stays the default.  `valid`: this wave has a frame.
template <class C>
__device__ __forceinline__ void k1b_wave(const int f, const bool valid, const uint8_t* __restrict__ frames,
                                         const u64* __restrict__ flags, const FrameGeom& gslot, const DetectParams& dp,
                                         mpe_detections* __restrict__ dets, int* __restrict__ worklist,
                                         const FrameWin* __restrict__ wins) {
  __shared__ K1bWaveLds<C> Wl[C::WAVES];
  __shared__ K1bFrameLds<C> Sl[C::WAVES];
  const int lane = threadIdx.x & 63;
  const int wv = C::WAVES > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  K1bWaveLds<C>& W = Wl[wv];
  K1bFrameLds<C>& S = Sl[wv];
  __syncthreads();  // (list mode: the previous group of this block is completely done)
  if (lane < MPE_MAX_KSIZE) W.taps[lane] = dp.taps[lane < dp.ksize ? lane : 0];
  const size_t slot_bytes = (size_t)gslot.rows * gslot.pitch;
  int roi_x, roi_y;
  const FrameGeom g = window_geom(gslot, wins, valid ? f : 0, dp, roi_x, roi_y);
  bool ready = false;
  if (valid) ready = k1b_front<C>(f, frames, slot_bytes, flags, g, dp, dets, worklist, W, S);
  if (lane == 0) {
    S.ready = ready ? 1 : 0;
    S.cols = g.cols;
    S.roi_x = roi_x;
    S.roi_y = roi_y;
  }
  __syncthreads();

  // ---- E: the contour phase.  One frame per block (the default): all lanes over the rows of all islands
  //      (cells_phase), and only islands it hands back — a hole, too large — are followed border by border, one lane
  //      per island.  Several frames per block (experiment builds): one lane per island of all the block's frames.
  if constexpr (C::WAVES == 1) {
    if (ready) {
      static_assert(sizeof(CellIsl) * C::ISL <= sizeof(W.pool), "the cell phase's island records live in the pixel pool");
      CellIsl* cs = reinterpret_cast<CellIsl*>(W.pool);  // (the thresholded pixels are dead once the bitmaps exist)
      const int nisl = S.nisl;
      if (lane < nisl) {
        const Island is = S.isl[lane];
        cs[lane].bm_off = is.bm_off;
        cs[lane].W = isl_words(is, g.cols, dp.ksize / 2);
        cs[lane].H = is.yhi - is.ylo + 1;
        cs[lane].ylo = is.ylo;
        cs[lane].xw0 = isl_xw0(is, dp.ksize / 2);
      }
      wave_sync();
      auto keep = [&](float mcx, float mcy, unsigned key) {
        const int k = atomicAdd(&S.nkept, 1);
        if (k < C::KEPT) {
          S.kx[k] = mcx;
          S.ky[k] = mcy;
          S.kkey[k] = key;
        }
      };
      const unsigned todo = cells_phase(S.nz, S.pm, S.ng, cs, nisl, lane, 64, dp, roi_x, roi_y, keep);
      if (todo && lane < nisl && ((todo >> lane) & 1u))  // (uniform `todo`; rare)
        scan_window(S.nz + cs[lane].bm_off, S.pm + cs[lane].bm_off, S.ng + cs[lane].bm_off, cs[lane].W, cs[lane].H,
                    cs[lane].ylo, cs[lane].xw0, dp, roi_x, roi_y, &S.over, keep);
    }
  } else if (wv == 0) {
    int base[C::WAVES + 1];
    base[0] = 0;
#pragma unroll
    for (int i = 0; i < C::WAVES; ++i) base[i + 1] = base[i] + (Sl[i].ready ? Sl[i].nisl : 0);
    for (int it = lane; it < base[C::WAVES]; it += 64) {
      int fi = 0;
#pragma unroll
      for (int i = 1; i < C::WAVES; ++i) fi += (it >= base[i]) ? 1 : 0;
      K1bFrameLds<C>& F = Sl[fi];
      const Island is = F.isl[it - base[fi]];
      const int H = is.yhi - is.ylo + 1;
      const int Wd = isl_words(is, F.cols, dp.ksize / 2);
      scan_window(F.nz + is.bm_off, F.pm + is.bm_off, F.ng + is.bm_off, Wd, H, is.ylo, isl_xw0(is, dp.ksize / 2), dp, F.roi_x,
                  F.roi_y, &F.over,
                  [&](float mcx, float mcy, unsigned key) {
                    const int k = atomicAdd(&F.nkept, 1);
                    if (k < C::KEPT) {
                      F.kx[k] = mcx;
                      F.ky[k] = mcy;
                      F.kkey[k] = key;
                    }
                  });
    }
  }
  __syncthreads();
  if (!ready) return;
#ifdef K1B_STOP_AFTER
  if (K1B_STOP_AFTER <= 5) {
    if (lane == 0) {
      dets[f].n = 0;
      dets[f].status = 0;
    }
    return;
  }
#endif
  if (C::KEPT < 2 * MPE_MAX_DETECTIONS && S.nkept > C::KEPT) {  // more blobs than this tier records
    if (lane == 0) k1b_hand_over(f, dets, worklist);
    return;
  }
  write_detections(S.kx, S.ky, S.kkey, S.nkept, C::KEPT, S.over, dp, dets + f, lane);
}

// wave w of block b works on frame b * C::WAVES + w
template <class C>
__global__ __launch_bounds__(64 * C::WAVES, C::MIN_WAVES) void k1b_blobs(const uint8_t* __restrict__ frames,
                                                           const u64* __restrict__ flags, FrameGeom g, DetectParams dp,
                                                           mpe_detections* __restrict__ dets,
                                                           int* __restrict__ worklist, int n_frames,
                                                           const FrameWin* __restrict__ wins) {
  const int f = blockIdx.x * C::WAVES + (int)(threadIdx.x >> 6);
  k1b_wave<C>(f, f < n_frames, frames, flags, g, dp, dets, worklist, wins);
}
// frames taken from a device work-list (those the smaller tier handed over)
template <class C>
__global__ __launch_bounds__(64 * C::WAVES, C::MIN_WAVES) void k1b_blobs_list(const uint8_t* __restrict__ frames,
                                                                const u64* __restrict__ flags, FrameGeom g,
                                                                DetectParams dp, mpe_detections* __restrict__ dets,
                                                                const int* __restrict__ in_list,
                                                                int* __restrict__ worklist,
                                                                const FrameWin* __restrict__ wins) {
  const int count = in_list[0];
  for (int b0 = blockIdx.x * C::WAVES; b0 < count; b0 += gridDim.x * C::WAVES) {  // (uniform over the block)
    const int w0 = b0 + (int)(threadIdx.x >> 6);
    k1b_wave<C>(w0 < count ? (in_list[1 + w0] & 0xFFFFFF) : 0, w0 < count, frames, flags, g, dp, dets, worklist, wins);
  }
}

// =============================================================================================
// K1b general path: frames the fast tiers handed over (more bright segments / bands / islands than their LDS pools
// hold: salt noise, glare, a dot grid).  One wave per frame, whole-frame bitmaps in a global scratch slab, exact on
// any frame.  Round 5 (on cluttered frames this tier was a cliff — 43 us of ONE LANE per
// frame on 32 waves of the whole chip, 11.6 k frames/s with 0.05 % salt noise):
//   * no thresholded copy of the frame: the blur reads the frame's own rows and applies THRESH_TOZERO on the fly
//     (PixWin::add), only around bright segments;
//   * the contour scan runs one LANE PER BAND — a maximal run of rows with a non-zero blurred pixel.  A component of
//     the blurred mask cannot cross an empty row, and a band cannot lie inside a hole of a component of another band,
//     so cvFindContours' raster scan decomposes exactly, as it does for the islands of the fast tiers; the kept blobs
//     are put back into raster order of their start pixels at the end (write_detections);
//   * up to 1024 slabs (1 GB of scratch at most) instead of 32.
// =============================================================================================
#define K1B_GEN_KEPT 512
#define K1B_GEN_BANDS 2048  // rows <= 4096 (make_geom): at most every other row starts a band

__host__ __device__ inline size_t k1b_gen_scratch_bytes(const FrameGeom& g) {
  const size_t bm = (size_t)(g.rows + 2) * g.wb * 8;
  const size_t todo = (size_t)g.rows * g.tw * 8;
  const size_t kept = (size_t)K1B_GEN_KEPT * 12;
  return ((3 * bm + todo + kept + 255) / 256) * 256;
}
// slabs = blocks of the launch: as many as 1 GB of scratch holds, 32 .. g_k1b_gen_blocks_cap (a process-wide tuning
// knob, option "k1b_general_blocks"; the kernel is bound by the latency of its global-memory bitmaps, so its rate
// follows the number of waves in flight)
static int g_k1b_gen_blocks_cap = 1024;
void k1b_set_general_blocks(int cap) { g_k1b_gen_blocks_cap = cap < 32 ? 32 : (cap > 8192 ? 8192 : cap); }
int k1b_get_general_blocks() { return g_k1b_gen_blocks_cap; }
static int k1b_gen_blocks(const FrameGeom& g) {
  const size_t n = ((size_t)1 << 30) / k1b_gen_scratch_bytes(g);
  const size_t cap = (size_t)g_k1b_gen_blocks_cap;
  return (int)(n < 32 ? 32 : (n > cap ? cap : n));
}

__global__ __launch_bounds__(64) void k1b_general(const uint8_t* __restrict__ frames, const u64* __restrict__ flags,
                                                 FrameGeom gslot, DetectParams dp, mpe_detections* __restrict__ dets,
                                                 const int* __restrict__ worklist, uint8_t* __restrict__ scratch,
                                                 const FrameWin* __restrict__ wins) {
  const FrameGeom& g = gslot;  // slab layout and flag indexing: the slot; rows / cols of a frame: its window (gl below)
  __shared__ int s_nkept, s_over, s_nband;
  __shared__ int s_taps[MPE_MAX_KSIZE];
  __shared__ u64 s_rowact[64];
  __shared__ short s_blo[K1B_GEN_BANDS], s_bhi[K1B_GEN_BANDS];
  const int lane = threadIdx.x;
  const int count = worklist[0];
  if ((int)blockIdx.x >= count) return;  // (the usual launch: nothing was handed over)
  if (lane < MPE_MAX_KSIZE) s_taps[lane] = dp.taps[lane < dp.ksize ? lane : 0];
  __syncthreads();
  const size_t slab = k1b_gen_scratch_bytes(g);
  uint8_t* base = scratch + (size_t)blockIdx.x * slab;
  const size_t bm_words = (size_t)(g.rows + 2) * g.wb;
  u64* nz = reinterpret_cast<u64*>(base);
  u64* pm = nz + bm_words;
  u64* ng = pm + bm_words;
  u64* todo = ng + bm_words;
  float* kx = reinterpret_cast<float*>(todo + (size_t)g.rows * g.tw);
  float* ky = kx + K1B_GEN_KEPT;
  unsigned* kkey = reinterpret_cast<unsigned*>(ky + K1B_GEN_KEPT);
  const int r = dp.ksize / 2;
  const int dc = (r + 15) / 16;
  const int spr = g.segs_per_row;
  const unsigned add = (unsigned)(255 - dp.thr) * 0x00010001u;

  for (int wi = blockIdx.x; wi < count; wi += gridDim.x) {
    const int f = worklist[1 + wi] & 0xFFFFFF;
    const uint8_t* frame = frames + (size_t)f * g.rows * g.pitch;
    int roi_x, roi_y;
    const FrameGeom gl = window_geom(gslot, wins, f, dp, roi_x, roi_y);
    if (lane == 0) {
      s_nkept = 0;
      s_over = 0;
      s_nband = 0;
    }
    s_rowact[lane] = 0;
    for (size_t i = lane; i < bm_words; i += 64) {
      nz[i] = 0;
      pm[i] = 0;
      ng[i] = 0;
    }
    for (size_t i = lane; i < (size_t)g.rows * g.tw; i += 64) todo[i] = 0;
    __threadfence_block();
    __syncthreads();
    // todo segments: neighbourhood of every bright segment
    {
      const size_t G0 = (size_t)f * g.segs_per_frame;
      const int nwin = (g.segs_per_frame + 63) >> 6;
      const size_t w0 = G0 >> 6;
      const int sh = (int)(G0 & 63);
      for (int i = lane; i < nwin; i += 64) {
        const u64 a = flags[w0 + i], b = flags[w0 + i + 1];
        u64 v = sh ? ((a >> sh) | (b << (64 - sh))) : a;
        const int rem = g.segs_per_frame - i * 64;
        if (rem < 64) v &= (1ull << rem) - 1;
        while (v) {
          const int s = i * 64 + __builtin_ctzll(v);
          v &= v - 1;
          const int y0 = s / spr, c0 = s - y0 * spr;
          for (int yy = max(0, y0 - r); yy <= min(gl.rows - 1, y0 + r); ++yy)
            for (int cc = max(0, c0 - dc); cc <= min(spr - 1, c0 + dc); ++cc)
              atomicOr(&todo[(size_t)yy * g.tw + (cc >> 6)], 1ull << (cc & 63));
        }
      }
    }
    __threadfence_block();
    __syncthreads();
    // blur (lane = row): the frame's own bytes, thresholded on the fly; rows that got a non-zero pixel become active
    const PixWin pw = {frame, 0, gl.rows, 0, g.pitch, add};
    for (int y = lane; y < gl.rows; y += 64) {
      u64* nzrow = nz + (size_t)(y + 1) * g.wb;
      for (int tw = 0; tw < g.tw; ++tw) {
        u64 tb = todo[(size_t)y * g.tw + tw];
        while (tb) {
          const int c = tw * 64 + __builtin_ctzll(tb);
          tb &= tb - 1;
          if (add)
            blur_to_bitmap<true>(pw, gl.rows, gl.cols, dp, s_taps, y, c, nzrow, 0);
          else  // (thr = 255: nothing passes the threshold — the flags say so already, no item arrives here)
            blur_to_bitmap<false>(pw, gl.rows, gl.cols, dp, s_taps, y, c, nzrow, 0);
        }
      }
    }
    __threadfence_block();
    __syncthreads();
    for (int y = lane; y < gl.rows; y += 64) {
      const u64* nzrow = nz + (size_t)(y + 1) * g.wb;
      u64 any = 0;
      for (int w = 0; w < g.wb; ++w) any |= nzrow[w];
      if (any) atomicOr(&s_rowact[y >> 6], 1ull << (y & 63));
    }
    __syncthreads();
    // bands = maximal runs of active rows (lane w owns word w of the row bitset; ranks by a wave prefix sum)
    {
      const int rw = (gl.rows + 63) >> 6;
      const u64 act = (lane < rw) ? s_rowact[lane] : 0;
      const u64 prevw = (lane > 0 && lane < rw) ? s_rowact[lane - 1] : 0;
      const u64 nextw = (lane + 1 < rw) ? s_rowact[lane + 1] : 0;
      u64 st = act & ~((act << 1) | (prevw >> 63));
      u64 en = act & ~((act >> 1) | (nextw << 63));
      const int cs = __builtin_popcountll(st), ce = __builtin_popcountll(en);
      int ps = cs, pe = ce;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int a = __shfl_up(ps, d), b = __shfl_up(pe, d);
        if (lane >= d) {
          ps += a;
          pe += b;
        }
      }
      int is_ = ps - cs, ie = pe - ce;
      while (st) {
        const int b = __builtin_ctzll(st);
        st &= st - 1;
        if (is_ < K1B_GEN_BANDS) s_blo[is_] = (short)(lane * 64 + b);
        ++is_;
      }
      while (en) {
        const int b = __builtin_ctzll(en);
        en &= en - 1;
        if (ie < K1B_GEN_BANDS) s_bhi[ie] = (short)(lane * 64 + b);
        ++ie;
      }
      if (lane == 63) s_nband = ps;
    }
    __syncthreads();
    const int nband = s_nband;
    auto keep = [&](float mcx, float mcy, unsigned key) {
      const int k = atomicAdd(&s_nkept, 1);
      if (k < K1B_GEN_KEPT) {
        kx[k] = mcx;
        ky[k] = mcy;
        kkey[k] = key;
      }
    };
    if (nband <= K1B_GEN_BANDS) {
      // one lane per band, the lanes' border followings in lock step (scan_window): slot 0 of a band's window is the
      // empty row above it, slots 1 .. H its rows, the separator below the empty row that ends it
      for (int b0 = 0; b0 < nband; b0 += 64) {  // (uniform: every lane enters scan_window, with H = 0 if it has no band)
        const int b = b0 + lane;
        const int lo = b < nband ? s_blo[b] : 0, H = b < nband ? s_bhi[b] - lo + 1 : 0;
        const size_t off = (size_t)lo * g.wb;
        scan_window(nz + off, pm + off, ng + off, g.wb, H, lo, 0, dp, roi_x, roi_y, &s_over, keep);
      }
    } else {  // (cannot happen for rows <= 4096; the literal whole-frame scan by one lane)
      scan_window(nz, pm, ng, g.wb, lane == 0 ? gl.rows : 0, 0, 0, dp, roi_x, roi_y, &s_over, keep);
    }
    __threadfence_block();
    __syncthreads();
    write_detections(kx, ky, kkey, s_nkept, K1B_GEN_KEPT, s_over, dp, dets + f, lane);
    __syncthreads();
  }
}

size_t k1b_scratch_bytes(const FrameGeom& g) { return k1b_gen_scratch_bytes(g) * (size_t)k1b_gen_blocks(g); }

hipError_t launch_k1b_blobs(const uint8_t* frames, const unsigned long long* flags, int n_frames, const FrameGeom& g,
                            const DetectParams& dp, mpe_detections* dets, int* worklist, uint8_t* scratch,
                            int blob_hint, hipStream_t s, const void* frame_windows, bool lists_zeroed,
                            bool first_tier_only) {
  const FrameWin* wins = static_cast<const FrameWin*>(frame_windows);
  if (first_tier_only) {
    // low-latency tracked frame: the small tier alone, nothing queued behind it.  A frame that overflows it is left
    // with status MPE_FRAME_TOO_MANY_ROWS and no work-list entry; the caller sees that in the record it fetches
    // anyway and repeats the frame through the whole chain (three launches and a memset less on every other frame)
    if (n_frames <= 0) return hipSuccess;
    if (blob_hint <= 0 || blob_hint > 8) return hipErrorInvalidValue;
    const int blocks = (n_frames + K1bSmall::WAVES - 1) / K1bSmall::WAVES;
    hipLaunchKernelGGL((k1b_blobs<K1bSmall>), dim3(blocks), dim3(64 * K1bSmall::WAVES), 0, s, frames, (const u64*)flags, g,
                       dp, dets, (int*)nullptr, n_frames, wins);
    return hipGetLastError();
  }
  // Three tiers, chained through device work-lists (no host round trip):
  //   small LDS pools (4 waves/SIMD) -> large LDS pools -> whole-frame window in global scratch
  // (Running the follow-up tiers on a side stream beside the voting kernel, whose blocks then waited for the few
  //  frames those tiers finish, was built and measured in round 3: the agent-scope release every blob wave needs for
  //  that hand-over — an L2 write-back — took the blob kernel from 0.34 to 2.0 ms per sub-batch.  Removed.)
  if (n_frames <= 0) return hipSuccess;
  int* list_a = worklist;                   // small -> large
  int* list_b = worklist + (n_frames + 1);  // large -> general
  hipError_t e = hipSuccess;
  if (!lists_zeroed) {  // both counters with ONE memset: list_b's counter sits right behind list_a's entries
    e = hipMemsetAsync(list_a, 0, (size_t)(n_frames + 2) * sizeof(int), s);
    if (e != hipSuccess) return e;
  }
  if (blob_hint > 0 && blob_hint <= 8) {  // (the small tier is cheap to try: frames that overflow it go on to the large one)
    const int blocks = (n_frames + K1bSmall::WAVES - 1) / K1bSmall::WAVES;
    hipLaunchKernelGGL((k1b_blobs<K1bSmall>), dim3(blocks), dim3(64 * K1bSmall::WAVES), 0, s, frames, (const u64*)flags, g, dp, dets,
                       list_a, n_frames, wins);
    const int grid = n_frames < 2048 ? n_frames : 2048;
    hipLaunchKernelGGL((k1b_blobs_list<K1bLarge>), dim3(grid), dim3(64), 0, s, frames, (const u64*)flags, g, dp,
                       dets, (const int*)list_a, list_b, wins);
  } else {
    hipLaunchKernelGGL((k1b_blobs<K1bLarge>), dim3(n_frames), dim3(64), 0, s, frames, (const u64*)flags, g, dp,
                       dets, list_b, n_frames, wins);
  }
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k1b_general, dim3(k1b_gen_blocks(g)), dim3(64), 0, s, frames, (const u64*)flags, g, dp, dets,
                     (const int*)list_b, scratch, wins);
  return hipGetLastError();
}

// =============================================================================================
// K2 — brute-force correspondence voting (pose_estimator.cpp:544-702)
// =============================================================================================
// lexicographic unranking of the idx-th 3-combination of {0..n-1}
__device__ __forceinline__ void unrank_combo3(int idx, int n, int& a, int& b, int& c) {
  a = 0;
  for (;;) {
    const int cnt = (n - 1 - a) * (n - 2 - a) / 2;  // combos starting with a
    if (idx < cnt) break;
    idx -= cnt;
    ++a;
  }
  b = a + 1;
  for (;;) {
    const int cnt = n - 1 - b;
    if (idx < cnt) break;
    idx -= cnt;
    ++b;
  }
  c = b + 1 + idx;
}

__device__ __forceinline__ V3 bearing(double u, double v, double fx, double fy, double cx, double cy) {
  V3 s = {(u - cx) / fx, (v - cy) / fy, 1.0};  // pose_estimator.cpp:288-301
  return vdiv(s, norm(s));
}

__device__ __forceinline__ double pick_root(const P3PCtx& c, int k) {
  return k == 0 ? c.root[0] : (k == 1 ? c.root[1] : (k == 2 ? c.root[2] : c.root[3]));
}
#define K2_THREADS 256
#define K2_TRI_CHUNK 64  // detection triples staged in LDS per pass
#define K2_TRI_CHUNK_SCAN 16
#define K2_LTAB 14  // doubles per marker permutation in the LDS copy of the table (scan-carrying variant)

// ---- marker-permutation table (frame independent) -------------------------------------------
// One entry per ordered marker triple (P1,P2,P3), in the reference's permutation order
// (combinations.cpp:131-244).  Holds everything of P3P::computePoses that depends on the world
// points only (p3p.cpp:124-141): the eta frame N, P1, p_1, p_2, d_12, the collinearity verdict,
// and the unused markers expressed in the eta frame, N (m - P1), ascending marker index.
//   [0..8] N rows, [9..11] P1, [12] p_1, [13] p_2, [14] d_12, [15] valid (1/0),
//   [16] p0 | p1 << 8 | p2 << 16 (as a double), [17] pad, [18 + 3u ..] m_eta[u],
//   behind them the same markers in single precision, 3 floats each (two per double; the plain variant's prefilter)
__host__ __device__ inline int k2_entry_f32_at(int n_m) { return 18 + 3 * (n_m - 3); }  // (in doubles)
__host__ __device__ inline int k2_entry_doubles(int n_m) { return k2_entry_f32_at(n_m) + (3 * (n_m - 3) + 1) / 2; }

__device__ __forceinline__ void perm_from_index(int pj, int n_m, int& p0, int& p1, int& p2) {
  int ma, mb, mc;
  unrank_combo3(pj / 6, n_m, ma, mb, mc);
  switch (pj % 6) {  // block rows [c b a],[c a b],[b c a],[b a c],[a b c],[a c b]
    case 0: p0 = mc; p1 = mb; p2 = ma; break;
    case 1: p0 = mc; p1 = ma; p2 = mb; break;
    case 2: p0 = mb; p1 = mc; p2 = ma; break;
    case 3: p0 = mb; p1 = ma; p2 = mc; break;
    case 4: p0 = ma; p1 = mb; p2 = mc; break;
    default: p0 = ma; p1 = mc; p2 = mb; break;
  }
}

// one entry of the table (layout above) -> e
__device__ __forceinline__ void k2_marker_entry(const SolveParams& sp, int pj, double* __restrict__ e) {
  const int n_m = sp.n_markers;
  int p0, p1, p2;
  perm_from_index(pj, n_m, p0, p1, p2);
  const V3 P1 = {sp.markers[3 * p0], sp.markers[3 * p0 + 1], sp.markers[3 * p0 + 2]};
  const V3 P2 = {sp.markers[3 * p1], sp.markers[3 * p1 + 1], sp.markers[3 * p1 + 2]};
  const V3 P3 = {sp.markers[3 * p2], sp.markers[3 * p2 + 1], sp.markers[3 * p2 + 2]};
  const bool valid = norm(cross(P2 - P1, P3 - P1)) != 0.0;  // p3p.cpp:77-80
  V3 n1 = P2 - P1;
  n1 = vdiv(n1, norm(n1));
  V3 n3 = cross(n1, P3 - P1);
  n3 = vdiv(n3, norm(n3));
  const V3 n2 = cross(n3, n1);
  const M3 N = {n1, n2, n3};
  const V3 P3n = mul(N, P3 - P1);
  e[0] = n1.x; e[1] = n1.y; e[2] = n1.z;
  e[3] = n2.x; e[4] = n2.y; e[5] = n2.z;
  e[6] = n3.x; e[7] = n3.y; e[8] = n3.z;
  e[9] = P1.x; e[10] = P1.y; e[11] = P1.z;
  e[12] = P3n.x;
  e[13] = P3n.y;
  e[14] = norm(P2 - P1);
  e[15] = valid ? 1.0 : 0.0;
  e[16] = (double)(p0 | (p1 << 8) | (p2 << 16));
  e[17] = 0.0;
  int u = 0;
  for (int m = 0; m < n_m; ++m) {
    if (m == p0 || m == p1 || m == p2) continue;
    const V3 mm = {sp.markers[3 * m], sp.markers[3 * m + 1], sp.markers[3 * m + 2]};
    const V3 me = mul(N, mm - P1);
    e[18 + 3 * u] = me.x;
    e[18 + 3 * u + 1] = me.y;
    e[18 + 3 * u + 2] = me.z;
    ++u;
  }
  float* ef = reinterpret_cast<float*>(e + k2_entry_f32_at(n_m));
  for (int i = 0; i < 3 * u; ++i) ef[i] = (float)e[18 + i];
  if (u & 1) ef[3 * u] = 0.f;
}

__global__ void k2_prep_markers(SolveParams sp, double* __restrict__ tab) {
  const int n_m = sp.n_markers;
  const int n_perms = n_m * (n_m - 1) * (n_m - 2);
  const int esz = k2_entry_doubles(n_m);
  for (int pj = blockIdx.x * blockDim.x + threadIdx.x; pj < n_perms; pj += gridDim.x * blockDim.x)
    k2_marker_entry(sp, pj, tab + (size_t)pj * esz);
}

hipError_t launch_k2_prep(const SolveParams& sp, double* tab, hipStream_t s) {
  if (sp.n_markers < 4) return hipSuccess;
  const int n_perms = sp.n_markers * (sp.n_markers - 1) * (sp.n_markers - 2);
  hipLaunchKernelGGL(k2_prep_markers, dim3((n_perms + 127) / 128), dim3(128), 0, s, sp, tab);
  return hipGetLastError();
}
size_t k2_table_bytes(int n_markers) {
  if (n_markers < 4) return 64;
  return (size_t)n_markers * (n_markers - 1) * (n_markers - 2) * k2_entry_doubles(n_markers) * sizeof(double);
}

// How many blocks of a frame share the marker permutations when each keeps its slice of the table in LDS (plain
// kernel): slices of at most ~7 KB (so that four 256-thread blocks with their back-projection columns still fit a CU),
// at most 16 of them, whole blocks of six permutations; 0 = no slicing (<= 5 markers: the whole table is 5 KB and the
// scan-carrying variant copies it; >= 11 markers: a 16th of the table is larger than that).
int k2_table_slices(int n_markers) {
  if (n_markers < 6 || n_markers > 10) return 0;
  const int n_perms = n_markers * (n_markers - 1) * (n_markers - 2);
  const size_t bytes = (size_t)n_perms * (k2_entry_doubles(n_markers) - 12) * sizeof(double);
  int n = (int)((bytes + 7167) / 7168);
  if (n < 2) n = 2;
  if (n > 16) n = 16;
  if (n > n_perms / 6) n = n_perms / 6;
  return n;
}

// ---- image scan riding inside the voting kernel ----------------------------------------------
// The voting kernel is FP64-VALU bound and leaves the memory pipeline idle; the image scan is HBM
// bound and needs almost no VALU.  Instead of running the two side by side as separate kernels
// (their waves then fight for VGPR space: three 168-VGPR voting waves fill a SIMD), every voting
// wave also streams a share of the NEXT sub-batch's pixels: `global_load_lds_dwordx4` (gfx950 LDS
// DMA) moves 16 B per lane straight from HBM into a per-wave LDS staging area — no VGPRs are held
// while the loads are in flight — and at a few "service points" between pieces of P3P arithmetic the
// wave tests the staged segments against the threshold (SWAR + ballot, the same arithmetic as
// k1a_scan), writes the flag words and starts the next round of loads.
#ifndef K2_SCAN_R
#define K2_SCAN_R 4  // 1 KiB wave-loads per round = KiB of staging LDS per wave
#endif
struct ScanArgs {
  const uint4* px;   // pixels of the region to scan, 16-byte segments
  u64* flags;        // one bit per segment
  int n_chunks;      // full chunks of 64 * K2_SCAN_R segments (the caller scans the remainder separately)
  ThrTest thr;       // threshold test constants (make_thr_test)
};
struct ScanRider {
  const uint4* px;
  u64* flags;
  uint4* stage;  // this wave's staging area in LDS: [K2_SCAN_R][64] segments
  int c, stride, n_chunks;
  ThrTest thr;
  bool pending;
  __device__ __forceinline__ void init(const ScanArgs& a, unsigned char* lds_stage) {
    const int waves_per_block = blockDim.x >> 6;
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform -> SGPRs
    px = a.px;
    flags = a.flags;
    thr = a.thr;
    n_chunks = a.n_chunks;
    stage = reinterpret_cast<uint4*>(lds_stage) + (size_t)wave_in_block * (K2_SCAN_R * 64);
    c = (int)blockIdx.x * waves_per_block + wave_in_block;
    stride = (int)gridDim.x * waves_per_block;
    pending = false;
  }
  // start the next round.  vmcnt counts in order, so an ordinary global load issued behind a round would wait
  // for the round's HBM latency: the voting loop therefore reads its tables from LDS only.
  __device__ __forceinline__ void issue() {
    if (pending || c >= n_chunks) return;
    asm volatile("" ::: "memory");
    // one global base address and one LDS base (M0) per round: the instruction's immediate offset moves BOTH
    // the memory address and the LDS address, and chunk layout == staging layout (1 KiB per load)
    const uint4* p = px + (size_t)c * (K2_SCAN_R * 64) + (threadIdx.x & 63) + 64 * (K2_SCAN_R / 2);
    uint4* l = stage + 64 * (K2_SCAN_R / 2);
    dma_rounds<0>(p, l);
    asm volatile("" ::: "memory");
    pending = true;
  }
  template <int K>
  static __device__ __forceinline__ void dma_rounds(const uint4* p, uint4* l) {
    if constexpr (K < K2_SCAN_R) {
      // cache policy sc0 | nt (aux = 1 | 2): the pixels are read exactly once — streaming them past the caches
      // took the fused kernel from 1.045 to 0.95 ms per 5.9 GB on MI355X (nt alone 0.98, sc0 alone 1.035)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)l, 16,
                                       (K - K2_SCAN_R / 2) * 1024, 3);
      dma_rounds<K + 1>(p, l);
    }
  }
  // test the staged round and write its flag words
  __device__ __forceinline__ void consume() {
    if (!pending) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA writes have landed in LDS
    const int lane = threadIdx.x & 63;
    u64 b[K2_SCAN_R];
    // Three levels, each a necessary condition for the next (bytewise OR >= every operand byte): the OR of the
    // whole round's words (most rounds of a dark frame stop here), then per segment, then the exact test.
    uint4 v[K2_SCAN_R];
    unsigned all = 0;
#pragma unroll
    for (int k = 0; k < K2_SCAN_R; ++k) {
      v[k] = stage[64 * k + lane];
      all |= v[k].x | v[k].y | v[k].z | v[k].w;
      b[k] = 0;
    }
    const unsigned hit = thr.sel ? gt_word_c<true>(all, thr.kk) : gt_word_c<false>(all, thr.kk);  // (uniform select)
    if (__ballot((hit & 0x80808080u) != 0)) {  // wave-uniform
#pragma unroll
      for (int k = 0; k < K2_SCAN_R; ++k) {
        b[k] = __ballot(maybe_gt16(v[k], thr) != 0);
        if (b[k]) b[k] = __ballot(any_gt16(v[k], thr) != 0);
      }
    }
    asm volatile("" ::: "memory");  // staging reads are done before the next round overwrites them
    if (lane == 0) {
      u64* out = flags + (size_t)c * K2_SCAN_R;
#pragma unroll
      for (int k = 0; k < K2_SCAN_R; ++k) out[k] = b[k];
    }
    pending = false;
    c += stride;
  }
  __device__ __forceinline__ void drain() {  // the rest of this wave's share, nothing to hide behind any more
    for (;;) {
      issue();
      if (!pending) break;
      consume();
    }
  }
};
// small helpers of the voting item; the host-tier build (tests/host/vote_host.cpp) brings its own one-lane versions
__device__ __forceinline__ bool k2_isfinite(double x) { return __builtin_isfinite(x); }
__device__ __forceinline__ f32x2 k2_pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float k2_fminf(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ unsigned k2_cvt_pk_u8(float x, unsigned byte, unsigned into) {  // v_cvt_pk_u8_f32
  return __builtin_amdgcn_cvt_pk_u8_f32(x, byte, into);
}
__device__ __forceinline__ float k2_rsqf(float x) { return __builtin_amdgcn_rsqf(x); }    // v_rsq_f32 (1 ulp)
__device__ __forceinline__ float k2_sqrtf(float x) { return __builtin_amdgcn_sqrtf(x); }  // v_sqrt_f32 (1 ulp)
struct NoRider {
  __device__ __forceinline__ void consume() {}
  __device__ __forceinline__ void issue() {}
  __device__ __forceinline__ void drain() {}
};

// Everything of computePoses that depends on the detection triple only (p3p.cpp:82-121, 143-154): the tau frame T
// (as K T^T, t[0..8]), f_1, f_2, b, f_1 / f_2 (t[9..12]) and the triple's indices + the swap flag (packed).
__device__ __forceinline__ void k2_triple_entry(const double (*iv)[3], int n_d, int idx, double fx, double fy, double cx,
                                                double cy, double* __restrict__ t, unsigned& packed) {
  int c0, c1, c2;
  unrank_combo3(idx, n_d, c0, c1, c2);
  const V3 fa = {iv[c0][0], iv[c0][1], iv[c0][2]}, fb = {iv[c1][0], iv[c1][1], iv[c1][2]},
           fc = {iv[c2][0], iv[c2][1], iv[c2][2]};
  V3 f1 = fa, f2 = fb;
  V3 e1 = f1;
  V3 e3 = cross(f1, f2);
  e3 = vdiv(e3, norm(e3));
  V3 e2 = cross(e3, e1);
  M3 T = {e1, e2, e3};
  V3 f3 = mul(T, fc);
  unsigned swap = 0;
  if (f3.z > 0.0) {
    swap = 1;
    f1 = fb;
    f2 = fa;
    e1 = f1;
    e3 = cross(f1, f2);
    e3 = vdiv(e3, norm(e3));
    e2 = cross(e3, e1);
    T = {e1, e2, e3};
    f3 = mul(T, fc);
  }
  const double cos_beta = dot(f1, f2);
  double b = 1 / (1 - cos_beta * cos_beta) - 1;
  b = (cos_beta < 0) ? -sqrt(b) : sqrt(b);
  // K T^T, row by row: a point w of the tau frame projects to (t[0..2].w, t[3..5].w) / (t[6..8].w) — the camera
  // matrix folded into the frame change once per triple instead of once per back-projection
  t[0] = fx * T.r0.x + cx * T.r0.z; t[1] = fx * T.r1.x + cx * T.r1.z; t[2] = fx * T.r2.x + cx * T.r2.z;
  t[3] = fy * T.r0.y + cy * T.r0.z; t[4] = fy * T.r1.y + cy * T.r1.z; t[5] = fy * T.r2.y + cy * T.r2.z;
  t[6] = T.r0.z; t[7] = T.r1.z; t[8] = T.r2.z;
  t[9] = f3.x / f3.z;
  t[10] = f3.y / f3.z;
  t[11] = b;
  t[12] = t[9] / t[10];
  packed = (unsigned)c0 | ((unsigned)c1 << 8) | ((unsigned)c2 << 16) | (swap << 24);
}

// The launch's list of hypotheses left to the strict arithmetic (VoteFixup) + the status word of the block's frame
struct K2SusDesc {
  unsigned* ctl;      // global: [0] entries appended by this launch, [1] entries lost to a full list (cumulative)
  u64* list;          // global list, K2_SUS_WORDS words per entry
  unsigned cap;       // entries the global list holds
  int* frame_status;  // the frame's detection-record status: set to MPE_FRAME_VOTE_LIST_FULL when an entry is lost
};
// What a voting work item reads of its frame and block (LDS in the kernels; plain arrays when the host-tier test runs
// this source, tests/test_vote_host.py).
struct K2Frame {
  const unsigned* trii;      // per staged triple: c0 | c1 << 8 | c2 << 16 | swap << 24
  const double (*tri)[13];   // per staged triple: K T^T rows, f_1, f_2, b, f_1 / f_2
  const double (*px)[2];     // undistorted detections
  const f32x2* pxf;          // the same in single precision (nearest-neighbour prefilter)
  double* q;                 // back-projections [2 * j + {0, 1}][lane] (plain variant)
  f32x2* qf;                 // their single-precision copies [j][lane]
  unsigned* hist;            // vote histogram of the frame
  const double* tab;         // marker-permutation table (global memory; plain variant)
  const double* ltab;        // its LDS copy, K2_LTAB doubles per permutation (scan-carrying variant)
  int n_d, nuo, nthr, tid, esz;
  double fx, fy, cx, cy, back_tol;
  float thr_pre;
  u64* vq;        // this wave's queue of deferred exact votes, K2_VQ_CAP entries of K2_VQ_WORDS u64 (scan variant;
                  // its fill count is a wave-uniform register of the caller)
  int vq_lanes;   // lanes that work the queue off together: 64 (1 when the host-tier test runs this source)
  int pj_base;    // first permutation of the table `tab` points at (0: the whole table; plain variant with LDS slices)
  // hypotheses this arithmetic does not decide itself (k2_sus_push): collected in a small list of the block (LDS: an
  // append there is a DS atomic — a global atomic with a return value would wait, in vmcnt order, for the scan rider's
  // loads in flight, and 30 - 45 % of the wave iterations have SOME lane that appends), moved to the launch's list in
  // global memory by k2_sus_flush at points where the whole block passes, worked off by k2_vote_fixup
  // The launch's list is described by a record in LDS (a copy of the kernel argument made once): its pointers are
  // only read when the block's list is flushed or full, so they do not sit in scalar registers across the voting loop
  // (the scan-carrying kernel uses every SGPR it has: with these fields passed by value it spilled 460 of them to
  // vector lanes and its launch went from 1.39 to 1.59 ms).
  const K2SusDesc* susd;
  bool fix;           // strict re-evaluation is on; false: every hypothesis is decided in this arithmetic
  int frame;          // index of the frame within the launch
  u64* sus_lds;       // the block's list
  unsigned* sus_lds_n;
  unsigned sus_lds_cap;
  // deferred plain variant: occupancy grid of the frame's detections, dilated by the prefilter radius (K2_GRID x K2_GRID
  // bits over their bounding box; cell (ix, iy) of a point (u, v): ix = (int)(u * ginv + gxo), iy likewise)
  const u64* grid;
  float ginv, gxo, gyo;
  const float (*trif)[12];  // per staged triple, single precision: the rows of G K T^T (k2_triple_f32), b
};

// i-th double of the LDS copy of the marker-permutation table (scan-carrying variant), K2_LTAB per permutation:
//   [0] p_1 [1] p_2 [2] d_12 [3] valid [4] packed marker indices [5..10] eta-frame unused markers (<= 2)
//   [11..13] the same markers in single precision (six floats, as the table entry packs them)
__device__ __forceinline__ double k2_ltab_value(const double* __restrict__ tab, int esz, int nuo, int i) {
  const int pe = i / K2_LTAB, fld = i - pe * K2_LTAB;
  if (fld >= 11) return (fld - 11 < (3 * nuo + 1) / 2) ? tab[(size_t)pe * esz + 18 + 3 * nuo + (fld - 11)] : 0.0;
  const int src = fld < 5 ? 12 + fld : 13 + fld;  // 12..16, 18..23
  return (fld < 5 + 3 * nuo) ? tab[(size_t)pe * esz + src] : 0.0;
}

// Nearest-neighbour prefilter: a detection can only vote if its exact distance to some back-projection is below
// tol; single precision places both points within 1e-3 px for any point that close to a detection (pixel
// coordinates < 4096), so "minimum single-precision distance <= tol (1 + 1e-4) + 0.05" is a safe necessary
// condition, and the exact double-precision search only runs for the few detections that pass it.
// (margin_px: 0.05 covers the rounding of double-precision points to single precision; the plain variant's deferred
//  path, whose back-projections are COMPUTED in single precision, passes 0.25)
__device__ __forceinline__ float k2_prefilter_threshold(double back_tol, double margin_px = 0.05) {
  const double tol_pre = back_tol * (1.0 + 1e-4) + margin_px;
  return (float)(tol_pre * tol_pre * (1.0 + 1e-5));
}

// ---- hypotheses handed to the strict arithmetic ------------------------------------------------------------------
// The fast arithmetic of this kernel (Newton-Raphson division / square root, Newton cube root, [R|C]-free
// back-projection with FMAs) and the strict one (IEEE operators, libm, the reference's statement order — k2_vote_strict,
// which shares its P3P with the validation kernel) differ by a few ulp per operation.  A vote can only come out
// differently where that difference is AMPLIFIED past the distance of a back-projection from the vote tolerance, so a
// hypothesis is not decided here but appended to a list, and re-evaluated by k2_vote_fixup with the strict functions,
// when
//   (a) a subtraction of Ferrari's method cancelled below 2^MPE_FERRARI_SUSPECT_EXP (9.3e-10) of its operands
//       (solve_quartic_lit2: whole hypothesis, all four roots; 0.06 % of the hypotheses), or, per root,
//       sin^2(theta) = 1 - root^2 or the vector (cn, cd) behind cot(alpha) is small for how well the quartic was
//       conditioned (K2_SUS_ROOT_BASE), or |cos(alpha)| < 1e-6 (the strict arithmetic takes it as
//       sqrt(1 - sin^2), which then has few digits);
//   (b) the distance of a detection to its nearest back-projection lies within 2^K2_SUS_BAND_EXP = 0.0156 px of the
//       tolerance (tested on the squares) — four times the 4e-3 px that (a) lets through — or the
//       nearest and the second nearest back-projection are that close to each other while in reach: only those
//       detections of that root go to the list; the other detections' votes (and whether any of them voted: the
//       triple's own three votes, pose_estimator.cpp:676-685) are cast here.
// Votes are integer adds, so the order in which the two kernels cast them does not matter.  The strict verdict
// REPLACES the fast one: with the list in place the histograms are those of k2_vote_strict (tests/soak_votes.py,
// test_default_votes_equal_strict_votes), at ~0.25 % of the hypotheses re-evaluated.  A full list (sized at > 100 times the
// expected rate by the host side) loses entries: those frames are voted again, whole, by the strict loop nest
// (k2_vote_relost; options "vote_fixup_overflow" / "vote_relost_frames" count the events and the frames).
#ifndef K2_SUS_BAND_EXP
#define K2_SUS_BAND_EXP (-6)  // the band's half width in pixels: 2^-6 = 0.0156 px, four times what (a) lets through
#endif
// per root, by how well the quartic was conditioned (MPE_QUARTIC_MID: roots good to ~2e-8, else to ~2e-11): the error
// of cos(theta) is divided by sin(theta) in the angle, that of (cn, cd) by its length relative to its operands
// With the quartic's worst cancellation 2^c (c = cancel_exp, > MPE_FERRARI_SUSPECT_EXP here) the two arithmetics' roots
// differ by up to ~2^(-52 - c); a back-projection moves by <= ~1800 px per unit of the ANGLE, so for a tenth of band
// (b), 4e-3 px, the angle may be off by 2.2e-6 = 2^-18.8: sin(theta) >= 2^(-52 - c + 18.8), i.e. sin^2(theta) and
// likewise |(cn, cd)|^2 / |operands|^2 must stay above 2^(-66 - 2c) — never less than 2^-33 (1.2e-10), at most 2^-12.
// Integer arithmetic on binary exponents (no double-precision literals: the loop has no scalar registers to spare).
#ifndef K2_SUS_ROOT_BASE
#define K2_SUS_ROOT_BASE (-66)
#define K2_SUS_ROOT_FLOOR (-33)
#define K2_SUS_COSA 1e-6f      // |cos(alpha)| below which a root is suspect
#endif
#define K2_SUS_WORDS 2
// entry: word 0 = frame | code << 32, word 1 = mask of the detections to decide; code = the hypothesis' detection and
// marker indices, the roots to evaluate (kmask) and whether the fast arithmetic already cast the triple's own votes
// for that root (any_fast; only with a single root in kmask)
__device__ __forceinline__ unsigned k2_sus_code(int c0, int c1, int c2, int p0, int p1, int p2, unsigned kmask, bool any_fast) {
  return (unsigned)(c0 | (c1 << 5) | (c2 << 10) | (p0 << 15) | (p1 << 19) | (p2 << 23)) | (kmask << 27) |
         ((unsigned)any_fast << 31);
}
// an entry that finds the global list full is LOST: its votes are cast nowhere.  The frame is marked
// (MPE_FRAME_VOTE_LIST_FULL in its status) and counted; k2_vote_relost, launched behind the fix-up kernel, votes every
// marked frame again with the strict kernel's loop nest (the histogram the default arithmetic has to equal anyway) and
// clears the mark — a full list costs time, never a pose.  The status only survives to the caller
// if that launch is skipped (it never is on the library's paths).
__device__ __forceinline__ void k2_sus_lost(const K2SusDesc& g) {
  atomicAdd(&g.ctl[1], 1u);
  *g.frame_status = MPE_FRAME_VOTE_LIST_FULL;
}
// append an entry to the block's list (to the global one directly when that is full: rare, slow, correct)
__device__ __forceinline__ void k2_sus_push(const K2Frame& F, unsigned code, unsigned detmask) {
  const u64 w0 = (u64)(unsigned)F.frame | ((u64)code << 32);
  const unsigned slot = atomicAdd(F.sus_lds_n, 1u);
  if (slot < F.sus_lds_cap) {
    F.sus_lds[(size_t)K2_SUS_WORDS * slot] = w0;
    F.sus_lds[(size_t)K2_SUS_WORDS * slot + 1] = (u64)detmask;
    return;
  }
  const K2SusDesc g = *F.susd;
  if (!g.ctl) return;  // (no list was supplied: launch_k2_vote refuses that for vote_arith != 0, see there)
  const unsigned gs = atomicAdd(&g.ctl[0], 1u);
  if (gs < g.cap) {
    g.list[(size_t)K2_SUS_WORDS * gs] = w0;
    g.list[(size_t)K2_SUS_WORDS * gs + 1] = (u64)detmask;
  } else {
    k2_sus_lost(g);
  }
}

// The exact half of the nearest-neighbour vote of ONE root of ONE hypothesis (pose_estimator.cpp:663-702) for the
// unused detections whose bit is set in `pass` (bit a = detection a got through the single-precision prefilter), the
// back-projections of the unused markers read through `qat(jj, u, v)`: exact double-precision search (first minimum),
// strict `< tol` decided on the squares (the square root is only taken inside the rounding band around tol^2), votes,
// and the triple's own three votes if any detection voted.  Detections in the suspect band (b) go to the list instead.
// The lane must be allowed to vote.
// cw = c0 | c1 << 8 | c2 << 16, pw = p0 | p1 << 8 | p2 << 16: the indices are unpacked where they are needed (the
// rare branches), so that they do not occupy six registers across the search
template <class QAt>
__device__ __forceinline__ void k2_vote_root_exact(const K2Frame& F, const unsigned cw, const unsigned pw,
                                                   unsigned pass, int k, QAt qat) {
  const double tol2 = F.back_tol * F.back_tol;
  // |d^2 - tol^2| <= 2 tol w with w = tol-independent 2^K2_SUS_BAND_EXP px
  const double band = F.fix ? ldexp(F.back_tol, K2_SUS_BAND_EXP + 1) : -1.0;  // (< 0: no detection ever is suspect)
  bool any = false;
  unsigned sus = 0;
  for (unsigned todo = pass; todo; todo &= todo - 1) {
    const int a = __builtin_ctz(todo);
    const double au = F.px[a][0], av = F.px[a][1];
    double best = INFINITY, second = INFINITY;  // (the runner-up: is the CHOICE of the marker safe?)
    int bj = 0;
    for (int jj = 0; jj < F.nuo; ++jj) {
      double bu, bv;
      qat(jj, bu, bv);
      const double du = au - bu, dv = av - bv;
      const double d2 = du * du + dv * dv;
      const bool nearer = d2 < best;
      second = nearer ? best : (d2 < second ? d2 : second);
      bj = nearer ? jj : bj;
      best = nearer ? d2 : best;
    }
    // suspect: the distance within the band around the tolerance, or a vote about to be cast for a marker whose
    // runner-up is as near (band < 0: screening is off, both tests are false)
    if (fabs(best - tol2) <= band || (best < tol2 && second - best <= band)) {
      sus |= 1u << a;
      continue;
    }
    bool within = best < tol2 * (1.0 - 1e-14);
    if (!within && best < tol2 * (1.0 + 1e-14)) within = sqrt(best) < F.back_tol;
    if (within) {
      // bj-th unused marker (ascending) -> marker index: skip over the sorted used indices
      const int p0 = pw & 0xFF, p1 = (pw >> 8) & 0xFF, p2 = (pw >> 16) & 0xFF;
      const int lo = min(p0, min(p1, p2)), hi = max(p0, max(p1, p2)), mid = p0 + p1 + p2 - lo - hi;
      int mi = bj;
      mi += (mi >= lo);
      mi += (mi >= mid);
      mi += (mi >= hi);
      atomicAdd(&F.hist[a * MPE_MAX_MARKERS + mi], 1u);
      any = true;
    }
  }
  if (sus)
    k2_sus_push(F, k2_sus_code(cw & 0xFF, (cw >> 8) & 0xFF, (cw >> 16) & 0xFF, pw & 0xFF, (pw >> 8) & 0xFF,
                               (pw >> 16) & 0xFF, 1u << k, any), sus);
  if (any) {  // pose_estimator.cpp:676-685
    atomicAdd(&F.hist[(cw & 0xFF) * MPE_MAX_MARKERS + (pw & 0xFF)], 1u);
    atomicAdd(&F.hist[((cw >> 8) & 0xFF) * MPE_MAX_MARKERS + ((pw >> 8) & 0xFF)], 1u);
    atomicAdd(&F.hist[((cw >> 16) & 0xFF) * MPE_MAX_MARKERS + ((pw >> 16) & 0xFF)], 1u);
  }
}
// ... with the (<= 2) back-projections passed by value (scan-carrying variant)
__device__ __forceinline__ void k2_vote_exact(const K2Frame& F, const unsigned cw, const unsigned pw, unsigned pass,
                                              double q0u, double q0v, double q1u, double q1v, int k) {
  k2_vote_root_exact(F, cw, pw, pass, k, [&](const int jj, double& bu, double& bv) {
    bu = jj == 0 ? q0u : q1u;
    bv = jj == 0 ? q0v : q1v;
  });
}

// Deferred exact votes (scan-carrying variant).  About 1 % of the (hypothesis, detection) pairs pass the prefilter,
// but in a wave of 64 independent hypotheses SOME lane does in every other iteration, and the whole wave then walks
// through the exact search and the vote with one or two lanes alive: that was 21 % of the voting kernel's time
// (0.68 -> 0.54 ms per 16 384 frames with everything behind the prefilter compiled out).  Instead a lane that has a
// candidate appends {back-projections, indices, prefilter mask} to its wave's small LDS queue (one entry per
// hypothesis root; slots come from a ballot, the fill count is a wave-uniform register) and the wave works the queue
// off with one entry per LANE whenever it is nearly full: the same exact test, the same votes (integer adds: any
// order).  A lane that finds the queue full votes on the spot.  Measured: 0.855 -> 0.833 ms per fused launch.
#define K2_VQ_CAP 28
#define K2_VQ_WORDS 2
// Round 4: the loop's back-projections are single precision (see K2SubF below), so an entry is {root, prefilter mask |
// staged triple, permutation, root number} and the exact evaluation rebuilds the root's back-substitution and its (<= 2)
// back-projections in double precision — the operations the loop itself used to run for every root.
// Suspect roots and hypotheses of the scan-carrying variant travel through the same queue (flag bits in the entry's
// index word) and reach the block's suspect list when the queue is worked off: a second, divergent append inside the
// root loop cost the kernel ~50 scalar-register reloads per root.
#define K2_VQ_ROOT_SUS (1u << 29)
#define K2_VQ_ITEM_SUS (1u << 30)
__device__ __forceinline__ u64 k2_vq_meta(int ti, int pj, int k, unsigned pass) {  // ti < 16, pj < 60
  return (u64)pass | ((u64)((unsigned)ti | ((unsigned)pj << 4) | ((unsigned)k << 10)) << 32);
}
__device__ __forceinline__ void k2_vote_flush(const K2Frame& F, int count);  // (behind k2_project_marker below)
// The block's list -> the launch's list in global memory: ONE returning global atomic per flush (thread 0), at a point
// every thread of the block passes and where no scan round is in flight.  s_base: one word of LDS for the broadcast.
#define K2_SUS_LDS_SCAN 30   // entries of the block's list, scan-carrying variant (one frame per block: ~4 on average)
#define K2_SUS_LDS_PLAIN 96  // plain variant, flushed after every chunk of staged triples
__device__ __forceinline__ void k2_sus_flush(const K2Frame& F, unsigned* s_base) {
  __syncthreads();
  const unsigned n = min(*F.sus_lds_n, F.sus_lds_cap);
  if (n == 0) return;  // (uniform over the block)
  const K2SusDesc g = *F.susd;
  if (!g.ctl) return;  // (uniform as well)
  if (F.tid == 0) *s_base = atomicAdd(&g.ctl[0], n);
  __syncthreads();
  const unsigned base = *s_base;
  for (unsigned i = (unsigned)F.tid; i < n; i += (unsigned)F.nthr) {
    if (base + i < g.cap) {
      g.list[(size_t)K2_SUS_WORDS * (base + i)] = F.sus_lds[(size_t)K2_SUS_WORDS * i];
      g.list[(size_t)K2_SUS_WORDS * (base + i) + 1] = F.sus_lds[(size_t)K2_SUS_WORDS * i + 1];
    } else {
      k2_sus_lost(g);
    }
  }
  __syncthreads();
  if (F.tid == 0) *F.sus_lds_n = 0;
  __syncthreads();
}

// ---- back-substitution of one root and back-projection of one marker (shared by the voting loop and by the deferred
//      evaluation of the plain variant, which must produce the same bits) -------------------------------------------
// p3p.cpp:193-213 without forming [R|C]:  cot_alpha = cn / cd;  sin_alpha = sqrt(1 / (cot^2 + 1)) = |cd| / hypot(cn, cd),
// cos_alpha = sign(cot) sqrt(1 - sin^2) = cn sign(cd) / hypot(cn, cd)
struct K2Sub {
  double cos_theta, sin_theta, cos_alpha, sin_alpha, Cx, Cy, Cz;
  double om, h2;  // 1 - root^2 and |(cn, cd)|^2: the quantities the suspect screen looks at
};
__device__ __forceinline__ K2Sub k2_back_substitute(double rt, double g1, double g2, double g3, double p_2, double d_12,
                                                    double b) {
  K2Sub S;
  const double cn = g1 - rt * p_2, cd = g2 * rt + g3;
  S.h2 = __builtin_fma(cn, cn, cd * cd);
  const double ih = rsqrt_nr(S.h2);
  S.cos_theta = rt;
  S.om = 1 - rt * rt;
  S.sin_theta = sqrt_nr(S.om);
  S.sin_alpha = fabs(cd) * ih;
  S.cos_alpha = (cd < 0 ? -cn : cn) * ih;
  const double dk = d_12 * __builtin_fma(S.sin_alpha, b, S.cos_alpha);
  const double sdk = S.sin_alpha * dk;
  S.Cx = S.cos_alpha * dk;
  S.Cy = S.cos_theta * sdk;
  S.Cz = S.sin_theta * sdk;
  return S;
}
// X_cam = T^T Rm (N (m - P1) - C_eta) through K: mk = the marker in the eta frame, tr = the rows of K T^T
__device__ __forceinline__ void k2_project_marker(const K2Sub& S, const double* mk, const double* tr, double& qu, double& qv) {
  const double v0 = mk[0] - S.Cx, v1 = mk[1] - S.Cy, v2 = mk[2] - S.Cz;
  const double g = __builtin_fma(S.cos_theta, v1, S.sin_theta * v2);
  const double w0 = -__builtin_fma(S.cos_alpha, v0, S.sin_alpha * g);
  const double w1 = __builtin_fma(S.sin_alpha, v0, -(S.cos_alpha * g));
  const double w2 = __builtin_fma(S.cos_theta, v2, -(S.sin_theta * v1));
  const double U = __builtin_fma(tr[0], w0, __builtin_fma(tr[1], w1, tr[2] * w2));  // K T^T w
  const double V = __builtin_fma(tr[3], w0, __builtin_fma(tr[4], w1, tr[5] * w2));
  const double Z = __builtin_fma(tr[6], w0, __builtin_fma(tr[7], w1, tr[8] * w2));
  const double iZ = rcp_nr(Z);
  qu = U * iZ;
  qv = V * iZ;
}

// works the scan-carrying variant's queue off, one entry per lane (at most one trip on the device: K2_VQ_CAP < 64)
__device__ __forceinline__ void k2_vote_flush(const K2Frame& F, int count) {
  wave_sync();
  const unsigned n = min((unsigned)count, (unsigned)K2_VQ_CAP);
  const unsigned lane = (unsigned)F.tid & 63u;
  for (unsigned i = lane; i < n; i += (unsigned)F.vq_lanes) {
    const u64* q = F.vq + (size_t)i * K2_VQ_WORDS;
    const double rt = __longlong_as_double((long long)q[0]);
    const u64 meta = q[1];
    const unsigned ix = (unsigned)(meta >> 32);
    const int ti = ix & 15, pj = (ix >> 4) & 63, k = (ix >> 10) & 3;
    const unsigned ii = F.trii[ti];
    const int c0 = ii & 0xFF, c1 = (ii >> 8) & 0xFF, c2 = (ii >> 16) & 0xFF;
    const int packed = (int)F.ltab[pj * K2_LTAB + 4];
    if (ix & (K2_VQ_ROOT_SUS | K2_VQ_ITEM_SUS)) {  // a suspect root / hypothesis on its way to the strict arithmetic
      const unsigned unused = (0xFFFFFFFFu >> (32 - F.n_d)) & ~((1u << c0) | (1u << c1) | (1u << c2));
      k2_sus_push(F, k2_sus_code(c0, c1, c2, packed & 0xFF, (packed >> 8) & 0xFF, (packed >> 16) & 0xFF,
                                 (ix & K2_VQ_ITEM_SUS) ? 0xFu : 1u << k, false), unused);
      continue;
    }
    const bool swap = (ii >> 24) & 1;
    const int r6 = pj % 6;
    const int pjs = swap ? (pj - r6 + (int)((0x134052u >> (4 * r6)) & 7u)) : pj;
    const double* lt = F.ltab + pjs * K2_LTAB;
    const double p_1 = lt[0], p_2 = lt[1], d_12 = lt[2];
    const double* tr = F.tri[ti];
    const double b = tr[11], f12 = tr[12];
    const double g1 = -f12 * p_1 + d_12 * b, g2 = -f12 * p_2, g3 = p_1 - d_12;
    const K2Sub S = k2_back_substitute(rt, g1, g2, g3, p_2, d_12, b);
    // isFinite([R C]) (pose_estimator.cpp:653; see the voting loop)
    if (!(k2_isfinite(S.Cx) && k2_isfinite(S.Cy) && k2_isfinite(S.Cz))) continue;
    double q0u = 0, q0v = 0, q1u = 0, q1v = 0;
    k2_project_marker(S, lt + 5, tr, q0u, q0v);
    if (F.nuo > 1) k2_project_marker(S, lt + 8, tr, q1u, q1v);
    k2_vote_exact(F, ii & 0xFFFFFFu, (unsigned)packed & 0xFFFFFFu, (unsigned)meta, q0u, q0v, q1u, q1v, k);
  }
  wave_sync();  // (the entries are read before the next ones overwrite them)
}

// the same back-substitution in single precision from (cn, cd, |(cn, cd)|^2, 1 - root^2) formed in double precision —
// the two cancellations of the step happen before the conversion — with -C_eta instead of C_eta
struct K2SubF {
  float ct, st, ca, sa, ncx, ncy, ncz;
};
__device__ __forceinline__ K2SubF k2_back_substitute_f32(float cn, float cd, float h2, float om, float rt, float d_12, float b) {
  K2SubF S;
  const float ih = k2_rsqf(h2);
  S.ct = rt;
  S.st = k2_sqrtf(om);
  S.sa = fabsf(cd) * ih;
  S.ca = (cd < 0 ? -cn : cn) * ih;
  const float ndk = -d_12 * __builtin_fmaf(S.sa, b, S.ca);
  const float nsdk = S.sa * ndk;
  S.ncx = S.ca * ndk;
  S.ncy = S.ct * nsdk;
  S.ncz = S.st * nsdk;
  return S;
}

// ---- plain variant, 3 .. 8 unused markers: single-precision back-projection + deferred exact evaluation -------------
// With 5 unused markers and 9 unused detections (C3) a root cost 150 double-precision operations for the
// back-projections, ten LDS column stores, and — in nearly every wave iteration, with a lane or two alive — the exact
// nearest-neighbour search: together 40 % of the kernel.  The roots themselves need double precision (the quartic
// cannot be screened in single precision: profiles/round3_study_f32_screen.json), but a back-projection FROM a
// double-precision root is well conditioned: it is evaluated in packed single precision for two markers at a time
// (the prefilter only asks "can this come within the tolerance": its margin grows from 0.05 to 0.25 px for the
// single-precision chain, whose error stays below 0.04 px for any point in front of or behind the camera with
// |z| >= 0.1 |X| — and a point closer to the image plane than that projects thousands of pixels away from every
// detection), and a (hypothesis, root) whose prefilter passes is appended to the wave's queue {root, indices, mask}.
// The queue is worked off with ONE ENTRY PER LANE: back-substitution and back-projections again, in double
// precision (k2_back_substitute / k2_project_marker: the same operations as the direct path), the exact search, the
// band screen and the votes (k2_vote_root_exact).
#ifndef K2_ON_GRID_COORD  // (test hook of the host-tier build: the single-precision chain against the double one)
#define K2_ON_GRID_COORD(...)
#endif
#define K2_DQ_CAP 128   // entries per wave (flushed above 64 at the end of an item; an entry that finds no room goes
                        // to the strict arithmetic's list instead)
#define K2_DQ_WORDS 2
#define K2_GRID 256     // occupancy grid: K2_GRID x K2_GRID bits (8 KB of LDS per block)
#define K2_GRID_WORDS (K2_GRID / 64)
__device__ __forceinline__ float k2_rcpf(float x) { return p3p_rcpf(x); }
__host__ __device__ constexpr bool k2_defers(bool scan, int np) { return !scan && np >= 2; }
// cell index iy * K2_GRID + ix of the grid coordinates (fx, fy): v_cvt_pk_u8_f32 converts, SATURATES to 0 .. 255 and
// packs in one instruction per coordinate (NaN -> 0); row / column 0 and 255 of the grid are never set, so everything
// outside the grid reads an empty cell
__device__ __forceinline__ unsigned k2_grid_cell(float fx, float fy) {
  return k2_cvt_pk_u8(fy, 1u, k2_cvt_pk_u8(fx, 0u, 0u));
}
// 1 if the cell of the grid coordinates is within the prefilter radius of a detection
__device__ __forceinline__ unsigned k2_grid_bit(const K2Frame& F, float fx, float fy) {
  const unsigned idx = k2_grid_cell(fx, fy);
  const unsigned word = reinterpret_cast<const unsigned*>(F.grid)[idx >> 5];
  return (word >> (idx & 31u)) & 1u;
}
// the block's grid: cells that a point within R of a detection can land in.  All threads; the grid must be zero; the
// parameters are returned through gp = {ginv, gxo, gyo}: grid coordinates of a pixel (u, v) = (u ginv + gxo, v ginv +
// gyo).  A coordinate c lands in cell floor(c) or floor(c) + 1 whatever the conversion's rounding, i.e. cell X takes
// points with c in [X - 1, X + 1): row Y of a detection's disc (centre (cx, cy), radius r, in cells) is set from
// floor(cx - hw) to floor(cx + hw) + 1, hw the disc's half-width over y in [Y - 1, Y + 1).  The single-precision chain
// that produces c is off by < 0.01 px (tests/test_vote_host.py; margin in R: 0.25 px).  248 cells span the detections'
// bounding box + 2 R, from cell 3 on, so that cells 0 and 255 — where everything outside the grid lands — stay empty.
__device__ __forceinline__ void k2_grid_build(const double (*px)[2], int n_d, double back_tol, u64* grid, float* gp, int tid,
                                              int nthr) {
  const float R = (float)(back_tol * (1.0 + 1e-4) + 0.25);
  float x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
  for (int a = 0; a < n_d; ++a) {  // (every thread: n_d <= 32 LDS reads, once per block)
    const float u = (float)px[a][0], v = (float)px[a][1];
    x0 = fminf(x0, u);
    x1 = fmaxf(x1, u);
    y0 = fminf(y0, v);
    y1 = fmaxf(y1, v);
  }
  const float span = fmaxf(x1 - x0, y1 - y0) + 2.0f * R;
  const float cell = fmaxf(span * (1.0f / (K2_GRID - 8)), 0.25f);
  const float inv = 1.0f / cell;
  const float ox = 3.0f - (x0 - R) * inv, oy = 3.0f - (y0 - R) * inv;
  if (tid == 0) {
    gp[0] = inv;
    gp[1] = ox;
    gp[2] = oy;
  }
  const float r = R * inv + 1e-3f;
  for (int a = tid; a < n_d; a += nthr) {
    const float u = (float)px[a][0], v = (float)px[a][1];
    if (!(u == u && v == v)) continue;
    const float cxg = u * inv + ox, cyg = v * inv + oy;  // (>= 3: truncation is floor)
    const int iy0 = max(1, (int)(cyg - r)), iy1 = min(K2_GRID - 2, (int)(cyg + r) + 1);
    for (int iy = iy0; iy <= iy1; ++iy) {
      const float dy = fmaxf(0.f, fmaxf((float)(iy - 1) - cyg, cyg - (float)(iy + 1)));
      const float hh = r * r - dy * dy;
      if (!(hh >= 0.f)) continue;
      const float hw = sqrtf(hh);
      const int ix0 = max(1, (int)(cxg - hw)), ix1 = min(K2_GRID - 2, (int)(cxg + hw) + 1);
      for (int w = ix0 >> 6; w <= (ix1 >> 6); ++w) {
        const int lo = max(ix0, 64 * w) - 64 * w, hi = min(ix1, 64 * w + 63) - 64 * w;
        const u64 m = (hi == 63 ? ~0ull : ((1ull << (hi + 1)) - 1)) & ~((1ull << lo) - 1);
        atomicOr(&grid[iy * K2_GRID_WORDS + w], m);
      }
    }
  }
}
// per staged triple, single precision: G K T^T with G = [ginv 0 gxo; 0 ginv gyo; 0 0 1] — a point's GRID coordinates are
// (U / Z, V / Z) of this matrix times the point in the tau frame.  Layout: rows 0 and 1 column-wise as pairs
// {T00, T10} {T01, T11} {T02, T12} (the operands of the packed instructions), row 2, b
__device__ __forceinline__ void k2_triple_f32(const double* T, const float* gp, float* o) {
  const double gi = gp[0], gx = gp[1], gy = gp[2];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    o[2 * c] = (float)(gi * T[c] + gx * T[6 + c]);
    o[2 * c + 1] = (float)(gi * T[3 + c] + gy * T[6 + c]);
    o[6 + c] = (float)T[6 + c];
  }
  o[9] = (float)T[11];
  o[10] = o[11] = 0.f;
}
// works the wave's queue off, one entry per lane: everything the voting loop knew about the root is rebuilt from the
// staged triple ti, the permutation pj and the root's value — back-substitution, double-precision back-projections,
// the single-precision prefilter over the unused detections, the exact search, band screen and votes
// `all` = false: only full passes (every lane an entry); what is left moves to the front of the queue and `count` says
// how many.  Called after every root with 64 or more entries queued, so that the queue (2 x 64 entries) never overflows.
template <int NP>
__device__ __forceinline__ void k2_defer_flush(const K2Frame& F, int& count, bool all) {
  wave_sync();
  const unsigned n = min((unsigned)count, (unsigned)K2_DQ_CAP);
  const unsigned W = (unsigned)F.vq_lanes;
  const unsigned lane = (unsigned)F.tid & (W - 1u);
  const unsigned n_do = all ? n : n - n % W;  // (wave-uniform)
  for (unsigned i = lane; i < n_do; i += W) {
    const u64* q = F.vq + (size_t)i * K2_DQ_WORDS;
    const double rt = __longlong_as_double((long long)q[0]);
    const unsigned meta = (unsigned)q[1];
    const int ti = meta & 0xFF, pj = (meta >> 8) & 0xFFF, k = (meta >> 20) & 3;
    const unsigned ii = F.trii[ti];
    const int c0 = ii & 0xFF, c1 = (ii >> 8) & 0xFF, c2 = (ii >> 16) & 0xFF;
    const bool swap = (ii >> 24) & 1;
    const int packed = (int)F.tab[(size_t)(pj - F.pj_base) * F.esz + 16];
    const int r6 = pj % 6;
    const int pjs = swap ? (pj - r6 + (int)((0x134052u >> (4 * r6)) & 7u)) : pj;
    const double* e = F.tab + (size_t)(pjs - F.pj_base) * F.esz;
    const double p_1 = e[12], p_2 = e[13], d_12 = e[14];
    const double* tr = F.tri[ti];
    const double b = tr[11], f12 = tr[12];
    const double g1 = -f12 * p_1 + d_12 * b, g2 = -f12 * p_2, g3 = p_1 - d_12;
    const K2Sub S = k2_back_substitute(rt, g1, g2, g3, p_2, d_12, b);
    // isFinite([R C]) (pose_estimator.cpp:653; see the voting loop): the loop's single-precision chain sends whatever
    // it cannot evaluate here
    if (!(k2_isfinite(S.Cx) && k2_isfinite(S.Cy) && k2_isfinite(S.Cz))) continue;
    f32x2 pfu[NP], pfv[NP];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      pfu[pp] = pfv[pp] = f32x2{INFINITY, INFINITY};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j = 2 * pp + h;
        if (j < F.nuo) {
          double qu, qv;
          k2_project_marker(S, e + 18 + 3 * j, tr, qu, qv);
          if (h == 0) {
            pfu[pp].x = (float)qu;
            pfv[pp].x = (float)qv;
          } else {
            pfu[pp].y = (float)qu;
            pfv[pp].y = (float)qv;
          }
        }
      }
    }
    const unsigned unused = (0xFFFFFFFFu >> (32 - F.n_d)) & ~((1u << c0) | (1u << c1) | (1u << c2));
    unsigned pass = 0;
    for (unsigned m = unused; m; m &= m - 1) {
      const int a = __builtin_ctz(m);
      const f32x2 af = F.pxf[a];
      float mn = INFINITY;
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        const f32x2 du = f32x2{af.x, af.x} - pfu[pp], dv = f32x2{af.y, af.y} - pfv[pp];
        const f32x2 d2 = k2_pk_fma(dv, dv, du * du);
        mn = k2_fminf(mn, k2_fminf(d2.x, d2.y));  // (a NaN distance never wins, as in the exact search)
      }
      pass |= (mn <= F.thr_pre) ? (1u << a) : 0u;
    }
    if (pass)
      k2_vote_root_exact(F, ii & 0xFFFFFFu, (unsigned)packed & 0xFFFFFFu, pass, k, [&](const int jj, double& bu, double& bv) {
        k2_project_marker(S, e + 18 + 3 * jj, tr, bu, bv);
      });
  }
  wave_sync();
  if (n_do < n) {  // (fewer than W entries: one per lane)
    u64 a = 0, b = 0;
    const bool mine = n_do + lane < n;
    if (mine) {
      a = F.vq[(size_t)(n_do + lane) * K2_DQ_WORDS];
      b = F.vq[(size_t)(n_do + lane) * K2_DQ_WORDS + 1];
    }
    wave_sync();
    if (mine) {
      F.vq[(size_t)lane * K2_DQ_WORDS] = a;
      F.vq[(size_t)lane * K2_DQ_WORDS + 1] = b;
    }
    wave_sync();
  }
  count = (int)(n - n_do);
}

// One work item = (staged detection triple ti, marker permutation pj): quartic coefficients (p3p.cpp:171-185),
// Ferrari, and for each root the back-projection of the unused markers and the nearest-neighbour votes
// (pose_estimator.cpp:596-702).  `live` = false: compute on, never vote (wave-uniform loop of the rider variant).
// NP (plain variant): the single-precision copies of the back-projections stay in REGISTERS as NP packed marker pairs
// (2 NP >= the number of unused markers) instead of LDS columns that every (detection, marker) pair of the prefilter
// would read again; 0 = the LDS columns (more than 8 unused markers).
template <bool SCAN, int NP = 0, class Rider>
__device__ __forceinline__ void k2_vote_item(const K2Frame& F, int ti, int pj, bool live, Rider& rider, int& vq_count) {
  const unsigned ii = F.trii[ti];
  const int c0 = ii & 0xFF, c1 = (ii >> 8) & 0xFF, c2 = (ii >> 16) & 0xFF;
  const bool swap = (ii >> 24) & 1;
  const int packed = SCAN ? (int)F.ltab[pj * K2_LTAB + 4]
                          : (int)F.tab[(size_t)(pj - F.pj_base) * F.esz + 16];  // marker indices of this permutation
  const int p0 = packed & 0xFF, p1 = (packed >> 8) & 0xFF, p2 = (packed >> 16) & 0xFF;
  const int r6 = pj % 6;
  const int pjs = swap ? (pj - r6 + (int)((0x134052u >> (4 * r6)) & 7u)) : pj;  // kSwapRow packed
  // e[12..] of the global table entry; in the scan-carrying variant e points into the LDS copy, shifted so
  // that the SAME indices work for p_1 p_2 d_12 valid (12..15), and the markers are read through lt below
  const double* lt = SCAN ? F.ltab + pjs * K2_LTAB : nullptr;
  const double* e = SCAN ? lt - 12 : F.tab + (size_t)(pjs - F.pj_base) * F.esz;
  // UNI: every lane of the wave runs the whole item (lanes without a valid hypothesis compute on harmlessly and are
  // barred from voting): the scan rider's rounds and the ballots that hand out queue slots need all 64 lanes
  constexpr bool UNI = SCAN || k2_defers(SCAN, NP);
  if (e[15] == 0.0) {  // collinear world points: computePoses returns -1
    if constexpr (UNI)
      live = false;
    else
      return;
  }
  const double p_1 = e[12], p_2 = e[13], d_12 = e[14];
  const double* tr = F.tri[ti];
  const double f_1 = tr[9], f_2 = tr[10], b = tr[11], f12 = tr[12];

  const double f_1_pw2 = f_1 * f_1, f_2_pw2 = f_2 * f_2;
  const double p_1_pw2 = p_1 * p_1, p_1_pw3 = p_1_pw2 * p_1, p_1_pw4 = p_1_pw3 * p_1;
  const double p_2_pw2 = p_2 * p_2, p_2_pw3 = p_2_pw2 * p_2, p_2_pw4 = p_2_pw3 * p_2;
  const double d_12_pw2 = d_12 * d_12, b_pw2 = b * b;
  const double F0 = -f_2_pw2 * p_2_pw4 - p_2_pw4 * f_1_pw2 - p_2_pw4;
  const double F1 = 2 * p_2_pw3 * d_12 * b + 2 * f_2_pw2 * p_2_pw3 * d_12 * b - 2 * f_2 * p_2_pw3 * f_1 * d_12;
  const double F2 = -f_2_pw2 * p_2_pw2 * p_1_pw2 - f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2 -
                    f_2_pw2 * p_2_pw2 * d_12_pw2 + f_2_pw2 * p_2_pw4 + p_2_pw4 * f_1_pw2 +
                    2 * p_1 * p_2_pw2 * d_12 + 2 * f_1 * f_2 * p_1 * p_2_pw2 * d_12 * b -
                    p_2_pw2 * p_1_pw2 * f_1_pw2 + 2 * p_1 * p_2_pw2 * f_2_pw2 * d_12 -
                    p_2_pw2 * d_12_pw2 * b_pw2 - 2 * p_1_pw2 * p_2_pw2;
  const double F3 = 2 * p_1_pw2 * p_2 * d_12 * b + 2 * f_2 * p_2_pw3 * f_1 * d_12 -
                    2 * f_2_pw2 * p_2_pw3 * d_12 * b - 2 * p_1 * p_2 * d_12_pw2 * b;
  const double F4 = -2 * f_2 * p_2_pw2 * f_1 * p_1 * d_12 * b + f_2_pw2 * p_2_pw2 * d_12_pw2 +
                    2 * p_1_pw3 * d_12 - p_1_pw2 * d_12_pw2 + f_2_pw2 * p_2_pw2 * p_1_pw2 - p_1_pw4 -
                    2 * f_2_pw2 * p_2_pw2 * p_1 * d_12 + p_2_pw2 * f_1_pw2 * p_1_pw2 +
                    f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2;
  rider.issue();  // P0: first scan round of the item (nothing is staged here: P6 consumed the last one)
  double root[4];
  int cancel_exp;
  solve_quartic_lit2(F0, F1, F2, F3, F4, root, [&]() {
    rider.consume();
    rider.issue();
  }, cancel_exp);
  rider.consume();  // P1
  rider.issue();
  // the detections outside the triple, as a bit mask (n_d <= 32)
  const unsigned unused = (0xFFFFFFFFu >> (32 - F.n_d)) & ~((1u << c0) | (1u << c1) | (1u << c2));
  const bool fix = F.fix;  // (uniform) suspect hypotheses are decided by the strict arithmetic
  // (a): Ferrari cancelled — all four roots to the list (scan-carrying variant: as a queue entry of the first root)
  bool item_sus = fix && cancel_exp < MPE_FERRARI_SUSPECT_EXP && live;
  if constexpr (!SCAN) {
    if (item_sus) {
      k2_sus_push(F, k2_sus_code(c0, c1, c2, p0, p1, p2, 0xFu, false), unused);
      if constexpr (UNI) {
        live = false;
        item_sus = false;
      } else {
        return;
      }
    }
  }
  // root-independent parts of cot_alpha (p3p.cpp:195-196), f_1/f_2 folded into one quotient
  const double g1 = -f12 * p_1 + d_12 * b, g2 = -f12 * p_2, g3 = p_1 - d_12;
  // (cn, cd) below is suspect when it keeps less than 1e-2 of its operands
  const int root_thr = max(K2_SUS_ROOT_FLOOR, K2_SUS_ROOT_BASE - 2 * cancel_exp);  // (binary exponent, see above)
  const int hs_exp = p3p_expo(__builtin_fma(g1, g1, p_2 * p_2) + __builtin_fma(g2, g2, g3 * g3)) + root_thr;
  const int om_exp = 1023 + root_thr;
  // scan-carrying variant: the first two of them (all of them in a 5-detection frame) stay in registers for the four
  // roots' prefilters; a missing second one sits at infinity and passes no test
  unsigned rest = unused, lsb0 = 0, lsb1 = 0;
  f32x2 af0 = {INFINITY, INFINITY}, af1 = {INFINITY, INFINITY};
  if constexpr (SCAN) {
    lsb0 = rest & (0u - rest);
    rest ^= lsb0;
    lsb1 = rest & (0u - rest);
    rest ^= lsb1;
    af0 = F.pxf[__builtin_ctz(lsb0 | 0x80000000u)];
    const f32x2 t1 = F.pxf[__builtin_ctz(lsb1 | 0x80000000u)];
    af1 = lsb1 ? t1 : af1;
  }
  // Everything behind the roots is evaluated with fused multiply-adds: this part never was in the reference's
  // operation order ([R|C]-free back-projection), its results feed only the `< tol` test, and a vote can only change
  // when a distance sits within ~1e-13 px of the tolerance.  (The quartic above keeps the literal order.)

#pragma unroll 1
  for (int k = 0; k < 4; ++k) {
    rider.consume();  // P2..P5 (no-op when nothing is staged)
    // next scan round: nothing of the voting loop waits on vmcnt (table and triples are in LDS)
    rider.issue();
    const double rt = k == 0 ? root[0] : (k == 1 ? root[1] : (k == 2 ? root[2] : root[3]));
    constexpr bool DEFER = k2_defers(SCAN, NP);
    constexpr bool F32 = DEFER || SCAN;  // single precision behind the root's two cancellations, exact evaluation deferred
    K2Sub S;    // p3p.cpp:193-213
    K2SubF Sf;  // (F32 variants)
    double om, h2;
    float cos_alpha_f;
    if constexpr (F32) {
      const double cn = g1 - rt * p_2, cd = g2 * rt + g3;
      h2 = __builtin_fma(cn, cn, cd * cd);
      om = 1 - rt * rt;
      Sf = k2_back_substitute_f32((float)cn, (float)cd, (float)h2, (float)om, (float)rt, (float)d_12, F.trif[ti][9]);
      cos_alpha_f = Sf.ca;
    } else {
      S = k2_back_substitute(rt, g1, g2, g3, p_2, d_12, b);
      om = S.om;
      h2 = S.h2;
      cos_alpha_f = (float)S.cos_alpha;
    }
    // (a), per root — BEFORE the finiteness test: a root within rounding of +-1 is finite in one arithmetic and NaN
    // (|root| > 1: sqrt of a negative number) in the other.  sin(theta) = sqrt(1 - root^2) at its branch point, (cn, cd)
    // cancelled, or cos(alpha) so small that the strict arithmetic's sqrt(1 - sin(alpha)^2) has no digits: this root
    // goes to the list with all unused detections
    const bool root_sus = fix && live && (item_sus || p3p_expo(om) < om_exp || p3p_expo(h2) < hs_exp ||
                                          fabsf(cos_alpha_f) < K2_SUS_COSA);
    bool root_listed = false;
    if constexpr (!SCAN) {
      if (root_sus) {
        k2_sus_push(F, k2_sus_code(c0, c1, c2, p0, p1, p2, 1u << k, false), unused);
        if constexpr (UNI)
          root_listed = true;
        else
          continue;
      }
    }
    // isFinite([R C]) (pose_estimator.cpp:653): a product is finite only if every factor is (0 * inf = NaN), so
    // R (products of the four sines / cosines with the finite frames) and C are finite iff C_eta is
    // (single-precision variants: a root outside [-1, 1] or NaN has no pose; whatever else is not finite in single
    //  precision goes to the queue, whose double-precision evaluation makes this test)
    bool finite_pose = true;
    if constexpr (F32) {
      finite_pose = om >= 0.0;
    } else if (!(k2_isfinite(S.Cx) && k2_isfinite(S.Cy) && k2_isfinite(S.Cz))) {
      if constexpr (UNI)
        finite_pose = false;
      else
        continue;
    }
    const bool may_vote = live && finite_pose && !root_listed;
    constexpr int NPA = NP > 0 ? NP : 1;
    f32x2 pfu[NPA], pfv[NPA];  // plain variant, NP > 0: (u, u) and (v, v) of marker pair p; missing markers at infinity
#pragma unroll
    for (int pp = 0; pp < NPA; ++pp) pfu[pp] = pfv[pp] = f32x2{INFINITY, INFINITY};
    auto back_project = [&](const int j) {
      const double* mk = e + 18 + 3 * j;
      double qu, qv;
      k2_project_marker(S, mk, tr, qu, qv);
      {
        F.q[(2 * j) * F.nthr + F.tid] = qu;
        F.q[(2 * j + 1) * F.nthr + F.tid] = qv;
        if constexpr (NP > 0) {
          if (j & 1) {
            pfu[j >> 1].y = (float)qu;
            pfv[j >> 1].y = (float)qv;
          } else {
            pfu[j >> 1].x = (float)qu;
            pfv[j >> 1].x = (float)qv;
          }
        } else {
          F.qf[j * F.nthr + F.tid] = f32x2{(float)qu, (float)qv};
        }
      }
    };
    unsigned grid_hits = 0;  // deferred plain variant: some back-projection falls into a cell near a detection
    f32x2 uv0 = {INFINITY, INFINITY}, uv1 = {INFINITY, INFINITY};  // scan-carrying variant: the (<= 2) back-projections
    bool chain_ok = true;  // the single-precision chain produced finite numbers
    if constexpr (F32) {
      // M = (G K T^T) Rm and -M C_eta once per root (22 instructions), then a marker's coordinates are 3 packed + 3 plain
      // multiply-adds, a reciprocal and a packed multiply.  Deferred plain variant: GRID coordinates (G = the map from
      // pixels to cells), the cell by one conversion per coordinate, one LDS read and one bit test (see K2_DQ_CAP above);
      // scan-carrying variant: pixels (G = 1).
      const float* tf = F.trif[ti];
      const f32x2 c0 = {tf[0], tf[1]}, c1 = {tf[2], tf[3]}, c2 = {tf[4], tf[5]};
      const float z0 = tf[6], z1 = tf[7], z2 = tf[8];
      const float ct = Sf.ct, st = Sf.st, ca = Sf.ca, sa = Sf.sa;
      const f32x2 M0 = k2_pk_fma(c1, f32x2{sa, sa}, c0 * f32x2{-ca, -ca});
      const f32x2 A = k2_pk_fma(c0, f32x2{sa, sa}, c1 * f32x2{ca, ca});
      const f32x2 M1 = k2_pk_fma(A, f32x2{-ct, -ct}, c2 * f32x2{-st, -st});
      const f32x2 M2 = k2_pk_fma(A, f32x2{-st, -st}, c2 * f32x2{ct, ct});
      const float Mz0 = __builtin_fmaf(z1, sa, -(z0 * ca));
      const float Az = __builtin_fmaf(z0, sa, z1 * ca);
      const float Mz1 = -__builtin_fmaf(ct, Az, st * z2);
      const float Mz2 = __builtin_fmaf(-st, Az, ct * z2);
      const f32x2 nd = k2_pk_fma(M0, f32x2{Sf.ncx, Sf.ncx}, k2_pk_fma(M1, f32x2{Sf.ncy, Sf.ncy}, M2 * f32x2{Sf.ncz, Sf.ncz}));
      const float ndz = __builtin_fmaf(Mz0, Sf.ncx, __builtin_fmaf(Mz1, Sf.ncy, Mz2 * Sf.ncz));
      // the markers in single precision
      const float* mf = SCAN ? reinterpret_cast<const float*>(lt + 11) : reinterpret_cast<const float*>(e + 18 + 3 * F.nuo);
      auto coords = [&](const int j) -> f32x2 {
        const float mx = mf[3 * j], my = mf[3 * j + 1], mz = mf[3 * j + 2];
        const f32x2 UV = k2_pk_fma(M0, f32x2{mx, mx}, k2_pk_fma(M1, f32x2{my, my}, k2_pk_fma(M2, f32x2{mz, mz}, nd)));
        const float Z = __builtin_fmaf(Mz0, mx, __builtin_fmaf(Mz1, my, __builtin_fmaf(Mz2, mz, ndz)));
        const float iZ = k2_rcpf(Z);
        return UV * f32x2{iZ, iZ};
      };
      if constexpr (SCAN) {
        uv0 = coords(0);
        if (F.nuo > 1) uv1 = coords(1);  // (uniform)
      } else {
#pragma unroll
        for (int j = 0; j < 2 * NP; ++j) {
          if (j < F.nuo) {  // (nuo is uniform over the block)
            const f32x2 g = coords(j);
            K2_ON_GRID_COORD(F, rt, g1, g2, g3, p_2, d_12, b, e + 18 + 3 * j, tr, g.x, g.y);
            grid_hits |= k2_grid_bit(F, g.x, g.y);
          }
        }
      }
      // anything this chain could not evaluate (overflow, underflow: (cn, cd) of 1e-20) is left to the double-precision one
      chain_ok = fabsf(nd.x + nd.y + ndz) < INFINITY;
      if (!chain_ok) grid_hits = 1u;
    } else if constexpr (NP > 0) {
#pragma unroll
      for (int j = 0; j < 2 * NP; ++j)
        if (j < F.nuo) back_project(j);  // (nuo is uniform over the block)
    } else {
      for (int j = 0; j < F.nuo; ++j) back_project(j);
    }
    // nearest back-projection for every unused detection (pose_estimator.cpp:862-906)
    if constexpr (SCAN) {
      // prefilter over all unused detections -> mask over the detections; the exact half is deferred (k2_vote_flush)
      // squared single-precision distances of one detection to marker 0 / marker 1, the smaller one against the
      // threshold (a NaN distance never wins, as in the exact search)
      auto near = [&](const f32x2 af) -> bool {
        const f32x2 d0 = af - uv0, d1 = af - uv1;
        const f32x2 s0 = d0 * d0, s1 = d1 * d1;
        return k2_fminf(s0.x + s0.y, s1.x + s1.y) <= F.thr_pre;
      };
      unsigned pass = (near(af0) ? lsb0 : 0u) | (near(af1) ? lsb1 : 0u);
      for (unsigned m = rest; m;) {  // (more than five detections)
        const unsigned lsb = m & (0u - m);
        const int a = __builtin_ctz(m);
        m ^= lsb;
        pass |= near(F.pxf[a]) ? lsb : 0u;
      }
      if (!chain_ok) pass = unused;  // (the double-precision evaluation decides)
      // slots by ballot: the queue's fill count is wave-uniform (a scalar register), no LDS atomic
      const bool want = root_sus || (pass != 0u && may_vote);
      const u64 bal = __ballot(want);
      if (bal != 0) {
        const unsigned slot = (unsigned)vq_count + (unsigned)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        vq_count += (int)__builtin_popcountll(bal);
        if (!want) {
        } else if (slot < (unsigned)K2_VQ_CAP) {
          u64* q = F.vq + (size_t)slot * K2_VQ_WORDS;
          q[0] = (u64)__double_as_longlong(rt);
          q[1] = k2_vq_meta(ti, pj, k, pass) | ((u64)(root_sus ? (item_sus ? K2_VQ_ITEM_SUS : K2_VQ_ROOT_SUS) : 0u) << 32);
        } else {
          // no room in the queue (6 or more lanes of the wave in one item: ~2e-5 of the hypotheses): this root, with
          // the detections that passed the prefilter, goes to the strict arithmetic like a suspect one — an exact
          // search inlined HERE, with the whole item's state alive, is what set the kernel's register peak
          k2_sus_push(F, k2_sus_code(c0, c1, c2, p0, p1, p2, item_sus ? 0xFu : 1u << k, false), root_sus ? unused : pass);
        }
      }
      if (item_sus) {  // (the whole hypothesis is on its way: nothing more of it here)
        live = false;
        item_sus = false;
      }
      continue;
    }
    if constexpr (k2_defers(SCAN, NP)) {
      // does ANY back-projection of this root fall into a cell near a detection?  (one grid lookup per marker instead of
      // a distance per detection x marker pair: 45 pairs at C3)  Then the root goes to the wave's queue: slots by
      // ballot (the fill count is wave-uniform); an entry beyond the queue's end goes to the strict arithmetic's list
      const bool hit = grid_hits != 0u;
      const bool want = hit && may_vote;
      const u64 bal = __ballot(want);
      if (bal != 0) {
        const unsigned slot = (unsigned)vq_count + (unsigned)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        vq_count += (int)__builtin_popcountll(bal);
        if (!want) {
        } else if (slot < (unsigned)K2_DQ_CAP) {
          u64* q = F.vq + (size_t)slot * K2_DQ_WORDS;
          q[0] = (u64)__double_as_longlong(rt);
          q[1] = (u64)((unsigned)ti | ((unsigned)pj << 8) | ((unsigned)k << 20));
        } else {  // (cannot happen with the flush below; kept as the capacity guard)
          k2_sus_push(F, k2_sus_code(c0, c1, c2, p0, p1, p2, 1u << k, false), unused);
        }
        if (vq_count >= F.vq_lanes) k2_defer_flush<NP>(F, vq_count, false);  // (wave-uniform)
      }
    } else {
      // the detections that are not part of the triple, ascending: a uniform trip count for the frame; the ones that get
      // through the single-precision prefilter are collected and decided together (k2_vote_root_exact)
      unsigned pass = 0;
      for (unsigned m = unused; m; m &= m - 1) {
        const int a = __builtin_ctz(m);
        const f32x2 af = F.pxf[a];
        float mn = INFINITY;
        if constexpr (NP > 0) {  // two markers per packed instruction, the back-projections from registers
#pragma unroll
          for (int pp = 0; pp < NP; ++pp) {
            const f32x2 du = f32x2{af.x, af.x} - pfu[pp], dv = f32x2{af.y, af.y} - pfv[pp];
            const f32x2 d2 = k2_pk_fma(dv, dv, du * du);
            mn = k2_fminf(mn, k2_fminf(d2.x, d2.y));  // (a NaN distance never wins, as in the exact search)
          }
        } else {  // (both coordinates per instruction, the back-projections from their LDS columns)
#pragma unroll 4
          for (int jj = 0; jj < F.nuo; ++jj) {
            const f32x2 qf = F.qf[jj * F.nthr + F.tid];
            f32x2 df = af - qf;
            df = df * df;
            const float d2f = df.x + df.y;
            mn = d2f < mn ? d2f : mn;  // (a NaN distance never wins, as in the exact search)
          }
        }
        pass |= (mn <= F.thr_pre) ? (1u << a) : 0u;
      }
      if (pass && may_vote)
        k2_vote_root_exact(F, ii & 0xFFFFFFu, (unsigned)packed & 0xFFFFFFu, pass, k, [&](const int jj, double& bu, double& bv) {
          bu = F.q[(2 * jj) * F.nthr + F.tid];
          bv = F.q[(2 * jj + 1) * F.nthr + F.tid];
        });
    }
  }
  rider.consume();  // P6: nothing of the scan is in flight while the next item fetches its table values
  if constexpr (SCAN) {
    if (vq_count >= K2_VQ_CAP - 8) {  // wave-uniform
      k2_vote_flush(F, vq_count);
      vq_count = 0;
    }
  }
}

// Voting kernel.  Work item = (detection triple, marker permutation).  Everything that depends
// only on the detection triple (tau frame T, f_1, f_2, b and the swap of p3p.cpp:100-121) is
// computed once per triple into LDS; everything that depends only on the marker permutation comes
// from the table above.  Per item: quartic coefficients (p3p.cpp:171-185), Ferrari, and for each
// root the back-projection of the unused markers WITHOUT forming [R|C]:
//     X_cam = R^T (m - C) = T^T Rm (N (m - P1) - C_eta),   Rm = the matrix of p3p.cpp:215-224,
// which is the same point as project2d(m, inverse(H)) of pose_estimator.cpp:660 up to rounding.
// Four waves per SIMD (<= 128 VGPRs) is what the voting kernels run at; with the suspect screening the allocator wants
// 130 - 138, and asked for four waves it keeps a handful of loop-invariant values (table pointers, parameters) in
// scratch OUTSIDE the item loop instead (checked in the ISA: every scratch access sits at loop depth <= 1; the plain
// instantiations, whose per-root part is single precision since round 4, need 113 - 120 and no scratch).
#ifndef K2_MIN_WAVES
#define K2_MIN_WAVES(NP) 4
#endif
// RANGE (forensics only, mpe_vote_items): frame f votes with the hypotheses whose flattened index — detection triple
// x P(n_m,3) + marker permutation, the reference's loop order — lies in [item_range[2f], item_range[2f+1]); the
// arithmetic of an item is the hot kernel's (same k2_vote_item).
template <bool SCAN, bool RANGE = false, int NP = 0>
__global__ __launch_bounds__(K2_THREADS, K2_MIN_WAVES(NP)) void k2_vote(mpe_detections* __restrict__ dets, SolveParams sp,
                                                      const double* __restrict__ tab, uint32_t* __restrict__ hist,
                                                      int splits, ScanArgs scan, const int* __restrict__ item_range,
                                                      int slice_tab, VoteFixup fixup) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ double s_px[MPE_MAX_DETECTIONS][2];
  __shared__ double s_iv[MPE_MAX_DETECTIONS][3];
  // detection triples staged per pass: 64, or 16 in the scan-carrying variant (LDS goes to the scan staging
  // and to an LDS copy of the marker table instead; more triples simply take more passes)
  constexpr int TRI = SCAN ? K2_TRI_CHUNK_SCAN : K2_TRI_CHUNK;
  __shared__ double s_tri[TRI][13];  // T rows (9), f_1, f_2, b, f_1/f_2
  __shared__ unsigned s_trii[TRI];   // c0 | c1 << 8 | c2 << 16 | swap << 24
  __shared__ unsigned s_hist[MPE_HIST_STRIDE];
  __shared__ f32x2 s_pxf[MPE_MAX_DETECTIONS];  // the detections in single precision (nearest-neighbour prefilter)
  constexpr int SUSN = SCAN ? K2_SUS_LDS_SCAN : K2_SUS_LDS_PLAIN;
  __shared__ u64 s_sus[SUSN * K2_SUS_WORDS];  // the block's list of hypotheses left to the strict arithmetic
  __shared__ unsigned s_sus_n, s_sus_base;
  __shared__ K2SusDesc s_susd;

  const int f = blockIdx.x / splits, part = blockIdx.x - f * splits;
  const int tid = threadIdx.x;
  const int nthr = blockDim.x;
  mpe_detections* d = dets + f;
  const int n_d = d->n, n_m = sp.n_markers;
  typename std::conditional<SCAN, ScanRider, NoRider>::type rider;
  if constexpr (SCAN) rider.init(scan, smem);  // (this variant keeps the back-projections in registers)
  // min_num_leds_detected_ (pose_estimator.h:78).  MPE_FRAME_VOTE_LIST_FULL is written by THIS launch (k2_sus_lost,
  // possibly by a sibling block of the same frame while this one starts): it must not decide the branch — threads of
  // one block could read different values around the barriers below — and such a frame is voted again anyway
  // (k2_vote_relost), so both values take the voting path.
  const int st_in = d->status;
  if (n_d < 4 || (st_in != 0 && st_in != MPE_FRAME_VOTE_LIST_FULL) || n_m < 4) {
    rider.drain();
    return;
  }

  for (int i = tid; i < MPE_HIST_STRIDE; i += nthr) s_hist[i] = 0;
  if (tid == 0) {
    s_sus_n = 0;
    s_susd = K2SusDesc{fixup.ctl, reinterpret_cast<u64*>(fixup.list), fixup.cap, &d->status};
  }
  if (tid < n_d) {
    const double u = d->undist_xy[2 * tid], v = d->undist_xy[2 * tid + 1];
    s_px[tid][0] = u;
    s_px[tid][1] = v;
    s_pxf[tid] = f32x2{(float)u, (float)v};
    const V3 b = bearing(u, v, sp.fx, sp.fy, sp.cx, sp.cy);
    s_iv[tid][0] = b.x;
    s_iv[tid][1] = b.y;
    s_iv[tid][2] = b.z;
  }
  __syncthreads();

  double* s_q = reinterpret_cast<double*>(smem);  // back-projections: [2*j + {0,1}][tid]
  const int n_combos = n_d * (n_d - 1) * (n_d - 2) / 6;
  const int n_perms = n_m * (n_m - 1) * (n_m - 2);
  const int nuo = n_m - 3;
  // single-precision copies of the back-projections behind the double ones: [j][tid] (plain variant).  Deferred plain
  // variant: no columns at all — the waves' queues take their place, the table slice follows them
  constexpr bool DEFER = k2_defers(SCAN, NP);
  // (deferred variant: [waves' queues][occupancy grid][table slice])
  u64* s_grid = reinterpret_cast<u64*>(smem + (size_t)(nthr >> 6) * K2_DQ_CAP * K2_DQ_WORDS * sizeof(u64));
  f32x2* s_qf = DEFER ? reinterpret_cast<f32x2*>(s_grid + K2_GRID * K2_GRID_WORDS)
                      : reinterpret_cast<f32x2*>(s_q + (size_t)2 * nuo * nthr);
  __shared__ float s_gp[4];
  constexpr bool F32 = DEFER || SCAN;             // (see k2_vote_item)
  __shared__ float s_trif[F32 ? TRI : 1][12];     // k2_triple_f32 of the staged triples
  if constexpr (SCAN) {  // pixels, not grid cells: G = 1
    if (tid == 0) {
      s_gp[0] = 1.f;
      s_gp[1] = 0.f;
      s_gp[2] = 0.f;
    }
  }
  if constexpr (DEFER) {
    for (int i = tid; i < K2_GRID * K2_GRID_WORDS; i += nthr) s_grid[i] = 0;
    __syncthreads();
    k2_grid_build(s_px, n_d, sp.back_tol, s_grid, s_gp, tid, nthr);
    __syncthreads();
  }
  const float thr_pre = k2_prefilter_threshold(sp.back_tol, F32 ? 0.25 : 0.05);
  const int esz = k2_entry_doubles(n_m);
  const double fx = sp.fx, fy = sp.fy, cx = sp.cx, cy = sp.cy;
  // block row with P1 <-> P2 exchanged: {2, 5, 0, 4, 3, 1}, packed 4 bits per row as 0x134052

  // Scan-carrying variant: the voting loop must not touch global memory (vmcnt counts in order, so any
  // ordinary load issued behind a scan round would wait for that round's HBM latency) -> the per-permutation
  // table values the loop needs are copied to LDS once per block:
  //   [0] p_1 [1] p_2 [2] d_12 [3] valid [4] packed marker indices [5..10] eta-frame unused markers (<= 2)
  double* s_tab = nullptr;
  u64* s_vq = nullptr;  // per-wave queue of deferred exact votes, behind the table copy
  if constexpr (DEFER) s_vq = reinterpret_cast<u64*>(smem) + (size_t)(tid >> 6) * (K2_DQ_CAP * K2_DQ_WORDS);
  if constexpr (SCAN) {
    s_tab = reinterpret_cast<double*>(smem + (size_t)(blockDim.x >> 6) * (K2_SCAN_R * 1024));
    s_vq = reinterpret_cast<u64*>(s_tab + (size_t)n_perms * K2_LTAB) + (size_t)(tid >> 6) * (K2_VQ_CAP * K2_VQ_WORDS);
    for (int i = tid; i < n_perms * K2_LTAB; i += nthr) s_tab[i] = k2_ltab_value(tab, esz, nuo, i);
    __syncthreads();
  }
  // Plain variant, 6 .. 10 markers: the per-permutation table (88 KB at 8 markers) does not fit the LDS of a block, and
  // reading it from global memory left the voting waves waiting (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES 0.21 at C3).  With
  // `slice_tab` the `splits` blocks of a frame divide the PERMUTATIONS among themselves (whole blocks of six rows, so
  // the swapped row of p3p.cpp:100-121 stays inside the slice) and each copies the fields the loop reads of its
  // slice — [12 .. esz) of every entry: p_1 p_2 d_12 valid indices pad, eta-frame markers — to LDS once.
  int p_lo = 0, n_perms_loc = n_perms, part_loc = part, splits_loc = splits;
  const double* tab_eff = tab;
  int esz_eff = esz;
  if constexpr (!SCAN) {
    if (slice_tab) {
      const int combos6 = n_perms / 6;
      p_lo = 6 * (int)(((long long)part * combos6) / splits);
      const int p_hi = 6 * (int)(((long long)(part + 1) * combos6) / splits);
      n_perms_loc = p_hi - p_lo;
      part_loc = 0;
      splits_loc = 1;
      const int LT = esz - 12;
      double* s_slice = reinterpret_cast<double*>(s_qf);  // (NP > 0: the single-precision columns are not used)
      for (int i = tid; i < n_perms_loc * LT; i += nthr) {
        const int pe = i / LT, fld = i - pe * LT;
        s_slice[i] = tab[(size_t)(p_lo + pe) * esz + 12 + fld];
      }
      __syncthreads();
      // (the item indexes the table with pj - pj_base: no pointer ever leaves the LDS allocation — an LDS pointer is a
      //  32-bit offset, and one that wraps below zero is not an address)
      tab_eff = s_slice - 12;  // so that tab_eff + (pj - p_lo) * LT + 12 is field 12 of permutation pj
      esz_eff = LT;
    }
  }
  const K2Frame F = {s_trii, s_tri, s_px, s_pxf, s_q,  s_qf, s_hist, tab_eff,     s_tab,   n_d,  nuo,
                     nthr,   tid,   esz_eff, fx, fy,   cx,   cy,     sp.back_tol, thr_pre, s_vq, 64, p_lo,
                     &s_susd, fixup.screen != 0u, f, s_sus, &s_sus_n, (unsigned)SUSN,
                     s_grid, s_gp[0], s_gp[1], s_gp[2], s_trif};
  int vq_count = 0;  // entries in this wave's queue (wave-uniform)
  for (int tc0 = 0; tc0 < n_combos; tc0 += TRI) {
    const int ntri = min(TRI, n_combos - tc0);
    if (tc0) __syncthreads();
    // ---- per-triple part of computePoses (p3p.cpp:82-121, 143-154)
    if (tid < ntri) {
      k2_triple_entry(s_iv, n_d, tc0 + tid, fx, fy, cx, cy, s_tri[tid], s_trii[tid]);
      if constexpr (F32) k2_triple_f32(s_tri[tid], s_gp, s_trif[tid]);
    }
    __syncthreads();

    // flattened (triple, permutation) index, advanced without divisions
    const int total = ntri * n_perms_loc;  // <= 64 * 3360
    const int stride = splits_loc * nthr;
    int t = part_loc * nthr + tid;
    int ti = t / n_perms_loc, pj = t - ti * n_perms_loc;  // (pj: within this block's permutation range)
    const int dti = stride / n_perms_loc, dpj = stride - dti * n_perms_loc;
    // With the scan rider on board the loop nest must stay wave-uniform (the rider's LDS-DMA rounds need all
    // 64 lanes at every call): every lane then runs the iteration count of the slowest one and lanes without
    // a valid item — past the end, collinear marker triple, non-finite root — compute on harmlessly and are
    // merely barred from voting (`live`).  Without a rider those lanes skip ahead as before.
    const int n_iter = (total - part_loc * nthr + stride - 1) / stride;
    constexpr bool UNI = SCAN || k2_defers(SCAN, NP);  // (see k2_vote_item)
    for (int it = 0; UNI ? (it < n_iter) : (t < total); ++it, t += stride, ti += dti, pj += dpj) {
      if (pj >= n_perms_loc) {
        pj -= n_perms_loc;
        ++ti;
      }
      bool live = true;
      const int ti_keep = ti, pj_keep = pj;
      if constexpr (UNI) {
        if (t >= total) {
          live = false;
          ti = 0;
          pj = 0;
        }
      }
      if constexpr (RANGE) {
        const int g = (tc0 + ti) * n_perms + p_lo + pj;
        if (g < item_range[2 * f] || g >= item_range[2 * f + 1]) {
          if constexpr (UNI) {
            live = false;
          } else {
            continue;
          }
        }
      }
      k2_vote_item<SCAN, NP>(F, ti, p_lo + pj, live, rider, vq_count);
      ti = ti_keep;
      pj = pj_keep;
    }
    if constexpr (k2_defers(SCAN, NP)) {  // (the staged triples the queue's entries refer to are about to be replaced)
      k2_defer_flush<NP>(F, vq_count, true);
    }
    if constexpr (SCAN) {
      if (tc0 + TRI < n_combos) {  // (more than 16 triples, i.e. more than 5 detections: the same)
        k2_vote_flush(F, vq_count);
        vq_count = 0;
      }
    }
    if constexpr (!SCAN) k2_sus_flush(F, &s_sus_base);  // (the scan-carrying variant: once, behind the rider's last round)
  }
  if constexpr (SCAN) k2_vote_flush(F, vq_count);  // what is left in this wave's queue of deferred votes
  rider.drain();
  if constexpr (SCAN) k2_sus_flush(F, &s_sus_base);
  __syncthreads();
  uint32_t* gh = hist + (size_t)f * MPE_HIST_STRIDE;
  if (splits == 1) {
    // this block owns the frame's histogram: plain stores of the rows the tail kernel reads (detections < n_d) —
    // the caller then needs no memset of the histogram buffer
    for (int i = tid; i < n_d * MPE_MAX_MARKERS; i += nthr) gh[i] = s_hist[i];
  } else {
    for (int i = tid; i < MPE_HIST_STRIDE; i += nthr) {
      const unsigned v = s_hist[i];
      if (v) atomicAdd(&gh[i], v);
    }
  }
}

// One hypothesis in the STRICT arithmetic: initialise()'s loop body (pose_estimator.cpp:596-702) for detection triple
// (c0, c1, c2) against marker permutation (p0, p1, p2) with the SAME device functions the validation kernel uses —
// p3p_prepare / solve_quartic / p3p_solution / make_projection / project, IEEE division and square root, libm cube
// root, the reference's statement order, [R|C] formed for every solution.  `kmask` selects the roots, `detmask` the
// detections to decide (k2_vote_strict: all four, every detection outside the triple; k2_vote_fixup: what the fast
// kernel left undecided), `triple_voted`: the triple's own three votes of that root have been cast already.
// q: 2 * (n_m - 3) doubles of scratch per lane, element i at q[i * qs].  vote(detection, marker) casts one vote.
template <class Vote>
__device__ __forceinline__ void k2_strict_item(const V3& fa, const V3& fb, const V3& fc, const double (*px)[2],
                                               const SolveParams& sp, int c0, int c1, int c2, int p0, int p1, int p2,
                                               unsigned kmask, unsigned detmask, bool triple_voted, double* q, int qs,
                                               Vote vote) {
  const int n_m = sp.n_markers, nuo = n_m - 3;
  const V3 wa = {sp.markers[3 * p0], sp.markers[3 * p0 + 1], sp.markers[3 * p0 + 2]},
           wb = {sp.markers[3 * p1], sp.markers[3 * p1 + 1], sp.markers[3 * p1 + 2]},
           wc = {sp.markers[3 * p2], sp.markers[3 * p2 + 1], sp.markers[3 * p2 + 2]};
  P3PCtx ctx;
  if (!p3p_prepare(fa, fb, fc, wa, wb, wc, ctx)) return;  // computePoses returned -1
#pragma unroll 1
  for (int k = 0; k < 4; ++k) {
    if (!((kmask >> k) & 1u)) continue;
    M3 R;
    V3 C;
    p3p_solution(ctx, pick_root(ctx, k), R, C);
    if (!rc_finite(R, C)) continue;  // pose_estimator.cpp:653
    const Proj P = make_projection(R, C, sp.fx, sp.fy, sp.cx, sp.cy);
    int j = 0;
    for (int m = 0; m < n_m; ++m) {  // unused markers, ascending (pose_estimator.cpp:621-661)
      if (m == p0 || m == p1 || m == p2) continue;
      double u, v;
      project(P, V3{sp.markers[3 * m], sp.markers[3 * m + 1], sp.markers[3 * m + 2]}, u, v);
      q[(2 * j) * qs] = u;
      q[(2 * j + 1) * qs] = v;
      ++j;
    }
    bool any = false;
    for (unsigned dm = detmask; dm; dm &= dm - 1) {  // unused detections, ascending (pose_estimator.cpp:576-597, 862-906)
      const int a = __builtin_ctz(dm);
      double best = INFINITY;
      int bj = 0;
      for (int jj = 0; jj < nuo; ++jj) {
        const double du = px[a][0] - q[(2 * jj) * qs], dv = px[a][1] - q[(2 * jj + 1) * qs];
        const double d2 = du * du + dv * dv;
        if (d2 < best) {
          best = d2;
          bj = jj;
        }
      }
      if (sqrt(best) < sp.back_tol) {  // strict <, pose_estimator.cpp:671,689
        int mi = -1, cnt = 0;
        for (int m = 0; m < n_m; ++m) {
          if (m == p0 || m == p1 || m == p2) continue;
          if (cnt == bj) mi = m;
          ++cnt;
        }
        vote(a, mi);
        any = true;
      }
    }
    if (any && !triple_voted) {  // pose_estimator.cpp:676-685
      vote(c0, p0);
      vote(c1, p1);
      vote(c2, p2);
    }
  }
}

// Strict voting kernel (option "vote_arith" = 0): initialise()'s loop nest (pose_estimator.cpp:565-702), every
// hypothesis through k2_strict_item.  No tables, no scan rider, about 2.5x the instructions of k2_vote: the reference
// point the default arithmetic is held against, selectable at run time.
// One frame's share `part` of `splits` of initialise()'s loop nest with the strict item, votes collected in LDS and
// then ADDED to (STORE = false) or STORED over (true: a whole frame by one block) the frame's histogram.
template <bool STORE>
__device__ __forceinline__ void k2_strict_frame(const mpe_detections* __restrict__ d, const SolveParams& sp,
                                                uint32_t* __restrict__ gh, int f, int part, int splits,
                                                const int* __restrict__ item_range, unsigned char* smem,
                                                double (*s_px)[2], double (*s_iv)[3], unsigned* s_hist) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int n_d = d->n, n_m = sp.n_markers;
  for (int i = tid; i < MPE_HIST_STRIDE; i += nthr) s_hist[i] = 0;
  if (tid < n_d) {
    const double u = d->undist_xy[2 * tid], v = d->undist_xy[2 * tid + 1];
    s_px[tid][0] = u;
    s_px[tid][1] = v;
    const V3 b = bearing(u, v, sp.fx, sp.fy, sp.cx, sp.cy);
    s_iv[tid][0] = b.x;
    s_iv[tid][1] = b.y;
    s_iv[tid][2] = b.z;
  }
  __syncthreads();
  double* s_q = reinterpret_cast<double*>(smem);  // back-projections: [2*j + {0,1}][tid]
  const int n_combos = n_d * (n_d - 1) * (n_d - 2) / 6;
  const int n_perms = n_m * (n_m - 1) * (n_m - 2);
  const long long total = (long long)n_combos * n_perms;
  for (long long t = (long long)part * nthr + tid; t < total; t += (long long)splits * nthr) {
    if (item_range && (t < item_range[2 * f] || t >= item_range[2 * f + 1])) continue;  // (forensics only)
    const int ti = (int)(t / n_perms), pj = (int)(t - (long long)ti * n_perms);
    int c0, c1, c2, p0, p1, p2;
    unrank_combo3(ti, n_d, c0, c1, c2);
    perm_from_index(pj, n_m, p0, p1, p2);
    const V3 fa = {s_iv[c0][0], s_iv[c0][1], s_iv[c0][2]}, fb = {s_iv[c1][0], s_iv[c1][1], s_iv[c1][2]},
             fc = {s_iv[c2][0], s_iv[c2][1], s_iv[c2][2]};
    const unsigned unused = (0xFFFFFFFFu >> (32 - n_d)) & ~((1u << c0) | (1u << c1) | (1u << c2));
    k2_strict_item(fa, fb, fc, s_px, sp, c0, c1, c2, p0, p1, p2, 0xFu, unused, false, s_q + tid, nthr,
                   [&](const int a, const int m) { atomicAdd(&s_hist[a * MPE_MAX_MARKERS + m], 1u); });
  }
  __syncthreads();
  for (int i = tid; i < MPE_HIST_STRIDE; i += nthr) {
    const unsigned v = s_hist[i];
    if constexpr (STORE) gh[i] = v;
    else if (v) atomicAdd(&gh[i], v);
  }
}
__global__ __launch_bounds__(K2_THREADS) void k2_vote_strict(const mpe_detections* __restrict__ dets, SolveParams sp,
                                                             uint32_t* __restrict__ hist, int splits,
                                                             const int* __restrict__ item_range) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ double s_px[MPE_MAX_DETECTIONS][2];
  __shared__ double s_iv[MPE_MAX_DETECTIONS][3];
  __shared__ unsigned s_hist[MPE_HIST_STRIDE];
  const int f = blockIdx.x / splits, part = blockIdx.x - f * splits;
  const mpe_detections* d = dets + f;
  if (d->n < 4 || d->status != 0 || sp.n_markers < 4) return;
  k2_strict_frame<false>(d, sp, hist + (size_t)f * MPE_HIST_STRIDE, f, part, splits, item_range, smem, s_px, s_iv, s_hist);
}

// Frames that lost a suspect entry to a full list (k2_sus_lost) are voted again, whole, with the strict loop nest: the
// histogram is STORED over whatever the fast launch and the fix-up kernel left, the mark is cleared, the tail then
// sees an ordinary frame.  A fixed grid: nothing lost since the last launch on this slot (ctl[1] == ctl[4], the rule)
// -> every block leaves after one load; otherwise the blocks stride over the launch's frames looking for the mark.
// The last block to finish records what has been handled (ctl[4]; ctl[5] counts the blocks).
__global__ __launch_bounds__(K2_THREADS) void k2_vote_relost(mpe_detections* __restrict__ dets, int n_frames, SolveParams sp,
                                                             uint32_t* __restrict__ hist, VoteFixup fx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ double s_px[MPE_MAX_DETECTIONS][2];
  __shared__ double s_iv[MPE_MAX_DETECTIONS][3];
  __shared__ unsigned s_hist[MPE_HIST_STRIDE];
  const unsigned lost = fx.ctl[1];
  if (lost == fx.ctl[4]) return;  // (only the last block of a launch writes ctl[4], after every block has read it)
  for (int f = blockIdx.x; f < n_frames; f += gridDim.x) {
    mpe_detections* d = dets + f;
    if (d->status != MPE_FRAME_VOTE_LIST_FULL) continue;  // (written by an earlier launch: uniform over the block)
    if (d->n >= 4 && sp.n_markers >= 4)
      k2_strict_frame<true>(d, sp, hist + (size_t)f * MPE_HIST_STRIDE, f, 0, 1, nullptr, smem, s_px, s_iv, s_hist);
    __syncthreads();
    if (threadIdx.x == 0) {
      d->status = 0;
      atomicAdd(&fx.ctl[6], 1u);  // frames voted again (cumulative)
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(&fx.ctl[5], 1u) == gridDim.x - 1) {
      fx.ctl[4] = lost;
      fx.ctl[5] = 0;
      __threadfence();
    }
  }
}

// The hypotheses a fast voting launch left undecided (k2_sus_push), one per lane, through k2_strict_item; their votes
// go straight into the frames' histograms in global memory (the voting launch has stored or added its own by then:
// same stream, or an event in between).  The last block to finish resets the list's fill count for the next launch
// and adds the number of entries to the cumulative counter (ctl[3]).
#define K2_FIX_THREADS 64
__global__ __launch_bounds__(K2_FIX_THREADS) void k2_vote_fixup(const mpe_detections* __restrict__ dets, SolveParams sp,
                                                                uint32_t* __restrict__ hist, VoteFixup fx) {
  __shared__ double s_q[2 * (MPE_MAX_MARKERS - 3) * K2_FIX_THREADS];
  const unsigned n = min(fx.ctl[0], fx.cap);
  if (n == 0) return;  // (nothing appended — every block sees the same count — and nothing to reset)
  const u64* list = reinterpret_cast<const u64*>(fx.list);
  const int tid = threadIdx.x;
  for (unsigned i = blockIdx.x * K2_FIX_THREADS + tid; i < n; i += gridDim.x * K2_FIX_THREADS) {
    const u64 w0 = list[(size_t)K2_SUS_WORDS * i];
    const unsigned detmask = (unsigned)list[(size_t)K2_SUS_WORDS * i + 1];
    const int f = (int)(unsigned)w0;
    const unsigned code = (unsigned)(w0 >> 32);
    const int c0 = code & 31, c1 = (code >> 5) & 31, c2 = (code >> 10) & 31;
    const int p0 = (code >> 15) & 15, p1 = (code >> 19) & 15, p2 = (code >> 23) & 15;
    const unsigned kmask = (code >> 27) & 15u;
    const bool triple_voted = (code >> 31) & 1u;
    const mpe_detections* d = dets + f;
    // the frame's detections, read straight from the record: k2_strict_item indexes px[a][0 / 1]
    const double (*px)[2] = reinterpret_cast<const double (*)[2]>(d->undist_xy);
    const V3 fa = bearing(px[c0][0], px[c0][1], sp.fx, sp.fy, sp.cx, sp.cy),
             fb = bearing(px[c1][0], px[c1][1], sp.fx, sp.fy, sp.cx, sp.cy),
             fc = bearing(px[c2][0], px[c2][1], sp.fx, sp.fy, sp.cx, sp.cy);
    uint32_t* gh = hist + (size_t)f * MPE_HIST_STRIDE;
    k2_strict_item(fa, fb, fc, px, sp, c0, c1, c2, p0, p1, p2, kmask, detmask, triple_voted, s_q + tid, K2_FIX_THREADS,
                   [&](const int a, const int m) { atomicAdd(&gh[a * MPE_MAX_MARKERS + m], 1u); });
  }
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    if (atomicAdd(&fx.ctl[2], 1u) == gridDim.x - 1) {  // every other block has read ctl[0] and finished
      fx.ctl[3] += n;
      fx.ctl[0] = 0;
      fx.ctl[2] = 0;
      __threadfence();
    }
  }
}

hipError_t launch_k2_fixup(mpe_detections* dets, int n_frames, const SolveParams& sp, uint32_t* hist, const VoteFixup& fx,
                           hipStream_t s) {
  if (!fx.ctl || fx.cap == 0 || sp.n_markers < 4) return hipSuccess;
  // (the entry count lives on the device: a fixed grid strides over it — wide, every entry is a single-wave chain of
  //  dependent FP64 operations (~30 us), and blocks beyond the count leave at once; ~0.15 % of the hypotheses)
  hipLaunchKernelGGL(k2_vote_fixup, dim3(2048), dim3(K2_FIX_THREADS), 0, s, dets, sp, hist, fx);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  // frames that lost an entry to a full list: voted again with the strict loop nest (nothing lost: 256 blocks, one load each)
  const size_t lds_strict = (size_t)(sp.n_markers - 3) * 2 * K2_THREADS * sizeof(double);
  hipLaunchKernelGGL(k2_vote_relost, dim3((unsigned)std::min(n_frames, 256)), dim3(K2_THREADS), lds_strict, s, dets,
                     n_frames, sp, hist, fx);
  return hipGetLastError();
}

hipError_t launch_k2_vote(mpe_detections* dets, int n_frames, const SolveParams& sp, const double* tab,
                          uint32_t* hist, int splits, int n_det_hint, hipStream_t s, const uint8_t* scan_px,
                          size_t scan_bytes, unsigned long long* scan_flags, int scan_thr, size_t* scanned_bytes,
                          const int* item_range, const VoteFixup* fixup) {
  if (scanned_bytes) *scanned_bytes = 0;
  // vote_arith 1: suspect hypotheses go to `fixup` (the caller launches launch_k2_fixup behind this kernel);
  // 2: the fast arithmetic decides everything itself (round-3 behaviour, for A/B measurements); 0: strict kernel
  VoteFixup fx = {nullptr, nullptr, 0u, 0u};
  if (sp.vote_arith != 0 && fixup && fixup->ctl && fixup->list) {
    fx = *fixup;
    fx.screen = sp.vote_arith == 1 ? 1u : 0u;
  }
  if (n_frames <= 0 || sp.n_markers < 4) return hipSuccess;
  // the fast kernels append to the list (suspects, or what their per-wave queues cannot hold): they need one
  if (sp.vote_arith != 0 && !fx.ctl) return hipErrorInvalidValue;
  int slice_tab = 0;
  if (splits < 0) {  // -(blocks per frame): the blocks share the marker permutations, table slices in LDS
    splits = -splits;
    slice_tab = 1;
  }
  if (splits < 1) splits = 1;
  const int nuo = sp.n_markers - 3;
  if (sp.vote_arith == 0) {  // strict arithmetic: the validation kernel's P3P, no tables, no scan rider
    const size_t lds_strict = (size_t)nuo * 2 * K2_THREADS * sizeof(double);
    hipLaunchKernelGGL(k2_vote_strict, dim3((unsigned)(n_frames * splits)), dim3(K2_THREADS), lds_strict, s, dets, sp,
                       hist, splits, item_range);
    return hipGetLastError();
  }
  // block size: the multiple of 64 (<= 256) that wastes the fewest lanes on the expected item count
  int threads = K2_THREADS;
  if (n_det_hint >= 4) {
    const long long nm = sp.n_markers;
    long long ntri = (long long)n_det_hint * (n_det_hint - 1) * (n_det_hint - 2) / 6;
    const long long chunk = (scan_px && nuo <= 2) ? K2_TRI_CHUNK_SCAN : K2_TRI_CHUNK;
    if (ntri > chunk) ntri = chunk;
    const long long items = ntri * nm * (nm - 1) * (nm - 2);
    double best = 1e30;
    for (int t = 64; t <= K2_THREADS; t += 64) {
      const long long per = (items + (long long)splits * t - 1) / ((long long)splits * t);
      const double waste = (double)(per * splits * t) / (double)items + 0.002 * (K2_THREADS / t);
      if (waste < best - 1e-9) {
        best = waste;
        threads = t;
      }
    }
  }
  // table slices: a block's items are 64 staged triples x its share of the permutations — full-size blocks (a 64-thread
  // block per slice measured 6 waves per CU and 170 instead of 118 ms per 16 384 C3 frames)
  if (slice_tab) threads = K2_THREADS;
  // plain kernel: 24 bytes of dynamic LDS per thread and unused marker (double + single precision back-projections) on
  // top of ~10.5 KB static; the block shrinks until both fit the 64 KB a block may have without an opt-in (14 - 16
  // markers: 192 / 128 threads)
  while (threads > 64 && (size_t)nuo * 24 * threads + 11 * 1024 > 64 * 1024) threads -= 64;
  size_t lds = (size_t)nuo * 2 * threads * sizeof(double) + (size_t)nuo * threads * sizeof(f32x2);
  ScanArgs sa = {nullptr, nullptr, 0, {0u, 0u}};
  const size_t chunk_bytes = (size_t)K2_SCAN_R * 1024;
  if (scan_px && nuo <= 2 && scan_bytes >= chunk_bytes && scan_bytes / chunk_bytes < 0x7fffffffull) {
    sa.px = reinterpret_cast<const uint4*>(scan_px);
    sa.flags = (u64*)scan_flags;
    sa.n_chunks = (int)(scan_bytes / chunk_bytes);
    sa.thr = make_thr_test(scan_thr);
    lds = (size_t)(threads / 64) * chunk_bytes +
          (size_t)sp.n_markers * (sp.n_markers - 1) * (sp.n_markers - 2) * K2_LTAB * sizeof(double) +
          (size_t)(threads / 64) * K2_VQ_CAP * K2_VQ_WORDS * sizeof(u64);
    if (scanned_bytes) *scanned_bytes = (size_t)sa.n_chunks * chunk_bytes;
    hipLaunchKernelGGL((k2_vote<true, false>), dim3((unsigned)(n_frames * splits)), dim3(threads), lds, s, dets, sp, tab, hist,
                       splits, sa, (const int*)nullptr, 0, fx);
  } else {
    // plain kernel: the prefilter's single-precision back-projections in registers as (nuo + 1) / 2 packed marker pairs
    // (up to 8 unused markers), else in LDS columns; the forensics instantiation (item_range) the same way
    const int np = nuo <= 8 ? (nuo + 1) / 2 : 0;
    const dim3 grid((unsigned)(n_frames * splits)), block(threads);
    const bool defer = k2_defers(false, np);
    if (defer)  // no back-projection columns: the waves' queues of deferred (hypothesis, root) entries + the grid
      lds = (size_t)(threads / 64) * K2_DQ_CAP * K2_DQ_WORDS * sizeof(u64) + (size_t)K2_GRID * K2_GRID_WORDS * sizeof(u64);
    // the table slice of a block lives where the (unused, NP > 0) single-precision columns would: make it fit
    if (slice_tab && np > 0) {
      const size_t slice = (size_t)6 * ((size_t)sp.n_markers * (sp.n_markers - 1) * (sp.n_markers - 2) / 6 / splits + 1) *
                           (k2_entry_doubles(sp.n_markers) - 12) * sizeof(double);
      const size_t have = defer ? 0 : (size_t)nuo * threads * sizeof(f32x2);
      if (slice > have) lds += slice - have;
    } else {
      slice_tab = 0;
    }
#define MPE_K2_PLAIN(NPV)                                                                                            \
  do {                                                                                                               \
    if (item_range)                                                                                                  \
      hipLaunchKernelGGL((k2_vote<false, true, NPV>), grid, block, lds, s, dets, sp, tab, hist, splits, sa, item_range, \
                         slice_tab, fx);                                                                             \
    else                                                                                                             \
      hipLaunchKernelGGL((k2_vote<false, false, NPV>), grid, block, lds, s, dets, sp, tab, hist, splits, sa,         \
                         (const int*)nullptr, slice_tab, fx);                                                        \
  } while (0)
    switch (np) {
      case 1: MPE_K2_PLAIN(1); break;
      case 2: MPE_K2_PLAIN(2); break;
      case 3: MPE_K2_PLAIN(3); break;
      case 4: MPE_K2_PLAIN(4); break;
      default: MPE_K2_PLAIN(0); break;
    }
#undef MPE_K2_PLAIN
  }
  return hipGetLastError();
}

// =============================================================================================
// K3 — per-frame tail: correspondences from the histogram, validation, Kabsch, Gauss-Newton
// =============================================================================================
struct T34 {  // rigid transform rows [R | t]
  double m[3][4];
};

__device__ __forceinline__ void project_T(const T34& T, const double* mk, double fx, double fy, double cx, double cy,
                                          double& u, double& v, double& X, double& Y, double& Z) {
  X = T.m[0][0] * mk[0] + T.m[0][1] * mk[1] + T.m[0][2] * mk[2] + T.m[0][3];
  Y = T.m[1][0] * mk[0] + T.m[1][1] * mk[1] + T.m[1][2] * mk[2] + T.m[1][3];
  Z = T.m[2][0] * mk[0] + T.m[2][1] * mk[1] + T.m[2][2] * mk[2] + T.m[2][3];
  u = (fx * X + cx * Z) / Z;
  v = (fy * Y + cy * Z) / Z;
}

// R = V U^T of H = U S V^T (computeTransformation, pose_estimator.cpp:916-922: JacobiSVD, no reflection
// guard) by a one-sided (Hestenes) Jacobi SVD: the columns of G = H V are rotated pairwise until they are
// orthogonal, U = G with normalised columns.  Same sweeps, thresholds and operation order as the CPU
// oracle's svd3.  A rank-deficient H (coplanar markers) leaves one column of G at rounding level; the
// reference's JacobiSVD gives that column of U the sign of its rounding residue (a coin flip between the
// rotation and its mirror image through the marker plane) — here, as in the oracle, every sigma <= 1e-12
// sigma_max is completed with the cross product of the other two columns (det U = +1).
__device__ void kabsch_rotation(const double Hm[3][3], double R[3][3]) {
  double G[3][3], V[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      G[i][j] = Hm[i][j];
      V[i][j] = (i == j) ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 60; ++sweep) {
    bool rotated = false;
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;  // (0,1) (0,2) (1,2)
      double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        alpha += G[i][p] * G[i][p];
        beta += G[i][q] * G[i][q];
        gamma += G[i][p] * G[i][q];
      }
      if (gamma == 0.0 || fabs(gamma) <= 1e-300 + 2.3e-16 * sqrt(alpha * beta)) continue;
      rotated = true;
      const double zeta = (beta - alpha) / (2.0 * gamma);
      const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
      const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double gp = G[i][p], gq = G[i][q];
        G[i][p] = c * gp - sn * gq;
        G[i][q] = sn * gp + c * gq;
        const double vp = V[i][p], vq = V[i][q];
        V[i][p] = c * vp - sn * vq;
        V[i][q] = sn * vp + c * vq;
      }
    }
    if (!rotated) break;
  }
  double sv[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) sv[j] = sqrt(G[0][j] * G[0][j] + G[1][j] * G[1][j] + G[2][j] * G[2][j]);
  const double smax = fmax(sv[0], fmax(sv[1], sv[2]));
  double U[3][3];
  int zero_col = -1, nzero = 0;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (sv[j] > smax * 1e-12 && sv[j] > 0) {
#pragma unroll
      for (int i = 0; i < 3; ++i) U[i][j] = G[i][j] / sv[j];
    } else {
#pragma unroll
      for (int i = 0; i < 3; ++i) U[i][j] = 0.0;
      zero_col = j;
      ++nzero;
    }
  }
  if (nzero == 1) {  // rank 2: u_zero = u_a x u_b, (zero, a, b) cyclic
#pragma unroll
    for (int z = 0; z < 3; ++z)
      if (z == zero_col) {
        const int a = (z + 1) % 3, b = (z + 2) % 3;
        U[0][z] = U[1][a] * U[2][b] - U[2][a] * U[1][b];
        U[1][z] = U[2][a] * U[0][b] - U[0][a] * U[2][b];
        U[2][z] = U[0][a] * U[1][b] - U[1][a] * U[0][b];
      }
  } else if (nzero > 1) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) U[i][j] = (i == j) ? 1.0 : 0.0;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double acc = V[i][0] * U[j][0];  // (V U^T)(i,j) = sum_k V(i,k) U(j,k), k ascending like the oracle's mul()
      acc += V[i][1] * U[j][1];
      acc += V[i][2] * U[j][2];
      R[i][j] = acc;
    }
}

// unpivoted LDL^T of a symmetric positive definite 6x6 (normal equations of GN)
struct LDL6 {
  double L[6][6];
  double D[6];
};
__device__ __forceinline__ void ldl6_factor(const double A[6][6], LDL6& F) {
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double d = A[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= F.L[j][k] * F.L[j][k] * F.D[k];
    F.D[j] = d;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double s = A[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= F.L[i][k] * F.L[j][k] * F.D[k];
      F.L[i][j] = s / d;
    }
  }
}
__device__ __forceinline__ void ldl6_solve(const LDL6& F, const double b[6], double x[6]) {
  double y[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= F.L[i][k] * y[k];
    y[i] = s;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) y[i] /= F.D[i];
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double s = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) s -= F.L[k][i] * x[k];
    x[i] = s;
  }
}

// exponentialMap(dT) * T   (pose_estimator.cpp:781, 962-994)
__device__ __forceinline__ void apply_exp(const double tw[6], T34& T) {
  const double ux = tw[0], uy = tw[1], uz = tw[2], wx = tw[3], wy = tw[4], wz = tw[5];
  const double theta = sqrt(wx * wx + wy * wy + wz * wz);
  const double th2 = theta * theta;
  double Rm[3][3], Vm[3][3];
  const double O[3][3] = {{0, -wz, wy}, {wz, 0, -wx}, {-wy, wx, 0}};
  double O2[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) O2[i][j] = O[i][0] * O[0][j] + O[i][1] * O[1][j] + O[i][2] * O[2][j];
  if (theta == 0) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Rm[i][j] = Vm[i][j] = (i == j) ? 1.0 : 0.0;
  } else {
    double st, ct;
    sincos(theta, &st, &ct);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        const double I = (i == j) ? 1.0 : 0.0;
        Rm[i][j] = I + O[i][j] / theta * st + O2[i][j] / th2 * (1 - ct);
        Vm[i][j] = I + (1 - ct) / th2 * O[i][j] + (theta - st) / (th2 * theta) * O2[i][j];
      }
  }
  const double t0 = Vm[0][0] * ux + Vm[0][1] * uy + Vm[0][2] * uz;
  const double t1 = Vm[1][0] * ux + Vm[1][1] * uy + Vm[1][2] * uz;
  const double t2 = Vm[2][0] * ux + Vm[2][1] * uy + Vm[2][2] * uz;
  const double tv[3] = {t0, t1, t2};
  T34 N;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 4; ++j) {
      double s = Rm[i][0] * T.m[0][j] + Rm[i][1] * T.m[1][j] + Rm[i][2] * T.m[2][j];
      if (j == 3) s += tv[i];
      N.m[i][j] = s;
    }
  }
  T = N;
}

// optimisePose (pose_estimator.cpp:733-792): Gauss-Newton on SE(3) over the n_c correspondence rows
// row(j, 0..2) = marker xyz, row(j, 3..4) = detection uv; T is updated in place, cov = A^-1 of the last iteration
// (row-major 6x6), returns the number of iterations.
template <class Row>
__device__ __forceinline__ int k3_gauss_newton(int n_c, Row row, double fx, double fy, double cx, double cy, T34& T,
                                               double* __restrict__ cov) {
  double A[6][6];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) A[r][c] = 0;
  int iters = 0;
  for (int it = 0; it < 500; ++it) {
    double b[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c) A[r][c] = 0;
    for (int j = 0; j < n_c; ++j) {
      const double mk[3] = {row(j, 0), row(j, 1), row(j, 2)};
      double u, v, x, y, z;
      project_T(T, mk, fx, fy, cx, cy, u, v, x, y, z);
      const double e0 = row(j, 3) - u, e1 = row(j, 4) - v;
      const double z_2 = z * z;
      // computeJacobian, pose_estimator.cpp:945-957
      double J0[6] = {0, 0, 0, 0, 0, 0}, J1[6] = {0, 0, 0, 0, 0, 0};
      J0[0] = 1 / z * fx;
      J0[2] = -x / z_2 * fx;
      J0[3] = -x * y / z_2 * fx;
      J0[4] = (1 + (x * x / z_2)) * fx;
      J0[5] = -y / z * fx;
      J1[1] = 1 / z * fy;
      J1[2] = -y / z_2 * fy;
      J1[3] = -(1 + y * y / z_2) * fy;
      J1[4] = x * y / z_2 * fy;
      J1[5] = x / z * fy;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
#pragma unroll
        for (int c = r; c < 6; ++c) A[r][c] += J0[r] * J0[c] + J1[r] * J1[c];  // A += J^T J (upper triangle)
        b[r] += J0[r] * e0 + J1[r] * e1;                                       // b += J^T e
      }
    }
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < r; ++c) A[r][c] = A[c][r];
    LDL6 F;
    ldl6_factor(A, F);
    double dT[6];
    ldl6_solve(F, b, dT);
    apply_exp(dT, T);
    iters = it + 1;
    double mx = -1;  // norm_max, pose_estimator.cpp:1073-1085
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double av = fabs(dT[r]);
      if (av > mx) mx = av;
    }
    if (mx <= 1e-13) break;
  }
  // pose_covariance_ = A.inverse() with the A of the last iteration (pose_estimator.cpp:790)
  LDL6 F;
  ldl6_factor(A, F);
#pragma unroll 1
  for (int c = 0; c < 6; ++c) {
    double e[6], x[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) e[r] = (r == c) ? 1.0 : 0.0;
    ldl6_solve(F, e, x);
#pragma unroll
    for (int r = 0; r < 6; ++r) cov[r * 6 + c] = x[r];
  }
  return iters;
}

#define K3_GROUP 16                 // lanes cooperating on one frame in the validation kernel
#define K3_FRAMES_PER_BLOCK 4       // one wave = 4 frames
#define K3B_THREADS 64              // refinement kernel: one lane per frame

// What the validation kernel hands to the refinement kernel, one record per frame (global memory).
struct TailMid {
  int n_c;             // rows of correspondences_
  int active;          // 1: computeTransformation / optimisePose run for this frame
  unsigned num_valid;  // P3P triples with a valid solution (pose_estimator.cpp:506)
  unsigned pad;
  unsigned char cm[MPE_MAX_MARKERS], cd[MPE_MAX_MARKERS];  // rows (marker, detection), 1-based
  double mean[3 * MPE_MAX_MARKERS];  // sum over the valid triples of inverse(H_best) * marker (not yet divided)
};
size_t k3_mid_bytes(int n_frames) { return (size_t)(n_frames > 0 ? n_frames : 1) * sizeof(TailMid); }

// ---------------------------------------------------------------------------------------------
// K3a  k3a_validate: 16 lanes per frame (4 frames per wave).  Lane 0 of a group builds the correspondences
// (histogram peeling, given rows, or nearest neighbour), then the C(n_c,3) P3P validations of
// checkCorrespondences run 16 at a time, one per lane, and are summed in combination order.
// MODE 0 / 1: as described.  MODE 2 (optimisePose alone): only the rows are parsed, no validation.
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(64) void k3a_validate(const mpe_detections* __restrict__ dets,
                                                   const uint32_t* __restrict__ hist, int n_frames, SolveParams sp,
                                                   mpe_result* __restrict__ results, uint32_t* __restrict__ corr_out,
                                                   const uint32_t* __restrict__ corr_in,
                                                   const double* __restrict__ nn_pred, double nn_tol,
                                                   TailMid* __restrict__ mid) {
  // dynamic LDS, sized for the actual marker count: per-lane contributions [4][16][3 n_m] and
  // back-projections [2 (rows - 3)][64]
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  const int nm3 = 3 * sp.n_markers;
  double* s_part_ = reinterpret_cast<double*>(smem3);
  double* s_q_ = s_part_ + K3_FRAMES_PER_BLOCK * K3_GROUP * nm3;
#define s_part(g_, l_, i_) s_part_[((g_)*K3_GROUP + (l_)) * nm3 + (i_)]
#define s_q(j_, t_) s_q_[(j_)*64 + (t_)]
  __shared__ double s_mean[K3_FRAMES_PER_BLOCK][MPE_MAX_MARKERS * 3];
  __shared__ double s_det[K3_FRAMES_PER_BLOCK][MPE_MAX_DETECTIONS][2];
  __shared__ double s_pred[K3_FRAMES_PER_BLOCK][MPE_MAX_MARKERS][2];
  __shared__ unsigned s_colmax[K3_FRAMES_PER_BLOCK][MPE_MAX_MARKERS];
  __shared__ unsigned char s_colrow[K3_FRAMES_PER_BLOCK][MPE_MAX_MARKERS];
  __shared__ double s_mk[MPE_MAX_MARKERS][3];
  __shared__ unsigned s_valid[K3_FRAMES_PER_BLOCK][K3_GROUP];
  __shared__ unsigned char s_cm[K3_FRAMES_PER_BLOCK][MPE_MAX_MARKERS], s_cd[K3_FRAMES_PER_BLOCK][MPE_MAX_MARKERS];
  __shared__ int s_nc[K3_FRAMES_PER_BLOCK];

  const int tid = threadIdx.x;
  const int grp = tid >> 4, l = tid & 15;
  const int f = blockIdx.x * K3_FRAMES_PER_BLOCK + grp;
  const bool live = f < n_frames;
  const mpe_detections* d = dets + (live ? f : 0);
  mpe_result* res = results + (live ? f : 0);
  const int n_d = live ? d->n : 0, n_m = sp.n_markers;
  const int dstatus = live ? d->status : 0;
  const uint32_t* H = hist + (size_t)(live ? f : 0) * MPE_HIST_STRIDE;
  const double fx = sp.fx, fy = sp.fy, cx = sp.cx, cy = sp.cy;

  // stage markers and detections in LDS; default output = identity pose, zero covariance
  if (tid < n_m) {
    s_mk[tid][0] = sp.markers[3 * tid];
    s_mk[tid][1] = sp.markers[3 * tid + 1];
    s_mk[tid][2] = sp.markers[3 * tid + 2];
  }
  for (int i = l; i < n_d; i += K3_GROUP) {
    s_det[grp][i][0] = d->undist_xy[2 * i];
    s_det[grp][i][1] = d->undist_xy[2 * i + 1];
  }
  if (nn_pred && live)  // (tracking path: lane 0's nearest-neighbour search below reads LDS, not a chain of global loads)
    for (int i = l; i < n_m; i += K3_GROUP) {
      s_pred[grp][i][0] = nn_pred[(size_t)f * 2 * MPE_MAX_MARKERS + 2 * i];
      s_pred[grp][i][1] = nn_pred[(size_t)f * 2 * MPE_MAX_MARKERS + 2 * i + 1];
    }
  if (live && !corr_in && !nn_pred && MODE != 2)  // (the histogram path: see lane 0 below)
    for (int c = l; c < n_m; c += K3_GROUP) {
      unsigned mv = 0, mr = 0;
      for (int r = 0; r < n_d; ++r) {
        const unsigned v = H[r * MPE_MAX_MARKERS + c];
        if (v > mv) {  // (first row of the column's maximum; an all-zero column keeps row 0)
          mv = v;
          mr = (unsigned)r;
        }
      }
      s_colmax[grp][c] = mv;
      s_colrow[grp][c] = (unsigned char)mr;
    }
  wave_sync();  // (the block is one wave)
  if (live && MODE != 2) {  // (MODE 2: results[f].T holds the start pose for the refinement kernel)
    for (int i = l; i < 16; i += K3_GROUP) res->T[i] = (i % 5 == 0) ? 1.0 : 0.0;
  }
  if (live) {
    for (int i = l; i < 36; i += K3_GROUP) res->cov[i] = 0.0;
    if (corr_out)
      for (int i = l; i < 2 * MPE_MAX_MARKERS; i += K3_GROUP) corr_out[(size_t)f * 2 * MPE_MAX_MARKERS + i] = 0;
  }

  // ---- lane 0 of the group: initialise()'s all-zero test (pose_estimator.cpp:704) and
  //      correspondencesFromHistogram (pose_estimator.cpp:344-370)
  if (l == 0) {
    int n_c = 0;
    bool go0 = live && dstatus == 0 && n_d >= 4 && n_m >= 4;
    if (MODE == 2) go0 = live && dstatus == 0 && n_d >= 1 && n_m >= 1;
    if (go0 && corr_in) {
      // tracking path: correspondences come from findCorrespondences (pose_estimator.cpp:372-392),
      // rows (marker, detection) terminated by a 0 marker; checkCorrespondences starts from them
      const uint32_t* ci = corr_in + (size_t)f * 2 * MPE_MAX_MARKERS;
      while (n_c < MPE_MAX_MARKERS && ci[2 * n_c] != 0) {
        s_cm[grp][n_c] = (unsigned char)ci[2 * n_c];
        s_cd[grp][n_c] = (unsigned char)ci[2 * n_c + 1];
        ++n_c;
      }
      go0 = false;
    }
    if (go0 && nn_pred) {
      // tracking path, correspondences found here: findCorrespondences (pose_estimator.cpp:372-392) —
      // nearest detection of every predicted marker pixel (first minimum wins), kept if within
      // nearest_neighbour_pixel_tolerance_
      for (int i = 0; i < n_m; ++i) {
        double best = __builtin_huge_val();
        int bj = 0;
        const double pu = s_pred[grp][i][0], pv = s_pred[grp][i][1];
        for (int j = 0; j < n_d; ++j) {
          const double du = pu - s_det[grp][j][0], dv = pv - s_det[grp][j][1];
          const double d2 = du * du + dv * dv;
          if (d2 < best) {
            best = d2;
            bj = j + 1;
          }
        }
        if (sqrt(best) <= nn_tol) {
          s_cm[grp][n_c] = (unsigned char)(i + 1);
          s_cd[grp][n_c] = (unsigned char)bj;
          ++n_c;
        }
      }
      go0 = false;
    }
    if (go0) {
      // The reference scans the whole histogram n_m times, column-major, for the first position of the maximum and
      // then zeroes that COLUMN (pose_estimator.cpp:349-368).  Only columns are ever removed, so a column's maximum
      // and the first row that reaches it never change: the group's lanes found them above (one column per lane, n_d
      // independent loads each instead of n_m * n_m * n_d dependent ones here), and a round is the first column, in
      // ascending order, with the largest value still standing.  A removed column stands at 0 with row 0, as in
      // the reference's scan, which matters when hist_thr is 0.
      bool any = false;
      for (int c = 0; c < n_m; ++c) any |= (s_colmax[grp][c] != 0);
      go0 = any;
    }
    if (go0) {
      unsigned removed = 0;  // zeroed columns
      for (int j = 0; j < n_m; ++j) {
        unsigned mv = 0;
        int ri = 0, ci = 0;
        bool first = true;
        for (int c = 0; c < n_m; ++c) {
          const bool gone = (removed >> c) & 1;
          const unsigned v = gone ? 0u : s_colmax[grp][c];
          if (first || v > mv) {
            mv = v;
            ri = gone ? 0 : (int)s_colrow[grp][c];
            ci = c;
            first = false;
          }
        }
        if (mv < sp.hist_thr) break;
        s_cm[grp][n_c] = (unsigned char)(ci + 1);
        s_cd[grp][n_c] = (unsigned char)(ri + 1);
        ++n_c;
        removed |= 1u << ci;
      }
    }
    s_nc[grp] = n_c;
    if (live) {
      res->n_det = n_d;
      res->n_corr = n_c;
      res->gn_iterations = 0;
      res->status = (dstatus != 0) ? dstatus : MPE_FRAME_NO_POSE;
      if (corr_out)
        for (int i = 0; i < n_c; ++i) {
          corr_out[(size_t)f * 2 * MPE_MAX_MARKERS + 2 * i] = s_cm[grp][i];
          corr_out[(size_t)f * 2 * MPE_MAX_MARKERS + 2 * i + 1] = s_cd[grp][i];
        }
    }
  }
  __syncthreads();
  const int n_c = s_nc[grp];
  const bool go = (MODE != 2) && n_c >= 4;

  // ---- checkCorrespondences (pose_estimator.cpp:394-542): the C(n_c,3) P3P validations run 16 at a time,
  //      one per lane of the group; after every round the lanes' inverse(H_best) * markers are added to the
  //      running sums IN COMBINATION ORDER (lane 0 first), i.e. in the reference's summation order for any n_c
  const int nu = n_c - 3;
  const int N = go ? n_c * (n_c - 1) * (n_c - 2) / 6 : 0;
  for (int v = l; v < nm3; v += K3_GROUP) s_mean[grp][v] = 0.0;
  unsigned num_valid = 0;
  for (int r0 = 0; __any(r0 < N); r0 += K3_GROUP) {  // wave-uniform trip count (the barriers below)
    const int ci = r0 + l;
    bool contributes = false;
    do {
      if (ci >= N) break;
      int a, b, c;
      unrank_combo3(ci, n_c, a, b, c);
      V3 fv[3], wp[3];
      {
        const int rows3[3] = {a, b, c};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int mi = s_cm[grp][rows3[k]] - 1, di = s_cd[grp][rows3[k]] - 1;
          wp[k] = {s_mk[mi][0], s_mk[mi][1], s_mk[mi][2]};
          fv[k] = bearing(s_det[grp][di][0], s_det[grp][di][1], fx, fy, cx, cy);
        }
      }
      P3PCtx ctx;
      if (!p3p_prepare(fv[0], fv[1], fv[2], wp[0], wp[1], wp[2], ctx)) break;
      double min_sq = INFINITY;
      int best = -1;
      bool found = false;
#pragma unroll 1
      for (int k = 0; k < 4; ++k) {
        M3 R;
        V3 C;
        p3p_solution(ctx, pick_root(ctx, k), R, C);
        if (!rc_finite(R, C)) continue;
        const Proj P = make_projection(R, C, fx, fy, cx, cy);
        // back-project the unused correspondences' markers, ascending row index
        for (int q = 0; q < nu; ++q) {
          int row = q;
          row += (row >= a);
          row += (row >= b);
          row += (row >= c);
          const int mi = s_cm[grp][row] - 1;
          double u, v;
          project(P, V3{s_mk[mi][0], s_mk[mi][1], s_mk[mi][2]}, u, v);
          s_q(2 * q, tid) = u;
          s_q(2 * q + 1, tid) = v;
        }
        // calculateSquaredReprojectionErrorAndCertainty (pose_estimator.cpp:303-342): greedy
        // global-minimum matching, column-major first minimum, rows = image points
        unsigned rowdone = 0, coldone = 0;
        double sq = 0;
        unsigned ncorr = 0;
        for (int it = 0; it < nu; ++it) {
          double mv = 0;
          int ri = 0, cj0 = 0;
          bool first = true;
          for (int cj = 0; cj < nu; ++cj) {
            const double bu = s_q(2 * cj, tid), bv = s_q(2 * cj + 1, tid);
            for (int rr = 0; rr < nu; ++rr) {
              double v;
              if (((rowdone >> rr) & 1) || ((coldone >> cj) & 1))
                v = INFINITY;
              else {
                int row = rr;
                row += (row >= a);
                row += (row >= b);
                row += (row >= c);
                const int di = s_cd[grp][row] - 1;
                const double du = s_det[grp][di][0] - bu, dv = s_det[grp][di][1] - bv;
                v = sqrt(du * du + dv * dv);
              }
              if (first || v < mv) {
                mv = v;
                ri = rr;
                cj0 = cj;
                first = false;
              }
            }
          }
          if (mv <= sp.back_tol) {
            sq += mv * mv;
            ++ncorr;
            rowdone |= 1u << ri;
            coldone |= 1u << cj0;
          } else
            break;
        }
        const double certainty = (double)ncorr / (double)nu;
        if (certainty >= sp.certainty_thr) {  // pose_estimator.cpp:494-502
          found = true;
          if (sq < min_sq) {
            min_sq = sq;
            best = k;
          }
        }
      }
      if (!found) break;
      if (best < 0) best = 0;  // unreachable: sq is always finite
      M3 R;
      V3 C;
      p3p_solution(ctx, pick_root(ctx, best), R, C);
      // inverse(H) * marker for ALL markers (pose_estimator.cpp:513-517)
      for (int jj = 0; jj < n_m; ++jj) {
        const V3 mk = {s_mk[jj][0] - C.x, s_mk[jj][1] - C.y, s_mk[jj][2] - C.z};
        const V3 pc = mulT(R, mk);  // R^T (m - C)
        s_part(grp, l, 3 * jj) = pc.x;
        s_part(grp, l, 3 * jj + 1) = pc.y;
        s_part(grp, l, 3 * jj + 2) = pc.z;
      }
      contributes = true;
    } while (false);
    s_valid[grp][l] = contributes ? 1u : 0u;
    __syncthreads();
    for (int v = l; v < nm3; v += K3_GROUP) {
      double sacc = s_mean[grp][v];
      for (int q = 0; q < K3_GROUP; ++q)
        if (s_valid[grp][q]) sacc += s_part(grp, q, v);
      s_mean[grp][v] = sacc;
    }
    for (int q = 0; q < K3_GROUP; ++q) num_valid += s_valid[grp][q];
    __syncthreads();
  }
  bool active = go && ((double)num_valid / (double)N >= sp.valid_corr_thr);
  if (MODE == 2) active = n_c >= 3;  // fewer rows leave the 6x6 normal equations singular
  if (live) {
    TailMid* m = mid + f;
    if (l == 0) {
      m->n_c = n_c;
      m->active = active ? 1 : 0;
      m->num_valid = num_valid;
      m->pad = 0;
    }
    if (l < n_c) {
      m->cm[l] = s_cm[grp][l];
      m->cd[l] = s_cd[grp][l];
    }
    if (active)
      for (int v = l; v < nm3; v += K3_GROUP) m->mean[v] = s_mean[grp][v];
  }
#undef s_part
#undef s_q
}

// ---------------------------------------------------------------------------------------------
// K3b  k3b_refine: ONE LANE PER FRAME.  computeTransformation (Kabsch, pose_estimator.cpp:908-930), then
// optimisePose (pose_estimator.cpp:733-792): Gauss-Newton on SE(3), the normal equations accumulated over
// the correspondences one after the other in row order — the reference's summation order —, unpivoted LDL^T,
// exponentialMap update, covariance = inverse of the last iteration's A (pose_estimator.cpp:790).
// Everything is lane-local: no shuffles, no barriers inside the iteration; lanes of frames without a pose
// (or whose iteration has converged) idle.  A 16 384-frame sub-batch is 256 waves of ~8 k instructions.
// MODE 0: Kabsch + GN.  MODE 1: Kabsch only (checkCorrespondences alone).  MODE 2: GN from results[f].T.
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(K3B_THREADS) void k3b_refine(const mpe_detections* __restrict__ dets, int n_frames,
                                                          SolveParams sp, mpe_result* __restrict__ results,
                                                          const TailMid* __restrict__ mid, int row_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3b[];
  // per-lane rows of (marker xyz, detection uv), [row][field][lane]: conflict-free, dynamic row index
  double* s_row = reinterpret_cast<double*>(smem3b);
#define ROW(r_, k_) s_row[((r_)*5 + (k_)) * K3B_THREADS + threadIdx.x]
  const int f = blockIdx.x * K3B_THREADS + threadIdx.x;
  if (f >= n_frames) return;
  const TailMid* m = mid + f;
  if (!m->active) return;
  const mpe_detections* d = dets + f;
  mpe_result* res = results + f;
  const int n_m = sp.n_markers;
  const int n_c = min(m->n_c, row_cap);
  const double fx = sp.fx, fy = sp.fy, cx = sp.cx, cy = sp.cy;
  for (int j = 0; j < n_c; ++j) {
    const int mi = m->cm[j] - 1, di = m->cd[j] - 1;
    ROW(j, 0) = sp.markers[3 * mi];
    ROW(j, 1) = sp.markers[3 * mi + 1];
    ROW(j, 2) = sp.markers[3 * mi + 2];
    ROW(j, 3) = d->undist_xy[2 * di];
    ROW(j, 4) = d->undist_xy[2 * di + 1];
  }
  T34 T;
  if (MODE == 2) {
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) T.m[r][c] = res->T[r * 4 + c];
  } else {
    // ---- computeTransformation (pose_estimator.cpp:908-930)
    const double nv = (double)m->num_valid;
    double mo[3] = {0, 0, 0}, mr[3] = {0, 0, 0};
    for (int i = 0; i < n_m; ++i)
      for (int k = 0; k < 3; ++k) {
        mo[k] += sp.markers[3 * i + k];
        mr[k] += m->mean[3 * i + k] / nv;
      }
    for (int k = 0; k < 3; ++k) {
      mo[k] /= (double)n_m;
      mr[k] /= (double)n_m;
    }
    double Hm[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int i = 0; i < n_m; ++i) {
      const double a[3] = {sp.markers[3 * i] - mo[0], sp.markers[3 * i + 1] - mo[1], sp.markers[3 * i + 2] - mo[2]};
      const double b[3] = {m->mean[3 * i] / nv - mr[0], m->mean[3 * i + 1] / nv - mr[1], m->mean[3 * i + 2] / nv - mr[2]};
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) Hm[r][c] += a[r] * b[c];
    }
    double X[3][3];
    kabsch_rotation(Hm, X);  // R = V U^T
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) T.m[r][c] = X[r][c];
      T.m[r][3] = mr[r] - (X[r][0] * mo[0] + X[r][1] * mo[1] + X[r][2] * mo[2]);
    }
  }

  // ---- optimisePose (pose_estimator.cpp:733-792)
  int iters = 0;
  if (MODE != 1)
    iters = k3_gauss_newton(n_c, [&](int j, int k) -> double { return ROW(j, k); }, fx, fy, cx, cy, T, res->cov);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) res->T[r * 4 + c] = T.m[r][c];
  res->T[12] = 0.0;
  res->T[13] = 0.0;
  res->T[14] = 0.0;
  res->T[15] = 1.0;
  res->gn_iterations = iters;
  res->status = MPE_FRAME_POSE;
#undef ROW
}

// ---------------------------------------------------------------------------------------------
// K3b for SMALL launches (tracked frames, a handful of streams in lock step): the same computeTransformation +
// optimisePose with 16 LANES PER FRAME.  One lane per frame (k3b_refine) is the right shape for 16 384 frames, but a
// single tracked frame then waits for ~18 000 dependent FP64 instructions of one lane (54 us).  Here a Gauss-Newton
// iteration is spread over the group: lane j computes the Jacobian rows of correspondence j, 27 lanes-slots sum the
// 21 + 6 entries of A = sum J^T J and b = sum J^T e over the correspondences IN ROW ORDER (the reference's summation
// order, as in the one-lane kernel), the LDL^T factorisation runs column by column with the five divisions of a column
// on five lanes, the 3x3 matrices of the exponential map one entry per lane.  Every scalar is computed by the same
// sequence of operations as in k3b_refine, so the two kernels return bit-identical poses, covariances and iteration
// counts (tested).  Groups of a wave converge at different iterations; a finished group idles through the others'
// synchronisation points.
// ---------------------------------------------------------------------------------------------
#define K3G_LANES 16
#define K3G_FRAMES 4
struct GnGroupLds {
  double J[MPE_MAX_MARKERS][14];  // per correspondence row: J0[6], J1[6], e0, e1
  double Ab[27];                  // upper triangle of A (21, row-major) then b (6)
  double L[6][6];
  double D[6];
  double O[9], O2[9], Rm[9], Vm[9];
  double T[12];
};
// index of A[r][c], r <= c, in the packed upper triangle
__device__ __forceinline__ int k3g_tri(int r, int c) { return r * 6 - (r * (r - 1)) / 2 + (c - r); }

template <int MODE>
__global__ __launch_bounds__(64) void k3b_refine_group(const mpe_detections* __restrict__ dets, int n_frames, SolveParams sp,
                                                       mpe_result* __restrict__ results, const TailMid* __restrict__ mid,
                                                       int row_cap) {
  __shared__ GnGroupLds s_g[K3G_FRAMES];
  const int tid = threadIdx.x;
  const int grp = tid >> 4, l = tid & 15;
  GnGroupLds& G = s_g[grp];
  const int f = blockIdx.x * K3G_FRAMES + grp;
  const bool in_range = f < n_frames;
  const TailMid* m = mid + (in_range ? f : 0);
  const bool live = in_range && m->active;
  const mpe_detections* d = dets + (in_range ? f : 0);
  mpe_result* res = results + (in_range ? f : 0);
  const int n_m = sp.n_markers;
  const int n_c = live ? min(m->n_c, row_cap) : 0;
  const double fx = sp.fx, fy = sp.fy, cx = sp.cx, cy = sp.cy;
  // this lane's correspondence row
  double mk[3] = {0, 0, 0}, du = 0, dv = 0;
  if (l < n_c) {
    const int mi = m->cm[l] - 1, di = m->cd[l] - 1;
    mk[0] = sp.markers[3 * mi];
    mk[1] = sp.markers[3 * mi + 1];
    mk[2] = sp.markers[3 * mi + 2];
    du = d->undist_xy[2 * di];
    dv = d->undist_xy[2 * di + 1];
  }
  // ---- start pose: computeTransformation (pose_estimator.cpp:908-930) by lane 0, or the given pose (MODE 2)
  if (live && l == 0) {
    T34 T0;
    if (MODE == 2) {
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) T0.m[r][c] = res->T[r * 4 + c];
    } else {
      const double nv = (double)m->num_valid;
      double mo[3] = {0, 0, 0}, mr[3] = {0, 0, 0};
      for (int i = 0; i < n_m; ++i)
        for (int k = 0; k < 3; ++k) {
          mo[k] += sp.markers[3 * i + k];
          mr[k] += m->mean[3 * i + k] / nv;
        }
      for (int k = 0; k < 3; ++k) {
        mo[k] /= (double)n_m;
        mr[k] /= (double)n_m;
      }
      double Hm[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      for (int i = 0; i < n_m; ++i) {
        const double a[3] = {sp.markers[3 * i] - mo[0], sp.markers[3 * i + 1] - mo[1], sp.markers[3 * i + 2] - mo[2]};
        const double b[3] = {m->mean[3 * i] / nv - mr[0], m->mean[3 * i + 1] / nv - mr[1], m->mean[3 * i + 2] / nv - mr[2]};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) Hm[r][c] += a[r] * b[c];
      }
      double X[3][3];
      kabsch_rotation(Hm, X);  // R = V U^T
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) T0.m[r][c] = X[r][c];
        T0.m[r][3] = mr[r] - (X[r][0] * mo[0] + X[r][1] * mo[1] + X[r][2] * mo[2]);
      }
    }
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) G.T[r * 4 + c] = T0.m[r][c];
  }
  wave_sync();
  T34 T;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) T.m[r][c] = live ? G.T[r * 4 + c] : ((r == c) ? 1.0 : 0.0);

  // ---- optimisePose (pose_estimator.cpp:733-792)
  bool done = !live || MODE == 1;
  int iters = 0;
  for (int it = 0; it < 500; ++it) {
    if (__builtin_amdgcn_ballot_w64(!done) == 0) break;  // every group of the wave has converged
    // (a) Jacobian rows, computeJacobian (pose_estimator.cpp:945-957): lane j <- correspondence j
    if (!done && l < n_c) {
      double u, v, x, y, z;
      project_T(T, mk, fx, fy, cx, cy, u, v, x, y, z);
      const double e0 = du - u, e1 = dv - v;
      const double z_2 = z * z;
      double* Jr = G.J[l];
      Jr[0] = 1 / z * fx;
      Jr[1] = 0;
      Jr[2] = -x / z_2 * fx;
      Jr[3] = -x * y / z_2 * fx;
      Jr[4] = (1 + (x * x / z_2)) * fx;
      Jr[5] = -y / z * fx;
      Jr[6] = 0;
      Jr[7] = 1 / z * fy;
      Jr[8] = -y / z_2 * fy;
      Jr[9] = -(1 + y * y / z_2) * fy;
      Jr[10] = x * y / z_2 * fy;
      Jr[11] = x / z * fy;
      Jr[12] = e0;
      Jr[13] = e1;
    }
    wave_sync();
    // (b) A = sum J^T J (upper triangle), b = sum J^T e, each entry summed over the rows in row order
    if (!done) {
      for (int q = l; q < 27; q += K3G_LANES) {
        double acc = 0;
        if (q < 21) {
          int r = 0, base = 0;
          while (q >= base + (6 - r)) {
            base += 6 - r;
            ++r;
          }
          const int c = r + (q - base);
          for (int j = 0; j < n_c; ++j) acc += G.J[j][r] * G.J[j][c] + G.J[j][6 + r] * G.J[j][6 + c];
        } else {
          const int r = q - 21;
          for (int j = 0; j < n_c; ++j) acc += G.J[j][r] * G.J[j][12] + G.J[j][6 + r] * G.J[j][13];
        }
        G.Ab[q] = acc;
      }
    }
    wave_sync();
    // (c) unpivoted LDL^T (ldl6_factor), column by column: lane j the pivot, lanes j+1..5 the column's divisions
    double Lrow[6] = {0, 0, 0, 0, 0, 0};  // row l of L (lanes 0..5)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if (!done && l == j) {
        double dd = G.Ab[k3g_tri(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) dd -= Lrow[k] * Lrow[k] * G.D[k];
        G.D[j] = dd;
      }
      wave_sync();
      if (!done && l > j && l < 6) {
        double sacc = G.Ab[k3g_tri(j, l)];  // A[l][j] = A[j][l]
#pragma unroll
        for (int k = 0; k < j; ++k) sacc -= Lrow[k] * G.L[j][k] * G.D[k];
        Lrow[j] = sacc / G.D[j];
        G.L[l][j] = Lrow[j];
      }
      wave_sync();
    }
    // (d) ldl6_solve, every lane for itself (a chain of 36 dependent operations: nothing to spread)
    double dT[6] = {0, 0, 0, 0, 0, 0};
    if (!done) {
      double yv[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        double sacc = G.Ab[21 + i];
#pragma unroll
        for (int k = 0; k < i; ++k) sacc -= G.L[i][k] * yv[k];
        yv[i] = sacc;
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) yv[i] /= G.D[i];
#pragma unroll
      for (int i = 5; i >= 0; --i) {
        double sacc = yv[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) sacc -= G.L[k][i] * dT[k];
        dT[i] = sacc;
      }
    }
    // (e) exponentialMap(dT) * T (apply_exp), the 3x3 matrices one entry per lane
    const double ux = dT[0], uy = dT[1], uz = dT[2], wx = dT[3], wy = dT[4], wz = dT[5];
    const double theta = sqrt(wx * wx + wy * wy + wz * wz);
    const double th2 = theta * theta;
    if (!done && l < 9) {
      const double Ov[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
      double o = Ov[0];
#pragma unroll
      for (int q = 1; q < 9; ++q) o = (l == q) ? Ov[q] : o;
      G.O[l] = o;
    }
    wave_sync();
    if (!done && l < 9) {
      const int i = l / 3, j = l - 3 * i;
      const double o2 = G.O[3 * i] * G.O[j] + G.O[3 * i + 1] * G.O[3 + j] + G.O[3 * i + 2] * G.O[6 + j];
      const double o = G.O[l];
      const double I = (i == j) ? 1.0 : 0.0;
      double rm = I, vm = I;
      if (theta != 0) {
        double st, ct;
        sincos(theta, &st, &ct);
        rm = I + o / theta * st + o2 / th2 * (1 - ct);
        vm = I + (1 - ct) / th2 * o + (theta - st) / (th2 * theta) * o2;
      }
      G.Rm[l] = rm;
      G.Vm[l] = vm;
    }
    wave_sync();
    double nvv = 0;
    if (!done && l < 12) {
      const int i = l >> 2, j = l & 3;
      double sacc = G.Rm[3 * i] * G.T[j] + G.Rm[3 * i + 1] * G.T[4 + j] + G.Rm[3 * i + 2] * G.T[8 + j];
      if (j == 3) sacc += G.Vm[3 * i] * ux + G.Vm[3 * i + 1] * uy + G.Vm[3 * i + 2] * uz;
      nvv = sacc;
    }
    wave_sync();  // (every read of the old pose is done)
    if (!done && l < 12) G.T[l] = nvv;
    wave_sync();
    if (!done) {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) T.m[r][c] = G.T[r * 4 + c];
      iters = it + 1;
      double mx = -1;  // norm_max, pose_estimator.cpp:1073-1085
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const double av = fabs(dT[r]);
        if (av > mx) mx = av;
      }
      if (mx <= 1e-13) done = true;
    }
  }
  if (!live) return;
  // pose_covariance_ = A.inverse() with the A of the last iteration (pose_estimator.cpp:790): its factors are still in
  // G.L / G.D; column c of the inverse on lane c
  if (MODE != 1 && l < 6) {
    double yv[6], xv[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      double sacc = (i == l) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < i; ++k) sacc -= G.L[i][k] * yv[k];
      yv[i] = sacc;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) yv[i] /= G.D[i];
#pragma unroll
    for (int i = 5; i >= 0; --i) {
      double sacc = yv[i];
#pragma unroll
      for (int k = i + 1; k < 6; ++k) sacc -= G.L[k][i] * xv[k];
      xv[i] = sacc;
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) res->cov[r * 6 + l] = xv[r];
  }
  if (l < 12) res->T[l] = G.T[l];  // (the group's latest pose)
  if (l == 0) {
    res->T[12] = 0.0;
    res->T[13] = 0.0;
    res->T[14] = 0.0;
    res->T[15] = 1.0;
    res->gn_iterations = iters;
    res->status = MPE_FRAME_POSE;
  }
}

hipError_t launch_k3_tail(const mpe_detections* dets, const uint32_t* hist, int n_frames, const SolveParams& sp,
                          mpe_result* results, uint32_t* corr_out, const uint32_t* corr_in, const double* nn_pred,
                          double nn_tol, void* mid_buf, hipStream_t s, int mode) {
  if (n_frames <= 0) return hipSuccess;
  const int refine_variant = sp.refine_variant;
  TailMid* mid = static_cast<TailMid*>(mid_buf);
  // explicit correspondences (corr_in) may hold up to MPE_MAX_MARKERS rows, also more than n_markers (repeated
  // markers are defined input for checkCorrespondences): size the row buffers for the row capacity
  const int rows = corr_in ? MPE_MAX_MARKERS : (sp.n_markers > 3 ? sp.n_markers : 4);
  const int nu = rows - 3;
  const size_t lds_a = ((size_t)K3_FRAMES_PER_BLOCK * K3_GROUP * 3 * sp.n_markers + (size_t)2 * nu * 64) * sizeof(double);
  const size_t lds_b = (size_t)rows * 5 * K3B_THREADS * sizeof(double);
  const dim3 grid_a((n_frames + K3_FRAMES_PER_BLOCK - 1) / K3_FRAMES_PER_BLOCK);
  const dim3 grid_b((n_frames + K3B_THREADS - 1) / K3B_THREADS);
  const dim3 grid_g((n_frames + K3G_FRAMES - 1) / K3G_FRAMES);
  // refinement: 16 lanes per frame while the launch is too small to fill the chip with one lane per frame (a tracked
  // frame, a few hundred streams in lock step), else one lane per frame; bit-identical results (refine_variant forces
  // one of them: 1 = lane, 2 = group)
  const bool group = refine_variant == 2 || (refine_variant == 0 && n_frames <= 2048);
#define K3_LAUNCH(M_)                                                                                              \
  do {                                                                                                             \
    hipLaunchKernelGGL(k3a_validate<M_>, grid_a, dim3(64), lds_a, s, dets, hist, n_frames, sp, results, corr_out,  \
                       corr_in, nn_pred, nn_tol, mid);                                                             \
    if (group)                                                                                                     \
      hipLaunchKernelGGL(k3b_refine_group<M_>, grid_g, dim3(64), 0, s, dets, n_frames, sp, results,                \
                         (const TailMid*)mid, rows);                                                               \
    else                                                                                                           \
      hipLaunchKernelGGL(k3b_refine<M_>, grid_b, dim3(K3B_THREADS), lds_b, s, dets, n_frames, sp, results,         \
                         (const TailMid*)mid, rows);                                                               \
  } while (0)
  if (mode == 1)
    K3_LAUNCH(1);
  else if (mode == 2)
    K3_LAUNCH(2);
  else
    K3_LAUNCH(0);
#undef K3_LAUNCH
  return hipGetLastError();
}

// =============================================================================================
// Primitive batches — P3P::computePoses / P3P::solveQuartic (p3p.h:110-127) for n independent
// problems, one lane each.  Not on the batch path (K2 / K3 inline the same device functions); they
// exist for callers of the static primitives and for stage-level parity tests of mpe_p3p.h.
// =============================================================================================
__global__ __launch_bounds__(64) void k_p3p_batch(const double* __restrict__ fv, const double* __restrict__ wp, int n,
                                                 double* __restrict__ sol, int* __restrict__ status) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* f = fv + (size_t)i * 9;
  const double* w = wp + (size_t)i * 9;
  const V3 f0 = {f[0], f[1], f[2]}, f1 = {f[3], f[4], f[5]}, f2 = {f[6], f[7], f[8]};
  const V3 w0 = {w[0], w[1], w[2]}, w1 = {w[3], w[4], w[5]}, w2 = {w[6], w[7], w[8]};
  P3PCtx c;
  if (!p3p_prepare(f0, f1, f2, w0, w1, w2, c)) {
    status[i] = -1;  // collinear world points, p3p.cpp:77-80; solutions left untouched like the reference
    return;
  }
  double* o = sol + (size_t)i * 48;
  for (int k = 0; k < 4; ++k) {
    M3 R;
    V3 C;
    p3p_solution(c, pick_root(c, k), R, C);
    double* q = o + 12 * k;
    q[0] = R.r0.x; q[1] = R.r0.y; q[2] = R.r0.z; q[3] = C.x;
    q[4] = R.r1.x; q[5] = R.r1.y; q[6] = R.r1.z; q[7] = C.y;
    q[8] = R.r2.x; q[9] = R.r2.y; q[10] = R.r2.z; q[11] = C.z;
  }
  status[i] = 0;
}

__global__ __launch_bounds__(64) void k_quartic_batch(const double* __restrict__ factors, int n, int variant,
                                                     double* __restrict__ roots) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* a = factors + (size_t)i * 5;
  double r[4];
  if (variant == 1)
    solve_quartic_lit2(a[0], a[1], a[2], a[3], a[4], r);  // the voting kernel's variant
  else
    solve_quartic(a[0], a[1], a[2], a[3], a[4], r);  // IEEE operators (validation kernel)
  for (int k = 0; k < 4; ++k) roots[(size_t)i * 4 + k] = r[k];
}

// =============================================================================================
// frame decode: sensor_msgs/Image payloads -> mono8 (what cv_bridge::toCvCopy(msg, MONO8) does for the node,
// monocular_pose_estimator.cpp:147).  HBM bound, one pass: 4 output pixels per lane and step.
//   bgr8 / rgb8 / bgra8 / rgba8: cv::cvtColor(..., COLOR_*2GRAY) for CV_8U — integer, 14 fractional bits,
//       Y = (B * 1868 + G * 9617 + R * 4899 + 2^13) >> 14   (OpenCV 2.4, 3.0 .. 3.4.1; from 3.4.2 on: 15 bits, see mpe.h)
//   mono16 (host byte order after cv_bridge's endianness fix): Mat::convertTo(CV_8U, 255. / 65535.) —
//       saturate_cast<uchar>((float)v * (float)(255. / 65535.)), i.e. round-half-even of the single-precision product
// =============================================================================================
__device__ __forceinline__ unsigned gray_px(unsigned c0, unsigned c1, unsigned c2, bool rgb) {
  const unsigned b = rgb ? c2 : c0, r = rgb ? c0 : c2;
  return (b * 1868u + c1 * 9617u + r * 4899u + (1u << 13)) >> 14;
}
__global__ __launch_bounds__(256) void k_to_mono8(const uint8_t* __restrict__ src, size_t src_stride, size_t src_frame_stride,
                                                  int encoding, int big_endian, int rows, int cols, long long n_rows_total,
                                                  uint8_t* __restrict__ dst) {
  const int quads = (cols + 3) >> 2;  // 4 output pixels per work item
  const long long total = n_rows_total * quads;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / quads;
    const int x0 = (int)(i - row * quads) * 4;
    const long long f = row / rows;
    const int y = (int)(row - f * rows);
    const uint8_t* s = src + (size_t)f * src_frame_stride + (size_t)y * src_stride;
    uint8_t* d = dst + ((size_t)f * rows + y) * cols + x0;
    const int n = min(4, cols - x0);
    unsigned out[4] = {0, 0, 0, 0};
    if (encoding == MPE_ENC_MONO16) {
      for (int k = 0; k < n; ++k) {
        const uint8_t* p = s + 2 * (size_t)(x0 + k);
        const unsigned v = big_endian ? ((unsigned)p[0] << 8 | p[1]) : ((unsigned)p[1] << 8 | p[0]);
        float r = rintf((float)v * (float)(255.0 / 65535.0));
        r = fminf(fmaxf(r, 0.f), 255.f);
        out[k] = (unsigned)r;
      }
    } else if (encoding == MPE_ENC_MONO8) {
      for (int k = 0; k < n; ++k) out[k] = s[x0 + k];
    } else {
      const int bpp = (encoding == MPE_ENC_BGRA8 || encoding == MPE_ENC_RGBA8) ? 4 : 3;
      const bool rgb = encoding == MPE_ENC_RGB8 || encoding == MPE_ENC_RGBA8;
      const uint8_t* p = s + (size_t)bpp * x0;
      if (n == 4 && ((reinterpret_cast<uintptr_t>(p) & 3) == 0)) {  // three or four aligned 32-bit loads
        const unsigned* w = reinterpret_cast<const unsigned*>(p);
        if (bpp == 3) {
          const unsigned w0 = w[0], w1 = w[1], w2 = w[2];
          out[0] = gray_px(w0 & 0xFF, (w0 >> 8) & 0xFF, (w0 >> 16) & 0xFF, rgb);
          out[1] = gray_px(w0 >> 24, w1 & 0xFF, (w1 >> 8) & 0xFF, rgb);
          out[2] = gray_px((w1 >> 16) & 0xFF, w1 >> 24, w2 & 0xFF, rgb);
          out[3] = gray_px((w2 >> 8) & 0xFF, (w2 >> 16) & 0xFF, w2 >> 24, rgb);
        } else {
          for (int k = 0; k < 4; ++k) out[k] = gray_px(w[k] & 0xFF, (w[k] >> 8) & 0xFF, (w[k] >> 16) & 0xFF, rgb);
        }
      } else {
        for (int k = 0; k < n; ++k) out[k] = gray_px(p[bpp * k], p[bpp * k + 1], p[bpp * k + 2], rgb);
      }
    }
    if (n == 4 && ((reinterpret_cast<uintptr_t>(d) & 3) == 0)) {
      *reinterpret_cast<unsigned*>(d) = out[0] | (out[1] << 8) | (out[2] << 16) | (out[3] << 24);
    } else {
      for (int k = 0; k < n; ++k) d[k] = (uint8_t)out[k];
    }
  }
}

hipError_t launch_to_mono8(const uint8_t* src, size_t src_stride, size_t src_frame_stride, int encoding, int big_endian,
                           int n_frames, int rows, int cols, uint8_t* dst, hipStream_t s) {
  if (n_frames <= 0 || rows <= 0 || cols <= 0) return hipSuccess;
  const long long n_rows = (long long)n_frames * rows;
  const long long items = n_rows * ((cols + 3) / 4);
  long long blocks = (items + 255) / 256;
  const long long most = (long long)device_cu_count() * 64;  // grid-stride beyond 64 blocks per CU
  if (blocks > most) blocks = most;
  hipLaunchKernelGGL(k_to_mono8, dim3((unsigned)blocks), dim3(256), 0, s, src, src_stride, src_frame_stride, encoding,
                     big_endian, rows, cols, n_rows, dst);
  return hipGetLastError();
}

// one wave that keeps a CU slot busy for `ticks` of the constant-rate counter (100 MHz): used once per
// handle to find two side streams that really run concurrently (see pick_concurrent_streams)
__global__ void k_spin(unsigned long long ticks, unsigned long long* sink) {
  const unsigned long long t0 = wall_clock64();
  unsigned long long t = t0;
  while (t - t0 < ticks) {
    __builtin_amdgcn_s_sleep(32);
    t = wall_clock64();
  }
  if (sink && threadIdx.x == 0) *sink = t - t0;
}

hipError_t launch_spin(unsigned long long ticks, hipStream_t s) {
  hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, ticks, (unsigned long long*)nullptr);
  return hipGetLastError();
}

hipError_t launch_p3p_batch(const double* fv, const double* wp, int n, double* sol, int* status, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_p3p_batch, dim3((n + 63) / 64), dim3(64), 0, s, fv, wp, n, sol, status);
  return hipGetLastError();
}

hipError_t launch_quartic_batch(const double* factors, int n, int variant, double* roots, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_quartic_batch, dim3((n + 63) / 64), dim3(64), 0, s, factors, n, variant, roots);
  return hipGetLastError();
}

}  // namespace mpe
