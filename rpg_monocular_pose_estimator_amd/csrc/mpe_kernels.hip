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
//   K3   k3_tail    : one lane per frame; histogram peeling, checkCorrespondences, Kabsch,
//                     Gauss-Newton refine + covariance.  (pose_estimator.cpp:344-370, 394-542,
//                     733-792, 908-994)
//
// FP64 everywhere on the geometry path (the reference is double; vote thresholds are knife
// edges), no MFMA (no dense contraction on this path), compiled with -ffp-contract=off.
#include "mpe_internal.h"
#include "mpe_p3p.h"

namespace mpe {

typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// =============================================================================================
// K1a — image scan
// =============================================================================================
// bytes > thr  <=>  byte + (255 - thr) carries into bit 8 of its 16-bit lane
__device__ __forceinline__ unsigned any_gt16(const uint4& v, unsigned add) {
  unsigned r = 0;
  r |= ((v.x & 0x00FF00FFu) + add) | (((v.x >> 8) & 0x00FF00FFu) + add);
  r |= ((v.y & 0x00FF00FFu) + add) | (((v.y >> 8) & 0x00FF00FFu) + add);
  r |= ((v.z & 0x00FF00FFu) + add) | (((v.z >> 8) & 0x00FF00FFu) + add);
  r |= ((v.w & 0x00FF00FFu) + add) | (((v.w >> 8) & 0x00FF00FFu) + add);
  return r & 0x01000100u;
}

#define K1A_UNROLL 4
__global__ __launch_bounds__(256) void k1a_scan(const uint4* __restrict__ px, u64* __restrict__ flags, size_t n_seg,
                                                unsigned add) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
  const size_t n_chunks = (n_seg + 64 * K1A_UNROLL - 1) / (64 * K1A_UNROLL);
  for (size_t c = wave; c < n_chunks; c += n_waves) {
    const size_t base = c * (64 * K1A_UNROLL) + lane;
    uint4 v[K1A_UNROLL];
#pragma unroll
    for (int k = 0; k < K1A_UNROLL; ++k) {
      const size_t idx = base + 64 * k;
      if (idx < n_seg) {
        const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(px) + idx);
        v[k] = make_uint4(t.x, t.y, t.z, t.w);
      } else {
        v[k] = make_uint4(0, 0, 0, 0);
      }
    }
    u64 b[K1A_UNROLL];
#pragma unroll
    for (int k = 0; k < K1A_UNROLL; ++k) b[k] = __ballot(any_gt16(v[k], add) != 0);
    if (lane == 0) {
      ulonglong2* out = reinterpret_cast<ulonglong2*>(flags + c * K1A_UNROLL);
      out[0] = make_ulonglong2(b[0], b[1]);
      out[1] = make_ulonglong2(b[2], b[3]);
    }
  }
}

hipError_t launch_k1a_scan(const uint8_t* frames, size_t n_bytes, unsigned long long* flags, int thr,
                           hipStream_t s) {
  const size_t n_seg = n_bytes / 16;
  if (n_seg == 0) return hipSuccess;
  int t = thr < -1 ? -1 : (thr > 255 ? 255 : thr);
  const unsigned add = (unsigned)(255 - t) * 0x00010001u;
  const size_t n_chunks = (n_seg + 64 * K1A_UNROLL - 1) / (64 * K1A_UNROLL);
  size_t blocks = (n_chunks + 3) / 4;  // 4 waves per block
  const size_t max_blocks = 256 * 8;   // 256 CUs x 8 blocks, grid-stride beyond
  if (blocks > max_blocks) blocks = max_blocks;
  hipLaunchKernelGGL(k1a_scan, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const uint4*>(frames),
                     (u64*)flags, n_seg, add);
  return hipGetLastError();
}

// =============================================================================================
// repack — copy an ROI of strided frames into the packed layout (pitch % 16 == 0, zero padded).
// Used for host frames with odd strides / widths and for ROI detection (the reference clones the
// ROI into a stand-alone matrix, led_detector.cpp:44).
// =============================================================================================
__global__ void repack_kernel(const uint8_t* __restrict__ src, size_t src_stride, size_t src_frame_stride,
                              int roi_x, int roi_y, int roi_w, int roi_h, uint8_t* __restrict__ dst, int dst_pitch) {
  const int f = blockIdx.z;
  const int y = blockIdx.y;
  const uint8_t* s = src + (size_t)f * src_frame_stride + (size_t)(roi_y + y) * src_stride + roi_x;
  uint8_t* d = dst + ((size_t)f * roi_h + y) * dst_pitch;
  for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < dst_pitch; x += gridDim.x * blockDim.x)
    d[x] = (x < roi_w) ? s[x] : (uint8_t)0;
}

hipError_t launch_repack(const uint8_t* src, size_t src_stride, size_t src_frame_stride, int n_frames, int roi_x,
                         int roi_y, int roi_w, int roi_h, uint8_t* dst, int dst_pitch, hipStream_t s) {
  if (n_frames <= 0 || roi_h <= 0) return hipSuccess;
  dim3 grid((dst_pitch + 255) / 256, roi_h, n_frames);
  hipLaunchKernelGGL(repack_kernel, grid, dim3(256), 0, s, src, src_stride, src_frame_stride, roi_x, roi_y, roi_w,
                     roi_h, dst, dst_pitch);
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
#define K1B_SEG_CAP 1024   // bright segments per frame kept in LDS
#define K1B_PIX_CAP 8192   // bytes of thresholded pixels per island window
#define K1B_BM_CAP 320     // u64 words per island bitmap
#define K1B_KEPT_CAP 64    // blobs that pass the shape filter (> MPE_MAX_DETECTIONS -> status)

struct BlobRec {
  long long a00, a10, a01;  // polygon sums: sum dxy, sum dxy*(x_{i-1}+x_i), sum dxy*(y_{i-1}+y_i)
  int xmin, xmax, ymin, ymax;
};

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
};

// fixed-point Gaussian for the 16 outputs x0..x0+15 of image row y, reading the LDS window.
// Returns the bit mask of outputs whose blurred value is non-zero: (sum + 2^15) >> 16 != 0.
template <int KS>
__device__ __forceinline__ unsigned blur_item_fast(const PixWin& w, int rows, int cols, int y, int c,
                                                   const int* __restrict__ taps) {
  constexpr int R = KS / 2;
  int acc[16];
#pragma unroll
  for (int x = 0; x < 16; ++x) acc[x] = 0;
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    const int yb = reflect101(y + i - R, rows) - w.ylo;
    if ((unsigned)yb >= (unsigned)w.H) continue;  // rows outside the band hold no bright pixel
    const uint4* p = reinterpret_cast<const uint4*>(w.pix + (size_t)yb * w.PW + 16 * (c - 1 - w.pwc0));
    const uint4 q0 = p[0], q1 = p[1], q2 = p[2];
    const unsigned q[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
    int t[16 + 2 * R];
#pragma unroll
    for (int j = 0; j < 16 + 2 * R; ++j) {
      const int k = 16 - R + j;
      t[j] = (int)((q[k >> 2] >> (8 * (k & 3))) & 0xFFu);
    }
    const int ky = taps[i];
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      int h = 0;
#pragma unroll
      for (int j = 0; j < KS; ++j) h += taps[j] * t[x + j];
      acc[x] += ky * h;
    }
  }
  unsigned m = 0;
#pragma unroll
  for (int x = 0; x < 16; ++x)
    if (acc[x] >= (1 << 15)) m |= 1u << x;
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
        if ((unsigned)so < (unsigned)w.PW) h += taps[j] * (int)w.pix[(size_t)yb * w.PW + so];
      }
      acc += taps[i] * h;
    }
    if (acc >= (1 << 15)) m |= 1u << x;
  }
  return m;
}

// three bits (x-1, x, x+1) of a bitmap row at bit index xb (>= 1)
__device__ __forceinline__ unsigned bits3(const u64* row, int xb) {
  const int lo = xb - 1, wi = lo >> 6, sh = lo & 63;
  const u64 a = row[wi], b = row[wi + 1];
  const u64 v = sh ? ((a >> sh) | (b << (64 - sh))) : a;
  return (unsigned)(v & 7);
}
// 8-neighbourhood occupancy, bit d = direction d non-zero; directions as OpenCV's chain codes:
// 0 E, 1 NE, 2 N, 3 NW, 4 W, 5 SW, 6 S, 7 SE (y grows downwards)
__device__ __forceinline__ unsigned neighbours(const u64* nz, int wb, int slot, int xb) {
  const unsigned u = bits3(nz + (size_t)(slot - 1) * wb, xb);
  const unsigned m = bits3(nz + (size_t)slot * wb, xb);
  const unsigned d = bits3(nz + (size_t)(slot + 1) * wb, xb);
  return ((m >> 2) & 1) | (((u >> 2) & 1) << 1) | (((u >> 1) & 1) << 2) | ((u & 1) << 3) | ((m & 1) << 4) |
         ((d & 1) << 5) | (((d >> 1) & 1) << 6) | (((d >> 2) & 1) << 7);
}
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
__device__ bool trace_outer_border(const u64* nz, u64* pm, u64* ng, int wb, int slot0, int xb0, int xoff, int yoff,
                                   PolyAcc& acc) {
  acc.init();
  unsigned nb = neighbours(nz, wb, slot0, xb0);
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
  const int x1b = xb0 + dir_dx(s), s1 = slot0 + dir_dy(s);
  int xb = xb0, slot = slot0;
  for (int step = 0; step < (1 << 20); ++step) {
    const int s_end = s;
    const unsigned m16 = nb | (nb << 8);
    const int k = __builtin_ctz(m16 >> (s + 1));
    const int sn = (s + 1 + k) & 7;
    if ((unsigned)(sn - 1) < (unsigned)s_end)
      set_bit(ng, wb, slot, xb);
    else
      set_bit(pm, wb, slot, xb);
    acc.emit(xb + xoff, slot + yoff);
    const int nxb = xb + dir_dx(sn), nslot = slot + dir_dy(sn);
    if (nxb == xb0 && nslot == slot0 && xb == x1b && slot == s1) {
      acc.close();
      return true;
    }
    slot = nslot;
    xb = nxb;
    s = (sn + 4) & 7;
    nb = neighbours(nz, wb, slot, xb);
  }
  return false;
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
__device__ __forceinline__ bool blob_filter(const BlobRec& b, const DetectParams& dp, float& mcx, float& mcy) {
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
  mcx = (float)(m10 / m00) + (float)dp.roi_x;
  mcy = (float)(m01 / m00) + (float)dp.roi_y;
  const double w = (double)width, h = (double)height;
  const double hw = (double)(width / 2), hh = (double)(height / 2);  // INTEGER halves (quirk A.6.2)
  const double pi = 3.1415926535897932384626433832795;
  return area >= dp.min_area && area <= dp.max_area && fabs(1 - fmin(w / h, h / w)) <= dp.max_wh &&
         fabs(1 - (area / (pi * (hw * hw)))) <= dp.max_circ && fabs(1 - (area / (pi * (hh * hh)))) <= dp.max_circ;
}

__global__ __launch_bounds__(64) void k1b_blobs(const uint8_t* __restrict__ frames, const u64* __restrict__ flags,
                                               FrameGeom g, DetectParams dp, mpe_detections* __restrict__ dets) {
  __shared__ __attribute__((aligned(16))) uint8_t s_pix[K1B_PIX_CAP];
  __shared__ u64 s_nz[K1B_BM_CAP], s_pm[K1B_BM_CAP], s_ng[K1B_BM_CAP];
  __shared__ unsigned s_seg[K1B_SEG_CAP];  // y << 16 | segment column
  __shared__ u64 s_rowact[64];
  __shared__ u64 s_colocc[4];
  __shared__ float s_kx[K1B_KEPT_CAP], s_ky[K1B_KEPT_CAP];
  __shared__ unsigned s_kkey[K1B_KEPT_CAP];
  __shared__ int s_nseg, s_nkept, s_over;

  const int lane = threadIdx.x;
  const int f = blockIdx.x;
  const uint8_t* frame = frames + (size_t)f * g.rows * g.pitch;
  mpe_detections* out = dets + f;
  const int r = dp.ksize / 2;
  const int dc = (r + 15) / 16;  // segment columns a bright segment can influence on each side
  const int spr = g.segs_per_row;
  const unsigned add = (unsigned)(255 - dp.thr) * 0x00010001u;

  s_rowact[lane] = 0;
  if (lane == 0) {
    s_nseg = 0;
    s_nkept = 0;
    s_over = 0;
  }
  __syncthreads();

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
          if (slot < K1B_SEG_CAP) s_seg[slot] = ((unsigned)y0 << 16) | (unsigned)c0;
          lds_set_range(s_rowact, max(0, y0 - r), min(g.rows - 1, y0 + r));
        }
      }
    }
  }
  __syncthreads();
  const int nseg_all = s_nseg;
  if (nseg_all == 0 || nseg_all > K1B_SEG_CAP) {
    if (lane == 0) {
      out->n = 0;
      out->status = nseg_all == 0 ? 0 : MPE_FRAME_TOO_MANY_ROWS;
    }
    return;
  }
  const int nseg = nseg_all;

  // ---- B: bands (maximal runs of active rows), then islands (runs of occupied segment columns)
  for (int rwi = 0; rwi < g.rw; ++rwi) {
    u64 act = s_rowact[rwi];
    while (act) {
      // band start inside this word (or continuing from the previous one is handled below)
      const int b0 = __builtin_ctzll(act);
      int ylo = rwi * 64 + b0;
      // extend downwards across words
      int yhi = ylo;
      {
        int wi = rwi;
        u64 run = act >> b0;  // bit 0 = row ylo
        int base = ylo;
        for (;;) {
          const u64 inv = ~run;
          const int len = inv ? __builtin_ctzll(inv) : 64;
          const int avail = 64 - (base & 63);
          if (len < avail) {
            yhi = base + len - 1;
            break;
          }
          // run reaches the end of this word: continue in the next one
          yhi = base + avail - 1;
          ++wi;
          if (wi >= g.rw) break;
          run = s_rowact[wi];
          base = wi * 64;
          if (!(run & 1)) break;
        }
      }
      // clear the band's bits in the loop state (only those inside word rwi matter for `act`)
      {
        const int end_in_word = min(yhi, rwi * 64 + 63) - rwi * 64;
        const u64 m = (end_in_word == 63 ? ~0ull : ((2ull << end_in_word) - 1)) & (~0ull << b0);
        act &= ~m;
      }
      const bool crosses = yhi > rwi * 64 + 63;

      // segment-column occupancy of the band
      if (lane < 4) s_colocc[lane] = 0;
      __syncthreads();
      for (int i = lane; i < nseg; i += 64) {
        const unsigned sg = s_seg[i];
        const int y = (int)(sg >> 16), c = (int)(sg & 0xFFFF);
        if (y >= ylo && y <= yhi) atomicOr(&s_colocc[c >> 6], 1ull << (c & 63));
      }
      __syncthreads();

      // islands: runs of set bits in the occupancy dilated by dc
      for (int cstart = 0; cstart < spr;) {
        // find next occupied column >= cstart
        int cfirst = -1;
        for (int wi = cstart >> 6; wi < 4 && wi * 64 < spr; ++wi) {
          u64 w = s_colocc[wi];
          if (wi == (cstart >> 6)) w &= ~0ull << (cstart & 63);
          if (w) {
            cfirst = wi * 64 + __builtin_ctzll(w);
            break;
          }
        }
        if (cfirst < 0) break;
        // extend while the gap to the next occupied column is <= 2*dc (dilated runs touch)
        int clast = cfirst;
        for (;;) {
          int nxt = -1;
          for (int c = clast + 1; c <= min(spr - 1, clast + 2 * dc + 1); ++c)
            if ((s_colocc[c >> 6] >> (c & 63)) & 1) {
              nxt = c;
              break;
            }
          if (nxt < 0) break;
          clast = nxt;
        }
        cstart = clast + 2 * dc + 2;
        const int clo = max(0, cfirst - dc), chi = min(spr - 1, clast + dc);  // output segment columns

        // ---- island window
        const int H = yhi - ylo + 1, S = H + 2;
        const int xw0 = 16 * clo;
        const int xhi = min(g.cols - 1, 16 * chi + 15);
        const int W = ((xhi - xw0 + 1) + 2 + 63) / 64 + 1;
        const int pwc0 = clo - 1, nps = chi - clo + 3, PW = 16 * nps;
        if (S * W > K1B_BM_CAP || H * PW > K1B_PIX_CAP) {
          if (lane == 0) s_over = 1;
          continue;
        }
        for (int i = lane; i < S * W; i += 64) {
          s_nz[i] = 0;
          s_pm[i] = 0;
          s_ng[i] = 0;
        }
        // stage the thresholded pixels (coalesced 16-byte loads)
        for (int i = lane; i < H * nps; i += 64) {
          const int yb = i / nps, sc = i - yb * nps;
          const int cg = pwc0 + sc;
          uint4 v = make_uint4(0, 0, 0, 0);
          if (cg >= 0 && cg < spr) {
            v = *reinterpret_cast<const uint4*>(frame + (size_t)(ylo + yb) * g.pitch + 16 * cg);
            v.x = tozero4(v.x, add);
            v.y = tozero4(v.y, add);
            v.z = tozero4(v.z, add);
            v.w = tozero4(v.w, add);
          }
          *reinterpret_cast<uint4*>(s_pix + (size_t)yb * PW + 16 * sc) = v;
        }
        __syncthreads();
        // blurred mask
        const PixWin pw = {s_pix, ylo, H, pwc0, PW};
        const int ncols = chi - clo + 1;
        for (int i = lane; i < H * ncols; i += 64) {
          const int yb = i / ncols, c = clo + (i - yb * ncols);
          const int x0 = 16 * c;
          if (x0 >= g.cols) continue;
          unsigned m;
          const bool interior = (x0 - r >= 0) && (x0 + 15 + r < g.cols);
          if (interior && dp.ksize == 5)
            m = blur_item_fast<5>(pw, g.rows, g.cols, ylo + yb, c, dp.taps);
          else if (interior && dp.ksize == 3)
            m = blur_item_fast<3>(pw, g.rows, g.cols, ylo + yb, c, dp.taps);
          else
            m = blur_item_generic(pw, g.rows, g.cols, ylo + yb, c, dp.taps, dp.ksize);
          if (m) {
            const int xb0 = x0 - xw0 + 1, wi = xb0 >> 6, shb = xb0 & 63;
            atomicOr(&s_nz[(size_t)(yb + 1) * W + wi], (u64)m << shb);
            if (shb > 48) atomicOr(&s_nz[(size_t)(yb + 1) * W + wi + 1], (u64)m >> (64 - shb));
          }
        }
        __syncthreads();

        // ---- OpenCV's raster scan for external contours (cvFindNextContour, RETR_EXTERNAL);
        //      sequential by construction: lane 0 walks the island's bitmap rows in raster order
        if (lane == 0) {
          int nk = s_nkept;
          for (int slot = 1; slot <= H; ++slot) {
            u64* nzrow = s_nz + (size_t)slot * W;
            u64* pmrow = s_pm + (size_t)slot * W;
            u64* ngrow = s_ng + (size_t)slot * W;
            int last_sign = 0;  // sign of the nearest marked pixel to the left (lnbd), 0 = none yet
            for (int w = 0; w < W - 1; ++w) {
              const u64 nzw = nzrow[w];
              if (!nzw) continue;
              const u64 leftnz = (nzw << 1) | (w ? (nzrow[w - 1] >> 63) : 0);
              u64 done = 0;
              for (;;) {
                const u64 pw_ = pmrow[w], gw = ngrow[w];
                const u64 cand = nzw & ~(pw_ | gw) & ~leftnz & ~done;  // unmarked 1 with a 0 on its left
                if (!cand) break;
                const int bb = __builtin_ctzll(cand);
                done |= (bb == 63) ? ~0ull : ((2ull << bb) - 1);
                const u64 below = (pw_ | gw) & ((1ull << bb) - 1);
                int sign = last_sign;
                if (below) {
                  const int hb = 63 - __builtin_clzll(below);
                  sign = ((gw >> hb) & 1) ? -1 : 1;
                }
                if (sign > 0) continue;  // inside an already traced outer border: not external
                PolyAcc acc;
                const int xb = w * 64 + bb;
                if (!trace_outer_border(s_nz, s_pm, s_ng, W, slot, xb, xw0 - 1, ylo - 1, acc)) s_over = 1;
                BlobRec br;
                br.a00 = acc.a00;
                br.a10 = acc.a10;
                br.a01 = acc.a01;
                br.xmin = acc.xmin;
                br.xmax = acc.xmax;
                br.ymin = acc.ymin;
                br.ymax = acc.ymax;
                float mcx, mcy;
                if (blob_filter(br, dp, mcx, mcy)) {
                  if (nk < K1B_KEPT_CAP) {
                    s_kx[nk] = mcx;
                    s_ky[nk] = mcy;
                    s_kkey[nk] = ((unsigned)(ylo + slot - 1) << 12) | (unsigned)(xb + xw0 - 1);
                  }
                  ++nk;
                }
              }
              const u64 mk = pmrow[w] | ngrow[w];
              if (mk) {
                const int hb = 63 - __builtin_clzll(mk);
                last_sign = ((ngrow[w] >> hb) & 1) ? -1 : 1;
              }
            }
          }
          s_nkept = nk;
        }
        __syncthreads();
      }  // islands
      if (crosses) {
        // the band continued into later words: drop its rows from them so they are not revisited
        for (int wi = rwi + 1; wi <= (yhi >> 6) && wi < g.rw; ++wi) {
          const int last = min(yhi, wi * 64 + 63) - wi * 64;
          const u64 m = last == 63 ? ~0ull : ((2ull << last) - 1);
          if (lane == 0) s_rowact[wi] &= ~m;
        }
        __syncthreads();
      }
    }
  }

  // ---- C: output in OpenCV's contour order (newest first = descending raster order of the
  //         start pixel), float32 centroid -> undistortPoints
  const int nk_all = s_nkept;
  const int nk = min(nk_all, K1B_KEPT_CAP);
  if (lane < nk) {
    const unsigned key = s_kkey[lane];
    int pos = 0;
    for (int j = 0; j < nk; ++j) pos += (s_kkey[j] > key) ? 1 : 0;
    if (pos < MPE_MAX_DETECTIONS) {
      const float mcx = s_kx[lane], mcy = s_ky[lane];
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
    if (s_over) st = MPE_FRAME_TOO_MANY_ROWS;
    out->status = st;
  }
}

size_t k1b_lds_bytes(const FrameGeom&) { return 0; }

hipError_t launch_k1b_blobs(const uint8_t* frames, const unsigned long long* flags, int n_frames, const FrameGeom& g,
                            const DetectParams& dp, mpe_detections* dets, hipStream_t s) {
  if (n_frames <= 0) return hipSuccess;
  hipLaunchKernelGGL(k1b_blobs, dim3(n_frames), dim3(64), 0, s, frames, (const u64*)flags, g, dp, dets);
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
__global__ __launch_bounds__(K2_THREADS) void k2_vote(const mpe_detections* __restrict__ dets, SolveParams sp,
                                                      uint32_t* __restrict__ hist, int splits) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ double s_px[MPE_MAX_DETECTIONS][2];
  __shared__ double s_iv[MPE_MAX_DETECTIONS][3];
  __shared__ double s_mk[MPE_MAX_MARKERS][3];
  __shared__ unsigned s_hist[MPE_HIST_STRIDE];

  const int f = blockIdx.x / splits, part = blockIdx.x - f * splits;
  const int tid = threadIdx.x;
  const mpe_detections* d = dets + f;
  const int n_d = d->n, n_m = sp.n_markers;
  if (n_d < 4 || d->status != 0 || n_m < 4) return;  // min_num_leds_detected_ (pose_estimator.h:78)

  const int nthr = blockDim.x;
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
  if (tid < n_m) {
    s_mk[tid][0] = sp.markers[3 * tid];
    s_mk[tid][1] = sp.markers[3 * tid + 1];
    s_mk[tid][2] = sp.markers[3 * tid + 2];
  }
  __syncthreads();

  double* s_q = reinterpret_cast<double*>(smem);  // back-projections: [2*j + {0,1}][tid]
  const int n_combos = n_d * (n_d - 1) * (n_d - 2) / 6;
  const int n_mcombos = n_m * (n_m - 1) * (n_m - 2) / 6;
  const int n_perms = n_mcombos * 6;
  const long long total = (long long)n_combos * n_perms;
  const int nuo = n_m - 3;

  for (long long t = (long long)part * nthr + tid; t < total; t += (long long)splits * nthr) {
    const int ci = (int)(t / n_perms), pj = (int)(t - (long long)ci * n_perms);
    int c0, c1, c2;
    unrank_combo3(ci, n_d, c0, c1, c2);
    int ma, mb, mc;
    unrank_combo3(pj / 6, n_m, ma, mb, mc);
    // combinations.cpp:131-244: block rows [c b a],[c a b],[b c a],[b a c],[a b c],[a c b]
    int p0, p1, p2;
    switch (pj % 6) {
      case 0: p0 = mc; p1 = mb; p2 = ma; break;
      case 1: p0 = mc; p1 = ma; p2 = mb; break;
      case 2: p0 = mb; p1 = mc; p2 = ma; break;
      case 3: p0 = mb; p1 = ma; p2 = mc; break;
      case 4: p0 = ma; p1 = mb; p2 = mc; break;
      default: p0 = ma; p1 = mc; p2 = mb; break;
    }
    const V3 fa = {s_iv[c0][0], s_iv[c0][1], s_iv[c0][2]}, fb = {s_iv[c1][0], s_iv[c1][1], s_iv[c1][2]},
             fc = {s_iv[c2][0], s_iv[c2][1], s_iv[c2][2]};
    const V3 wa = {s_mk[p0][0], s_mk[p0][1], s_mk[p0][2]}, wb = {s_mk[p1][0], s_mk[p1][1], s_mk[p1][2]},
             wc = {s_mk[p2][0], s_mk[p2][1], s_mk[p2][2]};
    P3PCtx ctx;
    if (!p3p_prepare(fa, fb, fc, wa, wb, wc, ctx)) continue;
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
      M3 R;
      V3 C;
      p3p_solution(ctx, pick_root(ctx, k), R, C);
      if (!rc_finite(R, C)) continue;
      const Proj P = make_projection(R, C, sp.fx, sp.fy, sp.cx, sp.cy);
      // back-project the unused markers (ascending marker index)
      int j = 0;
      for (int m = 0; m < n_m; ++m) {
        if (m == p0 || m == p1 || m == p2) continue;
        double u, v;
        project(P, V3{s_mk[m][0], s_mk[m][1], s_mk[m][2]}, u, v);
        s_q[(2 * j) * nthr + tid] = u;
        s_q[(2 * j + 1) * nthr + tid] = v;
        ++j;
      }
      // nearest back-projection for every unused detection (pose_estimator.cpp:862-906)
      bool any = false;
      for (int a = 0; a < n_d; ++a) {
        if (a == c0 || a == c1 || a == c2) continue;
        const double au = s_px[a][0], av = s_px[a][1];
        double best = INFINITY;
        int bj = 0;
        for (int jj = 0; jj < nuo; ++jj) {
          const double du = au - s_q[(2 * jj) * nthr + tid], dv = av - s_q[(2 * jj + 1) * nthr + tid];
          const double d2 = du * du + dv * dv;
          if (d2 < best) {
            best = d2;
            bj = jj;
          }
        }
        if (sqrt(best) < sp.back_tol) {  // strict <, pose_estimator.cpp:689
          // bj-th unused marker -> marker index
          int mi = -1, cnt = -1;
          for (int m = 0; m < n_m; ++m) {
            if (m == p0 || m == p1 || m == p2) continue;
            if (++cnt == bj) {
              mi = m;
              break;
            }
          }
          atomicAdd(&s_hist[a * MPE_MAX_MARKERS + mi], 1u);
          any = true;
        }
      }
      if (any) {  // pose_estimator.cpp:676-685
        atomicAdd(&s_hist[c0 * MPE_MAX_MARKERS + p0], 1u);
        atomicAdd(&s_hist[c1 * MPE_MAX_MARKERS + p1], 1u);
        atomicAdd(&s_hist[c2 * MPE_MAX_MARKERS + p2], 1u);
      }
    }
  }
  __syncthreads();
  uint32_t* gh = hist + (size_t)f * MPE_HIST_STRIDE;
  for (int i = tid; i < MPE_HIST_STRIDE; i += nthr) {
    const unsigned v = s_hist[i];
    if (v) atomicAdd(&gh[i], v);
  }
}

hipError_t launch_k2_vote(const mpe_detections* dets, int n_frames, const SolveParams& sp, uint32_t* hist, int splits,
                          int n_det_hint, hipStream_t s) {
  if (n_frames <= 0) return hipSuccess;
  if (splits < 1) splits = 1;
  const int nuo = sp.n_markers > 3 ? sp.n_markers - 3 : 1;
  // block size: the multiple of 64 (<= 256) that wastes the fewest lanes on the expected item count
  int threads = K2_THREADS;
  if (n_det_hint >= 4) {
    const long long nm = sp.n_markers;
    const long long items = (long long)n_det_hint * (n_det_hint - 1) * (n_det_hint - 2) / 6 * nm * (nm - 1) * (nm - 2);
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
  const size_t lds = (size_t)nuo * 2 * threads * sizeof(double);
  hipLaunchKernelGGL(k2_vote, dim3((unsigned)(n_frames * splits)), dim3(threads), lds, s, dets, sp, hist, splits);
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

// orthogonal polar factor of the 3x3 matrix X (scaled Newton iteration); for H = U S V^T this is
// V U^T when X = H^T — the rotation Eigen's JacobiSVD route produces at pose_estimator.cpp:916-922
// (no reflection guard: a negative determinant is kept, as in the reference).
__device__ void polar3(double X[3][3]) {
  for (int it = 0; it < 60; ++it) {
    // inverse transpose via cofactors
    double c00 = X[1][1] * X[2][2] - X[1][2] * X[2][1];
    double c01 = X[1][2] * X[2][0] - X[1][0] * X[2][2];
    double c02 = X[1][0] * X[2][1] - X[1][1] * X[2][0];
    double c10 = X[0][2] * X[2][1] - X[0][1] * X[2][2];
    double c11 = X[0][0] * X[2][2] - X[0][2] * X[2][0];
    double c12 = X[0][1] * X[2][0] - X[0][0] * X[2][1];
    double c20 = X[0][1] * X[1][2] - X[0][2] * X[1][1];
    double c21 = X[0][2] * X[1][0] - X[0][0] * X[1][2];
    double c22 = X[0][0] * X[1][1] - X[0][1] * X[1][0];
    double det = X[0][0] * c00 + X[0][1] * c01 + X[0][2] * c02;
    double id = 1.0 / det;
    double Y[3][3] = {{c00 * id, c01 * id, c02 * id}, {c10 * id, c11 * id, c12 * id}, {c20 * id, c21 * id, c22 * id}};
    double nx = 0, ny = 0;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        nx += X[i][j] * X[i][j];
        ny += Y[i][j] * Y[i][j];
      }
    double gam = sqrt(sqrt(ny / nx));
    double diff = 0;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double xn = 0.5 * (gam * X[i][j] + Y[i][j] / gam);
        diff += (xn - X[i][j]) * (xn - X[i][j]);
        X[i][j] = xn;
      }
    if (!(diff > 1e-30)) break;  // also leaves on NaN
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

#define K3_GROUP 16                 // lanes cooperating on one frame
#define K3_FRAMES_PER_BLOCK 4       // one wave = 4 frames
#define K3_NU_MAX (MPE_MAX_MARKERS - 3)

// butterfly sum over the 16 lanes of a group; every lane ends with the same total
__device__ __forceinline__ double group_sum(double v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}

__global__ __launch_bounds__(64) void k3_tail(const mpe_detections* __restrict__ dets,
                                              const uint32_t* __restrict__ hist, int n_frames, SolveParams sp,
                                              mpe_result* __restrict__ results, uint32_t* __restrict__ corr_out) {
  __shared__ double s_part[K3_FRAMES_PER_BLOCK][K3_GROUP][MPE_MAX_MARKERS * 3];
  __shared__ double s_mean[K3_FRAMES_PER_BLOCK][MPE_MAX_MARKERS * 3];
  __shared__ double s_q[2 * K3_NU_MAX][64];  // back-projections, [2*j + {0,1}][lane]
  __shared__ double s_det[K3_FRAMES_PER_BLOCK][MPE_MAX_DETECTIONS][2];
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
  if (live) {
    for (int i = l; i < 16; i += K3_GROUP) res->T[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int i = l; i < 36; i += K3_GROUP) res->cov[i] = 0.0;
    if (corr_out)
      for (int i = l; i < 2 * MPE_MAX_MARKERS; i += K3_GROUP) corr_out[(size_t)f * 2 * MPE_MAX_MARKERS + i] = 0;
  }
  for (int i = 0; i < 3 * MPE_MAX_MARKERS; ++i) s_part[grp][l][i] = 0.0;

  // ---- lane 0 of the group: initialise()'s all-zero test (pose_estimator.cpp:704) and
  //      correspondencesFromHistogram (pose_estimator.cpp:344-370)
  if (l == 0) {
    int n_c = 0;
    bool go0 = live && dstatus == 0 && n_d >= 4 && n_m >= 4;
    if (go0) {
      bool any = false;
      for (int r = 0; r < n_d; ++r)
        for (int c = 0; c < n_m; ++c) any |= (H[r * MPE_MAX_MARKERS + c] != 0);
      go0 = any;
    }
    if (go0) {
      unsigned removed = 0;  // zeroed columns
      for (int j = 0; j < n_m; ++j) {
        unsigned mv = 0;
        int ri = 0, ci = 0;
        bool first = true;
        for (int c = 0; c < n_m; ++c)
          for (int r = 0; r < n_d; ++r) {
            const unsigned v = ((removed >> c) & 1) ? 0u : H[r * MPE_MAX_MARKERS + c];
            if (first || v > mv) {
              mv = v;
              ri = r;
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
  const bool go = n_c >= 4;

  // ---- checkCorrespondences (pose_estimator.cpp:394-542): the C(n_c,3) P3P validations are
  //      spread over the 16 lanes; each lane sums inverse(H_best) * markers for its combinations
  unsigned my_valid = 0;
  const int nu = n_c - 3;
  const int N = go ? n_c * (n_c - 1) * (n_c - 2) / 6 : 0;
  for (int ci = l; ci < N; ci += K3_GROUP) {
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
    if (!p3p_prepare(fv[0], fv[1], fv[2], wp[0], wp[1], wp[2], ctx)) continue;
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
        s_q[2 * q][tid] = u;
        s_q[2 * q + 1][tid] = v;
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
          const double bu = s_q[2 * cj][tid], bv = s_q[2 * cj + 1][tid];
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
    if (!found) continue;
    ++my_valid;
    if (best < 0) best = 0;  // unreachable: sq is always finite
    M3 R;
    V3 C;
    p3p_solution(ctx, pick_root(ctx, best), R, C);
    // inverse(H) * marker for ALL markers (pose_estimator.cpp:513-517)
    for (int jj = 0; jj < n_m; ++jj) {
      const V3 mk = {s_mk[jj][0] - C.x, s_mk[jj][1] - C.y, s_mk[jj][2] - C.z};
      const V3 pc = mulT(R, mk);  // R^T (m - C)
      s_part[grp][l][3 * jj] += pc.x;
      s_part[grp][l][3 * jj + 1] += pc.y;
      s_part[grp][l][3 * jj + 2] += pc.z;
    }
  }
  s_valid[grp][l] = my_valid;
  __syncthreads();
  // ordered reduction over the lanes (= combination order while C(n_c,3) <= 16)
  for (int v = l; v < 3 * MPE_MAX_MARKERS; v += K3_GROUP) {
    double sacc = 0.0;
    for (int q = 0; q < K3_GROUP; ++q) sacc += s_part[grp][q][v];
    s_mean[grp][v] = sacc;
  }
  unsigned num_valid = 0;
  for (int q = 0; q < K3_GROUP; ++q) num_valid += s_valid[grp][q];
  __syncthreads();
  bool active = go && ((double)num_valid / (double)N >= sp.valid_corr_thr);

  // ---- computeTransformation (pose_estimator.cpp:908-930); evaluated by every lane of the group
  T34 T;
  if (active) {
    double mo[3] = {0, 0, 0}, mr[3] = {0, 0, 0};
    for (int i = 0; i < n_m; ++i)
      for (int k = 0; k < 3; ++k) {
        mo[k] += s_mk[i][k];
        mr[k] += s_mean[grp][3 * i + k] / (double)num_valid;
      }
    for (int k = 0; k < 3; ++k) {
      mo[k] /= (double)n_m;
      mr[k] /= (double)n_m;
    }
    double Hm[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int i = 0; i < n_m; ++i)
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
          Hm[r][c] += (s_mk[i][r] - mo[r]) * (s_mean[grp][3 * i + c] / (double)num_valid - mr[c]);
    double X[3][3];  // H^T
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) X[r][c] = Hm[c][r];
    polar3(X);  // R = V U^T
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) T.m[r][c] = X[r][c];
      T.m[r][3] = mr[r] - (X[r][0] * mo[0] + X[r][1] * mo[1] + X[r][2] * mo[2]);
    }
  }

  // ---- optimisePose (pose_estimator.cpp:733-792): Gauss-Newton on SE(3).  Lane j of the group
  //      evaluates correspondence j (J^T J and J^T e, upper triangle), a butterfly sum gives every
  //      lane the normal equations, and all lanes take the same LDL^T / exp-map step.
  double mkx = 0, mky = 0, mkz = 0, du_ = 0, dv_ = 0;
  bool has = false;
  if (active && l < n_c) {  // n_c <= MPE_MAX_MARKERS = K3_GROUP
    const int mi = s_cm[grp][l] - 1, di = s_cd[grp][l] - 1;
    mkx = s_mk[mi][0];
    mky = s_mk[mi][1];
    mkz = s_mk[mi][2];
    du_ = s_det[grp][di][0];
    dv_ = s_det[grp][di][1];
    has = true;
  }
  double A[6][6];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) A[r][c] = 0;
  int iters = 0;
  bool running = active;
  for (int it = 0; it < 500; ++it) {
    if (!__any(running)) break;
    double J0[6] = {0, 0, 0, 0, 0, 0}, J1[6] = {0, 0, 0, 0, 0, 0}, e0 = 0, e1 = 0;
    if (running && has) {
      const double mk[3] = {mkx, mky, mkz};
      double u, v, x, y, z;
      project_T(T, mk, fx, fy, cx, cy, u, v, x, y, z);
      e0 = du_ - u;
      e1 = dv_ - v;
      const double z_2 = z * z;
      // computeJacobian, pose_estimator.cpp:945-957
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
    }
    double An[6][6], b[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
#pragma unroll
      for (int c = r; c < 6; ++c) {
        An[r][c] = group_sum(J0[r] * J0[c] + J1[r] * J1[c]);
        An[c][r] = An[r][c];
      }
      b[r] = group_sum(J0[r] * e0 + J1[r] * e1);
    }
    if (running) {
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) A[r][c] = An[r][c];
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
      if (mx <= 1e-13) running = false;
    }
  }
  if (!active) return;
  // pose_covariance_ = A.inverse() with the A of the last iteration (pose_estimator.cpp:790);
  // lane c of the group solves for column c
  {
    LDL6 F;
    ldl6_factor(A, F);
    if (l < 6) {
      double e[6], x[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) e[r] = (r == l) ? 1.0 : 0.0;
      ldl6_solve(F, e, x);
#pragma unroll
      for (int r = 0; r < 6; ++r) res->cov[r * 6 + l] = x[r];
    }
  }
  if (l == 0) {
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) res->T[r * 4 + c] = T.m[r][c];
    res->gn_iterations = iters;
    res->status = MPE_FRAME_POSE;
  }
}

hipError_t launch_k3_tail(const mpe_detections* dets, const uint32_t* hist, int n_frames, const SolveParams& sp,
                          mpe_result* results, uint32_t* corr_out, hipStream_t s) {
  if (n_frames <= 0) return hipSuccess;
  hipLaunchKernelGGL(k3_tail, dim3((n_frames + K3_FRAMES_PER_BLOCK - 1) / K3_FRAMES_PER_BLOCK), dim3(64), 0, s, dets,
                     hist, n_frames, sp, results, corr_out);
  return hipGetLastError();
}

}  // namespace mpe
