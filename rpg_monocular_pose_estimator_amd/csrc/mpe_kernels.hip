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
// K1b — blob extraction, one wave per frame
// =============================================================================================
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

template <class F>
__device__ __forceinline__ void for_each_bright_seg(const u64* __restrict__ flags, size_t G0, int spf, int lane, F f) {
  const int nwin = (spf + 63) >> 6;
  const size_t w0 = G0 >> 6;
  const int sh = (int)(G0 & 63);
  for (int i = lane; i < nwin; i += 64) {
    u64 a = flags[w0 + i], b = flags[w0 + i + 1];
    u64 v = sh ? ((a >> sh) | (b << (64 - sh))) : a;
    const int rem = spf - i * 64;
    if (rem < 64) v &= (1ull << rem) - 1;
    while (v) {
      const int bit = __builtin_ctzll(v);
      v &= v - 1;
      f(i * 64 + bit);
    }
  }
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

struct RowMap {
  const u64* rowact;
  const u64* rowstart;
  const unsigned* rowpre;
  const unsigned* bandpre;
  __device__ __forceinline__ bool active(int y) const { return (rowact[y >> 6] >> (y & 63)) & 1; }
  // slot = (#active rows < y) + (#band starts <= y); slot 0 and the slot after every band are
  // all-zero separator rows, so border following can step +-1 slot without bounds checks.
  __device__ __forceinline__ int slot(int y) const {
    const int w = y >> 6, b = y & 63;
    return (int)(rowpre[w] + __builtin_popcountll(rowact[w] & ((1ull << b) - 1)) + bandpre[w] +
                 __builtin_popcountll(rowstart[w] & ((2ull << b) - 1)));
  }
};

// Fixed-point Gaussian of the thresholded image for 16 output pixels (x0..x0+15 of row y);
// returns the 16-bit mask of non-zero results: (sum + 2^15) >> 16 != 0  <=>  sum >= 2^15.
template <int KS>
__device__ __forceinline__ unsigned blur_item(const uint8_t* __restrict__ frame, int rows, int cols, int pitch, int y,
                                              int x0, int thr, const int* __restrict__ taps) {
  constexpr int R = KS / 2;
  int acc[16];
#pragma unroll
  for (int x = 0; x < 16; ++x) acc[x] = 0;
  const bool interior = (x0 - R >= 0) && (x0 + 15 + R < cols);
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    const int yy = reflect101(y + i - R, rows);
    const uint8_t* rowp = frame + (size_t)yy * pitch;
    int t[16 + 2 * R];
    if (interior) {
#pragma unroll
      for (int j = 0; j < 16 + 2 * R; ++j) {
        const int v = rowp[x0 - R + j];
        t[j] = v > thr ? v : 0;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16 + 2 * R; ++j) {
        const int v = rowp[reflect101(x0 - R + j, cols)];
        t[j] = v > thr ? v : 0;
      }
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
    if (acc[x] >= (1 << 15) && x0 + x < cols) m |= 1u << x;
  return m;
}

// any kernel size (sigma up to 6): direct double loop per output pixel
__device__ __noinline__ unsigned blur_item_generic(const uint8_t* __restrict__ frame, int rows, int cols, int pitch,
                                                   int y, int x0, int thr, const int* __restrict__ taps, int ks) {
  const int r = ks / 2;
  unsigned m = 0;
  for (int x = 0; x < 16; ++x) {
    if (x0 + x >= cols) break;
    int acc = 0;
    for (int i = 0; i < ks; ++i) {
      const int yy = reflect101(y + i - r, rows);
      const uint8_t* rowp = frame + (size_t)yy * pitch;
      int h = 0;
      for (int j = 0; j < ks; ++j) {
        const int v = rowp[reflect101(x0 + x + j - r, cols)];
        h += taps[j] * (v > thr ? v : 0);
      }
      acc += taps[i] * h;
    }
    if (acc >= (1 << 15)) m |= 1u << x;
  }
  return m;
}

// three bits (x-1, x, x+1) of a bitmap row at bit index xb (= x + 1 >= 1)
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
// "negative" (ng, takes precedence).  Returns false if the step bound was hit.
__device__ bool trace_outer_border(const u64* nz, u64* pm, u64* ng, int wb, int slot0, int xb0, int y0, PolyAcc& acc) {
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
    acc.emit(xb0 - 1, y0);
    acc.close();
    return true;
  }
  const int x1b = xb0 + dir_dx(s), y1 = y0 + dir_dy(s);
  int xb = xb0, y = y0, slot = slot0;
  for (int step = 0; step < (1 << 22); ++step) {
    const int s_end = s;
    const unsigned m16 = nb | (nb << 8);
    const int k = __builtin_ctz(m16 >> (s + 1));
    const int sn = (s + 1 + k) & 7;
    if ((unsigned)(sn - 1) < (unsigned)s_end)
      set_bit(ng, wb, slot, xb);
    else
      set_bit(pm, wb, slot, xb);
    acc.emit(xb - 1, y);
    const int nxb = xb + dir_dx(sn), ny = y + dir_dy(sn);
    if (nxb == xb0 && ny == y0 && xb == x1b && y == y1) {
      acc.close();
      return true;
    }
    slot += dir_dy(sn);
    xb = nxb;
    y = ny;
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
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ BlobRec s_blobs[MPE_MAX_RAW_BLOBS];
  __shared__ int s_nblobs, s_err;

  const int lane = threadIdx.x;
  const int f = blockIdx.x;
  const uint8_t* frame = frames + (size_t)f * g.rows * g.pitch;
  mpe_detections* out = dets + f;

  // LDS carve-up
  const size_t bm_words = (size_t)g.slot_cap * g.wb;
  u64* nz = reinterpret_cast<u64*>(smem);
  u64* pm = nz + bm_words;
  u64* ng = pm + bm_words;
  u64* todo = ng + bm_words;
  u64* rowact = todo + (size_t)g.slot_cap * g.tw;
  u64* rowstart = rowact + g.rw;
  u64* wordmask = rowstart + g.rw;
  const int wm_words = (int)((bm_words + 63) / 64) + 1;
  unsigned* rowpre = reinterpret_cast<unsigned*>(wordmask + wm_words);
  unsigned* bandpre = rowpre + g.rw;

  const int r = dp.ksize / 2;
  const int dc = (r + 15) / 16;  // segment columns a bright segment can influence on each side
  const size_t G0 = (size_t)f * g.segs_per_frame;

  for (int i = lane; i < g.rw; i += 64) rowact[i] = 0;
  if (lane == 0) {
    s_nblobs = 0;
    s_err = 0;
  }
  __syncthreads();

  // ---- A: rows within +-r of a bright segment become active
  for_each_bright_seg(flags, G0, g.segs_per_frame, lane, [&](int s) {
    const int y0 = s / g.segs_per_row;
    lds_set_range(rowact, max(0, y0 - r), min(g.rows - 1, y0 + r));
  });
  __syncthreads();

  // ---- B: rank / band prefix sums (rw <= 64 words: one per lane)
  int n_active, n_bands;
  {
    const u64 act = (lane < g.rw) ? rowact[lane] : 0;
    const u64 prevw = (lane > 0 && lane < g.rw) ? rowact[lane - 1] : 0;
    const u64 st = act & ~((act << 1) | (prevw >> 63));
    const int ca = __builtin_popcountll(act), cs = __builtin_popcountll(st);
    int pa = ca, ps = cs;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int ta = __shfl_up(pa, d), ts = __shfl_up(ps, d);
      if (lane >= d) {
        pa += ta;
        ps += ts;
      }
    }
    if (lane < g.rw) {
      rowpre[lane] = (unsigned)(pa - ca);
      bandpre[lane] = (unsigned)(ps - cs);
      rowstart[lane] = st;
    }
    n_active = __shfl(pa, 63);
    n_bands = __shfl(ps, 63);
  }
  const int nslots = n_active + n_bands + 1;  // separator rows included
  if (n_active == 0 || nslots > g.slot_cap) {
    if (lane == 0) {
      out->n = 0;
      out->status = (n_active == 0) ? 0 : MPE_FRAME_TOO_MANY_ROWS;
    }
    return;
  }
  __syncthreads();
  const RowMap rm = {rowact, rowstart, rowpre, bandpre};

  // ---- C: clear the bitmaps of the slots in use
  for (int i = lane; i < nslots * g.wb; i += 64) {
    nz[i] = 0;
    pm[i] = 0;
    ng[i] = 0;
  }
  for (int i = lane; i < nslots * g.tw; i += 64) todo[i] = 0;
  __syncthreads();

  // ---- D: output segments that can be non-zero after the blur
  for_each_bright_seg(flags, G0, g.segs_per_frame, lane, [&](int s) {
    const int y0 = s / g.segs_per_row, c0 = s - y0 * g.segs_per_row;
    const int ylo = max(0, y0 - r), yhi = min(g.rows - 1, y0 + r);
    const int clo = max(0, c0 - dc), chi = min(g.segs_per_row - 1, c0 + dc);
    const int slo = rm.slot(ylo);  // rows ylo..yhi are all active and contiguous -> consecutive slots
    for (int yy = ylo; yy <= yhi; ++yy)
      for (int cc = clo; cc <= chi; ++cc) atomicOr(&todo[(size_t)(slo + yy - ylo) * g.tw + (cc >> 6)], 1ull << (cc & 63));
  });
  __syncthreads();

  // ---- E/F: blur mask of the todo segments (lane owns rows y = lane mod 64)
  for (int y = lane; y < g.rows; y += 64) {
    if (!rm.active(y)) continue;
    const int sl = rm.slot(y);
    for (int tw = 0; tw < g.tw; ++tw) {
      u64 tb = todo[(size_t)sl * g.tw + tw];
      while (tb) {
        const int c = tw * 64 + __builtin_ctzll(tb);
        tb &= tb - 1;
        const int x0 = c * 16;
        if (x0 >= g.cols) continue;
        unsigned m;
        if (dp.ksize == 5)
          m = blur_item<5>(frame, g.rows, g.cols, g.pitch, y, x0, dp.thr, dp.taps);
        else if (dp.ksize == 3)
          m = blur_item<3>(frame, g.rows, g.cols, g.pitch, y, x0, dp.thr, dp.taps);
        else
          m = blur_item_generic(frame, g.rows, g.cols, g.pitch, y, x0, dp.thr, dp.taps, dp.ksize);
        if (m) {
          const int xb0 = x0 + 1, wi = xb0 >> 6, sh = xb0 & 63;
          atomicOr(&nz[(size_t)sl * g.wb + wi], (u64)m << sh);
          if (sh > 48) atomicOr(&nz[(size_t)sl * g.wb + wi + 1], (u64)m >> (64 - sh));
        }
      }
    }
  }
  __syncthreads();

  // ---- G: which bitmap words are non-zero (raster order = index order)
  {
    const int total = nslots * g.wb;
    for (int k = 0; k * 64 < total + 64; ++k) {
      const int idx = k * 64 + lane;
      const u64 b = __ballot(idx < total && nz[idx] != 0);
      if (lane == 0) wordmask[k] = b;
    }
  }
  __syncthreads();

  // ---- H: OpenCV's raster scan for external contours (cvFindNextContour, mode RETR_EXTERNAL),
  //         sequential by construction: lane 0 walks the non-zero words in raster order.
  if (lane == 0) {
    int nb = 0, err = 0;
    int slot = -1;
    for (int rwi = 0; rwi < g.rw; ++rwi) {
      u64 act = rowact[rwi];
      const u64 st = rowstart[rwi];
      while (act) {
        const int b = __builtin_ctzll(act);
        act &= act - 1;
        const int y = rwi * 64 + b;
        slot += ((st >> b) & 1) ? 2 : 1;
        // non-zero words of this bitmap row
        const size_t bitpos = (size_t)slot * g.wb;
        const int wi = (int)(bitpos >> 6), sh = (int)(bitpos & 63);
        u64 wm = sh ? ((wordmask[wi] >> sh) | (wordmask[wi + 1] << (64 - sh))) : wordmask[wi];
        if (g.wb < 64) wm &= (1ull << g.wb) - 1;
        int last_sign = 0;  // sign of the nearest marked pixel to the left (lnbd), 0 = none yet
        u64* nzrow = nz + (size_t)slot * g.wb;
        u64* pmrow = pm + (size_t)slot * g.wb;
        u64* ngrow = ng + (size_t)slot * g.wb;
        while (wm) {
          const int w = __builtin_ctzll(wm);
          wm &= wm - 1;
          const u64 nzw = nzrow[w];
          const u64 leftnz = (nzw << 1) | (w ? (nzrow[w - 1] >> 63) : 0);
          u64 done = 0;
          for (;;) {
            const u64 pw = pmrow[w], gw = ngrow[w];
            const u64 cand = nzw & ~(pw | gw) & ~leftnz & ~done;  // unmarked 1 with a 0 to its left
            if (!cand) break;
            const int bb = __builtin_ctzll(cand);
            done |= (bb == 63) ? ~0ull : ((2ull << bb) - 1);
            const u64 below = (pw | gw) & ((1ull << bb) - 1);
            int sign = last_sign;
            if (below) {
              const int hb = 63 - __builtin_clzll(below);
              sign = ((gw >> hb) & 1) ? -1 : 1;
            }
            if (sign > 0) continue;  // inside an already traced outer border: not external
            PolyAcc acc;
            if (!trace_outer_border(nz, pm, ng, g.wb, slot, w * 64 + bb, y, acc)) err = 1;
            if (nb < MPE_MAX_RAW_BLOBS) {
              BlobRec br;
              br.a00 = acc.a00;
              br.a10 = acc.a10;
              br.a01 = acc.a01;
              br.xmin = acc.xmin;
              br.xmax = acc.xmax;
              br.ymin = acc.ymin;
              br.ymax = acc.ymax;
              s_blobs[nb] = br;
            }
            ++nb;
          }
          const u64 mk = pmrow[w] | ngrow[w];
          if (mk) {
            const int hb = 63 - __builtin_clzll(mk);
            last_sign = ((ngrow[w] >> hb) & 1) ? -1 : 1;
          }
        }
      }
    }
    s_nblobs = nb;
    s_err = err;
  }
  __syncthreads();

  // ---- I: shape filter, float32 centroid, undistort; output order = OpenCV's contour order
  //         (newest contour first, i.e. reverse discovery order)
  const int nb_all = s_nblobs;
  const int nb = min(nb_all, MPE_MAX_RAW_BLOBS);
  int kept = 0;
  for (int base = ((nb - 1) / 64) * 64; base >= 0 && nb > 0; base -= 64) {
    const int i = base + (63 - lane);
    float mcx = 0.f, mcy = 0.f;
    bool ok = false;
    if (i < nb) ok = blob_filter(s_blobs[i], dp, mcx, mcy);
    const u64 bal = __ballot(ok);
    const int pos = kept + __builtin_popcountll(bal & ((1ull << lane) - 1));
    if (ok && pos < MPE_MAX_DETECTIONS) {
      float ux, uy;
      undistort_point(mcx, mcy, dp, ux, uy);
      out->dist_xy[2 * pos] = mcx;
      out->dist_xy[2 * pos + 1] = mcy;
      out->undist_xy[2 * pos] = (double)ux;
      out->undist_xy[2 * pos + 1] = (double)uy;
    }
    kept += __builtin_popcountll(bal);
  }
  if (lane == 0) {
    out->n = min(kept, MPE_MAX_DETECTIONS);
    int st = 0;
    if (kept > MPE_MAX_DETECTIONS) st = MPE_FRAME_TOO_MANY_DETECTIONS;
    if (nb_all > MPE_MAX_RAW_BLOBS || s_err) st = MPE_FRAME_TOO_MANY_BLOBS;
    out->status = st;
  }
}

size_t k1b_lds_bytes(const FrameGeom& g) {
  const size_t bm_words = (size_t)g.slot_cap * g.wb;
  const size_t wm_words = (bm_words + 63) / 64 + 1;
  size_t words = 3 * bm_words + (size_t)g.slot_cap * g.tw + 2 * (size_t)g.rw + wm_words;
  return words * 8 + 2 * (size_t)g.rw * 4 + 16;
}

hipError_t launch_k1b_blobs(const uint8_t* frames, const unsigned long long* flags, int n_frames, const FrameGeom& g,
                            const DetectParams& dp, mpe_detections* dets, hipStream_t s) {
  if (n_frames <= 0) return hipSuccess;
  const size_t lds = k1b_lds_bytes(g);
  static size_t configured = 0;
  if (lds > configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k1b_blobs),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    configured = lds;
  }
  hipLaunchKernelGGL(k1b_blobs, dim3(n_frames), dim3(64), lds, s, frames, (const u64*)flags, g, dp, dets);
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

  for (int i = tid; i < MPE_HIST_STRIDE; i += K2_THREADS) s_hist[i] = 0;
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

  for (long long t = (long long)part * K2_THREADS + tid; t < total; t += (long long)splits * K2_THREADS) {
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
      p3p_solution(ctx, ctx.root[k], R, C);
      if (!rc_finite(R, C)) continue;
      const Proj P = make_projection(R, C, sp.fx, sp.fy, sp.cx, sp.cy);
      // back-project the unused markers (ascending marker index)
      int j = 0;
      for (int m = 0; m < n_m; ++m) {
        if (m == p0 || m == p1 || m == p2) continue;
        double u, v;
        project(P, V3{s_mk[m][0], s_mk[m][1], s_mk[m][2]}, u, v);
        s_q[(2 * j) * K2_THREADS + tid] = u;
        s_q[(2 * j + 1) * K2_THREADS + tid] = v;
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
          const double du = au - s_q[(2 * jj) * K2_THREADS + tid], dv = av - s_q[(2 * jj + 1) * K2_THREADS + tid];
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
  for (int i = tid; i < MPE_HIST_STRIDE; i += K2_THREADS) {
    const unsigned v = s_hist[i];
    if (v) atomicAdd(&gh[i], v);
  }
}

hipError_t launch_k2_vote(const mpe_detections* dets, int n_frames, const SolveParams& sp, uint32_t* hist, int splits,
                          hipStream_t s) {
  if (n_frames <= 0) return hipSuccess;
  if (splits < 1) splits = 1;
  const int nuo = sp.n_markers > 3 ? sp.n_markers - 3 : 1;
  const size_t lds = (size_t)nuo * 2 * K2_THREADS * sizeof(double);
  hipLaunchKernelGGL(k2_vote, dim3((unsigned)(n_frames * splits)), dim3(K2_THREADS), lds, s, dets, sp, hist, splits);
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

#define K3_THREADS 64
__global__ __launch_bounds__(K3_THREADS) void k3_tail(const mpe_detections* __restrict__ dets,
                                                      const uint32_t* __restrict__ hist, int n_frames,
                                                      SolveParams sp, mpe_result* __restrict__ results,
                                                      uint32_t* __restrict__ corr_out) {
  const int f = blockIdx.x * K3_THREADS + threadIdx.x;
  if (f >= n_frames) return;
  const mpe_detections* d = dets + f;
  mpe_result* res = results + f;
  const int n_d = d->n, n_m = sp.n_markers;
  const uint32_t* H = hist + (size_t)f * MPE_HIST_STRIDE;
  const double fx = sp.fx, fy = sp.fy, cx = sp.cx, cy = sp.cy;

  // default output: identity pose, zero covariance, no pose
  for (int i = 0; i < 16; ++i) res->T[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int i = 0; i < 36; ++i) res->cov[i] = 0.0;
  res->n_det = n_d;
  res->n_corr = 0;
  res->gn_iterations = 0;
  res->status = (d->status != 0) ? d->status : MPE_FRAME_NO_POSE;
  if (corr_out)
    for (int i = 0; i < 2 * MPE_MAX_MARKERS; ++i) corr_out[(size_t)f * 2 * MPE_MAX_MARKERS + i] = 0;
  if (d->status != 0 || n_d < 4 || n_m < 4) return;

  // ---- initialise(): all-zero histogram -> 0   (pose_estimator.cpp:704)
  bool any = false;
  for (int r = 0; r < n_d; ++r)
    for (int c = 0; c < n_m; ++c) any |= (H[r * MPE_MAX_MARKERS + c] != 0);
  if (!any) return;

  // ---- correspondencesFromHistogram (pose_estimator.cpp:344-370)
  unsigned char cm[MPE_MAX_MARKERS], cd[MPE_MAX_MARKERS];  // 1-based (marker, detection)
  int n_c = 0;
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
    cm[n_c] = (unsigned char)(ci + 1);
    cd[n_c] = (unsigned char)(ri + 1);
    ++n_c;
    removed |= 1u << ci;
  }
  res->n_corr = n_c;
  if (corr_out)
    for (int i = 0; i < n_c; ++i) {
      corr_out[(size_t)f * 2 * MPE_MAX_MARKERS + 2 * i] = cm[i];
      corr_out[(size_t)f * 2 * MPE_MAX_MARKERS + 2 * i + 1] = cd[i];
    }

  // ---- checkCorrespondences (pose_estimator.cpp:394-542)
  if (n_c < 4) return;
  double mean[MPE_MAX_MARKERS][3];
  for (int i = 0; i < n_m; ++i) mean[i][0] = mean[i][1] = mean[i][2] = 0.0;
  const int nu = n_c - 3;
  unsigned N = 0, num_valid = 0;
  for (int a = 0; a < n_c; ++a)
    for (int b = a + 1; b < n_c; ++b)
      for (int c = b + 1; c < n_c; ++c) {
        ++N;
        const int rows3[3] = {a, b, c};
        V3 fv[3], wp[3];
        for (int k = 0; k < 3; ++k) {
          const int mi = cm[rows3[k]] - 1, di = cd[rows3[k]] - 1;
          wp[k] = {sp.markers[3 * mi], sp.markers[3 * mi + 1], sp.markers[3 * mi + 2]};
          fv[k] = bearing(d->undist_xy[2 * di], d->undist_xy[2 * di + 1], fx, fy, cx, cy);
        }
        P3PCtx ctx;
        if (!p3p_prepare(fv[0], fv[1], fv[2], wp[0], wp[1], wp[2], ctx)) continue;
        double min_sq = INFINITY;
        int best = -1;
        for (int k = 0; k < 4; ++k) {
          M3 R;
          V3 C;
          p3p_solution(ctx, ctx.root[k], R, C);
          if (!rc_finite(R, C)) continue;
          const Proj P = make_projection(R, C, fx, fy, cx, cy);
          // unused correspondences, ascending row index
          double bu[MPE_MAX_MARKERS], bv[MPE_MAX_MARKERS], iu[MPE_MAX_MARKERS], ivv[MPE_MAX_MARKERS];
          int q = 0;
          for (int l = 0; l < n_c; ++l) {
            if (l == a || l == b || l == c) continue;
            const int mi = cm[l] - 1, di = cd[l] - 1;
            project(P, V3{sp.markers[3 * mi], sp.markers[3 * mi + 1], sp.markers[3 * mi + 2]}, bu[q], bv[q]);
            iu[q] = d->undist_xy[2 * di];
            ivv[q] = d->undist_xy[2 * di + 1];
            ++q;
          }
          // calculateSquaredReprojectionErrorAndCertainty (pose_estimator.cpp:303-342): greedy
          // global-minimum matching, column-major first minimum, rows = image points
          unsigned rowdone = 0, coldone = 0;
          double sq = 0;
          unsigned ncorr = 0;
          for (int it = 0; it < nu; ++it) {
            double mv = 0;
            int ri = 0, ci = 0;
            bool first = true;
            for (int cj = 0; cj < nu; ++cj)
              for (int rr = 0; rr < nu; ++rr) {
                double v;
                if (((rowdone >> rr) & 1) || ((coldone >> cj) & 1))
                  v = INFINITY;
                else {
                  const double du = iu[rr] - bu[cj], dv = ivv[rr] - bv[cj];
                  v = sqrt(du * du + dv * dv);
                }
                if (first || v < mv) {
                  mv = v;
                  ri = rr;
                  ci = cj;
                  first = false;
                }
              }
            if (mv <= sp.back_tol) {
              sq += mv * mv;
              ++ncorr;
              rowdone |= 1u << ri;
              coldone |= 1u << ci;
            } else
              break;
          }
          const double certainty = (double)ncorr / (double)nu;
          if (certainty >= sp.certainty_thr) {  // pose_estimator.cpp:494-502
            if (best < 0) best = -2;              // valid_correspondence_found
            if (sq < min_sq) {
              min_sq = sq;
              best = k;
            }
          }
        }
        if (best == -1) continue;
        ++num_valid;
        if (best < 0) best = 0;  // unreachable: sq is always finite, so a valid solution always sets the index
        M3 R;
        V3 C;
        p3p_solution(ctx, ctx.root[best], R, C);
        // inverse(H) * marker for ALL markers (pose_estimator.cpp:513-517)
        for (int jj = 0; jj < n_m; ++jj) {
          const V3 mk = {sp.markers[3 * jj] - C.x, sp.markers[3 * jj + 1] - C.y, sp.markers[3 * jj + 2] - C.z};
          const V3 pc = mulT(R, mk);  // R^T (m - C)
          mean[jj][0] += pc.x;
          mean[jj][1] += pc.y;
          mean[jj][2] += pc.z;
        }
      }
  if (!((double)num_valid / (double)N >= sp.valid_corr_thr)) return;

  // ---- computeTransformation (pose_estimator.cpp:908-930)
  T34 T;
  {
    double mo[3] = {0, 0, 0}, mr[3] = {0, 0, 0};
    for (int i = 0; i < n_m; ++i)
      for (int k = 0; k < 3; ++k) {
        mean[i][k] = mean[i][k] / (double)num_valid;
        mo[k] += sp.markers[3 * i + k];
        mr[k] += mean[i][k];
      }
    for (int k = 0; k < 3; ++k) {
      mo[k] /= (double)n_m;
      mr[k] /= (double)n_m;
    }
    double Hm[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int i = 0; i < n_m; ++i)
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Hm[r][c] += (sp.markers[3 * i + r] - mo[r]) * (mean[i][c] - mr[c]);
    double X[3][3];  // H^T
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) X[r][c] = Hm[c][r];
    polar3(X);  // R = V U^T
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) T.m[r][c] = X[r][c];
      T.m[r][3] = mr[r] - (X[r][0] * mo[0] + X[r][1] * mo[1] + X[r][2] * mo[2]);
    }
  }

  // ---- optimisePose (pose_estimator.cpp:733-792): Gauss-Newton on SE(3)
  double A[6][6];
  int iters = 0;
  for (int it = 0; it < 500; ++it) {
    double b[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      b[r] = 0;
#pragma unroll
      for (int c = 0; c < 6; ++c) A[r][c] = 0;
    }
    for (int j = 0; j < n_c; ++j) {
      const int mi = cm[j] - 1, di = cd[j] - 1;
      double u, v, x, y, z;
      project_T(T, &sp.markers[3 * mi], fx, fy, cx, cy, u, v, x, y, z);
      const double e0 = d->undist_xy[2 * di] - u, e1 = d->undist_xy[2 * di + 1] - v;
      const double z_2 = z * z;
      // computeJacobian, pose_estimator.cpp:945-957
      const double J0[6] = {1 / z * fx, 0, -x / z_2 * fx, -x * y / z_2 * fx, (1 + (x * x / z_2)) * fx, -y / z * fx};
      const double J1[6] = {0, 1 / z * fy, -y / z_2 * fy, -(1 + y * y / z_2) * fy, x * y / z_2 * fy, x / z * fy};
#pragma unroll
      for (int r = 0; r < 6; ++r) {
#pragma unroll
        for (int c = 0; c < 6; ++c) A[r][c] += J0[r] * J0[c] + J1[r] * J1[c];
        b[r] += J0[r] * e0 + J1[r] * e1;
      }
    }
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
  {
    LDL6 F;
    ldl6_factor(A, F);
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double e[6] = {0, 0, 0, 0, 0, 0}, x[6];
      e[c] = 1.0;
      ldl6_solve(F, e, x);
#pragma unroll
      for (int r = 0; r < 6; ++r) res->cov[r * 6 + c] = x[r];
    }
  }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) res->T[r * 4 + c] = T.m[r][c];
  res->gn_iterations = iters;
  res->status = MPE_FRAME_POSE;
}

hipError_t launch_k3_tail(const mpe_detections* dets, const uint32_t* hist, int n_frames, const SolveParams& sp,
                          mpe_result* results, uint32_t* corr_out, hipStream_t s) {
  if (n_frames <= 0) return hipSuccess;
  hipLaunchKernelGGL(k3_tail, dim3((n_frames + K3_THREADS - 1) / K3_THREADS), dim3(K3_THREADS), 0, s, dets, hist,
                     n_frames, sp, results, corr_out);
  return hipGetLastError();
}

}  // namespace mpe
