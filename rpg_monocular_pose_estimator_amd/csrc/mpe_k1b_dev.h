// mpe_k1b_dev.h — K1b blob extraction, the DEVICE functions (blur, contour phases, capacity tiers, front phases, one
// frame per wave: k1b_wave).  Included by mpe_k1.hip (the blob kernels) and by mpe_k3.hip (k_track_frame: the tracked
// frame as ONE launch, round 6).  Part of the kernel sources: binding.device_source() reads it in place of the
// #include line.
#pragma once
#include "mpe_kernels_common.h"
namespace mpe {
//@k1b-dev-begin
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
  const u64* flags;      // raw mode: the image pass's flag bitmap (bit = a 16-byte segment holds a pixel above the
  size_t fbit0;          //   threshold) and the bit index of this window's first segment: a segment whose bit is clear
                         //   thresholds to sixteen zeros and is neither loaded nor thresholded
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
  // RAW (the general kernel: frame bytes and flag words in GLOBAL memory): the KS rows' flag words first, then the
  // pixel segments they allow, all loads of a stage in flight together — row by row an item waited for 2 KS dependent
  // round trips, and that tier runs too few waves to hide them (round 6: 31 k cycles per round of 64 items)
  uint4 rq[RAW ? KS : 1][3];
  unsigned rany = 0;
  if constexpr (RAW) {
    const int sc = c - 1 - w.pwc0, nsw = w.PW >> 4;
    u64 fa[KS], fm[KS], fb[KS];
    size_t g0s[KS];
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int yb = reflect101(y + i - R, rows) - w.ylo;
      const bool rowok = (unsigned)yb < (unsigned)w.H;
      g0s[i] = w.fbit0 + (size_t)(rowok ? yb : 0) * (size_t)nsw + (size_t)(sc + 1);  // (segment sc + 1: never negative)
      // (the neighbour segments' words only where those segments exist: column c - 1 of the frame's first row would
      //  index the flag stream at -1)
      fa[i] = (rowok && sc >= 0) ? w.flags[(g0s[i] - 1) >> 6] : 0ull;
      fm[i] = rowok ? w.flags[g0s[i] >> 6] : 0ull;
      fb[i] = (rowok && sc + 2 < nsw) ? w.flags[(g0s[i] + 1) >> 6] : 0ull;
    }
    const uint4 z4 = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int yb = reflect101(y + i - R, rows) - w.ylo;
      const bool rowok = (unsigned)yb < (unsigned)w.H;
      const size_t g0 = g0s[i];
      const bool in0 = rowok && (unsigned)sc < (unsigned)nsw && ((fa[i] >> ((g0 - 1) & 63)) & 1ull);
      const bool in1 = rowok && (unsigned)(sc + 1) < (unsigned)nsw && ((fm[i] >> (g0 & 63)) & 1ull);
      const bool in2 = rowok && (unsigned)(sc + 2) < (unsigned)nsw && ((fb[i] >> ((g0 + 1) & 63)) & 1ull);
      const uint4* p = reinterpret_cast<const uint4*>(w.pix + (size_t)(rowok ? yb : 0) * w.PW);
      rq[i][0] = in0 ? p[sc] : z4;
      rq[i][1] = in1 ? p[sc + 1] : z4;
      rq[i][2] = in2 ? p[sc + 2] : z4;
      rany |= (in0 || in1 || in2) ? (1u << i) : 0u;
    }
  }
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    const int yb = reflect101(y + i - R, rows) - w.ylo;
    if ((unsigned)yb >= (unsigned)w.H) continue;  // rows outside the band hold no bright pixel
    const int sc = c - 1 - w.pwc0, nsw = w.PW >> 4;  // window segment index of column c-1
    const uint4* p = reinterpret_cast<const uint4*>(w.pix + (size_t)yb * w.PW);
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    const bool in0 = (unsigned)sc < (unsigned)nsw, in1 = (unsigned)(sc + 1) < (unsigned)nsw, in2 = (unsigned)(sc + 2) < (unsigned)nsw;
    if (RAW) {  // only segments the image pass flagged can hold anything after the threshold
      if (!((rany >> i) & 1u)) continue;  // this row adds nothing to the sums (most rows of most items)
    }
    const uint4 q0 = RAW ? rq[RAW ? i : 0][0] : (in0 ? p[sc] : z4);
    const uint4 q1 = RAW ? rq[RAW ? i : 0][1] : (in1 ? p[sc + 1] : z4);
    const uint4 q2 = RAW ? rq[RAW ? i : 0][2] : (in2 ? p[sc + 2] : z4);
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
  const int pos1 = pos0 + dir_dy(s) * 65536 + dir_dx(s);  // (dy may be -1: no shift of a negative value)
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
    const int npos = pos + dy * 65536 + dx;
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
// COLS (round 6, the general tier's (band, column run) items): the scan only looks at the bit columns xb_lo .. xb_hi of
// the window — a maximal run of pixel columns that hold a set pixel in some row of the window, with an EMPTY column on
// either side.  No component crosses an empty column, and none beyond it can enclose one inside the run, so the raster
// scan decomposes there exactly as it does at an empty row: candidate and mark words are masked to the run (other runs
// of the band share its 64-bit words and are traced by other lanes at the same time: the marks are set atomically),
// the cursor moves over the run's words only.
template <bool COLS = false, class Emit>
__device__ __forceinline__ void scan_window(u64* nz, u64* pm, u64* ng, int W, int H, int ylo, int xw0,
                                            const DetectParams& dp, int roi_x, int roi_y, int* over, Emit emit,
                                            int xb_lo = 0, int xb_hi = 0) {
  const int w_lo = COLS ? (xb_lo >> 6) : 0, w_hi = COLS ? (xb_hi >> 6) : W - 1;
  auto wmask = [&](int w) -> u64 {
    if (!COLS) return ~0ull;
    u64 m = ~0ull;
    if (w == w_lo) m &= ~0ull << (xb_lo & 63);
    if (w == w_hi) m &= ~0ull >> (63 - (xb_hi & 63));
    return m;
  };
  int slot = 1, w = w_lo;
  int last_sign = 0;  // sign of the nearest marked pixel to the left (lnbd), 0 = none yet
  u64 done = 0;
  bool fin = H < 1;
  for (;;) {
    // ---- find: the next unmarked 1 with a 0 on its left that is not inside an already traced outer border
    bool have = false;
    int xb = 0;
    while (!have && !fin) {
      // (empty words — most of a window — only move the cursor: done is 0 on arrival, no mark can sit on them)
      if (COLS && w_lo == w_hi) {
        // a run inside one 64-bit word (the rule): four rows' words in flight at a time — the band's rows chain through
        // OTHER runs, so most of this run's rows are empty, and every empty row cost a dependent read
        const u64 m1 = wmask(w_lo);
        while (!fin && w == w_lo) {
          const u64* q = nz + (size_t)slot * W + w_lo;
          const u64 a0 = q[0] & m1;
          const u64 a1 = (slot + 1 <= H) ? (q[W] & m1) : 0;
          const u64 a2 = (slot + 2 <= H) ? (q[2 * W] & m1) : 0;
          const u64 a3 = (slot + 3 <= H) ? (q[3 * W] & m1) : 0;
          if (a0) break;
          const int adv = a1 ? 1 : (a2 ? 2 : (a3 ? 3 : 4));  // (an empty word only moves the cursor: done is 0 on arrival)
          slot += adv;
          last_sign = 0;
          fin = slot > H;
          if (adv < 4) break;
        }
      }
      while (!fin && (nz[slot * W + w] & wmask(w)) == 0) {
        if (++w > w_hi) {
          w = w_lo;
          last_sign = 0;
          fin = ++slot > H;
        }
      }
      if (fin) break;
      const int ro = slot * W + w;
      const u64 wm = wmask(w);
      const u64 nzw = nz[ro] & wm;
      const u64 pw_ = pm[ro] & wm, gw = ng[ro] & wm;
      // (COLS: the bit left of the run's first column is an empty column by construction)
      const u64 leftnz = (nzw << 1) | (w > w_lo ? (nz[ro - 1] >> 63) : 0);
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
        if (++w > w_hi) {
          w = w_lo;
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

// Every maximal run of set bits of a row of W words (the OR of a window's rows: its occupied columns) -> emit(xb_lo, xb_hi)
template <class EmitRun>
__device__ __forceinline__ void column_runs_of_words(const u64* occ, int W, EmitRun emit) {
  int start = -1;
  for (int w = 0; w < W; ++w) {
    const u64 v = occ[w];
    int pos = 0;
    while (pos < 64) {
      if (start < 0) {
        const u64 m = v >> pos;
        if (!m) break;
        const int t = __builtin_ctzll(m);
        start = 64 * w + pos + t;
        pos += t;
      } else {
        const u64 m = ~v >> pos;  // (the zeros shifted in from above read as "still set": the run goes on in the next word)
        if (!m) break;
        const int t = __builtin_ctzll(m);
        emit(start, 64 * w + pos + t - 1);
        start = -1;
        pos += t;
      }
    }
  }
  if (start >= 0) emit(start, 64 * W - 1);
}

// The runs of occupied bit columns of a window (slots 1 .. H, W <= MAXW words per row): OR of the rows, then every
// maximal run of set bits -> emit(xb_lo, xb_hi).  The general tier's (band, column run) items: scan_window<true>.
template <int MAXW, class EmitRun>
__device__ __forceinline__ void window_column_runs(const u64* nz, int W, int H, EmitRun emit) {
  u64 occ[MAXW];
#pragma unroll
  for (int w = 0; w < MAXW; ++w) occ[w] = 0;
  int r = 1;
  for (; r + 3 <= H; r += 4) {  // (four rows' loads in flight per trip)
#pragma unroll
    for (int w = 0; w < MAXW; ++w)
      if (w < W) {
        const u64* q = nz + (size_t)r * W + w;
        occ[w] |= (q[0] | q[W]) | (q[2 * W] | q[3 * W]);
      }
  }
  for (; r <= H; ++r) {
#pragma unroll
    for (int w = 0; w < MAXW; ++w)
      if (w < W) occ[w] |= nz[(size_t)r * W + w];
  }
  int start = -1;
#pragma unroll
  for (int w = 0; w < MAXW; ++w) {
    if (w >= W) continue;
    const u64 v = occ[w];
    int pos = 0;
    while (pos < 64) {
      if (start < 0) {
        const u64 m = v >> pos;
        if (!m) break;
        const int t = __builtin_ctzll(m);
        start = 64 * w + pos + t;
        pos += t;
      } else {
        const u64 m = ~v >> pos;  // (the zeros shifted in from above read as "still set": the run goes on in the next word)
        if (!m) break;
        const int t = __builtin_ctzll(m);
        emit(start, 64 * w + pos + t - 1);
        start = -1;
        pos += t;
      }
    }
  }
  if (start >= 0) emit(start, 64 * W - 1);
}

// ... and a column run cut again at the rows that are EMPTY within its columns xb0 .. xb1 (the window's rows hang together
// through OTHER runs): emit(first_slot, n_rows) for every maximal run of non-empty rows.  An empty row separates what is
// above it from what is below exactly as it does for whole bands, and the run's side columns are empty in every row.
template <class EmitRows>
__device__ __forceinline__ void window_run_rows(const u64* nz, int W, int H, int xb0, int xb1, EmitRows emit) {
  const int wl = xb0 >> 6, wh = xb1 >> 6;
  const u64 ml = ~0ull << (xb0 & 63), mh = ~0ull >> (63 - (xb1 & 63));
  int first = -1;
  auto row_any = [&](int r) -> u64 {
    if (r > H) return 0;
    u64 any = 0;
    for (int w = wl; w <= wh; ++w) {
      u64 v = nz[(size_t)r * W + w];
      if (w == wl) v &= ml;
      if (w == wh) v &= mh;
      any |= v;
    }
    return any;
  };
  auto step = [&](int r, u64 any) {
    if (any) {
      if (first < 0) first = r;
    } else if (first >= 0) {
      emit(first, r - first);
      first = -1;
    }
  };
  int r = 1;
  for (; r + 7 <= H + 1; r += 8) {  // (eight rows' words in flight per trip: the rows are independent reads)
    u64 a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = row_any(r + q);
#pragma unroll
    for (int q = 0; q < 8; ++q) step(r + q, a[q]);
  }
  for (; r + 3 <= H + 1; r += 4) {
    const u64 a0 = row_any(r), a1 = row_any(r + 1), a2 = row_any(r + 2), a3 = row_any(r + 3);
    step(r, a0);
    step(r + 1, a1);
    step(r + 2, a2);
    step(r + 3, a3);
  }
  for (; r <= H + 1; ++r) step(r, row_any(r));
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
struct BlurNoNote {
  __device__ __forceinline__ void operator()(int) const {}
};
// (note(word index): called for every bitmap word of the row that receives a bit — the general kernel lists them)
template <bool RAW = false, class Note = BlurNoNote>
__device__ __forceinline__ void blur_to_bitmap(const PixWin& pw, int rows, int cols, const DetectParams& dp,
                                               const int* taps, int y, int c, u64* nzrow, int xw0, Note note = Note()) {
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
    note(wi);
    if (shb > 48 && ((u64)m >> (64 - shb))) {
      atomicOr(&nzrow[wi + 1], (u64)m >> (64 - shb));
      note(wi + 1);
    }
  }
}

// Does bitmap row `cur` touch row `prev` (some set pixel of one 8-adjacent to a set pixel of the other)?  Rows of W
// 64-bit words, bit = pixel; the 3-neighbourhood of `prev` crosses word boundaries.
__device__ __forceinline__ bool k1b_rows_touch(const u64* cur, const u64* prev, int W) {
  u64 hit = 0;
  for (int w = 0; w < W; ++w) {
    const u64 p = prev[w];
    u64 dil = p | (p << 1) | (p >> 1);
    if (w > 0) dil |= prev[w - 1] >> 63;
    if (w + 1 < W) dil |= prev[w + 1] << 63;
    hit |= cur[w] & dil;
  }
  return hit != 0;
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
  short blo, bhi;      // segment columns the blur actually computes (clo .. chi, or without a neighbour column that
                       // provably blurs to zero: phase C2)
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
#ifndef K1B_SMALL_MIN_WAVES
#define K1B_SMALL_MIN_WAVES 4  // (experiment builds: 5 caps the kernel at 96 VGPRs, so that two side-scan waves per SIMD fit beside four of its own)
#endif
#ifndef K1B_SMALL_PIX
#define K1B_SMALL_PIX 4096
#endif
#ifndef K1B_SMALL_BM
#define K1B_SMALL_BM 208
#endif
struct K1bSmall {
  enum { PIX = K1B_SMALL_PIX, BM = K1B_SMALL_BM, SEG = 64, BAND = 8, ISL = 8, KEPT = 16, WAVES = K1B_SMALL_WAVES, MIN_WAVES = K1B_SMALL_MIN_WAVES };
};
struct K1bLarge {
  enum { PIX = 12288, BM = 704, SEG = 512, BAND = 32, ISL = 32, KEPT = 64, WAVES = 1, MIN_WAVES = 2 };
};

// (wave_sync: mpe_kernels_common.h)

template <class C>
struct K1bWaveLds {  // front-phase storage of one wave
  enum { SCRATCH = 4 * C::SEG + 512 + 32 * C::BAND + 4 * C::BAND, POOL = C::PIX > SCRATCH ? C::PIX : SCRATCH };
  __attribute__((aligned(16))) uint8_t pool[POOL];
  int taps[MPE_MAX_KSIZE];  // (taking the address of the by-value kernel argument would make the
                            //  compiler copy all of it to scratch)
  int nseg, nband;
  int blur_go;  // (k1b_front<C, true>: the block's other waves take their share of the blur's items)
#ifdef K1B_PHASE_CLOCKS  // (experiment builds: the shader clock at the phase boundaries of a frame, printed by block 0)
  unsigned long long clk[8];
#endif
};
#ifdef K1B_PHASE_CLOCKS
#define K1B_PHASE_STAMP(W, i) \
  if ((threadIdx.x & 63) == 0) (W).clk[i] = __builtin_amdgcn_s_memtime();
#else
#define K1B_PHASE_STAMP(W, i)
#endif
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
// Phase D of k1b_front — the blurred mask of every island, one (row, segment) item per thread `tid` of `nthr`: the wave
// of the frame alone (64), or every wave of the block (k1b_front<C, true>, the tracked frame's kernel).
template <class C>
__device__ __forceinline__ void k1b_blur_items(K1bWaveLds<C>& W, K1bFrameLds<C>& S, const FrameGeom& g, const DetectParams& dp,
                                               int tid, int nthr) {
  const Island* s_isl = S.isl;
  const int nisl = S.nisl, r = dp.ksize / 2;
  const int tot_blur = s_isl[nisl - 1].blur_end;
  for (int i = tid; i < tot_blur; i += nthr) {
    int k = 0;
    while (i >= s_isl[k].blur_end) ++k;
    const Island is = s_isl[k];
    const int li = i - (k ? s_isl[k - 1].blur_end : 0);
    const int ncols = is.bhi - is.blo + 1;
    const int yb = li / ncols, c = is.blo + (li - yb * ncols);
    const int H = is.yhi - is.ylo + 1;
    const int Wd = isl_words(is, g.cols, r);
    const PixWin pw = {W.pool + is.pix_off, is.ylo, H, is.cfirst, 16 * (is.clast - is.cfirst + 1), 0u, nullptr, 0};
    blur_to_bitmap(pw, g.rows, g.cols, dp, W.taps, is.ylo + yb, c, S.nz + is.bm_off + (size_t)(yb + 1) * Wd, isl_xw0(is, r));
  }
}

// false when the frame is finished (no bright pixel) or was handed to the next tier's work-list.
// HELP (the tracked frame's kernel, a block of several waves on ONE frame): the front phases are this wave's, the blur's
// items every wave's — a lone wave pays ~5 k cycles of latency per round of 64 items whatever the arithmetic (four rounds
// for five LED islands: 43 k of the blob extraction's 90 k cycles), and the CU's other SIMDs have nothing else to do.
template <class C, bool HELP = false>
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
  K1B_PHASE_STAMP(W, 1)
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
        is.blo = is.clo;
        is.bhi = is.chi;
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
  K1B_PHASE_STAMP(W, 2)

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
  K1B_PHASE_STAMP(W, 3)

  // ---- C2 (round 6): which neighbour columns does the blur have to compute at all?  The island's columns clo .. chi
  //      are its bright segments cfirst .. clast dilated by dc; with dc = 1 the column left of cfirst can only blur to
  //      something if a thresholded pixel sits in the FIRST r pixels of segment cfirst in some row of the band (an
  //      output x sees inputs x - r .. x + r), the column right of clast only through the LAST r pixels of clast —
  //      for an LED of ~5 pixels somewhere in a 16-pixel segment that is one case in four.  A column that cannot is left
  //      out of the blur's work list (its bitmap words stay zero, which is what its items would have written): 135 ->
  //      ~70 items for five LEDs, three wave passes -> two.  Only away from the image border (no BORDER_REFLECT_101
  //      read can reach the bright segments from that column) and for r <= 16.
  //      MEASURED AND NOT ADOPTED (profiles/round6_exp_blur_narrowing.txt, same box, four pairs): the blob kernel's
  //      window shrinks from 0.52 to 0.50 ms and the voting launch behind it grows from 1.72 to 1.80 — the step gets
  //      0.5 ms SLOWER (18.24 - 18.42 against 17.55 - 18.02 ms), as with every other change that only shortens the window
  //      (Schedules).  Detections bit-equal either way.  Kept for experiment builds.
#ifdef K1B_BLUR_NARROWING
  if (dc == 1) {
    int nbl = 0;
    if (lane < nisl) {
      Island is = s_isl[lane];
      const int H = is.yhi - is.ylo + 1, nbs = is.clast - is.cfirst + 1;
      const bool may_l = is.clo < is.cfirst && 16 * (is.cfirst - 1) >= r;
      const bool may_r = is.chi > is.clast && 16 * (is.clast + 2) + r <= g.cols;
      if (may_l || may_r) {
        unsigned left = 0, right = 0;
        const uint8_t* base = s_pix + is.pix_off;
        for (int yb = 0; yb < H; ++yb) {
          const uint8_t* rowp = base + (size_t)yb * 16 * nbs;
          for (int k = 0; k < r; ++k) {
            left |= rowp[k];
            right |= rowp[16 * nbs - 1 - k];
          }
        }
        if (may_l && !left) is.blo = is.cfirst;
        if (may_r && !right) is.bhi = is.clast;
        s_isl[lane].blo = is.blo;
        s_isl[lane].bhi = is.bhi;
      }
      nbl = H * (is.bhi - is.blo + 1);
    }
    int il = nbl;  // inclusive scan (nisl <= 32)
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
      const int e = __shfl_up(il, d);
      if (lane >= d) il += e;
    }
    if (lane < nisl) s_isl[lane].blur_end = il;
    wave_sync();
  }
#endif

  // ---- D: blurred mask of every island
  if constexpr (HELP) {
    if (lane == 0) W.blur_go = 1;
    __syncthreads();  // (the helper waves wait here, k1b_wave: islands, pixels and taps are theirs to read now)
    k1b_blur_items<C>(W, S, g, dp, (int)threadIdx.x, (int)blockDim.x);
    __syncthreads();  // (every wave's bits are in the bitmaps)
  } else {
    k1b_blur_items<C>(W, S, g, dp, lane, 64);
  }
  wave_sync();
  K1B_STOP_POINT(4, out)
  K1B_PHASE_STAMP(W, 4)
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
// HELP: the block has more waves than frames (C::WAVES == 1: one frame); wave 0 runs the frame, the others enter here
// too, take their share of the blur's items (k1b_front<C, true>) and leave.  Their barriers pair with wave 0's whatever
// path it takes: if it never reaches the blur (no bright pixel, a capacity hand-over) they meet its two barriers behind
// the front phases instead and find blur_go unset; if it does, they pass those two as well.
template <class C, bool HELP = false>
__device__ __forceinline__ void k1b_wave(const int f, const bool valid, const uint8_t* __restrict__ frames,
                                         const u64* __restrict__ flags, const FrameGeom& gslot, const DetectParams& dp,
                                         mpe_detections* __restrict__ dets, int* __restrict__ worklist,
                                         const FrameWin* __restrict__ wins) {
  __shared__ K1bWaveLds<C> Wl[C::WAVES];
  __shared__ K1bFrameLds<C> Sl[C::WAVES];
  static_assert(!HELP || C::WAVES == 1, "helper waves: one frame per block");
  const int lane = threadIdx.x & 63;
  const int wv = C::WAVES > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  K1bWaveLds<C>& W = Wl[wv];
  K1bFrameLds<C>& S = Sl[wv];
  if (HELP && threadIdx.x == 0) W.blur_go = 0;
  __syncthreads();  // (list mode: the previous group of this block is completely done)
  if (HELP && threadIdx.x >= 64) {
    int roi_x, roi_y;
    const FrameGeom gh = window_geom(gslot, wins, valid ? f : 0, dp, roi_x, roi_y);
    __syncthreads();  // wave 0: in front of the blur (or, if it never gets there, its barrier behind the front phases)
    const bool go = W.blur_go != 0;
    if (go) k1b_blur_items<C>(W, S, gh, dp, (int)threadIdx.x, (int)blockDim.x);
    __syncthreads();  // wave 0: behind the blur (or its barrier behind the contour phase)
    if (go) {         // ... and wave 0's two barriers behind the front phases: every wave of the block passes the same
      __syncthreads();  // number of barriers per frame, so a block may take frame after frame (k1b_blobs_list_few)
      __syncthreads();
    }
    return;
  }
  if (lane < MPE_MAX_KSIZE) W.taps[lane] = dp.taps[lane < dp.ksize ? lane : 0];
  const size_t slot_bytes = (size_t)gslot.rows * gslot.pitch;
  int roi_x, roi_y;
  const FrameGeom g = window_geom(gslot, wins, valid ? f : 0, dp, roi_x, roi_y);
  bool ready = false;
  K1B_PHASE_STAMP(W, 0)
  if (valid) ready = k1b_front<C, HELP>(f, frames, slot_bytes, flags, g, dp, dets, worklist, W, S);
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
  K1B_PHASE_STAMP(W, 5)
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
#ifdef K1B_PHASE_CLOCKS
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    printf("k1b phases (cycles): flags->segments %llu bands/islands %llu staging %llu blur %llu contours %llu records %llu\n",
           W.clk[1] - W.clk[0], W.clk[2] - W.clk[1], W.clk[3] - W.clk[2], W.clk[4] - W.clk[3], W.clk[5] - W.clk[4], t - W.clk[5]);
  }
#endif
}

//@k1b-dev-end
}  // namespace mpe
