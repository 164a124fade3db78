//@file-prologue
// mpe_k1.hip — K1a image scan, repack, K1b blob extraction (three tiers) (see mpe_kernels_common.h for the map of the kernel sources)
#include "mpe_kernels_common.h"

namespace mpe {
//@file-prologue-end
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

//@include-k1b-dev (the device functions of the blob extraction: mpe_k1b_dev.h, closed and re-opened namespace)
}  // namespace mpe
#include "mpe_k1b_dev.h"
namespace mpe {
//@include-k1b-dev-end
// wave w of block b works on frame b * C::WAVES + w
#ifdef K1B_NUM_VGPR  // (experiment builds: a VGPR cap below what the LDS-limited occupancy would allow the kernel)
#define K1B_VGPR_ATTR __attribute__((amdgpu_num_vgpr(K1B_NUM_VGPR)))
#else
#define K1B_VGPR_ATTR
#endif
template <class C>
__global__ __launch_bounds__(64 * C::WAVES, C::MIN_WAVES) K1B_VGPR_ATTR void k1b_blobs(const uint8_t* __restrict__ frames,
                                                           const u64* __restrict__ flags, FrameGeom g, DetectParams dp,
                                                           mpe_detections* __restrict__ dets,
                                                           int* __restrict__ worklist, int n_frames,
                                                           const FrameWin* __restrict__ wins) {
  const int f = blockIdx.x * C::WAVES + (int)(threadIdx.x >> 6);
  k1b_wave<C>(f, f < n_frames, frames, flags, g, dp, dets, worklist, wins);
}
// Few frames (a lone frame of the stage entries, the tracker's whole-image retries, small batches): a block of four
// waves per frame, waves 1 - 3 take their share of the blur's items (k1b_wave<C, true>, as in k_track_frame) — with fewer
// than four blocks per CU the other SIMDs idle, and a lone wave's blur is 43 k of its 90 k cycles.
#define K1B_HELP_THREADS 256
#define K1B_HELP_MAX_FRAMES 1024
template <class C>
__global__ __launch_bounds__(K1B_HELP_THREADS) void k1b_blobs_few(const uint8_t* __restrict__ frames,
                                                                   const u64* __restrict__ flags, FrameGeom g, DetectParams dp,
                                                                   mpe_detections* __restrict__ dets,
                                                                   int* __restrict__ worklist, const FrameWin* __restrict__ wins) {
  k1b_wave<C, true>((int)blockIdx.x, true, frames, flags, g, dp, dets, worklist, wins);
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

// ... and its four-wave form for launches of few frames (k1b_blobs_few): every wave of a block follows the list together
template <class C>
__global__ __launch_bounds__(K1B_HELP_THREADS) void k1b_blobs_list_few(const uint8_t* __restrict__ frames,
                                                                        const u64* __restrict__ flags, FrameGeom g,
                                                                        DetectParams dp, mpe_detections* __restrict__ dets,
                                                                        const int* __restrict__ in_list,
                                                                        int* __restrict__ worklist,
                                                                        const FrameWin* __restrict__ wins) {
  const int count = in_list[0];
  for (int w0 = blockIdx.x; w0 < count; w0 += gridDim.x)  // (uniform over the block)
    k1b_wave<C, true>(in_list[1 + w0] & 0xFFFFFF, true, frames, flags, g, dp, dets, worklist, wins);
}

// =============================================================================================
// K1b general path: frames the fast tiers handed over (more bright segments / bands / islands than their LDS pools
// hold: salt noise, glare, a dot grid).  One wave per frame, whole-frame bitmaps in a global scratch slab, exact on
// any frame.  Round 5 (on cluttered frames this tier was a cliff — 43 us of ONE LANE per
// frame on 32 waves of the whole chip, 11.6 k frames/s with 0.05 % salt noise):
//   * no thresholded copy of the frame: the blur reads the frame's own rows and applies THRESH_TOZERO on the fly
//     (PixWin::add) — and of the 3 x 5 segments under an item only those the image pass FLAGGED (all others threshold
//     to zeros: neither loaded nor thresholded, rows without one skipped); the (row, segment) items run as a flat
//     list, 64 at a time whatever their rows;
//   * the contour scan runs one LANE PER BAND — a maximal run of non-empty rows in which every row TOUCHES the one
//     above it (a set pixel 8-adjacent to a set pixel of the previous row).  A component of the blurred mask cannot
//     cross a boundary between two rows that do not touch, and a component below such a boundary cannot lie inside a
//     hole of one above it, so cvFindContours' raster scan decomposes exactly there, as it does at an empty row and for
//     the islands of the fast tiers; the kept blobs are put back into raster order of their start pixels at the end
//     (write_detections).  Empty rows alone cut nothing out of uniformly spread noise (0.05 % salt: every row is
//     within a blur radius of some pixel); non-touching row pairs cut such a frame into ~100 bands;
//   * up to 4096 slabs (1 GB of scratch at most) instead of 32: the kernel is bound by the latency of its bitmaps in
//     global memory, its rate follows the waves in flight (salt-noise leg, 32 768 frames: 1024 slabs 67.5 ms, 2048 45.5,
//     4096 42.5, 8192 37.2; the clean headline step, whose every sub-batch launches this tier empty, does not notice:
//     profiles/round5_exp_general_blocks.json).
// What is left: a band is still walked by ONE lane (its rows word by word, dependent bitmap reads from global memory),
// and a frame whose rows all touch — one big blob, a grid of lines — is one band.
// =============================================================================================
// Which neighbour columns of a bright segment (row y0, segment column c0) can blur to anything?  With dc = 1 the column
// on the left only through a thresholded pixel in the segment's FIRST r pixels, the one on the right only through its
// LAST r (an output x sees inputs x - r .. x + r; every other segment that could reach those columns marks them
// itself).  For an isolated bright pixel that drops two of three columns of to-do items in three cases of four — the
// blur was 35 % of a salt-noise frame, VALU bound at 16 waves per CU (2 472 items, ~900 instructions each).  Only away
// from the image border (no BORDER_REFLECT_101 read from the dropped column can reach the segment) and for r <= 16.
__device__ __forceinline__ void k1b_gen_narrow(const uint8_t* frame, int pitch, int cols, int y0, int c0, int r, int dc,
                                               unsigned add, int& cl, int& ch) {
  if (dc != 1 || !add) return;
  const bool may_l = cl < c0 && 16 * (c0 - 1) >= r, may_r = ch > c0 && 16 * (c0 + 2) + r <= cols;
  if (!may_l && !may_r) return;
  const uint4 v = *reinterpret_cast<const uint4*>(frame + (size_t)y0 * pitch + 16 * c0);
  const unsigned q[4] = {tozero4(v.x, add), tozero4(v.y, add), tozero4(v.z, add), tozero4(v.w, add)};
  unsigned left = 0, right = 0;
  for (int j = 0; j < r; ++j) {
    left |= (q[j >> 2] >> (8 * (j & 3))) & 0xFFu;
    const int t = 15 - j;
    right |= (q[t >> 2] >> (8 * (t & 3))) & 0xFFu;
  }
  if (may_l && !left) cl = c0;
  if (may_r && !right) ch = c0;
}

#define K1B_GEN_KEPT 512
#ifndef K1B_GEN_OCC
#define K1B_GEN_OCC 4  // waves per SIMD the kernel is compiled for (128 registers) and its LDS lists are sized for (10 KB)
#endif
#ifndef K1B_GEN_BANDS
#define K1B_GEN_BANDS 768  // (3 KB of LDS; a frame with more bands — only possible above 1 536 rows — is scanned whole by one lane)
#endif
#ifndef K1B_GEN_RUNS
#define K1B_GEN_RUNS 736    // (band, column run) items of a frame and their pieces (5.75 KB of LDS); more: one lane per band as in round 5
#endif
#define K1B_GEN_RUN_WORDS 16  // (window_column_runs<MAXW>: the lane-per-band form of the column occupancy, CPU-tier test only)

__host__ __device__ inline size_t k1b_gen_scratch_bytes(const FrameGeom& g) {
  const size_t bm = (size_t)(g.rows + 2) * g.wb * 8;
  const size_t todo = (size_t)g.rows * g.tw * 8;
  const size_t kept = (size_t)K1B_GEN_KEPT * 12;
  const size_t items = (size_t)g.rows * g.segs_per_row * 4;  // (row << 12 | segment column) of every to-do bit
  return ((3 * bm + todo + kept + items + 255) / 256) * 256;
}
// slabs = blocks of the launch: as many as 1 GB of scratch holds, 32 .. g_k1b_gen_blocks_cap (a process-wide tuning
// knob, option "k1b_general_blocks"; the kernel is bound by the latency of its global-memory bitmaps, so its rate
// follows the number of waves in flight)
static int g_k1b_gen_blocks_cap = 4096;
void k1b_set_general_blocks(int cap) { g_k1b_gen_blocks_cap = cap < 32 ? 32 : (cap > 8192 ? 8192 : cap); }
int k1b_get_general_blocks() { return g_k1b_gen_blocks_cap; }
// n_frames bounds them too (a one-frame tracked step used to reserve the 1 GB of a 4 096-slab launch)
static int k1b_gen_blocks(const FrameGeom& g, int n_frames) {
  size_t n = ((size_t)1 << 30) / k1b_gen_scratch_bytes(g);
  const size_t cap = (size_t)g_k1b_gen_blocks_cap;
  if (n < 32) n = 32;
  if (n > cap) n = cap;
  if (n_frames > 0 && n > (size_t)n_frames) n = (size_t)n_frames;
  return (int)n;
}

__global__ __launch_bounds__(64, K1B_GEN_OCC) void k1b_general(const uint8_t* __restrict__ frames, const u64* __restrict__ flags,
                                                 FrameGeom gslot, DetectParams dp, mpe_detections* __restrict__ dets,
                                                 const int* __restrict__ worklist, uint8_t* __restrict__ scratch,
                                                 const FrameWin* __restrict__ wins) {
  const FrameGeom& g = gslot;  // slab layout and flag indexing: the slot; rows / cols of a frame: its window (gl below)
  __shared__ int s_nkept, s_over, s_nband, s_nrun;
  __shared__ int s_taps[MPE_MAX_KSIZE];
  __shared__ u64 s_rows[128];
  u64* const s_rowact = s_rows;     // rows that hold a pixel / that touch the row above: bit y of 64 words each
  u64* const s_link = s_rows + 64;
  u64* const s_cor = s_rows;        // later (the row bitsets are dead by then): the occupied columns of 2 - 4 bands
  __shared__ short s_blo[K1B_GEN_BANDS], s_bhi[K1B_GEN_BANDS];
  __shared__ __attribute__((aligned(8))) short s_run4[4 * K1B_GEN_RUNS];  // the items: first row, rows, first / last bit column
  short* const s_rlo = s_run4;
  short* const s_rh = s_run4 + K1B_GEN_RUNS;
  short* const s_rx0 = s_run4 + 2 * K1B_GEN_RUNS;
  short* const s_rx1 = s_run4 + 3 * K1B_GEN_RUNS;
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
  unsigned* items = kkey + K1B_GEN_KEPT;
  unsigned* const s_seg = reinterpret_cast<unsigned*>(s_run4);  // (before the bands exist: 32-bit lists in the items' LDS)
  const int seg_cap = 2 * K1B_GEN_RUNS;
  const int r = dp.ksize / 2;
  const int dc = (r + 15) / 16;
  const int spr = g.segs_per_row;
  const unsigned add = (unsigned)(255 - dp.thr) * 0x00010001u;

#ifdef K1B_GEN_CLOCKS  // (experiment builds: the shader clock at the phase boundaries of block 0's first frame, printed)
  unsigned long long gclk[9];
#define K1B_GEN_STAMP(i) gclk[i] = __builtin_amdgcn_s_memtime();
#else
#define K1B_GEN_STAMP(i)
#endif
  for (int wi = blockIdx.x; wi < count; wi += gridDim.x) {
    K1B_GEN_STAMP(0)
    const int f = worklist[1 + wi] & 0xFFFFFF;
    const uint8_t* frame = frames + (size_t)f * g.rows * g.pitch;
    int roi_x, roi_y;
    const FrameGeom gl = window_geom(gslot, wins, f, dp, roi_x, roi_y);
    if (lane == 0) {
      s_nkept = 0;
      s_over = 0;
      s_nband = 0;
      s_nrun = 0;
    }
    s_rowact[lane] = 0;
    s_link[lane] = 0;
    for (size_t i = lane; i < bm_words; i += 64) {
      nz[i] = 0;
      pm[i] = 0;
      ng[i] = 0;
    }
    for (size_t i = lane; i < (size_t)g.rows * g.tw; i += 64) todo[i] = 0;
    __threadfence_block();
    __syncthreads();
    K1B_GEN_STAMP(1)
    // todo segments: neighbourhood of every bright segment
    {
      const size_t G0 = (size_t)f * g.segs_per_frame;
      const int nwin = (g.segs_per_frame + 63) >> 6;
      const size_t w0 = G0 >> 6;
      const int sh = (int)(G0 & 63);
      auto flag_word = [&](int i) -> u64 {  // the frame's flag bits 64 i .. 64 i + 63
        if (i >= nwin) return 0;
        const u64 a = flags[w0 + i], b = flags[w0 + i + 1];
        u64 v = sh ? ((a >> sh) | (b << (64 - sh))) : a;
        const int rem = g.segs_per_frame - i * 64;
        if (rem < 64) v &= (1ull << rem) - 1;
        return v;
      };
      auto mark = [&](int s) {
        const int y0 = s / spr, c0 = s - y0 * spr;
        int cl = max(0, c0 - dc), ch = min(spr - 1, c0 + dc);
        k1b_gen_narrow(frame, g.pitch, gl.cols, y0, c0, r, dc, add, cl, ch);
        for (int yy = max(0, y0 - r); yy <= min(gl.rows - 1, y0 + r); ++yy)
          for (int cc = cl; cc <= ch; ++cc)
            atomicOr(&todo[(size_t)yy * g.tw + (cc >> 6)], 1ull << (cc & 63));
      };
      // the bright segments as a LIST first (in the run lists' LDS, unused until the bands exist), then a lane per segment:
      // a lane per flag word walked its bits one dependent pixel read (k1b_gen_narrow) after the other — three of them in
      // the busiest of every 64 words of a salt frame, six rounds of words: ~18 round trips to cold pixels where the list
      // needs three.  Four flag words in flight per lane.
      for (int i0 = 0; i0 < nwin; i0 += 256) {
        u64 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = flag_word(i0 + 64 * u + lane);
#pragma unroll
        for (int u = 0; u < 4; ++u)
          while (v[u]) {
            const int s = (i0 + 64 * u + lane) * 64 + __builtin_ctzll(v[u]);
            v[u] &= v[u] - 1;
            const int at = atomicAdd(&s_nrun, 1);
            if (at < seg_cap) s_seg[at] = (unsigned)s;
          }
      }
      __syncthreads();
      const int nseg = s_nrun;  // (uniform)
      if (nseg <= seg_cap) {
        for (int i = lane; i < nseg; i += 64) mark((int)s_seg[i]);
      } else {  // (more bright segments than the list holds: a lane per flag word)
        for (int i = lane; i < nwin; i += 64) {
          u64 v = flag_word(i);
          while (v) {
            mark(i * 64 + __builtin_ctzll(v));
            v &= v - 1;
          }
        }
      }
    }
    __threadfence_block();
    __syncthreads();
    K1B_GEN_STAMP(2)
    // the to-do bits as a LIST of (row, segment column) items, so that the blur below runs them 64 at a time whatever
    // their rows (one lane per row ran as many rounds as the busiest row of every 64 has items)
    int n_items = 0;  // (uniform)
    for (int y0 = 0; y0 < gl.rows; y0 += 64) {
      const int y = y0 + lane;
      int cnt = 0;
      if (y < gl.rows)
        for (int tw = 0; tw < g.tw; ++tw) cnt += __builtin_popcountll(todo[(size_t)y * g.tw + tw]);
      int inc = cnt;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int a = __shfl_up(inc, d);
        if (lane >= d) inc += a;
      }
      int o = n_items + inc - cnt;
      if (y < gl.rows)
        for (int tw = 0; tw < g.tw; ++tw) {
          u64 tb = todo[(size_t)y * g.tw + tw];
          while (tb) {
            items[o++] = ((unsigned)y << 12) | (unsigned)(tw * 64 + __builtin_ctzll(tb));
            tb &= tb - 1;
          }
        }
      n_items += __shfl(inc, 63);
    }
    __threadfence_block();
    __syncthreads();
    K1B_GEN_STAMP(3)
    // blur: the frame's own bytes — only the segments the image pass flagged — thresholded on the fly
    const PixWin pw = {frame, 0, gl.rows, 0, g.pitch, add, flags, (size_t)f * g.segs_per_frame};
    if (lane == 0) s_nrun = 0;  // (the bright segments' list is used up: the same LDS now lists the bitmap words written)
    __syncthreads();
    for (int i = lane; i < n_items; i += 64) {
      const unsigned it = items[i];
      const int y = (int)(it >> 12), c = (int)(it & 0xFFFu);
      u64* nzrow = nz + (size_t)(y + 1) * g.wb;
      auto note = [&](int wi) {
        const int at = atomicAdd(&s_nrun, 1);
        if (at < seg_cap) s_seg[at] = ((unsigned)y << 8) | (unsigned)wi;
      };
      if (add)
        blur_to_bitmap<true>(pw, gl.rows, gl.cols, dp, s_taps, y, c, nzrow, 0, note);
      else  // (thr = 255: nothing passes the threshold — the flags say so already, no item arrives here)
        blur_to_bitmap<false>(pw, gl.rows, gl.cols, dp, s_taps, y, c, nzrow, 0, note);
    }
    __threadfence_block();
    __syncthreads();
    K1B_GEN_STAMP(4)
    // which rows hold a pixel, and which of them touch the row above (k1b_rows_touch, word by word): the bitmap read as
    // ONE array, consecutive lanes consecutive words, four loads in flight.  (A lane per row read 64 different cache
    // lines with every instruction; the kernel is bound by exactly that — ~4 000 memory instructions per frame, most of
    // them 64 lanes to 64 lines, one every ~70 cycles per CU with 12 waves resident: profiles/round6_exp_general_tier.txt 8.)
    const int nnz = s_nrun;  // (uniform) bitmap words the blur wrote bits into, as (row << 8 | word), with repetitions
    if (nnz <= seg_cap) {
      // ... from that list, two entries (eight loads) in flight per lane (a frame's words number a few hundred; reading
      // the whole bitmap for them — 90 wave-loads, then the rows above — was 46 dependent round trips of a salt frame's ~380)
      for (int i0 = 0; i0 < nnz; i0 += 128) {
        u64 cur[2], p0[2], pl[2], pr[2];
        int yy[2], ww[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int i = i0 + 64 * u + lane;
          const unsigned e = i < nnz ? s_seg[i] : 0u;
          yy[u] = (int)(e >> 8);
          ww[u] = (int)(e & 255u);
          const u64* prev = nz + (size_t)yy[u] * g.wb;  // (slot y holds row y - 1; slot 0 is the empty separator above row 0)
          const bool on = i < nnz;
          cur[u] = on ? prev[g.wb + ww[u]] : 0;
          p0[u] = on ? prev[ww[u]] : 0;
          pl[u] = (on && ww[u] > 0) ? prev[ww[u] - 1] : 0;
          pr[u] = (on && ww[u] + 1 < g.wb) ? prev[ww[u] + 1] : 0;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (cur[u]) {
            atomicOr(&s_rowact[yy[u] >> 6], 1ull << (yy[u] & 63));
            const u64 dil = p0[u] | (p0[u] << 1) | (p0[u] >> 1) | (pl[u] >> 63) | (pr[u] << 63);
            if (cur[u] & dil) atomicOr(&s_link[yy[u] >> 6], 1ull << (yy[u] & 63));
          }
      }
    } else {
      const int nw = gl.rows * g.wb;
      for (int i0 = 0; i0 < nw; i0 += 256) {
        u64 cur[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + 64 * u + lane;
          cur[u] = i < nw ? nz[(size_t)g.wb + i] : 0;  // (slot y + 1 holds row y; slot 0 is the empty separator above row 0)
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (cur[u]) {
            const int i = i0 + 64 * u + lane;
            const int y = i / g.wb, w = i - y * g.wb;
            atomicOr(&s_rowact[y >> 6], 1ull << (y & 63));
            const u64* prev = nz + (size_t)y * g.wb;
            const u64 p = prev[w];
            u64 dil = p | (p << 1) | (p >> 1);
            if (w > 0) dil |= prev[w - 1] >> 63;
            if (w + 1 < g.wb) dil |= prev[w + 1] << 63;
            if (cur[u] & dil) atomicOr(&s_link[y >> 6], 1ull << (y & 63));
          }
      }
    }
    __syncthreads();
    K1B_GEN_STAMP(5)
    // bands = maximal runs of non-empty rows in which every row TOUCHES the one above it (k1b_rows_touch): a band
    // starts at a non-empty row that is not linked to its predecessor, and ends in front of the next such row or empty
    // row (lane w owns word w of the row bitsets; ranks by a wave prefix sum)
    {
      const int rw = (gl.rows + 63) >> 6;
      const u64 act = (lane < rw) ? s_rowact[lane] : 0;
      const u64 lnk = (lane < rw) ? s_link[lane] : 0;
      const u64 nextl = (lane + 1 < rw) ? s_link[lane + 1] : 0;
      u64 st = act & ~lnk;                              // (a linked row's predecessor is non-empty)
      u64 en = act & ~((lnk >> 1) | (nextl << 63));     // the row below does not continue the band
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
    // (appended by up to 64 lanes at once: beyond K1B_GEN_KEPT blobs the kept SUBSET depends on the order the lanes
    //  arrive in; the frame then carries MPE_FRAME_TOO_MANY_DETECTIONS anyway and include/mpe.h says so)
    auto keep = [&](float mcx, float mcy, unsigned key) {
      const int k = atomicAdd(&s_nkept, 1);
      if (k < K1B_GEN_KEPT) {
        kx[k] = mcx;
        ky[k] = mcy;
        kkey[k] = key;
      }
    };
    K1B_GEN_STAMP(6)
    // Round 6: a band is cut again at its EMPTY COLUMNS — (band, run of occupied pixel columns) items, one lane each
    // (scan_window<true>).  One lane per band left 35 of 64 lanes busy on a salt-noise frame (~180 blobs) and the
    // longest band set the time: 47 % of the frame's cycles, every step a dependent read of the bitmaps.
#ifdef K1B_GEN_CLOCKS
    unsigned long long gclk_build = 0, gclk_t = 0;
    int gclk_items = 0;
#endif
    if (nband <= K1B_GEN_BANDS && g.wb <= 64) {
      // The bands are taken in BATCHES: runs are built until the list is half full (the pieces are appended behind them),
      // cut, scanned, and the list starts again — items are independent and the kept blobs are put into raster order at
      // the end, so a frame has as many items as it needs (a 1920-wide frame with 0.05 % noise: ~1 100; one list held
      // 736 and a frame beyond that fell back to a lane per band after building its runs for nothing).
      const int lw = g.wb <= 16 ? 4 : (g.wb <= 32 ? 5 : 6);  // (uniform) log2 of the lanes per bitmap row
      const int cw = lane & ((1 << lw) - 1), cj = lane >> lw, rg = 64 >> lw;
      const int nb = lw == 4 ? 4 : 2;  // bands in flight: their words share the 128 words of the row bitsets
      int b_next = 0;
      while (b_next < nband) {  // (uniform)
#ifdef K1B_GEN_CLOCKS
        gclk_t = __builtin_amdgcn_s_memtime();
#endif
        const int b_first = b_next;
        if (lane == 0) s_nrun = 0;
        __syncthreads();
        // the column occupancy of a band by the WHOLE wave, four bands in flight (two if a bitmap row has more than 16
        // words): with 16 / 32 / 64 lanes per row, lane (j, w) ORs word w of the band's rows j, j + 4 / 2 / 1, ...
        // (consecutive lanes consecutive words: a lane per band read its rows x words one scattered load after the
        // other, the tallest band setting the time — 1 200 of the frame's 4 000 memory instructions); then lanes 0 .. 3
        // cut their band's words into runs (column_runs_of_words).
        while (b_next < nband) {
          const int b0 = b_next;
          int blo[4], bh[4], hmax = 0;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int b = b0 + u;
            const bool on = u < nb && b < nband;
            blo[u] = on ? s_blo[b] : 0;
            bh[u] = on ? s_bhi[b] - blo[u] + 1 : 0;
            hmax = max(hmax, bh[u]);
          }
          u64 acc[4] = {0, 0, 0, 0};
          if (cw < g.wb)
            for (int kk = cj; kk < hmax; kk += 2 * rg) {  // (eight loads in flight)
              u64 t[2][4];
#pragma unroll
              for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                  const int k2 = kk + v * rg;
                  t[v][u] = k2 < bh[u] ? nz[(size_t)(blo[u] + 1 + k2) * g.wb + cw] : 0;
                }
#pragma unroll
              for (int u = 0; u < 4; ++u) acc[u] |= t[0][u] | t[1][u];
            }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (lw <= 4) acc[u] |= __shfl_xor(acc[u], 16);
            if (lw <= 5) acc[u] |= __shfl_xor(acc[u], 32);
            if (u < nb && lane < (1 << lw)) s_cor[(u << lw) + lane] = acc[u];
          }
          __syncthreads();
          if (lane < nb && b0 + lane < nband) {
            const int lo = s_blo[b0 + lane], H = s_bhi[b0 + lane] - lo + 1;
            column_runs_of_words(s_cor + (lane << lw), g.wb, [&](int x0, int x1) {
              const int i = atomicAdd(&s_nrun, 1);
              if (i < K1B_GEN_RUNS) {
                s_rlo[i] = (short)lo;
                s_rh[i] = (short)H;
                s_rx0[i] = (short)x0;
                s_rx1[i] = (short)x1;
              }
            });
          }
          __syncthreads();
          b_next += nb;
          if (s_nrun > K1B_GEN_RUNS * 3 / 8) break;  // (uniform) room for as many pieces again, and for the next bands' runs
        }
        const int nrun = s_nrun;  // (uniform)
        if (nrun > K1B_GEN_RUNS) {
          // (the batch's last bands alone overflowed the list — more than ~460 runs in four bands: a lane per band for
          //  this batch, the round-5 way; slot 0 of a band's window is the row above it — empty, or holding no neighbour
          //  of any pixel of the band — and the row below likewise: no border following leaves the band)
          const int b_end = min(b_next, nband);
          for (int c0 = b_first; c0 < b_end; c0 += 64) {  // (uniform: every lane enters scan_window, with H = 0 if it has no band)
            const int bb = c0 + lane;
            const int lo = bb < b_end ? s_blo[bb] : 0, H = bb < b_end ? s_bhi[bb] - lo + 1 : 0;
            const size_t off = (size_t)lo * g.wb;
            scan_window(nz + off, pm + off, ng + off, g.wb, H, lo, 0, dp, roi_x, roi_y, &s_over, keep);
          }
          __syncthreads();
          continue;
        }
        // ... and every run again at the rows that are EMPTY within its columns (window_run_rows; a lane per run): the
        // band's rows chain through OTHER runs — salt pixels three rows tall hang together over sixty rows — so a run
        // holds several blobs one above the other, and the longest such stack set the time of a round of 64 items
        // (~250 k cycles).  The pieces are appended behind the runs; if they do not fit, the runs themselves are scanned.
        // (done in the pass above by the band's one lane — a dependent read per row and run — the cut cost more than it
        //  saved: scan phase 0.8 -> 1.2 - 2.1 M cycles)
        for (int i0 = 0; i0 < nrun; i0 += 64) {
          const int i = i0 + lane;
          if (i < nrun) {
            const int lo = s_rlo[i], H = s_rh[i], x0 = s_rx0[i], x1 = s_rx1[i];
            window_run_rows(nz + (size_t)lo * g.wb, g.wb, H, x0, x1, [&](int first, int nrows) {
              const int j = atomicAdd(&s_nrun, 1);
              if (j < K1B_GEN_RUNS) {
                s_rlo[j] = (short)(lo + first - 1);
                s_rh[j] = (short)nrows;
                s_rx0[j] = (short)x0;
                s_rx1[j] = (short)x1;
              }
            });
          }
        }
        __syncthreads();
        const bool cut = s_nrun <= K1B_GEN_RUNS;                       // (uniform)
        const int item0 = cut ? nrun : 0, item1 = cut ? s_nrun : nrun;  // the items of this batch: [item0, item1)
#ifdef K1B_GEN_CLOCKS
        gclk_build += __builtin_amdgcn_s_memtime() - gclk_t;
        gclk_items += item1 - item0;
#endif
        for (int i0 = item0; i0 < item1; i0 += 64) {  // (uniform: every lane enters scan_window, with H = 0 if it has no item)
          const int i = i0 + lane;
          const int lo = i < item1 ? s_rlo[i] : 0, H = i < item1 ? s_rh[i] : 0;
          const size_t off = (size_t)lo * g.wb;
          scan_window<true>(nz + off, pm + off, ng + off, g.wb, H, lo, 0, dp, roi_x, roi_y, &s_over, keep,
                            i < item1 ? s_rx0[i] : 0, i < item1 ? s_rx1[i] : 0);
        }
        __syncthreads();
      }
    } else if (nband <= K1B_GEN_BANDS) {
      // (bitmap rows of more than 64 words — frames wider than 3 966 pixels: one lane per band)
      for (int b0 = 0; b0 < nband; b0 += 64) {  // (uniform: every lane enters scan_window, with H = 0 if it has no band)
        const int b = b0 + lane;
        const int lo = b < nband ? s_blo[b] : 0, H = b < nband ? s_bhi[b] - lo + 1 : 0;
        const size_t off = (size_t)lo * g.wb;
        scan_window(nz + off, pm + off, ng + off, g.wb, H, lo, 0, dp, roi_x, roi_y, &s_over, keep);
      }
    } else {  // (more bands than the list holds: the literal whole-frame scan by one lane)
      scan_window(nz, pm, ng, g.wb, lane == 0 ? gl.rows : 0, 0, 0, dp, roi_x, roi_y, &s_over, keep);
    }
    __threadfence_block();
    __syncthreads();
    K1B_GEN_STAMP(7)
    write_detections(kx, ky, kkey, s_nkept, K1B_GEN_KEPT, s_over, dp, dets + f, lane);
    __syncthreads();
    K1B_GEN_STAMP(8)
#ifdef K1B_GEN_CLOCKS
    if (blockIdx.x == 0 && lane == 0 && wi == 0)
      printf("k1b_general phases (cycles): clear %llu todo %llu items %llu (n=%d) blur %llu rows %llu bands %llu (n=%d) runs %llu (n=%d) scan %llu write %llu\n",
             gclk[1] - gclk[0], gclk[2] - gclk[1], gclk[3] - gclk[2], n_items, gclk[4] - gclk[3], gclk[5] - gclk[4],
             gclk[6] - gclk[5], nband, gclk_build, gclk_items, gclk[7] - gclk[6] - gclk_build, gclk[8] - gclk[7]);
#endif
  }
}

// ---------------------------------------------------------------------------------------------
// k1b_general_lds (round 6) — the same tier with the frame's three bitmaps, the to-do bitset and the band list in LDS:
// ONE block of four waves per CU, 139 KB of the CU's 160 KB at 752 x 480.  The phase clocks of k1b_general on a
// salt-noise frame (tools/general_tier_probe.py, -DK1B_GEN_CLOCKS: 3.4 M cycles, 47 % the contour scan, 35 % the blur)
// said why that kernel runs at a quarter of the VALU issue rate with 16 waves per CU: the scan walks its band word by
// word and follows borders pixel by pixel, every step a dependent read of bitmaps in GLOBAL memory (~1 600 cycles per
// step), and the blur's items each waited for ten round trips.  With the bitmaps in LDS a step is a ds_read; the four
// waves share the frame's blur items and its bands (a lane per band, 256 at a time); item list, flag scan, band
// building are one wave's work as before.  Frames whose bitmaps do not fit (1920 x 1200: 3 x 298 KB) keep k1b_general.
// The launch needs a CU's whole LDS per block, so it is only used once the previous call's work-list said that frames
// DO reach this tier (launch_k1b_blobs: general_lds) — an empty launch of the old kernel costs the clean step nothing.
// Same arithmetic, same decomposition, same records (tests: the general-tier suites, witness clutter vectors).
// MEASURED, NOT THE DEFAULT (option "general_lds" 0; profiles/round6_exp_general_tier.txt): a frame costs this kernel
// 1.2 M cycles where the slab kernel needs 3.4 M — but the slab kernel keeps 16 frames per CU in flight and this one ONE:
// salt noise 1.05 M -> 0.48 M fps, a saturated 64 x 64 patch 2.73 -> 2.93 M.  What is left per frame is the chain of one
// LANE walking its band (35 bands for ~180 blobs at 0.05 % salt noise: 35 of 256 lanes busy, the slowest band sets the
// time).  The next step is the VERDICT's: items of (band, run of occupied pixel columns) — exact, an empty column
// separates components as an empty row does — which needs the scan's candidate and mark words masked to the run and
// its mark updates atomic; not attempted in round 6.
// ---------------------------------------------------------------------------------------------
#define K1B_GENL_THREADS 256
__host__ __device__ inline size_t k1b_genl_lds_bytes(const FrameGeom& g) {
  const size_t bm = (size_t)(g.rows + 2) * g.wb * 8;
  const size_t todo = (size_t)g.rows * g.tw * 8;
  const size_t bands = ((size_t)2 * g.rows * sizeof(short) + 15) / 16 * 16;
  return 3 * bm + todo + bands;
}
static bool k1b_genl_fits(const FrameGeom& g) { return k1b_genl_lds_bytes(g) + 2048 <= (size_t)160 * 1024; }

__global__ __launch_bounds__(K1B_GENL_THREADS) void k1b_general_lds(const uint8_t* __restrict__ frames,
                                                                    const u64* __restrict__ flags, FrameGeom gslot,
                                                                    DetectParams dp, mpe_detections* __restrict__ dets,
                                                                    const int* __restrict__ worklist,
                                                                    uint8_t* __restrict__ scratch,
                                                                    const FrameWin* __restrict__ wins) {
  const FrameGeom& g = gslot;
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
  __shared__ int s_nkept, s_over, s_nband, s_nitems;
  __shared__ int s_taps[MPE_MAX_KSIZE];
  __shared__ u64 s_rowact[64], s_link[64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int count = worklist[0];
  if ((int)blockIdx.x >= count) return;
  if (tid < MPE_MAX_KSIZE) s_taps[tid] = dp.taps[tid < dp.ksize ? tid : 0];
  __syncthreads();
  const size_t bm_words = (size_t)(g.rows + 2) * g.wb;
  u64* nz = reinterpret_cast<u64*>(gsm);
  u64* pm = nz + bm_words;
  u64* ng = pm + bm_words;
  u64* todo = ng + bm_words;
  short* s_blo = reinterpret_cast<short*>(todo + (size_t)g.rows * g.tw);
  short* s_bhi = s_blo + g.rows;
  // the slab of k1b_general (same layout, its bitmap area unused): kept blobs and the item list stay in global memory
  const size_t slab = k1b_gen_scratch_bytes(g);
  uint8_t* base = scratch + (size_t)blockIdx.x * slab;
  float* kx = reinterpret_cast<float*>(reinterpret_cast<u64*>(base) + 3 * bm_words + (size_t)g.rows * g.tw);
  float* ky = kx + K1B_GEN_KEPT;
  unsigned* kkey = reinterpret_cast<unsigned*>(ky + K1B_GEN_KEPT);
  unsigned* items = kkey + K1B_GEN_KEPT;
  const int r = dp.ksize / 2;
  const int dc = (r + 15) / 16;
  const int spr = g.segs_per_row;
  const unsigned add = (unsigned)(255 - dp.thr) * 0x00010001u;

  for (int wi = blockIdx.x; wi < count; wi += gridDim.x) {
    const int f = worklist[1 + wi] & 0xFFFFFF;
    const uint8_t* frame = frames + (size_t)f * g.rows * g.pitch;
    int roi_x, roi_y;
    const FrameGeom gl = window_geom(gslot, wins, f, dp, roi_x, roi_y);
    if (tid == 0) {
      s_nkept = 0;
      s_over = 0;
      s_nband = 0;
      s_nitems = 0;
    }
    if (tid < 64) {
      s_rowact[tid] = 0;
      s_link[tid] = 0;
    }
    for (size_t i = tid; i < 3 * bm_words + (size_t)g.rows * g.tw; i += K1B_GENL_THREADS) nz[i] = 0;  // (nz, pm, ng, todo: contiguous)
    __syncthreads();
    // todo segments: neighbourhood of every bright segment
    {
      const size_t G0 = (size_t)f * g.segs_per_frame;
      const int nwin = (g.segs_per_frame + 63) >> 6;
      const size_t w0 = G0 >> 6;
      const int sh = (int)(G0 & 63);
      for (int i = tid; i < nwin; i += K1B_GENL_THREADS) {
        const u64 a = flags[w0 + i], b = flags[w0 + i + 1];
        u64 v = sh ? ((a >> sh) | (b << (64 - sh))) : a;
        const int rem = g.segs_per_frame - i * 64;
        if (rem < 64) v &= (1ull << rem) - 1;
        while (v) {
          const int sgm = i * 64 + __builtin_ctzll(v);
          v &= v - 1;
          const int y0 = sgm / spr, c0 = sgm - y0 * spr;
          int cl = max(0, c0 - dc), ch = min(spr - 1, c0 + dc);
          k1b_gen_narrow(frame, g.pitch, gl.cols, y0, c0, r, dc, add, cl, ch);
          for (int yy = max(0, y0 - r); yy <= min(gl.rows - 1, y0 + r); ++yy)
            for (int cc = cl; cc <= ch; ++cc)
              atomicOr(&todo[(size_t)yy * g.tw + (cc >> 6)], 1ull << (cc & 63));
        }
      }
    }
    __syncthreads();
    // the to-do bits as a LIST of (row, segment column) items (wave 0: ranks by a wave prefix sum, 64 rows at a time)
    if (wv == 0) {
      int n_items = 0;  // (uniform)
      for (int y0 = 0; y0 < gl.rows; y0 += 64) {
        const int y = y0 + lane;
        int cnt = 0;
        if (y < gl.rows)
          for (int tw = 0; tw < g.tw; ++tw) cnt += __builtin_popcountll(todo[(size_t)y * g.tw + tw]);
        int inc = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int a = __shfl_up(inc, d);
          if (lane >= d) inc += a;
        }
        int o = n_items + inc - cnt;
        if (y < gl.rows)
          for (int tw = 0; tw < g.tw; ++tw) {
            u64 tb = todo[(size_t)y * g.tw + tw];
            while (tb) {
              items[o++] = ((unsigned)y << 12) | (unsigned)(tw * 64 + __builtin_ctzll(tb));
              tb &= tb - 1;
            }
          }
        n_items += __shfl(inc, 63);
      }
      if (lane == 0) s_nitems = n_items;
    }
    __threadfence_block();
    __syncthreads();
    const int n_items = s_nitems;
    // blur: the frame's own bytes — only the segments the image pass flagged — thresholded on the fly
    const PixWin pw = {frame, 0, gl.rows, 0, g.pitch, add, flags, (size_t)f * g.segs_per_frame};
    for (int i = tid; i < n_items; i += K1B_GENL_THREADS) {
      const unsigned it = items[i];
      const int y = (int)(it >> 12), c = (int)(it & 0xFFFu);
      u64* nzrow = nz + (size_t)(y + 1) * g.wb;
      if (add)
        blur_to_bitmap<true>(pw, gl.rows, gl.cols, dp, s_taps, y, c, nzrow, 0);
      else
        blur_to_bitmap<false>(pw, gl.rows, gl.cols, dp, s_taps, y, c, nzrow, 0);
    }
    __syncthreads();
    for (int y = tid; y < gl.rows; y += K1B_GENL_THREADS) {
      const u64* nzrow = nz + (size_t)(y + 1) * g.wb;
      u64 any = 0;
      for (int w = 0; w < g.wb; ++w) any |= nzrow[w];
      if (any) {
        atomicOr(&s_rowact[y >> 6], 1ull << (y & 63));
        if (k1b_rows_touch(nzrow, nzrow - g.wb, g.wb)) atomicOr(&s_link[y >> 6], 1ull << (y & 63));
      }
    }
    __syncthreads();
    // bands (see k1b_general): wave 0, lane w owns word w of the row bitsets
    if (wv == 0) {
      const int rw = (gl.rows + 63) >> 6;
      const u64 act = (lane < rw) ? s_rowact[lane] : 0;
      const u64 lnk = (lane < rw) ? s_link[lane] : 0;
      const u64 nextl = (lane + 1 < rw) ? s_link[lane + 1] : 0;
      u64 st = act & ~lnk;
      u64 en = act & ~((lnk >> 1) | (nextl << 63));
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
      while (st) {  // (a band has at least one row: never more bands than rows, the lists' capacity)
        const int b = __builtin_ctzll(st);
        st &= st - 1;
        s_blo[is_++] = (short)(lane * 64 + b);
      }
      while (en) {
        const int b = __builtin_ctzll(en);
        en &= en - 1;
        s_bhi[ie++] = (short)(lane * 64 + b);
      }
      if (lane == 63) s_nband = ps;
    }
    __syncthreads();
    const int nband = s_nband;
    auto keep = [&](float mcx, float mcy, unsigned key) {  // (see k1b_general on the subset kept beyond K1B_GEN_KEPT)
      const int k = atomicAdd(&s_nkept, 1);
      if (k < K1B_GEN_KEPT) {
        kx[k] = mcx;
        ky[k] = mcy;
        kkey[k] = key;
      }
    };
    // one lane per band, 256 bands at a time (every lane enters scan_window, with H = 0 if it has no band)
    for (int b0 = 0; b0 < nband; b0 += K1B_GENL_THREADS) {
      const int b = b0 + tid;
      const int lo = b < nband ? s_blo[b] : 0, H = b < nband ? s_bhi[b] - lo + 1 : 0;
      const size_t off = (size_t)lo * g.wb;
      scan_window(nz + off, pm + off, ng + off, g.wb, H, lo, 0, dp, roi_x, roi_y, &s_over, keep);
    }
    __threadfence_block();
    __syncthreads();
    if (wv == 0) write_detections(kx, ky, kkey, s_nkept, K1B_GEN_KEPT, s_over, dp, dets + f, lane);
    __syncthreads();
  }
}

size_t k1b_scratch_bytes(const FrameGeom& g, int n_frames) {
  return k1b_gen_scratch_bytes(g) * (size_t)k1b_gen_blocks(g, n_frames);
}

hipError_t launch_k1b_blobs(const uint8_t* frames, const unsigned long long* flags, int n_frames, const FrameGeom& g,
                            const DetectParams& dp, mpe_detections* dets, int* worklist, uint8_t* scratch,
                            size_t scratch_bytes, int blob_hint, hipStream_t s, const void* frame_windows,
                            bool lists_zeroed,
                            bool first_tier_only, bool general_lds) {
  const FrameWin* wins = static_cast<const FrameWin*>(frame_windows);
  if (first_tier_only) {
    // low-latency tracked frame: the small tier alone, nothing queued behind it.  A frame that overflows it is left
    // with status MPE_FRAME_TOO_MANY_ROWS and no work-list entry; the caller sees that in the record it fetches
    // anyway and repeats the frame through the whole chain (three launches and a memset less on every other frame)
    if (n_frames <= 0) return hipSuccess;
    if (blob_hint <= 0 || blob_hint > 8) return hipErrorInvalidValue;
    const int blocks = (n_frames + K1bSmall::WAVES - 1) / K1bSmall::WAVES;
    if (K1bSmall::WAVES == 1 && n_frames <= K1B_HELP_MAX_FRAMES)
      hipLaunchKernelGGL((k1b_blobs_few<K1bSmall>), dim3(n_frames), dim3(K1B_HELP_THREADS), 0, s, frames, (const u64*)flags, g,
                         dp, dets, (int*)nullptr, wins);
    else
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
    if (K1bSmall::WAVES == 1 && n_frames <= K1B_HELP_MAX_FRAMES)
      hipLaunchKernelGGL((k1b_blobs_few<K1bSmall>), dim3(n_frames), dim3(K1B_HELP_THREADS), 0, s, frames, (const u64*)flags, g, dp,
                         dets, list_a, wins);
    else
      hipLaunchKernelGGL((k1b_blobs<K1bSmall>), dim3(blocks), dim3(64 * K1bSmall::WAVES), 0, s, frames, (const u64*)flags, g, dp, dets,
                         list_a, n_frames, wins);
    const int grid = n_frames < 2048 ? n_frames : 2048;
    if (n_frames <= K1B_HELP_MAX_FRAMES)
      hipLaunchKernelGGL((k1b_blobs_list_few<K1bLarge>), dim3(grid), dim3(K1B_HELP_THREADS), 0, s, frames, (const u64*)flags, g,
                         dp, dets, (const int*)list_a, list_b, wins);
    else
      hipLaunchKernelGGL((k1b_blobs_list<K1bLarge>), dim3(grid), dim3(64), 0, s, frames, (const u64*)flags, g, dp,
                         dets, (const int*)list_a, list_b, wins);
  } else if (n_frames <= K1B_HELP_MAX_FRAMES) {
    hipLaunchKernelGGL((k1b_blobs_few<K1bLarge>), dim3(n_frames), dim3(K1B_HELP_THREADS), 0, s, frames, (const u64*)flags, g, dp,
                       dets, list_b, wins);
  } else {
    hipLaunchKernelGGL((k1b_blobs<K1bLarge>), dim3(n_frames), dim3(64), 0, s, frames, (const u64*)flags, g, dp,
                       dets, list_b, n_frames, wins);
  }
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  // one slab per block: the grid follows the scratch the CALLER reserved (the process-wide knob may have moved since)
  size_t gen_blocks = (size_t)k1b_gen_blocks(g, n_frames);
  const size_t slabs = scratch_bytes / k1b_gen_scratch_bytes(g);
  if (slabs < 1) return hipErrorInvalidValue;
  if (gen_blocks > slabs) gen_blocks = slabs;
  if (general_lds && k1b_genl_fits(g)) {
    // frames DO reach this tier (the caller's reading of the previous call's work-list): the LDS-resident kernel, a
    // block of four waves per CU with the frame's bitmaps in the CU's LDS
    const size_t lds = k1b_genl_lds_bytes(g);
    static size_t attr_set = 0;  // (the opt-in above 64 KB of dynamic LDS, once per size)
    if (lds > attr_set) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(k1b_general_lds), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
      if (e != hipSuccess) return e;
      attr_set = lds;
    }
    const size_t cus = (size_t)device_cu_count();
    if (gen_blocks > cus) gen_blocks = cus;
    hipLaunchKernelGGL(k1b_general_lds, dim3((unsigned)gen_blocks), dim3(K1B_GENL_THREADS), lds, s, frames, (const u64*)flags,
                       g, dp, dets, (const int*)list_b, scratch, wins);
    return hipGetLastError();
  }
  // ... and no more blocks than the device holds at once: a block walks the work-list with the grid as its stride, so the
  // blocks of a second round would start when the first round's have done ALL their frames (4 096 blocks on the 3 072
  // resident ones of 256 CUs: the last quarter of a 16 384-frame list ran on a quarter of the machine, 7.9 ms where
  // 5.3 rounds of frames need ~4)
  static int resident = 0;
  if (!resident) {
    int per_cu = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k1b_general), 64, 0);
    if (e != hipSuccess) return e;
    resident = (per_cu > 0 ? per_cu : 1) * device_cu_count();
  }
  static const bool full_grid = getenv("MPE_K1B_GEN_FULL_GRID") != nullptr;  // (A/B of the cap: round6_exp_general_tier.txt 8)
  if (!full_grid && gen_blocks > (size_t)resident) gen_blocks = (size_t)resident;
  hipLaunchKernelGGL(k1b_general, dim3((unsigned)gen_blocks), dim3(64), 0, s, frames, (const u64*)flags, g, dp, dets,
                     (const int*)list_b, scratch, wins);
  return hipGetLastError();
}

//@file-epilogue
}  // namespace mpe
//@file-epilogue-end
