//@file-prologue
// mpe_k2.hip — K2 brute-force correspondence voting (tables, fast / strict / fix-up kernels) (see mpe_kernels_common.h for the map of the kernel sources)
#include "mpe_kernels_common.h"
#include "mpe_k2_head.h"
namespace mpe {
//@file-prologue-end
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
// Scan-carrying variant with many detections (CG, round 6).  With 21 detections the per-root prefilter — a single-
// precision distance per unused detection and marker, 8 instructions each, 144 per root — was a quarter of the kernel.
// A bit grid as above does not help here: the unused markers of a 5-LED rig back-project NEAR the triple's own
// detections, where the clutter is, and every fifth root hits a set cell (measured: the wave's queue overflowed into
// the strict list, 1.6 M -> 8.3 M entries per step).  Instead each cell of a coarse grid (32 x 32 over the detections'
// bounding box, 4 KB of LDS) holds the MASK of the detections whose prefilter disc can reach it: one lookup per marker
// gives the handful of candidates, and only those get the distance test — the same `pass` mask as the loop over all
// detections, bit for bit (the candidates are a superset of the detections within the prefilter radius).
#define K2_CG_MIN_DETECTIONS 9  // the launcher picks the variant from its detection-count hint
#define K2_CGRID 32
__device__ __forceinline__ unsigned k2_cgrid_candidates(const unsigned* cmask, float fx, float fy) {
  const unsigned ix = min(k2_cvt_pk_u8(fx, 0u, 0u), (unsigned)(K2_CGRID - 1));  // (saturating conversions: NaN, negative
  const unsigned iy = min(k2_cvt_pk_u8(fy, 0u, 0u), (unsigned)(K2_CGRID - 1));  //  -> cell 0, large -> the last: empty)
  return cmask[iy * K2_CGRID + ix];
}
// cells 2 .. K2_CGRID - 3 span the detections' bounding box + 2 R; a coordinate c lands in cell floor(c) or floor(c) + 1
// whatever the conversion's rounding, so detection a (centre (cx, cy), radius r, in cells) is entered in the cells
// floor(c - r) .. floor(c + r) + 1 of both axes (the square around its disc); cells 0 and K2_CGRID - 1 stay empty.
// All threads; cmask must be zero; gp = {ginv, gxo, gyo}: cell coordinates of a pixel (u, v) = (u ginv + gxo, v ginv + gyo)
__device__ __forceinline__ void k2_cgrid_build(const double (*px)[2], int n_d, double back_tol, unsigned* cmask, float* gp,
                                               int tid, int nthr) {
  const float R = (float)((back_tol * (1.0 + 1e-4) + 0.25) * 1.0001 + 1e-3);
  float x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
  for (int a = 0; a < n_d; ++a) {
    const float u = (float)px[a][0], v = (float)px[a][1];
    x0 = fminf(x0, u);
    x1 = fmaxf(x1, u);
    y0 = fminf(y0, v);
    y1 = fmaxf(y1, v);
  }
  const float span = fmaxf(x1 - x0, y1 - y0) + 2.0f * R;
  const float cell = fmaxf(span * (1.0f / (K2_CGRID - 6)), 0.25f);
  const float inv = 1.0f / cell;
  const float ox = 2.0f - (x0 - R) * inv, oy = 2.0f - (y0 - R) * inv;
  if (tid == 0) {
    gp[0] = inv;
    gp[1] = ox;
    gp[2] = oy;
  }
  const float r = R * inv + 1e-3f;
  for (int a = tid; a < n_d; a += nthr) {
    const float u = (float)px[a][0], v = (float)px[a][1];
    if (!(u == u && v == v)) continue;
    const float cxg = u * inv + ox, cyg = v * inv + oy;  // (>= 2: truncation is floor)
    const int ix0 = max(1, (int)(cxg - r)), ix1 = min(K2_CGRID - 2, (int)(cxg + r) + 1);
    const int iy0 = max(1, (int)(cyg - r)), iy1 = min(K2_CGRID - 2, (int)(cyg + r) + 1);
    for (int iy = iy0; iy <= iy1; ++iy)
      for (int ix = ix0; ix <= ix1; ++ix) atomicOr(&cmask[iy * K2_CGRID + ix], 1u << a);
  }
}
// the block's grid: cells that a point within R of a detection can land in.  All threads; the grid must be zero; the
// parameters are returned through gp = {ginv, gxo, gyo}: grid coordinates of a pixel (u, v) = (u ginv + gxo, v ginv +
// gyo).  A coordinate c lands in cell floor(c) or floor(c) + 1 whatever the conversion's rounding, i.e. cell X takes
// points with c in [X - 1, X + 1): row Y of a detection's disc (centre (cx, cy), radius r, in cells) is set from
// floor(cx - hw) to floor(cx + hw) + 1, hw the disc's half-width over y in [Y - 1, Y + 1).  The single-precision chain
// that produces c is off by < 0.01 px (tests/test_vote_host.py; margin in R: 0.25 px).  248 cells span the detections'
// bounding box + 2 R, from cell 3 on, so that cells 0 and 255 — where everything outside the grid lands — stay empty.
template <int DIM = K2_GRID>
__device__ __forceinline__ void k2_grid_build(const double (*px)[2], int n_d, double back_tol, u64* grid, float* gp, int tid,
                                              int nthr) {
  constexpr int WORDS = DIM / 64;
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
  const float cell = fmaxf(span * (1.0f / (DIM - 8)), 0.25f);
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
    const int iy0 = max(1, (int)(cyg - r)), iy1 = min(DIM - 2, (int)(cyg + r) + 1);
    for (int iy = iy0; iy <= iy1; ++iy) {
      const float dy = fmaxf(0.f, fmaxf((float)(iy - 1) - cyg, cyg - (float)(iy + 1)));
      const float hh = r * r - dy * dy;
      if (!(hh >= 0.f)) continue;
      const float hw = sqrtf(hh);
      const int ix0 = max(1, (int)(cxg - hw)), ix1 = min(DIM - 2, (int)(cxg + hw) + 1);
      for (int w = ix0 >> 6; w <= (ix1 >> 6); ++w) {
        const int lo = max(ix0, 64 * w) - 64 * w, hi = min(ix1, 64 * w + 63) - 64 * w;
        const u64 m = (hi == 63 ? ~0ull : ((1ull << (hi + 1)) - 1)) & ~((1ull << lo) - 1);
        atomicOr(&grid[iy * WORDS + w], m);
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
template <bool SCAN, int NP = 0, bool CG = false, class Rider>
__device__ __forceinline__ void k2_vote_item(const K2Frame& F, int ti, int pj, bool live, Rider& rider, int& vq_count) {
  // (Round 6, measured and NOT kept: `#pragma clang fp contract(fast)` in this item and in solve_quartic_lit2 — the
  //  VERDICT's "253 avoidable instructions" — fuses 45 of them (one per product chain: 2 962 -> 2 927 VALU in the C3
  //  listing), moves no rate, and breaks default == strict on 25 of 65 536 C3 frames: the coefficients F0 .. F4 are the
  //  one place where the fast and the strict arithmetic agree BIT FOR BIT today, which is why the suspect screen only
  //  has to watch Ferrari's own cancellations.  profiles/round6_parity_soak_votes_contract_fast.json)
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
      unsigned pass;
      if constexpr (CG) {
        // many detections: the candidates of the (<= 2) back-projections' cells, then the same distance test as below
        // for those alone (F.grid / ginv / gxo / gyo describe the candidate grid in this variant)
        const unsigned* cm = reinterpret_cast<const unsigned*>(F.grid);
        unsigned cand = k2_cgrid_candidates(cm, __builtin_fmaf(uv0.x, F.ginv, F.gxo), __builtin_fmaf(uv0.y, F.ginv, F.gyo));
        if (F.nuo > 1)  // (uniform)
          cand |= k2_cgrid_candidates(cm, __builtin_fmaf(uv1.x, F.ginv, F.gxo), __builtin_fmaf(uv1.y, F.ginv, F.gyo));
        pass = 0u;
        for (unsigned m = cand & unused; m;) {
          const unsigned lsb = m & (0u - m);
          const int a = __builtin_ctz(m);
          m ^= lsb;
          pass |= near(F.pxf[a]) ? lsb : 0u;
        }
        if (!chain_ok) pass = unused;  // (the double-precision evaluation decides)
      } else {
        pass = (near(af0) ? lsb0 : 0u) | (near(af1) ? lsb1 : 0u);
        for (unsigned m = rest; m;) {  // (more than five detections)
          const unsigned lsb = m & (0u - m);
          const int a = __builtin_ctz(m);
          m ^= lsb;
          pass |= near(F.pxf[a]) ? lsb : 0u;
        }
        if (!chain_ok) pass = unused;  // (the double-precision evaluation decides)
      }
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
#define K2_FAST_HIST (MPE_FAST_VOTE_DETECTIONS * MPE_MAX_MARKERS)  // histogram rows a fast voting block keeps in LDS
// CG (scan-carrying variant only): frames with many detections — a 128 x 128-bit occupancy grid of the detections decides
// per root whether any unused detection can be near a back-projection (see K2_CGRID)
template <bool SCAN, bool RANGE = false, int NP = 0, bool CG = false>
__global__ __launch_bounds__(K2_THREADS, K2_MIN_WAVES(NP)) void k2_vote(mpe_detections* __restrict__ dets, SolveParams sp,
                                                      const double* __restrict__ tab, uint32_t* __restrict__ hist,
                                                      int splits, ScanArgs scan, const int* __restrict__ item_range,
                                                      int slice_tab, VoteFixup fixup) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ double s_px[MPE_FAST_VOTE_DETECTIONS][2];  // (wider frames leave below)
  __shared__ double s_iv[MPE_FAST_VOTE_DETECTIONS][3];
  // detection triples staged per pass: 64, or 16 in the scan-carrying variant (LDS goes to the scan staging
  // and to an LDS copy of the marker table instead; more triples simply take more passes)
  constexpr int TRI = SCAN ? K2_TRI_CHUNK_SCAN : K2_TRI_CHUNK;
  __shared__ double s_tri[TRI][13];  // T rows (9), f_1, f_2, b, f_1/f_2
  __shared__ unsigned s_trii[TRI];   // c0 | c1 << 8 | c2 << 16 | swap << 24
  __shared__ unsigned s_hist[K2_FAST_HIST];
  __shared__ f32x2 s_pxf[MPE_FAST_VOTE_DETECTIONS];  // the detections in single precision (nearest-neighbour prefilter)
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
  // More detections than this kernel's 32-bit detection masks and 5-bit suspect codes hold (33 .. MPE_MAX_DETECTIONS):
  // the frame is left to the strict loop nest (k2_vote_relost, launched behind the fix-up kernel), whose histogram the
  // fast arithmetic has to equal anyway.  The mark is the one of a frame that lost a suspect entry; ctl[7] counts.
  if (n_d > MPE_FAST_VOTE_DETECTIONS) {
    if (part == 0) {  // (the re-vote ADDS its blocks' votes to these rows)
      uint32_t* gw = hist + (size_t)f * MPE_HIST_STRIDE;
      for (int i = tid; i < n_d * MPE_MAX_MARKERS; i += nthr) gw[i] = 0;
      if (tid == 0) {
        atomicAdd(&fixup.ctl[7], 1u);
        d->status = MPE_FRAME_VOTE_LIST_FULL;
      }
    }
    rider.drain();
    return;
  }

  for (int i = tid; i < K2_FAST_HIST; i += nthr) s_hist[i] = 0;
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
  __shared__ unsigned s_cmask[CG ? K2_CGRID * K2_CGRID : 2];
  __shared__ float s_cgp[4];
  if constexpr (CG) {
    for (int i = tid; i < K2_CGRID * K2_CGRID; i += nthr) s_cmask[i] = 0;
    __syncthreads();
    k2_cgrid_build(s_px, n_d, sp.back_tol, s_cmask, s_cgp, tid, nthr);
    __syncthreads();
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
                     CG ? reinterpret_cast<u64*>(s_cmask) : s_grid, CG ? s_cgp[0] : s_gp[0], CG ? s_cgp[1] : s_gp[1], CG ? s_cgp[2] : s_gp[2],
                     s_trif};
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
      k2_vote_item<SCAN, NP, CG>(F, ti, p_lo + pj, live, rider, vq_count);
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
    for (int i = tid; i < K2_FAST_HIST; i += nthr) {
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
template <bool GLIBC, class Vote>
__device__ __forceinline__ void k2_strict_item(const V3& fa, const V3& fb, const V3& fc, const double (*px)[2],
                                               const SolveParams& sp, int c0, int c1, int c2, int p0, int p1, int p2,
                                               unsigned kmask, u64 detmask, bool triple_voted, double* q, int qs,
                                               Vote vote) {
  const int n_m = sp.n_markers, nuo = n_m - 3;
  const V3 wa = {sp.markers[3 * p0], sp.markers[3 * p0 + 1], sp.markers[3 * p0 + 2]},
           wb = {sp.markers[3 * p1], sp.markers[3 * p1 + 1], sp.markers[3 * p1 + 2]},
           wc = {sp.markers[3 * p2], sp.markers[3 * p2 + 1], sp.markers[3 * p2 + 2]};
  P3PCtx ctx;
  // (GLIBC — vote_arith 3 / 4: the three complex powers of the quartic as libstdc++ / glibc evaluate them, mpe_ddmath.h;
  //  a template argument, so that the kernels of the other arithmetics are the code they were)
  if (!p3p_prepare(fa, fb, fc, wa, wb, wc, ctx, GLIBC)) return;  // computePoses returned -1
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
    for (u64 dm = detmask; dm; dm &= dm - 1) {  // unused detections, ascending (pose_estimator.cpp:576-597, 862-906)
      const int a = __builtin_ctzll(dm);
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
template <bool STORE, bool GLIBC>
__device__ __forceinline__ void k2_strict_frame(const mpe_detections* __restrict__ d, const SolveParams& sp,
                                                uint32_t* __restrict__ gh, int f, int part, int splits,
                                                const int* __restrict__ item_range, unsigned char* smem,
                                                double (*s_px)[2], double (*s_iv)[3], unsigned* s_hist) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int n_d = d->n, n_m = sp.n_markers;
  for (int i = tid; i < MPE_HIST_WORDS; i += nthr) s_hist[i] = 0;
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
    const u64 unused = (~0ull >> (64 - n_d)) & ~((1ull << c0) | (1ull << c1) | (1ull << c2));  // (n_d <= 64)
    k2_strict_item<GLIBC>(fa, fb, fc, s_px, sp, c0, c1, c2, p0, p1, p2, 0xFu, unused, false, s_q + tid, nthr,
                   [&](const int a, const int m) { atomicAdd(&s_hist[a * MPE_MAX_MARKERS + m], 1u); });
  }
  __syncthreads();
  for (int i = tid; i < MPE_HIST_WORDS; i += nthr) {
    const unsigned v = s_hist[i];
    if constexpr (STORE) gh[i] = v;
    else if (v) atomicAdd(&gh[i], v);
  }
}
template <bool GLIBC>
__global__ __launch_bounds__(K2_THREADS) void k2_vote_strict(const mpe_detections* __restrict__ dets, SolveParams sp,
                                                             uint32_t* __restrict__ hist, int splits,
                                                             const int* __restrict__ item_range) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ double s_px[MPE_MAX_DETECTIONS][2];
  __shared__ double s_iv[MPE_MAX_DETECTIONS][3];
  __shared__ unsigned s_hist[MPE_HIST_WORDS];
  const int f = blockIdx.x / splits, part = blockIdx.x - f * splits;
  const mpe_detections* d = dets + f;
  if (d->n < 4 || d->status != 0 || sp.n_markers < 4) return;
  k2_strict_frame<false, GLIBC>(d, sp, hist + (size_t)f * MPE_HIST_STRIDE, f, part, splits, item_range, smem, s_px, s_iv, s_hist);
}

// Frames that lost a suspect entry to a full list (k2_sus_lost) are voted again, whole, with the strict loop nest: the
// histogram is STORED over whatever the fast launch and the fix-up kernel left, the mark is cleared, the tail then
// sees an ordinary frame.  Frames too WIDE for the fast kernels (more than MPE_FAST_VOTE_DETECTIONS detections; they
// carry the same mark and zeroed histogram rows) are voted here for the first time: every block of the launch takes
// its share of such a frame's hypotheses and ADDS its votes (one frame of 64 detections and 5 markers is 2.5 M P3P
// solves), and the last block to finish clears those marks.
// Nothing marked since the last launch on this slot (ctl[1] + ctl[7] == ctl[4], the rule) -> every block leaves after
// three loads; otherwise the blocks stride over the launch's frames looking for the mark.  The last block to finish
// records what has been handled (ctl[4]; ctl[5] counts the blocks).
template <bool GLIBC>
__global__ __launch_bounds__(K2_THREADS) void k2_vote_relost(mpe_detections* __restrict__ dets, int n_frames, SolveParams sp,
                                                             uint32_t* __restrict__ hist, VoteFixup fx,
                                                             const int* __restrict__ item_range) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ double s_px[MPE_MAX_DETECTIONS][2];
  __shared__ double s_iv[MPE_MAX_DETECTIONS][3];
  __shared__ unsigned s_hist[MPE_HIST_WORDS];
  __shared__ int s_last;
  const unsigned lost = fx.ctl[1] + fx.ctl[7];
  if (lost == fx.ctl[4]) return;  // (only the last block of a launch writes ctl[4], after every block has read it)
  const bool any_wide = fx.ctl[7] != 0u;  // (cumulative: wide frames have passed through this slot at some time)
  for (int f = any_wide ? 0 : (int)blockIdx.x; f < n_frames; f += any_wide ? 1 : (int)gridDim.x) {
    mpe_detections* d = dets + f;
    if (d->status != MPE_FRAME_VOTE_LIST_FULL) continue;  // (written by an earlier launch: uniform over the block)
    const bool wide = d->n > MPE_FAST_VOTE_DETECTIONS;
    uint32_t* gh = hist + (size_t)f * MPE_HIST_STRIDE;
    if (wide) {  // every block: its share of the frame; the mark stays until all of them are done
      if (sp.n_markers >= 4)
        k2_strict_frame<false, GLIBC>(d, sp, gh, f, (int)blockIdx.x, (int)gridDim.x, item_range, smem, s_px, s_iv, s_hist);
      __syncthreads();
      continue;
    }
    if (f % (int)gridDim.x != (int)blockIdx.x) continue;  // a narrow frame: one block, the histogram stored
    if (d->n >= 4 && sp.n_markers >= 4)
      k2_strict_frame<true, GLIBC>(d, sp, gh, f, 0, 1, item_range, smem, s_px, s_iv, s_hist);  // (forensics: the range)
    __syncthreads();
    if (threadIdx.x == 0) {
      d->status = 0;
      atomicAdd(&fx.ctl[6], 1u);  // frames voted again (cumulative)
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    __threadfence();
    s_last = atomicAdd(&fx.ctl[5], 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  if (any_wide) {  // every block's votes are in: the wide frames become ordinary ones
    for (int f = threadIdx.x; f < n_frames; f += blockDim.x) {
      mpe_detections* d = dets + f;
      if (d->status == MPE_FRAME_VOTE_LIST_FULL && d->n > MPE_FAST_VOTE_DETECTIONS) {
        d->status = 0;
        atomicAdd(&fx.ctl[6], 1u);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    fx.ctl[4] = lost;
    fx.ctl[5] = 0;
    __threadfence();
  }
}

// The hypotheses a fast voting launch left undecided (k2_sus_push), one per lane, through k2_strict_item; their votes
// go straight into the frames' histograms in global memory (the voting launch has stored or added its own by then:
// same stream, or an event in between).  The last block to finish resets the list's fill count for the next launch
// and adds the number of entries to the cumulative counter (ctl[3]).
#define K2_FIX_THREADS 64
template <bool GLIBC>
__global__ __launch_bounds__(K2_FIX_THREADS) void k2_vote_fixup(const mpe_detections* __restrict__ dets, SolveParams sp,
                                                                uint32_t* __restrict__ hist, VoteFixup fx) {
  __shared__ double s_q[2 * (MPE_MAX_MARKERS - 3) * K2_FIX_THREADS];
  const unsigned n = min(fx.ctl[0], fx.cap);
  if (n == 0) return;  // (nothing appended — every block sees the same count — and nothing to reset)
  const u64* list = reinterpret_cast<const u64*>(fx.list);
  const int tid = threadIdx.x;
  for (unsigned i = blockIdx.x * K2_FIX_THREADS + tid; i < n; i += gridDim.x * K2_FIX_THREADS) {
    const u64 w0 = list[(size_t)K2_SUS_WORDS * i];
    const u64 detmask = list[(size_t)K2_SUS_WORDS * i + 1];  // (32 bits: the fast kernels only vote frames that narrow)
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
    k2_strict_item<GLIBC>(fa, fb, fc, px, sp, c0, c1, c2, p0, p1, p2, kmask, detmask, triple_voted, s_q + tid, K2_FIX_THREADS,
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
                           hipStream_t s, int relost_blocks, const int* item_range) {
  if (!fx.ctl || fx.cap == 0 || sp.n_markers < 4 || n_frames <= 0) return hipSuccess;
  // (the entry count lives on the device: a fixed grid strides over it — wide, every entry is a single-wave chain of
  //  dependent FP64 operations (~30 us), and blocks beyond the count leave at once; ~0.15 % of the hypotheses)
  const bool glibc = vote_arith_glibc_pow(sp.vote_arith);
  if (glibc)
    hipLaunchKernelGGL(k2_vote_fixup<true>, dim3(2048), dim3(K2_FIX_THREADS), 0, s, dets, sp, hist, fx);
  else
    hipLaunchKernelGGL(k2_vote_fixup<false>, dim3(2048), dim3(K2_FIX_THREADS), 0, s, dets, sp, hist, fx);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  // frames that lost an entry to a full list: voted again with the strict loop nest.  A SMALL grid: the launch sits in
  // the tail chain of every sub-batch, beside the next voting launch whose pending blocks it has to queue behind —
  // 256 blocks that only read two words took 0.39 ms there (profiles/round5_bench_kernel_stats.csv of the first
  // collection; 3 us serialised).  The lists are sized so that nothing is lost as a rule.
  const size_t lds_strict = (size_t)(sp.n_markers - 3) * 2 * K2_THREADS * sizeof(double);
  // relost_blocks: the caller's choice — 32 where the launch sits beside a voting launch and nothing has ever been
  // lost, the whole chip once frames have been (mpe_schedule.cpp: relost_grid)
  if (relost_blocks < 1) relost_blocks = 32;
  if (glibc)
    hipLaunchKernelGGL(k2_vote_relost<true>, dim3((unsigned)relost_blocks), dim3(K2_THREADS), lds_strict, s, dets,
                       n_frames, sp, hist, fx, item_range);
  else
    hipLaunchKernelGGL(k2_vote_relost<false>, dim3((unsigned)relost_blocks), dim3(K2_THREADS), lds_strict, s, dets,
                       n_frames, sp, hist, fx, item_range);
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
  if (!vote_arith_is_strict(sp.vote_arith) && fixup && fixup->ctl && fixup->list) {
    fx = *fixup;
    fx.screen = vote_arith_screens(sp.vote_arith) ? 1u : 0u;
  }
  if (n_frames <= 0 || sp.n_markers < 4) return hipSuccess;
  // the fast kernels append to the list (suspects, or what their per-wave queues cannot hold): they need one
  if (!vote_arith_is_strict(sp.vote_arith) && !fx.ctl) return hipErrorInvalidValue;
  int slice_tab = 0;
  if (splits < 0) {  // -(blocks per frame): the blocks share the marker permutations, table slices in LDS
    splits = -splits;
    slice_tab = 1;
  }
  if (splits < 1) splits = 1;
  const int nuo = sp.n_markers - 3;
  if (vote_arith_is_strict(sp.vote_arith)) {  // strict arithmetic: the validation kernel's P3P, no tables, no scan rider
    const size_t lds_strict = (size_t)nuo * 2 * K2_THREADS * sizeof(double);
    if (vote_arith_glibc_pow(sp.vote_arith))
      hipLaunchKernelGGL(k2_vote_strict<true>, dim3((unsigned)(n_frames * splits)), dim3(K2_THREADS), lds_strict, s, dets,
                         sp, hist, splits, item_range);
    else
      hipLaunchKernelGGL(k2_vote_strict<false>, dim3((unsigned)(n_frames * splits)), dim3(K2_THREADS), lds_strict, s, dets,
                         sp, hist, splits, item_range);
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
    if (n_det_hint >= K2_CG_MIN_DETECTIONS)
      hipLaunchKernelGGL((k2_vote<true, false, 0, true>), dim3((unsigned)(n_frames * splits)), dim3(threads), lds, s, dets, sp,
                         tab, hist, splits, sa, (const int*)nullptr, 0, fx);
    else
      hipLaunchKernelGGL((k2_vote<true, false>), dim3((unsigned)(n_frames * splits)), dim3(threads), lds, s, dets, sp, tab,
                         hist, splits, sa, (const int*)nullptr, 0, fx);
  } else if (nuo <= 2 && !item_range && splits == 1) {
    // <= 5 markers and nothing to scan (small batches, single frames, the tracker's brute-force initialisation, the
    // stage-level vote entry): the scan-carrying variant all the same, with an EMPTY rider — every service point is a
    // no-op (ScanRider::issue / consume return at once with n_chunks = 0).  Its single-precision head and per-wave
    // queue make it 16 % faster than the plain <= 5-marker kernel, which pays the suspect screen without them: 7.77
    // against 9.26 ms per 262 144 frames, 486 against 579 us per 16 384 (round 5, call r5u; round 3's kernel without
    // the screen: 528).  The plain kernel remains for the forensics entry (item ranges) and explicit vote_splits.
    sa.thr = make_thr_test(scan_thr);
    lds = (size_t)(threads / 64) * chunk_bytes +
          (size_t)sp.n_markers * (sp.n_markers - 1) * (sp.n_markers - 2) * K2_LTAB * sizeof(double) +
          (size_t)(threads / 64) * K2_VQ_CAP * K2_VQ_WORDS * sizeof(u64);
    if (n_det_hint >= K2_CG_MIN_DETECTIONS)
      hipLaunchKernelGGL((k2_vote<true, false, 0, true>), dim3((unsigned)(n_frames * splits)), dim3(threads), lds, s, dets, sp,
                         tab, hist, splits, sa, (const int*)nullptr, 0, fx);
    else
      hipLaunchKernelGGL((k2_vote<true, false>), dim3((unsigned)(n_frames * splits)), dim3(threads), lds, s, dets, sp, tab,
                         hist, splits, sa, (const int*)nullptr, 0, fx);
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

//@file-epilogue
}  // namespace mpe
//@file-epilogue-end
