// Host build of the DEVICE voting source (test scaffolding, CPU tier): mpe_p3p.h as it is and, cut out of
// mpe_k2.hip at test time (vote_extract.inc), the index arithmetic, the marker-permutation table entry, the
// per-triple part of computePoses and the voting work item (quartic coefficients, Ferrari in the fast arithmetic,
// back-projection without forming [R|C], single-precision prefilter, exact nearest-neighbour votes) — i.e. what
// k2_vote<false> runs per lane, driven here by one "lane" over all (triple, permutation) items of a frame.
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <vector>
#include "mpe.h"
#include "mpe_p3p.h"
using std::max;
using std::min;
struct f32x2 {  // clang's ext_vector_type(2) float, as far as the voting item uses it
  float x, y;
};
static inline f32x2 operator-(f32x2 a, f32x2 b) { return f32x2{a.x - b.x, a.y - b.y}; }
static inline f32x2 operator*(f32x2 a, f32x2 b) { return f32x2{a.x * b.x, a.y * b.y}; }
static inline f32x2 k2_pk_fma(f32x2 a, f32x2 b, f32x2 c) { return f32x2{std::fmaf(a.x, b.x, c.x), std::fmaf(a.y, b.y, c.y)}; }
static inline float k2_fminf(float a, float b) { return std::fminf(a, b); }
static inline f32x2 operator+(f32x2 a, f32x2 b) { return f32x2{a.x + b.x, a.y + b.y}; }
static inline unsigned k2_cvt_pk_u8(float x, unsigned byte, unsigned into) {  // saturating, NaN -> 0
  const float r = std::nearbyintf(x);
  const unsigned v = !(r > 0.f) ? 0u : (r >= 255.f ? 255u : (unsigned)r);
  return (into & ~(0xFFu << (8 * byte))) | (v << (8 * byte));
}
static inline float k2_rsqf(float x) { return 1.0f / std::sqrt(x); }
static inline float k2_sqrtf(float x) { return std::sqrt(x); }
static inline bool k2_isfinite(double x) { return std::isfinite(x); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) {
  const unsigned o = *p;
  *p += v;
  return o;
}
typedef unsigned long long u64;
static inline void atomicOr(u64* p, u64 v) { *p |= v; }
static inline void atomicOr(unsigned* p, unsigned v) { *p |= v; }
static inline void wave_sync() {}
static inline void __syncthreads() {}
static inline double __longlong_as_double(long long v) {
  double d;
  std::memcpy(&d, &v, 8);
  return d;
}
static inline long long __double_as_longlong(double d) {
  long long v;
  std::memcpy(&v, &d, 8);
  return v;
}
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
static inline u64 __ballot(bool b) { return b ? 1ull : 0ull; }                       // a wave of one lane
static inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned, unsigned base) { return base; }  // no lower lanes
static inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned, unsigned base) { return base; }
// test hook: grid coordinates of a back-projection from the voting loop's single-precision chain against the
// double-precision functions the deferred evaluation uses; g_f32_err[0] = largest difference in PIXELS among the points
// the double-precision chain puts inside the grid, [1] = number of such points, [2] = back-projections within the vote
// tolerance of a detection, [3] = those of them whose grid cell is NOT set (must be 0: the loop would drop a vote)
static double g_f32_err[4];
#define K2_ON_GRID_COORD(F, rt, g1, g2, g3, p_2, d_12, b, mk, tr, gx, gy) \
  host_grid_coord(F, rt, g1, g2, g3, p_2, d_12, b, mk, tr, gx, gy)
namespace mpe {
struct K2Frame;
static void host_grid_coord(const K2Frame& F, double rt, double g1, double g2, double g3, double p_2, double d_12, double b,
                            const double* mk, const double* tr, float gx, float gy);
#include "vote_extract.inc"
static void host_grid_coord(const K2Frame& F, double rt, double g1, double g2, double g3, double p_2, double d_12, double b,
                            const double* mk, const double* tr, float gx, float gy) {
  const K2Sub S = k2_back_substitute(rt, g1, g2, g3, p_2, d_12, b);
  double qu, qv;
  k2_project_marker(S, mk, tr, qu, qv);
  // a back-projection that votes (within the tolerance of a detection) must find its cell set
  for (int a = 0; a < F.n_d; ++a) {
    const double du = qu - F.px[a][0], dv = qv - F.px[a][1];
    if (du * du + dv * dv < F.back_tol * F.back_tol) {
      g_f32_err[2] += 1;
      if (!k2_grid_bit(F, gx, gy)) g_f32_err[3] += 1;
      break;
    }
  }
  const double ex = qu * F.ginv + F.gxo, ey = qv * F.ginv + F.gyo;
  if (!(ex >= 0 && ex <= K2_GRID && ey >= 0 && ey <= K2_GRID)) return;
  const double err = std::fmax(std::fabs(ex - gx), std::fabs(ey - gy)) / F.ginv;
  g_f32_err[1] += 1;
  if (!(err <= g_f32_err[0])) g_f32_err[0] = err;  // (a NaN sticks)
}
}
extern "C" void host_vote_f32_err(double* out, int reset) {
  for (int i = 0; i < 4; ++i) out[i] = g_f32_err[i];
  if (reset) g_f32_err[0] = g_f32_err[1] = g_f32_err[2] = g_f32_err[3] = 0;
}
using namespace mpe;
// [0] suspect-list entries of the last call, [1] of which whole hypotheses (Ferrari), [2] entries lost to a full
// list, [3] the frame was marked MPE_FRAME_VOTE_LIST_FULL
static unsigned g_stats[4];
extern "C" void host_vote_stats(unsigned* out) {
  for (int i = 0; i < 4; ++i) out[i] = g_stats[i];
}

// variant 0: k2_vote<false> (back-projections in the LDS columns, votes on the spot); variant 1: k2_vote<true> as far as
// one lane can run it (LDS copy of the table, back-projections by value, exact votes deferred through the queue)
extern "C" int host_vote(const double* undist_xy, int n_d, const double* markers, int n_m, const double* k4,
                         double back_tol, unsigned* hist /* MPE_FAST_VOTE_DETECTIONS x MPE_MAX_MARKERS */, int variant) {
  if (n_d < 4 || n_m < 4 || n_d > MPE_FAST_VOTE_DETECTIONS || n_m > MPE_MAX_MARKERS) return -1;
  SolveParams sp;
  std::memset(&sp, 0, sizeof(sp));
  sp.n_markers = n_m;
  for (int i = 0; i < 3 * n_m; ++i) sp.markers[i] = markers[i];
  sp.fx = k4[0];
  sp.fy = k4[1];
  sp.cx = k4[2];
  sp.cy = k4[3];
  sp.back_tol = back_tol;
  const int n_perms = n_m * (n_m - 1) * (n_m - 2), n_combos = n_d * (n_d - 1) * (n_d - 2) / 6, nuo = n_m - 3;
  const int esz = k2_entry_doubles(n_m);
  std::vector<double> tab((size_t)n_perms * esz);
  for (int pj = 0; pj < n_perms; ++pj) k2_marker_entry(sp, pj, tab.data() + (size_t)pj * esz);
  double px[MPE_FAST_VOTE_DETECTIONS][2], iv[MPE_FAST_VOTE_DETECTIONS][3];
  f32x2 pxf[MPE_FAST_VOTE_DETECTIONS];
  for (int i = 0; i < n_d; ++i) {
    const double u = undist_xy[2 * i], v = undist_xy[2 * i + 1];
    px[i][0] = u;
    px[i][1] = v;
    pxf[i] = f32x2{(float)u, (float)v};
    const V3 b = bearing(u, v, sp.fx, sp.fy, sp.cx, sp.cy);
    iv[i][0] = b.x;
    iv[i][1] = b.y;
    iv[i][2] = b.z;
  }
  std::vector<double> tri((size_t)n_combos * 13);
  std::vector<unsigned> trii(n_combos);
  for (int i = 0; i < n_combos; ++i) k2_triple_entry(iv, n_d, i, sp.fx, sp.fy, sp.cx, sp.cy, tri.data() + (size_t)i * 13, trii[i]);
  std::vector<double> q(2 * nuo);
  std::vector<f32x2> qf(nuo);
  std::memset(hist, 0, sizeof(unsigned) * MPE_FAST_VOTE_DETECTIONS * MPE_MAX_MARKERS);
  std::vector<double> ltab((size_t)n_perms * K2_LTAB);
  for (size_t i = 0; i < ltab.size(); ++i) ltab[i] = k2_ltab_value(tab.data(), esz, nuo, (int)i);
  std::vector<u64> vq((size_t)std::max(K2_VQ_CAP * K2_VQ_WORDS, K2_DQ_CAP * K2_DQ_WORDS), 0);
  // variants 10 / 11: as 0 / 1 with the suspect list in place, worked off by k2_strict_item afterwards (what
  // k2_vote_fixup does on the device); variant 20: every hypothesis through k2_strict_item (= k2_vote_strict)
  // variants 30 / 31: the same with a block list of four entries — nearly every append goes straight to the launch's
  // list instead; variants 40 / 41: that one holds four entries as well — entries are LOST, counted, and the frame is
  // marked MPE_FRAME_VOTE_LIST_FULL (g_stats[3])
  const bool tiny_global = variant == 40 || variant == 41;
  if (tiny_global) variant -= 10;
  const bool tiny_list = variant == 30 || variant == 31;
  if (tiny_list) variant -= 20;
  const bool fixup = variant == 10 || variant == 11;
  if (fixup) variant -= 10;
  g_stats[0] = g_stats[1] = g_stats[2] = g_stats[3] = 0;
  if (variant == 20) {
    std::vector<double> qs(2 * nuo);
    for (int ti = 0; ti < n_combos; ++ti)
      for (int pj = 0; pj < n_perms; ++pj) {
        int c0, c1, c2, p0, p1, p2;
        unrank_combo3(ti, n_d, c0, c1, c2);
        perm_from_index(pj, n_m, p0, p1, p2);
        const V3 fa = {iv[c0][0], iv[c0][1], iv[c0][2]}, fb = {iv[c1][0], iv[c1][1], iv[c1][2]},
                 fc = {iv[c2][0], iv[c2][1], iv[c2][2]};
        const unsigned unused = (0xFFFFFFFFu >> (32 - n_d)) & ~((1u << c0) | (1u << c1) | (1u << c2));
        k2_strict_item<false>(fa, fb, fc, px, sp, c0, c1, c2, p0, p1, p2, 0xFu, unused, false, qs.data(), 1,
                       [&](const int a, const int m) { hist[a * MPE_MAX_MARKERS + m] += 1u; });
      }
    return 0;
  }
  const unsigned sus_cap = fixup ? (tiny_global ? 1u : 1u << 16) : 0u;
  const unsigned lds_cap = tiny_list ? 4u : 1u << 16;
  std::vector<u64> sus_list((size_t)K2_SUS_WORDS * (sus_cap + 1), 0), sus_lds((size_t)K2_SUS_WORDS * lds_cap, 0);
  unsigned sus_ctl[4] = {0, 0, 0, 0}, sus_lds_n = 0;
  std::vector<u64> grid((size_t)K2_GRID * K2_GRID_WORDS, 0);  // occupancy grid of the deferred plain variant
  float gp[4] = {0, 0, 0, 0};
  k2_grid_build(px, n_d, sp.back_tol, grid.data(), gp, 0, 1);
  if (variant == 1) {  // scan-carrying variant: pixel coordinates (G = 1)
    gp[0] = 1.f;
    gp[1] = gp[2] = 0.f;
  }
  std::vector<float> trif((size_t)n_combos * 12);
  for (int i = 0; i < n_combos; ++i) k2_triple_f32(tri.data() + (size_t)i * 13, gp, trif.data() + (size_t)i * 12);
  int frame_status = 0;
  const K2SusDesc susd = {sus_ctl, sus_list.data(), sus_cap, &frame_status};
  const K2Frame F = {trii.data(), reinterpret_cast<const double(*)[13]>(tri.data()), px, pxf, q.data(), qf.data(),
                     hist, tab.data(), ltab.data(), n_d, nuo, 1, 0, esz, sp.fx, sp.fy, sp.cx, sp.cy, sp.back_tol,
                     k2_prefilter_threshold(sp.back_tol, (variant == 0 && k2_defers(false, nuo <= 8 ? (nuo + 1) / 2 : 0)) ? 0.25 : 0.05),
                     vq.data(), 1, 0, &susd, fixup, 0,
                     sus_lds.data(), &sus_lds_n, lds_cap, grid.data(), gp[0], gp[1], gp[2],
                     reinterpret_cast<const float(*)[12]>(trif.data())};
  struct Fix {
    const K2Frame& F;
    const SolveParams& sp;
    const double (*iv)[3];
    const double (*px)[2];
    unsigned* hist;
    int nuo;
    ~Fix() {  // k2_sus_flush (the block's list -> the launch's), then the strict re-evaluation (k2_vote_fixup)
      if (!F.fix) return;
      unsigned s_base = 0;
      k2_sus_flush(F, &s_base);
      std::vector<double> qs(2 * nuo);
      const K2SusDesc& g = *F.susd;
      const unsigned n = std::min(g.ctl[0], g.cap);
      g_stats[0] += n;
      g_stats[2] += g.ctl[1];
      g_stats[3] = *g.frame_status == MPE_FRAME_VOTE_LIST_FULL ? 1u : 0u;
      for (unsigned i = 0; i < n; ++i) {
        const u64 w0 = g.list[(size_t)K2_SUS_WORDS * i];
        const unsigned detmask = (unsigned)g.list[(size_t)K2_SUS_WORDS * i + 1];
        const unsigned code = (unsigned)(w0 >> 32);
        const int c0 = code & 31, c1 = (code >> 5) & 31, c2 = (code >> 10) & 31;
        const int p0 = (code >> 15) & 15, p1 = (code >> 19) & 15, p2 = (code >> 23) & 15;
        const unsigned kmask = (code >> 27) & 15u;
        if (kmask == 0xFu) g_stats[1] += 1;
        const V3 fa = {iv[c0][0], iv[c0][1], iv[c0][2]}, fb = {iv[c1][0], iv[c1][1], iv[c1][2]},
                 fc = {iv[c2][0], iv[c2][1], iv[c2][2]};
        unsigned* h = hist;
        k2_strict_item<false>(fa, fb, fc, px, sp, c0, c1, c2, p0, p1, p2, kmask, detmask, (code >> 31) & 1u, qs.data(), 1,
                       [&](const int a, const int m) { h[a * MPE_MAX_MARKERS + m] += 1u; });
      }
    }
  } fix_at_exit = {F, sp, iv, px, hist, nuo};
  NoRider rider;
  if (variant == 1) {
    if (nuo > 2) return -2;  // the scan-carrying variant keeps at most two back-projections (in registers)
    // as on the device: 16 staged triples at a time (a queue entry names its triple by the index within the chunk), the
    // queue worked off before the next chunk is staged
    int vq_count = 0;
    for (int tc0 = 0; tc0 < n_combos; tc0 += K2_TRI_CHUNK_SCAN) {
      K2Frame Fc = F;
      Fc.trii = F.trii + tc0;
      Fc.tri = F.tri + tc0;
      Fc.trif = F.trif + tc0;
      for (int ti = 0; ti < std::min(K2_TRI_CHUNK_SCAN, n_combos - tc0); ++ti)
        for (int pj = 0; pj < n_perms; ++pj) k2_vote_item<true>(Fc, ti, pj, true, rider, vq_count);
      k2_vote_flush(Fc, vq_count);
      vq_count = 0;
    }
  } else {
    int unused = 0;  // (np >= 2: the fill count of the deferred-evaluation queue)
    // the instantiation the launcher would pick: (nuo + 1) / 2 marker pairs in registers up to 8 unused markers
    const int np = nuo <= 8 ? (nuo + 1) / 2 : 0;
    struct FlushAtExit {  // what is left in the deferred queue (np >= 2), before the suspect list is worked off
      const K2Frame& F;
      int& count;
      bool on;
      int np;
      ~FlushAtExit() {
        if (!on) return;
        switch (np) {
          case 1: k2_defer_flush<1>(F, count, true); break;
          case 2: k2_defer_flush<2>(F, count, true); break;
          case 3: k2_defer_flush<3>(F, count, true); break;
          default: k2_defer_flush<4>(F, count, true); break;
        }
      }
    } flush_at_exit = {F, unused, variant != 2 && k2_defers(false, np), np};
    for (int ti = 0; ti < n_combos; ++ti)
      for (int pj = 0; pj < n_perms; ++pj) {
        switch (variant == 2 ? 0 : np) {   // variant 2: force the LDS-column prefilter
          case 1: k2_vote_item<false, 1>(F, ti, pj, true, rider, unused); break;
          case 2: k2_vote_item<false, 2>(F, ti, pj, true, rider, unused); break;
          case 3: k2_vote_item<false, 3>(F, ti, pj, true, rider, unused); break;
          case 4: k2_vote_item<false, 4>(F, ti, pj, true, rider, unused); break;
          default: k2_vote_item<false, 0>(F, ti, pj, true, rider, unused); break;
        }
      }
  }
  return 0;
}

// n detection sets in one call (tests/soak_votes_host.py): det n x MPE_FAST_VOTE_DETECTIONS x 2, hist n x MPE_FAST_VOTE_DETECTIONS
// x MPE_MAX_MARKERS
extern "C" int host_vote_batch(const double* det, const int* n_det, int n, const double* markers, int n_m, const double* k4,
                               double back_tol, unsigned* hist, int variant, unsigned* stats_sum /* 3, optional */) {
  if (stats_sum) stats_sum[0] = stats_sum[1] = stats_sum[2] = 0;
  for (int i = 0; i < n; ++i) {
    unsigned* h = hist + (size_t)i * MPE_FAST_VOTE_DETECTIONS * MPE_MAX_MARKERS;
    if (n_det[i] < 4) {
      std::memset(h, 0, sizeof(unsigned) * MPE_FAST_VOTE_DETECTIONS * MPE_MAX_MARKERS);
      continue;
    }
    const int rc = host_vote(det + (size_t)i * 2 * MPE_FAST_VOTE_DETECTIONS, n_det[i], markers, n_m, k4, back_tol, h, variant);
    if (rc) return rc;
    if (stats_sum)
      for (int j = 0; j < 3; ++j) stats_sum[j] += g_stats[j];
  }
  return 0;
}
