#!/usr/bin/env python3
"""Section 4 of DESIGN.md and the clutter table of section 3 from the committed collection (profiles/round6_bench.json,
round6_pmc.json, round6_bench_headline_only.json, round6_bench_arith1.json): the text between the `<!-- measured:… -->`
markers of DESIGN.md is replaced.  python tools/write_design_measurement.py"""
import json
import os
import re

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda n: os.path.join(R, "profiles", n)
b = json.load(open(P("round6_bench.json")))
p = json.load(open(P("round6_pmc.json")))
ho = json.load(open(P("round6_bench_headline_only.json"))) if os.path.exists(P("round6_bench_headline_only.json")) else None
a1 = json.load(open(P("round6_bench_arith1.json"))) if os.path.exists(P("round6_bench_arith1.json")) else None
a2 = json.load(open(P("round6_bench_arith2.json"))) if os.path.exists(P("round6_bench_arith2.json")) else None
oc, cl, tr, rf = b["other_configs"], b["clutter"], b["tracked"], b["roofline"]
M = lambda v: "%.2f M" % (v / 1e6)
k = lambda v: "%.1f k" % (v / 1e3)
solves = lambda nd: nd * (nd - 1) * (nd - 2) // 6 * 60
ps = lambda key, nd: p["k2_vote_valu"][key]["valu_insts_per_frame"] * 64.0 / solves(nd)
slots = rf.get("timed_region_by_slot") or []
lm = [s["launch_ms"] for s in slots]
gp = [s["gap_before_ms"] for s in slots]
par = b["parity"]
sec4 = '''`python bench.py` = C2 (`BASELINE.json configs[1]`): 262 144 distinct synthetic 752×480 frames (95 GB, resident in HBM),
5 LEDs, demo.launch parameters.  A step = one submission of the streaming entry + the D2H copy of its 113 MB of pose
records into pinned host memory on a consumer stream.  All numbers: `profiles/round6_*`, one collection with the
committed binary (`profiles/collect.sh 6` → `install.py 6`; the counter file carries the fingerprint of the kernel + host
sources, `%s`, and the bench line says whether it matches: %s).  This text is generated from those files
(`tools/write_design_measurement.py`).
* **value: %s frames/s** (%.2f ms per step, median %.2f; `round6_bench.json`)%s.  Boxes — and runs on one box — differ by
  ± 3 %% (17.3 – 18.4 ms over the round's boxes and A/B runs; runs on ONE box fall into two modes, ~17.4 and ~18.3 ms), so
  the round's claims are interleaved same-box A/Bs: `vote_arith` 3 / 1 / 2 17.82 / 17.83 / 17.82 ms in the mean of five runs each
  (`round6_exp_vote_arith3_step_cost_final.txt`%s).  %.0f %% of the frames yield a pose
  (the rest fail in the reference algorithm too).
* `parity`: the records of the LAST TIMED submission as they arrived in host memory, %d frames against the oracle: %d
  status mismatches, %d poses within %.1e m / %.1e rad, %d unexplained — and none "explained" either: the
  allowance for the Ferrari corner is gone (§8).
* `step_hbm`: 94.6 GB ÷ step = %.2f TB/s = %.2f of spec for the whole step.
* `roofline` = `k2_vote<scan>`: 8.52 GB per launch (72 %% of a sub-batch) = the algorithmic bytes of §3 K1a × the 23 593
  frames' worth one launch scans; `frac` %.3f is the LOWER of `frac_hip_events` %.3f (pairs of events around all 160
  fused launches inside the timed region: %.3f ms) and `frac_rocprofv3` %.3f (`rocprofv3 --kernel-trace
  --kernel-include-regex '.*k2_vote<true.*' --stats` of the same command, `round6_bench_vote_only_kernel_stats.csv`:
  %.3f ms — inflated by the first launch of every submission, a tracer artefact: `round6_vote_launch_outliers.json`).
  `traffic` %.3f GB = %.3f × algorithmic (FETCH_SIZE × 2 + WRITE_SIZE passes of the timed launch shape,
  `round6_pmc_timed_*.csv`).  `timed_region_by_slot`: the eight launches of a submission %.2f – %.2f ms, the windows in
  front of them %.2f – %.2f ms.  %d VALU + %d SALU per frame voted, %.0f %% of the wave cycles waiting, %.2f GHz.
* `false_hint_leg`: the same steps with an announcement that does not come true: %.2f ms per step.
* `cpu_baseline` (`kind: "port"`, the oracle on the box's %d usable cores, 16 384 frames): %s fps, single thread %.0f fps — a
  reported baseline, not the target.  `host_streamed` (PCIe-inclusive, never `value`): %s fps = %.0f GB/s.
* **`other_configs`** (each 5 steps after 2 warm-up, 512-frame oracle sample: 0 status mismatches, 0 unexplained): C1 (4
  LEDs, 65 536 frames) **%s fps**, `k2_vote<scan>` HBM %.2f; C3 (8 LEDs / 12 detections, 16 384 frames) **%s fps**,
  `k2_vote<plain>` FP64 issue %.2f (spec clock; %d VALU per P3P solve); C3 at tolerance 2 **%s**, %.2f; C4 (1920×1200,
  16 384 frames = 38 GB) **%s fps**, HBM %.2f.  **`clutter`**: the table in §3.  **`tracked`**: one stream %.3f ms per frame
  (the oracle's tracker on one core %.3f), lock step 8 / 64 streams %s / %s fps; **`latency_ms_one_frame`** (one
  brute-force frame, host memory in, record out) %.3f ms: §1.  The whole line takes %.0f s of the driver's clock.  Every
  leg also rides in `config` as flat scalars (`C3_fps`, `C3_frac`, `d16_fps`, `trk1_ms`, `legs_unexplained` …): the
  driver's record keeps only those (VERDICT round 5, item 4).
* N > 1 (`--gpus N`): the line carries `ranks_seen` (from the process group), `per_rank_fps` and `shard_parity` — the
  first 64 frames of EVERY rank's shard, re-created on rank 0 from the rank's seeds and checked against the oracle on
  the records as they arrived through the gather; the gloo world-size-2 / 3 / 8 CPU tests assert the same fields on the
  plumbing run.
''' % (p["source_fingerprint"], rf["counters"]["from_a_build_of_these_sources"], M(b["value"]), b["ms_per_step"], b["ms_per_step_median"],
       ("; minutes later on the same box the headline leg alone: %s, %.2f ms" % (M(ho["value"]), ho["ms_per_step"])) if ho else "",
       ("; in the collection itself, headline leg alone: 3 → %.2f, 1 → %.2f, 2 (fast alone) → %.2f ms" % (ho["ms_per_step"], a1["ms_per_step"], a2["ms_per_step"])) if (ho and a1 and a2) else "",
       100 * b["poses_found_frac"], par["frames"], par["status_mismatches"], par["poses_compared"], par["pos_max_m"], par["rot_max_rad"],
       par["mismatches_unexplained"], b["step_hbm"]["achieved_GBps"] / 1e3, b["step_hbm"]["frac_of_spec"],
       rf["frac"], rf["frac_hip_events"], rf["avg_launch_ms"], rf["frac_rocprofv3"], rf["rocprofv3_avg_launch_ms"],
       rf["traffic"] / 1e9, rf["traffic"] / rf["bytes_per_launch"], min(lm), max(lm), min(gp), max(gp),
       p["k2_vote_valu"]["fused_C2"]["valu_insts_per_frame"], p["k2_vote_valu"]["fused_C2"]["salu_insts_per_frame"],
       100 * p["k2_vote_valu"]["fused_C2"]["wait_inst_any_over_wave_cycles"], p["k2_vote_valu"]["fused_C2"]["effective_clock_GHz"],
       b["false_hint_leg"]["ms_per_step"], b["cpu_baseline"]["cores"], k(b["cpu_baseline"]["value"]), b["cpu_baseline"]["single_thread_fps"],
       k(b["host_streamed_fps"]), b["host_streamed_fps"] * 360960 / 1e9,
       M(oc["C1"]["value"]), oc["C1"]["roofline"]["frac"], k(oc["C3"]["value"]), oc["C3"]["roofline"]["frac"],
       p["k2_vote_valu"]["C3"]["valu_insts_per_frame"] * 64 / 73920, k(oc["C3_tol2"]["value"]), oc["C3_tol2"]["roofline"]["frac"],
       M(oc["C4"]["value"]), oc["C4"]["roofline"]["frac"], tr["one_stream"]["latency_ms_per_frame"], 1e3 / tr["one_stream"]["cpu_one_core_fps"],
       k(tr["lockstep_8"]["fps"]), k(tr["lockstep_64"]["fps"]), b["latency_ms_one_frame"]["pinned"]["median_ms"], b["total_s"])


def leg(name):
    v = cl[name]
    r = v.get("roofline") or {}
    return v["value"], r.get("bound"), r.get("frac")


d4, d16, salt, patch = leg("d4"), leg("d16"), leg("salt"), leg("patch")
g = p.get("k1b_general_salt") or {}
clutter = '''| leg | frames/s (round 5) | what grows | voting launch / dominant kernel: bound, fraction |
|---|---|---|---|
| clean (headline) | %s (14.61 M) | — | HBM %.2f |
| 4 distractor spots (9 detections) | %s (3.37 M) | C(9,3) = 84 triples × 60: 8.4 × the hypotheses | FP64 issue %.2f (%d VALU per solve; round 5: 1 668) |
| 16 distractor spots (21 detections) | **%s** (164 k) | 1 330 triples × 60 = 79 800 P3P per frame (C3's 73 920) | FP64 issue %.2f (%d VALU per solve; round 5: 2 496) |
| 0.05 %% salt noise (~180 isolated bright pixels) | %s (979 k) | every frame in the general tier | the blob tiers: bound by the request rate of scattered accesses (`k1b_general`: %d VALU per frame; round 5: 140 621) |
| saturated 64×64 patch | %s (2.80 M) | one 68-row island per frame | blob tiers |

Every leg: 256-frame oracle sample, 0 status mismatches, 0 unexplained.  With distractors it is the VOTING that grows —
C(n_d,3) — not the blob tiers (the reference's own cost model, `pose_estimator.cpp:565-702`).  Round 5 paid for every unused
detection in every root's prefilter (2 496 VALU per solve at 21 detections against 1 500 on a clean frame); the grid of
detection masks (K2 below) makes a solve cost the same ~1 400 at 9 and at 21 detections.  The fractions are against 614 G
wave-instructions/s (1 024 SIMDs × the 2.4 GHz SPEC clock ÷ 4).  The salt leg's gain is the general tier's (band, column
run, row piece) items, narrowed to-do columns, batched and coalesced loads, resident-block grid and 4 waves per SIMD
(`round6_exp_general_tier.txt`, K1b below).
''' % (M(b["value"]), rf["frac"], M(d4[0]), d4[2] or 0, ps("C2_d4", 9), k(d16[0]), d16[2] or 0, ps("C2_d16", 21),
       (M(salt[0]) if salt[0] >= 1e6 else k(salt[0])), g.get("valu_insts_per_frame", 0), M(patch[0]))

path = os.path.join(R, "DESIGN.md")
s = open(path).read()
for tag, txt in (("section4", sec4), ("clutter", clutter)):
    a, z = "<!-- measured:%s:begin -->\n" % tag, "<!-- measured:%s:end -->\n" % tag
    assert a in s and z in s, tag
    s = s[:s.index(a) + len(a)] + txt + s[s.index(z):]
open(path, "w").write(s)
print("DESIGN.md: measured sections written")
