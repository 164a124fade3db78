#!/usr/bin/env python3
"""Fill README.md's measured paragraph from the committed collection (profiles/round6_bench.json, round6_pmc.json):
   python tools/write_readme.py > README.md      (the numbers in the prose are then the collection's, not retyped)"""
import json
import os

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
b = json.load(open(os.path.join(R, "profiles", "round6_bench.json")))
p = json.load(open(os.path.join(R, "profiles", "round6_pmc.json")))
oc, cl, tr = b["other_configs"], b["clutter"], b["tracked"]
M = lambda v: "%.2f M" % (v / 1e6)
k = lambda v: "%.0f k" % (v / 1e3)
ps = lambda key, nd: p["k2_vote_valu"][key]["valu_insts_per_frame"] * 64.0 / (nd * (nd - 1) * (nd - 2) // 6 * 60)
print('''# mpe-mi355x — MI355X-native compute back-end for monocular LED pose estimation

From-scratch AMD Instinct MI355X (gfx950 / CDNA4) implementation of the per-frame hot path of
`uzh-rpg/rpg_monocular_pose_estimator`: LED blob detection on a mono8 frame (threshold → fixed-point
Gaussian → external-contour polygon centroids → undistortion), brute-force LED/detection
correspondence search (every detection triple × marker permutation solved with Kneip's P3P and
scored by reprojection voting), correspondence validation, and Gauss-Newton PnP refinement with
pose covariance — behind a C ABI (`include/mpe.h`) and a source-compatible `PoseEstimator` facade.

* `rpg_monocular_pose_estimator_amd/csrc/` — hand-written HIP kernels (`mpe_k1.hip` + `mpe_k1b_dev.h` image scan + blob extraction, `mpe_k2.hip` voting, `mpe_k3.hip` validate / refine / tracked frame, `mpe_p3p.h`, `mpe_ddmath.h`) + the C ABI (`mpe_host.h`, `mpe_schedule.cpp`, `mpe_options.cpp`, `mpe_track_abi.cpp`, `mpe_abi.cpp`, `mpe_tracker.cpp` → `libmpe_hip.so`)
* `compat/` — C++ facade with every public class / method name of the reference's library (`PoseEstimator`, `LEDDetector`, `P3P`, `Combinations`, `Visualization`), ROS-free example + replay CLI, `compat/ros/` node / nodelet source
* `rpg_monocular_pose_estimator_amd/` — ctypes binding, Python mirror of the facade, synthetic workload, sharding helpers
* `oracle/` — dependency-free CPU restatement of the reference path (**test infrastructure only**)
* `tests/` — oracle known-answer tests, an independent numpy witness of the whole path + the golden vectors it made, ABI tests, the device source of the blob extraction, of the P3P / tail geometry and of the libstdc++ / glibc restatement compiled for the host against the oracle and this image's libm (all CPU), and HIP-vs-oracle parity tests (`-m gpu`)
* `bench.py` (headline: batched brute force), `bench_streams.py` (independent tracked camera streams), `__graft_entry__.py`, `profiles/`, `tools/` — measurement
* `DESIGN.md` (path, kernels, rooflines), `INTEGRATION.md` (how to bind it), `SURVEY.md` (scope contract)
''')
print('''On one MI355X (round 6, `profiles/round6_*`; every figure below is in the default `python bench.py` line, i.e. under
the driver's clock, with its own roofline and oracle parity sample, and as a flat key of `config`): **%s frames/s** at
752×480 / 5 LEDs with brute-force initialisation on every frame (%.2f ms per 262 144 frames in the collection's default
line; 17.05 – 17.9 ms over the round's boxes; target 50 k fps; rounds 5 … 1: 14.6 M, 14.7 M, 13.9 M, 13.3 M, 10.0 M),
records delivered to pinned host memory inside the step.  Every pixel is read once — 72 %% of them by the FP64 voting
kernel itself through LDS DMA, the rest by a one-block-per-CU side scan that streams beside it — at %.2f TB/s = %.2f of
the HBM spec over the WHOLE step (1.10 – 1.13 × the stand-alone scan of the same pixels, the floor of this design; 1.011 ×
the algorithmic bytes by the counters).
**Results are the CPU reference build's**: detections bit for bit; since round 6 the vote histograms equal the oracle's
on every soaked frame ALSO in the unstable corner of the reference's Ferrari solver (0 of 65 536 C2 and 0 of 16 384 C3
frames differ; rounds 3 – 5: 1 and 27) — the strict voting item now evaluates `std::pow(complex, double)` as libstdc++ /
glibc do (`csrc/mpe_ddmath.h`, pinned against this image's libm in the CPU tier) — the default arithmetic equals the strict
kernel on 1 048 576 C2 + 65 536 C3 + 294 912 cluttered frames, poses agree to ~1e-14 m, 0 status mismatches in 524 288 /
4 096 / 8 192 / 32 768 soaked frames of C2 / C3 / C4 / C1; up to 64 detections per frame (was 32).
Other configs: %s fps at 4 LEDs, %s fps at 1920×1200, %s / %s fps at 8 LEDs / 12 detections (73 920 P3P solves per frame,
tolerance 5 / 2 px; %.2f of the FP64 issue roof at the spec clock).  **Clutter**: 4 distractor spots %s fps, 16 distractors
**%s** (round 5: 164 k — the voting kernel now looks its back-projections up in a grid of detection masks: %.0f VALU per P3P
solve instead of 2 496), 0.05 %% salt noise %s, a saturated 64×64 patch %s.  The stateful tracking path is one kernel launch
per frame: %.3f ms for one stream (one CPU core: %.3f — the per-phase cycle table is in DESIGN.md §1), %s / %s tracked
frames/s for 8 / 64 streams driven in lock step from one host thread.  CPU baseline (the restated reference path on the
box's %d cores): %s fps, %s single-threaded.
''' % (M(b["value"]), b["ms_per_step"], b["step_hbm"]["achieved_GBps"] / 1e3, b["step_hbm"]["frac_of_spec"],
       M(oc["C1"]["value"]), M(oc["C4"]["value"]), k(oc["C3"]["value"]), k(oc["C3_tol2"]["value"]), oc["C3"]["roofline"]["frac"],
       M(cl["d4"]["value"]), k(cl["d16"]["value"]), ps("C2_d16", 21), k(cl["salt"]["value"]) if cl["salt"]["value"] < 1e6 else M(cl["salt"]["value"]),
       M(cl["patch"]["value"]), tr["one_stream"]["latency_ms_per_frame"], 1e3 / tr["one_stream"]["cpu_one_core_fps"],
       k(tr["lockstep_8"]["fps"]), k(tr["lockstep_64"]["fps"]), b["cpu_baseline"]["cores"], k(b["cpu_baseline"]["value"]),
       k(b["cpu_baseline"]["single_thread_fps"]) if b["cpu_baseline"]["single_thread_fps"] > 1e3 else "%.0f" % b["cpu_baseline"]["single_thread_fps"]))
print('''```
python -c "import __graft_entry__ as g; g.build()"     # hipcc + g++, no GPU needed
python -m pytest tests -q -m "not gpu"                 # CPU suite
python -m pytest tests -q -m gpu                       # on an MI355X
python bench.py                                        # one JSON line: headline + every config, clutter, tracked streams (24 s)
python bench.py --headline-only                        # the C2 leg alone
python bench.py --gpus 8                               # launches its own 8 ranks (one per GPU, RCCL pose gather)
python bench_streams.py --streams 8                    # stateful estimator, 8 camera streams, one thread each
python bench_streams.py --streams 64 --lockstep        # 64 streams in lock step: one launch per time step
bash profiles/collect.sh 6 && python profiles/install.py 6   # (on an MI355X) the round's whole evidence, then condense it
tools/ab.sh NAME REPS "base build_variants/X/libmpe_hip.so" "bench args"   # same-box A/B of kernel variants (tools/build_variant.sh)
tools/host_sanitize.sh                                 # the device code's CPU tier (tests/host/*.cpp) under ASan + UBSan
tools/leg_trace.sh NAME --clutter salt                 # (on an MI355X) rocprofv3 kernel table of one bench leg; tools/leg_pmc.sh: one counter pass
```''')
