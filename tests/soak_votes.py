"""Vote-histogram parity soak with forensics: detection sets of N synthetic frames (HIP detection, bit-exact vs the
oracle) through the HIP voting in THREE arithmetics (option "vote_arith": 1 = the default: the fast kernel with its
suspects re-evaluated by the strict functions, k2_vote_fixup; 0 = the strict kernel; 2 = the fast kernel alone, round 3's
default) and through the oracle's voting (frame-parallel on the host cores).  Counted: frames whose histogram differs
from the oracle's, per arithmetic, and — the round-4 claim — frames on which the DEFAULT differs from STRICT (must be
0: exit code 2 otherwise).  Every frame that differs from the oracle is SAVED (detections + both histograms ->
<out>.npz) and CLASSIFIED (tests/forensics.py): both paths are asked for every hypothesis' own votes, the difference is
traced to the hypotheses that cast it, and each of those must sit in the unstable corner of the reference's Ferrari
solver (cancellation < 1e-12) or be one on which the oracle's own P3P answer moves under a 1-ulp change of an input.
Exit code 1 if a mismatch of the default or the strict arithmetic stays unexplained (the fast arithmetic alone also
fails outside that corner — near-degenerate detection triples — which is listed, not counted).
usage (on an MI355X): python tests/soak_votes.py [frames [config [out_prefix]]]      -> one JSON line
MPE_SOAK_STRICT=0 skips the strict kernel (2.5x the time; at C3 it is run on the first MPE_SOAK_STRICT_FRAMES frames);
MPE_SOAK_ORACLE=0 skips the oracle and the fast-alone arithmetic: only DEFAULT against STRICT, at GPU speed (large N)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import rpg_monocular_pose_estimator_amd as mpe  # noqa: E402
from rpg_monocular_pose_estimator_amd import synth  # noqa: E402
import oracle  # noqa: E402
import forensics  # noqa: E402

oracle.build()
from oracle import binding as orc  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
CONFIG = sys.argv[2] if len(sys.argv) > 2 else "C2"
OUT = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/soak_votes_%s" % CONFIG
CH = min(N, 32768 if CONFIG != "C3" else 2048)
cfg = dict(synth.CONFIGS[CONFIG])
if os.environ.get("MPE_SOAK_DISTRACTORS"):  # cluttered frames (the occupancy-grid variant of the <= 5-marker kernel)
    cfg["n_distractors"] = int(os.environ["MPE_SOAK_DISTRACTORS"])
rows, cols = cfg["rows"], cfg["cols"]
K, D = synth.camera_for(rows, cols)
markers = np.asarray(cfg["markers"])
dev = torch.device("cuda", 0)
h = mpe.Handle(0)
P = mpe.demo_params()
TOL = 5.0
cores = len(os.sched_getaffinity(0))
STRICT = os.environ.get("MPE_SOAK_STRICT", "1") != "0"
ORACLE = os.environ.get("MPE_SOAK_ORACLE", "1") != "0"
# MPE_SOAK_PAIR="3,4": the same soak for the arithmetics with the reference library's powers (round 6): "default" then
# means vote_arith 3 and "strict" vote_arith 4
A_DEF, A_STRICT = [int(x) for x in os.environ.get("MPE_SOAK_PAIR", "1,0").split(",")]
STRICT_FRAMES = int(os.environ.get("MPE_SOAK_STRICT_FRAMES", str(N if CONFIG != "C3" else min(N, 4096))))
diff = {0: 0, 1: 0, 2: 0}
cells = {0: 0, 1: 0, 2: 0}
default_vs_strict = 0
default_vs_strict_frames = []
strict_frames = 0
saved = []
tot = 0
t0 = time.time()
h.get_option("vote_fixup_items")
for part in range(max(1, N // CH)):
    _, spots = synth.make_scenes_batch(cfg, CH, seed=7100 + part)
    frames = synth.render_frames_torch(spots, rows, cols, cfg["spot_sigma"], dev, seed=8100 + part)
    torch.cuda.synchronize()
    det = h.detect_batch(frames, K, D, P)
    nd = det["n"].astype(np.int32)
    dets = det["undist_xy"].reshape(CH, mpe.MAX_DETECTIONS, 2)
    ref = orc.vote_batch(dets, nd, markers, K, TOL, n_threads=cores) if ORACLE else None
    got = {}
    do_strict = STRICT and tot < STRICT_FRAMES
    for arith in ((1, 2, 0) if ORACLE else (1, 0)):
        if arith == 0 and not do_strict:
            continue
        h.set_option("vote_arith", {1: A_DEF, 0: A_STRICT}.get(arith, arith))
        got[arith] = h.vote_batch([dets[i, :nd[i]] for i in range(CH)], markers, K, TOL)
        for i in (range(CH) if ORACLE else ()):
            r = ref[i, :nd[i], :len(markers)] if nd[i] >= 4 else np.zeros((nd[i], len(markers)), np.uint32)
            g = got[arith][i] if nd[i] >= 4 else np.zeros_like(r)
            if not np.array_equal(g, r):
                diff[arith] += 1
                cells[arith] += int((g != r).sum())
                c = forensics.classify_mismatch(dets[i, :nd[i]], markers, K, TOL, orc, h)
                saved.append({"part": part, "frame": i, "vote_arith": arith, "det": dets[i, :nd[i]].copy(),
                              "hip": g.copy(), "oracle": r.copy(), "verdict": c})
    h.set_option("vote_arith", 1)
    if do_strict:
        strict_frames += CH
        for i in range(CH):
            if nd[i] >= 4 and not np.array_equal(got[1][i], got[0][i]):
                default_vs_strict += 1
                default_vs_strict_frames.append([part, i])
    tot += CH
    print(part, tot, diff, default_vs_strict, round(time.time() - t0), flush=True)
# the claim is about the default and the strict arithmetic; what the fast arithmetic ALONE (vote_arith 2, kept for A/B
# measurements) gets wrong outside the Ferrari corner is reported separately — it is what the round-4 screen exists for
unexplained = [s for s in saved if not s["verdict"]["unstable"] and s["vote_arith"] != 2]
fast_alone_other = [s for s in saved if not s["verdict"]["unstable"] and s["vote_arith"] == 2]
if saved:
    os.makedirs(os.path.dirname(OUT) or ".", exist_ok=True)
    np.savez(OUT + ".npz", **{"det_%d" % k: s["det"] for k, s in enumerate(saved)},
             **{"hip_%d" % k: s["hip"] for k, s in enumerate(saved)},
             **{"oracle_%d" % k: s["oracle"] for k, s in enumerate(saved)},
             meta=json.dumps([{k: v for k, v in s.items() if k not in ("det", "hip", "oracle")} for s in saved]))
print(json.dumps({"config": CONFIG, "frames": tot, "vote_arith_of_default_and_strict": [A_DEF, A_STRICT], "p3p_solves_per_frame": "C(n_det,3) x P(n_markers,3)",
                  "oracle_compared": ORACLE,
                  "frames_with_a_different_histogram_vs_oracle": {
                      "default (vote_arith 1: fast + strict re-evaluation of suspects)": diff[1],
                      "strict (vote_arith 0)": diff[0] if strict_frames else None,
                      "fast alone (vote_arith 2, round 3's default)": diff[2]} if ORACLE else None,
                  "differing_cells": {"default": cells[1], "strict": cells[0], "fast alone": cells[2]},
                  "frames_compared_default_vs_strict": strict_frames,
                  "frames_default_differs_from_strict": default_vs_strict,
                  "frames_default_differs_from_strict_idx": default_vs_strict_frames[:20],
                  "vote_fixup_items": h.get_option("vote_fixup_items"),
                  "vote_fixup_overflow": h.get_option("vote_fixup_overflow"),
                  "mismatches_classified_unstable": len(saved) - len(unexplained) - len(fast_alone_other),
                  "mismatches_unexplained": len(unexplained),
                  "fast_alone_mismatches_outside_the_ferrari_corner": [[s["part"], s["frame"]] for s in fast_alone_other],
                  "mismatching_frames_saved_to": (OUT + ".npz") if saved else None,
                  "verdicts": [{k: v for k, v in s.items() if k not in ("det", "hip", "oracle")} for s in saved]}))
sys.exit(2 if default_vs_strict else (1 if unexplained else 0))
