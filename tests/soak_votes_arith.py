"""Round 6 vote-histogram soak: how many frames differ from the CPU oracle in each voting arithmetic.
  vote_arith 0  strict kernel, exact products / cbrt(hypot) for the quartic's complex powers (rounds 1 - 5)
             1  default: fast kernel + strict re-evaluation of its suspects (same powers as 0)
             3  as 1, 4 as 0, with the powers evaluated as libstdc++ / glibc do (csrc/mpe_ddmath.h)
Counted per arithmetic: frames (and cells) whose histogram differs from the oracle's; and frames on which the screened
fast kernel differs from its own strict kernel (1 vs 0, 3 vs 4: both should be 0 — the screen's thresholds were
derived for arithmetic 0; for 3 vs 4 the number says whether they need widening).
usage (on an MI355X): python tests/soak_votes_arith.py [frames [config [out.json]]]      -> one JSON line"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import rpg_monocular_pose_estimator_amd as mpe  # noqa: E402
from rpg_monocular_pose_estimator_amd import synth  # noqa: E402
import oracle  # noqa: E402

oracle.build()
from oracle import binding as orc  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
CONFIG = sys.argv[2] if len(sys.argv) > 2 else "C2"
OUT = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/soak_votes_arith_%s.json" % CONFIG
ARITHS = [int(a) for a in os.environ.get("MPE_SOAK_ARITHS", "0,1,3,4").split(",")]
CH = min(N, 32768 if CONFIG != "C3" else 2048)
cfg = synth.CONFIGS[CONFIG]
rows, cols = cfg["rows"], cfg["cols"]
K, D = synth.camera_for(rows, cols)
markers = np.asarray(cfg["markers"])
dev = torch.device("cuda", 0)
h = mpe.Handle(0)
P = mpe.demo_params()
TOL = float(os.environ.get("MPE_SOAK_TOL", "5.0"))
cores = len(os.sched_getaffinity(0))
diff = {a: 0 for a in ARITHS}
cells = {a: 0 for a in ARITHS}
frames_idx = {a: [] for a in ARITHS}
pair = {"1_vs_0": 0, "3_vs_4": 0, "4_vs_0": 0}
t_arith = {a: 0.0 for a in ARITHS}
tot = 0
t0 = time.time()
for part in range(max(1, N // CH)):
    _, spots = synth.make_scenes_batch(cfg, CH, seed=7100 + part)
    frames = synth.render_frames_torch(spots, rows, cols, cfg["spot_sigma"], dev, seed=8100 + part)
    torch.cuda.synchronize()
    det = h.detect_batch(frames, K, D, P)
    nd = det["n"].astype(np.int32)
    dets = det["undist_xy"].reshape(CH, mpe.MAX_DETECTIONS, 2)
    ref = orc.vote_batch(dets, nd, markers, K, TOL, n_threads=cores)
    got = {}
    lst = [dets[i, :nd[i]] for i in range(CH)]
    for a in ARITHS:
        h.set_option("vote_arith", a)
        t1 = time.time()
        got[a] = h.vote_batch(lst, markers, K, TOL)
        t_arith[a] += time.time() - t1
        for i in range(CH):
            if nd[i] < 4:
                continue
            r = ref[i, :nd[i], :len(markers)]
            if not np.array_equal(got[a][i], r):
                diff[a] += 1
                cells[a] += int((got[a][i] != r).sum())
                frames_idx[a].append([part, i])
    h.set_option("vote_arith", 1)
    for name, (x, y) in (("1_vs_0", (1, 0)), ("3_vs_4", (3, 4)), ("4_vs_0", (4, 0))):
        if x in got and y in got:
            pair[name] += sum(1 for i in range(CH) if nd[i] >= 4 and not np.array_equal(got[x][i], got[y][i]))
    tot += CH
    print(part, tot, diff, pair, round(time.time() - t0), flush=True)
out = {"config": CONFIG, "frames": tot, "back_projection_pixel_tolerance": TOL,
       "frames_with_a_histogram_different_from_the_oracle": {str(a): diff[a] for a in ARITHS},
       "differing_cells": {str(a): cells[a] for a in ARITHS},
       "frames_differing_between_arithmetics": pair,
       "differing_frames_first_20": {str(a): frames_idx[a][:20] for a in ARITHS},
       "vote_batch_seconds": {str(a): round(t_arith[a], 2) for a in ARITHS},
       "vote_fixup_items": h.get_option("vote_fixup_items"), "vote_fixup_overflow": h.get_option("vote_fixup_overflow"),
       "arithmetics": {"0": "strict kernel, exact products / cbrt(hypot)", "1": "fast + strict fix-up (default of rounds 4 - 5)",
                       "3": "fast + strict fix-up, libstdc++ / glibc powers", "4": "strict kernel, libstdc++ / glibc powers"}}
os.makedirs(os.path.dirname(OUT) or ".", exist_ok=True)
json.dump(out, open(OUT, "w"), indent=1)
print(json.dumps(out))
