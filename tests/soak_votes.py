"""Vote-histogram parity soak: detection sets of N synthetic frames (HIP detection, bit-exact vs the oracle) through
the HIP voting kernel in BOTH arithmetics (option "vote_arith": 1 fast, 0 strict) and through the oracle's
voting (frame-parallel on the host cores); counts the frames whose histogram differs anywhere.
usage (on an MI355X): python tests/soak_votes.py [frames [config]]      -> one JSON line"""
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

N = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
CONFIG = sys.argv[2] if len(sys.argv) > 2 else "C2"
CH = min(N, 32768)
cfg = synth.CONFIGS[CONFIG]
rows, cols = cfg["rows"], cfg["cols"]
K, D = synth.camera_for(rows, cols)
markers = np.asarray(cfg["markers"])
dev = torch.device("cuda", 0)
h = mpe.Handle(0)
P = mpe.demo_params()
diff = {0: 0, 1: 0}
cells = {0: 0, 1: 0}
tot = 0
t0 = time.time()
for part in range(N // CH):
    _, spots = synth.make_scenes_batch(cfg, CH, seed=7100 + part)
    frames = synth.render_frames_torch(spots, rows, cols, cfg["spot_sigma"], dev, seed=8100 + part)
    torch.cuda.synchronize()
    det = h.detect_batch(frames, K, D, P)
    nd = det["n"].astype(np.int32)
    dets = det["undist_xy"].reshape(CH, mpe.MAX_DETECTIONS, 2)
    ref = orc.vote_batch(dets, nd, markers, K, 5.0, n_threads=16)
    for arith in (1, 0):
        h.set_option("vote_arith", arith)
        got = h.vote_batch([dets[i, :nd[i]] for i in range(CH)], markers, K, 5.0)
        for i in range(CH):
            r = ref[i, :nd[i], :len(markers)] if nd[i] >= 4 else np.zeros((nd[i], len(markers)), np.uint32)
            g = got[i] if nd[i] >= 4 else np.zeros_like(r)
            if not np.array_equal(g, r):
                diff[arith] += 1
                cells[arith] += int((g != r).sum())
    tot += CH
    print(part, tot, diff, round(time.time() - t0), flush=True)
print(json.dumps({"config": CONFIG, "frames": tot, "p3p_solves_per_frame": "C(n_det,3) x P(n_markers,3)",
                  "frames_with_a_different_histogram": {"fast (vote_arith 1)": diff[1], "strict (vote_arith 0)": diff[0]},
                  "differing_cells": {"fast": cells[1], "strict": cells[0]}}))
