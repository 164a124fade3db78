#!/usr/bin/env python
"""Generates tests/golden/witness_*.npz — golden vectors of the hot path produced by the INDEPENDENT witness
(tests/witness_pipeline.py: numpy / scipy / Python complex, written from SURVEY.md Appendix A-C; it shares no
code with oracle/ or the product).  Same layout as the oracle-made files of make_golden.py, so the same two
tests consume them: tests/test_golden_cpu.py (oracle vs these vectors) and
tests/test_gpu_parity.py::test_hip_against_committed_golden_vectors (HIP path vs these vectors, -m gpu).

Frames are not stored: (config, seed) regenerates them bit-exactly through rpg_monocular_pose_estimator_amd.synth;
a SHA-1 per frame detects generator drift.  Pure Python: a C3 frame (73 920 P3P solves) takes about a minute.

    python tests/golden/make_witness_golden.py
"""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import witness_pipeline as W  # noqa: E402
from rpg_monocular_pose_estimator_amd import synth  # noqa: E402

CASES = [  # name, config, frames, seed, back-projection tolerance
    ("witness_c1_demo4", "C1", 8, 9101, 5.0),
    ("witness_c2_five_leds", "C2", 16, 9102, 5.0),
    ("witness_c3_eight_leds_tol5", "C3", 2, 9103, 5.0),
    ("witness_c3_eight_leds_tol2", "C3", 3, 9104, 2.0),
    ("witness_c4_1920x1200", "C4", 2, 9105, 5.0),
    # 5 LEDs + 4 distractor spots: 9 detections, C(9,3) x 60 = 5 040 P3P solves per frame, most of them on triples
    # that contain an outlier -- the <= 5-marker voting kernels with more detections than markers
    ("witness_c2_four_distractors", "C2", 16, 9106, 5.0, 4),
]
MAXD, MAXM = 32, 16


def build(name, config, n, seed, tol, n_distractors=None):
    d = synth.make_frames(config if n_distractors is None else dict(synth.CONFIGS[config], n_distractors=n_distractors),
                          n, seed)
    P = dict(synth.DEMO_PARAMS, back_projection_pixel_tolerance=tol)
    n_m = len(d["markers"])
    out = dict(config=config, seed=seed, n=n, tol=tol, made_by="tests/witness_pipeline.py",
               sha1=np.array([hashlib.sha1(f.tobytes()).hexdigest() for f in d["frames"]]),
               n_det=np.zeros(n, np.int32), dist_xy=np.zeros((n, MAXD, 2), np.float32),
               undist_xy=np.zeros((n, MAXD, 2)), hist=np.zeros((n, MAXD, MAXM), np.uint32),
               corr=np.zeros((n, MAXM, 2), np.uint32), n_corr=np.zeros(n, np.int32),
               status=np.zeros(n, np.int32), T=np.zeros((n, 4, 4)), cov=np.zeros((n, 6, 6)),
               gn_iterations=np.zeros(n, np.int32))
    if n_distractors is not None:
        out["n_distractors"] = n_distractors
    for i in range(n):
        t0 = time.time()
        r = W.estimate_frame(d["frames"][i], d["markers"], d["K"], d["D"], P)
        k = len(r["und"])
        out["n_det"][i] = k
        out["dist_xy"][i, :k] = r["dist"]
        out["undist_xy"][i, :k] = r["und"]
        out["hist"][i, :k, :n_m] = r["hist"]
        out["n_corr"][i] = len(r["corr"])
        out["corr"][i, :len(r["corr"])] = r["corr"]
        out["status"][i] = r["status"]
        out["T"][i] = r["T"]
        out["cov"][i] = r["cov"]
        out["gn_iterations"][i] = r["gn_iterations"]
        print("  %s frame %d: %d detections, status %d, %.1f s" % (name, i, k, r["status"], time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "poses", int((out["status"] == 0).sum()), "/", n, flush=True)


if __name__ == "__main__":
    only = sys.argv[1:]
    for c in CASES:
        if not only or c[0] in only:
            build(*c)
