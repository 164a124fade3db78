#!/usr/bin/env python
"""Generates tests/golden/witness_seq_*.npz — golden vectors of the TRACKING path (estimateBodyPose as a state machine:
prediction, ROI, nearest-neighbour correspondences, whole-image retry, brute-force fallback) produced by the independent
witness (tests/witness_pipeline.py::Tracker, written from pose_estimator.cpp:62-147 / 232-276 / 372-392 / 794-848 /
996-1064 and led_detector.cpp:114-224 — it shares no code with oracle/ or the product).  Consumed by
tests/test_golden_cpu.py (the oracle's tracker vs these vectors) and tests/test_gpu_parity.py (the HIP tracker vs these
vectors, -m gpu).  Frames are not stored: (config, frames, seed, dropout) regenerates them through
rpg_monocular_pose_estimator_amd.synth.make_sequence; a SHA-1 per frame detects generator drift.

    python tests/golden/make_witness_seq_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import witness_pipeline as W  # noqa: E402
from rpg_monocular_pose_estimator_amd import synth  # noqa: E402

CASES = [  # name, config, frames, seed, LED drop-out frames (fewer than 4 LEDs visible: whole-image retry, stale state)
    ("witness_seq_c2_tracking", "C2", 36, 9201, (9, 20, 21)),
    ("witness_seq_c1_demo4", "C1", 30, 9202, (12,)),
    ("witness_seq_c2_long_dropout", "C2", 24, 9203, (6, 7, 8, 9, 10)),
    # 0.1 % / 0.3 % of every frame's pixels saturated: isolated bright pixels in the ROI and around it (the blob tiers
    # behind the first one, with per-frame ROI windows on the device), a drop-out in the middle
    ("witness_seq_c2_salt", "C2", 24, 9204, (11,), 0.001),
    ("witness_seq_c2_salt_dense", "C2", 20, 9205, (), 0.003),
]


def build(name, config, n, seed, dropout, salt=0.0):
    seq = synth.make_sequence(config, n, seed=seed, dropout=dropout, salt=salt)
    tr = W.Tracker(seq["markers"], seq["K"], seq["D"], dict(synth.DEMO_PARAMS))
    out = dict(config=config, seed=seed, n=n, dropout=np.array(dropout, np.int32), made_by="tests/witness_pipeline.py::Tracker",
               sha1=np.array([hashlib.sha1(f.tobytes()).hexdigest() for f in seq["frames"]]),
               updated=np.zeros(n, np.int32), roi=np.zeros((n, 4), np.int32), it_since_initialized=np.zeros(n, np.int32),
               n_det=np.zeros(n, np.int32), n_corr=np.zeros(n, np.int32), used_bruteforce=np.zeros(n, np.int32),
               T=np.zeros((n, 4, 4)), cov=np.zeros((n, 6, 6)))
    if salt:
        out["salt"] = salt
    for k in range(n):
        r = tr.estimate(seq["frames"][k], seq["times"][k])
        out["updated"][k] = int(r["updated"])
        out["roi"][k] = r["roi"]
        out["it_since_initialized"][k] = r["it_since_initialized"]
        out["n_det"][k] = r["n_det"]
        out["n_corr"][k] = r["n_corr"]
        out["used_bruteforce"][k] = int(r["used_bruteforce"])
        out["T"][k] = r["T"]
        out["cov"][k] = r["cov"]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "updated", int(out["updated"].sum()), "of", n, "brute force", int(out["used_bruteforce"].sum()),
          "whole-image frames", int((out["roi"][:, 2] == seq["cols"]).sum()))


if __name__ == "__main__":
    only = sys.argv[1:]
    for c in CASES:
        if not only or c[0] in only:
            build(*c)
