#!/usr/bin/env python
"""Generates tests/golden/*.npz — golden vectors of the hot path produced by the CPU oracle.

The reference cannot be built or imported here (needs Eigen/OpenCV/ROS) and ships no fixtures, so
the vectors come from the oracle (pinned by tests/test_oracle_kat.py).  Frames are not stored: each
file holds the scene description (config, seed) from which rpg_monocular_pose_estimator_amd.synth
regenerates them bit-exactly, plus a SHA-1 per frame to detect generator drift.

    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle  # noqa: E402
from rpg_monocular_pose_estimator_amd import synth  # noqa: E402

CASES = [  # name, config, frames, seed, back-projection tolerance
    ("c1_demo4", "C1", 8, 9001, 5.0),
    ("c2_five_leds", "C2", 24, 9002, 5.0),
    ("c3_eight_leds_tol5", "C3", 3, 9003, 5.0),
    ("c3_eight_leds_tol2", "C3", 4, 9004, 2.0),
    ("c4_1920x1200", "C4", 2, 9005, 5.0),
]
MAXD, MAXM = 32, 16


def build(name, config, n, seed, tol):
    d = synth.make_frames(config, n, seed)
    P = oracle.make_params(back_projection_pixel_tolerance=tol)
    n_m = len(d["markers"])
    out = dict(config=config, seed=seed, n=n, tol=tol,
               sha1=np.array([hashlib.sha1(f.tobytes()).hexdigest() for f in d["frames"]]),
               n_det=np.zeros(n, np.int32), dist_xy=np.zeros((n, MAXD, 2), np.float32),
               undist_xy=np.zeros((n, MAXD, 2)), hist=np.zeros((n, MAXD, MAXM), np.uint32),
               corr=np.zeros((n, MAXM, 2), np.uint32), n_corr=np.zeros(n, np.int32),
               status=np.zeros(n, np.int32), T=np.zeros((n, 4, 4)), cov=np.zeros((n, 6, 6)),
               gn_iterations=np.zeros(n, np.int32))
    for i in range(n):
        und, dist = oracle.find_leds(d["frames"][i], P, d["K"], d["D"])
        r = oracle.solve_bruteforce(und, d["markers"], d["K"], P)
        k = len(und)
        out["n_det"][i] = k
        out["dist_xy"][i, :k] = dist
        out["undist_xy"][i, :k] = und
        out["hist"][i, :k, :n_m] = r["hist"]
        out["n_corr"][i] = r["n_corr"]
        out["corr"][i, :r["n_corr"]] = r["corr"]
        out["status"][i] = r["status"]
        out["T"][i] = r["T"]
        out["cov"][i] = r["cov"]
        out["gn_iterations"][i] = r["gn_iterations"]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "poses", int((out["status"] == 0).sum()), "/", n)


SEQUENCES = [  # name, config, frames, seed, drop-out frames
    ("seq_c1_demo_replay", "C1", 40, 9101, (12, 13, 27)),
    ("seq_c2_tracking", "C2", 30, 9102, (9,)),
]


def build_sequence(name, config, n, seed, dropout):
    """The stateful estimator (tracking path) over a smooth synthetic trajectory = the ROS-free
    counterpart of the demo.launch bag replay (BASELINE config C1)."""
    d = synth.make_sequence(config, n, seed=seed, dropout=dropout)
    tr = oracle.Tracker(d["markers"], d["K"], d["D"], oracle.make_params())
    out = dict(config=config, seed=seed, n=n, dropout=np.array(dropout, np.int32), times=d["times"],
               sha1=np.array([hashlib.sha1(f.tobytes()).hexdigest() for f in d["frames"]]),
               updated=np.zeros(n, np.int32), roi=np.zeros((n, 4), np.int32), it=np.zeros(n, np.int32),
               n_det=np.zeros(n, np.int32), n_corr=np.zeros(n, np.int32), bruteforce=np.zeros(n, np.int32),
               T=np.zeros((n, 4, 4)), cov=np.zeros((n, 6, 6)))
    for k in range(n):
        r = tr.estimate(d["frames"][k], d["times"][k])
        out["updated"][k] = r["updated"]
        out["roi"][k] = r["roi"]
        out["it"][k] = r["it_since_initialized"]
        out["n_det"][k] = r["n_det"]
        out["n_corr"][k] = r["n_corr"]
        out["bruteforce"][k] = r["used_bruteforce"]
        out["T"][k] = r["T"]
        out["cov"][k] = r["cov"]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "poses", int(out["updated"].sum()), "/", n, "tracked", int((out["updated"] & (1 - out["bruteforce"])).sum()))


if __name__ == "__main__":
    oracle.build()
    for c in CASES:
        build(*c)
    for c in SEQUENCES:
        build_sequence(*c)
