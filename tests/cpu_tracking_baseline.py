#!/usr/bin/env python
"""Frames/s of the CPU oracle's stateful estimator (tracking path) on one synthetic stream, one thread — the CPU
figure next to bench_streams.py's GPU figure in DESIGN.md.  (The oracle is test infrastructure: it lives under
tests/ and is never used by the product or by bench_streams.py.)   usage: python tests/cpu_tracking_baseline.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from rpg_monocular_pose_estimator_amd import synth  # noqa: E402


def main():
    oracle.build()
    from oracle import binding as orc
    d = synth.make_sequence("C2", 40, seed=900)
    order = np.concatenate([np.arange(40), np.arange(38, 0, -1)])
    idx = np.resize(order, 400)
    frames = np.ascontiguousarray(d["frames"][idx])
    tr = orc.Tracker(d["markers"], d["K"], d["D"], orc.make_params())
    t0 = time.perf_counter()
    n_pose = sum(int(tr.estimate(frames[k], 0.02 * k)["updated"]) for k in range(len(frames)))
    dt = time.perf_counter() - t0
    print(json.dumps({"cpu_tracking_fps_one_thread": len(frames) / dt, "frames": len(frames), "poses": n_pose}))


if __name__ == "__main__":
    main()
