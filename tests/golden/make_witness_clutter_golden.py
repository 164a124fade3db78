#!/usr/bin/env python
"""Generates tests/golden/witness_clutter.npz (and witness_clutter_C4.npz, the same at 1920x1200) — DETECTIONS of cluttered C2 frames (salt noise, a saturated patch, a
ring enclosing the LEDs, a dot grid, distractor spots: rpg_monocular_pose_estimator_amd.synth.make_clutter_frames) by
the INDEPENDENT witness (tests/witness_pipeline.py::find_leds: scipy connected components + Moore boundary tracing;
it shares no code with oracle/ or the product).  Round 5 rewrote the general blob tier — the kernel such frames end
up in — so these vectors pin that tier, and the oracle's findLeds on such frames, to a second implementation.
Consumers: tests/test_golden_cpu.py::test_oracle_against_the_witness_on_cluttered_frames and
tests/test_gpu_parity.py::test_hip_against_the_witness_on_cluttered_frames (-m gpu).

Frames are not stored: (kind, seed) regenerates them bit-exactly; a SHA-1 per frame detects generator drift.

    python tests/golden/make_witness_clutter_golden.py
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

CASES = [(kind, 3, 9200 + i) for i, kind in enumerate(synth.CLUTTER_KINDS)]  # kind, frames, seed
# the same at 1920x1200 (C4): bitmaps of 3 x 288 KB per frame in the general tier, 30 flag words per row instead of 12,
# islands that may not fit the large LDS tier's window
CASES_C4 = [(kind, 1, 9300 + i) for i, kind in enumerate(("salt", "patch", "ring", "grid", "d4"))]
THRESHOLDS = (140, 60)
MAXD = 256


def main():
    make("C2", CASES, "witness_clutter.npz")
    make("C4", CASES_C4, "witness_clutter_C4.npz")


def make(config, cases, name):
    P = synth.DEMO_PARAMS
    kinds, seeds, idx, thrs, sha, n_det, dist, und = [], [], [], [], [], [], [], []
    for kind, n, seed in cases:
        d = synth.make_clutter_frames(kind, n, seed, config)
        for i in range(n):
            for thr in THRESHOLDS:
                u, ds = W.find_leds(d["frames"][i], thr, P["gaussian_sigma"], P["min_blob_area"], P["max_blob_area"],
                                    P["max_width_height_distortion"], P["max_circular_distortion"], d["K"], d["D"])
                assert len(u) <= MAXD, (kind, i, thr, len(u))
                kinds.append(kind); seeds.append(seed); idx.append(i); thrs.append(thr)
                sha.append(hashlib.sha1(d["frames"][i].tobytes()).hexdigest())
                n_det.append(len(u))
                a = np.zeros((MAXD, 2), np.float32); a[:len(ds)] = ds
                b = np.zeros((MAXD, 2)); b[:len(u)] = u
                dist.append(a); und.append(b)
                print(kind, i, thr, len(u))
    out = os.path.join(HERE, name)
    np.savez_compressed(out, made_by="tests/witness_pipeline.py::find_leds", config=config, kind=np.array(kinds), seed=np.array(seeds),
                        frame=np.array(idx), threshold=np.array(thrs), sha1=np.array(sha), n_det=np.array(n_det, np.int32),
                        dist_xy=np.array(dist), undist_xy=np.array(und))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
