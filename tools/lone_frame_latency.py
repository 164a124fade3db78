"""Latency of one frame through the stage entries and the whole brute-force path (host numpy frame in, record out):
   MPE_LIB=<libmpe_hip.so> python tools/lone_frame_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import rpg_monocular_pose_estimator_amd as mpe
from rpg_monocular_pose_estimator_amd import synth
d = synth.make_frames("C2", 64, seed=5)
h = mpe.Handle(0)
P = mpe.demo_params()
for name, fn in (("find_leds (one frame)", lambda i: h.find_leds(d["frames"][i % 64], P, d["K"], d["D"])),
                 ("detect_batch (one frame)", lambda i: h.detect_batch(d["frames"][i % 64][None], d["K"], d["D"], P)),
                 ("estimate_batch (one frame, brute force)", lambda i: h.estimate_batch(d["frames"][i % 64][None], d["markers"], d["K"], d["D"], P)),
                 ("estimate_batch (64 frames)", lambda i: h.estimate_batch(d["frames"], d["markers"], d["K"], d["D"], P))):
    for i in range(20): fn(i)
    best = []
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(200): fn(i)
        best.append((time.perf_counter() - t0) / 200)
    print("%-44s %.4f ms" % (name, min(best) * 1e3), flush=True)
