"""Latency of ONE cluttered frame in the blob tiers (the general tier's worst cases): the noise frames of
tests/test_gpu_parity.py::test_general_tier_band_scan_on_random_clutter, each detected alone and as 256 copies.
  MPE_LIB=<libmpe_hip.so> [MPE_MAX_DET=32 for a round-5 library] python tools/dense_frame_probe.py [rows cols]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import rpg_monocular_pose_estimator_amd as mpe
from rpg_monocular_pose_estimator_amd import synth

rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (480, 752)
rng = np.random.default_rng(77)
K, D = synth.camera_for(rows, cols)
h = mpe.Handle(0)
P = mpe.demo_params()
for dens in (0.0002, 0.0005, 0.002, 0.01, 0.03):
    f = (rng.random((rows, cols)) < dens).astype(np.uint8) * rng.integers(150, 256, (rows, cols)).astype(np.uint8)
    for thr in (140, 60):
        P.threshold_value = thr
        for copies in (1, 256):
            fr = torch.from_numpy(f).cuda()[None].repeat(copies, 1, 1).contiguous()
            best = 1e9
            for k in range(4):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                r = h.detect_batch(fr, K, D, P)
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            print("density %.4f thr %3d  %3d frame(s): %8.3f ms  (%d detections, status %d)" % (
                dens, thr, copies, best * 1e3, int(r["n"][0]), int(r["status"][0])), flush=True)
