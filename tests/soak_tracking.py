"""Tracking-path parity soak (BASELINE configs[4]): S independent camera streams of F frames
each through the lock-step batch replay of the HIP path (mpe_tracker_run_sequences_batch: ROI detection,
findCorrespondences, checkCorrespondences, optimisePose, whole-image retries, brute-force re-initialisation — one device
submission per time step for all streams) and, stream by stream, through the oracle's restatement of
PoseEstimator::estimateBodyPose (pose_estimator.cpp:62-147).  Every frame of every stream is compared: updated flag,
ROI rectangle, it_since_initialized, detection / correspondence counts, brute-force flag: equal; pose <= 1e-4 m /
1e-3 rad.  A stream that diverges is reported with its first differing frame (everything after it follows from the
state the two estimators no longer share) and its frames are saved.  Exit code 1 on any difference.
usage (on an MI355X): python tests/soak_tracking.py [streams [frames [config [out_prefix]]]]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import rpg_monocular_pose_estimator_amd as mpe  # noqa: E402
from rpg_monocular_pose_estimator_amd import synth  # noqa: E402
import oracle  # noqa: E402
from util import pose_diff  # noqa: E402

oracle.build()
from oracle import binding as orc  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
F = int(sys.argv[2]) if len(sys.argv) > 2 else 160
CONFIG = sys.argv[3] if len(sys.argv) > 3 else "C2"
OUT = sys.argv[4] if len(sys.argv) > 4 else "gpurun_out/soak_tracking_%s" % CONFIG
POS_TOL_M, ROT_TOL_RAD = 1e-4, 1e-3
SALT = float(os.environ.get("MPE_SOAK_SALT", "0"))  # fraction of saturated pixels on every second stream (the general blob tier inside ROIs)
BASE = 40  # a smooth 40-frame trajectory played forwards and backwards keeps the target inside the image

rng = np.random.default_rng(2024)
order = np.concatenate([np.arange(BASE), np.arange(BASE - 2, 0, -1)])
idx = np.resize(order, F)
seqs = []
t0 = time.time()
for s in range(S):
    # LED drop-outs on two thirds of the streams: two LEDs only in those frames (whole-image retry, re-initialisation)
    drop = tuple(int(x) for x in rng.choice(np.arange(4, BASE - 2), size=int(rng.integers(1, 4)), replace=False)) if s % 3 else ()
    d = synth.make_sequence(CONFIG, BASE, seed=7100 + s, dropout=drop, salt=SALT if s % 2 else 0.0)
    seqs.append(dict(frames=np.ascontiguousarray(d["frames"][idx]), times=np.arange(F) * 0.02, markers=d["markers"],
                     K=d["K"], D=d["D"], drop=drop, rows=d["frames"].shape[1], cols=d["frames"].shape[2]))
t_make = time.time() - t0

h = mpe.Handle(0)
P = mpe.demo_params()
trackers = [mpe.Tracker(h, seqs[0]["markers"], seqs[0]["K"], seqs[0]["D"], P) for _ in range(S)]
t0 = time.time()
rec, info = mpe.tracker_run_sequences_batch(trackers, [q["frames"] for q in seqs], seqs[0]["times"])
t_hip = time.time() - t0

t0 = time.time()
frames_compared = poses_compared = n_brute = n_retry = n_roi = 0
worst_pos = worst_rot = 0.0
diverged = []
for s in range(S):
    to = orc.Tracker(seqs[s]["markers"], seqs[s]["K"], seqs[s]["D"], orc.make_params())
    for k in range(F):
        ro = to.estimate(seqs[s]["frames"][k], seqs[s]["times"][k])
        why = None
        if (rec["status"][s, k] == 0) != ro["updated"]:
            why = "updated flag"
        elif tuple(int(x) for x in info[s, k, 0:4]) != tuple(ro["roi"]):
            why = "roi"
        elif int(info[s, k, 4]) != ro["it_since_initialized"]:
            why = "it_since_initialized"
        elif int(info[s, k, 5]) != ro["n_det"] or int(info[s, k, 6]) != ro["n_corr"]:
            why = "detection / correspondence count"
        elif bool(info[s, k, 7]) != ro["used_bruteforce"]:
            why = "brute-force flag"
        elif ro["updated"]:
            dp, dr = pose_diff(rec["T"][s, k].reshape(4, 4), ro["T"])
            worst_pos, worst_rot = max(worst_pos, dp), max(worst_rot, dr)
            poses_compared += 1
            if dp > POS_TOL_M or dr > ROT_TOL_RAD:
                why = "pose (%.3g m, %.3g rad)" % (dp, dr)
        if why:
            diverged.append({"stream": s, "first_differing_frame": k, "what": why, "dropout_frames": list(seqs[s]["drop"])})
            break
        frames_compared += 1
        n_brute += int(info[s, k, 7])
        n_retry += int(k > 0 and info[s, k, 2] == seqs[s]["cols"] and info[s, k, 4] >= 1)
        n_roi += int(info[s, k, 2] < seqs[s]["cols"])
t_orc = time.time() - t0

if diverged:
    os.makedirs(os.path.dirname(OUT) or ".", exist_ok=True)
    np.savez_compressed(OUT + ".npz", **{"frames_stream_%d" % d["stream"]: seqs[d["stream"]]["frames"] for d in diverged[:4]})
print(json.dumps({
    "config": CONFIG, "streams": S, "frames_per_stream": F, "tracked_frames": S * F,
    "frames_compared_equal": frames_compared, "streams_diverged": len(diverged), "diverged": diverged[:16],
    "poses_compared": poses_compared, "worst_position_difference_m": worst_pos, "worst_rotation_difference_rad": worst_rot,
    "bruteforce_initialisations": n_brute, "whole_image_retries": n_retry, "roi_frames": n_roi,
    "streams_with_dropouts": sum(1 for q in seqs if q["drop"]),
    "compared": "updated flag, ROI rectangle, it_since_initialized, n_det, n_corr, brute-force flag: equal; pose <= 1e-4 m / 1e-3 rad",
    "seconds": {"make_sequences": round(t_make, 1), "hip_lockstep_replay": round(t_hip, 2), "oracle": round(t_orc, 1)},
    "frames_saved_to": (OUT + ".npz") if diverged else None}))
for t in trackers:
    t.close()
h.close()
sys.exit(1 if diverged else 0)
