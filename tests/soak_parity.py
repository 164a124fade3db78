"""End-to-end parity soak with forensics: N frames through the HIP path (default fused schedule) and through the
oracle.  Every frame whose status differs, or whose poses differ by more than 1e-4 m, is SAVED (pixels, detections,
both records -> <out>.npz) and CLASSIFIED (tests/forensics.py::classify_end_to_end): the difference is traced to the
hypotheses of the voting, or to the P3P solves of the validation, on which the two paths differ, and each of those
must be a witnessed instability of the reference algorithm itself (the oracle's own answer moves under a 1-ulp change
of an input) or sit in the cancellation corner of its Ferrari solver.  Exit code 1 if one stays unexplained.
usage (on an MI355X): [MPE_VOTE_ARITH=0|1] [MPE_SOAK_CLUTTER=d4|d16|salt] python tests/soak_parity.py [frames [config [chunk [out_prefix]]]]
MPE_VOTE_ARITH: option "vote_arith" of the handle (3 = the default since round 6; see include/mpe.h)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import rpg_monocular_pose_estimator_amd as mpe  # noqa: E402
from rpg_monocular_pose_estimator_amd import synth  # noqa: E402
import oracle  # noqa: E402
import forensics  # noqa: E402

oracle.build()
from oracle import binding as orc  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
CONFIG = sys.argv[2] if len(sys.argv) > 2 else "C2"
CH = min(N, int(sys.argv[3]) if len(sys.argv) > 3 else 65536)
ARITH = int(os.environ.get("MPE_VOTE_ARITH", "3"))
OUT = sys.argv[4] if len(sys.argv) > 4 else "gpurun_out/soak_parity_%s_arith%d" % (CONFIG, ARITH)
TOL_PX = float(os.environ.get("MPE_BACK_TOL", "5"))
cfg = dict(synth.CONFIGS[CONFIG])
CLUTTER = os.environ.get("MPE_SOAK_CLUTTER", "")  # d4 / d16: distractor spots; salt: 0.05 % isolated saturated pixels
if CLUTTER in ("d4", "d16"):
    cfg["n_distractors"] = int(CLUTTER[1:])
rows, cols = cfg["rows"], cfg["cols"]
K, D = synth.camera_for(rows, cols)
markers = np.asarray(cfg["markers"])
dev = torch.device("cuda", 0)
h = mpe.Handle(0)
P = mpe.demo_params(back_projection_pixel_tolerance=TOL_PX)
PO = orc.make_params(back_projection_pixel_tolerance=TOL_PX)
h.set_option("vote_arith", ARITH)
st = torch.cuda.Stream(device=dev)
h.set_stream(st.cuda_stream)
hf = mpe.Handle(0)  # a second handle for the forensics (own stream, default options except vote_arith)
hf.set_option("vote_arith", ARITH)
cores = len(os.sched_getaffinity(0))
tot = st_mis = pose_mis = n_pose = 0
worst = 0.0
saved = []
t0 = time.time()
for part in range(max(1, N // CH)):
    _, spots = synth.make_scenes_batch(cfg, CH, seed=7000 + part)
    frames = synth.render_frames_torch(spots, rows, cols, cfg["spot_sigma"], dev, seed=8000 + part)
    if CLUTTER == "salt":
        g = torch.Generator(device=dev)
        g.manual_seed(9000 + part)
        for a in range(0, CH, 1024):  # (in slices: the random field of a whole chunk is 4 bytes a pixel)
            fr = frames[a:a + 1024]
            fr[torch.rand(fr.shape, generator=g, device=dev) < 0.0005] = 255
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        out = torch.zeros(CH * mpe.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        h.estimate_batch_device(frames.data_ptr(), CH, rows, cols, markers, K, D, P, out.data_ptr())
    st.synchronize()
    got = np.frombuffer(out.cpu().numpy().tobytes(), mpe.RESULT_DTYPE)
    host_frames = frames.cpu().numpy()
    ref = orc.estimate_batch(host_frames, markers, K, D, PO, n_threads=cores)
    tot += CH
    bad_status = got["status"] != ref["status"]
    st_mis += int(bad_status.sum())
    ok = (got["status"] == 0) & (ref["status"] == 0)
    d = np.zeros(CH)
    d[ok] = np.linalg.norm(got["T"][ok][:, [3, 7, 11]] - ref["T"][ok][:, [3, 7, 11]], axis=1)
    n_pose += int(ok.sum())
    pose_mis += int((d > 1e-4).sum())
    worst = max(worst, float(d.max()) if ok.any() else 0.0)
    for i in np.nonzero(bad_status | (d > 1e-4))[0]:
        und, _ = orc.find_leds(host_frames[i], PO, K, D)
        v = forensics.classify_end_to_end(hf, orc, und, markers, K, P, PO)
        saved.append({"part": part, "frame": int(i), "hip_status": int(got["status"][i]), "oracle_status": int(ref["status"][i]),
                      "dpos_m": float(d[i]), "verdict": v, "pixels": host_frames[i].copy(), "det": np.asarray(und, float)})
    print(part, tot, st_mis, pose_mis, worst, round(time.time() - t0), flush=True)
unexplained = [s for s in saved if not s["verdict"]["unstable"]]
meta = [{k: v for k, v in s.items() if k not in ("pixels", "det")} for s in saved]
if saved:
    os.makedirs(os.path.dirname(OUT) or ".", exist_ok=True)
    np.savez_compressed(OUT + ".npz", **{"pixels_%d" % k: s["pixels"] for k, s in enumerate(saved)},
                        **{"det_%d" % k: s["det"] for k, s in enumerate(saved)}, meta=json.dumps(meta))
print(json.dumps({"config": CONFIG, "clutter": CLUTTER or None, "frames": tot, "back_projection_pixel_tolerance": TOL_PX, "status_mismatches": st_mis,
                  "poses_compared": n_pose, "pose_mismatches_gt_1e-4m": pose_mis, "worst_position_difference_m": worst,
                  "mismatches_classified_unstable": len(saved) - len(unexplained),
                  "mismatches_unexplained": len(unexplained),
                  "mismatching_frames_saved_to": (OUT + ".npz") if saved else None, "verdicts": meta,
                  "schedule": h.get_option("last_schedule"), "vote_arith": ARITH,
                  "vote_arith_meaning": "3 = fast kernel + strict fix-up with the reference library's complex powers (default); 1 the same with exact powers; 4 / 0 their strict kernels"}))
sys.exit(1 if unexplained else 0)
