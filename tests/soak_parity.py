"""End-to-end parity soak: N frames through the HIP path (default fused schedule) and through the oracle.
usage (on an MI355X): [MPE_VOTE_ARITH=0|1] python tests/soak_parity.py [frames [config [chunk]]]
MPE_VOTE_ARITH: option "vote_arith" of the handle (1 = fast voting arithmetic, the default; 0 = strict).
python tests/soak_parity.py 1048576   -> profiles/round2_parity_soak_*.json"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rpg_monocular_pose_estimator_amd as mpe
from rpg_monocular_pose_estimator_amd import synth
import oracle
oracle.build()
from oracle import binding as orc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
CONFIG = sys.argv[2] if len(sys.argv) > 2 else "C2"
CH = min(N, int(sys.argv[3]) if len(sys.argv) > 3 else 65536)
cfg = synth.CONFIGS[CONFIG]; rows, cols = cfg["rows"], cfg["cols"]
K, D = synth.camera_for(rows, cols); markers = np.asarray(cfg["markers"])
dev = torch.device("cuda", 0)
h = mpe.Handle(0); P = mpe.demo_params()
ARITH = int(os.environ.get("MPE_VOTE_ARITH", "1"))
h.set_option("vote_arith", ARITH)
st = torch.cuda.Stream(device=dev); h.set_stream(st.cuda_stream)
tot = st_mis = pose_mis = n_pose = 0
worst = 0.0
t0 = time.time()
for part in range(N // CH):
    _, spots = synth.make_scenes_batch(cfg, CH, seed=7000 + part)
    frames = synth.render_frames_torch(spots, rows, cols, cfg["spot_sigma"], dev, seed=8000 + part)
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        out = torch.zeros(CH * mpe.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        h.estimate_batch_device(frames.data_ptr(), CH, rows, cols, markers, K, D, P, out.data_ptr())
    st.synchronize()
    got = np.frombuffer(out.cpu().numpy().tobytes(), mpe.RESULT_DTYPE)
    ref = orc.estimate_batch(frames.cpu().numpy(), markers, K, D, orc.make_params(), n_threads=16)
    tot += CH
    st_mis += int((got["status"] != ref["status"]).sum())
    ok = (got["status"] == 0) & (ref["status"] == 0)
    d = np.linalg.norm(got["T"][ok][:, [3, 7, 11]] - ref["T"][ok][:, [3, 7, 11]], axis=1)
    n_pose += int(ok.sum())
    pose_mis += int((d > 1e-4).sum())
    worst = max(worst, float(d.max()) if len(d) else 0.0)
    print(part, tot, st_mis, pose_mis, worst, round(time.time() - t0), flush=True)
print(json.dumps({"config": CONFIG, "frames": tot, "status_mismatches": st_mis, "poses_compared": n_pose, "pose_mismatches_gt_1e-4m": pose_mis,
                  "worst_position_difference_m": worst, "schedule": h.get_option("last_schedule"),
                  "vote_arith": ARITH, "vote_arith_meaning": "1 = fast (Newton-Raphson div / sqrt, Newton cube root), 0 = strict (IEEE, validation kernel's P3P)"}))
