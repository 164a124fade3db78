import sys, json, time
sys.path.insert(0, '/root/repo')
import numpy as np
import rpg_monocular_pose_estimator_amd as mpe
from rpg_monocular_pose_estimator_amd import synth
seq = synth.make_sequence("C2", 300, seed=5)
h = mpe.Handle(0)
t = mpe.Tracker(h, seq["markers"], seq["K"], seq["D"], mpe.demo_params())
t.run_sequence(seq["frames"][:8], seq["times"][:8]); t.reset()
h.set_option("track_phase_clocks", 1)
t0 = time.perf_counter()
rec, info = t.run_sequence(seq["frames"], seq["times"])
dt = (time.perf_counter() - t0) / len(seq["frames"])
cyc = [h.get_option("track_phase_cycles_%d" % i) for i in range(4)]
print(json.dumps({"ms_per_frame": dt * 1e3, "phase_cycles_scan_blobs_validate_refine": cyc, "sum": sum(cyc), "poses": int((rec["status"] == 0).sum())}))
