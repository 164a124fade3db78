import sys; sys.path.insert(0,'/root/repo')
import numpy as np, time, torch
import rpg_monocular_pose_estimator_amd as mpe
from rpg_monocular_pose_estimator_amd import synth
d = synth.make_clutter_frames("salt", 64, seed=3)
h = mpe.Handle(0)
fr = torch.from_numpy(d["frames"]).cuda().repeat(64,1,1).contiguous()  # 4096 frames
for k in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter()
    r = h.detect_batch(fr, d["K"], d["D"], mpe.demo_params())
    torch.cuda.synchronize(); print("detect_batch 4096 salt frames: %.2f ms" % ((time.perf_counter()-t0)*1e3), int(r["n"].mean()))
