#!/usr/bin/env python
"""Experiment (round 5): do two INTERLEAVED pipelines fill each other's blob windows?  Two handles, two caller
streams, each streaming half of the batch (schedule 6 inside each); against one handle on the whole batch.
  python profiles/experiments/two_pipelines.py [frames_total] [steps]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rpg_monocular_pose_estimator_amd as mpe
from rpg_monocular_pose_estimator_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 15
dev = torch.device("cuda", 0)
cfg = synth.CONFIGS["C2"]
rows, cols = cfg["rows"], cfg["cols"]
K, D = synth.camera_for(rows, cols)
markers = np.asarray(cfg["markers"])
_, spots = synth.make_scenes_batch(cfg, B, seed=1000)
frames = synth.render_frames_torch(spots, rows, cols, cfg["spot_sigma"], dev, seed=77)
P = mpe.demo_params()
rec = mpe.RESULT_DTYPE.itemsize


def run(n_pipes, opts=()):
    per = B // n_pipes
    hs, streams, outs = [], [], []
    for i in range(n_pipes):
        h = mpe.Handle(0)
        s = torch.cuda.Stream(device=dev)
        h.set_stream(s.cuda_stream)
        for k, v in opts:
            h.set_option(k, v)
        hs.append(h)
        streams.append(s)
        outs.append(torch.zeros(per * rec, dtype=torch.uint8, device=dev))

    def step():
        for i, h in enumerate(hs):
            fr = frames[i * per:(i + 1) * per]
            with torch.cuda.stream(streams[i]):
                h.estimate_batch_device_submit(fr.data_ptr(), per, rows, cols, markers, K, D, P, outs[i].data_ptr(),
                                               fr.data_ptr(), per)
                h.estimate_batch_device_collect(streams[i].cuda_stream)
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / STEPS
    sched = [h.get_option("last_schedule") for h in hs]
    conc = [h.get_option("streams_concurrent") for h in hs]
    for h in hs:
        h.close()
    return {"pipelines": n_pipes, "opts": dict(opts), "ms_per_step": dt * 1e3, "fps": B / dt, "schedules": sched,
            "streams_concurrent": conc}


res = [run(1), run(2), run(1), run(2), run(2, (("scan_split_pct", 0),)), run(3)]
print(json.dumps({"frames_total": B, "steps": STEPS, "runs": res}))
