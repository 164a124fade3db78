#!/usr/bin/env python3
"""What fills the window between two voting launches on the caller's stream?  From a rocprofv3 --kernel-trace csv of the
headline bench: for every scan-carrying voting launch, the kernels of the blob chain that ran since the previous one
(k1b_blobs<K1bSmall>, k1b_blobs_list<K1bLarge>, k1b_general) as offsets from the end of the previous voting launch.
  usage: summarize_period.py <dir with *_kernel_trace.csv> [out.json]"""
import csv
import glob
import json
import os
import sys

import numpy as np


def main():
    rows = []
    for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "mpe::" in n:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
    rows.sort()
    votes = [(a, b) for a, b, n in rows if "k2_vote<true" in n]
    names = {"blobs": "k1b_blobs<mpe::K1bSmall>", "list": "k1b_blobs_list<mpe::K1bLarge>", "general": "k1b_general"}
    acc = {k: {"start": [], "dur": []} for k in names}
    win, vd = [], []
    for i in range(1, len(votes)):
        p_end, v_start = votes[i - 1][1], votes[i][0]
        if v_start - p_end > 3e6:  # (a step boundary with other work in between: not a window)
            continue
        win.append((v_start - p_end) * 1e-3)
        vd.append((votes[i][1] - votes[i][0]) * 1e-3)
        for k, pat in names.items():
            hit = [(a, b) for a, b, n in rows if pat in n and p_end - 50000 <= a < v_start]
            if hit:
                a, b = hit[-1]
                acc[k]["start"].append((a - p_end) * 1e-3)
                acc[k]["dur"].append((b - a) * 1e-3)
    out = {"windows": len(win), "window_us_mean": float(np.mean(win)), "window_us_median": float(np.median(win)),
           "vote_us_mean": float(np.mean(vd)),
           "chain": {k: {"n": len(v["start"]), "start_after_previous_vote_end_us": float(np.mean(v["start"])) if v["start"] else None,
                         "duration_us": float(np.mean(v["dur"])) if v["dur"] else None} for k, v in acc.items()}}
    c = out["chain"]
    if all(c[k]["n"] for k in c):
        out["vote_start_after_general_end_us"] = out["window_us_mean"] - (c["general"]["start_after_previous_vote_end_us"] + c["general"]["duration_us"])
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
