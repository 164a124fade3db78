#!/usr/bin/env python3
"""Which launches of the scan-carrying voting kernel are the slow ones?  (its duration under the tracer
has sigma 0.56 ms on a 1.81 ms mean, max 3.9 ms — "which launches are the 2x ones is not explained anywhere".)
Reads a rocprofv3 --kernel-trace csv that holds ONLY that kernel (collect_round5.sh traces it alone), puts the launches
in start order, and reports the duration by position within a step (8 launches per streaming submission: sub-batch
slots 0..7), the outliers (> 1.5 x median) with their slot and step, and the gap in front of every launch.
  usage: summarize_vote_trace.py <dir with *_kernel_trace.csv> <launches per step> [out.json]"""
import csv
import glob
import json
import os
import sys

import numpy as np


def main():
    d, per = sys.argv[1], int(sys.argv[2])
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k2_vote<true" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort()
    if not rows:
        sys.exit("no k2_vote<true launches under " + d)
    st = np.array([a for a, _ in rows], float)
    en = np.array([b for _, b in rows], float)
    dur = (en - st) * 1e-6  # ms
    gap = np.r_[0.0, (st[1:] - en[:-1]) * 1e-6]
    med = float(np.median(dur))
    slot = np.arange(len(dur)) % per
    by_slot = [{"slot": int(s), "launches": int((slot == s).sum()), "mean_ms": float(dur[slot == s].mean()),
                "max_ms": float(dur[slot == s].max()), "mean_gap_before_ms": float(gap[slot == s].mean())} for s in range(per)]
    out_i = np.nonzero(dur > 1.5 * med)[0]
    out = {"kernel": "k2_vote<true, false, 0>", "launches": int(len(dur)), "launches_per_step": per, "median_ms": med,
           "mean_ms": float(dur.mean()), "std_ms": float(dur.std()), "max_ms": float(dur.max()),
           "by_slot": by_slot,
           "outliers_gt_1.5x_median": [{"launch": int(i), "step": int(i // per), "slot": int(i % per), "ms": float(dur[i]),
                                        "gap_before_ms": float(gap[i])} for i in out_i[:64]],
           "outlier_count": int(len(out_i)),
           "outlier_slots_histogram": {str(s): int((slot[out_i] == s).sum()) for s in range(per)},
           "outlier_steps": sorted({int(i // per) for i in out_i})[:64]}
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(txt)
    else:
        print(txt)


if __name__ == "__main__":
    main()
