#!/usr/bin/env python3
"""Condense a rocprofv3 *_counter_collection.csv into per-kernel means (one row per kernel x counter).

usage: summarize_pmc.py <counter_collection.csv> [<out.csv>]   — only mpe:: kernels are kept."""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    agg = collections.defaultdict(lambda: [0, 0.0])
    # rocprofv3 emits one row per (dispatch, counter); sum first over rows of the same dispatch+counter
    per = collections.defaultdict(float)
    for r in rows:
        name = r["Kernel_Name"].split("(")[0]
        if not name.startswith("mpe::") and "mpe::" not in name:
            continue
        per[(r["Dispatch_Id"], name, r["Counter_Name"])] += float(r["Counter_Value"])
    for (_, name, ctr), v in per.items():
        agg[(name, ctr)][0] += 1
        agg[(name, ctr)][1] += v
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(out)
    w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
    for (name, ctr), (n, tot) in sorted(agg.items()):
        w.writerow([name, ctr, n, "%.1f" % (tot / n)])


if __name__ == "__main__":
    main()
