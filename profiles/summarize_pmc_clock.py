#!/usr/bin/env python3
"""Per-kernel means of a rocprofv3 `--kernel-trace --pmc ...` pass, plus the EFFECTIVE CLOCK of every kernel =
GRBM_GUI_ACTIVE / (End_Timestamp - Start_Timestamp) when that counter was collected (MI355X_MICROARCH.md, "DVFS
give-back").   usage: summarize_pmc_clock.py <dir with *_counter_collection.csv and *_kernel_trace.csv> [out.csv]"""
import collections
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not cc:
        sys.exit("no counter_collection.csv under " + d)
    dur = {}
    for f in kt:
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (r["Kernel_Name"].split("(")[0], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    per = collections.defaultdict(float)
    names = {}
    for f in cc:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            if "mpe::" not in name:
                continue
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = name
    agg = collections.defaultdict(lambda: [0, 0.0])
    clk = collections.defaultdict(list)
    for (did, ctr), v in per.items():
        a = agg[(names[did], ctr)]
        a[0] += 1
        a[1] += v
        if ctr == "GRBM_GUI_ACTIVE" and did in dur and dur[did][1] > 0:
            clk[names[did]].append((v, dur[did][1]))
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(out)
    w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
    for (name, ctr), (n, tot) in sorted(agg.items()):
        w.writerow([name, ctr, n, "%.1f" % (tot / n)])
    for name, lst in sorted(clk.items()):
        cyc = sum(c for c, _ in lst)
        ns = sum(t for _, t in lst)
        w.writerow([name, "effective_clock_GHz (GRBM_GUI_ACTIVE / duration)", len(lst), "%.4f" % (cyc / ns)])
        w.writerow([name, "mean_duration_ns (counter pass, kernels serialised)", len(lst), "%.1f" % (ns / len(lst))])


if __name__ == "__main__":
    main()
