#!/usr/bin/env python3
"""Condense gpurun_out/final2/ (written by collect_round2.sh) into the committed files profiles/round2_*."""
import csv
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, "gpurun_out", "final2") + "/"
P = os.path.join(ROOT, "profiles") + "/round2_"


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def keep_mpe(src, dst):
    rows = list(csv.reader(open(src)))
    with open(dst, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(rows[0])
        for r in rows[1:]:
            if "mpe::" in r[0]:
                w.writerow(r)


def val(f, kernel_suffix, counter):
    for r in csv.DictReader(open(f)):
        if r["kernel"].endswith(kernel_suffix) and r["counter"] == counter:
            return float(r["mean_per_dispatch"])
    raise KeyError((f, kernel_suffix, counter))


def hbm(kernel, label, fetch_csv, write_csv, frames, where):
    fe, wr = val(fetch_csv, kernel, "FETCH_SIZE"), val(write_csv, kernel, "WRITE_SIZE")
    return {"kernel": label, "rows": 480, "cols": 752, "frames_per_launch": frames, "FETCH_SIZE_KB": fe,
            "WRITE_SIZE_KB": wr, "fetch_bytes": fe * 1024 * 2, "write_bytes": wr * 1024,
            "hbm_bytes_per_frame": (fe * 1024 * 2 + wr * 1024) / frames, "algorithmic_bytes_per_frame": 480 * 752,
            "from": where,
            "note": "FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950; separate --pmc passes"}


def main():
    json.dump(last_json(F + "bench.json"), open(P + "bench.json", "w"), indent=1)
    for c in ("C1", "C3", "C4", "mode3"):
        json.dump(last_json(F + "bench_%s.json" % c), open(P + "bench_%s.json" % c, "w"), indent=1)
    for n in ("streams1", "streams8", "lockstep8", "lockstep64", "lockstep128", "lockstep256", "lockstep64g2", "lockstep256g4"):
        json.dump(last_json(F + n + ".json"), open(P + "bench_%s.json" % n, "w"), indent=1)
    keep_mpe(F + "stats/s_kernel_stats.csv", P + "bench_kernel_stats.csv")
    keep_mpe(F + "stats_seq/s_kernel_stats.csv", P + "bench_sequential_kernel_stats.csv")
    keep_mpe(F + "stats_c3/s_kernel_stats.csv", P + "bench_C3_kernel_stats.csv")
    keep_mpe(F + "stats_lockstep/s_kernel_stats.csv", P + "bench_lockstep64_kernel_stats.csv")
    for a, b in (("pmc_fetch", "fused_fetch_size"), ("pmc_write", "fused_write_size"), ("pmc_sq", "fused_sq"),
                 ("pmc1_fetch", "sequential_fetch_size"), ("pmc1_write", "sequential_write_size"),
                 ("pmc1_sq", "sequential_sq"), ("pmc3_sq", "C3_sq")):
        shutil.copy(F + a + "_summary.csv", P + "pmc_" + b + ".csv")
    out = {
        # default schedule 6: 20 % of the next sub-batch is scanned by a side k1a_scan, the rider of this launch scans
        # the other 80 % = 4 731 174 912 B = 13 107.2 frames' worth of pixels (what bench.py reports as bytes_per_launch)
        "k2_vote_scan": hbm("k2_vote<true>", "k2_vote<true> (voting kernel carrying 80 % of the image scan of the next sub-batch)",
                            F + "pmc_fetch_summary.csv", F + "pmc_write_summary.csv", 16384 * 0.8, "round2_pmc_fused_*.csv"),
        "k1a_scan": hbm("k1a_scan", "k1a_scan", F + "pmc1_fetch_summary.csv", F + "pmc1_write_summary.csv", 16384,
                        "round2_pmc_sequential_*.csv"),
        "k2_vote_valu": {
            "C2": {"kernel": "k2_vote<false>", "frames_per_launch": 16384, "from": "round2_pmc_sequential_sq.csv",
                   "valu_insts_per_frame": val(F + "pmc1_sq_summary.csv", "k2_vote<false>", "SQ_INSTS_VALU") / 16384,
                   "salu_insts_per_frame": val(F + "pmc1_sq_summary.csv", "k2_vote<false>", "SQ_INSTS_SALU") / 16384},
            "C3": {"kernel": "k2_vote<false>", "frames_per_launch": 16384, "from": "round2_pmc_C3_sq.csv",
                   "valu_insts_per_frame": val(F + "pmc3_sq_summary.csv", "k2_vote<false>", "SQ_INSTS_VALU") / 16384,
                   "salu_insts_per_frame": val(F + "pmc3_sq_summary.csv", "k2_vote<false>", "SQ_INSTS_SALU") / 16384},
            "fused_C2": {"kernel": "k2_vote<true>", "frames_per_launch": 16384, "from": "round2_pmc_fused_sq.csv",
                         "valu_insts_per_frame": val(F + "pmc_sq_summary.csv", "k2_vote<true>", "SQ_INSTS_VALU") / 16384},
        },
    }
    json.dump(out, open(P + "pmc.json", "w"), indent=1)
    for n in ("soak_fast", "soak_strict", "soak_fast_c3", "soak_votes"):
        json.dump(last_json(F + n + ".json"), open(P + "parity_%s.json" % n, "w"), indent=1)


if __name__ == "__main__":
    main()
