#!/usr/bin/env python3
"""Condense gpurun_out/final4/ (written by collect_round4.sh) into the committed files profiles/round4_*."""
import csv
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, "gpurun_out", "final4") + "/"
P = os.path.join(ROOT, "profiles") + "/round4_"
DIMS = {"C1": (480, 752), "C2": (480, 752), "C3": (480, 752), "C4": (1200, 1920)}


def last_json(path):
    lines = [ln for ln in open(path).read().strip().splitlines() if ln.startswith("{")]
    return json.loads(lines[-1])


def keep_mpe(src, dst):
    rows = list(csv.reader(open(src)))
    with open(dst, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(rows[0])
        for r in rows[1:]:
            if "mpe::" in r[0]:
                w.writerow(r)


def val(f, kernel_prefix, counter):
    """mean per dispatch of `counter` for the kernel whose name starts with `kernel_prefix` (template arguments vary)"""
    for r in csv.DictReader(open(f)):
        name = r["kernel"].replace("void ", "")
        if name.startswith("mpe::" + kernel_prefix) and r["counter"].startswith(counter):
            return float(r["mean_per_dispatch"])
    raise KeyError((f, kernel_prefix, counter))


def bench_line_of(log):
    out = None
    for ln in open(log):
        if ln.startswith('{"metric"'):
            out = json.loads(ln)
    return out


def hbm(kernel, label, fetch_csv, write_csv, frames, where, dims):
    fe, wr = val(fetch_csv, kernel, "FETCH_SIZE"), val(write_csv, kernel, "WRITE_SIZE")
    return {"kernel": label, "rows": dims[0], "cols": dims[1], "frames_per_launch": frames, "FETCH_SIZE_KB": fe,
            "WRITE_SIZE_KB": wr, "fetch_bytes": fe * 1024 * 2, "write_bytes": wr * 1024,
            "hbm_bytes_per_frame": (fe * 1024 * 2 + wr * 1024) / frames, "algorithmic_bytes_per_frame": dims[0] * dims[1],
            "from": where,
            "note": "FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950; separate --pmc passes"}


def valu(f, kernel, label, frames, where):
    return {"kernel": label, "frames_per_launch": frames, "from": where,
            "valu_insts_per_frame": val(f, kernel, "SQ_INSTS_VALU") / frames,
            "salu_insts_per_frame": val(f, kernel, "SQ_INSTS_SALU") / frames,
            "wait_inst_any_over_wave_cycles": val(f, kernel, "SQ_WAIT_INST_ANY") / val(f, kernel, "SQ_WAVE_CYCLES"),
            # rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs of the MI355X: effective clock = cycles / 8 / duration
            "effective_clock_GHz": val(f, kernel, "effective_clock_GHz") / 8.0}


def fused_shape(log, cfg):
    """what one fused launch of a counter pass scanned, from that process's own bench line"""
    b = bench_line_of(log)
    r, c = DIMS[cfg]
    return int(b["kernel_ms"]["frames_per_launch"]), b["roofline"]["bytes_per_launch"] / float(r * c), b["config"]["schedule"], \
        b["roofline"]["bytes_per_launch"]


def stats_avg(path, kernel_sub):
    for r in csv.DictReader(open(path)):
        if kernel_sub in r["Name"]:
            return float(r["AverageNs"]) * 1e-6, int(r["Calls"])
    raise KeyError((path, kernel_sub))


def main():
    for n in ("bench", "bench_r3", "bench_nostream", "bench_arith1", "bench_arith2", "bench_C1", "bench_C3", "bench_C3_tol2", "bench_C4"):
        if os.path.exists(F + n + ".json") and os.path.getsize(F + n + ".json") > 2:
            json.dump(last_json(F + n + ".json"), open(P + n + ".json", "w"), indent=1)
    for n in ("streams1", "streams8", "lockstep8", "lockstep64", "lockstep256", "lockstep256g4t4", "lockstep512g8t8"):
        if os.path.exists(F + n + ".json") and os.path.getsize(F + n + ".json") > 2:
            json.dump(last_json(F + n + ".json"), open(P + "bench_%s.json" % n, "w"), indent=1)
    for a, b in (("stats", "bench_kernel_stats"), ("stats_vote", "bench_vote_only_kernel_stats"),
                 ("stats_seq", "bench_sequential_kernel_stats"), ("stats_c3", "bench_C3_kernel_stats"),
                 ("stats_streams1", "bench_streams1_kernel_stats"), ("stats_lockstep", "bench_lockstep64_kernel_stats")):
        src = F + a + "/s_kernel_stats.csv"
        if os.path.exists(src):
            keep_mpe(src, P + b + ".csv")
    for a, b in (("pmc_fetch", "timed_fetch_size"), ("pmc_write", "timed_write_size"), ("pmc_sq", "timed_sq"),
                 ("pmc1_fetch", "sequential_fetch_size"), ("pmc1_write", "sequential_write_size"),
                 ("pmc1_sq", "sequential_sq"), ("pmc3_sq", "C3_sq"), ("pmcC4_fetch", "C4_fetch_size"),
                 ("pmcC4_write", "C4_write_size"), ("pmcC1_fetch", "C1_fetch_size"), ("pmcC1_write", "C1_write_size")):
        if os.path.exists(F + a + "_summary.csv"):
            shutil.copy(F + a + "_summary.csv", P + "pmc_" + b + ".csv")
    shutil.copy(F + "pytest_gpu.log", P + "pytest_gpu.txt")
    fpl, rider_frames, sched, rider_bytes = fused_shape(F + "pmc_fetch.log", "C2")
    out = {
        "source_fingerprint": open(F + "source_fingerprint.txt").read().strip(),
        "k2_vote_scan": hbm("k2_vote<true", "k2_vote<true> (voting kernel of a %d-frame sub-batch carrying %.0f frames' worth "
                            "of the image scan of the next one; schedule in the counter pass: %s, side streams taken as "
                            "concurrent without the probe: the launch shapes of the timed run)" % (fpl, rider_frames, sched),
                            F + "pmc_fetch_summary.csv", F + "pmc_write_summary.csv", rider_frames, "round4_pmc_timed_*.csv",
                            DIMS["C2"]),
        "k1a_scan": hbm("k1a_scan", "k1a_scan", F + "pmc1_fetch_summary.csv", F + "pmc1_write_summary.csv", 16384,
                        "round4_pmc_sequential_*.csv", DIMS["C2"]),
        "k2_vote_valu": {
            "C2": valu(F + "pmc1_sq_summary.csv", "k2_vote<false", "k2_vote<false, false, 1>", 16384, "round4_pmc_sequential_sq.csv"),
            "C3": valu(F + "pmc3_sq_summary.csv", "k2_vote<false", "k2_vote<false, false, 3> (table slices in LDS, deferred "
                       "exact evaluation)", 16384, "round4_pmc_C3_sq.csv"),
            "fused_C2": dict(valu(F + "pmc_sq_summary.csv", "k2_vote<true", "k2_vote<true>", fpl, "round4_pmc_timed_sq.csv"),
                             frames_scanned_per_launch=rider_frames, schedule_in_the_counter_pass=sched),
        },
        "k1b_blobs": valu(F + "pmc1_sq_summary.csv", "k1b_blobs<mpe::K1bSmall>", "k1b_blobs<K1bSmall>", 16384,
                          "round4_pmc_sequential_sq.csv"),
        "k2_vote_fixup": valu(F + "pmc1_sq_summary.csv", "k2_vote_fixup", "k2_vote_fixup", 16384, "round4_pmc_sequential_sq.csv"),
        "k1a_scan_valu": valu(F + "pmc1_sq_summary.csv", "k1a_scan", "k1a_scan", 16384, "round4_pmc_sequential_sq.csv"),
        "by_config": {},
    }
    # the rocprofv3 clock of the dominant kernel, from the pass that traces ONLY that kernel
    ms, calls = stats_avg(F + "stats_vote/s_kernel_stats.csv", "k2_vote<true")
    traced = bench_line_of(F + "stats_vote.log")
    out["k2_vote_scan"].update({"rocprof_avg_launch_ms": ms, "rocprof_launches": calls,
                                "rocprof_bytes_per_launch": traced["roofline"]["bytes_per_launch"],
                                "rocprof_process_ms_per_step": traced["ms_per_step"],
                                "rocprof_process_hip_event_launch_ms": traced["roofline"]["avg_launch_ms"],
                                "rocprof_from": "round4_bench_vote_only_kernel_stats.csv (rocprofv3 --kernel-trace "
                                                "--kernel-include-regex 'k2_vote<true' --stats of the bench command)"})
    for cfg in ("C1", "C4"):
        if os.path.exists(F + "pmc%s_fetch_summary.csv" % cfg):
            fpl_c, rf, sc, _ = fused_shape(F + "pmc%s_fetch.log" % cfg, cfg)
            out["by_config"][cfg] = {"k2_vote_scan": hbm("k2_vote<true", "k2_vote<true> at %s (%d frames voted, %.0f frames' worth "
                                                         "scanned per launch, schedule %s)" % (cfg, fpl_c, rf, sc),
                                                         F + "pmc%s_fetch_summary.csv" % cfg, F + "pmc%s_write_summary.csv" % cfg,
                                                         rf, "round4_pmc_%s_*.csv" % cfg, DIMS[cfg])}
    json.dump(out, open(P + "pmc.json", "w"), indent=1)
    for n in ("soak_votes_C2", "soak_votes_C3", "soak_parity_C2", "soak_parity_C3", "soak_parity_C3_tol2", "soak_parity_C4",
              "soak_parity_C1", "soak_tracking"):
        if os.path.exists(F + n + ".log"):
            try:
                json.dump(last_json(F + n + ".log"), open(P + "parity_%s.json" % n, "w"), indent=1)
            except Exception as e:
                print("no JSON line in", n, e)
    print("installed")


if __name__ == "__main__":
    main()
