#!/usr/bin/env python3
"""Condense gpurun_out/final<ROUND>/ (written by `collect.sh <ROUND>`) into the committed files profiles/round<ROUND>_*.
usage: python profiles/install.py <ROUND>      (round 6 on; rounds 1 - 5 keep their own install_round<N>.py)"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = int(sys.argv[1]) if len(sys.argv) > 1 else 6
F = os.path.join(ROOT, "gpurun_out", "final%d" % ROUND) + "/"
P = os.path.join(ROOT, "profiles") + "/round%d_" % ROUND
RN = "round%d_" % ROUND
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
            "lds_insts_per_frame": val(f, kernel, "SQ_INSTS_LDS") / frames,
            "wait_inst_any_over_wave_cycles": val(f, kernel, "SQ_WAIT_INST_ANY") / val(f, kernel, "SQ_WAVE_CYCLES"),
            # rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs of the MI355X: effective clock = cycles / 8 / duration
            "effective_clock_GHz": val(f, kernel, "effective_clock_GHz") / 8.0,
            "mean_duration_ms_in_the_counter_pass": val(f, kernel, "mean_duration_ns") * 1e-6}


def one_launch(f, kernel, label):
    """a single-frame kernel of the tracked step: per launch"""
    w = val(f, kernel, "SQ_WAVES")
    return {"kernel": label, "waves": w, "valu_insts": val(f, kernel, "SQ_INSTS_VALU"), "salu_insts": val(f, kernel, "SQ_INSTS_SALU"),
            "lds_insts": val(f, kernel, "SQ_INSTS_LDS"), "wave_cycles": val(f, kernel, "SQ_WAVE_CYCLES"),
            "wait_inst_any": val(f, kernel, "SQ_WAIT_INST_ANY"), "active_inst_valu": val(f, kernel, "SQ_ACTIVE_INST_VALU"),
            "duration_us_in_the_counter_pass": val(f, kernel, "mean_duration_ns") * 1e-3,
            "effective_clock_GHz": val(f, kernel, "effective_clock_GHz") / 8.0}


def fused_shape(log, cfg):
    """what one fused launch of a counter pass scanned, from that process's own bench line"""
    b = bench_line_of(log)
    r, c = DIMS[cfg]
    return int(b["kernel_ms"]["frames_per_launch"]), b["roofline"]["bytes_per_launch"] / float(r * c), b["config"]["schedule"], \
        b["roofline"]["bytes_per_launch"]


def frames_per_launch(log, expect=None):
    """frames of ONE kernel launch in a counter pass, from that process's own bench line (a 32768-frame step is two
    16384-frame sub-batches, a 16384-frame one two of 8192 unless --pipeline 1: dividing per-dispatch means by the
    step's frames under-counts per-frame figures by 2, which the round's first five collections did for the cluttered
    legs)"""
    n = int(bench_line_of(log)["kernel_ms"]["frames_per_launch"])
    assert expect is None or n == expect, (log, n, expect)
    return n


def stats_avg(path, kernel_sub):
    for r in csv.DictReader(open(path)):
        if kernel_sub in r["Name"]:
            return float(r["AverageNs"]) * 1e-6, int(r["Calls"])
    raise KeyError((path, kernel_sub))


def main():
    for n in ("bench", "bench_headline_only", "bench_arith1", "bench_arith2"):
        if os.path.exists(F + n + ".json") and os.path.getsize(F + n + ".json") > 2:
            json.dump(last_json(F + n + ".json"), open(P + n + ".json", "w"), indent=1)
    for a, b in (("stats", "bench_kernel_stats"), ("stats_vote", "bench_vote_only_kernel_stats"),
                 ("stats_c3", "bench_C3_kernel_stats"), ("stats_salt", "bench_salt_kernel_stats"),
                 ("stats_streams1", "bench_streams1_kernel_stats")):
        src = F + a + "/s_kernel_stats.csv"
        if os.path.exists(src):
            keep_mpe(src, P + b + ".csv")
    for a, b in (("pmc_fetch", "timed_fetch_size"), ("pmc_write", "timed_write_size"), ("pmc_sq", "timed_sq"),
                 ("pmc1_fetch", "sequential_fetch_size"), ("pmc1_write", "sequential_write_size"),
                 ("pmc1_sq", "sequential_sq"), ("pmc3_sq", "C3_sq"), ("pmc3t2_sq", "C3_tol2_sq"), ("pmcd4_sq", "C2_d4_sq"),
                 ("pmcd16_sq", "C2_d16_sq"), ("pmcsalt_sq", "C2_salt_sq"), ("pmcC4_fetch", "C4_fetch_size"),
                 ("pmcC4_write", "C4_write_size"), ("pmcC1_fetch", "C1_fetch_size"), ("pmcC1_write", "C1_write_size"),
                 ("pmc_track", "tracked_frame_sq")):
        if os.path.exists(F + a + "_summary.csv"):
            shutil.copy(F + a + "_summary.csv", P + "pmc_" + b + ".csv")
    if os.path.exists(F + "period_summary.json"):
        shutil.copy(F + "period_summary.json", P + "blob_window_timeline.json")
    if os.path.exists(F + "vote_trace_summary.json"):
        shutil.copy(F + "vote_trace_summary.json", P + "vote_launch_outliers.json")
    if os.path.exists(F + "pytest_gpu.log"):
        shutil.copy(F + "pytest_gpu.log", P + "pytest_gpu.txt")
    fpl, rider_frames, sched, rider_bytes = fused_shape(F + "pmc_fetch.log", "C2")
    out = {
        "source_fingerprint": open(F + "source_fingerprint.txt").read().strip(),
        "k2_vote_scan": hbm("k2_vote<true", "k2_vote<true> (voting kernel of a %d-frame sub-batch carrying %.0f frames' worth "
                            "of the image scan of the next one; schedule in the counter pass: %s, side streams taken as "
                            "concurrent without the probe: the launch shapes of the timed run)" % (fpl, rider_frames, sched),
                            F + "pmc_fetch_summary.csv", F + "pmc_write_summary.csv", rider_frames, RN + "pmc_timed_*.csv",
                            DIMS["C2"]),
        "k1a_scan": hbm("k1a_scan", "k1a_scan", F + "pmc1_fetch_summary.csv", F + "pmc1_write_summary.csv", 16384,
                        RN + "pmc_sequential_*.csv", DIMS["C2"]),
        "k2_vote_valu": {
            # (--pipeline 1, nothing to scan: since the round's last kernel change the scan-carrying variant with an empty
            #  rider; before it the plain <= 5-marker kernel k2_vote<false, false, 1>)
            "C2": valu(F + "pmc1_sq_summary.csv", "k2_vote<true", "k2_vote<true, false, 0> with an empty rider (no pixels to scan)",
                       frames_per_launch(F + "pmc1_sq.log", 16384), RN + "pmc_sequential_sq.csv"),
            "fused_C2": dict(valu(F + "pmc_sq_summary.csv", "k2_vote<true", "k2_vote<true>", fpl, RN + "pmc_timed_sq.csv"),
                             frames_scanned_per_launch=rider_frames, schedule_in_the_counter_pass=sched),
        },
        "k1b_blobs": valu(F + "pmc1_sq_summary.csv", "k1b_blobs<mpe::K1bSmall>", "k1b_blobs<K1bSmall>", 16384,
                          RN + "pmc_sequential_sq.csv"),
        "k2_vote_fixup": valu(F + "pmc1_sq_summary.csv", "k2_vote_fixup", "k2_vote_fixup", 16384, RN + "pmc_sequential_sq.csv"),
        "by_config": {},
    }
    for key, f, kern, label, frames in (
            ("C3", "pmc3_sq", "k2_vote<false", "k2_vote<false, false, 3> (table slices in LDS, deferred exact evaluation)", 16384),
            ("C3_tol2", "pmc3t2_sq", "k2_vote<false", "k2_vote<false, false, 3> at back_projection_pixel_tolerance 2", 16384),
            ("C2_d4", "pmcd4_sq", "k2_vote<true", "k2_vote<true> on frames with 4 distractor spots (9 detections)", None),
            ("C2_d16", "pmcd16_sq", "k2_vote<true", "k2_vote<true> on frames with 16 distractor spots (21 detections)", None)):
        if os.path.exists(F + f + "_summary.csv"):
            try:
                out["k2_vote_valu"][key] = valu(F + f + "_summary.csv", kern, label, frames_per_launch(F + f + ".log", frames),
                                                RN + "pmc_%s_sq.csv" % key)
            except KeyError as e:
                print("no counters for", key, e)
    if os.path.exists(F + "pmcsalt_sq_summary.csv"):
        try:
            out["k1b_general_salt"] = valu(F + "pmcsalt_sq_summary.csv", "k1b_general", "k1b_general on frames with 0.05 % salt "
                                           "noise (every frame reaches this tier)", frames_per_launch(F + "pmcsalt_sq.log"),
                                           RN + "pmc_C2_salt_sq.csv")
        except KeyError as e:
            print("no counters for k1b_general", e)
    if os.path.exists(F + "pmc_track_summary.csv"):
        tr = {}
        # (round 6: a tracked frame is ONE kernel, k_track_frame; the four names of rounds 3 - 5 only appear in the
        #  repeat path of a frame the small blob tier cannot hold)
        for kern, label in (("k_track_frame", "k_track_frame (scan + blobs + validate + refine)"), ("k1a_scan", "k1a_scan"),
                            ("k1b_blobs<mpe::K1bSmall>", "k1b_blobs<K1bSmall>"), ("k3a_validate", "k3a_validate"),
                            ("k3b_refine_group", "k3b_refine_group")):
            try:
                tr[label] = one_launch(F + "pmc_track_summary.csv", kern, label)
            except KeyError as e:
                print("tracked frame: no counters for", kern, e)
        out["tracked_frame"] = tr
    # the rocprofv3 clock of the dominant kernel, from the pass that traces ONLY that kernel
    ms, calls = stats_avg(F + "stats_vote/s_kernel_stats.csv", "k2_vote<true")
    traced = bench_line_of(F + "stats_vote.log")
    out["k2_vote_scan"].update({"rocprof_avg_launch_ms": ms, "rocprof_launches": calls,
                                "rocprof_bytes_per_launch": traced["roofline"]["bytes_per_launch"],
                                "rocprof_process_ms_per_step": traced["ms_per_step"],
                                "rocprof_process_hip_event_launch_ms": traced["roofline"]["avg_launch_ms"],
                                "rocprof_from": RN + "bench_vote_only_kernel_stats.csv (rocprofv3 --kernel-trace "
                                                "--kernel-include-regex '.*k2_vote<true.*' --stats of the bench command)"})
    for cfg in ("C1", "C4"):
        if os.path.exists(F + "pmc%s_fetch_summary.csv" % cfg):
            fpl_c, rf, sc, _ = fused_shape(F + "pmc%s_fetch.log" % cfg, cfg)
            out["by_config"][cfg] = {"k2_vote_scan": hbm("k2_vote<true", "k2_vote<true> at %s (%d frames voted, %.0f frames' worth "
                                                         "scanned per launch, schedule %s)" % (cfg, fpl_c, rf, sc),
                                                         F + "pmc%s_fetch_summary.csv" % cfg, F + "pmc%s_write_summary.csv" % cfg,
                                                         rf, RN + "pmc_%s_*.csv" % cfg, DIMS[cfg])}
    json.dump(out, open(P + "pmc.json", "w"), indent=1)
    for n in ("soak_votes_arith_C2", "soak_votes_arith_C3", "soak_general_tier"):
        if os.path.exists(F + n + ".json"):
            shutil.copy(F + n + ".json", P + "parity_%s.json" % n)
    for n in ("soak_votes_C2", "soak_34_C2", "soak_34_C3", "soak_34_d16", "soak_34_d4", "soak_parity_C2", "soak_parity_C3", "soak_parity_C4", "soak_parity_C1", "soak_parity_clutter_salt", "soak_parity_clutter_d4", "soak_parity_clutter_d16", "soak_parity_clutter_salt_C4", "soak_parity_clutter_d4_C3", "soak_tracking", "soak_tracking_salt"):
        if os.path.exists(F + n + ".log"):
            try:
                json.dump(last_json(F + n + ".log"), open(P + "parity_%s.json" % n, "w"), indent=1)
            except Exception as e:
                print("no JSON line in", n, e)
    print("installed")


if __name__ == "__main__":
    main()
