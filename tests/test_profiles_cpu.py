"""The committed counter evidence is internally consistent: every per-frame figure in profiles/round5_pmc.json divides a
per-DISPATCH mean by the frames of ONE launch.  (Until the round's last collection the cluttered legs were divided by
the frames of a step — two launches — and every figure derived from them was half the truth; SQ_WAVES of the same
dispatches would have said so.)"""
import csv
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
# the latest collection (profiles/collect.sh <round> -> profiles/install.py <round>)
RN = "round6_" if os.path.exists(os.path.join(PROF, "round6_pmc.json")) else "round5_"


def _mean(csv_name, kernel_sub, counter):
    for r in csv.DictReader(open(os.path.join(PROF, csv_name))):
        if kernel_sub in r["kernel"] and r["counter"] == counter:
            return float(r["mean_per_dispatch"]), int(r["dispatches"])
    raise KeyError((csv_name, kernel_sub, counter))


@pytest.fixture(scope="module")
def pmc():
    return json.load(open(os.path.join(PROF, "%spmc.json" % RN)))


@pytest.mark.parametrize("key,tol", [("C2", 0.0), ("fused_C2", 0.0), ("C2_d4", 0.0), ("C2_d16", 0.03)])
def test_frames_per_launch_agree_with_the_waves_of_the_dispatches(pmc, key, tol):
    """k2_vote<true> at 5 markers is launched with one 128-thread block per frame: SQ_WAVES per dispatch = 2 x frames.
    (d16: the first launch of the pass runs without the tier hint and a few frames are voted by table slices.)"""
    v = pmc["k2_vote_valu"][key]
    waves, _ = _mean(v["from"], "k2_vote<true", "SQ_WAVES")
    # (round 6: the launcher sizes the block from the detection-count hint — 128 threads for 5 detections, 192 for the
    #  960 items of a 16-triple chunk at 9 / 21 detections: 2 or 3 waves per frame)
    per_frame = 3.0 if (RN == "round6_" and key in ("C2_d4", "C2_d16")) else 2.0
    assert abs(waves / per_frame - v["frames_per_launch"]) <= tol * v["frames_per_launch"], (key, waves, v["frames_per_launch"])
    valu, _ = _mean(v["from"], "k2_vote<true", "SQ_INSTS_VALU")
    assert abs(valu / v["frames_per_launch"] - v["valu_insts_per_frame"]) < 1e-6 * v["valu_insts_per_frame"]


def test_per_solve_instruction_counts_against_the_detections(pmc):
    """P3P solves per frame: C(n_d,3) x 60 at 5 markers.  The nearest-detection search grows with n_d, so a solve costs
    MORE instructions with more detections, never fewer (the halved counts had d4 at 834 against the clean 1 500)."""
    per_solve = {}
    for key, n_d in (("C2_d4", 9), ("C2_d16", 21)):
        solves = n_d * (n_d - 1) * (n_d - 2) // 6 * 60
        per_solve[key] = pmc["k2_vote_valu"][key]["valu_insts_per_frame"] * 64.0 / solves
    clean = pmc["k2_vote_valu"]["C2"]["valu_insts_per_frame"] * 64.0 / 600.0  # (every frame with 5 detections: upper bound)
    if RN == "round6_":
        # the candidate-mask grid (k2_vote<true, false, 0, true>): a solve no longer pays for every unused detection —
        # 1 400 VALU at 9 and at 21 detections (round 5: 1 668 / 2 496), below the clean frame's 1 500 (whose waves idle
        # through more of their lanes: 600 hypotheses on 128 threads)
        for k in per_solve:
            assert 0.85 * clean < per_solve[k] < 1.1 * clean, (clean, per_solve)
        return
    assert 0.9 * clean < per_solve["C2_d4"] < per_solve["C2_d16"] < 2 * clean, (clean, per_solve)


def test_issue_fractions_of_the_counter_passes_stay_under_the_roof(pmc):
    """VALU wave-instructions per second of each counter pass against 1024 SIMDs x its own clock / 4."""
    for key, v in pmc["k2_vote_valu"].items():
        rate = v["valu_insts_per_frame"] * v["frames_per_launch"] / (v["mean_duration_ms_in_the_counter_pass"] * 1e-3)
        roof = 1024 * v["effective_clock_GHz"] * 1e9 / 4.0
        assert 0.2 < rate / roof < 1.0, (key, rate / roof)
    g = pmc["k1b_general_salt"]
    rate = g["valu_insts_per_frame"] * g["frames_per_launch"] / (g["mean_duration_ms_in_the_counter_pass"] * 1e-3)
    assert rate / (1024 * g["effective_clock_GHz"] * 1e9 / 4.0) < 0.5  # (latency bound: far from the issue roof)


def test_scan_traffic_is_the_algorithmic_bytes_and_a_little(pmc):
    """HBM bytes per scanned frame from FETCH_SIZE x 2 + WRITE_SIZE: never below the frame itself, within 2 % of it."""
    for v in [pmc["k2_vote_scan"], pmc["k1a_scan"]] + [c["k2_vote_scan"] for c in pmc["by_config"].values()]:
        ratio = v["hbm_bytes_per_frame"] / v["algorithmic_bytes_per_frame"]
        assert 1.0 <= ratio < 1.02, (v["kernel"], ratio)


def test_the_bench_line_of_the_collection_used_these_counters(pmc):
    b = json.load(open(os.path.join(PROF, RN + "bench.json")))
    assert b["roofline"]["counters"]["from_a_build_of_these_sources"] is True
    import rpg_monocular_pose_estimator_amd as mpe
    assert pmc["source_fingerprint"] == mpe.source_fingerprint(), "profiles/%spmc.json is not of this tree's kernels" % RN
    assert b["legs_failed"] == []
    for leg in ("d4", "d16"):
        r = b["clutter"][leg]["roofline"]
        assert r["bound"] == "fp64_valu" and 0.5 < r["frac"] < 1.0, (leg, r)
