"""The DEVICE voting source — what one lane of k2_vote runs for a (detection triple, marker permutation) item, plus the
marker-permutation table and the per-triple part of computePoses — compiled for the HOST and driven over all items of
a frame: the vote histogram must equal the oracle's `initialise()` loop (pose_estimator.cpp:544-702) cell by cell.
Only the hardware reciprocal / rsqrt seeds of the fast arithmetic are replaced (tests/host/stub/hip/hip_runtime.h)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from rpg_monocular_pose_estimator_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rpg_monocular_pose_estimator_amd", "csrc")
MAX_DET, MAX_MARK = 32, 16


def _cut(text, begin, end):
    i = text.index(begin)
    return text[i:text.index(end, i)]


def _build(d, tiny_queues=False):
    import rpg_monocular_pose_estimator_amd as mpe
    hip = mpe.device_source()
    internal = open(os.path.join(CSRC, "mpe_internal.h")).read()
    inc = _cut(internal, "struct SolveParams {", "#define MPE_HIST_STRIDE")
    inc += _cut(hip, "// lexicographic unranking of the idx-th 3-combination", "#define K2_THREADS")
    inc += _cut(hip, "#define K2_TRI_CHUNK 64", "__global__ void k2_prep_markers(")
    inc += _cut(hip, "struct NoRider {", "// Voting kernel.  Work item =")   # incl. the deferred-vote queue
    inc += _cut(hip, "// One hypothesis in the STRICT arithmetic", "// Strict voting kernel (option")
    if tiny_queues:  # capacities at which the "no room: vote on the spot" paths run all the time
        assert "#define K2_VQ_CAP 28" in inc
        inc = inc.replace("#define K2_VQ_CAP 28", "#define K2_VQ_CAP 9")
    with open(os.path.join(d, "vote_extract.inc"), "w") as fh:
        fh.write(inc)
    so = os.path.join(d, "libvote_host.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", *os.environ.get("MPE_HOST_CXXFLAGS", "").split(),
                           *os.environ.get("MPE_HOST_CXXFLAGS", "").split(), "-I", str(d),
                           "-I", os.path.join(ROOT, "tests", "host", "stub"), "-I", CSRC,
                           "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "host", "vote_host.cpp"), "-o", so])
    lib = C.CDLL(so)
    lib.host_vote.restype = C.c_int
    return lib


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    return _build(tmp_path_factory.mktemp("vote_host"))


@pytest.fixture(scope="module")
def host_tiny(tmp_path_factory):
    return _build(tmp_path_factory.mktemp("vote_host_tiny"), tiny_queues=True)


@pytest.fixture(scope="module")
def orc():
    import oracle
    oracle.build()
    from oracle import binding
    return binding


def _host_hist(lib, det, markers, K, tol, variant=0):
    det = np.ascontiguousarray(det, float)
    markers = np.ascontiguousarray(markers, float)
    k4 = np.array([K[0][0], K[1][1], K[0][2], K[1][2]], float)
    hist = np.zeros((MAX_DET, MAX_MARK), np.uint32)
    rc = lib.host_vote(det.ctypes.data_as(C.c_void_p), len(det), markers.ctypes.data_as(C.c_void_p), len(markers),
                       k4.ctypes.data_as(C.c_void_p), C.c_double(tol), hist.ctypes.data_as(C.c_void_p), variant)
    assert rc == 0
    return hist[:len(det), :len(markers)]


@pytest.mark.parametrize("config,n_frames", [("C1", 40), ("C2", 60), ("C3", 2)])
def test_device_voting_source_on_the_host(host, orc, config, n_frames):
    """Synthetic frames of the BASELINE configs: detections from the oracle's findLeds, votes from the device source."""
    d = synth.make_frames(config, n_frames, seed=2024)
    P = orc.make_params()
    n_votes = 0
    n_done = 0
    for i in range(n_frames):
        und, _ = orc.find_leds(d["frames"][i], P, d["K"], d["D"])
        if len(und) < 4 or len(und) > MAX_DET:
            continue
        ref = orc.vote_histogram(und, d["markers"], d["K"], P.back_projection_pixel_tolerance)
        got = _host_hist(host, und, d["markers"], d["K"], P.back_projection_pixel_tolerance)
        assert np.array_equal(got, ref), (config, i, np.argwhere(got != ref)[:5])
        if len(d["markers"]) <= 5:   # the scan-carrying variant: prefilter mask, deferred exact votes, queue + flush
            got1 = _host_hist(host, und, d["markers"], d["K"], P.back_projection_pixel_tolerance, 1)
            assert np.array_equal(got1, ref), (config, i, "scan variant", np.argwhere(got1 != ref)[:5])
        n_votes += int(ref.sum())
        n_done += 1
    assert n_done >= n_frames // 2 and n_votes > 0, (n_done, n_votes)


def test_device_voting_source_random_detections(host, orc):
    """Detections that do NOT come from the marker set (most hypotheses vote for nothing, some by accident), planar and
    general rigs, tolerances 1 / 3 / 5 px."""
    rng = np.random.default_rng(77)
    cfg = synth.CONFIGS["C2"]
    K, _ = synth.camera_for(cfg["rows"], cfg["cols"])
    for it in range(60):
        n_m = int(rng.integers(4, 7))
        markers = rng.uniform(-0.15, 0.15, (n_m, 3))
        if it % 3 == 0:
            markers[:, 2] = 0.0
        n_d = int(rng.integers(4, 9))
        det = np.column_stack([rng.uniform(250, 500, n_d), rng.uniform(150, 330, n_d)])
        tol = [1.0, 3.0, 5.0][it % 3]
        ref = orc.vote_histogram(det, markers, K, tol)
        got = _host_hist(host, det, markers, K, tol)
        assert np.array_equal(got, ref), (it, np.argwhere(got != ref)[:5])
        if n_m <= 5:
            got1 = _host_hist(host, det, markers, K, tol, 1)
            assert np.array_equal(got1, ref), (it, "scan variant", np.argwhere(got1 != ref)[:5])


def test_device_voting_source_with_full_queues(host_tiny, orc):
    """The same source built with a 5-entry vote queue (flush threshold 1): the queue is worked off after nearly every
    push and lanes that find no room vote on the spot — histograms unchanged."""
    for config, n_frames in (("C2", 12), ("C3", 1)):
        d = synth.make_frames(config, n_frames, seed=31)
        P = orc.make_params()
        for i in range(n_frames):
            und, _ = orc.find_leds(d["frames"][i], P, d["K"], d["D"])
            if len(und) < 4:
                continue
            ref = orc.vote_histogram(und, d["markers"], d["K"], P.back_projection_pixel_tolerance)
            for variant in ((0, 1) if len(d["markers"]) <= 5 else (0,)):
                got = _host_hist(host_tiny, und, d["markers"], d["K"], P.back_projection_pixel_tolerance, variant)
                assert np.array_equal(got, ref), (config, i, variant)


def _stats(lib):
    st = np.zeros(4, np.uint32)
    lib.host_vote_stats(st.ctypes.data_as(C.c_void_p))
    return int(st[0]), int(st[1]), int(st[2]), int(st[3])


@pytest.mark.parametrize("config,n_frames", [("C1", 30), ("C2", 60), ("C3", 2)])
def test_fast_votes_with_strict_fixup_equal_strict_votes(host, orc, config, n_frames):
    """The default voting arithmetic = the fast work item + the strict re-evaluation of the hypotheses it appends to its
    suspect list (k2_sus_push / k2_strict_item, what k2_vote_fixup runs on the device): the histogram must be the one
    the strict functions produce for EVERY hypothesis (variant 20 = k2_vote_strict's loop), cell by cell — and the
    oracle's.  Both kernel variants; the list must have been used (some entries) and never full."""
    d = synth.make_frames(config, n_frames, seed=99)
    P = orc.make_params()
    tol = P.back_projection_pixel_tolerance
    entries = 0
    for i in range(n_frames):
        und, _ = orc.find_leds(d["frames"][i], P, d["K"], d["D"])
        if len(und) < 4 or len(und) > MAX_DET:
            continue
        strict = _host_hist(host, und, d["markers"], d["K"], tol, 20)
        assert np.array_equal(strict, orc.vote_histogram(und, d["markers"], d["K"], tol)), (config, i, "strict vs oracle")
        for variant in ((10, 11) if len(d["markers"]) <= 5 else (10,)):
            got = _host_hist(host, und, d["markers"], d["K"], tol, variant)
            n, whole, lost, marked = _stats(host)
            assert np.array_equal(got, strict), (config, i, variant, np.argwhere(got != strict)[:5])
            assert lost == 0 and marked == 0
            entries += n
    assert entries > 0


def test_fixup_on_the_saved_unstable_frames(host, orc):
    """The frames on which round 3's fast arithmetic and the oracle disagreed (tests/data/unstable_det_r3_*.npy, one
    hypothesis each in the corner of Ferrari's method): with the suspect list the fast path hands exactly those
    hypotheses to the strict functions and ends up with the strict histogram."""
    cfg = synth.CONFIGS["C2"]
    K, _ = synth.camera_for(cfg["rows"], cfg["cols"])
    markers = np.ascontiguousarray(cfg["markers"], float)
    data = os.path.join(ROOT, "tests", "data")
    files = sorted(f for f in os.listdir(data) if f.startswith("unstable_det_r3_") or f.startswith("vote_regression_det"))
    assert files
    for f in files:
        det = np.load(os.path.join(data, f))
        strict = _host_hist(host, det, markers, K, 5.0, 20)
        for variant in (10, 11):
            got = _host_hist(host, det, markers, K, 5.0, variant)
            assert np.array_equal(got, strict), (f, variant, np.argwhere(got != strict)[:5])
            assert _stats(host)[1] >= 1, f   # at least one whole hypothesis went to the strict functions


def test_fixup_lists_full(host, orc):
    """The block's list holds four entries: appends go straight to the launch's list instead — same histogram as the
    strict loop.  That one holds a single entry: entries are LOST (their votes are cast nowhere), which is
    counted and marks the frame MPE_FRAME_VOTE_LIST_FULL — a capacity overrun is never silent."""
    d = synth.make_frames("C2", 12, seed=5)
    P = orc.make_params()
    tol = P.back_projection_pixel_tolerance
    seen_lost = seen_entries = 0
    for i in range(12):
        und, _ = orc.find_leds(d["frames"][i], P, d["K"], d["D"])
        if len(und) < 4:
            continue
        strict = _host_hist(host, und, d["markers"], d["K"], tol, 20)
        for variant in (30, 31):
            got = _host_hist(host, und, d["markers"], d["K"], tol, variant)
            n, whole, lost, marked = _stats(host)
            assert np.array_equal(got, strict), (i, variant)
            assert lost == 0 and marked == 0
            seen_entries += n
        for variant in (40, 41):
            _host_hist(host, und, d["markers"], d["K"], tol, variant)
            n, whole, lost, marked = _stats(host)
            assert n <= 1 and (lost > 0) == (marked == 1)
            seen_lost += lost
    assert seen_lost > 0 and seen_entries > 10


def test_single_precision_grid_coordinates_stay_inside_the_margin(host, orc):
    """Deferred plain variant (6 .. 11 markers): the voting loop decides "can this root's back-projections be near a
    detection" from grid coordinates computed in SINGLE precision (M = G K T^T Rm per root, 6 multiply-adds and a
    reciprocal per marker); the occupancy grid is dilated by the vote tolerance + 0.25 px for that chain's error.  Here:
    every back-projection the double-precision chain puts inside the grid (the only ones that can vote), on detection
    sets of C3 scenes and of random rigs, differs from it by less than 0.05 px, and every one that lies within the tolerance of a detection finds its cell set
    (the grid marks, row by row, the cells a disc of that radius can reach)."""
    host.host_vote_f32_err.restype = None
    out = (C.c_double * 4)()
    host.host_vote_f32_err(out, 1)
    rng = np.random.default_rng(11)
    cfg = synth.CONFIGS["C3"]
    K, D = synth.camera_for(cfg["rows"], cfg["cols"])
    _, spots = synth.make_scenes_batch(cfg, 6, seed=77)
    for i in range(6):
        und = np.asarray(orc.undistort_points(spots[i].astype(np.float32), K, D), np.float32).astype(float)
        _host_hist(host, und[rng.permutation(len(und))][:10], cfg["markers"], K, 5.0, 0)
    for n_m in (6, 7, 9):      # random rigs, detections that do not come from them
        markers = rng.uniform(-0.2, 0.2, (n_m, 3))
        det = np.stack([rng.uniform(50, 700, 7), rng.uniform(50, 430, 7)], 1)
        _host_hist(host, det, markers, K, 5.0, 0)
    host.host_vote_f32_err(out, 1)
    assert out[1] > 1e5, out[1]            # points compared
    assert out[0] < 0.05, (out[0], out[1])   # (NaN fails too)
    # the occupancy grid itself: every back-projection within the vote tolerance of a detection finds its cell set
    assert out[2] > 100 and out[3] == 0, (out[2], out[3])


def test_near_degenerate_detection_triple_goes_to_the_strict_functions(host, orc):
    """A C3 detection set from the round-4 soak (16 384 frames, tests/soak_votes.py) on which the fast arithmetic ALONE
    loses the four votes of one hypothesis — detections (3, 4, 7) x markers (6, 2, 7): the third bearing lies 4.8e-8 off
    the plane of the other two, f_1 = -2e7, the quartic has a double root — although the oracle is stable under 1-ulp
    changes of its inputs (not the Ferrari corner).  The screen hands the hypothesis to the strict functions: the default
    equals the strict loop and the oracle."""
    cfg = synth.CONFIGS["C3"]
    K, _ = synth.camera_for(cfg["rows"], cfg["cols"])
    markers = np.ascontiguousarray(cfg["markers"], float)
    det = np.load(os.path.join(ROOT, "tests", "data", "c3_degenerate_triple_det.npy"))
    ref = orc.vote_histogram(det, markers, K, 5.0)
    strict = _host_hist(host, det, markers, K, 5.0, 20)
    default = _host_hist(host, det, markers, K, 5.0, 10)
    assert np.array_equal(strict, ref)
    assert np.array_equal(default, strict)
