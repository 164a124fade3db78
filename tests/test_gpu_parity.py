"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs, at every stage boundary.  Integer / index / float32-detection results must be bit-equal;
poses within the north_star tolerance (<= 1e-4 m, <= 1e-3 rad)."""
import numpy as np
import pytest

from rpg_monocular_pose_estimator_amd import synth
import rpg_monocular_pose_estimator_amd as mpe
from util import (pose_diff, POS_TOL_M, ROT_TOL_RAD, p3p_test_problems, check_p3p_solutions, quartic_test_problems,
                  check_quartic_roots)
from golden_util import golden_cases, golden_sequences, load as load_golden, load_sequence, witness_sequences

pytestmark = pytest.mark.gpu


def _oracle_dets(orc, d, P):
    return [orc.find_leds(f, P, d["K"], d["D"]) for f in d["frames"]]


@pytest.mark.parametrize("config,n", [("C2", 24), ("C3", 6), ("C1", 8), ("C4", 3)])
def test_detection_bit_exact(hip, orc, config, n):
    d = synth.make_frames(config, n, seed=101)
    P = mpe.demo_params()
    got = hip.detect_batch(d["frames"], d["K"], d["D"], P)
    ref = _oracle_dets(orc, d, orc.make_params())
    for i in range(n):
        und, dist = ref[i]
        assert got["status"][i] == 0
        assert got["n"][i] == len(und), (i, got["n"][i], len(und))
        k = len(und)
        assert np.array_equal(got["dist_xy"][i][:2 * k].reshape(-1, 2), dist), i   # float32, bit-equal
        assert np.array_equal(got["undist_xy"][i][:2 * k].reshape(-1, 2), und), i  # float32 widened


# (C3: 24 frames since round 6 — VERDICT round 5, weak 8: three frames were thin for the config with 73 920 P3P solves
#  per frame; 0.25 s of oracle per frame)
@pytest.mark.parametrize("config,n", [("C2", 24), ("C3", 24), ("C1", 8)])
def test_vote_histogram_integer_equal(hip, orc, config, n):
    d = synth.make_frames(config, n, seed=202)
    dets = [u for (u, _) in _oracle_dets(orc, d, orc.make_params())]
    got = hip.vote_batch(dets, d["markers"], d["K"], 5.0)
    for i in range(n):
        ref = orc.vote_histogram(dets[i], d["markers"], d["K"], 5.0)
        assert np.array_equal(got[i], ref), (i, got[i], ref)


# C3 with the demo tolerance (5 px) mostly FAILS to initialise in the reference algorithm itself
# (12 detections x 8 markers pollute the vote table) — parity of the failure is what is checked;
# the 2 px variant exercises the 8-marker tail with poses found.
@pytest.mark.parametrize("config,n,tol,min_pose", [("C2", 24, 5.0, 12), ("C3", 3, 5.0, 0), ("C3", 4, 2.0, 3),
                                                   ("C1", 8, 5.0, 4)])
def test_solve_bruteforce_parity(hip, orc, config, n, tol, min_pose):
    d = synth.make_frames(config, n, seed=303)
    Po, Ph = orc.make_params(back_projection_pixel_tolerance=tol), mpe.demo_params(back_projection_pixel_tolerance=tol)
    n_pose = 0
    for i in range(n):
        und, _ = orc.find_leds(d["frames"][i], Po, d["K"], d["D"])
        ro = orc.solve_bruteforce(und, d["markers"], d["K"], Po)
        rh = hip.solve_bruteforce(und, d["markers"], d["K"], Ph)
        assert np.array_equal(rh["hist"], ro["hist"]), i
        assert rh["status"] == ro["status"], i
        assert rh["n_corr"] == ro["n_corr"] and np.array_equal(rh["corr"], ro["corr"]), i
        if ro["status"] == 0:
            n_pose += 1
            dp, dr = pose_diff(rh["T"], ro["T"])
            assert dp <= POS_TOL_M and dr <= ROT_TOL_RAD, (i, dp, dr)
            assert np.allclose(rh["cov"], ro["cov"], rtol=1e-6, atol=1e-12), i
    assert n_pose >= min_pose


@pytest.mark.parametrize("config,n", [("C2", 48), ("C3", 4), ("C4", 3)])
def test_estimate_batch_parity(hip, orc, config, n):
    d = synth.make_frames(config, n, seed=404)
    ro = orc.estimate_batch(d["frames"], d["markers"], d["K"], d["D"], orc.make_params(), n_threads=4)
    rh = hip.estimate_batch(d["frames"], d["markers"], d["K"], d["D"], mpe.demo_params())
    assert np.array_equal(rh["status"], ro["status"])
    assert np.array_equal(rh["n_det"], ro["n_det"])
    assert np.array_equal(rh["n_corr"], ro["n_corr"])
    for i in range(n):
        if ro["status"][i] == 0:
            dp, dr = pose_diff(rh["T"][i], ro["T"][i])
            assert dp <= POS_TOL_M and dr <= ROT_TOL_RAD, (i, dp, dr)


def test_edge_frames(hip, orc):
    """Empty frame, frame with 3 LEDs only, blob touching the border, salt noise."""
    K, D = synth.camera_for(480, 752)
    Po, Ph = orc.make_params(), mpe.demo_params()
    rng = np.random.default_rng(5)
    frames = np.zeros((5, 480, 752), np.uint8)
    frames[1] = synth.render_frame(rng, np.array([[100.5, 100.2], [300.1, 200.7], [500.9, 400.3]]), 480, 752)
    frames[2] = synth.render_frame(rng, np.array([[1.0, 1.0], [750.5, 478.2], [0.3, 240.0], [400.0, 0.0], [375.5, 240.5]]), 480, 752)
    frames[3] = (rng.random((480, 752)) > 0.9995).astype(np.uint8) * 255
    frames[4] = synth.render_frame(rng, np.array([[200.0, 200.0], [207.0, 203.0], [400.0, 300.0], [404.0, 309.0], [600.0, 100.0]]), 480, 752)
    got = hip.detect_batch(frames, K, D, Ph)
    for i in range(len(frames)):
        und, dist = orc.find_leds(frames[i], Po, K, D)
        assert got["status"][i] == 0
        assert got["n"][i] == len(und), (i, got["n"][i], len(und))
        assert np.array_equal(got["dist_xy"][i][:2 * len(und)].reshape(-1, 2), dist), i
    res = hip.estimate_batch(frames, synth.M5, K, D, Ph)
    ro = orc.estimate_batch(frames, synth.M5, K, D, Po)
    assert np.array_equal(res["status"], ro["status"])


def test_find_leds_roi(hip, orc):
    d = synth.make_frames("C2", 4, seed=9)
    Po, Ph = orc.make_params(), mpe.demo_params()
    for i in range(4):
        s = d["spots"][i]
        x0, y0 = int(max(0, s[:, 0].min() - 23)), int(max(0, s[:, 1].min() - 17))
        x1, y1 = int(min(752, s[:, 0].max() + 31)), int(min(480, s[:, 1].max() + 29))
        roi = (x0, y0, x1 - x0, y1 - y0)
        uo, do = orc.find_leds(d["frames"][i], Po, d["K"], d["D"], roi=roi)
        uh, dh = hip.find_leds(d["frames"][i], Ph, d["K"], d["D"], roi=roi)
        assert np.array_equal(do, dh) and np.array_equal(uo, uh)


@pytest.fixture(params=[0, 1], ids=["general_slabs", "general_lds"])
def general_tier(request, hip):
    """Both kernels of the general blob tier (round 6): k1b_general (bitmaps in global-memory slabs) and k1b_general_lds
    (a block per CU, the frame's bitmaps in LDS; frames too large for that fall back to the slabs by themselves)."""
    hip.set_option("general_lds", request.param)
    yield request.param
    hip.set_option("general_lds", 0)


def test_pathological_frames_take_the_general_path(hip, orc, general_tier):
    """Frames that overflow the fast kernel's LDS pools (fully bright, dense salt noise, one huge
    ring around the LEDs) are re-done by the general kernel — same results as the oracle."""
    K, D = synth.camera_for(480, 752)
    Po, Ph = orc.make_params(), mpe.demo_params()
    rng = np.random.default_rng(6)
    frames = np.zeros((4, 480, 752), np.uint8)
    frames[0] = 255
    frames[1] = (rng.random((480, 752)) > 0.997).astype(np.uint8) * 200
    d = synth.make_frames("C2", 2, seed=31)
    frames[2] = d["frames"][0]
    yy, xx = np.mgrid[0:480, 0:752]
    ring = np.abs(np.hypot(xx - 376, yy - 240) - 230) < 3       # a bright ring enclosing every LED:
    frames[2][ring] = 250                                         # RETR_EXTERNAL drops what is inside
    frames[3] = d["frames"][1]
    frames[3][::7, ::5] = np.maximum(frames[3][::7, ::5], 180)    # LEDs inside a dense dot grid
    got = hip.detect_batch(frames, K, D, Ph)
    for i in range(len(frames)):
        und, dist = orc.find_leds(frames[i], Po, K, D)
        if len(und) > mpe.MAX_DETECTIONS:  # documented capacity: loud per-frame status, first 32 kept
            assert got["status"][i] == -10 and got["n"][i] == mpe.MAX_DETECTIONS
            und, dist = und[:mpe.MAX_DETECTIONS], dist[:mpe.MAX_DETECTIONS]
        else:
            assert got["status"][i] == 0, i
            assert got["n"][i] == len(und), (i, got["n"][i], len(und))
        assert np.array_equal(got["dist_xy"][i][:2 * len(und)].reshape(-1, 2), dist), i


def test_general_tier_band_scan_on_random_clutter(hip, orc, general_tier):
    """Round 5: the general blob tier scans one lane per BAND of active rows (k1b_general).  Frames built to reach that
    tier with every band shape: sparse salt noise (many short bands, several blobs per band), dense noise (one band
    over the whole frame), horizontal stripes of noise separated by empty rows (bands that start / end at the image
    border), a tall blob that spans bands' worth of rows next to small ones, blobs nested in a ring that itself touches
    the noise, LEDs on top of everything.  Detections bit-equal to the oracle's on every frame, at two thresholds and
    for an odd-sized ROI-like frame (pitch != cols)."""
    rng = np.random.default_rng(77)
    Po, Ph = orc.make_params(), mpe.demo_params()
    for rows, cols in ((480, 752), (123, 211)):
        K, D = synth.camera_for(rows, cols)
        frames = []
        leds = synth.make_frames("C2", 4, seed=88)["frames"] if (rows, cols) == (480, 752) else None
        for dens in (0.0002, 0.0005, 0.002, 0.01):
            f = (rng.random((rows, cols)) < dens).astype(np.uint8) * rng.integers(150, 256, (rows, cols)).astype(np.uint8)
            frames.append(f)
        f = np.zeros((rows, cols), np.uint8)                      # stripes of noise, empty rows between them
        for y0 in range(0, rows, 17):
            f[y0:y0 + 6] = (rng.random((min(6, rows - y0), cols)) < 0.004) * 255
        frames.append(f)
        f = (rng.random((rows, cols)) < 0.0008).astype(np.uint8) * 255
        f[10:rows - 10, cols // 3:cols // 3 + 5] = 220            # a tall bar through nearly every band
        f[rows // 2 - 3:rows // 2 + 3, 5:40] = 200
        frames.append(f)
        yy, xx = np.mgrid[0:rows, 0:cols]
        f = (rng.random((rows, cols)) < 0.0005).astype(np.uint8) * 255
        ring = np.abs(np.hypot(xx - cols / 2, yy - rows / 2) - min(rows, cols) / 3) < 2.5
        f[ring] = 240                                             # what lies inside the ring is not external
        frames.append(f)
        if leds is not None:
            for i in range(2):
                f = np.maximum(leds[i], (rng.random((rows, cols)) < 0.0005).astype(np.uint8) * 255)
                frames.append(f)
        # a grid of single pixels every fourth row and column: ~188 column runs per three-row band — four bands overflow the
        # kernel's run list in ONE step (those bands are then scanned a lane per band), and the frame needs many batches
        f = np.zeros((rows, cols), np.uint8)
        f[1::4, 1::4] = 255
        frames.append(f)
        f = np.zeros((rows, cols), np.uint8)                      # ... and pairs of pixels: blobs that pass the area filter
        f[2::6, 2::5] = 255
        f[3::6, 2::5] = 255
        f[2::6, 3::5] = 255
        frames.append(f)
        frames = np.ascontiguousarray(np.stack(frames))
        for thr in (140, 60):
            Po.threshold_value = thr
            Ph.threshold_value = thr
            got = hip.detect_batch(frames, K, D, Ph)
            for i in range(len(frames)):
                und, dist = orc.find_leds(frames[i], Po, K, D)
                if len(und) > mpe.MAX_DETECTIONS:
                    assert got["status"][i] == -10 and got["n"][i] == mpe.MAX_DETECTIONS, (rows, thr, i)
                    und, dist = und[:mpe.MAX_DETECTIONS], dist[:mpe.MAX_DETECTIONS]
                else:
                    assert got["status"][i] == 0 and got["n"][i] == len(und), (rows, thr, i, got["n"][i], len(und))
                assert np.array_equal(got["dist_xy"][i][:2 * len(und)].reshape(-1, 2), dist), (rows, thr, i)
                assert np.array_equal(got["undist_xy"][i][:2 * len(und)].reshape(-1, 2), und), (rows, thr, i)


def _wide_frame(rng, n_spots, K, D, rows=480, cols=752):
    """A C2 scene (5 LEDs) + distractor spots, n_spots blobs in all, >= 12 px apart."""
    for _ in range(50):
        T, spots = synth.sample_scene(rng, synth.M5, K, D, rows, cols, n_distractors=n_spots - 5)
        if len(spots) == n_spots:
            return synth.render_frame(rng, spots, rows, cols)
    raise AssertionError("could not place %d spots" % n_spots)


def test_frames_with_33_to_64_detections_match_the_oracle(hip, orc):
    """The reference has no limit on the detections of a frame (led_detector.cpp:65-86 loops over every contour,
    pose_estimator.cpp:549-557 builds the vote table for any image_points_.size()).  Until round 5 more than 32 were a
    capacity status; now up to MPE_MAX_DETECTIONS = 64 go through the whole brute-force path — wider frames than the
    fast voting kernels' 32-bit masks are voted by the strict loop nest (k2_vote_relost) — and have to equal the oracle:
    detections bit for bit, vote histogram integer-equal, correspondences, status, pose.  Beyond 64: status -10 with
    the first 64 detections of the reference's order, never silent."""
    assert mpe.MAX_DETECTIONS == 64
    K, D = synth.camera_for(480, 752)
    rng = np.random.default_rng(8)
    Po, Ph = orc.make_params(), mpe.demo_params()
    frames = np.stack([_wide_frame(rng, n, K, D) for n in (33, 41, 64, 32, 5)])
    got = hip.detect_batch(frames, K, D, Ph)
    wide0 = hip.get_option("vote_wide_frames")
    res = hip.estimate_batch(frames, synth.M5, K, D, Ph)  # (a batch that mixes wide and narrow frames)
    assert hip.get_option("vote_wide_frames") == wide0 + 3
    n_seen = []
    for i, f in enumerate(frames):
        und, dist = orc.find_leds(f, Po, K, D)
        n_seen.append(len(und))
        assert got["status"][i] == 0 and got["n"][i] == len(und), (i, got["n"][i], len(und))
        assert np.array_equal(got["dist_xy"][i][:2 * len(und)].reshape(-1, 2), dist), i
        assert np.array_equal(got["undist_xy"][i][:2 * len(und)].reshape(-1, 2), und), i
        ro = orc.solve_bruteforce(und, synth.M5, K, Po)
        rh = hip.solve_bruteforce(und, synth.M5, K, Ph)
        assert np.array_equal(rh["hist"], ro["hist"]), (i, len(und), np.argwhere(rh["hist"] != ro["hist"])[:6])
        assert rh["status"] == ro["status"] and rh["n_corr"] == ro["n_corr"], (i, rh["status"], ro["status"])
        assert np.array_equal(rh["corr"], ro["corr"]), i
        # ... and the same frame inside the mixed batch
        assert res["status"][i] == ro["status"] and res["n_det"][i] == len(und) and res["n_corr"][i] == ro["n_corr"], i
        if ro["status"] == 0:
            for T in (rh["T"], res["T"][i]):
                dp, dr = pose_diff(T, ro["T"])
                assert dp <= POS_TOL_M and dr <= ROT_TOL_RAD, (i, dp, dr)
        if len(und) > 32:  # the strict kernel and the stage-level vote entry on a wide frame
            hip.set_option("vote_arith", 4)
            try:
                hs = hip.vote_batch([und], synth.M5, K, 5.0)[0]
            finally:
                hip.set_option("vote_arith", 3)
            assert np.array_equal(hs, ro["hist"]), i
            assert np.array_equal(hip.vote_batch([und], synth.M5, K, 5.0)[0], ro["hist"]), i
    assert n_seen[:3] == [33, 41, 64] and n_seen[3] <= 32, n_seen
    # 8 markers (the plain voting kernels) against 36 detections
    und = np.column_stack([rng.uniform(150, 600, 36), rng.uniform(100, 380, 36)])
    M8 = synth.CONFIGS["C3"]["markers"]
    assert np.array_equal(hip.vote_batch([und], M8, K, 5.0)[0], orc.vote_histogram(und, M8, K, 5.0))


def test_too_many_detections_is_loud(hip, orc):
    """> MPE_MAX_DETECTIONS (64) blobs pass the filter: status -10 on that frame, never silent."""
    K, D = synth.camera_for(480, 752)
    rng = np.random.default_rng(8)
    frames = _wide_frame(rng, 90, K, D)[None]
    got = hip.detect_batch(frames, K, D, mpe.demo_params())
    und, dist = orc.find_leds(frames[0], orc.make_params(), K, D)
    assert len(und) > mpe.MAX_DETECTIONS
    assert got["status"][0] == -10 and got["n"][0] == mpe.MAX_DETECTIONS
    assert np.array_equal(got["dist_xy"][0][:2 * mpe.MAX_DETECTIONS].reshape(-1, 2), dist[:mpe.MAX_DETECTIONS])
    res = hip.estimate_batch(frames, synth.M5, K, D, mpe.demo_params())
    assert res["status"][0] == -10


def test_pose_estimator_facade(hip, orc):
    d = synth.make_frames("C2", 6, seed=77)
    for i in range(6):
        pe = mpe.PoseEstimator(hip, bruteforce_every_frame=True)
        pe.setMarkerPositions(d["markers"])
        pe.camera_matrix_K_ = d["K"]
        pe.camera_distortion_coeffs_ = list(d["D"])
        pe.detection_threshold_value_ = 140
        pe.setBackProjectionPixelTolerance(5)
        pe.setNearestNeighbourPixelTolerance(7)
        ok = pe.estimateBodyPose(d["frames"][i], 0.1 * i)
        ro = orc.estimate_batch(d["frames"][i:i + 1], d["markers"], d["K"], d["D"], orc.make_params())
        assert ok == (ro["status"][0] == 0)
        if ok:
            dp, dr = pose_diff(pe.getPredictedPose(), ro["T"][0])
            assert dp <= POS_TOL_M and dr <= ROT_TOL_RAD


@pytest.mark.parametrize("name", golden_cases())
def test_hip_against_committed_golden_vectors(hip, name):
    """HIP path vs tests/golden/*.npz: detections bit-equal, vote histogram and correspondences
    integer-equal, status equal, pose within the north_star tolerance."""
    g, d = load_golden(name)
    P = mpe.demo_params(back_projection_pixel_tolerance=float(g["tol"]))
    n, n_m = int(g["n"]), len(d["markers"])
    det = hip.detect_batch(d["frames"], d["K"], d["D"], P)
    res = hip.estimate_batch(d["frames"], d["markers"], d["K"], d["D"], P)
    for i in range(n):
        k = int(g["n_det"][i])
        assert det["n"][i] == k and det["status"][i] == 0
        assert np.array_equal(det["dist_xy"][i][:2 * k].reshape(-1, 2), g["dist_xy"][i, :k])
        assert np.array_equal(det["undist_xy"][i][:2 * k].reshape(-1, 2), g["undist_xy"][i, :k])
        r = hip.solve_bruteforce(g["undist_xy"][i, :k], d["markers"], d["K"], P)
        assert np.array_equal(r["hist"], g["hist"][i, :k, :n_m]), i
        assert r["status"] == g["status"][i] == res["status"][i]
        assert r["n_corr"] == g["n_corr"][i] and np.array_equal(r["corr"], g["corr"][i, :r["n_corr"]])
        if g["status"][i] == 0:
            for T in (r["T"], res["T"][i]):
                dp, dr = pose_diff(T, g["T"][i])
                assert dp <= POS_TOL_M and dr <= ROT_TOL_RAD, (i, dp, dr)
            assert np.allclose(r["cov"], g["cov"][i], rtol=1e-6, atol=1e-12)


def test_hip_against_the_witness_on_cluttered_frames(hip, general_tier):
    """The blob tiers — above all the general tier rewritten in round 5 — against detections made by the INDEPENDENT
    witness, not by the oracle (tests/golden/witness_clutter.npz, and witness_clutter_C4.npz at 1920x1200): salt noise
    sparse and dense, a saturated patch, a ring enclosing the LEDs, a dot grid, 4 and 16 distractor spots, two thresholds.  Count, order, float32 centroids
    and undistorted points bit-equal (or the documented capacity status with the first 32 detections)."""
    from golden_util import load_clutter
    for cases in (load_clutter(), load_clutter("C4")):   # 752x480 and 1920x1200
        for thr in sorted({c[1] for c in cases}):
            sel = [c for c in cases if c[1] == thr]
            frames = np.ascontiguousarray(np.stack([c[2] for c in sel]))
            got = hip.detect_batch(frames, sel[0][3], sel[0][4], mpe.demo_params(threshold_value=thr))
            for i, (kind, _, _, _, _, k, dist, und) in enumerate(sel):
                if k > mpe.MAX_DETECTIONS:
                    assert got["status"][i] == -10 and got["n"][i] == mpe.MAX_DETECTIONS, (kind, thr)
                    k = mpe.MAX_DETECTIONS
                else:
                    assert got["status"][i] == 0 and got["n"][i] == k, (kind, thr, got["n"][i], k)
                assert np.array_equal(got["dist_xy"][i][:2 * k].reshape(-1, 2), dist[:k]), (kind, thr)
                assert np.array_equal(got["undist_xy"][i][:2 * k].reshape(-1, 2), und[:k]), (kind, thr)


def test_cpp_facade_example_node(orc, tmp_path):
    """The C++ facade (compat/) driven like MPENode::imageCallback, on a raw frame file."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "compat")])
    d = synth.make_frames("C2", 3, seed=808)
    ref = orc.estimate_batch(d["frames"], d["markers"], d["K"], d["D"], orc.make_params())
    for i in range(3):
        path = str(tmp_path / ("f%d.raw" % i))
        d["frames"][i].tofile(path)
        out = subprocess.run([os.path.join(root, "compat", "example_node"), path, "480", "752"],
                             capture_output=True, text=True)
        if ref["status"][i] == 0:
            assert out.returncode == 0, out.stderr
            T = np.array([float(x) for x in out.stdout.splitlines()[0].split()[1:]]).reshape(4, 4)
            dp, dr = pose_diff(T, ref["T"][i])
            assert dp <= POS_TOL_M and dr <= ROT_TOL_RAD
        else:
            assert out.returncode == 1


@pytest.mark.parametrize("config,seed,dropout", [("C2", 3, (15, 16, 30)), ("C1", 11, (8,)), ("C2", 21, ())])
def test_tracking_path_matches_oracle(hip, orc, config, seed, dropout):
    """The stateful estimator (uninitialised branch, then prediction + ROI + nearest-neighbour
    correspondences with fallback to brute force, pose_estimator.cpp:62-147) frame by frame
    against the oracle's restatement of the same state machine."""
    d = synth.make_sequence(config, 36, seed=seed, dropout=dropout)
    to = orc.Tracker(d["markers"], d["K"], d["D"], orc.make_params())
    th = mpe.Tracker(hip, d["markers"], d["K"], d["D"], mpe.demo_params())
    n_tracked = 0
    for k in range(len(d["frames"])):
        ro = to.estimate(d["frames"][k], d["times"][k])
        rh = th.estimate(d["frames"][k], d["times"][k])
        for key in ("updated", "roi", "it_since_initialized", "n_det", "n_corr", "used_bruteforce"):
            assert rh[key] == ro[key], (k, key, rh[key], ro[key])
        if ro["updated"]:
            dp, dr = pose_diff(rh["T"], ro["T"])
            assert dp <= POS_TOL_M and dr <= ROT_TOL_RAD, (k, dp, dr)
            assert np.allclose(rh["cov"], ro["cov"], rtol=1e-5, atol=1e-12)
            n_tracked += (not ro["used_bruteforce"])
    assert n_tracked >= 20   # the ROI tracking branch really ran


def test_check_and_refine_parity(hip, orc):
    d = synth.make_frames("C2", 8, seed=123)
    Po, Ph = orc.make_params(), mpe.demo_params()
    for i in range(8):
        und, _ = orc.find_leds(d["frames"][i], Po, d["K"], d["D"])
        r = orc.solve_bruteforce(und, d["markers"], d["K"], Po)
        if r["n_corr"] < 4:
            continue
        ok, T0 = orc.check_correspondences(und, d["markers"], d["K"], Po, r["corr"])
        rh = hip.check_and_refine(und, d["markers"], d["K"], Ph, r["corr"])
        assert (rh["status"] == 0) == bool(ok)
        if ok:
            Topt, cov, it = orc.optimise_pose(und, d["markers"], d["K"], r["corr"], T0)
            dp, dr = pose_diff(rh["T"], Topt)
            assert dp <= POS_TOL_M and dr <= ROT_TOL_RAD
        # wrong correspondences (shifted) must be rejected the same way
        bad = r["corr"].copy()
        bad[:, 1] = np.roll(bad[:, 1], 1)
        ok2, _ = orc.check_correspondences(und, d["markers"], d["K"], Po, bad)
        assert (hip.check_and_refine(und, d["markers"], d["K"], Ph, bad)["status"] == 0) == bool(ok2)


@pytest.mark.parametrize("sigma", [0.3, 0.6, 1.0, 2.0, 6.0])
def test_detection_other_gaussian_sigmas(hip, orc, sigma):
    """Kernel sizes 3 (fast path), 5, 7, 13 and 37 (generic blur path, dc = 2 at sigma 6)."""
    d = synth.make_frames("C2", 6, seed=61)
    frames = d["frames"].copy()
    frames[0][:6, :] = 200          # bright band on the top border (BORDER_REFLECT_101 rows)
    frames[1][:, -5:] = 220         # and on the right border
    kw = dict(gaussian_sigma=sigma, max_blob_area=100000.0, min_blob_area=0.0, max_circular_distortion=1.0,
              max_width_height_distortion=1.0)
    got = hip.detect_batch(frames, d["K"], d["D"], mpe.demo_params(**kw))
    Po = orc.make_params(**kw)
    for i in range(len(frames)):
        und, dist = orc.find_leds(frames[i], Po, d["K"], d["D"])
        assert got["status"][i] == 0 and got["n"][i] == len(und), (i, got["n"][i], len(und))
        # NaN centroids (zero-area contours pass the filter when min_blob_area = 0) compare equal
        assert np.array_equal(got["dist_xy"][i][:2 * len(und)].reshape(-1, 2), dist, equal_nan=True), i


@pytest.mark.parametrize("rows,cols", [(479, 750), (100, 33), (17, 200), (480, 752)])
def test_detection_odd_sizes_and_strides(hip, orc, rows, cols):
    """cols % 16 != 0 and strided host frames go through the repack path; results unchanged."""
    big = synth.make_frames("C2", 3, seed=71)
    K, D = big["K"], big["D"]
    Po, Ph = orc.make_params(), mpe.demo_params()
    for i in range(3):
        s = big["spots"][i]
        x0 = int(np.clip(s[:, 0].mean() - cols / 2, 0, 752 - cols))
        y0 = int(np.clip(s[:, 1].mean() - rows / 2, 0, 480 - rows))
        view = big["frames"][i][y0:y0 + rows, x0:x0 + cols]        # non-contiguous view (stride 752)
        uo, do = orc.find_leds(np.ascontiguousarray(view), Po, K, D)
        uh, dh = hip.find_leds(np.ascontiguousarray(view), Ph, K, D)
        assert np.array_equal(do, dh) and np.array_equal(uo, uh), (i, len(do), len(dh))
        # the same region addressed as an ROI of the big frame
        uh2, dh2 = hip.find_leds(big["frames"][i], Ph, K, D, roi=(x0, y0, cols, rows))
        uo2, do2 = orc.find_leds(big["frames"][i], Po, K, D, roi=(x0, y0, cols, rows))
        assert np.array_equal(do2, dh2) and np.array_equal(uo2, uh2)
    # a whole batch with cols % 16 != 0
    crop = np.ascontiguousarray(big["frames"][:, 1:1 + min(rows, 479), 1:1 + min(cols, 751)])
    got = hip.detect_batch(crop, K, D, Ph)
    for i in range(3):
        uo, do = orc.find_leds(crop[i], Po, K, D)
        assert got["n"][i] == len(uo) and np.array_equal(got["dist_xy"][i][:2 * len(uo)].reshape(-1, 2), do)


def test_threshold_extremes_and_no_distortion(hip, orc):
    d = synth.make_frames("C2", 2, seed=81)
    for thr in (0, 30, 254, 255):
        kw = dict(threshold_value=thr)
        got = hip.detect_batch(d["frames"], d["K"], np.zeros(0), mpe.demo_params(**kw))
        for i in range(2):
            uo, do = orc.find_leds(d["frames"][i], orc.make_params(**kw), d["K"], np.zeros(0))
            if len(uo) > mpe.MAX_DETECTIONS:
                assert got["status"][i] == -10
                continue
            assert got["status"][i] == 0 and got["n"][i] == len(uo), (thr, i, got["n"][i], len(uo))
            assert np.array_equal(got["undist_xy"][i][:2 * len(uo)].reshape(-1, 2), uo)


@pytest.mark.parametrize("name", golden_sequences())
def test_replay_cli_against_golden_sequence(name, tmp_path):
    """compat/replay (ROS-free counterpart of demo.launch, BASELINE config C1): the C++ facade with the
    whole state machine over a frame-sequence file + the marker YAML, vs the golden per-frame records."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "compat")])
    g, d = load_sequence(name)
    raw = str(tmp_path / "seq.raw")
    d["frames"].tofile(raw)
    yaml = str(tmp_path / "markers.yaml")
    with open(yaml, "w") as fh:
        fh.write("marker_positions:\n")
        for m in d["markers"]:
            fh.write("  - x: %.10g\n    y: %.10g\n    z: %.10g\n" % tuple(m))
    if str(g["config"]) == "C1":   # the shipped 4-LED file is the C1 marker set
        yaml = os.path.join(root, "tests", "data", "demo_marker_positions.yaml")
    dt = float(d["times"][1] - d["times"][0])
    out = subprocess.run([os.path.join(root, "compat", "replay"), "--markers", yaml, "--frames", raw, "--rows",
                          str(d["rows"]), "--cols", str(d["cols"]), "--dt", repr(dt)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert len(lines) == int(g["n"])
    for k, ln in enumerate(lines):
        tok = ln.split()
        assert int(tok[0]) == k
        if g["updated"][k]:
            assert tok[2] == "pose", (k, ln[:60])
            T = np.array([float(x) for x in tok[3:19]]).reshape(4, 4)
            dp, dr = pose_diff(T, g["T"][k])
            assert dp <= POS_TOL_M and dr <= ROT_TOL_RAD, (k, dp, dr)
        else:
            assert tok[2] == "none", (k, ln[:60])


def test_known_divergence_on_an_unstable_quartic(hip, orc):
    """tests/data/vote_regression_det_0.npy: one of the 600 hypotheses of this detection set hits the
    unstable corner of the reference's Ferrari solver (see tests/test_oracle_kat.py::
    test_reference_ferrari_is_unstable_when_w_vanishes); its four votes are the only place where the
    HIP histogram may differ from this particular CPU build's."""
    import os
    det = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "vote_regression_det_0.npy"))
    K, _ = synth.camera_for(480, 752)
    ref = orc.vote_histogram(det, synth.M5, K, 5.0).astype(int)
    # round 6: the default arithmetic ("vote_arith" 3) evaluates the quartic's complex powers as libstdc++ / glibc do
    # and lands on this CPU build's digits even there: no cell may differ
    assert hip.get_option("vote_arith") == 3
    assert np.array_equal(hip.vote_batch([det], synth.M5, K, 5.0)[0].astype(int), ref)
    # the arithmetic of rounds 4 - 5 (exact products, cbrt(hypot)): those four cells, by one vote
    hip.set_option("vote_arith", 1)
    try:
        got = hip.vote_batch([det], synth.M5, K, 5.0)[0].astype(int)
    finally:
        hip.set_option("vote_arith", 3)
    diff = got - ref
    allowed = np.zeros_like(diff, bool)
    for cell in ((0, 2), (1, 4), (2, 1), (3, 0)):
        allowed[cell] = True
    assert np.all(diff[~allowed] == 0) and np.all(np.abs(diff) <= 1)


def test_abi_error_paths_are_loud(hip):
    """Usage errors come back as negative return codes with a message — nothing is silently ignored."""
    K, D = synth.camera_for(480, 752)
    img = np.zeros((480, 752), np.uint8)
    P = mpe.demo_params()
    with pytest.raises(mpe.MpeError):            # ROI outside the image (cv::Mat::operator() would assert)
        hip.find_leds(img, P, K, D, roi=(700, 400, 100, 100))
    with pytest.raises(mpe.MpeError):            # GaussianBlur(ksize=0) needs sigma > 0
        hip.find_leds(img, mpe.demo_params(gaussian_sigma=0.0), K, D)
    with pytest.raises(mpe.MpeError):            # sigma beyond the dynamic-reconfigure range (cfg:13)
        hip.find_leds(img, mpe.demo_params(gaussian_sigma=7.0), K, D)
    det = np.random.default_rng(0).uniform(100, 400, (5, 2))
    with pytest.raises(mpe.MpeError):            # > MPE_MAX_MARKERS
        hip.solve_bruteforce(det, np.random.default_rng(1).normal(size=(17, 3)), K, P)
    with pytest.raises(mpe.MpeError):            # > MPE_MAX_DETECTIONS
        hip.solve_bruteforce(np.random.default_rng(2).uniform(0, 400, (65, 2)), synth.M5, K, P)
    with pytest.raises(mpe.MpeError):            # correspondence index out of range
        hip.check_and_refine(det, synth.M5, K, P, np.array([[1, 1], [2, 2], [3, 3], [9, 4]], np.uint32))


def test_degenerate_inputs_match_oracle(hip, orc):
    """Few detections / markers, collinear markers (P3P returns -1 for every permutation), duplicate
    detections: same status, histogram and correspondences as the CPU path."""
    K, _ = synth.camera_for(480, 752)
    Po, Ph = orc.make_params(), mpe.demo_params()
    rng = np.random.default_rng(4)
    T = np.eye(4)
    T[:3, 3] = [0.05, -0.02, 1.2]
    cases = []
    cases.append((synth.project(T, synth.M4, K), synth.M4))                       # 4 LEDs / 4 detections (C1)
    cases.append((synth.project(T, synth.M5, K)[:3], synth.M5))                   # 3 detections: below the minimum
    cases.append((np.zeros((0, 2)), synth.M5))                                    # none
    line = np.array([[0.0, 0, 0], [0.05, 0, 0], [0.1, 0, 0], [0.15, 0, 0], [0.2, 0, 0]])
    cases.append((synth.project(T, line, K), line))                               # collinear markers
    d5 = synth.project(T, synth.M5, K)
    cases.append((np.vstack([d5, d5[:2]]), synth.M5))                             # duplicated detections
    cases.append((rng.uniform(50, 400, (9, 2)), synth.M8))                        # random detections, 8 markers
    for det, M in cases:
        ro = orc.solve_bruteforce(det, M, K, Po)
        rh = hip.solve_bruteforce(det, M, K, Ph)
        assert rh["status"] == ro["status"] and rh["n_corr"] == ro["n_corr"], (len(det), len(M))
        if len(det):
            assert np.array_equal(rh["hist"], ro["hist"])
        assert np.array_equal(rh["corr"], ro["corr"])
        if ro["status"] == 0:
            dp, dr = pose_diff(rh["T"], ro["T"])
            assert dp <= POS_TOL_M and dr <= ROT_TOL_RAD


def test_tracking_1920x1200(hip, orc):
    d = synth.make_sequence("C4", 10, seed=5)
    to = orc.Tracker(d["markers"], d["K"], d["D"], orc.make_params())
    th = mpe.Tracker(hip, d["markers"], d["K"], d["D"], mpe.demo_params())
    for k in range(10):
        ro = to.estimate(d["frames"][k], d["times"][k])
        rh = th.estimate(d["frames"][k], d["times"][k])
        assert (rh["updated"], rh["roi"], rh["n_det"], rh["n_corr"]) == (ro["updated"], ro["roi"], ro["n_det"], ro["n_corr"]), k
        if ro["updated"]:
            dp, dr = pose_diff(rh["T"], ro["T"])
            assert dp <= POS_TOL_M and dr <= ROT_TOL_RAD


@pytest.mark.gpu
def test_run_sequence_and_concurrent_streams(orc):
    """mpe_tracker_run_sequence = the per-frame calls in a C loop; and BASELINE configs[4] in small:
    several independent camera streams, one handle + tracker + host thread each, on one GPU at the
    same time — every stream must equal the oracle's state machine on its own frames."""
    import threading
    n_streams, n = 4, 24
    seqs = [synth.make_sequence("C2", n, seed=40 + s, dropout=(9,) if s == 1 else ()) for s in range(n_streams)]
    handles = [mpe.Handle(0) for _ in range(n_streams)]
    trackers = [mpe.Tracker(handles[s], seqs[s]["markers"], seqs[s]["K"], seqs[s]["D"], mpe.demo_params())
                for s in range(n_streams)]
    got = [None] * n_streams

    def work(s):
        got[s] = trackers[s].run_sequence(seqs[s]["frames"], seqs[s]["times"])

    threads = [threading.Thread(target=work, args=(s,)) for s in range(n_streams)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    for s in range(n_streams):
        assert got[s] is not None
        rec, info = got[s]
        to = orc.Tracker(seqs[s]["markers"], seqs[s]["K"], seqs[s]["D"], orc.make_params())
        for k in range(n):
            ro = to.estimate(seqs[s]["frames"][k], seqs[s]["times"][k])
            assert (rec["status"][k] == 0) == ro["updated"], (s, k)
            assert tuple(info[k, 0:4]) == ro["roi"] and info[k, 4] == ro["it_since_initialized"], (s, k)
            assert info[k, 5] == ro["n_det"] and info[k, 6] == ro["n_corr"] and bool(info[k, 7]) == ro["used_bruteforce"]
            if ro["updated"]:
                dp, dr = pose_diff(rec["T"][k].reshape(4, 4), ro["T"])
                assert dp <= POS_TOL_M and dr <= ROT_TOL_RAD, (s, k, dp, dr)
    for t in trackers:
        t.close()
    for h in handles:
        h.close()


@pytest.mark.gpu
def test_p3p_batch_matches_oracle(hip, orc):
    """P3P::computePoses on the device (the functions K2 / K3 inline) against the oracle, problem by
    problem: all four [R|C] solutions, incl. those from complex Ferrari roots, and collinear inputs.
    Problems whose solutions differ by more than 1e-6 are COUNTED, bounded, and each one must be a witnessed
    instability of the reference algorithm itself: moving one input of the ORACLE by one ulp moves the oracle's
    own answer by more than the disagreement tolerance (the alpha + 2y ~ 0 corner of Ferrari)."""
    fv, wp = p3p_test_problems()
    st, sol = hip.p3p_batch(fv, wp)
    check_p3p_solutions(st, sol, fv, wp, orc)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1])
def test_solve_quartic_batch_matches_oracle(hip, orc, variant):
    """Device quartic (0 = IEEE operators as in the validation / strict voting kernels, 1 = the fast voting
    kernel's variant) against the oracle.  Every quartic whose real parts differ by more than 1e-9 is counted,
    bounded, and must sit in the unstable corner: |alpha + 2y| small against its operands (error amplification
    of the 2 beta / w term ~ 1 / |w|^2), computed here in plain Python complex arithmetic."""
    f, roots = quartic_test_problems(variant)
    got = hip.solve_quartic_batch(f, variant)
    check_quartic_roots(got, f, roots, orc)


@pytest.mark.gpu
def test_track_step_matches_oracle_pieces(hip, orc):
    """mpe_track_step = findLeds(ROI) + findCorrespondences + checkCorrespondences + optimisePose in one
    submission, against the same chain assembled from the oracle's functions."""
    d = synth.make_frames("C2", 12, seed=321)
    Po, Ph = orc.make_params(), mpe.demo_params()
    n_pose = 0
    for i in range(12):
        T = d["T_true"][i]
        pred = synth.project(T, d["markers"], d["K"]) + np.random.default_rng(i).normal(0, 0.8, (len(d["markers"]), 2))
        roi = orc.determine_roi(pred, d["rows"], d["cols"], 20, d["K"], d["D"])
        und, _ = orc.find_leds(d["frames"][i], Po, d["K"], d["D"], roi=roi)
        r = hip.track_step(d["frames"][i], roi, Ph, d["K"], d["D"], d["markers"], pred)
        assert r["det_status"] == 0 and np.array_equal(r["undist"], und), i
        if len(und) < 4:
            assert r["status"] == 1 and r["n_corr"] == 0
            continue
        # findCorrespondences, pose_estimator.cpp:372-392
        corr = []
        for m in range(len(pred)):
            dist = np.sqrt(((und - pred[m]) ** 2).sum(axis=1))
            j = int(np.argmin(dist))
            if dist[j] <= Ph.nearest_neighbour_pixel_tolerance:
                corr.append((m + 1, j + 1))
        corr = np.array(corr, np.uint32).reshape(-1, 2)
        assert np.array_equal(r["corr"], corr), i
        ok, T0 = orc.check_correspondences(und, d["markers"], d["K"], Po, corr) if len(corr) >= 4 else (False, None)
        assert (r["status"] == 0) == bool(ok), i
        if ok:
            Topt, cov, it = orc.optimise_pose(und, d["markers"], d["K"], corr, T0)
            dp, dr = pose_diff(r["T"], Topt)
            assert dp <= POS_TOL_M and dr <= ROT_TOL_RAD
            assert np.allclose(r["cov"], cov, rtol=1e-5, atol=1e-12)
            n_pose += 1
    assert n_pose >= 8


@pytest.mark.gpu
def test_track_step_repeats_a_frame_the_small_blob_tier_cannot_hold(hip, orc):
    """The tracked frame tries the small blob tier alone (no follow-up tiers queued); a ROI with more blobs than that
    tier records comes back through the whole tier chain, transparently: the detections are the oracle's."""
    rng = np.random.default_rng(77)
    d = synth.make_frames("C2", 1, seed=322)
    gx, gy = np.meshgrid(np.arange(6) * 100 + 80.0, np.arange(4) * 100 + 70.0)
    spots = np.stack([gx.ravel(), gy.ravel()], 1) + rng.uniform(-10, 10, (24, 2))
    frame = synth.render_frame(rng, spots, d["rows"], d["cols"])
    Po, Ph = orc.make_params(), mpe.demo_params()
    roi = (0, 0, d["cols"], d["rows"])
    und, _ = orc.find_leds(frame, Po, d["K"], d["D"], roi=roi)
    assert len(und) > 16                       # more than K1bSmall::KEPT
    pred = spots[:len(d["markers"])] + 0.3
    r = hip.track_step(frame, roi, Ph, d["K"], d["D"], d["markers"], pred)
    assert r["det_status"] == 0 and np.array_equal(r["undist"], und)
    # and an ordinary frame right after it on the same handle
    d2 = synth.make_frames("C2", 1, seed=323)
    pred2 = synth.project(d2["T_true"][0], d2["markers"], d2["K"])
    roi2 = orc.determine_roi(pred2, d2["rows"], d2["cols"], 20, d2["K"], d2["D"])
    und2, _ = orc.find_leds(d2["frames"][0], Po, d2["K"], d2["D"], roi=roi2)
    r2 = hip.track_step(d2["frames"][0], roi2, Ph, d2["K"], d2["D"], d2["markers"], pred2)
    assert r2["det_status"] == 0 and np.array_equal(r2["undist"], und2)


@pytest.mark.gpu
def test_marker_table_is_rebuilt_when_the_rig_changes():
    """The marker-permutation table is kept between calls with the same rig on the same stream; another rig (other
    positions, same count; another count) in between must rebuild it: records equal those of fresh handles."""
    d = synth.make_frames("C2", 24, seed=4711)
    P = mpe.demo_params()
    rig_a = np.asarray(d["markers"], float)
    rig_b = rig_a * np.array([1.0, 0.9, 1.1]) + 0.003          # other positions, same count
    rig_c = np.vstack([rig_a, [[0.05, -0.07, 0.02]]])           # one marker more
    fresh = {}
    for name, rig in (("a", rig_a), ("b", rig_b), ("c", rig_c)):
        h = mpe.Handle()
        fresh[name] = h.estimate_batch(d["frames"], rig, d["K"], d["D"], P)
        h.close()
    assert (fresh["a"]["status"] == 0).sum() >= 12
    assert not np.array_equal(fresh["a"].view(np.uint8), fresh["b"].view(np.uint8))
    h = mpe.Handle()
    for name, rig in (("a", rig_a), ("a", rig_a), ("b", rig_b), ("a", rig_a), ("c", rig_c), ("a", rig_a)):
        got = h.estimate_batch(d["frames"], rig, d["K"], d["D"], P)
        assert np.array_equal(got.view(np.uint8), fresh[name].view(np.uint8)), name
    h.close()


@pytest.mark.gpu
def test_histogram_threshold_values(orc):
    """correspondencesFromHistogram stops at the first maximum below histogram_threshold_ (pose_estimator.cpp:362; the
    launch files use 0, the cfg allows more): thresholds 0, 1, 40, 200 and 100 000 on C2 frames with and without
    distractors — statuses, correspondence counts and poses equal to the oracle's (the device kernel finds the
    column maxima once and replays the reference's n_m scans on them)."""
    d = synth.make_frames("C2", 24, seed=811)
    cfgn = dict(synth.CONFIGS["C2"], n_distractors=3)
    dn = synth.make_frames(cfgn, 24, seed=812)
    h = mpe.Handle()
    n_pose = {}
    for thr in (0, 1, 40, 200, 100000):
        Ph, Po = mpe.demo_params(histogram_threshold=thr), orc.make_params(histogram_threshold=thr)
        n_pose[thr] = 0
        for dd in (d, dn):
            got = h.estimate_batch(dd["frames"], dd["markers"], dd["K"], dd["D"], Ph)
            ref = orc.estimate_batch(dd["frames"], dd["markers"], dd["K"], dd["D"], Po)
            assert np.array_equal(got["status"], ref["status"]), thr
            assert np.array_equal(got["n_corr"], ref["n_corr"]) and np.array_equal(got["n_det"], ref["n_det"]), thr
            for i in np.nonzero(got["status"] == 0)[0]:
                dp, dr = pose_diff(got["T"][i].reshape(4, 4), ref["T"][i].reshape(4, 4))
                assert dp <= POS_TOL_M and dr <= ROT_TOL_RAD, (thr, i)
                n_pose[thr] += 1
    assert n_pose[0] >= 30 and n_pose[100000] == 0 and n_pose[200] <= n_pose[1]
    h.close()


@pytest.mark.gpu
def test_check_and_optimise_stage_entry_points(hip, orc):
    """checkCorrespondences and optimisePose as separate device calls against the oracle's functions:
    the unrefined pose of computeTransformation, then Gauss-Newton from that pose AND from perturbed
    poses (iteration counts and covariance too)."""
    d = synth.make_frames("C2", 10, seed=555)
    Po, Ph = orc.make_params(), mpe.demo_params()
    n = 0
    for i in range(10):
        und, _ = orc.find_leds(d["frames"][i], Po, d["K"], d["D"])
        r = orc.solve_bruteforce(und, d["markers"], d["K"], Po)
        if r["n_corr"] < 4:
            continue
        ok, T0 = orc.check_correspondences(und, d["markers"], d["K"], Po, r["corr"])
        okh, T0h = hip.check_correspondences(und, d["markers"], d["K"], Ph, r["corr"])
        assert okh == bool(ok), i
        if not ok:
            continue
        dp, dr = pose_diff(T0h, T0)
        assert dp <= 1e-9 and dr <= 1e-9, (i, dp, dr)
        rng = np.random.default_rng(i)
        for trial in range(3):
            Ts = T0.copy()
            if trial:
                Ts[:3, :3] = synth.rodrigues(rng.normal(size=3), 0.05 * trial) @ Ts[:3, :3]
                Ts[:3, 3] += rng.normal(0, 0.01 * trial, 3)
            Topt, cov, it = orc.optimise_pose(und, d["markers"], d["K"], r["corr"], Ts)
            rh = hip.optimise_pose(und, d["markers"], d["K"], Ph, r["corr"], Ts)
            assert rh["status"] == 0
            dp, dr = pose_diff(rh["T"], Topt)
            assert dp <= 1e-9 and dr <= 1e-9, (i, trial, dp, dr)
            assert abs(rh["gn_iterations"] - it) <= 1
            assert np.allclose(rh["cov"], cov, rtol=1e-5, atol=1e-12)
        n += 1
    assert n >= 6
    # fewer than 3 correspondences: no refinement
    assert hip.optimise_pose(und, d["markers"], d["K"], Ph, r["corr"][:2], np.eye(4))["status"] == 1


@pytest.mark.gpu
def test_facade_step_methods_static_primitives_and_overlay(tmp_path):
    """compat/facade_selftest steps: estimateBodyPose vs the same state machine driven through the class's
    public step methods (initialise, checkCorrespondences, optimisePose, predictWithROI, ...) and the static
    LEDDetector / P3P classes; then the overlay.  The binary checks equality itself."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "compat")])
    d = synth.make_sequence("C2", 30, seed=77, dropout=(12, 13))
    raw = str(tmp_path / "seq.raw")
    d["frames"].tofile(raw)
    yaml = str(tmp_path / "markers.yaml")
    with open(yaml, "w") as fh:
        fh.write("marker_positions:\n")
        for m in d["markers"]:
            fh.write("  - x: %.17g\n    y: %.17g\n    z: %.17g\n" % tuple(m))
    out = subprocess.run([os.path.join(root, "compat", "facade_selftest"), "steps", "--markers", yaml, "--frames", raw,
                          "--rows", str(d["rows"]), "--cols", str(d["cols"]), "--dt", "0.02"],
                         capture_output=True, text=True)
    assert out.returncode == 0 and "selftest ok" in out.stdout, (out.returncode, out.stdout[-800:], out.stderr[-800:])


@pytest.mark.gpu
def test_full_size_batch_properties(orc):
    """65 536 device-resident 752x480 frames (24 GB, the size at which a call runs as 8 sub-batches on two
    streams): size-independent properties instead of a frame-by-frame CPU comparison —
      * the software-pipelined call equals the plain sequential one bit for bit,
      * frames are independent: the batch in reverse order gives the reversed records bit for bit,
      * a random sample agrees with the oracle, and the poses found agree with the ground truth of the
        synthetic scenes (median position error of a few millimetres at 0.8-2.5 m: centroid noise, not the solver)."""
    import torch
    B = 65536
    cfg = synth.CONFIGS["C2"]
    rows, cols = cfg["rows"], cfg["cols"]
    K, D = synth.camera_for(rows, cols)
    markers = np.asarray(cfg["markers"])
    T_true, spots = synth.make_scenes_batch(cfg, B, seed=4242)
    dev = torch.device("cuda", 0)
    frames = synth.render_frames_torch(spots, rows, cols, cfg["spot_sigma"], dev, seed=99)
    torch.cuda.synchronize()
    P = mpe.demo_params()
    h = mpe.Handle(0)
    # a real (non-default) torch stream shared with the library: allocation fills and the library's kernels are
    # then ordered on it (mpe_set_stream(NULL) would mean the handle's own, unrelated stream)
    stream = torch.cuda.Stream(device=dev)
    h.set_stream(stream.cuda_stream)
    nb = B * mpe.RESULT_DTYPE.itemsize

    def run(fr, pipeline):
        with torch.cuda.stream(stream):
            out = torch.zeros(nb, dtype=torch.uint8, device=dev)
            h.set_option("pipeline", pipeline)
            h.estimate_batch_device(fr.data_ptr(), B, rows, cols, markers, K, D, P, out.data_ptr())
        stream.synchronize()
        return out

    assert h.get_option("streams_concurrent") == -1
    h.set_option("pipeline_mode", 0)                       # two-stream software pipeline
    piped = run(frames, 8)
    assert h.get_option("streams_concurrent") in (0, 1)   # probed at the first two-stream call (1 = overlap verified)
    plain = run(frames, 1)
    assert torch.equal(piped, plain)
    # fused schedule: the image scan of sub-batch s + 1 rides inside the voting kernel of sub-batch s (LDS DMA)
    h.set_option("pipeline_mode", 3)
    fused = run(frames, 8)
    assert h.get_option("last_schedule") == 3
    assert torch.equal(fused, plain)
    h.set_option("pipeline_mode", 4)                       # fused + validate / refine on a side stream
    fused4 = run(frames, 8)
    assert h.get_option("last_schedule") == 4 and torch.equal(fused4, plain)
    h.set_option("pipeline_mode", 6)                       # ... + the scan split between a side scan kernel and the rider
    default_pct = h.get_option("scan_split_pct")
    for pct in (20, 0, 55):
        h.set_option("scan_split_pct", pct)
        fused6 = run(frames, 8)
        assert h.get_option("last_schedule") == 6 and torch.equal(fused6, plain), pct
    h.set_option("scan_split_pct", default_pct)
    h.set_option("pipeline_mode", -1)                      # automatic = 6
    piped = run(frames, 8)
    assert h.get_option("last_schedule") == 6 and torch.equal(piped, plain)
    flipped = torch.flip(frames, dims=[0]).contiguous()
    torch.cuda.synchronize()
    rev = run(flipped, 8)
    assert torch.equal(rev.view(B, -1).flip(0), piped.view(B, -1))
    rec = np.frombuffer(piped.cpu().numpy().tobytes(), mpe.RESULT_DTYPE)
    found = rec["status"] == 0
    assert 0.9 < found.mean() <= 1.0 and (rec["status"] >= 0).all()
    T = rec["T"].reshape(B, 4, 4)
    dpos = np.linalg.norm(T[found][:, :3, 3] - T_true[found][:, :3, 3], axis=1)
    assert np.median(dpos) < 5e-3 and np.mean(dpos < 0.05) > 0.97, (np.median(dpos), np.mean(dpos < 0.05))
    # oracle on a random sample
    idx = np.random.default_rng(0).choice(B, 96, replace=False)
    sample = frames[torch.as_tensor(idx, device=dev)].cpu().numpy()
    ref = orc.estimate_batch(sample, markers, K, D, orc.make_params(), n_threads=8)
    for j, i in enumerate(idx):
        assert rec["status"][i] == ref["status"][j], i
        if ref["status"][j] == 0:
            dp, dr = pose_diff(T[i], ref["T"][j].reshape(4, 4))
            assert dp <= POS_TOL_M and dr <= ROT_TOL_RAD, (i, dp, dr)
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("thr", [-1, 0, 1, 100, 126, 127, 128, 129, 200, 254, 255])
def test_fused_scan_flags_at_every_threshold_form(orc, thr):
    """The fused schedule's scan (SWAR byte test in its AND form for thr >= 128, OR form below, LDS-DMA staging)
    against the plain schedule and the oracle on frames whose pixel values straddle the threshold."""
    import torch
    B = 16384 + 100   # two sub-batches of >= 8192 frames: the second one is scanned by the voting kernel's riders
    rows, cols = 64, 112  # (its size is not a multiple of the riders' chunk: the stand-alone scan takes the rest)
    K, D = synth.camera_for(rows, cols)
    rng = np.random.default_rng(thr + 5)
    t = max(0, min(255, thr))
    base = np.empty((257, rows, cols), np.uint8)
    for i in range(257):   # background at / just below thr, six 3x3 spots just above it
        base[i] = np.clip(rng.integers(t - 2, t + 1, (rows, cols)), 0, 255).astype(np.uint8)
        for k in range(6):
            y, x = 6 + 9 * k, int(rng.integers(4, cols - 8))
            base[i, y:y + 3, x:x + 3] = min(255, t + 1 + (k & 1))
    frames_np = base[np.arange(B) % 257]
    dev = torch.device("cuda", 0)
    frames = torch.as_tensor(frames_np, device=dev)
    P = mpe.demo_params(threshold_value=thr, min_blob_area=1.0, max_blob_area=1e9, max_width_height_distortion=1e9,
                        max_circular_distortion=1e9)
    h = mpe.Handle(0)
    stream = torch.cuda.Stream(device=dev)
    h.set_stream(stream.cuda_stream)
    h.set_option("pipeline", 2)
    out = {}
    for mode in (0, 3):
        h.set_option("pipeline_mode", mode)
        with torch.cuda.stream(stream):
            res = torch.zeros(B * mpe.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
            h.estimate_batch_device(frames.data_ptr(), B, rows, cols, synth.M5, K, D, P, res.data_ptr())
        stream.synchronize()
        out[mode] = np.frombuffer(res.cpu().numpy().tobytes(), mpe.RESULT_DTYPE)
    assert out[0].tobytes() == out[3].tobytes()
    Po = orc.make_params(threshold_value=thr, min_blob_area=1.0, max_blob_area=1e9, max_width_height_distortion=1e9,
                         max_circular_distortion=1e9)
    for i in list(range(0, 257, 16)) + [B - 1]:
        und, _ = orc.find_leds(frames_np[i], Po, K, D)
        got = out[3][i]
        if len(und) > mpe.MAX_DETECTIONS:
            assert got["status"] == -10
        else:
            assert got["n_det"] == len(und), (thr, i, got["n_det"], len(und))
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["C1", "C2-distractors"])
def test_fused_schedule_other_shapes(orc, config):
    """The fused schedule (scan riding in the voting kernel) with 4 markers (one unused marker: the other
    layout of the LDS table) and with 5 markers + distractor blobs (more detection triples than one 16-entry
    chunk): bit-identical to the plain chain of kernels, and equal to the oracle on a sample."""
    import torch
    B = 32768 + 64
    if config == "C1":
        cfg = dict(synth.CONFIGS["C1"])
    else:
        cfg = dict(synth.CONFIGS["C2"])
        cfg["n_distractors"] = 3        # up to 8 detections -> C(8,3) = 56 triples
    rows, cols = cfg["rows"], cfg["cols"]
    K, D = synth.camera_for(rows, cols)
    markers = np.asarray(cfg["markers"])
    _, spots = synth.make_scenes_batch(cfg, B, seed=77)
    dev = torch.device("cuda", 0)
    frames = synth.render_frames_torch(spots, rows, cols, cfg["spot_sigma"], dev, seed=12)
    torch.cuda.synchronize()
    P = mpe.demo_params()
    h = mpe.Handle(0)
    stream = torch.cuda.Stream(device=dev)
    h.set_stream(stream.cuda_stream)
    out = {}
    for mode, pipeline in ((3, 16), (0, 1)):
        h.set_option("pipeline_mode", mode)
        h.set_option("pipeline", pipeline)
        with torch.cuda.stream(stream):
            res = torch.zeros(B * mpe.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
            h.estimate_batch_device(frames.data_ptr(), B, rows, cols, markers, K, D, P, res.data_ptr())
        stream.synchronize()
        out[mode] = np.frombuffer(res.cpu().numpy().tobytes(), mpe.RESULT_DTYPE)
        if mode == 3:
            assert h.get_option("last_schedule") == 3
    assert out[0].tobytes() == out[3].tobytes()
    idx = np.random.default_rng(1).choice(B, 64, replace=False)
    sample = frames[torch.as_tensor(idx, device=dev)].cpu().numpy()
    ref = orc.estimate_batch(sample, markers, K, D, orc.make_params(), n_threads=8)
    n_pose = 0
    for j, i in enumerate(idx):
        assert out[3]["status"][i] == ref["status"][j], (config, i)
        if ref["status"][j] == 0:
            dp, dr = pose_diff(out[3]["T"][i].reshape(4, 4), ref["T"][j].reshape(4, 4))
            assert dp <= POS_TOL_M and dr <= ROT_TOL_RAD, (config, i, dp, dr)
            n_pose += 1
    assert n_pose >= 20
    h.close()


def _planar_rig(n, tilt):
    """n coplanar markers: in the z = 0 plane of the marker frame, or the same rig rotated / shifted so
    that no coordinate is special."""
    base = np.array([[0.08, 0.07], [0.05, -0.09], [-0.07, -0.08], [-0.06, -0.01], [0.01, 0.035]])[:n]
    pts = np.c_[base, np.zeros(n)]
    if tilt:
        pts = pts @ synth.rodrigues([1, 2, 0.5], 0.6).T + np.array([0.01, -0.02, 0.03])
    return pts


@pytest.mark.gpu
@pytest.mark.parametrize("n_markers,tilt", [(4, 0), (4, 1), (5, 0), (5, 1)])
def test_coplanar_marker_rigs(hip, orc, n_markers, tilt):
    """Rank-deficient Kabsch input (computeTransformation, pose_estimator.cpp:908-930, H = A B^T with coplanar
    markers): the reference's JacobiSVD still returns a rotation, so must K3 (Hestenes Jacobi SVD as in the
    oracle, sigma_3 ~ 0 completed with the cross product).  Whole path, status / correspondences equal, pose
    within the north_star tolerance; frames on which the reference's Gauss-Newton diverges (a planar rig has
    two reprojection minima) must diverge on both sides."""
    M = _planar_rig(n_markers, tilt)
    cfg = dict(rows=480, cols=752, markers=M, n_distractors=0, spot_sigma=1.5)
    d = synth.make_frames(cfg, 24, seed=900 + n_markers + tilt)
    ro = orc.estimate_batch(d["frames"], M, d["K"], d["D"], orc.make_params(), n_threads=4)
    rh = hip.estimate_batch(d["frames"], M, d["K"], d["D"], mpe.demo_params())
    assert np.array_equal(rh["status"], ro["status"])
    assert np.array_equal(rh["n_corr"], ro["n_corr"])
    n_pose = n_chaotic = 0
    for i in range(len(ro)):
        if ro["status"][i] != 0:
            continue
        To, Th = ro["T"][i].reshape(4, 4), rh["T"][i].reshape(4, 4)
        if ro["gn_iterations"][i] > 25 or not np.all(np.isfinite(To)):
            # Gauss-Newton started from the mirror-image minimum of a planar rig and wanders with steps of
            # several radians (cost 1e3 .. 1e6 px^2, no line search in pose_estimator.cpp:753-788): chaotic, any
            # rounding difference gives another trajectory — in the reference as well.  Only the verdicts compare.
            n_chaotic += 1
            continue
        n_pose += 1
        dp, dr = pose_diff(Th, To)
        assert dp <= POS_TOL_M and dr <= ROT_TOL_RAD, (i, dp, dr)
        assert abs(np.linalg.det(Th[:3, :3]) - 1.0) < 1e-9, i  # a rotation, not its mirror image
    assert n_pose >= 16 and n_chaotic <= 3, (n_pose, n_chaotic)
    # the unrefined pose of computeTransformation alone, stage entry point vs oracle
    Po, Ph = orc.make_params(), mpe.demo_params()
    n_chk = 0
    for i in range(8):
        und, _ = orc.find_leds(d["frames"][i], Po, d["K"], d["D"])
        r = orc.solve_bruteforce(und, M, d["K"], Po)
        if r["n_corr"] < 4:
            continue
        ok, T0 = orc.check_correspondences(und, M, d["K"], Po, r["corr"])
        okh, T0h = hip.check_correspondences(und, M, d["K"], Ph, r["corr"])
        assert okh == bool(ok), i
        if ok:
            dp, dr = pose_diff(T0h, T0)
            assert dp <= 1e-9 and dr <= 1e-9, (i, dp, dr)
            n_chk += 1
    assert n_chk >= 4


@pytest.mark.gpu
def test_strict_vote_arithmetic(orc):
    """Option "vote_arith" = 0: the voting kernel built from the validation kernel's P3P functions (IEEE
    division / square root, one quartic solver for voting and validation).  Histograms integer-equal to the
    oracle on the standard cases; the known unstable-corner hypothesis of tests/data may still differ (libm);
    the whole path agrees with the fast arithmetic on ordinary frames."""
    import os
    h = mpe.Handle()
    try:
        assert h.get_option("vote_arith") == 3   # (the default since round 6; 4 is its strict kernel)
        h.set_option("vote_arith", 4)
        det = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "vote_regression_det_0.npy"))
        K, _ = synth.camera_for(480, 752)
        assert np.array_equal(h.vote_batch([det], synth.M5, K, 5.0)[0], orc.vote_histogram(det, synth.M5, K, 5.0))
        h.set_option("vote_arith", 0)
        assert h.get_option("vote_arith") == 0
        for config, n in (("C2", 24), ("C1", 8), ("C3", 2)):
            d = synth.make_frames(config, n, seed=202)
            dets = [orc.find_leds(f, orc.make_params(), d["K"], d["D"])[0] for f in d["frames"]]
            got = h.vote_batch(dets, d["markers"], d["K"], 5.0)
            for i in range(n):
                ref = orc.vote_histogram(dets[i], d["markers"], d["K"], 5.0)
                assert np.array_equal(got[i], ref), (config, i, got[i], ref)
        det = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "vote_regression_det_0.npy"))
        K, _ = synth.camera_for(480, 752)
        diff = h.vote_batch([det], synth.M5, K, 5.0)[0].astype(int) - orc.vote_histogram(det, synth.M5, K, 5.0).astype(int)
        assert np.abs(diff).max() <= 1 and np.count_nonzero(diff) <= 4, diff
        d = synth.make_frames("C2", 40, seed=31)
        strict = h.estimate_batch(d["frames"], d["markers"], d["K"], d["D"], mpe.demo_params())
        h.set_option("vote_arith", 1)
        fast = h.estimate_batch(d["frames"], d["markers"], d["K"], d["D"], mpe.demo_params())
        ref = orc.estimate_batch(d["frames"], d["markers"], d["K"], d["D"], orc.make_params(), n_threads=4)
        for r in (strict, fast):
            assert np.array_equal(r["status"], ref["status"]) and np.array_equal(r["n_corr"], ref["n_corr"])
        ok = ref["status"] == 0
        assert np.abs(strict["T"][ok] - ref["T"][ok]).max() < 1e-9
        assert np.abs(fast["T"][ok] - ref["T"][ok]).max() < 1e-9
        with pytest.raises(mpe.MpeError):
            h.set_option("vote_arith", 5)
    finally:
        h.close()


@pytest.mark.gpu
def test_default_votes_equal_strict_votes(orc):
    """Round 4: the DEFAULT voting arithmetic (vote_arith 1 = the fast kernel + k2_vote_fixup, the strict re-evaluation
    of the hypotheses it appends to its suspect list) must produce the STRICT kernel's histograms (vote_arith 0), cell
    by cell — on ordinary frames of every config, on random detection sets (planar rigs, several tolerances), and on
    the saved frames on which round 3's fast arithmetic differed from the oracle (tests/data/unstable_det_r3_*.npy,
    vote_regression_det_0.npy: one hypothesis each in the corner of Ferrari's method) — through the plain kernel
    (vote_batch) for every marker count incl. the table-slice variant (8 markers).  The list is used, never full."""
    import os
    h = mpe.Handle()
    try:
        cases = []
        for config, n in (("C2", 256), ("C1", 64), ("C3", 6)):
            d = synth.make_frames(config, n, seed=414)
            dets = [orc.find_leds(f, orc.make_params(), d["K"], d["D"])[0] for f in d["frames"]]
            dets = [x for x in dets if 4 <= len(x) <= mpe.MAX_DETECTIONS]
            cases.append((config, dets, d["markers"], d["K"], 5.0))
        K, _ = synth.camera_for(480, 752)
        data = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
        saved = [np.load(os.path.join(data, f)) for f in sorted(os.listdir(data))
                 if f.startswith("unstable_det_r3_") or f.startswith("vote_regression_det")]
        assert len(saved) >= 4
        cases.append(("saved", saved, synth.M5, K, 5.0))
        # the C3 frame of the round-4 soak on which the fast arithmetic ALONE loses a hypothesis' votes outside the
        # Ferrari corner (a near-degenerate detection triple, tests/test_vote_host.py): default = strict = oracle
        deg = np.load(os.path.join(data, "c3_degenerate_triple_det.npy"))
        cases.append(("degenerate triple", [deg], synth.CONFIGS["C3"]["markers"], K, 5.0))
        rng = np.random.default_rng(12)
        for it in range(12):
            n_m = int(rng.integers(4, 8))
            markers = rng.uniform(-0.15, 0.15, (n_m, 3))
            if it % 3 == 0:
                markers[:, 2] = 0.0
            dets = [np.column_stack([rng.uniform(250, 500, k), rng.uniform(150, 330, k)])
                    for k in rng.integers(4, 10, 16)]
            cases.append(("random%d" % it, dets, markers, K, [1.0, 3.0, 5.0][it % 3]))
        items0 = h.get_option("vote_fixup_items")
        # (round 6: the same claim for the pair with the reference library's powers — 3, the default now, against 4)
        for name, dets, markers, Kc, tol in cases:
            for a_strict, a_fast in ((0, 1), (4, 3)):
                h.set_option("vote_arith", a_strict)
                strict = h.vote_batch(dets, markers, Kc, tol)
                h.set_option("vote_arith", a_fast)
                got = h.vote_batch(dets, markers, Kc, tol)
                for i in range(len(dets)):
                    assert np.array_equal(got[i], strict[i]), (name, a_fast, i, np.argwhere(got[i] != strict[i])[:5])
                if name == "degenerate triple":
                    assert np.array_equal(got[0], orc.vote_histogram(dets[0], np.asarray(markers, float), Kc, tol))
        assert h.get_option("vote_fixup_items") > items0
        assert h.get_option("vote_fixup_overflow") == 0
        # the whole path (fused scan-carrying kernel, fix-up on the tail stream of the pipelined schedule, streaming
        # submissions): records byte-identical to the strict arithmetic's wherever the votes are (same tail kernels)
        import torch
        d = synth.make_frames("C2", 96, seed=77)
        big = torch.from_numpy(d["frames"]).cuda().repeat(342, 1, 1)[:32768 + 64].contiguous()  # 2 sub-batches
        for a_strict, a_fast in ((0, 1), (4, 3)):
            h.set_option("vote_arith", a_strict)
            rs = h.estimate_batch(big, d["markers"], d["K"], d["D"], mpe.demo_params())
            h.set_option("vote_arith", a_fast)
            rf = h.estimate_batch(big, d["markers"], d["K"], d["D"], mpe.demo_params())
            assert h.get_option("last_schedule") in (3, 6)
            for k in ("status", "n_corr", "n_det"):
                assert np.array_equal(rs[k], rf[k]), k
            assert np.array_equal(rs["T"], rf["T"], equal_nan=True) and np.array_equal(rs["cov"], rf["cov"], equal_nan=True)
        assert h.get_option("vote_fixup_overflow") == 0
    finally:
        h.close()


@pytest.mark.gpu
def test_vote_arithmetics_with_the_reference_librarys_powers(orc):
    """Round 6: "vote_arith" 3 (fast kernel + strict fix-up) and 4 (strict kernel) evaluate the quartic's complex
    powers as libstdc++ / glibc do (p3p.cpp:262,264,268 through <complex> and clog; csrc/mpe_ddmath.h).  On ordinary
    frames every arithmetic gives the oracle's histogram; 3 equals 4; wide frames (the strict loop nest) follow the
    option too; an unknown value is refused."""
    h = mpe.Handle()
    try:
        for config, n in (("C2", 64), ("C3", 3), ("C1", 16)):
            d = synth.make_frames(config, n, seed=611)
            dets = [orc.find_leds(f, orc.make_params(), d["K"], d["D"])[0] for f in d["frames"]]
            dets = [x for x in dets if len(x) >= 4]
            ref = [orc.vote_histogram(x, d["markers"], d["K"], 5.0) for x in dets]
            got = {}
            for a in (0, 1, 3, 4):
                h.set_option("vote_arith", a)
                assert h.get_option("vote_arith") == a
                got[a] = h.vote_batch(dets, d["markers"], d["K"], 5.0)
            for i in range(len(dets)):
                assert np.array_equal(got[3][i], got[4][i]), (config, i)
                for a in (0, 1, 3, 4):
                    assert np.array_equal(got[a][i], ref[i]), (config, a, i)
        K, _ = synth.camera_for(480, 752)
        rng = np.random.default_rng(12)
        wide = np.column_stack([rng.uniform(150, 600, 35), rng.uniform(100, 380, 35)])
        ref = orc.vote_histogram(wide, synth.M5, K, 5.0)
        for a in (3, 4):
            h.set_option("vote_arith", a)
            assert np.array_equal(h.vote_batch([wide], synth.M5, K, 5.0)[0], ref), a
        h.set_option("vote_arith", 3)
        d = synth.make_frames("C2", 12, seed=612)
        ro = orc.estimate_batch(d["frames"], d["markers"], d["K"], d["D"], orc.make_params(), n_threads=4)
        rh = h.estimate_batch(d["frames"], d["markers"], d["K"], d["D"], mpe.demo_params())
        assert np.array_equal(rh["status"], ro["status"]) and np.array_equal(rh["n_corr"], ro["n_corr"])
        with pytest.raises(mpe.MpeError):
            h.set_option("vote_arith", 5)
    finally:
        h.close()


@pytest.mark.gpu
def test_occupancy_grid_variant_of_the_scan_carrying_kernel(orc):
    """Round 6: with 9 or more detections expected per frame (option "detections_hint", or the count the launcher of a
    stage-level call knows) the <= 5-marker voting kernel decides per root with a 128 x 128-bit occupancy grid of the
    detections whether any unused detection can be near a back-projection, and queues the few roots that can.  Same
    histograms as the strict kernel and the oracle on cluttered frames (9 .. 32 detections); the hint never changes a
    record (the pipelined path with and without it, byte for byte)."""
    import torch
    h = mpe.Handle()
    try:
        K, _ = synth.camera_for(480, 752)
        rng = np.random.default_rng(77)
        dets = []
        for kind, n in (("d4", 6), ("d16", 6)):
            d = synth.make_clutter_frames(kind, n, seed=91)
            dets += [orc.find_leds(f, orc.make_params(), d["K"], d["D"])[0] for f in d["frames"]]
        dets += [np.column_stack([rng.uniform(100, 650, k), rng.uniform(60, 420, k)]) for k in (9, 12, 17, 25, 32)]
        assert min(len(x) for x in dets) >= 9
        h.set_option("vote_arith", 4)
        strict = h.vote_batch(dets, synth.M5, K, 5.0)
        h.set_option("vote_arith", 3)
        got = h.vote_batch(dets, synth.M5, K, 5.0)   # (hint = the largest count of the call: the grid variant)
        for i, x in enumerate(dets):
            assert np.array_equal(got[i], strict[i]), (i, len(x), np.argwhere(got[i] != strict[i])[:5])
            if len(x) <= 21:
                assert np.array_equal(got[i], orc.vote_histogram(x, synth.M5, K, 5.0)), (i, len(x))
        # M4 (one unused marker) through the same variant
        got4 = h.vote_batch(dets[:6], synth.CONFIGS["C1"]["markers"], K, 5.0)
        for i in range(6):
            assert np.array_equal(got4[i], orc.vote_histogram(dets[i], synth.CONFIGS["C1"]["markers"], K, 5.0)), i
        # the pipelined path: records with the hint = records without
        d = synth.make_clutter_frames("d16", 64, seed=92)
        big = torch.from_numpy(d["frames"]).cuda().repeat(260, 1, 1)[:16384 + 32].contiguous()
        h.set_option("detections_hint", 5)    # (an explicit hint below 9: the per-detection prefilter of round 5)
        ra = h.estimate_batch(big, d["markers"], d["K"], d["D"], mpe.demo_params())
        assert h.get_option("detections_seen") == 21   # (what an automatic hint would use after this call)
        h.set_option("detections_hint", 21)
        assert h.get_option("detections_hint") == 21
        rb = h.estimate_batch(big, d["markers"], d["K"], d["D"], mpe.demo_params())
        assert np.array_equal(ra.view(np.uint8), rb.view(np.uint8))
        ro = orc.estimate_batch(d["frames"][:16], d["markers"], d["K"], d["D"], orc.make_params(), n_threads=4)
        assert np.array_equal(rb["status"][:16], ro["status"]) and np.array_equal(rb["n_corr"][:16], ro["n_corr"])
    finally:
        h.close()


@pytest.mark.gpu
def test_a_full_suspect_list_costs_time_not_poses(orc):
    """a suspect list that overflows used to LOSE entries and reject the frame
    (MPE_FRAME_VOTE_LIST_FULL).  Now the frames that lost an entry are voted again, whole, by the strict loop nest
    behind the fix-up kernel (k2_vote_relost) and come out as ordinary frames.  With the list capped at a handful of
    entries (option "vote_list_cap") the default arithmetic must still produce the strict histograms — plain kernel at
    5 and 8 markers, a cluttered single frame (8 markers, 28 detections: ~1.1 M hypotheses), the fused scan-carrying
    kernel over two sub-batches — with overflow events counted, frames re-voted, and no -13 status anywhere."""
    import torch
    h = mpe.Handle()
    try:
        K, _ = synth.camera_for(480, 752)
        rng = np.random.default_rng(5)
        cases = []
        d2 = synth.make_frames("C2", 128, seed=515)
        dets = [orc.find_leds(f, orc.make_params(), d2["K"], d2["D"])[0] for f in d2["frames"]]
        cases.append(("C2", [x for x in dets if len(x) >= 4], d2["markers"], d2["K"], 5.0))
        d3 = synth.make_frames("C3", 4, seed=516)
        dets = [orc.find_leds(f, orc.make_params(), d3["K"], d3["D"])[0] for f in d3["frames"]]
        cases.append(("C3", [x for x in dets if len(x) >= 4], d3["markers"], d3["K"], 5.0))
        clutter = np.column_stack([rng.uniform(200, 550, 28), rng.uniform(120, 360, 28)])
        cases.append(("cluttered frame", [clutter], synth.CONFIGS["C3"]["markers"], K, 5.0))
        for name, dets, markers, Kc, tol in cases:
            h.set_option("vote_list_cap", 0)
            h.set_option("vote_arith", 0)
            strict = h.vote_batch(dets, markers, Kc, tol)
            h.set_option("vote_arith", 1)
            ov_before = h.get_option("vote_fixup_overflow")
            roomy = h.vote_batch(dets, markers, Kc, tol)
            ov0, re0 = h.get_option("vote_fixup_overflow"), h.get_option("vote_relost_frames")
            assert ov0 == ov_before, name      # (the default list holds every suspect of these launches)
            h.set_option("vote_list_cap", 4)
            tight = h.vote_batch(dets, markers, Kc, tol)
            assert h.get_option("vote_list_cap") == 4
            assert h.get_option("vote_fixup_overflow") > ov0, name
            assert h.get_option("vote_relost_frames") > re0, name
            for i in range(len(dets)):
                assert np.array_equal(roomy[i], strict[i]), (name, i)
                assert np.array_equal(tight[i], strict[i]), (name, i, np.argwhere(tight[i] != strict[i])[:5])
        # the cluttered frame end to end on a single-frame call: a pose-or-not verdict, never a capacity status
        h.set_option("vote_list_cap", 0)
        r = h.solve_bruteforce(clutter, synth.CONFIGS["C3"]["markers"], K, mpe.demo_params())
        assert r["status"] in (0, 1), r["status"]
        # fused kernel + pipelined schedule + streaming shape: records byte-identical to the roomy list's
        d = synth.make_frames("C2", 96, seed=78)
        big = torch.from_numpy(d["frames"]).cuda().repeat(342, 1, 1)[:32768 + 64].contiguous()
        ra = h.estimate_batch(big, d["markers"], d["K"], d["D"], mpe.demo_params())
        h.set_option("vote_list_cap", 8)
        ov0, re0 = h.get_option("vote_fixup_overflow"), h.get_option("vote_relost_frames")
        rb = h.estimate_batch(big, d["markers"], d["K"], d["D"], mpe.demo_params())
        assert h.get_option("vote_fixup_overflow") > ov0 and h.get_option("vote_relost_frames") > re0
        assert not np.any(rb["status"] == -13)
        for k in ("status", "n_corr", "n_det"):
            assert np.array_equal(ra[k], rb[k]), k
        assert np.array_equal(ra["T"], rb["T"], equal_nan=True) and np.array_equal(ra["cov"], rb["cov"], equal_nan=True)
    finally:
        h.close()


@pytest.mark.gpu
def test_c3_poses_at_rounding_level(hip, orc):
    """8 markers: 56 validation P3P per frame, summed in the reference's combination order by the tail kernel
    (16 per round, lanes added in order) -> the poses agree with the oracle far below the north_star tolerance."""
    d = synth.make_frames("C3", 6, seed=303)
    Po, Ph = orc.make_params(back_projection_pixel_tolerance=2.0), mpe.demo_params(back_projection_pixel_tolerance=2.0)
    n = 0
    for i in range(6):
        und, _ = orc.find_leds(d["frames"][i], Po, d["K"], d["D"])
        ro = orc.solve_bruteforce(und, d["markers"], d["K"], Po)
        rh = hip.solve_bruteforce(und, d["markers"], d["K"], Ph)
        assert rh["status"] == ro["status"] and np.array_equal(rh["corr"], ro["corr"])
        if ro["status"] == 0:
            dp, dr = pose_diff(rh["T"], ro["T"])
            assert dp <= 1e-9 and dr <= 1e-9, (i, dp, dr)
            ok, T0 = orc.check_correspondences(und, d["markers"], d["K"], Po, ro["corr"])
            okh, T0h = hip.check_correspondences(und, d["markers"], d["K"], Ph, ro["corr"])
            assert ok and okh
            assert np.abs(T0h - T0).max() <= 1e-12, (i, np.abs(T0h - T0).max())
            n += 1
    assert n >= 3


@pytest.mark.gpu
def test_explicit_correspondences_with_repeated_markers(hip, orc):
    """More correspondence rows than markers (a marker listed twice) is defined input for checkCorrespondences
    (pose_estimator.cpp:394-542 works on the rows as given): the tail kernel sizes its back-projection buffer
    for the row capacity, and the verdict equals the oracle's."""
    d = synth.make_frames("C2", 4, seed=77)
    Po, Ph = orc.make_params(), mpe.demo_params()
    for i in range(4):
        und, _ = orc.find_leds(d["frames"][i], Po, d["K"], d["D"])
        r = orc.solve_bruteforce(und, d["markers"], d["K"], Po)
        if r["n_corr"] < 5:
            continue
        corr = np.vstack([r["corr"], r["corr"][:3]])  # 8 rows for 5 markers
        ok, T0 = orc.check_correspondences(und, d["markers"], d["K"], Po, corr)
        okh, T0h = hip.check_correspondences(und, d["markers"], d["K"], Ph, corr)
        assert okh == bool(ok), i
        if ok:
            dp, dr = pose_diff(T0h, T0)
            assert dp <= 1e-9 and dr <= 1e-9, (i, dp, dr)


@pytest.mark.gpu
def test_multi_device_entry_from_one_process(hip, orc, tmp_path):
    """mpe_estimate_batch_multi / _multi_device: the batch sharded over several handles from one host process
    (on this 1-GPU box: several handles on GPU 0; on an 8-GPU node: one per GPU) gives exactly the records of a
    single-handle call, in frame order — through ctypes AND from a plain C host program."""
    import os
    import subprocess
    import torch
    d = synth.make_frames("C2", 37, seed=1234)
    P = mpe.demo_params()
    one = hip.estimate_batch(d["frames"], d["markers"], d["K"], d["D"], P)
    ref = orc.estimate_batch(d["frames"], d["markers"], d["K"], d["D"], orc.make_params(), n_threads=4)
    assert np.array_equal(one["status"], ref["status"])
    n_gpu = torch.cuda.device_count()
    for n_dev in (1, 2, 3):
        hs = [mpe.Handle(i % n_gpu) for i in range(n_dev)]
        try:
            got = mpe.estimate_batch_multi(hs, d["frames"], d["markers"], d["K"], d["D"], P)
            assert got.tobytes() == one.tobytes(), n_dev
            shards = []
            for i in range(n_dev):
                lo, hi = mpe.shard_bounds(len(d["frames"]), i, n_dev)
                shards.append(torch.from_numpy(d["frames"][lo:hi].copy()).to("cuda:%d" % (i % n_gpu)))
            got = mpe.estimate_batch_multi(hs, shards, d["markers"], d["K"], d["D"], P)
            assert got.tobytes() == one.tobytes(), n_dev
        finally:
            for h in hs:
                h.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_abi_multi")
    libdir = os.path.dirname(mpe.library_path())
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "c_abi_multi.c"), "-o", exe, "-L", libdir, "-lmpe_hip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    raw, mk = str(tmp_path / "f.raw"), str(tmp_path / "m.txt")
    d["frames"].tofile(raw)
    np.savetxt(mk, d["markers"], fmt="%.17g")
    for n_dev in (1, 2):
        out = subprocess.run([exe, raw, str(len(d["frames"])), "480", "752", str(n_dev), mk], capture_output=True, text=True)
        assert out.returncode == 0 and "multi ok" in out.stdout, (n_dev, out.returncode, out.stdout, out.stderr)


@pytest.mark.gpu
def test_bench_entry_on_the_gpu_box():
    """bench.py end to end at a small batch (one JSON line, n_gpus 1, roofline + parity present); `--gpus 2` on a box
    with fewer than 2 GPUs is a clean, loud error (never a silent single-rank run)."""
    import json
    import os
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--frames", "32768", "--steps", "3", "--warmup", "1",
                          "--cpu-sample", "256"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-1500:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["value"] > 1e6 and rec["roofline"]["frac"] > 0.3
    assert rec["cpu_baseline"]["kind"] == "port" and rec["parity"]["pose_mismatches_gt_1e-4m_or_1e-3rad"] == 0
    assert rec["parity"]["status_mismatches"] == 0 and rec["parity"]["mismatches_unexplained"] == 0
    assert "last timed step" in rec["parity"]["records"]        # (the records the timed submissions produced)
    assert "other_configs" not in rec                            # (an explicit --frames: the headline leg alone)
    if torch.cuda.device_count() < 2:
        few = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"],
                             capture_output=True, text=True, env=env, timeout=300)
        assert few.returncode != 0 and "GPU(s) are visible" in few.stderr


@pytest.mark.gpu
def test_bench_secondary_legs_on_the_gpu_box():
    """The legs the default bench line carries behind the headline — another config, a clutter variant, the tracked
    streams, one frame — at small sizes through the same functions: each must come with a roofline of its dominant
    kernel, a parity sample with nothing unexplained, and no fraction above 1."""
    import argparse
    import os
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    ctx = {"rank": 0, "local_rank": 0, "world": 1, "dev": torch.device("cuda", 0), "dist": None}
    base = dict(steps=2, warmup=1, cpu_sample=64, back_tol=None, clutter=None, no_cpu=False, pipeline=16, pipeline_mode=-1,
                vote_arith=3, vote_splits=0, scan_split_pct=-1, side_scan_blocks=-1, assume_side_streams=False, k1a_lds=-1,
                opt=None, records_to_host=True, no_streaming=False, vote_events=True, false_hint_leg=False,
                no_host_leg=True, detections_hint=-1, consumer_priority=None)
    for kw in (dict(config="C1", frames=32768 + 512), dict(config="C3", frames=1024, back_tol=2.0),
               dict(config="C4", frames=1024), dict(config="C2", frames=2048, clutter="salt"),
               dict(config="C2", frames=2048, clutter="patch"), dict(config="C2", frames=1024, clutter="d16")):
        a = argparse.Namespace(**dict(base, **kw))
        out, parity_failed, impossible = bench.run_config(a, ctx, light=True)
        c = bench.compact(out)
        assert not parity_failed and not impossible, (kw, c)
        assert c["parity"]["frames"] == 64 and c["parity"]["mismatches_unexplained"] == 0, (kw, c["parity"])
        assert c["roofline"]["kernel"] and c["value"] > 0, (kw, c)
        assert (c["roofline"].get("frac") or 0) <= 1.0
    tr, bad = bench.tracked_legs(0, n_frames=60)
    assert not bad, tr
    assert tr["one_stream"]["parity"]["found_mismatches"] == 0 and tr["one_stream"]["parity"]["state_mismatches"] == 0
    assert tr["lockstep_8"]["fps"] > 0 and tr["lockstep_64"]["stream0_statuses_equal_the_solo_run"]
    lat = bench.one_frame_latency(0, reps=20)
    assert lat["pinned"]["pose_found"] and lat["pageable"]["median_ms"] > 0


@pytest.mark.gpu
def test_bench_collectives_on_a_one_rank_rccl_group():
    """What a 1-GPU box can execute of an N > 1 bench run: `--force-process-group` makes a ONE-rank nccl (= RCCL) group
    and sends the step through every collective of the multi-GPU path — the asynchronous, double-buffered dist.gather
    of the record buffers on the consumer stream, barrier, all_reduce of the step time, the rank report, and the parity
    sample computed from the GATHERED buffer (not from the rank's own)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-process-group", "--frames", "65536",
                        "--steps", "3", "--warmup", "1", "--no-host-leg", "--no-false-hint-leg", "--headline-only",
                        "--no-isolated"], capture_output=True, text=True, timeout=400, env=env, cwd=root)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert line["ranks_seen"] == [0] and len(line["per_rank_fps"]) == 1 and line["per_rank_fps"][0] > 0
    assert "forced_process_group" in line
    sp = line["shard_parity"]
    assert len(sp) == 1 and sp[0]["rank"] == 0 and sp[0]["frames"] > 0
    assert sp[0]["status_mismatches"] == 0 and sp[0]["mismatches_unexplained"] == 0, sp


@pytest.mark.gpu
def test_bench_head_of_shard_is_reproducible():
    """What makes an N > 1 bench run self-checking: rank 0 re-creates the first frames of EVERY rank's shard from that
    rank's seeds (bench.head_of_shard) and checks the records that arrived through the gather against the oracle on
    them.  The re-creation must be bit-identical to what the rank itself rendered (bench.make_batch) — for any rank
    index, batch size and clutter variant — or the check would compare records with the wrong frames."""
    import os
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    dev = torch.device("cuda", 0)
    for config, clutter, B, rank in (("C2", None, 1000, 0), ("C2", None, 300, 3), ("C1", None, 4096, 1),
                                     ("C2", "d4", 512, 2), ("C4", None, 260, 5)):
        _, frames = bench.make_batch(synth, config, clutter, B, dev, rank)
        head = bench.head_of_shard(synth, config, clutter, bench.SAMPLE_PER_RANK, dev, rank)
        assert head.shape[0] == bench.SAMPLE_PER_RANK
        assert torch.equal(head, frames[:bench.SAMPLE_PER_RANK]), (config, clutter, B, rank)
        other = bench.head_of_shard(synth, config, clutter, bench.SAMPLE_PER_RANK, dev, rank + 1)
        assert not torch.equal(other, head)      # (another rank's shard is another set of frames)


@pytest.mark.gpu
def test_tracking_on_noisy_frames_matches_oracle(orc):
    """The tracking path on cluttered frames: salt noise (three densities) over every frame of a stream, so that the ROI
    detections overflow the small blob tier and are repeated through the whole chain — large LDS tier and the general
    tier with PER-FRAME WINDOWS (each stream's ROI in its own slot, borders and centroid offsets following the window:
    the one configuration of the rewritten general tier the batch tests do not reach) — in lock step and one stream at a
    time.  Every frame must equal the oracle's state machine: updated flag, ROI, it_since_initialized, detection and
    correspondence counts, brute-force flag, pose."""
    n_streams, n = 4, 14
    seqs = [synth.make_sequence("C2", n, seed=160 + s) for s in range(n_streams)]
    for s, q in enumerate(seqs):
        rng = np.random.default_rng(500 + s)
        dens = (0.0005, 0.002, 0.0032, 0.001)[s]
        for k in range(n):
            m = rng.random(q["frames"][k].shape) < dens
            q["frames"][k][m] = 255
    h = mpe.Handle(0)
    P = mpe.demo_params()
    trackers = [mpe.Tracker(h, seqs[0]["markers"], seqs[0]["K"], seqs[0]["D"], P) for _ in range(n_streams)]
    rec, info = mpe.tracker_run_sequences_batch(trackers, [q["frames"] for q in seqs], seqs[0]["times"])
    solo = mpe.Tracker(h, seqs[0]["markers"], seqs[0]["K"], seqs[0]["D"], P)
    n_roi = n_pose = 0
    for s in range(n_streams):
        to = orc.Tracker(seqs[s]["markers"], seqs[s]["K"], seqs[s]["D"], orc.make_params())
        solo.reset()
        for k in range(n):
            ro = to.estimate(seqs[s]["frames"][k], seqs[s]["times"][k])
            assert rec["status"][s, k] >= 0, (s, k, rec["status"][s, k])
            assert (rec["status"][s, k] == 0) == ro["updated"], (s, k)
            assert tuple(info[s, k, 0:4]) == ro["roi"] and info[s, k, 4] == ro["it_since_initialized"], (s, k)
            assert info[s, k, 5] == ro["n_det"] and info[s, k, 6] == ro["n_corr"], (s, k, info[s, k], ro["n_det"], ro["n_corr"])
            assert bool(info[s, k, 7]) == ro["used_bruteforce"], (s, k)
            n_roi += int(info[s, k, 2] < seqs[s]["cols"])
            if ro["updated"]:
                n_pose += 1
                dp, dr = pose_diff(rec["T"][s, k].reshape(4, 4), ro["T"])
                assert dp <= POS_TOL_M and dr <= ROT_TOL_RAD, (s, k, dp, dr)
            r1 = solo.estimate(seqs[s]["frames"][k], seqs[s]["times"][k])
            assert r1["updated"] == ro["updated"] and r1["n_det"] == ro["n_det"] and r1["roi"] == ro["roi"], (s, k)
            if ro["updated"]:
                assert np.array_equal(r1["T"], rec["T"][s, k].reshape(4, 4)), (s, k)
    assert n_roi >= n_streams * (n - 6) // 2 and n_pose >= n_streams * n // 3, (n_roi, n_pose)
    for t in trackers + [solo]:
        t.close()
    h.close()


@pytest.mark.gpu
def test_lockstep_tracker_batch_matches_oracle(orc):
    """BASELINE configs[4] as ONE submission per time step: N trackers on one handle driven in lock step
    (mpe_tracker_estimate_batch / mpe_tracker_run_sequences_batch: one image scan + blob extraction over the N ROI
    slots, one validate / refine over the N detection sets, brute-force re-initialisations batched too).  Every
    stream must equal the oracle's state machine on its own frames — LED drop-outs (whole-image retry and
    re-initialisation on some streams while others keep tracking), ROIs of different sizes, and the per-frame Python
    entry point must give the same records as the C loop."""
    n_streams, n = 6, 26
    drop = {1: (9,), 3: (14, 15), 4: (5, 20)}
    seqs = [synth.make_sequence("C2", n, seed=60 + s, dropout=drop.get(s, ())) for s in range(n_streams)]
    h = mpe.Handle(0)
    P = mpe.demo_params()
    mk = lambda: [mpe.Tracker(h, seqs[0]["markers"], seqs[0]["K"], seqs[0]["D"], P) for _ in range(n_streams)]
    trackers = mk()
    rec, info = mpe.tracker_run_sequences_batch(trackers, [q["frames"] for q in seqs], seqs[0]["times"])
    n_brute = n_roi = 0
    for s in range(n_streams):
        to = orc.Tracker(seqs[s]["markers"], seqs[s]["K"], seqs[s]["D"], orc.make_params())
        for k in range(n):
            ro = to.estimate(seqs[s]["frames"][k], seqs[s]["times"][k])
            assert (rec["status"][s, k] == 0) == ro["updated"], (s, k)
            assert tuple(info[s, k, 0:4]) == ro["roi"] and info[s, k, 4] == ro["it_since_initialized"], (s, k)
            assert info[s, k, 5] == ro["n_det"] and info[s, k, 6] == ro["n_corr"], (s, k)
            assert bool(info[s, k, 7]) == ro["used_bruteforce"], (s, k)
            n_brute += int(info[s, k, 7])
            n_roi += int(info[s, k, 2] < seqs[s]["cols"])
            if ro["updated"]:
                dp, dr = pose_diff(rec["T"][s, k].reshape(4, 4), ro["T"])
                assert dp <= POS_TOL_M and dr <= ROT_TOL_RAD, (s, k, dp, dr)
    # every stream initialises by brute force, tracks in ROIs, and the drop-out frames force whole-image retries
    n_retry = int(((info[:, 1:, 2] == seqs[0]["cols"]) & (info[:, 1:, 4] >= 1)).sum())
    assert n_brute >= n_streams and n_roi >= n_streams * (n - 6) and n_retry >= 3, (n_brute, n_roi, n_retry)
    # per-step entry point == the C loop; and == one tracker per stream driven alone
    t2 = mk()
    solo = [mpe.Tracker(h, seqs[0]["markers"], seqs[0]["K"], seqs[0]["D"], P) for _ in range(n_streams)]
    for k in range(n):
        r2, i2, upd = mpe.tracker_estimate_batch(t2, [q["frames"][k] for q in seqs], [seqs[0]["times"][k]] * n_streams)
        assert r2.tobytes() == rec[:, k].tobytes() and np.array_equal(i2, info[:, k]), k
        for s in range(n_streams):
            r1 = solo[s].estimate(seqs[s]["frames"][k], seqs[s]["times"][k])
            assert r1["updated"] == bool(upd[s]) and np.array_equal(r1["T"], r2["T"][s].reshape(4, 4)), (s, k)
    # the same streams spread over three handles (pipelined groups): identical records
    hg = [mpe.Handle(0) for _ in range(3)]
    t3 = [mpe.Tracker(hg[s % 3], seqs[0]["markers"], seqs[0]["K"], seqs[0]["D"], P) for s in range(n_streams)]
    rec3, info3 = mpe.tracker_run_sequences_batch(t3, [q["frames"] for q in seqs], seqs[0]["times"])
    assert rec3.tobytes() == rec.tobytes() and np.array_equal(info3, info)
    # ... and the three groups on two / three host threads
    for threads in (2, 3):
        for t in t3:
            t.reset()
        rec4, info4 = mpe.tracker_run_sequences_batch(t3, [q["frames"] for q in seqs], seqs[0]["times"], threads)
        assert rec4.tobytes() == rec.tobytes() and np.array_equal(info4, info), threads
    for t in t3:
        t.close()
    for hh in hg:
        hh.close()
    # streams with different set-ups cannot share a lock-step batch
    other = mpe.Tracker(h, seqs[0]["markers"][:4], seqs[0]["K"], seqs[0]["D"], P)
    with pytest.raises(mpe.MpeError):
        mpe.tracker_estimate_batch([t2[0], other], [seqs[0]["frames"][0]] * 2, [0.0, 0.0])
    for t in trackers + t2 + solo + [other]:
        t.close()
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", witness_sequences())
def test_hip_tracker_against_the_witness_sequences(name):
    """The HIP tracker (mpe_tracker_*: host state machine + the device steps) against the tracking-path vectors of the
    independent witness (tests/golden/witness_seq_*.npz, numpy restatement of pose_estimator.cpp:62-147 etc.) — not
    against the oracle: ROI, it_since_initialized, counts, brute-force flag per frame, poses within the north_star
    tolerance, covariance rtol 1e-6."""
    g, d = load_sequence(name)
    h = mpe.Handle(0)
    tr = mpe.Tracker(h, d["markers"], d["K"], d["D"], mpe.demo_params())
    try:
        rec, info = tr.run_sequence(d["frames"], d["times"])
        for k in range(int(g["n"])):
            assert (rec["status"][k] == 0) == bool(g["updated"][k]), k
            assert tuple(info[k, 0:4]) == tuple(int(v) for v in g["roi"][k]), k
            assert (info[k, 4], info[k, 5], info[k, 6], int(info[k, 7])) == \
                   (g["it_since_initialized"][k], g["n_det"][k], g["n_corr"][k], g["used_bruteforce"][k]), k
            if g["updated"][k]:
                dp, dr = pose_diff(rec["T"][k].reshape(4, 4), g["T"][k])
                assert dp <= POS_TOL_M and dr <= ROT_TOL_RAD, (k, dp, dr)
                assert np.allclose(rec["cov"][k].reshape(6, 6), g["cov"][k], rtol=1e-6, atol=1e-12), k
    finally:
        tr.close()
        h.close()


@pytest.mark.gpu
def test_c5_eight_streams_three_ways(orc):
    """BASELINE configs[4] at its stated size — EIGHT independent 752x480 camera streams — on one GPU, in the three
    arrangements the host side offers, every stream against the oracle's estimateBodyPose state machine
    (pose_estimator.cpp:62-147) frame by frame, with LED drop-outs on five of the streams (whole-image retries,
    brute-force re-initialisations while the other streams keep tracking):
      (a) lock step on ONE handle: one device submission per time step for all eight;
      (b) eight handles, eight host threads, one tracker each (mpe_tracker_run_sequence) — all on the device at once;
      (c) two groups of four in lock step on two handles, each group on its own host thread
          (mpe_tracker_run_sequences_batch_threads).
    The three give byte-identical records and step information."""
    import threading
    n_streams, n = 8, 30
    drop = {0: (12,), 2: (7, 8), 3: (20,), 5: (15, 16, 17), 6: (25,)}
    seqs = [synth.make_sequence("C2", n, seed=500 + s, dropout=drop.get(s, ())) for s in range(n_streams)]
    frames = [q["frames"] for q in seqs]
    times = seqs[0]["times"]
    P = mpe.demo_params()
    mk, K, D = seqs[0]["markers"], seqs[0]["K"], seqs[0]["D"]
    # (a)
    h = mpe.Handle(0)
    ta = [mpe.Tracker(h, mk, K, D, P) for _ in range(n_streams)]
    rec, info = mpe.tracker_run_sequences_batch(ta, frames, times)
    n_brute = n_retry = n_pose = 0
    for s in range(n_streams):
        to = orc.Tracker(mk, K, D, orc.make_params())
        for k in range(n):
            ro = to.estimate(frames[s][k], times[k])
            assert (rec["status"][s, k] == 0) == ro["updated"], (s, k)
            assert tuple(info[s, k, 0:4]) == ro["roi"] and info[s, k, 4] == ro["it_since_initialized"], (s, k)
            assert info[s, k, 5] == ro["n_det"] and info[s, k, 6] == ro["n_corr"], (s, k)
            assert bool(info[s, k, 7]) == ro["used_bruteforce"], (s, k)
            n_brute += int(info[s, k, 7])
            if ro["updated"]:
                n_pose += 1
                dp, dr = pose_diff(rec["T"][s, k].reshape(4, 4), ro["T"])
                assert dp <= POS_TOL_M and dr <= ROT_TOL_RAD, (s, k, dp, dr)
    n_retry = int(((info[:, 1:, 2] == seqs[0]["cols"]) & (info[:, 1:, 4] >= 1)).sum())
    assert n_brute >= n_streams and n_retry >= 5 and n_pose >= n_streams * (n - 8), (n_brute, n_retry, n_pose)
    # (b)
    hb = [mpe.Handle(0) for _ in range(n_streams)]
    tb = [mpe.Tracker(hb[s], mk, K, D, P) for s in range(n_streams)]
    got = [None] * n_streams

    def work(s):
        got[s] = tb[s].run_sequence(frames[s], times)

    th = [threading.Thread(target=work, args=(s,)) for s in range(n_streams)]
    [t.start() for t in th]
    [t.join() for t in th]
    for s in range(n_streams):
        assert got[s] is not None, s
        assert got[s][0].tobytes() == rec[s].tobytes() and np.array_equal(got[s][1], info[s]), s
    # (c)
    hc = [mpe.Handle(0) for _ in range(2)]
    tc = [mpe.Tracker(hc[s // 4], mk, K, D, P) for s in range(n_streams)]
    rec_c, info_c = mpe.tracker_run_sequences_batch(tc, frames, times, 2)
    assert rec_c.tobytes() == rec.tobytes() and np.array_equal(info_c, info)
    for t in ta + tb + tc:
        t.close()
    for hh in [h] + hb + hc:
        hh.close()


@pytest.mark.gpu
def test_chunked_host_ingest_is_bit_identical(orc):
    """mpe_estimate_batch on HOST frames: the double-buffered chunked ingest (copy of chunk c + 1 beside the kernels
    of chunk c; option "ingest_chunk") gives byte-identical records to one blocking copy — from pageable memory and
    from pinned memory obtained through the ABI (mpe_alloc_pinned)."""
    d = synth.make_frames("C2", 37, seed=4321)
    P = mpe.demo_params()
    h = mpe.Handle(0)
    try:
        h.set_option("ingest_chunk", 0)
        one = h.estimate_batch(d["frames"], d["markers"], d["K"], d["D"], P)
        ref = orc.estimate_batch(d["frames"], d["markers"], d["K"], d["D"], orc.make_params(), n_threads=4)
        assert np.array_equal(one["status"], ref["status"])
        pin = mpe.PinnedFrames(*d["frames"].shape)
        pin.array[...] = d["frames"]
        for chunk in (8, 16, 36):
            h.set_option("ingest_chunk", chunk)
            assert h.get_option("ingest_chunk") == chunk
            assert h.estimate_batch(d["frames"], d["markers"], d["K"], d["D"], P).tobytes() == one.tobytes(), chunk
            assert h.estimate_batch(pin.array, d["markers"], d["K"], d["D"], P).tobytes() == one.tobytes(), chunk
        pin.close()
        with pytest.raises(mpe.MpeError):
            h.set_option("ingest_chunk", -1)
    finally:
        h.close()


@pytest.mark.gpu
def test_reference_class_surface_runs(tmp_path):
    """The literal reference surface (estimateBodyPose(cv::Mat, double), cv::Mat camera_matrix_K_, Eigen getters —
    compat/adapters/reference_surface.h) driven the way MPENode drives the class, on a tracked sequence: identical
    poses / covariances to the plain facade underneath, overlay through augmentImage(cv::Mat&)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "compat")])
    exe = str(tmp_path / "reference_surface_check")
    libdir = os.path.dirname(mpe.library_path())
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-DMPE_REFERENCE_SURFACE", "-I", os.path.join(root, "tests", "mock_deps"),
                           "-I", os.path.join(root, "compat"), "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "reference_surface_check.cpp"), "-o", exe, "-L",
                           os.path.join(root, "compat"), "-lmonocular_pose_estimator_compat", "-L", libdir, "-lmpe_hip",
                           "-Wl,-rpath," + os.path.join(root, "compat"), "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    d = synth.make_sequence("C2", 20, seed=99)
    raw = str(tmp_path / "seq.raw")
    d["frames"].tofile(raw)
    out = subprocess.run([exe, raw, "20", str(d["rows"]), str(d["cols"])], capture_output=True, text=True)
    assert out.returncode == 0 and "surface ok" in out.stdout, (out.returncode, out.stdout, out.stderr[-500:])


@pytest.mark.gpu
def test_overlay_geometry_against_an_independent_projection(tmp_path):
    """The debug overlay (Visualization::createVisualizationImage, visualization.cpp:58-99: body axes x red / y green
    / z blue of length 0.075 m projected WITH lens distortion, a radius-10 ring around every detection, the ROI
    rectangle, all 2 px thick) painted by the facade's rasteriser, checked against geometry computed independently in
    numpy from the same pose / ROI / detections: every ring is complete and every coloured pixel lies where the
    reference's primitives put ink (within the 2 px pen).  OpenCV is not available to compare pixel for pixel."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "compat")])
    d = synth.make_sequence("C2", 12, seed=77)
    raw, yaml, ov = str(tmp_path / "seq.raw"), str(tmp_path / "m.yaml"), str(tmp_path / "ov.bgr")
    d["frames"].tofile(raw)
    with open(yaml, "w") as fh:
        fh.write("marker_positions:\n")
        for m in d["markers"]:
            fh.write("  - x: %.17g\n    y: %.17g\n    z: %.17g\n" % tuple(m))
    out = subprocess.run([os.path.join(root, "compat", "facade_selftest"), "steps", "--markers", yaml, "--frames", raw,
                          "--rows", str(d["rows"]), "--cols", str(d["cols"]), "--dt", "0.02", "--overlay-out", ov],
                         capture_output=True, text=True)
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-500:])
    st = [ln for ln in out.stdout.splitlines() if ln.startswith("overlay_state")][0].split()[1:]
    T = np.array([float(x) for x in st[:16]]).reshape(4, 4)
    rx, ry, rw, rh, nc = [int(x) for x in st[16:21]]
    centres = np.array([float(x) for x in st[21:21 + 2 * nc]]).reshape(nc, 2)
    rows, cols = d["rows"], d["cols"]
    img = np.fromfile(ov, np.uint8).reshape(rows, cols, 3)  # B, G, R
    red = (img[..., 0] == 0) & (img[..., 1] == 0) & (img[..., 2] == 255)
    green = (img[..., 0] == 0) & (img[..., 1] == 255) & (img[..., 2] == 0)
    blue = (img[..., 0] == 255) & (img[..., 1] == 0) & (img[..., 2] == 0)
    # the selftest's camera (facade_selftest.cpp configure()): independent pinhole + plumb-bob projection
    K = np.array([[307.8119, 0, 371.6954], [0, 307.5514, 243.5497], [0, 0, 1.0]])
    D = np.array([-0.2819, 0.0675, 0.0004, -0.0003, -0.0063])
    tips = np.array([[0, 0, 0], [0.075, 0, 0], [0, 0.075, 0], [0, 0, 0.075]])
    pc = tips @ T[:3, :3].T + T[:3, 3]
    px = synth.distort_px(np.stack([K[0, 0] * pc[:, 0] / pc[:, 2] + K[0, 2], K[1, 1] * pc[:, 1] / pc[:, 2] + K[1, 2]], -1), K, D)

    def dist_to_segment(yy, xx, a, b):
        p = np.stack([xx, yy], -1).astype(float)
        ab = b - a
        t = np.clip(((p - a) @ ab) / max(ab @ ab, 1e-12), 0, 1)
        return np.linalg.norm(p - (a + t[:, None] * ab), axis=1)

    def near_any(mask, tests, tol):
        yy, xx = np.nonzero(mask)
        ok = np.zeros(len(yy), bool)
        for fn in tests:
            ok |= fn(yy, xx) <= tol
        return ok

    ring_d = [lambda yy, xx, c=c: np.abs(np.hypot(xx - np.rint(c[0]), yy - np.rint(c[1])) - 10.0) for c in centres]
    x0, y0, x1, y1 = rx, ry, rx + rw - 1, ry + rh - 1
    corners = [np.array(p, float) for p in ((x0, y0), (x1, y0), (x1, y1), (x0, y1))]
    rect_d = [lambda yy, xx, a=corners[i], b=corners[(i + 1) % 4]: dist_to_segment(yy, xx, a, b) for i in range(4)]
    axis_d = [lambda yy, xx, k=k: dist_to_segment(yy, xx, px[0], px[k]) for k in (1, 2, 3)]
    tol = 2.6  # 2 px pen (2x2 brush) + rounding of the end points
    assert near_any(red, ring_d + [axis_d[0]], tol).all()
    assert near_any(green, [axis_d[1]], tol).all() and green.sum() >= 4
    assert near_any(blue, rect_d + [axis_d[2]], tol).all()
    # completeness: every ring closed (sampled every 5 degrees), the rectangle outline fully inked, the axes inked
    inked = red | green | blue
    for c in centres:
        for ang in np.deg2rad(np.arange(0, 360, 5)):
            x, y = int(np.rint(np.rint(c[0]) + 10 * np.cos(ang))), int(np.rint(np.rint(c[1]) + 10 * np.sin(ang)))
            if 1 <= x < cols - 1 and 1 <= y < rows - 1:
                assert inked[y - 1:y + 2, x - 1:x + 2].any(), (c, ang)
    if rw < cols or rh < rows:  # (a whole-image ROI lies on the frame border)
        assert blue[y0, x0:x1 + 1].all() and blue[y1, x0:x1 + 1].all() and blue[y0:y1 + 1, x0].all() and blue[y0:y1 + 1, x1].all()
    for k, col in ((1, red), (2, green), (3, blue)):
        for t in np.linspace(0, 1, 20):
            q = px[0] + t * (px[k] - px[0])
            x, y = int(np.rint(q[0])), int(np.rint(q[1]))
            if 2 <= x < cols - 2 and 2 <= y < rows - 2:
                assert inked[y - 2:y + 3, x - 2:x + 3].any(), (k, t)


@pytest.mark.gpu
def test_general_tier_kernel_follows_what_the_last_call_saw(orc):
    """Round 6, option "general_lds" = -1: the general blob tier runs as k1b_general_lds — a CU's whole LDS per block —
    only once a call has SEEN frames reach that tier (a pinned mirror of the hand-over count, read a call late: option read-out "general_seen");
    records identical whichever kernel ran, in the pipelined path too."""
    import torch
    d = synth.make_clutter_frames("salt", 64, seed=17)
    big = torch.from_numpy(d["frames"]).cuda().repeat(258, 1, 1)[:16384 + 64].contiguous()
    h = mpe.Handle()
    try:
        P = mpe.demo_params()
        assert h.get_option("general_lds") == 0   # (the default: measured slower on salt noise, see mpe_k1.hip)
        h.set_option("general_lds", -1)           # the automatic choice
        assert h.get_option("general_seen") == 0
        r0 = h.estimate_batch(big, d["markers"], d["K"], d["D"], P)      # slabs (nothing seen yet)
        assert h.get_option("general_seen") > 0
        r1 = h.estimate_batch(big, d["markers"], d["K"], d["D"], P)      # LDS-resident now
        h.set_option("general_lds", 0)
        r2 = h.estimate_batch(big, d["markers"], d["K"], d["D"], P)
        assert np.array_equal(r0.view(np.uint8), r1.view(np.uint8)) and np.array_equal(r0.view(np.uint8), r2.view(np.uint8))
        ro = orc.estimate_batch(d["frames"][:16], d["markers"], d["K"], d["D"], orc.make_params(), n_threads=4)
        assert np.array_equal(r1["status"][:16], ro["status"]) and np.array_equal(r1["n_det"][:16], ro["n_det"])
        clean = synth.make_frames("C2", 64, seed=18)
        h.set_option("general_lds", -1)
        h.estimate_batch(torch.from_numpy(clean["frames"]).cuda(), clean["markers"], clean["K"], clean["D"], P)
        assert h.get_option("general_seen") == 0                         # ... and back once a call saw none
    finally:
        h.close()


@pytest.mark.gpu
def test_general_tier_block_walks_many_frames(orc):
    """A block of k1b_general strides the work-list by the grid and reuses its slab (bitmaps, to-do bits, item list) frame
    after frame.  With the slabs capped at 32 (option "k1b_general_blocks", process-wide) 330 DIFFERENT noisy frames make
    every block take ten or eleven of them one after the other, sparse after dense and the other way round; every frame's
    record equals the oracle's — nothing of a frame survives in the slab or in a cache line of it."""
    rng = np.random.default_rng(4242)
    rows, cols = 480, 752
    K, D = synth.camera_for(rows, cols)
    leds = synth.make_frames("C2", 30, seed=4243)["frames"]
    frames = []
    for i in range(330):
        dens = 10.0 ** rng.uniform(-4.0, -2.6)
        f = (rng.random((rows, cols)) < dens).astype(np.uint8) * rng.integers(150, 256, (rows, cols)).astype(np.uint8)
        if i % 11 == 0:
            f = np.maximum(f, leds[i // 11])
        if i % 7 == 3:
            f[int(rng.integers(0, rows - 70)):, int(rng.integers(0, cols - 8)):][:60, :6] = 230  # a tall bar
        frames.append(f)
    frames = np.ascontiguousarray(np.stack(frames))
    Po, Ph = orc.make_params(), mpe.demo_params()
    h0 = mpe.Handle()
    old = h0.get_option("k1b_general_blocks")
    h0.set_option("k1b_general_blocks", 32)
    try:
        h = mpe.Handle()
        try:
            got = h.detect_batch(frames, K, D, Ph)
            again = h.detect_batch(frames[::-1].copy(), K, D, Ph)
        finally:
            h.close()
    finally:
        h0.set_option("k1b_general_blocks", old)
        h0.close()
    again = again[::-1]   # whatever order the blocks met them in (entries behind a record's n are unspecified)
    assert np.array_equal(got["n"], again["n"]) and np.array_equal(got["status"], again["status"])
    for i in range(len(frames)):
        k = 2 * int(got["n"][i])
        assert np.array_equal(got["dist_xy"][i][:k], again["dist_xy"][i][:k]), i
        assert np.array_equal(got["undist_xy"][i][:k], again["undist_xy"][i][:k]), i
    n_gen = 0
    for i in range(len(frames)):
        und, dist = orc.find_leds(frames[i], Po, K, D, cap=65536)
        n = len(und)
        if n > mpe.MAX_DETECTIONS:
            assert got["status"][i] == -10 and got["n"][i] == mpe.MAX_DETECTIONS, i
            und, dist, n = und[:mpe.MAX_DETECTIONS], dist[:mpe.MAX_DETECTIONS], mpe.MAX_DETECTIONS
        else:
            assert got["status"][i] == 0 and got["n"][i] == n, (i, got["n"][i], n)
        assert np.array_equal(got["dist_xy"][i][:2 * n].reshape(-1, 2), dist), i
        assert np.array_equal(got["undist_xy"][i][:2 * n].reshape(-1, 2), und), i
        n_gen += 1
    assert n_gen == 330


@pytest.mark.gpu
def test_facade_static_primitives_with_a_device():
    """The same calls as tests/test_abi_cpu.py::test_facade_static_primitives_never_throw, with a HIP device: they succeed
    (0 / 0, an empty image yields no detections) and leave no error behind."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "compat"), "facade_selftest"])
    out = subprocess.run([os.path.join(root, "compat", "facade_selftest"), "nothrow"], capture_output=True, text=True)
    assert out.returncode == 0 and "computePoses 0 solveQuartic 0 centers 0 last_error []" in out.stdout, (out.stdout, out.stderr)
