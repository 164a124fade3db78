"""The CPU oracle against the committed golden vectors (tests/golden/*.npz, made by
tests/golden/make_golden.py): guards the oracle itself against regressions.  CPU only."""
import numpy as np
import pytest

from golden_util import golden_cases, golden_sequences, load, load_clutter, load_sequence, witness_sequences
from util import pose_diff


@pytest.mark.parametrize("name", golden_cases())
def test_oracle_reproduces_golden(orc, name):
    g, d = load(name)
    P = orc.make_params(back_projection_pixel_tolerance=float(g["tol"]))
    n_m = len(d["markers"])
    for i in range(int(g["n"])):
        und, dist = orc.find_leds(d["frames"][i], P, d["K"], d["D"])
        k = int(g["n_det"][i])
        assert len(und) == k
        assert np.array_equal(dist, g["dist_xy"][i, :k]) and np.array_equal(und, g["undist_xy"][i, :k])
        r = orc.solve_bruteforce(und, d["markers"], d["K"], P)
        assert np.array_equal(r["hist"], g["hist"][i, :k, :n_m])
        assert r["status"] == g["status"][i] and r["n_corr"] == g["n_corr"][i]
        assert np.array_equal(r["corr"], g["corr"][i, :r["n_corr"]])
        if r["status"] == 0:
            dp, dr = pose_diff(r["T"], g["T"][i])
            assert dp < 1e-9 and dr < 1e-9
            assert np.allclose(r["cov"], g["cov"][i], rtol=1e-6, atol=1e-14)


@pytest.mark.parametrize("name", golden_sequences())
def test_oracle_tracker_reproduces_golden_sequence(orc, name):
    g, d = load_sequence(name)
    tr = orc.Tracker(d["markers"], d["K"], d["D"], orc.make_params())
    for k in range(int(g["n"])):
        r = tr.estimate(d["frames"][k], d["times"][k])
        assert r["updated"] == bool(g["updated"][k]) and r["roi"] == tuple(g["roi"][k]), k
        assert (r["it_since_initialized"], r["n_det"], r["n_corr"], int(r["used_bruteforce"])) == \
               (g["it"][k], g["n_det"][k], g["n_corr"][k], g["bruteforce"][k]), k
        if r["updated"]:
            dp, dr = pose_diff(r["T"], g["T"][k])
            assert dp < 1e-9 and dr < 1e-9


@pytest.mark.parametrize("name", witness_sequences())
def test_oracle_tracker_against_the_witness_sequences(orc, name):
    """The tracking path of the oracle (estimateBodyPose as a state machine) against vectors made by the INDEPENDENT
    witness (tests/witness_pipeline.py::Tracker, written from the reference sources, numpy): prediction, ROI rectangle,
    nearest-neighbour correspondences, whole-image retry on drop-outs, brute-force flag, counts and poses per frame."""
    g, d = load_sequence(name)
    tr = orc.Tracker(d["markers"], d["K"], d["D"], orc.make_params())
    n_roi = 0
    for k in range(int(g["n"])):
        r = tr.estimate(d["frames"][k], d["times"][k])
        assert r["updated"] == bool(g["updated"][k]) and r["roi"] == tuple(int(v) for v in g["roi"][k]), k
        assert (r["it_since_initialized"], r["n_det"], r["n_corr"], int(r["used_bruteforce"])) == \
               (g["it_since_initialized"][k], g["n_det"][k], g["n_corr"][k], g["used_bruteforce"][k]), k
        n_roi += int(r["roi"][2] < d["cols"])
        if r["updated"]:
            dp, dr = pose_diff(r["T"], g["T"][k])
            assert dp < 1e-9 and dr < 1e-9, (k, dp, dr)
            assert np.allclose(r["cov"], g["cov"][k], rtol=1e-6, atol=1e-14), k
    # (a sequence is mostly ROI frames — except the dense-salt one, whose 12 - 28 detections per frame never validate:
    #  every frame is a whole-image brute-force attempt, in the witness and in the oracle alike)
    assert n_roi >= int(g["n"]) // 2 or int(g["updated"].sum()) == 0


def test_witness_tracker_primitives_against_the_oracle(orc):
    """logarithmMap / determineROI / distortPoints of the witness (numpy, from pose_estimator.cpp:996-1064 and
    led_detector.cpp:114-224) against the oracle's on random inputs incl. the special cases (identity rotation, zero
    translation, predictions outside the image, NaN)."""
    import witness_pipeline as W
    from rpg_monocular_pose_estimator_amd import synth
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(3)
    K, D = synth.camera_for(480, 752)
    for it in range(200):
        T = np.eye(4)
        if it % 7:
            T[:3, :3] = Rotation.from_rotvec(rng.normal(size=3) * rng.uniform(1e-9, 2.5)).as_matrix()
        if it % 5:
            T[:3, 3] = rng.normal(size=3)
        assert np.allclose(W.logarithm_map(T), orc.logarithm_map(T), rtol=1e-12, atol=1e-15), it
        px = rng.uniform(-200, 1000, (5, 2))
        if it % 11 == 0:
            px[0, 0] = np.nan
        assert tuple(W.determine_roi(px, 480, 752, 20, K, D)) == tuple(orc.determine_roi(px, 480, 752, 20, K, D)), it


def test_oracle_against_the_witness_on_cluttered_frames(orc):
    """findLeds on what a real camera adds — salt noise (sparse and dense), a saturated patch, a ring that encloses the
    LEDs (RETR_EXTERNAL), a dot grid, distractor spots — at two thresholds: the oracle's detections (count, order,
    float32 centroids, undistorted points) equal the INDEPENDENT witness's (tests/golden/witness_clutter.npz: scipy
    connected components + Moore tracing, tests/witness_pipeline.py::find_leds).  The same vectors pin the HIP path's
    general blob tier on the GPU (test_hip_against_the_witness_on_cluttered_frames)."""
    cases = load_clutter()
    assert len(cases) == 42 and {c[0] for c in cases} == {"salt", "salt_dense", "patch", "ring", "grid", "d4", "d16"}
    c4 = load_clutter("C4")   # the same kinds at 1920x1200
    assert len(c4) == 10 and all(c[2].shape == (1200, 1920) for c in c4)
    for kind, thr, frame, K, D, k, dist, und in cases + c4:
        P = orc.make_params(threshold_value=thr)
        u, ds = orc.find_leds(frame, P, K, D)
        assert len(u) == k, (kind, thr, len(u), k)
        assert np.array_equal(ds, dist) and np.array_equal(u, und), (kind, thr)
