"""The CPU oracle against the committed golden vectors (tests/golden/*.npz, made by
tests/golden/make_golden.py): guards the oracle itself against regressions.  CPU only."""
import numpy as np
import pytest

from golden_util import golden_cases, golden_sequences, load, load_sequence
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
