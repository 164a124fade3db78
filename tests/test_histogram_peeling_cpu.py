"""correspondencesFromHistogram (pose_estimator.cpp:344-370) scans the whole vote histogram once per marker for the first
position of the maximum, column-major, and then zeroes that COLUMN.  The device kernel (k3a_validate) does not repeat
the scans: since only columns are ever removed, a column's maximum and the first row that reaches it never change, so
every lane finds them for one column and the rounds are replayed on those n_m pairs (a removed column stands at value 0
with row 0, which is what the reference's scan sees — it matters when the threshold is 0).  This test holds that FORM,
restated in Python exactly as the kernel has it, against the oracle's literal restatement of the reference on random
histograms: sparse, dense, tied maxima, all-zero columns and rows, thresholds 0 / 1 / mid / above everything."""
import numpy as np
import pytest

import oracle

oracle.build()
from oracle import binding as orc  # noqa: E402


def column_maxima_form(hist, threshold):
    """The kernel's form (mpe_k3.hip, k3a_validate, histogram path): hist rows = detections, columns = markers."""
    n_d, n_m = hist.shape
    colmax = np.zeros(n_m, np.uint64)
    colrow = np.zeros(n_m, np.int64)
    for c in range(n_m):               # one lane per column
        mv, mr = 0, 0
        for r in range(n_d):
            v = int(hist[r, c])
            if v > mv:                 # first row of the column's maximum; an all-zero column keeps row 0
                mv, mr = v, r
        colmax[c], colrow[c] = mv, mr
    if not colmax.any():               # initialise()'s all-zero test
        return np.zeros((0, 2), np.uint32)
    removed = 0
    out = []
    for _ in range(n_m):
        mv, ri, ci, first = 0, 0, 0, True
        for c in range(n_m):
            gone = (removed >> c) & 1
            v = 0 if gone else int(colmax[c])
            if first or v > mv:
                mv, ri, ci, first = v, (0 if gone else int(colrow[c])), c, False
        if mv < threshold:
            break
        out.append((ci + 1, ri + 1))
        removed |= 1 << ci
    return np.array(out, np.uint32).reshape(-1, 2)


@pytest.mark.parametrize("seed", range(6))
def test_column_maxima_form_equals_the_reference_scan(seed):
    rng = np.random.default_rng(seed)
    n_cases = 0
    for _ in range(400):
        n_d, n_m = int(rng.integers(1, 13)), int(rng.integers(1, 9))
        kind = rng.integers(0, 5)
        if kind == 0:
            h = rng.integers(0, 4, (n_d, n_m))                         # many ties and zeros
        elif kind == 1:
            h = rng.integers(0, 500, (n_d, n_m))
        elif kind == 2:
            h = np.where(rng.random((n_d, n_m)) < 0.15, rng.integers(1, 60, (n_d, n_m)), 0)   # sparse
        elif kind == 3:
            h = rng.integers(0, 50, (n_d, n_m))
            h[:, rng.integers(0, n_m)] = 0                             # an all-zero column
            h[rng.integers(0, n_d), :] = 0                             # an all-zero row
        else:
            h = np.zeros((n_d, n_m), np.int64)
            if rng.random() < 0.7:
                h[rng.integers(0, n_d), rng.integers(0, n_m)] = int(rng.integers(1, 9))
        h = h.astype(np.uint32)
        for thr in (0, 1, int(h.max() // 2 + 1), int(h.max()) + 1):
            ref = np.asarray(orc.correspondences_from_histogram(h, thr), np.uint32).reshape(-1, 2)
            if not h.any():
                # (the reference never gets here with an all-zero histogram: initialise() returns first,
                #  pose_estimator.cpp:704; the kernel applies that test in front of the peeling)
                continue
            got = column_maxima_form(h, thr)
            assert np.array_equal(got, ref), (h, thr, got, ref)
            n_cases += 1
    assert n_cases > 1000
