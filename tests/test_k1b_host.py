"""The DEVICE source of the blob extraction (threshold, fixed-point blur, Suzuki border following, polygon sums,
shape filter, undistortion: everything of K1b that is not wave plumbing), compiled for the HOST and checked against
the oracle's findLeds — the CPU tier has no GPU, but it can still run the very code the GPU runs.

The region is cut out of the kernel sources (rpg_monocular_pose_estimator_amd/csrc/mpe_k1.hip; binding.device_source) at test time (tests/host/k1b_host.cpp
holds the one-lane shims and the whole-frame flow of the device's general tier).  Reference being matched:
led_detector.cpp:35-112 through oracle.find_leds."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from rpg_monocular_pose_estimator_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rpg_monocular_pose_estimator_amd", "csrc")


def _cut(text, begin, end):
    i = text.index(begin)
    return text[i:text.index(end, i)]


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    d = tmp_path_factory.mktemp("k1b_host")
    import rpg_monocular_pose_estimator_amd as mpe
    hip = mpe.device_source()
    internal = open(os.path.join(CSRC, "mpe_internal.h")).read()
    with open(os.path.join(d, "k1a_extract.inc"), "w") as fh:
        fh.write(_cut(hip, "struct ThrTest {", "#ifndef K1A_UNROLL"))
    inc = _cut(internal, "struct DetectParams {", "struct SolveParams {")
    inc += _cut(hip, "struct BlobRec {", "// final stage: kept blobs")
    with open(os.path.join(d, "k1b_extract.inc"), "w") as fh:
        fh.write(inc)
    so = os.path.join(d, "libk1b_host.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", *os.environ.get("MPE_HOST_CXXFLAGS", "").split(), "-I", str(d),
                           os.path.join(ROOT, "tests", "host", "k1b_host.cpp"), "-o", so])
    lib = C.CDLL(so)
    lib.host_find_leds.restype = C.c_int
    return lib


def _host_find_leds(lib, orc, img, P, K, D, roi_xy=(0, 0)):
    taps = np.ascontiguousarray(orc.gaussian_kernel_q8(P.gaussian_sigma), np.int32)
    shape = np.array([P.min_blob_area, P.max_blob_area, P.max_width_height_distortion, P.max_circular_distortion])
    Kf = np.ascontiguousarray(np.asarray(K, float).reshape(9))
    Df = np.ascontiguousarray(np.asarray(D, float).reshape(-1))
    cap = 4096
    dist = np.zeros((cap, 2), np.float32)
    und = np.zeros((cap, 2))
    img = np.ascontiguousarray(img, np.uint8)
    n = lib.host_find_leds(img.ctypes.data_as(C.c_void_p), img.shape[0], img.shape[1], int(P.threshold_value),
                           taps.ctypes.data_as(C.c_void_p), len(taps), shape.ctypes.data_as(C.c_void_p),
                           Kf.ctypes.data_as(C.c_void_p), Df.ctypes.data_as(C.c_void_p), len(Df), roi_xy[0], roi_xy[1],
                           dist.ctypes.data_as(C.c_void_p), und.ctypes.data_as(C.c_void_p), cap)
    assert n >= 0
    return und[:n], dist[:n]


@pytest.fixture(scope="module")
def orc():
    import oracle
    oracle.build()
    from oracle import binding
    return binding


def _random_image(rng, rows, cols, kind):
    img = rng.integers(0, 60, (rows, cols)).astype(np.uint8)  # below the threshold
    if kind == "spots":  # LED-like Gaussian spots, some on the border (mirrored blur taps)
        yy, xx = np.mgrid[0:rows, 0:cols]
        for _ in range(rng.integers(1, 9)):
            cx, cy = rng.uniform(-2, cols + 2), rng.uniform(-2, rows + 2)
            s = rng.uniform(0.8, 4.0)
            img = np.maximum(img, (255 * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))).astype(np.uint8))
    elif kind == "noise":  # salt: single pixels, thin lines, touching components, holes after the blur
        m = rng.random((rows, cols)) < rng.uniform(0.01, 0.2)
        img[m] = rng.integers(100, 256, m.sum())
    else:  # rings and bars: nested components and holes (RETR_EXTERNAL must skip what lies inside)
        yy, xx = np.mgrid[0:rows, 0:cols]
        for _ in range(rng.integers(1, 5)):
            cx, cy, r = rng.uniform(0, cols), rng.uniform(0, rows), rng.uniform(5, 30)
            d = np.hypot(xx - cx, yy - cy)
            for rr in np.arange(r, 4, -rng.uniform(6, 9)):  # concentric rings ...
                img[(d > rr - 1.5) & (d < rr + 1.0)] = 255
            if rng.random() < 0.7:
                img[(d < 1.8)] = 230                         # ... around a disc
        if rng.random() < 0.5:
            y0 = rng.integers(0, rows)
            img[y0:y0 + 1, rng.integers(0, cols // 2):] = 200
    return img


@pytest.mark.parametrize("kind", ["spots", "noise", "rings"])
def test_device_blob_source_on_the_host_equals_the_oracle(host, orc, kind):
    rng = np.random.default_rng({"spots": 1, "noise": 2, "rings": 3}[kind])
    n_blobs = 0
    for it in range(400):
        rows, cols = int(rng.integers(8, 90)), int(rng.integers(18, 150))
        img = _random_image(rng, rows, cols, kind)
        K, D = synth.camera_for(rows, cols)
        if it % 3 == 0:  # wide-open shape filter: every traced contour with a non-degenerate box is reported
            P = orc.make_params(min_blob_area=0.0, max_blob_area=1e9, max_width_height_distortion=1e9,
                                max_circular_distortion=1e9, gaussian_sigma=[0.6, 0.5, 0.85, 0.2][it % 12 // 3])  # 5, 3, 7 taps; [0, 256, 0]
        else:
            P = orc.make_params(gaussian_sigma=0.6 if it % 2 else 0.4)
        roi_xy = (int(rng.integers(0, 50)), int(rng.integers(0, 50))) if it % 4 == 1 else (0, 0)
        # (the oracle adds the ROI origin of the window it is given; here the window is the whole small image and the
        #  origin is only the float offset of led_detector.cpp:74)
        ref_und, ref_dist = orc.find_leds(np.pad(img, ((roi_xy[1], 0), (roi_xy[0], 0))), P, K, D,
                                          roi=(roi_xy[0], roi_xy[1], cols, rows))
        und, dist = _host_find_leds(host, orc, img, P, K, D, roi_xy)
        assert len(dist) == len(ref_dist), (kind, it, len(dist), len(ref_dist))
        assert np.array_equal(dist, ref_dist, equal_nan=True), (kind, it)   # float32 centroids, bit for bit, same order
        # (a zero-area contour that the wide-open filter lets through has a 0 / 0 centroid on both sides)
        assert np.array_equal(und, ref_und, equal_nan=True), (kind, it)     # undistorted: float32 stored as double
        n_blobs += len(dist)
    assert n_blobs > 100, n_blobs


def test_blurred_value_exactly_at_the_rounding_boundary(host, orc):
    """GaussianBlur keeps a pixel iff (sum + 2^15) >> 16 != 0, i.e. sum >= 2^15.  With taps [1, 42, 170, 42, 1] three dim
    pixels (1 at the output position, 2 one step diagonally, 2 two steps to the side) give exactly
    170*170 + 2*42*42 + 2*170 = 32768; the mask pixel they switch on touches the halo of a bright block three pixels
    away, so it changes that blob's contour and centroid: '>=' and '>' give different answers."""
    img = np.zeros((40, 48), np.uint8)
    img[10:20, 10:20] = 255              # its blurred mask reaches x = 21 (two pixels beyond the block)
    y, x = 15, 22                        # diagonal neighbour of the halo: 8-connected to the blob
    img[y, x] = 1
    img[y + 1, x + 1] = 2
    img[y, x + 2] = 2
    K, D = synth.camera_for(40, 48)
    P = orc.make_params(threshold_value=0, gaussian_sigma=0.6, min_blob_area=0.0, max_blob_area=1e9,
                        max_width_height_distortion=1e9, max_circular_distortion=1e9)
    assert list(orc.gaussian_kernel_q8(0.6)) == [1, 42, 170, 42, 1]
    blurred, mask = orc.blur_mask(img, 0, 0.6)
    assert mask[y, x] and not mask[y - 1, x] and not mask[y + 1, x]   # the bridge between the halo and (y, x + 1)
    ref_und, ref_dist = orc.find_leds(img, P, K, D)
    und, dist = _host_find_leds(host, orc, img, P, K, D)
    assert len(ref_dist) == 1 and np.array_equal(dist, ref_dist) and np.array_equal(und, ref_und)
    img[y, x] = 0                                      # below the boundary: the bridge is gone and the answer changes
    _, dist2 = _host_find_leds(host, orc, img, P, K, D)
    assert not np.array_equal(dist2, dist)


def test_scan_threshold_arithmetic_exhaustively(host):
    """The image scan flags a 16-byte segment iff one of its bytes exceeds the threshold (strictly, THRESH_TOZERO,
    led_detector.cpp:44) — computed with three SWAR operations per word on the device.  Every threshold from -1 to 255
    (and the clamps beyond) against the byte-wise definition, on segments with one byte at every value around the
    threshold, random segments, all-equal segments; the cheap OR test must never miss a hit."""
    rng = np.random.default_rng(8)
    host.host_scan_tests.restype = C.c_int
    for thr in list(range(-1, 256)) + [-5, 300]:
        t = min(255, max(-1, thr))
        segs = []
        for v in {max(0, t - 1), max(0, t), min(255, t + 1), 0, 127, 128, 255}:
            for pos in (0, 5, 15):
                s = np.full(16, rng.integers(0, max(1, t + 1)) if t >= 0 else 0, np.uint8)  # background <= thr
                s[pos] = v
                segs.append(s)
            segs.append(np.full(16, v, np.uint8))
        segs += [rng.integers(0, 256, 16).astype(np.uint8) for _ in range(20)]
        for s in segs:
            s = np.ascontiguousarray(s)
            want = bool((s.astype(int) > t).any())
            r = host.host_scan_tests(s.ctypes.data_as(C.c_void_p), thr)
            assert bool(r & 1) == want, (thr, s)
            assert (r & 2) or not want, (thr, s)        # maybe_gt16 is a necessary condition
            assert bool(r & 4) == want, (thr, s)        # ... and the compile-time AND / OR forms agree
            assert (r & 8) or not want, (thr, s)


def test_cell_sums_equal_the_border_trace(host):
    """The contour phase without border following (cells_phase: flood per component, Green sums over the 2 x 2 block
    cells of row pairs, Euler-number hole test) against the literal Suzuki-Abe trace (scan_window) on random masks: LED-like
    discs, dense noise with one-pixel-wide bridges and diagonal contacts, rings (holes: the island must go back to the
    trace), several islands side by side and stacked, islands too big for the phase.  Raw contour sums (up to
    orientation), bounding boxes, start keys and the filtered blobs must all agree; the phase must decide most islands
    itself and hand back those with holes."""
    from scipy import ndimage
    host.host_cells_vs_trace.restype = C.c_int
    rng = np.random.default_rng(2024)
    shape_real = np.array([10.0, 200.0, 0.5, 0.5])
    shape_open = np.array([0.0, 1e9, 1.0, 1e9])
    n_comp = n_fb = n_isl = n_hole_isl = 0

    def run(mask, wins, shape):
        nonlocal n_comp, n_fb, n_isl
        mask = np.ascontiguousarray(mask, np.uint8)
        w = np.ascontiguousarray(np.asarray(wins, np.int32).reshape(-1, 4))
        fb = C.c_int(0)
        r = host.host_cells_vs_trace(mask.ctypes.data_as(C.c_void_p), mask.shape[0], mask.shape[1],
                                     w.ctypes.data_as(C.c_void_p), len(w), shape.ctypes.data_as(C.c_void_p), C.byref(fb))
        assert r >= 0, (wins, np.argwhere(mask)[:10])
        n_comp += r
        n_fb += fb.value
        n_isl += len(w)
        return fb.value

    yy, xx = np.mgrid[0:40, 0:120]
    for it in range(1500):
        kind = it % 5
        rows, cols = int(rng.integers(4, 40)), int(rng.integers(4, 120))
        if kind == 0:      # LED-like discs, sometimes touching
            m = np.zeros((rows, cols), bool)
            for _ in range(int(rng.integers(1, 4))):
                cy, cx, rad = rng.uniform(0, rows), rng.uniform(0, cols), rng.uniform(1.0, 6.0)
                m |= (yy[:rows, :cols] - cy) ** 2 + (xx[:rows, :cols] - cx) ** 2 <= rad ** 2
        elif kind == 1:    # dense noise: thin bridges, diagonal contacts, small holes
            m = rng.random((rows, cols)) < rng.uniform(0.2, 0.9)
        elif kind == 2:    # closed noise: blobs with bays
            m = ndimage.binary_closing(rng.random((rows, cols)) < 0.35)
        elif kind == 3:    # a ring (hole) beside a disc
            m = np.zeros((rows, cols), bool)
            cy, cx = rows / 2, cols / 3
            d2 = (yy[:rows, :cols] - cy) ** 2 + (xx[:rows, :cols] - cx) ** 2
            m |= (d2 <= 36) & (d2 >= 9)
            m |= (yy[:rows, :cols] - cy) ** 2 + (xx[:rows, :cols] - 2.2 * cx) ** 2 <= 9
        else:              # sparse specks
            m = rng.random((rows, cols)) < 0.03
        shape = shape_real if it % 2 else shape_open
        run(m, [[0, rows, 0, cols]], shape)
        # the same content as two or three independent islands: column bands with an empty gap, or row bands
        if cols >= 30 and it % 3 == 0:
            c1, c2 = cols // 3, 2 * cols // 3
            m2 = m.copy()
            m2[:, c1 - 1:c1 + 1] = False
            m2[:, c2 - 1:c2 + 1] = False
            run(m2, [[0, rows, 0, c1], [0, rows, c1, c2], [0, rows, c2, cols]], shape)
        if rows >= 12 and it % 3 == 1:
            r1 = rows // 2
            m2 = m.copy()
            m2[r1 - 1:r1 + 1, :] = False
            run(m2, [[0, r1, 0, cols], [r1, rows, 0, cols]], shape)
    # a ring always goes back to the trace; a plain disc never does
    ring = ((yy - 20) ** 2 + (xx - 20) ** 2 <= 64) & ((yy - 20) ** 2 + (xx - 20) ** 2 >= 16)
    assert run(ring[:, :60], [[0, 40, 0, 60]], shape_open) == 1
    disc = (yy - 20) ** 2 + (xx - 20) ** 2 <= 64
    assert run(disc[:, :60], [[0, 40, 0, 60]], shape_open) == 0
    # an island beyond the phase's capacity (rows x words) goes back as a whole
    big = np.zeros((100, 200), bool)
    big[5:90, 10:190] = rng.random((85, 180)) < 0.5
    assert run(big, [[0, 100, 0, 200]], shape_open) == 1
    assert n_comp > 5000 and n_isl > 2000
    assert 0 < n_fb < 0.6 * n_isl, (n_fb, n_isl)


def test_band_scans_equal_the_whole_frame_scan(host):
    """Round 5: the general blob tier (k1b_general) scans one lane per BAND — a maximal run of rows that hold a set
    pixel — instead of one lane over the whole frame.  The decomposition must be exact for findContours(RETR_EXTERNAL):
    same components, raw contour sums, bounding boxes, start keys, filtered blobs.  The device's scan_window (cut out
    of the kernel source) is run over the whole bitmap and then once per band, called as the kernel calls it, on random
    masks: specks of several densities (many short bands, several blobs per band), noise stripes, tall bars through
    many bands' worth of rows, rings with blobs nested inside (not external) and outside, blobs touching the image
    borders, dense noise (one band), widths on both sides of a 64-bit word boundary."""
    from scipy import ndimage
    host.host_bands_vs_whole.restype = C.c_int
    rng = np.random.default_rng(515)
    shape_real = np.array([10.0, 200.0, 0.5, 0.5])
    shape_open = np.array([0.0, 1e9, 1.0, 1e9])
    n_comp = n_bands = n_multi = 0
    for it in range(1200):
        kind = it % 6
        rows = int(rng.integers(3, 70))
        cols = int(rng.choice([rng.integers(3, 60), 61, 62, 63, 64, 65, rng.integers(66, 200)]))
        yy, xx = np.mgrid[0:rows, 0:cols]
        if kind == 0:      # specks: many bands
            m = rng.random((rows, cols)) < rng.choice([0.002, 0.01, 0.03])
            m = ndimage.binary_dilation(m, iterations=int(rng.integers(0, 3))) if rng.random() < 0.7 else m
        elif kind == 1:    # noise stripes separated by empty rows, first / last stripe at the border
            m = np.zeros((rows, cols), bool)
            y = 0
            while y < rows:
                h = int(rng.integers(1, 6))
                m[y:y + h] = rng.random((min(h, rows - y), cols)) < 0.15
                y += h + int(rng.integers(1, 4))
        elif kind == 2:    # a tall bar through the specks
            m = rng.random((rows, cols)) < 0.01
            x0 = int(rng.integers(0, cols))
            m[int(rng.integers(0, rows // 2 + 1)):rows - int(rng.integers(0, rows // 3 + 1)), x0:x0 + 2] = True
        elif kind == 3:    # a ring with blobs inside and outside
            m = rng.random((rows, cols)) < 0.02
            rad = min(rows, cols) / 2.5
            d = np.hypot(yy - rows / 2, xx - cols / 2)
            m |= np.abs(d - rad) < 1.2
        elif kind == 4:    # dense noise: one band, holes, diagonal contacts
            m = rng.random((rows, cols)) < rng.uniform(0.2, 0.7)
        else:              # discs, some cut by the borders
            m = np.zeros((rows, cols), bool)
            for _ in range(int(rng.integers(1, 7))):
                cy, cx, rad = rng.uniform(-2, rows + 2), rng.uniform(-2, cols + 2), rng.uniform(0.8, 6.0)
                m |= (yy - cy) ** 2 + (xx - cx) ** 2 <= rad ** 2
        mask = np.ascontiguousarray(m, np.uint8)
        shape = shape_real if it % 2 else shape_open
        nb = C.c_int(0)
        r = host.host_bands_vs_whole(mask.ctypes.data_as(C.c_void_p), rows, cols, shape.ctypes.data_as(C.c_void_p), C.byref(nb))
        assert r >= 0, (it, kind, rows, cols)
        n_comp += r
        n_bands += nb.value
        n_multi += int(nb.value > 1)
    assert n_comp > 8000 and n_bands > 3000 and n_multi > 500, (n_comp, n_bands, n_multi)
    # the comparison has teeth: a band window that starts one row late must be caught
    os.environ["K1B_HOST_BREAK_BANDS"] = "1"
    try:
        m = np.zeros((20, 30), np.uint8)
        m[3:15, 5:12] = 1
        nb = C.c_int(0)
        assert host.host_bands_vs_whole(m.ctypes.data_as(C.c_void_p), 20, 30, shape_open.ctypes.data_as(C.c_void_p), C.byref(nb)) == -1
    finally:
        del os.environ["K1B_HOST_BREAK_BANDS"]


def test_column_run_scans_equal_the_whole_frame_scan(host):
    """Round 6: the general blob tier cuts every band again at its EMPTY COLUMNS and scans (band, column run) items, one
    lane each (window_column_runs + scan_window<true>, cut out of the kernel source and called as the kernel calls
    them; the runs of a band in reverse order: on the device they run concurrently).  Exact for
    findContours(RETR_EXTERNAL): same components, raw contour sums, bounding boxes, start keys, filtered blobs as the
    whole-frame scan — specks (many runs per band, several sharing a 64-bit word), stripes, bars, rings with blobs
    inside (a ring's columns are one run: what it encloses stays inside it), dense noise, discs cut by the borders,
    widths around the word boundaries."""
    from scipy import ndimage
    host.host_runs_vs_whole.restype = C.c_int
    rng = np.random.default_rng(616)
    shape_real = np.array([10.0, 200.0, 0.5, 0.5])
    shape_open = np.array([0.0, 1e9, 1.0, 1e9])
    n_comp = n_runs = n_multi = 0
    for it in range(1500):
        kind = it % 6
        rows = int(rng.integers(3, 70))
        cols = int(rng.choice([rng.integers(3, 60), 61, 62, 63, 64, 65, 126, 127, 128, 129, rng.integers(66, 400)]))
        if it % 25 == 7:  # bitmap rows of 17 .. 64 words (a wave holds a row's words on 32 / 64 lanes)
            cols = int(rng.choice([957, 958, 959, 1022, 1920, 1983, rng.integers(960, 3900), 3966]))
        yy, xx = np.mgrid[0:rows, 0:cols]
        if kind == 0:
            m = rng.random((rows, cols)) < rng.choice([0.002, 0.01, 0.03])
            m = ndimage.binary_dilation(m, iterations=int(rng.integers(0, 3))) if rng.random() < 0.7 else m
        elif kind == 1:
            m = np.zeros((rows, cols), bool)
            x = 0
            while x < cols:   # noise stripes separated by empty COLUMNS of width 1 .. 3
                w = int(rng.integers(1, 9))
                m[:, x:x + w] = rng.random((rows, min(w, cols - x))) < 0.2
                x += w + int(rng.integers(1, 4))
        elif kind == 2:
            m = rng.random((rows, cols)) < 0.01
            y0 = int(rng.integers(0, rows))
            m[y0:y0 + 2, int(rng.integers(0, cols // 2 + 1)):cols - int(rng.integers(0, cols // 3 + 1))] = True  # a wide bar
        elif kind == 3:
            m = rng.random((rows, cols)) < 0.02
            rad = min(rows, cols) / 2.5
            d = np.hypot(yy - rows / 2, xx - cols / 2)
            m |= np.abs(d - rad) < 1.2
        elif kind == 4:
            m = rng.random((rows, cols)) < rng.uniform(0.05, 0.6)
        else:
            m = np.zeros((rows, cols), bool)
            for _ in range(int(rng.integers(1, 9))):
                cy, cx, rad = rng.uniform(-2, rows + 2), rng.uniform(-2, cols + 2), rng.uniform(0.8, 6.0)
                m |= (yy - cy) ** 2 + (xx - cx) ** 2 <= rad ** 2
        mask = np.ascontiguousarray(m, np.uint8)
        shape = shape_real if it % 2 else shape_open
        nr = C.c_int(0)
        r = host.host_runs_vs_whole(mask.ctypes.data_as(C.c_void_p), rows, cols, shape.ctypes.data_as(C.c_void_p), C.byref(nr))
        assert r >= 0, (it, kind, rows, cols, r)
        n_comp += r
        n_runs += nr.value
        n_multi += int(nr.value > 3)
    assert n_comp > 10000 and n_runs > 8000 and n_multi > 700, (n_comp, n_runs, n_multi)
    os.environ["K1B_HOST_BREAK_RUNS"] = "1"   # the comparison has teeth: a run that starts one column late must be caught
    try:
        m = np.zeros((20, 30), np.uint8)
        m[3:15, 5:12] = 1
        nr = C.c_int(0)
        assert host.host_runs_vs_whole(m.ctypes.data_as(C.c_void_p), 20, 30, shape_open.ctypes.data_as(C.c_void_p), C.byref(nr)) == -1
    finally:
        del os.environ["K1B_HOST_BREAK_RUNS"]


def test_raw_frame_blur_equals_the_blur_of_the_thresholded_copy(host, orc):
    """Round 5: the general blob tier no longer copies the frame; its blur reads the frame's own rows, loads only the
    segments the image pass FLAGGED, applies THRESH_TOZERO on the fly and skips rows without a flagged segment
    (blur_item_fast<.., RAW>).  Against the blur of a thresholded copy (the LDS tiers' input) the non-zero bitmaps must
    be identical: random frames with spots, salt noise, bright patches at the borders (the mirrored-border variant and
    its byte-wise fallback), widths that are not multiples of 16, 3- and 5-tap kernels, thresholds on both sides of
    128, a frame whose first flag bit is not word aligned."""
    host.host_blur_raw_vs_copy.restype = C.c_int
    rng = np.random.default_rng(99)
    total = 0
    for it in range(300):
        rows, cols = int(rng.integers(6, 60)), int(rng.choice([rng.integers(17, 120), 32, 48, 64, 33, 47]))
        img = rng.integers(0, 60, (rows, cols)).astype(np.uint8)
        kind = it % 4
        if kind == 0:
            img[rng.random((rows, cols)) < 0.01] = 255
        elif kind == 1:
            for _ in range(int(rng.integers(1, 5))):
                y, x = int(rng.integers(0, rows)), int(rng.integers(0, cols))
                img[max(0, y - 2):y + 3, max(0, x - 2):x + 3] = rng.integers(100, 256)
        elif kind == 2:   # bright pixels on the borders
            img[:, 0] = rng.integers(0, 256, rows)
            img[:, cols - 1] = rng.integers(0, 256, rows)
            img[0, :] = rng.integers(0, 256, cols)
            img[rows - 1, :] = rng.integers(0, 256, cols)
        else:
            img = rng.integers(0, 256, (rows, cols)).astype(np.uint8)
        thr = int(rng.choice([20, 100, 127, 128, 140, 200, 254]))
        sigma = float(rng.choice([0.4, 0.6, 0.8]))
        taps = np.ascontiguousarray(orc.gaussian_kernel_q8(sigma), np.int32)
        img = np.ascontiguousarray(img)
        r = host.host_blur_raw_vs_copy(img.ctypes.data_as(C.c_void_p), rows, cols, thr, taps.ctypes.data_as(C.c_void_p), len(taps))
        assert r >= 0, (it, rows, cols, thr, sigma)
        total += r
    assert total > 50000
