"""CPU-only vote-histogram soak of the DEVICE voting source: the work item of k2_vote (cut out of mpe_kernels.hip and
compiled for the host, as tests/test_vote_host.py does) against the oracle's initialise() loop on N synthetic
detection sets — the undistorted LED projections of random scenes, rounded to float32 as findLeds delivers them, in
random order.  Counts the frames whose histogram differs anywhere and classifies each one: a frame is "unstable" when
one of its hypotheses sits in the corner of the reference's Ferrari solver where |alpha + 2y| cancels (tests/util.py
ferrari_w), i.e. where two builds of the reference itself disagree.
usage: python tests/soak_votes_host.py [frames [config [variant]]]     -> one JSON line (no GPU needed)"""
import ctypes as C
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from rpg_monocular_pose_estimator_amd import synth  # noqa: E402
import oracle  # noqa: E402
import test_vote_host  # noqa: E402

MAX_DET, MAX_MARK = 32, 16


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    config = sys.argv[2] if len(sys.argv) > 2 else "C2"
    variant = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    oracle.build()
    from oracle import binding as orc
    import forensics
    cfg = synth.CONFIGS[config]
    K, D = synth.camera_for(cfg["rows"], cfg["cols"])
    markers = np.ascontiguousarray(cfg["markers"], float)
    with tempfile.TemporaryDirectory() as d:
        lib = test_vote_host._build(d)
        lib.host_vote_batch.restype = C.c_int
        _, spots = synth.make_scenes_batch(cfg, n, seed=4242)
        rng = np.random.default_rng(5)
        nd = spots.shape[1]
        det = np.zeros((n, MAX_DET, 2))
        for i in range(n):
            und = orc.undistort_points(spots[i].astype(np.float32), K, D)
            det[i, :nd] = np.asarray(und, np.float32).astype(float)[rng.permutation(nd)]
        n_det = np.full(n, nd, np.int32)
        t0 = time.time()
        ref = orc.vote_batch(det, n_det, markers, K, 5.0, n_threads=os.cpu_count() or 1)
        t1 = time.time()
        k4 = np.array([K[0][0], K[1][1], K[0][2], K[1][2]], float)
        got = np.zeros((n, MAX_DET, MAX_MARK), np.uint32)
        rc = lib.host_vote_batch(det.ctypes.data_as(C.c_void_p), n_det.ctypes.data_as(C.c_void_p), n,
                                 markers.ctypes.data_as(C.c_void_p), len(markers), k4.ctypes.data_as(C.c_void_p),
                                 C.c_double(5.0), got.ctypes.data_as(C.c_void_p), variant)
        assert rc == 0, rc
        t2 = time.time()
    bad = [i for i in range(n) if not np.array_equal(got[i, :nd, :len(markers)], ref[i, :nd])]
    cls = [forensics.classify_frame(det[i, :nd], markers, K, 5.0, orc) for i in bad]
    print(json.dumps({"config": config, "frames": n, "variant": variant,
                      "frames_with_a_different_histogram": len(bad),
                      "mismatches_classified_unstable": sum(1 for c in cls if c["unstable"]),
                      "mismatches_unexplained": sum(1 for c in cls if not c["unstable"]),
                      "frames_idx": bad[:20], "classification": cls[:20],
                      "oracle_s": round(t1 - t0, 1), "host_s": round(t2 - t1, 1)}))
    return 1 if any(not c["unstable"] for c in cls) else 0


if __name__ == "__main__":
    sys.exit(main())
