"""CPU-only vote-histogram soak of the DEVICE voting source: the work item of k2_vote (cut out of mpe_k2.hip and
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

        def run(var):
            out = np.zeros((n, MAX_DET, MAX_MARK), np.uint32)
            stats = np.zeros(3, np.uint32)
            rc = lib.host_vote_batch(det.ctypes.data_as(C.c_void_p), n_det.ctypes.data_as(C.c_void_p), n,
                                     markers.ctypes.data_as(C.c_void_p), len(markers), k4.ctypes.data_as(C.c_void_p),
                                     C.c_double(5.0), out.ctypes.data_as(C.c_void_p), var, stats.ctypes.data_as(C.c_void_p))
            assert rc == 0, rc
            return out, stats
        got, _ = run(variant)               # the fast arithmetic deciding everything itself (round 3)
        fixed, stats = run(variant + 10)    # ... with its suspects re-evaluated by the strict functions (default now)
        strict, _ = run(20)                 # every hypothesis through the strict functions
        t2 = time.time()
    nm = len(markers)

    def differing(a, b):
        return [i for i in range(n) if not np.array_equal(a[i, :nd, :nm], b[i, :nd, :nm])]
    bad = differing(got, ref)
    bad_fixed_vs_strict = differing(fixed, strict)
    bad_fast_vs_strict = differing(got, strict)
    bad_strict = differing(strict, ref)
    bad_fixed = differing(fixed, ref)
    cls = [forensics.classify_frame(det[i, :nd], markers, K, 5.0, orc) for i in sorted(set(bad) | set(bad_fixed))]
    n_comb = nd * (nd - 1) * (nd - 2) // 6
    hyp = n * n_comb * nm * (nm - 1) * (nm - 2)
    print(json.dumps({"config": config, "frames": n, "variant": variant, "hypotheses": hyp,
                      "fast_alone_vs_oracle": len(bad), "fast_alone_vs_strict": len(bad_fast_vs_strict),
                      "fast_with_fixup_vs_strict": len(bad_fixed_vs_strict),
                      "fast_with_fixup_vs_oracle": len(bad_fixed), "strict_vs_oracle": len(bad_strict),
                      "suspect_entries": int(stats[0]), "suspect_whole_hypotheses": int(stats[1]),
                      "suspect_list_full": int(stats[2]), "suspect_rate_per_hypothesis": float(stats[0]) / hyp,
                      "mismatches_classified_unstable": sum(1 for c in cls if c["unstable"]),
                      "mismatches_unexplained": sum(1 for c in cls if not c["unstable"]),
                      "frames_idx": bad[:20], "frames_fixed_vs_strict": bad_fixed_vs_strict[:20],
                      "classification": cls[:20],
                      "oracle_s": round(t1 - t0, 1), "host_s": round(t2 - t1, 1)}))
    return 1 if (bad_fixed_vs_strict or any(not c["unstable"] for c in cls)) else 0


if __name__ == "__main__":
    sys.exit(main())
