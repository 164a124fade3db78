import glob
import hashlib
import os

import numpy as np

from rpg_monocular_pose_estimator_amd import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
                  if not os.path.basename(p).startswith(("seq_", "witness_seq_", "witness_clutter")))


def witness_sequences():
    """Tracking-path vectors made by the independent witness (tests/golden/make_witness_seq_golden.py)."""
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "witness_seq_*.npz")))


def golden_sequences():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "seq_*.npz")))


def load_sequence(name):
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    d = synth.make_sequence(str(g["config"]), int(g["n"]), seed=int(g["seed"]), dropout=tuple(int(x) for x in g["dropout"]),
                            salt=float(g["salt"]) if "salt" in g else 0.0)
    sha = [hashlib.sha1(f.tobytes()).hexdigest() for f in d["frames"]]
    assert sha == [str(s) for s in g["sha1"]], "synthetic sequence generator drifted from the golden scenes"
    return g, d


def load(name):
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    cfg = str(g["config"])
    if "n_distractors" in g:   # (a BASELINE config with another number of distractor spots)
        cfg = dict(synth.CONFIGS[cfg], n_distractors=int(g["n_distractors"]))
    d = synth.make_frames(cfg, int(g["n"]), int(g["seed"]))
    sha = [hashlib.sha1(f.tobytes()).hexdigest() for f in d["frames"]]
    assert sha == [str(s) for s in g["sha1"]], "synthetic generator drifted from the golden scenes"
    return g, d


def load_clutter(config="C2"):
    """tests/golden/witness_clutter.npz (detections of cluttered C2 frames by the independent witness; config "C4":
    witness_clutter_C4.npz, 1920x1200) ->
    list of (kind, threshold, frame (rows, cols) uint8, K, D, n_det, dist_xy (n,2) f32, undist_xy (n,2) f64);
    frames regenerated from (kind, seed) and checked against their SHA-1."""
    import hashlib
    from rpg_monocular_pose_estimator_amd import synth
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                             "witness_clutter.npz" if config == "C2" else "witness_clutter_%s.npz" % config))
    cache, out = {}, []
    for j in range(len(g["kind"])):
        kind, seed, i, thr = str(g["kind"][j]), int(g["seed"][j]), int(g["frame"][j]), int(g["threshold"][j])
        if (kind, seed) not in cache:
            cache[(kind, seed)] = synth.make_clutter_frames(kind, 3 if config == "C2" else 1, seed, config)
        d = cache[(kind, seed)]
        f = d["frames"][i]
        assert hashlib.sha1(f.tobytes()).hexdigest() == str(g["sha1"][j]), "clutter frame generator drifted: %s %d" % (kind, i)
        k = int(g["n_det"][j])
        out.append((kind, thr, f, d["K"], d["D"], k, g["dist_xy"][j, :k], g["undist_xy"][j, :k]))
    return out
