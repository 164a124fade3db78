import glob
import hashlib
import os

import numpy as np

from rpg_monocular_pose_estimator_amd import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
                  if not os.path.basename(p).startswith(("seq_", "witness_seq_")))


def witness_sequences():
    """Tracking-path vectors made by the independent witness (tests/golden/make_witness_seq_golden.py)."""
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "witness_seq_*.npz")))


def golden_sequences():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "seq_*.npz")))


def load_sequence(name):
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    d = synth.make_sequence(str(g["config"]), int(g["n"]), seed=int(g["seed"]), dropout=tuple(int(x) for x in g["dropout"]))
    sha = [hashlib.sha1(f.tobytes()).hexdigest() for f in d["frames"]]
    assert sha == [str(s) for s in g["sha1"]], "synthetic sequence generator drifted from the golden scenes"
    return g, d


def load(name):
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    d = synth.make_frames(str(g["config"]), int(g["n"]), int(g["seed"]))
    sha = [hashlib.sha1(f.tobytes()).hexdigest() for f in d["frames"]]
    assert sha == [str(s) for s in g["sha1"]], "synthetic generator drifted from the golden scenes"
    return g, d
