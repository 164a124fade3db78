import glob
import hashlib
import os

import numpy as np

from rpg_monocular_pose_estimator_amd import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load(name):
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    d = synth.make_frames(str(g["config"]), int(g["n"]), int(g["seed"]))
    sha = [hashlib.sha1(f.tobytes()).hexdigest() for f in d["frames"]]
    assert sha == [str(s) for s in g["sha1"]], "synthetic generator drifted from the golden scenes"
    return g, d
