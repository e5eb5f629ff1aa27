"""Shared helpers of the Adam parity tests: replay a golden case (tests/golden/ref_adam_golden.npz, produced by the
reference's own scene/OurAdam.py) through an implementation."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_adam_golden.npz")
KEYS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
LRS = dict(xyz=1.6e-4, f_dc=2.5e-3, f_rest=1.25e-4, opacity=5e-2, scaling=5e-3, rotation=1e-3)
EPS = 1e-15


def load():
    return np.load(GOLDEN)


def case_names(z):
    return [str(n) for n in z["case_names"]]


def meta(z, name):
    P, steps, wd, seed = z[f"{name}.meta"]
    return int(P), int(steps), float(wd), int(seed)


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
