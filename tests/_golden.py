"""Load tests/golden/<case>.npz fixtures (made by tests/golden/make_golden.py from the unmodified
reference) into (case, expected) dicts."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names(prefix=""):
    out = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    return [n for n in out if n.startswith(prefix)]


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    case, exp = {"name": name}, {}
    for k in z.files:
        v = z[k]
        if k.startswith("in__"):
            case[k[4:]] = np.ascontiguousarray(v)
        elif k.startswith("p__"):
            case[k[3:]] = v.item() if v.ndim == 0 else v
        elif k.startswith("out__"):
            exp[k[5:]] = v
    return case, exp


def rel_err(a, b):
    """max |a-b| / max(|b|): the 'within X rel of the CPU reference' measure used for float outputs."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    den = max(float(np.max(np.abs(b))), 1e-30)
    return float(np.max(np.abs(a - b))) / den
