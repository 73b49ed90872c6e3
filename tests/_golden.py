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


# Element-wise bar next to the per-tensor norm: every element within rtol of ITS OWN reference value,
# plus a floor of afrac of the tensor's largest magnitude for the elements that are sums cancelling to (almost) nothing --
# rel_err alone leaves small-magnitude elements unconstrained.
ELEM_RTOL, ELEM_AFRAC = 1e-4, 1e-5


def elem_err(a, b, rtol=ELEM_RTOL, afrac=ELEM_AFRAC):
    """(worst |a-b| / (rtol |b| + afrac max|b|), flat index of that element, a there, b there); <= 1 passes."""
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    if a.size == 0:
        return 0.0, -1, 0.0, 0.0
    bound = rtol * np.abs(b) + afrac * max(float(np.max(np.abs(b))), 1e-30)
    ratio = np.abs(a - b) / bound
    i = int(np.argmax(ratio))
    return float(ratio[i]), i, float(a[i]), float(b[i])


WORST = {}  # (tag) -> (rel_err, elem ratio): filled by assert_close, printed by tests/conftest.py at the end of a session


def assert_close(tag, a, b, tol=1e-4, rtol=ELEM_RTOL, afrac=ELEM_AFRAC):
    """both bars: max|a-b| / max|b| <= tol, and every element within rtol |b| + afrac max|b|; names the worst offender"""
    err = rel_err(a, b)
    ratio, i, av, bv = elem_err(a, b, rtol, afrac)
    cls = str(tag).split("/")[-1]
    old = WORST.get(cls, (0.0, 0.0))
    WORST[cls] = (max(old[0], err), max(old[1], ratio))
    assert err <= tol, "%s: rel err %.3g > %.1g" % (tag, err, tol)
    assert ratio <= 1.0, ("%s: element %d is %.9g, reference %.9g: |diff| %.3g = %.2f x (%.0e |ref| + %.0e max|ref|)"
                          % (tag, i, av, bv, abs(av - bv), ratio, rtol, afrac))
    return err, ratio
