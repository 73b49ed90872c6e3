import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import _golden as G
from tests import _hip_cases as HC
name = sys.argv[1] if len(sys.argv) > 1 else "conv_abs"
case, exp = G.load(name)
got = HC.run_hip(case, exp)
y, e = got["y"], exp["y"]
d = np.abs(y - e)
print(name, "shape", y.shape, "max diff", d.max())
bad = np.argwhere(d > 1e-4)
print("bad count", len(bad), "of", y.size)
print(bad[:40].tolist())
np.set_printoptions(precision=3, suppress=True, linewidth=200)
print("got[0,0]\n", y[0, 0]); print("exp[0,0]\n", e[0, 0])
print("got[0,1]\n", y[0, 1]); print("exp[0,1]\n", e[0, 1])
print("got[1,2]\n", y[1, 2]); print("exp[1,2]\n", e[1, 2])
