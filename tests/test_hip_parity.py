"""GPU parity tests: the HIP path (through the C-ABI of include/bcnn_hip.h) against
 (1) the golden vectors taken from the unmodified reference (tests/golden), and
 (2) the CPU oracle on larger seeded inputs.
Bars (BASELINE.json north_star): pooling indices bit-exact; conv / batch-norm within 1e-4 relative
(max |a-b| / max |b| per tensor); everything else elementwise-tight."""
import numpy as np
import pytest

from tests import _golden as G

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4          # conv / batchnorm / depthwise / gemm tensors
ACT_TOL = 2e-6          # activation map (exp/log evaluated in double on both sides)
VAR_KEYS = {"saved_var", "run_var", "dvar"}  # E[x^2]-E[x]^2 cancels: compare with abs + rel tolerance


def _check(name, got, exp, tol):
    for key, want in exp.items():
        if key == "dy":
            continue
        have = got[key]
        assert have.shape == want.shape, (name, key, have.shape, want.shape)
        if want.dtype.kind == "i":
            assert np.array_equal(have, want), "%s/%s: indices differ" % (name, key)
            continue
        assert np.array_equal(np.isnan(have), np.isnan(want)), (name, key, "NaN pattern")
        h, w = np.nan_to_num(have), np.nan_to_num(want)
        if key in VAR_KEYS:
            assert np.allclose(h, w, rtol=1e-4, atol=1e-6), (name, key, np.abs(h - w).max())
        else:
            # per-tensor norm AND the element-wise bar |a - b| <= tol |b| + tol / 10 max|b| (worst offender named)
            G.assert_close("%s/%s" % (name, key), h, w, tol, rtol=tol, afrac=tol / 10)


@pytest.mark.parametrize("name", G.names())
def test_hip_matches_reference_golden(name):
    from tests import _hip_cases as HC
    case, exp = G.load(name)
    got = HC.run_hip(case, exp)
    op = str(case["op"])
    if op == "maxpool":
        # bit-exact: values, int32 indices, and the backward scatter (same addition order)
        assert np.array_equal(got["indexes"], exp["indexes"])
        assert np.array_equal(got["y"], exp["y"], equal_nan=True)
        assert np.array_equal(got["dx"], exp["dx"], equal_nan=True)
        return
    _check(name, got, exp, ACT_TOL if op == "act" else REL_TOL)
