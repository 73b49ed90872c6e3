"""CPU (not-gpu) tests: the C restatement oracle/bcnn_oracle.c against the golden vectors taken
from the unmodified reference. This is what pins the oracle (the reference
ships no test vectors of its own, so the pins are reference outputs generated in the build
container)."""
import numpy as np
import pytest

from oracle import orc_bind
from tests import _golden as G

# The restatement keeps the reference's summation order, so agreement is at rounding level.
# (Not asserted bit-exact: libm exp/log may be contracted differently by the two compilations.)
TOL = 2e-6


@pytest.mark.parametrize("name", G.names())
def test_oracle_matches_reference_golden(name, oracle_lib):
    case, exp = G.load(name)
    got = orc_bind.run_oracle(case, exp)
    assert set(exp) <= set(got) | {"dy"}, (sorted(exp), sorted(got))
    for key, want in exp.items():
        if key == "dy":
            continue
        have = got[key]
        assert have.shape == want.shape, (key, have.shape, want.shape)
        if want.dtype.kind == "i":
            assert np.array_equal(have, want), "%s: index mismatch" % key  # bit-exact
        else:
            assert np.all(np.isfinite(have) == np.isfinite(want)), key
            err = G.rel_err(np.nan_to_num(have), np.nan_to_num(want))
            assert err <= TOL, "%s/%s rel err %.3g" % (name, key, err)


def test_oracle_bit_exact_on_integer_and_order_sensitive_paths(oracle_lib):
    """maxpool value/indices/backward and conv (K <= 384, same summation order) must be bit-equal."""
    for name in G.names("maxpool"):
        case, exp = G.load(name)
        got = orc_bind.run_oracle(case, exp)
        assert np.array_equal(got["indexes"], exp["indexes"])
        assert np.array_equal(got["y"], exp["y"], equal_nan=True)
        assert np.array_equal(got["dx"], exp["dx"], equal_nan=True)
    for name in ("conv_k3s1p1", "conv_groups2", "conv_k1s2_quirk1", "conv_bias_one_quirk2"):
        case, exp = G.load(name)
        got = orc_bind.run_oracle(case, exp)
        for key in ("y", "dw", "db", "dx"):
            assert np.array_equal(got[key], exp[key]), (name, key)


def test_sgd_update_matches_formula(oracle_lib):
    """A.9 / bcnn_learner.c:67-83: momentum is carried inside the gradient buffers."""
    rs = np.random.RandomState(0)
    w = rs.randn(37).astype(np.float32); b = rs.randn(5).astype(np.float32)
    dw = rs.randn(37).astype(np.float32); db = rs.randn(5).astype(np.float32)
    w0, b0, dw0, db0 = w.copy(), b.copy(), dw.copy(), db.copy()
    B, lr, mom, dec = 16, np.float32(0.003), np.float32(0.9), np.float32(5e-4)
    orc_bind.lib().orc_sgd_update(orc_bind.P(w), orc_bind.P(b), orc_bind.P(dw), orc_bind.P(db), 37, 5,
                                  B, lr, mom, dec)
    eb = b0 + (-lr / B) * db0
    g = dw0 + (dec * B) * w0
    ew = w0 + (-lr / B) * g
    assert np.allclose(b, eb, rtol=1e-6) and np.allclose(db, db0 * mom, rtol=1e-6)
    assert np.allclose(w, ew, rtol=1e-6) and np.allclose(dw, g * mom, rtol=1e-6)
