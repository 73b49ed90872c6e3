"""The pins pin themselves: every tests/golden/*.npz is re-made from the unmodified reference (oracle/_ref/libbcnn_ref.so,
compiled from /root/reference by oracle/Makefile) and has to come out identical, inputs being the stored ones. Runs where
the reference tree is mounted (the build container); on the GPU box, which has no /root/reference, the library built here
travels along and the check still runs, without it the test is skipped."""
import numpy as np
import pytest

from tests import _golden as G


@pytest.mark.parametrize("name", G.names())
def test_fixture_equals_what_the_reference_produces_today(name):
    from oracle import ref_bind as rb
    if not rb.available():
        pytest.skip("oracle/_ref/libbcnn_ref.so not built (needs /root/reference)")
    from oracle import ref_cases as rc
    case, exp = G.load(name)
    got = rc.run_ref(case)
    assert set(got) == set(exp), (sorted(got), sorted(exp))
    for k in exp:
        a, b = np.asarray(got[k]), np.asarray(exp[k])
        assert a.shape == b.shape and a.dtype == b.dtype, (k, a.shape, b.shape, a.dtype, b.dtype)
        assert np.array_equal(a, b, equal_nan=True), (k, float(np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64)))))
