"""CPU-only tests of the host-side mirror (filter design etc.) against the golden vectors."""
import numpy as np
import pytest

from f2cnn_amd.gammatone import filters


@pytest.mark.parametrize("C", [8, 64, 128])
def test_filter_design_matches_reference(golden, C):
    cf = filters.centre_freqs(16000, C, 100)
    np.testing.assert_allclose(cf, golden[f"g1_cf_{C}"], rtol=1e-14, atol=0)
    co = filters.make_erb_filters(16000, cf)
    ref = golden[f"g1_coefs_{C}"]
    assert co.shape == ref.shape
    np.testing.assert_allclose(co, ref, rtol=1e-12, atol=0)
    assert np.all(co[:, 5] == 0) and np.all(co[:, 6] == 1)


def test_erb_space_defaults():
    assert filters.erb_space().shape == (100,)
    assert abs(filters.erb_point(100, 8000, 1) - 100) < 1e-9 and abs(filters.erb_point(100, 8000, 0) - 8000) < 1e-9
