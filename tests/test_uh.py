"""CPU: host-side unit-hydrograph setup against the reference's own basinUH / make_uh output
(stored in the golden fixtures; defaults fshape=2.5, tscale=86400, velo=1.5, diff=5000 of
route/ancillary_data/param.nml.default)."""
import numpy as np
import pytest

from helpers import load_golden
from mizuroute_amd import uh as uhmod


@pytest.mark.parametrize("name", ["cameo50_irf", "tree150_all"])
def test_basin_uh_and_make_uh_match_reference(name):
    net, z = load_golden(name)
    dt = float(z["dt"])
    frac = uhmod.basin_uh(dt, 2.5, 86400.0)
    assert frac.shape == z["frac_future"].shape
    assert np.allclose(frac, z["frac_future"], rtol=1e-12, atol=1e-18)
    off, u = uhmod.make_uh(net.params["RLENGTH"], dt, 1.5, 5000.0)
    assert np.array_equal(off, z["uh_offset"])
    assert np.allclose(u, z["uh"], rtol=1e-11, atol=1e-18)


def test_gammp_known_values():
    # P(1, x) = 1 - exp(-x);  P(a, 0) = 0
    for x in (0.1, 1.0, 3.5):
        assert abs(uhmod.gammp(1.0, x) - (1 - np.exp(-x))) < 1e-14
    assert uhmod.gammp(2.5, 0.0) == 0.0
