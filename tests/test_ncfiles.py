"""CPU: the restart / history NetCDF layout (mizuroute_amd/ncfiles.py) written and read back with
scipy only; names, dimensions and padding are the reference's (write_restart_pio.f90, read_restart.f90)."""
import numpy as np
from scipy.io import netcdf_file

from mizuroute_amd import api, ncfiles


def fake_state(N=7, seed=0):
    rng = np.random.default_rng(seed)
    off = np.concatenate([[0], np.cumsum(rng.integers(1, 5, N))]).astype(np.int32)
    nw = rng.integers(1, 21, N).astype(np.int32)
    pad = lambda: np.where(np.arange(api.WCAP)[None, :] < nw[:, None], rng.random((N, api.WCAP)), -9999.0)
    rf = np.zeros((N, api.WCAP), np.int32); rf[:, 0] = 1
    st = dict(basin_q=rng.random(N), qfuture=rng.random((N, 5)), volume_irf=rng.random(N), volume_kwt=rng.random(N),
              volume_mc=rng.random(N), irf_qfuture=rng.random(int(off[-1])), numWaves=nw, qwave=pad(), tentry=pad(), texit=pad(),
              routed=rf, q_sub_mc=rng.random((N, 2)))
    return st, off


def test_restart_file_round_trip_and_layout(tmp_path):
    st, off = fake_state()
    N = st["basin_q"].size
    path = str(tmp_path / "case.r.2001-01-02-00000.nc")
    ncfiles.write_restart_file(path, st, np.arange(101, 101 + N), off, (82800.0, 86400.0), restart_time=86400.0)
    f = netcdf_file(path, "r", mmap=False)
    assert f.version_byte == 2                                            # 64bit_offset, public_var.f90:54
    assert f.dimensions["wave"] == 20 and f.dimensions["tbound"] == 2 and f.dimensions["seg"] == N
    assert f.variables["qwave"].dimensions == ("wave", "seg")            # Fortran (seg, wave)
    assert f.variables["irf_qfuture"].dimensions == ("tdh_irf", "seg") and f.variables["qfuture"].dimensions == ("tdh", "seg")
    assert f.variables["q_sub_mc"].dimensions == ("mol_mc", "seg")
    q = f.variables["qwave"][:].T
    nw = st["numWaves"]
    assert all((q[e, nw[e]:] == -9999.0).all() for e in range(N))         # padding, write_restart_pio.f90:1105-1106
    assert (f.variables["qwave_mod"][:] == -9999.0).all()
    assert np.array_equal(f.variables["numQF"][:], np.diff(off))
    f.close()
    back = ncfiles.read_restart_file(path)
    for k in ("basin_q", "qfuture", "volume_irf", "volume_kwt", "volume_mc", "irf_qfuture", "numWaves", "q_sub_mc"):
        assert np.array_equal(back[k], st[k]), k
    live = np.arange(api.WCAP)[None, :] < nw[:, None]
    for k in ("qwave", "tentry", "texit"):
        assert np.array_equal(back[k][live], st[k][live]), k
    assert np.array_equal(back["routed"][live], st["routed"][live])
    assert np.allclose(back["time_bound"], (82800.0, 86400.0))


def test_history_file_layout(tmp_path):
    class Dom:   # stand-in for RoutingDomain: device-side interval means
        N = 5
        def mean_q(self, m, reset=False): return np.arange(5) + 0.123456789 * m
        def flux(self, m, which): return np.ones(5) * m
    path = str(tmp_path / "case.h.2001-01-01-00000.nc")
    w = ncfiles.HistoryWriter(path, np.arange(5) + 1, [api.KWT, api.IRF], volumes=True)
    w.append(0.0, 3600.0, Dom()); w.append(3600.0, 7200.0, Dom(), stamp_offset=1800.0); w.close()
    f = netcdf_file(path, "r", mmap=False)
    assert f.variables["KWTroutedRunoff"].dimensions == ("time", "seg") and f.variables["KWTroutedRunoff"][:].dtype.itemsize == 4
    assert f.variables["KWTroutedRunoff"][:].shape == (2, 5) and "IRFvolume" in f.variables
    # time = start of the aggregated interval (+ stamp offset), both ends in time_bounds (historyFile.f90:349-373)
    assert np.array_equal(f.variables["time"][:], [0.0, 3600.0 + 1800.0])
    assert np.array_equal(f.variables["time_bounds"][:], [[0.0, 3600.0], [3600.0, 7200.0]])
    assert np.allclose(f.variables["IRFroutedRunoff"][1], np.arange(5) + 0.123456789, rtol=1e-7)
    f.close()
