"""Stand-alone driver (mizuroute_amd/standalone.py): control file, parameter namelist, network
augmentation from a topology file, and -- on the GPU -- a whole run from files against the same run
through the API."""
import os

import numpy as np
import pytest
from scipy.io import netcdf_file

import mizuroute_amd as m
from mizuroute_amd import standalone


def write_case(tmp, net, runoff_mm_s, dt, route_opt="2", shuffle_seed=3, remap=None, new_file="single", extra="", lakes=None, wm=None, solute=None):
    """Topology, forcing (HM HRUs = RN HRUs in shuffled order), control file and namelist of a synthetic case."""
    rng = np.random.default_rng(shuffle_seed)
    N = net.N
    hru_id = (np.arange(N) + 50001).astype(np.int32)
    seg_of_hru = np.zeros(N, np.int32)
    for r in range(N):
        seg_of_hru[net.hruIndex[net.hruOffset[r]:net.hruOffset[r + 1]] - 1] = net.reachId[r]
    f = netcdf_file(os.path.join(tmp, "ntopo.nc"), "w", version=2)
    f.createDimension("seg", N); f.createDimension("hru", N)
    def var(name, typ, dim, data):
        v = f.createVariable(name, typ, (dim,)); v[:] = data
    down_id = np.where(net.downIndex > 0, net.reachId[np.maximum(net.downIndex, 1) - 1], -1).astype(np.int32)
    var("seg_id", "i", "seg", net.reachId.astype(np.int32)); var("tosegment", "i", "seg", down_id)
    var("Length", "d", "seg", net.params["RLENGTH"]); var("Slope", "d", "seg", net.params["R_SLOPE"])
    var("hruid", "i", "hru", hru_id); var("seg_hru_id", "i", "hru", seg_of_hru); var("Basin_Area", "d", "hru", net.params["BASAREA"])
    if lakes is not None:      # lake flags and parameters under the reference's own names (popMetadat.f90:124-232)
        from mizuroute_amd.lakepar import LAKE_PAR
        isl = np.zeros(N, np.int32); isl[lakes["reach"] - 1] = 1
        mt = np.zeros(N, np.int32); mt[lakes["reach"] - 1] = lakes["model_type"]
        var("islake", "i", "seg", isl); var("lakeModelType", "i", "seg", mt)
        if "targ_vol" in lakes:
            tv = np.zeros(N, np.int32); tv[lakes["reach"] - 1] = lakes["targ_vol"]
            var("LakeTargVol", "i", "seg", tv)
        for i, k in enumerate(LAKE_PAR):
            a = np.zeros(N); a[lakes["reach"] - 1] = lakes["par"][i]
            var(k, "d", "seg", a)
    f.close()
    perm = rng.permutation(N)                       # forcing file lists the HRUs in another order
    steps = runoff_mm_s.shape[0]
    g = netcdf_file(os.path.join(tmp, "runoff.nc"), "w", version=2)
    g.createDimension("time", None); g.createDimension("hru", N)
    t = g.createVariable("time", "d", ("time",)); t.units = "hours since 2001-01-01 00:00:00"
    h = g.createVariable("hru_id", "i", ("hru",)); h[:] = hru_id[perm]
    q = g.createVariable("RUNOFF", "d", ("time", "hru"))
    if lakes is not None:
        ev = g.createVariable("evap", "d", ("time", "hru")); pr = g.createVariable("precip", "d", ("time", "hru"))
    if solute is not None:
        so = g.createVariable("solute", "d", ("time", "hru"))
    for k in range(steps):
        t[k] = k * dt / 3600.0
        q[k, :] = runoff_mm_s[k, perm]
        if solute is not None:
            so[k, :] = solute[k, perm]
        if lakes is not None:
            ev[k, :] = lakes["evap"][k, perm] * 1000.0; pr[k, :] = lakes["precip"][k, perm] * 1000.0      # mm/s like the runoff
    g.close()
    if wm is not None:         # water-management file: a subset of the reaches in its own order, on its own (coarser) step
        segs, flux, vol, dt_wm = wm
        w = netcdf_file(os.path.join(tmp, "wm.nc"), "w", version=2)
        w.createDimension("time", None); w.createDimension("seg", segs.size)
        tw = w.createVariable("time", "d", ("time",)); tw.units = "hours since 2001-01-01 00:00:00"
        sid = w.createVariable("seg_id", "i", ("seg",)); sid[:] = net.reachId[segs].astype(np.int32)
        fx = w.createVariable("abs_inj", "d", ("time", "seg")); tvv = w.createVariable("target_vol", "d", ("time", "seg"))
        for k in range(flux.shape[0]):
            tw[k] = k * dt_wm / 3600.0
            fx[k, :] = flux[k]; tvv[k, :] = vol[k]
        w.close()
    open(os.path.join(tmp, "param.nml"), "w").write("&HSLOPE\n fshape = 2.5\n tscale = 86400\n/\n&IRF_UH\n velo = 1.5\n diff = 5000.0\n/\n&KWT\n mann_n = 0.01\n wscale = 0.001\n/\n")
    end = np.datetime64("2001-01-01T00:00:00") + np.timedelta64(int((steps - 1) * dt), "s")
    ctl = f"""! synthetic case
<case_name>      synth        ! name
<sim_start>      2001-01-01 00:00:00  ! start
<sim_end>        {str(end).replace('T', ' ')}  ! end
<route_opt>      {route_opt}   ! methods
<doesBasinRoute> 1
<dt_qsim>        {int(dt)}
<dt_ro>          {int(dt)}
<ancil_dir>      {tmp}/
<input_dir>      {tmp}/
<output_dir>     {tmp}/out/
<fname_ntopOld>  ntopo.nc
<fname_qsim>     runoff.nc
<vname_qsim>     RUNOFF
<vname_time>     time
<vname_hruid>    hru_id
<units_qsim>     mm/s
<is_remap>       F
<param_nml>      param.nml
<varname_area>      Basin_Area
<varname_length>    Length
<varname_slope>     Slope
<varname_HRUid>     hruid
<varname_hruSegId>  seg_hru_id
<varname_segId>     seg_id
<varname_downSegId> tosegment
<restart_write>  last
<outputFrequency> 6
<newFileFrequency> {new_file}
{extra}"""
    path = os.path.join(tmp, "synth.control")
    open(path, "w").write(ctl)
    return path


def test_control_namelist_and_units(tmp_path):
    net = m.make_network(40, seed=4)
    path = write_case(str(tmp_path), net, np.zeros((3, 40)), 3600.0)
    ctl = standalone.read_control(path)
    assert ctl["case_name"] == "synth" and ctl["route_opt"] == "2" and ctl["sim_start"] == "2001-01-01 00:00:00"
    assert ctl["varname_downSegId"] == "tosegment" and ctl["outputFrequency"] == "6"
    nml = standalone.read_param_nml(os.path.join(str(tmp_path), "param.nml"))
    assert nml["fshape"] == 2.5 and nml["tscale"] == 86400.0 and nml["velo"] == 1.5 and nml["mann_n"] == 0.01 and nml["wscale"] == 0.001
    assert standalone.unit_factors("mm/s") == (1.0, 1.0e-3) and standalone.unit_factors("m/s") == (1.0, 1.0)
    tc, lc = standalone.unit_factors("mm/day")
    assert lc == 1.0e-3 and tc == 1.0 / 86400.0


def test_network_augmentation_matches_the_generator(tmp_path):
    """Upstream lists, HRU weights, BASAREA/TOTAREA, goodBas and the hydraulic geometry recomputed from the
    raw topology file equal what the synthetic generator (same formulas as process_ntopo.f90) holds."""
    net = m.make_network(700, seed=12, p3=0.05, zero_area_frac=0.1)
    path = write_case(str(tmp_path), net, np.zeros((2, 700)), 3600.0)
    ctl = standalone.read_control(path)
    got, hru_id = standalone.build_network(ctl, standalone.read_param_nml(os.path.join(str(tmp_path), "param.nml")))
    assert np.array_equal(got.downIndex, net.downIndex) and np.array_equal(got.upOffset, net.upOffset)
    assert np.array_equal(got.upIndex, net.upIndex) and np.array_equal(got.upGood, net.upGood)
    assert np.array_equal(got.hruOffset, net.hruOffset) and np.array_equal(got.hruIndex, net.hruIndex)
    pos = net.params["TOTAREA"] > 0      # (the generator gives zero-area reaches a 1 mm wide channel; the reference formula gives 0)
    for k in net.PARAM_ORDER:
        sel = pos if k in ("R_WIDTH", "R_STORAGE") else slice(None)
        if k in ("TOTAREA", "R_WIDTH", "R_STORAGE"):      # the generator adds the upstream areas level by level, the driver in the
            assert np.allclose(got.params[k][sel], net.params[k][sel], rtol=1e-13, atol=0.0), k      # reference's order (pinned in test_oracle_vs_ref.py)
        else:
            assert np.array_equal(got.params[k][sel], net.params[k][sel]), k
    w = got.hruWeight[net.params["BASAREA"][np.repeat(np.arange(net.N), np.diff(net.hruOffset))] > 0]
    assert np.array_equal(w, np.ones_like(w))


@pytest.mark.gpu
def test_run_from_files_equals_api_run(tmp_path, hip_lib):
    from mizuroute_amd import uh as uhmod
    net = m.make_network(1500, seed=13)
    dt, steps = 3600.0, 48
    ro = m.make_runoff(net.H, steps, seed=14, storm_prob=0.03, storm_amp=3e-6)       # m/s
    path = write_case(str(tmp_path), net, ro * 1000.0, dt, route_opt="21")            # file holds mm/s
    out = standalone.run(path, window=16, log=lambda *_: None)
    assert out["steps"] == steps
    frac = uhmod.basin_uh(dt, 2.5, 86400.0)
    uh_off, uhv = uhmod.make_uh(net.params["RLENGTH"], dt, 1.5, 5000.0)
    dom = m.RoutingDomain(net, dt, [m.KWT, m.IRF], frac_future=frac, uh_offset=uh_off, uh=uhv, max_window=16)
    Q = dom.run((ro * 1000.0) * 1.0 * 1.0e-3)          # the driver converts with time_conv * length_conv on the device
    f = netcdf_file(out["history"], "r", mmap=False)
    got = f.variables["KWTroutedRunoff"][:]
    irf = f.variables["IRFroutedRunoff"][:]
    assert got.shape == (steps // 6, net.N)
    want = Q.reshape(steps // 6, 6, 2, net.N).sum(axis=1) / 6.0
    assert np.allclose(got, want[:, 0], rtol=2e-6, atol=1e-12) and np.allclose(irf, want[:, 1], rtol=2e-6, atol=1e-12)
    # records are stamped with the START of their interval, both ends in time_bounds (historyFile.f90:349-373)
    assert np.array_equal(f.variables["time"][:], np.arange(steps // 6) * 6 * dt)
    assert np.array_equal(f.variables["time_bounds"][:], np.stack([np.arange(steps // 6) * 6 * dt, (np.arange(steps // 6) + 1) * 6 * dt], axis=1))
    f.close()
    st = __import__("mizuroute_amd.ncfiles", fromlist=["x"]).read_restart_file(out["restart"])
    assert np.array_equal(st["numWaves"], dom.kwt_state()[0])
    # the restart file is named by the restart time = end of the last step (write_restart_pio.f90:207-253)
    assert os.path.basename(out["restart"]).endswith(".r.2001-01-03-00000.nc"), out["restart"]
    # a window shorter than, and not a divisor of, the output interval must give the same records
    os.makedirs(str(tmp_path / "w4"))
    path4 = write_case(str(tmp_path / "w4"), net, ro * 1000.0, dt, route_opt="21")
    out4 = standalone.run(path4, window=4, log=lambda *_: None)
    f4 = netcdf_file(out4["history"], "r", mmap=False)
    assert np.array_equal(f4.variables["KWTroutedRunoff"][:], got) and np.array_equal(f4.variables["IRFroutedRunoff"][:], irf)
    f4.close()


@pytest.mark.gpu
def test_history_files_per_period_and_restart_pointer(tmp_path, hip_lib):
    """<newFileFrequency> daily: one history file per day, named by the day (write_simoutput_pio.f90:328-380); the other history
    variables through control keys (read_control.f90:239-262); rpointer.rof names the last restart and history files
    (io_rpointfile.f90:23-75)."""
    net = m.make_network(600, seed=15)
    dt, steps = 3600.0, 48
    ro = m.make_runoff(net.H, steps, seed=16, storm_prob=0.03, storm_amp=3e-6)
    path = write_case(str(tmp_path), net, ro * 1000.0, dt, route_opt="21", new_file="daily",
                      extra="<outputInflow> T\n<dlayRunoff> T\n<instRunoff> T\n<KWTvolume> T\n")
    out = standalone.run(path, window=7, log=lambda *_: None)
    names = [os.path.basename(p) for p in out["history_files"]]
    assert names == ["synth.h.2001-01-01-00000.nc", "synth.h.2001-01-02-00000.nc"], names
    tot = 0
    for k, p in enumerate(out["history_files"]):
        f = netcdf_file(p, "r", mmap=False)
        assert f.variables["KWTroutedRunoff"][:].shape == (4, net.N)
        for name in ("basRunoff", "instRunoff", "dlayRunoff", "KWTinflow", "IRFinflow", "KWTvolume"):
            assert name in f.variables, name
        assert np.array_equal(f.variables["time"][:], (np.arange(4) + 4 * k) * 6 * dt)
        assert (f.variables["basRunoff"][:] > 0).all() and f.variables["basRunoff"][:].shape == (4, net.H)
        tot += f.variables["KWTroutedRunoff"][:].shape[0]
        f.close()
    assert tot == steps // 6
    lines = open(out["rpointer"]).read().split()
    assert os.path.basename(out["rpointer"]) == "rpointer.rof" and lines == [out["restart"], out["history_files"][-1]]


@pytest.mark.gpu
def test_run_from_files_with_a_mapping_file(tmp_path, hip_lib, oracle_lib):
    """is_remap = T: the forcing is on 2000 hydrologic-model polygons and reaches the 1200 river-network HRUs
    through a mapping file (remap_1D_runoff on the device); compared with the API run fed the oracle's remap."""
    from mizuroute_amd import uh as uhmod
    from mizuroute_amd.synthetic import make_remap, make_source_runoff
    tmp = str(tmp_path)
    net = m.make_network(1200, seed=21)
    dt, steps, n_src = 3600.0, 36, 2000
    path = write_case(tmp, net, np.zeros((steps, net.N)), dt, route_opt="1")
    mp = make_remap(net.H, n_src, 0, seed=22, missing_frac=0.0)
    sim = np.abs(make_source_runoff(steps, n_src, 0, seed=23)) + 1e-9            # mm/s, no fill values: every HRU gets runoff
    hru_id = (np.arange(net.N) + 50001).astype(np.int32)                          # ids written by write_case
    src_id = mp["src_id"].astype(np.int32)
    g = netcdf_file(os.path.join(tmp, "runoff_hm.nc"), "w", version=2)
    g.createDimension("time", None); g.createDimension("hm", n_src)
    t = g.createVariable("time", "d", ("time",)); t.units = "hours since 2001-01-01 00:00:00"
    h = g.createVariable("hm_id", "i", ("hm",)); h[:] = src_id
    q = g.createVariable("RUNOFF", "d", ("time", "hm"))
    for k in range(steps):
        t[k] = float(k); q[k, :] = sim[k]
    g.close()
    f = netcdf_file(os.path.join(tmp, "map.nc"), "w", version=2)
    f.createDimension("hru", mp["hru_ix"].size); f.createDimension("data", mp["weight"].size)
    rn = np.where(mp["hru_ix"] > 0, hru_id[np.maximum(mp["hru_ix"], 1) - 1], 999999999 % (2 ** 31 - 1)).astype(np.int32)
    v = f.createVariable("RN_hruId", "i", ("hru",)); v[:] = rn
    v = f.createVariable("nOverlaps", "i", ("hru",)); v[:] = np.where(mp["num_qhru"] < 0, 0, mp["num_qhru"]).astype(np.int32)
    v = f.createVariable("weight", "d", ("data",)); v[:] = mp["weight"]
    v = f.createVariable("overlapHruId", "i", ("data",)); v[:] = mp["qhru_id"].astype(np.int32)
    f.close()
    ctl = open(path).read().replace("<is_remap>       F", "<is_remap>       T").replace("<fname_qsim>     runoff.nc", "<fname_qsim>     runoff_hm.nc")
    ctl = ctl.replace("<vname_hruid>    hru_id", "<vname_hruid>    hm_id")
    ctl += "<fname_remap>          map.nc\n<vname_hruid_in_remap> RN_hruId\n<vname_weight>         weight\n<vname_qhruid>         overlapHruId\n<vname_num_qhru>       nOverlaps\n"
    open(path, "w").write(ctl)
    out = standalone.run(path, window=12, log=lambda *_: None)
    # the same through the API: oracle remap (rows whose HRU is not in the network were written with a foreign id)
    mp2 = dict(mp); mp2["num_qhru"] = np.where(mp["num_qhru"] < 0, 0, mp["num_qhru"]).astype(np.int32)
    rc, basin = oracle_lib.remap_runoff(mp2, sim)
    assert rc == 0
    frac = uhmod.basin_uh(dt, 2.5, 86400.0)
    uh_off, uhv = uhmod.make_uh(net.params["RLENGTH"], dt, 1.5, 5000.0)
    dom = m.RoutingDomain(net, dt, [m.IRF], frac_future=frac, uh_offset=uh_off, uh=uhv, max_window=12, length_conv=1.0e-3)
    Q = dom.run(basin)
    got = netcdf_file(out["history"], "r", mmap=False).variables["IRFroutedRunoff"][:]
    want = Q[:, 0].reshape(steps // 6, 6, net.N).mean(axis=1)
    assert np.allclose(got, want, rtol=2e-6, atol=1e-12)


@pytest.mark.gpu
def test_run_from_files_with_gridded_forcing(tmp_path, hip_lib, oracle_lib):
    """Gridded runoff, runoff(time, lat, lon), through a mapping file with i_index / j_index (remap_2D_runoff on the device;
    read_runoff.f90 option 3): the run from files equals the API run fed the oracle's remap of the same grid."""
    from mizuroute_amd import uh as uhmod
    from mizuroute_amd.synthetic import make_remap, make_source_runoff
    tmp = str(tmp_path)
    net = m.make_network(900, seed=31)
    dt, steps, nx, ny = 3600.0, 24, 40, 30
    path = write_case(tmp, net, np.zeros((steps, net.N)), dt, route_opt="1")
    mp = make_remap(net.H, nx, ny, seed=32, missing_frac=0.0)
    sim = np.abs(make_source_runoff(steps, nx, ny, seed=33)) + 1e-9              # [steps, ny * nx] (or [steps, ny, nx]) mm/s
    grid = np.asarray(sim).reshape(steps, ny, nx)
    hru_id = (np.arange(net.N) + 50001).astype(np.int32)
    g = netcdf_file(os.path.join(tmp, "runoff_grid.nc"), "w", version=2)
    g.createDimension("time", None); g.createDimension("lat", ny); g.createDimension("lon", nx)
    t = g.createVariable("time", "d", ("time",)); t.units = "hours since 2001-01-01 00:00:00"
    q = g.createVariable("RUNOFF", "d", ("time", "lat", "lon"))
    for k in range(steps):
        t[k] = float(k); q[k, :, :] = grid[k]
    g.close()
    f = netcdf_file(os.path.join(tmp, "map2d.nc"), "w", version=2)
    f.createDimension("hru", mp["hru_ix"].size); f.createDimension("data", mp["weight"].size)
    rn = np.where(mp["hru_ix"] > 0, hru_id[np.maximum(mp["hru_ix"], 1) - 1], 999999).astype(np.int32)
    v = f.createVariable("RN_hruId", "i", ("hru",)); v[:] = rn
    v = f.createVariable("nOverlaps", "i", ("hru",)); v[:] = np.where(mp["num_qhru"] < 0, 0, mp["num_qhru"]).astype(np.int32)
    v = f.createVariable("weight", "d", ("data",)); v[:] = mp["weight"]
    v = f.createVariable("i_index", "i", ("data",)); v[:] = mp["i_index"].astype(np.int32)
    v = f.createVariable("j_index", "i", ("data",)); v[:] = mp["j_index"].astype(np.int32)
    f.close()
    ctl = open(path).read().replace("<is_remap>       F", "<is_remap>       T").replace("<fname_qsim>     runoff.nc", "<fname_qsim>     runoff_grid.nc")
    ctl += "<fname_remap>          map2d.nc\n<vname_hruid_in_remap> RN_hruId\n<vname_weight>         weight\n<vname_num_qhru>       nOverlaps\n<vname_i_index> i_index\n<vname_j_index> j_index\n"
    open(path, "w").write(ctl)
    out = standalone.run(path, window=10, log=lambda *_: None)
    mp2 = dict(mp); mp2["num_qhru"] = np.where(mp["num_qhru"] < 0, 0, mp["num_qhru"]).astype(np.int32)
    rc, basin = oracle_lib.remap_runoff(mp2, grid)
    assert rc == 0
    frac = uhmod.basin_uh(dt, 2.5, 86400.0)
    uh_off, uhv = uhmod.make_uh(net.params["RLENGTH"], dt, 1.5, 5000.0)
    dom = m.RoutingDomain(net, dt, [m.IRF], frac_future=frac, uh_offset=uh_off, uh=uhv, max_window=12, length_conv=1.0e-3)
    Q = dom.run(basin)
    got = netcdf_file(out["history"], "r", mmap=False).variables["IRFroutedRunoff"][:]
    want = Q[:, 0].reshape(steps // 6, 6, net.N).mean(axis=1)
    assert np.allclose(got, want, rtol=2e-6, atol=1e-12)


def test_time_map_follows_the_reference_rule():
    """timeMap_sim_forc: one record when the step lies inside a forcing interval, overlap fractions otherwise."""
    tm = standalone.time_map
    assert tm(0.0, 3600.0, 3600.0, 10, 1) == ([0], None) and tm(0.0, 3600.0, 3600.0, 10, 10) == ([9], None)
    assert tm(0.0, 3600.0, 10800.0, 4, 2) == ([0], None) and tm(0.0, 3600.0, 10800.0, 4, 4) == ([1], None)     # 3-hourly forcing, hourly steps
    r, f = tm(0.0, 10800.0, 3600.0, 12, 2)                                                                  # hourly forcing, 3-hourly steps
    assert r == [3, 4, 5] and np.allclose(f, [1 / 3, 1 / 3, 1 / 3])
    r, f = tm(1800.0, 3600.0, 3600.0, 10, 1)                                                                # half an hour out of phase
    assert r == [0, 1] and np.allclose(f, [0.5, 0.5])
    with pytest.raises(ValueError):
        tm(0.0, 3600.0, 3600.0, 3, 4)


@pytest.mark.gpu
def test_run_from_files_with_lakes_and_water_management(tmp_path, hip_lib):
    """<is_lake_sim>, <is_flux_wm>, <is_vol_wm> from files: lake flags / parameters in the topology file, evaporation and
    precipitation beside the runoff, abstraction / injection and target volumes in a water-management file on a 3-hourly
    step for a subset of the reaches (get_basin_runoff.f90:106-250) -- against the same run through the API."""
    from mizuroute_amd import uh as uhmod
    from mizuroute_amd.synthetic import make_lakes
    net = m.make_network(1200, seed=31)
    dt, steps, dt_wm = 3600.0, 48, 10800.0
    ro = m.make_runoff(net.H, steps, seed=32, storm_prob=0.03, storm_amp=3e-6)
    lakes = make_lakes(net, steps, dt, seed=6, frac=0.02, input_option=0, target_frac=0.5)
    rng = np.random.default_rng(8)
    others = np.setdiff1d(np.arange(net.N), lakes["reach"] - 1)
    segs = np.concatenate([rng.choice(others, 60, replace=False), lakes["reach"] - 1])
    rng.shuffle(segs)
    nrec = int(steps * dt / dt_wm)
    flux = rng.uniform(-0.02, 0.05, (nrec, segs.size))                       # m3/s taken (+) or injected (-)
    vol3 = lakes["wm_vol"][::3][:nrec][:, segs]                              # target volumes, 3-hourly
    path = write_case(str(tmp_path), net, ro * 1000.0, dt, route_opt="5", lakes=lakes, wm=(segs, flux, vol3, dt_wm),
                      extra="<is_lake_sim> T\n<is_flux_wm> T\n<is_vol_wm> T\n<LakeInputOption> 0\n<calendar> noleap\n<fname_wm> wm.nc\n<vname_flux_wm> abs_inj\n"
                            "<vname_vol_wm> target_vol\n<vname_time_wm> time\n<vname_segid_wm> seg_id\n<dt_wm> 10800\n<floodplain> T\n")
    out = standalone.run(path, window=16, log=lambda *_: None)
    # the same through the API: per-step arrays as the driver builds them (every 3-hourly record covers three steps)
    wm_flux = np.full((steps, net.N), -9999.0); wm_vol = np.zeros((steps, net.N))
    for k in range(steps):
        wm_flux[k, segs] = flux[k // 3]
        wm_vol[k, segs] = vol3[k // 3]
    lk = dict(lakes); lk["wm_vol"] = wm_vol
    frac = uhmod.basin_uh(dt, 2.5, 86400.0)
    uh_off, uhv = uhmod.make_uh(net.params["RLENGTH"], dt, 1.5, 5000.0)
    ctl = standalone.read_control(path)
    net_f, _ = standalone.build_network(ctl, standalone.read_param_nml(os.path.join(str(tmp_path), "param.nml")))
    dom = m.RoutingDomain(net_f, dt, [m.DW], frac_future=frac, uh_offset=uh_off, uh=uhv, max_window=16, lakes=lk, is_flux_wm=1)
    Q = dom.run(ro, wm_flux=wm_flux)
    f = netcdf_file(out["history"], "r", mmap=False)
    got = f.variables["DWroutedRunoff"][:]
    want = Q.reshape(steps // 6, 6, 1, net.N).sum(axis=1)[:, 0] / 6.0
    assert np.allclose(got, want, rtol=5e-6, atol=1e-10), float(np.abs(got - want).max())
    lake0 = lakes["reach"] - 1
    assert np.abs(want[:, lake0]).max() > 0                                   # the lakes do release water
    f.close()
    # and the lakes matter: the same files without <is_lake_sim> give another answer at the lake outlets
    os.makedirs(str(tmp_path / "nolake"))
    p2 = write_case(str(tmp_path / "nolake"), net, ro * 1000.0, dt, route_opt="5")
    o2 = standalone.run(p2, window=16, log=lambda *_: None)
    f2 = netcdf_file(o2["history"], "r", mmap=False)
    assert not np.allclose(f2.variables["DWroutedRunoff"][:][:, lake0], got[:, lake0], rtol=1e-3)
    f2.close()


def test_sort_flux_and_scale_forcing_follow_the_reference_rules():
    """sort_flux (process_remap.f90:268-316) and scale_forcing (get_basin_runoff.f90:375-425) on the host side"""
    ix = np.array([3, -9999, 1], dtype=np.int64)
    assert np.array_equal(standalone.sort_flux(ix, np.array([5.0, 7.0, -2.0]), 4, False), [-2.0, -9999.0, 5.0, -9999.0])
    assert np.array_equal(standalone.sort_flux(ix, np.array([5.0, 7.0, -2.0]), 4, True), [0.0, 0.0, 5.0, 0.0])
    a = np.array([1.0, -9999.0, 2.0])
    assert np.array_equal(standalone.scale_forcing(a, -9999.0, -9999.0), a)                    # neither given: untouched
    assert np.array_equal(standalone.scale_forcing(a, 2.0, -9999.0), [2.0, -9999.0, 4.0])      # missing values stay
    assert np.array_equal(standalone.scale_forcing(a, -9999.0, 0.5), [1.5, -9999.0, 2.5])
    assert standalone.suppressed(0.0, -9999.0) and standalone.suppressed(0.0, 0.0) and not standalone.suppressed(1.0, 0.0)


def test_river_network_subset_mode(tmp_path):
    """<seg_outlet>: the reaches upstream of (and including) a segment and their HRUs go to <fname_ntopNew>; the new file
    builds the same sub-network the generator holds (upstream closure, areas, downstream ids)."""
    net = m.make_network(900, seed=21)
    nup_tot = np.zeros(net.N, int)
    order = np.argsort(-standalone.hops_to_outlet(net.downIndex.astype(np.int64) - 1))
    for i in order:                                   # reaches upstream, leaves first
        d = net.downIndex[i] - 1
        if d >= 0:
            nup_tot[d] += nup_tot[i] + 1
    pick = int(np.argsort(nup_tot)[-20])              # a segment with a sizeable sub-basin that is not the whole network
    path = write_case(str(tmp_path), net, np.zeros((2, net.N)), 3600.0, extra=f"<seg_outlet> {int(net.reachId[pick])}\n<fname_ntopNew> ntopo_sub.nc\n")
    out = standalone.run(path, log=lambda *_: None)
    assert out["reaches"] == nup_tot[pick] + 1 and out["hrus"] == out["reaches"]
    ctl = standalone.read_control(path)
    ctl["fname_ntopOld"] = "ntopo_sub.nc"
    sub, hru_id = standalone.build_network(ctl, standalone.read_param_nml(os.path.join(str(tmp_path), "param.nml")))
    assert sub.N == out["reaches"] and (sub.downIndex == 0).sum() == 1
    assert int(sub.reachId[np.nonzero(sub.downIndex == 0)[0][0]]) == int(net.reachId[pick])
    pos = {int(x): i for i, x in enumerate(net.reachId)}
    full = np.array([pos[int(x)] for x in sub.reachId])
    assert np.array_equal(sub.params["BASAREA"], net.params["BASAREA"][full])
    assert np.allclose(sub.params["TOTAREA"], net.params["TOTAREA"][full], rtol=1e-13, atol=0.0)      # everything upstream came along (the generator sums in another order)
    assert np.array_equal(np.diff(sub.upOffset), np.diff(net.upOffset)[full])


@pytest.mark.gpu
def test_run_from_files_with_a_constituent(tmp_path, hip_lib):
    """<tracer> T: the constituent flux beside the runoff, g/hour units (read_control.f90:476-506), the history variables
    localSolute / soluteFlux / soluteMass against the same run through the API."""
    from mizuroute_amd import uh as uhmod
    net = m.make_network(1000, seed=51)
    dt, steps = 3600.0, 36
    ro = m.make_runoff(net.H, steps, seed=52, storm_prob=0.05, storm_amp=3e-6)
    rng = np.random.default_rng(53)
    sol = rng.uniform(0.0, 5.0, (steps, net.H))                                  # g/hour/m2 in the file
    path = write_case(str(tmp_path), net, ro * 1000.0, dt, route_opt="51", solute=sol,
                      extra="<tracer> T\n<vname_solute> solute\n<units_cc> g/hour\n")
    out = standalone.run(path, window=7, log=lambda *_: None)
    ctl = standalone.read_control(path)
    net_f, _ = standalone.build_network(ctl, standalone.read_param_nml(os.path.join(str(tmp_path), "param.nml")))
    frac = uhmod.basin_uh(dt, 2.5, 86400.0)
    uh_off, uhv = uhmod.make_uh(net.params["RLENGTH"], dt, 1.5, 5000.0)
    dom = m.RoutingDomain(net_f, dt, [m.DW, m.IRF], frac_future=frac, uh_offset=uh_off, uh=uhv, max_window=12)
    dom.set_tracer(sol, time_conv=1.0 / 3600.0, mass_conv=1000.0)
    dom.run(ro)
    F = dom.solute_flux.reshape(steps // 6, 6, 2, net.N).sum(axis=1) / 6.0
    f = netcdf_file(out["history"], "r", mmap=False)
    assert np.allclose(f.variables["soluteFlux"][:], F[:, 0], rtol=5e-6, atol=1e-9)           # the diffusive wave's, under the reference's name
    assert np.allclose(f.variables["IRFsoluteFlux"][:], F[:, 1], rtol=5e-6, atol=1e-9)
    assert np.allclose(f.variables["soluteMass"][:][-1], dom.solute_state(m.DW, 1), rtol=5e-6, atol=1e-6)
    assert f.variables["localSolute"][:].max() > 0 and F.max() > 0
    f.close()


@pytest.mark.gpu
def test_constituent_run_restarted_from_a_file_continues(tmp_path, hip_lib):
    """<tracer> T with <restart_write> last and <fname_state_in>: the restart file must carry the constituent state
    (tfuture, solute_mass), and a run cut in two must end where the uninterrupted run ends."""
    net = m.make_network(800, seed=61)
    dt, steps, cut = 3600.0, 36, 18
    ro = m.make_runoff(net.H, steps, seed=62, storm_prob=0.05, storm_amp=3e-6)
    sol = np.random.default_rng(63).uniform(0.0, 5.0, (steps, net.H))
    base = "<tracer> T\n<vname_solute> solute\n<units_cc> g/hour\n<restart_write> last\n<outputFrequency> 6\n"
    t = lambda k: str(np.datetime64("2001-01-01T00:00:00") + np.timedelta64(int(k * dt), "s")).replace("T", " ")
    whole = write_case(str(tmp_path), net, ro * 1000.0, dt, route_opt="15", solute=sol, extra=base + "<case_name> whole\n")
    out_w = standalone.run(whole, window=7, log=lambda *_: None)
    first = write_case(str(tmp_path), net, ro * 1000.0, dt, route_opt="15", solute=sol,
                       extra=base + f"<case_name> first\n<sim_end> {t(cut - 1)}\n")
    out_1 = standalone.run(first, window=5, log=lambda *_: None)
    f = netcdf_file(out_1["restart"], "r", mmap=False)
    assert "tfuture" in f.variables and any(k.startswith("solute_mass") for k in f.variables), list(f.variables)
    f.close()
    second = write_case(str(tmp_path), net, ro * 1000.0, dt, route_opt="15", solute=sol,
                        extra=base + f"<case_name> second\n<sim_start> {t(cut)}\n<fname_state_in> {os.path.basename(out_1['restart'])}\n")
    out_2 = standalone.run(second, window=7, log=lambda *_: None)
    fw, f2 = netcdf_file(out_w["restart"], "r", mmap=False), netcdf_file(out_2["restart"], "r", mmap=False)
    for k in fw.variables:
        a, b = np.asarray(fw.variables[k].data), np.asarray(f2.variables[k].data)
        assert np.array_equal(a, b), k
    fw.close(); f2.close()


@pytest.mark.gpu
def test_run_from_files_with_gauge_observations(tmp_path, hip_lib):
    """<qmodOption> 1: gauge metadata csv + observation file (sites as character arrays, 3-hourly times), against the same
    run through the API."""
    from mizuroute_amd import uh as uhmod
    from mizuroute_amd.synthetic import make_gauges
    net = m.make_network(900, seed=71)
    dt, steps = 3600.0, 36
    ro = m.make_runoff(net.H, steps, seed=72, storm_prob=0.05, storm_amp=3e-6)
    da = make_gauges(net, steps, n_gauge=30, seed=5, every=3, blend=6, trend=2)
    tmp = str(tmp_path)
    names = [f"G{i:05d}" for i in range(da["gauge_reach"].size)]
    with open(os.path.join(tmp, "gages.csv"), "w") as fp:
        fp.write("gage_id, reach_id, lat, lon\n")
        for nm, r in zip(names, da["gauge_reach"]):
            fp.write(f"{nm}, {int(net.reachId[r - 1]) if r > 0 else 987654321}, 0.0, 0.0\n")
    rec = np.nonzero(da["have"])[0]
    g = netcdf_file(os.path.join(tmp, "obs.nc"), "w", version=2)
    g.createDimension("time", None); g.createDimension("site", len(names)); g.createDimension("strlen", 10)
    t = g.createVariable("time", "d", ("time",)); t.units = "hours since 2001-01-01 00:00:00"
    sv = g.createVariable("site", "c", ("site", "strlen"))
    for i, nm in enumerate(names):
        sv[i, :] = np.array(list(nm.ljust(10)), dtype="S1")
    fl = g.createVariable("flow", "d", ("time", "site")); fl._FillValue = -999.0
    for k, it in enumerate(rec):
        t[k] = it * dt / 3600.0
        fl[k, :] = np.where(np.isnan(da["obs"][it]), -999.0, da["obs"][it])
    g.close()
    path = write_case(tmp, net, ro * 1000.0, dt, route_opt="15",
                      extra="<qmodOption> 1\n<qBlendPeriod> 6\n<QerrTrend> 2\n<gageMetaFile> gages.csv\n<fname_gageObs> obs.nc\n"
                            "<vname_gageFlow> flow\n<vname_gageSite> site\n<vname_gageTime> time\n")
    out = standalone.run(path, window=8, log=lambda *_: None)
    ctl = standalone.read_control(path)
    net_f, _ = standalone.build_network(ctl, standalone.read_param_nml(os.path.join(tmp, "param.nml")))
    frac = uhmod.basin_uh(dt, 2.5, 86400.0)
    uh_off, uhv = uhmod.make_uh(net.params["RLENGTH"], dt, 1.5, 5000.0)
    dom = m.RoutingDomain(net_f, dt, [m.IRF, m.DW], frac_future=frac, uh_offset=uh_off, uh=uhv, max_window=12)
    dom.set_da(da)
    Q = dom.run(ro)
    want = Q.reshape(steps // 6, 6, 2, net.N).sum(axis=1) / 6.0
    f = netcdf_file(out["history"], "r", mmap=False)
    assert np.allclose(f.variables["IRFroutedRunoff"][:], want[:, 0], rtol=5e-6, atol=1e-10)
    assert np.allclose(f.variables["DWroutedRunoff"][:], want[:, 1], rtol=5e-6, atol=1e-10)
    f.close()
    dom2 = m.RoutingDomain(net_f, dt, [m.IRF], frac_future=frac, uh_offset=uh_off, uh=uhv, max_window=12)
    assert np.abs(dom2.run(ro)[:, 0] - Q[:, 0]).max() > 0                     # the observations did change the answer
