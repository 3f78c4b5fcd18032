"""Restart and history files in the reference's NetCDF layout (host side of the hot path).

Restart: variables, dimensions and padding follow write_restart_pio.f90:259-392,723-1290 and are
read back as read_restart.f90:17-742 does, so that a file written here restarts the reference and
vice versa.  History: <case>.h.yyyy-mm-dd-sssss.nc with reachID, time and the time-MEAN discharge
of every active method as float32 (histVars_data.f90:229-231,273-281, historyFile.f90:434-534,
docs/source/users_guide/Output_files.rst:14-48).

The reference writes through ParallelIO with `pio_netcdf_format = "64bit_offset"` (public_var.f90:54),
i.e. classic NetCDF CDF-2, which scipy.io.netcdf_file(version=2) reads and writes; no netCDF library
is needed.  A Fortran array declared (seg, wave) is the NetCDF variable with dimensions (wave, seg).
The reference's own writer cannot run in this image (it needs PIO), so the FILE layout is taken from
its source, not pinned by execution; the STATE that travels through a file is pinned by the
round-trip tests (tests/test_ncfiles.py, tests/test_gpu_parity.py::test_restart_continues_bit_exact).
"""
from __future__ import annotations

import numpy as np
from scipy.io import netcdf_file

from . import api

REAL_MISSING, INT_MISSING = -9999.0, -9999           # public_var.f90:44-45
MAXQPAR = 20                                         # public_var.f90:36, dimension "wave"
_SUFFIX = {api.IRF: "irf", api.KWT: "kwt", api.KW: "kw", api.MC: "mc", api.DW: "dw"}
_MOLDIM = {api.KW: "mol_kw", api.MC: "mol_mc", api.DW: "mol_dw"}
HIST_Q = {api.SUM: "sumUpstreamRunoff", api.IRF: "IRFroutedRunoff", api.KWT: "KWTroutedRunoff",
          api.KW: "KWroutedRunoff", api.MC: "MCroutedRunoff", api.DW: "DWroutedRunoff"}     # popMetadat.f90:240-245
HIST_VOL = {api.IRF: "IRFvolume", api.KWT: "KWTvolume", api.KW: "KWvolume", api.MC: "MCvolume", api.DW: "DWvolume"}
_PFX = {api.IRF: "IRF", api.KWT: "KWT", api.KW: "KW", api.MC: "MC", api.DW: "DW"}
HIST_HEIGHT = {m: p + "height" for m, p in _PFX.items()}                 # popMetadat.f90:251-255
HIST_FLOOD = {m: ("IRf" if m == api.IRF else p) + "floodVolume" for m, p in _PFX.items()}     # :256-260 (the reference spells IRffloodVolume)
HIST_INFLOW = {m: p + "inflow" for m, p in _PFX.items()}                 # :261-265


def _var(f, name, typ, dims, data, **att):
    v = f.createVariable(name, typ, dims)
    for k, a in att.items():
        setattr(v, k, a)
    if dims:
        v[:] = data
    else:
        v.data[...] = data          # (netcdf_variable.assignValue indexes a 0-d array with [:])
    return v


def collect_state(dom) -> dict:
    """Everything read_restart.f90 restores, in caller order, as plain arrays."""
    st = {"basin_q": dom.flux(dom.methods[0], api.F_BASIN_QR1)}
    if dom.does_basin_route == 1:
        st["qfuture"] = dom.basin_state()
    for m in dom.methods:
        if m == api.SUM:
            continue
        st[f"volume_{_SUFFIX[m]}"] = dom.flux(m, api.F_VOL1)
        if m == api.IRF:
            st["irf_qfuture"] = dom.irf_state()
        elif m == api.KWT:
            nw, qf, ti, tr, rf = dom.kwt_state()
            st.update(numWaves=nw, qwave=qf, tentry=ti, texit=tr, routed=rf)
        else:
            st[f"q_sub_{_SUFFIX[m]}"] = dom.mol_state(m)
    if getattr(dom, "tracer_on", False):      # constituent: tfuture(seg, tdh) and solute_mass(seg) (write_restart_pio.f90:941-971,1292-)
        ts = dom.tracer_state()
        if "tfuture" in ts:
            st["tfuture"] = ts["tfuture"]
        for m, a in ts["mass"].items():
            st["solute_mass" if m == api.DW else f"solute_mass_{_SUFFIX[m]}"] = a      # the reference keeps the diffusive wave's only
    return st


def apply_state(dom, st: dict):
    dom.set_basin_state(st.get("qfuture"), st["basin_q"])
    for m in dom.methods:
        if m == api.SUM:
            continue
        dom.set_volume(m, st[f"volume_{_SUFFIX[m]}"])
        if m == api.IRF:
            dom.set_irf_state(st["irf_qfuture"])
        elif m == api.KWT:
            dom.set_kwt_state(st["numWaves"], st["qwave"], st["tentry"], st["texit"], st["routed"])
        else:
            dom.set_mol_state(m, st[f"q_sub_{_SUFFIX[m]}"])
    if getattr(dom, "tracer_on", False) and any(k.startswith("solute_mass") or k == "tfuture" for k in st):
        ts = {"mass": {}}
        if "tfuture" in st:
            ts["tfuture"] = st["tfuture"]
        for m in dom.methods:
            key = "solute_mass" if m == api.DW else f"solute_mass_{_SUFFIX.get(m, '')}"
            if m != api.SUM and key in st:
                ts["mass"][m] = st[key]
        dom.set_tracer_state(ts)


def write_restart_file(path, st: dict, reach_id, uh_offset, time_bound, restart_time=0.0,
                       time_units="seconds since 1970-01-01 00:00:00", calendar="standard"):
    """st: dict from collect_state (KWT arrays padded to api.WCAP columns)."""
    N = len(reach_id)
    f = netcdf_file(path, "w", version=2)
    f.createDimension("seg", N)
    f.createDimension("tbound", 2)
    _var(f, "nNodes", "i", (), 1, long_name="Number of MPI tasks", units="-")
    _var(f, "reachID", "i", ("seg",), np.asarray(reach_id, np.int32), long_name="reach ID", units="-")
    _var(f, "restart_time", "d", (), float(restart_time), long_name="resatart time", units=time_units, calendar=calendar)
    _var(f, "time_bound", "f", ("tbound",), np.asarray(time_bound, np.float32), long_name="time bound at last time step", units="sec")
    _var(f, "basin_q", "d", ("seg",), st["basin_q"], long_name="basin routed flow", units="m3/s")
    if "qfuture" in st:
        f.createDimension("tdh", st["qfuture"].shape[1])
        _var(f, "qfuture", "d", ("tdh", "seg"), st["qfuture"].T, long_name="future flow series", units="m3/s")
    for key in sorted(k for k in st if k.startswith("volume_")):
        _var(f, key, "d", ("seg",), st[key], long_name="volume in reach/lake", units="m3")
    if "tfuture" in st:
        if "tdh" not in f.dimensions:
            f.createDimension("tdh", st["tfuture"].shape[1])
        _var(f, "tfuture", "d", ("tdh", "seg"), st["tfuture"].T, long_name="future tracer mass series", units="mg/s")
    for key in sorted(k for k in st if k.startswith("solute_mass")):
        _var(f, key, "d", ("seg",), st[key], long_name="mass in reach/lake", units="mg")
    if "irf_qfuture" in st:
        off = np.asarray(uh_offset, np.int64)
        nq = np.diff(off).astype(np.int32)
        pad = np.full((N, int(nq.max())), REAL_MISSING)                    # write_restart_pio.f90:1008-1009
        for e in range(N):
            pad[e, :nq[e]] = st["irf_qfuture"][off[e]:off[e + 1]]
        f.createDimension("tdh_irf", pad.shape[1])
        _var(f, "numQF", "i", ("seg",), nq, long_name="number of future q time steps in a reach", units="-")
        _var(f, "irf_qfuture", "d", ("tdh_irf", "seg"), pad.T, long_name="future flow series", units="m3/s")
    if "numWaves" in st:
        nw = np.asarray(st["numWaves"], np.int32)
        f.createDimension("wave", MAXQPAR)
        live = np.arange(MAXQPAR)[None, :] < nw[:, None]
        _var(f, "numWaves", "i", ("seg",), nw, long_name="number of waves in a reach", units="-")
        for name, key, unit, desc in (("tentry", "tentry", "s", "time when a wave enters a segment"),
                                      ("texit", "texit", "s", "time when a wave is expected to exit a segment"),
                                      ("qwave", "qwave", "m2/s", "flow of a wave"),
                                      ("qwave_mod", None, "m2/s", "modified flow of a wave")):
            a = st[key][:, :MAXQPAR] if key else np.full((N, MAXQPAR), REAL_MISSING)   # QM is always -9999 on this path
            _var(f, name, "d", ("wave", "seg"), np.where(live, a, REAL_MISSING).T, long_name=desc, units=unit)
        _var(f, "routed", "i", ("wave", "seg"), np.where(live, st["routed"][:, :MAXQPAR], INT_MISSING).astype(np.int32).T,
             long_name="routing flag", units="-")
    for m, dim in _MOLDIM.items():
        key = f"q_sub_{_SUFFIX[m]}"
        if key in st:
            f.createDimension(dim, st[key].shape[1])
            _var(f, key, "d", (dim, "seg"), st[key].T, long_name="flow at computational molecule", units="m3/s")
    f.close()


def read_restart_file(path) -> dict:
    f = netcdf_file(path, "r", mmap=False)
    v = f.variables
    st = {"reachID": v["reachID"][:].copy(), "time_bound": v["time_bound"][:].astype(np.float64), "basin_q": v["basin_q"][:].copy()}
    N = st["reachID"].size
    if "qfuture" in v:
        st["qfuture"] = v["qfuture"][:].T.copy()
    if "tfuture" in v:
        st["tfuture"] = v["tfuture"][:].T.copy()
    for key in v:
        if key.startswith("solute_mass"):
            st[key] = v[key][:].copy()
    for k in v:
        if k.startswith("volume_") or k.startswith("q_sub_"):
            st[k] = v[k][:].T.copy() if v[k][:].ndim == 2 else v[k][:].copy()
    if "irf_qfuture" in v:
        nq = v["numQF"][:]
        pad = v["irf_qfuture"][:].T
        st["numQF"] = nq.copy()
        st["irf_qfuture"] = np.concatenate([pad[e, :nq[e]] for e in range(N)]) if N else np.zeros(0)
    if "numWaves" in v:
        nw = v["numWaves"][:].copy()
        st["numWaves"] = nw
        for k in ("qwave", "tentry", "texit"):
            a = np.full((N, api.WCAP), REAL_MISSING)
            a[:, :MAXQPAR] = v[k][:].T
            st[k] = a
        r = np.zeros((N, api.WCAP), np.int32)
        r[:, :MAXQPAR] = np.where(v["routed"][:].T == 1, 1, 0)          # read_restart.f90:464-465
        st["routed"] = r
    f.close()
    # NetCDF classic is big-endian: hand back native arrays
    return {k: (a.astype(a.dtype.newbyteorder("=")) if isinstance(a, np.ndarray) else a) for k, a in st.items()}


def write_restart(path, dom, reach_id, time_bound, **kw):
    write_restart_file(path, collect_state(dom), reach_id, getattr(dom, "uh_offset", None), time_bound, **kw)


def read_restart(path, dom):
    """Restore a freshly initialised domain from a restart file; returns time_bound (TSEC of the last step)."""
    st = read_restart_file(path)
    if st["reachID"].size != dom.N:
        raise ValueError("restart file and domain differ in the number of reaches")
    apply_state(dom, st)
    return st["time_bound"]


class HistoryWriter:
    """<case>.h.*.nc: one record per output interval, time-mean discharge (and optionally the last
    volume) per active method, float32, dimensions (time, seg) (historyFile.f90:434-534).  `time` is the START
    of the aggregated interval plus histTimeStamp_offset and `time_bounds` holds both ends
    (historyFile.f90:349-373, histVars_data.f90:179-183)."""

    def __init__(self, path, reach_id, methods, time_units="seconds since 1970-01-01 00:00:00", calendar="standard", volumes=False,
                 inflow=False, height=False, runoff=False, hru_id=None, solute=False):
        """volumes: <M>volume (last value); inflow: <M>inflow (outputInflow); height: <M>height and <M>floodVolume (floodplain);
        runoff: instRunoff, dlayRunoff and basRunoff(hru) -- the last three need the domain built with history=H_* flags;
        solute: localSolute, and per method soluteFlux (interval mean) / soluteMass (last value) -- the reference writes the
        diffusive wave's under these names (popMetadat.f90:266-268), the other methods' get the method's prefix."""
        self.f = netcdf_file(path, "w", version=2)
        self.f.createDimension("time", None)
        self.f.createDimension("seg", len(reach_id))
        self.f.createDimension("tbound", 2)
        self.methods, self.n = list(methods), 0
        t = self.f.createVariable("time", "d", ("time",)); t.units = time_units; t.calendar = calendar; t.long_name = "time"; t.bounds = "time_bounds"
        tb = self.f.createVariable("time_bounds", "d", ("time", "tbound")); tb.units = time_units; tb.calendar = calendar; tb.long_name = "time interval endpoints"
        _var(self.f, "reachID", "i", ("seg",), np.asarray(reach_id, np.int32), long_name="reach ID", units="-")
        self.vars = {}
        # runoff: True = all three, or the names wanted (the reference's defaults: basRunoff on, instRunoff and dlayRunoff off,
        # read_control.f90:239-241, popMetadat.f90:238-239)
        want = ("basRunoff", "instRunoff", "dlayRunoff") if runoff is True else tuple(runoff or ())
        self.runoff = want
        if "basRunoff" in want:
            if hru_id is None:
                raise ValueError("basRunoff needs hru_id")
            self.f.createDimension("hru", len(hru_id))
            _var(self.f, "basinID", "i", ("hru",), np.asarray(hru_id, np.int32), long_name="basin ID", units="-")
        for name, dims, unit in (("basRunoff", ("time", "hru"), "m/s"), ("instRunoff", ("time", "seg"), "m3/s"), ("dlayRunoff", ("time", "seg"), "m3/s")):
            if name in want:
                v = self.f.createVariable(name, "f", dims); v.units = unit
        for m in self.methods:
            q = self.f.createVariable(HIST_Q[m], "f", ("time", "seg")); q.units = "m3/s"
            self.vars[(m, api.M_Q)] = q
            if volumes and m in HIST_VOL:
                w = self.f.createVariable(HIST_VOL[m], "f", ("time", "seg")); w.units = "m3"
                self.vars[(m, "v")] = w
            if inflow and m in HIST_INFLOW:
                w = self.f.createVariable(HIST_INFLOW[m], "f", ("time", "seg")); w.units = "m3/s"
                self.vars[(m, api.M_INFLOW)] = w
            if height and m in HIST_HEIGHT:
                w = self.f.createVariable(HIST_HEIGHT[m], "f", ("time", "seg")); w.units = "m"
                self.vars[(m, api.M_HEIGHT)] = w
                w = self.f.createVariable(HIST_FLOOD[m], "f", ("time", "seg")); w.units = "m3"
                self.vars[(m, api.M_FLOODVOL)] = w

        self.solute = solute
        if solute:
            v = self.f.createVariable("localSolute", "f", ("time", "seg")); v.units = "mg/s"
            for m in self.methods:
                if m == api.SUM:
                    continue
                pre = "" if m == api.DW else HIST_Q[m].replace("routedRunoff", "")
                v = self.f.createVariable(pre + "soluteFlux", "f", ("time", "seg")); v.units = "mg/s"
                v = self.f.createVariable(pre + "soluteMass", "f", ("time", "seg")); v.units = "mg"

    def append_solute(self, local_mean, flux_mean, mass):
        """constituent of the record append() writes next: local_mean [seg], flux_mean / mass {method: [seg]}"""
        self.f.variables["localSolute"][self.n, :] = np.asarray(local_mean, np.float32)
        for m in flux_mean:
            pre = "" if m == api.DW else HIST_Q[m].replace("routedRunoff", "")
            self.f.variables[pre + "soluteFlux"][self.n, :] = np.asarray(flux_mean[m], np.float32)
            self.f.variables[pre + "soluteMass"][self.n, :] = np.asarray(mass[m], np.float32)

    def append(self, t_begin, t_end, dom, stamp_offset=0.0):
        """Write the means accumulated on the device over [t_begin, t_end] and reset them (histVars finalize / refresh)."""
        self.f.variables["time"][self.n] = t_begin + stamp_offset
        self.f.variables["time_bounds"][self.n, :] = (t_begin, t_end)
        for (m, which), var in self.vars.items():
            if which == "v":
                var[self.n, :] = dom.flux(m, api.F_VOL1).astype(np.float32)
            elif which == api.M_Q and not hasattr(dom, "mean"):
                var[self.n, :] = dom.mean_q(m, reset=True).astype(np.float32)        # stand-ins in tests
            else:
                var[self.n, :] = dom.mean(m, which).astype(np.float32)
        for name, which in (("basRunoff", api.M_BAS_RUNOFF), ("instRunoff", api.M_INST_RUNOFF), ("dlayRunoff", api.M_DLAY_RUNOFF)):
            if name in self.runoff:
                self.f.variables[name][self.n, :] = dom.mean(0, which).astype(np.float32)
        if hasattr(dom, "reset_means"):
            dom.reset_means()
        self.n += 1

    def close(self):
        self.f.close()
