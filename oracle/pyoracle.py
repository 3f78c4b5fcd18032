"""ctypes wrapper of oracle/libmzr_oracle.so (the plain-C restatement; see mzr_oracle.h).

TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libmzr_oracle.so")

SUM, IRF, KWT, KW, MC, DW = 0, 1, 2, 3, 4, 5
F_Q, F_VOL0, F_VOL1, F_INFLOW, F_ELE, F_FLOODVOL, F_WB, F_BASIN_QR1, F_BASIN_QR0, F_BASIN_QI = range(10)
WCAP = 32
NMOL = {KW: 20, MC: 2, DW: 20}


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "libmzr_oracle.so"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        ip = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
        dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_int, ip, ip, ip, ip, ip, ip, dp, dp]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_get_kwt_paths.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
        L.orc_config.argtypes = [C.c_void_p, C.c_double, C.c_int, ip, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]
        L.orc_set_uh.argtypes = [C.c_void_p, C.c_int, dp, C.c_void_p, C.c_void_p]
        L.orc_step.argtypes = [C.c_void_p, C.c_double, C.c_double, dp, C.c_void_p]
        L.orc_run.argtypes = [C.c_void_p, C.c_int, C.c_double, dp, C.c_void_p, C.c_void_p]
        L.orc_run_wm.argtypes = [C.c_void_p, C.c_int, C.c_double, dp, dp, C.c_void_p, C.c_void_p]
        L.orc_set_lakes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, ip, ip, dp]
        L.orc_set_lake_target.argtypes = [C.c_void_p, ip, C.c_int, C.c_int, dp]
        L.orc_set_da.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, ip, C.c_int, ip, dp]
        L.orc_set_tracer.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int, dp]
        L.orc_get_solute.argtypes = [C.c_void_p, C.c_int, dp, dp]
        L.orc_hist_get.argtypes = [C.c_void_p, C.c_int, C.c_int, dp]
        L.orc_hist_refresh.argtypes = [C.c_void_p]
        L.orc_run_lake.argtypes = [C.c_void_p, C.c_int, C.c_double, dp, C.c_void_p, dp, dp, ip, C.c_void_p, C.c_void_p]
        L.orc_last_error.restype = C.c_char_p
        L.orc_last_error.argtypes = [C.c_void_p]
        L.orc_get_flux.argtypes = [C.c_void_p, C.c_int, C.c_int, dp]
        L.orc_get_kwt_state.argtypes = [C.c_void_p, ip, dp, dp, dp, ip]
        L.orc_set_kwt_state.argtypes = [C.c_void_p, ip, dp, dp, dp, ip]
        L.orc_get_irf_state.argtypes = [C.c_void_p, dp]
        L.orc_get_mol_state.argtypes = [C.c_void_p, C.c_int, dp]
        L.orc_get_basin_state.argtypes = [C.c_void_p, dp]
        LL = C.POINTER(C.c_longlong)
        L.orc_get_kwt_traffic.argtypes = [C.c_void_p, LL, LL, LL, LL, LL, LL]
        _lib = L
    return _lib


class Oracle:
    """One routing domain on the CPU oracle."""

    def __init__(self, net, dt, methods, frac_future, uh_offset=None, uh=None, does_basin_route=1,
                 hw_drain_point=2, min_length_route=0.0, runoff_min=0.0, is_flux_wm=0):
        L = lib()
        self.net, self.N, self.H = net, net.N, net.H
        self.methods = list(methods)
        self.h = L.orc_create(net.N, net.H, net.downIndex, net.upOffset, net.upIndex, net.upGood,
                              net.hruOffset, net.hruIndex, net.hruWeight, net.param_matrix())
        m = np.asarray(self.methods, dtype=np.int32)
        rc = L.orc_config(self.h, float(dt), len(m), m, does_basin_route, hw_drain_point,
                          float(min_length_route), float(runoff_min), int(is_flux_wm))
        if rc:
            raise RuntimeError(self.error())
        self.ntdh = len(frac_future)
        ff = np.ascontiguousarray(frac_future, dtype=np.float64)
        if uh_offset is not None:
            self.uh_offset = np.ascontiguousarray(uh_offset, dtype=np.int32)
            self.uh = np.ascontiguousarray(uh, dtype=np.float64)
            L.orc_set_uh(self.h, self.ntdh, ff, self.uh_offset.ctypes.data, self.uh.ctypes.data)
        else:
            self.uh_offset = None
            L.orc_set_uh(self.h, self.ntdh, ff, None, None)
        self.dt = float(dt)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_destroy(self.h)
            self.h = None

    def error(self):
        return lib().orc_last_error(self.h).decode()

    def step(self, T0, T1, runoff):
        return lib().orc_step(self.h, float(T0), float(T1), np.ascontiguousarray(runoff, dtype=np.float64), None)

    def run(self, runoff, t_start=0.0, want_vol=False, wm_flux=None):
        runoff = np.ascontiguousarray(runoff, dtype=np.float64)
        n = runoff.shape[0]
        Q = np.zeros((n, len(self.methods), self.N))
        V = np.zeros((n, len(self.methods), self.N)) if want_vol else None
        if wm_flux is not None:
            rc = lib().orc_run_wm(self.h, n, float(t_start), runoff, np.ascontiguousarray(wm_flux, dtype=np.float64),
                                  Q.ctypes.data, V.ctypes.data if want_vol else None)
        else:
            rc = lib().orc_run(self.h, n, float(t_start), runoff, Q.ctypes.data, V.ctypes.data if want_vol else None)
        if rc:
            raise RuntimeError(f"oracle ierr={rc}: {self.error()}")
        return (Q, V) if want_vol else Q

    def set_da(self, da, first_step=0):
        """direct insertion of gauge observations (qmodOption = 1): da = dict(blend, trend, gauge_reach[nGauge] (1-based),
        have[nSteps], obs[nSteps, nGauge]); row 0 belongs to step first_step of the run."""
        c = lambda a, t: np.ascontiguousarray(a, dtype=t)
        self._da = (c(da["gauge_reach"], np.int32), c(da["have"], np.int32), c(da["obs"], np.float64))     # kept alive
        return lib().orc_set_da(self.h, int(da["blend"]), int(da["trend"]), self._da[0].size, self._da[0], int(first_step), self._da[1], self._da[2])

    def run_tracer(self, runoff, solute, t_start=0.0, time_conv=1.0, mass_conv=1.0):
        """tracer = T: routes runoff [nSteps, H] with the basin constituent flux solute [nSteps, H]; returns
        (Q, flux, mass) [nSteps, nRoutes, N] -- REACH_Q, reach_solute_flux, reach_solute_mass(1) after every step."""
        runoff = np.ascontiguousarray(runoff, dtype=np.float64)
        self._solute = np.ascontiguousarray(solute, dtype=np.float64)
        n, R = runoff.shape[0], len(self.methods)
        if lib().orc_set_tracer(self.h, float(time_conv), float(mass_conv), 0, self._solute):
            raise RuntimeError("orc_set_tracer")
        Q, F, M = (np.zeros((n, R, self.N)) for _ in range(3))
        q1 = np.zeros((1, R, self.N))
        for it in range(n):
            rc = lib().orc_run(self.h, 1, float(t_start) + it * self.dt, runoff[it:it + 1], q1.ctypes.data, None)
            if rc:
                raise RuntimeError(f"oracle ierr={rc}: {self.error()}")
            Q[it] = q1[0]
            for ix in range(R):
                f, mm = np.zeros(self.N), np.zeros(self.N)
                lib().orc_get_solute(self.h, ix, f, mm)
                F[it, ix], M[it, ix] = f, mm
        return Q, F, M

    def set_lakes(self, lakes):
        """lakes: dict as oracle.casefile.write_case(lakes=...)."""
        c = lambda a, t: np.ascontiguousarray(a, dtype=t)
        self._lakes = lakes
        rc = self._set_lakes(lakes)
        if rc == 0 and "targ_vol" in lakes:       # target-volume lakes: flags, jump start, REACH_WM_VOL[nSteps][N]
            self._wmvol = c(lakes["wm_vol"], np.float64)
            rc = lib().orc_set_lake_target(self.h, c(lakes["targ_vol"], np.int32), int(lakes.get("vol_jumpstart", 0)), 0, self._wmvol)
        return rc

    def _set_lakes(self, lakes):
        c = lambda a, t: np.ascontiguousarray(a, dtype=t)
        return lib().orc_set_lakes(self.h, int(lakes["input_option"]), int(lakes["calendar_id"]), len(lakes["reach"]),
                                   c(lakes["reach"], np.int32), c(lakes["model_type"], np.int32), c(lakes["par"], np.float64))

    def run_lake(self, runoff, lakes, t_start=0.0, want_vol=False, wm_flux=None):
        runoff = np.ascontiguousarray(runoff, dtype=np.float64)
        n = runoff.shape[0]
        Q = np.zeros((n, len(self.methods), self.N))
        V = np.zeros((n, len(self.methods), self.N)) if want_vol else None
        c = lambda a, t: np.ascontiguousarray(a, dtype=t)
        wm = c(wm_flux, np.float64) if wm_flux is not None else None
        rc = lib().orc_run_lake(self.h, n, float(t_start), runoff, wm.ctypes.data if wm is not None else None,
                                c(lakes["evap"], np.float64), c(lakes["precip"], np.float64), c(lakes["ymd"], np.int32),
                                Q.ctypes.data, V.ctypes.data if want_vol else None)
        if rc:
            raise RuntimeError(f"oracle ierr={rc}: {self.error()}")
        return (Q, V) if want_vol else Q

    def hist(self, route, which):
        """interval mean of a history variable since the last hist_refresh (mzr_oracle.h orc_hist_get)"""
        out = np.zeros(self.H if which == 12 else self.N)
        if lib().orc_hist_get(self.h, route, which, out):
            raise RuntimeError("oracle: no history accumulated")
        return out

    def hist_refresh(self):
        lib().orc_hist_refresh(self.h)

    def flux(self, route, which):
        out = np.zeros(self.N)
        lib().orc_get_flux(self.h, route, which, out)
        return out

    def kwt_state(self):
        nw = np.zeros(self.N, np.int32)
        qf = np.zeros((self.N, WCAP)); ti = np.zeros((self.N, WCAP)); tr = np.zeros((self.N, WCAP))
        rf = np.zeros((self.N, WCAP), np.int32)
        lib().orc_get_kwt_state(self.h, nw, qf, ti, tr, rf)
        return nw, qf, ti, tr, rf

    def irf_state(self):
        out = np.zeros(int(self.uh_offset[-1]))
        lib().orc_get_irf_state(self.h, out)
        return out

    def mol_state(self, method):
        out = np.zeros((self.N, NMOL[method]))
        lib().orc_get_mol_state(self.h, method, out)
        return out

    def basin_state(self):
        out = np.zeros((self.N, self.ntdh))
        lib().orc_get_basin_state(self.h, out)
        return out

    PATHS = ("shock_merges", "merged_leaving", "merged_staying", "exit_time_fixes", "duplicate_times",
             "removes", "removes_over_64", "confluences_over_2", "tstart_fixes")

    def kwt_paths(self):
        """How often the less common branches of kwt_rch ran since creation (coverage evidence for the parity tests)."""
        v = (C.c_longlong * 9)()
        lib().orc_get_kwt_paths(self.h, v)
        return dict(zip(self.PATHS, list(v)))

    def kwt_traffic(self):
        v = [C.c_longlong(0) for _ in range(6)]
        lib().orc_get_kwt_traffic(self.h, *[C.byref(x) for x in v])
        return dict(zip(("w_in", "w_up", "w_out", "n_head", "n_route", "n_edges"), [x.value for x in v]))


def _ip(a):
    return np.ascontiguousarray(a, dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int))


def remap_runoff(mp, sim):
    """process_remap.f90 remap_runoff over a series: sim [nSteps, n1] (1-D) or [nSteps, n2, n1] (grid);
    returns (ierr, basinRunoff[nSteps, H]) with the array carried from step to step like the reference's."""
    L = lib()
    dp = C.POINTER(C.c_double); lp = C.POINTER(C.c_longlong)
    sim = np.ascontiguousarray(sim, dtype=np.float64)
    H = int(mp["H"]); nSteps = sim.shape[0]
    out = np.zeros((nSteps, H)); cur = np.zeros(H)
    w = np.ascontiguousarray(mp["weight"], dtype=np.float64)
    hix, num = _ip(mp["hru_ix"]), _ip(mp["num_qhru"])
    rc = 0
    for t in range(nSteps):
        if sim.ndim == 2:
            qid = np.ascontiguousarray(mp["qhru_id"], dtype=np.int64); sid = np.ascontiguousarray(mp["src_id"], dtype=np.int64)
            L.orc_remap_1d.restype = C.c_int
            r = L.orc_remap_1d(C.c_int(len(mp["hru_ix"])), hix, num, _ip(mp["qhru_ix"]), qid.ctypes.data_as(lp), sid.ctypes.data_as(lp),
                               w.ctypes.data_as(dp), sim[t].ctypes.data_as(dp), cur.ctypes.data_as(dp))
        else:
            r = L.orc_remap_2d(C.c_int(len(mp["hru_ix"])), hix, num, _ip(mp["i_index"]), _ip(mp["j_index"]), w.ctypes.data_as(dp),
                               C.c_int(sim.shape[2]), C.c_int(sim.shape[1]), sim[t].ctypes.data_as(dp), cur.ctypes.data_as(dp))
        rc = rc or r
        out[t] = cur
    return rc, out


def sort_flux(ix_in, flux, H, remove_negatives=True):
    """process_remap.f90 sort_flux over a series: flux [nSteps, nIn] -> [nSteps, H]."""
    L = lib()
    dp = C.POINTER(C.c_double)
    flux = np.ascontiguousarray(flux, dtype=np.float64)
    out = np.zeros((flux.shape[0], H))
    for t in range(flux.shape[0]):
        L.orc_sort_flux(C.c_int(flux.shape[1]), _ip(ix_in), flux[t].ctypes.data_as(dp), C.c_int(int(remove_negatives)), C.c_int(H), out[t].ctypes.data_as(dp))
    return out
