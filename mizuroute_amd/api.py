"""Python host side above the C-ABI of libmzr_hip.so (include/mzr.h).

Mirrors the call sequence of the reference's driver around its hot path:
  init_route_method / put_data_struct / init_state_data -> RoutingDomain(...)
  main_route(basinRunoff, ..., RCHFLX, RCHSTA)           -> RoutingDomain.step(T0, T1, runoff)
  (new) a whole window of steps, time-skewed on device   -> RoutingDomain.run(runoff[nSteps, H])
Error behaviour follows the reference: an integer ierr plus a message chain
(main_route.f90:93,159); here a non-zero ierr raises MzrError(ierr, message).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

SUM, IRF, KWT, KW, MC, DW = 0, 1, 2, 3, 4, 5
F_Q, F_VOL0, F_VOL1, F_INFLOW, F_ELE, F_FLOODVOL, F_WB, F_BASIN_QR1, F_BASIN_QR0, F_BASIN_QI = range(10)
H_INFLOW, H_HEIGHT, H_RUNOFF = 1, 2, 4
M_Q, M_INFLOW, M_HEIGHT, M_FLOODVOL, M_INST_RUNOFF, M_DLAY_RUNOFF, M_BAS_RUNOFF = 0, 1, 2, 3, 10, 11, 12
WCAP = 32
NMOL = {KW: 20, MC: 2, DW: 20}

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path() -> str:
    # MZR_LIB: an experiment build of the same library (tools/build_variant.sh: instruction-accounting, timing and debugging builds)
    return os.environ.get("MZR_LIB") or os.path.join(_HERE, "lib", "libmzr_hip.so")


def build_library(verbose: bool = False) -> str:
    """Compile the HIP kernels + C-ABI for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    subprocess.run(cmd, check=True, stdout=None if verbose else subprocess.DEVNULL)
    return lib_path()


class MzrConfig(C.Structure):
    _fields_ = [("dt", C.c_double), ("nRoutes", C.c_int), ("routeMethods", C.c_int * 6),
                ("doesBasinRoute", C.c_int), ("hw_drain_point", C.c_int),
                ("min_length_route", C.c_double), ("runoffMin", C.c_double), ("negRunoffTol", C.c_double),
                ("time_conv", C.c_double), ("length_conv", C.c_double), ("maxWindow", C.c_int), ("device", C.c_int),
                ("is_flux_wm", C.c_int), ("lakeMemoryPerMethod", C.c_int), ("mcTailTol", C.c_double), ("sweepShare", C.c_double),
                ("stepBatch", C.c_int), ("sweepPriority", C.c_int), ("sweepTimeout", C.c_double)]


_lib = None

EXPORTS = ["mzr_default_config", "mzr_create", "mzr_destroy", "mzr_last_error", "mzr_set_network",
           "mzr_set_param", "mzr_set_uh", "mzr_set_frac_future", "mzr_init_state", "mzr_step", "mzr_run",
           "mzr_run_dev", "mzr_sync", "mzr_get_flux", "mzr_get_window_q", "mzr_get_mean_q",
           "mzr_get_kwt_state", "mzr_set_kwt_state", "mzr_get_irf_state", "mzr_get_mol_state",
           "mzr_get_basin_state", "mzr_get_schedule", "mzr_set_profiling", "mzr_get_timing", "mzr_get_timing_range",
           "mzr_get_kwt_traffic", "mzr_set_boundary", "mzr_boundary_size", "mzr_export_boundary_dev", "mzr_export_boundary_prev_dev", "mzr_get_export_lag", "mzr_wait_export", "mzr_wait_import",
           "mzr_import_boundary_dev", "mzr_set_wm_flux", "mzr_set_lakes", "mzr_set_lake_forcing", "mzr_set_lake_forcing_dev", "mzr_set_lake_target", "mzr_set_wm_vol", "mzr_get_global_wb", "mzr_set_da", "mzr_set_obs", "mzr_set_tracer", "mzr_set_solute", "mzr_get_solute", "mzr_get_window_solute", "mzr_get_tracer_state", "mzr_set_tracer_state",
           "mzr_set_remap", "mzr_set_sort_map", "mzr_remap_runoff_dev", "mzr_run_src_dev",
           "mzr_set_irf_state", "mzr_set_mol_state", "mzr_set_basin_state", "mzr_set_volume",
           "mzr_get_sweep_info", "mzr_run_async", "mzr_run_async_f32", "mzr_comm_unique_id", "mzr_comm_init", "mzr_comm_send", "mzr_comm_recv",
           "mzr_comm_recv_many", "mzr_comm_destroy", "mzr_comm_last_error", "mzr_comm_sync", "mzr_set_history", "mzr_get_mean",
           "mzr_reset_means", "mzr_get_sweep_arrivals", "mzr_get_sweep_retries", "mzr_get_sweep_clock"]


def load_library():
    """dlopen libmzr_hip.so; fails loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(mizuroute_amd has no CPU fallback)")
    # PyTorch bundles its own HIP runtime; when both live in one process torch must initialise its
    # device context first or it no longer finds the GPU.  (Pure C/Fortran hosts are unaffected.)
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    L = C.CDLL(path)
    ip = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
    dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
    vp, ci, cd = C.c_void_p, C.c_int, C.c_double
    L.mzr_default_config.argtypes = [C.POINTER(MzrConfig)]
    L.mzr_default_config.restype = None
    L.mzr_create.argtypes = [C.POINTER(MzrConfig), C.POINTER(vp)]
    L.mzr_destroy.argtypes = [vp]
    L.mzr_last_error.argtypes = [vp, C.c_char_p, ci]
    L.mzr_set_network.argtypes = [vp, ci, ci, ip, ip, ip, vp, ip, ip, dp, vp]
    L.mzr_set_param.argtypes = [vp, C.c_char_p, dp]
    L.mzr_set_uh.argtypes = [vp, ip, dp]
    L.mzr_set_frac_future.argtypes = [vp, ci, dp]
    L.mzr_init_state.argtypes = [vp]
    L.mzr_set_boundary.argtypes = [vp, ci, ip, ci, ip, ip]
    L.mzr_boundary_size.argtypes = [vp, ci, ci]
    L.mzr_boundary_size.restype = C.c_longlong
    L.mzr_export_boundary_dev.argtypes = [vp, vp]
    L.mzr_export_boundary_prev_dev.argtypes = [vp, vp]
    L.mzr_get_export_lag.argtypes = [vp]
    L.mzr_wait_export.argtypes = [vp]
    L.mzr_wait_import.argtypes = [vp]
    L.mzr_import_boundary_dev.argtypes = [vp, ci, vp, ci, ci]
    L.mzr_step.argtypes = [vp, cd, cd, dp]
    L.mzr_run.argtypes = [vp, ci, cd, dp]
    L.mzr_run_dev.argtypes = [vp, ci, cd, vp]
    L.mzr_run_async.argtypes = [vp, ci, cd, vp]
    L.mzr_run_async_f32.argtypes = [vp, ci, cd, vp]
    L.mzr_comm_unique_id.argtypes = [C.c_char_p]
    L.mzr_comm_init.argtypes = [ci, ci, C.c_char_p, ci, C.POINTER(vp)]
    L.mzr_comm_send.argtypes = [vp, vp, vp, C.c_longlong, ci]
    L.mzr_comm_recv.argtypes = [vp, vp, vp, C.c_longlong, ci]
    L.mzr_comm_recv_many.argtypes = [vp, vp, ci, C.POINTER(vp), C.POINTER(C.c_longlong), C.POINTER(ci)]
    L.mzr_comm_destroy.argtypes = [vp]
    L.mzr_set_history.argtypes = [vp, ci]
    L.mzr_get_mean.argtypes = [vp, ci, ci, dp]
    L.mzr_reset_means.argtypes = [vp]
    L.mzr_comm_sync.argtypes = [vp]
    L.mzr_comm_last_error.argtypes = [C.c_char_p, ci]
    L.mzr_sync.argtypes = [vp]
    L.mzr_set_wm_flux.argtypes = [vp, ci, dp]
    L.mzr_get_global_wb.argtypes = [vp, ci, dp]
    L.mzr_set_da.argtypes = [vp, ci, ci, ci, ip]
    L.mzr_set_obs.argtypes = [vp, ci, ip, dp]
    L.mzr_set_tracer.argtypes = [vp, ci, C.c_double, C.c_double]
    L.mzr_set_solute.argtypes = [vp, ci, dp]
    L.mzr_get_solute.argtypes = [vp, ci, ci, dp]
    L.mzr_get_window_solute.argtypes = [vp, ci, dp]
    L.mzr_get_tracer_state.argtypes = [vp, ci, vp, vp]
    L.mzr_set_tracer_state.argtypes = [vp, ci, vp, vp]
    L.mzr_set_irf_state.argtypes = [vp, dp]
    L.mzr_set_mol_state.argtypes = [vp, ci, dp]
    L.mzr_set_basin_state.argtypes = [vp, vp, vp]
    L.mzr_set_volume.argtypes = [vp, ci, dp]
    L.mzr_set_remap.argtypes = [vp, ci, ci, ip, ip, ci, vp, vp, vp, dp, ci, ci, vp, vp]
    L.mzr_set_sort_map.argtypes = [vp, ci, ip, ci]
    L.mzr_remap_runoff_dev.argtypes = [vp, ci, vp, vp]
    L.mzr_run_src_dev.argtypes = [vp, ci, cd, vp]
    L.mzr_set_lakes.argtypes = [vp, ci, ci, ci, ip, ip, dp]
    L.mzr_set_lake_target.argtypes = [vp, ip, ci]
    L.mzr_set_wm_vol.argtypes = [vp, ci, dp]
    L.mzr_set_lake_forcing.argtypes = [vp, ci, vp, vp, ip, ip, ip]
    L.mzr_set_lake_forcing_dev.argtypes = [vp, ci, vp, vp, ip, ip, ip]
    L.mzr_get_flux.argtypes = [vp, ci, ci, dp]
    L.mzr_get_window_q.argtypes = [vp, ci, dp]
    L.mzr_get_mean_q.argtypes = [vp, ci, dp, ci]
    L.mzr_get_kwt_state.argtypes = [vp, ip, dp, dp, dp, ip]
    L.mzr_set_kwt_state.argtypes = [vp, ip, dp, dp, dp, ip]
    L.mzr_get_irf_state.argtypes = [vp, dp]
    L.mzr_get_mol_state.argtypes = [vp, ci, dp]
    L.mzr_get_basin_state.argtypes = [vp, dp]
    L.mzr_get_schedule.argtypes = [vp, C.POINTER(ci), C.POINTER(ci)]
    L.mzr_get_sweep_info.argtypes = [vp, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]
    L.mzr_get_sweep_arrivals.argtypes = [vp, C.POINTER(ci), C.POINTER(ci), C.POINTER(C.c_longlong)]
    L.mzr_get_sweep_clock.argtypes = [vp, ci, C.POINTER(C.c_double), C.POINTER(ci), ci]
    L.mzr_set_profiling.argtypes = [vp, ci]
    LL = C.POINTER(C.c_longlong)
    L.mzr_get_timing.argtypes = [vp, ci, LL, C.POINTER(cd), LL, ci]
    L.mzr_get_timing_range.argtypes = [vp, ci, C.POINTER(cd), C.POINTER(cd), ci]
    L.mzr_get_kwt_traffic.argtypes = [vp, LL, LL, LL, LL, LL, LL, ci]
    _lib = L
    return L


class MzrError(RuntimeError):
    def __init__(self, ierr, message):
        super().__init__(f"ierr={ierr}: {message}")
        self.ierr, self.message = ierr, message


class RoutingDomain:
    """One routing domain (a whole network, or one sub-basin partition) resident on one GPU."""

    def __init__(self, net, dt, methods, frac_future=None, uh_offset=None, uh=None, does_basin_route=1,
                 hw_drain_point=2, min_length_route=0.0, runoff_min=0.0, max_window=64, device=0,
                 export_reaches=None, halo_reaches=None, halo_good=None, is_flux_wm=0, lakes=None,
                 time_conv=1.0, length_conv=1.0, history=0, sweep_share=1.0, step_batch=1, sweep_priority=0, lake_memory_per_method=0):
        L = load_library()
        self.L, self.net, self.N, self.H = L, net, net.N, net.H
        self.methods = list(methods)
        self.dt = float(dt)
        self.does_basin_route = int(does_basin_route)
        cfg = MzrConfig()
        L.mzr_default_config(C.byref(cfg))
        cfg.dt = float(dt); cfg.nRoutes = len(self.methods)
        for i, m in enumerate(self.methods):
            cfg.routeMethods[i] = int(m)
        cfg.doesBasinRoute = int(does_basin_route); cfg.hw_drain_point = int(hw_drain_point)
        cfg.min_length_route = float(min_length_route); cfg.runoffMin = float(runoff_min)
        cfg.maxWindow = int(max_window); cfg.device = int(device); cfg.is_flux_wm = int(is_flux_wm)
        cfg.time_conv = float(time_conv); cfg.length_conv = float(length_conv)   # runoff units -> m/s
        cfg.stepBatch = int(step_batch)         # mzr_step: steps put aside and routed together (1 = every call routes its step)
        cfg.sweepShare = float(sweep_share)     # share of the device's wavefront slots this domain's persistent sweeps fill
        cfg.lakeMemoryPerMethod = int(lake_memory_per_method)   # several methods + Hanasaki memory: accept per-method copies (the reference shares one)
        cfg.sweepPriority = int(sweep_priority)  # 1: its sweeps' wavefronts go first on every SIMD (a small, deep domain beside a large one)
        self.is_flux_wm = int(is_flux_wm)
        self.max_window = int(max_window)
        self.h = C.c_void_p()
        self._check(L.mzr_create(C.byref(cfg), C.byref(self.h)))
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        self._keep = [i32(net.upGood), i32(net.reachId)]
        self._check(L.mzr_set_network(self.h, net.N, net.H, i32(net.downIndex), i32(net.upOffset), i32(net.upIndex),
                                      self._keep[0].ctypes.data, i32(net.hruOffset), i32(net.hruIndex),
                                      f64(net.hruWeight), self._keep[1].ctypes.data))
        for name in net.PARAM_ORDER:
            self._check(L.mzr_set_param(self.h, name.encode(), f64(net.params[name])))
        if frac_future is not None:
            ff = f64(frac_future)
            self.ntdh_bas = len(ff)
            self._check(L.mzr_set_frac_future(self.h, len(ff), ff))
        if uh_offset is not None:
            self.uh_offset = i32(uh_offset)
            self._check(L.mzr_set_uh(self.h, self.uh_offset, f64(uh)))
        self.lakes = lakes
        if lakes is not None:
            self._check(L.mzr_set_lakes(self.h, int(lakes["input_option"]), int(lakes["calendar_id"]), len(lakes["reach"]),
                                        i32(lakes["reach"]), i32(lakes["model_type"]), f64(lakes["par"])))
            if "targ_vol" in lakes:      # lakes that follow a target volume (is_vol_wm)
                self._check(L.mzr_set_lake_target(self.h, i32(lakes["targ_vol"]), int(lakes.get("vol_jumpstart", 0))))
        self.n_export = 0 if export_reaches is None else len(export_reaches)
        self.n_halo = 0 if halo_reaches is None else len(halo_reaches)
        if self.n_export or self.n_halo:
            ex = i32(export_reaches if export_reaches is not None else [])
            ha = i32(halo_reaches if halo_reaches is not None else [])
            hg = i32(halo_good if halo_good is not None else np.ones(len(ha)))
            self._check(L.mzr_set_boundary(self.h, len(ex), ex, len(ha), ha, hg))
        if history:      # history sums beyond discharge: H_INFLOW | H_HEIGHT | H_RUNOFF
            self._check(L.mzr_set_history(self.h, int(history)))
        self._check(L.mzr_init_state(self.h))

    # ---- plumbing
    def _check(self, rc):
        if rc != 0:
            buf = C.create_string_buffer(4096)
            self.L.mzr_last_error(self.h, buf, 4096)
            raise MzrError(rc, buf.value.decode(errors="replace"))

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.L.mzr_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- the hot path
    def step(self, T0, T1, runoff):
        """One main_route call: TSEC(1:2) = (T0, T1), runoff[nHru] in m/s."""
        self._check(self.L.mzr_step(self.h, float(T0), float(T1), np.ascontiguousarray(runoff, dtype=np.float64)))

    def run(self, runoff, t_start=0.0, wm_flux=None, first_step=0):
        """runoff[nSteps, nHru] (and wm_flux[nSteps, nRch] if is_flux_wm); returns
        REACH_Q[nSteps, nRoutes, nRch] (caller's reach order).  first_step: position of runoff[0] in the lake forcing of
        self.lakes (a run handed over in several calls)."""
        runoff = np.ascontiguousarray(runoff, dtype=np.float64)
        n = runoff.shape[0]
        out = np.zeros((n, len(self.methods), self.N))
        done = 0
        while done < n:
            w = min(self.max_window, n - done)
            if self.lakes is not None:
                self.set_lake_forcing(first_step + done, w)
            if self.is_flux_wm:
                self._check(self.L.mzr_set_wm_flux(self.h, w, np.ascontiguousarray(wm_flux[done:done + w], dtype=np.float64)))
            if getattr(self, "da", None) is not None:
                self.set_obs(self._da_done, w); self._da_done += w
            tr = getattr(self, "solute", None)
            if tr is not None:
                self._check(self.L.mzr_set_solute(self.h, w, np.ascontiguousarray(tr[self._sol_done:self._sol_done + w])))
            self._check(self.L.mzr_run(self.h, w, float(t_start) + done * self.dt, runoff[done:done + w]))
            for ix, m in enumerate(self.methods):
                buf = np.zeros((w, self.N))
                self._check(self.L.mzr_get_window_q(self.h, m, buf))
                out[done:done + w, ix, :] = buf
            if tr is not None:
                if self._sol_done == 0 or getattr(self, "solute_flux", None) is None or self.solute_flux.shape[0] != n:
                    self.solute_flux = np.zeros((n, len(self.methods), self.N))
                for ix, m in enumerate(self.methods):
                    if m != SUM:
                        buf = np.zeros((w, self.N))
                        self._check(self.L.mzr_get_window_solute(self.h, m, buf))
                        self.solute_flux[done:done + w, ix, :] = buf
                self._sol_done += w
            done += w
        return out

    def set_tracer(self, solute, time_conv=1.0, mass_conv=1.0):
        """Constituent routing (tracer = T): solute [nSteps, nHru] is the basin mass flux of the steps run() will route from
        now on (None switches it off); run() then also keeps reach_solute_flux of every step in self.solute_flux
        [nSteps, nRoutes, nRch] (zeros for the runoff accumulation)."""
        self.solute, self._sol_done = (None if solute is None else np.ascontiguousarray(solute, dtype=np.float64)), 0
        self.tracer_on = solute is not None
        self._check(self.L.mzr_set_tracer(self.h, int(solute is not None), float(time_conv), float(mass_conv)))

    def enable_tracer(self, time_conv=1.0, mass_conv=1.0):
        """Switch constituent routing on for a host that hands the solute over window by window itself (mzr_set_solute),
        as the stand-alone driver does: restart files then carry tfuture / solute_mass (ncfiles gates on tracer_on)."""
        self.solute, self._sol_done = None, 0
        self.tracer_on = True
        self._check(self.L.mzr_set_tracer(self.h, 1, float(time_conv), float(mass_conv)))

    def tracer_state(self):
        """dict(tfuture [nRch, ntdhBas] if the hillslope is routed, mass {method: [nRch]}) -- what a restart file keeps"""
        st = {"mass": {}}
        if self.does_basin_route == 1:
            tf = np.zeros((self.N, self.ntdh_bas))
            self._check(self.L.mzr_get_tracer_state(self.h, -1, tf.ctypes.data_as(C.c_void_p), None))
            st["tfuture"] = tf
        for m in self.methods:
            if m != SUM:
                a = np.zeros(self.N)
                self._check(self.L.mzr_get_tracer_state(self.h, m, None, a.ctypes.data_as(C.c_void_p)))
                st["mass"][m] = a
        return st

    def set_tracer_state(self, st):
        if "tfuture" in st:
            tf = np.ascontiguousarray(st["tfuture"], dtype=np.float64)
            self._check(self.L.mzr_set_tracer_state(self.h, -1, tf.ctypes.data_as(C.c_void_p), None))
        for m, a in st["mass"].items():
            a = np.ascontiguousarray(a, dtype=np.float64)
            self._check(self.L.mzr_set_tracer_state(self.h, m, None, a.ctypes.data_as(C.c_void_p)))

    def solute_state(self, method, which=0):
        out = np.zeros(self.N)
        self._check(self.L.mzr_get_solute(self.h, method, which, out))
        return out

    def set_da(self, da):
        """Direct insertion of gauge observations (qmodOption = 1): da = dict(blend, trend, gauge_reach[nGauge] (1-based),
        have[nSteps], obs[nSteps, nGauge]) for the steps run() will route from now on; None switches it off."""
        self.da, self._da_done = da, 0
        if da is None:
            self._check(self.L.mzr_set_da(self.h, 10, 1, 0, np.zeros(1, dtype=np.int32)))
        else:
            g = np.ascontiguousarray(da["gauge_reach"], dtype=np.int32)
            self._check(self.L.mzr_set_da(self.h, int(da["blend"]), int(da["trend"]), g.size, g))

    def set_obs(self, first, w):
        """observations of steps [first, first + w) of self.da for the next window"""
        self._check(self.L.mzr_set_obs(self.h, int(w), np.ascontiguousarray(self.da["have"][first:first + w], dtype=np.int32),
                                       np.ascontiguousarray(self.da["obs"][first:first + w], dtype=np.float64)))

    def set_solute(self, w, solute):
        """basin constituent mass flux [w, nHru] of the next window (constituent routing on: enable_tracer)"""
        self._check(self.L.mzr_set_solute(self.h, int(w), np.ascontiguousarray(solute, dtype=np.float64)))

    def set_wm_flux(self, w, wm_flux):
        """REACH_WM_FLUX [w, nRch] of the next window (is_flux_wm; -9999 = no data for the reach)."""
        self._check(self.L.mzr_set_wm_flux(self.h, int(w), np.ascontiguousarray(wm_flux, dtype=np.float64)))

    def set_lake_forcing(self, first, w, evap_dev_ptr=None, precip_dev_ptr=None):
        """Upload evaporation/precipitation and the calendar of steps [first, first+w) of self.lakes.  With device pointers
        ([w, nHru] each, e.g. filled by remap_device) the two fluxes stay on the device; with LakeInputOption = 1 they
        are not used and not moved."""
        lk = self.lakes
        ymd = np.asarray(lk["ymd"][first:first + w], dtype=np.int64)
        mdays = np.array([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31])
        leap = (lk["calendar_id"] == 1) & (((ymd[:, 0] % 4 == 0) & (ymd[:, 0] % 100 != 0)) | (ymd[:, 0] % 400 == 0))
        cum = np.concatenate([[0], np.cumsum(mdays)])[ymd[:, 1] - 1]
        doy = cum + ymd[:, 2] + (leap & (ymd[:, 1] > 2))
        c = lambda a, t: np.ascontiguousarray(a, dtype=t)
        cal = (c(ymd[:, 1], np.int32), c(ymd[:, 2], np.int32), c(doy, np.int32))
        if evap_dev_ptr is not None:
            self._check(self.L.mzr_set_lake_forcing_dev(self.h, int(w), C.c_void_p(int(evap_dev_ptr)), C.c_void_p(int(precip_dev_ptr)), *cal))
        elif int(lk["input_option"]) == 1:
            self._check(self.L.mzr_set_lake_forcing(self.h, int(w), None, None, *cal))
        else:
            ev, pr = c(lk["evap"][first:first + w], np.float64), c(lk["precip"][first:first + w], np.float64)
            self._check(self.L.mzr_set_lake_forcing(self.h, int(w), ev.ctypes.data, pr.ctypes.data, *cal))
        if "targ_vol" in lk:      # REACH_WM_VOL of the window
            self._check(self.L.mzr_set_wm_vol(self.h, int(w), c(lk["wm_vol"][first:first + w], np.float64)))

    def run_device(self, n_steps, t_start, runoff_dev_ptr):
        """Asynchronous window on device-resident runoff [n_steps, nHru] (e.g. a torch tensor's data_ptr())."""
        self._check(self.L.mzr_run_dev(self.h, int(n_steps), float(t_start), C.c_void_p(int(runoff_dev_ptr))))

    def run_async(self, n_steps, t_start, runoff_host_ptr):
        """Asynchronous window on host-resident (page-locked) runoff [n_steps, nHru]: copy and routing overlap
        with the window before."""
        self._check(self.L.mzr_run_async(self.h, int(n_steps), float(t_start), C.c_void_p(int(runoff_host_ptr))))

    def run_async_f32(self, n_steps, t_start, runoff_host_ptr):
        """As run_async, forcing in single precision as the files store it (widened on the device behind the copy)."""
        self._check(self.L.mzr_run_async_f32(self.h, int(n_steps), float(t_start), C.c_void_p(int(runoff_host_ptr))))

    def sync(self):
        self._check(self.L.mzr_sync(self.h))

    # ---- partition boundary records (device pointers; see include/mzr.h for the wire format)
    def boundary_size(self, n_steps, n_reach):
        return int(self.L.mzr_boundary_size(self.h, int(n_steps), int(n_reach)))

    def export_boundary(self, rec_dev_ptr):
        self._check(self.L.mzr_export_boundary_dev(self.h, C.c_void_p(int(rec_dev_ptr))))

    def export_lag(self):
        """True while the last window's final launches are kept back for the next window (overlapping windows): its boundary
        record is then exported one window later, by export_boundary_prev after the next run of the same length."""
        return bool(self.L.mzr_get_export_lag(self.h))

    def export_boundary_prev(self, rec_dev_ptr):
        """the record of the window BEFORE the last one (mzr_export_boundary_prev_dev)"""
        self._check(self.L.mzr_export_boundary_prev_dev(self.h, C.c_void_p(int(rec_dev_ptr))))

    def wait_export(self):
        """the host waits for the last export's record, not for what has been queued since"""
        self._check(self.L.mzr_wait_export(self.h))

    def import_boundary(self, n_steps, rec_dev_ptr, n_src, halo_base):
        self._check(self.L.mzr_import_boundary_dev(self.h, int(n_steps), C.c_void_p(int(rec_dev_ptr)), int(n_src), int(halo_base)))

    def wait_import(self):
        """the host waits until the last import_boundary has read its record -- and for nothing the handle keeps back (overlapping windows stay)"""
        self._check(self.L.mzr_wait_import(self.h))

    # ---- results / state
    def flux(self, method, which=F_Q):
        out = np.zeros(self.N)
        self._check(self.L.mzr_get_flux(self.h, method, which, out))
        return out

    def global_wb(self, method):
        """comp_global_wb of the last routed step: dict of the seven domain sums [m3] and the error term."""
        out = np.zeros(8)
        self._check(self.L.mzr_get_global_wb(self.h, method, out))
        return dict(zip(("dVol", "lateral", "precip", "take_actual", "evaporation", "outflow", "take_demand", "error"), out))

    def window_q(self, method, n_steps):
        out = np.zeros((n_steps, self.N))
        self._check(self.L.mzr_get_window_q(self.h, method, out))
        return out

    def mean_q(self, method, reset=False):
        out = np.zeros(self.N)
        self._check(self.L.mzr_get_mean_q(self.h, method, out, int(reset)))
        return out

    def mean(self, method, which):
        """interval mean of a history variable (include/mzr.h MZR_M_*); needs history=... at construction except for M_Q"""
        out = np.zeros(self.H if which == M_BAS_RUNOFF else self.N)
        self._check(self.L.mzr_get_mean(self.h, int(method), int(which), out))
        return out

    def reset_means(self):
        self._check(self.L.mzr_reset_means(self.h))

    def kwt_state(self):
        nw = np.zeros(self.N, np.int32)
        qf = np.zeros((self.N, WCAP)); ti = np.zeros((self.N, WCAP)); tr = np.zeros((self.N, WCAP))
        rf = np.zeros((self.N, WCAP), np.int32)
        self._check(self.L.mzr_get_kwt_state(self.h, nw, qf, ti, tr, rf))
        return nw, qf, ti, tr, rf

    def set_kwt_state(self, nw, qf, ti, tr, rf):
        c = lambda a, t: np.ascontiguousarray(a, dtype=t)
        self._check(self.L.mzr_set_kwt_state(self.h, c(nw, np.int32), c(qf, np.float64), c(ti, np.float64),
                                             c(tr, np.float64), c(rf, np.int32)))

    # restart: the same layouts back in (read_restart.f90:152-742)
    def set_irf_state(self, qfuture):
        self._check(self.L.mzr_set_irf_state(self.h, np.ascontiguousarray(qfuture, dtype=np.float64)))

    def set_mol_state(self, method, q):
        self._check(self.L.mzr_set_mol_state(self.h, method, np.ascontiguousarray(q, dtype=np.float64)))

    def set_basin_state(self, qfuture, basin_q):
        a = None if qfuture is None else np.ascontiguousarray(qfuture, dtype=np.float64)
        b = None if basin_q is None else np.ascontiguousarray(basin_q, dtype=np.float64)
        ptr = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)
        self._check(self.L.mzr_set_basin_state(self.h, ptr(a), ptr(b)))

    def set_volume(self, method, vol):
        self._check(self.L.mzr_set_volume(self.h, method, np.ascontiguousarray(vol, dtype=np.float64)))

    def irf_state(self):
        out = np.zeros(int(self.uh_offset[-1]))
        self._check(self.L.mzr_get_irf_state(self.h, out))
        return out

    def mol_state(self, method):
        out = np.zeros((self.N, NMOL[method]))
        self._check(self.L.mzr_get_mol_state(self.h, method, out))
        return out

    def basin_state(self):
        out = np.zeros((self.N, self.ntdh_bas))
        self._check(self.L.mzr_get_basin_state(self.h, out))
        return out

    # ---- schedule / measurement
    def schedule(self):
        a, b = C.c_int(0), C.c_int(0)
        self._check(self.L.mzr_get_schedule(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def sweep_arrivals(self):
        """(arrived, joined) wavefronts of the last KWT sweep launch, and the histogram of start delays since init (bucket k: < 2^k x 10 ns)"""
        a, j = C.c_int(0), C.c_int(0)
        hist = (C.c_longlong * 32)()
        self._check(self.L.mzr_get_sweep_arrivals(self.h, C.byref(a), C.byref(j), hist))
        return a.value, j.value, [int(x) for x in hist]

    def sweep_clock(self, max_n=1024, reset=False):
        """durations [ms] of the latest KWT sweep launches on the device's own clock (first wavefront in -> last wavefront out), oldest first"""
        ms = (C.c_double * max(1, max_n))()
        n = C.c_int(0)
        self._check(self.L.mzr_get_sweep_clock(self.h, max_n, ms, C.byref(n), 1 if reset else 0))
        return [float(ms[k]) for k in range(n.value)]

    def sweep_retries(self):
        """windows whose persistent KWT sweep gave up (ierr 93) and that were routed again through one launch per stage"""
        n = C.c_longlong(0)
        self._check(self.L.mzr_get_sweep_retries(self.h, C.byref(n)))
        return n.value

    def sweep_info(self):
        """(wavefronts of the persistent KWT sweep, wavefronts the device holds at once, items dealt to them)"""
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        self._check(self.L.mzr_get_sweep_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    # ---- forcing remap (process_remap.f90:32-316)
    def set_remap(self, mp):
        """mp: mapping-file content as a dict (see synthetic.make_remap): hru_ix, num_qhru, weight, n1, n2
        and qhru_ix [+ qhru_id, src_id] (polygon vector) or i_index, j_index (grid)."""
        i32 = lambda k: np.ascontiguousarray(mp[k], dtype=np.int32) if k in mp else None
        i64 = lambda k: np.ascontiguousarray(mp[k], dtype=np.int64) if k in mp else None
        ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        kind = 2 if mp.get("n2", 0) > 0 else 1
        hix, num, w = i32("hru_ix"), i32("num_qhru"), np.ascontiguousarray(mp["weight"], dtype=np.float64)
        q, ii, jj, qid, sid = i32("qhru_ix"), i32("i_index"), i32("j_index"), i64("qhru_id"), i64("src_id")
        self._check(self.L.mzr_set_remap(self.h, kind, hix.size, hix, num, w.size, ptr(q), ptr(ii), ptr(jj), w,
                                         int(mp["n1"]), int(mp.get("n2", 0)), ptr(qid), ptr(sid)))

    def set_sort_map(self, ix_in, remove_negatives=True):
        ix = np.ascontiguousarray(ix_in, dtype=np.int32)
        self._check(self.L.mzr_set_sort_map(self.h, ix.size, ix, int(bool(remove_negatives))))

    def remap_device(self, n_steps, src_dev_ptr, dst_dev_ptr):
        self._check(self.L.mzr_remap_runoff_dev(self.h, int(n_steps), C.c_void_p(int(src_dev_ptr)), C.c_void_p(int(dst_dev_ptr))))

    def run_source_device(self, n_steps, t_start, src_dev_ptr):
        """remap + run of one window from the hydrologic model's own runoff layer (device memory)."""
        self._check(self.L.mzr_run_src_dev(self.h, int(n_steps), float(t_start), C.c_void_p(int(src_dev_ptr))))

    def set_profiling(self, mode):
        """mode: 0 off, 1 HIP events around every stage launch (timing()), 2 KWT particle-traffic
        counters (kwt_traffic(); device atomics, not for timed runs), 3 both."""
        self._check(self.L.mzr_set_profiling(self.h, int(mode)))

    def timing(self, method, reset=False):
        n, ms, rs = C.c_longlong(0), C.c_double(0), C.c_longlong(0)
        self._check(self.L.mzr_get_timing(self.h, method, C.byref(n), C.byref(ms), C.byref(rs), int(reset)))
        lo, hi = C.c_double(0), C.c_double(0)
        self._check(self.L.mzr_get_timing_range(self.h, method, C.byref(lo), C.byref(hi), int(reset)))
        return dict(launches=n.value, kernel_ms=ms.value, reach_steps=rs.value, min_ms=lo.value, max_ms=hi.value)

    def kwt_traffic(self, reset=False):
        v = [C.c_longlong(0) for _ in range(6)]
        self._check(self.L.mzr_get_kwt_traffic(self.h, *[C.byref(x) for x in v], int(reset)))
        return dict(zip(("w_in", "w_up", "w_out", "n_head", "n_route", "n_edges"), [x.value for x in v]))


class Comm:
    """The library's own boundary-record transport (RCCL point-to-point, include/mzr.h mzr_comm_*): what a host
    without torch uses.  unique_id() on rank 0, the 128 bytes to every rank by the host's own means, then Comm(...)."""

    @staticmethod
    def unique_id() -> bytes:
        L = load_library()
        buf = C.create_string_buffer(128)
        Comm._check(L, L.mzr_comm_unique_id(buf))
        return buf.raw

    @staticmethod
    def _check(L, rc):
        if rc:
            buf = C.create_string_buffer(512)
            L.mzr_comm_last_error(buf, 512)
            raise MzrError(rc, buf.value.decode(errors="replace"))

    def __init__(self, rank, n_ranks, uid: bytes, device=0):
        self.L = load_library()
        self.c = C.c_void_p()
        self._check(self.L, self.L.mzr_comm_init(int(rank), int(n_ranks), C.create_string_buffer(uid, 128), int(device), C.byref(self.c)))
        self.rank, self.n_ranks = rank, n_ranks

    def send(self, dom, dev_ptr, n, peer):
        self._check(self.L, self.L.mzr_comm_send(self.c, dom.h, C.c_void_p(int(dev_ptr)), int(n), int(peer)))

    def recv(self, dom, dev_ptr, n, peer):
        self._check(self.L, self.L.mzr_comm_recv(self.c, dom.h, C.c_void_p(int(dev_ptr)), int(n), int(peer)))

    def recv_many(self, dom, triples):
        """triples: (device pointer, doubles, peer) -- one grouped call, transfers side by side"""
        k = len(triples)
        ptrs = (C.c_void_p * k)(*[C.c_void_p(int(t[0])) for t in triples])
        ns = (C.c_longlong * k)(*[int(t[1]) for t in triples])
        peers = (C.c_int * k)(*[int(t[2]) for t in triples])
        self._check(self.L, self.L.mzr_comm_recv_many(self.c, dom.h, k, ptrs, ns, peers))

    def sync(self):
        self._check(self.L, self.L.mzr_comm_sync(self.c))

    def close(self):
        if self.c:
            self.L.mzr_comm_destroy(self.c)
            self.c = C.c_void_p()

