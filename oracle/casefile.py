"""Binary "case file": one routing problem (network, parameters, unit hydrographs, runoff series)
in a flat little-endian layout that a Fortran `access='stream'` reader consumes directly.

Read by mizuroute_amd/fortran/mzr_demo.f90 (the Fortran host demo of the C-ABI) and by the test
harness that drives the reference solvers.  Layout (all int32 / float64):
  magic 'MZRC', version
  N, H, nSteps, nRoutes, routeMethods[6], doesBasinRoute, hw_drain_point, nUp, nHru, nOrder, nBranch,
  uhSource, ntdhBas, nUh, dumpEvery, isFluxWm, isLakeSim
  dt, min_length_route, runoffMin, fshape, tscale, velo, diff, t_start
  downIndex[N] reachId[N] upOffset[N+1] upIndex[nUp] upGood[nUp] hruOffset[N+1] hruIndex[nHru] hruWeight[nHru]
  par[11][N]   (RiverNetwork.PARAM_ORDER)
  orderOffset[nOrder+1] branchOffset[nBranch+1] seg[N]       (a processing schedule for CPU drivers)
  if uhSource == 1: fracFuture[ntdhBas] uhOffset[N+1] uh[nUh]
  runoff[nSteps][H]
  if isFluxWm == 1: wmflux[nSteps][N]   (REACH_WM_FLUX, + abstraction / - injection, m3/s)
  if isLakeSim == 1: LakeInputOption, calendarId (0 noleap, 1 standard), nLake,
                     ymd[nSteps][3] (year, month, day of every step),
                     lakeReach[nLake] (1-based), lakeModelType[nLake], lakePar[NLAKEPAR][nLake],
                     evap[nSteps][H], precip[nSteps][H]
"""
from __future__ import annotations

import struct

import numpy as np

MAGIC_IN = 1297765955
MAGIC_DA = 1145133645       # b'MZAD': trailing gauge-observation section (direct insertion)
MAGIC_TR = 1381259853       # b'MZTR': trailing constituent section (tracer)

# lake parameter rows (RPARAM fields, dataTypes.f90:196-254), in this order
from mizuroute_amd.lakepar import LAKE_PAR, NLAKEPAR  # noqa: F401  (order of the lake parameter rows)


def serial_schedule(net):
    """One order, one branch, all reaches upstream->downstream (a valid serial schedule)."""
    order = net.topo_order() + 1
    return np.array([0, 1], np.int32), np.array([0, net.N], np.int32), order.astype(np.int32)



def write_case(path, net, runoff, dt, methods, does_basin_route=1, hw_drain_point=2,
               min_length_route=0.0, runoff_min=0.0, fshape=2.5, tscale=86400.0, velo=1.5, diff=5000.0,
               t_start=0.0, uh=None, schedule=None, dump_every=1, wm_flux=None, lakes=None, da=None, solute=None):
    """lakes: None or dict(input_option, calendar_id, ymd[nSteps,3], reach[nLake] (1-based), model_type[nLake],
    par[NLAKEPAR, nLake], evap[nSteps,H], precip[nSteps,H])."""
    """uh: None -> the harness calls the reference's basinUH/make_uh; else (frac, uhOffset, uh).
    da: None or dict(blend, trend, gauge_reach[nGauge] (1-based), have[nSteps], obs[nSteps, nGauge]) -> qmodOption = 1.
    solute: None or basin constituent mass flux [nSteps, H] -> tracer = T."""
    runoff = np.ascontiguousarray(runoff, dtype=np.float64)
    n_steps = runoff.shape[0]
    orderOffset, branchOffset, seg = schedule if schedule is not None else serial_schedule(net)
    m = list(methods) + [-1] * (6 - len(methods))
    with open(path, "wb") as f:
        f.write(struct.pack("<2i", MAGIC_IN, 3))
        ints = [net.N, net.H, n_steps, len(methods)] + m + [does_basin_route, hw_drain_point,
                int(net.upOffset[-1]), int(net.hruOffset[-1]), len(orderOffset) - 1, len(branchOffset) - 1,
                1 if uh is not None else 0, len(uh[0]) if uh is not None else 0,
                int(uh[1][-1]) if uh is not None else 0, int(dump_every), 1 if wm_flux is not None else 0,
                (2 if "targ_vol" in lakes else 1) if lakes is not None else 0]      # 2: + target-volume section
        f.write(struct.pack(f"<{len(ints)}i", *ints))
        f.write(struct.pack("<8d", dt, min_length_route, runoff_min, fshape, tscale, velo, diff, t_start))
        for a in (net.downIndex, net.reachId, net.upOffset, net.upIndex, net.upGood, net.hruOffset, net.hruIndex):
            f.write(np.ascontiguousarray(a, dtype="<i4").tobytes())
        f.write(np.ascontiguousarray(net.hruWeight, dtype="<f8").tobytes())
        # par(N,11) column-major == [11][N] row-major
        f.write(np.ascontiguousarray(net.param_matrix(), dtype="<f8").tobytes())
        for a in (orderOffset, branchOffset, seg):
            f.write(np.ascontiguousarray(a, dtype="<i4").tobytes())
        if uh is not None:
            f.write(np.ascontiguousarray(uh[0], dtype="<f8").tobytes())
            f.write(np.ascontiguousarray(uh[1], dtype="<i4").tobytes())
            f.write(np.ascontiguousarray(uh[2], dtype="<f8").tobytes())
        # runoff(H, nSteps) column-major == [nSteps][H] row-major
        f.write(runoff.astype("<f8").tobytes())
        if wm_flux is not None:
            f.write(np.ascontiguousarray(wm_flux, dtype="<f8").tobytes())
        if lakes is not None:
            nl = len(lakes["reach"])
            f.write(struct.pack("<3i", int(lakes["input_option"]), int(lakes["calendar_id"]), nl))
            f.write(np.ascontiguousarray(lakes["ymd"], dtype="<i4").tobytes())
            f.write(np.ascontiguousarray(lakes["reach"], dtype="<i4").tobytes())
            f.write(np.ascontiguousarray(lakes["model_type"], dtype="<i4").tobytes())
            f.write(np.ascontiguousarray(lakes["par"], dtype="<f8").tobytes())
            f.write(np.ascontiguousarray(lakes["evap"], dtype="<f8").tobytes())
            f.write(np.ascontiguousarray(lakes["precip"], dtype="<f8").tobytes())
            if "targ_vol" in lakes:      # NETOPO%LakeTargVol flags, is_vol_wm_jumpstart, REACH_WM_VOL[nSteps][N]
                f.write(np.ascontiguousarray(lakes["targ_vol"], dtype="<i4").tobytes())
                f.write(struct.pack("<i", int(lakes.get("vol_jumpstart", 0))))
                f.write(np.ascontiguousarray(lakes["wm_vol"], dtype="<f8").tobytes())
        if da is not None:
            g = np.ascontiguousarray(da["gauge_reach"], dtype="<i4")
            f.write(struct.pack("<4i", MAGIC_DA, g.size, int(da["blend"]), int(da["trend"])))
            f.write(g.tobytes())
            f.write(np.ascontiguousarray(da["have"], dtype="<i4").tobytes())
            f.write(np.ascontiguousarray(da["obs"], dtype="<f8").tobytes())      # [nSteps][nGauge] == val(nGauge, nSteps)
        if solute is not None:
            f.write(struct.pack("<i", MAGIC_TR))
            f.write(np.ascontiguousarray(solute, dtype="<f8").tobytes())         # [nSteps][H] == solute(H, nSteps)


