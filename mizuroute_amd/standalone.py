"""Stand-alone run from a mizuRoute control file: the host side that the reference's
`route_runoff.exe <control file>` provides (standalone/route_runoff.f90, read_control.f90,
read_param.f90, process_ntopo.f90 / network_topo.f90 set-up, get_basin_runoff.f90 forcing,
historyFile.f90, write_restart_pio.f90), in front of the device hot path.

    python -m mizuroute_amd.standalone <control_file> [--window W]

Scope (what a run of the reference's SAMPLE.control needs, nothing more):
  * control file `<tag> value ! comment` (read_control.f90) and the parameter namelist
    &HSLOPE fshape,tscale / &IRF_UH velo,diff / &KWT mann_n,wscale (read_param.f90);
  * river network from `<fname_ntopOld>`: ids, downstream ids, length, slope, HRU ids / areas / their
    segment (names through the <varname_*> dictionary; defaults of popMetadat.f90:124-170), with the
    augmentation the reference computes at start-up: upstream lists (network_topo.f90), HRU weights =
    area fractions (:188), BASAREA / TOTAREA (:748-764), goodBas (:771-775), hydraulic geometry
    R_WIDTH = wscale*sqrt(TOTAREA), R_DEPTH = high_depth without flood plain, R_MAN_N = mann_n,
    R_STORAGE (process_ntopo.f90:174-205), slope floor min_slope (:360), unit hydrographs (uh.py);
  * forcing from `<fname_qsim>` (one NetCDF file, or a text file listing them) on the simulation
    step or on any other regular step `<dt_ro>` (timeMap_sim_forc: overlap-weighted records), on the
    river-network HRUs in file order (is_remap = F -> sort_flux) or remapped (is_remap = T, 1-D);
  * lakes (`<is_lake_sim> T`): flags `islake`, `lakeModelType`, `LakeTargVol` and the Doll / Hanasaki / HYPE parameters
    from the topology file (popMetadat.f90:124-232 names, `<varname_*>` overrides), evaporation and precipitation from
    the forcing files (`<vname_evapo>`, `<vname_precip>`, `<LakeInputOption>`, scale / offset / sign keys;
    get_basin_runoff.f90:136-205), `<lakeRegulate> F` = every lake natural; water management (`<is_flux_wm>`,
    `<is_vol_wm>`, `<is_vol_wm_jumpstart>`): `<fname_wm>` file(s) on their own step `<dt_wm>`, fluxes and target volumes
    sorted onto the reaches by `<vname_segid_wm>` (get_basin_runoff.f90:106-109, 207-250);
  * river network subset mode (`<seg_outlet>` > 0): the part of the network upstream of a segment written to `<fname_ntopNew>`;
  * constituent routing (`<tracer> T`): `<vname_solute>` beside the runoff, `<units_cc>` mass / time, through the runoff's mapping;
    history variables `localSolute`, `soluteFlux`, `soluteMass` (popMetadat.f90:266-268; other methods than DW with their prefix);
  * direct insertion of gauge observations (`<qmodOption> 1`, `<qBlendPeriod>`, `<QerrTrend>`): `<gageMetaFile>` + `<fname_gageObs>`;
  * history file(s) `<case_name>.h.<start>.nc` at `<outputFrequency>` (a multiple of the step or
    `daily`), one file per run (`<newFileFrequency> single`), restart in / out (`<fname_state_in>`,
    `<restart_write> last`).
NetCDF files are classic (CDF-1/2) through scipy -- the image has no netCDF-4/HDF5 library.  The file
conventions are taken from the reference's sources and documentation; they are not pinned by running
the reference's own I/O (it needs ParallelIO).
"""
from __future__ import annotations

import datetime as _dt
import os
import re
import sys

import numpy as np
from scipy.io import netcdf_file

from . import api, ncfiles, uh as uhmod
from .synthetic import HIGH_DEPTH, RiverNetwork, build_upstream_csr, hops_to_outlet

MIN_SLOPE = 1.0e-6            # public_var.f90:30
VERY_SMALL = np.finfo(np.float64).tiny
DEFAULT_NAMES = dict(varname_area="area", varname_length="length", varname_slope="slope", varname_HRUid="HRUid",
                     varname_hruSegId="hruSegId", varname_segId="segId", varname_downSegId="downSegId")


def read_control(path: str) -> dict:
    """`<tag>  value   ! comment` lines (read_control.f90); later tags win."""
    out = {}
    for line in open(path):
        s = line.strip()
        if not s.startswith("<"):
            continue
        m = re.match(r"<([^>]+)>\s*([^!]*)", s)
        if m:
            out[m.group(1).strip()] = m.group(2).strip()
    return out


def read_param_nml(path: str) -> dict:
    """&HSLOPE / &IRF_UH / &KWT namelist groups with scalar values (read_param.f90)."""
    out = dict(fshape=2.5, tscale=86400.0, velo=1.5, diff=5000.0, mann_n=0.01, wscale=0.001, dscale=0.0036)
    for k, v in re.findall(r"(\w+)\s*=\s*([-+0-9.eEdD]+)", open(path).read()):
        out[k] = float(v.replace("d", "e").replace("D", "e"))
    return out


def _truth(s: str) -> bool:
    return s.strip().upper() in ("T", ".TRUE.", "TRUE")


def _parse_time(s: str) -> _dt.datetime:
    s = s.strip()
    for fmt in ("%Y-%m-%d %H:%M:%S", "%Y-%m-%d %H:%M", "%Y-%m-%d"):
        try:
            return _dt.datetime.strptime(s, fmt)
        except ValueError:
            pass
    raise ValueError(f"cannot parse time '{s}'")


def _time_axis(var) -> np.ndarray:
    """Seconds since 1970-01-01 of a CF time variable (`<unit> since <date>`)."""
    units = var.units.decode() if isinstance(var.units, bytes) else var.units
    m = re.match(r"\s*(\w+)\s+since\s+(.*)", units)
    scale = {"seconds": 1.0, "second": 1.0, "sec": 1.0, "minutes": 60.0, "hours": 3600.0, "hour": 3600.0, "days": 86400.0, "day": 86400.0}[m.group(1).lower()]
    ref = _parse_time(m.group(2).strip().replace("T", " ").split(".")[0])
    return (ref - _dt.datetime(1970, 1, 1)).total_seconds() + np.asarray(var[:], dtype=np.float64) * scale


def unit_factors(units: str):
    """runoff [units] -> m/s: (time_conv, length_conv) as get_basin_runoff / init_model_data derive them."""
    length, time = [u.strip().lower() for u in units.split("/")]
    lc = {"mm": 1.0e-3, "m": 1.0}[length]
    tc = {"s": 1.0, "sec": 1.0, "second": 1.0, "h": 1.0 / 3600.0, "hr": 1.0 / 3600.0, "hour": 1.0 / 3600.0, "d": 1.0 / 86400.0, "day": 1.0 / 86400.0}[time]
    return tc, lc


def time_map(start_ro_sec: float, dt: float, dt_ro: float, n_ro: int, ix_time: int):
    """timeMap_sim_forc (get_basin_runoff.f90): which forcing records overlap simulation step ix_time (1-based) and
    with what weight.  start_ro_sec = simulation start minus forcing start [s].  Returns (records, fracs):
    0-based record indices and their weights, fracs = None when one record covers the whole step (the reference
    then uses that record as it is)."""
    very_small = 1.0e-12
    lo = start_ro_sec + dt * (ix_time - 1)
    hi = lo + dt
    edge = lambda i: dt_ro * i                     # frcLapse(i+1), i = 0..n_ro
    front = next((i for i in range(1, n_ro + 1) if lo < edge(i)), None)
    end = next((i for i in range(1, n_ro + 1) if hi < edge(i) or abs(hi - edge(i)) < very_small), None)
    if front is None or end is None:
        raise ValueError("forcing files do not cover the simulation period")
    if front > end:
        raise ValueError("timeMap_sim_forc/index of idxFront lower than idxEnd")
    if front == end:
        return [front - 1], None
    recs, fracs = [], []
    for i in range(front, end + 1):
        recs.append(i - 1)
        fracs.append((edge(i) - lo) / dt if i == front else (hi - edge(i - 1)) / dt if i == end else (edge(i) - edge(i - 1)) / dt)
    return recs, fracs


def build_network(ctl: dict, nml: dict):
    """River network + parameters from the topology file, augmented as the reference does at start-up."""
    name = lambda k: ctl.get(k, DEFAULT_NAMES[k])
    f = netcdf_file(os.path.join(ctl.get("ancil_dir", ""), ctl["fname_ntopOld"]), "r", mmap=False)
    v = f.variables
    seg_id = np.asarray(v[name("varname_segId")][:], dtype=np.int64)
    down_id = np.asarray(v[name("varname_downSegId")][:], dtype=np.int64)
    length = np.asarray(v[name("varname_length")][:], dtype=np.float64)
    slope = np.asarray(v[name("varname_slope")][:], dtype=np.float64)
    hru_id = np.asarray(v[name("varname_HRUid")][:], dtype=np.int64)
    hru_seg = np.asarray(v[name("varname_hruSegId")][:], dtype=np.int64)
    hru_area = np.asarray(v[name("varname_area")][:], dtype=np.float64)
    f.close()
    return augment_topology(seg_id, down_id, length, slope, hru_id, hru_seg, hru_area, nml), hru_id


def augment_topology(seg_id, down_id, length, slope, hru_id, hru_seg, hru_area, nml):
    """What the reference derives at start-up from the raw topology (augment_ntopo, process_ntopo.f90:39-266, with
    hru2segment / up2downSegment / reachOrder / reach_list of network_topo.f90): downstream indices, upstream lists in
    reach order, the HRUs of every reach in file order with their area weights (network_topo.f90:188), BASAREA / TOTAREA,
    goodBas (upstream reaches with a contributing area), hydraulic geometry from wscale (no floodplain: depth = high_depth),
    channel storage, and the slope floor of put_data_struct (process_ntopo.f90:274).  Compared with the compiled reference
    routines in the test test_network_augmentation_matches_the_reference."""
    seg_id, down_id = np.asarray(seg_id, dtype=np.int64), np.asarray(down_id, dtype=np.int64)
    length, slope = np.asarray(length, dtype=np.float64), np.asarray(slope, dtype=np.float64)
    hru_id, hru_seg, hru_area = np.asarray(hru_id, dtype=np.int64), np.asarray(hru_seg, dtype=np.int64), np.asarray(hru_area, dtype=np.float64)
    N, H = seg_id.size, hru_id.size
    ix = {int(s): i for i, s in enumerate(seg_id)}
    downIndex = np.array([ix.get(int(dn), -1) + 1 for dn in down_id], dtype=np.int32)   # 0: outlet / not in the network
    upOffset, upIndex = build_upstream_csr(downIndex)
    # HRUs per segment in file order, weights = area fractions (network_topo.f90:188)
    seg_of_hru = np.array([ix.get(int(s), -1) for s in hru_seg], dtype=np.int64)
    order = np.argsort(seg_of_hru, kind="stable")
    order = order[seg_of_hru[order] >= 0]
    cnt = np.bincount(seg_of_hru[order], minlength=N)
    hruOffset = np.zeros(N + 1, dtype=np.int32); hruOffset[1:] = np.cumsum(cnt)
    hruIndex = (order + 1).astype(np.int32)
    # the reference sums a reach's HRU areas one after the other in file order (network_topo.f90:160-190)
    # (position by position over the ragged lists: the j-th HRU of every reach at once -- the same left-to-right additions)
    basarea = np.zeros(N)
    for j in range(int(cnt.max()) if cnt.size else 0):
        has = np.nonzero(cnt > j)[0]
        basarea[has] = basarea[has] + hru_area[order[hruOffset[has] + j]]
    w = np.where(basarea[seg_of_hru[order]] > 0, hru_area[order] / np.where(basarea[seg_of_hru[order]] > 0, basarea[seg_of_hru[order]], 1.0), 0.0)
    down0 = downIndex.astype(np.int64) - 1
    dist = hops_to_outlet(down0)
    # area above a reach: its upstream reaches' total areas added in upstream-list order, headwaters first (reachOrder)
    # (level by level from the headwaters, and within a level the j-th upstream reach of every reach at once: the same additions in
    # the same order as the reference's loop over the upstream list)
    totarea = basarea.copy()
    nup = np.diff(upOffset).astype(np.int64)
    by_level = np.argsort(-dist, kind="stable")
    bounds = np.concatenate([[0], np.cumsum(np.bincount(int(dist.max()) - dist, minlength=int(dist.max()) + 1))]) if N else np.zeros(1, np.int64)
    for lv in range(bounds.size - 1):
        rs = by_level[bounds[lv]:bounds[lv + 1]]
        ups = np.zeros(rs.size)
        for j in range(int(nup[rs].max()) if rs.size else 0):
            has = np.nonzero(nup[rs] > j)[0]
            ups[has] = ups[has] + totarea[upIndex[upOffset[rs[has]] + j] - 1]
        totarea[rs] = basarea[rs] + ups
    width = nml["wscale"] * np.sqrt(totarea)
    rdepth = np.full(N, HIGH_DEPTH)
    side = np.zeros(N)
    params = dict(R_SLOPE=np.maximum(slope, MIN_SLOPE), R_MAN_N=np.full(N, nml["mann_n"]), R_WIDTH=width, R_DEPTH=rdepth,
                  RLENGTH=length, R_STORAGE=rdepth * (width + side * rdepth) * length, SIDE_SLOPE=side,
                  FLDP_SLOPE=np.full(N, 1000.0), BASAREA=basarea, TOTAREA=totarea, MINFLOW=np.zeros(N))
    good = (totarea > VERY_SMALL).astype(np.int32)
    return RiverNetwork(N=N, H=H, downIndex=downIndex, reachId=seg_id.astype(np.int32), upOffset=upOffset, upIndex=upIndex,
                        # (goodBasin of every upstream slot follows the reach's OWN total area: network_topo.f90:771-775)
                        upGood=np.repeat(good, np.diff(upOffset)).astype(np.int32),
                        hruOffset=hruOffset, hruIndex=hruIndex, hruWeight=w, params=params)


def write_subset(ctl: dict, log=print) -> dict:
    """River-network subset mode (`<seg_outlet>` > 0; process_ntopo.f90:236 reach_mask, init_model_data.f90:718-745): the
    reaches upstream of -- and including -- the outlet segment and their HRUs are written to `<fname_ntopNew>` with every
    variable of the input topology file, and the run stops there ("run again using the new network topology file")."""
    name = lambda k: ctl.get(k, DEFAULT_NAMES[k])
    src = netcdf_file(os.path.join(ctl.get("ancil_dir", ""), ctl["fname_ntopOld"]), "r", mmap=False)
    v = src.variables
    seg_id = np.asarray(v[name("varname_segId")][:], dtype=np.int64)
    down_id = np.asarray(v[name("varname_downSegId")][:], dtype=np.int64)
    hru_seg = np.asarray(v[name("varname_hruSegId")][:], dtype=np.int64)
    out_id = int(ctl["seg_outlet"])
    ix = {int(x): i for i, x in enumerate(seg_id)}
    if out_id not in ix:
        raise ValueError(f"<seg_outlet> {out_id} is not a segment of {ctl['fname_ntopOld']}")
    ups = [[] for _ in range(seg_id.size)]
    for i, dn in enumerate(down_id):
        j = ix.get(int(dn), -1)
        if j >= 0:
            ups[j].append(i)
    keep = np.zeros(seg_id.size, bool)
    stack = [ix[out_id]]
    while stack:
        i = stack.pop()
        if not keep[i]:
            keep[i] = True
            stack.extend(ups[i])
    seg_sel = np.nonzero(keep)[0]
    hru_sel = np.nonzero(np.isin(hru_seg, seg_id[seg_sel]))[0]
    seg_dim, hru_dim = v[name("varname_segId")].dimensions[0], v[name("varname_HRUid")].dimensions[0]
    path = os.path.join(ctl.get("ancil_dir", ""), ctl["fname_ntopNew"])
    dst = netcdf_file(path, "w", version=2)
    for dname, dlen in src.dimensions.items():
        dst.createDimension(dname, seg_sel.size if dname == seg_dim else hru_sel.size if dname == hru_dim else dlen)
    for vname, var in v.items():
        o = dst.createVariable(vname, var.data.dtype.char if var.data.dtype.char != "l" else "i", var.dimensions)
        for a in getattr(var, "_attributes", {}):
            setattr(o, a, getattr(var, a))
        data = np.asarray(var[:])
        if var.dimensions and var.dimensions[0] == seg_dim:
            data = data[seg_sel]
        elif var.dimensions and var.dimensions[0] == hru_dim:
            data = data[hru_sel]
        if vname == name("varname_downSegId"):       # the outlet of the subset drains nowhere
            data = np.where(np.isin(data, seg_id[seg_sel]), data, -1).astype(data.dtype)
            data[np.nonzero(seg_id[seg_sel] == out_id)[0]] = -1
        o[:] = data
    dst.close(); src.close()
    log(f"river network subset mode: {seg_sel.size} of {seg_id.size} reaches, {hru_sel.size} HRUs -> {path}; run again using the new network topology file")
    return dict(subset=path, reaches=int(seg_sel.size), hrus=int(hru_sel.size))


def _forcing_files(ctl: dict):
    p = os.path.join(ctl.get("input_dir", ""), ctl["fname_qsim"])
    if p.endswith(".nc"):
        return [p]
    return [os.path.join(ctl.get("input_dir", ""), ln.strip()) for ln in open(p) if ln.strip()]


REAL_MISSING = -9999.0


def _wm_files(ctl: dict):
    p = os.path.join(ctl.get("input_dir", ""), ctl["fname_wm"])
    if p.endswith(".nc"):
        return [p]
    return [os.path.join(ctl.get("input_dir", ""), ln.strip()) for ln in open(p) if ln.strip()]


class ForcingSeries:
    """Records of one variable family over one or several NetCDF files, on their own regular step, mapped onto the
    simulation steps (read_forcing_data + timeMap_sim_forc, get_basin_runoff.f90:262-372): a step inside one record
    takes the record as it is, a step across records their overlap-weighted sum."""

    def __init__(self, files, vname_time: str, dt_in: float, t0_sec: float):
        self.handles = [netcdf_file(p, "r", mmap=False) for p in files]
        taxis = np.concatenate([_time_axis(h.variables[vname_time]) for h in self.handles])
        self.frec = np.concatenate([np.full(h.variables[vname_time].shape[0], i) for i, h in enumerate(self.handles)])
        self.lrec = np.concatenate([np.arange(h.variables[vname_time].shape[0]) for h in self.handles])
        self.start_sec = t0_sec - float(taxis[0])         # the first record covers [taxis[0], taxis[0] + dt_in)
        self.n, self.dt_in = taxis.size, dt_in

    def var(self, name):
        return self.handles[0].variables[name]

    def step(self, name: str, dt: float, ix_time: int) -> np.ndarray:
        """the variable at simulation step ix_time (1-based), flattened ((lat, lon) -> j-major cells)"""
        recs, fracs = time_map(self.start_sec, dt, self.dt_in, self.n, ix_time)
        get = lambda i: np.asarray(self.handles[self.frec[i]].variables[name][self.lrec[i]], dtype=np.float64).reshape(-1)
        if fracs is None:
            return get(recs[0])
        acc = np.zeros_like(get(recs[0]))
        for i, fr in zip(recs, fracs):
            acc = acc + fr * get(i)
        return acc

    def close(self):
        for h in self.handles:
            h.close()


def scale_forcing(a: np.ndarray, scale: float, offset: float) -> np.ndarray:
    """scale_forcing (get_basin_runoff.f90:375-425): only when one of the two is given; missing values stay."""
    vs = 1.0e-12
    if abs(scale - REAL_MISSING) < vs and abs(offset - REAL_MISSING) < vs:
        return a
    sc = 1.0 if abs(scale - REAL_MISSING) < vs else scale
    of = 0.0 if abs(offset - REAL_MISSING) < vs else offset
    return np.where(np.abs(a - REAL_MISSING) > vs, sc * a + of, a)


def suppressed(scale: float, offset: float) -> bool:
    """scale 0 and offset 0 or absent: the flux is not read and taken as zero (get_basin_runoff.f90:140-142, 176-178)"""
    vs = 1.0e-12
    return abs(scale) < vs and (abs(offset) < vs or abs(offset - REAL_MISSING) < vs)


def sort_flux(ix_in: np.ndarray, flux: np.ndarray, n: int, remove_negatives: bool) -> np.ndarray:
    """process_remap.f90:268-316 on the host (water-management series are nSeg values per step): entries of the data
    set land on position ix_in (1-based, < 1 = not in the network), everything else is realMissing; negative values
    (realMissing too) become 0 when asked."""
    out = np.full(n, REAL_MISSING)
    ok = ix_in >= 1
    out[ix_in[ok] - 1] = flux[ok]
    if remove_negatives:
        out[out < 0.0] = 0.0
    return out


def read_lakes(ctl: dict, net, n_steps: int) -> dict:
    """Lake flags and parameters of the topology file (process_ntopo.f90 / popMetadat.f90 names) as the dictionary
    api.RoutingDomain takes: reach (1-based), model_type, par[NLAKEPAR, nLake], input_option, calendar_id."""
    from .lakepar import LAKE_PAR, NLAKEPAR
    f = netcdf_file(os.path.join(ctl.get("ancil_dir", ""), ctl["fname_ntopOld"]), "r", mmap=False)
    v = f.variables
    name = lambda k: ctl.get("varname_" + k, k)
    if name("islake") not in v:
        raise ValueError(f"<is_lake_sim> T but the topology file has no '{name('islake')}'")
    islake = np.asarray(v[name("islake")][:], dtype=np.int64) == 1
    reach = (np.nonzero(islake)[0] + 1).astype(np.int32)
    mtype = np.asarray(v[name("lakeModelType")][:], dtype=np.int32)[islake] if name("lakeModelType") in v else np.ones(reach.size, np.int32)
    if not _truth(ctl.get("lakeRegulate", "T")):
        mtype = np.ones(reach.size, np.int32)              # every lake natural (Doll), public_var.f90:105
    par = np.zeros((NLAKEPAR, reach.size))
    for i, k in enumerate(LAKE_PAR):
        if name(k) in v:
            par[i] = np.asarray(v[name(k)][:], dtype=np.float64)[islake]
    out = dict(reach=reach, model_type=mtype, par=par, input_option=int(ctl.get("LakeInputOption", 0)),
               calendar_id=0 if ctl.get("calendar", "standard").strip().lower() in ("noleap", "365_day") else 1)
    if _truth(ctl.get("is_vol_wm", "F")):
        flag = np.asarray(v[name("LakeTargVol")][:], dtype=np.int32)[islake] if name("LakeTargVol") in v else np.zeros(reach.size, np.int32)
        out.update(targ_vol=flag, vol_jumpstart=int(_truth(ctl.get("is_vol_wm_jumpstart", "F"))))
    f.close()
    return out


def read_gauges(ctl: dict, net, t_beg, dt: float, n_steps: int) -> dict:
    """Gauge observations for direct insertion (`<qmodOption> 1`): `<gageMetaFile>` (csv with a header naming `gage_id` and
    `reach_id`, gageMeta_data.f90:52-68) links the sites of `<fname_gageObs>` (`<vname_gageSite>` character array,
    `<vname_gageTime>`, `<vname_gageFlow>(time, site)`) to reaches (obs_data.f90:717-743); a simulation step has observations
    when its start time is one of the file's times (get_time_ix).  Returns gauge_reach (1-based, -9999 = not in the
    network), have[nSteps], obs[nSteps, nGauge]."""
    import csv
    with open(os.path.join(ctl.get("ancil_dir", ""), ctl["gageMetaFile"])) as fp:
        rows = list(csv.DictReader(fp, skipinitialspace=True))
    site2reach = {r["gage_id"].strip(): int(float(r["reach_id"])) for r in rows}
    f = netcdf_file(os.path.join(ctl.get("ancil_dir", ""), ctl["fname_gageObs"]), "r", mmap=False)
    raw = np.asarray(f.variables[ctl.get("vname_gageSite", "site")][:])
    sites = [b"".join(np.atleast_1d(x).astype("S1").tolist()).decode().strip().strip("\x00") for x in raw] if raw.dtype.kind == "S" and raw.ndim == 2 \
        else [str(x).strip() for x in raw]
    tobs = _time_axis(f.variables[ctl.get("vname_gageTime", "time")])
    flow = np.asarray(f.variables[ctl.get("vname_gageFlow", "flow")][:], dtype=np.float64)
    fill = getattr(f.variables[ctl.get("vname_gageFlow", "flow")], "_FillValue", None)
    if fill is not None:
        flow = np.where(flow == fill, np.nan, flow)
    f.close()
    pos = {int(x): i + 1 for i, x in enumerate(net.reachId)}
    gauge_reach = np.array([pos.get(site2reach.get(sn, -1), -9999) for sn in sites], dtype=np.int32)
    t0 = (t_beg - _dt.datetime(1970, 1, 1)).total_seconds()
    rec = {float(t): i for i, t in enumerate(tobs)}
    have = np.zeros(n_steps, np.int32); obs = np.full((n_steps, len(sites)), np.nan)
    for k in range(n_steps):
        i = rec.get(t0 + k * dt)
        if i is not None:
            have[k] = 1; obs[k] = flow[i]
    return dict(gauge_reach=gauge_reach, have=have, obs=obs)


def run(control_path: str, window: int = 1024, device: int = 0, log=print) -> dict:
    ctl = read_control(control_path)
    if int(float(ctl.get("seg_outlet", -9999))) > 0:
        return write_subset(ctl, log)
    nml = read_param_nml(os.path.join(ctl.get("ancil_dir", ""), ctl["param_nml"]))
    net, hru_id = build_network(ctl, nml)
    dt = float(ctl.get("dt_qsim", 86400))
    dt_ro = float(ctl.get("dt_ro", dt))
    methods = [int(c) for c in ctl.get("route_opt", "0")]
    t_beg, t_end = _parse_time(ctl["sim_start"]), _parse_time(ctl["sim_end"])
    epoch = _dt.datetime(1970, 1, 1)
    n_steps = int(round((t_end - t_beg).total_seconds() / dt)) + 1          # both ends included (init_model_data.f90)
    tc, lc = unit_factors(ctl.get("units_qsim", "m/s"))
    frac = uhmod.basin_uh(dt, nml["fshape"], nml["tscale"])
    uh_off, uhv = uhmod.make_uh(net.params["RLENGTH"], dt, nml["velo"], nml["diff"])
    W = max(1, min(window, n_steps))
    # options of the reference this driver does not implement stop the run instead of being ignored (read_control.f90)
    qmod = str(ctl.get("qmodOption", "0")).strip()
    if qmod not in ("F", "0", "", "1"):
        raise ValueError(f"<qmodOption> = {qmod}: expected 0 (none) or 1 (direct insertion), main_route.f90:125-148")
    qmod = qmod == "1"
    tracer = _truth(ctl.get("tracer", "F"))
    if tracer:      # units of the constituent flux, read_control.f90:476-506
        cc = ctl.get("units_cc", "mg/s")
        if "/" not in cc:
            raise ValueError(f'expect the character "/" exists in the mass flux units string [units={cc}]')
        c_mass, c_time = [u.strip() for u in cc.split("/", 1)]
        mass_conv = {"mg": 1.0, "g": 1000.0, "kg": 1000000.0}[c_mass]
        time_conv_sol = {"d": 1.0 / 86400.0, "day": 1.0 / 86400.0, "h": 1.0 / 3600.0, "hr": 1.0 / 3600.0, "hour": 1.0 / 3600.0, "s": 1.0, "sec": 1.0, "second": 1.0}[c_time]
    is_lake, is_flux_wm = _truth(ctl.get("is_lake_sim", "F")), _truth(ctl.get("is_flux_wm", "F"))
    is_vol_wm = _truth(ctl.get("is_vol_wm", "F")) and is_lake
    if is_vol_wm and int(ctl.get("LakeInputOption", 0)) == 1:
        # the reference reads the target volumes inside its "LakeInputOption 0 or 2" branch only (get_basin_runoff.f90:138-229):
        # with option 1 its lakes would follow volumes that were never read
        raise ValueError("<is_vol_wm> T needs <LakeInputOption> 0 or 2: the target volumes are read together with the lake fluxes")
    lakes = read_lakes(ctl, net, n_steps) if is_lake else None
    # history variables beyond discharge and volume (read_control.f90:239-262, histVars_data.f90): basRunoff defaults to T
    # (instRunoff is forced off without hillslope routing, read_control.f90:708)
    runoff_vars = tuple(k for k, dflt in (("basRunoff", "T"), ("instRunoff", "F"), ("dlayRunoff", "F")) if _truth(ctl.get(k, dflt))
                        and not (k == "instRunoff" and int(ctl.get("doesBasinRoute", 1)) == 0))
    want_runoff = len(runoff_vars) > 0
    want_inflow, want_height = _truth(ctl.get("outputInflow", "F")), _truth(ctl.get("floodplain", "F"))
    hflags = (api.H_RUNOFF if want_runoff else 0) | (api.H_INFLOW if want_inflow else 0) | (api.H_HEIGHT if want_height else 0)
    dom = api.RoutingDomain(net, dt, methods, frac_future=frac, uh_offset=uh_off, uh=uhv, max_window=W, device=device,
                            does_basin_route=int(ctl.get("doesBasinRoute", 1)), hw_drain_point=int(ctl.get("hw_drain_point", 2)),
                            min_length_route=float(ctl.get("min_length_route", 0.0)), time_conv=tc, length_conv=lc, history=hflags,
                            lakes=lakes, is_flux_wm=int(is_flux_wm))
    gauges = None
    if qmod:
        gauges = read_gauges(ctl, net, t_beg, dt, n_steps)
        dom.set_da(dict(blend=int(ctl.get("qBlendPeriod", 10)), trend=int(ctl.get("QerrTrend", 1)), **gauges))
    if tracer:
        dom.enable_tracer(time_conv_sol, mass_conv)      # (before read_restart: the file's tfuture / solute_mass are taken only then)
        sol_sum = {mm: np.zeros(net.N) for mm in methods if mm != api.SUM}
        sol_local = np.zeros(net.N)
    # ---- forcing: concatenate the files' time axes, find the record of every simulation step
    t0 = (t_beg - epoch).total_seconds()
    fro = ForcingSeries(_forcing_files(ctl), ctl["vname_time"], dt_ro, t0)
    handles = fro.handles
    fwm = None
    if is_flux_wm or is_vol_wm:
        fwm = ForcingSeries(_wm_files(ctl), ctl.get("vname_time_wm", "time"), float(ctl.get("dt_wm", dt)), t0)
        wm_seg = np.asarray(fwm.var(ctl["vname_segid_wm"])[:], dtype=np.int64)
        pos_seg = {int(x): i + 1 for i, x in enumerate(net.reachId)}
        wm_ix = np.array([pos_seg.get(int(x), -9999) for x in wm_seg], dtype=np.int64)        # match_index, model_setup.f90:897
    fnum = lambda k: float(str(ctl.get(k, REAL_MISSING)).lower().replace("d", "e"))
    sc_ro, of_ro = fnum("scale_factor_runoff"), fnum("offset_value_runoff")
    sc_ep, of_ep, sc_pr, of_pr = fnum("scale_factor_Ep"), fnum("offset_value_Ep"), fnum("scale_factor_prec"), fnum("offset_value_prec")
    remap = _truth(ctl.get("is_remap", "F"))
    if remap:
        m = netcdf_file(os.path.join(ctl.get("ancil_dir", ""), ctl["fname_remap"]), "r", mmap=False)
        mv = m.variables
        pos_rn = {int(x): i + 1 for i, x in enumerate(hru_id)}
        hix = np.array([pos_rn.get(int(x), -9999) for x in mv[ctl["vname_hruid_in_remap"]][:]], dtype=np.int32)
        num = np.asarray(mv[ctl["vname_num_qhru"]][:], dtype=np.int32)
        wgt = np.asarray(mv[ctl["vname_weight"]][:], dtype=np.float64)
        vi, vj = ctl.get("vname_i_index", "i_index"), ctl.get("vname_j_index", "j_index")
        if vi in mv and vj in mv:
            # gridded runoff, runoff(time, lat, lon): the mapping names grid boxes (remap_2D_runoff, process_remap.f90:107-177;
            # read_runoff.f90 option 3); i runs along <dname_xlon>, j along <dname_ylat>
            qv = handles[0].variables[ctl["vname_qsim"]]
            if len(qv.shape) != 3:
                raise ValueError("the mapping file has i_index / j_index but the runoff variable is not (time, lat, lon)")
            n2, n1 = int(qv.shape[1]), int(qv.shape[2])
            mp = dict(hru_ix=hix, num_qhru=num, weight=wgt, i_index=np.asarray(mv[vi][:], dtype=np.int32), j_index=np.asarray(mv[vj][:], dtype=np.int32),
                      n1=n1, n2=n2, H=net.H)
        else:
            src_id = np.asarray(handles[0].variables[ctl["vname_hruid"]][:], dtype=np.int64)
            pos_src = {int(x): i + 1 for i, x in enumerate(src_id)}
            qid = np.asarray(mv[ctl["vname_qhruid"]][:], dtype=np.int64)
            mp = dict(hru_ix=hix, num_qhru=num, weight=wgt,
                      qhru_ix=np.array([pos_src.get(int(x), -9999) for x in qid], dtype=np.int32), qhru_id=qid, src_id=src_id,
                      n1=src_id.size, n2=0, H=net.H)
        m.close()
        dom.set_remap(mp)
    else:
        src_id = np.asarray(handles[0].variables[ctl["vname_hruid"]][:], dtype=np.int64)
        pos_rn = {int(x): i + 1 for i, x in enumerate(hru_id)}
        dom.set_sort_map(np.array([pos_rn.get(int(x), -9999) for x in src_id], dtype=np.int32), True)
    # ---- restart in
    t_first = 0.0
    state_in = ctl.get("fname_state_in", "coldstart")
    if state_in and state_in.lower() != "coldstart" and not state_in.isupper():
        tb = ncfiles.read_restart(os.path.join(ctl.get("output_dir", ""), state_in), dom)
        t_first = float(tb[1])
        log(f"restart from {state_in}: time_bound = {tb}")
    # ---- history
    of = ctl.get("outputFrequency", "1")
    every = int(round(86400.0 / dt)) if of == "daily" else int(of)
    os.makedirs(ctl.get("output_dir", "."), exist_ok=True)
    # one history file per <newFileFrequency> period (single / daily / monthly / yearly, default yearly), named by the period's
    # first step (write_simoutput_pio.f90:328-380, period-start stamp)
    nff = ctl.get("newFileFrequency", "yearly").strip()
    if nff not in ("single", "daily", "monthly", "yearly"):
        raise ValueError(f"<newFileFrequency> {nff}: expected single, daily, monthly or yearly")

    def period(t):
        return {"single": 0, "daily": (t.year, t.month, t.day), "monthly": (t.year, t.month), "yearly": t.year}[nff]

    def hist_name(t):
        stamp = {"single": f"{t:%Y-%m-%d}-{t.hour * 3600 + t.minute * 60 + t.second:05d}", "daily": f"{t:%Y-%m-%d}-{t.hour * 3600 + t.minute * 60 + t.second:05d}",
                 "monthly": f"{t:%Y-%m}", "yearly": f"{t:%Y}"}[nff]
        return os.path.join(ctl.get("output_dir", ""), f"{ctl['case_name']}.h.{stamp}.nc")

    def hist_open(t):
        return ncfiles.HistoryWriter(hist_name(t), net.reachId, methods, time_units=f"seconds since {t_beg:%Y-%m-%d %H:%M:%S}",
                                     volumes=any(_truth(ctl.get(k, "F")) for k in ncfiles.HIST_VOL.values()), inflow=want_inflow, height=want_height,
                                     runoff=runoff_vars, hru_id=hru_id, solute=tracer)

    hfiles = [hist_name(t_beg)]
    hname = hfiles[0]
    hist = hist_open(t_beg)
    hist_period = period(t_beg)
    import torch
    dev = torch.device("cuda", device)
    done = 0
    qname = ctl["vname_qsim"]
    while done < n_steps:
        w = min(W, n_steps - done, every - (done % every))                 # a window never straddles an output record
        rows = [scale_forcing(fro.step(qname, dt, done + k + 1), sc_ro, of_ro) for k in range(w)]   # forcing of every simulation step
        if lakes is not None:
            # lake fluxes of the window: evaporation / precipitation through the same mapping as runoff (on the device),
            # the calendar of every step, the target volumes (get_basin_runoff.f90:136-229)
            lk = dom.lakes
            days = [t_beg + _dt.timedelta(seconds=(done + k) * dt) for k in range(w)]
            lk["ymd"] = np.array([[d.year, d.month, d.day] for d in days], dtype=np.int64)
            flux_dev = {}
            for key, vname, sc, of, flip in (("evap", ctl.get("vname_evapo", "evap"), sc_ep, of_ep, _truth(ctl.get("is_Ep_upward_negative", "F"))),
                                             ("precip", ctl.get("vname_precip", "precip"), sc_pr, of_pr, False)):
                dstf = torch.zeros((w, net.H), dtype=torch.float64, device=dev)
                torch.cuda.synchronize()      # (the fill is done before the library's own stream writes into it)
                flux_dev[key] = dstf
                if lk["input_option"] == 1 or suppressed(sc, of):
                    continue
                a = np.stack([fro.step(vname, dt, done + k + 1) for k in range(w)])
                if flip:
                    a = scale_forcing(a, -1.0, 0.0)
                a = scale_forcing(a, sc, of)
                srcf = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
                dom.remap_device(w, srcf.data_ptr(), dstf.data_ptr())
            torch.cuda.synchronize()      # (torch's fills and copies run on its stream; the library's streams do not wait for it)
            dom.sync()
            if is_vol_wm:
                lk["wm_vol"] = np.stack([sort_flux(wm_ix, fwm.step(ctl["vname_vol_wm"], dt, done + k + 1), net.N, True) for k in range(w)])
            dom.set_lake_forcing(0, w, flux_dev["evap"].data_ptr(), flux_dev["precip"].data_ptr())
        if is_flux_wm:
            dom.set_wm_flux(w, np.stack([sort_flux(wm_ix, fwm.step(ctl["vname_flux_wm"], dt, done + k + 1), net.N, False) for k in range(w)]))
        if tracer:      # the constituent takes the runoff's path from the file to the river-network HRUs (get_basin_runoff.f90:111-134)
            a = np.stack([fro.step(ctl.get("vname_solute", "solute"), dt, done + k + 1) for k in range(w)])
            srcs = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            dsts = torch.empty((w, net.H), dtype=torch.float64, device=dev)
            dom.remap_device(w, srcs.data_ptr(), dsts.data_ptr())
            dom.sync()
            dom._check(dom.L.mzr_set_solute(dom.h, w, np.ascontiguousarray(dsts.cpu().numpy())))
        if gauges is not None:
            dom.set_obs(done, w)
        src = torch.from_numpy(np.ascontiguousarray(np.stack(rows))).to(dev)
        dom.run_source_device(w, t_first + done * dt, src.data_ptr())
        dom.sync()
        if tracer:      # interval sums of the constituent fluxes (the history file holds their means)
            buf = np.zeros((w, net.N))
            dom._check(dom.L.mzr_get_window_solute(dom.h, -1, buf)); sol_local += buf.sum(axis=0)
            for mm in sol_sum:
                dom._check(dom.L.mzr_get_window_solute(dom.h, mm, buf)); sol_sum[mm] += buf.sum(axis=0)
        done += w
        if done % every == 0:      # time = start of the aggregated interval (+ <histTimeStamp_offset>), historyFile.f90:367
            t_rec = t_beg + _dt.timedelta(seconds=(done - every) * dt)
            if period(t_rec) != hist_period:          # the record opens a new period: new file
                hist.close()
                hist, hist_period = hist_open(t_rec), period(t_rec)
                hfiles.append(hist_name(t_rec))
            if tracer:
                hist.append_solute(sol_local / every, {mm: v / every for mm, v in sol_sum.items()}, {mm: dom.solute_state(mm, 1) for mm in sol_sum})
                sol_local[:] = 0.0
                for v in sol_sum.values():
                    v[:] = 0.0
            hist.append((done - every) * dt, done * dt, dom, stamp_offset=float(ctl.get("histTimeStamp_offset", 0.0)))
    hist.close()
    out = dict(history=hname, history_files=hfiles, steps=n_steps, reaches=net.N)
    if ctl.get("restart_write", "never").lower() == "last":
        t_rst = t_end + _dt.timedelta(seconds=dt)        # the restart time is the END of the last step (write_restart_pio.f90:207-253,771)
        rname = os.path.join(ctl.get("output_dir", ""), f"{ctl['case_name']}.r.{t_rst:%Y-%m-%d}-{t_rst.hour * 3600 + t_rst.minute * 60 + t_rst.second:05d}.nc")
        ncfiles.write_restart(rname, dom, net.reachId, (t_first + (n_steps - 1) * dt, t_first + n_steps * dt), restart_time=t_first + n_steps * dt)
        out["restart"] = rname
        # restart pointer file: names of the last restart and history files (io_rpointfile.f90:23-75)
        with open(os.path.join(ctl.get("restart_dir", ctl.get("output_dir", "")), ctl.get("rpntfil", "rpointer.rof")), "w") as fp:
            fp.write(rname + "\n" + hfiles[-1] + "\n")
        out["rpointer"] = fp.name
    fro.close()
    if fwm is not None:
        fwm.close()
    dom.close()
    log(f"routed {net.N} reaches x {n_steps} steps -> {hname}")
    return out


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("control")
    ap.add_argument("--window", type=int, default=1024)
    a = ap.parse_args()
    run(a.control, a.window)
