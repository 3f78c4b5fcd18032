"""Seeded synthetic river networks and runoff forcing.

The reference ships no test data (route/input/ holds only a README), so every parity case and
the benchmark run on networks grown here.  The generator is a level-synchronous random
binary-merge (Shreve) tree grown upstream from the outlets, P(2 upstreams) ~ 0.48, P(1) = 0.04,
P(0) ~ 0.48, with a mild width controller so the network reaches N reaches with roughly
`depth` hops along the longest path instead of dying out like a critical branching process.

Reach attributes follow the reference's river-network file variables
(docs/source/users_guide/Input_files.rst:58-118) and the defaults of
route/ancillary_data/param.nml.default:1-12; derived attributes follow
process_ntopo.f90:176-197 (width = wscale*sqrt(totalArea), storage = area(depth)*length).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

HIGH_DEPTH = 100000.0  # globalData.f90:189


@dataclass
class RiverNetwork:
    """Arrays use the reference's conventions: reach / HRU indices are 1-based, downIndex <= 0
    marks an outlet, immediate upstreams are stored CSR-style in UREACHI order."""

    N: int
    H: int
    downIndex: np.ndarray      # int32 [N]
    reachId: np.ndarray        # int32 [N]
    upOffset: np.ndarray       # int32 [N+1]
    upIndex: np.ndarray        # int32 [nUp]
    upGood: np.ndarray         # int32 [nUp]   goodBas flag per upstream slot
    hruOffset: np.ndarray      # int32 [N+1]
    hruIndex: np.ndarray       # int32 [nHru]
    hruWeight: np.ndarray      # float64 [nHru]
    params: dict = field(default_factory=dict)   # name -> float64 [N]

    PARAM_ORDER = ("R_SLOPE", "R_MAN_N", "R_WIDTH", "R_DEPTH", "RLENGTH", "R_STORAGE",
                   "SIDE_SLOPE", "FLDP_SLOPE", "BASAREA", "TOTAREA", "MINFLOW")

    def param_matrix(self) -> np.ndarray:
        """[11, N] row-major in PARAM_ORDER."""
        return np.ascontiguousarray(np.stack([self.params[k] for k in self.PARAM_ORDER]).astype(np.float64))

    def topo_order(self) -> np.ndarray:
        """A processing order (0-based reach indices), upstream before downstream."""
        down = self.downIndex.astype(np.int64) - 1
        dist = hops_to_outlet(down)
        return np.argsort(-dist, kind="stable").astype(np.int32)

    def n_levels(self) -> int:
        return int(hops_to_outlet(self.downIndex.astype(np.int64) - 1).max()) + 1


def hops_to_outlet(down0: np.ndarray) -> np.ndarray:
    """Number of hops from each reach to its outlet (outlet = 0). down0: 0-based, <0 = outlet."""
    n = down0.shape[0]
    dist = np.zeros(n, dtype=np.int64)
    ptr = down0.copy()
    active = np.nonzero(ptr >= 0)[0]
    # pointer-jumping would be O(log depth); a plain frontier walk from the outlets is simpler:
    # build children lists and BFS.
    order = np.argsort(down0, kind="stable")
    sorted_down = down0[order]
    starts = np.searchsorted(sorted_down, np.arange(n), side="left")
    ends = np.searchsorted(sorted_down, np.arange(n), side="right")
    frontier = np.nonzero(down0 < 0)[0]
    d = 0
    while frontier.size:
        dist[frontier] = d
        cnt = ends[frontier] - starts[frontier]
        tot = int(cnt.sum())
        if tot == 0:
            break
        base = np.repeat(starts[frontier], cnt)
        off = np.arange(tot) - np.repeat(np.cumsum(cnt) - cnt, cnt)
        frontier = order[base + off]
        d += 1
    del active, ptr
    return dist


def build_upstream_csr(downIndex: np.ndarray):
    """UREACHI lists from downstream indices (1-based in, 1-based out), ordered by reach index."""
    n = downIndex.shape[0]
    down0 = downIndex.astype(np.int64) - 1
    has = down0 >= 0
    src = np.nonzero(has)[0]
    dst = down0[has]
    order = np.argsort(dst, kind="stable")
    counts = np.bincount(dst, minlength=n)
    upOffset = np.zeros(n + 1, dtype=np.int32)
    upOffset[1:] = np.cumsum(counts)
    upIndex = (src[order] + 1).astype(np.int32)
    return upOffset, upIndex


def make_network(N: int, seed: int = 20240529, depth: int | None = None, n_outlets: int | None = None,
                 floodplain: bool = False, zero_area_frac: float = 0.0, p3: float = 0.0) -> RiverNetwork:
    rng = np.random.default_rng(seed)
    if depth is None:
        depth = max(4, int(round(2.6 * np.sqrt(N))))          # ~800 at 100k, ~4.5k at 3M
    if n_outlets is None:
        n_outlets = max(1, N // 50000)
    n_outlets = min(n_outlets, N)
    w_target = max(2.0, N / depth)
    down = np.full(N, 0, dtype=np.int64)                       # 1-based downstream index, 0 = outlet
    frontier = np.arange(n_outlets, dtype=np.int64)
    n_made = n_outlets
    while n_made < N:
        w = frontier.size
        if w == 0:                                             # everything died out: new coastal outlet
            frontier = np.array([n_made], dtype=np.int64)
            n_made += 1
            continue
        p2 = float(np.clip(0.48 + 0.15 * (1.0 - w / w_target), 0.30, 0.66))
        p1 = 0.04
        u = rng.random(w)
        nchild = np.where(u < p2, 2, np.where(u < p2 + p1, 1, 0))
        if p3 > 0:                                             # occasional triple confluence
            nchild = np.where(rng.random(w) < p3, 3, nchild)
        if w < 4:
            nchild = np.maximum(nchild, 1)
        tot = int(nchild.sum())
        if tot == 0:
            nchild[rng.integers(w)] = 1
            tot = 1
        if n_made + tot > N:                                   # trim the last level
            keep = N - n_made
            cs = np.cumsum(nchild)
            nchild = np.where(cs <= keep, nchild, np.maximum(0, nchild - (cs - keep)))
            tot = int(nchild.sum())
        parents = np.repeat(frontier, nchild)
        children = np.arange(n_made, n_made + tot, dtype=np.int64)
        down[children] = parents + 1
        n_made += tot
        frontier = children
    # shuffle reach numbering so that array order carries no topological information
    perm = rng.permutation(N)                                   # new index of old reach i is perm[i]
    down_new = np.zeros(N, dtype=np.int64)
    has = down > 0
    down_new[perm[has]] = perm[down[has] - 1] + 1
    downIndex = down_new.astype(np.int32)
    upOffset, upIndex = build_upstream_csr(downIndex)

    length = np.clip(np.exp(rng.normal(np.log(3000.0), 0.6, N)), 200.0, 30000.0)
    slope = np.maximum(np.exp(rng.uniform(np.log(1e-4), np.log(5e-2), N)), 1e-6)   # min_slope, public_var.f90:30
    basarea = np.exp(rng.normal(np.log(2.5e7), 0.5, N))
    if zero_area_frac > 0:
        # a few headwater reaches without any contributing area (exercises goodBas = .false.)
        head = np.nonzero(np.diff(upOffset) == 0)[0]
        z = head[rng.random(head.size) < zero_area_frac]
        basarea[z] = 0.0
    # accumulate total area upstream -> downstream
    totarea = basarea.copy()
    down0 = downIndex.astype(np.int64) - 1
    dist = hops_to_outlet(down0)
    for d in range(int(dist.max()), 0, -1):
        idx = np.nonzero(dist == d)[0]
        np.add.at(totarea, down0[idx], totarea[idx])
    wscale, dscale = 0.001, 0.0006
    width = wscale * np.sqrt(np.maximum(totarea, 1.0))
    if floodplain:
        rdepth = dscale * np.sqrt(np.maximum(totarea, 1.0))
    else:
        rdepth = np.full(N, HIGH_DEPTH)
    side = np.zeros(N)
    storage = rdepth * (width + side * rdepth) * length        # hydraulic.f90:207-238 storage()
    params = dict(R_SLOPE=slope, R_MAN_N=np.full(N, 0.01), R_WIDTH=width, R_DEPTH=rdepth,
                  RLENGTH=length, R_STORAGE=storage, SIDE_SLOPE=side, FLDP_SLOPE=np.full(N, 1000.0),
                  BASAREA=basarea, TOTAREA=totarea, MINFLOW=np.zeros(N))
    # goodBas: all upstream slots of a reach share the flag "own total area > verySmall"
    # (network_topo.f90:769-775)
    good_reach = (totarea > np.finfo(np.float64).tiny).astype(np.int32)
    upGood = np.repeat(good_reach, np.diff(upOffset)).astype(np.int32)
    hruOffset = np.arange(N + 1, dtype=np.int32)               # one HRU per reach, weight 1
    hruIndex = np.arange(1, N + 1, dtype=np.int32)
    hruWeight = np.ones(N)
    return RiverNetwork(N=N, H=N, downIndex=downIndex, reachId=np.arange(1001, 1001 + N, dtype=np.int32),
                        upOffset=upOffset, upIndex=upIndex, upGood=upGood, hruOffset=hruOffset,
                        hruIndex=hruIndex, hruWeight=hruWeight, params=params)


def make_star_network(k: int = 5, depth: int = 6, seed: int = 0, identical: bool = True) -> RiverNetwork:
    """k perfect binary trees of `depth` levels whose roots meet in ONE confluence reach that drains
    to an outlet reach.  Reaches of the same tree and level share all parameters, so that with
    spatially uniform runoff sibling tributaries deliver particles with identical times (the
    duplicate-time branch of qexmul_rch), while the k trees differ from each other, so that the
    k-way confluence receives k particle lists.  identical=False draws every reach's parameters
    independently instead: no duplicate times, every list grows to MAXQPAR, and the k-way confluence
    holds more than 64 particles before remove_rch."""
    rng = np.random.default_rng(seed)
    per = 2 ** depth - 1
    N = k * per + 2
    down = np.zeros(N, dtype=np.int64)               # 1-based, 0 = outlet
    level = np.zeros(N, dtype=np.int64)
    tree = np.zeros(N, dtype=np.int64)
    conf, outlet = N - 2, N - 1
    down[conf] = outlet + 1
    for j in range(k):
        base = j * per
        for i in range(per):                         # heap order: children of i are 2i+1, 2i+2
            down[base + i] = (conf if i == 0 else base + (i - 1) // 2) + 1
            level[base + i] = int(np.floor(np.log2(i + 1)))
            tree[base + i] = j
    level[conf] = level[outlet] = -1
    tree[conf] = tree[outlet] = k
    downIndex = down.astype(np.int32)
    upOffset, upIndex = build_upstream_csr(downIndex)
    lvl_len = np.clip(np.exp(rng.normal(np.log(3000.0), 0.4, (k + 1, depth + 1))), 500.0, 12000.0)
    lvl_slope = np.exp(rng.uniform(np.log(5e-4), np.log(2e-2), (k + 1, depth + 1)))
    length = lvl_len[tree, level + 1]
    slope = lvl_slope[tree, level + 1]
    if not identical:
        length = np.clip(np.exp(rng.normal(np.log(3000.0), 0.6, N)), 200.0, 30000.0)
        slope = np.exp(rng.uniform(np.log(1e-4), np.log(5e-2), N))
    basarea = np.full(N, 2.5e7) * (1.0 + 0.1 * tree)
    totarea = basarea.copy()
    down0 = downIndex.astype(np.int64) - 1
    dist = hops_to_outlet(down0)
    for d in range(int(dist.max()), 0, -1):
        idx = np.nonzero(dist == d)[0]
        np.add.at(totarea, down0[idx], totarea[idx])
    width = 0.001 * np.sqrt(totarea)
    rdepth = np.full(N, HIGH_DEPTH)
    side = np.zeros(N)
    params = dict(R_SLOPE=slope, R_MAN_N=np.full(N, 0.01), R_WIDTH=width, R_DEPTH=rdepth,
                  RLENGTH=length, R_STORAGE=rdepth * width * length, SIDE_SLOPE=side,
                  FLDP_SLOPE=np.full(N, 1000.0), BASAREA=basarea, TOTAREA=totarea, MINFLOW=np.zeros(N))
    upGood = np.ones(upIndex.size, dtype=np.int32)
    return RiverNetwork(N=N, H=N, downIndex=downIndex, reachId=np.arange(1001, 1001 + N, dtype=np.int32),
                        upOffset=upOffset, upIndex=upIndex, upGood=upGood,
                        hruOffset=np.arange(N + 1, dtype=np.int32), hruIndex=np.arange(1, N + 1, dtype=np.int32),
                        hruWeight=np.ones(N), params=params)


IMISS = -9999   # integerMissing, public_var.f90:44


def make_remap(H: int, n1: int, n2: int = 0, seed: int = 0, max_overlap: int = 6, missing_frac: float = 0.03) -> dict:
    """A mapping file's content (dataTypes.f90:132-143) from a runoff layer to H river-network HRUs:
    n2 = 0: the runoff layer is a vector of n1 polygons (remap_1D_runoff); n2 > 0: an n1 x n2 grid
    (remap_2D_runoff).  Rows come in shuffled order, a few rows name HRUs that are not in the network
    (hru_ix = integerMissing), a few overlaps name polygons/cells that are not in the runoff file,
    most rows have weights that sum to one and some do not (the renormalisation branch)."""
    rng = np.random.default_rng(seed)
    n_extra = max(1, int(missing_frac * H))
    hru_ix = np.concatenate([rng.permutation(H) + 1, np.full(n_extra, IMISS)]).astype(np.int32)
    hru_ix = hru_ix[rng.permutation(hru_ix.size)]
    nMap = hru_ix.size
    num = rng.integers(0, max_overlap + 1, nMap).astype(np.int32)
    num[rng.random(nMap) < 0.01] = 0
    extra = np.nonzero(hru_ix == IMISS)[0]
    if extra.size:
        num[extra[0]] = IMISS                      # "num_qhru missing too": the cursor does not advance
    n_ov = int(num[num > 0].sum())
    w = rng.random(n_ov) + 0.05
    row = np.repeat(np.arange(nMap)[num > 0], num[num > 0])
    tot = np.bincount(row, weights=w, minlength=nMap)
    norm = rng.random(nMap) < 0.8                   # these rows carry normalised weights
    w = np.where(norm[row], w / tot[row], w)
    out = dict(hru_ix=hru_ix, num_qhru=num, weight=w, n1=n1, n2=n2, H=H)
    if n2 == 0:
        q = rng.integers(1, n1 + 1, n_ov).astype(np.int32)
        src_id = (rng.permutation(n1) + 700001).astype(np.int64)
        qid = src_id[q - 1].copy()
        q[rng.random(n_ov) < missing_frac] = IMISS
        out.update(qhru_ix=q, qhru_id=qid, src_id=src_id)
    else:
        ii = rng.integers(1, n1 + 1, n_ov).astype(np.int32)
        jj = rng.integers(1, n2 + 1, n_ov).astype(np.int32)
        bad = rng.random(n_ov) < missing_frac        # cells outside the runoff grid
        ii[bad & (rng.random(n_ov) < 0.5)] = n1 + 3
        jj[bad & (ii <= n1)] = 0
        out.update(i_index=ii, j_index=jj)
    return out


def make_source_runoff(n_steps: int, n1: int, n2: int = 0, seed: int = 0) -> np.ndarray:
    """runoff of the hydrologic model's own layer, [n_steps, n1] or [n_steps, n2, n1] (grid rows in the
    reference's memory order), with a few negative / fill values that the remap must skip."""
    rng = np.random.default_rng(seed)
    shape = (n_steps, n1) if n2 == 0 else (n_steps, n2, n1)
    ro = 1e-8 * (1.0 + rng.random(shape)) + np.where(rng.random(shape) < 0.02, 1e-6 * rng.random(shape), 0.0)
    ro[rng.random(shape) < 0.02] = -9999.0
    ro[rng.random(shape) < 0.01] = -1e-7            # inside the tolerance (-1e-6): still used
    return np.ascontiguousarray(ro)


def make_runoff(H: int, n_steps: int, seed: int = 7, t0: int = 0, base: float = 1e-8,
                storm_prob: float = 0.01, storm_amp: float = 1e-6) -> np.ndarray:
    """runoff[t, h] in m/s: low seasonal base flow plus sparse storm pulses."""
    rng = np.random.default_rng(seed)
    phi = rng.uniform(0, 2 * np.pi, H)
    t = np.arange(t0, t0 + n_steps, dtype=np.float64)[:, None]
    ro = base * (1.0 + np.sin(2 * np.pi * t / 168.0 + phi[None, :]))
    rng2 = np.random.default_rng(seed + 1000003 * (t0 + 1))
    pulses = rng2.random((n_steps, H))
    amp = rng2.random((n_steps, H))
    ro += np.where(pulses < storm_prob, storm_amp * amp * amp, 0.0)
    return np.ascontiguousarray(ro)


def step_dates(n_steps: int, dt: float, start=(2001, 1, 1), calendar_id: int = 0) -> np.ndarray:
    """(year, month, day) of the START of every step; calendar_id 0 = noleap, 1 = standard."""
    mdays = [31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31]
    y, mo, d = start
    out = np.zeros((n_steps, 3), dtype=np.int32)
    sec = 0.0
    for t in range(n_steps):
        out[t] = (y, mo, d)
        sec += dt
        while sec >= 86400.0:
            sec -= 86400.0
            leap = calendar_id == 1 and ((y % 4 == 0 and y % 100 != 0) or y % 400 == 0)
            nd = mdays[mo - 1] + (1 if (mo == 2 and leap) else 0)
            d += 1
            if d > nd:
                d = 1; mo += 1
                if mo > 12:
                    mo = 1; y += 1
    return out


def make_lakes(net: RiverNetwork, n_steps: int, dt: float, seed: int = 5, frac: float = 0.02, calendar_id: int = 0,
               input_option: int = 0, memory: bool = False, start=(2001, 1, 1), demand_memory: bool = False,
               target_frac: float = 0.0, vol_jumpstart: bool = False, forcing: bool = True) -> dict:
    """Synthetic lakes/reservoirs (Doll 70 %, Hanasaki 25 %, HYPE 5 %, plus an
    endorheic one), parameters in the ranges of docs/source/users_guide/lake.rst.  A lake must be
    the only upstream of its outlet reach (kwt_route.f90:551-553)."""
    from .lakepar import LAKE_PAR, NLAKEPAR
    rng = np.random.default_rng(seed)
    down0 = net.downIndex.astype(np.int64) - 1
    nup = np.diff(net.upOffset)
    elig = np.nonzero(((down0 < 0) | (nup[np.maximum(down0, 0)] == 1)) & (nup > 0))[0]
    # no lake directly below another lake's outlet chain collision: keep lakes non-adjacent
    rng.shuffle(elig)
    chosen, taken = [], set()
    for r in elig:
        if len(chosen) >= max(4, int(frac * net.N)):
            break
        d = int(down0[r])
        ups = set(int(u) - 1 for u in net.upIndex[net.upOffset[r]:net.upOffset[r + 1]])
        if r in taken or d in taken or (ups & taken):
            continue
        chosen.append(int(r)); taken.update([int(r), d] + list(ups))
    reach = np.array(sorted(chosen), dtype=np.int32) + 1
    nl = reach.size
    u = rng.random(nl)
    model = np.where(u < 0.65, 1, np.where(u < 0.9, 2, np.where(u < 0.97, 3, 0))).astype(np.int32)
    if nl >= 3:
        model[:3] = [1, 2, 3]                          # make sure every type occurs
    # an endorheic lake has no outflow, so it can only sit at an outlet (KWT aborts on zero flow below it)
    is_out = down0[reach - 1] < 0
    model = np.where((model == 0) & ~is_out, 1, model).astype(np.int32)
    if is_out.any():
        model[np.nonzero(is_out)[0][0]] = 0
    ix = {k: i for i, k in enumerate(LAKE_PAR)}
    par = np.zeros((NLAKEPAR, nl))
    area = net.params["TOTAREA"][reach - 1]
    qmean = 2e-8 * area                                 # rough mean inflow [m3/s]
    par[ix["D03_MaxStorage"]] = qmean * 86400.0 * rng.uniform(20, 200, nl)
    par[ix["D03_Coefficient"]] = rng.uniform(0.005, 0.05, nl)
    par[ix["D03_Power"]] = rng.uniform(1.0, 2.0, nl)
    par[ix["D03_S0"]] = par[ix["D03_MaxStorage"]] * rng.uniform(0.0, 0.2, nl)
    par[ix["HYP_E_zero"]] = 0.0; par[ix["HYP_E_min"]] = 2.0; par[ix["HYP_E_lim"]] = 6.0; par[ix["HYP_E_emr"]] = 9.0
    par[ix["HYP_A_avg"]] = qmean * 86400.0 * 30 / 9.0 + 1e4
    par[ix["HYP_Qrate_emr"]] = qmean * 3; par[ix["HYP_Erate_emr"]] = 1.5
    par[ix["HYP_Qrate_prim"]] = qmean * 1.2; par[ix["HYP_Qrate_amp"]] = 0.3; par[ix["HYP_Qrate_phs"]] = 100
    par[ix["HYP_prim_F"]] = 1; par[ix["HYP_Qsim_mode"]] = (rng.random(nl) < 0.5)
    par[ix["H06_Smax"]] = qmean * 86400.0 * rng.uniform(30, 600, nl)
    par[ix["H06_alpha"]] = 0.85; par[ix["H06_envfact"]] = 0.1; par[ix["H06_S_ini"]] = par[ix["H06_Smax"]] * 0.8
    par[ix["H06_c1"]] = 0.1; par[ix["H06_c2"]] = 0.9; par[ix["H06_exponent"]] = 2.0; par[ix["H06_denominator"]] = 0.5
    par[ix["H06_c_compare"]] = 0.5; par[ix["H06_frac_Sdead"]] = 0.1; par[ix["H06_E_rel_ini"]] = 1.0
    season = 1.0 + 0.5 * np.sin(2 * np.pi * (np.arange(12) + 0.5) / 12.0)
    for mth in range(12):
        par[ix["H06_I_Jan"] + mth] = qmean * season[mth]
        par[ix["H06_D_Jan"] + mth] = qmean * 0.3 * season[(mth + 6) % 12]
    par[ix["H06_purpose"]] = (rng.random(nl) < 0.5)
    par[ix["H06_I_mem_F"]] = 1.0 if memory else 0.0
    par[ix["H06_D_mem_F"]] = 1.0 if demand_memory else 0.0       # the demand is REACH_WM_FLUX: needs is_flux_wm
    par[ix["H06_I_mem_L"]] = 1; par[ix["H06_D_mem_L"]] = 1
    rngf = np.random.default_rng(seed + 77)
    precip = evap = None      # (forcing=False: LakeInputOption 1 runs at full size, where [steps][HRU] host arrays are too much)
    if forcing:
        precip = 3e-8 * (1.0 + rngf.random((n_steps, net.H)))
        evap = 2e-8 * (1.0 + rngf.random((n_steps, net.H)))
    out = dict(input_option=input_option, calendar_id=calendar_id, ymd=step_dates(n_steps, dt, start, calendar_id),
               reach=reach, model_type=model, par=par, evap=evap, precip=precip)
    if target_frac > 0:     # is_vol_wm: some lakes follow a prescribed volume (REACH_WM_VOL per step and reach)
        rt = np.random.default_rng(seed + 99)
        flag = (rt.random(nl) < target_frac).astype(np.int32)
        flag[:2] = 1                                        # at least the first two lakes
        t = np.arange(n_steps)[:, None]
        base = qmean[None, :] * 86400.0 * 40.0
        vol = np.zeros((n_steps, net.N))
        vol[:, reach - 1] = base * (0.6 + 0.35 * np.sin(2 * np.pi * t / 37.0 + np.arange(nl)[None, :]))
        out.update(targ_vol=flag, vol_jumpstart=int(vol_jumpstart), wm_vol=vol)
    return out


def make_gauges(net: RiverNetwork, n_steps: int, n_gauge: int = 40, seed: int = 3, every: int = 3, blend: int = 10, trend: int = 2,
                q_scale: float = 5.0) -> dict:
    """Synthetic gauge observations for direct insertion (qmodOption = 1, data_assimilation.f90): gauges on non-headwater
    reaches (one entry points outside the network), an observation time every `every`-th step with a gap in the middle
    longer than the blending period, values around q_scale * TOTAREA-scaled flows with some missing (NaN) and negative."""
    rng = np.random.default_rng(seed)
    nup = np.diff(net.upOffset)
    cand = np.nonzero(nup > 0)[0]
    reach = (rng.choice(cand, min(n_gauge, cand.size), replace=False) + 1).astype(np.int32)
    reach[-1] = -9999                                     # a gauge that is not linked to any reach
    have = np.zeros(n_steps, np.int32); have[::every] = 1
    have[n_steps // 3: n_steps // 3 + 2 * blend + 3] = 0    # a gap: the error decays and is dropped
    area = np.where(reach > 0, net.params["TOTAREA"][np.maximum(reach, 1) - 1], 1.0)
    obs = q_scale * 2e-8 * area[None, :] * (0.5 + rng.random((n_steps, reach.size)))
    obs[rng.random(obs.shape) < 0.1] = np.nan
    obs[rng.random(obs.shape) < 0.05] = -1.0
    return dict(blend=int(blend), trend=int(trend), gauge_reach=reach, have=have, obs=obs)
