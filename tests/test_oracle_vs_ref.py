"""CPU, build container only: the C oracle against the reference harness on freshly generated
cases (skipped where oracle/_ref/ref_route is not present)."""
import numpy as np
import pytest

from mizuroute_amd.synthetic import make_network, make_runoff
from oracle import refrun

pytestmark = pytest.mark.skipif(not refrun.available(), reason="oracle/_ref/ref_route not built")


@pytest.mark.parametrize("N,seed,dt,steps,kw", [
    (60, 1, 3600.0, 60, dict()),
    (500, 2, 3600.0, 100, dict(p3=0.05)),
    (500, 3, 86400.0, 40, dict(zero_area_frac=0.1)),
    (2000, 4, 1800.0, 150, dict()),
])
def test_all_methods_bit_exact(N, seed, dt, steps, kw, oracle_lib):
    net = make_network(N, seed=seed, **kw)
    ro = make_runoff(net.H, steps, seed=seed + 10, storm_prob=0.05, storm_amp=3e-6)
    methods = [0, 1, 2, 3, 4, 5]
    out = refrun.run_case(net, ro, dt, methods)
    orc = oracle_lib.Oracle(net, dt, methods, out["frac_future"], out["uh_offset"], out["uh"])
    if out["ierr"] != 0:
        # the reference aborts (e.g. kinwav_rch 'zero flow' below a zero-area headwater,
        # kwt_route.f90:1365): the oracle must fail with the same code at the same step
        for it in range(out["ierr_step"]):
            rc = orc.step(it * dt, (it + 1) * dt, ro[it])
            assert rc == (out["ierr"] if it == out["ierr_step"] - 1 else 0), (it, rc, orc.error())
        methods = [m for m in methods if m != 2]
        out = refrun.run_case(net, ro, dt, methods)
        assert out["ierr"] == 0
        orc = oracle_lib.Oracle(net, dt, methods, out["frac_future"], out["uh_offset"], out["uh"])
        assert np.array_equal(orc.run(ro), out["Q"])
        return
    Q, V = orc.run(ro, want_vol=True)
    assert np.array_equal(Q, out["Q"])
    assert np.array_equal(V, out["VOL"])
    nw, qf, ti, tr, rf = orc.kwt_state()
    assert np.array_equal(nw, out["state"][2]["nw"])


@pytest.mark.parametrize("kind,must", [("duplicates", ("duplicate_times", "shock_merges", "merged_leaving", "merged_staying", "exit_time_fixes")),
                                       ("over64", ("removes_over_64", "confluences_over_2"))])
def test_kwt_rare_branches_bit_exact(kind, must, oracle_lib):
    from helpers import star_case
    net, ro, dt = star_case(kind)
    out = refrun.run_case(net, ro, dt, [2])
    assert out["ierr"] == 0
    orc = oracle_lib.Oracle(net, dt, [2], out["frac_future"], out["uh_offset"], out["uh"])
    assert np.array_equal(orc.run(ro), out["Q"])
    assert np.array_equal(orc.kwt_state()[0], out["state"][2]["nw"])
    paths = orc.kwt_paths()
    for k in must:
        assert paths[k] > 0, (k, paths)     # the case does exercise the branch


def test_hw_drain_top_and_no_basin_route(oracle_lib):
    net = make_network(300, seed=9)
    ro = make_runoff(net.H, 50, seed=19, storm_prob=0.05)
    methods = [1, 3, 4, 5, 2]
    out = refrun.run_case(net, ro, 3600.0, methods, hw_drain_point=1, does_basin_route=0)
    orc = oracle_lib.Oracle(net, 3600.0, methods, out["frac_future"], out["uh_offset"], out["uh"],
                            hw_drain_point=1, does_basin_route=0)
    assert np.array_equal(orc.run(ro), out["Q"])


def test_openmp_schedule_gives_same_answer():
    net = make_network(800, seed=5)
    ro = make_runoff(net.H, 30, seed=6)
    a = refrun.run_case(net, ro, 3600.0, [2, 1])
    b = refrun.run_case(net, ro, 3600.0, [2, 1], nthreads=4, schedule=refrun.level_schedule(net))
    assert np.array_equal(a["Q"], b["Q"])
    c = refrun.run_case(net, ro, 3600.0, [2, 1], nthreads=4, schedule=refrun.streamorder_schedule(net))
    assert np.array_equal(a["Q"], c["Q"])


def test_water_management_bit_exact(oracle_lib):
    """is_flux_wm: abstraction cascade, injection and missing values in IRF/KW/MC/DW; for KWT a small
    abstraction (incl. the reference's celerity index shift in extract_from_rch) is bit-exact and an
    injection makes the reference fail in qexmul_rch -- the oracle must fail with the same code."""
    net = make_network(400, seed=31)
    net.params["MINFLOW"] = np.full(net.N, 1e-4)
    steps = 40
    ro = make_runoff(net.H, steps, seed=5, storm_prob=0.05, storm_amp=3e-6)
    rng = np.random.default_rng(3)
    wm = np.zeros((steps, net.N))
    sel = rng.random(net.N) < 0.3
    wm[:, sel] = rng.uniform(0.0, 0.5, sel.sum())[None, :] * (1 + np.sin(np.arange(steps))[:, None])
    wm[:, rng.random(net.N) < 0.1] = -0.1
    wm[:, rng.random(net.N) < 0.03] = -9999.0
    methods = [1, 3, 4, 5]
    out = refrun.run_case(net, ro, 3600.0, methods, wm_flux=wm)
    assert out["ierr"] == 0
    orc = oracle_lib.Oracle(net, 3600.0, methods, out["frac_future"], out["uh_offset"], out["uh"], is_flux_wm=1)
    Q, V = orc.run(ro, want_vol=True, wm_flux=wm)
    assert np.array_equal(Q, out["Q"]) and np.array_equal(V, out["VOL"])
    for label, val, expect_err in (("abstraction", -0.001, 0), ("injection", 0.2, 20)):
        wk = np.zeros((steps, net.N)); wk[:, rng.random(net.N) < 0.2] = val
        out = refrun.run_case(net, ro, 3600.0, [2], wm_flux=wk)
        assert out["ierr"] == expect_err, (label, out["stdout"])
        orc = oracle_lib.Oracle(net, 3600.0, [2], out["frac_future"], out["uh_offset"], out["uh"], is_flux_wm=1)
        if expect_err == 0:
            assert np.array_equal(orc.run(ro, wm_flux=wk), out["Q"])
        else:
            with pytest.raises(RuntimeError, match="ierr=20"):
                orc.run(ro, wm_flux=wk)


@pytest.mark.parametrize("memory,opt,cal,start", [(False, 0, 0, (2001, 2, 20)), (True, 2, 1, (2004, 2, 20)), (False, 1, 0, (2001, 12, 20))])
def test_lakes_bit_exact(memory, opt, cal, start, oracle_lib):
    """Lakes and reservoirs (endorheic, Doll03, Hanasaki06 with/without inflow memory, HYPE), every
    LakeInputOption, both calendars: the oracle reproduces the reference bit for bit, KWT included."""
    from mizuroute_amd.synthetic import make_lakes
    net = make_network(800, seed=41, n_outlets=6)
    steps, dt = 60, 21600.0
    ro = make_runoff(net.H, steps, seed=5, storm_prob=0.05, storm_amp=3e-6)
    lakes = make_lakes(net, steps, dt, seed=5, frac=0.02, memory=memory, input_option=opt, calendar_id=cal, start=start)
    assert set(lakes["model_type"]) == {0, 1, 2, 3}
    for methods in ([1, 3, 4, 5, 0], [2]):
        out = refrun.run_case(net, ro, dt, methods, lakes=lakes)
        assert out["ierr"] == 0, out["stdout"]
        orc = oracle_lib.Oracle(net, dt, methods, out["frac_future"], out["uh_offset"], out["uh"])
        orc.set_lakes(lakes)
        Q, V = orc.run_lake(ro, lakes, want_vol=True)
        assert np.array_equal(Q, out["Q"]) and np.array_equal(V, out["VOL"]), methods


@pytest.mark.parametrize("jump", [False, True])
def test_target_volume_lakes_and_demand_memory_bit_exact(jump, oracle_lib):
    """is_vol_wm: some lakes follow a prescribed volume (lake_route.f90:139-142,197-205, with and without the jump start);
    Hanasaki demand memory (lake_route.f90:288-331) fed by REACH_WM_FLUX (is_flux_wm).  Oracle == reference, bit for bit."""
    from mizuroute_amd.synthetic import make_lakes
    net = make_network(800, seed=43, n_outlets=6)
    steps, dt = 80, 21600.0
    ro = make_runoff(net.H, steps, seed=6, storm_prob=0.05, storm_amp=3e-6)
    lakes = make_lakes(net, steps, dt, seed=7, frac=0.03, memory=True, input_option=2, calendar_id=1, start=(2004, 2, 10),
                       demand_memory=True, target_frac=0.4, vol_jumpstart=jump)
    assert lakes["targ_vol"].sum() >= 2 and (lakes["model_type"][lakes["targ_vol"] == 0] == 2).any()
    rng = np.random.default_rng(8)
    wm = np.full((steps, net.N), -9999.0)          # realMissing: no water management at this reach
    lr = lakes["reach"] - 1
    wm[:, lr] = 0.3e-8 * net.params["TOTAREA"][lr][None, :] * (1.0 + np.sin(np.arange(steps) / 9.0))[:, None] * (rng.random((steps, lr.size)) - 0.15)
    # (KWT cannot route below a lake that releases nothing -- kinwav_rch's "zero flow", as below an endorheic lake -- so the
    # particle method gets the demand memory without target volumes)
    no_targ = {k: v for k, v in lakes.items() if k not in ("targ_vol", "vol_jumpstart", "wm_vol")}
    for methods, lk in (([1, 5], lakes), ([2], no_targ)):
        out = refrun.run_case(net, ro, dt, methods, lakes=lk, wm_flux=wm)
        assert out["ierr"] == 0, out["stdout"]
        orc = oracle_lib.Oracle(net, dt, methods, out["frac_future"], out["uh_offset"], out["uh"], is_flux_wm=1)
        orc.set_lakes(lk)
        Q, V = orc.run_lake(ro, lk, want_vol=True, wm_flux=wm)
        assert np.array_equal(Q, out["Q"]) and np.array_equal(V, out["VOL"]), methods
        if "targ_vol" in lk:
            tl = lr[lk["targ_vol"] != 0]
            assert (V[-1, 0, tl] <= lk["wm_vol"][-1, tl]).all(), "a target-volume lake never ends a step above its target"


# ---- forcing remap (process_remap.f90:32-316) against the reference's own routines -------------------
@pytest.mark.parametrize("hw_drain,basin_route,zfrac", [(2, 1, 0.0), (1, 1, 0.05), (2, 0, 0.0)])
def test_tracer_bit_exact(hw_drain, basin_route, zfrac, oracle_lib):
    """tracer = T (main_route.f90:161-172,204-236,392-401, basinUH.f90:130-137, tracer.f90:43-207): constituent mass flux
    through the HRU mapping, the hillslope fold and every routing method (KWT included), zero-runoff HRUs, both headwater
    pour points, with and without hillslope routing: flux and mass of every step against the unmodified reference."""
    net = make_network(700, seed=21, zero_area_frac=zfrac)
    steps, dt = 60, 3600.0
    ro = make_runoff(net.H, steps, seed=22, storm_prob=0.05, storm_amp=3e-6)
    rng = np.random.default_rng(23)
    if zfrac > 0 or basin_route == 0:
        ro[:, rng.random(net.H) < 0.1] = 0.0                                # HRUs without runoff: their constituent is dropped (and KWT stops on zero flow)
    sol = rng.uniform(0.0, 2e-3, (steps, net.H)) * (rng.random((steps, net.H)) < 0.7)
    methods = [0, 1, 2, 3, 4, 5]
    out = refrun.run_case(net, ro, dt, methods, solute=sol, hw_drain_point=hw_drain, does_basin_route=basin_route)
    if out["ierr"] != 0:       # KWT 'zero flow' below a zero-area headwater: leave KWT out as the other tests do
        methods = [0, 1, 3, 4, 5]
        out = refrun.run_case(net, ro, dt, methods, solute=sol, hw_drain_point=hw_drain, does_basin_route=basin_route)
    assert out["ierr"] == 0, out["stdout"]
    assert not (zfrac == 0 and basin_route == 1) or 2 in methods          # KWT is part of the comparison where the case allows it
    orc = oracle_lib.Oracle(net, dt, methods, out["frac_future"], out["uh_offset"], out["uh"], does_basin_route=basin_route, hw_drain_point=hw_drain)
    Q, F, M = orc.run_tracer(ro, sol)
    assert np.array_equal(Q, out["Q"])
    assert np.array_equal(F, out["SOLFLUX"]) and np.array_equal(M, out["SOLMASS"])
    assert F[:, 1:].max() > 0 and (F[:, 0] == 0).all()                      # routed by every method but the runoff accumulation


@pytest.mark.parametrize("trend", [1, 2, 3, 4])
def test_direct_insertion_bit_exact(trend, oracle_lib):
    """qmodOption = 1 (main_route.f90:125-148, data_assimilation.f90:28-97): observations at gauges every third step, a gap
    longer than the blending period, missing and negative values, a gauge outside the network; all four error trends.
    The harness hands the observations to the UNMODIFIED main_route through the gageObs interface."""
    from mizuroute_amd.synthetic import make_gauges
    net = make_network(800, seed=11)
    steps, dt = 90, 3600.0
    ro = make_runoff(net.H, steps, seed=12, storm_prob=0.05, storm_amp=3e-6)
    da = make_gauges(net, steps, n_gauge=60, seed=trend, every=3, blend=10, trend=trend)
    methods = [0, 1, 3, 4, 5]
    out = refrun.run_case(net, ro, dt, methods, da=da)
    assert out["ierr"] == 0, out["stdout"]
    orc = oracle_lib.Oracle(net, dt, methods, out["frac_future"], out["uh_offset"], out["uh"])
    assert orc.set_da(da) == 0
    Q, V = orc.run(ro, want_vol=True)
    assert np.array_equal(Q, out["Q"]) and np.array_equal(V, out["VOL"])
    plain = refrun.run_case(net, ro, dt, methods)
    g = da["gauge_reach"][:-1] - 1
    assert np.abs(plain["Q"][:, 1:, g] - out["Q"][:, 1:, g]).max() > 0          # the insertion does change the routed flow
    assert np.array_equal(plain["Q"][:, 0], out["Q"][:, 0])                      # ... but not the runoff accumulation


@pytest.mark.parametrize("H,n1,n2,seed", [(500, 700, 0, 3), (3000, 2500, 0, 4), (800, 40, 30, 5), (2000, 90, 64, 6)])
def test_remap_runoff_bit_exact(H, n1, n2, seed, oracle_lib):
    from mizuroute_amd.synthetic import make_remap, make_source_runoff
    if not refrun.remap_available():
        pytest.skip("oracle/_ref/ref_remap not built")
    mp = make_remap(H, n1, n2, seed=seed)
    sim = make_source_runoff(6, n1, n2, seed=seed + 1)
    ierr, ref = refrun.run_remap(mp, sim)
    rc, orc = oracle_lib.remap_runoff(mp, sim)
    assert ierr == 0 and rc == 0
    assert np.array_equal(ref, orc)
    assert (orc != 0).mean() > 0.5


def test_remap_id_mismatch_is_the_reference_error(oracle_lib):
    from mizuroute_amd.synthetic import make_remap, make_source_runoff
    if not refrun.remap_available():
        pytest.skip("oracle/_ref/ref_remap not built")
    mp = make_remap(300, 400, 0, seed=8)
    k = int(np.nonzero(mp["qhru_ix"] > 0)[0][50])
    mp["qhru_id"][k] += 1                      # process_remap.f90:217-220 -> ierr = 20
    sim = make_source_runoff(2, 400, 0, seed=9)
    ierr, _ = refrun.run_remap(mp, sim)
    rc, _ = oracle_lib.remap_runoff(mp, sim)
    assert ierr == 20 and rc == 20


def test_sort_flux_bit_exact(oracle_lib):
    from mizuroute_amd.synthetic import make_source_runoff
    if not refrun.remap_available():
        pytest.skip("oracle/_ref/ref_remap not built")
    rng = np.random.default_rng(1)
    ix = (rng.permutation(500) + 1).astype(np.int32)
    ix[::17] = -9999
    ix[3] = ix[4]                                 # two file entries for one HRU: the later one wins
    fl = make_source_runoff(3, 500, 0, seed=6)
    for rm in (True, False):
        ierr, ref = refrun.run_remap(dict(ix_in=ix, H=520), fl, remove_negatives=rm)
        assert ierr == 0
        assert np.array_equal(ref, oracle_lib.sort_flux(ix, fl, 520, remove_negatives=rm))


# ---- the reference's start-up routines for the river network (oracle/_ref/ref_topo): augmentation and MPI domains
def _raw_topology(net, seed, zero_area=0.0, orphans=0):
    """what a topology file holds for `net`: ids, downstream ids, lengths, slopes, and the HRUs in a shuffled order (some with
    zero area, some draining to a segment that is not in the network)"""
    rng = np.random.default_rng(seed)
    N = net.N
    hru_id = (np.arange(N) + 50001).astype(np.int64)
    seg_of_hru = np.zeros(N, np.int64)
    for r in range(N):
        seg_of_hru[net.hruIndex[net.hruOffset[r]:net.hruOffset[r + 1]] - 1] = net.reachId[r]
    area = net.params["BASAREA"].copy()
    # a reach may have several HRUs: split some areas over two HRUs of the same reach
    extra = rng.choice(N, size=N // 5, replace=False)
    hru_id = np.concatenate([hru_id, 900001 + np.arange(extra.size)])
    seg_of_hru = np.concatenate([seg_of_hru, net.reachId[extra].astype(np.int64)])
    area = np.concatenate([area, rng.uniform(1e5, 5e6, extra.size)])
    if zero_area > 0:
        area[rng.random(area.size) < zero_area] = 0.0
    perm = rng.permutation(hru_id.size)
    down_id = np.where(net.downIndex > 0, net.reachId[np.maximum(net.downIndex, 1) - 1], -1).astype(np.int64)
    return dict(seg_id=net.reachId.astype(np.int64), down_id=down_id, length=net.params["RLENGTH"], slope=net.params["R_SLOPE"],
                hru_id=hru_id[perm], hru_seg=seg_of_hru[perm], hru_area=area[perm])


@pytest.mark.parametrize("N,seed,zero_area", [(60, 3, 0.0), (700, 4, 0.0), (2500, 5, 0.3)])
def test_network_augmentation_matches_the_reference(N, seed, zero_area):
    """mizuroute_amd.standalone.augment_topology against the UNMODIFIED augment_ntopo (process_ntopo.f90:39-266) and the
    network_topo.f90 routines it calls: downstream indices, upstream lists, HRU lists and weights, areas, goodBasin,
    hydraulic geometry and storage -- integers equal, reals bit for bit."""
    from oracle import refrun
    if not refrun.topo_available():
        pytest.skip("oracle/_ref/ref_topo not built")
    import mizuroute_amd as m
    from mizuroute_amd import standalone
    net0 = m.make_network(N, seed=seed)
    raw = _raw_topology(net0, seed + 100, zero_area)
    nml = dict(wscale=0.0017, mann_n=0.03)
    ref = refrun.run_topo(**raw, n_nodes=1, wscale=nml["wscale"], mann_n=nml["mann_n"], irf=True)
    net = standalone.augment_topology(nml=nml, **raw)
    assert np.array_equal(np.where(net.downIndex > 0, net.downIndex, -1), ref["downSegIndex"])
    assert np.array_equal(np.diff(net.upOffset), ref["nUp"]) and np.array_equal(np.diff(net.hruOffset), ref["nHRU"])
    for r in range(net.N):
        assert np.array_equal(net.upIndex[net.upOffset[r]:net.upOffset[r + 1]], ref["upSegIndices"][r]), r
        assert np.array_equal(net.upGood[net.upOffset[r]:net.upOffset[r + 1]], ref["goodBasin"][r]), r
        assert np.array_equal(net.hruIndex[net.hruOffset[r]:net.hruOffset[r + 1]], ref["hruContribIx"][r]), r
        if net.params["BASAREA"][r] > 0.0:
            assert np.array_equal(net.hruWeight[net.hruOffset[r]:net.hruOffset[r + 1]], ref["weight"][r]), r
        else:      # HRUs without any area: the reference divides 0 by 0 (network_topo.f90:188), here the weights are 0
            assert np.isnan(ref["weight"][r]).all() and (net.hruWeight[net.hruOffset[r]:net.hruOffset[r + 1]] == 0.0).all(), r
    for name, key in (("BASAREA", "basArea"), ("TOTAREA", "totalArea"), ("R_WIDTH", "width"), ("R_DEPTH", "depth"), ("R_STORAGE", "storage"),
                      ("R_MAN_N", "man_n"), ("FLDP_SLOPE", "floodplainSlope")):
        assert np.array_equal(net.params[name], ref[key]), name
    from mizuroute_amd.partition import subtree_sizes
    assert np.array_equal(subtree_sizes(net), ref["nAllUp"])
    # the reach unit hydrographs of the same call (make_uh)
    from mizuroute_amd import uh as uhmod
    off, uhv = uhmod.make_uh(net.params["RLENGTH"], 3600.0, 1.5, 5000.0)
    for r in range(net.N):      # (numpy's exp / power against flang's: last bits)
        assert off[r + 1] - off[r] == ref["timeDelayHist"][r].size and np.allclose(uhv[off[r]:off[r + 1]], ref["timeDelayHist"][r], rtol=1e-12, atol=1e-300), r


@pytest.mark.parametrize("N,seed,nodes", [(300, 5, 4), (3000, 6, 8), (3000, 7, 3), (20000, 8, 8), (500, 9, 2), (400, 10, 1)])
def test_domain_decomposition_matches_the_reference(N, seed, nodes):
    """mizuroute_amd.partition against the UNMODIFIED mpi_domain_decomposition (classify_river_basin + assign_node,
    domain_decomposition.f90:41-163,450-590,724-819): the same domains in the same order, the same reaches in each, the
    same node for each (ties between equally large domains included), and the partitions that follow from it."""
    from oracle import refrun
    if not refrun.topo_available():
        pytest.skip("oracle/_ref/ref_topo not built")
    import mizuroute_amd as m
    from mizuroute_amd.partition import reference_domains, partition_network
    net = m.make_network(N, seed=seed)
    raw = _raw_topology(net, seed + 200)
    ref = refrun.run_topo(**raw, n_nodes=nodes, irf=False)
    from oracle.nr_indexx import nr_indexx      # the reference's (unstable) index sort, restated with the test infrastructure
    kind, outlet, size, node, is_main, root_of = reference_domains(net, nodes, sort_index=nr_indexx)
    doms = [d for d in ref["domains"] if d["basinType"] != 3]
    assert len(doms) == kind.size
    assert [d["basinType"] for d in doms] == list(kind)
    assert [d["segIndex"].size for d in doms] == list(size)
    assert [d["idNode"] for d in doms] == list(node)
    for k, d in enumerate(doms):
        mine = np.nonzero(is_main)[0] if kind[k] == 2 else np.nonzero(root_of == outlet[k])[0]
        assert np.array_equal(np.sort(d["segIndex"]) - 1, mine), k
    P = partition_network(net, nodes, build_for=[], sort_index=nr_indexx)
    want = np.zeros(net.N, np.int64)
    for d in doms:
        want[d["segIndex"] - 1] = max(d["idNode"], 0)
    assert np.array_equal(P.part_of_reach, want)
    assert np.array_equal(P.is_mainstem, is_main)
    # the product's default (a stable argsort instead of the reference's unstable indexx): the same domains, and node loads
    # that differ from the reference's only by which of several equally large domains went where
    k2, o2, s2, node2, m2, r2 = reference_domains(net, nodes)
    assert np.array_equal(k2, kind) and np.array_equal(o2, outlet) and np.array_equal(s2, size) and np.array_equal(m2, is_main)
    tied = np.array([sz for sz in np.unique(size[kind == 1]) if (size[kind == 1] == sz).sum() > 1] or [0])
    for nd in range(-1, nodes):
        assert abs(int(size[node2 == nd].sum()) - int(size[node == nd].sum())) <= int(tied.max()) * nodes, nd
