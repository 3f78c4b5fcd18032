"""GPU parity: the HIP path (through the C-ABI of libmzr_hip.so) against
  (a) the reference's own outputs in tests/golden/ and
  (b) the CPU oracle on freshly generated cases,
plus size-independent properties at the benchmark size.

Tolerance: BASELINE.json asks for per-reach discharge within 1e-6 relative of the reference.  The
device uses ROCm's FP64 pow/sqrt, which differ from glibc's in the last bits, so FP64 results are
compared with REL_TOL = 1e-6 (helpers.REL_TOL); integer results (particle counts) must be exact.
"""
import os

import numpy as np
import pytest

import mizuroute_amd as m
from helpers import GOLDEN_CASES, REL_TOL, golden_lakes, load_golden, parity_report

pytestmark = pytest.mark.gpu


def _filled(torch, n):
    """n zeroed doubles on the device, the fill FINISHED: torch fills on its own stream, and the library packs records on streams of its
    own that do not wait for it -- a fill that is still queued when the pack kernel runs wipes the record (seen as one partitioned test
    in a few failing on a fresh box, where torch's first fill launch is slow)"""
    t = torch.zeros(int(n), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    return t


def domain_from_golden(net, z, **kw):
    methods = [int(x) for x in z["methods"]]
    return m.RoutingDomain(net, float(z["dt"]), methods, frac_future=z["frac_future"],
                           uh_offset=z["uh_offset"], uh=z["uh"], lakes=golden_lakes(z), **kw)


@pytest.mark.parametrize("name", GOLDEN_CASES)
@pytest.mark.parametrize("window,sweep", [(1000, "1"), (7, "1"), (1, "1"), (1000, "0"), (7, "0"), (7, "1w"), (1, "1w")])
def test_matches_reference_golden(name, window, sweep, hip_lib, monkeypatch):
    """window: steps per call (1 = mzr_step-like); sweep: "1" the persistent sweeps (k_sweep_kwt, and k_sweep_route for the
    Eulerian methods whatever the window length: progress counters instead of kernel boundaries), "0" one launch per stage
    (k_stage_kwt, k_stage).  "1w": the flavour
    with four particle slots per lane of the 4-lane class (windows of 7 and of 1 step: the reaches are regrouped after the second window)."""
    if sweep == "1w":      # the sweep flavour whose 4-lane groups hold 15 entries (kernels_kwt_wide.hip; large domains pick it by themselves)
        monkeypatch.setenv("MZR_KWT_KC_WIDE_RUN", "1")
        sweep = "1"
    monkeypatch.setenv("MZR_KWT_SWEEP", sweep)
    monkeypatch.setenv("MZR_ROUTE_SWEEP", sweep)
    net, z = load_golden(name)
    dom = domain_from_golden(net, z, max_window=window)
    Q = dom.run(z["runoff"])
    methods = [int(x) for x in z["methods"]]
    for ix, meth in enumerate(methods):
        rep = parity_report(z["ref_Q"][:, ix, :], Q[:, ix, :])
        print(name, "method", meth, rep)
        assert rep["max_rel"] <= REL_TOL, (name, meth, rep)
    # hillslope delay and SUM involve only + and *: bit-exact
    assert np.array_equal(dom.flux(methods[0], m.api.F_BASIN_QR1), z["ref_QR1"][-1])
    assert np.array_equal(dom.basin_state(), z["ref_basin_qfuture"])
    if 0 in methods:
        assert np.array_equal(Q[:, methods.index(0), :], z["ref_Q"][:, methods.index(0), :])
    if 1 in methods:
        assert np.array_equal(Q[:, methods.index(1), :], z["ref_Q"][:, methods.index(1), :])
        assert np.array_equal(dom.irf_state(), z["ref_state_1_irf_qfuture"])
    if 2 in methods:
        nw, qf, ti, tr, rf = dom.kwt_state()
        assert np.array_equal(nw, z["ref_state_2_nw"]), "particle counts differ"
        mask = np.arange(rf.shape[1])[None, :] < nw[:, None]
        assert np.array_equal(rf[mask], z["ref_state_2_rf"][mask]), "routed flags differ"
        for got, key in ((qf, "qf"), (ti, "ti"), (tr, "tr")):
            ref = z[f"ref_state_2_{key}"]
            assert np.allclose(got[mask], ref[mask], rtol=REL_TOL, atol=0), key
    if golden_lakes(z) is not None:       # lake storage follows the reference, too
        for meth in methods:
            if meth != 0:
                assert np.allclose(dom.flux(meth, m.api.F_VOL1), z[f"ref_state_{meth}_VOL1"], rtol=REL_TOL, atol=1e-6), meth
    for meth in (3, 4, 5):
        if meth in methods:
            assert np.allclose(dom.mol_state(meth), z[f"ref_state_{meth}_mol"], rtol=REL_TOL, atol=1e-300)
            ix = methods.index(meth)
            for which, key in ((m.api.F_VOL1, "VOL1"), (m.api.F_ELE, "ELE"), (m.api.F_INFLOW, "INFLOW")):
                assert np.allclose(dom.flux(meth, which), z[f"ref_state_{meth}_{key}"], rtol=REL_TOL, atol=1e-12), (meth, key)
    dom.close()


def test_step_by_step_equals_window(hip_lib):
    net, z = load_golden("tree150_all")
    dt = float(z["dt"])
    a = domain_from_golden(net, z, max_window=64)
    Qa = a.run(z["runoff"][:24])
    b = domain_from_golden(net, z, max_window=1)
    for it in range(24):
        b.step(it * dt, (it + 1) * dt, z["runoff"][it])
        for ix, meth in enumerate(a.methods):
            assert np.array_equal(b.flux(meth), Qa[it, ix]), (it, meth)


def test_pipelined_steps_equal_a_window(hip_lib):
    """mzr_step with stepBatch > 1 puts the steps aside and routes them as windows: the host's time loop
    (standalone/route_runoff.f90:80-108) at the speed of the windows.  N steps handed over one by one must leave the same
    bits as one N-step window and as N synchronous steps -- also when the batch does not divide N (a getter flushes what
    is pending), when a gap in time or a step of another length interrupts the sequence, and for every method."""
    net, z = load_golden("tree150_all")
    dt = float(z["dt"])
    ro = z["runoff"]
    n = 50
    a = domain_from_golden(net, z, max_window=64)
    Qa = a.run(ro[:n])
    b = domain_from_golden(net, z, max_window=64, step_batch=16)      # 3 full batches + 2 steps flushed by the getter
    c = domain_from_golden(net, z, max_window=64, step_batch=1)
    for it in range(n):
        b.step(it * dt, (it + 1) * dt, ro[it])
        c.step(it * dt, (it + 1) * dt, ro[it])
    for ix, meth in enumerate(a.methods):
        assert np.array_equal(b.flux(meth), Qa[n - 1, ix]), meth
        assert np.array_equal(c.flux(meth), Qa[n - 1, ix]), meth
        assert np.array_equal(b.mean_q(meth), a.mean_q(meth)), meth
    assert all(np.array_equal(x, y) for x, y in zip(a.kwt_state(), b.kwt_state()))
    assert np.array_equal(a.irf_state(), b.irf_state())
    assert np.array_equal(a.basin_state(), b.basin_state())
    # an interrupted sequence: 5 steps, a step that is shorter than dt, 6 more steps (each part is a window of its own)
    d1 = domain_from_golden(net, z, max_window=64, step_batch=8)
    d2 = domain_from_golden(net, z, max_window=64, step_batch=1)
    t = 0.0
    for it in range(12):
        length = 0.5 * dt if it == 5 else dt
        for d in (d1, d2):
            d.step(t, t + length, ro[it])
        t += length
    for meth in a.methods:
        assert np.array_equal(d1.flux(meth), d2.flux(meth)), meth
    assert all(np.array_equal(x, y) for x, y in zip(d1.kwt_state(), d2.kwt_state()))


@pytest.mark.parametrize("case", ["irf_target_volume_lakes", "kw_mc_lakes"])
def test_pipelined_steps_carry_their_lake_forcing_abstractions_observations_and_constituent(case, hip_lib):
    """stepBatch > 1 in a configuration with per-step inputs beside the runoff: the host's loop stays as the reference's driver
    has it (per step: lake evaporation / precipitation and calendar, REACH_WM_FLUX, REACH_WM_VOL, gauge observations, basin
    constituent -- main_route.f90:115-148,161-172 --, then the step); the one-step calls of the setters put their rows aside
    with the step, and the batch is routed as one window.  Same bits as one step per call, and as windows handed over whole;
    a getter in the middle of a batch, and a one-step setter followed by a window call instead of mzr_step, change nothing."""
    from mizuroute_amd import uh as uhmod
    from mizuroute_amd.synthetic import make_gauges, make_lakes
    net = m.make_network(1200, seed=71, n_outlets=5, floodplain=True)
    steps, dt = 45, 21600.0
    ro = m.make_runoff(net.H, steps, seed=72, storm_prob=0.05, storm_amp=3e-6)
    frac = uhmod.basin_uh(dt, 2.5, 86400.0)
    off, v = uhmod.make_uh(net.params["RLENGTH"], dt, 1.5, 5000.0)
    sol = np.random.default_rng(73).uniform(0.0, 5.0, (steps, net.H))
    da = make_gauges(net, steps, n_gauge=60, seed=74, every=3, blend=6, trend=2)
    if case == "irf_target_volume_lakes":
        methods, wm_on = [m.IRF], 1
        lakes = make_lakes(net, steps, dt, seed=7, frac=0.03, memory=True, input_option=2, calendar_id=1, start=(2004, 2, 10),
                           demand_memory=True, target_frac=0.4, vol_jumpstart=1)
        lr = lakes["reach"] - 1
        wm = np.full((steps, net.N), -9999.0)
        wm[:, lr] = 0.3e-8 * net.params["TOTAREA"][lr][None, :] * (1.0 + np.sin(np.arange(steps) / 9.0))[:, None] * (np.random.default_rng(8).random((steps, lr.size)) - 0.15)
    else:
        methods, wm_on, wm = [m.KW, m.MC], 0, None
        lakes = make_lakes(net, steps, dt, seed=7, frac=0.03, input_option=0)

    def build(**kw):
        dom = m.RoutingDomain(net, dt, methods, frac_future=frac, uh_offset=off, uh=v, max_window=16, lakes=lakes, is_flux_wm=wm_on, **kw)
        dom.set_da(da)
        return dom

    def hand_over(dom, it):      # what the reference's driver reads per step, one-step calls
        dom.set_lake_forcing(it, 1)
        if wm_on:
            dom.set_wm_flux(1, wm[it:it + 1])
        dom.set_obs(it, 1)
        dom.set_solute(1, sol[it:it + 1])

    def results(dom):
        out = []
        for meth in methods:
            out += [dom.flux(meth), dom.flux(meth, m.api.F_VOL1), dom.solute_state(meth, 0), dom.solute_state(meth, 1), dom.mean_q(meth, reset=False)]
        return out

    a = build(); a.set_tracer(sol, time_conv=1.0 / 3600.0, mass_conv=1000.0)      # whole windows
    one, bat = build(step_batch=1), build(step_batch=16)
    for d in (one, bat):
        d.enable_tracer(time_conv=1.0 / 3600.0, mass_conv=1000.0)
    mid = None
    for it in range(steps):
        for d in (one, bat):
            hand_over(d, it)
            d.step(it * dt, (it + 1) * dt, ro[it])
        if it == 20:      # a getter in the middle of the second batch routes what is pending
            mid = [results(one), results(bat)]
    assert all(np.array_equal(x, y) for x, y in zip(*mid))
    Qa = a.run(ro, wm_flux=wm)
    ra, r1, rb = results(a), results(one), results(bat)
    assert all(np.array_equal(x, y) for x, y in zip(r1, rb)), "batched steps against one step per call"
    assert all(np.array_equal(x, y) for x, y in zip(ra, rb)), "batched steps against whole windows"
    assert max(np.abs(x).max() for x in rb) > 0
    # rows put aside for a step that then arrives as a window call: handed to their setters as they are
    it = steps - 1
    w1, w2 = build(step_batch=16), build(step_batch=1)
    for d in (w1, w2):
        d.enable_tracer(time_conv=1.0 / 3600.0, mass_conv=1000.0)
        for k in range(3):
            hand_over(d, k)
            if k < 2:
                d.step(k * dt, (k + 1) * dt, ro[k])
            else:
                d._check(d.L.mzr_run(d.h, 1, k * dt, np.ascontiguousarray(ro[k:k + 1])))
    assert all(np.array_equal(x, y) for x, y in zip(results(w1), results(w2)))


@pytest.mark.parametrize("N,seed,dt,kw", [
    (3000, 21, 3600.0, dict(p3=0.03)),
    (20000, 22, 3600.0, dict()),
    (3000, 23, 86400.0, dict()),
])
def test_kwt_vs_oracle_fresh_case(N, seed, dt, kw, hip_lib, oracle_lib):
    net = m.make_network(N, seed=seed, **kw)
    steps = 96
    ro = m.make_runoff(net.H, steps, seed=seed + 1, storm_prob=0.03, storm_amp=3e-6)
    ff = np.array([0.5, 0.3, 0.2])
    orc = oracle_lib.Oracle(net, dt, [2], ff)
    Qo = orc.run(ro)
    dom = m.RoutingDomain(net, dt, [m.KWT], frac_future=ff, max_window=40)
    Qg = dom.run(ro)
    rep = parity_report(Qo[:, 0], Qg[:, 0])
    print("kwt fresh", N, dt, rep, "stages", dom.schedule())
    assert rep["max_rel"] <= REL_TOL, rep
    assert np.array_equal(dom.kwt_state()[0], orc.kwt_state()[0])
    paths = orc.kwt_paths()   # the storms do drive kinwav_rch through its shock branches
    assert paths["shock_merges"] > 0 and paths["merged_leaving"] > 0 and paths["removes"] > 0, paths
    # particle-traffic counters used by the roofline model agree with the oracle's for the last step
    dom.set_profiling(2)
    dom.kwt_traffic(reset=True)
    orc_t0 = None
    ro2 = m.make_runoff(net.H, 1, seed=seed + 2)
    orc.step(steps * dt, (steps + 1) * dt, ro2[0])
    dom.step(steps * dt, (steps + 1) * dt, ro2[0])
    to, tg = orc.kwt_traffic(), dom.kwt_traffic()
    assert to == tg, (to, tg)


@pytest.mark.parametrize("kind,must", [("duplicates", ("duplicate_times", "shock_merges", "merged_leaving", "merged_staying", "exit_time_fixes")),
                                       ("over64", ("removes_over_64", "confluences_over_2"))])
def test_kwt_rare_branches_vs_oracle(kind, must, hip_lib, oracle_lib):
    """Serial fall-back paths of the lane-group kernel: duplicate particle times across tributaries
    (cursor-walk merge), exit-time ordering fixes, shock merges, the k-way merge of a confluence
    of more than two reaches and thinning of more than 64 particles."""
    from helpers import star_case
    net, ro, dt = star_case(kind)
    ff = np.array([0.5, 0.3, 0.2])
    orc = oracle_lib.Oracle(net, dt, [2], ff)
    Qo = orc.run(ro)
    paths = orc.kwt_paths()
    for k in must:
        assert paths[k] > 0, (k, paths)
    dom = m.RoutingDomain(net, dt, [m.KWT], frac_future=ff, max_window=32)
    Qg = dom.run(ro)
    rep = parity_report(Qo[:, 0], Qg[:, 0])
    print("kwt rare branches", kind, rep, paths)
    assert rep["max_rel"] <= REL_TOL, rep
    assert np.array_equal(dom.kwt_state()[0], orc.kwt_state()[0])


def test_eulerian_methods_vs_oracle_fresh_case(hip_lib, oracle_lib):
    net = m.make_network(4000, seed=31, floodplain=True)
    ro = m.make_runoff(net.H, 60, seed=32, storm_prob=0.05, storm_amp=2e-5)   # large pulses: overbank flow
    ff = np.array([0.6, 0.4])
    uh_off = np.arange(0, 3 * net.N + 1, 3, dtype=np.int32)
    uh = np.tile(np.array([0.2, 0.5, 0.3]), net.N)
    methods = [0, 1, 3, 4, 5]
    orc = oracle_lib.Oracle(net, 3600.0, methods, ff, uh_off, uh, hw_drain_point=1)
    Qo = orc.run(ro)
    dom = m.RoutingDomain(net, 3600.0, methods, frac_future=ff, uh_offset=uh_off, uh=uh, hw_drain_point=1, max_window=25)
    Qg = dom.run(ro)
    for ix, meth in enumerate(methods):
        rep = parity_report(Qo[:, ix], Qg[:, ix])
        print("method", meth, rep)
        assert rep["max_rel"] <= REL_TOL, (meth, rep)
    assert (orc.flux(methods.index(4), oracle_lib.F_FLOODVOL) > 0).any(), "case should exercise the floodplain branch"


def test_reference_error_codes_surface(hip_lib):
    # a reach below a single zero-area headwater has zero flow: kinwav_rch raises ierr=20
    # (kwt_route.f90:1365-1368); negative runoff raises ierr=20 in basin2reach (process_remap.f90:397)
    net = m.make_network(500, seed=3, zero_area_frac=0.1)
    ro = m.make_runoff(net.H, 2, seed=13)
    dom = m.RoutingDomain(net, 86400.0, [m.KWT], frac_future=np.array([1.0]))
    with pytest.raises(m.MzrError) as e:
        dom.run(ro)
    assert e.value.ierr == 20 and "zero flow" in e.value.message
    net2 = m.make_network(100, seed=4)
    dom2 = m.RoutingDomain(net2, 3600.0, [m.SUM], frac_future=np.array([1.0]))
    bad = m.make_runoff(net2.H, 1, seed=1)
    bad[0, 5] = -1.0
    with pytest.raises(m.MzrError) as e2:
        dom2.run(bad)
    assert e2.value.ierr == 20 and "negative runoff" in e2.value.message


def test_window_split_invariance_at_benchmark_size(hip_lib):
    """Size-independent property at the BASELINE size (~100k reaches, KWT): the result must not
    depend on how the time axis is cut into windows (bit-exact), and water must be conserved:
    over a long run the outlets discharge what the hillslopes delivered, minus channel storage."""
    net = m.make_network(100000, seed=20240529)
    steps = 48
    ro = m.make_runoff(net.H, steps, seed=7, storm_prob=0.01, storm_amp=1e-6)
    ff = np.array([0.4, 0.3, 0.2, 0.1])
    a = m.RoutingDomain(net, 3600.0, [m.KWT, m.SUM], frac_future=ff, max_window=48)
    b = m.RoutingDomain(net, 3600.0, [m.KWT, m.SUM], frac_future=ff, max_window=5)
    Qa, Qb = a.run(ro), b.run(ro)
    assert np.array_equal(Qa, Qb)
    assert np.isfinite(Qa).all() and (Qa >= 0).all()
    na, nb = a.kwt_state()[0], b.kwt_state()[0]
    assert np.array_equal(na, nb) and na.max() <= 20 and na.min() >= 1
    # SUM is exact accumulation: outlet discharge == sum of lateral inflows of the whole basin
    outlets = net.downIndex <= 0
    qr1 = a.flux(m.KWT, m.api.F_BASIN_QR1)
    assert np.isclose(Qa[-1, 1, outlets].sum(), qr1.sum(), rtol=1e-9)


def test_fortran_host_drives_the_c_abi(hip_lib, oracle_lib, tmp_path):
    """The ISO_C_BINDING module (mizuroute_amd/fortran/mzr_c.f90) and a Fortran time loop calling
    mzr_step once per step -- the reference's own driver structure -- reproduce the oracle."""
    import os
    import subprocess
    from oracle import refrun
    exe = os.path.join(os.path.dirname(m.lib_path()), "..", "fortran", "mzr_demo")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    net = m.make_network(300, seed=77)
    ro = m.make_runoff(net.H, 20, seed=78, storm_prob=0.05, storm_amp=3e-6)
    ff = np.array([0.5, 0.3, 0.2])
    uh_off = np.arange(0, 2 * net.N + 1, 2, dtype=np.int32)
    uh = np.tile(np.array([0.6, 0.4]), net.N)
    methods = [2, 1, 0]
    case, out = str(tmp_path / "case.bin"), str(tmp_path / "out.bin")
    refrun.write_case(case, net, ro, 3600.0, methods, uh=(ff, uh_off, uh))
    res = subprocess.run([exe, case, out], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    raw = np.fromfile(out, dtype=np.uint8)
    n, steps, nr = np.frombuffer(raw[:12].tobytes(), dtype="<i4")
    Q = np.frombuffer(raw[12:].tobytes(), dtype="<f8").reshape(steps, nr, n)
    orc = oracle_lib.Oracle(net, 3600.0, methods, ff, uh_off, uh)
    Qo = orc.run(ro)
    for ix in range(nr):
        rep = parity_report(Qo[:, ix], Q[:, ix])
        assert rep["max_rel"] <= REL_TOL, (methods[ix], rep)


@pytest.mark.parametrize("kc_wide", ["0", "1"])
def test_partitioned_network_equals_whole(kc_wide, hip_lib, monkeypatch):
    """Sub-basin partitioning with one-way boundary records (mainstem on partition 0) reproduces the
    unpartitioned run bit for bit: every reach sees exactly the upstream records it would have seen.
    kc_wide = 1: the partitions route through the wide flavour of the KWT sweep (the one the 375 k-reach shards of the
    north-star configuration pick; here by force, with export and halo reaches: its FULL instantiation), the whole
    network through the default one.
    All partitions run on this one GPU; the transport is an in-process loopback (the RCCL path is
    the same code with torch.distributed send/recv, tests/test_partition.py covers it over gloo)."""
    import torch
    from mizuroute_amd.partition import PartitionedRouter, partition_network
    net = m.make_network(6000, seed=8, p3=0.02)
    nparts, W, steps = 3, 16, 32
    P = partition_network(net, nparts)
    assert P.main is not None and sum(d.export_local.size for d in P.trib) > 0
    ro = m.make_runoff(net.H, steps, seed=9, storm_prob=0.05, storm_amp=3e-6)
    ff = np.array([0.5, 0.3, 0.2])
    uh_off = np.arange(0, 2 * net.N + 1, 2, dtype=np.int32)
    uh = np.tile(np.array([0.6, 0.4]), net.N)
    methods = [m.KWT, m.IRF, m.SUM]
    whole = m.RoutingDomain(net, 3600.0, methods, frac_future=ff, uh_offset=uh_off, uh=uh, max_window=W)
    Qw = whole.run(ro)
    steps2 = 6 * W
    ro2 = m.make_runoff(net.H, steps2, seed=10, storm_prob=0.05, storm_amp=3e-6)
    whole2 = m.RoutingDomain(net, 3600.0, methods, frac_future=ff, uh_offset=uh_off, uh=uh, max_window=W)
    Qw2 = whole2.run(ro2)
    monkeypatch.setenv("MZR_KWT_KC_WIDE_RUN", kc_wide)

    box = {}

    def make(spec, **kw):
        g = spec.reach_global
        off = np.zeros(g.size + 1, np.int32); off[1:] = np.cumsum(np.diff(uh_off)[g])
        u = np.concatenate([uh[uh_off[x]:uh_off[x + 1]] for x in g])
        return m.RoutingDomain(spec.net, 3600.0, methods, frac_future=ff, uh_offset=off, uh=u, max_window=W, **kw)

    routers = []
    for rank in range(nparts):
        class T:
            def __init__(self, me): self.me = me
            def send(self, t, dst): box[(self.me, dst)] = t.clone()
            def recv(self, t, src): t.copy_(box[(src, 0)]); torch.cuda.synchronize()
        routers.append(PartitionedRouter(P, rank, make, T(rank), lambda n: _filled(torch, n), W))
    Q = np.full((steps, len(methods), net.N), np.nan)
    dev = torch.device("cuda")
    for w0 in range(0, steps, W):
        for rank in list(range(1, nparts)) + [0]:          # tributary partitions first, mainstem owner last
            r = routers[rank]
            rt = torch.from_numpy(np.ascontiguousarray(ro[w0:w0 + W][:, r.trib_spec.hru_global])).to(dev) if r.trib is not None else None
            rm = torch.from_numpy(np.ascontiguousarray(ro[w0:w0 + W][:, r.main_spec.hru_global])).to(dev) if r.main is not None else None
            r.run_window(W, w0 * 3600.0, rt.data_ptr() if rt is not None else 0, rm.data_ptr() if rm is not None else 0)
            r.sync()
            for dom, spec in ((r.trib, r.trib_spec), (r.main, r.main_spec)):
                if dom is None:
                    continue
                for ix, meth in enumerate(methods):
                    q = dom.window_q(meth, W)
                    Q[w0:w0 + W, ix, spec.reach_global[:spec.n_real]] = q[:, :spec.n_real]
    assert not np.isnan(Q).any()
    for ix, meth in enumerate(methods):
        assert np.array_equal(Q[:, ix], Qw[:, ix]), f"method {meth} differs between partitioned and whole-network runs"

    # the same with the exchange pipelined by one window (no sync between windows, as bench.py runs it):
    # interval means and the last window must come out the same
    box.clear()
    routers2 = []
    for rank in range(nparts):
        class T2:
            def __init__(self, me): self.me = me
            def send(self, t, dst): box.setdefault((self.me, dst), []).append(t.clone())
            def recv(self, t, src): t.copy_(box[(src, 0)].pop(0)); torch.cuda.synchronize()
        routers2.append(PartitionedRouter(P, rank, make, T2(rank), lambda n: _filled(torch, n), W))
    order = list(range(1, nparts)) + [0]
    for w0 in range(0, steps2, W):
        for rank in order:
            r = routers2[rank]
            rt = torch.from_numpy(np.ascontiguousarray(ro2[w0:w0 + W][:, r.trib_spec.hru_global])).to(dev) if r.trib is not None else None
            rm = torch.from_numpy(np.ascontiguousarray(ro2[w0:w0 + W][:, r.main_spec.hru_global])).to(dev) if r.main is not None else None
            r.run_window(W, w0 * 3600.0, rt.data_ptr() if rt is not None else 0, rm.data_ptr() if rm is not None else 0, keep=(rt, rm))
    for rank in order:
        routers2[rank].sync()
    for rank in order:
        r = routers2[rank]
        for dom, spec in ((r.trib, r.trib_spec), (r.main, r.main_spec)):
            if dom is None:
                continue
            g = spec.reach_global[:spec.n_real]
            for ix, meth in enumerate(methods):
                assert np.array_equal(dom.window_q(meth, W)[:, :spec.n_real], Qw2[-W:, ix][:, g]), (rank, meth)
                assert np.array_equal(dom.mean_q(meth)[:spec.n_real], whole2.mean_q(meth)[g]), (rank, meth)


@pytest.mark.parametrize("methods,lakes", [([1, 5], False), ([4, 3], False), ([5], True)])
def test_partitioned_eulerian_domains_keep_their_overlapping_windows(methods, lakes, hip_lib):
    """Windows of the Eulerian methods overlap (the last launches of window k go out with the first ones of window k + 1).  A
    tributary domain that ships boundary records keeps that: the record of window k is packed behind the START of window k + 1
    from the rows the library keeps (mzr_export_boundary_prev_dev), not right behind window k (which makes the library issue the
    kept-back launches on their own).  Same bits as the whole network: every window of every reach, and the interval means."""
    import torch
    from mizuroute_amd.partition import PartitionedRouter, partition_network, lakes_for_domain
    from mizuroute_amd.synthetic import make_lakes
    net = m.make_network(5000, seed=18, p3=0.02)
    nparts, nwin = 3, 5
    P = partition_network(net, nparts)
    assert P.main is not None and sum(d.export_local.size for d in P.trib) > 0
    ff = np.array([0.5, 0.3, 0.2])
    uh_off = np.arange(0, 2 * net.N + 1, 2, dtype=np.int32)
    uh = np.tile(np.array([0.6, 0.4]), net.N)
    probe = m.RoutingDomain(net, 3600.0, methods, frac_future=ff, uh_offset=uh_off, uh=uh, max_window=8)
    W = max(64, int(probe.schedule()[0]) + 8)          # at least as many steps as the deepest domain has stages: the windows overlap
    probe.close()
    steps = nwin * W
    lk = make_lakes(net, steps, 3600.0, seed=3, frac=0.01, input_option=1) if lakes else None
    ro = m.make_runoff(net.H, steps, seed=19, storm_prob=0.05, storm_amp=3e-6)
    whole = m.RoutingDomain(net, 3600.0, methods, frac_future=ff, uh_offset=uh_off, uh=uh, max_window=W, lakes=lk)
    Qw = whole.run(ro)

    def make(spec, **kw):
        g = spec.reach_global
        off = np.zeros(g.size + 1, np.int32); off[1:] = np.cumsum(np.diff(uh_off)[g])
        u = np.concatenate([uh[uh_off[x]:uh_off[x + 1]] for x in g])
        return m.RoutingDomain(spec.net, 3600.0, methods, frac_future=ff, uh_offset=off, uh=u, max_window=W,
                               lakes=lakes_for_domain(lk, spec, net.N) if lk is not None else None, **kw)

    box = {}
    dev = torch.device("cuda")
    routers = []
    for rank in range(nparts):
        class T:
            def __init__(self, me): self.me = me
            def send(self, t, dst): box.setdefault((self.me, dst), []).append(t.clone())
            def recv(self, t, src): t.copy_(box[(src, 0)].pop(0)); torch.cuda.synchronize()
        # (torch.empty: a record is packed on the library's streams, which do not wait for a fill kernel on torch's; main_thread: rank 0
        # queues its mainstem window from a host thread of its own)
        routers.append(PartitionedRouter(P, rank, make, T(rank), lambda n: torch.empty(n, dtype=torch.float64, device="cuda"), W, main_thread=True))
    order = list(range(1, nparts)) + [0]
    late = 0
    got = {}
    for k in range(nwin):
        w0 = k * W
        for rank in order:
            r = routers[rank]
            for dom in (r.trib, r.main):
                if dom is not None and dom.lakes is not None:
                    dom.set_lake_forcing(w0, W)
            rt = torch.from_numpy(np.ascontiguousarray(ro[w0:w0 + W][:, r.trib_spec.hru_global])).to(dev) if r.trib is not None else None
            rm = torch.from_numpy(np.ascontiguousarray(ro[w0:w0 + W][:, r.main_spec.hru_global])).to(dev) if r.main is not None else None
            r.run_window(W, w0 * 3600.0, rt.data_ptr() if rt is not None else 0, rm.data_ptr() if rm is not None else 0, keep=(rt, rm))
            late += int(r._late)
    for rank in order:
        routers[rank].sync()
    assert late > 0, "no tributary domain kept its last launches back: the overlap this test is about did not happen"
    # round 6: the MAINSTEM domain's windows overlap too (its imported halo discharge is kept twice; PartitionedRouter waits for the
    # import with wait_import, not sync): a window costs W launches per method, not nStages + W - 1 (a pair of windows is one launch)
    main = routers[0].main
    nS_main = int(main.schedule()[0])
    n_launch = int(main.timing(methods[0])["launches"])
    assert nS_main >= 8 and n_launch <= nwin * W + nS_main, (n_launch, nwin, W, nS_main)
    for rank in order:
        r = routers[rank]
        for dom, spec in ((r.trib, r.trib_spec), (r.main, r.main_spec)):
            if dom is None:
                continue
            g = spec.reach_global[:spec.n_real]
            for ix, meth in enumerate(methods):
                assert np.array_equal(dom.window_q(meth, W)[:, :spec.n_real], Qw[-W:, ix][:, g]), (rank, meth)
                assert np.array_equal(dom.mean_q(meth)[:spec.n_real], whole.mean_q(meth)[g]), (rank, meth)


def test_water_management_fluxes(hip_lib, oracle_lib):
    """is_flux_wm: abstraction cascade / injection in IRF, KW, MC, DW (irf_route.f90:118-142) and
    extract_from_rch in KWT (kwt_route.f90:351-455), incl. missing values (-9999)."""
    net = m.make_network(600, seed=31)
    net.params["MINFLOW"] = np.full(net.N, 1e-4)
    steps = 50
    ro = m.make_runoff(net.H, steps, seed=5, storm_prob=0.05, storm_amp=3e-6)
    rng = np.random.default_rng(3)
    wm = np.zeros((steps, net.N))
    sel = rng.random(net.N) < 0.3
    wm[:, sel] = rng.uniform(0.0, 0.5, sel.sum())[None, :] * (1 + np.sin(np.arange(steps))[:, None])
    inj = rng.random(net.N) < 0.1
    wm[:, inj] = -rng.uniform(0.0, 0.2, inj.sum())[None, :]
    wm[:, rng.random(net.N) < 0.03] = -9999.0
    ff = np.array([0.5, 0.3, 0.2])
    uh_off = np.arange(0, 3 * net.N + 1, 3, dtype=np.int32)
    uh = np.tile(np.array([0.2, 0.5, 0.3]), net.N)
    methods = [1, 3, 4, 5]
    orc = oracle_lib.Oracle(net, 3600.0, methods, ff, uh_off, uh, is_flux_wm=1)
    Qo = orc.run(ro, wm_flux=wm)
    dom = m.RoutingDomain(net, 3600.0, methods, frac_future=ff, uh_offset=uh_off, uh=uh, max_window=20, is_flux_wm=1)
    Qg = dom.run(ro, wm_flux=wm)
    for ix, meth in enumerate(methods):
        rep = parity_report(Qo[:, ix], Qg[:, ix])
        assert rep["max_rel"] <= REL_TOL, (meth, rep)
        assert np.allclose(dom.flux(meth, m.api.F_WB), orc.flux(ix, oracle_lib.F_WB), rtol=1e-6, atol=1e-6)
    # KWT: small abstraction (Qtake < 0) succeeds ...
    wk = np.zeros((steps, net.N)); wk[:, rng.random(net.N) < 0.2] = -0.001
    orc2 = oracle_lib.Oracle(net, 3600.0, [2], ff, is_flux_wm=1)
    Qo2 = orc2.run(ro, wm_flux=wk)
    dom2 = m.RoutingDomain(net, 3600.0, [m.KWT], frac_future=ff, max_window=20, is_flux_wm=1)
    Qg2 = dom2.run(ro, wm_flux=wk)
    rep = parity_report(Qo2[:, 0], Qg2[:, 0])
    assert rep["max_rel"] <= REL_TOL and (Qo2 != oracle_lib.Oracle(net, 3600.0, [2], ff).run(ro)).any(), rep
    # ... and a window without fluxes is refused
    with pytest.raises(m.MzrError):
        dom2.L.mzr_run  # noqa: B018
        dom2._check(dom2.L.mzr_run(dom2.h, 1, 0.0, np.ascontiguousarray(ro[:1])))


# ---- forcing remap (process_remap.f90:32-316): integer gather + ordered FP64 sums -> bit-exact -----------
@pytest.mark.parametrize("n1,n2,seed", [(2500, 0, 4), (90, 64, 6)])
def test_remap_runoff_vs_oracle(n1, n2, seed, hip_lib, oracle_lib):
    import torch
    from mizuroute_amd.synthetic import make_remap, make_source_runoff
    net = m.make_network(3000, seed=2)
    mp = make_remap(net.H, n1, n2, seed=seed)
    steps = 21
    sim = make_source_runoff(steps, n1, n2, seed=seed + 1)
    rc, want = oracle_lib.remap_runoff(mp, sim)
    assert rc == 0
    kw = dict(frac_future=np.array([0.6, 0.4]), uh_offset=np.arange(0, 2 * net.N + 1, 2, dtype=np.int32),
              uh=np.tile(np.array([0.7, 0.3]), net.N), max_window=32)   # IRF: HRUs without a mapping row carry zero runoff
    dom = m.RoutingDomain(net, 3600.0, [m.IRF], **kw)
    dom.set_remap(mp)
    src = torch.from_numpy(sim).cuda()
    dst = torch.full((steps, net.H), -1.0, dtype=torch.float64, device="cuda")
    dom.remap_device(steps, src.data_ptr(), dst.data_ptr())
    dom.sync()
    assert np.array_equal(dst.cpu().numpy(), want)
    # remap + routing in one call == routing of the remapped runoff
    a = dom.run(want)
    b = m.RoutingDomain(net, 3600.0, [m.IRF], **kw)
    b.set_remap(mp)
    b.run_source_device(steps, 0.0, src.data_ptr())
    b.sync()
    assert np.array_equal(b.window_q(m.IRF, steps), a[:, 0])
    # the reference's id check surfaces as its error
    bad = dict(mp)
    if n2 == 0:
        bad["qhru_id"] = mp["qhru_id"].copy()
        bad["qhru_id"][int(np.nonzero(mp["qhru_ix"] > 0)[0][7])] += 1
        with pytest.raises(m.MzrError) as e:
            dom.set_remap(bad)
        assert e.value.ierr == 20 and "mismatch in HRU ids" in str(e.value)


def test_sort_flux_vs_oracle(hip_lib, oracle_lib):
    import torch
    from mizuroute_amd.synthetic import make_source_runoff
    net = m.make_network(520, seed=2)
    rng = np.random.default_rng(1)
    ix = (rng.permutation(500) + 1).astype(np.int32)
    ix[::17] = -9999
    ix[3] = ix[4]
    fl = make_source_runoff(5, 500, 0, seed=6)
    dom = m.RoutingDomain(net, 3600.0, [m.IRF], frac_future=np.array([1.0]), uh_offset=np.arange(net.N + 1, dtype=np.int32), uh=np.ones(net.N))
    src = torch.from_numpy(fl).cuda()
    dst = torch.empty((5, net.H), dtype=torch.float64, device="cuda")
    for rmneg in (True, False):
        dom.set_sort_map(ix, rmneg)
        dom.remap_device(5, src.data_ptr(), dst.data_ptr())
        dom.sync()
        assert np.array_equal(dst.cpu().numpy(), oracle_lib.sort_flux(ix, fl, net.H, remove_negatives=rmneg))


def test_mc_substep_tail_closed_form(hip_lib, oracle_lib, monkeypatch):
    """Muskingum-Cunge on short, flat reaches (Courant number 20-150 -> as many sub-steps per step, mc_route.f90:262-300):
    the sub-step sum with its tail in closed form (default) and iterated to the end (MZR_MC_TAIL_TOL=0), both against
    the oracle's literal loop and against each other."""
    net = m.make_network(3000, seed=91)
    rng = np.random.default_rng(4)
    pick = rng.choice(net.N, 400, replace=False)
    net.params["RLENGTH"][pick] = rng.uniform(150.0, 600.0, pick.size)
    net.params["R_SLOPE"][pick] = rng.uniform(2e-4, 1e-3, pick.size)
    dt, steps = 3600.0, 60
    ro = m.make_runoff(net.H, steps, seed=92, storm_prob=0.05, storm_amp=4e-6)
    ff = np.array([0.6, 0.4])
    orc = oracle_lib.Oracle(net, dt, [m.MC], ff, None, None)
    Qo = orc.run(ro)[:, 0]
    Q = {}
    for tol in ("0", None):
        if tol is None:
            monkeypatch.delenv("MZR_MC_TAIL_TOL", raising=False)
        else:
            monkeypatch.setenv("MZR_MC_TAIL_TOL", tol)
        dom = m.RoutingDomain(net, dt, [m.MC], frac_future=ff, max_window=32)
        Q[tol] = dom.run(ro)[:, 0]
        rep = parity_report(Qo, Q[tol])
        print("MC tail tol", tol, rep)
        assert rep["max_rel"] <= REL_TOL, (tol, rep)
        dom.close()
    big = np.abs(Qo) > 1e-12
    rel = np.abs(Q[None] - Q["0"])[big] / np.abs(Qo[big])
    assert rel.max() < 1e-8, float(rel.max())          # second order in the distance to the fixed point: ~1e-10 observed
    assert parity_report(Qo, Q["0"])["max_rel"] < 1e-10  # without the tail only rounding separates the two


def test_channel_table_equals_values_computed_per_step(hip_lib, monkeypatch):
    """Round 6: what KW, Muskingum-Cunge and DW derive from a reach's channel parameters alone (square and cube roots, bankfull
    area / perimeter / discharge, the overbank coefficients) can be computed once into a table by the very code that runs every
    reach-step (kernels_route.hip d_chan; MZR_CHAN_TABLE=1 -- off by default, it measured slower on the c4 shard): the same bits with
    the table and without, floodplains included; and a parameter set after the first window makes the table again."""
    net = m.make_network(5000, seed=33, floodplain=True)
    dt, steps = 3600.0, 40
    ro = m.make_runoff(net.H, steps, seed=34, storm_prob=0.05, storm_amp=2e-5)      # pulses large enough for overbank flow
    ff = np.array([0.6, 0.4])
    methods = [m.KW, m.MC, m.DW]
    out = {}
    for tab in ("1", None):      # "1": computed per reach-step (the default), None: through the table
        monkeypatch.setenv("MZR_CHAN_TABLE", "0" if tab == "1" else "1")
        dom = m.RoutingDomain(net, dt, methods, frac_future=ff, hw_drain_point=1, max_window=20)
        Q1 = dom.run(ro[:20])
        slope2 = net.params["R_SLOPE"] * 1.5
        dom._check(dom.L.mzr_set_param(dom.h, b"R_SLOPE", np.ascontiguousarray(slope2)))      # (no KWT: the state stays)
        Q2 = dom.run(ro[20:])
        out[tab] = (Q1, Q2, [dom.flux(x, m.api.F_FLOODVOL) for x in methods])
        dom.close()
    for a, b in zip(out["1"][:2], out[None][:2]):
        assert np.array_equal(a, b)
    assert all(np.array_equal(a, b) for a, b in zip(out["1"][2], out[None][2]))
    assert (out[None][2][1] > 0).any(), "the case should exercise the floodplain branch"
    # the changed slope changed the second window (the table was made again)
    monkeypatch.setenv("MZR_CHAN_TABLE", "1")
    dom = m.RoutingDomain(net, dt, methods, frac_future=ff, hw_drain_point=1, max_window=20)
    Qs = dom.run(ro)
    assert np.array_equal(Qs[:20], out[None][0]) and not np.array_equal(Qs[20:], out[None][1])


@pytest.mark.parametrize("case", ["plain", "lakes_wm"])
def test_route_sweep_equals_stage_launches(case, hip_lib, monkeypatch):
    """The persistent sweep of the Eulerian methods (k_sweep_route: tickets, per-reach progress, sc1 accesses, fences around
    lake state) against one launch per stage: the same arithmetic in the same order, so every result is bit-identical --
    discharge of every step, volumes, solver state, history sums -- for windows of 1, 5 and 64 steps."""
    from mizuroute_amd import uh as uhmod
    from mizuroute_amd.synthetic import make_lakes
    net = m.make_network(6000, seed=71, floodplain=(case != "plain"))
    dt, steps = 3600.0, 70
    ro = m.make_runoff(net.H, steps, seed=72, storm_prob=0.04, storm_amp=3e-6)
    ff = np.array([0.5, 0.3, 0.2])
    uh_off, uhv = uhmod.make_uh(net.params["RLENGTH"], dt, 1.5, 5000.0)
    if case == "plain":
        methods, kw, wm = [m.SUM, m.IRF, m.KW, m.MC, m.DW], {}, None
    else:       # one method per domain with lakes (the Hanasaki parameters are the method's own here)
        rng = np.random.default_rng(5)
        wm = np.where(rng.random((steps, net.N)) < 0.1, rng.uniform(-0.01, 0.02, (steps, net.N)), -9999.0)
        methods, kw = [m.DW], dict(lakes=make_lakes(net, steps, dt, seed=8, frac=0.03, input_option=0, memory=True), is_flux_wm=1)
    out = {}
    for sweep, window in (("0", 64), ("1", 64), ("1", 5), ("1", 1)):
        monkeypatch.setenv("MZR_ROUTE_SWEEP", sweep)
        dom = m.RoutingDomain(net, dt, methods, frac_future=ff, uh_offset=uh_off, uh=uhv, max_window=window, history=m.api.H_INFLOW | m.api.H_HEIGHT, **kw)
        Q = dom.run(ro, wm_flux=wm)
        res = [Q] + [dom.flux(me, f) for me in methods for f in (m.api.F_VOL1, m.api.F_WB, m.api.F_INFLOW)] + [dom.mean_q(me) for me in methods]
        res += [dom.mol_state(me) for me in methods if me in (m.KW, m.MC, m.DW)]
        if m.IRF in methods:
            res.append(dom.irf_state())
        out[(sweep, window)] = res
        dom.close()
    ref = out[("0", 64)]
    for key, res in out.items():
        for a, b in zip(ref, res):
            assert np.array_equal(a, b), key


@pytest.mark.parametrize("trend", [2, 3, 4])
def test_direct_insertion_vs_oracle(trend, hip_lib, oracle_lib, monkeypatch):
    """qmodOption = 1: gauge observations inserted into IRF, KW, MC and DW (main_route.f90:125-148, data_assimilation.f90:28-97)
    against the oracle (pinned to the reference harness in tests/test_oracle_vs_ref.py), windows that do not divide the
    observation cycle, both launch forms bit-identical to each other."""
    from mizuroute_amd import uh as uhmod
    from mizuroute_amd.synthetic import make_gauges
    net = m.make_network(3000, seed=41)
    dt, steps = 3600.0, 80
    ro = m.make_runoff(net.H, steps, seed=42, storm_prob=0.05, storm_amp=3e-6)
    ff = np.array([0.5, 0.3, 0.2])
    uh_off, uhv = uhmod.make_uh(net.params["RLENGTH"], dt, 1.5, 5000.0)
    da = make_gauges(net, steps, n_gauge=120, seed=trend, every=3, blend=10, trend=trend)
    da["gauge_reach"][1] = da["gauge_reach"][0]                    # two gauges on one reach: the later one wins
    methods = [m.SUM, m.IRF, m.KW, m.MC, m.DW]
    orc = oracle_lib.Oracle(net, dt, methods, ff, uh_off, uhv)
    assert orc.set_da(da) == 0
    Qo = orc.run(ro)
    Q = {}
    for sweep in ("0", "1"):
        monkeypatch.setenv("MZR_ROUTE_SWEEP", sweep)
        dom = m.RoutingDomain(net, dt, methods, frac_future=ff, uh_offset=uh_off, uh=uhv, max_window=7)
        dom.set_da(da)
        Q[sweep] = dom.run(ro)
        for ix, meth in enumerate(methods):
            rep = parity_report(Qo[:, ix], Q[sweep][:, ix])
            print("direct insertion, trend", trend, "sweep", sweep, "method", meth, rep)
            assert rep["max_rel"] <= REL_TOL, (meth, rep)
        dom.close()
    assert np.array_equal(Q["0"], Q["1"])
    dom = m.RoutingDomain(net, dt, methods, frac_future=ff, uh_offset=uh_off, uh=uhv, max_window=7)
    Qplain = dom.run(ro)
    g = da["gauge_reach"][:-1] - 1
    assert np.abs(Qplain[:, 1:, g] - Q["0"][:, 1:, g]).max() > 0     # the insertion does change the routed flow
    with pytest.raises(m.api.MzrError):                              # switched on, but no observations handed over
        dom.set_da(da); dom.da = None
        dom.run(ro[:7])
    dom.close()


def test_lakes_direct_insertion_and_constituent_in_partitioned_domains(hip_lib):
    """Lakes / reservoirs, gauge observations and a constituent in a network cut into sub-basin partitions: every domain gets the
    lakes and the gauges among the reaches it routes (partition.lakes_for_domain, gauges_for_domain) and the constituent of its
    HRUs; the discharge a lake releases or an observation corrects, and the constituent flux of the tributary outlets
    (tracer.f90:43-138), travel to the mainstem in the boundary record.  Bit-identical to the unpartitioned run."""
    import torch
    from mizuroute_amd import uh as uhmod
    from mizuroute_amd.partition import partition_network, lakes_for_domain, gauges_for_domain
    from mizuroute_amd.synthetic import make_gauges, make_lakes
    net = m.make_network(6000, seed=18, p3=0.02, floodplain=True)
    dt, W, steps, nparts = 3600.0, 12, 36, 3
    ro = m.make_runoff(net.H, steps, seed=19, storm_prob=0.05, storm_amp=3e-6)
    ff = np.array([0.5, 0.3, 0.2])
    uh_off, uhv = uhmod.make_uh(net.params["RLENGTH"], dt, 1.5, 5000.0)
    lakes = make_lakes(net, steps, dt, seed=5, frac=0.02, input_option=0, memory=True)
    da = make_gauges(net, steps, n_gauge=150, seed=2, every=3, blend=6, trend=2)
    methods = [m.IRF, m.DW]
    sol = np.random.default_rng(20).uniform(0.0, 5.0, (steps, net.H))
    # (two methods and Hanasaki reservoirs with memory: per-method copies of the mutable parameters, accepted knowingly --
    # test_hanasaki_memory_with_several_methods_is_refused)
    whole = m.RoutingDomain(net, dt, methods, frac_future=ff, uh_offset=uh_off, uh=uhv, max_window=W, lakes=lakes, lake_memory_per_method=1)
    whole.set_da(da)
    whole.set_tracer(sol, time_conv=1.0 / 3600.0, mass_conv=1000.0)
    Qw = whole.run(ro)
    Fw = whole.solute_flux.copy()
    P = partition_network(net, nparts)
    assert P.main is not None
    n_lake = sum(0 if (lk := lakes_for_domain(lakes, d, net.N)) is None else lk["reach"].size for d in P.trib + [P.main])
    assert n_lake == lakes["reach"].size and lakes_for_domain(lakes, P.main, net.N) is not None      # lakes on the mainstem too

    def build(spec, **kw):
        g = spec.reach_global
        cnt = np.diff(uh_off)[g]
        off = np.zeros(g.size + 1, np.int32); off[1:] = np.cumsum(cnt)
        u = np.concatenate([uhv[uh_off[x]:uh_off[x + 1]] for x in g])
        dom = m.RoutingDomain(spec.net, dt, methods, frac_future=ff, uh_offset=off, uh=u, max_window=W,
                              lakes=lakes_for_domain(lakes, spec, net.N), lake_memory_per_method=1, **kw)
        dom.set_da(gauges_for_domain(da, spec, net.N))
        dom.set_tracer(sol[:, spec.hru_global] if spec.hru_global.size else np.zeros((steps, 1)), time_conv=1.0 / 3600.0, mass_conv=1000.0)
        return dom

    Q = np.full((steps, len(methods), net.N), np.nan)
    F = np.full((steps, len(methods), net.N), np.nan)
    recs = {}
    doms = [(p, sp, build(sp, export_reaches=sp.export_local)) for p, sp in enumerate(P.trib) if sp.n_real > 0]
    main = build(P.main, halo_reaches=P.main.halo_local, halo_good=P.main.halo_good)
    for w0 in range(0, steps, W):
        for p, sp, dom in doms:
            q = dom.run(ro[w0:w0 + W][:, sp.hru_global], t_start=w0 * dt, first_step=w0)
            Q[w0:w0 + W][:, :, sp.reach_global[:sp.n_real]] = q[:, :, :sp.n_real]
            F[w0:w0 + W][:, :, sp.reach_global[:sp.n_real]] = dom.solute_flux[:, :, :sp.n_real]
            if sp.export_local.size:
                rec = _filled(torch, dom.boundary_size(W, sp.export_local.size))
                dom.export_boundary(rec.data_ptr()); dom.sync()
                recs[p] = rec
        for p in range(nparts):
            base, n = P.main.halo_base[p]
            if n:
                main.import_boundary(W, recs[p].data_ptr(), n, base)
        main.sync()
        q = main.run(ro[w0:w0 + W][:, P.main.hru_global], t_start=w0 * dt, first_step=w0)
        Q[w0:w0 + W][:, :, P.main.reach_global[:P.main.n_real]] = q[:, :, :P.main.n_real]
        F[w0:w0 + W][:, :, P.main.reach_global[:P.main.n_real]] = main.solute_flux[:, :, :P.main.n_real]
    assert not np.isnan(Q).any() and not np.isnan(F).any()
    for ix, meth in enumerate(methods):
        assert np.array_equal(Q[:, ix], Qw[:, ix]), meth
        assert np.array_equal(F[:, ix], Fw[:, ix]), ("constituent", meth)
    assert Fw.max() > 0


@pytest.mark.parametrize("hw_drain,basin_route,window", [(2, 1, 9), (1, 1, 64), (2, 0, 1)])
def test_tracer_vs_oracle(hw_drain, basin_route, window, hip_lib, oracle_lib):
    """tracer = T: the constituent through the HRU mapping, the hillslope delay and every routing method, KWT included
    (main_route.f90:161-172,204-236,392-401, tracer.f90:43-207), windows that cut the series anywhere: flux of every step
    and the mass left in every reach against the oracle (bit-identical to the reference harness, tests/test_oracle_vs_ref.py)."""
    from mizuroute_amd import uh as uhmod
    net = m.make_network(2500, seed=61)
    dt, steps = 3600.0, 50
    ro = m.make_runoff(net.H, steps, seed=62, storm_prob=0.05, storm_amp=3e-6)
    rng = np.random.default_rng(63)
    sol = rng.uniform(0.0, 2e-3, (steps, net.H)) * (rng.random((steps, net.H)) < 0.7)
    ff = np.array([0.5, 0.3, 0.2])
    uh_off, uhv = uhmod.make_uh(net.params["RLENGTH"], dt, 1.5, 5000.0)
    methods = [m.SUM, m.IRF, m.KWT, m.KW, m.MC, m.DW]
    orc = oracle_lib.Oracle(net, dt, methods, ff, uh_off, uhv, does_basin_route=basin_route, hw_drain_point=hw_drain)
    Qo, Fo, Mo = orc.run_tracer(ro, sol)
    dom = m.RoutingDomain(net, dt, methods, frac_future=ff, uh_offset=uh_off, uh=uhv, max_window=window,
                          does_basin_route=basin_route, hw_drain_point=hw_drain)
    dom.set_tracer(sol)
    Q = dom.run(ro)
    F = dom.solute_flux
    assert (F[:, 0] == 0).all() and F[:, 1:].max() > 0
    for ix, meth in enumerate(methods):
        repq, repf = parity_report(Qo[:, ix], Q[:, ix]), parity_report(Fo[:, ix], F[:, ix], floor=1e-12)
        print("tracer", hw_drain, basin_route, "method", meth, repq, repf)
        assert repq["max_rel"] <= REL_TOL and repf["max_rel"] <= REL_TOL, (meth, repq, repf)
        if meth != m.SUM:
            mass = dom.solute_state(meth, 1)
            assert np.allclose(mass, Mo[-1, ix], rtol=REL_TOL, atol=1e-9), meth
    with pytest.raises(m.api.MzrError):          # on, but no constituent handed over for the window
        dom.solute = None
        dom.run(ro[:window])
    dom.close()


def test_tracer_restart_continues_bit_exact(tmp_path, hip_lib):
    """a constituent run cut in two with the state going through a restart file (tfuture, solute_mass; write_restart_pio.f90:
    941-971,1292-) gives the same fluxes as the uninterrupted run"""
    from mizuroute_amd import ncfiles, uh as uhmod
    net = m.make_network(1500, seed=81)
    dt, n1, n2 = 3600.0, 30, 25
    ro = m.make_runoff(net.H, n1 + n2, seed=82, storm_prob=0.05, storm_amp=3e-6)
    sol = np.random.default_rng(83).uniform(0.0, 2e-3, (n1 + n2, net.H))
    ff = np.array([0.5, 0.3, 0.2])
    uh_off, uhv = uhmod.make_uh(net.params["RLENGTH"], dt, 1.5, 5000.0)
    methods = [m.KWT, m.IRF, m.DW]
    mk = lambda: m.RoutingDomain(net, dt, methods, frac_future=ff, uh_offset=uh_off, uh=uhv, max_window=16)
    whole = mk(); whole.set_tracer(sol); whole.run(ro)
    Fw = whole.solute_flux.copy()
    a = mk(); a.set_tracer(sol[:n1]); a.run(ro[:n1])
    path = str(tmp_path / "tr.r.nc")
    ncfiles.write_restart(path, a, net.reachId, ((n1 - 1) * dt, n1 * dt), restart_time=n1 * dt)
    st = ncfiles.read_restart_file(path)
    assert "tfuture" in st and "solute_mass" in st and "solute_mass_kwt" in st and "solute_mass_irf" in st
    b = mk(); b.set_tracer(sol[n1:])
    ncfiles.read_restart(path, b)
    b.run(ro[n1:], t_start=n1 * dt)
    assert np.array_equal(b.solute_flux, Fw[n1:])
    for meth in methods:
        assert np.array_equal(b.solute_state(meth, 1), whole.solute_state(meth, 1)), meth


def test_global_water_balance(hip_lib):
    """comp_global_wb (water_balance.f90:191-323) of the last step: the seven sums restated from the per-reach fields, and
    the global error equal to the sum of the per-reach errors (comp_reach_wb, :22-112) -- inside the domain every
    outflow is somebody's inflow -- for plain reaches, with water management, and with lakes."""
    from mizuroute_amd.synthetic import make_lakes
    net = m.make_network(3000, seed=55)
    dt, steps = 3600.0, 20
    ro = m.make_runoff(net.H, steps, seed=56, storm_prob=0.05, storm_amp=3e-6)
    ff = np.array([0.5, 0.3, 0.2])
    outlet = net.downIndex <= 0
    rng = np.random.default_rng(3)
    wm = np.where(rng.random((steps, net.N)) < 0.1, rng.uniform(-0.01, 0.02, (steps, net.N)), 0.0)
    lakes = make_lakes(net, steps, dt, seed=5, frac=0.02, input_option=0)
    from mizuroute_amd import uh as uhmod
    uh_off, uhv = uhmod.make_uh(net.params["RLENGTH"], dt, 1.5, 5000.0)
    for meth, kw, wmf in ((m.IRF, {}, None), (m.DW, dict(is_flux_wm=1), wm), (m.KW, dict(lakes=lakes), None)):
        dom = m.RoutingDomain(net, dt, [meth], frac_future=ff, uh_offset=uh_off, uh=uhv, max_window=8, **kw)
        dom.run(ro, wm_flux=wmf)
        g = dom.global_wb(meth)
        vol1, vol0 = dom.flux(meth, m.api.F_VOL1), dom.flux(meth, m.api.F_VOL0)
        assert np.isclose(g["dVol"], (vol1 - vol0).sum(), rtol=1e-12, atol=1e-6)
        assert np.isclose(g["lateral"], dom.flux(meth, m.api.F_BASIN_QR1).sum() * dt, rtol=1e-12)
        assert np.isclose(g["outflow"], -dom.flux(meth, m.api.F_Q)[outlet].sum() * dt, rtol=1e-12)
        if wmf is not None:
            assert np.isclose(g["take_demand"], -wm[-1].sum() * dt, rtol=1e-12) and g["take_actual"] != 0.0
        else:
            assert g["take_demand"] == 0.0 and g["take_actual"] == 0.0
        if "lakes" in kw:
            assert g["precip"] > 0.0 and g["evaporation"] < 0.0
        else:
            assert g["precip"] == 0.0 and g["evaporation"] == 0.0
        assert g["error"] == g["dVol"] - (g["lateral"] + g["precip"] + g["take_actual"] + g["evaporation"] + g["outflow"])
        wb = dom.flux(meth, m.api.F_WB)
        scale = np.abs(vol1).sum() + abs(g["lateral"])
        if wmf is None:
            assert abs(g["error"] - wb.sum()) <= 1e-9 * scale, (meth, g["error"], wb.sum())
        else:      # (the per-reach balance counts an injection both in the lateral flow and in the take, like the reference's)
            assert abs(g["error"]) <= 1e-9 * scale, (meth, g["error"])
        dom.close()


# ---- restart / history files (ncfiles.py): state through a file == state kept on the device ----------
def test_restart_continues_bit_exact(tmp_path, hip_lib):
    from mizuroute_amd import ncfiles, uh as uhmod
    net = m.make_network(2500, seed=17)
    dt, n1, n2 = 3600.0, 40, 35
    ro = m.make_runoff(net.H, n1 + n2, seed=18, storm_prob=0.03, storm_amp=3e-6)
    ff = np.array([0.5, 0.3, 0.2])
    uh_off, uhv = uhmod.make_uh(net.params["RLENGTH"], dt, 1.5, 5000.0)
    methods = [m.KWT, m.IRF, m.KW, m.MC, m.DW]
    mk = lambda: m.RoutingDomain(net, dt, methods, frac_future=ff, uh_offset=uh_off, uh=uhv, max_window=48)
    whole = mk()
    Qw = np.concatenate([whole.run(ro[:n1]), whole.run(ro[n1:], t_start=n1 * dt)])
    a = mk()
    a.run(ro[:n1])
    path = str(tmp_path / "case.r.nc")
    ncfiles.write_restart(path, a, net.reachId, ((n1 - 1) * dt, n1 * dt), restart_time=n1 * dt)
    b = mk()
    tb = ncfiles.read_restart(path, b)
    assert tb[1] == n1 * dt
    Qb = b.run(ro[n1:], t_start=tb[1])
    assert np.array_equal(Qb, Qw[n1:])
    for meth in methods:
        assert np.array_equal(b.flux(meth, m.api.F_VOL1), whole.flux(meth, m.api.F_VOL1)), meth
    assert np.array_equal(b.kwt_state()[0], whole.kwt_state()[0])
    # history: interval means accumulated on the device, written as float32
    hpath = str(tmp_path / "case.h.nc")
    c = mk()
    hw = ncfiles.HistoryWriter(hpath, net.reachId, methods)
    for k in range(3):
        Q = c.run(ro[k * 24:(k + 1) * 24], t_start=k * 24 * dt)
        hw.append(k * 24 * dt, (k + 1) * 24 * dt, c)
        want = Q.mean(axis=0)
    hw.close()
    from scipy.io import netcdf_file
    f = netcdf_file(hpath, "r", mmap=False)
    got = f.variables["KWTroutedRunoff"][:]
    assert got.shape == (3, net.N)
    assert np.allclose(got[2], want[0], rtol=1e-6)
    for ix, meth in enumerate(methods):        # every method's mean is over its own 24 steps (one reset must not disturb the others)
        assert np.allclose(f.variables[ncfiles.HIST_Q[meth]][:][2], want[ix], rtol=1e-6), meth
    f.close()


@pytest.mark.parametrize("class_b_max,class_c_max,kc_wide,solo", [("0", "0", "0", ""), ("64", "0", "0", ""), ("64", "64", "0", ""), ("0", "64", "0", ""),
                                                                  ("64", "64", "1", ""), ("24", "13", "1", ""), ("0", "13", "1", ""),
                                                                  ("0", "0", "0", "6;1"), ("0", "0", "1", "9;2"), ("12", "6", "1", "14;3")])
def test_kwt_lane_classes_give_the_same_answer(class_b_max, class_c_max, kc_wide, solo, hip_lib, monkeypatch):
    """Routed reaches are served by 16, 8 or 4 lanes depending on a host-side guess of their particle
    count; a wrong guess is caught in the wavefront (wide fall-back).  Forcing every reach into any one
    class must not change a bit.  kc_wide: the sweep flavour whose 4-lane groups hold 15 entries instead of 11
    (MZR_KWT_KC_WIDE_RUN; large domains pick it by themselves).  solo = "n;k": 16-lane reaches that needed n entries or more share
    their pass with k - 1 others at most, the other lane groups of the pass stand empty (MZR_KWT_SOLO_MIN / MZR_KWT_SOLO_PER)."""
    net = m.make_network(4000, seed=51)
    ro = m.make_runoff(net.H, 120, seed=52, storm_prob=0.03, storm_amp=3e-6)
    ff = np.array([0.5, 0.3, 0.2])
    ref = m.RoutingDomain(net, 3600.0, [m.KWT], frac_future=ff, max_window=30)
    Qr = ref.run(ro)
    monkeypatch.setenv("MZR_KWT_CLASSB_MAX", class_b_max)
    monkeypatch.setenv("MZR_KWT_CLASSC_MAX", class_c_max)
    monkeypatch.setenv("MZR_KWT_KC_WIDE_RUN", kc_wide)
    if solo:
        monkeypatch.setenv("MZR_KWT_SOLO_MIN", solo.split(";")[0])
        monkeypatch.setenv("MZR_KWT_SOLO_PER", solo.split(";")[1])
    dom = m.RoutingDomain(net, 3600.0, [m.KWT], frac_future=ff, max_window=30)
    Qd = dom.run(ro)
    assert np.array_equal(Qd, Qr)
    assert all(np.array_equal(a, b) for a, b in zip(dom.kwt_state(), ref.kwt_state()))


# ---- degenerate sizes: one reach, a chain of two, windows of one step, eight-way confluence ---------------
@pytest.mark.parametrize("N", [1, 2, 3, 5, 9])
def test_tiny_networks_vs_oracle(N, hip_lib, oracle_lib):
    net = m.make_network(N, seed=N)
    ro = m.make_runoff(net.H, 30, seed=3, storm_prob=0.1, storm_amp=3e-6)
    ff = np.array([0.6, 0.4])
    uh_off, uhv = np.arange(net.N + 1, dtype=np.int32), np.ones(net.N)
    methods = [m.KWT, m.IRF, m.DW]
    orc = oracle_lib.Oracle(net, 3600.0, methods, ff, uh_off, uhv)
    Qo = orc.run(ro)
    for win in (1, 7, 64):
        dom = m.RoutingDomain(net, 3600.0, methods, frac_future=ff, uh_offset=uh_off, uh=uhv, max_window=win)
        Qg = dom.run(ro)
        for ix in range(len(methods)):
            rep = parity_report(Qo[:, ix], Qg[:, ix])
            assert rep["max_rel"] <= REL_TOL, (N, win, methods[ix], rep)
        assert np.array_equal(dom.kwt_state()[0], orc.kwt_state()[0])
        dom.close()


def test_eight_way_confluence_and_the_upstream_limit(hip_lib, oracle_lib):
    from mizuroute_amd.synthetic import make_star_network
    net = make_star_network(8, 3, seed=2, identical=False)      # eight tributaries meet in one reach: MZR_MAX_UPSTREAM
    ro = m.make_runoff(net.H, 40, seed=5, storm_prob=0.05, storm_amp=3e-6)
    ff = np.array([0.5, 0.5])
    orc = oracle_lib.Oracle(net, 3600.0, [2], ff)
    Qo = orc.run(ro)
    dom = m.RoutingDomain(net, 3600.0, [m.KWT], frac_future=ff, max_window=16)
    rep = parity_report(Qo[:, 0], dom.run(ro)[:, 0])
    assert rep["max_rel"] <= REL_TOL, rep
    assert np.array_equal(dom.kwt_state()[0], orc.kwt_state()[0])
    with pytest.raises(m.MzrError) as e:                           # nine: refused at set-up, not silently wrong
        m.RoutingDomain(make_star_network(9, 2, seed=2), 3600.0, [m.KWT], frac_future=ff)
    assert "at most 8 immediate upstream" in str(e.value)


def test_long_single_step_run_equals_windows(hip_lib):
    """mzr_step (one main_route call per model time step, the coupled-model use) regroups the KWT lane
    classes every 512 steps; 600 single steps must equal the same steps run as windows."""
    net = m.make_network(1500, seed=61)
    steps, dt = 600, 3600.0
    ro = m.make_runoff(net.H, steps, seed=62, storm_prob=0.03, storm_amp=3e-6)
    ff = np.array([0.5, 0.3, 0.2])
    a = m.RoutingDomain(net, dt, [m.KWT], frac_future=ff, max_window=64)
    Qa = a.run(ro)
    b = m.RoutingDomain(net, dt, [m.KWT], frac_future=ff, max_window=1)
    for it in range(steps):
        b.step(it * dt, (it + 1) * dt, ro[it])
        if it % 97 == 0 or it == steps - 1:
            assert np.array_equal(b.flux(m.KWT), Qa[it, 0]), it
    assert all(np.array_equal(x, y) for x, y in zip(a.kwt_state(), b.kwt_state()))


def test_long_window_with_chunked_hillslope_prepass(hip_lib):
    """Windows longer than 2048 steps produce the hillslope series in 1024-step chunks on a second
    stream behind the routing sweep; the result must not depend on that."""
    from mizuroute_amd import uh as uhmod
    net = m.make_network(2000, seed=71)
    steps = 5000
    ro = m.make_runoff(net.H, steps, seed=72, storm_prob=0.02, storm_amp=3e-6)
    ff = uhmod.basin_uh(3600.0, 2.5, 86400.0)
    a = m.RoutingDomain(net, 3600.0, [m.KWT, m.IRF], frac_future=ff, uh_offset=np.arange(net.N + 1, dtype=np.int32), uh=np.ones(net.N), max_window=512)
    b = m.RoutingDomain(net, 3600.0, [m.KWT, m.IRF], frac_future=ff, uh_offset=np.arange(net.N + 1, dtype=np.int32), uh=np.ones(net.N), max_window=4096)
    Qa, Qb = a.run(ro), b.run(ro)
    assert np.array_equal(Qa, Qb)
    assert np.array_equal(a.basin_state(), b.basin_state())
    assert all(np.array_equal(x, y) for x, y in zip(a.kwt_state(), b.kwt_state()))


# ---- bench.py as the driver runs it for N > 1: two ranks (here on one GPU, gloo transport) must route what one rank routes
@pytest.mark.parametrize("ranks", [2, 8])
def test_bench_two_ranks_equal_one_rank(tmp_path, hip_lib, ranks):
    """`bench.py --gpus N` launched as the driver launches it (torch.distributed.run, one process per rank -- here all on the one
    GPU of the box, records over gloo): N = 2 and N = 8, the rank count of the north-star configuration (7 peers, recv_many, the
    reference's assign_node with 8 nodes).  Interval means and particle counts of every reach equal the one-rank run's, bit for bit."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    total = 6000 if ranks == 2 else 12000
    # (no --config: N > 1 takes c3, the north-star network family, as the driver's run does; the reaches per rank are cut down for the test)
    common = ["--reaches", str(total // ranks), "--window", "48", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-roofline",
              "--no-h2d", "--no-single-step", "--no-configs"]
    env = dict(os.environ, MZR_BENCH_SINGLE_DEVICE="1", MZR_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = str(tmp_path / "one")
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--config", "c3", "--reaches", str(total)] + common[2:] + ["--dump", one],
                        cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    two = str(tmp_path / "many")
    # two launch shapes: N = 2 through torch.distributed.run (the driver's documented shape), N = 8 as the PLAIN command
    # `python bench.py --gpus 8 ...` with no WORLD_SIZE -- bench.py launches its eight ranks itself
    launcher = ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
                 "--master-port", str(port)] if ranks == 2 else [sys.executable])
    env2 = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if ranks != 2:
        env2.pop("MZR_BENCH_BACKEND")          # the plain command picks gloo itself when all ranks share one device
    r2 = subprocess.run(launcher + [os.path.join(root, "bench.py"), "--gpus", str(ranks)] + common + ["--dump", two],
                        cwd=root, env=env2, capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-3000:]
    line = r2.stdout.strip().splitlines()[-1]      # the LAST stdout line is the one the driver parses
    import json
    assert len(line) < 4096
    j = json.loads(line)
    assert j["n_gpus"] == ranks and j["config"]["reaches_total"] == total and j["value"] > 0
    assert j["backend"] == "gloo" and j["rccl_ranks"] == 0      # (one GPU here: RCCL carries nothing; on N GPUs rccl_ranks = N)
    assert j["config"]["baseline_config"] == "c3" and "sub-basin partitions" in j["config"]["workload"]
    a = np.load(one + ".rank0.npz")
    parts = [np.load(f"{two}.rank{r}.npz") for r in range(ranks)]
    reach = np.concatenate([p["reach"] for p in parts]); q = np.concatenate([p["q"] for p in parts]); nw = np.concatenate([p["nw"] for p in parts])
    assert np.array_equal(np.sort(reach), np.arange(total)), "every reach is routed by exactly one rank"
    order = np.argsort(reach)
    assert np.array_equal(nw[order], a["nw"]), "particle counts differ between the partitioned and the one-rank run"
    assert np.array_equal(q[order], a["q"]), "interval means differ between the partitioned and the one-rank run"


# ---- the library's own transport (mzr_comm_*, RCCL loaded at run time) ----------------------------------------
def test_comm_single_rank(hip_lib):
    """Loads librccl, creates and destroys a one-rank communicator (all a single GPU allows); bad peers are refused."""
    import torch
    uid = m.api.Comm.unique_id()
    assert len(uid) == 128
    c = m.api.Comm(0, 1, uid, device=0)
    net = m.make_network(50, seed=3)
    dom = m.RoutingDomain(net, 3600.0, [m.IRF], frac_future=[1.0], uh_offset=np.arange(net.N + 1, dtype=np.int32), uh=np.ones(net.N), max_window=4)
    buf = torch.zeros(8, dtype=torch.float64, device="cuda")
    with pytest.raises(m.MzrError):
        c.send(dom, buf.data_ptr(), 8, 0)          # a rank cannot send to itself
    c.sync(); c.close()


_COMM_PAIR = r"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
import mizuroute_amd as m
rank, path = int(sys.argv[2]), sys.argv[3]
torch.cuda.set_device(rank)
if rank == 0:
    open(path + ".tmp", "wb").write(m.api.Comm.unique_id()); os.rename(path + ".tmp", path)
while not os.path.exists(path):
    time.sleep(0.05)
c = m.api.Comm(rank, 2, open(path, "rb").read(), device=rank)
net = m.make_network(50, seed=3)
dom = m.RoutingDomain(net, 3600.0, [m.IRF], frac_future=[1.0], uh_offset=np.arange(net.N + 1, dtype=np.int32), uh=np.ones(net.N), max_window=4, device=rank)
buf = torch.arange(1000, dtype=torch.float64, device=f"cuda:{rank}") * (1 + rank)
if rank == 1:
    c.send(dom, buf.data_ptr(), buf.numel(), 0); c.sync()
else:
    got = torch.zeros(1000, dtype=torch.float64, device="cuda:0")
    c.recv_many(dom, [(got.data_ptr(), got.numel(), 1)]); dom.sync(); c.sync()
    assert torch.equal(got.cpu(), torch.arange(1000, dtype=torch.float64) * 2), "record did not arrive intact"
c.close()
print("ok", rank)
"""


def test_comm_two_ranks(tmp_path, hip_lib):
    """A boundary record from rank 1 to rank 0 through mzr_comm_send / mzr_comm_recv_many (needs two GPUs)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "pair.py"; script.write_text(_COMM_PAIR)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    ps = [subprocess.Popen([sys.executable, str(script), root, str(r), str(tmp_path / "uid")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in (0, 1)]
    for p in ps:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0 and "ok" in out, err[-2000:]


# ---- lakes: target volumes (is_vol_wm) and the Hanasaki demand memory ---------------------------------------------
@pytest.mark.parametrize("jump,window", [(False, 16), (True, 5)])
def test_target_volume_lakes_and_demand_memory_vs_oracle(jump, window, hip_lib, oracle_lib):
    """lake_route.f90:139-142,197-205 (target volume, jump start) and :288-331 (demand memory fed by REACH_WM_FLUX) on the
    device against the oracle (itself bit-identical to the reference, tests/test_oracle_vs_ref.py)."""
    from mizuroute_amd.synthetic import make_lakes
    net = m.make_network(800, seed=43, n_outlets=6)
    steps, dt = 80, 21600.0
    ro = m.make_runoff(net.H, steps, seed=6, storm_prob=0.05, storm_amp=3e-6)
    lakes = make_lakes(net, steps, dt, seed=7, frac=0.03, memory=True, input_option=2, calendar_id=1, start=(2004, 2, 10),
                       demand_memory=True, target_frac=0.4, vol_jumpstart=jump)
    rng = np.random.default_rng(8)
    wm = np.full((steps, net.N), -9999.0)
    lr = lakes["reach"] - 1
    wm[:, lr] = 0.3e-8 * net.params["TOTAREA"][lr][None, :] * (1.0 + np.sin(np.arange(steps) / 9.0))[:, None] * (rng.random((steps, lr.size)) - 0.15)
    from mizuroute_amd import uh as uhmod
    frac = uhmod.basin_uh(dt, 2.5, 86400.0)
    off, v = uhmod.make_uh(net.params["RLENGTH"], dt, 1.5, 5000.0)
    no_targ = {k: val for k, val in lakes.items() if k not in ("targ_vol", "vol_jumpstart", "wm_vol")}
    # one method per domain: with several, the reference's methods share the mutable Hanasaki parameters through RPARAM and so
    # influence each other (the memory is fed once per method and step); here every method keeps its own copy (include/mzr.h)
    for methods, lk in (([m.IRF], lakes), ([m.DW], lakes), ([m.KWT], no_targ)):
        dom = m.RoutingDomain(net, dt, methods, frac_future=frac, uh_offset=off, uh=v, max_window=window, lakes=lk, is_flux_wm=1)
        Qg = dom.run(ro, wm_flux=wm)
        orc = oracle_lib.Oracle(net, dt, methods, frac, off, v, is_flux_wm=1)
        orc.set_lakes(lk)
        Qo, Vo = orc.run_lake(ro, lk, want_vol=True, wm_flux=wm)
        for ix, meth in enumerate(methods):
            rep = parity_report(Qo[:, ix], Qg[:, ix])
            assert rep["max_rel"] <= REL_TOL, (meth, rep)
            vol = dom.flux(meth, m.api.F_VOL1)
            assert np.allclose(vol[lr], Vo[-1, ix, lr], rtol=REL_TOL, atol=1e-6), meth


# ---- history variables beyond discharge (histVars_data.f90:154-305) -------------------------------------------------------
def test_history_means_vs_oracle(tmp_path, hip_lib, oracle_lib):
    """instRunoff, dlayRunoff, basRunoff, and per method routedRunoff, inflow, height, floodVolume (interval means) and volume
    (last value): device accumulators against the oracle's restatement of aggregate / finalize, over two output intervals
    cut by windows that do not divide them; then the same through the history file."""
    from scipy.io import netcdf_file
    from mizuroute_amd import ncfiles, uh as uhmod
    net = m.make_network(1500, seed=61, floodplain=True)
    dt, steps, every = 3600.0, 48, 24
    ro = m.make_runoff(net.H, steps, seed=62, storm_prob=0.05, storm_amp=4e-6)
    frac = uhmod.basin_uh(dt, 2.5, 86400.0)
    off, v = uhmod.make_uh(net.params["RLENGTH"], dt, 1.5, 5000.0)
    methods = [m.SUM, m.IRF, m.KWT, m.MC, m.DW]
    A = m.api
    dom = m.RoutingDomain(net, dt, methods, frac_future=frac, uh_offset=off, uh=v, max_window=10, history=A.H_INFLOW | A.H_HEIGHT | A.H_RUNOFF)
    orc = oracle_lib.Oracle(net, dt, methods, frac, off, v)
    hpath = str(tmp_path / "case.h.nc")
    hw = ncfiles.HistoryWriter(hpath, net.reachId, methods, volumes=True, inflow=True, height=True, runoff=True, hru_id=np.arange(net.H) + 1)
    done = 0
    for k in range(steps // every):
        orc.run(ro[k * every:(k + 1) * every], t_start=k * every * dt)
        while done < (k + 1) * every:
            w = min(10, (k + 1) * every - done)
            dom.run(ro[done:done + w], t_start=done * dt)
            done += w
        for ix, meth in enumerate(methods):
            for which, name in ((A.M_Q, "discharge"), (A.M_INFLOW, "inflow"), (A.M_HEIGHT, "height"), (A.M_FLOODVOL, "floodVolume")):
                got, want = dom.mean(meth, which), orc.hist(ix, which)
                assert np.allclose(got, want, rtol=REL_TOL, atol=1e-12), (k, meth, name, np.abs(got - want).max())
            assert np.allclose(dom.flux(meth, A.F_VOL1), orc.hist(ix, 4), rtol=REL_TOL, atol=1e-9), (k, meth, "volume")
        for which in (A.M_INST_RUNOFF, A.M_DLAY_RUNOFF, A.M_BAS_RUNOFF):
            assert np.array_equal(dom.mean(0, which), orc.hist(0, which)), (k, which)       # sums of the same numbers in the same order
        hw.append(k * every * dt, (k + 1) * every * dt, dom)
        orc.hist_refresh()
    hw.close()
    f = netcdf_file(hpath, "r", mmap=False)
    for name in ("instRunoff", "dlayRunoff", "basRunoff", "sumUpstreamRunoff", "DWinflow", "MCheight", "DWfloodVolume", "IRffloodVolume", "KWTinflow", "IRFvolume"):
        assert name in f.variables and f.variables[name][:].shape[0] == 2 and f.variables[name][:].dtype.itemsize == 4, name
    assert f.variables["basRunoff"].dimensions == ("time", "hru")
    f.close()


def test_hanasaki_memory_with_several_methods_is_refused(hip_lib):
    """lake_route.f90:258-276,360: the reference keeps the mutable Hanasaki parameters (I_months, D_months, E_rel_ini) once per lake
    in RPARAM, so with several active methods every method feeds and reads the same memory.  This library routes the methods side
    by side with a copy each: it says so (ierr 20) instead of deviating silently, unless the caller accepts it
    (mzr_config.lakeMemoryPerMethod); one method, or reservoirs without memory, are not affected."""
    from mizuroute_amd.synthetic import make_lakes
    net = m.make_network(800, seed=3)
    ff = np.array([0.6, 0.4])
    off, v = np.arange(net.N + 1, dtype=np.int32), np.ones(net.N)
    mem = make_lakes(net, 4, 3600.0, seed=5, frac=0.05, input_option=1, memory=True)
    assert (mem["model_type"] == 2).any()
    with pytest.raises(m.MzrError) as e:
        m.RoutingDomain(net, 3600.0, [m.IRF, m.DW], frac_future=ff, uh_offset=off, uh=v, max_window=4, lakes=mem)
    assert e.value.ierr == 20 and "lakeMemoryPerMethod" in e.value.message
    m.RoutingDomain(net, 3600.0, [m.IRF, m.DW], frac_future=ff, uh_offset=off, uh=v, max_window=4, lakes=mem, lake_memory_per_method=1).close()
    m.RoutingDomain(net, 3600.0, [m.DW], frac_future=ff, uh_offset=off, uh=v, max_window=4, lakes=mem).close()
    plain = make_lakes(net, 4, 3600.0, seed=5, frac=0.05, input_option=1, memory=False)
    m.RoutingDomain(net, 3600.0, [m.IRF, m.DW], frac_future=ff, uh_offset=off, uh=v, max_window=4, lakes=plain).close()


def test_kwt_lake_at_a_tributary_outlet_reaches_the_mainstem_as_a_lake(hip_lib):
    """kwt_route.f90:540-559 (getusq_rch): the reach below a lake takes the lake's outflow as ONE particle, and the lake must be its
    only upstream reach ("lake outlet reach should have one upstream lake", ierr 10 otherwise).  In a partitioned network the
    lake may be the outlet of a tributary domain and the reach below it a mainstem reach that only sees a halo: the halo carries
    the flag (mzr_set_boundary haloGood bit 1, partition.halo_flags), so the mainstem domain does what the whole network does --
    here the refusal, since a tributary joins the mainstem at a confluence."""
    from mizuroute_amd.partition import partition_network, halo_flags, lakes_for_domain
    from mizuroute_amd.synthetic import make_lakes
    net = m.make_network(3000, seed=61)
    P = partition_network(net, 2)
    assert P.main is not None
    tmpl = make_lakes(net, 8, 3600.0, seed=5, frac=0.01, input_option=1)
    src = next(sp for sp in P.trib if sp.n_real and sp.export_local.size)
    lake_g = int(src.reach_global[src.export_local[0] - 1])                     # a tributary outlet, global 0-based
    lakes = dict(tmpl, reach=np.array([lake_g + 1], np.int32), model_type=np.array([1], np.int32), par=np.ascontiguousarray(tmpl["par"][:, :1]))
    ro = m.make_runoff(net.H, 8, seed=3, storm_prob=0.05, storm_amp=2e-6)
    ff = np.array([0.6, 0.4])
    # the whole network refuses: the reach below the lake is a confluence
    whole = m.RoutingDomain(net, 3600.0, [m.KWT], frac_future=ff, max_window=8, lakes=lakes)
    with pytest.raises(m.MzrError) as e:
        whole.run(ro)
    assert e.value.ierr == 10
    # so does the mainstem domain, which only sees the lake as a halo reach
    hf = halo_flags(lakes, P.main)
    assert ((hf & 2) != 0).sum() == 1 and ((hf & 1) == (P.main.halo_good != 0)).all()
    main = m.RoutingDomain(P.main.net, 3600.0, [m.KWT], frac_future=ff, max_window=8, halo_reaches=P.main.halo_local, halo_good=hf,
                           lakes=lakes_for_domain(lakes, P.main, net.N))
    import torch
    for sp in P.trib:                                   # the tributaries' records of the window, as the exchange would deliver them
        if not (sp.n_real and sp.export_local.size):
            continue
        td = m.RoutingDomain(sp.net, 3600.0, [m.KWT], frac_future=ff, max_window=8, export_reaches=sp.export_local,
                             lakes=lakes_for_domain(lakes, sp, net.N))
        td.run(ro[:, sp.hru_global])
        rec = _filled(torch, td.boundary_size(8, sp.export_local.size))
        td.export_boundary(rec.data_ptr()); td.sync()
        base, n = P.main.halo_base[sp.part]
        main.import_boundary(8, rec.data_ptr(), n, base); main.sync()
        td.close()
    with pytest.raises(m.MzrError) as e2:
        main.run(ro[:, P.main.hru_global])
    assert e2.value.ierr == 10


def test_boundary_record_that_does_not_fit_is_refused(hip_lib):
    """The boundary record carries what the sender packed (routing methods, steps, reaches, constituent on / off): a receiver that
    expects something else -- here the constituent switched on on one side only, and a record of another window length -- raises
    ierr 20 instead of reading the record at the wrong offsets (what replaces mpi_process.f90:1245-1329 has no MPI datatype to
    catch it)."""
    import torch
    from mizuroute_amd.partition import partition_network
    net = m.make_network(3000, seed=62)
    P = partition_network(net, 2)
    src = next(sp for sp in P.trib if sp.n_real and sp.export_local.size)
    ff = np.array([0.6, 0.4])
    W = 6
    ro = m.make_runoff(net.H, W, seed=3, storm_prob=0.05, storm_amp=2e-6)
    trib = m.RoutingDomain(src.net, 3600.0, [m.KWT], frac_future=ff, max_window=W, export_reaches=src.export_local)
    trib.run(ro[:, src.hru_global])
    n = src.export_local.size
    rec = _filled(torch, trib.boundary_size(W, n))
    trib.export_boundary(rec.data_ptr()); trib.sync()
    base, cnt = P.main.halo_base[src.part]
    assert cnt == n

    def mainstem(tracer):
        d = m.RoutingDomain(P.main.net, 3600.0, [m.KWT], frac_future=ff, max_window=W, halo_reaches=P.main.halo_local, halo_good=P.main.halo_good)
        if tracer:
            d.set_tracer(np.zeros((W, max(1, P.main.hru_global.size))), time_conv=1.0, mass_conv=1.0)
        return d

    ok = mainstem(False)
    ok.import_boundary(W, rec.data_ptr(), n, base); ok.sync()              # the record fits
    bad = mainstem(True)                                                      # constituent on one side only
    bad.import_boundary(W, rec.data_ptr(), n, base)
    with pytest.raises(m.MzrError) as e:
        bad.sync()
    assert e.value.ierr == 20 and "record" in e.value.message
    bad2 = mainstem(False)                                                    # another window length
    bad2.import_boundary(W - 1, rec.data_ptr(), n, base)
    with pytest.raises(m.MzrError) as e:
        bad2.sync()
    assert e.value.ierr == 20
    # round 6: a record carries what the importer reads.  KWT: header | Q | BASIN_QR[W+1] | counts | 2 x 21 particle rows; an Eulerian
    # method: header | Q and nothing else (mpi_process.f90:1245-1329 ships the outlets' fluxes) -- and the two are not mistaken for
    # each other: the KWT part is flagged in the header
    assert trib.boundary_size(W, n) == 4 + (W + (W + 1) + W + 2 * W * 21) * n
    uh_off = np.arange(src.net.N + 1, dtype=np.int32)
    tr_irf = m.RoutingDomain(src.net, 3600.0, [m.IRF, m.DW], frac_future=ff, uh_offset=uh_off, uh=np.ones(src.net.N), max_window=W, export_reaches=src.export_local)
    assert tr_irf.boundary_size(W, n) == 4 + 2 * W * n
    tr_irf.run(ro[:, src.hru_global])
    rec2 = _filled(torch, tr_irf.boundary_size(W, n))
    tr_irf.export_boundary(rec2.data_ptr()); tr_irf.sync()
    hdr = rec2[:4].cpu().numpy()
    assert hdr[1] == 2 and hdr[2] == W and hdr[3] == n and rec.cpu().numpy()[3] == n + 2.0 ** 31
    assert np.array_equal(rec2[4:4 + W * n].cpu().numpy().reshape(W, n), tr_irf.window_q(m.IRF, W)[:, src.export_local - 1])
    mo = np.arange(P.main.net.N + 1, dtype=np.int32)
    main_e = m.RoutingDomain(P.main.net, 3600.0, [m.IRF, m.DW], frac_future=ff, uh_offset=mo, uh=np.ones(P.main.net.N), max_window=W,
                             halo_reaches=P.main.halo_local, halo_good=P.main.halo_good)
    main_e.import_boundary(W, rec2.data_ptr(), n, base); main_e.wait_import(); main_e.sync()      # fits
    bad3 = mainstem(False)                                                    # an Eulerian record offered to a KWT domain
    with pytest.raises(m.MzrError):
        bad3.import_boundary(W, rec2.data_ptr(), n, base)      # (one method against two: refused)
        bad3.sync()
