"""GPU: the per-GPU shard sizes of BASELINE.json configs C3-C5 (8 GPUs: ~375 k reaches KWT, ~625 k reaches
IRF + Muskingum-Cunge, ~375 k reaches diffusive wave with 1 % lakes and floodplains) under size-independent
properties -- a window cut anywhere gives the same bits, particle lists stay within MAXQPAR, discharge is finite
and non-negative, the per-reach water balance closes to the reference's own warning thresholds
(water_balance.f90:22-112: 2e-5, lakes 2e-2, relative to the volumes involved) -- and oracle parity of the same
configurations at 50 k reaches."""
import os

import numpy as np
import pytest

import mizuroute_amd as m
from mizuroute_amd.synthetic import make_lakes
from helpers import REL_TOL, parity_report

pytestmark = pytest.mark.gpu

DT = 3600.0


def _uh(net, dt=DT):
    from mizuroute_amd import uh as uhmod
    frac = uhmod.basin_uh(dt, 2.5, 86400.0)
    off, v = uhmod.make_uh(net.params["RLENGTH"], dt, 1.5, 5000.0)
    return frac, off, v


def _forcing(net, steps, seed=7):
    import torch
    import bench
    ro = bench.device_runoff(torch, net.H, steps, 0, seed, torch.device("cuda", 0))
    torch.cuda.synchronize()
    return ro


def _run_split(dom, ro, cuts, lakes=None):
    """route `ro` (device tensor [steps, H]) in windows of the given lengths"""
    t = 0
    for w in cuts:
        if lakes is not None:
            dom.set_lake_forcing(t, w)
        dom.run_device(w, t * DT, ro[t:t + w].data_ptr())
        dom.sync()
        t += w


def test_c3_shard_kwt_375k(hip_lib):
    net = m.make_network(375_000, seed=20240529)
    frac, _, _ = _uh(net)
    steps = 96
    ro = _forcing(net, steps)
    a = m.RoutingDomain(net, DT, [m.KWT], frac_future=frac, max_window=steps)
    b = m.RoutingDomain(net, DT, [m.KWT], frac_future=frac, max_window=steps)
    _run_split(a, ro, [steps])
    _run_split(b, ro, [7, 1, 40, 48])
    sa, sb = a.kwt_state(), b.kwt_state()
    assert all(np.array_equal(x, y) for x, y in zip(sa, sb)), "a window cut changed the particle state"
    Qa, Qb = a.flux(m.KWT, m.api.F_Q), b.flux(m.KWT, m.api.F_Q)
    assert np.array_equal(Qa, Qb)
    nw = sa[0]
    assert nw.min() >= 1 and nw.max() <= 20, (nw.min(), nw.max())
    assert np.isfinite(Qa).all() and (Qa >= 0).all()
    assert a.sweep_info()[2] > 10_000          # items per launch: the shard really is past the 100 k benchmark
    # total discharge at the outlets never exceeds what has entered the network so far (no water from nowhere)
    qr = a.flux(m.KWT, m.api.F_BASIN_QR1)
    assert np.isfinite(qr).all() and (qr >= 0).all()


def test_c4_shard_irf_mc_625k(hip_lib):
    net = m.make_network(625_000, seed=20240530)
    frac, off, v = _uh(net)
    steps = 48
    ro = _forcing(net, steps)
    kw = dict(frac_future=frac, uh_offset=off, uh=v, max_window=steps)
    a = m.RoutingDomain(net, DT, [m.IRF, m.MC], **kw)
    b = m.RoutingDomain(net, DT, [m.IRF, m.MC], **kw)
    _run_split(a, ro, [steps])
    _run_split(b, ro, [5, 19, 24])
    for meth in (m.IRF, m.MC):
        Qa, Qb = a.flux(meth, m.api.F_Q), b.flux(meth, m.api.F_Q)
        assert np.array_equal(Qa, Qb), meth
        assert np.isfinite(Qa).all() and (Qa >= 0).all()
        wb, vol = a.flux(meth, m.api.F_WB), a.flux(meth, m.api.F_VOL1)
        rel = np.abs(wb) / np.maximum(vol, 1.0)
        assert rel.max() < 2e-5, (meth, float(rel.max()))
    assert np.array_equal(a.irf_state(), b.irf_state())
    assert np.array_equal(a.mol_state(m.MC), b.mol_state(m.MC))


def test_c5_shard_dw_lakes_375k(hip_lib):
    net = m.make_network(375_000, seed=20240531, floodplain=True)
    frac, off, v = _uh(net)
    steps = 32
    lakes = make_lakes(net, steps, DT, seed=9, frac=0.01, input_option=1)
    assert lakes["reach"].size >= 3000                     # 1 % of the reaches
    ro = _forcing(net, steps)
    kw = dict(frac_future=frac, uh_offset=off, uh=v, max_window=steps, lakes=lakes)
    a = m.RoutingDomain(net, DT, [m.DW], **kw)
    b = m.RoutingDomain(net, DT, [m.DW], **kw)
    _run_split(a, ro, [steps], lakes)
    _run_split(b, ro, [3, 13, 16], lakes)
    Qa, Qb = a.flux(m.DW, m.api.F_Q), b.flux(m.DW, m.api.F_Q)
    assert np.array_equal(Qa, Qb)
    assert np.isfinite(Qa).all() and (Qa >= 0).all()
    assert np.array_equal(a.flux(m.DW, m.api.F_VOL1), b.flux(m.DW, m.api.F_VOL1))
    assert np.array_equal(a.mol_state(m.DW), b.mol_state(m.DW))
    is_lake = np.zeros(net.N, bool); is_lake[lakes["reach"] - 1] = True
    wb, vol = a.flux(m.DW, m.api.F_WB), a.flux(m.DW, m.api.F_VOL1)
    rel = np.abs(wb) / np.maximum(vol, 1.0)
    assert rel[~is_lake].max() < 2e-5 and rel[is_lake].max() < 2e-2, (float(rel[~is_lake].max()), float(rel[is_lake].max()))
    fv = a.flux(m.DW, m.api.F_FLOODVOL)
    assert np.isfinite(fv).all() and (fv >= 0).all()


@pytest.mark.parametrize("cfg", ["c2_c3_kwt", "c4_irf_mc", "c5_dw_lakes"])
def test_config_parity_50k(cfg, hip_lib, oracle_lib):
    """the physics of every BASELINE configuration at 50 k reaches against the C oracle (which is pinned to the reference)"""
    steps = 30
    if cfg == "c5_dw_lakes":
        net = m.make_network(50_000, seed=77, floodplain=True)
        lakes = make_lakes(net, steps, DT, seed=5, frac=0.01, input_option=1)
        methods = [m.DW]
    else:
        net = m.make_network(50_000, seed=78)
        lakes = None
        methods = [m.KWT] if cfg == "c2_c3_kwt" else [m.IRF, m.MC]
    frac, off, v = _uh(net)
    ro = m.make_runoff(net.H, steps, seed=12, storm_prob=0.02, storm_amp=2e-6)
    dom = m.RoutingDomain(net, DT, methods, frac_future=frac, uh_offset=off, uh=v, max_window=16, lakes=lakes)
    Qg = dom.run(ro)
    orc = oracle_lib.Oracle(net, DT, methods, frac, off, v)
    if lakes is not None:
        orc.set_lakes(lakes)
        Qo = orc.run_lake(ro, lakes)
    else:
        Qo = orc.run(ro)
    for ix, meth in enumerate(methods):
        rep = parity_report(Qo[:, ix], Qg[:, ix])
        print(cfg, meth, rep)
        assert rep["max_rel"] <= REL_TOL, (cfg, meth, rep)
    if m.KWT in methods:
        assert np.array_equal(dom.kwt_state()[0], orc.kwt_state()[0])


def test_bench_operating_point_sweep_equals_stage_launches(hip_lib, monkeypatch):
    """The driver's bench command at its own operating point: 100 000 reaches, windows of 16 384 steps, 25 windows queued
    without a synchronisation in between (5 + 20, as `bench.py --steps 20 --warmup 5` does), with the regroupings that
    fall into them (before windows 2, 3, 11 and 19).  The persistent sweep must finish (no ierr 93) and leave the same bits
    as one launch per stage (MZR_KWT_SWEEP=0): particle state, last discharge and the interval mean of every reach -- in both flavours
    of the sweep (three and four particle slots per lane of the 4-lane class)."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    N, W, NWIN = 100_000, 16384, 25
    net = m.make_network(N, seed=20240529)
    frac, _, _ = _uh(net)
    pool = [bench.device_runoff(torch, net.H, W, k * W, 7, dev) for k in range(2)]
    torch.cuda.synchronize()

    def route(sweep, wide=False):
        monkeypatch.setenv("MZR_KWT_SWEEP", "1" if sweep else "0")
        monkeypatch.setenv("MZR_KWT_KC_WIDE_RUN", "1" if wide else "0")
        dom = m.RoutingDomain(net, DT, [m.KWT], frac_future=frac, max_window=W)
        for k in range(NWIN):
            dom.run_device(W, k * W * DT, pool[k % 2].data_ptr())
            if k == 4:
                dom.sync()                      # end of the bench's warm-up
        dom.sync()
        out = dom.kwt_state(), dom.flux(m.KWT, m.api.F_Q), dom.mean_q(m.KWT)
        assert dom.sweep_info()[2] > 1000
        dom.close()
        return out

    sa, Qa, Ma = route(True)
    sb, Qb, Mb = route(False)
    assert all(np.array_equal(x, y) for x, y in zip(sa, sb)), "particle state differs between the sweep and one launch per stage"
    assert np.array_equal(Qa, Qb) and np.array_equal(Ma, Mb)
    # the sweep flavour the 375 k-reach shards pick by themselves (4-lane groups of 15 entries, kernels_kwt_wide.hip), here by force
    sw, Qw, Mw = route(True, wide=True)
    assert all(np.array_equal(x, y) for x, y in zip(sw, sb)), "particle state differs between the wide sweep flavour and one launch per stage"
    assert np.array_equal(Qw, Qb) and np.array_equal(Mw, Mb)
    assert np.isfinite(Qa).all() and (Qa >= 0).all() and sa[0].max() <= 20


@pytest.mark.parametrize("config", ["c3", "c4", "c5"])
def test_full_size_network_partitioned_equals_whole(config, hip_lib):
    """BASELINE.json configs[2..4] at FULL size -- 3 M reaches KWT, 5 M reaches IRF + Muskingum-Cunge, 3 M reaches diffusive wave with
    floodplains and 30 000 lakes / reservoirs -- cut into eight sub-basin partitions by the reference's decomposition
    (domain_decomposition.f90:41-163 = mizuroute_amd/partition.py), every partition routed as a domain of its own, the boundary
    records of the tributary outlets replayed through the halo reaches of the mainstem domain (what replaces the per-step
    gather / scatter of mpi_process.f90:1245-1329): interval means of every method and the particle counts of every reach equal
    the unpartitioned network's, bit for bit, over two windows of 256 steps (bench.py's Loopback.parity, the same code the
    `configs` objects of the bench line come from).  Property checks on the deep, narrow mainstem domain on top."""
    import torch
    import bench
    from mizuroute_amd import uh as uhmod
    lb = bench.Loopback(torch, m, uhmod, config, 8)
    P, net = lb.P, lb.net
    assert net.N == bench.FULL[config]
    # the decomposition covers every reach exactly once
    seen = np.zeros(net.N, np.int32)
    for sp in P.trib:
        if sp.n_real:
            np.add.at(seen, sp.reach_global[:sp.n_real], 1)
    assert P.main is not None and P.main.n_real > 1000
    np.add.at(seen, P.main.reach_global[:P.main.n_real], 1)
    assert seen.min() == 1 and seen.max() == 1
    assert sum(1 for sp in P.trib if sp.n_real) == 8
    rep, whole, res = lb.parity(256, 2)
    print(config, rep, {k: (v["reaches"], v["stages"]) for k, v in res["domains"].items()})
    assert rep["partitioned_equals_whole_bit_for_bit"], rep
    assert rep["max_abs_diff"] == 0.0
    # the mainstem domain: thousands of dependent stages, fed by thousands of halo reaches
    dm = res["domains"]["main"]
    assert dm["stages"] > 1000 and dm["halos"] > 1000 and dm["reaches"] == P.main.n_real
    g = P.main.reach_global[:P.main.n_real]
    for mm in lb.methods:
        q = res["mean_part"][mm]
        assert np.isfinite(q).all() and (q >= 0).all()
        # a mainstem reach carries more than the mean reach does (it drains more than 1/8 of the network)
        assert q[g].mean() > 10.0 * q.mean()
    if m.KWT in lb.methods:
        nw = res["nw_part"]
        assert nw.min() >= 1 and nw.max() <= 20
        assert nw[g].mean() > nw.mean()          # long particle lists sit on the main stems


@pytest.mark.parametrize("cfg", ["irf_mc", "dw_lakes"])
def test_windows_shorter_than_the_network_is_deep_overlap_too(cfg, hip_lib, monkeypatch):
    """Round 6: a window of fewer steps than the network has stages keeps its last launches back as well -- the first
    (nStages - 1 - W) of them go out on their own in front of the next window, the last W ride with the next window's launches
    (a mainstem domain of thousands of stages routed in windows of 2 048 is the case this is for: nStages - 1 launches per window
    instead of nStages + W - 1).  Windows of 60-400 steps on a network ~450 stages deep, one of them longer than the network is deep:
    the same bits as MZR_OVERLAP_WINDOWS=0, and exactly one launch fewer per kept-back launch that rode along."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    lakes = None
    if cfg == "dw_lakes":
        net = m.make_network(30_000, seed=31, floodplain=True)
        methods = [m.DW]
    else:
        net = m.make_network(30_000, seed=32)
        methods = [m.IRF, m.MC]
    frac, off, v = _uh(net)
    cuts = [300, 200, 60, 400, 700, 150, 150]
    total = sum(cuts)
    if cfg == "dw_lakes":
        lakes = make_lakes(net, total, DT, seed=9, frac=0.01, input_option=1)
    ro = bench.device_runoff(torch, net.H, total, 0, 7, dev)
    torch.cuda.synchronize()

    def route(overlap):
        monkeypatch.setenv("MZR_OVERLAP_WINDOWS", "1" if overlap else "0")
        monkeypatch.setenv("MZR_STEP_BLOCK", "1")
        dom = m.RoutingDomain(net, DT, methods, frac_future=frac, uh_offset=off, uh=v, max_window=max(cuts), lakes=lakes)
        nS = int(dom.schedule()[0])
        assert 300 < nS <= 512, nS
        t = 0
        for w in cuts:
            if lakes is not None:
                dom.set_lake_forcing(t, w)
            dom.run_device(w, t * DT, ro[t:t + w].data_ptr())
            t += w
        dom.sync()
        out = {}
        for mm in methods:
            out[("Q", mm)] = dom.flux(mm, m.api.F_Q)
            out[("mean", mm)] = dom.mean_q(mm)
            out[("vol", mm)] = dom.flux(mm, m.api.F_VOL1)
            if mm in (m.MC, m.DW):
                out[("mol", mm)] = dom.mol_state(mm)
        if m.IRF in methods:
            out["irf"] = dom.irf_state()
        launches = int(dom.timing(methods[0])["launches"])
        dom.close()
        return out, launches, nS

    (a, la, nS), (b, lb_, _) = route(True), route(False)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert lb_ == sum(nS + w - 1 for w in cuts)
    assert la == lb_ - sum(min(nS - 1, w) for w in cuts[1:]), (la, lb_, nS)


@pytest.mark.parametrize("cfg", ["irf_mc", "dw_lakes", "kw_sum"])
def test_overlapping_windows_equal_windows_one_after_the_other(cfg, hip_lib, monkeypatch):
    """Overlapping windows of the Eulerian methods (kernels_route.hip k_stage_pair; mc_route.f90:46-416, irf_route.f90:40-264,
    dfw_route.f90:49-370 stay what they are): the launches in which window k drains are issued together with the launches in
    which window k+1 fills.  Windows queued back to back -- of different lengths, with a getter in between (which makes the
    library issue the kept-back launches on their own) and a window shorter than the network is deep (which cannot take
    them along) -- must leave the same bits as MZR_OVERLAP_WINDOWS=0: discharge, volumes, solver state, interval means,
    lake state included.  The same for several steps per launch (MZR_STEP_BLOCK; default 4 where the windows are long enough)."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    lakes = None
    if cfg == "dw_lakes":
        net = m.make_network(30_000, seed=31, floodplain=True)
        methods = [m.DW]
    else:
        net = m.make_network(30_000, seed=32)
        methods = [m.IRF, m.MC] if cfg == "irf_mc" else [m.KW, m.SUM]
    frac, off, v = _uh(net)
    cuts = [700, 640, 512, 3, 600, 700]
    total = sum(cuts)
    if cfg == "dw_lakes":
        lakes = make_lakes(net, total, DT, seed=9, frac=0.01, input_option=1)
    ro = bench.device_runoff(torch, net.H, total, 0, 7, dev)
    torch.cuda.synchronize()

    def route(overlap, block=None):
        monkeypatch.setenv("MZR_OVERLAP_WINDOWS", "1" if overlap else "0")
        if block is None:
            monkeypatch.delenv("MZR_STEP_BLOCK", raising=False)
        else:
            monkeypatch.setenv("MZR_STEP_BLOCK", str(block))
        dom = m.RoutingDomain(net, DT, methods, frac_future=frac, uh_offset=off, uh=v, max_window=max(cuts), lakes=lakes)
        assert dom.schedule()[0] <= 512, dom.schedule()
        t = 0
        mid = None
        for k, w in enumerate(cuts):
            if lakes is not None:
                dom.set_lake_forcing(t, w)
            dom.run_device(w, t * DT, ro[t:t + w].data_ptr())
            t += w
            if k == 1:
                mid = dom.flux(methods[0], m.api.F_Q).copy()      # a getter between two queued windows
        dom.sync()
        out = {"mid": mid}
        for mm in methods:
            out[("Q", mm)] = dom.flux(mm, m.api.F_Q)
            out[("mean", mm)] = dom.mean_q(mm)
            if mm != m.SUM:
                out[("vol", mm)] = dom.flux(mm, m.api.F_VOL1)
                out[("wb", mm)] = dom.flux(mm, m.api.F_WB)
        if m.IRF in methods:
            out["irf"] = dom.irf_state()
        for mm in methods:
            if mm in (m.MC, m.DW, m.KW):
                out[("mol", mm)] = dom.mol_state(mm)
        out["qr1"] = dom.flux(methods[0], m.api.F_BASIN_QR1)
        dom.close()
        return out

    a, b = route(True), route(False)
    assert a.keys() == b.keys()
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert np.isfinite(a[("Q", methods[0])]).all()
    # several steps of a reach per launch (stage_reach_block: the skew of the schedule counted in blocks of steps), with block
    # lengths that divide no window, overlapping and not: the same bits again
    for overlap, block in ((True, 2), (True, 5), (False, 7)):
        c = route(overlap, block)
        for k in a:
            assert np.array_equal(c[k], b[k]), (k, overlap, block)


def test_host_forcing_windows_f64_and_f32_equal_resident_forcing(hip_lib):
    """mzr_run_async (the loop of standalone/route_runoff.f90:80-108 with the read hidden) and mzr_run_async_f32 (forcing as the
    files store it, single precision, widened on the device as get_nc widens it into real(dp), read_runoff.f90:264-306): windows
    handed over in page-locked host memory, queued without a synchronisation in between, against the same windows resident on
    the device -- same bits (the forcing here is exactly representable in single precision, as a float file variable is)."""
    import torch
    dev = torch.device("cuda", 0)
    net = m.make_network(20_000, seed=41)
    frac, off, v = _uh(net)
    W, K = 96, 5
    ro32 = (torch.rand((K * W, net.H), dtype=torch.float32) * 2e-7).pin_memory()
    ro64 = ro32.to(torch.float64).pin_memory()
    rod = ro64.to(dev)
    torch.cuda.synchronize()
    res = []
    for mode in ("dev", "f64", "f32"):
        dom = m.RoutingDomain(net, DT, [m.KWT, m.IRF], frac_future=frac, uh_offset=off, uh=v, max_window=W)
        for k in range(K):
            if mode == "dev":
                dom.run_device(W, k * W * DT, rod[k * W:(k + 1) * W].data_ptr())
            elif mode == "f64":
                dom.run_async(W, k * W * DT, ro64[k * W:(k + 1) * W].data_ptr())
            else:
                dom.run_async_f32(W, k * W * DT, ro32[k * W:(k + 1) * W].data_ptr())
        dom.sync()
        res.append((dom.flux(m.KWT, m.api.F_Q), dom.flux(m.IRF, m.api.F_Q), dom.mean_q(m.KWT), dom.kwt_state()[0]))
        dom.close()
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert np.array_equal(a, b)
    assert np.isfinite(res[0][0]).all() and res[0][0].max() > 0


def test_window_whose_sweep_gave_up_is_routed_again(hip_lib, monkeypatch):
    """ierr 93 (a wavefront of the persistent KWT sweep sees no progress on what it waits for and gives up instead of hanging the
    device) is provoked here with a watchdog of one clock tick.  A caller that synchronises every window does
    not see it: the library goes back to the state the window started from and routes it through one launch per stage
    (k_stage_kwt), and the results are those of a run that never used the sweep -- bit for bit."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    net = m.make_network(30_000, seed=51)
    frac, _, _ = _uh(net)
    W, K = 96, 3
    ro = bench.device_runoff(torch, net.H, K * W, 0, 7, dev)
    torch.cuda.synchronize()

    def route(sweep, timeout):
        monkeypatch.setenv("MZR_KWT_SWEEP", "1" if sweep else "0")
        monkeypatch.setenv("MZR_SWEEP_TIMEOUT_S", timeout)
        dom = m.RoutingDomain(net, DT, [m.KWT], frac_future=frac, max_window=W, history=m.api.H_RUNOFF)
        for k in range(K):
            dom.run_device(W, k * W * DT, ro[k * W:(k + 1) * W].data_ptr())
            dom.sync()
        # (the runoff history sums of a window routed again -- instantaneous, delayed and basin runoff -- are made again too)
        hist = [dom.mean(m.KWT, w) for w in (m.api.M_INST_RUNOFF, m.api.M_DLAY_RUNOFF, m.api.M_BAS_RUNOFF)]
        out = dom.kwt_state(), dom.flux(m.KWT, m.api.F_Q), dom.mean_q(m.KWT), dom.sweep_retries(), hist
        dom.close()
        return out

    sa, Qa, Ma, ra, Ha = route(True, "1e-8")
    sb, Qb, Mb, rb, Hb = route(False, "8")
    assert ra >= 1 and rb == 0, (ra, rb)
    assert all(np.array_equal(x, y) for x, y in zip(sa, sb))
    assert np.array_equal(Qa, Qb) and np.array_equal(Ma, Mb)
    assert all(np.array_equal(x, y) and np.abs(x).max() > 0 for x, y in zip(Ha, Hb))


def test_short_window_stall_beside_another_sweep_is_reported(hip_lib, monkeypatch):
    """with KWT and an Eulerian method in a window of at most 8 steps both go through persistent sweeps on one
    stream; after a KWT stall k_sweep_route returns at once, so a retry of the KWT part alone would leave IRF behind without
    a word.  The retry must not be armed there: the caller gets ierr 93."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    net = m.make_network(20_000, seed=52)
    frac, off, v = _uh(net)
    W = 4
    ro = bench.device_runoff(torch, net.H, W, 0, 7, dev)
    torch.cuda.synchronize()
    monkeypatch.setenv("MZR_KWT_SWEEP", "1")
    monkeypatch.setenv("MZR_ROUTE_SWEEP", "1")
    monkeypatch.setenv("MZR_SWEEP_TIMEOUT_S", "1e-8")
    dom = m.RoutingDomain(net, DT, [m.KWT, m.IRF], frac_future=frac, uh_offset=off, uh=v, max_window=W)
    dom.run_device(W, 0.0, ro.data_ptr())
    with pytest.raises(m.MzrError) as ei:
        dom.sync()
    assert ei.value.ierr == 93
    assert dom.sweep_retries() == 0
    dom.close()


def test_sweep_times_itself_on_the_device_clock(hip_lib):
    """mzr_get_sweep_clock: every launch of the persistent KWT sweep leaves the device clock of its first wavefront in and its last
    wavefront out; the durations are positive, one per window, and their sum fits the wall time of the windows."""
    import time
    import torch
    import bench
    dev = torch.device("cuda", 0)
    net = m.make_network(30_000, seed=53)
    frac, _, _ = _uh(net)
    W, K = 256, 4
    ro = bench.device_runoff(torch, net.H, W, 0, 7, dev)
    torch.cuda.synchronize()
    dom = m.RoutingDomain(net, DT, [m.KWT], frac_future=frac, max_window=W)
    dom.run_device(W, 0.0, ro.data_ptr()); dom.sync()
    assert len(dom.sweep_clock(reset=True)) == 1
    t0 = time.perf_counter()
    for k in range(1, 1 + K):
        dom.run_device(W, k * W * DT, ro.data_ptr())
    dom.sync()
    wall_ms = (time.perf_counter() - t0) * 1e3
    ms = dom.sweep_clock()
    assert len(ms) == K and all(x > 0 for x in ms), ms
    assert sum(ms) <= wall_ms * 1.001, (ms, wall_ms)
    assert dom.sweep_clock(2) == ms[-2:]
    dom.close()


def test_queue_of_windows_is_taken_back_to_the_one_that_failed(hip_lib, monkeypatch):
    """Five windows queued without a synchronisation in between (what bench.py does with twenty); the persistent KWT sweep of the third
    gives up (MZR_SWEEP_FAIL_AT=2: a watchdog of one clock tick for that window alone).  Every kernel of the two windows behind it
    returns at once; mzr_sync goes back to the state the third window started from, routes it through one launch per stage and
    queues the two others again: one retry, and the bits of a run that never used the sweep."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    net = m.make_network(30_000, seed=54)
    frac, _, _ = _uh(net)
    W, K = 96, 5
    ro = bench.device_runoff(torch, net.H, K * W, 0, 7, dev)
    torch.cuda.synchronize()

    def route(sweep, fail_at):
        monkeypatch.setenv("MZR_KWT_SWEEP", "1" if sweep else "0")
        if fail_at is None:
            monkeypatch.delenv("MZR_SWEEP_FAIL_AT", raising=False)
        else:
            monkeypatch.setenv("MZR_SWEEP_FAIL_AT", str(fail_at))
        dom = m.RoutingDomain(net, DT, [m.KWT], frac_future=frac, max_window=W)
        for k in range(K):
            dom.run_device(W, k * W * DT, ro[k * W:(k + 1) * W].data_ptr())
        dom.sync()
        out = dom.kwt_state(), dom.flux(m.KWT, m.api.F_Q), dom.mean_q(m.KWT), dom.basin_state(), dom.sweep_retries()
        dom.close()
        return out

    sa, Qa, Ma, Ba, ra = route(True, 2)
    sb, Qb, Mb, Bb, rb = route(False, None)
    assert ra == 1 and rb == 0, (ra, rb)
    assert all(np.array_equal(x, y) for x, y in zip(sa, sb))
    assert np.array_equal(Qa, Qb) and np.array_equal(Ma, Mb) and np.array_equal(Ba, Bb)


@pytest.mark.parametrize("wide", ["0", "1"])
def test_c2_full_size_against_the_oracle(wide, hip_lib, oracle_lib, monkeypatch):
    """BASELINE.json configs[1] at its FULL size -- the bench's own 100 000-reach network -- against the C oracle (which is pinned to the
    reference): 60 steps in windows of 16 (three regroupings: all three lane classes and the fall-backs between them), storms strong enough
    to fill the particle lists and thin them, both flavours of the persistent sweep (three / four particle slots per lane of the 4-lane
    class: kernels_kwt.hip / kernels_kwt_wide.hip).  Discharge of every reach and step within 1e-6 relative, particle counts equal."""
    monkeypatch.setenv("MZR_KWT_KC_WIDE_RUN", wide)
    net = m.make_network(100_000, seed=20240529)
    frac, _, _ = _uh(net)
    steps = 60
    ro = m.make_runoff(net.H, steps, seed=12, storm_prob=0.02, storm_amp=2e-6)
    dom = m.RoutingDomain(net, DT, [m.KWT], frac_future=frac, max_window=16)
    Qg = dom.run(ro)
    orc = oracle_lib.Oracle(net, DT, [m.KWT], frac, np.arange(net.N + 1, dtype=np.int32), np.ones(net.N))
    Qo = orc.run(ro)
    rep = parity_report(Qo[:, 0], Qg[:, 0])
    print("c2 full size, wide 4-lane class", wide, rep)
    assert rep["max_rel"] <= REL_TOL, rep
    assert np.array_equal(dom.kwt_state()[0], orc.kwt_state()[0])
    assert dom.kwt_state()[0].max() >= 19      # lists full: thinning took place
    dom.close()
