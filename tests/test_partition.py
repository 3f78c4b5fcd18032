"""CPU: sub-basin partitioner invariants and the N>1 boundary-exchange plumbing over gloo
(world_size 2, no GPU: the compute domains are replaced by a recording stand-in, the transport is
the real torch.distributed point-to-point path)."""
import os
import socket

import numpy as np
import pytest

import mizuroute_amd as m
from mizuroute_amd.partition import PartitionedRouter, partition_network, subtree_sizes


@pytest.mark.parametrize("N,parts", [(500, 2), (3000, 4), (3000, 8), (200, 1)])
def test_partition_invariants(N, parts):
    net = m.make_network(N, seed=5 + parts, p3=0.02)
    P = partition_network(net, parts)
    down0 = net.downIndex.astype(np.int64) - 1
    cnt = subtree_sizes(net)
    assert cnt.max() <= N and (cnt >= 1).all()
    # mainstem rule of the reference (domain_decomposition.f90:507-519)
    if parts > 1:
        assert np.array_equal(P.is_mainstem, cnt > N // parts)
    # every reach is routed in exactly one domain
    seen = np.zeros(N, int)
    for d in P.trib:
        seen[d.reach_global[:d.n_real]] += 1
    if P.main is not None:
        seen[P.main.reach_global[:P.main.n_real]] += 1
    assert (seen == 1).all()
    # mainstem is closed downstream; tributary domains are closed upstream
    ms = np.nonzero(P.is_mainstem)[0]
    assert all(down0[r] < 0 or P.is_mainstem[down0[r]] for r in ms)
    for d in P.trib:
        loc = set(int(g) for g in d.reach_global)
        for g in d.reach_global:
            ups = net.upIndex[net.upOffset[g]:net.upOffset[g + 1]] - 1
            assert all(int(u) in loc for u in ups)
        # local network keeps UREACHI order and parameters
        sub = d.net
        for k, g in enumerate(d.reach_global):
            ups_l = sub.upIndex[sub.upOffset[k]:sub.upOffset[k + 1]] - 1
            ups_g = net.upIndex[net.upOffset[g]:net.upOffset[g + 1]] - 1
            assert np.array_equal(d.reach_global[ups_l], ups_g)
            assert sub.params["R_WIDTH"][k] == net.params["R_WIDTH"][g]
    # exports of all partitions == halos of the mainstem domain, in (partition, export) order
    if P.main is not None:
        halos_g = P.main.reach_global[P.main.n_real:]
        exp_g = np.concatenate([d.reach_global[d.export_local - 1] for d in P.trib])
        assert np.array_equal(halos_g, exp_g)
        for g in halos_g:
            assert P.is_mainstem[down0[g]] and not P.is_mainstem[g]
        for p, (base, n) in P.main.halo_base.items():
            assert n == P.trib[p].export_local.size
    # load balance: no partition above 1.6x the mean unless a single tributary forces it
    loads = np.array([d.n_real for d in P.trib], float)
    if P.main is not None:
        loads[0] += P.main.n_real
    if parts > 1:
        biggest = max(cnt[r] for r in range(N) if not P.is_mainstem[r] and (down0[r] < 0 or P.is_mainstem[down0[r]]))
        assert loads.max() <= max(1.6 * loads.mean(), biggest + 1)


class RecordingDomain:
    """Stand-in for RoutingDomain on CPU: fabricates a deterministic boundary record per export
    reach and remembers what it is asked to import."""

    def __init__(self, spec, export_reaches=None, halo_reaches=None, halo_good=None, sweep_share=1.0, sweep_priority=0):
        import torch
        self.torch, self.spec = torch, spec
        self.exp = np.asarray(export_reaches if export_reaches is not None else [], dtype=np.int64)
        self.imports = {}
        self.ran = []
        self.calls = []

    HDR, MAGIC = 4, 20260930.0      # the record's header (mzr_host.hip MZR_REC_HDR / MZR_REC_MAGIC): {layout version, nRoutes, steps, reaches + flags}
    KWT_FLAG = 2147483648.0         # ... reaches + 2^31 when the record carries the KWT part

    LAG = False      # True: a domain whose windows overlap (an Eulerian method) -- the record of a window is packed one window later

    def boundary_size(self, w, n):
        # mzr_boundary_size, one method: header | Q[R][W][nB] and, with KWT, | qlat[W+1][nB] | obN[W][nB] | obQ[W][21][nB] | obT[W][21][nB].
        # The overlapping (LAG) domain stands for an Eulerian method: its record is the discharge alone
        return self.HDR + (w * n if self.LAG else (1 * w + (w + 1) + w + 2 * w * 21) * n)

    def tag(self, n):
        return float(n) + (0.0 if self.LAG else self.KWT_FLAG)

    def run_device(self, w, t_start, ptr):
        self.ran.append((w, t_start)); self.calls.append(("run", w))

    def export_boundary(self, ptr):
        self.calls.append(("export",))

    def export_lag(self):
        return self.LAG

    def export_boundary_prev(self, ptr):
        assert self.LAG
        self.calls.append(("export_prev",))

    def wait_export(self):
        self.calls.append(("wait_export",))

    def fabricate(self, w):
        n = self.exp.size
        ids = self.spec.net.reachId[self.exp - 1].astype(np.float64)
        rec = np.zeros(self.boundary_size(w, n))
        rec[:self.HDR] = (self.MAGIC, 1.0, float(w), self.tag(n))
        rec[self.HDR: self.HDR + w * n] = (np.arange(w)[:, None] * 1000.0 + ids[None, :]).ravel()   # Q[t][b] = 1000 t + id
        return self.torch.from_numpy(rec)

    def import_boundary(self, w, ptr, n, base):
        self.imports[base] = (w, n)

    def sync(self):
        self.calls.append(("sync",))


def _worker(rank, world, port, q, lag=False):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = m.make_network(1500, seed=42)
    P = partition_network(net, world)
    got = {}

    class Transport:
        def send(self, t, dst): dist.send(t, dst)
        def recv(self, t, src):
            dist.recv(t, src)
            got[src] = t.clone()
        def recv_many(self, pairs):      # the records of all peers at once, as the RCCL transports do (bench.py, mzr_comm_recv_many)
            works = dist.batch_isend_irecv([dist.P2POp(dist.irecv, t, src) for t, src in pairs])
            for wk in works:
                wk.wait()
            for t, src in pairs:
                got[src] = t.clone()

    domains = {}
    RecordingDomain.LAG = lag

    def make(spec, **kw):
        d = RecordingDomain(spec, **kw); domains[spec.kind] = d
        return d

    router = PartitionedRouter(P, rank, make, Transport(), lambda n: torch.zeros(n, dtype=torch.float64), max_window=4)
    w = 4
    # replace the (no-op) device export by the fabricated record
    if router.trib is not None:
        router.alloc = lambda n, _d=router.trib: _d.fabricate(w) if n == _d.boundary_size(w, _d.exp.size) else torch.zeros(n, dtype=torch.float64)
    # two windows: the exchange of window 0 rides behind the start of window 1, sync() flushes window 1
    router.run_window(w, 0.0, 0, 0)
    if rank == 0 and P.main is not None:
        assert domains["main"].ran == []                  # nothing has travelled yet
    router.run_window(w, w * 3600.0, 0, 0)
    ok = True
    if rank == 0 and P.main is not None:
        ok &= domains["main"].ran == [(w, 0.0)]
    router.sync()
    if rank == 0 and P.main is not None:
        main = domains["main"]
        for p in range(1, world):
            base, n = P.main.halo_base[p]
            if n == 0:
                continue
            ok &= main.imports.get(base) == (w, n)
            ids = P.trib[p].net.reachId[P.trib[p].export_local - 1].astype(np.float64)
            expect = (np.arange(w)[:, None] * 1000.0 + ids[None, :]).ravel()
            H = RecordingDomain.HDR
            ok &= bool(np.array_equal(got[p][:H].numpy(), np.array([RecordingDomain.MAGIC, 1.0, w, main.tag(n)])))      # the header the importer checks
            ok &= got[p].numel() == main.boundary_size(w, n)
            ok &= got[p].numel() == H + (w * n if lag else (w + (w + 1) + w + 2 * w * 21) * n)      # an Eulerian record is the discharge alone
            ok &= bool(np.array_equal(got[p][H: H + w * n].numpy(), expect))
        ok &= main.ran == [(w, 0.0), (w, w * 3600.0)]
    if router.trib is not None and router.trib.exp.size and P.main is not None:
        calls = [c[0] for c in router.trib.calls if c[0] != "sync"]
        if lag and router._may_lag:      # window 0's record is packed behind the START of window 1 (no synchronisation in between), window 1's by sync();
                                         # (rank 0 routes two domains on one GPU and exports right behind its window)
            ok &= calls == ["run", "run", "export_prev", "wait_export", "export"]
            ok &= [c[0] for c in router.trib.calls][:2] == ["run", "run"]
        else:
            ok &= calls == ["run", "export", "run", "export"]
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,lag", [(2, False), (8, False), (2, True)])
def test_boundary_exchange_over_gloo(world, lag):
    """The N > 1 path of PartitionedRouter over torch.distributed (gloo): two ranks, and the eight of the north-star configuration
    (seven peers, all their records received at once, the reference's assign_node with eight nodes).  lag: tributary domains whose
    windows overlap -- the record of window k is packed behind the start of window k + 1 (export_boundary_prev), the same records
    arrive in the same order."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, lag)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


def test_nr_indexx_sorts_like_the_reference_routine():
    """oracle.nr_indexx restates nr_utils.f90:114-190 (pinned to the compiled routine through the domain tests of
    test_oracle_vs_ref.py); here: it IS a sort index, for short arrays (insertion sort) and long ones (quicksort), with ties."""
    from oracle.nr_indexx import nr_indexx
    rng = np.random.default_rng(1)
    for n in (1, 2, 15, 16, 17, 100, 5000):
        for hi in (3, 50, 10**6):
            a = rng.integers(0, hi, n)
            ix = nr_indexx(a)
            assert sorted(ix.tolist()) == list(range(n))
            assert np.all(np.diff(a[ix]) >= 0)


def test_lakes_and_gauges_are_dealt_to_the_domains_that_route_them():
    import mizuroute_amd as m
    from mizuroute_amd.partition import partition_network, lakes_for_domain, gauges_for_domain, mainstem_cost
    from mizuroute_amd.synthetic import make_gauges, make_lakes
    net = m.make_network(8000, seed=4, p3=0.02)
    lakes = make_lakes(net, 10, 3600.0, seed=5, frac=0.02, input_option=0, target_frac=0.3)
    da = make_gauges(net, 10, n_gauge=60, seed=2)
    for main_cost in (0.0, mainstem_cost(net, 4, 64)):
        P = partition_network(net, 4, main_cost=main_cost)
        seen_l, seen_g = [], []
        for d in P.trib + [P.main]:
            lk = lakes_for_domain(lakes, d, net.N)
            if lk is not None:
                g = d.reach_global[lk["reach"] - 1]
                seen_l += g.tolist()
                sel = np.searchsorted(lakes["reach"] - 1, g)
                assert np.array_equal(lk["par"], lakes["par"][:, sel]) and np.array_equal(lk["model_type"], lakes["model_type"][sel])
                assert lk["evap"].shape == (10, d.hru_global.size) and np.array_equal(lk["evap"], lakes["evap"][:, d.hru_global])
                assert np.array_equal(lk["wm_vol"], lakes["wm_vol"][:, d.reach_global]) and np.array_equal(lk["targ_vol"], lakes["targ_vol"][sel])
            gd = gauges_for_domain(da, d, net.N)
            loc = gd["gauge_reach"]
            seen_g += d.reach_global[loc[loc > 0] - 1].tolist()
            assert gd["obs"] is da["obs"] and loc.size == da["gauge_reach"].size
        assert sorted(seen_l) == sorted((lakes["reach"] - 1).tolist())                      # every lake exactly once
        assert sorted(seen_g) == sorted((da["gauge_reach"][da["gauge_reach"] > 0] - 1).tolist())
        assert P.part_of_reach[P.is_mainstem].max() == 0
