"""Sub-basin partitioning of a river network across GPUs and the per-window boundary exchange.

Replaces, on the reference side,
  mpi_domain_decomposition / classify_river_basin / assign_node
                                  route/build/src/domain_decomposition.f90:41-163,450-590,724-819
  the per-step gather/scatter of  mpi_route  (route/build/src/mpi_process.f90:1245-1329)

Same decomposition as the reference (`reference_domains` below restates classify_river_basin and assign_node and is
compared with the compiled reference routines): a reach is MAINSTEM when more than N/nParts reaches lie upstream of its
outlet, itself counted (domain_decomposition.f90:507-519); every sub-tree hanging on the mainstem (and every basin
without mainstem) is a TRIBUTARY domain; the smallest tributaries go to partition 0 -- which also routes the mainstem --
until they hold an even share, the others are dealt largest-first onto the least-loaded of the other partitions.

What differs from the reference is the exchange.  There, rank 0 gathers outlet discharge and KWT
particles every step and scatters the stripped particles back.  Here every reach strips its own
routed particles, so data only flows downstream: a tributary partition routes
a whole window of steps, packs the boundary records of its outlet reaches once
(`mzr_export_boundary_dev`) and sends them point-to-point to partition 0, whose mainstem domain
replays them through halo reaches (`mzr_import_boundary_dev`).  One message per partition per
window, no return message, and tributary partitions can run ahead of the mainstem.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .synthetic import RiverNetwork, hops_to_outlet


# Rank 0 sweeps its tributary domain and the mainstem domain side by side on one GPU.  The mainstem is a hundredth of the work
# and one long chain of dependent passes: it gets the wavefronts its items per level ask for (a few hundred of 4 000) and the
# highest wave priority (mzr_config.sweepPriority), the tributary sweep everything else.
MAIN_SWEEP_SHARE = 0.12


@dataclass
class Domain:
    part: int                       # owning partition (rank)
    kind: str                       # "trib" | "main"
    net: RiverNetwork               # local network (1-based local indices)
    reach_global: np.ndarray        # [n_local] global 0-based reach index of every local reach (halos included)
    hru_global: np.ndarray          # [H_local] global 0-based HRU index of every local HRU
    n_real: int                     # local reaches [0, n_real) are routed here; the rest are halo
    export_local: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))   # 1-based
    halo_local: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))     # 1-based
    halo_good: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    halo_base: dict = field(default_factory=dict)   # source partition -> (first halo slot, count)


@dataclass
class Partition:
    n_parts: int
    trib: list                      # Domain per partition (may have zero reaches)
    main: Domain | None             # mainstem domain (owned by partition 0) or None
    part_of_reach: np.ndarray       # [N] partition that routes each reach
    is_mainstem: np.ndarray         # [N] bool


def subtree_sizes(net: RiverNetwork) -> np.ndarray:
    """Number of reaches upstream of each reach, itself included."""
    down0 = net.downIndex.astype(np.int64) - 1
    dist = hops_to_outlet(down0)
    cnt = np.ones(net.N, dtype=np.int64)
    for d in range(int(dist.max()), 0, -1):
        idx = np.nonzero(dist == d)[0]
        np.add.at(cnt, down0[idx], cnt[idx])
    return cnt


def reference_domains(net: RiverNetwork, n_nodes: int, main_cost: float = 0.0, sort_index=None):
    """The reference's MPI domains and their nodes (domain_decomposition.f90): `classify_river_basin` / `decomposeDomain`
    (:450-590, :600-720) make, in this order, one tributary domain per basin whose outlet has at most nSeg/nNodes reaches
    upstream (itself counted), the mainstem domain (every reach with more than that), and one tributary domain per reach
    that drains into the mainstem; `assign_node` (:724-819) gives the smallest tributaries to node 0 until they hold more
    than nTribSeg/nNodes reaches, takes the LAST nNodes-1 domains of the list back out (by position in the list, as the
    source does), and deals what is left, largest first, the mainstem to the root (-1) and every tributary to the node
    of 1..nNodes-1 with the least work so far.  Returns (kind[nDom] 1 tributary / 2 mainstem, outlet reach [nDom] (0-based,
    -1 for the mainstem), size[nDom], node[nDom], is_mainstem[N], root_of[N]).
    main_cost (not in the reference; 0 = its rule): what routing the mainstem costs the root, in tributary reaches.  On CPUs
    the few thousand mainstem reaches are a rounding error of the root's share; on a GPU they are a chain of thousands of
    dependent stages that costs half a tributary share (`mainstem_cost`), so the root's even share of small tributaries is cut
    to (nTribSeg - (nNodes-1) main_cost) / nNodes.  The domains themselves and every result stay the same.
    sort_index: the index sort `assign_node` ranks the domains with.  The reference's `indexx` (nr_utils.f90:114-177) is not
    stable, so WHICH of two equally large domains it serves first is an accident of that routine; here a stable argsort is
    used (results do not depend on it: same domains, same loads up to the size of a tied domain).  The tests pass the
    line-for-line restatement kept with the test infrastructure (nr_indexx, beside the C restatement of the solvers) to compare node for node with the
    compiled reference."""
    N = net.N
    down0 = net.downIndex.astype(np.int64) - 1
    cnt = subtree_sizes(net)
    max_segs = N // max(1, n_nodes)
    is_main = cnt > max_segs                            # :507-519 (allUpSegIndices holds the reach itself)
    kind, outlet, size = [], [], []
    for r in np.nonzero(down0 < 0)[0]:                  # basins without mainstem, in reach order
        if cnt[r] <= max_segs:
            kind.append(1); outlet.append(int(r)); size.append(int(cnt[r]))
    if is_main.any():
        kind.append(2); outlet.append(-1); size.append(int(is_main.sum()))
        trib_out = (~is_main) & (down0 >= 0) & is_main[np.maximum(down0, 0)]      # lgc_tributary_outlet
        for r in np.nonzero(trib_out)[0]:
            kind.append(1); outlet.append(int(r)); size.append(int(cnt[r]))
    kind, outlet, size = np.array(kind, np.int64), np.array(outlet, np.int64), np.array(size, np.int64)
    n_dom = kind.size
    rank = np.argsort(size, kind="stable") if sort_index is None else np.asarray(sort_index(size))
    node = np.full(n_dom, -99, np.int64)
    assigned = np.zeros(n_dom, bool)
    n_even = int(size[kind == 1].sum()) // max(1, n_nodes)
    if main_cost > 0.0 and n_nodes > 1 and (kind == 2).any():
        n_even = max(0, int((float(size[kind == 1].sum()) - (n_nodes - 1) * main_cost) / n_nodes))
    small = 0
    for ixx in rank:
        if kind[ixx] == 1:
            small += int(size[ixx]); node[ixx] = 0; assigned[ixx] = True
            if small > n_even:
                break
    if n_nodes > 1:
        assigned[max(0, n_dom - n_nodes + 1):] = False    # isAssigned(nDomain-nNodes+2:nDomain) = .false.
    work = np.zeros(max(0, n_nodes - 1), np.int64)
    for ixx in rank[::-1]:
        if assigned[ixx]:
            continue
        if kind[ixx] == 2:
            node[ixx] = -1
        else:
            k = int(np.argmin(work)) if work.size else -1       # minloc: the first minimum
            if k >= 0:
                work[k] += int(size[ixx])
            node[ixx] = k + 1
        assigned[ixx] = True
    # root (domain outlet) of every non-mainstem reach
    root_of = np.full(N, -1, dtype=np.int64)
    outs = outlet[kind == 1]
    root_of[outs] = outs
    dist = hops_to_outlet(down0)
    for r in np.argsort(dist, kind="stable"):
        if root_of[r] < 0 and not is_main[r]:
            root_of[r] = root_of[down0[r]]
    return kind, outlet, size, node, is_main, root_of


def _local_network(net: RiverNetwork, real: np.ndarray, halos: np.ndarray) -> tuple:
    """Local RiverNetwork over reaches `real` (routed here) followed by `halos` (tributary outlets
    routed elsewhere).  UREACHI order is preserved; a reach whose downstream is not local becomes an
    outlet; halo reaches keep their parameters but have no upstreams and no HRUs."""
    loc = np.concatenate([real, halos]).astype(np.int64)
    n_loc = loc.size
    g2l = np.full(net.N, -1, dtype=np.int64)
    g2l[loc] = np.arange(n_loc)
    down0 = net.downIndex.astype(np.int64) - 1
    dl = np.where(down0[loc] >= 0, g2l[np.maximum(down0[loc], 0)], -1)
    downIndex = (dl + 1).astype(np.int32)
    is_halo = np.zeros(n_loc, bool); is_halo[real.size:] = True
    upOff = [0]; upIdx = []; upGood = []
    hruOff = [0]; hruIdx = []; hruW = []
    hru_g = []
    for k, g in enumerate(loc):
        if not is_halo[k]:
            for e in range(net.upOffset[g], net.upOffset[g + 1]):
                u = g2l[net.upIndex[e] - 1]
                assert u >= 0, "upstream reach missing from the domain"
                upIdx.append(u + 1); upGood.append(int(net.upGood[e]))
            for e in range(net.hruOffset[g], net.hruOffset[g + 1]):
                hru_g.append(int(net.hruIndex[e]) - 1); hruW.append(float(net.hruWeight[e]))
                hruIdx.append(len(hru_g))
        upOff.append(len(upIdx)); hruOff.append(len(hruIdx))
    params = {k: np.ascontiguousarray(v[loc]) for k, v in net.params.items()}
    sub = RiverNetwork(N=n_loc, H=max(1, len(hru_g)), downIndex=downIndex, reachId=net.reachId[loc].astype(np.int32),
                       upOffset=np.array(upOff, np.int32), upIndex=np.array(upIdx, np.int32),
                       upGood=np.array(upGood, np.int32), hruOffset=np.array(hruOff, np.int32),
                       hruIndex=np.array(hruIdx, np.int32), hruWeight=np.array(hruW, np.float64), params=params)
    return sub, loc, np.array(hru_g, np.int64)


def mainstem_cost(net: RiverNetwork, n_parts: int, window: int, level_s: float = 38e-6, reach_steps_per_s: float = 2.9e9) -> float:
    """What the mainstem domain of an n_parts-way decomposition costs per window, in tributary reaches (KWT on MI355X,
    profiles/r03_loopback_c3.json): it is a chain of (its stages + window) dependent levels of ~38 us each, while a tributary
    domain routes ~2.9e9 reach-steps/s."""
    down0 = net.downIndex.astype(np.int64) - 1
    is_main = subtree_sizes(net) > (net.N // max(1, n_parts))
    if not is_main.any():
        return 0.0
    dist = hops_to_outlet(down0)
    depth = int(dist[is_main].max()) + 1
    return (depth + window) * level_s * reach_steps_per_s / window


def halo_flags(lakes: dict, spec: "Domain") -> np.ndarray:
    """halo_good of a mainstem domain with the lakes of the whole network marked (bit 1): a tributary outlet that is a lake where
    it is routed reaches the mainstem reach below it as a lake does (kwt_route.f90:540-559; mzr_set_boundary)."""
    hg = np.asarray(spec.halo_good, dtype=np.int32).copy()
    if lakes is None or hg.size == 0:
        return hg
    is_lake = np.zeros(int(max(spec.reach_global.max(), np.asarray(lakes["reach"]).max())) + 1, bool)
    is_lake[np.asarray(lakes["reach"], dtype=np.int64) - 1] = True
    halo_global = spec.reach_global[spec.n_real:]
    return hg | (2 * is_lake[halo_global]).astype(np.int32)


def lakes_for_domain(lakes: dict, spec: "Domain", n_reach_global: int):
    """The lakes / reservoirs of the whole network (dict of synthetic.make_lakes or standalone.read_lakes: 1-based global
    reaches, parameters per lake, evaporation / precipitation per HRU, target volumes per reach) as ONE domain sees them: the
    lakes among the reaches it routes, with the forcing columns of its HRUs.  A lake is routed where it lies; the domain
    downstream sees its discharge through the boundary record like any other tributary outlet.  None if the domain has none."""
    if lakes is None:
        return None
    g2l = np.full(n_reach_global, -1, dtype=np.int64)
    g2l[spec.reach_global[:spec.n_real]] = np.arange(spec.n_real)
    loc = g2l[np.asarray(lakes["reach"], dtype=np.int64) - 1]
    sel = np.nonzero(loc >= 0)[0]
    if sel.size == 0:
        return None
    sel = sel[np.argsort(loc[sel], kind="stable")]
    out = dict(input_option=lakes["input_option"], calendar_id=lakes["calendar_id"], ymd=lakes["ymd"],
               reach=(loc[sel] + 1).astype(np.int32), model_type=np.asarray(lakes["model_type"])[sel].astype(np.int32),
               par=np.ascontiguousarray(np.asarray(lakes["par"])[:, sel]))
    for k in ("evap", "precip"):
        if k in lakes and lakes[k] is not None:
            out[k] = np.ascontiguousarray(np.asarray(lakes[k])[:, spec.hru_global]) if spec.hru_global.size else np.zeros((len(lakes["ymd"]), 1))
    if "targ_vol" in lakes:
        out.update(targ_vol=np.asarray(lakes["targ_vol"])[sel].astype(np.int32), vol_jumpstart=lakes.get("vol_jumpstart", 0),
                   wm_vol=np.ascontiguousarray(np.asarray(lakes["wm_vol"])[:, spec.reach_global]))
    return out


def gauges_for_domain(da: dict, spec: "Domain", n_reach_global: int):
    """Direct insertion of gauge observations (qmodOption 1) as ONE domain sees it: the same gauges and observation columns,
    every gauge linked to its reach if the domain routes that reach and to none otherwise (what `mzr_set_da` takes for a gauge
    outside the network).  The corrected discharge of a tributary outlet reaches the mainstem through its boundary record."""
    if da is None:
        return None
    g2l = np.full(n_reach_global, -1, dtype=np.int64)
    g2l[spec.reach_global[:spec.n_real]] = np.arange(spec.n_real)
    gr = np.asarray(da["gauge_reach"], dtype=np.int64)
    loc = np.where(gr >= 1, g2l[np.maximum(gr, 1) - 1], -1)
    out = dict(da)
    out["gauge_reach"] = np.where(loc >= 0, loc + 1, -9999).astype(np.int32)
    return out


def partition_network(net: RiverNetwork, n_parts: int, build_for=None, main_cost: float = 0.0, sort_index=None) -> Partition:
    """build_for: partitions whose Domain objects (local networks) are materialised; None = all.
    A rank of a multi-GPU job passes [rank]; the assignment itself is always computed in full.
    main_cost, sort_index: see reference_domains (0 = the reference's assignment; None = stable argsort)."""
    N = net.N
    down0 = net.downIndex.astype(np.int64) - 1
    # domains and their nodes exactly as the reference makes them (pinned against the compiled reference routines,
    # the test test_domain_decomposition_matches_the_reference); partition p = node p, the mainstem
    # (node -1, "handled in root proc") goes to partition 0
    kind, outlet, size, node, is_main, root_of = reference_domains(net, n_parts, main_cost, sort_index)
    roots = np.sort(outlet[kind == 1])
    part_of_root = {int(o): int(max(nd, 0)) for o, nd, k in zip(outlet, node, kind) if k == 1}
    part_of_reach = np.zeros(N, dtype=np.int64)
    nm = ~is_main
    part_of_reach[nm] = np.array([part_of_root[int(x)] for x in root_of[nm]], dtype=np.int64) if nm.any() else 0
    part_of_reach[is_main] = 0
    trib = []
    exports = []                                       # per partition: global indices of export reaches
    want = set(range(n_parts)) if build_for is None else set(build_for)
    root_part = np.array([part_of_root[int(r)] for r in roots], dtype=np.int64) if roots.size else np.zeros(0, np.int64)
    for p in range(n_parts):
        ex_g = roots[(root_part == p) & (down0[roots] >= 0)].astype(np.int64)
        exports.append(ex_g)
        if p not in want:
            trib.append(Domain(part=p, kind="trib", net=None, reach_global=np.zeros(0, np.int64),
                               hru_global=np.zeros(0, np.int64), n_real=int((nm & (part_of_reach == p)).sum()),
                               export_local=np.zeros(ex_g.size, np.int32)))
            continue
        real = np.nonzero(nm & (part_of_reach == p))[0]
        sub, loc, hru_g = _local_network(net, real, np.zeros(0, np.int64))
        g2l = np.full(N, -1, dtype=np.int64); g2l[loc] = np.arange(loc.size)
        ex_l = (g2l[ex_g] + 1).astype(np.int32)
        trib.append(Domain(part=p, kind="trib", net=sub, reach_global=loc, hru_global=hru_g, n_real=real.size,
                           export_local=ex_l))
    main = None
    if is_main.any() and 0 not in want:
        hb, base = {}, 0
        for p in range(n_parts):
            hb[p] = (base, int(exports[p].size)); base += int(exports[p].size)
        main = Domain(part=0, kind="main", net=None, reach_global=np.zeros(0, np.int64), hru_global=np.zeros(0, np.int64),
                      n_real=int(is_main.sum()), halo_base=hb)
    if is_main.any() and 0 in want:
        real = np.nonzero(is_main)[0]
        halos = np.concatenate(exports) if exports else np.zeros(0, np.int64)
        sub, loc, hru_g = _local_network(net, real, halos)
        halo_local = (np.arange(real.size, real.size + halos.size) + 1).astype(np.int32)
        up_cnt_good = np.array([int(net.upGood[net.upOffset[g]:net.upOffset[g + 1]].sum()) for g in halos], dtype=np.int32)
        base, hb = 0, {}
        for p in range(n_parts):
            hb[p] = (base, int(exports[p].size)); base += int(exports[p].size)
        main = Domain(part=0, kind="main", net=sub, reach_global=loc, hru_global=hru_g, n_real=real.size,
                      halo_local=halo_local, halo_good=(up_cnt_good > 0).astype(np.int32), halo_base=hb)
    return Partition(n_parts=n_parts, trib=trib, main=main, part_of_reach=part_of_reach, is_mainstem=is_main)


class PartitionedRouter:
    """Drives the domains of ONE partition (rank) window by window.

    transport: object with send(tensor, dst) / recv(tensor, src) (torch.distributed P2P over RCCL
    in production; an in-process loopback in the single-GPU test).  alloc(n): n doubles of device memory with
    no work pending on them (torch.empty, not torch.zeros: a record is packed on the library's own streams,
    which do not wait for a fill kernel on torch's).  make_domain(domain, **kw) builds
    the compute object for a Domain (RoutingDomain in production).
    """

    def __init__(self, part: Partition, rank: int, make_domain, transport, alloc, max_window: int, main_thread: bool = False):
        self.part, self.rank, self.transport, self.alloc, self.W = part, rank, transport, alloc, max_window
        td = part.trib[rank]
        self.trib_spec = td
        self.main_spec = part.main if (rank == 0 and part.main is not None) else None
        # rank 0 routes its tributary window k and the mainstem window k-1 side by side on one GPU: the two persistent
        # sweeps share the device's wavefront slots out (a sweep whose grid does not fit the device can stall)
        both = td.n_real > 0 and self.main_spec is not None
        self.trib = make_domain(td, export_reaches=td.export_local, sweep_share=(1.0 - MAIN_SWEEP_SHARE) if both else 1.0) if td.n_real > 0 else None
        self.main = None
        if self.main_spec is not None:
            ms = self.main_spec
            self.main = make_domain(ms, halo_reaches=ms.halo_local, halo_good=ms.halo_good, sweep_share=MAIN_SWEEP_SHARE if both else 1.0,
                                    sweep_priority=1 if both else 0)
        self.n_routes = None
        self._pending = None            # (w, t_start, runoff_main_ptr, record, keep) of the window whose exchange is still due
        self._late = False              # ... and whose record has not been packed yet (overlapping windows: export_boundary_prev)
        # where two domains share a GPU (rank 0) the tributary domain exports right behind its window: a domain that keeps a window
        # queued ahead holds the hardware queues its neighbour's launches need (measured on the c4 network: 1.33 s per window of
        # rank 0 with the record one window later against 0.71 s)
        self._may_lag = not both
        # main_thread: rank 0 queues its mainstem window from a host thread of its own, beside the tributary window.  For domains of the
        # Eulerian methods, whose windows are thousands of launches: a launch blocks the calling thread while its stream's queue is
        # full, so one thread queues the two domains one after the other however many streams they have (c4 network, windows of 2 048:
        # 0.77 s per window of rank 0 from one thread, 0.62 s from two).  Not for KWT domains (one launch per window).
        self._main_thread = bool(main_thread) and both

    def _rec_size(self, dom, w, n):
        return dom.boundary_size(w, n)

    def run_window(self, w, t_start, runoff_trib_ptr, runoff_main_ptr, keep=None):
        """One window of w steps.  Pointers are device pointers to [w, H_local] runoff of the
        tributary domain and (rank 0) the mainstem domain; `keep` is any object that must stay alive
        until the window has been routed everywhere (the tensors behind the pointers).

        Pipelined by one window: the tributary domain starts window k at once, and only then the
        boundary records of window k-1 travel (send on the tributary ranks; receive, import and
        mainstem window k-1 on rank 0), so the exchange and the host time of the mainstem launches
        hide behind the tributary sweep of window k.  sync() flushes the window still in flight.
        All ranks must call run_window / sync in the same order.

        Where the tributary domain's windows overlap (Eulerian methods: the last launches of window k-1 go out
        with the first ones of window k, RoutingDomain.export_lag), the record of window k-1 is packed AFTER window
        k has been queued -- from the rows the library keeps, as soon as those launches are out (export_boundary_prev)
        -- instead of right behind window k-1, which would make the library issue the kept-back launches on their own
        and lose the overlap.  Same records, same order, same depth of the pipeline."""
        part = self.part
        n_exp = self.trib_spec.export_local.size
        ships = bool(n_exp) and part.main is not None
        prev = self._pending
        late = prev is not None and self._late            # the record of window k-1 has not been packed yet
        if prev is not None and self.trib is not None:
            if late and w == prev[0]:
                pass                                      # ... and comes out of this window's first launches
            else:
                self.trib.sync()                          # window k-1 and its export are complete (a sync issues launches kept back)
                if late:                                  # a window of another length does not overlap with it: the record now
                    rec_prev = self.alloc(self.trib.boundary_size(prev[0], n_exp))
                    self.trib.export_boundary(rec_prev.data_ptr()); self.trib.sync()
                    prev = (prev[0], prev[1], prev[2], rec_prev, prev[4]); late = False
        rec = None
        self._late = False
        worker = None
        if prev is not None and self._main_thread:       # (rank 0, two domains: the record of window k-1 is complete -- trib.sync() above)
            import threading
            box = []
            def _main_side(prev=prev):
                try:
                    self._exchange(*prev)
                except BaseException as e:                # handed to the calling thread
                    box.append(e)
            worker = threading.Thread(target=_main_side); worker.start()
            prev = None
        # Rank 0 with both domains queued from ONE thread (KWT: a window is one persistent launch per domain): the mainstem's window k-1
        # goes out BEFORE the tributary's window k.  The mainstem's sweep is a few hundred wavefronts; launched right behind the
        # tributary's, while the dispatcher is still placing that sweep's thousands of workgroups, one launch in ten or so saw its
        # wavefronts start late -- and a sweep wavefront that starts 20 us behind the first of its launch does not join (DESIGN.md
        # 2.4): the mainstem window then ran on 64 wavefronts, 1.2-1.4 s instead of 0.3 (round 6, bench.py --loopback --config c3).
        # Launched first it is resident before the large sweep arrives.  Same records, same results.
        if prev is not None and self.trib is not None and self.main is not None and not self._main_thread and not late:
            self._exchange(*prev)
            prev = None
        if self.trib is not None:
            self.trib.run_device(w, t_start, runoff_trib_ptr)
            if ships:
                if late:
                    rec_prev = self.alloc(self.trib.boundary_size(prev[0], n_exp))
                    self.trib.export_boundary_prev(rec_prev.data_ptr())
                    self.trib.wait_export()               # (the record only: window k runs on)
                    prev = (prev[0], prev[1], prev[2], rec_prev, prev[4])
                if self._may_lag and getattr(self.trib, "export_lag", None) is not None and self.trib.export_lag():
                    self._late = True
                else:
                    rec = self.alloc(self.trib.boundary_size(w, n_exp))
                    self.trib.export_boundary(rec.data_ptr())
        if worker is not None:
            worker.join()
            if box:
                raise box[0]
        if prev is not None:
            self._exchange(*prev)
        self._pending = (w, t_start, runoff_main_ptr, rec, keep) if part.main is not None else None

    def _exchange(self, w, t_start, runoff_main_ptr, rec, keep):
        """Boundary records of a finished tributary window -> mainstem halos -> mainstem window."""
        part = self.part
        if self.rank != 0:
            if rec is not None:
                self.transport.send(rec, 0)
            return
        # all records at once where the transport can (one link per peer: the transfers run side by side),
        # one after the other otherwise
        bufs = {}
        for p in range(1, part.n_parts):
            base, n = self.main_spec.halo_base[p]
            if n:
                bufs[p] = self.alloc(self.main.boundary_size(w, n))
        recv_many = getattr(self.transport, "recv_many", None)
        if recv_many is not None and len(bufs) > 1:
            recv_many([(bufs[p], p) for p in sorted(bufs)])
        else:
            for p in sorted(bufs):
                self.transport.recv(bufs[p], p)
        for p in range(part.n_parts):
            base, n = self.main_spec.halo_base[p]
            if n == 0:
                continue
            buf = rec if p == 0 else bufs[p]
            self.main.import_boundary(w, buf.data_ptr(), n, base)
        # the imports have consumed the buffers.  (wait_import, not sync: a mainstem domain of the Eulerian methods keeps the last
        # launches of its window back for the next one -- its windows overlap like a tributary domain's -- and sync would issue them)
        wait = getattr(self.main, "wait_import", None)
        if wait is not None:
            wait()
        else:
            self.main.sync()
        self.main.run_device(w, t_start, runoff_main_ptr)

    def sync(self):
        if self._pending is not None:
            if self.trib is not None:
                self.trib.sync()
            prev, self._pending = self._pending, None
            if self._late and self.trib is not None:      # the last window's record: its kept-back launches have just been issued
                rec = self.alloc(self.trib.boundary_size(prev[0], self.trib_spec.export_local.size))
                self.trib.export_boundary(rec.data_ptr()); self.trib.sync()
                prev = (prev[0], prev[1], prev[2], rec, prev[4])
            self._late = False
            self._exchange(*prev)
        if self.trib is not None:
            self.trib.sync()
        if self.main is not None:
            self.main.sync()
