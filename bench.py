#!/usr/bin/env python
"""Benchmark of the routing hot path on MI355X: reaches*timesteps/s on the BASELINE.json workload.

  python bench.py [--gpus N] [--steps K] [--warmup W]          (N=1)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): synthetic HDMA-CONUS-like sub-basin, ~100k reaches per GPU,
KWT routing (route_opt 2), dt = 3600 s, hillslope unit-hydrograph delay on (fshape 2.5,
tscale 86400 s).  A bench "step" is one pass of the hot path over one BATCH of synthetic forcing =
one forcing window of `window_steps` model time steps (16384 by default; one main_route call per
model time step in the reference; the device sweeps a window time-skewed over the stages,
DESIGN.md 2): cold start, W untimed batches, then exactly K timed batches.  `value` counts MODEL
time steps: reaches x K x window_steps / elapsed.  Forcing windows are generated on the device
before the timed region (two of them, used alternately), so `value` is the HBM-resident rate;
`value_with_h2d` (N = 1) repeats the timed region with every window handed over in page-locked host
memory (mzr_run_async: copy on its own stream behind the sweep of the window before), and
`single_step` times mzr_step, one main_route-equivalent call per model time step (the coupled-model use).

One JSON line on rank 0 with the contract fields plus
  "roofline":     HBM roofline of the dominant kernel (KWT stage sweep): algorithmic bytes from the
                  device particle counters x SURVEY.md 8(d) byte model, / summed kernel time from HIP
                  events recorded around every stage launch on the library's stream;
  "cpu_baseline": the reference's own Fortran solvers (oracle/_ref, unmodified sources) timed on the
                  host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_REACH = 100_000          # reaches per GPU (BASELINE.json configs[1])
DT = 3600.0
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


LINE_LIMIT = 4096          # bytes of the LAST stdout line (the driver keeps a bounded tail of stdout and parses that line)


def _r(x, sig=6):
    """floats to `sig` significant digits (the detail file keeps every digit)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        return float(f"{x:.{sig}g}")
    return x


def _pick(d, keys):
    return {k: _r(d.get(k)) for k in keys if isinstance(d, dict) and k in d}


def _cpu_summary(c):
    """cpu_baseline of the line: value / cores / kind / cpu_model and the two MPI-shaped forms as {value, cores}; prose stays in the detail"""
    if not isinstance(c, dict):
        return None
    out = _pick(c, ("value", "unit", "cores", "kind", "host_cores", "cpu_model", "one_thread"))
    s_ = c.get("sample")
    if isinstance(s_, str):
        out["sample"] = s_[:160]
    for k in ("mpi_like", "mpi_like_cores"):
        v = c.get(k)
        if isinstance(v, dict) and v.get("value") is not None:
            out[k] = _pick(v, ("value", "cores", "processes", "threads_per_process"))
    return out


def _roof_summary(r):
    if not isinstance(r, dict):
        return None
    return _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "bytes_per_reach_step",
                     "avg_launch_us", "launches", "min_launch_us", "max_launch_us"))


def _config_summary(c):
    """one flat object per full-size configuration"""
    if not isinstance(c, dict):
        return None
    if c.get("value") is None:
        return {"value": None, "error": str(c.get("error"))[:160]}
    if not isinstance(c.get("model_8gpu"), dict) and "rank0_over_slowest" in c:      # already a summary (a line fed back in): as it is
        return {k: _r(v) for k, v in c.items()}
    roof = c.get("roofline") or {}
    mod = c.get("model_8gpu") or {}
    cpu = c.get("cpu_baseline") or {}
    par = c.get("parity") or {}
    main = (c.get("domains") or {}).get("main") or {}
    slow = mod.get("slowest_tributary_s")
    out = {"value": _r(c.get("value")), "window_steps": c.get("window_steps"), "kernel": roof.get("kernel"), "frac": _r(roof.get("frac"), 4),
           "avg_launch_us": _r(roof.get("avg_launch_us"), 5), "parity": par.get("partitioned_equals_whole_bit_for_bit"),
           "model_8gpu": _r(mod.get("value")), "rank0_over_slowest": _r(mod.get("rank0_side_by_side_s") / slow, 4) if slow else None,
           "record_bytes_per_window": main.get("record_bytes_per_window"),
           "cpu": _r(cpu.get("value")), "cpu_cores": cpu.get("cores")}
    for k in ("mpi_like", "mpi_like_cores"):
        v = cpu.get(k)
        if isinstance(v, dict) and v.get("value") is not None:
            out["cpu_" + k] = {"value": _r(v["value"]), "cores": v.get("cores")}
    return out


def compact_line(detail, detail_path="bench_detail.json"):
    """The ONE line the driver parses, <= LINE_LIMIT bytes: contract fields, `roofline`, `cpu_baseline` and one flat summary per
    full-size configuration.  Per-launch lists, per-domain arrays, histograms and prose live in the detail object (bench_detail.json)."""
    cfg = detail.get("config") or {}
    out = {k: _r(detail.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                          "vs_baseline", "dtype", "data")}
    c2 = _pick(cfg, ("baseline_config", "route_opt", "reaches_total", "reaches_per_gpu", "window_steps", "stages", "simulated_years_per_wallclock_day",
                     "kernel_time_fraction"))
    c2 = {"workload": str(cfg.get("workload", ""))[:420], **c2}
    out["config"] = c2
    for k in ("rccl_ranks", "backend"):
        if k in detail:
            out[k] = detail[k]
    out["roofline"] = _roof_summary(detail.get("roofline"))
    out["cpu_baseline"] = _cpu_summary(detail.get("cpu_baseline"))
    for k in ("value_with_h2d", "value_with_h2d_f64"):
        if isinstance(detail.get(k), (int, float)):
            out[k] = _r(detail[k])
    ss = detail.get("single_step")
    if isinstance(ss, dict):
        out["single_step"] = {"value": _r(ss.get("value")), "pipelined": _r(ss["pipelined"].get("value")) if isinstance(ss.get("pipelined"), dict) else None}
    if detail.get("kwt_sweep_retries") is not None:
        out["kwt_sweep_retries"] = detail["kwt_sweep_retries"]
    if isinstance(detail.get("configs"), dict):
        out["configs"] = {k: _config_summary(v) for k, v in detail["configs"].items()}
    out["detail"] = detail_path
    out["error"] = None if detail.get("error") is None else str(detail["error"])[:300]
    line = json.dumps(out, separators=(",", ":"))
    if len(line) > LINE_LIMIT:      # never print a line the driver cannot keep: drop the optional parts, largest first
        for k in ("configs", "single_step", "value_with_h2d_f64", "value_with_h2d"):
            out.pop(k, None)
            line = json.dumps(out, separators=(",", ":"))
            if len(line) <= LINE_LIMIT:
                break
    return line


def emit(detail):
    """detail -> bench_detail.json (beside bench.py, and under gpurun_out/ so that it comes back from a GPU box), compact line -> stdout (last line)"""
    paths = [os.path.join(ROOT, "bench_detail.json")]
    god = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(god):
        paths.append(os.path.join(god, "bench_detail.json"))
    for p_ in paths:
        try:
            with open(p_, "w") as f:
                json.dump(detail, f)
        except OSError:
            pass
    sys.stdout.flush()
    print(compact_line(detail), flush=True)


# BASELINE.json configs as one GPU sees them: c2 = the headline (~100 k reaches, KWT); c3-c5 = one of eight shards
# (~3 M / 8 KWT; ~5 M / 8 IRF + Muskingum-Cunge; ~3 M / 8 diffusive wave, 1 % lakes, floodplains).
# `dominant` = the method whose kernel the roofline object describes, `bytes` = SURVEY.md 8(d) bytes per reach-step
# of that method for U immediate upstream reaches (KWT: from the particle counters instead).
CONFIGS = {
    "c2": dict(reaches=100_000, methods="2", window=16384, workload="synthetic HDMA-CONUS-like sub-basin, KWT (route_opt 2), dt 3600 s, hillslope UH on"),
    "c3": dict(reaches=375_000, methods="2", window=8192, workload="one of 8 shards of a ~3 M-reach HDMA-CONUS-like network, KWT (route_opt 2), dt 3600 s, hillslope UH on"),
    "c4": dict(reaches=625_000, methods="14", window=3072, dominant=4, bytes=lambda U: 152 + 12 * U,
               workload="one of 8 shards of a ~5 M-reach MERIT-like network, IRF-UH + Muskingum-Cunge (route_opt 14), dt 3600 s, hillslope UH on"),
    "c5": dict(reaches=375_000, methods="5", window=2048, dominant=5, bytes=lambda U: 440 + 12 * U, lakes=0.01, floodplain=True,
               workload="one of 8 shards of a ~3 M-reach HDMA-CONUS-like network, diffusive wave (route_opt 5), 1 % lakes (Doll / Hanasaki / HYPE), "
                        "floodplains, dt 3600 s, hillslope UH on"),
}


def kwt_bytes(tr):
    """Algorithmic bytes of the KWT sweep from particle counters:
    routed reach: 25*(W_in + W_up + W_out) + 68 + 44*U ; headwater reach: 53."""
    return 25 * (tr["w_in"] + tr["w_up"] + tr["w_out"]) + 68 * tr["n_route"] + 44 * tr["n_edges"] + 53 * tr["n_head"]


def device_runoff(torch, H, n_steps, t0, seed, device):
    """runoff[t, h] [m/s] on the device: low seasonal base flow + sparse storm pulses."""
    g = torch.Generator(device=device); g.manual_seed(seed)
    phi = torch.rand(H, generator=g, device=device, dtype=torch.float64) * (2 * np.pi)
    g2 = torch.Generator(device=device); g2.manual_seed(seed * 1000003 + t0 + 17)
    t = torch.arange(t0, t0 + n_steps, device=device, dtype=torch.float64)[:, None]
    ro = 1e-8 * (1.0 + torch.sin(2 * np.pi * t / 168.0 + phi[None, :]))
    pulse = torch.rand((n_steps, H), generator=g2, device=device, dtype=torch.float64)
    amp = torch.rand((n_steps, H), generator=g2, device=device, dtype=torch.float64)
    ro += torch.where(pulse < 0.01, 1e-6 * amp * amp, torch.zeros_like(amp))
    return ro.contiguous()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(net, frac, runoff, spinup_steps=240, sample_steps=240, methods=(2,), uh=None, lakes=None, one_thread=True):
    """Reference Fortran solvers (oracle/_ref/ref_route) on the host cores, bounded sample of the SAME
    forcing the GPU leg routes: `sample_steps` steps timed after `spinup_steps` untimed ones (particle lists at
    steady state), OpenMP over the reference's own stream-order branches at 16 threads (the best count of the
    round-1 sweep 1/8/16/32/64; fewer if the host has fewer cores), plus a short 1-thread run."""
    from oracle import refrun
    if not refrun.available():
        return None
    cores = os.cpu_count() or 1
    uh_off, uhv = uh if uh is not None else (np.arange(net.N + 1, dtype=np.int32), np.ones(net.N))
    methods = list(methods)
    nt = max(1, min(16, cores))
    sched = refrun.streamorder_schedule(net)
    kw = dict(uh=(frac, uh_off, uhv), dump_every=0)

    def lk(n):      # lake forcing of the first n steps
        return {} if lakes is None else dict(lakes=dict(lakes, evap=lakes["evap"][:n], precip=lakes["precip"][:n], ymd=lakes["ymd"][:n]))

    n_all = spinup_steps + sample_steps
    r = refrun.run_case(net, runoff[:n_all], DT, methods, nthreads=nt, schedule=sched, time_from=spinup_steps, **kw, **lk(n_all))
    one = refrun.run_case(net, runoff[:36], DT, methods, nthreads=1, time_from=24, **kw, **lk(36)) if one_thread else {"ierr": 0, "reach_steps_per_s": None}
    if r["ierr"] or one["ierr"]:
        raise RuntimeError(f"reference harness ierr {r['ierr']} at step {r['ierr_step']}: {r['stdout'][-300:]}")
    # (the harness counts a reach-step per active method, as `value` does)
    return {"value": r["reach_steps_per_s"], "unit": "reaches*timesteps/s", "cores": nt, "kind": "reference",
            "host_cores": cores, "cpu_model": cpu_model(),
            "one_thread": one["reach_steps_per_s"],
            "sample": f"same {net.N}-reach network and forcing, route_opt {''.join(str(x) for x in methods)}, {sample_steps} steps timed after {spinup_steps} spin-up "
                      f"steps at {nt} OpenMP threads over the reference's stream-order branches (main_route.f90:356-405); "
                      f"unmodified reference kwt_route.f90/main_route.f90 built with flang -O2; one_thread = 12 steps "
                      f"timed after 24 (lists not yet at steady state: an upper bound for one core)"}


def cpu_by_threads(net, frac, runoff, methods=(2,), uh=None, lakes=None, counts=(1, 16, 64, 128), spin=48, smp=48):
    """the OpenMP form of the reference (one process, its stream-order branches dealt to threads) at several thread counts:
    why cpu_baseline quotes 16"""
    from oracle import refrun
    if not refrun.available():
        return None
    cores = os.cpu_count() or 1
    uh_off, uhv = uh if uh is not None else (np.arange(net.N + 1, dtype=np.int32), np.ones(net.N))
    sched = refrun.streamorder_schedule(net)
    out = {}
    for nt in counts:
        if nt > cores:
            continue
        sp, sm = (12, 12) if nt == 1 else (spin, smp)
        kw = dict(uh=(frac, uh_off, uhv), dump_every=0)
        if lakes is not None:
            kw["lakes"] = dict(lakes, evap=lakes["evap"][:sp + sm], precip=lakes["precip"][:sp + sm], ymd=lakes["ymd"][:sp + sm])
        try:
            r = refrun.run_case(net, runoff[:sp + sm], DT, list(methods), nthreads=nt, schedule=sched if nt > 1 else None, time_from=sp, **kw)
            out[str(nt)] = None if r["ierr"] else r["reach_steps_per_s"]
        except Exception as e:
            out[str(nt)] = f"failed: {e}"
    out["sample"] = f"{smp} steps timed after {spin} (1 thread: 12 after 12), same network and forcing"
    return out


def cpu_mpi_like(domains, frac, runoff_of, methods, uh_of=None, lakes_of=None, threads=1, spin=48, smp=48, also_one_thread=True):
    """The reference's MPI form (mpi_process.f90:1088-1342) as far as it can run here: every tributary domain of the
    reference's own decomposition in a PROCESS of its own, all at once (that is what its ranks do between two exchanges), each
    with `threads` OpenMP threads; the mainstem domain (rank 0's extra, serial after the exchange in the reference) is left out,
    which flatters the CPU.  value = reach-steps of all processes / the slowest process's timed seconds."""
    from oracle import refrun
    from concurrent.futures import ThreadPoolExecutor
    if not refrun.available():
        return None
    doms = [d for d in domains if d.n_real > 0]

    def one(dm):
        net = dm.net
        uh_off, uhv = uh_of(dm) if uh_of is not None else (np.arange(net.N + 1, dtype=np.int32), np.ones(net.N))
        kw = dict(uh=(frac, uh_off, uhv), dump_every=0)
        lk = lakes_of(dm) if lakes_of is not None else None
        if lk is not None:
            kw["lakes"] = lk
        sched = refrun.streamorder_schedule(net) if threads > 1 else None
        r = refrun.run_case(net, runoff_of(dm), DT, list(methods), nthreads=threads, schedule=sched, time_from=spin, **kw)
        if r["ierr"]:
            raise RuntimeError(f"reference harness ierr {r['ierr']}")
        rs = float(net.N) * smp * len(methods)
        return rs, rs / r["reach_steps_per_s"]

    best = None
    tried = {}
    for thr in sorted({1, max(1, threads)} if also_one_thread else {max(1, threads)}):
        threads = thr
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=len(doms)) as ex:
            res = list(ex.map(one, doms))
        wall = time.perf_counter() - t0
        total = sum(a for a, _ in res)
        slowest = max(b for _, b in res)
        tried[str(thr)] = total / slowest
        if best is None or total / slowest > best[0]:
            best = (total / slowest, thr, slowest, wall, res)
    value, threads, slowest, wall, res = best
    # why P ranks need not be P times one rank: value = all reach-steps / the SLOWEST process's time, so it is bounded by the largest
    # domain of the decomposition (balance = mean process time / slowest; the reference's rule caps a tributary domain at N / P reaches,
    # which at large P leaves few big domains and many tiny ones) and by what P processes streaming particle lists leave each other
    # of the host's memory system (seconds per reach-step of the slowest process against the 1-process figure in `one_thread`)
    times = [b for _, b in res]
    return {"value": value, "unit": "reaches*timesteps/s", "processes": len(doms), "threads_per_process": threads, "by_threads_per_process": tried,
            "cores": len(doms) * threads, "kind": "reference", "timed_s_slowest_process": slowest, "wall_s_with_case_io": wall,
            "balance": float(np.mean(times) / slowest), "reaches_largest_domain": int(max(d.net.N for d in doms)),
            "reach_steps_per_s_of_the_slowest_process": float(max(a / b for a, b in res if b == slowest)),
            "sample": f"{len(doms)} tributary domains of the reference's decomposition, one ref_route process each, side by side, {smp} steps timed after {spin}; "
                      "mainstem domain and the per-step gather / scatter of mpi_route left out"}


def cpu_mpi_like_cores(net, frac, runoff_of, methods, uh_of=None, lakes_of=None, spin=48, smp=48):
    """The strongest form of the reference this host can run (its OpenMP scales badly: 16 threads are its best, 64 run three times
    slower): P single-thread processes over the reference's own P-way decomposition, P = the host's physical cores (half the logical
    ones), every rank's tributary domains in one ref_route process, all side by side.  The mainstem domain and the per-step gather /
    scatter of mpi_route (mpi_process.f90:1245-1329) are left out, which flatters the CPU."""
    from mizuroute_amd.partition import partition_network
    cores = os.cpu_count() or 16
    P = max(8, min(128, cores // 2))
    t0 = time.perf_counter()
    Pc = partition_network(net, P)
    t_part = time.perf_counter() - t0
    r = cpu_mpi_like(Pc.trib, frac, lambda dm: runoff_of(dm, spin + smp), methods, uh_of=uh_of, lakes_of=lakes_of, threads=1, spin=spin, smp=smp, also_one_thread=False)
    if r is not None:
        r["ranks"] = P
        r["partition_s"] = t_part
        r["sample"] = (f"the reference's {P}-way decomposition (domain_decomposition.f90:41-163), one single-thread ref_route process per rank (P = physical cores of "
                       f"this host: {cores} logical / 2), side by side, {smp} steps timed after {spin}; mainstem domain and per-step gather / scatter left out")
    return r


FULL = {"c3": 3_000_000, "c4": 5_000_000, "c5": 3_000_000}      # reaches of the 8-GPU configurations (BASELINE.json configs[2..4])


class Loopback:
    """The full-size network of an 8-GPU configuration on ONE GPU: cut into sub-basin partitions by the reference's rule
    (mizuroute_amd/partition.py = domain_decomposition.f90), every partition routed as its own domain one after the other,
    boundary records of the tributary outlets handed to the mainstem domain through device memory (what RCCL carries between
    GPUs).  parity(): short windows against the unpartitioned network, bit for bit (interval means, particle counts);
    timing(): windows of the configuration's length, per-domain sweep time, and what eight GPUs would take: every rank its
    tributary domain, rank 0 also the mainstem one window behind.  Used by `bench.py --loopback`, by the `configs` objects of
    the default bench line and by tests/test_gpu_scale.py."""

    def __init__(self, torch, m, uhmod, config, nparts=8, reaches=0, window=0, balance=False):
        from mizuroute_amd.partition import partition_network, mainstem_cost
        self.torch, self.m, self.config = torch, m, config
        self.dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        self.cfg = cfg = CONFIGS[config]
        self.methods = [int(c) for c in cfg["methods"]]
        self.nparts = nparts
        N = reaches or FULL[config]
        self.frac = uhmod.basin_uh(DT, 2.5, 86400.0)
        t0 = time.perf_counter()
        self.net = net = m.make_network(N, seed=20240529, floodplain=bool(cfg.get("floodplain")))
        self.Wcfg = window or cfg["window"]
        self.mc = mainstem_cost(net, nparts, self.Wcfg) if (balance and self.methods == [m.KWT]) else 0.0
        self.P = partition_network(net, nparts, main_cost=self.mc)
        self.t_setup = time.perf_counter() - t0
        self.need_uh = any(x != m.KWT for x in self.methods)
        self.uh_off, self.uhv = uhmod.make_uh(net.params["RLENGTH"], DT, 1.5, 5000.0) if self.need_uh else (None, None)
        self.lakes = None
        if cfg.get("lakes"):      # c5: 1 % of the reaches are lakes / reservoirs (Doll, Hanasaki, HYPE); one window of lake forcing, reused
            from mizuroute_amd.synthetic import make_lakes
            self.lakes = make_lakes(net, max(self.Wcfg, 256), DT, seed=9, frac=cfg["lakes"], input_option=1, forcing=False)

    def uh_of(self, spec):
        if not self.need_uh:
            return {}
        g = spec.reach_global
        cnt = np.diff(self.uh_off)[g]
        off = np.zeros(g.size + 1, np.int32); off[1:] = np.cumsum(cnt)
        idx = np.repeat(self.uh_off[g].astype(np.int64), cnt) + (np.arange(int(cnt.sum())) - np.repeat(off[:-1].astype(np.int64), cnt))
        return dict(uh_offset=off, uh=self.uhv[idx])

    def make(self, spec, W, **kw):
        from mizuroute_amd.partition import lakes_for_domain
        lk = lakes_for_domain(self.lakes, spec, self.net.N) if self.lakes is not None else None
        return self.m.RoutingDomain(spec.net, DT, self.methods, frac_future=self.frac, max_window=W, device=0, lakes=lk, **self.uh_of(spec), **kw)

    def forcing(self, W, t0s, cols=None, shared=True):
        """shared: one forcing for the whole network, a domain takes the columns of its HRUs (parity); otherwise a forcing of
        the domain's own (timing: the whole network's window would not fit beside the domains)"""
        torch = self.torch
        if not shared:
            return device_runoff(torch, len(cols), W, t0s, 7 + len(cols) % 97, self.dev)
        ro = device_runoff(torch, self.net.H, W, t0s, 7, self.dev)
        return ro if cols is None else ro[:, torch.as_tensor(cols, device=self.dev, dtype=torch.long)].contiguous()

    def route_partitioned(self, W, K, timing, extra=0):
        """all windows of one domain, then the next; returns per-reach interval means / particle counts, the times and the
        boundary records (kept on the device only when `timing`)"""
        torch, m, net, P, methods = self.torch, self.m, self.net, self.P, self.methods
        mean = {mm: np.zeros(net.N) for mm in methods}
        nw = np.zeros(net.N, np.int64)
        recs = {}                                      # (partition, window) -> boundary record on the device
        times = {}
        for p in range(self.nparts):
            sp = P.trib[p]
            if sp.n_real == 0:
                continue
            dom = self.make(sp, W, export_reaches=sp.export_local)
            tw = []
            ships = bool(sp.export_local.size) and P.main is not None
            # (`extra` more windows of the partitions 1..: their records -- a few outlets each -- feed the mainstem in the longer
            # side-by-side run of rank 0, whose own partition 0 produces its record live)
            # Windows are queued the way PartitionedRouter queues them: no synchronisation between them where the domain's windows
            # overlap (Eulerian methods: the record of window k - 1 is packed behind the START of window k, export_boundary_prev);
            # a window's time = from one record to the next.
            late = None                                  # window whose record has not been packed yet
            lagged = False
            # timing: two forcing windows made once and used in turn (two windows are in flight; a fresh 15 GB tensor per window beside
            # the one still in use fragments the allocator's pool until the largest domains no longer fit)
            ros = [self.forcing(W, k * W, sp.hru_global, shared=False) for k in range(2)] if timing else None
            torch.cuda.synchronize(); t_prev = time.perf_counter()
            for k in range(K + (extra if p > 0 else 0)):
                ro = ros[k % 2] if timing else self.forcing(W, k * W, sp.hru_global, shared=True)
                torch.cuda.current_stream().synchronize()      # (torch made it on ITS stream)
                if dom.lakes is not None:
                    dom.set_lake_forcing(0, W)
                dom.run_device(W, k * W * DT, ro.data_ptr())
                if late is not None:
                    rec = torch.empty(dom.boundary_size(W, sp.export_local.size), dtype=torch.float64, device=self.dev)
                    dom.export_boundary_prev(rec.data_ptr()); dom.wait_export()
                    recs[(p, late)] = rec; late = None
                    now = time.perf_counter(); tw.append(now - t_prev); t_prev = now
                if ships and timing and dom.export_lag():
                    late = k; lagged = True
                    continue
                dom.sync()
                now = time.perf_counter(); tw.append(now - t_prev)
                if ships:
                    rec = torch.empty(dom.boundary_size(W, sp.export_local.size), dtype=torch.float64, device=self.dev)
                    dom.export_boundary(rec.data_ptr()); dom.sync()
                    recs[(p, k)] = rec
                t_prev = time.perf_counter()
            if late is not None:                          # the last window: its kept-back launches go out with the synchronisation
                dom.sync()
                tw.append(time.perf_counter() - t_prev)
                rec = torch.empty(dom.boundary_size(W, sp.export_local.size), dtype=torch.float64, device=self.dev)
                dom.export_boundary(rec.data_ptr()); dom.sync()
                recs[(p, late)] = rec
            del ro, ros
            times[f"trib{p}"] = dict(reaches=int(sp.n_real), stages=dom.schedule()[0], exports=int(sp.export_local.size), s_per_window=tw, record_one_window_later=lagged)
            for mm in methods:
                mean[mm][sp.reach_global[:sp.n_real]] = dom.mean_q(mm)[:sp.n_real]
            if m.KWT in methods:
                nw[sp.reach_global[:sp.n_real]] = dom.kwt_state()[0][:sp.n_real]
            dom.close(); del dom
            torch.cuda.empty_cache()
        if P.main is not None:
            ms = P.main
            dom = self.make(ms, W, halo_reaches=ms.halo_local, halo_good=ms.halo_good)
            tw = []
            for k in range(K):
                ro = self.forcing(W, k * W, ms.hru_global, shared=not timing)
                for p in range(self.nparts):
                    base, n = ms.halo_base[p]
                    if n:
                        dom.import_boundary(W, recs[(p, k)].data_ptr(), n, base)
                # (no dom.sync() between the windows: the mainstem's windows overlap too since round 6 -- a synchronisation of the
                # handle would issue the launches it keeps back for the next window; the device-wide one below does not)
                dom.wait_import()
                if dom.lakes is not None:
                    dom.set_lake_forcing(0, W)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                dom.run_device(W, k * W * DT, ro.data_ptr()); torch.cuda.synchronize()
                tw.append(time.perf_counter() - t1)
                del ro
            dom.sync()
            times["main"] = dict(reaches=int(ms.n_real), halos=int(ms.halo_local.size), stages=dom.schedule()[0], s_per_window=tw,
                                 record_bytes_per_window=int(sum(r.numel() for (p, k), r in recs.items() if k == 0) * 8))
            for mm in methods:
                mean[mm][ms.reach_global[:ms.n_real]] = dom.mean_q(mm)[:ms.n_real]
            if m.KWT in methods:
                nw[ms.reach_global[:ms.n_real]] = dom.kwt_state()[0][:ms.n_real]
            dom.close(); del dom
        if not timing:
            recs.clear()
        torch.cuda.empty_cache()
        return mean, nw, times, recs

    def parity(self, Wa=256, Ka=2):
        """the partitioned network against the whole one: Ka windows of Wa steps; returns the report and both sets of results"""
        torch, m, net, methods = self.torch, self.m, self.net, self.methods
        extra = dict(uh_offset=self.uh_off, uh=self.uhv) if self.need_uh else {}
        whole = m.RoutingDomain(net, DT, methods, frac_future=self.frac, max_window=Wa, device=0, lakes=self.lakes, **extra)
        tw = []
        for k in range(Ka):
            ro = self.forcing(Wa, k * Wa)
            if self.lakes is not None:
                whole.set_lake_forcing(0, Wa)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            whole.run_device(Wa, k * Wa * DT, ro.data_ptr()); whole.sync()
            tw.append(time.perf_counter() - t1)
            del ro
        mean_w = {mm: whole.mean_q(mm) for mm in methods}
        nw_w = whole.kwt_state()[0] if m.KWT in methods else None
        whole_info = dict(stages=whole.schedule()[0], window_steps=Wa, s_per_window=tw, value=float(net.N) * Wa * len(methods) / tw[-1])
        whole.close(); del whole
        torch.cuda.empty_cache()
        mean_p, nw_p, times, _ = self.route_partitioned(Wa, Ka, False)
        same = all(np.array_equal(mean_p[mm], mean_w[mm]) for mm in methods) and (nw_w is None or np.array_equal(nw_p, nw_w))
        rep = {"partitioned_equals_whole_bit_for_bit": bool(same), "window_steps": Wa, "windows": Ka,
               "max_abs_diff": float(max(np.abs(mean_p[mm] - mean_w[mm]).max() for mm in methods))}
        return rep, whole_info, dict(mean_whole=mean_w, mean_part=mean_p, nw_whole=nw_w, nw_part=nw_p, domains=times)

    def timing(self, W, K, side_by_side=True):
        """K windows of W steps of every domain (the first is dropped as warm-up); rank 0's two domains side by side"""
        torch, m, net, P, methods = self.torch, self.m, self.net, self.P, self.methods
        from mizuroute_amd.partition import lakes_for_domain
        EXTRA = 4 if side_by_side else 0
        _, _, times, recs = self.route_partitioned(W, K, True, extra=EXTRA)
        # (a domain whose records come one window later: entry 0 = window 0 and the head of window 1, the last entry = the rest of the
        # last window; the entries between are whole windows)
        med = lambda d: float(np.median(d["s_per_window"][1:-1] if d.get("record_one_window_later") and len(d["s_per_window"]) >= 3 else d["s_per_window"][1:]))
        trib = {k: med(v) for k, v in times.items() if k.startswith("trib")}
        t_main = med(times["main"]) if "main" in times else 0.0
        one_gpu = sum(trib.values()) + t_main
        # rank 0 as it really runs (PartitionedRouter): its tributary window k and the mainstem window k-1 SIDE BY SIDE on one GPU;
        # the other partitions' records are the ones measured above
        t_rank0 = trib.get("trib0", 0.0) + t_main
        if side_by_side and P.main is not None and P.trib[0].n_real > 0:
            sp, ms = P.trib[0], P.main
            lk_t = lakes_for_domain(self.lakes, sp, net.N) if self.lakes is not None else None
            lk_m = lakes_for_domain(self.lakes, ms, net.N) if self.lakes is not None else None
            share = main_sweep_share(ms, methods, m)
            d_t = m.RoutingDomain(sp.net, DT, methods, frac_future=self.frac, max_window=W, device=0, sweep_share=1.0 - share, export_reaches=sp.export_local,
                                  lakes=lk_t, **self.uh_of(sp))
            d_m = m.RoutingDomain(ms.net, DT, methods, frac_future=self.frac, max_window=W, device=0, sweep_share=share, sweep_priority=1, halo_reaches=ms.halo_local,
                                  halo_good=ms.halo_good, lakes=lk_m, **self.uh_of(ms))
            ro_t = [self.forcing(W, k * W, sp.hru_global, shared=False) for k in range(2)]
            ro_m = [self.forcing(W, k * W, ms.hru_global, shared=False) for k in range(2)]
            rec0 = [torch.empty(d_t.boundary_size(W, sp.export_local.size), dtype=torch.float64, device=self.dev) for _ in range(2)]
            tw, tw_trib, tw_join = [], [], []
            import threading
            two_threads = m.KWT not in methods
            kwt_leg = m.KWT in methods
            KS = K + EXTRA      # (two fresh domains: their first windows hold the regroupings and table builds)
            # (rank 0's tributary domain exports right behind its window, as PartitionedRouter does where two domains share a GPU: a
            # domain that keeps a window queued ahead holds the hardware queues its neighbour's launches need -- measured on c4: 1.33 s
            # per window with the record one window later against 0.71 s)
            for k in range(KS + 1):
                torch.cuda.synchronize(); t1 = time.perf_counter()
                def main_side(k=k):                          # the mainstem follows one window behind
                    for p in range(self.nparts):
                        base, n = ms.halo_base[p]
                        if n:
                            d_m.import_boundary(W, (rec0[(k - 1) % 2] if p == 0 else recs[(p, k - 1)]).data_ptr(), n, base)
                    if not kwt_leg:
                        d_m.wait_import()                    # (the record buffers may go; the mainstem's overlapping windows stay)
                    if d_m.lakes is not None:
                        d_m.set_lake_forcing(0, W)
                    d_m.run_device(W, (k - 1) * W * DT, ro_m[(k - 1) % 2].data_ptr())
                # Eulerian methods: a window is thousands of launches, and a launch blocks its thread while the stream's queue is full --
                # the mainstem's window is queued from a host thread of its own (PartitionedRouter, main_thread)
                th = None
                if k >= 1 and two_threads:
                    th = threading.Thread(target=main_side); th.start()
                main_first = kwt_leg and os.environ.get("MZR_BENCH_MAIN_FIRST", "1") != "0"
                if k >= 1 and main_first:      # KWT: the small sweep is launched -- and resident -- before the large one's thousands of workgroups are dispatched
                    main_side()
                if k < KS:
                    if d_t.lakes is not None:
                        d_t.set_lake_forcing(0, W)
                    d_t.run_device(W, k * W * DT, ro_t[k % 2].data_ptr())
                if th is not None:
                    th.join()
                elif k >= 1 and not main_first:
                    main_side()
                if k < KS:
                    d_t.sync()
                    t_trib = time.perf_counter() - t1
                    d_t.export_boundary(rec0[k % 2].data_ptr())
                d_t.sync()
                if kwt_leg:
                    d_m.sync()                            # (a KWT mainstem has nothing to keep back: the handle's own synchronisation, as in rounds 3-5)
                else:
                    torch.cuda.synchronize()              # (not d_m.sync(): the mainstem keeps its window's last launches back for the next one)
                if 1 <= k < KS:
                    tw.append(time.perf_counter() - t1); tw_trib.append(t_trib)
                    if m.KWT in methods:      # wavefronts of the mainstem's last sweep launch that arrived / joined (a launch whose wavefronts start late joins with few)
                        try:
                            a_ = d_m.sweep_arrivals(); c_ = d_m.sweep_clock(1); c2_ = d_t.sweep_clock(1)
                            tw_join.append([int(a_[0]), int(a_[1]), round(c_[-1], 1) if c_ else None, round(c2_[-1], 1) if c2_ else None])
                        except Exception:
                            pass
            d_m.sync()
            t_rank0 = float(np.median(tw[len(tw) // 2:]))      # (the first windows of two fresh domains hold their regroupings and table builds)
            times["rank0_side_by_side"] = dict(s_per_window=tw, s_until_the_tributary_window_is_done=tw_trib, mainstem_sweep_wavefronts_arrived_joined=tw_join, sweep_share_mainstem=share, sweep_priority_mainstem=1, mainstem_queued_from_its_own_host_thread=two_threads,
                                               what="tributary window k and mainstem window k-1 of rank 0 queued together; median of the later half of the windows")
            d_t.close(); d_m.close()
        recs.clear()
        torch.cuda.empty_cache()
        crit = max(max(trib.values()), t_rank0)
        model = {"what": "one domain per GPU as measured here; window time = max(slowest tributary rank, rank 0 with its tributary and the mainstem side by side); "
                         "the boundary records travel behind the next window's sweep; medians over the timed windows",
                 "s_per_window": crit, "value": float(net.N) * W * len(methods) / crit,
                 "slowest_tributary_s": max(trib.values()), "rank0_tributary_plus_mainstem_one_after_the_other_s": trib.get("trib0", 0.0) + t_main,
                 "rank0_side_by_side_s": t_rank0, "balance": min(trib.values()) / max(trib.values())}
        return dict(one_gpu_s=one_gpu, value=float(net.N) * W * len(methods) / one_gpu, domains=times, model_8gpu=model, window_steps=W, windows_timed=K - 1)

    def roofline(self, W, p=1):
        """HBM roofline of the configuration's dominant kernel on tributary domain p (a full per-GPU share of the network):
        HIP events around the kernel's launches on one window (after two untimed ones), algorithmic bytes from the particle
        counters of one more window (KWT) or from SURVEY.md 8(d)'s byte model (Eulerian methods)"""
        torch, m, cfg = self.torch, self.m, self.cfg
        sp = self.P.trib[p]
        DOM = cfg.get("dominant", m.KWT)
        dom = self.make(sp, W, export_reaches=sp.export_local)
        ros = [self.forcing(W, k * W, sp.hru_global, shared=False) for k in range(2)]
        torch.cuda.synchronize()      # (torch made them on ITS stream; the library routes on its own)
        def win(k):
            if dom.lakes is not None:
                dom.set_lake_forcing(0, W)
            dom.run_device(W, k * W * DT, ros[k % 2].data_ptr()); dom.sync()
        win(0); win(1)
        clock_ms = None
        if self.methods == [m.KWT]:
            # the sweep times itself (mzr_get_sweep_clock: first wavefront in -> last wavefront out on the device's clock): four windows
            dom.sweep_clock(reset=True)
            for k in range(2, 6):
                win(k)
            clock_ms = [x for x in dom.sweep_clock(4) if x > 0]
            pt = {"launches": len(clock_ms), "kernel_ms": float(sum(clock_ms)), "reach_steps": float(sp.net.N) * W * len(clock_ms)}
            dom.set_profiling(2); dom.kwt_traffic(reset=True)
            win(6)
            dom.set_profiling(0)
            tr = dom.kwt_traffic(reset=True)
            per_rs = kwt_bytes(tr) / max(1, tr["n_route"] + tr["n_head"])
            kernel = "k_sweep_kwt"
        else:
            dom.timing(DOM, reset=True); dom.set_profiling(1)
            win(2)
            dom.set_profiling(0)
            pt = dom.timing(DOM, reset=True)
            U = float(sp.net.upIndex.size) / sp.net.N
            per_rs = float(cfg["bytes"](U))
            kernel = f"k_stage<{DOM}>"
        launches = max(1, pt["launches"])
        achieved = per_rs * pt["reach_steps"] / (pt["kernel_ms"] * 1e-3) / 1e9 if pt["kernel_ms"] > 0 else 0.0
        sw = dom.sweep_info() if m.KWT in self.methods else None
        dom.close(); del dom, ros
        torch.cuda.empty_cache()
        return {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None, "traffic_source": None, "algorithmic_bytes_per_launch": per_rs * pt["reach_steps"] / launches,
                "bytes_per_reach_step": per_rs, "avg_launch_us": pt["kernel_ms"] / launches * 1e3, "launches": launches,
                "launch_us": [x * 1e3 for x in clock_ms] if clock_ms else None,
                "timed_by": "device clock of the sweep itself (mzr_get_sweep_clock)" if clock_ms else "HIP events around every stage launch",
                "domain": f"trib{p}: {sp.n_real} reaches, window of {W} steps", "kwt_sweep": sw}

    def cpu(self, n_spin, n_smp):
        """the reference's own solvers on the host cores, a bounded sample of the SAME full-size network"""
        net, methods = self.net, self.methods
        try:
            ro_cpu = device_runoff(self.torch, net.H, n_spin + n_smp, 0, 7, self.dev).cpu().numpy()
            lk = None
            if self.lakes is not None:
                lk = dict(self.lakes, evap=np.zeros((n_spin + n_smp, net.H)), precip=np.zeros((n_spin + n_smp, net.H)))
                lk["ymd"] = self.lakes["ymd"][:n_spin + n_smp]
            cpu = cpu_baseline(net, self.frac, ro_cpu, n_spin, n_smp, methods, (self.uh_off, self.uhv) if self.need_uh else None, lk, one_thread=False)
            cpu["sample"] = (f"the FULL {net.N}-reach network, route_opt {self.cfg['methods']}, {n_smp} steps timed after {n_spin}, "
                             f"{cpu['cores']} OpenMP threads over the reference's stream-order branches; unmodified reference solvers, flang -O2")
            # the MPI shape: the eight tributary domains as eight processes side by side (threads so that the host's cores are used)
            try:
                from mizuroute_amd.partition import lakes_for_domain
                thr = max(1, min(16, (os.cpu_count() or 8) // 8))

                def lk_of(dm):
                    if self.lakes is None:
                        return None
                    l2 = lakes_for_domain(self.lakes, dm, net.N)
                    if l2 is None:
                        return None
                    n_ = n_spin + n_smp
                    return dict(l2, evap=np.zeros((n_, max(1, dm.hru_global.size))), precip=np.zeros((n_, max(1, dm.hru_global.size))), ymd=self.lakes["ymd"][:n_])

                def uh_pair(dm):
                    u = self.uh_of(dm)
                    return (u["uh_offset"], u["uh"]) if u else (np.arange(dm.net.N + 1, dtype=np.int32), np.ones(dm.net.N))

                cpu["mpi_like"] = cpu_mpi_like(self.P.trib, self.frac, lambda dm: ro_cpu[:, dm.hru_global], methods,
                                               uh_of=uh_pair, lakes_of=lk_of if self.lakes is not None else None, threads=thr, spin=n_spin, smp=n_smp,
                                               also_one_thread=n_smp >= 32)
            except Exception as e:
                cpu["mpi_like"] = {"value": None, "sample": f"failed: {type(e).__name__}: {e}"}
            try:      # one single-thread rank per physical core over the P-way decomposition of the FULL network, 48 + 48 steps
                # (in the default line's `configs` objects for the KWT network only -- c3, the north-star configuration: 60-90 s per
                # configuration, and the default bench has 900 s; `--loopback` reports it for every configuration)
                if n_smp < 32 and methods != [self.m.KWT]:
                    raise StopIteration("left out of the default line's time budget: bench.py --loopback --config " + self.config)
                n2 = 96
                ro2 = ro_cpu if ro_cpu.shape[0] >= n2 else device_runoff(self.torch, net.H, n2, 0, 7, self.dev).cpu().numpy()

                def lk2(dm):
                    if self.lakes is None:
                        return None
                    l2 = lakes_for_domain(self.lakes, dm, net.N)
                    if l2 is None:
                        return None
                    return dict(l2, evap=np.zeros((n2, max(1, dm.hru_global.size))), precip=np.zeros((n2, max(1, dm.hru_global.size))), ymd=self.lakes["ymd"][:n2])

                cpu["mpi_like_cores"] = cpu_mpi_like_cores(net, self.frac, lambda dm, n: ro2[:n, dm.hru_global], methods, uh_of=uh_pair,
                                                           lakes_of=lk2 if self.lakes is not None else None)
                del ro2
            except StopIteration:      # (not measured in this run: no key rather than a null)
                pass
            except Exception as e:
                cpu["mpi_like_cores"] = {"value": None, "sample": f"failed: {type(e).__name__}: {e}"}
        except Exception as e:
            cpu = {"value": None, "unit": "reaches*timesteps/s", "cores": 0, "kind": "reference", "sample": f"failed: {e}"}
        return cpu


def main_sweep_share(ms, methods, m):
    """share of the device's wavefront slots the mainstem domain's persistent sweep gets beside the tributary sweep of the same
    rank.  The mainstem is a hundredth of the rank's work but one long chain of dependent passes (its stages + W levels): it needs
    about as many wavefronts as it has items per level (every reach is active in every level of the skewed schedule, ~7 reaches
    per item) and, above all, its passes must not queue behind the tributary's on the SIMDs (sweep_priority = 1)."""
    from mizuroute_amd.partition import MAIN_SWEEP_SHARE
    return MAIN_SWEEP_SHARE


def loopback_bench(args, torch, m, uhmod):
    """`bench.py --loopback --config c3|c4|c5 --partitions 8`: parity and timing of the full-size network (class Loopback)"""
    lb = Loopback(torch, m, uhmod, args.config, args.partitions or 8, args.reaches, args.window, args.balance)
    net, cfg, methods, nparts, P = lb.net, lb.cfg, lb.methods, lb.nparts, lb.P
    out = {"metric": "reaches*timesteps/s", "unit": "reaches*timesteps/s", "n_gpus": 1, "higher_is_better": True, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"FULL {args.config} network ({net.N} reaches, route_opt {cfg['methods']}{', ' + str(lb.lakes['reach'].size) + ' lakes' if lb.lakes is not None else ''}) in {nparts} sub-basin partitions (reference decomposition), "
                                  "all on one GPU, boundary records through device memory", "baseline_config": args.config, "reaches_total": net.N,
                      "partitions": nparts, "mainstem_reaches": int(P.is_mainstem.sum()), "setup_s": lb.t_setup,
                      "assignment": "reference (assign_node)" if lb.mc == 0.0 else f"rank 0's tributary share cut by the mainstem's cost of {lb.mc:.0f} reaches"}}
    rep, whole_info, _ = lb.parity(256, 2)
    same = rep["partitioned_equals_whole_bit_for_bit"]
    out["config"]["whole_network"] = whole_info
    out["parity"] = rep
    W, K = args.window or min(cfg["window"], 4096), max(2, args.steps)      # (all partitions and their records on ONE GPU)
    tm = lb.timing(W, K)
    out.update({"value": tm["value"], "steps": K, "warmup": 1, "ms_per_step": tm["one_gpu_s"] * 1e3, "scaling": "strong",
                "vs_baseline": None, "error": None if same else "partitioned run differs from the whole network"})
    out["config"].update({"window_steps": W, "domains": tm["domains"]})
    out["cpu_baseline"] = None if args.no_cpu_baseline else lb.cpu(args.cpu_spinup, args.cpu_sample)
    out["model_8gpu"] = tm["model_8gpu"]
    print(json.dumps(out))


def main():
    # forcing windows of many different sizes come and go (one per domain of a full-size configuration): without expandable
    # segments the caching allocator keeps a 12 GB block of every size it has seen
    os.environ.setdefault("PYTORCH_ALLOC_CONF", "expandable_segments:True")
    os.environ.setdefault("PYTORCH_HIP_ALLOC_CONF", "expandable_segments:True")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4, help="timed batches (forcing windows)")
    ap.add_argument("--warmup", type=int, default=1, help="untimed batches")
    ap.add_argument("--window", type=int, default=0,
                    help="model time steps per batch; 0 = what --config says (16384 for c2), fewer when --steps is large (about 2M model steps in total)")
    ap.add_argument("--reaches", type=int, default=0, help="reaches per GPU (default: what --config says)")
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="BASELINE.json configuration.  --gpus 1: c2 (default, the metric's headline: 100 k reaches KWT) or the per-GPU shard of c3 / c4 / c5.  "
                         "--gpus N > 1: c3 (default: ONE network of N x 375 k reaches -- the ~3 M-reach north-star network at N = 8 -- cut into N sub-basin "
                         "partitions by the reference's decomposition, windows of 8192) or c2 (N x 100 k reaches, windows of 16 384)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dump", default="", help="write <DUMP>.rank<r>.npz with the per-reach interval mean of REACH_Q and the particle "
                    "counts of the reaches this rank routes (tests: a partitioned run must equal the one-rank run); forcing is then "
                    "generated for the whole network and sliced per domain, so that every partitioning routes the same thing")
    ap.add_argument("--no-h2d", action="store_true", help="skip the host-forcing leg (value_with_h2d)")
    ap.add_argument("--no-single-step", action="store_true", help="skip the mzr_step leg (single_step)")
    ap.add_argument("--partitions", type=int, default=0,
                    help="with --loopback: route the FULL-SIZE network of --config (c3 ~3 M reaches KWT, c4 ~5 M IRF + MC, c5 ~3 M DW) cut into "
                         "this many sub-basin partitions by the reference's decomposition, all of them on this one GPU")
    ap.add_argument("--loopback", action="store_true", help="see --partitions: boundary records go through device memory instead of RCCL")
    ap.add_argument("--balance", action="store_true", help="with --loopback: cut rank 0's share of small tributaries by what the mainstem costs it "
                    "(partition.mainstem_cost; the reference's assignment gives rank 0 an even share plus the mainstem).  With --gpus N > 1 "
                    "this is the default (same domains, same results, rank 0 level with the others)")
    ap.add_argument("--reference-assignment", action="store_true", help="with --gpus N > 1: the reference's assign_node as it is (the default since round 5; "
                    "--balance cuts rank 0's share of small tributaries by the mainstem's cost instead)")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` objects of the default line (full-size c3 / c4 / c5 in 8 partitions on this GPU)")
    ap.add_argument("--configs", default="c3,c4,c5", help="which full-size configurations the default line carries")
    ap.add_argument("--cpu-spinup-configs", type=int, default=16, help="untimed steps of the CPU baseline of a `configs` object (full network: about a second per step)")
    ap.add_argument("--cpu-sample-configs", type=int, default=24, help="timed steps of the CPU baseline of a `configs` object")
    ap.add_argument("--cpu-spinup", type=int, default=48, help="with --loopback: untimed steps of the CPU baseline on the full network")
    ap.add_argument("--cpu-sample", type=int, default=48, help="with --loopback: timed steps of the CPU baseline on the full network")
    ap.add_argument("--event-roofline", action="store_true", help="also time K more windows with HIP events on the sweep launches (the round-4 measurement; "
                    "roofline.avg_launch_us comes from the sweep's own device clock inside the K timed windows either way)")
    ap.add_argument("--no-roofline", action="store_true",
                    help="skip the event-timed and the traffic-counter windows (used under rocprofv3)")
    args = ap.parse_args()

    import torch
    import mizuroute_amd as m
    from mizuroute_amd import uh as uhmod

    if args.loopback:
        return loopback_bench(args, torch, m, uhmod)

    # `--gpus N` is the rank count.  Launched through torch.distributed.run (WORLD_SIZE set) it must agree with it; launched PLAIN
    # (`python bench.py --gpus N`, no WORLD_SIZE) the process launches the N ranks itself, one per visible GPU, and passes their exit
    # code on (mpi_process.f90:1088-1342 is what the N ranks replace).  Fewer than N devices is an error unless
    # MZR_BENCH_SINGLE_DEVICE=1 (all ranks on cuda:0, records over gloo: the test of the N-rank protocol on a one-GPU box).
    single_dev = bool(os.environ.get("MZR_BENCH_SINGLE_DEVICE"))
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        ndev = torch.cuda.device_count()
        if ndev < args.gpus and not single_dev:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {ndev} GPU(s) visible; one rank per GPU is the rule "
                             "(MZR_BENCH_SINGLE_DEVICE=1 runs all ranks on cuda:0 over gloo, a protocol test and not a measurement)")
        import socket
        import subprocess
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if single_dev:
            env.setdefault("MZR_BENCH_BACKEND", "gloo")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        raise SystemExit(subprocess.call(cmd, env=env))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} launched with WORLD_SIZE={world}: the two must agree")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if single_dev:      # several ranks share cuda:0 (gloo transport)
        local_rank = 0
    elif world > 1 and torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible)")
    dist = None
    backend = "nccl"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MZR_BENCH_BACKEND", "gloo" if single_dev else "nccl")   # "gloo" only on a one-GPU box
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        world = dist.get_world_size()                   # what the communicator reports is what the line says
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    # ---- workload.  N = 1: BASELINE.json configs[1], one ~100k-reach sub-basin.  N > 1 (weak
    # scaling): ONE network of N x 100k reaches partitioned by the reference's rule (mainstem =
    # reaches with more than total/N reaches upstream, tributary sub-basins dealt largest-first;
    # mizuroute_amd/partition.py); tributary partitions ship their outlet reaches' boundary records to
    # the mainstem owner (rank 0) once per window over RCCL point-to-point.
    frac = uhmod.basin_uh(DT, 2.5, 86400.0)
    K, KW = max(1, args.steps), max(0, args.warmup)
    # N = 1: BASELINE.json configs[1] (c2).  N > 1: configs[2] (c3), the north-star network -- N x 375 k reaches, i.e. the ~3 M-reach
    # CONUS-scale network at N = 8 -- partitioned as the reference partitions it (domain_decomposition.f90:41-163)
    if args.config is None:
        args.config = "c2" if world == 1 else "c3"
    cfg = CONFIGS[args.config]
    if world > 1 and args.config not in ("c2", "c3"):
        raise SystemExit("--gpus N > 1 routes a KWT network (c3, or c2's reaches per GPU); the c4 / c5 shards run with --gpus 1, their full networks with --loopback")
    methods = [int(c) for c in cfg["methods"]]
    kwt_run = methods == [m.KWT]
    DOM = cfg.get("dominant", m.KWT)          # the method the roofline object describes
    W = args.window
    if W <= 0:   # keep the whole run at about two million model time steps whatever K the caller asks for
        W = cfg["window"]
        while W > 128 and W * (K + KW) > (1 << 21):
            W //= 2
    n_reach = args.reaches or cfg["reaches"]
    net = m.make_network(n_reach * world, seed=20240529, floodplain=bool(cfg.get("floodplain")))
    workload = cfg["workload"]
    if world > 1:
        workload = (f"ONE synthetic HDMA-CONUS-like network of {net.N} reaches ({n_reach} per GPU"
                    + ("; BASELINE.json configs[2]: the ~3 M-reach network at 8 GPUs" if args.config == "c3" else "; BASELINE.json configs[1]'s reaches per GPU")
                    + f"), KWT (route_opt 2), dt 3600 s, hillslope UH on, cut into {world} sub-basin partitions by the reference's domain decomposition "
                      "(domain_decomposition.f90:41-163), one partition per GPU, mainstem on rank 0, one boundary-record message per partition and window over RCCL p2p")
    router = None
    lakes = None
    if world == 1:
        extra = {}
        if not kwt_run:
            uh_off, uhv = uhmod.make_uh(net.params["RLENGTH"], DT, 1.5, 5000.0)
            extra = dict(uh_offset=uh_off, uh=uhv)
        if cfg.get("lakes"):
            from mizuroute_amd.synthetic import make_lakes
            lakes = make_lakes(net, W, DT, seed=9, frac=cfg["lakes"], input_option=1)     # one window of lake forcing, used for every window
            extra["lakes"] = lakes
        dom = m.RoutingDomain(net, DT, methods, frac_future=frac, max_window=W, device=local_rank, **extra)
        doms = [(dom, net.H)]
    else:
        from mizuroute_amd.partition import PartitionedRouter, partition_network
        from mizuroute_amd.partition import mainstem_cost
        P = partition_network(net, world, build_for=[rank], main_cost=mainstem_cost(net, world, W) if (args.balance and not args.reference_assignment) else 0.0)

        lib_comm = None
        if os.environ.get("MZR_BENCH_TRANSPORT") == "lib" and backend == "nccl":     # the library's own RCCL transport (mzr_comm_*)
            uid = [m.api.Comm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            lib_comm = m.api.Comm(rank, world, uid[0], device=local_rank)

        class LibTransport:
            """records travel through mzr_comm_send / mzr_comm_recv_many; `router` is filled in once it exists"""
            router, keep = None, None

            def send(self, t, dst):
                lib_comm.sync()                      # the buffer of the send before this one may go
                self.keep = t
                lib_comm.send(self.router.trib, t.data_ptr(), t.numel(), dst)

            def recv(self, t, src):
                lib_comm.recv(self.router.main, t.data_ptr(), t.numel(), src)

            def recv_many(self, pairs):
                lib_comm.recv_many(self.router.main, [(t.data_ptr(), t.numel(), src) for t, src in pairs])

        class Transport:
            def send(self, t, dst):
                dist.send(t if backend == "nccl" else t.cpu(), dst)

            def recv(self, t, src):
                if backend == "nccl":
                    dist.recv(t, src)
                else:                                   # gloo moves host tensors
                    h = torch.empty(t.shape, dtype=t.dtype)
                    dist.recv(h, src)
                    t.copy_(h)
                torch.cuda.current_stream().synchronize()

            def recv_many(self, pairs):
                """All boundary records of a window at once (one xGMI link per peer, transfers side by side)."""
                if backend != "nccl":
                    for t, src in pairs:
                        self.recv(t, src)
                    return
                works = dist.batch_isend_irecv([dist.P2POp(dist.irecv, t, src) for t, src in pairs])
                for wk in works:
                    wk.wait()
                torch.cuda.current_stream().synchronize()

        def make(spec, **kw):
            return m.RoutingDomain(spec.net, DT, [m.KWT], frac_future=frac, max_window=W, device=local_rank, **kw)

        transport = LibTransport() if lib_comm is not None else Transport()
        router = PartitionedRouter(P, rank, make, transport,
                                   lambda n: torch.empty(n, dtype=torch.float64, device=dev), W)
        transport.router = router
        dom = router.trib if router.trib is not None else router.main
        doms = [(d, sp.net.H) for d, sp in ((router.trib, router.trib_spec), (router.main, router.main_spec)) if d is not None]
    n_stages, max_width = dom.schedule()

    def gen(n_steps, t0):
        if args.dump:      # one forcing for the whole network, every domain takes the columns of its HRUs
            glob = device_runoff(torch, net.H, n_steps, t0, 7, dev)
            if router is None:
                return [glob]
            specs = [sp for d_, sp in ((router.trib, router.trib_spec), (router.main, router.main_spec)) if d_ is not None]
            return [glob[:, torch.as_tensor(sp.hru_global, device=dev, dtype=torch.long)].contiguous() for sp in specs]
        return [device_runoff(torch, H, n_steps, t0, 7 + rank + 101 * i, dev) for i, (_, H) in enumerate(doms)]

    pool = [gen(W, 0), gen(W, W)]          # two forcing windows, used alternately
    state = {"batch": 0, "t": 0.0}          # batches routed so far, simulation clock [s]

    def run_batches(nb):
        for _ in range(nb):
            k = state["batch"]
            ros = pool[k % 2]
            t_start = state["t"]
            if router is None:
                if lakes is not None:
                    dom.set_lake_forcing(0, W)
                dom.run_device(W, t_start, ros[0].data_ptr())
            else:
                pt = ros[0].data_ptr() if router.trib is not None else 0
                pm = ros[-1].data_ptr() if router.main is not None else 0
                router.run_window(W, t_start, pt, pm, keep=ros)
            state["batch"] = k + 1
            state["t"] = t_start + W * DT

    def sync_all():
        if router is not None:
            router.sync()            # flushes the window whose boundary exchange is still in flight
            return
        for d, _ in doms:
            d.sync()

    # The line below is printed whatever happens: a failure of the timed region (e.g. ierr 93, the sweep's watchdog) ends up
    # in its "error" field with value null, and the process exits with 1.
    error = None
    elapsed, value, tm = None, None, None
    events_in_timed = False
    events_behind = False
    try:
        torch.cuda.synchronize()
        run_batches(KW)
        sync_all()
        dom.timing(DOM, reset=True)
        # KWT on one GPU: the sweep is ONE launch per window, so the HIP events around it (two per window, on the library's
        # stream) ride in the timed region itself: roofline.achieved is the average over exactly the K timed launches
        # ... but NOT while `value` is timed: timing-enabled events on the sweep's stream slow some windows by up to a quarter (447 ->
        # 510-600 ms, in a pattern with a period of eight windows whose strength differs from process to process; six processes
        # without events: 3.28-3.30 x 10^9 every one; with events, as markers or attached to the dispatch: 2.77-3.30;
        # profiles/r04_experiments.md).  So the K timed windows run without them and K more windows right behind carry the events.
        events_in_timed = False
        # Round 5: the sweep times itself -- its first wavefront in and its last wavefront out leave the device's 100 MHz clock in two
        # words per launch (mzr_get_sweep_clock) -- so roofline.avg_launch_us is measured INSIDE the K timed windows with no host event
        # anywhere near the stream.  --event-roofline keeps the old leg (K more windows with HIP events) beside it.
        events_behind = world == 1 and kwt_run and not args.no_roofline and args.event_roofline
        if kwt_run:
            dom.sweep_clock(reset=True)

        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_batches(K)
        sync_all()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        clock_ms = [x for x in dom.sweep_clock(K)] if kwt_run else None      # the K timed launches of this rank's (tributary) domain
        tm = dom.timing(DOM, reset=True)
        if dist is not None:
            tmax = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        total_reach_steps = float(net.N) * K * W * len(methods)      # every active method routes every reach every step
        value = total_reach_steps / elapsed
    except Exception as e:
        error = f"{type(e).__name__}: {e}"
    if error is not None:
        if rank == 0:
            print(json.dumps({"metric": "reaches*timesteps/s", "value": None, "unit": "reaches*timesteps/s", "n_gpus": world, "steps": K, "warmup": KW,
                              "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                              "config": {"workload": workload, "baseline_config": args.config, "window_steps": W}, "roofline": None,
                              "cpu_baseline": None, "error": error}))
        sys.stdout.flush()
        os._exit(1)

    if args.dump:      # per-reach results of everything routed so far, by global reach index
        parts = []
        if router is None:
            parts.append((np.arange(net.N), dom.mean_q(m.KWT), dom.kwt_state()[0]))
        else:
            for d_, sp in ((router.trib, router.trib_spec), (router.main, router.main_spec)):
                if d_ is not None:
                    parts.append((np.asarray(sp.reach_global[:sp.n_real]), d_.mean_q(m.KWT)[:sp.n_real], d_.kwt_state()[0][:sp.n_real]))
        np.savez(f"{args.dump}.rank{rank}.npz", reach=np.concatenate([p_[0] for p_ in parts]), q=np.concatenate([p_[1] for p_ in parts]),
                 nw=np.concatenate([p_[2] for p_ in parts]))

    # ---- the same timed region with the forcing handed over in page-locked host memory (N = 1): two host
    # windows, copied by the library on its own stream while the window before is routed
    value_h2d, value_h2d_f64, h2d_info = None, None, None
    if world == 1 and not args.no_h2d and lakes is None:
        # (a) the forcing as the reference's files store it -- single precision, which get_nc widens into real(dp)
        # (standalone/read_runoff.f90:264-306): mzr_run_async_f32 moves half the bytes and widens on the device; (b) double precision
        for tag in ("f32", "f64"):
            try:
                dt_ = torch.float32 if tag == "f32" else torch.float64
                hosts = [torch.empty((W, net.H), dtype=dt_).pin_memory() for _ in range(2)]
                for hb, ro in zip(hosts, pool):
                    hb.copy_(ro[0])
                torch.cuda.synchronize()
                call = dom.run_async_f32 if tag == "f32" else dom.run_async
                k0 = state["batch"]
                call(W, state["t"], hosts[k0 % 2].data_ptr())      # one untimed window fills the pipeline
                dom.sync()
                state["t"] += W * DT
                t1 = time.perf_counter()
                for k in range(k0 + 1, k0 + 1 + K):
                    call(W, state["t"], hosts[k % 2].data_ptr())
                    state["t"] += W * DT
                dom.sync()
                el = time.perf_counter() - t1
                v_ = float(net.N) * K * W * len(methods) / el
                state["batch"] = k0 + 1 + K
                nbytes = float(W) * net.H * (4 if tag == "f32" else 8)
                if tag == "f32":
                    value_h2d = v_
                    h2d_info = {"forcing": "float32 in page-locked host memory (as the forcing files store it), widened to f64 on the device behind the copy",
                                "bytes_per_window": nbytes, "host_to_device_GBps_needed": nbytes * K / el / 1e9}
                else:
                    value_h2d_f64 = v_
                    h2d_info["f64_bytes_per_window"] = nbytes
                    h2d_info["f64_GBps_achieved_or_needed"] = nbytes * K / el / 1e9
                del hosts
            except Exception as e:   # reported, never required
                if tag == "f32":
                    value_h2d = f"failed: {e}"
                else:
                    value_h2d_f64 = f"failed: {e}"

    # ---- one main_route-equivalent call per model time step (mzr_step), the reference driver's loop
    # (standalone/route_runoff.f90:80-108): (a) stepBatch = 1, every call routes its step and synchronises (the
    # coupled-model use); (b) stepBatch = 4096 on a second handle: the calls put their rows aside, the library routes
    # them as windows, results are fetched once at the end (output frequency) -- host rows, H2D and the Python call
    # overhead of every step included
    single = None
    if world == 1 and not args.no_single_step and lakes is None:
        n1 = 40
        try:
            rows = pool[0][0][:n1].cpu().numpy()
            tb = state["t"]
            dom.step(tb, tb + DT, rows[0])
            t1 = time.perf_counter()
            for k in range(1, n1):
                dom.step(tb + k * DT, tb + (k + 1) * DT, rows[k])
            el1 = time.perf_counter() - t1
            state["t"] = tb + n1 * DT
            single = {"value": float(net.N) * (n1 - 1) * len(methods) / el1, "unit": "reaches*timesteps/s", "ms_per_model_timestep": el1 / (n1 - 1) * 1e3,
                      "steps": n1 - 1, "what": "mzr_step, stepBatch 1: host forcing row -> device, one sweep over the stages, results synchronised every step"}
        except Exception as e:
            single = {"value": None, "what": f"failed: {e}"}
        try:
            nb, batch = 3 * 4096, 4096
            rows = pool[1][0][:4096].cpu().numpy()
            dom2 = m.RoutingDomain(net, DT, methods, frac_future=frac, max_window=batch, device=local_rank, step_batch=batch, **extra)
            for k in range(batch):                      # one untimed batch: buffers, tables, regrouping
                dom2.step(k * DT, (k + 1) * DT, rows[k % 4096])
            dom2.sync()
            t1 = time.perf_counter()
            for k in range(batch, batch + nb):
                dom2.step(k * DT, (k + 1) * DT, rows[k % 4096])
            q_last = dom2.flux(methods[0])               # fetching a result routes what is pending and synchronises
            el2 = time.perf_counter() - t1
            single["pipelined"] = {"value": float(net.N) * nb * len(methods) / el2, "unit": "reaches*timesteps/s", "ms_per_model_timestep": el2 / nb * 1e3,
                                   "steps": nb, "step_batch": batch, "finite": bool(np.isfinite(q_last).all()),
                                   "what": "mzr_step, stepBatch 4096: one call per model time step with its host forcing row, routed as windows of 4096, "
                                           "results fetched once at the end"}
            dom2.close()
        except Exception as e:   # reported, never required
            single["pipelined"] = f"failed: {e}"

    # ---- roofline of the dominant kernel (KWT stage sweep), measured live with HIP events around
    # every stage launch on the library's stream, on the window that follows the timed region
    roof, ktf = None, None
    post_error = None
    value_events = None
    try:
        clock_roof = kwt_run and not args.no_roofline      # KWT: the launches of the timed region timed themselves (device clock)
        pt_clock = None
        if clock_roof:
            ok = [x for x in (clock_ms or []) if x > 0]
            dom_N = net.N if router is None else (router.trib_spec.net.N if router.trib is not None else router.main_spec.net.N)
            if ok:
                pt_clock = {"kernel_ms": float(sum(ok)), "launches": len(ok), "reach_steps": float(dom_N) * W * len(ok), "min_ms": min(ok), "max_ms": max(ok)}
                ktf = pt_clock["kernel_ms"] * 1e-3 / elapsed if elapsed else None
        if args.no_roofline or clock_roof:
            pass
        elif world > 1:      # every rank takes part in the profiled window (the exchange is collective)
            if rank != 0:
                dist.barrier()
                run_batches(1)
                sync_all()
        pt_events = None
        if rank == 0 and not args.no_roofline and events_behind:      # the same K windows once more, HIP events on every sweep launch
            dom.timing(DOM, reset=True)
            dom.set_profiling(1)
            torch.cuda.synchronize()
            t_prof = time.perf_counter()
            run_batches(K)
            sync_all()
            torch.cuda.synchronize()
            t_prof = time.perf_counter() - t_prof
            dom.set_profiling(0)
            pt_events = dom.timing(DOM, reset=True)
            value_events = float(net.N) * K * W * len(methods) / t_prof
        elif rank == 0 and not args.no_roofline and not clock_roof:
            dom.timing(DOM, reset=True)
            dom.set_profiling(1)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t_prof = time.perf_counter()
            run_batches(1)
            sync_all()
            t_prof = time.perf_counter() - t_prof
            dom.set_profiling(0)
            pt = dom.timing(DOM, reset=True)
            ktf = pt["kernel_ms"] * 1e-3 / t_prof if t_prof > 0 else None
            if not kwt_run:      # Eulerian solvers: SURVEY.md 8(d) byte model of the dominant method, one lane per reach, one launch per stage
                U = float(net.upIndex.size) / net.N
                per_rs = float(cfg["bytes"](U))
                launches = max(1, pt["launches"])
                achieved = per_rs * pt["reach_steps"] / (pt["kernel_ms"] * 1e-3) / 1e9 if pt["kernel_ms"] > 0 else 0.0
                roof = {"bound": "hbm", "kernel": f"k_stage<{DOM}>", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": None, "algorithmic_bytes_per_launch": per_rs * pt["reach_steps"] / launches, "bytes_per_reach_step": per_rs,
                        "avg_launch_us": pt["kernel_ms"] / launches * 1e3, "launches": launches,
                        "note": "FP64-transcendental kernel (Newton normal depth, fifth-root powers): far from the HBM roofline by construction; "
                                "FP64 instruction counts per kernel are in profiles/*_pmc.md"}
        # particle-traffic counters (device atomics) are collected on one more window so that they do
        # not disturb the event-timed launches; bytes per reach-step of that window x the reach-steps
        # of the timed window = algorithmic bytes of the timed window
        if world > 1 and rank != 0 and not args.no_roofline:
            dist.barrier()
            run_batches(1)
            sync_all()
        if rank == 0 and not args.no_roofline and kwt_run:
            dom.set_profiling(2)
            dom.kwt_traffic(reset=True)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            run_batches(1)
            sync_all()
            dom.set_profiling(0)
            tr = dom.kwt_traffic(reset=True)
            per_rs = kwt_bytes(tr) / max(1, tr["n_route"] + tr["n_head"])
            pt = pt_clock
            if pt is None:
                raise RuntimeError("the sweep's device clock returned no launch of the timed region")
            bytes_total = per_rs * pt["reach_steps"]
            launches = max(1, pt["launches"])
            avg_ms = pt["kernel_ms"] / launches
            achieved = bytes_total / (pt["kernel_ms"] * 1e-3) / 1e9 if pt["kernel_ms"] > 0 else 0.0
            sw = dom.sweep_info()
            ev = None
            if pt_events is not None and pt_events["launches"] > 0:      # --event-roofline: the round-4 measurement beside it
                ev = {"avg_launch_us": pt_events["kernel_ms"] / pt_events["launches"] * 1e3, "launches": pt_events["launches"],
                      "min_launch_us": pt_events.get("min_ms", 0.0) * 1e3, "max_launch_us": pt_events.get("max_ms", 0.0) * 1e3, "value_with_events": value_events,
                      "what": f"{K} more windows right behind the timed ones with HIP events attached to the sweep launches (the events themselves slow some windows)"}
            roof = {"bound": "hbm", "kernel": "k_sweep_kwt" if sw[0] > 0 and os.environ.get("MZR_KWT_SWEEP", "1") != "0" else "k_stage_kwt",
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS,
                    # HBM traffic from the PMC counters needs rocprofv3 passes of its own (MI355X_MICROARCH.md): not measurable in this run.
                    # The per-round figure for this workload is in profiles/ (r05*_summary.md, kwt_hbm_traffic.json); null here by rule.
                    "traffic": None,
                    "algorithmic_bytes_per_launch": bytes_total / launches,
                    "bytes_per_reach_step": per_rs,
                    "avg_launch_us": avg_ms * 1e3, "launches": launches,
                    "min_launch_us": pt["min_ms"] * 1e3, "max_launch_us": pt["max_ms"] * 1e3,
                    "launch_us": [x * 1e3 for x in (clock_ms or [])],
                    "frac_of_shortest_launch": (bytes_total / launches / (pt["min_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if pt["min_ms"] else None,
                    "timed_over": (f"the {launches} sweep launches of the K timed windows themselves, each timed on the device's own 100 MHz clock: first wavefront in -> "
                                   "last wavefront out (mzr_get_sweep_clock); no HIP events in or near the timed region"),
                    "hip_events": ev,
                    "particles_per_routed_reach": (tr["w_in"] + tr["w_up"] + tr["w_out"]) / max(1, tr["n_route"])}


    except Exception as e:      # reported in "error"; the headline value above stands
        post_error = f"roofline legs: {type(e).__name__}: {e}"
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # reported at N=1 only
        try:
            n_spin = 240 if args.config == "c2" else 48                                     # the shards are 4-6 x larger: a bounded sample
            ro_cpu = device_runoff(torch, net.H, 2 * n_spin, 0, 7, dev).cpu().numpy()      # the first steps of the GPU leg's forcing
            cpu = cpu_baseline(net, frac, ro_cpu, n_spin, n_spin, methods, None if kwt_run else (uh_off, uhv), lakes)
            if cpu is not None and args.config == "c2":
                cpu["by_threads"] = cpu_by_threads(net, frac, ro_cpu, methods, None if kwt_run else (uh_off, uhv), lakes)
                try:      # the MPI shape on this network: the reference's decomposition for 8 ranks, one process per tributary domain
                    from mizuroute_amd.partition import partition_network
                    P8 = partition_network(net, 8)
                    thr = max(1, min(16, (os.cpu_count() or 8) // 8))
                    cpu["mpi_like"] = cpu_mpi_like(P8.trib, frac, lambda dm: ro_cpu[:96, dm.hru_global], methods, threads=thr, spin=48, smp=48)
                except Exception as e:
                    cpu["mpi_like"] = {"value": None, "sample": f"failed: {type(e).__name__}: {e}"}
                try:      # ... and at the reference's natural rank count on this host: one single-thread rank per physical core over the P-way decomposition
                    cpu["mpi_like_cores"] = cpu_mpi_like_cores(net, frac, lambda dm, n: ro_cpu[:n, dm.hru_global], methods)
                except Exception as e:
                    cpu["mpi_like_cores"] = {"value": None, "sample": f"failed: {type(e).__name__}: {e}"}
        except Exception as e:   # the baseline is reported, never required
            cpu = {"value": None, "unit": "reaches*timesteps/s", "cores": 0, "kind": "reference", "sample": f"failed: {e}"}

    # ---- BASELINE.json configs[2..4] at FULL size on this one GPU (N = 1, default configuration only): the 3 M / 5 M / 3 M + lakes
    # networks cut into eight partitions by the reference's decomposition and routed domain after domain, boundary records through
    # device memory (class Loopback); per configuration: parity against the unpartitioned network, median window times of every
    # domain over >= 4 timed windows, roofline of the dominant kernel on one tributary domain, the reference on the full network
    sweep_geom = dict(zip(("wavefronts", "device_wavefront_slots", "items_per_launch"), dom.sweep_info())) if world == 1 and m.KWT in methods else None
    sweep_arr = dict(zip(("arrived_last", "joined_last", "start_delay_hist_log2_10ns"), dom.sweep_arrivals())) if world == 1 and m.KWT in methods else None
    sweep_retries = dom.sweep_retries() if m.KWT in methods else None      # windows routed again after an ierr 93 (none in a healthy run)
    configs = None
    if rank == 0 and world == 1 and args.config == "c2" and not args.no_configs:
        configs = {}
        try:
            dom.close()
        except Exception:
            pass
        del pool
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        for cname in args.configs.split(","):
            t_c = time.perf_counter()
            try:
                lb = Loopback(torch, m, uhmod, cname, 8)
                rep, whole_info, _ = lb.parity(128, 1)
                # (c3: the KWT records of all eight partitions -- particle rows of 4 400 outlets -- stay on this one GPU: windows of 4 096.  c4 / c5: the
                # configuration's own window; their records are the outlets' discharge alone since round 6, 0.1-0.2 GB per window, so c4's windows of
                # 3 072 -- longer than its tributary domains are deep: they overlap -- fit beside rank 0's two domains)
                Wc = min(CONFIGS[cname]["window"], 4096)
                tmc = lb.timing(Wc, 5)
                roofc = lb.roofline(Wc)
                cpuc = None if args.no_cpu_baseline else lb.cpu(args.cpu_spinup_configs, args.cpu_sample_configs)
                configs[cname] = {"workload": f"FULL {cname} network: {lb.net.N} reaches, route_opt {lb.cfg['methods']}"
                                              + (f", {lb.lakes['reach'].size} lakes / reservoirs" if lb.lakes is not None else "")
                                              + ", 8 sub-basin partitions (reference decomposition) one after the other on this GPU",
                                  "value": tmc["value"], "unit": "reaches*timesteps/s", "window_steps": Wc, "windows_timed": tmc["windows_timed"],
                                  "s_per_window_all_domains": tmc["one_gpu_s"], "parity": rep, "whole_network": whole_info,
                                  "domains": {k: {kk: vv for kk, vv in v.items() if kk != "what"} for k, v in tmc["domains"].items()},
                                  "model_8gpu": tmc["model_8gpu"], "roofline": roofc, "cpu_baseline": cpuc,
                                  "setup_s": lb.t_setup, "wall_s": None}
                del lb
            except Exception as e:
                configs[cname] = {"value": None, "error": f"{type(e).__name__}: {e}"}
            gc.collect()
            torch.cuda.empty_cache()
            configs[cname]["wall_s"] = time.perf_counter() - t_c

    if rank == 0:
        out = {
            "metric": "reaches*timesteps/s", "value": value, "unit": "reaches*timesteps/s",
            "n_gpus": world, "rccl_ranks": (world if (dist is not None and backend == "nccl") else 0), "backend": (backend if dist is not None else None),
            "steps": K, "warmup": KW,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "baseline_config": args.config, "route_opt": cfg["methods"],
                       "reaches_per_gpu": net.N // world, "reaches_total": net.N, "stages": n_stages,
                       "max_stage_width": max_width,
                       "step": "one forcing window (batch) of window_steps model time steps",
                       "window_steps": W, "model_timesteps_timed": K * W, "ms_per_model_timestep": elapsed / (K * W) * 1e3,
                       "simulated_years_per_wallclock_day": (K * W * DT / 31536000.0) / (elapsed / 86400.0),
                       "kernel_time_fraction": ktf,
                       "kwt_sweep": sweep_geom,
                       "parallelism": ("1 domain" if world == 1 else
                                       f"{world} sub-basin partitions (the reference's domains; " + ("its node assignment" if (args.reference_assignment or not args.balance) else
                                       "rank 0's share of small tributaries cut by the mainstem's cost") + "), mainstem on rank 0, "
                                       "one boundary-record message per partition per window over RCCL p2p")},
            "value_resident": value, "value_with_h2d": value_h2d, "value_with_h2d_f64": value_h2d_f64, "h2d": h2d_info, "single_step": single,
            "kwt_sweep_arrivals": sweep_arr, "kwt_sweep_retries": sweep_retries,
            "roofline": roof, "cpu_baseline": cpu, "configs": configs, "error": post_error,
        }
        emit(out)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
