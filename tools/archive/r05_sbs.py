"""round 5 (inside gpurun): rank 0 of a full-size Eulerian configuration, tributary domain and mainstem side by side, with the host time of every call
   python tools/r05_sbs.py c4 3072"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mizuroute_amd as m
from mizuroute_amd import uh as uhmod
import bench
from mizuroute_amd.partition import lakes_for_domain
cfgname, W = sys.argv[1], int(sys.argv[2])
lb = bench.Loopback(torch, m, uhmod, cfgname, 8, 0, W)
P, net, methods = lb.P, lb.net, lb.methods
sp, ms = P.trib[0], P.main
DT = bench.DT
mk = lambda spec, **kw: m.RoutingDomain(spec.net, DT, methods, frac_future=lb.frac, max_window=W, device=0, lakes=lakes_for_domain(lb.lakes, spec, net.N) if lb.lakes is not None else None, **lb.uh_of(spec), **kw)
def create():
    share = bench.main_sweep_share(ms, methods, m)
    d_t = mk(sp, export_reaches=sp.export_local, sweep_share=1.0 - share)
    d_m = mk(ms, halo_reaches=ms.halo_local, halo_good=ms.halo_good, sweep_share=share, sweep_priority=1)
    print("stages trib0", d_t.schedule(), "main", d_m.schedule(), "reaches", sp.n_real, ms.n_real, flush=True)
    ro_t = [lb.forcing(W, k * W, sp.hru_global, shared=False) for k in range(2)]
    ro_m = [lb.forcing(W, k * W, ms.hru_global, shared=False) for k in range(2)]
    rec0 = [torch.empty(d_t.boundary_size(W, sp.export_local.size), dtype=torch.float64, device=lb.dev) for _ in range(2)]
    return d_t, d_m, ro_t, ro_m, rec0
if not os.environ.get('LATE_CREATE'):
    d_t, d_m, ro_t, ro_m, rec0 = create()
# records of the other partitions: zeros with the right header are not accepted; reuse rank 0's own record layout is per partition -> route the others once
recs = {}
hoard = []
for p in range(1, 8):
    s2 = P.trib[p]
    base, n = ms.halo_base[p]
    if not n: continue
    d = mk(s2, export_reaches=s2.export_local)
    ro = lb.forcing(W, 0, s2.hru_global, shared=False); torch.cuda.synchronize()
    d.run_device(W, 0.0, ro.data_ptr()); d.sync()
    r = torch.empty(d.boundary_size(W, s2.export_local.size), dtype=torch.float64, device=lb.dev)
    d.export_boundary(r.data_ptr()); d.sync(); recs[p] = r
    if os.environ.get('KEEP_RECS'): hoard.extend(torch.empty_like(r) for _ in range(6))
    d.close(); del d, ro; torch.cuda.empty_cache()
if os.environ.get('LATE_CREATE'):
    d_t, d_m, ro_t, ro_m, rec0 = create()
def stamp(lbl, t0, out): out.append((lbl, round(time.perf_counter() - t0, 3)))
for mode in sys.argv[3:] or ["plain"]:
    for k in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter(); o = []
        th = None
        if mode == "thread" and k >= 1:      # the mainstem's calls from a host thread of their own (ctypes releases the GIL): a launch blocks while its stream's queue is full
            import threading
            def main_calls(k=k):
                for p in range(8):
                    base, n = ms.halo_base[p]
                    if n: d_m.import_boundary(W, (rec0[(k - 1) % 2] if p == 0 else recs[p]).data_ptr(), n, base)
                d_m.run_device(W, (k - 1) * W * DT, ro_m[(k - 1) % 2].data_ptr())
            th = threading.Thread(target=main_calls); th.start()
        d_t.run_device(W, k * W * DT, ro_t[k % 2].data_ptr()); stamp("t.run", t0, o)
        if th is not None: th.join(); stamp("m.join", t0, o)
        if mode == "tsync_first": d_t.sync(); stamp("t.sync0", t0, o)
        if k >= 1 and th is None:
            for p in range(8):
                base, n = ms.halo_base[p]
                if n: d_m.import_boundary(W, (rec0[(k - 1) % 2] if p == 0 else recs[p]).data_ptr(), n, base)
            stamp("m.import", t0, o)
            if mode == "main_sweep": os.environ["MZR_ROUTE_SWEEP"] = "1"
            d_m.run_device(W, (k - 1) * W * DT, ro_m[(k - 1) % 2].data_ptr()); stamp("m.run", t0, o)
            os.environ.pop("MZR_ROUTE_SWEEP", None)
        d_t.sync(); stamp("t.sync", t0, o)
        d_t.export_boundary(rec0[k % 2].data_ptr()); d_t.sync(); stamp("t.export", t0, o)
        d_m.sync(); stamp("m.sync", t0, o)
        print(mode, k, o, flush=True)
    print(mode, "mainstem mean_q checksum", [float(np.sum(d_m.mean_q(mm))) for mm in methods], [float(np.max(d_m.mean_q(mm))) for mm in methods], flush=True)
