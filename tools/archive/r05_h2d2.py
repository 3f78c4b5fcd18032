"""round 5: the host-forcing leg -- where do the 10 % go?  per window: the sweep's duration on the device clock, host time of the calls"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mizuroute_amd as m
from mizuroute_amd import uh as uhmod
sys.argv = sys.argv[:1]
import bench
N, W, K = 100000, 16384, 6
net = m.make_network(N, seed=20240529)
frac = uhmod.basin_uh(3600.0, 2.5, 86400.0)
dev = torch.device("cuda", 0)
dom = m.RoutingDomain(net, 3600.0, [m.KWT], frac_future=frac, max_window=W)
pool = [bench.device_runoff(torch, net.H, W, k * W, 7, dev) for k in range(2)]
torch.cuda.synchronize()
t = 0.0
for k in range(3):
    dom.run_device(W, t, pool[k % 2].data_ptr()); t += W * 3600.0
dom.sync()
for mode in ("dev", "f32", "f64", "f32-syncEach"):
    hosts = None
    if mode != "dev":
        dt_ = torch.float32 if mode.startswith("f32") else torch.float64
        hosts = [torch.empty((W, net.H), dtype=dt_).pin_memory() for _ in range(2)]
        for hb, ro in zip(hosts, pool):
            hb.copy_(ro)
        torch.cuda.synchronize()
    call = {"dev": None, "f32": dom.run_async_f32, "f64": dom.run_async, "f32-syncEach": dom.run_async_f32}[mode]
    # one untimed window
    if call: call(W, t, hosts[0].data_ptr())
    else: dom.run_device(W, t, pool[0].data_ptr())
    t += W * 3600.0; dom.sync(); dom.sweep_clock(reset=True)
    h0 = dom.sweep_arrivals()[2]
    t0 = time.perf_counter(); tc = []
    for k in range(1, K + 1):
        a = time.perf_counter()
        if call: call(W, t, hosts[k % 2].data_ptr())
        else: dom.run_device(W, t, pool[k % 2].data_ptr())
        tc.append((time.perf_counter() - a) * 1e3)
        t += W * 3600.0
        if mode.endswith("syncEach"): dom.sync()
    dom.sync()
    el = time.perf_counter() - t0
    print(f"{mode:14s} {N * K * W / el:.4g} reach-steps/s  {el / K * 1e3:.1f} ms per window; sweep on the device clock: " + " ".join(f"{x:.0f}" for x in dom.sweep_clock(K)) + "; host ms per call: " + " ".join(f"{x:.0f}" for x in tc))
    a, j, h1 = dom.sweep_arrivals()
    dh = [y - x for x, y in zip(h0, h1)]
    print("    last launch arrived / joined:", a, j, " start delays of the mode's launches (bucket k: < 2^k x 10 ns):", {k: v for k, v in enumerate(dh) if v})
    del hosts
