"""Throughput of every routing method on the benchmark network (MI355X): reach-steps/s per method,
one method per domain, windows of W steps, forcing resident on the device."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mizuroute_amd as m
from mizuroute_amd import uh as uhmod
sys.argv = sys.argv[:1]
import bench

N, W, NWIN = int(os.environ.get('NR', '100000')), int(os.environ.get('WW', '1024')), 3
net = m.make_network(N, seed=20240529)
frac = uhmod.basin_uh(3600.0, 2.5, 86400.0)
uh_off, uh = uhmod.make_uh(net.params["RLENGTH"], 3600.0, 1.5, 5000.0)
dev = torch.device("cuda", 0)
out = {}
sel = os.environ.get("METHODS", "")
for name, meth in (("SUM", [m.SUM]), ("IRF", [m.IRF]), ("KWT", [m.KWT]), ("KW", [m.KW]), ("MC", [m.MC]), ("DW", [m.DW]),
                   ("SUM+IRF+KWT+KW+DW in one domain (one stream per method)", [m.SUM, m.IRF, m.KWT, m.KW, m.DW])):
    if sel and name not in sel.split(","):
        continue
    dom = m.RoutingDomain(net, 3600.0, meth, frac_future=frac, uh_offset=uh_off, uh=uh, max_window=W)
    ros = [bench.device_runoff(torch, net.H, W, k * W, 7, dev) for k in range(NWIN + 1)]
    torch.cuda.synchronize()
    dom.run_device(W, 0.0, ros[0].data_ptr()); dom.sync()
    t0 = time.perf_counter()
    for k in range(NWIN):
        dom.run_device(W, (k + 1) * W * 3600.0, ros[k + 1].data_ptr())
    dom.sync()
    dt = time.perf_counter() - t0
    out[name] = dict(reach_steps_per_s=N * W * NWIN * len(meth) / dt, ms_per_step=dt / (W * NWIN) * 1e3)
    dom.close()
print(json.dumps(out))
