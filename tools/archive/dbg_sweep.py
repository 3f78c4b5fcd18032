import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import mizuroute_amd as m
from mizuroute_amd import uh as uhmod
sys.argv = sys.argv[:1]
import bench
N = int(os.environ.get("NR", "100000")); W = int(os.environ.get("WW", "512"))
net = m.make_network(N, seed=20240529)
frac = uhmod.basin_uh(3600.0, 2.5, 86400.0)
dom = m.RoutingDomain(net, 3600.0, [m.KWT], frac_future=frac, max_window=W)
print("sweep info", dom.sweep_info(), "stages", dom.schedule())
dev = torch.device("cuda", 0)
for k in range(int(os.environ.get("NW", "4"))):
    ro = bench.device_runoff(torch, net.H, W, k * W, 7, dev); torch.cuda.synchronize()
    t0 = time.time()
    dom.run_device(W, k * W * 3600.0, ro.data_ptr()); dom.sync()
    dt = time.time() - t0
    print("window", k, "ok %.1f ms  %.3g reach-steps/s" % (dt * 1e3, N * W / dt), "sweep", dom.sweep_info())
