"""Debug (GPU box, under rocprofv3 --kernel-trace --memory-copy-trace): six forcing windows handed over in page-locked host memory
(mzr_run_async_f32, or MODE=f64 / MODE=dev), so that the trace shows where the time between two sweeps goes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mizuroute_amd as m
from mizuroute_amd import uh as uhmod
sys.argv = sys.argv[:1]
import bench
dev = torch.device("cuda", 0)
N, W = 100000, 16384
mode = os.environ.get("MODE", "f32")
net = m.make_network(N, seed=20240529)
frac = uhmod.basin_uh(3600.0, 2.5, 86400.0)
dom = m.RoutingDomain(net, 3600.0, [m.KWT], frac_future=frac, max_window=W)
ro = bench.device_runoff(torch, net.H, W, 0, 7, dev); torch.cuda.synchronize()
dt = torch.float32 if mode == "f32" else torch.float64
hosts = [torch.empty((W, net.H), dtype=dt).pin_memory() for _ in range(2)]
for hb in hosts:
    hb.copy_(ro)
torch.cuda.synchronize()
call = {"f32": dom.run_async_f32, "f64": dom.run_async}.get(mode)
k = 0
def one():
    global k
    if call is None:
        dom.run_device(W, k * W * 3600.0, ro.data_ptr())
    else:
        call(W, k * W * 3600.0, hosts[k % 2].data_ptr())
    k += 1
for _ in range(3):
    one()
dom.sync()
t0 = time.perf_counter()
for _ in range(5):
    one()
dom.sync()
print(mode, "5 windows: %.3f s per window" % ((time.perf_counter() - t0) / 5))
