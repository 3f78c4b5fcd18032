import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mizuroute_amd as m
from mizuroute_amd import uh as uhmod
sys.argv = sys.argv[:1]
import bench
N = int(os.environ.get("NR", "20000")); W = int(os.environ.get("WW", "2048"))
net = m.make_network(N, seed=20240529)
frac = uhmod.basin_uh(3600.0, 2.5, 86400.0)
dev = torch.device("cuda", 0)
pool = [bench.device_runoff(torch, net.H, W, 0, 7, dev), bench.device_runoff(torch, net.H, W, W, 7, dev)]
torch.cuda.synchronize()
doms = [m.RoutingDomain(net, 3600.0, [m.KWT], frac_future=frac, max_window=W) for _ in range(2)]
hosts = [torch.empty((W, net.H), dtype=torch.float64).pin_memory() for _ in range(2)]
for hb, ro in zip(hosts, pool):
    hb.copy_(ro)
torch.cuda.synchronize()
nw = 6
for k in range(nw):
    doms[0].run_device(W, k * W * 3600.0, pool[k % 2].data_ptr())
doms[0].sync()
for k in range(nw):
    doms[1].run_async(W, k * W * 3600.0, hosts[k % 2].data_ptr())
    if os.environ.get("SYNC_EACH"): doms[1].sync()
doms[1].sync()
a, b = doms[0].kwt_state(), doms[1].kwt_state()
print("state equal:", all(np.array_equal(x, y) for x, y in zip(a, b)))
