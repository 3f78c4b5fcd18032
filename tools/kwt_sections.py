"""Debug: per-section wave-cycle breakdown of the KWT stage kernel.
Build the library with `make -C mizuroute_amd/csrc clean all EXTRA=-DMZR_KWT_TIMING` first."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mizuroute_amd as m
from mizuroute_amd import uh as uhmod
sys.argv = sys.argv[:1]
import bench
net = m.make_network(100000, seed=20240529)
frac = uhmod.basin_uh(3600.0, 2.5, 86400.0)
W = int(os.environ.get('WW', '512'))
dom = m.RoutingDomain(net, 3600.0, [m.KWT], frac_future=frac, max_window=W)
dev = torch.device("cuda", 0)
ro = bench.device_runoff(torch, net.H, W, 0, 7, dev); torch.cuda.synchronize()
dom.run_device(W, 0.0, ro.data_ptr()); dom.sync()
buf = (C.c_ulonglong * 32)()
dom.L.mzr_debug_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
dom.L.mzr_debug_cycles(dom.h, buf, 1)
ro = bench.device_runoff(torch, net.H, W, W, 7, dev); torch.cuda.synchronize()
dom.run_device(W, W * 3600.0, ro.data_ptr()); dom.sync()
dom.L.mzr_debug_cycles(dom.h, buf, 1)
names = ["0 setup/need", "1 load own + merge", "2 min/inflow", "3 remove", "4 celerity pow", "5 shock search", "6 routing loop", "7 interp + stores"]
tot = max(1, sum(buf[i] for i in range(8)) + sum(buf[i] for i in range(16, 20)) + buf[21] + buf[22] + buf[25] + buf[26])
for i, n in enumerate(names):
    print(f"{n:22s} {buf[i]:16d}  {100.0*buf[i]/tot:5.1f}%")
for i, n in zip(list(range(16, 20)) + [21, 25, 26, 22], ["7a count routed", "7b interp", "7c Q_END + scalar stores", "7d outbox stores", "P1 wait for up / downstream (all, narrow passes)", "P1b wait for the own step (16 lanes)", "P2a at-rest stores, drain, publish own (16 lanes)", "P2 drain + publish"]):
    print(f"{n:22s} {buf[i]:16d}  {100.0*buf[i]/tot:5.1f}%")
print("(7 = at-rest stores only when the 7a-7d stamps are present)")
print("dependency polls that had to wait", buf[23], "spin iterations", buf[24])
print("stamped wave passes", buf[20], "cycles per pass", tot / max(1, buf[20]))
print("slow merges", buf[8], "exit-time fixes", buf[9], "deferred to next round", buf[10], "sampled routed reach-steps (1/16 of blocks)", buf[11],
      "mean LDS need", buf[12] / max(1, buf[11]), "thinned", buf[13], "particles removed", buf[14], "shock merges", buf[15])
print("size histogram (<=4,<=8,<=12,<=16,<=20,<=32,<=48,>48; with -DMZR_KWT_HIST the counters 8..15 hold this instead):", [buf[8 + i] for i in range(8)])
if buf[3] and buf[6]:
    print("with -DMZR_KWT_HIST: class A waves mean %.0f max %d cycles (n=%d); class B waves mean %.0f max %d cycles (n=%d)" % (buf[2] / buf[3], buf[4], buf[3], buf[5] / buf[6], buf[7], buf[6]))
