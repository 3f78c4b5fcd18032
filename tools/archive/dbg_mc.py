"""Debug (library built with EXTRA=-DMZR_MC_STATS): histogram of Muskingum-Cunge sub-steps per reach-step, asked for and executed."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mizuroute_amd as m
from mizuroute_amd import uh as uhmod
sys.argv = sys.argv[:1]
import bench
N, W = 100000, 256
net = m.make_network(N, seed=20240529)
frac = uhmod.basin_uh(3600.0, 2.5, 86400.0)
dev = torch.device("cuda", 0)
dom = m.RoutingDomain(net, 3600.0, [m.KWT, m.MC], frac_future=frac, max_window=W)
dom.L.mzr_debug_cycles.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
for k in range(2):
    ro = bench.device_runoff(torch, net.H, W, k * W, 7, dev); torch.cuda.synchronize()
    dom.run_device(W, k * W * 3600.0, ro.data_ptr()); dom.sync()
    if k == 0:
        out = (ctypes.c_ulonglong * 32)(); dom.L.mzr_debug_cycles(dom.h, out, 1)
out = (ctypes.c_ulonglong * 32)(); dom.L.mzr_debug_cycles(dom.h, out, 1)
a = np.array(list(out), dtype=np.int64)
print("bin (log2)      :", " ".join("%9d" % (1 << i) for i in range(10)))
print("executed        :", " ".join("%9d" % v for v in a[:10]))
print("asked for ntSub :", " ".join("%9d" % v for v in a[16:26]))
L = net.params["RLENGTH"]; print("length percentiles (m) 0.1 1 10 50:", np.percentile(L, [0.1, 1, 10, 50]))


raw = (ctypes.c_ulonglong * N)()
dom.L.mzr_debug_raw.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong, ctypes.POINTER(ctypes.c_ulonglong)]
assert dom.L.mzr_debug_raw(dom.h, 32 * 1024 + 64, N, raw) == 0
per = np.array(list(raw), dtype=np.float64) / (2 * W)          # both windows (the per-reach sums are not reset)
print("mean executed sub-steps per step, per reach: max %.1f  p99.99 %.1f p99.9 %.1f p99 %.1f p90 %.1f mean %.2f" % (
    per.max(), np.percentile(per, 99.99), np.percentile(per, 99.9), np.percentile(per, 99), np.percentile(per, 90), per.mean()))
print("reaches with mean >= 16:", int((per >= 16).sum()), " >= 8:", int((per >= 8).sum()))

mx = (ctypes.c_ulonglong * 4096)()
assert dom.L.mzr_debug_raw(dom.h, 32 * 1024 + 64 + 200000, 4096, mx) == 0
mx = np.array(list(mx), dtype=np.float64); mx = mx[mx > 0]
print("per-launch maximum of executed sub-steps: mean %.1f p50 %.1f p90 %.1f max %.0f over %d launches" % (mx.mean(), np.percentile(mx, 50), np.percentile(mx, 90), mx.max(), mx.size))
