"""Debug: per-pass records of the persistent KWT sweep (library built with EXTRA=-DMZR_KWT_TIMING)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mizuroute_amd as m
from mizuroute_amd import uh as uhmod
sys.argv = sys.argv[:1]
import bench
N = int(os.environ.get("NR", "100000")); W = int(os.environ.get("WW", "2048"))
net = m.make_network(N, seed=20240529)
frac = uhmod.basin_uh(3600.0, 2.5, 86400.0)
dom = m.RoutingDomain(net, 3600.0, [m.KWT], frac_future=frac, max_window=W)
dev = torch.device("cuda", 0)
for k in range(3):
    ro = bench.device_runoff(torch, net.H, W, k * W, 7, dev); torch.cuda.synchronize()
    dom.run_device(W, k * W * 3600.0, ro.data_ptr()); dom.sync()
buf = np.zeros((65536, 16), dtype=np.uint32)
dom.L.mzr_debug_records.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
n = dom.L.mzr_debug_records(dom.h, buf.ctypes.data_as(C.c_void_p), 65536)
r = buf[:n].astype(np.int64)
names = ["wait deps", "setup/loads", "merge", "min+remove", "pow+shock", "routing", "count+interp", "stores", "drain"]
tot = r[:, 3:12].sum(axis=1)
print("records", n)
for G in (4, 8, 16):
    sel = r[:, 0] == G
    if not sel.any(): continue
    t = tot[sel]; c = t - r[sel, 3]
    print(f"G={G}: passes {sel.sum()}  total cycles mean {t.mean():.0f} p50 {np.percentile(t,50):.0f} p90 {np.percentile(t,90):.0f} p99 {np.percentile(t,99):.0f} max {t.max()}")
    print(f"        without the dependency wait: mean {c.mean():.0f} p50 {np.percentile(c,50):.0f} p90 {np.percentile(c,90):.0f} p99 {np.percentile(c,99):.0f} max {c.max()}")
    for lo, hi in ((0, 8), (9, 16), (17, 24), (25, 32), (33, 40), (41, 48), (49, 64)):
        s2 = sel & (r[:, 1] >= lo) & (r[:, 1] <= hi)
        if s2.sum() < 5: continue
        row = " ".join(f"{nm} {r[s2, 3 + i].mean():.0f}" for i, nm in enumerate(names))
        print(f"   size {lo}-{hi}: n {s2.sum()}  removed(mean) {r[s2,2].mean():.1f}  compute {(tot[s2]-r[s2,3]).mean():.0f} | {row} | of merge: wait for rows + staging {r[s2,12].mean():.0f}, rank loops {r[s2,13].mean():.0f}, rest of the loop {r[s2,14].mean():.0f}")

# the slowest passes (without their dependency wait): what the window's longest chain is made of
sel = r[:, 0] == 16
if sel.any():
    c = tot - r[:, 3]
    idx = np.argsort(np.where(sel, c, -1))[::-1][:12]
    for i in idx:
        print("   slowest: size %d removed %d compute %d | " % (r[i, 1], r[i, 2], c[i]) + " ".join(f"{nm} {r[i, 3 + k]}" for k, nm in enumerate(names)))
    big = sel & (r[:, 1] >= 45)
    if big.sum():
        print(f"passes holding a list of >= 45 entries: n {big.sum()}  compute mean {c[big].mean():.0f} p50 {np.percentile(c[big], 50):.0f} max {c[big].max()}  removed mean {r[big, 2].mean():.1f}; "
              f"{W} steps x mean = {W * c[big].mean() / 2.0e9 * 1e3:.1f} ms at 2.0 GHz")
