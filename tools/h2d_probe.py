"""Debug (GPU box): how fast a forcing window crosses PCIe alone and beside the persistent KWT sweep."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mizuroute_amd as m
from mizuroute_amd import uh as uhmod
sys.argv = sys.argv[:1]
import bench
dev = torch.device("cuda", 0)
N, W = 100000, 16384
net = m.make_network(N, seed=20240529)
frac = uhmod.basin_uh(3600.0, 2.5, 86400.0)
dom = m.RoutingDomain(net, 3600.0, [m.KWT], frac_future=frac, max_window=W)
ro = bench.device_runoff(torch, net.H, W, 0, 7, dev); torch.cuda.synchronize()
for k in range(3):
    dom.run_device(W, k * W * 3600.0, ro.data_ptr()); dom.sync()
t0 = time.perf_counter(); dom.run_device(W, 3 * W * 3600.0, ro.data_ptr()); dom.sync(); tw = time.perf_counter() - t0
print("window alone %.3f s" % tw)
for dt, nm in ((torch.float32, "f32"), (torch.float64, "f64")):
    hb = torch.empty((W, net.H), dtype=dt).pin_memory(); hb.fill_(1e-8)
    db = torch.empty((W, net.H), dtype=dt, device=dev)
    st = torch.cuda.Stream()
    gb = hb.numel() * hb.element_size() / 1e9
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.cuda.stream(st):
            db.copy_(hb, non_blocking=True)
        st.synchronize(); tc = time.perf_counter() - t0
    print(f"{nm}: {gb:.2f} GB alone in {tc:.3f} s = {gb / tc:.1f} GB/s")
    # in chunks on two streams
    st2 = torch.cuda.Stream()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    nchunk = 8; rows = W // nchunk
    for c in range(nchunk):
        with torch.cuda.stream(st if c % 2 == 0 else st2):
            db[c * rows:(c + 1) * rows].copy_(hb[c * rows:(c + 1) * rows], non_blocking=True)
    st.synchronize(); st2.synchronize(); tc2 = time.perf_counter() - t0
    print(f"{nm}: two streams, 8 chunks: {gb / tc2:.1f} GB/s")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dom.run_device(W, 4 * W * 3600.0, ro.data_ptr())
    with torch.cuda.stream(st):
        db.copy_(hb, non_blocking=True)
    st.synchronize(); tc = time.perf_counter() - t0
    dom.sync(); tw2 = time.perf_counter() - t0
    print(f"{nm}: beside the sweep: copy done after {tc:.3f} s ({gb / tc:.1f} GB/s), window done after {tw2:.3f} s")
    del hb, db
