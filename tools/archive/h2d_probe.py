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
kk = [3]
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
    kk[0] += 1
    dom.run_device(W, kk[0] * W * 3600.0, ro.data_ptr())
    with torch.cuda.stream(st):
        db.copy_(hb, non_blocking=True)
    st.synchronize(); tc = time.perf_counter() - t0
    dom.sync(); tw2 = time.perf_counter() - t0
    print(f"{nm}: beside the sweep: copy done after {tc:.3f} s ({gb / tc:.1f} GB/s), window done after {tw2:.3f} s")
    del hb, db

# the library's own pipeline: K windows handed over in page-locked host memory, queued back to back
for dt, nm, call in ((torch.float32, "f32", dom.run_async_f32), (torch.float64, "f64", dom.run_async)):
    hosts = [torch.empty((W, net.H), dtype=dt).pin_memory() for _ in range(2)]
    for hb in hosts:
        hb.copy_(ro)
    K = 6
    kk[0] += 1; call(W, kk[0] * W * 3600.0, hosts[0].data_ptr()); dom.sync()
    t0 = time.perf_counter(); marks = []
    for k in range(K):
        kk[0] += 1
        call(W, kk[0] * W * 3600.0, hosts[(k + 1) % 2].data_ptr())
        marks.append(time.perf_counter() - t0)
    dom.sync(); el = time.perf_counter() - t0
    print(f"{nm}: {K} windows queued back to back: {el / K:.3f} s per window; the calls returned after", " ".join(f"{x:.3f}" for x in marks), "s; sweep wavefronts arrived / joined:", dom.sweep_arrivals()[:2])
    # the same with a synchronisation after every window (no overlap of copy and sweep at all)
    t0 = time.perf_counter()
    for k in range(3):
        kk[0] += 1
        call(W, kk[0] * W * 3600.0, hosts[k % 2].data_ptr()); dom.sync()
    print(f"{nm}: one window at a time: {(time.perf_counter() - t0) / 3:.3f} s per window")
    del hosts
t0 = time.perf_counter()
for k in range(6):
    kk[0] += 1
    dom.run_device(W, kk[0] * W * 3600.0, ro.data_ptr())
dom.sync()
print("resident, 6 windows queued: %.3f s per window" % ((time.perf_counter() - t0) / 6), dom.sweep_arrivals()[:2])
