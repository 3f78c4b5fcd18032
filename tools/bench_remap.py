"""Throughput of the forcing-remap kernel (kernels_remap.hip) against the HBM roofline.
Algorithmic bytes per (HRU, step): 8 written + 8 gathered per valid overlap (weights and indices are
re-used across the steps of a tile).  Run on the GPU box: python tools/bench_remap.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mizuroute_amd as m

H, W, REPS = 100000, 1024, 20
net = m.make_network(H, seed=20240529)
out = {}
for name, n1, n2 in (("polygons_150k", 150000, 0), ("grid_464x224", 464, 224)):
    mp = m.make_remap(net.H, n1, n2, seed=11)
    dom = m.RoutingDomain(net, 3600.0, [m.SUM], frac_future=np.array([1.0]), max_window=8)
    dom.set_remap(mp)
    nsrc = n1 if n2 == 0 else n1 * n2
    src = torch.rand((W, nsrc), dtype=torch.float64, device="cuda") * 1e-7
    dst = torch.empty((W, net.H), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    dom.remap_device(W, src.data_ptr(), dst.data_ptr()); dom.sync()
    t0 = time.perf_counter()
    for _ in range(REPS):
        dom.remap_device(W, src.data_ptr(), dst.data_ptr())
    dom.sync()
    dt = (time.perf_counter() - t0) / REPS
    valid = int(((mp["qhru_ix"] if n2 == 0 else mp["i_index"]) > 0).sum())
    by = 8.0 * W * (net.H + valid)
    out[name] = dict(ms_per_window=dt * 1e3, hru_steps_per_s=net.H * W / dt, algorithmic_GBps=by / dt / 1e9,
                     frac_of_8TBps=by / dt / 8e12, overlaps_per_hru=valid / net.H, source_cells=nsrc)
print(json.dumps(out))
