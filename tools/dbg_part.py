import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import mizuroute_amd as m
from mizuroute_amd.partition import PartitionedRouter, partition_network
net = m.make_network(6000, seed=8, p3=0.02)
nparts, W, steps = 3, 16, 32
P = partition_network(net, nparts)
ro = m.make_runoff(net.H, steps, seed=9, storm_prob=0.05, storm_amp=3e-6)
ff = np.array([0.5, 0.3, 0.2])
uh_off = np.arange(0, 2 * net.N + 1, 2, dtype=np.int32); uh = np.tile(np.array([0.6, 0.4]), net.N)
methods = [m.KWT, m.IRF, m.SUM]
whole = m.RoutingDomain(net, 3600.0, methods, frac_future=ff, uh_offset=uh_off, uh=uh, max_window=W)
Qw = whole.run(ro)
box = {}
def make(spec, **kw):
    g = spec.reach_global
    off = np.zeros(g.size + 1, np.int32); off[1:] = np.cumsum(np.diff(uh_off)[g])
    u = np.concatenate([uh[uh_off[x]:uh_off[x + 1]] for x in g])
    return m.RoutingDomain(spec.net, 3600.0, methods, frac_future=ff, uh_offset=off, uh=u, max_window=W, **kw)
routers = []
for rank in range(nparts):
    class T:
        def __init__(self, me): self.me = me
        def send(self, t, dst): box[(self.me, dst)] = t.clone()
        def recv(self, t, src): t.copy_(box[(src, 0)]); torch.cuda.synchronize()
    routers.append(PartitionedRouter(P, rank, make, T(rank), lambda n: torch.zeros(n, dtype=torch.float64, device="cuda"), W))
Q = np.full((steps, len(methods), net.N), np.nan)
dev = torch.device("cuda")
for w0 in range(0, steps, W):
    for rank in list(range(1, nparts)) + [0]:
        r = routers[rank]
        rt = torch.from_numpy(np.ascontiguousarray(ro[w0:w0 + W][:, r.trib_spec.hru_global])).to(dev) if r.trib is not None else None
        rm = torch.from_numpy(np.ascontiguousarray(ro[w0:w0 + W][:, r.main_spec.hru_global])).to(dev) if r.main is not None else None
        r.run_window(W, w0 * 3600.0, rt.data_ptr() if rt is not None else 0, rm.data_ptr() if rm is not None else 0)
        r.sync()
        for dom, spec in ((r.trib, r.trib_spec), (r.main, r.main_spec)):
            if dom is None: continue
            for ix, meth in enumerate(methods):
                q = dom.window_q(meth, W)
                Q[w0:w0 + W, ix, spec.reach_global[:spec.n_real]] = q[:, :spec.n_real]
for ix, meth in enumerate(methods):
    bad = Q[:, ix] != Qw[:, ix]
    print('method', meth, 'mismatch count', bad.sum(), 'of', bad.size)
    if bad.any():
        tt, rr = np.nonzero(bad)
        print(' first step', tt.min(), 'reaches mainstem frac', P.is_mainstem[rr].mean(), 'n distinct reaches', len(set(rr)))
        r0 = rr[tt == tt.min()][:5]
        for r in r0: print('  reach', r, 'main', P.is_mainstem[r], 'part', P.part_of_reach[r], 'Q', Q[tt.min(), ix, r], 'Qw', Qw[tt.min(), ix, r], 'nup', net.upOffset[r+1]-net.upOffset[r])
print('n mainstem', P.is_mainstem.sum(), 'exports', [d.export_local.size for d in P.trib])
