import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import mizuroute_amd as m
from helpers import load_golden, golden_lakes
net, z = load_golden("lakes500")
lakes = golden_lakes(z)
methods = [int(x) for x in z["methods"]]
lk = lakes["reach"] - 1
down0 = net.downIndex.astype(int) - 1
for meths in ([1], [5], [2]):
    dom = m.RoutingDomain(net, float(z["dt"]), meths, frac_future=z["frac_future"], uh_offset=z["uh_offset"], uh=z["uh"], lakes=lakes, max_window=1000)
    Q = dom.run(z["runoff"])
    ref = z["ref_Q"][:, methods.index(meths[0]), :]
    rel = np.abs(Q[:, 0] - ref) / np.maximum(np.abs(ref), 1e-30)
    bad = rel > 1e-9
    print("method", meths, "max rel", rel.max(), "bad count", bad.sum())
    if bad.any():
        tt, rr = np.nonzero(bad)
        t0 = tt.min(); r0 = sorted(set(rr[tt == t0]))
        print(" first bad step", t0, "reaches", r0[:10])
        for r in r0[:6]:
            print("  reach", r, "is lake", r in set(lk), "type", (lakes["model_type"][list(lk).index(r)] if r in set(lk) else None),
                  "down is lake", down0[r] in set(lk), "ups lakes", [int(u - 1) in set(lk) for u in net.upIndex[net.upOffset[r]:net.upOffset[r + 1]]],
                  "Q", Q[t0, 0, r], "ref", ref[t0, r])
print("---- no lakes passed to the device, same network (sanity)")
dom = m.RoutingDomain(net, float(z["dt"]), [1], frac_future=z["frac_future"], uh_offset=z["uh_offset"], uh=z["uh"], max_window=1000)
Q = dom.run(z["runoff"][:3])
from oracle import pyoracle
orc = pyoracle.Oracle(net, float(z["dt"]), [1], z["frac_future"], z["uh_offset"], z["uh"])
Qo = orc.run(z["runoff"][:3])
print("no-lake run vs oracle identical:", np.array_equal(Q, Qo))
dom = m.RoutingDomain(net, float(z["dt"]), [1], frac_future=z["frac_future"], uh_offset=z["uh_offset"], uh=z["uh"], lakes=lakes, max_window=1000)
Q2 = dom.run(z["runoff"][:3])
qr1 = dom.flux(1, m.api.F_BASIN_QR1)
orc2 = pyoracle.Oracle(net, float(z["dt"]), [1], z["frac_future"], z["uh_offset"], z["uh"]); orc2.set_lakes(lakes)
lk3 = dict(lakes); 
Qo2 = orc2.run_lake(z["runoff"][:3], {**lakes, "evap": lakes["evap"][:3], "precip": lakes["precip"][:3], "ymd": lakes["ymd"][:3]})
print("lake run vs oracle: max rel", (np.abs(Q2 - Qo2) / np.maximum(np.abs(Qo2), 1e-30)).max())
qo = orc2.flux(0, pyoracle.F_BASIN_QR1)
badq = np.nonzero(qr1 != qo)[0]
print("BASIN_QR1 mismatches", badq.size, badq[:10], "lake reaches", sorted(lk)[:10])
hw = np.nonzero(np.diff(net.upOffset) == 0)[0]
print("headwater Q mismatches step0:", (Q2[0, 0, hw] != Qo2[0, 0, hw]).sum(), "of", hw.size)
