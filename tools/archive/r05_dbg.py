"""debugging aid (round 5): the blocked flavour of the KWT sweep against the one-step flavour, step by step"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mizuroute_amd as m
from helpers import load_golden, golden_lakes

def run(name, W, env):
    for k, v in env.items():
        os.environ[k] = v
    net, z = load_golden(name)
    methods = [int(x) for x in z["methods"]]
    dom = m.RoutingDomain(net, float(z["dt"]), methods, frac_future=z["frac_future"], uh_offset=z["uh_offset"], uh=z["uh"], lakes=golden_lakes(z), max_window=W)
    try:
        Q = dom.run(z["runoff"])
    except Exception as e:
        print("   ", env, "FAILED", str(e)[:200]); Q = None
    for k in env:
        os.environ.pop(k)
    return net, z, Q, methods

name, W = sys.argv[1], int(sys.argv[2])
net, z, Qa, methods = run(name, W, {"MZR_KWT_KBLK_RUN": "1"})
ix = methods.index(2)
for env in ({"MZR_KWT_KBLK_RUN": "4"}, {"MZR_KWT_KBLK_RUN": "4", "MZR_KWT_CLASSB_MAX": "0", "MZR_KWT_CLASSC_MAX": "0"},
            {"MZR_KWT_KBLK_RUN": "4", "MZR_KWT_CLASSB_MAX": "64", "MZR_KWT_CLASSC_MAX": "0"}, {"MZR_KWT_KBLK_RUN": "4", "MZR_KWT_CLASSB_MAX": "64", "MZR_KWT_CLASSC_MAX": "64"},
            {"MZR_KWT_KBLK_RUN": "4", "MZR_KWT_CLASSB_MAX": "20", "MZR_KWT_CLASSC_MAX": "0"}):
    _, _, Qb, _ = run(name, W, env)
    if Qb is None:
        continue
    bad = np.argwhere(Qa[:, ix, :] != Qb[:, ix, :])
    print(env, "differences:", len(bad), "first", bad[:6].tolist() if len(bad) else None)
    if len(bad):
        t, r = bad[0]
        print("   step", t, "(in window:", t % W, ") reach", r, "K1", Qa[t, ix, r], "K4", Qb[t, ix, r], " nUp", int(net.upOffset[r + 1] - net.upOffset[r]))
