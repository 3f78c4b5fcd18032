"""debugging aid (round 5): state after every window, blocked against one-step flavour"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mizuroute_amd as m
from helpers import load_golden, golden_lakes

name, W = sys.argv[1], int(sys.argv[2])
extra = dict(kv.split("=") for kv in sys.argv[3:])
net, z = load_golden(name)
methods = [2]
dt = float(z["dt"])
ro = z["runoff"]
def make(env):
    for k, v in env.items():
        os.environ[k] = v
    return m.RoutingDomain(net, dt, methods, frac_future=z["frac_future"], max_window=W)
a = make({"MZR_KWT_KBLK_RUN": "1"})
b = make(dict({"MZR_KWT_KBLK_RUN": "4"}, **extra))
nwin = ro.shape[0] // W
for k in range(nwin):
    os.environ["MZR_KWT_KBLK_RUN"] = "1"
    Qa = a.run(ro[k * W:(k + 1) * W], t_start=k * W * dt)
    os.environ["MZR_KWT_KBLK_RUN"] = "4"
    Qb = b.run(ro[k * W:(k + 1) * W], t_start=k * W * dt)
    sa, sb = a.kwt_state(), b.kwt_state()
    dq = np.argwhere(Qa[:, 0, :] != Qb[:, 0, :])
    dn = np.nonzero(sa[0] != sb[0])[0]
    mask = np.arange(sa[1].shape[1])[None, :] < sa[0][:, None]
    dqf = np.argwhere((sa[1] != sb[1]) & mask)
    dti = np.argwhere((sa[2] != sb[2]) & mask)
    print(f"window {k}: Q diffs {len(dq)} first {dq[:4].tolist()}; nw diffs {dn[:6].tolist()}; qf diffs {len(dqf)} {dqf[:4].tolist()}; ti diffs {len(dti)} {dti[:4].tolist()}; max nw {sa[0].max()}")
    if len(dq):
        t, r = dq[0]
        print("    nw of that reach", sa[0][r], sb[0][r], "Q", Qa[t, 0, r], Qb[t, 0, r])
        break
