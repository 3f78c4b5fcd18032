"""Randomised KWT parity sweep (GPU vs the CPU oracle): network size, time step, confluence mix, storm
intensity, window length and lane-class threshold are drawn at random.  python tools/fuzz_kwt.py [n] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import mizuroute_amd as m
from oracle import pyoracle
from helpers import parity_report

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst = 0.0
for c in range(n_cases):
    N = int(rng.integers(300, 6000)); dt = float(rng.choice([900.0, 3600.0, 10800.0, 86400.0]))
    p3 = float(rng.choice([0.0, 0.0, 0.05, 0.2])); steps = int(rng.integers(40, 160)); W = int(rng.choice([1, 5, 16, 64]))
    prob = float(rng.choice([0.0, 0.01, 0.05, 0.2])); amp = float(rng.choice([1e-7, 1e-6, 1e-5]))
    cb = str(rng.choice(["0", "7", "16", "64"]))
    os.environ["MZR_KWT_CLASSB_MAX"] = cb
    net = m.make_network(N, seed=int(rng.integers(1 << 30)), p3=p3)
    ro = m.make_runoff(net.H, steps, seed=int(rng.integers(1 << 30)), storm_prob=prob, storm_amp=amp)
    ff = np.array([0.5, 0.3, 0.2])
    orc = pyoracle.Oracle(net, dt, [2], ff)
    try:
        Qo = orc.run(ro)
    except Exception as e:
        print(c, "oracle error (reference error path)", str(e)[:80]); continue
    dom = m.RoutingDomain(net, dt, [m.KWT], frac_future=ff, max_window=W)
    Qg = dom.run(ro)
    rep = parity_report(Qo[:, 0], Qg[:, 0])
    same_counts = bool(np.array_equal(dom.kwt_state()[0], orc.kwt_state()[0]))
    worst = max(worst, rep["max_rel"])
    print(c, dict(N=N, dt=dt, p3=p3, steps=steps, W=W, prob=prob, amp=amp, classB=cb), "max_rel %.2e" % rep["max_rel"], "counts", same_counts, orc.kwt_paths()["shock_merges"])
    assert rep["max_rel"] <= 1e-6 and same_counts
    dom.close()
print("worst max_rel %.3e over %d cases" % (worst, n_cases))
