"""Build hygiene of the reference harness: the unmodified reference solvers compiled at -O0 beside the -O2
build the oracle is pinned to, both run on the cases of the golden fixtures and on a fresh 3 000-reach case with every
method; prints the drift between the two builds (max relative difference of the routed discharge per method, and whether
the KWT particle counts agree).  Test infrastructure; needs /root/reference and flang.
    python oracle/check_ref_O0.py          ->  oracle/_ref/O0_DRIFT.txt
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    o0 = os.path.join(HERE, "_ref", "O0")
    env = dict(os.environ, MZR_REF_OUT=o0, MZR_REF_FFLAGS="-O0 -ffp-contract=off -fopenmp")
    subprocess.check_call([os.path.join(HERE, "build_ref.sh")], env=env)
    import mizuroute_amd as m
    from mizuroute_amd import uh as uhmod
    from oracle import refrun
    lines = []

    def both(net, ro, dt, methods, **kw):
        out = []
        for exe in (os.path.join(HERE, "_ref", "ref_route"), os.path.join(o0, "ref_route")):
            refrun.EXE = exe
            out.append(refrun.run_case(net, ro, dt, methods, **kw))
        return out

    net = m.make_network(3000, seed=31, p3=0.03)
    dt, steps = 3600.0, 72
    ro = m.make_runoff(net.H, steps, seed=32, storm_prob=0.05, storm_amp=3e-6)
    frac = uhmod.basin_uh(dt, 2.5, 86400.0)
    off, uhv = uhmod.make_uh(net.params["RLENGTH"], dt, 1.5, 5000.0)
    methods = [0, 1, 2, 3, 4, 5]
    a, b = both(net, ro, dt, methods, uh=(frac, off, uhv), dump_every=1)
    assert a["ierr"] == 0 and b["ierr"] == 0, (a["ierr"], b["ierr"])
    Qa, Qb = a["Q"], b["Q"]
    for ix, mm in enumerate(methods):
        qa, qb = Qa[:, ix], Qb[:, ix]
        mask = np.abs(qa) > 1e-12
        rel = (np.abs(qa - qb)[mask] / np.abs(qa)[mask]).max() if mask.any() else 0.0
        lines.append(f"method {mm}: max relative difference of REACH_Q between -O2 and -O0 over {steps} steps x {net.N} reaches = {rel:.3e}"
                     f" ({'bit-identical' if np.array_equal(qa, qb) else 'differs'})")
    if "numWaves" in a and "numWaves" in b:
        lines.append(f"KWT particle counts equal: {bool(np.array_equal(a['numWaves'], b['numWaves']))}")
    txt = "\n".join(lines)
    print(txt)
    open(os.path.join(HERE, "_ref", "O0_DRIFT.txt"), "w").write(txt + "\n")


if __name__ == "__main__":
    main()
