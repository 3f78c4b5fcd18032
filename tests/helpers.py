"""Shared test helpers: fixtures -> RiverNetwork, error metrics."""
import os

import numpy as np

from mizuroute_amd.synthetic import RiverNetwork

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_CASES = ["cameo50_irf", "tree150_all", "tree400_kwt", "tree200_kwt_daily", "lakes500_kwt", "lakes300_dw"]
REL_TOL = 1e-6          # BASELINE.json north_star: discharge within 1e-6 relative of the reference


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    par = z["params"]
    net = RiverNetwork(N=int(z["N"]), H=int(z["H"]), downIndex=z["downIndex"], reachId=z["reachId"],
                       upOffset=z["upOffset"], upIndex=z["upIndex"], upGood=z["upGood"],
                       hruOffset=z["hruOffset"], hruIndex=z["hruIndex"], hruWeight=z["hruWeight"],
                       params={k: par[i].copy() for i, k in enumerate(RiverNetwork.PARAM_ORDER)})
    return net, z


def golden_lakes(z):
    """lakes dict of a fixture (None if the case has no lakes)."""
    if "lake_reach" not in z.files:
        return None
    return {k[5:]: (int(z[k]) if z[k].ndim == 0 else z[k]) for k in z.files if k.startswith("lake_")}


def rel_err(a, b, floor=1e-9):
    """max relative error over entries where |reference| > floor (SURVEY.md 8d parity report)."""
    a, b = np.asarray(a), np.asarray(b)
    m = np.abs(a) > floor
    if not m.any():
        return 0.0
    return float((np.abs(a - b)[m] / np.abs(a)[m]).max())


def parity_report(ref, got, floor=1e-9):
    ref, got = np.asarray(ref), np.asarray(got)
    m = np.abs(ref) > floor
    r = np.abs(ref - got)[m] / np.abs(ref)[m]
    return dict(max_rel=float(r.max()) if r.size else 0.0,
                p999=float(np.quantile(r, 0.999)) if r.size else 0.0,
                frac_within=float((r <= REL_TOL).mean()) if r.size else 1.0,
                bit_identical=float((ref == got).mean()))


def star_case(kind):
    """Inputs that drive kwt_rch through branches ordinary networks rarely reach.
    'duplicates': symmetric tributaries + uniform runoff -> equal particle times across tributaries
                  (qexmul_rch drops the duplicate, kwt_route.f90:916-918) and shocks;
    'over64':     a five-way confluence of saturated tributaries with a daily step -> more than 64
                  particles enter remove_rch at once."""
    import numpy as np
    from mizuroute_amd.synthetic import make_star_network, make_runoff
    if kind == "duplicates":
        net = make_star_network(5, 6, seed=5, identical=True)
        steps, dt = 120, 3600.0
        rng = np.random.default_rng(9)
        t = np.arange(steps)
        series = 2e-8 * (1 + np.sin(2 * np.pi * t / 24.0)) + np.where(rng.random(steps) < 0.15, 3e-6 * rng.random(steps), 0.0) + 1e-9
        ro = np.ascontiguousarray(np.repeat(series[:, None], net.H, axis=1))
    else:
        net = make_star_network(5, 6, seed=5, identical=False)
        steps, dt = 30, 86400.0
        ro = make_runoff(net.H, steps, seed=3, storm_prob=0.05, storm_amp=3e-6)
    return net, ro, dt
