"""Host-side, once-per-run setup products the hot path consumes.

  basin_uh  -- FRAC_FUTURE, the hillslope gamma-distribution time-delay histogram
               (route/build/src/process_param.f90:13-92 `basinUH`, incomplete gamma function
               route/build/src/gamma_func.f90 `gammp/gser/gcf/gammln`)
  make_uh   -- per-reach impulse-response unit hydrographs from the Saint-Venant solution of
               Lohmann et al. (1996) (route/build/src/process_param.f90:99-262 `make_uh`)

Checked against the reference's own output in tests/test_uh.py (fixtures carry FRAC_FUTURE and UH
computed by the reference routines).
"""
from __future__ import annotations

import math

import numpy as np

_EPS = np.finfo(np.float64).eps
_FPMIN = np.finfo(np.float64).tiny / _EPS
_COEF = (76.18009172947146, -86.50532032941677, 24.01409824083091,
         -1.231739572450155, 0.1208650973866179e-2, -0.5395239384953e-5)


def gammln(xx: float) -> float:
    x = xx
    tmp = x + 5.5
    tmp = (x + 0.5) * math.log(tmp) - tmp
    s = 0.0
    for j, c in enumerate(_COEF):
        s += c / (x + 1.0 + j)
    return tmp + math.log(2.5066282746310005 * (1.000000000190015 + s) / x)


def _gser(a: float, x: float) -> float:
    if x == 0.0:
        return 0.0
    ap, summ = a, 1.0 / a
    dl = summ
    for _ in range(100):
        ap += 1.0
        dl = dl * x / ap
        summ += dl
        if abs(dl) < abs(summ) * _EPS:
            break
    return summ * math.exp(-x + a * math.log(x) - gammln(a))


def _gcf(a: float, x: float) -> float:
    if x == 0.0:
        return 1.0
    b = x + 1.0 - a
    c = 1.0 / _FPMIN
    d = 1.0 / b
    h = d
    for i in range(1, 101):
        an = -i * (i - a)
        b += 2.0
        d = an * d + b
        if abs(d) < _FPMIN:
            d = _FPMIN
        c = b + an / c
        if abs(c) < _FPMIN:
            c = _FPMIN
        d = 1.0 / d
        dl = d * c
        h *= dl
        if abs(dl - 1.0) <= _EPS:
            break
    return math.exp(-x + a * math.log(x) - gammln(a)) * h


def gammp(a: float, x: float) -> float:
    """Regularised lower incomplete gamma function P(a, x) (gamma_func.f90:19-35)."""
    return _gser(a, x) if x < a + 1.0 else 1.0 - _gcf(a, x)


def basin_uh(dt: float, fshape: float, tscale: float) -> np.ndarray:
    """FRAC_FUTURE(1:ntdh): fraction of hillslope runoff reaching the channel in future steps."""
    cumprob = gammp(fshape, dt / tscale)
    if cumprob > 0.999:
        ntdh_try = 1.999
    else:
        lo, hi = 1.0, 1000.0
        ntdh_try = 0.5 * (lo + hi)
        for itry in range(1, 101):
            cumprob = gammp(fshape, dt * ntdh_try / tscale)
            if cumprob < 0.99:
                lo = ntdh_try
            if cumprob > 0.999:
                hi = ntdh_try
            if 0.99 < cumprob < 0.999:
                break
            ntdh_try = 0.5 * (lo + hi)
            if itry == 100:
                raise RuntimeError("basinUH/cannot identify the maximum number of bins for the tdh")
    ntdh = int(math.ceil(ntdh_try))
    frac = np.zeros(ntdh)
    psave = 0.0
    for j in range(1, ntdh + 1):
        cum = gammp(fshape, (float(j) * dt) / tscale)
        frac[j - 1] = max(0.0, cum - psave)
        psave = cum
    return frac / frac.sum()


def make_uh(length: np.ndarray, dt: float, velo: float, diff: float):
    """Per-reach unit hydrographs at the simulation step; returns (uhOffset[N+1], uh[sum ntdh])."""
    length = np.asarray(length, dtype=np.float64)
    n_seg = length.shape[0]
    dTUH, nHr = 3600.0, 240
    nTsub = int(math.ceil(dt / dTUH))
    fr = np.zeros(nHr)
    fr[:nTsub] = 1.0 / nTsub
    sec = dTUH * np.arange(1, nHr + 1)
    offs = np.zeros(n_seg + 1, dtype=np.int32)
    chunks = []
    for i in range(n_seg):
        L = length[i]
        if velo > 0.0:
            pot = ((velo * sec - L) ** 2.0) / (4.0 * diff * sec)
            H = np.where(pot > 69.0, 0.0, 1.0 / (2.0 * np.sqrt(3.14159265359 * diff * sec)) * L * np.exp(-np.minimum(pot, 700.0)))
        else:
            H = np.zeros(nHr)
        inte = float(np.cumsum(H)[-1])
        UHM = H / inte if inte > 0.0 else H
        cs = np.cumsum(UHM)
        # (the thresholds are default-real literals in the source: process_param.f90:184,190,222)
        idx = np.nonzero(cs > float(np.float32(0.99999)))[0]
        iHrLast = int(idx[0]) + 1 if idx.size else nHr
        csr = np.cumsum(UHM[::-1])
        idx = np.nonzero(csr > float(np.float32(0.99999)))[0]
        iHrStrt = nHr - int(idx[0]) if idx.size else 1
        UHQ = np.zeros(nHr)
        for jHr in range(1, nHr + 1):
            acc = 0.0
            for iHr in range(iHrStrt, iHrLast + 1):
                if jHr - iHr > 0:
                    if jHr - iHr <= nTsub:
                        acc += fr[jHr - iHr - 1] * UHM[iHr - 1]
                else:
                    break
            UHQ[jHr - 1] = acc
        inte = float(np.cumsum(UHQ)[-1])
        if inte > 0.0:
            UHQ = UHQ / inte
        cs = np.cumsum(UHQ)
        idx = np.nonzero(cs > float(np.float32(0.9999)))[0]
        iHrLast = int(idx[0]) + 1 if idx.size else nHr
        UHQ = UHQ / cs[iHrLast - 1]
        ntdh = (iHrLast + nTsub - 1) // nTsub
        u = np.zeros(ntdh)
        for jHr in range(1, iHrLast + 1):
            u[(jHr + nTsub - 1) // nTsub - 1] += UHQ[jHr - 1]
        chunks.append(u)
        offs[i + 1] = offs[i] + ntdh
    return offs, np.concatenate(chunks) if chunks else np.zeros(0)
