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
    """Per-reach unit hydrographs at the simulation step; returns (uhOffset[N+1], uh[sum ntdh]).
    process_param.f90:99-262 (make_uh), all reaches at once: the sums run in the source's order (np.cumsum is sequential
    along a row), so the result is what the per-reach loop gives, bit for bit."""
    length = np.asarray(length, dtype=np.float64)
    n_seg = length.shape[0]
    dTUH, nHr = 3600.0, 240
    nTsub = int(math.ceil(dt / dTUH))
    fr = np.zeros(nHr)
    fr[:nTsub] = 1.0 / nTsub
    sec = dTUH * np.arange(1, nHr + 1)
    thr5, thr4 = float(np.float32(0.99999)), float(np.float32(0.9999))      # default-real literals in the source (:184,190,222)
    offs = np.zeros(n_seg + 1, dtype=np.int64)
    chunks = []
    hrs = np.arange(1, nHr + 1)
    for c0 in range(0, n_seg, 65536):
        L = length[c0:c0 + 65536][:, None]
        n = L.shape[0]
        if velo > 0.0:
            pot = ((velo * sec[None, :] - L) ** 2.0) / (4.0 * diff * sec[None, :])
            H = np.where(pot > 69.0, 0.0, 1.0 / (2.0 * np.sqrt(3.14159265359 * diff * sec[None, :])) * L * np.exp(-np.minimum(pot, 700.0)))
        else:
            H = np.zeros((n, nHr))
        inte = np.cumsum(H, axis=1)[:, -1:]
        UHM = np.where(inte > 0.0, H / np.where(inte > 0.0, inte, 1.0), H)
        cs = np.cumsum(UHM, axis=1)
        hit = cs > thr5
        iHrLast = np.where(hit.any(axis=1), hit.argmax(axis=1) + 1, nHr)
        csr = np.cumsum(UHM[:, ::-1], axis=1)
        hit = csr > thr5
        iHrStrt = np.where(hit.any(axis=1), nHr - hit.argmax(axis=1), 1)
        inside = (hrs[None, :] >= iHrStrt[:, None]) & (hrs[None, :] <= iHrLast[:, None])
        UHMm = np.where(inside, UHM, 0.0)
        # UHQ(jHr) = sum over iHr = iHrStrt..iHrLast (ascending) of fr(jHr-iHr) * UHM(iHr) for 0 < jHr-iHr <= nTsub
        UHQ = np.zeros((n, nHr))
        for d in range(min(nTsub, nHr - 1), 0, -1):
            UHQ[:, d:] = UHQ[:, d:] + fr[d - 1] * UHMm[:, :nHr - d]
        inte = np.cumsum(UHQ, axis=1)[:, -1:]
        UHQ = np.where(inte > 0.0, UHQ / np.where(inte > 0.0, inte, 1.0), UHQ)
        cs = np.cumsum(UHQ, axis=1)
        hit = cs > thr4
        iHrLast = np.where(hit.any(axis=1), hit.argmax(axis=1) + 1, nHr)
        UHQ = UHQ / cs[np.arange(n), iHrLast - 1][:, None]
        ntdh = (iHrLast + nTsub - 1) // nTsub
        UHQ = np.where(hrs[None, :] <= iHrLast[:, None], UHQ, 0.0)
        nT = (nHr + nTsub - 1) // nTsub
        pad = np.zeros((n, nT * nTsub)); pad[:, :nHr] = UHQ
        grp = pad.reshape(n, nT, nTsub)
        u = np.zeros((n, nT))
        for k in range(nTsub):          # hours of a step are added one after the other
            u = u + grp[:, :, k]
        keep = np.arange(nT)[None, :] < ntdh[:, None]
        chunks.append(u[keep])
        offs[c0 + 1:c0 + n + 1] = offs[c0] + np.cumsum(ntdh)
    return offs.astype(np.int32), np.concatenate(chunks) if chunks else np.zeros(0)
