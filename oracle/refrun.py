"""Run the reference-solver harness oracle/_ref/ref_route (unmodified reference Fortran solvers +
shim modules, see oracle/README.md) on a case and parse its binary dump.

TEST INFRASTRUCTURE ONLY.  Used (a) in this container to pin the C restatement and to generate
the committed fixtures in tests/golden/, (b) by bench.py's cpu_baseline leg ("kind": "reference").
Nothing here reads /root/reference at run time; only oracle/build_ref.sh does, at build time.
"""
from __future__ import annotations

import os
import re
import struct
import subprocess
import tempfile

import numpy as np

from oracle.casefile import MAGIC_IN, serial_schedule, write_case  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.environ.get("MZR_REF_EXE", os.path.join(HERE, "_ref", "ref_route"))
MAGIC_OUT = 1297765967
WCAP = 32
NMOL = {3: 20, 4: 2, 5: 20}


def available() -> bool:
    return os.path.exists(EXE) and os.access(EXE, os.X_OK)


def build():
    subprocess.check_call([os.path.join(HERE, "build_ref.sh")])


def level_schedule(net):
    """orders = hop-distance levels, every reach its own branch: exposes all parallelism the
    reference's OpenMP loop (main_route.f90:356-405) can use on this network."""
    from mizuroute_amd.synthetic import hops_to_outlet
    dist = hops_to_outlet(net.downIndex.astype(np.int64) - 1)
    order = np.argsort(-dist, kind="stable")
    levels = dist[order]
    nlev = int(dist.max()) + 1
    counts = np.bincount(int(dist.max()) - levels, minlength=nlev)
    orderOffset = np.zeros(nlev + 1, np.int32)
    orderOffset[1:] = np.cumsum(counts)
    branchOffset = np.arange(net.N + 1, dtype=np.int32)
    return orderOffset, branchOffset, (order + 1).astype(np.int32)


def streamorder_schedule(net):
    """The reference's own intra-rank parallel schedule (domain_decomposition.f90:242-445,
    `stream_order`): one "order" per Strahler stream order, one branch per connected run of reaches
    of that order, processed upstream -> downstream inside a branch and in parallel across branches
    (main_route.f90:356-405)."""
    N = net.N
    down0 = net.downIndex.astype(np.int64) - 1
    order = net.topo_order()                      # upstream -> downstream
    so = np.zeros(N, dtype=np.int64)
    upOff, upIdx = net.upOffset, net.upIndex.astype(np.int64) - 1
    for r in order:
        ups = upIdx[upOff[r]:upOff[r + 1]]
        if ups.size == 0:
            so[r] = 1
        else:
            o = so[ups]
            mx = o.max()
            so[r] = mx + 1 if (o == mx).sum() >= 2 else mx
    head = np.ones(N, dtype=bool)                  # a branch starts where no upstream has the same order
    for r in range(N):
        ups = upIdx[upOff[r]:upOff[r + 1]]
        if ups.size and (so[ups] == so[r]).any():
            head[r] = False
    branches = {}
    for r in order:
        if head[r]:
            chain = [r]
            d = down0[r]
            while d >= 0 and so[d] == so[r]:
                chain.append(d)
                d = down0[d]
            branches.setdefault(int(so[r]), []).append(chain)
    orderOffset, branchOffset, seg = [0], [0], []
    for o in sorted(branches):
        for ch in branches[o]:
            seg.extend(ch)
            branchOffset.append(len(seg))
        orderOffset.append(len(branchOffset) - 1)
    return (np.array(orderOffset, np.int32), np.array(branchOffset, np.int32), (np.array(seg, np.int64) + 1).astype(np.int32))


class _Reader:
    def __init__(self, buf):
        self.b, self.p = buf, 0

    def i(self, n=1):
        n = int(n)      # (header fields come back as numpy int32: 8 * n overflowed on the 10 GB dump of a 5 M-reach case)
        v = np.frombuffer(self.b, "<i4", n, self.p); self.p += 4 * n
        return v if n > 1 else int(v[0])

    def d(self, n=1):
        n = int(n)
        v = np.frombuffer(self.b, "<f8", n, self.p); self.p += 8 * n
        return v.copy() if n > 1 else float(v[0])


def read_output(path, methods):
    r = _Reader(open(path, "rb").read())
    magic, N, n_steps, n_routes, ntdh_bas, dump_every = (int(x) for x in r.i(6))
    assert magic == MAGIC_OUT
    out = dict(N=N, n_steps=n_steps, steps=[], Q=[], VOL=[], QR1=[])
    while True:
        it = r.i()
        if it == -1:
            break
        q = r.d(N * n_routes).reshape(n_routes, N)
        v = r.d(N * n_routes).reshape(n_routes, N)
        qr1 = r.d(N)
        out["steps"].append(it); out["Q"].append(q); out["VOL"].append(v); out["QR1"].append(qr1)
    out["ierr"], out["ierr_step"] = r.i(), r.i()
    out["wall"] = r.d()
    out["Q"] = np.array(out["Q"]); out["VOL"] = np.array(out["VOL"]); out["QR1"] = np.array(out["QR1"])
    out["frac_future"] = r.d(ntdh_bas) if ntdh_bas > 1 else np.array([r.d()])
    out["uh_offset"] = r.i(N + 1).astype(np.int32)
    out["uh"] = r.d(int(out["uh_offset"][-1]))
    out["basin_qfuture"] = r.d(N * ntdh_bas).reshape(N, ntdh_bas)
    out["state"] = {}
    for ix in range(n_routes):
        m = r.i()
        st = {}
        flux = r.d(N * 7).reshape(N, 7)
        for k, name in enumerate(("Q", "VOL0", "VOL1", "INFLOW", "ELE", "FLOODVOL", "WB")):
            st[name] = flux[:, k].copy()
        if m == 1:
            st["irf_qfuture"] = r.d(int(out["uh_offset"][-1]))
        elif m == 2:
            nw = np.zeros(N, np.int32); w = np.zeros((N, 4, WCAP))
            for i in range(N):
                nw[i] = r.i()
                w[i] = r.d(4 * WCAP).reshape(4, WCAP)
            st["nw"] = nw; st["qf"] = w[:, 0]; st["ti"] = w[:, 1]; st["tr"] = w[:, 2]; st["rf"] = w[:, 3].astype(np.int32)
        elif m in NMOL:
            st["mol"] = r.d(N * NMOL[m]).reshape(N, NMOL[m])
        out["state"][m] = st
    return out


def run_case(net, runoff, dt, methods, nthreads=1, keep=None, time_from=0, **kw):
    """Write a case, run the reference harness, return the parsed dump (+ 'stdout').
    time_from: leading steps that run but are excluded from reach_steps_per_s (spin-up)."""
    if not available():
        raise FileNotFoundError(EXE)
    tmpdir = keep or tempfile.mkdtemp(prefix="mzrref_")
    case, outp = os.path.join(tmpdir, "case.bin"), os.path.join(tmpdir, "out.bin")
    write_case(case, net, runoff, dt, methods, **kw)
    env = dict(os.environ, OMP_NUM_THREADS=str(nthreads), MZR_REF_SKIP=str(int(time_from)))
    res = subprocess.run([EXE, case, outp, str(nthreads)], capture_output=True, text=True, env=env)
    if res.returncode != 0:
        raise RuntimeError(f"ref_route failed rc={res.returncode}: {res.stdout}\n{res.stderr}")
    out = read_output(outp, methods)
    out["stdout"] = res.stdout
    if os.path.exists(outp + ".tr"):        # constituent fluxes of every dumped step: flux and mass per method, lateral flux
        r = _Reader(open(outp + ".tr", "rb").read())
        N, nr = out["N"], len(methods)
        fl, ms, bs = [], [], []
        while r.p < len(r.b):
            r.i()
            fl.append(r.d(N * nr).reshape(nr, N)); ms.append(r.d(N * nr).reshape(nr, N)); bs.append(r.d(N))
        out["SOLFLUX"], out["SOLMASS"], out["BASIN_SOLUTE"] = np.array(fl), np.array(ms), np.array(bs)
        if keep is None:
            os.remove(outp + ".tr")
    mt = re.search(r"reach_steps_per_s=\s*([0-9.E+\-]+)", res.stdout)
    out["reach_steps_per_s"] = float(mt.group(1)) if mt else None
    if keep is None:
        os.remove(case); os.remove(outp); os.rmdir(tmpdir)
    return out


# ---- forcing remap harness (oracle/_ref/ref_remap: the reference's remap_runoff / sort_flux) ----------
EXE_REMAP = os.path.join(HERE, "_ref", "ref_remap")
MAGIC_REMAP = 1380798800


def remap_available() -> bool:
    return os.path.exists(EXE_REMAP) and os.access(EXE_REMAP, os.X_OK)


def run_remap(mp, sim, kind=None, remove_negatives=True):
    """mp: dict from mizuroute_amd.synthetic.make_remap (kind 1/2) or dict(ix_in=..., H=...) (kind 3).
    sim: [nSteps, n1] or [nSteps, n2, n1].  Returns (ierr, basinRunoff[nSteps, H])."""
    if kind is None:
        kind = 3 if "ix_in" in mp else (2 if mp.get("n2", 0) > 0 else 1)
    nSteps = sim.shape[0]
    n1 = sim.shape[-1]
    n2 = sim.shape[1] if kind == 2 else 0
    H = int(mp["H"])
    tmpdir = tempfile.mkdtemp(prefix="mzrremap_")
    case, outp = os.path.join(tmpdir, "case.bin"), os.path.join(tmpdir, "out.bin")
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32).tobytes()
    with open(case, "wb") as f:
        nMap = len(mp["hru_ix"]) if kind != 3 else 0
        nOv = len(mp["weight"]) if kind != 3 else 0
        f.write(struct.pack("9i", MAGIC_REMAP, kind, nMap, nOv, n1, n2, H, nSteps, int(bool(remove_negatives))))
        if kind in (1, 2):
            f.write(i32(mp["hru_ix"])); f.write(i32(mp["num_qhru"]))
            if kind == 1:
                f.write(i32(mp["qhru_ix"]))
                f.write(np.ascontiguousarray(mp["qhru_id"], dtype=np.int64).tobytes())
                f.write(np.ascontiguousarray(mp["src_id"], dtype=np.int64).tobytes())
            else:
                f.write(i32(mp["i_index"])); f.write(i32(mp["j_index"]))
            f.write(np.ascontiguousarray(mp["weight"], dtype=np.float64).tobytes())
        else:
            f.write(i32(mp["ix_in"]))
        f.write(np.ascontiguousarray(sim, dtype=np.float64).tobytes())
    res = subprocess.run([EXE_REMAP, case, outp], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"ref_remap failed rc={res.returncode}: {res.stdout}\n{res.stderr}")
    raw = open(outp, "rb").read()
    ierr = struct.unpack("i", raw[:4])[0]
    out = np.frombuffer(raw, dtype=np.float64, offset=4, count=H * nSteps).reshape(nSteps, H).copy()
    os.remove(case); os.remove(outp); os.rmdir(tmpdir)
    return ierr, out


# ---- the reference's start-up routines for the river network (oracle/_ref/ref_topo: augment_ntopo, mpi_domain_decomposition)
TOPO_EXE = os.path.join(HERE, "_ref", "ref_topo")


def topo_available():
    return os.path.exists(TOPO_EXE)


def run_topo(seg_id, down_id, length, slope, hru_id, hru_seg, hru_area, n_nodes=1, dt=3600.0, fshape=2.5, tscale=86400.0,
             velo=1.5, diff=5000.0, mann_n=0.01, wscale=0.001, dscale=0.0036, floodplain=False, irf=True, workdir=None):
    """Raw topology (what the topology file holds) through the UNMODIFIED augment_ntopo and mpi_domain_decomposition.
    Returns per-reach scalars and ragged lists (1-based indices as the reference holds them) and the MPI domains."""
    import tempfile
    seg_id, down_id = np.asarray(seg_id, np.int64), np.asarray(down_id, np.int64)
    n_seg, n_hru = seg_id.size, np.asarray(hru_id).size
    tmp = workdir or tempfile.mkdtemp(prefix="mzr_topo_")
    fin, fout = os.path.join(tmp, "topo_case.txt"), os.path.join(tmp, "topo_out.txt")
    with open(fin, "w") as f:
        f.write(f"{n_seg} {n_hru} {int(n_nodes)}\n")
        f.write(f"{dt!r} {fshape!r} {tscale!r} {velo!r} {diff!r} {mann_n!r} {wscale!r} {dscale!r} {int(floodplain)} {int(irf)}\n")
        for a, fmt in ((seg_id, "%d"), (down_id, "%d"), (length, "%.17g"), (slope, "%.17g"), (hru_id, "%d"), (hru_seg, "%d"), (hru_area, "%.17g")):
            f.write(" ".join(fmt % x for x in np.asarray(a)) + "\n")
    r = subprocess.run([TOPO_EXE, fin, fout], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(fout):
        raise RuntimeError(f"ref_topo failed ({r.returncode}): {r.stdout[-400:]} {r.stderr[-400:]}")
    lines = open(fout).read().split("\n")
    it = iter(lines)
    head = next(it).split()
    assert head[0] == "augment_ntopo" and int(head[1]) == 0, head
    tot = [int(x) for x in next(it).split()]
    sc = np.array([next(it).split() for _ in range(n_seg)], dtype=object)
    out = dict(tot_hru=tot[0], tot_upseg=tot[1], tot_upstream=tot[2], tot_uh=tot[3],
               downSegIndex=sc[:, 1].astype(np.int64), nHRU=sc[:, 2].astype(np.int64), nUp=sc[:, 3].astype(np.int64),
               nAllUp=sc[:, 4].astype(np.int64), rchOrder=sc[:, 5].astype(np.int64), streamOrder=sc[:, 6].astype(np.int64),
               basArea=sc[:, 7].astype(np.float64), upsArea=sc[:, 8].astype(np.float64), totalArea=sc[:, 9].astype(np.float64),
               width=sc[:, 10].astype(np.float64), depth=sc[:, 11].astype(np.float64), storage=sc[:, 12].astype(np.float64),
               man_n=sc[:, 13].astype(np.float64), floodplainSlope=sc[:, 14].astype(np.float64),
               hruContribIx=[], weight=[], upSegIndices=[], goodBasin=[], allUpSegIndices=[], timeDelayHist=[])
    ints = lambda s: np.array(s.split(), dtype=np.int64)
    flts = lambda s: np.array(s.split(), dtype=np.float64)
    for _ in range(n_seg):
        out["hruContribIx"].append(ints(next(it))); out["weight"].append(flts(next(it)))
        out["upSegIndices"].append(ints(next(it))); out["goodBasin"].append(ints(next(it)))
        out["allUpSegIndices"].append(ints(next(it))); out["timeDelayHist"].append(flts(next(it)))
    head = next(it).split()
    assert head[0] == "mpi_domain_decomposition" and int(head[1]) == 0, head
    n_dom, n_contrib = [int(x) for x in next(it).split()]
    doms = []
    for _ in range(n_dom):
        bt, node, ns, nh = [int(x) for x in next(it).split()]
        doms.append(dict(basinType=bt, idNode=node, segIndex=ints(next(it)), hruIndex=ints(next(it))))
    out["domains"], out["nContribHRU"] = doms, n_contrib
    return out
