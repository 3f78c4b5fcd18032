#!/usr/bin/env python
"""Turn a gpurun_out/<tag>/ profile bundle (tools/profile_bundle.sh) into the tracked summaries under profiles/:
  <tag>_kernel_stats.csv, <tag>_kernel_stats_400k.csv   rocprofv3 --kernel-trace --stats
  <tag>_bench.json, <tag>_bench_400k.json               the un-profiled bench lines of the same build
  <tag>_pmc.md / <tag>_pmc.json                         SQ counters per kernel (separate --pmc passes) + derived utilisation
  <tag>_calib.json                                      FETCH_SIZE / WRITE_SIZE against known byte counts (tools/calib_hbm.hip)
  <tag>_summary.md                                      the table the roofline numbers come from
  kwt_hbm_traffic.json                                  HBM bytes per sweep launch, read by bench.py (roofline.traffic)
HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE in separate passes, KiB; the read and
write factors are the ones measured by the calibration kernels of the same bundle (8 bytes per lane, contiguous and
160-byte row gathers), not assumed."""
import csv
import json
import os
import re
import shutil
import sys

tag = sys.argv[1]
src, dst = os.path.join("gpurun_out", tag), "profiles"
os.makedirs(dst, exist_ok=True)


def kname(full):
    return re.sub(r"<.*", "", full.split("(")[0].replace("void ", "")).strip()


def sweep_launches(size):
    """durations [ms] of every k_sweep_kwt launch of the --kernel-trace run, in launch order"""
    sub = "stats" if size == "100k" else "stats_400k"
    out = []
    for root, _, files in os.walk(os.path.join(src, sub)):
        for fn in files:
            if fn.endswith("kernel_trace.csv"):
                rows = [r for r in csv.DictReader(open(os.path.join(root, fn))) if "k_sweep_kwt" in r["Kernel_Name"]]
                rows.sort(key=lambda r: int(r["Start_Timestamp"]))
                out = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    return out


def last_json(path):
    try:
        return json.loads([l for l in open(path).read().strip().splitlines() if l.startswith("{")][-1])
    except Exception:
        return None


calib = json.load(open(os.path.join(src, "calib.json")))
KNOWN = {"k_copy8": (2 ** 31, 2 ** 31), "k_copy16": (2 ** 31, 2 ** 31), "k_rows8": ((2 ** 28 // 20) * 164, (2 ** 28 // 20) * 8)}
cal = {}
for k, (rd, wr) in KNOWN.items():
    c = calib.get(k, {})
    if "FETCH_SIZE_KiB_per_launch" in c:
        cal[k] = dict(known_read_bytes=rd, known_write_bytes=wr, FETCH_SIZE_KiB=c["FETCH_SIZE_KiB_per_launch"], WRITE_SIZE_KiB=c["WRITE_SIZE_KiB_per_launch"],
                      read_factor=rd / (c["FETCH_SIZE_KiB_per_launch"] * 1024), write_factor=wr / (c["WRITE_SIZE_KiB_per_launch"] * 1024))
json.dump(cal, open(os.path.join(dst, f"{tag}_calib.json"), "w"), indent=1)
# contiguous 8-byte lanes: the factor that turns FETCH_SIZE into bytes that crossed the memory side (row gathers
# then show their over-fetch as traffic above the useful bytes, which is what "traffic" is for)
RF = cal.get("k_copy8", {}).get("read_factor", 2.0)
WF = cal.get("k_copy8", {}).get("write_factor", 1.0)
rows = cal.get("k_rows8", {})

lines = [f"# Profile {tag}", "",
         "Bundle: `tools/profile_bundle.sh " + tag + "` on one MI355X; every counter group in its own `rocprofv3 --pmc` run, kernel times",
         "from a separate `rocprofv3 --kernel-trace --stats` run of the same command.", "",
         f"HBM-counter calibration (`tools/calib_hbm.hip`, 2 GiB buffers): FETCH_SIZE x {RF:.3f} = bytes read, WRITE_SIZE x {WF:.3f} = bytes written",
         "for 8-byte-per-lane contiguous access (`k_copy8`); the 16-byte case (`k_copy16`) gives the same factors; scattered 160-byte rows",
         f"(`k_rows8`) fetch {rows.get('FETCH_SIZE_KiB', 0) * 1024 * RF / max(1, rows.get('known_read_bytes', 1)):.2f} x their useful bytes. Details: `{tag}_calib.json`.", ""]
out_traffic = None
pm_all = {}
for size, sfx, cmd in (("100k", "", "--window 16384 --steps 1 --warmup 1"), ("400k", "_400k", "--reaches 400000 --window 2048 --steps 2 --warmup 2")):
    sdir = os.path.join(src, "stats" + sfx)
    if not os.path.exists(os.path.join(sdir, "k_kernel_stats.csv")):
        continue
    shutil.copy(os.path.join(sdir, "k_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats{sfx}.csv"))
    bench = last_json(os.path.join(src, "bench" + sfx + ".json"))
    if bench:
        json.dump(bench, open(os.path.join(dst, f"{tag}_bench{sfx}.json"), "w"), indent=1)
    pm = json.load(open(os.path.join(src, f"{tag}_{size}_pmc.json")))
    pm = {kname(k): v for k, v in pm.items()}
    pm_all[size] = {k: v for k, v in pm.items() if k.startswith("k_")}
    stats = {}
    for r in csv.DictReader(open(os.path.join(sdir, "k_kernel_stats.csv"))):      # instantiations of one kernel template are one row
        k = kname(r["Name"])
        if k in stats:
            a = stats[k]
            a["Calls"] = str(int(a["Calls"]) + int(r["Calls"])); a["TotalDurationNs"] = str(float(a["TotalDurationNs"]) + float(r["TotalDurationNs"]))
            a["AverageNs"] = str(float(a["TotalDurationNs"]) / int(a["Calls"]))
        else:
            stats[k] = dict(r)
    lines += [f"## {size} reaches (`bench.py --no-cpu-baseline --no-roofline --no-h2d --no-single-step {cmd}`)", "",
              "| kernel | calls | avg us | total ms | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | HBM MB/launch (calibrated) |", "|---|---|---|---|---|---|---|"]
    extra = ""
    for k, r in stats.items():
        if not k.startswith("k_"):
            continue
        p = pm.get(k, {})
        n = max(1, p.get("_dispatches", 1))
        f, w = p.get("FETCH_SIZE", 0.0) / n, p.get("WRITE_SIZE", 0.0) / n
        if k == "k_sweep_kwt" and "_last" in p and "FETCH_SIZE" in p["_last"]:      # the steady window: the sum also holds the census launch and the cold first window
            f, w = p["_last"]["FETCH_SIZE"], p["_last"].get("WRITE_SIZE", w)
        hbm = (RF * f + WF * w) * 1024
        lines.append(f"| {k} | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e6:.2f} | {f:.0f} | {w:.0f} | {hbm/1e6:.2f} |")
        if k == "k_sweep_kwt" and bench:
            rf = bench.get("roofline") or {}
            per = sweep_launches(size)
            extra = (f"`k_sweep_kwt` at {size}: launches of the profiled command (rocprofv3 kernel trace, ms): {', '.join('%.2f' % x for x in per)} -- the census launch of "
                     f"mzr_init_state, the cold first window (every reach still in class A), then the steady window(s); the bench line's HIP-event time of a steady window is "
                     f"{rf.get('avg_launch_us', float('nan'))/1e3:.2f} ms; FETCH / WRITE of the LAST launch; "
                     f"algorithmic {rf.get('algorithmic_bytes_per_launch', 0)/1e6:.0f} MB per launch vs {hbm/1e6:.0f} MB at the memory side "
                     f"({hbm / max(1.0, rf.get('algorithmic_bytes_per_launch', 1.0)):.2f} x).")
            if size == "100k":
                cfg = bench["config"]
                out_traffic = dict(tag=tag, reaches=cfg["reaches_per_gpu"], window=cfg["window_steps"], hbm_bytes_per_launch=hbm, fetch_kib_per_launch=f,
                                   write_kib_per_launch=w, read_factor=RF, write_factor=WF,
                                   note="(read_factor*FETCH_SIZE + write_factor*WRITE_SIZE)*1024 per k_sweep_kwt launch, separate rocprofv3 --pmc passes; factors from "
                                        "the calibration kernels of the same bundle")
    lines += ["", extra, ""]
    if bench:
        lines += ["Bench line of the same build (un-profiled run):", "", "```", json.dumps(bench), "```", ""]
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
if out_traffic:
    json.dump(out_traffic, open(os.path.join(dst, "kwt_hbm_traffic.json"), "w"), indent=1)

# ---- SQ counters
json.dump(pm_all, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
L = [f"# SQ counters {tag}", "", "One `rocprofv3 --pmc` pass per group (tools/pmc.sh), summed over all dispatches of the profiled command; SQ_* cycle counters are in",
     "quad-cycles (MI355X_MICROARCH.md).  Derived: VALU issue utilisation = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs);",
     "wave residency = SQ_WAVE_CYCLES x 4 / the same denominator (wavefronts per SIMD on average); issue / wait shares are of SQ_WAVE_CYCLES.", ""]
for size, pm in pm_all.items():
    L += [f"## {size} reaches", "", "| kernel | VALU util | waves/SIMD | issuing | wait (s_waitcnt etc.) | wait on issue | VALU insts | SALU insts | LDS insts | VMEM rd / wr insts | FP64 share of VALU | lanes active per VALU inst |",
          "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for k, p in pm.items():
        if "SQ_WAVE_CYCLES" not in p or p["SQ_WAVE_CYCLES"] < 1e6:
            continue
        den = p["GRBM_GUI_ACTIVE"] / 8 * 1024
        fp64 = sum(p.get(c, 0) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
        L.append(f"| {k} | {p['SQ_ACTIVE_INST_VALU']*4/den:.2f} | {p['SQ_WAVE_CYCLES']*4/den:.2f} | {p.get('SQ_ACTIVE_INST_ANY',0)/p['SQ_WAVE_CYCLES']:.2f} | "
                 f"{p.get('SQ_WAIT_ANY',0)/p['SQ_WAVE_CYCLES']:.2f} | {p.get('SQ_WAIT_INST_ANY',0)/p['SQ_WAVE_CYCLES']:.2f} | {p.get('SQ_INSTS_VALU',0):.3g} | "
                 f"{p.get('SQ_INSTS_SALU',0):.3g} | {p.get('SQ_INSTS_LDS',0):.3g} | {p.get('SQ_INSTS_VMEM_RD',0):.3g} / {p.get('SQ_INSTS_VMEM_WR',0):.3g} | "
                 f"{fp64/max(1,p.get('SQ_INSTS_VALU',1)):.2f} | {p.get('SQ_THREAD_CYCLES_VALU',0)/max(1,p.get('SQ_ACTIVE_INST_VALU',1)):.1f} |")
    L += ["", "Raw sums: `" + f"{tag}_pmc.json" + "`.", ""]
open(os.path.join(dst, f"{tag}_pmc.md"), "w").write("\n".join(L) + "\n")
print("\n".join(l for l in lines if not l.startswith("{"))[:6000]); print("\n".join(L))
