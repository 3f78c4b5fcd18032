#!/usr/bin/env python
"""Turn a gpurun_out/<tag>/ profile bundle (tools/profile_bundle.sh) into the tracked summary under
profiles/: kernel stats CSV, HBM PMC totals per kernel, the bench line, and
profiles/kwt_hbm_traffic.json (read by bench.py for roofline.traffic).

HBM bytes per launch follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are
collected in separate --pmc passes, are in KiB, and on gfx950 FETCH_SIZE reports half of the bytes
of a wide coalesced read, so reads are doubled:  traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import collections
import csv
import json
import os
import shutil
import re
import sys


def kname(full):
    """'void k_stage_kwt<false, 1024>(MzrDev, ...)' -> 'k_stage_kwt' (template instances pooled)."""
    return re.sub(r"<.*", "", full.split("(")[0].replace("void ", "")).strip()

tag = sys.argv[1]
src = os.path.join("gpurun_out", tag)
dst = "profiles"
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "stats", "k_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(dst, f"{tag}_bench.json"), "w"), indent=1)


def pmc(name, counter):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    path = os.path.join(src, name, "k_counter_collection.csv")
    if not os.path.exists(path):
        path += ".part"
    for row in csv.DictReader(open(path)):
        if row.get("Counter_Name") != counter:
            continue
        k = kname(row["Kernel_Name"])
        agg[k] += float(row["Counter_Value"]); cnt[k] += 1
    return agg, cnt


fetch, nf = pmc("pmc_fetch", "FETCH_SIZE")
write, nw = pmc("pmc_write", "WRITE_SIZE")
stats = {kname(r["Name"]): r for r in csv.DictReader(open(os.path.join(src, "stats", "k_kernel_stats.csv")))}
lines = [f"# Profile {tag}", "",
         "Command: `python bench.py --no-cpu-baseline --no-roofline --window 8192 --steps 1 --warmup 1` under",
         "`rocprofv3 --kernel-trace --stats` and, separately, `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE`.", "",
         "| kernel | calls | avg us | total ms | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | HBM MB/launch (2*F+W) |",
         "|---|---|---|---|---|---|---|"]
out = {}
for k, r in stats.items():
    if not k.startswith("k_"):
        continue
    f = fetch.get(k, 0.0) / max(1, nf.get(k, 1)); w = write.get(k, 0.0) / max(1, nw.get(k, 1))
    hbm = (2 * f + w) * 1024
    lines.append(f"| {k} | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e6:.2f} | {f:.0f} | {w:.0f} | {hbm/1e6:.2f} |")
    out[k] = dict(avg_us=float(r["AverageNs"]) / 1e3, fetch_kib=f, write_kib=w, hbm_bytes=hbm)
rf = bench.get("roofline") or {}
# The profiled command routes an untimed first window (cold start, lane classes not yet formed) and then
# the timed ones; the HIP-event figure of the bench line is for a window in steady state.  Compare like
# with like: the per-launch average of the LAST window's KWT launches in the kernel trace.
steady = None
tpath = os.path.join(src, "stats", "k_kernel_trace.csv")
if os.path.exists(tpath) and rf.get("launches"):
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(tpath)) if "k_stage_kwt" in r["Kernel_Name"]]
    n = int(rf["launches"])
    if len(dur) >= n:
        steady = sum(dur[-n:]) / n / 1e3
lines += ["", "Bench line of the same build (un-profiled run):", "", "```", json.dumps(bench), "```", "",
          f"KWT stage kernel: HIP-event average {rf.get('avg_launch_us', float('nan')):.1f} us vs rocprofv3 "
          f"{out.get('k_stage_kwt', {}).get('avg_us', float('nan')):.1f} us per launch over all windows"
          + (f", {steady:.1f} us over the last (steady) window" if steady else "") + "; algorithmic "
          f"{rf.get('algorithmic_bytes_per_launch', 0)/1e6:.2f} MB/launch vs HBM counters "
          f"{out.get('k_stage_kwt', {}).get('hbm_bytes', 0)/1e6:.2f} MB/launch."]
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
if "k_stage_kwt" in out:
    cfg = bench["config"]
    json.dump(dict(tag=tag, reaches=cfg["reaches_per_gpu"], window=8192, hbm_bytes_per_launch=out["k_stage_kwt"]["hbm_bytes"],
                   fetch_kib_per_launch=out["k_stage_kwt"]["fetch_kib"], write_kib_per_launch=out["k_stage_kwt"]["write_kib"],
                   note="(2*FETCH_SIZE + WRITE_SIZE)*1024, separate rocprofv3 --pmc passes, bench.py --window 8192"),
              open(os.path.join(dst, "kwt_hbm_traffic.json"), "w"), indent=1)
print("\n".join(lines))
