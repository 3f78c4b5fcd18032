#!/usr/bin/env python
"""gpurun_out/<tag>_methods/ (tools/profile_methods.sh) -> profiles/<tag>_methods.md: reach-steps/s per Eulerian method,
kernel time per launch, and the SQ counters of k_stage<M> with the FP64 share of the vector peak they imply."""
import csv, json, os, re, sys
tag = sys.argv[1]
src, dst = f"gpurun_out/{tag}_methods", "profiles"
bm = json.loads([l for l in open(f"{src}/bench_methods.json") if l.startswith("{")][-1])
pm = json.load(open(f"{src}/{tag}_methods_pmc.json"))
stats = {re.sub(r"\(.*", "", r["Name"]).replace("void ", "").strip(): r for r in csv.DictReader(open(f"{src}/stats/k_kernel_stats.csv"))}
stats_full = dict(stats)
names = {"k_stage<0>": "SUM", "k_stage<1>": "IRF", "k_stage<3>": "KW", "k_stage<4>": "MC", "k_stage<5>": "DW"}
N = 100000
L = [f"# Eulerian methods, {tag}", "", "`tools/profile_methods.sh` on one MI355X: 100 000 reaches, windows of 1024 steps, one launch per stage (`k_stage<M>`; from round 4 on also `k_stage_pair<M>`, the launches that serve two overlapping windows at once: both are summed here), one method per domain;",
     "kernel times from `rocprofv3 --kernel-trace --stats`, counters from separate `--pmc` passes summed over all launches.  FP64 peak used: 78.6 TFLOP/s vector",
     "(MI355X_MICROARCH.md) = 39.3 x 10^12 FP64 lane-instructions/s with an FMA counted once; HBM 8 TB/s.", "",
     "| method | reach-steps/s | avg launch us | VALU insts / reach-step (lanes) | FP64 share of VALU | FP64 lane-inst/s (share of peak) | VALU util | waves/SIMD | algorithmic GB/s (share of HBM peak) |",
     "|---|---|---|---|---|---|---|---|---|"]
BYTES = {"SUM": 16 + 8 * 1, "IRF": 24 * 12 + 12 + 56, "KW": 440 + 12, "MC": 152 + 12, "DW": 440 + 12}
for k, nm in names.items():
    # one launch per stage (k_stage<M>) and, since round 4, the launches that serve two overlapping windows (k_stage_pair<M>)
    kp = k.replace("k_stage<", "k_stage_pair<")
    pre = (k[:-1], kp[:-1])      # "k_stage<4" matches k_stage<4> and k_stage<4, false> (round 4: instantiations with / without the rare branches)
    hit = lambda name: any(name.replace("void ", "").startswith(x + ">") or name.replace("void ", "").startswith(x + ",") for x in pre)
    ps = [v for kk, v in pm.items() if hit(kk)]
    sts = [v for kk, v in stats_full.items() if hit(kk)]
    if not ps or not sts or nm not in bm:
        continue
    p = {}
    for q in ps:
        for c, v in q.items():
            if isinstance(v, (int, float)):
                p[c] = p.get(c, 0) + v
    rs = bm[nm]["reach_steps_per_s"]
    calls, tot_ns = sum(int(x["Calls"]) for x in sts), sum(float(x["TotalDurationNs"]) for x in sts)
    fp64 = sum(p.get(c, 0) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
    lanes = p.get("SQ_THREAD_CYCLES_VALU", 0) / max(1, p.get("SQ_ACTIVE_INST_VALU", 1))
    den = p["GRBM_GUI_ACTIVE"] / 8 * 1024
    # reach-steps routed by the profiled command: 4 windows of 1024 steps
    rsteps = N * 1024 * 4
    fp64_rate = fp64 * lanes / (tot_ns * 1e-9)
    L.append(f"| {nm} | {rs:.3g} | {tot_ns/calls/1e3:.1f} | {p['SQ_INSTS_VALU']*lanes/rsteps:.0f} | {fp64/max(1,p['SQ_INSTS_VALU']):.2f} | {fp64_rate:.3g} ({fp64_rate/39.3e12:.2f}) | "
             f"{p['SQ_ACTIVE_INST_VALU']*4/den:.2f} | {p['SQ_WAVE_CYCLES']*4/den:.2f} | {rs*BYTES[nm]/1e9:.0f} ({rs*BYTES[nm]/8e12:.2f}) |")
L += ["", "Raw: `" + f"{tag}_methods_pmc.json`, `{tag}_methods_kernel_stats.csv`, `{tag}_methods_bench.json`.", ""]
open(f"{dst}/{tag}_methods.md", "w").write("\n".join(L) + "\n")
json.dump({k: v for k, v in pm.items() if "k_stage" in k}, open(f"{dst}/{tag}_methods_pmc.json", "w"), indent=1)
json.dump(bm, open(f"{dst}/{tag}_methods_bench.json", "w"), indent=1)
import shutil; shutil.copy(f"{src}/stats/k_kernel_stats.csv", f"{dst}/{tag}_methods_kernel_stats.csv")
print("\n".join(L))
