#!/bin/bash
# usage: tools_pmc.sh <outdir> <counters...>   (run inside gpurun)
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
out=$1; shift
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/$out -o pmc -- python bench.py --no-cpu-baseline --no-roofline --window 1024 --steps 1 --warmup 1 $MZR_PMC_ARGS > gpurun_out/$out.log 2>&1
ls gpurun_out/$out | head
python - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/$out/*counter_collection.csv")
print(f)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"][:40]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    cnt[(k,row["Counter_Name"])] += 1
for k, d in agg.items():
    if "k_stage" in k or "hillslope" in k or "basin" in k:
        print(k, {c: (v, cnt[(k,c)]) for c, v in d.items()})
PY
