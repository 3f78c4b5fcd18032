#!/bin/bash
# Round profile bundle (run inside gpurun): bench line, rocprofv3 kernel stats, SQ / HBM PMC passes (separate runs),
# the HBM-counter calibration, and the same for a 400 k-reach domain (past the Infinity Cache).
# usage: tools/profile_bundle.sh <tag>        then, here: python tools/summarize_profile.py <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
o=gpurun_out/$tag; rm -rf $o; mkdir -p $o
python bench.py --no-configs > $o/bench.json 2> $o/bench.err
tail -c 600 $o/bench.json
ARGS="--no-cpu-baseline --no-roofline --no-h2d --no-single-step --no-configs --window 16384 --steps 2 --warmup 3"      # (two regroupings behind it: the last windows are steady)
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o k -- python bench.py $ARGS > $o/stats.log 2>&1
bash tools/pmc.sh ${tag}_100k "python bench.py $ARGS" > $o/pmc_100k.log 2>&1
# 400 k reaches: rows no longer fit the 256 MiB Infinity Cache
ARGS4="--no-cpu-baseline --no-h2d --no-single-step --no-configs --reaches 400000 --window 2048 --steps 2 --warmup 2"
python bench.py $ARGS4 > $o/bench_400k.json 2> $o/bench_400k.err
tail -c 400 $o/bench_400k.json
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_400k -o k -- python bench.py $ARGS4 --no-roofline > $o/stats_400k.log 2>&1
bash tools/pmc.sh ${tag}_400k "python bench.py $ARGS4 --no-roofline" > $o/pmc_400k.log 2>&1
# counter calibration on known byte counts
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/calib_hbm.hip -o /tmp/calib_hbm && {
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/calib_f -o c -- /tmp/calib_hbm > $o/calib.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $o/calib_w -o c -- /tmp/calib_hbm >> $o/calib.log 2>&1
}
python - <<PY
import csv, glob, collections, json
res = {}
for name, ctr in (("calib_f", "FETCH_SIZE"), ("calib_w", "WRITE_SIZE")):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob("$o/%s/**/*counter_collection.csv" % name, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr:
                k = r["Kernel_Name"].split("(")[0]; agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in agg: res.setdefault(k, {})[ctr + "_KiB_per_launch"] = agg[k] / cnt[k]
json.dump(res, open("$o/calib.json", "w"), indent=1); print(res)
PY
cp gpurun_out/${tag}_100k_pmc.json gpurun_out/${tag}_400k_pmc.json $o/ 2>/dev/null
find $o -name "*kernel_trace.csv" -size +20M -delete
find $o -name "*counter_collection.csv" -delete
ls $o
