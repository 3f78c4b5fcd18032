#!/bin/bash
# Debug (inside gpurun): HBM-side bytes per k_sweep_kwt launch (FETCH_SIZE / WRITE_SIZE in separate passes)
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
o=gpurun_out/traffic; rm -rf $o; mkdir -p $o
ARGS="--no-cpu-baseline --no-roofline --no-h2d --no-single-step --window 16384 --steps 1 --warmup 1"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $o/$c -o p -- python bench.py $ARGS > $o/$c.log 2>&1
done
python - <<PY
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob("$o/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                k = r["Kernel_Name"].split("(")[0][:40]; agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in agg:
        if "sweep_kwt" in k: print(c, k, "KiB per launch %.0f" % (agg[k] / cnt[k]), "launches", cnt[k])
PY
rm -rf $o/FETCH_SIZE $o/WRITE_SIZE
