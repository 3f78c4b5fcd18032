#!/bin/bash
# Debug (inside gpurun): VALU / SALU / LDS instruction counts of k_sweep_kwt for the current build at c2 and the c3 shard, and the speed
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
o=gpurun_out/valu; rm -rf $o; mkdir -p $o
count() {
  lab=$1; shift
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $o/$lab -o p -- python bench.py --no-cpu-baseline --no-roofline --no-h2d --no-single-step --no-configs "$@" > $o/$lab.log 2>&1
  python - <<PY
import csv, glob, collections
agg = collections.defaultdict(float)
for f in glob.glob("$o/$lab/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_sweep_kwt" in r["Kernel_Name"]: agg[r["Counter_Name"]] += float(r["Counter_Value"])
print("$lab", " ".join(f"{k}={v:.4g}" for k, v in sorted(agg.items())))
PY
  rm -rf $o/$lab
}
count c2 --window 4096 --steps 2 --warmup 3
count c3 --config c3 --window 1024 --steps 2 --warmup 3
run() { echo "=== $*"; env $* 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('value %.4g  ms/step %.2f' % (j['value'] or 0, j['ms_per_step'] or 0), j.get('error'))
    elif 'rror' in l: print(l.rstrip())
"; }
B="python bench.py --no-cpu-baseline --no-single-step --no-configs --no-h2d --no-roofline"
run X=1 $B --steps 3 --warmup 2
run X=1 $B --config c3 --steps 4 --warmup 3
