#!/bin/bash
# Debug (inside gpurun): lane-class thresholds of the KWT sweep at the c2 (100 k, latency-bound) and c3-shard (375 k, throughput-bound) operating points
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "=== $*"; env $* 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('value %.4g  ms/step %.2f' % (j['value'], j['ms_per_step']))
    elif 'rror' in l: print(l.rstrip())
"; }
B="python bench.py --no-cpu-baseline --no-h2d --no-single-step --no-roofline --no-configs"
for v in "X=1" "MZR_KWT_CLASSB_MAX=24" "MZR_KWT_CLASSB_MAX=28" "MZR_KWT_CLASSC_MAX=11" "MZR_KWT_CLASSB_MAX=24 MZR_KWT_CLASSC_MAX=11"; do
  run $v $B --steps 3 --warmup 2
  run $v $B --config c3 --steps 4 --warmup 3
done
