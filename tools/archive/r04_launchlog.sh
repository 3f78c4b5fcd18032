#!/bin/bash
# Debug (inside gpurun): per-window duration of k_sweep_kwt in the driver's c2 leg, several fresh processes
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
  MZR_LAUNCH_LOG=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline --no-h2d --no-single-step 2> gpurun_out/launchlog_$i.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('run value %.4g frac %s' % (j['value'] or 0, (j.get('roofline') or {}).get('frac')))
"
  grep "^mzr launch" gpurun_out/launchlog_$i.err | awk '{printf "%s ", $6} END {print ""}'
done
