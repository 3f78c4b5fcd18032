#!/bin/bash
# Debug (inside gpurun): the driver's c2 leg without the per-window events, fresh processes
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3 4 5 6; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline --no-h2d --no-single-step --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('run (no events) value %.4g ms/step %.1f' % (j['value'] or 0, j['ms_per_step']))
"
done
