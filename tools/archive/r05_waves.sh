#!/bin/bash
# round 5 (inside gpurun): is the c2 window bound by throughput or by its longest chain?  fewer wavefronts, same work
cd ${GRAFT_REPO_ROOT:-/root/repo}
B="python bench.py --no-cpu-baseline --no-h2d --no-single-step --no-configs"
show() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j.get('roofline') or {}
        print('value %.4g ms/step %.1f frac %s launch_us %s sweep %s err %s' % (j['value'] or 0, j['ms_per_step'] or 0, r.get('frac'), r.get('avg_launch_us'), j['config'].get('kwt_sweep'), j.get('error')))
"; }
for w in 4008 3000 2048 1024; do
  echo "=== waves $w c2 K=1"; MZR_KWT_KBLK_RUN=1 MZR_KWT_SWEEP_WAVES=$w $B --steps 3 --warmup 2 2>&1 | show
done
for w in 4008 2048; do
  echo "=== waves $w c3 K=1"; MZR_KWT_KBLK_RUN=1 MZR_KWT_SWEEP_WAVES=$w $B --config c3 --steps 3 --warmup 3 2>&1 | show
done
