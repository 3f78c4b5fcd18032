#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or overlapping or mc or route_sweep" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "mc_substep or c4" 2>&1 | tail -4
B="python bench.py --no-cpu-baseline --no-h2d --no-single-step --no-configs --config c4"
for hm in 0 4 3 6 10; do
  echo "=== MZR_MC_HEAVY_MIN=$hm"; MZR_MC_HEAVY_MIN=$hm $B --steps 5 --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j.get('roofline') or {}
print('value %.4g ms/step %.1f frac %s launch_us %s err %s' % (j['value'] or 0, j['ms_per_step'] or 0, r.get('frac'), r.get('avg_launch_us'), j.get('error')))"
done
