#!/bin/bash
# round 5 (inside gpurun): wavefronts per SIMD of the sweep -- workgroups of two wavefronts, 8 KB of LDS each
cd ${GRAFT_REPO_ROOT:-/root/repo}
B="python bench.py --no-cpu-baseline --no-h2d --no-single-step --no-configs"
show() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j.get('roofline') or {}
        print('value %.4g ms/step %.1f frac %s launch_us %s sweep %s err %s' % (j['value'] or 0, j['ms_per_step'] or 0, r.get('frac'), r.get('avg_launch_us'), j['config'].get('kwt_sweep'), j.get('error')))
"; }
for spec in "$@"; do
  lib=${spec%%:*}; k=${spec##*:}
  export MZR_LIB=$PWD/mizuroute_amd/$lib/libmzr_hip.so
  echo "=== $lib KBLK_RUN=$k c2"; MZR_KWT_KBLK_RUN=$k $B --steps 4 --warmup 2 2>&1 | show
  echo "=== $lib KBLK_RUN=$k c3"; MZR_KWT_KBLK_RUN=$k $B --config c3 --steps 4 --warmup 3 2>&1 | show
done
