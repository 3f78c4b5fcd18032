#!/bin/bash
# round 5 (inside gpurun): bench c2 / c3 under environment settings given as arguments ("A=1,B=2" per run; "-" = none); MZR_LIB_VAR=name picks a variant build
cd ${GRAFT_REPO_ROOT:-/root/repo}
B="python bench.py --no-cpu-baseline --no-h2d --no-single-step --no-configs"
show() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j.get('roofline') or {}
        print('value %.4g ms/step %.1f frac %.4f launch_ms %.1f min %.1f max %.1f err %s' % (j['value'] or 0, j['ms_per_step'] or 0, r.get('frac') or 0, (r.get('avg_launch_us') or 0) / 1e3, (r.get('min_launch_us') or 0) / 1e3, (r.get('max_launch_us') or 0) / 1e3, j.get('error')))
    elif 'rror' in l: print(l.rstrip()[:300])
"; }
for spec in "$@"; do
  envs=$(echo "$spec" | tr ',' ' '); [ "$spec" = "-" ] && envs="X=1"
  lib=lib; for e in $envs; do case $e in MZR_LIB_VAR=*) lib=lib_var/${e#MZR_LIB_VAR=};; esac; done
  export MZR_LIB=$PWD/mizuroute_amd/$lib/libmzr_hip.so
  for c in ${CONFIGS:-c2 c3}; do
    w=2; [ $c = c3 ] && w=3
    echo "=== [$spec] $c"; env $envs $B --config $c --steps ${STEPS:-4} --warmup $w 2>&1 | show
  done
done
