#!/bin/bash
# round 5 (inside gpurun): LDS pool of 208 entries (8 KB per wavefront: five wavefronts per SIMD) against 240, one step per visit
cd ${GRAFT_REPO_ROOT:-/root/repo}
B="python bench.py --no-cpu-baseline --no-h2d --no-single-step --no-configs"
show() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j.get('roofline') or {}
        print('value %.4g ms/step %.1f frac %s launch_us %s sweep %s err %s' % (j['value'] or 0, j['ms_per_step'] or 0, r.get('frac'), r.get('avg_launch_us'), j['config'].get('kwt_sweep'), j.get('error')))
"; }
for lib in lib lib_var/pool208; do
  for k in 1 4; do
    export MZR_LIB=$PWD/mizuroute_amd/$lib/libmzr_hip.so
    echo "=== $lib KBLK_RUN=$k c2"; MZR_KWT_KBLK_RUN=$k $B --steps 4 --warmup 2 2>&1 | show
    echo "=== $lib KBLK_RUN=$k c3"; MZR_KWT_KBLK_RUN=$k $B --config c3 --steps 4 --warmup 3 2>&1 | show
  done
done
export MZR_LIB=$PWD/mizuroute_amd/lib_var/timing/libmzr_hip.so
echo "######## sections at W=16384, one step per visit"
MZR_KWT_KBLK_RUN=1 WW=16384 python tools/kwt_sections.py 2>&1 | grep -v amdgpu.ids | tail -22 | head -19
