#!/bin/bash
# Debug (inside gpurun): outbox ring of 4 + split 16-lane pass + MC slow-reach blocks
cd ${GRAFT_REPO_ROOT:-/root/repo}
(time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "kwt or golden" ) > gpurun_out/r04_ring_tests.log 2>&1
tail -3 gpurun_out/r04_ring_tests.log
run() { echo "=== $*"; env $* 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j.get('roofline') or {}
        print('value %.4g  ms/step %.2f  frac %s launch_us %s h2d %s %s' % (j['value'] or 0, j['ms_per_step'] or 0, r.get('frac'), r.get('avg_launch_us'), j.get('value_with_h2d'), j.get('value_with_h2d_f64')), j.get('error'))
    elif 'rror' in l: print(l.rstrip())
"; }
B="python bench.py --no-cpu-baseline --no-single-step --no-configs --no-h2d"
run X=1 $B --steps 4 --warmup 2
run X=1 $B --steps 4 --warmup 2
run X=1 $B --config c3 --steps 4 --warmup 3
run MZR_MC_SLOW_MIN=0 $B --config c4 --steps 6 --warmup 3
run MZR_MC_SLOW_MIN=6 $B --config c4 --steps 6 --warmup 3
run MZR_MC_SLOW_MIN=4 $B --config c4 --steps 6 --warmup 3
python tools/h2d_probe.py 2>&1 | grep -v amdgpu
(time timeout 900 python -m pytest tests/test_gpu_scale.py -x -q -k "overlapping or c4_shard or config_parity or mc_substep" ) > gpurun_out/r04_ring_tests2.log 2>&1
tail -3 gpurun_out/r04_ring_tests2.log
