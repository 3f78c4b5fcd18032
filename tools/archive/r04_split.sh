#!/bin/bash
# Debug (inside gpurun): the split 16-lane pass of the KWT sweep -- parity tests, then c2 / c3 benches
cd ${GRAFT_REPO_ROOT:-/root/repo}
(time timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q ) > gpurun_out/r04_split_parity.log 2>&1
tail -5 gpurun_out/r04_split_parity.log
(time timeout 1500 python -m pytest tests/test_gpu_scale.py -x -q -k "c3_shard or config_parity or operating_point") > gpurun_out/r04_split_scale.log 2>&1
tail -5 gpurun_out/r04_split_scale.log
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
