#!/bin/bash
# Debug (inside gpurun): overlapping windows of the Eulerian methods -- parity tests, then the c4 / c5 shard benches with and without
cd ${GRAFT_REPO_ROOT:-/root/repo}
(time timeout 900 python -m pytest tests/test_gpu_scale.py -x -q -k "overlapping or host_forcing" ) > gpurun_out/r04_overlap_tests.log 2>&1
tail -5 gpurun_out/r04_overlap_tests.log
(time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "kwt or golden" ) > gpurun_out/r04_kwt_tests.log 2>&1
tail -5 gpurun_out/r04_kwt_tests.log
run() { echo "=== $*"; env $* 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j.get('roofline') or {}
        print('value %.4g  ms/step %.2f  frac %s launch_us %s launches %s h2d %s %s' % (j['value'], j['ms_per_step'], r.get('frac'), r.get('avg_launch_us'), r.get('launches'), j.get('value_with_h2d'), j.get('value_with_h2d_f64')))
    elif 'rror' in l: print(l.rstrip())
"; }
B="python bench.py --no-cpu-baseline --no-single-step --no-configs"
for c in c4 c5; do
  run MZR_OVERLAP_WINDOWS=0 $B --no-h2d --config $c --steps 6 --warmup 2
  run MZR_OVERLAP_WINDOWS=1 $B --no-h2d --config $c --steps 6 --warmup 2
done
run X=1 $B --no-h2d --config c3 --steps 4 --warmup 3
run X=1 $B --steps 4 --warmup 2
