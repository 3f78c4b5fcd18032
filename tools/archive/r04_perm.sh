#!/bin/bash
# Debug (inside gpurun): lane permutation of the Eulerian stage kernels (IRF by taps, MC by sub-steps)
cd ${GRAFT_REPO_ROOT:-/root/repo}
(time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -k "not kwt_vs and not operating_point and not full_size and not c3_shard" ) > gpurun_out/r04_perm_tests.log 2>&1
tail -4 gpurun_out/r04_perm_tests.log
METHODS=IRF,MC python tools/bench_methods.py 2>&1 | tail -1
MZR_LANE_PERM=0 METHODS=IRF,MC python tools/bench_methods.py 2>&1 | tail -1
run() { echo "=== $*"; env $* 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j.get('roofline') or {}
        print('value %.4g  ms/step %.2f frac %s launch_us %s' % (j['value'] or 0, j['ms_per_step'] or 0, r.get('frac'), r.get('avg_launch_us')), j.get('error'))
    elif 'rror' in l: print(l.rstrip())
"; }
B="python bench.py --no-cpu-baseline --no-single-step --no-configs --no-h2d"
run MZR_LANE_PERM=0 $B --config c4 --steps 4 --warmup 3
run MZR_LANE_PERM=1 $B --config c4 --steps 4 --warmup 3
