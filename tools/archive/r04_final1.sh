#!/bin/bash
# Debug (inside gpurun): basin2reach per XCD, 20-window runs with / without the events, MC counters at shard size
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
(time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or hillslope or basin or remap or step_by_step") > gpurun_out/r04_final1_tests.log 2>&1
tail -3 gpurun_out/r04_final1_tests.log
run() { echo "=== $*"; env $* 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j.get('roofline') or {}
        print('value %.4g  ms/step %.2f frac %s launch_ms %s' % (j['value'] or 0, j['ms_per_step'] or 0, r.get('frac'), (r.get('avg_launch_us') or 0) / 1e3), j.get('error'))
    elif 'rror' in l: print(l.rstrip())
"; }
B="python bench.py --no-cpu-baseline --no-single-step --no-configs --no-h2d"
run X=1 $B --steps 20 --warmup 5
run X=1 $B --steps 20 --warmup 5 --no-roofline
run X=1 $B --steps 4 --warmup 2
o=gpurun_out/final1_stats; rm -rf $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o -o k -- python bench.py --no-cpu-baseline --no-single-step --no-configs --no-h2d --no-roofline --steps 2 --warmup 2 > gpurun_out/final1_stats.log 2>&1
grep -E "k_basin2reach|k_hillslope_out|k_kwt_window_init|k_sweep_kwt|k_accum" $o/*/k_kernel_stats.csv $o/k_kernel_stats.csv 2>/dev/null | cut -c1-200
find $o -name "*kernel_trace.csv" -delete
NR=625000 WW=3072 METHODS=MC bash tools/pmc.sh mc625 "python tools/bench_methods.py" 2>&1 | grep "k_stage" | cut -c1-1500
