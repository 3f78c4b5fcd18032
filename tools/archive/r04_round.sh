#!/bin/bash
# Debug (inside gpurun): after the KWT / overlap changes -- h2d pipeline probe, c3 loop-back (rank 0 with priority), c4 / c5 shards, 100 k methods
cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/h2d_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r04_h2d_probe.txt
cat gpurun_out/r04_h2d_probe.txt
run() { echo "=== $*"; env $* 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j.get('roofline') or {}
        print('value %.4g  ms/step %.2f frac %s' % (j['value'] or 0, j['ms_per_step'] or 0, r.get('frac')), j.get('error'))
        if j.get('model_8gpu'): print(json.dumps(j['model_8gpu'])); print(json.dumps({k: v.get('s_per_window') for k, v in j['config']['domains'].items()}))
    elif 'rror' in l: print(l.rstrip())
"; }
B="python bench.py --no-cpu-baseline --no-single-step --no-configs --no-h2d"
run X=1 $B --config c4 --steps 4 --warmup 2
run X=1 $B --config c5 --steps 6 --warmup 2
run X=1 python bench.py --loopback --config c3 --partitions 8 --steps 6 --no-cpu-baseline
METHODS=IRF,KW,MC,DW,SUM python tools/bench_methods.py 2>&1 | tail -1
MZR_OVERLAP_WINDOWS=0 METHODS=IRF,KW,MC,DW,SUM python tools/bench_methods.py 2>&1 | tail -1
