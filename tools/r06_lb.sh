#!/bin/bash
# round 6: loopback of one full-size configuration, the side-by-side leg of rank 0 window by window.  CONFIG=c3 [REPS=2] tools/r06_lb.sh
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_lb; mkdir -p $O
for i in $(seq 1 ${REPS:-1}); do
MZR_LIB=${MZR_LIB:-} python bench.py --loopback --config ${CONFIG:-c3} --no-cpu-baseline --steps 5 > $O/lb.json 2> $O/lb.err
python - $O/lb.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
m=j['model_8gpu']; d=j['config']['domains']; r=d['rank0_side_by_side']
print('value %.4g'%j['value'], 'W', j['config']['window_steps'], 'model8 %.4g'%m['value'], 'slowest %.4f sbs %.4f'%(m['slowest_tributary_s'], m['rank0_side_by_side_s']))
print(' sbs ', [round(x,3) for x in r['s_per_window']])
print(' trib', [round(x,3) for x in r.get('s_until_the_tributary_window_is_done',[])])
print(' main sweep arrived/joined', r.get('mainstem_sweep_wavefronts_arrived_joined'))
PY
grep -c 'gave up' $O/lb.err
done
