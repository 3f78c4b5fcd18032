#!/bin/bash
# round 5 (inside gpurun): the bench's host-forcing leg with the product build and with a variant build (lib_var/<name>), turn about
#   tools/r05_h2d4.sh <variant> [reps]
cd ${GRAFT_REPO_ROOT:-/root/repo}
v=$1; reps=${2:-3}
for rep in $(seq $reps); do for lib in lib lib_var/$v; do
  MZR_LIB=$PWD/mizuroute_amd/$lib/libmzr_hip.so python bench.py --no-cpu-baseline --no-configs --no-single-step --steps 6 --warmup 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$lib: value %.4g  value_with_h2d %.4g  ratio %.4f  retries %s  error %s' % (j['value'], j['value_with_h2d'], j['value_with_h2d'] / j['value'], j.get('kwt_sweep_retries'), j.get('error')))
"
done; done
