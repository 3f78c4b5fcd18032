#!/bin/bash
# Debug (inside gpurun): A/B of environment switches on the default bench.  usage: tools/envab.sh "VAR=val" "VAR=val VAR2=val" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "=== $*"; env $* python bench.py --no-cpu-baseline --no-h2d --no-single-step --no-roofline --steps 3 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('value %.4g  ms/step %.2f' % (j['value'], j['ms_per_step']))
    elif 'rror' in l: print(l.rstrip())
"; }
for v in "$@"; do run $v; done
