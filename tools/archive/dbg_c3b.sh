#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1; do
  python bench.py --steps 3 --warmup 1 --no-h2d --no-single-step --configs c3 2> gpurun_out/dbg_c3b_$i.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); c = (j.get('configs') or {}).get('c3') or {}
        print('run value %.4g c3 %s err %s' % (j['value'] or 0, c.get('value'), c.get('error')))
"
done
