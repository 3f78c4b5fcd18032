#!/bin/bash
# round 5 (inside gpurun): the bench's host-forcing leg with and without the sweep launch's head start over the next window's copy
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for hs in 1 0; do
  MZR_H2D_HEAD_START=$hs python bench.py --no-cpu-baseline --no-configs --no-single-step --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('head start $hs: value %.4g  value_with_h2d %.4g  ratio %.4f' % (j['value'], j['value_with_h2d'], j['value_with_h2d'] / j['value']))
"
done; done
