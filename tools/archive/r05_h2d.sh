#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for e in X=1 MZR_H2D_UNCACHED=1; do
  echo "=== $e"; env $e python bench.py --no-cpu-baseline --no-single-step --no-configs --no-roofline --steps 6 --warmup 2 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('resident %.4g  h2d f32 %s  h2d f64 %s' % (j['value'], j['value_with_h2d'], j['value_with_h2d_f64']))"
done
