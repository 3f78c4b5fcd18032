#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fresh_case or traffic or lane_classes" 2>&1 | tail -5
python bench.py --no-cpu-baseline --no-h2d --no-single-step --no-configs --steps 4 --warmup 2 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('value %.4g frac %.4f launch_ms %.1f bytes/rs %.2f particles %.2f sweep %s' % (j['value'], r['frac'], r['avg_launch_us'] / 1e3, r['bytes_per_reach_step'], r['particles_per_routed_reach'], j['config']['kwt_sweep']))"
