#!/bin/bash
# Debug (inside gpurun): steps per launch at 100 k reaches and windows of 16 384 steps (launch-bound regime)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for kb in 1 4 8 16; do
  echo "KB=$kb 100k W=16384: $(MZR_STEP_BLOCK=$kb NR=100000 WW=16384 METHODS=IRF,KW,DW python tools/bench_methods.py 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.readline()); print({k: '%.3g' % v['reach_steps_per_s'] for k, v in j.items()})")"
done
for kb in 1 2 4; do
  echo "KB=$kb 625k W=4096: $(MZR_STEP_BLOCK=$kb NR=625000 WW=4096 METHODS=IRF,KW,DW python tools/bench_methods.py 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.readline()); print({k: '%.3g' % v['reach_steps_per_s'] for k, v in j.items()})")"
done
