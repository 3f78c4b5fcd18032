#!/bin/bash
# Debug (inside gpurun): steps per launch of the Eulerian stage kernels (MZR_STEP_BLOCK): parity subset, then speed per method
cd ${GRAFT_REPO_ROOT:-/root/repo}
MZR_STEP_BLOCK=3 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -m gpu -k "not kwt and not c3 and not c4 and not c5 and not soak" > gpurun_out/block_parity.log 2>&1
grep -E "passed|failed|error" gpurun_out/block_parity.log | tail -3
for kb in 1 2 4 8; do
  echo "KB=$kb 625k W=2048: $(MZR_STEP_BLOCK=$kb NR=625000 WW=2048 METHODS=IRF,MC,DW python tools/bench_methods.py 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.readline()); print({k: '%.3g' % v['reach_steps_per_s'] for k, v in j.items()})")"
done
for kb in 1 4; do
  echo "KB=$kb 100k W=4096: $(MZR_STEP_BLOCK=$kb NR=100000 WW=4096 METHODS=IRF,KW,MC,DW python tools/bench_methods.py 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.readline()); print({k: '%.3g' % v['reach_steps_per_s'] for k, v in j.items()})")"
done
