#!/bin/bash
# Debug (inside gpurun): Muskingum-Cunge throughput and parity against the oracle for several tail tolerances
cd ${GRAFT_REPO_ROOT:-/root/repo}
for tol in ${TOLS:-0 1e-12 1e-11 1e-10}; do
  echo "=== MZR_MC_TAIL_TOL=$tol"
  MZR_MC_TAIL_TOL=$tol METHODS=MC python tools/bench_methods.py 2>&1 | tail -1
  MZR_MC_TAIL_TOL=$tol python -m pytest tests/test_gpu_scale.py -q -k "parity_50k and c4" -s 2>&1 | grep -E "c4_irf_mc 4|passed|failed"
done
