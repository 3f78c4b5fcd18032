#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not bench_two_ranks" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "not full_size" 2>&1 | tail -3
tools/r05_env.sh -
