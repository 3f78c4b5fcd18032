#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "taken_back or gave_up or stall_beside or times_itself or operating_point" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden" 2>&1 | tail -5
