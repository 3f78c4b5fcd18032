#!/bin/bash
# Debug (inside gpurun): Eulerian stage kernels as one-wavefront workgroups placed per XCD, FULL = false really used
cd ${GRAFT_REPO_ROOT:-/root/repo}
(time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -k "not kwt_vs and not operating_point and not full_size and not c3_shard and not golden and not star and not rare") > gpurun_out/r04_euler2_tests.log 2>&1
tail -3 gpurun_out/r04_euler2_tests.log
METHODS=IRF,KW,MC,DW,SUM python tools/bench_methods.py 2>&1 | tail -1
NR=625000 WW=3072 METHODS=IRF,MC python tools/bench_methods.py 2>&1 | tail -1
NR=375000 WW=2048 METHODS=DW python tools/bench_methods.py 2>&1 | tail -1
for c in c4 c5; do python bench.py --no-cpu-baseline --no-single-step --no-configs --no-h2d --config $c --steps 4 --warmup 3 2>&1 | tail -1 | cut -c1-240; done
