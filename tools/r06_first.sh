#!/bin/bash
# round 6, first GPU call: the changed tests, the plain `--gpus 8` command on a one-GPU box, a short default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06_first
O=gpurun_out/r06_first
timeout 900 python -m pytest tests/test_gpu_scale.py -q -m gpu -k "gave_up or queue_of_windows" > $O/retry.log 2>&1; echo "retry rc $?" >> $O/retry.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "two_ranks_equal_one_rank" > $O/ranks.log 2>&1; echo "ranks rc $?" >> $O/ranks.log
python bench.py --gpus 8 --steps 1 --warmup 0 > $O/plain8.out 2> $O/plain8.err; echo "plain8 rc $?" >> $O/plain8.err
timeout 1500 python bench.py --steps 4 --warmup 1 --configs c5 > $O/bench.out 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
tail -c 4200 $O/bench.out | tail -1 | wc -c >> $O/bench.err
cp bench_detail.json $O/ 2>/dev/null
tail -3 $O/retry.log $O/ranks.log $O/plain8.err; tail -1 $O/bench.out
