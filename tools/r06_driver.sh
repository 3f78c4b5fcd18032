#!/bin/bash
# round 6: the driver's own commands -- the GPU suite, smoke(), bench.py as the driver runs it
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_driver; mkdir -p $O
if [ -z "$NOSUITE" ]; then
  ( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $O/suite.log 2>&1; echo "suite rc $?" >> $O/suite.log; tail -n 6 $O/suite.log
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -n 1 $O/smoke.log
fi
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.out 2> $O/bench.err; echo "bench rc $?"
cp bench_detail.json $O/
tail -n 1 $O/bench.out | wc -c; tail -n 1 $O/bench.out; tail -n 4 $O/bench.err
