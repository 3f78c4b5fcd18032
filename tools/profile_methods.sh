#!/bin/bash
# Run inside gpurun: the Eulerian methods on the 100 k-reach benchmark network (tools/bench_methods.py, windows of 1024 steps,
# one launch per stage): throughput, rocprofv3 kernel stats and SQ counters (one --pmc pass per group).
# usage: tools/profile_methods.sh <tag>     then, here: python tools/summarize_methods.py <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
o=gpurun_out/${tag}_methods; rm -rf $o; mkdir -p $o
export METHODS=IRF,KW,MC,DW,SUM
python tools/bench_methods.py > $o/bench_methods.json 2> $o/bench.err
tail -1 $o/bench_methods.json
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o k -- python tools/bench_methods.py > $o/stats.log 2>&1
bash tools/pmc.sh ${tag}_methods "python tools/bench_methods.py" > $o/pmc.log 2>&1
cp gpurun_out/${tag}_methods_pmc.json $o/ 2>/dev/null
find $o -name "*kernel_trace.csv" -size +20M -delete
ls $o
