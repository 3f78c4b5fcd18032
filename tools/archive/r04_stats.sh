#!/bin/bash
# Round end (inside gpurun): rocprofv3 kernel stats of the driver's c2 legs on the round's last build
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
o=gpurun_out/r04e_stats; rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o -o k -- python bench.py --gpus 1 --steps 8 --warmup 5 --no-cpu-baseline --no-h2d --no-single-step --no-configs > $o/bench.json 2> $o/bench.err
tail -c 300 $o/bench.json
f=$(find $o -name "k_kernel_stats.csv" | head -1); cp $f gpurun_out/r04e_kernel_stats.csv; head -8 gpurun_out/r04e_kernel_stats.csv | cut -c1-160
find $o -name "*kernel_trace.csv" -delete
