#!/bin/bash
# Round profile bundle (run inside gpurun): bench line, rocprofv3 kernel stats, HBM PMC passes.
# usage: tools_profile.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/$tag
python bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
tail -c 3000 gpurun_out/$tag/bench.json
ARGS="--no-cpu-baseline --no-roofline --window 8192 --steps 1 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/stats -o k -- python bench.py $ARGS > gpurun_out/$tag/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$tag/pmc_fetch -o k -- python bench.py $ARGS > gpurun_out/$tag/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/$tag/pmc_write -o k -- python bench.py $ARGS > gpurun_out/$tag/pmc_write.log 2>&1
find gpurun_out/$tag -name "*.csv" | head -20
# keep only what is small enough to travel back
find gpurun_out/$tag -name "*kernel_trace.csv" -size +20M -delete
find gpurun_out/$tag -name "*counter_collection.csv" -size +30M -exec sh -c 'head -c 30000000 "$1" > "$1.part"; rm "$1"' _ {} \;
