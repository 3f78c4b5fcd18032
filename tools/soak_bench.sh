#!/bin/bash
# Soak of the driver's bench command: N fresh processes of `python3 bench.py --gpus 1 --steps 20 --warmup 5`
# (CPU baseline left out: it runs after the GPU legs and only costs time), one line per run in
# gpurun_out/<tag>/soak.log: run index, exit code, wall seconds, and the value or the error text.
#   tools/soak_bench.sh <tag> <N> [extra bench arguments]
tag=${1:-soak}; n=${2:-12}; shift 2
out=gpurun_out/$tag; mkdir -p $out
: > $out/soak.log
pass=0
for i in $(seq 1 $n); do
  t0=$(date +%s)
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline "$@" > $out/run_$i.out 2> $out/run_$i.err
  rc=$?
  t1=$(date +%s)
  val=$(python3 - "$out/run_$i.out" <<'EOF'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value=%.4g h2d=%s single=%s frac=%s resc=%s err=%s" % (j.get("value") or 0, j.get("value_with_h2d"), (j.get("single_step") or {}).get("value"),
          (j.get("roofline") or {}).get("frac"), j.get("sweep_rescues_per_xcc"), j.get("error")))
except Exception as e:
    print("no-json")
EOF
)
  [ $rc -eq 0 ] && pass=$((pass+1))
  echo "run $i rc=$rc wall=$((t1 - t0)) $val $(grep -h "ierr=\|MZR STALL" $out/run_$i.err | tail -3 | tr '\n' ' ')" >> $out/soak.log
done
echo "passed $pass / $n" >> $out/soak.log
cat $out/soak.log
