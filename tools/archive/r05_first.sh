#!/bin/bash
# round 5, first GPU run of the blocked sweep: parity first, then speed of both flavours
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not bench_two_ranks" 2>&1 | tail -25 ) > gpurun_out/r05_first_tests.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "not full_size" 2>&1 | tail -25 ) >> gpurun_out/r05_first_tests.log 2>&1
B="python bench.py --no-cpu-baseline --no-h2d --no-single-step --no-configs"
for env in "X=1" "MZR_KWT_KBLK_RUN=1"; do
  echo "=== $env $B --steps 4 --warmup 2" >> gpurun_out/r05_first_bench.log
  ( env $env timeout 600 $B --steps 4 --warmup 2 2>&1 | tail -3 ) >> gpurun_out/r05_first_bench.log
  echo "=== $env $B --config c3 --steps 4 --warmup 3" >> gpurun_out/r05_first_bench.log
  ( env $env timeout 600 $B --config c3 --steps 4 --warmup 3 2>&1 | tail -3 ) >> gpurun_out/r05_first_bench.log
done
tail -5 gpurun_out/r05_first_tests.log
python - <<'PY'
import json
for l in open("gpurun_out/r05_first_bench.log"):
    if l.startswith("==="): print(l.strip())
    elif l.startswith("{"):
        j = json.loads(l); r = j.get("roofline") or {}
        print("value %.4g ms/step %.1f frac %s launch_us %s err %s" % (j["value"] or 0, j["ms_per_step"] or 0, r.get("frac"), r.get("avg_launch_us"), j.get("error")))
PY
