#!/bin/bash
# round 6: the KWT sweep's two operating points (c2: 100 k reaches, windows of 16 384; c3 shard: 375 k, windows of 8 192), the sweep's
# own device clock per window.  TAG=name [PARITY=1] [CONFIGS="c2 c3"] tools/r06_perf.sh
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${TAG:-x}; O=gpurun_out/r06_perf_$TAG; mkdir -p $O
if [ -n "$PARITY" ]; then
  timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "golden or kwt or lane_classes or fresh or rare or confluence" > $O/parity.log 2>&1; echo "parity rc $?" >> $O/parity.log
  tail -n 3 $O/parity.log
fi
for c in ${CONFIGS:-c2 c3}; do
  timeout 900 python bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline --no-h2d --no-single-step --no-configs > $O/$c.out 2> $O/$c.err
  cp bench_detail.json $O/${c}_detail.json 2>/dev/null
  python - "$O/$c.out" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=j.get("roofline") or {}
    print(j["config"]["baseline_config"], "value %.4g"%j["value"], "ms/window %.1f"%j["ms_per_step"], "launch ms %.1f (%.1f..%.1f)"%(r.get("avg_launch_us",0)/1e3, r.get("min_launch_us",0)/1e3, r.get("max_launch_us",0)/1e3), "frac %.4f"%r.get("frac",0), "B/rs %.1f"%r.get("bytes_per_reach_step",0), "err", j.get("error"))
except Exception as e:
    print("no line:", e)
PY
done
