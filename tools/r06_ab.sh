#!/bin/bash
# round 6: A/B of library builds (tools/build_variant.sh <name> "<flags>" here, then on the GPU box:)
#   VARIANTS="product ring8 ..." [CONFIGS="c2 c3"] [STEPS=4] tools/r06_ab.sh
# the sweep's own device clock per window at the two KWT operating points, per variant, turn about (REPS times)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_ab; mkdir -p $O
for rep in $(seq 1 ${REPS:-1}); do
for v in ${VARIANTS:-product}; do
  lib=mizuroute_amd/lib_var/$v/libmzr_hip.so; [ "$v" = product ] && lib=mizuroute_amd/lib/libmzr_hip.so
  [ -f $lib ] || { echo "$v: no $lib"; continue; }
  for c in ${CONFIGS:-c2 c3}; do
    MZR_LIB=$PWD/$lib timeout 900 python bench.py --config $c --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline --no-h2d --no-single-step --no-configs $ARGS > $O/${v}_$c.out 2> $O/${v}_$c.err
    python - "$O/${v}_$c.out" "$v" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=j.get("roofline") or {}
    print("%-12s"%sys.argv[2], j["config"]["baseline_config"], "value %.4g"%j["value"], "ms/window %.1f"%j["ms_per_step"], "launch ms %.1f (%.1f..%.1f)"%(r.get("avg_launch_us",0)/1e3, r.get("min_launch_us",0)/1e3, r.get("max_launch_us",0)/1e3), "frac %.4f"%r.get("frac",0), "err", j.get("error"))
except Exception as e:
    print(sys.argv[2], "no line:", e)
PY
  done
done
done
