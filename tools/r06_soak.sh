#!/bin/bash
# round 6: soak of the c2 bench with the automatic watchdog (1.8 s without progress at c2): no retry may happen in a healthy run
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_soak; mkdir -p $O
for i in $(seq 1 ${RUNS:-6}); do
  python bench.py --steps ${STEPS:-12} --warmup 2 --no-cpu-baseline --no-single-step --no-configs > $O/$i.out 2> $O/$i.err
  python - $O/$i.out $i <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=j.get("roofline") or {}
print("run", sys.argv[2], "value %.4g"%(j["value"] or 0), "h2d %.4g"%(j.get("value_with_h2d") or 0), "launch ms %.1f..%.1f"%((r.get("min_launch_us") or 0)/1e3,(r.get("max_launch_us") or 0)/1e3), "retries", j.get("kwt_sweep_retries"), "err", j.get("error"))
PY
  grep -h "gave up" $O/$i.err | head -2
done
