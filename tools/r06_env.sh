#!/bin/bash
# round 6: one bench configuration under several environments.  CONFIG=c4 ENVS="A=1;B=2 C=3" (sets separated by spaces, variables inside a set by ';')
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_env; mkdir -p $O
i=0
for e in "" $ENVS; do
  i=$((i+1))
  ( IFS=';'; for kv in $e; do export "$kv"; done; timeout 900 python bench.py --config ${CONFIG:-c4} --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline --no-h2d --no-single-step --no-configs $ARGS > $O/$i.out 2> $O/$i.err )
  python - "$O/$i.out" "[$e]" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=j.get("roofline") or {}
    print(sys.argv[2], j["config"]["baseline_config"], "value %.4g"%(j["value"] or 0), "ms/window %.1f"%(j["ms_per_step"] or 0), r.get("kernel"), "launch us %.1f"%(r.get("avg_launch_us") or 0), "frac %.4f"%(r.get("frac") or 0), "err", j.get("error"))
except Exception as e:
    print(sys.argv[2], "no line:", e)
PY
done
