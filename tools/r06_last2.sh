#!/bin/bash
# round 6: Muskingum-Cunge, the blocks of the reaches with many sub-steps at wave priority 3 (library variant lib_var/mcprio) against the product, c4 shard
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_last2; mkdir -p $O
for v in base mcprio base mcprio; do
  ( [ $v = mcprio ] && export MZR_LIB=$PWD/mizuroute_amd/lib_var/mcprio/libmzr_hip.so; timeout 900 python bench.py --config c4 --steps 4 --warmup 1 --no-cpu-baseline --no-h2d --no-single-step --no-configs > $O/$v.out 2> $O/$v.err )
  python - "$O/$v.out" $v <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=j.get("roofline") or {}
    print(sys.argv[2], "value %.4g"%j["value"], "ms/window %.1f"%j["ms_per_step"], r.get("kernel"), "launch us %.1f"%(r.get("avg_launch_us") or 0), "err", j.get("error"))
except Exception as e:
    print(sys.argv[2], "no line:", e)
PY
done
