#!/bin/bash
# round 6, last A/B: the C / B class cut at c2 (11 against 13), k_basin2reach with 2 / 8 steps per lane (library variants)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_last; mkdir -p $O
run() {  # name, env...
  n=$1; shift
  ( for kv in "$@"; do export "$kv"; done; timeout 600 python bench.py --config c2 --steps 6 --warmup 1 --no-cpu-baseline --no-h2d --no-single-step --no-configs > $O/$n.out 2> $O/$n.err )
  python - "$O/$n.out" "$n" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=j.get("roofline") or {}
    print(sys.argv[2], "value %.4g"%j["value"], "ms/window %.1f"%j["ms_per_step"], "launch us %.1f"%(r.get("avg_launch_us") or 0), "err", j.get("error"))
except Exception as e:
    print(sys.argv[2], "no line:", e)
PY
}
run base1; run c11a MZR_KWT_CLASSC_MAX=11; run base2; run c11b MZR_KWT_CLASSC_MAX=11
run bt2a MZR_LIB=$PWD/mizuroute_amd/lib_var/bt2/libmzr_hip.so; run bt8a MZR_LIB=$PWD/mizuroute_amd/lib_var/bt8/libmzr_hip.so; run base3; run bt2b MZR_LIB=$PWD/mizuroute_amd/lib_var/bt2/libmzr_hip.so; run bt8b MZR_LIB=$PWD/mizuroute_amd/lib_var/bt8/libmzr_hip.so
