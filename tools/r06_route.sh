#!/bin/bash
# round 6: the Eulerian stage kernels' head (topology words in one round trip, two upstream rows in flight): parity, then c5 / c4 shards, base library against the new one
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_route; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "golden or eulerian or mc_ or channel_table or partitioned or route_sweep or overlapping or lakes or tiny or methods" > $O/parity.log 2>&1; echo "parity rc $?" >> $O/parity.log
tail -n 3 $O/parity.log
for c in ${SHARDS:-c5 c4}; do
  for v in base new base new; do
    ( [ $v = base ] && export MZR_LIB=$PWD/mizuroute_amd/lib_var/base/libmzr_hip.so; timeout 900 python bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline --no-h2d --no-single-step --no-configs > $O/$c.$v.out 2> $O/$c.$v.err )
    python - "$O/$c.$v.out" $v <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=j.get("roofline") or {}
    print(sys.argv[2], j["config"]["baseline_config"], "value %.4g"%j["value"], "ms/window %.1f"%j["ms_per_step"], r.get("kernel"), "launch us %.1f"%(r.get("avg_launch_us") or 0), "frac %.4f"%(r.get("frac") or 0), "err", j.get("error"))
except Exception as e:
    print("no line:", e)
PY
  done
done
