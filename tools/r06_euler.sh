#!/bin/bash
# round 6: the Eulerian side -- parity tests that touch MC / KW / DW, partitioned domains and boundary records; loopback of c4 / c5
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_euler; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "golden or eulerian or mc_ or channel_table or partitioned or route_sweep or overlapping or lakes or boundary or record or comm" > $O/parity.log 2>&1; echo "parity rc $?" >> $O/parity.log
tail -n 4 $O/parity.log
for c in $LOOPBACK; do
  timeout 1200 python bench.py --loopback --config $c --no-cpu-baseline --steps 5 > $O/lb_$c.json 2> $O/lb_$c.err; echo "loopback $c rc $?"
  python - $O/lb_$c.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    m=j["model_8gpu"]; d=j["config"]["domains"]
    print("value %.4g"%j["value"], "W", j["config"]["window_steps"], "parity", j["parity"]["partitioned_equals_whole_bit_for_bit"], "model8 %.4g"%m["value"], "slowest trib %.4f rank0 sbs %.4f ratio %.3f"%(m["slowest_tributary_s"], m["rank0_side_by_side_s"], m["rank0_side_by_side_s"]/m["slowest_tributary_s"]), "record bytes", d["main"].get("record_bytes_per_window"), "main s", [round(x,4) for x in d["main"]["s_per_window"][-3:]], "trib1", [round(x,4) for x in d["trib1"]["s_per_window"][1:4]], "err", j.get("error"))
except Exception as e:
    print("no line", e)
PY
done
for c in $SHARDS; do
  timeout 900 python bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline --no-h2d --no-single-step --no-configs > $O/$c.out 2> $O/$c.err
  python - "$O/$c.out" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=j.get("roofline") or {}
    print(j["config"]["baseline_config"], "value %.4g"%j["value"], "ms/window %.1f"%j["ms_per_step"], r.get("kernel"), "launch us %.1f"%(r.get("avg_launch_us") or 0), "frac %.4f"%(r.get("frac") or 0), "err", j.get("error"))
except Exception as e:
    print("no line:", e)
PY
done
