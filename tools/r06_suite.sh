#!/bin/bash
# round 6: the whole GPU suite as the driver runs it, then (optional) the loopback of one full-size configuration
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_suite; mkdir -p $O
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $O/suite.log 2>&1; echo "suite rc $?" >> $O/suite.log
tail -n 8 $O/suite.log
for c in $LOOPBACK; do
  timeout 1200 python bench.py --loopback --config $c --no-cpu-baseline --steps 5 > $O/lb_$c.json 2> $O/lb_$c.err; echo "loopback $c rc $?"
  python - $O/lb_$c.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    m=j["model_8gpu"]; d=j["config"]["domains"]
    print("value %.4g"%j["value"], "W", j["config"]["window_steps"], "parity", j["parity"]["partitioned_equals_whole_bit_for_bit"], "model8 %.4g"%m["value"], "slowest trib %.4f rank0 sbs %.4f ratio %.3f"%(m["slowest_tributary_s"], m["rank0_side_by_side_s"], m["rank0_side_by_side_s"]/m["slowest_tributary_s"]), "record bytes", d["main"].get("record_bytes_per_window"), "main s", d["main"]["s_per_window"][-3:], "err", j.get("error"))
except Exception as e:
    print("no line", e)
PY
done
