#!/bin/bash
# round 5 (inside gpurun): loopback timing of a full-size configuration under environment settings ("A=1,B=2" per run; "-" = none)
cd ${GRAFT_REPO_ROOT:-/root/repo}
c=${CONFIG:-c4}
for spec in "$@"; do
  envs=$(echo "$spec" | tr ',' ' '); [ "$spec" = "-" ] && envs="X=1"
  env $envs python bench.py --loopback --config $c --steps ${STEPS:-3} --no-cpu-baseline ${ARGS:-} > gpurun_out/lb_tmp.json 2> gpurun_out/lb_tmp.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("gpurun_out/lb_tmp.json") if l.startswith("{")][-1])
    print("=== [$spec] $c value %.4g err %s same %s" % (j["value"], j["error"], j["parity"]["partitioned_equals_whole_bit_for_bit"]))
    print({k:[round(x,3) for x in v["s_per_window"]] for k,v in j["config"]["domains"].items()})
    m=j["model_8gpu"]; print({k:(round(v,4) if isinstance(v,float) else v) for k,v in m.items() if k!="what"})
except Exception as e:
    print("=== [$spec] failed", e); print(open("gpurun_out/lb_tmp.err").read()[-600:])
PY
done
