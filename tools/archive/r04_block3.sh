#!/bin/bash
# Debug (inside gpurun): blocked / unblocked instantiations: speed at 100 k (windows of 16 384 and 1024) and the c4 / c5 legs
cd ${GRAFT_REPO_ROOT:-/root/repo}
MZR_STEP_BLOCK=3 timeout 600 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "overlapping" 2>&1 | grep -E "passed|failed|error" | tail -2
for kb in 1 x; do
  [ $kb = x ] && unset MZR_STEP_BLOCK || export MZR_STEP_BLOCK=$kb
  echo "KB=$kb 100k W=16384: $(NR=100000 WW=16384 METHODS=IRF,KW,DW python tools/bench_methods.py 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.readline()); print({k: '%.3g' % v['reach_steps_per_s'] for k, v in j.items()})")"
done
unset MZR_STEP_BLOCK
echo "default 100k W=1024: $(NR=100000 WW=1024 METHODS=IRF,KW,MC,DW python tools/bench_methods.py 2>/dev/null | tail -1)"
python bench.py --steps 3 --warmup 1 --no-h2d --no-single-step --no-cpu-baseline --configs c4,c5 2> gpurun_out/dbg_c4.err > gpurun_out/dbg_c4.json
python - <<PY
import json
for l in open("gpurun_out/dbg_c4.json"):
    if l.startswith("{"):
        j = json.loads(l)
        for k, c in j["configs"].items():
            m8 = c.get("model_8gpu") or {}
            print(k, c.get("value"), c.get("error"), {x: round(m8[x], 4) for x in m8 if x.endswith("_s")}, (c.get("roofline") or {}).get("frac"))
PY
