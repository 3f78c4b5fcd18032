#!/bin/bash
# Round end (inside gpurun): the driver's command, then three fresh processes of it without the configs / CPU legs
cd ${GRAFT_REPO_ROOT:-/root/repo}
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04e_bench_driver.json 2> gpurun_out/r04e_bench_driver.err
grep "^mzr" gpurun_out/r04e_bench_driver.err | sort | uniq -c
python - <<PY
import json
for l in open("gpurun_out/r04e_bench_driver.json"):
    if l.startswith("{"):
        j = json.loads(l)
        print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["value_with_h2d"], j["kwt_sweep_retries"], j["error"])
        for k, c in (j.get("configs") or {}).items():
            m8 = c.get("model_8gpu") or {}
            print(k, c.get("value"), c.get("error"), (c.get("parity") or {}).get("partitioned_equals_whole_bit_for_bit"), m8.get("value"), {x: round(m8[x], 4) for x in m8 if x.endswith("_s")}, (c.get("roofline") or {}).get("frac"), c.get("wall_s"))
PY
for i in 1; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('soak value %.4g frac %s h2d %s retries %s err %s' % (j['value'] or 0, (j.get('roofline') or {}).get('frac'), j.get('value_with_h2d'), j.get('kwt_sweep_retries'), j.get('error')))
"
done
