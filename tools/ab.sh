#!/bin/bash
# Debug (run inside gpurun): A/B of build variants.  usage: tools/ab.sh "<bench args>" "<EXTRA flags 1>" "<EXTRA flags 2>" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/ab
args=$1; shift
i=0
for v in "$@"; do
  make -C mizuroute_amd/csrc clean >/dev/null; make -C mizuroute_amd/csrc all EXTRA="$v" -j8 > gpurun_out/ab/build_$i.log 2>&1 || { echo "BUILD FAILED [$v]"; tail -5 gpurun_out/ab/build_$i.log; }
  echo "=== [$v]"
  python bench.py --no-cpu-baseline $args 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j.get('roofline') or {}
        print('value %.4g  ms/step %.2f  frac %s  launch_us %s' % (j['value'], j['ms_per_step'], r.get('frac'), r.get('avg_launch_us')))
    elif 'rror' in l: print(l.rstrip())
"
  i=$((i+1))
done
make -C mizuroute_amd/csrc clean >/dev/null; make -C mizuroute_amd/csrc all -j8 >/dev/null 2>&1
