#!/bin/bash
# Debug (inside gpurun): class-B cut and capacity after the LDS thinning
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/ab
run() { echo "=== $*"; env $* 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('value %.4g  ms/step %.2f' % (j['value'] or 0, j['ms_per_step'] or 0), j.get('error'))
    elif 'rror' in l: print(l.rstrip())
"; }
B="python bench.py --no-cpu-baseline --no-single-step --no-configs --no-h2d --no-roofline"
i=0
build() { make -C mizuroute_amd/csrc clean >/dev/null; make -C mizuroute_amd/csrc all EXTRA="$1" -j8 > gpurun_out/ab/build_$i.log 2>&1 || { echo "BUILD FAILED [$1]"; tail -5 gpurun_out/ab/build_$i.log; }; i=$((i+1)); echo "##### build [$1]"; }
build ""
run X=1 $B --steps 3 --warmup 2
run MZR_KWT_CLASSB_MAX=20 $B --steps 3 --warmup 2
run MZR_KWT_CLASSB_MAX=28 $B --steps 3 --warmup 2
run X=1 $B --config c3 --steps 4 --warmup 3
run MZR_KWT_CLASSB_MAX=24 $B --config c3 --steps 4 --warmup 3
run MZR_KWT_CLASSB_MAX=30 $B --config c3 --steps 4 --warmup 3
build "-DMZR_KWT_POOL=320 -DMZR_KWT_OCC=4 -DMZR_KWT_KTB=5"
run MZR_KWT_CLASSB_MAX=28 $B --steps 3 --warmup 2
run MZR_KWT_CLASSB_MAX=34 $B --steps 3 --warmup 2
run MZR_KWT_CLASSB_MAX=28 $B --config c3 --steps 4 --warmup 3
run MZR_KWT_CLASSB_MAX=34 $B --config c3 --steps 4 --warmup 3
make -C mizuroute_amd/csrc clean >/dev/null; make -C mizuroute_amd/csrc all -j8 >/dev/null 2>&1
