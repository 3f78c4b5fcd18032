#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/ab
i=0
for v in "" "-DMZR_STAGE_WG=64"; do
  make -C mizuroute_amd/csrc clean >/dev/null; make -C mizuroute_amd/csrc all EXTRA="$v" -j8 > gpurun_out/ab/build_f$i.log 2>&1 || { echo "BUILD FAILED [$v]"; tail -5 gpurun_out/ab/build_f$i.log; }; i=$((i+1))
  echo "##### build [$v]"
  for lp in 1 0; do
    echo "lane perm $lp"
    MZR_LANE_PERM=$lp METHODS=IRF,KW,MC,DW python tools/bench_methods.py 2>&1 | tail -1 | cut -c1-400
    MZR_LANE_PERM=$lp NR=625000 WW=3072 METHODS=IRF,MC python tools/bench_methods.py 2>&1 | tail -1
  done
done
make -C mizuroute_amd/csrc clean >/dev/null; make -C mizuroute_amd/csrc all -j8 >/dev/null 2>&1
for c in c4 c5; do python bench.py --no-cpu-baseline --no-single-step --no-configs --no-h2d --config $c --steps 4 --warmup 3 2>&1 | tail -1 | cut -c1-200; done
