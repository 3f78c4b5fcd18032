#!/bin/bash
# round 5 (inside gpurun): PC sampling of the KWT sweep (rocprofv3 beta), aggregated by source line and by instruction.
#   METHOD=stochastic|host_trap  INTERVAL=<cycles | us>  LIBVAR=glines
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
o=gpurun_out/pcs; rm -rf $o; mkdir -p $o
m=${METHOD:-stochastic}; unit=cycles; iv=${INTERVAL:-1048576}
[ $m = host_trap ] && unit=time
export MZR_LIB=$PWD/mizuroute_amd/lib_var/${LIBVAR:-glines}/libmzr_hip.so
timeout 900 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $m --pc-sampling-unit $unit --pc-sampling-interval $iv --kernel-trace --output-format csv -d $o/raw -o p -- \
  python bench.py --no-cpu-baseline --no-roofline --no-h2d --no-single-step --no-configs ${ARGS:---window 4096 --steps 2 --warmup 3} > $o/run.log 2>&1
echo "exit $?"; tail -5 $o/run.log
find $o/raw -type f | xargs ls -la
python tools/pcs_agg.py $o/raw > $o/agg.txt 2>&1; head -60 $o/agg.txt
for f in $(find $o/raw -name "*pc_sampling*.csv"); do head -2000 $f > $o/$(basename $f).head; done
rm -rf $o/raw
