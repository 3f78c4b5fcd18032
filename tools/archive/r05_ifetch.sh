#!/bin/bash
# round 5 (inside gpurun): do the 16-lane passes run faster when every wavefront runs the same instantiation (all reaches in class A)?
cd ${GRAFT_REPO_ROOT:-/root/repo}
export MZR_LIB=$PWD/mizuroute_amd/lib_var/timing/libmzr_hip.so
for e in "X=1" "MZR_KWT_CLASSB_MAX=0 MZR_KWT_CLASSC_MAX=0"; do
  echo "######## $e"
  env $e MZR_KWT_SWEEP_WAVES=99999 WW=4096 python tools/kwt_records.py 2>&1 | grep -v amdgpu.ids | grep -A12 "^G=16" | head -14
done
