#!/bin/bash
# round 6: per-pass section cycles of the sweep (timing build, tools/build_variant.sh / lib_var/timing), with and without the heaviest reaches alone in their pass
cd "$GRAFT_REPO_ROOT" || exit 1
export MZR_LIB=$PWD/mizuroute_amd/lib_var/timing/libmzr_hip.so
for e in ${ENVS:-"MZR_KWT_SOLO_MIN=0" "MZR_KWT_SOLO_MIN=45"}; do
  echo "######## $e"
  env $e WW=${WW:-4096} timeout 600 python tools/kwt_records.py 2>&1 | grep -v amdgpu.ids | tail -40
done
