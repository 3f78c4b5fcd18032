#!/bin/bash
# round 5 (inside gpurun): per-section wave cycles and spin counts of the sweep (timing build), flavours and window lengths given as K:W pairs
cd ${GRAFT_REPO_ROOT:-/root/repo}
export MZR_LIB=$PWD/mizuroute_amd/lib_var/timing/libmzr_hip.so
for spec in "$@"; do
  k=${spec%%:*}; w=${spec##*:}
  echo "######## MZR_KWT_KBLK_RUN=$k W=$w"
  MZR_KWT_KBLK_RUN=$k WW=$w python tools/kwt_sections.py 2>&1 | grep -v amdgpu.ids | tail -22 | head -19
done
