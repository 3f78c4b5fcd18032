#!/bin/bash
# Debug (run inside gpurun): per-section wave cycles and per-class wave life of the KWT stage kernel.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/sections
for v in MZR_KWT_TIMING; do
  make -C mizuroute_amd/csrc clean >/dev/null; make -C mizuroute_amd/csrc all EXTRA=-D$v -j8 > gpurun_out/sections/build_$v.log 2>&1
  WW=2048 python tools/kwt_sections.py > gpurun_out/sections/$v.txt 2>&1
  MZR_KWT_SWEEP_WAVES=${SWW:-99999} python tools/kwt_records.py > gpurun_out/sections/records.txt 2>&1; cat gpurun_out/sections/records.txt
  tail -20 gpurun_out/sections/$v.txt
done
make -C mizuroute_amd/csrc clean >/dev/null; make -C mizuroute_amd/csrc all -j8 >/dev/null 2>&1
