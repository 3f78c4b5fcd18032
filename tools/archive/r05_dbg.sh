#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
touch mizuroute_amd/csrc/kernels_kwt.hip; make -C mizuroute_amd/csrc EXTRA="-DMZR_DBG_HANDOVER=2 -DMZR_DBG_TRACE=50400" -j4 2>&1 | grep -E "error"
python tools/r05_dbg2.py tree150_all 7 MZR_KWT_CLASSB_MAX=20 MZR_KWT_CLASSC_MAX=0 > gpurun_out/r05_trace.log 2>&1
grep -v "^TR" gpurun_out/r05_trace.log | grep -v amdgpu.ids | tail -5
grep -c "^TR" gpurun_out/r05_trace.log
