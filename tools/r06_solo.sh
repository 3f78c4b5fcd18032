#!/bin/bash
# round 6, VERDICT item 4: the heaviest class-A reaches alone (or two) in their pass -- MZR_KWT_SOLO_MIN / MZR_KWT_SOLO_PER (kwt_regroup).
# parity under the switch first (bit-identical by construction: a hole is a lane group without a step), then c2 (or CONFIG) per setting.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_solo; mkdir -p $O
if [ -z "$NOPARITY" ]; then
MZR_KWT_SOLO_MIN=8 MZR_KWT_SOLO_PER=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kwt or KWT or golden or sweep or lane_classes" > $O/parity1.log 2>&1; tail -2 $O/parity1.log
MZR_KWT_SOLO_MIN=33 MZR_KWT_SOLO_PER=1 timeout 900 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "c2_full_size" > $O/parity2.log 2>&1; tail -2 $O/parity2.log
fi
export MZR_KWT_CLASS_LOG=1
CONFIG=${CONFIG:-c2} STEPS=${STEPS:-6} ENVS="${ENVS:-MZR_KWT_SOLO_MIN=41 MZR_KWT_SOLO_MIN=33 MZR_KWT_SOLO_MIN=29;MZR_KWT_SOLO_PER=2 MZR_KWT_SOLO_MIN=25;MZR_KWT_SOLO_PER=2 MZR_KWT_SOLO_MIN=1000}" bash tools/r06_env.sh
for f in gpurun_out/r06_env/*.err; do grep "kwt classes" $f | tail -1 | cut -c1-200; done
