#!/bin/bash
# round 5 (inside gpurun): instruction counts and wave cycles of k_sweep_kwt, one step per visit against blocks of steps
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
o=gpurun_out/valu; rm -rf $o; mkdir -p $o
count() {
  lab=$1; shift; envs=$1; shift
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"; do
    env $envs rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $o/$lab -o p -- python bench.py --no-cpu-baseline --no-roofline --no-h2d --no-single-step --no-configs "$@" > $o/$lab.log 2>&1
    python - <<PY
import csv, glob, collections
agg = collections.defaultdict(float)
for f in glob.glob("$o/$lab/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_sweep_kwt" in r["Kernel_Name"]: agg[r["Counter_Name"]] += float(r["Counter_Value"])
print("$lab [$envs]", " ".join(f"{k}={v:.4g}" for k, v in sorted(agg.items())))
PY
    rm -rf $o/$lab
  done
}
count c2 "MZR_KWT_KBLK_RUN=1" --window 4096 --steps 2 --warmup 3
count c2 "MZR_KWT_KBLK_RUN=4" --window 4096 --steps 2 --warmup 3
count c3 "MZR_KWT_KBLK_RUN=1" --config c3 --window 1024 --steps 2 --warmup 3
count c3 "MZR_KWT_KBLK_RUN=4" --config c3 --window 1024 --steps 2 --warmup 3
