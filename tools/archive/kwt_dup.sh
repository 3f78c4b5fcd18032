#!/bin/bash
# Debug (inside gpurun): VALU instructions of the sections of the KWT pass.  A build executes one section twice (same results);
# the difference of SQ_INSTS_VALU of k_sweep_kwt to the plain build is what the section costs.  usage: tools/kwt_dup.sh [tag]
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-dup}
o=gpurun_out/$tag; rm -rf $o; mkdir -p $o
count() {   # $1 = label, $2.. = bench args
  lab=$1; shift
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $o/$lab -o p -- python bench.py --no-cpu-baseline --no-roofline --no-h2d --no-single-step --no-configs "$@" > $o/$lab.log 2>&1
  python - <<PY
import csv, glob, collections
agg = collections.defaultdict(float)
for f in glob.glob("$o/$lab/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_sweep_kwt" in r["Kernel_Name"]: agg[r["Counter_Name"]] += float(r["Counter_Value"])
print("$lab", " ".join(f"{k}={v:.4g}" for k, v in sorted(agg.items())))
PY
  rm -rf $o/$lab
}
for v in ${VARIANTS:-"" "-DMZR_DUP_MERGE=2" "-DMZR_DUP_THIN=2" "-DMZR_DUP_KINWAV=2" "-DMZR_DUP_INTERP=2" "-DMZR_DUP_COUNT=2" "-DMZR_DUP_STORES=2" "-DMZR_DUP_WAIT=2" "-DMZR_DUP_STAGE=2" "-DMZR_THIN_LDS=0"}; do
  make -C mizuroute_amd/csrc clean >/dev/null; make -C mizuroute_amd/csrc all EXTRA="$v" -j8 > $o/build.log 2>&1 || { echo "BUILD FAILED [$v]"; tail -5 $o/build.log; }
  l=$(echo "base$v" | tr -d ' =-' )
  count ${l}_c2 --window 4096 --steps 2 --warmup 3
  count ${l}_c3 --config c3 --window 1024 --steps 2 --warmup 3
done
make -C mizuroute_amd/csrc clean >/dev/null; make -C mizuroute_amd/csrc all -j8 >/dev/null 2>&1
