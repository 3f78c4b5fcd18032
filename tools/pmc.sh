#!/bin/bash
# Run inside gpurun: SQ / cache counters per kernel, one rocprofv3 pass per counter group (separate from --stats runs).
# usage: tools/pmc.sh <tag> "<command>"      -> gpurun_out/<tag>_pmc.json  (summed per kernel and counter)
tag=$1; shift
cmd=$1
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
groups=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"
 "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"
 "SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES"
 "SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU"
 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64"
 "FETCH_SIZE" "WRITE_SIZE"
)
i=0
for g in "${groups[@]}"; do
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out/g$i -o p -- $cmd > $out/g$i.log 2>&1
  i=$((i+1))
done
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
last = collections.defaultdict(dict)      # value of the LAST dispatch of every kernel (the steady window; the sum also holds the census launch and the cold first window)
for f in glob.glob("$out/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:60]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
        d = int(row.get("Dispatch_Id", 0) or 0)
        if d >= last[k].get(row["Counter_Name"], (-1, 0.0))[0]:
            last[k][row["Counter_Name"]] = (d, float(row["Counter_Value"]))
res = {k: {c: v for c, v in d.items()} for k, d in agg.items()}
for k in res:
    res[k]["_dispatches"] = max(cnt[(k, c)] for c in agg[k])
    res[k]["_last"] = {c: v[1] for c, v in last[k].items()}
json.dump(res, open("gpurun_out/${tag}_pmc.json", "w"), indent=1)
for k, d in sorted(res.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:6]:
    print(k, json.dumps(d))
PY
rm -rf $out/g*/
