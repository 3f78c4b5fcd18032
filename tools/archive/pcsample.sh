#!/bin/bash
# Run inside gpurun: PC sampling of the KWT sweep (library built with line tables), histogram by source line.
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
make -C mizuroute_amd/csrc clean >/dev/null; make -C mizuroute_amd/csrc all EXTRA="-gline-tables-only" -j8 > gpurun_out/pcs_build.log 2>&1
rm -rf gpurun_out/pcs; mkdir -p gpurun_out/pcs
method=${PCS_METHOD:-host_trap}; unit=${PCS_UNIT:-time}; interval=${PCS_INTERVAL:-1}
NR=100000 WW=2048 NW=3 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit --pc-sampling-interval $interval --output-format csv -d gpurun_out/pcs -o s -- python tools/dbg_sweep.py > gpurun_out/pcs.log 2>&1
tail -5 gpurun_out/pcs.log
find gpurun_out/pcs -type f | head; 
f=$(find gpurun_out/pcs -name "*pc_sampling*.csv" | head -1)
[ -n "$f" ] && { head -3 $f; wc -l $f; }
python - <<'PY'
import csv, glob, collections, re, subprocess, os
fs = glob.glob("gpurun_out/pcs/**/*pc_sampling*.csv", recursive=True)
if not fs: raise SystemExit("no pc sampling output")
rows = list(csv.DictReader(open(fs[0])))
print(len(rows), rows[0].keys() if rows else None)
cnt = collections.Counter()
for r in rows:
    key = r.get("Instruction_Comment") or r.get("Instruction") or ""
    cnt[key] += 1
out = open("gpurun_out/pcs_hist.txt", "w")
for k, v in cnt.most_common(4000): out.write(f"{v}\t{k}\n")
PY
find gpurun_out/pcs -name "*.csv" -size +20M -delete
make -C mizuroute_amd/csrc clean >/dev/null; make -C mizuroute_amd/csrc all -j8 >/dev/null 2>&1
