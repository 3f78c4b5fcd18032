#!/bin/bash
# Debug (inside gpurun): timeline of the host-forcing pipeline (kernel + memory-copy trace), sweeps and copies with their gaps
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
for mode in dev f32 f64; do
  o=gpurun_out/h2dtrace_$mode; rm -rf $o; mkdir -p $o
  MODE=$mode rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $o -o t -- python tools/h2d_trace.py > $o/run.log 2>&1
  tail -1 $o/run.log
  python - <<PY
import csv, glob
rows = []
for f in glob.glob("$o/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + n))
for f in glob.glob("$o/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "copy"))[:28]))
rows.sort()
big = [r for r in rows if r[1] - r[0] > 5e6]       # longer than 5 ms
t0 = big[0][0] if big else 0
last_end = None
for s, e, n in big[-40:]:
    print("%9.1f ms  +%8.1f ms  %s" % ((s - t0) / 1e6, (e - s) / 1e6, n))
PY
  find $o -name "*.csv" -size +5M -delete
done
