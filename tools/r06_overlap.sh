#!/bin/bash
# round 6: do the IRF and the Muskingum-Cunge launches of a c4 step overlap?  kernel trace of two windows, start / end of every launch
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_overlap; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o k -- python bench.py --config c4 --steps ${STEPS:-1} --warmup ${WARMUP:-1} --window ${WIN:-1024} --no-cpu-baseline --no-h2d --no-single-step --no-configs --no-roofline > $O/run.log 2>&1
python - "$O" <<'PY'
import csv, glob, sys, collections
rows=[]
for f in glob.glob(sys.argv[1]+"/tr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "k_stage" in n:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MC" if "ILi4E" in n or "<4" in n else "IRF" if "ILi1E" in n or "<1" in n else n[:20], r.get("Queue_Id","?"), r.get("Stream_Id","?")))
rows.sort()
print("launches", len(rows), collections.Counter(x[2] for x in rows), "queues", collections.Counter((x[2],x[3]) for x in rows))
mid=rows[len(rows)//2: len(rows)//2+40]
t0=mid[0][0]
for s,e,k,q,st in mid: print("%-4s q%s start %8.1f us  end %8.1f us  dur %6.1f" % (k,q,(s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3))
half=rows[int(len(rows)*float(__import__("os").environ.get("LO","0.55"))):int(len(rows)*float(__import__("os").environ.get("HI","0.75")))]
import statistics
for k in ("IRF","MC"):
    d=[(e-s)/1e3 for s,e,kk,*_ in half if kk==k]; st=[s for s,e,kk,*_ in half if kk==k]
    per=[(b-a)/1e3 for a,b in zip(st,st[1:])]
    print(k, "steady: dur mean %.1f median %.1f max %.1f ; period mean %.1f" % (statistics.mean(d), statistics.median(d), max(d), statistics.mean(per)))
span=(max(e for s,e,*_ in half)-min(s for s,e,*_ in half))/1e3
busy={k:sum(e-s for s,e,kk,*_ in half if kk==k)/1e3 for k in ("IRF","MC")}
print("span us %.0f  sum of durations IRF %.0f MC %.0f  launches %d" % (span, busy["IRF"], busy["MC"], len(half)))
PY
rm -rf $O/tr
