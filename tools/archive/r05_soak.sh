#!/bin/bash
# round 5 (inside gpurun): the multi-second launches of round 4 -- N processes of the driver's c2 legs under rocprofv3 --kernel-trace;
# per launch of k_sweep_kwt: the duration the profiler saw against the duration on the device's own clock (first wavefront in -> last
# wavefront out).  A launch that is long for the profiler and short on the device clock spent the difference BEFORE its first wavefront.
#   tools/r05_soak.sh <N> [bench arguments]
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
n=${1:-4}; shift
o=gpurun_out/r05_soak; rm -rf $o; mkdir -p $o
for i in $(seq 1 $n); do
  rocprofv3 --kernel-trace --output-format csv -d $o/p$i -o p -- python bench.py --no-cpu-baseline --no-h2d --no-single-step --no-configs --steps 12 --warmup 3 "$@" > $o/run_$i.out 2> $o/run_$i.err
  python - $o $i <<'PY'
import csv, glob, json, sys
o, i = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(f"{o}/p{i}/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sw = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_sweep_kwt" in r["Kernel_Name"]]
heads = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_sweep_heads" in r["Kernel_Name"]]
prof = [(e - s) / 1e6 for s, e in sw]
gap = [(sw[k][0] - heads[k][1]) / 1e6 if k < len(heads) else None for k in range(len(sw))]      # from the end of the heads kernel to the sweep's start
try:
    j = json.loads([l for l in open(f"{o}/run_{i}.out") if l.startswith("{")][-1])
    dev = [x / 1e3 for x in (j.get("roofline") or {}).get("launch_us") or []]
    val, err = j.get("value"), j.get("error")
except Exception as e:
    dev, val, err = [], None, str(e)
big = [k for k, p in enumerate(prof) if p > 1.5 * sorted(prof)[len(prof) // 2]]
print(f"run {i}: value {val} err {err}; k_sweep_kwt launches {len(prof)} (the first is the census), profiler ms: " + " ".join(f"{p:.0f}" for p in prof))
print(f"        device clock ms (the K timed windows): " + " ".join(f"{d:.0f}" for d in dev))
print(f"        gap heads->sweep ms: " + " ".join("-" if g is None else f"{g:.2f}" for g in gap) + f"; launches beyond 1.5 x the median: {big}")
PY
  rm -rf $o/p$i
done
