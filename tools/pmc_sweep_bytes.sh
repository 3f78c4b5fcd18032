cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
ARGS="--no-cpu-baseline --no-roofline --no-h2d --no-single-step --window 16384 --steps 2 --warmup 1"
out=gpurun_out/pmc_fw; rm -rf $out; mkdir -p $out
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/g0 -o p -- python bench.py $ARGS > $out/g0.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/g1 -o p -- python bench.py $ARGS > $out/g1.log 2>&1
python - <<PY
import csv, glob
for f in sorted(glob.glob("gpurun_out/pmc_fw/g*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "k_sweep_kwt" in row["Kernel_Name"]:
            print(row["Counter_Name"], row.get("Dispatch_Id"), row["Counter_Value"])
PY
