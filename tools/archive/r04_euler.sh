#!/bin/bash
# Debug (inside gpurun): Eulerian stage kernels after the IRF load batching and the FULL = false instantiations; MC at 4 / 5 wavefronts per SIMD
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/ab
(time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -k "not kwt_vs and not operating_point and not full_size and not c3_shard and not golden") > gpurun_out/r04_euler_tests.log 2>&1
tail -3 gpurun_out/r04_euler_tests.log
i=0
for v in "" "-DMZR_MC_WAVES=4" "-DMZR_MC_WAVES=5"; do
  make -C mizuroute_amd/csrc clean >/dev/null; make -C mizuroute_amd/csrc all EXTRA="$v" -j8 > gpurun_out/ab/build_e$i.log 2>&1 || { echo "BUILD FAILED [$v]"; tail -5 gpurun_out/ab/build_e$i.log; }; i=$((i+1))
  echo "##### build [$v]"
  METHODS=IRF,MC,DW python tools/bench_methods.py 2>&1 | tail -1
  NR=625000 WW=3072 METHODS=IRF,MC python tools/bench_methods.py 2>&1 | tail -1
done
make -C mizuroute_amd/csrc clean >/dev/null; make -C mizuroute_amd/csrc all -j8 >/dev/null 2>&1
python bench.py --no-cpu-baseline --no-single-step --no-configs --no-h2d --config c4 --steps 4 --warmup 3 2>&1 | tail -1 | cut -c1-300
