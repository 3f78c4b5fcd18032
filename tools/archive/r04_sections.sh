cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/sections
(time timeout 1200 python -m pytest tests/test_gpu_scale.py -x -q -k full_size -s) > gpurun_out/r04_fullsize_tests.log 2>&1
make -C mizuroute_amd/csrc clean >/dev/null; make -C mizuroute_amd/csrc all EXTRA=-DMZR_KWT_TIMING -j8 > gpurun_out/sections/build.log 2>&1
WW=2048 timeout 600 python tools/kwt_sections.py > gpurun_out/sections/sections_100k.txt 2>&1
MZR_KWT_SWEEP_WAVES=99999 NR=100000 timeout 600 python tools/kwt_records.py > gpurun_out/sections/records_100k.txt 2>&1
MZR_KWT_SWEEP_WAVES=256 NR=100000 timeout 600 python tools/kwt_records.py > gpurun_out/sections/records_100k_alone.txt 2>&1
MZR_KWT_SWEEP_WAVES=99999 NR=375000 WW=1024 timeout 600 python tools/kwt_records.py > gpurun_out/sections/records_375k.txt 2>&1
