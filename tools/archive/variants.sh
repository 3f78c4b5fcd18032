cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "== $*"; env "$@" 2>&1 | tail -2 | cut -c1-400; }
run python bench.py --no-cpu-baseline --no-roofline --no-h2d --no-single-step --steps 4 --warmup 1
run python bench.py --no-cpu-baseline --no-roofline --no-h2d --no-single-step --steps 4 --warmup 1
run python bench.py --no-cpu-baseline --no-roofline --no-h2d --no-single-step --steps 2 --warmup 2
run env MZR_KWT_SWEEP=0 python bench.py --no-cpu-baseline --no-roofline --no-h2d --no-single-step --steps 4 --warmup 1
