cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --no-roofline --window 8192 --steps 2 --warmup 2 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('value %.4g  ms/step %.2f' % (j['value'], j['ms_per_step']))
    elif 'rror' in l: print(l.rstrip())
"; }
run X=1
run MZR_KWT_SWEEP_WAVES=3072
run MZR_KWT_SWEEP_WAVES=2048
run MZR_KWT_SWEEP=0
