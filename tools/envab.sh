cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "=== $*"; env "$@" python bench.py --no-cpu-baseline --no-h2d --no-single-step --no-roofline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('value %.4g  ms/step %.2f' % (j['value'], j['ms_per_step']))
    elif 'rror' in l: print(l.rstrip())
"; }
run A=1
run MZR_KWT_CLASSB_MAX=24
run MZR_KWT_CLASSB_MAX=28
run MZR_KWT_CLASSB_MAX=16
run MZR_KWT_CLASSC_MAX=11
run MZR_KWT_CLASSC_MAX=7
run MZR_KWT_OCC=4
run MZR_KWT_OCC=6
