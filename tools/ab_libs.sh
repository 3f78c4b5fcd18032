for n in base occ4 sleep2 sleep8 base; do
  if [ $n != base ]; then cp mizuroute_amd/lib/libmzr_hip_$n.so mizuroute_amd/lib/libmzr_hip.so; else cp mizuroute_amd/lib/libmzr_hip_base.so mizuroute_amd/lib/libmzr_hip.so; fi
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-h2d --no-single-step --no-roofline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', j['value'], j['error'])"
done
