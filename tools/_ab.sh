run() { label=$1; shift; env "$@" | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', j['value'], j['ms_per_step'], j['roofline']['frac'])"; }
for rep in 1 2; do
  run "ce reg " python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-extra 2>/dev/null
  run "ce 2pass" TASU_CE_REG=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-extra 2>/dev/null
done
