run() { label=$1; shift; env "$@" | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', j['value'], j['ms_per_step'], j['roofline']['frac'])"; }
for rep in 1 2; do
  run "drop0.05 default" python bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-decode --no-extra --drop-prob 0.05 2>/dev/null
  run "drop0.05 SK=0   " TASU_GEMM_SK=0 python bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-decode --no-extra --drop-prob 0.05 2>/dev/null
  run "batch12  default" python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-decode --no-extra --batch 12 2>/dev/null
  run "batch12  SK=0   " TASU_GEMM_SK=0 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-decode --no-extra --batch 12 2>/dev/null
  run "batch8   default" python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-decode --no-extra --batch 8 2>/dev/null
  run "batch8   SK=0   " TASU_GEMM_SK=0 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-decode --no-extra --batch 8 2>/dev/null
done
