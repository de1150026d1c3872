# Alternating-process A/B of one environment switch on the headline step (same box, same code): prints ms_per_step per run.
#   bash tools/ab_env.sh TASU_UPLOAD_PACK 0 1 [reps] [extra bench.py flags]
set -u; cd "${GRAFT_REPO_ROOT:-.}"
VAR=$1; A=$2; B=$3; REPS=${4:-3}; shift 4 2>/dev/null || shift $#
for i in $(seq $REPS); do for v in $A $B; do
  ms=$(env $VAR=$v python bench.py --no-cpu-baseline --no-extra --no-decode --steps 30 --warmup 5 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['gemm_ms_per_step'])")
  echo "$VAR=$v ms_per_step gemm_ms: $ms"
done; done
