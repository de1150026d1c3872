cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/lora
timeout 900 python -m pytest tests/test_gpu_lora.py -x -q 2>&1 | tail -5
rm -rf gpurun_out/lora/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lora/prof -- python3 bench.py --lora --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-extra --no-graphs > gpurun_out/lora/prof.json 2> gpurun_out/lora/prof.err
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/lora/prof/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
# per-shape durations of rank kernel: group by grid size
import collections
d=collections.defaultdict(list)
for r in rows:
    if 'rank_gemm' in r['Kernel_Name']:
        d[(r['Kernel_Name'][:60], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'))].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(d.items()):
    print(k, len(v), sum(v)/len(v)/1e3)
PY
find gpurun_out/lora -name "*kernel_trace.csv" -delete
