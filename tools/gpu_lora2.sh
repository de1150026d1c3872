cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/lora
rm -rf gpurun_out/lora/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lora/prof -- python3 bench.py --lora --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-extra --no-graphs > gpurun_out/lora/prof.json 2> gpurun_out/lora/prof.err
find gpurun_out/lora -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/lora/prof/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total ms/step", tot/1e6/7)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:24]:
    print(f"{float(r['TotalDurationNs'])/1e6/7:8.2f} ms/step {int(r['Calls'])/7:7.1f} calls/step {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:100]}")
PY
