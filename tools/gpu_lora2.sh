cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/lora
timeout 600 python bench.py --lora --steps 10 --warmup 3 --no-cpu-baseline --no-decode --no-extra > gpurun_out/lora/bench_lora.json 2> gpurun_out/lora/bench_lora.err
tail -c 1500 gpurun_out/lora/bench_lora.json; tail -5 gpurun_out/lora/bench_lora.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lora/prof -- python3 bench.py --lora --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-extra --no-graphs > gpurun_out/lora/prof.json 2> gpurun_out/lora/prof.err
find gpurun_out/lora -name "*kernel_trace.csv" -delete
python tools/summarize_profile.py $(find gpurun_out/lora/prof -name "*kernel_stats.csv") 2>/dev/null | head -40
