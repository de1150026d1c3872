cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/art
timeout 900 python bench.py > gpurun_out/art/bench_r03.json 2> gpurun_out/art/bench_r03.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/art/prof_bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-extra --no-graphs > gpurun_out/art/r03_bench_under_rocprof.json 2> gpurun_out/art/prof_bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/art/prof_decode -- python3 tools/bench_paths.py decode 16 > gpurun_out/art/r03_decode_under_rocprof.json 2> gpurun_out/art/prof_decode.err
find gpurun_out/art -name "*kernel_trace.csv" -delete
find gpurun_out/art -name "*.csv" | head; du -sh gpurun_out/art
tail -c 600 gpurun_out/art/bench_r03.json
