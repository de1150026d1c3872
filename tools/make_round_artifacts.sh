set -euo pipefail; cd "${GRAFT_REPO_ROOT:?run through gpurun (it exports GRAFT_REPO_ROOT)}"; export TMPDIR=/tmp; set +e   # (the runs below report their own exit codes)
R=${R:-r05}
rm -rf gpurun_out/art; mkdir -p gpurun_out/art
timeout 1500 python bench.py > gpurun_out/art/bench_$R.json 2> gpurun_out/art/bench_$R.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/art/prof_bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-extra --no-graphs > gpurun_out/art/${R}_bench_under_rocprof.json 2> gpurun_out/art/prof_bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/art/prof_decode -- python3 tools/bench_paths.py decode 16 > gpurun_out/art/${R}_decode_under_rocprof.json 2> gpurun_out/art/prof_decode.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/art/prof_audio -- python3 bench.py --path audio --blank-biased --blank-bias 13.0 --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-extra --no-graphs > gpurun_out/art/${R}_audio_under_rocprof.json 2> gpurun_out/art/prof_audio.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/art/prof_lora -- python3 bench.py --lora --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-extra --no-graphs > gpurun_out/art/${R}_lora_under_rocprof.json 2> gpurun_out/art/prof_lora.err
find gpurun_out/art -name "*kernel_trace.csv" -delete
find gpurun_out/art -name "*.csv" | head -20; du -sh gpurun_out/art
tail -c 1200 gpurun_out/art/bench_$R.json
