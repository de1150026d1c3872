cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/c1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_audio -- python3 bench.py --path audio --blank-biased --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-extra --no-graphs > $O/audio_under_rocprof.json 2> $O/prof_audio.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-extra --no-graphs > $O/bench_under_rocprof.json 2> $O/prof_bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_decode -- python3 tools/bench_paths.py decode 16 > $O/decode_under_rocprof.json 2> $O/prof_decode.err
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.csv" | head; du -sh $O; tail -3 $O/pytest.log
