cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/c5; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "encoder or gemm" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py -x -q -k "encoder or audio or benchmark_shape" >> $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 600 python bench.py --path audio --blank-biased --blank-bias 13.0 --steps 8 --warmup 3 --no-cpu-baseline --no-decode --no-extra > $O/audio_bb.json 2> $O/audio_bb.err
TASU_ATTN_QW2_FROM=500 timeout 600 python bench.py --path audio --blank-biased --blank-bias 13.0 --steps 8 --warmup 3 --no-cpu-baseline --no-decode --no-extra > $O/audio_bb_qw2.json 2> $O/audio_bb_qw2.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_audio -- python3 bench.py --path audio --blank-biased --blank-bias 13.0 --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-extra --no-graphs > $O/audio_under_rocprof.json 2> $O/prof_audio.err
find $O -name "*kernel_trace.csv" -delete
python - <<'PY'
import json
for f in ("audio_bb","audio_bb_qw2","audio_under_rocprof"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/c5/{f}.json") if l.startswith("{")][-1]); print(f, d["value"], d["ms_per_step"], d["config"]["seq_len"], d["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
PY
