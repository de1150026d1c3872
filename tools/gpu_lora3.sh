cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_lora.py -x -q 2>&1 | tail -5
timeout 600 python bench.py --lora --steps 10 --warmup 3 --no-cpu-baseline --no-decode --no-extra 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['gemm_ms_per_step'], r['roofline']['launches_per_step'])"
timeout 600 python bench.py --lora --steps 10 --warmup 3 --no-cpu-baseline --no-decode --no-extra --no-graphs 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('eager', r['value'], r['ms_per_step'])"
