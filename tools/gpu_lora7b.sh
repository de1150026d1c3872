cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --model qwen2.5-7b --lora --steps 4 --warmup 2 --no-cpu-baseline --no-decode --no-extra 2>&1 | tail -c 600
