cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_entrypoints.py -x -q -k "graph_replay_on_a_corpus" 2>&1 | tail -40
