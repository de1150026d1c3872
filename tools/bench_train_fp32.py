"""fp32 training step alone (bench.train_fp32_leg): python tools/bench_train_fp32.py [B]  -- for rocprofv3 --kernel-trace --stats."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

r = bench.train_fp32_leg(0, int(sys.argv[1]) if len(sys.argv) > 1 else 16)
r["roofline"].pop("note", None)
print(json.dumps(r))
