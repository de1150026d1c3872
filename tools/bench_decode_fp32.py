"""fp32 decode leg alone (bench.decode_fp32_leg): python tools/bench_decode_fp32.py [B [qwen2.5-7b [new_tokens]]]  -- for rocprofv3 --kernel-trace --stats."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

r = bench.decode_fp32_leg(0, int(sys.argv[1]) if len(sys.argv) > 1 else 16, new_tokens=int(sys.argv[3]) if len(sys.argv) > 3 else 200,
                          model_name=sys.argv[2] if len(sys.argv) > 2 else "qwen2.5-1.5b")
r["roofline"].pop("note", None)
print(json.dumps(r))
