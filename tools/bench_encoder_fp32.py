"""The frozen SenseVoice encoder + CTC softmax in fp32 (ps_slm_amd/encoder.py: encoder_posterior_fp32), 16 utterances x 500 feature
frames: milliseconds per pass -- what an fp32 AUDIO decode pays in front of its prompt pass (the bench's decode_fp32 leg is text-only).
    python tools/bench_encoder_fp32.py [B]      # under rocprofv3 --kernel-trace --stats for the per-kernel split"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ps_slm_amd.encoder import encoder_posterior_fp32  # noqa: E402
from ps_slm_amd.model import Geometry, TasuModel  # noqa: E402
from ps_slm_amd.ops import HipOps  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
geo = Geometry.qwen25_1p5b()
m = TasuModel(geo, HipOps(), "cuda")
m.llm.keep_f32 = True
m.arith = "fp32"
m.init_random(seed=1, with_encoder=True)
g = torch.Generator().manual_seed(2)
feats = torch.randn(B, 500, geo.feat_dim, generator=g)
lens = torch.full((B,), 500, dtype=torch.int64)
for _ in range(2):
    encoder_posterior_fp32(m, feats, lens)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 3
for _ in range(n):
    encoder_posterior_fp32(m, feats, lens)
e1.record()
torch.cuda.synchronize()
print(json.dumps({"B": B, "frames": 500, "encoder_fp32_ms": round(e0.elapsed_time(e1) / n, 2)}))
