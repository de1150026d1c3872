import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.ops import HipOps
ops = HipOps()
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
m, n, k = 4096, 17920, 1536
sets = [(torch.randn(m, k, device="cuda").to(torch.bfloat16), torch.randn(n, k, device="cuda").to(torch.bfloat16),
         torch.empty(m, n, device="cuda", dtype=torch.bfloat16)) for _ in range(8)]
torch.cuda.synchronize()
for i in range(24):
    a, b, c = sets[i % 8]
    side.wait_stream(main)
    ops.cache_prefetch(sets[(i + 1) % 8][1], side, 512, 0)
    ops.gemm(a, b, c, m, n, k)
main.wait_stream(side)
torch.cuda.synchronize()
