import sys, time, torch
sys.path.insert(0, '/root/repo')
from ps_slm_amd.ops import HipOps
ops = HipOps()
pos = torch.arange(64, device="cuda", dtype=torch.int32)
cos = torch.zeros(64, 64, device="cuda"); sin = torch.zeros(64, 64, device="cuda")
x = torch.randn(4096, 1536, device="cuda"); w = torch.ones(1536, device="cuda"); y = torch.zeros(4096, 1536, device="cuda", dtype=torch.bfloat16); r = torch.zeros(4096, device="cuda")
for name, fn in (("rope_table (tiny)", lambda: ops.rope_table(pos, cos, sin, 128, 1e6)), ("rmsnorm_fwd 4096x1536 (8us)", lambda: ops.rmsnorm_fwd(x, w, y, r, 1e-6))):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host issue {1e6*(t1-t0)/2000:.2f} us/launch, total {1e6*(t2-t0)/2000:.2f} us/launch")
