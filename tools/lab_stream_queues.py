"""Lab: which torch streams does the hardware run NEXT TO the default stream?  A spin kernel on the default stream and one on the
candidate: wall time ~ one spin = concurrent, ~ two = same hardware queue (HIP maps streams onto a few hardware queues)."""
import time, json, torch
torch.cuda.init()
x = torch.zeros(1, device="cuda")
CY = 20_000_000
def spin_pair(s):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    torch.cuda._sleep(CY)
    with torch.cuda.stream(s):
        torch.cuda._sleep(CY)
    torch.cuda.synchronize(); return time.perf_counter() - t0
torch.cuda._sleep(CY); torch.cuda.synchronize()
t0 = time.perf_counter(); torch.cuda._sleep(CY); torch.cuda.synchronize(); one = time.perf_counter() - t0
streams = [torch.cuda.Stream() for _ in range(12)]
res = {"one_spin_ms": round(one * 1e3, 2), "pair_ms": [round(spin_pair(s) * 1e3, 2) for s in streams]}
hi = [torch.cuda.Stream(priority=-1) for _ in range(4)]
res["pair_ms_high_priority"] = [round(spin_pair(s) * 1e3, 2) for s in hi]
# pairs among the created streams
def pair2(a, b):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(a): torch.cuda._sleep(CY)
    with torch.cuda.stream(b): torch.cuda._sleep(CY)
    torch.cuda.synchronize(); return time.perf_counter() - t0
res["s0_vs_others_ms"] = [round(pair2(streams[0], s) * 1e3, 2) for s in streams[1:8]]
print(json.dumps(res))
