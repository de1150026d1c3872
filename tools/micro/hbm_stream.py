"""Practical streaming rates of this box's HBM with the framework's own vendor kernels (context for the 8 TB/s datasheet peak that
roofline.frac of the decode leg is quoted against): read-only (sum), copy (read + write), fill (write-only) on 4-GiB buffers, far
beyond the 256-MB Infinity Cache."""
import json, torch

n = 1 << 31                                     # bf16 elements: 4 GiB
x = torch.ones(n, dtype=torch.bfloat16, device="cuda")
y = torch.empty_like(x)


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


xi = x.view(torch.int32)
out = {"read_sum_TBps": round(2 * n / t(lambda: xi.sum()) / 1e12, 2),
       "copy_read_plus_write_TBps": round(4 * n / t(lambda: y.copy_(x)) / 1e12, 2),
       "fill_write_TBps": round(2 * n / t(lambda: y.fill_(1.0)) / 1e12, 2)}
print(json.dumps(out))
