"""Clock and power while the fp32 lm_head GEMM (64 rows) runs back to back for a few seconds: is the fp32 decode step power-bound?
    (lab build: make -C ps_slm_amd/csrc lab; TASU_LIB_PATH=ps_slm_amd/libtasu_hip_lab.so)
    TASU_F32_STREAM_DBG={0,1,2,4} python tools/micro/f32_stream_clocks.py      # 1 = no weight loads, 2 = no MFMAs, 4 = real bits, no traffic
Samples `rocm-smi --showclocks --showpower` from a thread while the stream is busy."""
import json
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ps_slm_amd.ops import HipOps  # noqa: E402

M, D, V = 64, 1536, 151936
ops = HipOps()
g = torch.Generator(device="cuda").manual_seed(1)
xn = torch.randn(M, D, device="cuda", generator=g)
ws = torch.empty(2 * 64 * V, device="cuda")
logits = torch.empty(M, V, device="cuda")
heads = [torch.randn(V, D, device="cuda", generator=g) * 0.02 for _ in range(2)]
samples, stop = [], False


def sample():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
        pw = re.search(r"Power \(W\): ([\d.]+)", out) or re.search(r"Average Graphics Package Power \(W\): ([\d.]+)", out)
        samples.append((int(sclk.group(1)) if sclk else None, float(pw.group(1)) if pw else None))
        time.sleep(0.2)


for h in heads:
    ops.f32_gemm(xn, h, logits, M, V, D, ws=ws)
torch.cuda.synchronize()
th = threading.Thread(target=sample)
th.start()
t0 = time.time()
n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < 4.0:
    for _ in range(50):
        ops.f32_gemm(xn, heads[n & 1], logits, M, V, D, ws=ws)
        n += 1
    torch.cuda.synchronize()
e1.record()
torch.cuda.synchronize()
stop = True
th.join()
print(json.dumps({"dbg": os.environ.get("TASU_F32_STREAM_DBG", "0"), "stream": os.environ.get("TASU_F32_STREAM", "1"),
                  "us_per_gemm_incl_finisher": round(e0.elapsed_time(e1) * 1e3 / n, 1), "samples_sclk_MHz_power_W": samples[1:-1]}))
