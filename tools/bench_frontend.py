"""Times the audio front end (fbank + LFR/CMVN kernels) on 16 x 30 s utterances, HIP events on the launch stream."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ps_slm_amd.frontend import WavFrontend

fe = WavFrontend(cmvn=(np.zeros(560, np.float32), np.ones(560, np.float32)))
n = 16000 * 30
waves = [torch.randn(n, device="cuda") * 0.1 for _ in range(16)]
for w in waves:
    fe(w)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps):
    for w in waves:
        fe(w)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"front end, 16 x 30 s: {ms:.3f} ms per batch ({16 * 30 / (ms / 1e3):.0f} x real time); "
      f"waveform bytes {16 * n * 4 / 1e6:.1f} MB -> features {16 * 500 * 560 * 4 / 1e6:.1f} MB")
