"""Side streams that the hardware really runs NEXT TO the main stream.

HIP maps streams onto a handful of hardware queues (4 per device by default) in creation order, the default stream included;
two streams on one hardware queue execute strictly one after the other.  torch hands out its pooled streams round-robin, so every
fourth ``torch.cuda.Stream()`` shares the default stream's queue -- and which caller gets that one depends on how many streams
the process created before (measured: the adapters' weight-gradient stream landed on it after two earlier workloads in one
process, and the LoRA step went from 48.0 to 52.8 ms; `tools/lab_stream_queues.py` shows the period-4 pattern).  The overlap
designs of this package (gradient exchange under backward, the adapters' weight gradients under the dgrad chain, the frozen encoder
under the previous decoder step) therefore ask for their stream here: candidates are probed with two short spin kernels -- wall
time of the pair ~ one spin means concurrent -- against the current stream and against the side streams already handed out."""
import time

import torch

_SPIN_CYCLES = 400_000            # ~0.17 ms at 2.4 GHz: long against launch latency, short enough to probe a dozen pairs
_state = {}                       # device index -> dict(reps=[one stream per hardware queue other than the main one], next=round-robin position)


def _pair_seconds(a, b):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(a):
        torch.cuda._sleep(_SPIN_CYCLES)
    with torch.cuda.stream(b):
        torch.cuda._sleep(_SPIN_CYCLES)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def runs_concurrently(a, b):
    """True when a spin kernel on ``a`` and one on ``b`` overlap in time (different hardware queues)."""
    _pair_seconds(a, b)                                   # first use of a stream creates its queue: not timed
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(a):
        torch.cuda._sleep(_SPIN_CYCLES)
    torch.cuda.synchronize()
    one = time.perf_counter() - t0
    pair = min(_pair_seconds(a, b) for _ in range(3))
    return pair < 1.5 * one


def _classes(idx):
    """The pooled candidates grouped by hardware queue (probed once per device): one representative stream per queue that is NOT
    the current stream's queue."""
    st = _state.get(idx)
    if st is not None:
        return st
    main = torch.cuda.current_stream()
    reps = []
    for _ in range(12):
        c = torch.cuda.Stream(device=idx)
        if not runs_concurrently(main, c):
            continue                                      # shares the main stream's queue
        if all(runs_concurrently(r, c) for r in reps):
            reps.append(c)                                # a queue not seen yet
        if len(reps) == 3:                                # 4 hardware queues per device: main + 3
            break
    st = _state[idx] = dict(reps=reps, next=0)
    return st


def side_stream(device):
    """A stream for work that is to overlap with the CURRENT stream of ``device``: callers get the device's other hardware
    queues in turn (the first three calls three different queues -- gradient exchange, adapters' weight gradients, encoder ahead --
    later calls share them round-robin).  None on a CPU device.  Never call inside a hipGraph capture (the first call per device
    launches probe kernels and synchronises)."""
    device = torch.device(device)
    if device.type != "cuda":
        return None
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if not hasattr(torch.cuda, "_sleep"):                 # (no spin kernel to probe with: any stream)
        return torch.cuda.Stream(device=idx)
    with torch.cuda.device(idx):
        st = _classes(idx)
        if not st["reps"]:                                # (a runtime with one hardware queue: overlap is impossible anyway)
            return torch.cuda.Stream(device=idx)
        s_ = st["reps"][st["next"] % len(st["reps"])]
        st["next"] += 1
        return s_
