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
_state = {}                       # device index -> dict(candidates=[...], taken=[...])


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


def side_stream(device):
    """A stream for work that is to overlap with the CURRENT stream of ``device`` (and with the side streams handed out before).
    None on a CPU device.  Never call inside a hipGraph capture (it launches and synchronises)."""
    device = torch.device(device)
    if device.type != "cuda":
        return None
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if not hasattr(torch.cuda, "_sleep"):                 # (no spin kernel to probe with: any stream)
        return torch.cuda.Stream(device=idx)
    with torch.cuda.device(idx):
        st = _state.setdefault(idx, dict(candidates=[], taken=[]))
        main = torch.cuda.current_stream()
        best = None
        for n in range(12):
            if n >= len(st["candidates"]):
                st["candidates"].append(torch.cuda.Stream(device=idx))
            c = st["candidates"][n]
            if any(c is t for t in st["taken"]):
                continue
            if not runs_concurrently(main, c):
                continue
            if best is None:
                best = c                                  # overlaps with the main stream at least
            if all(runs_concurrently(t, c) for t in st["taken"]):
                best = c
                break
        if best is None:                                  # (a runtime with one hardware queue: overlap is then impossible anyway)
            best = torch.cuda.Stream(device=idx)
        st["taken"].append(best)
        return best
