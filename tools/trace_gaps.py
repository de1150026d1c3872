"""Where a hipGraph-replayed step's wall time goes BETWEEN kernels: reads a rocprofv3 --kernel-trace CSV of a graph-replay run
(`rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-decode
--no-extra`), takes the dispatches of the last replays on the busiest queue, and prints per step: kernel time, idle time between
consecutive kernels, and the idle time attributed to the kernel that FOLLOWS each gap (the launch whose start-up it is).

  python tools/trace_gaps.py DIR_OR_CSV [steps_to_average]
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|void |\(.*", "", n)
    if n.startswith("_ZN"):
        n = re.sub(r"_ZN12_GLOBAL__N_1\d+|_ZN\d+[a-z_]+\d+", "", n)
        n = re.sub(r"E[PKv].*", "", n)
    return n


def main():
    src = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    path = src if src.endswith(".csv") else sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = list(csv.DictReader(open(path)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
    # a step = the span between two consecutive adamw_kernel launches (the step's last kernel); take the last `steps` of them
    ends = [i for i, e in enumerate(ev) if "adamw_kernel" in e[2]]
    if len(ends) < steps + 1:
        raise SystemExit(f"only {len(ends)} steps in the trace")
    lo, hi = ends[-steps - 1] + 1, ends[-1] + 1
    seg = ev[lo:hi]
    busy = sum(e[1] - e[0] for e in seg)
    span = seg[-1][1] - ev[lo - 1][1]
    gap_by, n_by, dur_by = defaultdict(int), defaultdict(int), defaultdict(int)
    prev_end = ev[lo - 1][1]
    overlap = 0
    for s, e, n in seg:
        g = s - prev_end
        k = short(n)
        if g >= 0:
            gap_by[k] += g
        else:
            overlap += -g
        n_by[k] += 1
        dur_by[k] += e - s
        prev_end = max(prev_end, e)
    print(f"{path}\n{steps} steps, {len(seg) / steps:.0f} kernels per step")
    print(f"per step: span {span / steps / 1e6:.3f} ms, kernel time {busy / steps / 1e6:.3f} ms, idle between kernels "
          f"{(span - busy + overlap) / steps / 1e6:.3f} ms, overlapped {overlap / steps / 1e6:.3f} ms")
    print("| kernel | calls/step | ms/step | avg us | gap in front: ms/step | avg gap us |\n|---|---|---|---|---|---|")
    for k in sorted(dur_by, key=lambda k: -dur_by[k] - gap_by[k]):
        c = n_by[k] / steps
        print(f"| {k} | {c:.1f} | {dur_by[k] / steps / 1e6:.3f} | {dur_by[k] / n_by[k] / 1e3:.1f} | {gap_by[k] / steps / 1e6:.3f} | "
              f"{gap_by[k] / n_by[k] / 1e3:.2f} |")


if __name__ == "__main__":
    main()
