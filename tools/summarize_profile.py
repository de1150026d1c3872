"""Turns a rocprofv3 --kernel-trace --stats CSV into the per-step markdown summary kept under profiles/."""
import csv, re, sys
path, steps, title = sys.argv[1], int(sys.argv[2]), sys.argv[3]
rows = list(csv.DictReader(open(path)))
out = [f"# {title}", f"# per-step figures = totals / {steps} steps (warm-up + timed); framework init-time kernels omitted", "",
       "| kernel | calls/step | ms/step | avg us | % of step kernels |", "|---|---|---|---|---|"]
tot, lines = 0.0, []
for r in rows:
    n = r["Name"]
    if "at::native" in n or "rocclr" in n:
        continue
    short = re.sub(r"\(anonymous namespace\)::|void |\(.*", "", n)
    if n.startswith("_ZN"):
        short = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
        short = re.sub(r"E[PK].*", "", short)
    if "gemm_nt" in n:
        short = "gemm_nt_kernel" + n[n.index("<"):n.index(">") + 1]
    ms = int(r["TotalDurationNs"]) / 1e6 / steps
    tot += ms
    lines.append((short, int(r["Calls"]) / steps, ms, float(r["AverageNs"]) / 1e3))
for s, c, ms, a in lines:
    out.append(f"| {s} | {c:.1f} | {ms:.3f} | {a:.1f} | {100 * ms / tot:.1f} |")
out.append(f"| **total** | | **{tot:.2f}** | | |")
g = [l for l in lines if l[0].startswith("gemm_nt") or "gemm_pipe_kernel" in l[0] or "gemm_pp_kernel" in l[0]]
if g:
    out += ["", f"GEMM kernels (gemm_pp_kernel / gemm_pipe_kernel / gemm_nt_kernel, all instantiations): {sum(l[1] for l in g):.0f} launches/step, {sum(l[2] for l in g):.2f} ms/step, "
            f"average launch {1e3 * sum(l[2] for l in g) / sum(l[1] for l in g):.1f} us"]
print("\n".join(out))
