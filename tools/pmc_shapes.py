"""Per-shape memory-side bytes of the training step's GEMMs from two rocprofv3 --pmc passes of tools/gemm_shapes_run.py
(FETCH_SIZE, WRITE_SIZE; gfx950 corrections of MI355X_MICROARCH.md: FETCH_SIZE x2 for 16-B/lane streaming reads, WRITE_SIZE exact;
both in KB) -> JSON with measured / algorithmic bytes per shape."""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_shapes_run import REPS, SHAPES


def per_call(dirname, counter):
    """Counter sums of the GEMM dispatches of every call (dispatches between two marker launches, in dispatch order)."""
    f = glob.glob(dirname + "/*/*counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    calls, cur, n = [], 0.0, 0
    for r in rows:
        name = r["Kernel_Name"]
        if "gemm_pipe_kernel" in name or "gemm_pp_kernel" in name or "gemm_nt_kernel" in name:
            cur += float(r["Counter_Value"])
            n += 1
        elif "cast_f32_bf16_kernel" in name and n:
            calls.append((cur, n))
            cur, n = 0.0, 0
    return calls


fetch_c, write_c = per_call(sys.argv[1], "FETCH_SIZE"), per_call(sys.argv[2], "WRITE_SIZE")
assert len(fetch_c) == len(write_c) == REPS * len(SHAPES), (len(fetch_c), len(write_c))
fetch, write, launches = [c[0] for c in fetch_c], [c[0] for c in write_c], [c[1] for c in fetch_c]
out = {"command": "rocprofv3 --pmc FETCH_SIZE (resp. WRITE_SIZE) --kernel-trace --output-format csv -- python3 tools/gemm_shapes_run.py "
                  "(two separate passes; cold, rotating operand sets; the last of 3 launches per shape is reported)",
       "correction": "gfx950: FETCH_SIZE x2 (16-B/lane streaming reads are tallied at half), WRITE_SIZE exact; fabric-side L2 misses, "
                     "Infinity-Cache hits included", "shapes": {}}
for i, (name, m, n, k, mode, sw) in enumerate(SHAPES):
    j = i * REPS + REPS - 1
    rows = 2 * n if sw else n
    out_bytes = m * rows * (2 if mode == 0 else 4) + (m * n * 2 if sw else 0) + (m * n * 4 if mode == 2 else 0)
    algo = (m * k + rows * k) * 2 + out_bytes + (m * n * 4 if mode == 2 else 0)
    meas_f, meas_w = 2 * fetch[j] * 1024, write[j] * 1024
    out["shapes"][name] = {"M": m, "N": rows, "K": k, "fetch_bytes": int(meas_f), "write_bytes": int(meas_w),
                           "kernel_launches": launches[j], "algorithmic_bytes": int(algo), "ratio": round((meas_f + meas_w) / algo, 2)}
per_layer = {"qkv", "o", "gate_up+swiglu", "down", "d_down", "d_gate_up", "d_o", "d_qkv"}
tot = sum((28 if k in per_layer else 1) * (v["fetch_bytes"] + v["write_bytes"]) for k, v in out["shapes"].items())
cnt = sum(28 if k in per_layer else 1 for k in out["shapes"])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ps_slm_amd._lib import gemm_source_hash  # noqa: E402
out["gemm_source_hash"] = gemm_source_hash()            # the kernels these counters describe (bench.py checks it)
out["traffic_bytes_per_launch"] = int(tot / cnt)       # per GEMM CALL (a split call is two kernel launches): call-count-weighted mean of one step's GEMMs
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
