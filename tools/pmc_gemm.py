"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --output-format csv) of bench.py into
profiles/r01_gemm_pmc.json: memory-side bytes per GEMM launch with the gfx950 corrections of MI355X_MICROARCH.md."""
import csv, glob, json, sys

def per_launch(dirname, counter):
    f = glob.glob(dirname + "/*/*counter_collection.csv")[0]
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter or "gemm" not in r["Kernel_Name"] or "skinny" in r["Kernel_Name"]:
            continue
        tot += float(r["Counter_Value"])
        n += 1
    return tot / max(n, 1), n

fetch_kb, n1 = per_launch(sys.argv[1], "FETCH_SIZE")
write_kb, n2 = per_launch(sys.argv[2], "WRITE_SIZE")
out = {"kernel": "tasu_pipe::gemm_pipe_kernel + gemm_nt_kernel (all instantiations)", "launches": n1,
       "command": "rocprofv3 --pmc FETCH_SIZE (resp. WRITE_SIZE) --kernel-trace --output-format csv -- python3 bench.py --steps 2 "
                  "--warmup 1 --no-cpu-baseline --no-decode --no-graphs (two separate passes)",
       "FETCH_SIZE_KB_per_launch": round(fetch_kb, 1), "WRITE_SIZE_KB_per_launch": round(write_kb, 1),
       "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of 16-B/lane streaming reads (MI355X_MICROARCH.md, HBM section) "
                     "-> doubled; WRITE_SIZE exact",
       "traffic_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024),
       "note": "memory-side (fabric) bytes of the L2s; Infinity-Cache hits are included, so this is an upper bound on HBM bytes"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
