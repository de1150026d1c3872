"""Per-kernel memory-side bytes from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --output-format csv) of ANY program:
kernels whose name matches the regex are grouped by (short name, grid size) -> launches, fetch / write bytes per launch
(gfx950: FETCH_SIZE doubled for wide streaming reads, WRITE_SIZE exact: MI355X_MICROARCH.md, HBM section).
    python tools/pmc_kernels.py <fetch_dir> <write_dir> '<regex>' <out.json> ['<command description>']"""
import collections, csv, glob, json, re, sys


def per_kernel(dirname, counter, rx):
    f = glob.glob(dirname + "/**/*counter_collection.csv", recursive=True)[0]
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        m = rx.search(r["Kernel_Name"])
        if m:
            d[(m.group(0), r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
    return d


rx = re.compile(sys.argv[3])
fe, wr = per_kernel(sys.argv[1], "FETCH_SIZE", rx), per_kernel(sys.argv[2], "WRITE_SIZE", rx)
out = []
for key in sorted(set(fe) | set(wr)):
    f, w = fe.get(key, [0.0]), wr.get(key, [0.0])
    out.append({"kernel": key[0], "grid": key[1], "launches": len(f), "fetch_bytes_per_launch": int(2 * 1024 * sum(f) / len(f)),
                "write_bytes_per_launch": int(1024 * sum(w) / len(w))})
json.dump({"command": sys.argv[5] if len(sys.argv) > 5 else "rocprofv3 --pmc FETCH_SIZE (resp. WRITE_SIZE) --kernel-trace --output-format csv (two passes)",
           "correction": "gfx950: FETCH_SIZE doubled (16-B/lane streaming reads are tallied at half), WRITE_SIZE exact; memory-side bytes of the L2s, "
                         "Infinity-Cache hits included (MI355X_MICROARCH.md, HBM)", "kernels": out}, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out)[:1500])
