"""Two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --output-format csv) of tools/lora_kernels_run.py -> memory-side bytes per
launch of the round-4 kernels next to their algorithmic bytes (gfx950: FETCH_SIZE doubled, MI355X_MICROARCH.md)."""
import collections, csv, glob, json, re, sys
def per_kernel(dirname, counter):
    f = glob.glob(dirname + "/**/*counter_collection.csv", recursive=True)[0]
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            m = re.search(r"(rank_gemm_tn_kernel<[^>]*>|rank_gemm_kernel<[^>]*>|lora_apply_kernel<[^>]*>|psd_[a-z_]+_kernel|dropout_[a-z_0-9]+_kernel|lora_refresh_kernel)", r["Kernel_Name"])
            if m:
                d[(m.group(1), r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
    return d
fe, wr = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = []
for key in sorted(set(fe) | set(wr)):
    name, grid = key
    f = fe.get(key, [0.0]); w = wr.get(key, [0.0])
    out.append({"kernel": name, "grid": grid, "launches": len(f), "fetch_bytes_per_launch": int(2 * 1024 * sum(f) / len(f)),
                "write_bytes_per_launch": int(1024 * sum(w) / len(w))})
json.dump({"command": "rocprofv3 --pmc FETCH_SIZE (resp. WRITE_SIZE) --kernel-trace --output-format csv -- python3 tools/lora_kernels_run.py (two passes)",
           "correction": "gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md); memory-side bytes of the L2s, Infinity-Cache hits included",
           "algorithmic": {"rank_gemm K=1536": "A 12.6 MB + B 0.2 MB read, 0.5 MB written", "rank_gemm K=8960": "A 73.4 MB + B 1.1 MB read, 0.5 MB written",
                           "rank_gemm_tn out=1536 / 8960 (K = 4096 rows)": "At 12.6 / 73.4 MB + B 0.5 MB read, 0.4 / 2.3 MB written",
                           "lora_apply N=1536": "y 12.6 MB read + 12.6 MB written (+ u 0.5 MB, W 0.2 MB)", "lora_apply N=8960": "y 73.4 MB read + 73.4 MB written",
                           "psd_logit_stats": "logits 401 MB read once (running max / sum)", "psd_gather_softmax": "kept frames' logits + fp32 rows written"},
           "kernels": out}, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
