"""MFMA-busy evidence for the GEMM launches from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES,
SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY, GRBM_GUI_ACTIVE; --output-format csv) of bench.py -> profiles/r01_gemm_mfma_pmc.json.
MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): cycles in which a SIMD's matrix pipe is
busy over the cycles the kernel had on all 1024 SIMDs (MI355X_MICROARCH.md: GRBM_GUI_ACTIVE is summed over the 8 XCDs)."""
import csv, glob, json, sys, collections

# usage: pmc_mfma.py <pass dir> [<pass dir> ...] <out.json>   (counters may be split over passes of the same command)
acc = collections.defaultdict(float)
n = 0
for d in sys.argv[1:-1]:
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    seen = set()
    for r in csv.DictReader(open(f)):
        if "gemm" not in r["Kernel_Name"] or "skinny" in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        seen.add(r["Dispatch_Id"])
    n = len(seen)
cyc = acc["GRBM_GUI_ACTIVE"] / 8.0
out = {"kernel": "tasu_pipe::gemm_pipe_kernel + gemm_nt_kernel (all instantiations)", "launches": n,
       "command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
                  "GRBM_GUI_ACTIVE (two passes: the first four counters, then the three SQ_WAIT/ACTIVE ones) --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode --no-graphs",
       "sums_over_gemm_launches": {k: v for k, v in acc.items()},
       "kernel_cycles_per_launch": round(cyc / max(n, 1), 1),
       "mfma_busy_fraction": round(acc["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0), 4) if cyc else None,
       "wave_cycles_split": {"wait_any": round(acc["SQ_WAIT_ANY"] / max(acc["SQ_WAVE_CYCLES"], 1), 4),
                             "wait_inst_any": round(acc["SQ_WAIT_INST_ANY"] / max(acc["SQ_WAVE_CYCLES"], 1), 4),
                             "active_inst_any": round(acc["SQ_ACTIVE_INST_ANY"] / max(acc["SQ_WAVE_CYCLES"], 1), 4)},
       "note": "a 16x16x32 bf16 MFMA occupies its SIMD's matrix pipe for 16 cycles; the loader waves of gemm_pipe_kernel are "
               "parked in s_waitcnt / s_barrier by design, which is what wait_any mostly counts"}
json.dump(out, open(sys.argv[-1], "w"), indent=1)
print(json.dumps(out)[:900])
