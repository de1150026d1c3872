set -uo pipefail; cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp
A=gpurun_out/art; mkdir -p $A; R=r05
RP="rocprofv3 --kernel-trace --output-format csv"
rm -rf $A/prof_decode7b $A/prof_decode
timeout 600 $RP --stats -d $A/prof_decode7b -- python3 tools/bench_paths.py decode 16 qwen2.5-7b > $A/${R}_decode7b_under_rocprof.json 2> $A/prof_decode7b.err
timeout 600 $RP --stats -d $A/prof_decode -- python3 tools/bench_paths.py decode 16 > $A/${R}_decode_under_rocprof.json 2> $A/prof_decode.err
for p in decode decode7b; do f=$(find $A/prof_$p -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $A/${R}_${p}_kernel_stats.csv; done
find $A -name "*kernel_trace.csv" -delete; find $A -name "*agent_info.csv" -delete
timeout 1500 python bench.py > $A/bench_$R.json 2> $A/bench_$R.err
python -c "
import json; d=json.load(open('$A/bench_$R.json')); print(json.dumps(d['digest']))"
timeout 600 python tools/ab_decode_7b.py 2>/dev/null | head -6 > $A/r05_decode7b_ab.txt; cat $A/r05_decode7b_ab.txt
timeout 300 python tools/bench_attn_sp.py 2>/dev/null > $A/r05_attn_kernels.txt; cat $A/r05_attn_kernels.txt | head -3
