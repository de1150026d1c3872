# Last pass of a round: decode kernel statistics (1.5B, 7B), the bench line (timed), the same-process A/B logs.  Run through gpurun.
set -uo pipefail; cd "${GRAFT_REPO_ROOT:?run through gpurun}"; export TMPDIR=/tmp
A=gpurun_out/art; mkdir -p $A; R=${R:-r06}
RP="rocprofv3 --kernel-trace --output-format csv"
rm -rf $A/prof_decode7b $A/prof_decode
timeout 600 $RP --stats -d $A/prof_decode7b -- python3 tools/bench_paths.py decode 16 qwen2.5-7b > $A/${R}_decode7b_under_rocprof.json 2> $A/prof_decode7b.err
timeout 600 $RP --stats -d $A/prof_decode -- python3 tools/bench_paths.py decode 16 > $A/${R}_decode_under_rocprof.json 2> $A/prof_decode.err
for p in decode decode7b; do f=$(find $A/prof_$p -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $A/${R}_${p}_kernel_stats.csv; done
find $A -name "*kernel_trace.csv" -delete; find $A -name "*agent_info.csv" -delete
t0=$(date +%s); timeout 1500 python bench.py > $A/bench_$R.json 2> $A/bench_$R.err; echo "bench.py wall seconds: $(( $(date +%s) - t0 ))" | tee $A/bench_${R}_wall.txt
python -c "
import json; d=json.load(open('$A/bench_$R.json')); print(json.dumps(d['digest']))"
timeout 600 python tools/ab_decode_7b.py 2>/dev/null | head -6 > $A/${R}_decode7b_ab.txt; cat $A/${R}_decode7b_ab.txt
for v in 0 1 0 1; do echo "TASU_DEC_PRENORM_IN=$v $(TASU_DEC_PRENORM_IN=$v python tools/bench_paths.py decode 16 2>/dev/null | cut -c100-190)"; done | tee $A/${R}_decode_prenorm_ab.txt
for v in 0 1 0 1; do echo "TASU_DEC_PRENORM=$v $(TASU_DEC_PRENORM=$v python tools/bench_paths.py decode 16 2>/dev/null | cut -c100-190)"; done | tee -a $A/${R}_decode_prenorm_ab.txt
