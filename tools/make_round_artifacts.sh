set -euo pipefail; cd "${GRAFT_REPO_ROOT:?run through gpurun (it exports GRAFT_REPO_ROOT)}"; export TMPDIR=/tmp; set +e   # (the runs below report their own exit codes)
# Round artifacts in one GPU call: the bench line, rocprofv3 kernel statistics of the four measured programs, and the PMC passes
# (counters in their own runs, --kernel-trace only: never combined with the trace domains).  PART=bench|stats|pmc|all
R=${R:-r06}; PART=${PART:-all}
A=gpurun_out/art; mkdir -p $A
RP="rocprofv3 --kernel-trace --output-format csv"
if [ "$PART" = all ] || [ "$PART" = bench ]; then
  timeout 1500 python bench.py > $A/bench_$R.json 2> $A/bench_$R.err
  tail -c 1500 $A/bench_$R.json
fi
if [ "$PART" = all ] || [ "$PART" = stats ]; then
  timeout 600 $RP --stats -d $A/prof_bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-extra --no-graphs > $A/${R}_bench_under_rocprof.json 2> $A/prof_bench.err
  timeout 600 $RP --stats -d $A/prof_decode -- python3 tools/bench_paths.py decode 16 > $A/${R}_decode_under_rocprof.json 2> $A/prof_decode.err
  timeout 600 $RP --stats -d $A/prof_audio -- python3 bench.py --path audio --blank-biased --blank-bias 13.0 --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-extra --no-graphs > $A/${R}_audio_under_rocprof.json 2> $A/prof_audio.err
  timeout 600 $RP --stats -d $A/prof_lora -- python3 bench.py --lora --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-extra --no-graphs > $A/${R}_lora_under_rocprof.json 2> $A/prof_lora.err
  timeout 600 $RP --stats -d $A/prof_decode7b -- python3 tools/bench_paths.py decode 16 qwen2.5-7b > $A/${R}_decode7b_under_rocprof.json 2> $A/prof_decode7b.err
  timeout 600 $RP --stats -d $A/prof_decodef32 -- python3 tools/bench_decode_fp32.py > $A/${R}_decodef32_under_rocprof.json 2> $A/prof_decodef32.err
  for p in bench decode audio lora decode7b decodef32; do f=$(find $A/prof_$p -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $A/${R}_${p}_kernel_stats.csv; done
  # the graph-replayed step's timeline: kernel time against idle time between kernels (tools/trace_gaps.py)
  timeout 600 $RP -d $A/prof_gaps -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-decode --no-extra > /dev/null 2> $A/prof_gaps.err
  python tools/trace_gaps.py $A/prof_gaps 5 > $A/${R}_step_gaps.txt 2>&1
fi
if [ "$PART" = all ] || [ "$PART" = pmc ]; then
  # per-shape GEMM traffic (cold rotating operands), FETCH and WRITE in separate passes
  timeout 600 $RP --pmc FETCH_SIZE -d $A/pmc_gemm_f -- python3 tools/gemm_shapes_run.py > /dev/null 2> $A/pmc_gemm_f.err
  timeout 600 $RP --pmc WRITE_SIZE -d $A/pmc_gemm_w -- python3 tools/gemm_shapes_run.py > /dev/null 2> $A/pmc_gemm_w.err
  python tools/pmc_shapes.py $A/pmc_gemm_f $A/pmc_gemm_w $A/${R}_gemm_pmc.json > /dev/null 2> $A/pmc_shapes.err
  # matrix-pipe occupancy of the step's GEMM launches
  timeout 600 $RP --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $A/pmc_mfma1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode --no-extra --no-graphs > /dev/null 2> $A/pmc_mfma1.err
  timeout 600 $RP --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES -d $A/pmc_mfma2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode --no-extra --no-graphs > /dev/null 2> $A/pmc_mfma2.err
  python tools/pmc_mfma.py $A/pmc_mfma1 $A/pmc_mfma2 $A/${R}_gemm_mfma_pmc.json > /dev/null 2> $A/pmc_mfma.err
  # the decode step's kernels
  timeout 600 $RP --pmc FETCH_SIZE -d $A/pmc_dec_f -- python3 tools/bench_paths.py decode 16 > /dev/null 2> $A/pmc_dec_f.err
  timeout 600 $RP --pmc WRITE_SIZE -d $A/pmc_dec_w -- python3 tools/bench_paths.py decode 16 > /dev/null 2> $A/pmc_dec_w.err
  python tools/pmc_kernels.py $A/pmc_dec_f $A/pmc_dec_w 'stream_gemm_kernel<[^>]*>|stream_finish_norm[a-z_]*kernel<[^>]*>|attn_decode[a-z_]*|topk_[a-z]+_kernel|beam_update_kernel|rmsnorm_fwd_reg_kernel<[^>]*>|decode_step_prologue[a-z_]*' $A/${R}_decode_pmc.json "rocprofv3 --pmc FETCH_SIZE (resp. WRITE_SIZE) --kernel-trace --output-format csv -- python3 tools/bench_paths.py decode 16 (two passes)" > /dev/null 2> $A/pmc_dec.err
  # the audio-SFT step (encoder GEMMs, SANM attention, FSMN, PSD, frontend)
  timeout 600 $RP --pmc FETCH_SIZE -d $A/pmc_aud_f -- python3 bench.py --path audio --blank-biased --blank-bias 13.0 --steps 3 --warmup 1 --no-cpu-baseline --no-decode --no-extra --no-graphs > /dev/null 2> $A/pmc_aud_f.err
  timeout 600 $RP --pmc WRITE_SIZE -d $A/pmc_aud_w -- python3 bench.py --path audio --blank-biased --blank-bias 13.0 --steps 3 --warmup 1 --no-cpu-baseline --no-decode --no-extra --no-graphs > /dev/null 2> $A/pmc_aud_w.err
  python tools/pmc_kernels.py $A/pmc_aud_f $A/pmc_aud_w 'gemm_pp_kernel<[^>]*>|gemm_pipe_kernel<[^>]*>|attn_fwd_kernel<[^>]*>|fsmn_ln_rows_kernel|psd_[a-z_]+_kernel|layernorm_[a-z_]+kernel[<a-z>]*' $A/${R}_audio_pmc.json "rocprofv3 --pmc FETCH_SIZE (resp. WRITE_SIZE) --kernel-trace --output-format csv -- python3 bench.py --path audio --blank-biased --blank-bias 13.0 --steps 3 --warmup 1 --no-graphs ... (two passes)" > /dev/null 2> $A/pmc_aud.err
fi
find $A -name "*kernel_trace.csv" -delete; find $A -name "*counter_collection.csv" -delete; find $A -name "*agent_info.csv" -delete
ls $A | head -40; du -sh $A
