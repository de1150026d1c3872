# kernel-only times of the fp32 decode-step GEMMs per shape (rocprofv3 kernel trace of tools/bench_f32_stream.py), one run per
# (the switches live in the lab build: make -C ps_slm_amd/csrc lab; the script loads ps_slm_amd/libtasu_hip_lab.so)
# setting: DBGS="0 1 2" (TASU_F32_STREAM_DBG probes) and/or KSS="1 2 3" (TASU_F32_STREAM_KS); TILE=1 adds the tile kernel.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
make -C ps_slm_amd/csrc lab > /dev/null 2>&1; export TASU_LIB_PATH=$GRAFT_REPO_ROOT/ps_slm_amd/libtasu_hip_lab.so
run() {   # tag, env assignment
  env $2 true
  ( export $2; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/f32s -o $1 -- python3 tools/bench_f32_stream.py > gpurun_out/f32s_$1.log 2>&1 )
  python3 - $1 <<P
import csv,collections,glob,sys
tag=sys.argv[1]
f=glob.glob("gpurun_out/f32s/**/%s_kernel_trace.csv"%tag,recursive=True)[0]
g=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "f32_stream" in r["Kernel_Name"] or "f32_gemm_kernel" in r["Kernel_Name"]:
        g[(r["Kernel_Name"][10:46], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(g.items()):
    v=sorted(v); n=len(v)
    print(tag,k,n,"q10",round(v[n//10],1),"med",round(v[n//2],1),"q90",round(v[9*n//10],1))
P
}
for d in ${DBGS:-}; do run d$d TASU_F32_STREAM_DBG=$d; done
for k in ${KSS:-}; do run k$k TASU_F32_STREAM_KS=$k; done
if [ -n "${TILE:-}" ]; then run tile TASU_F32_STREAM=0; fi
