# LDS counters of the attention kernels (one shape), for the current build and, if given, another library
set -euo pipefail; cd "${GRAFT_REPO_ROOT:?run through gpurun (it exports GRAFT_REPO_ROOT)}"; export TMPDIR=/tmp; set +e   # (the runs below report their own exit codes)
O=gpurun_out/pmc_attn; mkdir -p $O
cat > /tmp/one_shape.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from ps_slm_amd.ops import HipOps
ops = HipOps(); HD = 128; bf = torch.bfloat16
B, S, H, G = 16, 256, 12, 2
M, LD, Spad = B * S, (H + 2 * G) * HD, 256
qkv = torch.randn(M, LD, device="cuda").to(bf); dout = torch.randn(M, H * HD, device="cuda").to(bf)
km = torch.ones(B, Spad, dtype=torch.uint8, device="cuda")
out, lse, delta = torch.zeros(M, H * HD, dtype=bf, device="cuda"), torch.zeros(B * H * Spad, device="cuda"), torch.zeros(B * H * Spad, device="cuda")
cos, sin = torch.ones(M, 64, device="cuda"), torch.zeros(M, 64, device="cuda")
dqkv = torch.zeros(M, LD, dtype=bf, device="cuda"); dkp, dvp = torch.zeros(M, H * HD, device="cuda"), torch.zeros(M, H * HD, device="cuda")
for _ in range(3):
    ops.attn_fwd(qkv, None, km, out, lse, B, S, H, G, HD ** -0.5, True)
    ops.attn_bwd_prep(dout, out, delta, None, B, S, H)
    for k in (1, 2):
        ops.attn_bwd_rope(qkv, km, dout, lse, delta, cos, sin, dqkv, dkp, dvp, B, S, H, G, HD ** -0.5, True, k)
torch.cuda.synchronize()
PY
for tag in new old; do
  lib=ps_slm_amd/libtasu_hip.so; [ $tag = old ] && lib=ps_slm_amd/libtasu_hip_oldswz.so
  [ -f $lib ] || continue
  TASU_LIB_PATH=$lib rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/$tag -- python3 /tmp/one_shape.py > /dev/null 2> $O/$tag.err
  python3 - $tag <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/pmc_attn/{tag}/*/*counter_collection.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0][-40:]
    if "attn" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print(tag, k, {c: round(v / n[(k, c)]) for c, v in d.items()})
PY
done
