"""Is a GEMM shape bound by what crosses the fabric?  Runs it on its real operands (cold, rotating sets) and on ALIASED operands
(lda = 0 / ldb = 0: every row of A / B is the same K bytes, so that operand costs the fabric nothing while the kernel issues the very
same loads, MFMAs and stores).  A shape whose time drops with an operand aliased is paying for that operand's fabric traffic."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.ops import HipOps

ops = HipOps()
bf = torch.bfloat16
SHAPES = {"d_gate_up": (4096, 1536, 17920), "gate_up_plain": (4096, 17920, 1536), "down": (4096, 1536, 8960), "d_down": (4096, 8960, 1536),
          "qkv": (4096, 2048, 1536)}
if len(sys.argv) > 1 and sys.argv[1] == "encoder":          # the SANM encoder's GEMMs (16 x 504 frames)
    SHAPES = {"enc_qkv": (8064, 1536, 512), "enc_out": (8064, 512, 512), "enc_w1": (8064, 2048, 512), "enc_w2": (8064, 512, 2048)}
NSETS = 6


def timed(fn, n=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, (M, N, K) in SHAPES.items():
    A = [torch.randn(M, K, device="cuda").to(bf) for _ in range(NSETS)]
    B = [(torch.randn(N, K, device="cuda") * K ** -0.5).to(bf) for _ in range(NSETS)]
    C = torch.empty(M, N, dtype=bf, device="cuda")
    out = {}
    for tag, la, lb in (("real", None, None), ("A_aliased", 0, None), ("B_aliased", None, 0), ("both_aliased", 0, 0)):
        out[tag] = round(timed(lambda i=0: ops.gemm(A[i % NSETS], B[i % NSETS], C, M, N, K, lda=la, ldb=lb)), 1)
    print(json.dumps({"shape": name, "M": M, "N": N, "K": K, "us": out}), flush=True)
    del A, B, C
