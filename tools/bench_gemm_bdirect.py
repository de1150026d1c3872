"""Prototype timing + correctness of the B-bypass GEMM (gemm_pipe.hip built with -DTASU_EXP_B_DIRECT: the weight operand in
fragment order, loaded global -> registers by the MFMA waves, 256 x 96 tiles) against the shipped kernel on the same shapes.
Run twice, once per library:
  TASU_GEMM_KERNEL=pipe TASU_GEMM_BN=96 python tools/bench_gemm_bdirect.py shipped
  TASU_LIB_PATH=ps_slm_amd/libtasu_exp_D.so TASU_GEMM_KERNEL=pipe TASU_GEMM_BN=96 python tools/bench_gemm_bdirect.py bdirect
Cold rotating operand sets, HIP events, results checked against torch.matmul."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.ops import HipOps

mode = sys.argv[1] if len(sys.argv) > 1 else "shipped"
ops = HipOps()
bf = torch.bfloat16


def frag_order(w, bn=96):
    """[N, K] row-major -> [tile_n][wave column][K-step][k-half][j][lane group][row][8] (gemm_pipe.hip, TASU_EXP_B_DIRECT)."""
    N, K = w.shape
    tn = (N + bn - 1) // bn
    wp = torch.zeros(tn * bn, K, dtype=w.dtype, device=w.device)
    wp[:N] = w
    ni = bn // 32
    return wp.view(tn, 2, ni, 16, K // 64, 2, 4, 8).permute(0, 1, 4, 5, 2, 6, 3, 7).contiguous()


for name, M, N, K in [("d_down", 4096, 8960, 1536), ("gate_up-like", 4096, 17920, 1536), ("lm_head-like", 2048, 151936 // 8, 1536),
                      ("down-like", 4096, 1536, 8960), ("sq8192", 8192, 8192, 8192)]:
    per_set = 2 * (M * K + N * K + M * N)
    nsets = max(2, min(12, -(-(3 << 29) // per_set)))
    sets = []
    for i in range(nsets):
        a = torch.randn(M, K, device="cuda").to(bf)
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(bf)
        sets.append((a, w, frag_order(w) if mode == "bdirect" else w, torch.empty(M, N, device="cuda", dtype=bf)))
    run = lambda s: ops.gemm(s[0], s[2] if mode == "bdirect" else s[1], s[3], M, N, K, ldb=K)
    for s in sets:
        run(s)
    torch.cuda.synchronize()
    a, w, _, c = sets[0]
    ref = a.float() @ w.float().t()
    err = float((c.float() - ref).abs().max() / ref.abs().max())
    del ref
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 30
    e0.record()
    for i in range(iters):
        run(sets[i % nsets])
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{mode:8s} {name:14s} M={M} N={N} K={K}  {ms * 1e3:8.1f} us  {2 * M * N * K / ms / 1e9:7.1f} TF/s   rel err {err:.1e}", flush=True)
    del sets
    torch.cuda.empty_cache()
