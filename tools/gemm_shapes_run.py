"""The profiled program of the per-shape GEMM traffic measurement: launches every GEMM shape of the 1.5B benchmark step REPS
times in a fixed order on rotating (cold) operand sets; a tiny marker launch (cast_f32_bf16_kernel) follows every GEMM call, so
that the GEMM dispatches between two markers are one call (the column-split policy makes two launches out of some calls).
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir1> -- python3 tools/gemm_shapes_run.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dir2> -- python3 tools/gemm_shapes_run.py
    python tools/pmc_shapes.py <dir1> <dir2> profiles/r03_gemm_pmc.json"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.ops import GEMM_BF16, GEMM_F32, GEMM_RESID, HipOps

REPS = 3
M = 4096
# (name, M, N, K, mode, swiglu): the launches of one training step (labelled-rows lm_head: 2048 rows)
SHAPES = [("qkv", M, 2048, 1536, GEMM_BF16, False), ("o", M, 1536, 1536, GEMM_RESID, False), ("gate_up+swiglu", M, 8960, 1536, GEMM_BF16, True),
          ("down", M, 1536, 8960, GEMM_RESID, False), ("d_down", M, 8960, 1536, GEMM_BF16, False), ("d_gate_up", M, 1536, 17920, GEMM_BF16, False),
          ("d_o", M, 1536, 1536, GEMM_BF16, False), ("d_qkv", M, 1536, 2048, GEMM_BF16, False), ("lm_head_labelled", 2048, 151936, 1536, GEMM_BF16, False),
          ("wgrad1", 2048, 25088, 1664, GEMM_F32, False), ("proj1", 1664, 2048, 25088, GEMM_BF16, False)]

if __name__ == "__main__":
    ops = HipOps()
    bf = torch.bfloat16
    for name, m, n, k, mode, sw in SHAPES:
        rows = 2 * n if sw else n
        a = [torch.randn(m, k, device="cuda").to(bf) for _ in range(REPS)]
        b = [(torch.randn(rows, k, device="cuda") * k ** -0.5).to(bf) for _ in range(REPS)]
        c = torch.empty(m, rows, device="cuda", dtype=torch.float32 if mode != GEMM_BF16 else bf)
        r = torch.zeros(m, n, device="cuda") if mode == GEMM_RESID else None
        act = torch.empty(m, n, device="cuda", dtype=bf) if sw else None
        mk_in, mk_out = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda", dtype=bf)
        torch.cuda.synchronize()
        for i in range(REPS):
            if sw:
                ops.gemm_gate_up_swiglu(a[i], b[i], c, act, m, n, k)
            else:
                ops.gemm(a[i], b[i], c, m, n, k, resid=r, mode=mode)
            ops.cast_bf16(mk_in, mk_out)                   # marker: end of this call's dispatches
        torch.cuda.synchronize()
        del a, b, c
        torch.cuda.empty_cache()
