"""Thin operator layer over the C-ABI (include/tasu_hip.h).  Tensors are containers only: every method passes
``data_ptr()`` + sizes + the current HIP stream to libtasu_hip.so and returns nothing (outputs are
caller-allocated).  There is no eager/PyTorch fallback: construction fails if the library is missing.

The method set is the operator interface of the model code (ps_slm_amd/model.py); tests exercise the same
model code on CPU by injecting tests/fake_ops.py, a torch-CPU double with identical signatures.
"""
import contextlib
import ctypes
import os

import torch

from . import _lib

GEMM_BF16, GEMM_F32, GEMM_RESID = 0, 1, 2
GEMM_WS_BYTES = (64 << 20) + 4096 * 4          # tasu_gemm_nt_bf16_ws workspace: counters + split-K partial tiles
LN_BWD_SPLIT = 16


class F32Fragments:
    """An fp32 weight matrix [N, K] re-laid for the decode step's streaming GEMM (tasu_f32_to_fragment_order); the f32_gemm*
    wrappers pass it with ldw = TASU_F32_LDW_FRAGMENT.  ``rows``: the row-major matrix it was made from (problems the streaming
    kernel does not serve -- more than 64 rows -- take that one: ``HipOps.f32_weight``)."""
    __slots__ = ("t", "N", "K", "rows")

    def __init__(self, t, N, K, rows):
        self.t, self.N, self.K, self.rows = t, N, K, rows


def _wl(w):
    """(pointer, leading dimension) of an fp32 weight operand"""
    return (_p(w.t), -1) if isinstance(w, F32Fragments) else (_p(w), w.stride(0))


class TasuOpError(RuntimeError):
    pass


def _p(t):
    return None if t is None else t.data_ptr()


class HipOps:
    name = "hip"

    def __init__(self):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise TasuOpError("HipOps needs a ROCm device (torch.cuda.is_available() is False)")
        # split-K workspace of tasu_gemm_nt_bf16_ws: zeroed arrival counters + fp32 partial tiles (include/tasu_hip.h).
        # Allocated up front (never inside a hipGraph capture); all GEMMs of one HipOps run on one stream.
        self.gemm_ws = torch.zeros(GEMM_WS_BYTES, dtype=torch.uint8, device="cuda")
        self._gemm_ws_alt = {}         # further ones for launch sequences that may run on another stream (alt_workspace)
        self.topk_ws = torch.empty(1024 * 16 * 34, dtype=torch.float32, device="cuda")   # tasu_logprob_topk partials, M <= 1024
        # decode-step GEMMs: the single-launch weight-streaming kernels (csrc/gemm_stream.hip) where they serve the shape,
        # the split-K + finish kernels (csrc/gemm_skinny.hip) otherwise.
        self.use_stream = True          # (tests / tools flip the attribute for A/B runs)
        self.dec_frag = False          # set by begin_decode(): decode activations travel in fragment order
        self.dec_frag_act = False      # ... including the MLP activation that feeds the down projection
        self.dec_down_slabs = True     # down projection as K-range slabs + tasu_stream_finish_norm (False: split-K kernels)
        self.dec_split_order = (1, 7, 5, 2, 3, 4, 6, 8, 13)   # K ranges tried in this order (K = 8960: 7 x 1280 before 5 x 1792; A/B: swap;
                                                              # 13: Qwen2.5-7B's K = 18944 = 12 x 1536 + 512, a ragged split)
        # the post-attention norm without a launch: the o projection writes bf16(norm_w . x) + per-tile sums of squares, gate|up scales
        # its accumulators by rstd (tasu_gemm_stream_resid_prenorm / _swiglu_rstd): 6 launches per layer, 1.684 -> 1.620 ms per
        # position at 1.5B (MI355X, alternating processes on one box).  No hand-off inside a launch (the variant below lost to that).
        # False / TASU_DEC_PRENORM=0: the separate norm launch (A/B, tests).
        self.dec_prenorm = os.environ.get("TASU_DEC_PRENORM", "1") != "0"
        self.dec_prenorm_in = os.environ.get("TASU_DEC_PRENORM_IN", "1") != "0"     # ... and the input norm (slab finish -> q|k|v)
        self.dec_sumsq = None
        self.dec_sumsq_in = None
        self.dec_stream_7b = True      # K = 3584 in one range / K = 18944 as a ragged split on the streaming kernels (False: A/B runs,
                                       # the split-K kernels of gemm_skinny.hip as before round 5)
        self._frag = {}                # row-major weight address -> (fragment-order copy, the row-major tensor)
        self.dec_prologue = True       # the position's five set-up launches as one (tasu_decode_step_prologue)
        self.attn_kernel = {"fwd": "policy", "bwd": "policy"}     # tools / tests: force an attention kernel family for A/B runs
        # the RMSNorm behind the o / down projection inside the projection's launch (tasu_gemm_stream_norm: write-through tiles, a
        # ticket per workgroup, the last arrivers normalise the rows).  Bit-identical, 5 launches per layer instead of 7 -- and
        # SLOWER: 1.776 against 1.702 ms per position (MI355X, round 5): ~100-250 returning atomics on one counter, the poll and
        # the sc1 round trip for the row cost what the launch boundary they replace costs.  Off; kept for A/B runs.
        self.dec_fused_norm = False
        self.norm_sync = torch.zeros(4, dtype=torch.int32, device="cuda")     # its two ticket words (zero between launches)

    # ------------------------------------------------------------------ plumbing
    @contextlib.contextmanager
    def alt_workspace(self, who):
        """GEMMs issued inside use the split-K / stream-K workspace named ``who`` instead of the common one.  include/tasu_hip.h:
        launches sharing a workspace must be ordered on one stream -- the frozen encoder's pass may run on a side stream next to the
        decoder step's GEMMs (TasuModel.prefetch_encoder), and so do the adapters' weight gradients (ps_slm_amd/lora.py; ranks
        above 64 go through the tile kernels), so their launches (eager or captured) never touch the workspace of the rest.
        Allocated on first use, which is an eager pass (never inside a capture: the first pass of a shape is eager)."""
        if who not in self._gemm_ws_alt:
            self._gemm_ws_alt[who] = torch.zeros(GEMM_WS_BYTES, dtype=torch.uint8, device="cuda")
        keep, self.gemm_ws = self.gemm_ws, self._gemm_ws_alt[who]
        try:
            yield
        finally:
            self.gemm_ws = keep

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    @staticmethod
    def _chk(rc, what):
        if rc != 0:
            raise TasuOpError(f"{what} failed with code {rc} ({'bad argument' if rc == 1 else 'launch failure'})")

    # ------------------------------------------------------------------ GEMM & layout
    def gemm(self, a, b, c, M, N, K, bias=None, resid=None, mode=GEMM_BF16, lda=None, ldb=None, ldc=None):
        """c[M,N] = a[M,K] @ b[N,K]^T (+bias).  a/b bf16 (leading dims default to their row strides)."""
        lda = a.stride(0) if lda is None else lda
        ldb = b.stride(0) if ldb is None else ldb
        ldc = c.stride(0) if ldc is None else ldc
        self._chk(self.lib.tasu_gemm_nt_bf16_ws(_p(a), lda, _p(b), ldb, _p(c), ldc, _p(bias), _p(resid), M, N, K, mode,
                                                _p(self.gemm_ws), GEMM_WS_BYTES, self._stream()), "tasu_gemm_nt_bf16_ws")

    def gemm_bias_relu(self, a, b, c, M, N, K, bias):
        """c = relu(a @ b^T + bias), bf16: the FFN's w_1 + ReLU of the SANM encoder in one launch (tasu_gemm_bias_relu_bf16)."""
        self._chk(self.lib.tasu_gemm_bias_relu_bf16(_p(a), a.stride(0), _p(b), b.stride(0), _p(c), c.stride(0), _p(bias), M, N, K,
                                                    _p(self.gemm_ws), GEMM_WS_BYTES, self._stream()), "tasu_gemm_bias_relu_bf16")

    GEMM_KERNELS = {"pp256": 1, "pipe128": 2, "pipe192": 3, "pipe96": 4}

    def gemm_on(self, kernel, a, b, c, M, N, K, bias=None, resid=None, mode=GEMM_BF16):
        """gemm() on a NAMED kernel (tasu_gemm_nt_bf16_kernel), regardless of the dispatcher's tile policy: tests and tuning."""
        self._chk(self.lib.tasu_gemm_nt_bf16_kernel(_p(a), a.stride(0), _p(b), b.stride(0), _p(c), c.stride(0), _p(bias), _p(resid),
                                                    M, N, K, mode, self.GEMM_KERNELS[kernel], self._stream()), "tasu_gemm_nt_bf16_kernel")

    def gemm_streamk(self, a, b, c, M, N, K, bias=None, resid=None, mode=GEMM_BF16):
        """gemm() on the 256 x 256 kernel with the stream-K schedule (tasu_gemm_nt_bf16_streamk): tests and tuning."""
        self._chk(self.lib.tasu_gemm_nt_bf16_streamk(_p(a), a.stride(0), _p(b), b.stride(0), _p(c), c.stride(0), _p(bias), _p(resid),
                                                     M, N, K, mode, _p(self.gemm_ws), GEMM_WS_BYTES, self._stream()),
                  "tasu_gemm_nt_bf16_streamk")

    def gemm_splitk(self, a, b, c, M, N, K, ksplit, ws):
        """c[M,N] (bf16) = a[M,K] @ b[N,K]^T with the K range cut into ``ksplit`` work items per output tile (fp32 partial
        matrices in ``ws``, summed in order, one bf16 rounding): for outputs that cannot fill the chip behind a very long K."""
        self._chk(self.lib.tasu_gemm_nt_bf16_splitk(_p(a), a.stride(0), _p(b), b.stride(0), _p(ws), N, M, N, K, ksplit,
                                                    self._stream()), "tasu_gemm_nt_bf16_splitk")
        self._chk(self.lib.tasu_sum_slabs_bf16(_p(ws), ksplit, M * N, _p(c), M * N, self._stream()), "tasu_sum_slabs_bf16")

    def gemm_slabs(self, a, b, ws, M, N, K, ksplit):
        """ws[ks][M, N] (fp32) = a[:, ks-th K range] @ b[:, ks-th K range]^T on 256 x 256 tiles (tasu_gemm_nt_bf16_slabs); the
        consumer sums the slabs."""
        self._chk(self.lib.tasu_gemm_nt_bf16_slabs(_p(a), a.stride(0), _p(b), b.stride(0), _p(ws), N, M, N, K, ksplit,
                                                   self._stream()), "tasu_gemm_nt_bf16_slabs")

    def sum_slabs(self, ws, n_slabs, out, n):
        self._chk(self.lib.tasu_sum_slabs_bf16(_p(ws), n_slabs, n, _p(out), n, self._stream()), "tasu_sum_slabs_bf16")

    def gemm_gate_up_swiglu(self, a, wgu, gu, act, M, I, K):
        """gu[M,2I] = a @ wgu^T and act[M,I] = swiglu(gu) in one launch (training step)."""
        if act.stride(0) != I:                          # act is the head of a wider operand buffer
            return self._chk(self.lib.tasu_gemm_gate_up_swiglu_ld(_p(a), a.stride(0), _p(wgu), wgu.stride(0), _p(gu), _p(act), act.stride(0), M, I,
                                                                  K, _p(self.gemm_ws), GEMM_WS_BYTES, self._stream()), "tasu_gemm_gate_up_swiglu_ld")
        self._chk(self.lib.tasu_gemm_gate_up_swiglu_ws(_p(a), a.stride(0), _p(wgu), wgu.stride(0), _p(gu), _p(act), M, I, K,
                                                       _p(self.gemm_ws), GEMM_WS_BYTES, self._stream()),
                  "tasu_gemm_gate_up_swiglu_ws")

    def gemm_qkv_rope(self, a, wqkv, bias, qkv, cos, sin, M, H, G, K):
        """qkv[M, (H+2G)*128] = rope(a @ wqkv^T + bias) in one launch (tasu_gemm_qkv_rope): the training step's / prefill's
        q|k|v projection with the rotary embedding of the q and k heads in the GEMM's epilogue."""
        self._chk(self.lib.tasu_gemm_qkv_rope(_p(a), a.stride(0), _p(wqkv), wqkv.stride(0), _p(bias), _p(qkv), _p(cos), _p(sin), M, H, G,
                                              K, _p(self.gemm_ws), GEMM_WS_BYTES, self._stream()), "tasu_gemm_qkv_rope")

    def gemm_dswiglu(self, dy, wd_t, gu, dgu, dact_ws, M, I, K):
        """dgu[M,2I] = swiglu_bwd(dy @ wd_t^T, gu): the down projection's input gradient with the SwiGLU backward in the GEMM's
        epilogue (tasu_gemm_dswiglu); dact_ws: bf16 [M, I] scratch for the shapes the fused epilogue does not serve."""
        self._chk(self.lib.tasu_gemm_dswiglu(_p(dy), dy.stride(0), _p(wd_t), wd_t.stride(0), _p(gu), _p(dgu), _p(dact_ws), M, I, K,
                                             _p(self.gemm_ws), GEMM_WS_BYTES, self._stream()), "tasu_gemm_dswiglu")

    # ------------------------------------------------------------------ decode-step GEMMs
    def _stream_split(self, K):
        """K ranges the streaming kernels can take this K in (0 = not served)."""
        if not self.use_stream or (not self.dec_stream_7b and K in (3584, 18944)):
            return 0
        for ks in self.dec_split_order:
            if self.lib.tasu_stream_supported(K, ks):        # (equal ranges, or ks - 1 equal ranges and a shorter last one)
                return ks
        return 0

    def begin_decode(self, D, HHD, I):
        """Called once per generate(): decides whether the decode step's bf16 activations (normed hidden state, attention
        output, MLP activation) travel between its kernels in FRAGMENT ORDER (include/tasu_hip.h) -- possible when the
        streaming kernels serve every GEMM of the layer.  Returns that decision."""
        self.dec_frag = bool(self._stream_split(D) == 1 and self._stream_split(HHD) == 1 and D % 32 == 0 and HHD % 32 == 0)
        # the MLP activation (the down projection's input): fragment order when the down projection runs on the streaming
        # kernels -- in one K range, or (K = 8960 = 5 x 1792) as K-range slabs + tasu_stream_finish_norm.  The slabs are the
        # default since the two row halves of a K range share an XCD's L2 and the slabs are row-major (1.89 vs 2.0 ms per
        # position at 1.5B against the split-K kernels of gemm_skinny.hip; the dec_down_slabs attribute selects, for A/B runs).
        if self.dec_prenorm and self.dec_sumsq is None:    # (here, not at first use: a decode step may be under hipGraph capture)
            self.dec_sumsq = torch.zeros(4096 // 16 * 64, dtype=torch.float32, device="cuda")
            self.dec_sumsq_in = torch.zeros(8, 4096 // 16 * 64, dtype=torch.float32, device="cuda")    # per 64-row chunk (input norms)
        ks_down = self._stream_split(I)
        # (the same predicate gemm_skinny_norm routes by: a K-split down projection reads a fragment-order activation only on the
        # slab route, whose finish serves N = D in 256 * {1, 2, 6, 7, 14} -- ADVICE r5: any other geometry keeps row-major)
        self.dec_frag_act = bool(self.dec_frag and I % 32 == 0 and
                                 (ks_down == 1 or (ks_down > 1 and self.dec_down_slabs and self._slab_finish_serves(D))))
        return self.dec_frag

    @staticmethod
    def _slab_finish_serves(N):
        """Output widths tasu_stream_finish_norm / _prenorm serve (a wave per 256 columns of a row)."""
        return N % 256 == 0 and N // 256 in (1, 2, 6, 7, 14)

    def end_decode(self):
        self.dec_frag = self.dec_frag_act = False

    def decode_frag_possible(self, D, HHD):
        """The predicate begin_decode() uses for fragment-order operands: only then is a fragment-order weight copy ever read."""
        return bool(self._stream_split(D) == 1 and self._stream_split(HHD) == 1 and D % 32 == 0 and HHD % 32 == 0)

    def forget_decode_weights(self, ptrs):
        """Drops the fragment-order copies registered for these row-major weight addresses (a model reloading its weights)."""
        for p in ptrs:
            self._frag.pop(p, None)

    def register_decode_weight(self, w, kind, N, H=0, G=0, slabs_ok=False):
        """Load-time: a fragment-order copy of a decode-step weight (kind: 'plain' | 'swiglu' | 'qkv'), looked up by the
        row-major tensor's address when a streaming GEMM is called with it.  Only for a K the streaming kernels take in ONE
        range -- or, ``slabs_ok`` (the down projection), in K-range slabs: no other copy is ever handed out by _wf()."""
        ks = self._stream_split(w.shape[1])
        if not self.use_stream or w.data_ptr() in self._frag or ks == 0 or (ks > 1 and not slabs_ok):
            return
        out = torch.empty(w.numel() if kind != "plain" else ((N + 15) // 16) * 16 * w.shape[1], dtype=torch.bfloat16, device=w.device)
        code = {"plain": 0, "swiglu": 2, "qkv": 3}[kind]
        self._chk(self.lib.tasu_to_fragment_order(_p(w), w.stride(0), _p(out), code, N, w.shape[1], H, G, self._stream()),
                  "tasu_to_fragment_order")
        self._frag[w.data_ptr()] = (out, w)          # keeps the row-major tensor alive: its address is the key

    def _wf(self, w):
        """(weight pointer, layout flag): the fragment-order copy while a decode runs in fragment order (the activations and
        the weights of a streaming GEMM change layout together), else the row-major tensor."""
        f = self._frag.get(w.data_ptr()) if self.dec_frag else None
        return (f[0], 1) if f is not None else (w, 0)

    def dec_rmsnorm(self, x, w, y, eps):
        """y = rmsnorm(x) for the decode step: fragment order when the layer runs on the streaming kernels."""
        M, D = x.shape
        if self.dec_frag:
            return self._chk(self.lib.tasu_rmsnorm_fwd_frag(_p(x), _p(w), _p(y), M, D, eps, self._stream()), "tasu_rmsnorm_fwd_frag")
        self._chk(self.lib.tasu_rmsnorm_fwd(_p(x), _p(w), _p(y), None, M, D, eps, self._stream()), "tasu_rmsnorm_fwd")

    def gemm_skinny(self, a, b, c, M, N, K, ws, bias=None, resid=None, mode=GEMM_BF16):
        """M <= 64 weight-streaming GEMM (decode step); ws: fp32 workspace tensor."""
        if mode == GEMM_BF16 and resid is None and self._stream_split(K) == 1 and c.stride(0) % 4 == 0:
            wf, wflag = self._wf(b)
            return self._chk(self.lib.tasu_gemm_stream_bf16(_p(a), a.stride(0), _p(wf), b.stride(0), _p(c), c.stride(0), _p(bias),
                                                            None, M, N, K, mode, int(self.dec_frag), wflag, self._stream()),
                             "tasu_gemm_stream_bf16")
        self._chk(self.lib.tasu_gemm_skinny_bf16(_p(a), a.stride(0), _p(b), b.stride(0), _p(c), c.stride(0), _p(bias),
                                                 _p(resid), M, N, K, mode, _p(ws), 0 if ws is None else ws.numel(),
                                                 self._stream()), "tasu_gemm_skinny_bf16")

    def transpose(self, src, dst, R, C, Rpad, Cpad):
        self._chk(self.lib.tasu_transpose_bf16(_p(src), src.stride(0), _p(dst), dst.stride(0), R, C, Rpad, Cpad,
                                               self._stream()), "tasu_transpose_bf16")

    def cast_bf16(self, src, dst):
        self._chk(self.lib.tasu_cast_f32_bf16(_p(src), _p(dst), src.numel(), self._stream()), "tasu_cast_f32_bf16")

    # ------------------------------------------------------------------ norms
    def rmsnorm_fwd(self, x, w, y, rstd, eps):
        M, D = x.shape
        if y.stride(0) != D:                            # y is the head of a wider operand buffer ([x | rank activations])
            return self._chk(self.lib.tasu_rmsnorm_fwd_ld(_p(x), _p(w), _p(y), y.stride(0), _p(rstd), M, D, eps, self._stream()), "tasu_rmsnorm_fwd_ld")
        self._chk(self.lib.tasu_rmsnorm_fwd(_p(x), _p(w), _p(y), _p(rstd), M, D, eps, self._stream()), "tasu_rmsnorm_fwd")

    def rmsnorm_bwd(self, dy, x, w, rstd, dx, dx_bf16, accumulate):
        M, D = x.shape
        self._chk(self.lib.tasu_rmsnorm_bwd(_p(dy), _p(x), _p(w), _p(rstd), _p(dx), _p(dx_bf16), int(accumulate), M, D,
                                            self._stream()), "tasu_rmsnorm_bwd")

    def rmsnorm_fwd_rows(self, x, src_rows, w, y, rstd, eps):
        """y[i] = rmsnorm(x[src_rows[i]]) (zero row where src_rows[i] < 0); y / rstd are compact."""
        n, D = y.shape
        self._chk(self.lib.tasu_rmsnorm_fwd_rows(_p(x), _p(src_rows), _p(w), _p(y), _p(rstd), n, D, eps, self._stream()),
                  "tasu_rmsnorm_fwd_rows")

    def rmsnorm_bwd_rows(self, dy, x, w, rstd, slot, dx, dx_bf16):
        """dx[m] = rmsnorm dgrad of compact row slot[m] (dy, rstd compact) or 0 where slot[m] < 0."""
        M, D = x.shape
        self._chk(self.lib.tasu_rmsnorm_bwd_rows(_p(dy), _p(x), _p(w), _p(rstd), _p(slot), _p(dx), _p(dx_bf16), M, D,
                                                 self._stream()), "tasu_rmsnorm_bwd_rows")

    def rmsnorm_bwd_rows_resid(self, dy, x, w, rstd, slot, resid, dx, dx_bf16):
        """dx[m] = resid[slot[m]] + rmsnorm dgrad of compact row slot[m] (dy, rstd, resid compact) or 0 where slot[m] < 0."""
        M, D = x.shape
        self._chk(self.lib.tasu_rmsnorm_bwd_rows_resid(_p(dy), _p(x), _p(w), _p(rstd), _p(slot), _p(resid), _p(dx), _p(dx_bf16), M, D,
                                                       self._stream()), "tasu_rmsnorm_bwd_rows_resid")

    def scale_softmax_rows(self, s, p, R, V, denom, stats=None):
        """p[r, :V] = bf16(softmax(bf16(s[r, :V] / denom))), pad columns zero (cross-attention projector); stats [R, 2] fp32
        receives the row's (max, 1 / sum) for softmax_bwd_rows."""
        self._chk(self.lib.tasu_scale_softmax_rows_bf16(_p(s), _p(p), _p(stats), R, V, s.stride(0), denom, self._stream()),
                  "tasu_scale_softmax_rows_bf16")

    def softmax_bwd_rows(self, s, stats, dp, ds, R, V, denom):
        """ds = bf16(bf16(P32 * (dp - sum(P32 * dp))) / denom) with P32 = the fp32 softmax recomputed from the saved scores s."""
        self._chk(self.lib.tasu_softmax_bwd_rows_bf16(_p(s), _p(stats), _p(dp), _p(ds), R, V, s.stride(0), denom, self._stream()),
                  "tasu_softmax_bwd_rows_bf16")

    def layernorm_fwd(self, x, gamma, beta, y, mean, rstd, R, D, eps):
        self._chk(self.lib.tasu_layernorm_fwd(_p(x), x.stride(0), _p(gamma), _p(beta), _p(y), y.stride(0),
                                              int(y.dtype == torch.float32), _p(mean), _p(rstd), R, D, eps,
                                              self._stream()), "tasu_layernorm_fwd")

    def layernorm_bwd_params(self, dy, x, mean, rstd, dgamma, dbeta, ws, R, D):
        self._chk(self.lib.tasu_layernorm_bwd_params(_p(dy), dy.stride(0), _p(x), x.stride(0), _p(mean), _p(rstd),
                                                     _p(dgamma), _p(dbeta), _p(ws), R, D, self._stream()),
                  "tasu_layernorm_bwd_params")

    def colsum(self, x, out, R, Cn):
        self._chk(self.lib.tasu_colsum_bf16(_p(x), x.stride(0), _p(out), R, Cn, self._stream()), "tasu_colsum_bf16")

    # ------------------------------------------------------------------ rope + attention
    def rope_table(self, pos, cos, sin, head_dim, theta):
        self._chk(self.lib.tasu_rope_table(_p(pos), _p(cos), _p(sin), pos.numel(), head_dim, theta, self._stream()),
                  "tasu_rope_table")

    def rope_fwd(self, qkv, cos, sin, qt, kt, vt, B, S, H, G):
        self._chk(self.lib.tasu_rope_fwd(_p(qkv), _p(cos), _p(sin), _p(qt), _p(kt), _p(vt), B, S, H, G, self._stream()),
                  "tasu_rope_fwd")

    def rope_bwd(self, dqkv, dk_part, dv_part, cos, sin, B, S, H, G):
        self._chk(self.lib.tasu_rope_bwd(_p(dqkv), _p(dk_part), _p(dv_part), _p(cos), _p(sin), B, S, H, G,
                                         self._stream()), "tasu_rope_bwd")

    def attn_fwd(self, qkv, vt, key_mask, out, lse, B, S, H, G, scale, causal):
        if self.attn_kernel["fwd"] != "policy":
            return self.attn_fwd_on(self.attn_kernel["fwd"], qkv, key_mask, out, lse, B, S, H, G, scale, causal)
        self._chk(self.lib.tasu_attn_fwd(_p(qkv), _p(vt), _p(key_mask), _p(out), _p(lse), B, S, H, G, scale, int(causal),
                                         self._stream()), "tasu_attn_fwd")

    ATTN_KERNELS = {"policy": 0, "tiled": 1, "gqa": 2, "sp": 3}

    def attn_fwd_on(self, kernel, qkv, key_mask, out, lse, B, S, H, G, scale, causal):
        """attn_fwd on a NAMED kernel (tasu_attn_fwd_kernel): "tiled" (attention.hip) or "sp" (single pass, Spad <= 256)."""
        self._chk(self.lib.tasu_attn_fwd_kernel(_p(qkv), _p(key_mask), _p(out), _p(lse), B, S, H, G, scale, int(causal),
                                                self.ATTN_KERNELS[kernel], self._stream()), "tasu_attn_fwd_kernel")

    def attn_bwd_fused(self, qkv, key_mask, dout, out, lse, delta, cos, sin, dqkv, dk_part, dv_part, B, S, H, G, scale, causal,
                       kernel=None):
        """The whole attention backward (delta, dQ / dK / dV, rotary backward) behind one entry point (tasu_attn_bwd_fused): the
        single-pass kernels where Spad <= 256, else attn_bwd_prep + attn_bwd_rope.  dk_part / dv_part: fp32 [M, H * 128]."""
        kernel = kernel or self.attn_kernel["bwd"]
        self._chk(self.lib.tasu_attn_bwd_fused(_p(qkv), _p(key_mask), _p(dout), _p(out), _p(lse), _p(delta), _p(cos), _p(sin), _p(dqkv),
                                               _p(dk_part), _p(dv_part), B, S, H, G, scale, int(causal), self.ATTN_KERNELS[kernel],
                                               self._stream()), "tasu_attn_bwd_fused")

    def attn_bwd_prep(self, dout, out, delta, dout_t, B, S, H):
        self._chk(self.lib.tasu_attn_bwd_prep(_p(dout), _p(out), _p(delta), _p(dout_t), B, S, H, self._stream()),
                  "tasu_attn_bwd_prep")

    def attn_bwd_dq(self, qkv, kt, key_mask, dout, lse, delta, dqkv, B, S, H, G, scale, causal):
        self._chk(self.lib.tasu_attn_bwd_dq(_p(qkv), _p(kt), _p(key_mask), _p(dout), _p(lse), _p(delta), _p(dqkv), B, S,
                                            H, G, scale, int(causal), self._stream()), "tasu_attn_bwd_dq")

    def attn_bwd_dkv(self, qkv, qt, key_mask, dout, dout_t, lse, delta, dk_part, dv_part, B, S, H, G, scale, causal):
        self._chk(self.lib.tasu_attn_bwd_dkv(_p(qkv), _p(qt), _p(key_mask), _p(dout), _p(dout_t), _p(lse), _p(delta),
                                             _p(dk_part), _p(dv_part), B, S, H, G, scale, int(causal), self._stream()),
                  "tasu_attn_bwd_dkv")

    def attn_bwd(self, qkv, qt, kt, key_mask, dout, dout_t, lse, delta, dqkv, dk_part, dv_part, B, S, H, G, scale, causal):
        """attn_bwd_dq + attn_bwd_dkv in one launch."""
        self._chk(self.lib.tasu_attn_bwd(_p(qkv), _p(qt), _p(kt), _p(key_mask), _p(dout), _p(dout_t), _p(lse), _p(delta),
                                         _p(dqkv), _p(dk_part), _p(dv_part), B, S, H, G, scale, int(causal), self._stream()),
                  "tasu_attn_bwd")

    def attn_bwd_rope(self, qkv, key_mask, dout, lse, delta, cos, sin, dqkv, dk_part, dv_part, B, S, H, G, scale, causal, kernel=0):
        """attn_bwd + rope_bwd behind one entry point: dqkv = the finished gradient of the unrotated q | k | v projection.  kernel:
        0 policy, 1 per-head kernels (two launches), 2 the GQA kernel (one launch; dk_part / dv_part untouched)."""
        self._chk(self.lib.tasu_attn_bwd_rope(_p(qkv), _p(key_mask), _p(dout), _p(lse), _p(delta), _p(cos), _p(sin), _p(dqkv),
                                              _p(dk_part), _p(dv_part), B, S, H, G, scale, int(causal), int(kernel), self._stream()),
                  "tasu_attn_bwd_rope")

    # ------------------------------------------------------------------ activations
    def swiglu_fwd(self, gu, act, M, I):
        self._chk(self.lib.tasu_swiglu_fwd(_p(gu), _p(act), M, I, self._stream()), "tasu_swiglu_fwd")

    def swiglu_bwd(self, dact, gu, dgu, M, I):
        self._chk(self.lib.tasu_swiglu_bwd(_p(dact), _p(gu), _p(dgu), M, I, self._stream()), "tasu_swiglu_bwd")

    def silu_fwd(self, x, y):
        self._chk(self.lib.tasu_silu_fwd(_p(x), _p(y), x.numel(), self._stream()), "tasu_silu_fwd")

    def silu_bwd(self, dy, x, dx):
        self._chk(self.lib.tasu_silu_bwd(_p(dy), _p(x), _p(dx), x.numel(), self._stream()), "tasu_silu_bwd")

    def relu_bwd(self, dy, x, dx):
        self._chk(self.lib.tasu_relu_bwd(_p(dy), _p(x), _p(dx), x.numel(), self._stream()), "tasu_relu_bwd")

    def relu_fwd(self, x, y):
        self._chk(self.lib.tasu_relu_fwd(_p(x), _p(y), x.numel(), self._stream()), "tasu_relu_fwd")

    # ------------------------------------------------------------------ LoRA
    def gemm_rank(self, a, b, c, M, N, K, f32=False, transposed=False):
        """c[M, N] = a[M, K] @ b[N, K]^T for N <= 64 (csrc/gemm_rank.hip); ``transposed``: c holds C^T [N, M]."""
        self._chk(self.lib.tasu_gemm_nt_rank(_p(a), a.stride(0), _p(b), b.stride(0), _p(c), c.stride(0), M, N, K, int(f32), int(transposed),
                                             self._stream()), "tasu_gemm_nt_rank")

    def gemm_rank_tn(self, at, b, c, M, N, K, transposed=False):
        """c[M, N] (fp32) = at[K, M]^T @ b[N, K]^T for N <= 64: A given K-major (tasu_gemm_tn_rank); ``transposed``: c holds C^T."""
        self._chk(self.lib.tasu_gemm_tn_rank(_p(at), at.stride(0), _p(b), b.stride(0), _p(c), c.stride(0), M, N, K, int(transposed),
                                             self._stream()), "tasu_gemm_tn_rank")

    def lora_apply(self, y, u, w, M, N, R, s=1.0, p=0.0, rng=None, sid=0, x_in=None, x_out=None):
        """y[M, N] = bf16(y + mask * bf16(s * bf16(u[M, R] @ w[N, R]^T))) [, x_out = x_in + y] in one pass over y (tasu_lora_apply)."""
        self._chk(self.lib.tasu_lora_apply(_p(y), y.stride(0), _p(u), u.stride(0), _p(w), w.stride(0), M, N, R, float(s), float(p), _p(rng),
                                           int(sid), _p(x_in), _p(x_out), 0 if x_in is None else x_in.stride(0), self._stream()),
                  "tasu_lora_apply")

    # ---- one launch per adapted group (csrc/lora.hip, csrc/gemm_rank.hip: the members' kernels batched)
    @staticmethod
    def _ptrs(ts):
        return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])

    @staticmethod
    def _ints(vs):
        return (ctypes.c_int * len(vs))(*[int(v) for v in vs])

    def gemm_rank_group(self, As, Bs, Cs, M, N, Ks):
        """Cs[t][M, N] = bf16(As[t][M, Ks[t]] @ Bs[t][N, Ks[t]]^T) for the members t of a group in one launch (tasu_gemm_nt_rank_group);
        the Cs share their row stride."""
        ldc = Cs[0].stride(0)
        assert all(c.stride(0) == ldc for c in Cs)
        self._chk(self.lib.tasu_gemm_nt_rank_group(len(As), self._ptrs(As), self._ints([a.stride(0) for a in As]), self._ptrs(Bs),
                                                   self._ints([b.stride(0) for b in Bs]), self._ptrs(Cs), ldc, M, N, self._ints(Ks),
                                                   self._stream()), "tasu_gemm_nt_rank_group")

    def lora_apply_group(self, y, us, ws, M, N, R, sids, s=1.0, p=0.0, rng=None):
        """lora_apply for the members of a group, in order, in one pass over y (tasu_lora_apply_group: the same bits)."""
        ldu, ldw = us[0].stride(0), ws[0].stride(0)
        assert all(u.stride(0) == ldu for u in us) and all(w.stride(0) == ldw for w in ws)
        self._chk(self.lib.tasu_lora_apply_group(_p(y), y.stride(0), len(us), self._ptrs(us), ldu, self._ptrs(ws), ldw, self._ints(sids), M, N, R,
                                                 float(s), float(p), _p(rng), self._stream()), "tasu_lora_apply_group")

    def lora_dropout_norm_group(self, x, w, rstd, dsts, M, D, p, rng, sids):
        self._chk(self.lib.tasu_lora_dropout_norm_group(_p(x), _p(w), _p(rstd), len(dsts), self._ptrs(dsts), self._ints(sids), M, D, float(p),
                                                        _p(rng), self._stream()), "tasu_lora_dropout_norm_group")

    def scale_bf16(self, src, dst, s):
        self._chk(self.lib.tasu_scale_bf16(_p(src), _p(dst), float(s), src.numel(), self._stream()), "tasu_scale_bf16")

    def lora_dropout(self, src, dst, p, rng, sid):
        """dst = dropout(src) for [M, C] matrices (row strides honoured; a 1-D tensor is one row); mask index m * C + c."""
        M, C = (1, src.numel()) if src.dim() == 1 else src.shape
        lds, ldd = (C, C) if src.dim() == 1 else (src.stride(0), dst.stride(0))
        self._chk(self.lib.tasu_lora_dropout(_p(src), lds, _p(dst), ldd, M, C, float(p), _p(rng), int(sid), self._stream()), "tasu_lora_dropout")

    def lora_dropout_norm(self, x, w, rstd, dst, M, D, p, rng, sid):
        self._chk(self.lib.tasu_lora_dropout_norm(_p(x), _p(w), _p(rstd), _p(dst), M, D, float(p), _p(rng), int(sid), self._stream()),
                  "tasu_lora_dropout_norm")

    def copy_rows(self, src, dst, M, C):
        """dst[:M, :C] = src[:M, :C] (bf16, both with their own row strides)."""
        self._chk(self.lib.tasu_copy_rows_bf16(_p(src), src.stride(0), _p(dst), dst.stride(0), M, C, self._stream()), "tasu_copy_rows_bf16")

    def lora_refresh(self, pb, table, n_entries, total_tiles):
        self._chk(self.lib.tasu_lora_refresh(_p(pb), _p(table), n_entries, total_tiles, self._stream()), "tasu_lora_refresh")

    def rng_advance(self, rng):
        self._chk(self.lib.tasu_rng_advance(_p(rng), self._stream()), "tasu_rng_advance")

    # ------------------------------------------------------------------ loss
    def ce_fwd_bwd(self, logits, shift_labels, M, V, row_loss, row_hit, row_argmax, dlogits, inv_count):
        self._chk(self.lib.tasu_ce_fwd_bwd(_p(logits), logits.stride(0), _p(shift_labels), M, V, _p(row_loss),
                                           _p(row_hit), _p(row_argmax), _p(dlogits), _p(inv_count), self._stream()),
                  "tasu_ce_fwd_bwd")

    def ce_reduce(self, row_loss, row_hit, shift_labels, M, out):
        self._chk(self.lib.tasu_ce_reduce(_p(row_loss), _p(row_hit), _p(shift_labels), M, _p(out), self._stream()),
                  "tasu_ce_reduce")

    # ------------------------------------------------------------------ front end / merge
    def posterior_build(self, ids, alpha, out, R, V):
        self._chk(self.lib.tasu_posterior_build(_p(ids), _p(alpha), _p(out), out.stride(0), R, V, self._stream()),
                  "tasu_posterior_build")

    def embed_merge(self, table, proj, kind, idx, x, M, D):
        self._chk(self.lib.tasu_embed_merge_fwd(_p(table), _p(proj), _p(kind), _p(idx), _p(x), M, D, self._stream()),
                  "tasu_embed_merge_fwd")

    def merge_bwd(self, dx, audio_rows, dproj, n, D):
        self._chk(self.lib.tasu_merge_bwd(_p(dx), _p(audio_rows), _p(dproj), n, D, self._stream()), "tasu_merge_bwd")

    # ------------------------------------------------------------------ optimizer
    def adamw(self, p, g, m, v, p_bf16, lr, beta1, beta2, eps, wd, step, grad_scale):
        self._chk(self.lib.tasu_adamw(_p(p), _p(g), _p(m), _p(v), _p(p_bf16), p.numel(), float(lr), beta1, beta2, eps, wd,
                                      step, grad_scale, self._stream()), "tasu_adamw")

    # ------------------------------------------------------------------ encoder / PSD
    def sinusoid_pe(self, x, y, B, T, D, scale):
        self._chk(self.lib.tasu_sinusoid_pe(_p(x), _p(y), B, T, D, scale, self._stream()), "tasu_sinusoid_pe")

    def fsmn_fwd(self, v, ldv, w, lens, out, B, T, D, ksize, accumulate):
        self._chk(self.lib.tasu_fsmn_fwd(_p(v), ldv, _p(w), _p(lens), _p(out), B, T, D, ksize, int(accumulate),
                                         self._stream()), "tasu_fsmn_fwd")

    def fsmn_ln_fwd(self, v, ldv, w, lens, x, gamma, beta, xn, B, T, D, ksize, eps):
        """x += fsmn(v); xn = LayerNorm(x) (bf16) -- one launch for D = 512 / kernel 11 (tasu_fsmn_ln_fwd)."""
        self._chk(self.lib.tasu_fsmn_ln_fwd(_p(v), ldv, _p(w), _p(lens), _p(x), _p(gamma), _p(beta), _p(xn), xn.stride(0), B, T, D, ksize,
                                            eps, self._stream()), "tasu_fsmn_ln_fwd")

    def softmax_rows(self, x, y, R, V):
        self._chk(self.lib.tasu_softmax_rows(_p(x), int(x.dtype == torch.bfloat16), x.stride(0), _p(y), y.stride(0), R, V,
                                             self._stream()), "tasu_softmax_rows")

    def psd_frame_stats(self, post, lens, fid, fblank, B, T, bstride, V, blank_id):
        self._chk(self.lib.tasu_psd_frame_stats(_p(post), post.stride(0), _p(lens), _p(fid), _p(fblank), B, T, bstride, V,
                                                blank_id, self._stream()), "tasu_psd_frame_stats")

    def psd_plan(self, fid, fblank, lens, seg_start, seg_len, new_lens, B, T, blank_id, thr):
        self._chk(self.lib.tasu_psd_plan(_p(fid), _p(fblank), _p(lens), _p(seg_start), _p(seg_len), _p(new_lens), B, T,
                                         blank_id, thr, self._stream()), "tasu_psd_plan")

    def psd_gather(self, post, seg_start, seg_len, new_lens, out, B, T, bstride, Tout, V):
        self._chk(self.lib.tasu_psd_gather(_p(post), post.stride(0), _p(seg_start), _p(seg_len), _p(new_lens), _p(out),
                                           out.stride(0), B, T, bstride, Tout, V, self._stream()), "tasu_psd_gather")

    def psd_logit_stats(self, logits, lens, fid, fblank, fstat, B, T, bstride, V, blank_id):
        self._chk(self.lib.tasu_psd_logit_stats(_p(logits), logits.stride(0), _p(lens), _p(fid), _p(fblank), _p(fstat), B, T, bstride, V,
                                                blank_id, self._stream()), "tasu_psd_logit_stats")

    def psd_gather_softmax(self, logits, fstat, seg_start, seg_len, new_lens, out, B, T, bstride, Tout, V):
        self._chk(self.lib.tasu_psd_gather_softmax(_p(logits), logits.stride(0), _p(fstat), _p(seg_start), _p(seg_len), _p(new_lens),
                                                   _p(out), out.stride(0), B, T, bstride, Tout, V, self._stream()), "tasu_psd_gather_softmax")

    # ------------------------------------------------------------------ decode loop
    def kv_fill(self, qkv, kc, vc, B, S, H, G, nb, ctx):
        self._chk(self.lib.tasu_kv_fill(_p(qkv), _p(kc), _p(vc), B, S, H, G, nb, ctx, self._stream()), "tasu_kv_fill")

    def kv_append(self, qkv, kc, vc, pos, M, H, G, ctx):
        self._chk(self.lib.tasu_kv_append(_p(qkv), _p(kc), _p(vc), _p(pos), M, H, G, ctx, self._stream()), "tasu_kv_append")

    def gemm_skinny_norm(self, a, b, c, resid, M, N, K, norm_w, y, eps, ws, prenorm_slot=None):
        """c (fp32) = resid + bf16(a @ b^T); y = rmsnorm(c, norm_w) -- decode-step projection with the next norm fused.
        prenorm_slot (int; the K-range-slab route only, prenorm_in_ok): y = bf16(norm_w * c) WITHOUT rstd, and the per-tile sums of
        squares are returned for gemm_skinny_qkv_rope(sumsq=...) (tasu_stream_finish_prenorm); else returns None."""
        ks = self._stream_split(K) if N % 16 == 0 else 0
        # layout of the INPUT: a one-range K reads a buffer written in the layer's activation order (dec_frag); a longer K is
        # the down projection, whose input layout is dec_frag_act
        a_frag = int(self.dec_frag if ks == 1 else self.dec_frag_act)
        y_frag = int(self.dec_frag)
        need = ks * (N // 16) * 1024 if ks > 1 else 0
        if ks > 1 and (not (self.dec_down_slabs or a_frag) or not self._slab_finish_serves(N) or ws is None or ws.numel() < need):
            if a_frag:
                # begin_decode promised the producer a fragment-order reader: the row-major split-K kernel below would misread it
                raise TasuOpError(f"gemm_skinny_norm: the activation is in fragment order but the K-range-slab route cannot run "
                                  f"(N={N}, K={K}, {ks} ranges, workspace {0 if ws is None else ws.numel()} of {need} floats)")
            ks = 0                                          # split-K kernels of gemm_skinny.hip (see begin_decode; the slab
                                                            # finish serves N = 256 * {1, 2, 6, 7, 14})
        wf, wflag = (self._wf(b) if a_frag else (b, 0))
        if self.dec_fused_norm and ks >= 1 and N in (256, 1536) and c.stride(0) == N and y.stride(0) == N and \
                (ks == 1 or (ws is not None and ws.numel() >= ks * (N // 16) * 1024)):
            # one launch: projection + residual (+ slab sum) + the norm of the finished rows by the last workgroups to arrive
            return self._chk(self.lib.tasu_gemm_stream_norm(_p(a), a.stride(0), _p(wf), b.stride(0), _p(c), _p(resid), M, N, K, ks,
                                                            _p(ws) if ks > 1 else None, ws.numel() if (ks > 1 and ws is not None) else 0,
                                                            _p(norm_w), _p(y), eps, a_frag, wflag, y_frag, _p(self.norm_sync), self._stream()),
                             "tasu_gemm_stream_norm")
        if ks == 1:
            # one launch for the projection + residual add, one for the norm (the sum of squares needs the whole row)
            self._chk(self.lib.tasu_gemm_stream_bf16(_p(a), a.stride(0), _p(wf), b.stride(0), _p(c), N, None, _p(resid), M, N, K,
                                                     GEMM_RESID, a_frag, wflag, self._stream()), "tasu_gemm_stream_bf16")
            return self.dec_rmsnorm(c[:M], norm_w, y, eps)
        if ks > 1 and ws is not None and ws.numel() >= ks * (N // 16) * 1024:
            self._chk(self.lib.tasu_gemm_stream_slabs(_p(a), a.stride(0), _p(wf), b.stride(0), _p(ws), ws.numel(), M, N, K, ks,
                                                      a_frag, wflag, self._stream()), "tasu_gemm_stream_slabs")
            if prenorm_slot is not None:
                ssq = self.dec_sumsq_in[prenorm_slot]
                self._chk(self.lib.tasu_stream_finish_prenorm(_p(ws), ks, _p(c), _p(resid), M, N, _p(norm_w), _p(y), y_frag, _p(ssq),
                                                              self._stream()), "tasu_stream_finish_prenorm")
                return ssq
            return self._chk(self.lib.tasu_stream_finish_norm(_p(ws), ks, _p(c), _p(resid), M, N, _p(norm_w), _p(y), eps, y_frag,
                                                              self._stream()), "tasu_stream_finish_norm")
        self._chk(self.lib.tasu_gemm_skinny_norm(_p(a), a.stride(0), _p(b), b.stride(0), _p(c), _p(resid), M, N, K, _p(norm_w),
                                                 _p(y), eps, y_frag, _p(ws), 0 if ws is None else ws.numel(), self._stream()),
                  "tasu_gemm_skinny_norm")

    def gemm_skinny_qkv_rope(self, a, wqkv, bias, qkv, M, H, G, K, cos, sin, kc, vc, pos, ctx, ws, sumsq=None, eps=0.0):
        """qkv = rope(a @ wqkv^T + bias); k, v appended to the cache at pos -- one call per decode-step layer.  sumsq: a is the slab
        finish's bf16(norm_w * x); the rows' rstd comes from the K / 16 partial sums of squares."""
        if sumsq is not None:
            wf, wflag = self._wf(wqkv)
            return self._chk(self.lib.tasu_gemm_stream_qkv_rope_rstd(_p(a), a.stride(0), _p(wf), wqkv.stride(0), _p(bias), _p(qkv), M, H, G, K,
                                                                     _p(cos), _p(sin), _p(kc), _p(vc), _p(pos), ctx, _p(sumsq), K // 16, eps,
                                                                     int(self.dec_frag), wflag, self._stream()),
                             "tasu_gemm_stream_qkv_rope_rstd")
        if self._stream_split(K) == 1:
            wf, wflag = self._wf(wqkv)
            return self._chk(self.lib.tasu_gemm_stream_qkv_rope(_p(a), a.stride(0), _p(wf), wqkv.stride(0), _p(bias), _p(qkv), M, H,
                                                                G, K, _p(cos), _p(sin), _p(kc), _p(vc), _p(pos), ctx,
                                                                int(self.dec_frag), wflag, self._stream()),
                             "tasu_gemm_stream_qkv_rope")
        self._chk(self.lib.tasu_gemm_skinny_qkv_rope(_p(a), a.stride(0), _p(wqkv), wqkv.stride(0), _p(bias), _p(qkv), M, H, G, K,
                                                     _p(cos), _p(sin), _p(kc), _p(vc), _p(pos), ctx, _p(ws),
                                                     0 if ws is None else ws.numel(), self._stream()),
                  "tasu_gemm_skinny_qkv_rope")

    def prenorm_ok(self, D, HHD, I):
        """decode.py asks: may the post-attention norm travel inside its neighbours (gemm_skinny_prenorm + gemm_skinny_swiglu(sumsq=))?"""
        # (K <= 2048 only: at K = 3584 the streaming kernels have no register to spare -- measured slower with it)
        return bool(self.dec_prenorm and self.dec_frag and D % 16 == 0 and I % 8 == 0 and max(D, HHD) <= 2048 and
                    self._stream_split(HHD) == 1 and self._stream_split(D) == 1)

    def prenorm_in_ok(self, D, I):
        """... and the layer's INPUT norm inside the slab finish and the q|k|v projection (gemm_skinny_norm(prenorm_slot=) +
        gemm_skinny_qkv_rope(sumsq=)): the down projection must run as K-range slabs."""
        ks = self._stream_split(I)
        return bool(self.dec_prenorm and self.dec_prenorm_in and self.dec_frag and self.dec_frag_act and self.dec_down_slabs and ks > 1 and
                    D % 256 == 0 and D <= 2048 and self._stream_split(D) == 1)

    def gemm_skinny_prenorm(self, a, b, c, resid, M, N, K, norm_w, yw):
        """c (fp32) = resid + bf16(a @ b^T); yw = bf16(norm_w * c) in the consumer's operand order; returns the per-tile sums of
        squares [N / 16][64] for gemm_skinny_swiglu(sumsq=...)."""
        if self.dec_sumsq is None:
            self.dec_sumsq = torch.zeros(4096 // 16 * 64, dtype=torch.float32, device=c.device)
        wf, wflag = self._wf(b)
        self._chk(self.lib.tasu_gemm_stream_resid_prenorm(_p(a), a.stride(0), _p(wf), b.stride(0), _p(c), _p(resid), M, N, K, _p(norm_w), _p(yw),
                                                          int(self.dec_frag), _p(self.dec_sumsq), int(self.dec_frag), wflag, self._stream()),
                  "tasu_gemm_stream_resid_prenorm")
        return self.dec_sumsq

    def gemm_skinny_swiglu(self, a, wgu, act, M, I, K, ws, sumsq=None, eps=0.0):
        """act[M, I] = swiglu(a[M,K] @ wgu[2I,K]^T) in one launch (decode step).  sumsq: a is gemm_skinny_prenorm's yw; the rows'
        rstd comes from the K / 16 partial sums of squares."""
        if sumsq is not None:
            wf, wflag = self._wf(wgu)
            return self._chk(self.lib.tasu_gemm_stream_swiglu_rstd(_p(a), a.stride(0), _p(wf), wgu.stride(0), _p(act), act.stride(0), M, I, K,
                                                                   _p(sumsq), K // 16, eps, int(self.dec_frag), wflag,
                                                                   int(self.dec_frag_act), self._stream()), "tasu_gemm_stream_swiglu_rstd")
        if self._stream_split(K) == 1 and I % 8 == 0 and act.stride(0) % 4 == 0:
            wf, wflag = self._wf(wgu)
            return self._chk(self.lib.tasu_gemm_stream_swiglu(_p(a), a.stride(0), _p(wf), wgu.stride(0), _p(act), act.stride(0), M, I,
                                                              K, int(self.dec_frag), wflag, int(self.dec_frag_act), self._stream()),
                             "tasu_gemm_stream_swiglu")
        self._chk(self.lib.tasu_gemm_skinny_swiglu(_p(a), a.stride(0), _p(wgu), wgu.stride(0), _p(act), act.stride(0), M, I, K,
                                                   _p(ws), 0 if ws is None else ws.numel(), self._stream()),
                  "tasu_gemm_skinny_swiglu")

    def decode_step_prologue(self, table, ids, x, norm_w, xn, eps, pos, cos, sin, head_dim, theta, index, index_tmp, src_row, lens,
                             n_beams, M, D, ctx):
        """Start of a generated position: embedding rows, the first layer's input norm, the RoPE factors and the beam reorder of
        the cache row index -- one launch in fragment-order decodes (tasu_decode_step_prologue), else the five separate ones."""
        if self.dec_frag and self.dec_prologue and D % 256 == 0 and D // 256 in (1, 2, 6, 7, 14) and head_dim == 128 and n_beams <= 5:
            return self._chk(self.lib.tasu_decode_step_prologue(_p(table), _p(ids), _p(x), _p(norm_w), _p(xn), eps, _p(pos), _p(cos),
                                                                _p(sin), theta, _p(index), _p(src_row), _p(lens), n_beams, M, D, ctx,
                                                                self._stream()), "tasu_decode_step_prologue")
        self.kv_index_reorder(index, index_tmp, src_row, lens, M, ctx)
        self.kv_index_reorder(index_tmp, index, None, lens, M, ctx)
        self.embed_rows(table, ids, x, M, D)
        self.rope_table(pos, cos, sin, head_dim, theta)
        for m0 in range(0, M, 64):
            self.dec_rmsnorm(x[m0:m0 + 64], norm_w, xn[m0:m0 + 64], eps)

    def rope_append(self, qkv, cos, sin, kc, vc, pos, M, H, G, ctx):
        self._chk(self.lib.tasu_rope_append(_p(qkv), _p(cos), _p(sin), _p(kc), _p(vc), _p(pos), M, H, G, ctx, self._stream()),
                  "tasu_rope_append")

    def kv_index_init(self, index, B, nb, S, ctx):
        self._chk(self.lib.tasu_kv_index_init(_p(index), B, nb, S, ctx, self._stream()), "tasu_kv_index_init")

    def kv_index_reorder(self, src, dst, src_row, lens, M, ctx):
        self._chk(self.lib.tasu_kv_index_reorder(_p(src), _p(dst), _p(src_row), _p(lens), M, ctx, self._stream()),
                  "tasu_kv_index_reorder")

    def attn_decode(self, qkv, kc, vc, index, kstart, lens, out, M, H, G, ctx, scale):
        self._chk(self.lib.tasu_attn_decode(_p(qkv), _p(kc), _p(vc), _p(index), _p(kstart), _p(lens), _p(out), M, H, G, ctx, scale,
                                            int(self.dec_frag), self._stream()), "tasu_attn_decode")

    def logprob_topk(self, logits, M, V, k, banned, n_banned, out_val, out_idx):
        need = M * 16 * (2 + 2 * k)
        if self.topk_ws.numel() < need:
            raise TasuOpError(f"logprob_topk: M={M}, k={k} exceeds the preallocated workspace")
        self._chk(self.lib.tasu_logprob_topk(_p(logits), logits.stride(0), M, V, k, _p(banned), n_banned, _p(out_val),
                                             _p(out_idx), _p(self.topk_ws), self.topk_ws.numel(), self._stream()),
                  "tasu_logprob_topk")

    def beam_update(self, vals, idx, bs, first):
        """One position of the device-side beam search (tasu_beam_update); ``bs``: ps_slm_amd.decode.DeviceBeam."""
        self._chk(self.lib.tasu_beam_update(_p(vals), _p(idx), _p(bs.run_scores), _p(bs.fin_scores), _p(bs.fin_len), _p(bs.fin_par),
                                            _p(bs.fin_tok), _p(bs.is_fin), _p(bs.unsat), _p(bs.bp_tok), _p(bs.bp_par),
                                            _p(bs.len_pow), _p(bs.ctl), _p(bs.done_host), _p(bs.valid), _p(bs.next_ids),
                                            _p(bs.next_src), _p(bs.next_pos), _p(bs.next_slot), _p(bs.next_lens), _p(bs.banned),
                                            bs.B, bs.nb, bs.max_new, bs.eos, bs.min_length, bs.S, int(first), self._stream()),
                  "tasu_beam_update")

    # ------------------------------------------------------------------ fp32 arithmetic mode of the decode path (csrc/fp32.hip)
    def f32_gemm(self, a, w, c, M, N, K, bias=None, resid=None, act=0, ws=None):
        """c[M, N] = [resid +] act(a[M, K] @ w[N, K]^T + bias), everything fp32 (tasu_f32_gemm_nt); ``ws``: fp32 workspace for the
        K-range slabs of narrow outputs (16 * M * N floats cover every split)."""
        self._chk(self.lib.tasu_f32_gemm_nt(_p(a), a.stride(0), *_wl(w), _p(c), c.stride(0), _p(bias), _p(resid), M, N, K, act,
                                            _p(ws), ws.numel() if ws is not None else 0, self._stream()), "tasu_f32_gemm_nt")

    def f32_to_fragments(self, w):
        """The fragment-order copy of an fp32 [N, K] matrix (K % 16 == 0) for the decode step's streaming GEMM."""
        N, K = w.shape
        out = torch.empty(((N + 15) // 16) * 16 * K, device=w.device, dtype=torch.float32)
        self._chk(self.lib.tasu_f32_to_fragment_order(_p(w), w.stride(0), _p(out), N, K, self._stream()), "tasu_f32_to_fragment_order")
        return F32Fragments(out, N, K, w)

    def f32_weight(self, w, M, ws):
        """The operand to hand an f32_gemm* wrapper for a product with M rows: the fragment-order copy where the streaming kernel
        serves the problem (tasu_f32_gemm_streams), the row-major matrix otherwise."""
        if isinstance(w, F32Fragments):
            return w if self.lib.tasu_f32_gemm_streams(M, w.N, w.K, ws.numel() if ws is not None else 0) == 1 else w.rows
        return w

    def f32_gemm_stream(self, a, w, c, M, N, K, ks, bias=None, resid=None, act=0, ws=None):
        """f32_gemm forced onto the weight-streaming kernel of the decode step (M <= 64) with K slices of 16 ks per wave
        (tasu_f32_gemm_stream: tests and tools)."""
        self._chk(self.lib.tasu_f32_gemm_stream(_p(a), a.stride(0), *_wl(w), _p(c), c.stride(0), _p(bias), _p(resid), M, N, K, act,
                                                ks, _p(ws), ws.numel() if ws is not None else 0, self._stream()), "tasu_f32_gemm_stream")

    def f32_gemm_resid_rmsnorm(self, a, w, x, norm_w, y, M, N, K, eps, ws, resid=None, bias=None):
        """x = resid + a @ w^T (+ bias); y = rmsnorm(x, norm_w): a decoder layer's o / down projection with the norm behind it
        (tasu_f32_gemm_resid_rmsnorm: one launch for the slab sum and the norm when the problem splits)."""
        self._chk(self.lib.tasu_f32_gemm_resid_rmsnorm(_p(a), a.stride(0), *_wl(w), _p(x), x.stride(0), _p(bias), _p(resid),
                                                       _p(norm_w), _p(y), M, N, K, eps, _p(ws), ws.numel() if ws is not None else 0,
                                                       self._stream()), "tasu_f32_gemm_resid_rmsnorm")

    def f32_gemm_swiglu(self, a, wgu, gu, act, M, I, K, ws):
        self._chk(self.lib.tasu_f32_gemm_swiglu(_p(a), a.stride(0), *_wl(wgu), _p(gu), _p(act), M, I, K, _p(ws),
                                                ws.numel() if ws is not None else 0, self._stream()), "tasu_f32_gemm_swiglu")

    def f32_gemm_qkv_rope(self, a, wqkv, bias, qkv, cos, sin, M, H, G, K, ws, kc=None, vc=None, slot=None, ctx=0):
        self._chk(self.lib.tasu_f32_gemm_qkv_rope(_p(a), a.stride(0), *_wl(wqkv), _p(bias), _p(qkv), _p(cos), _p(sin), M, H, G,
                                                  K, _p(kc), _p(vc), _p(slot), ctx, _p(ws), ws.numel() if ws is not None else 0,
                                                  self._stream()), "tasu_f32_gemm_qkv_rope")

    def f32_rmsnorm(self, x, w, y, M, D, eps):
        self._chk(self.lib.tasu_f32_rmsnorm(_p(x), _p(w), _p(y), M, D, eps, self._stream()), "tasu_f32_rmsnorm")

    def f32_rope(self, qkv, cos, sin, M, H, G, kc=None, vc=None, slot=None, ctx=0, inverse=False):
        self._chk(self.lib.tasu_f32_rope(_p(qkv), _p(cos), _p(sin), M, H, G, _p(kc), _p(vc), _p(slot), ctx, int(inverse), self._stream()),
                  "tasu_f32_rope")

    def f32_kv_fill(self, qkv, kc, vc, B, S, H, G, nb, ctx):
        self._chk(self.lib.tasu_f32_kv_fill(_p(qkv), _p(kc), _p(vc), B, S, H, G, nb, ctx, self._stream()), "tasu_f32_kv_fill")

    def f32_attn_prefill(self, qkv, kstart, out, B, S, H, G, scale, klen=None):
        """klen None: causal over keys [kstart[b], s]; else bidirectional over keys [0, klen[b]) (the SANM encoder)."""
        self._chk(self.lib.tasu_f32_attn_prefill(_p(qkv), _p(kstart), _p(klen), _p(out), B, S, H, G, scale, self._stream()),
                  "tasu_f32_attn_prefill")

    def f32_fsmn(self, v, ldv, w, lens, out, B, T, D, ksize):
        self._chk(self.lib.tasu_f32_fsmn(_p(v), ldv, _p(w), _p(lens), _p(out), B, T, D, ksize, self._stream()), "tasu_f32_fsmn")

    def f32_attn_decode(self, qkv, kc, vc, index, kstart, lens, out, M, H, G, ctx, scale):
        self._chk(self.lib.tasu_f32_attn_decode(_p(qkv), _p(kc), _p(vc), _p(index), _p(kstart), _p(lens), _p(out), M, H, G, ctx, scale,
                                                self._stream()), "tasu_f32_attn_decode")

    def f32_swiglu(self, gu, act, M, I):
        self._chk(self.lib.tasu_f32_swiglu(_p(gu), _p(act), M, I, self._stream()), "tasu_f32_swiglu")

    def f32_embed_merge(self, table, proj, kind, idx, x, M, D):
        self._chk(self.lib.tasu_f32_embed_merge(_p(table), _p(proj), proj.stride(0), _p(kind), _p(idx), _p(x), M, D, self._stream()),
                  "tasu_f32_embed_merge")

    def f32_ce(self, logits, labels, M, V, row_loss, row_hit, row_argmax=None, row_lse=None, dlogits=None, inv_count=None):
        self._chk(self.lib.tasu_f32_ce(_p(logits), logits.stride(0), _p(labels), M, V, _p(row_loss), _p(row_hit), _p(row_argmax),
                                       _p(row_lse), _p(dlogits), _p(inv_count), self._stream()), "tasu_f32_ce")

    # fp32 training step: backward kernels (csrc/fp32_train.hip)
    def f32_rmsnorm_bwd(self, dy, x, w, dx, M, D, eps, accumulate):
        self._chk(self.lib.tasu_f32_rmsnorm_bwd(_p(dy), _p(x), _p(w), _p(dx), M, D, eps, int(accumulate), self._stream()), "tasu_f32_rmsnorm_bwd")

    def f32_swiglu_bwd(self, dact, gu, dgu, M, I):
        self._chk(self.lib.tasu_f32_swiglu_bwd(_p(dact), _p(gu), _p(dgu), M, I, self._stream()), "tasu_f32_swiglu_bwd")

    def f32_silu(self, x, out, dy=None):
        self._chk(self.lib.tasu_f32_silu(_p(x), _p(dy), _p(out), x.numel(), self._stream()), "tasu_f32_silu")

    def f32_colsum(self, x, out, R, Cn):
        self._chk(self.lib.tasu_f32_colsum(_p(x), x.stride(0), _p(out), R, Cn, self._stream()), "tasu_f32_colsum")

    def f32_layernorm_bwd_params(self, dy, x, mean, rstd, dgamma, dbeta, R, D):
        self._chk(self.lib.tasu_f32_layernorm_bwd_params(_p(dy), dy.stride(0), _p(x), x.stride(0), _p(mean), _p(rstd), _p(dgamma), _p(dbeta),
                                                         R, D, self._stream()), "tasu_f32_layernorm_bwd_params")

    def f32_transpose(self, src, dst, R, Cn, Rpad):
        self._chk(self.lib.tasu_f32_transpose(_p(src), src.stride(0), _p(dst), dst.stride(0), R, Cn, Rpad, self._stream()), "tasu_f32_transpose")

    def f32_gather_rows(self, dx, rows, out, n, D):
        self._chk(self.lib.tasu_f32_gather_rows(_p(dx), _p(rows), _p(out), n, D, self._stream()), "tasu_f32_gather_rows")

    def f32_attn_bwd(self, qkv, dout, kstart, dqkv, lse_ws, delta_ws, B, S, H, G, scale):
        self._chk(self.lib.tasu_f32_attn_bwd(_p(qkv), _p(dout), _p(kstart), _p(dqkv), _p(lse_ws), _p(delta_ws), B, S, H, G, scale,
                                             self._stream()), "tasu_f32_attn_bwd")

    def f32_logprob_topk(self, logits, M, V, k, banned, n_banned, out_val, out_idx, ws=None):
        """``ws``: M * 16 * (2 + 2 k) floats -- with it the row is split over 16 workgroups (the decode step's 64 rows)."""
        self._chk(self.lib.tasu_f32_logprob_topk(_p(logits), logits.stride(0), M, V, k, _p(banned), n_banned, _p(out_val), _p(out_idx),
                                                 _p(ws), ws.numel() if ws is not None else 0, self._stream()), "tasu_f32_logprob_topk")

    # ------------------------------------------------------------------ audio front end
    def fbank(self, wave, n_samples, scale, win, shift, window, mel, n_mels, preemph, out):
        self._chk(self.lib.tasu_fbank(_p(wave), n_samples, scale, win, shift, _p(window), _p(mel), n_mels, preemph, _p(out),
                                      self._stream()), "tasu_fbank")

    def lfr_cmvn(self, fb, T, D, lfr_m, lfr_n, means, scales, out):
        self._chk(self.lib.tasu_lfr_cmvn(_p(fb), T, D, lfr_m, lfr_n, _p(means), _p(scales), _p(out), self._stream()),
                  "tasu_lfr_cmvn")

    def embed_rows(self, table, ids, x, M, D):
        self._chk(self.lib.tasu_embed_rows(_p(table), _p(ids), _p(x), M, D, self._stream()), "tasu_embed_rows")
