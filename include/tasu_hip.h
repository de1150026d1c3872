/* tasu_hip.h -- C-ABI of libtasu_hip.so: the MI355X (gfx950) kernels behind the TASU alignment hot path
 * (SenseVoiceSmall -> CTC posterior -> LinearSiLU projector -> Qwen2.5 decoder -> CE, AdamW).
 *
 * The reference (PigeonDan1/ps-slm) is pure Python and has NO FFI of its own; its plugin boundary is
 * `model_factory(train_config, model_config, **kwargs)` (Multitask/model/ps-slm.py:130-181, selected by
 * Multitask/aispeech_asr_config.py:28 through Multitask/utils/model_utils.py:9-33).  Each entry point below
 * replaces the PyTorch-eager arithmetic of one span of that model's forward/backward; the span is cited on
 * every declaration.  INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * TRAINING CONTRACT.  These entry points are forward AND backward kernels called in a hand-scheduled order by
 * ps_slm_amd.model.TasuModel; no torch operator runs across this boundary.  The reference's loop
 * (Multitask/finetune_deepspeed.py:127-149, Multitask/utils/deepspeed_utils.py:205-236: an optimizer / `deepspeed.initialize` over
 * `model.parameters()`, `engine.backward(loss)`, `engine.step()`) reaches them through ONE autograd node: `outputs.loss` is the result
 * of ps_slm_amd.ps_slm._HipStep, whose backward runs the whole hand-scheduled backward and hands the trainable leaves their slices of
 * the flat gradient bucket (tests/test_engine_cpu.py, tests/test_gpu_engine.py).  ps_slm_amd.engine.TasuEngine (same backward / step
 * surface; tasu_adamw + tasu_comm_* / tasu_allreduce_f32 below) is the fast path: overlapped RCCL exchange, fused AdamW.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is DEVICE memory unless stated; no ownership transfer;
 *     the caller allocates every output and workspace.
 *   - `stream` is a hipStream_t (NULL = default stream); every call is asynchronous on it.
 *   - return value: 0 = TASU_OK, 1 = bad argument (nothing launched), 2 = launch failure.
 *   - bf16 = IEEE bfloat16 stored as uint16; "rows" are contiguous along the last dimension; `ld*` are
 *     leading dimensions in ELEMENTS.
 *   - token-major activations: [M, D] with M = B*S.
 */
#ifndef TASU_HIP_H_
#define TASU_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TASU_ABI_VERSION 16
int tasu_abi_version(void);

/* ---------------------------------------------------------------------------------------------- GEMM
 * C[M,N] (+)= A[M,K] . B[N,K]^T (+ bias[N]);  A, B, bias bf16;  fp32 accumulation on MFMA.
 * Requirements: K % 64 == 0 (pad K with zeros in the layout), lda % 8 == 0, ldb % 8 == 0, A/B 16-B aligned.
 * Replaces every nn.Linear on the path: transformers modeling_qwen2.py:206-208 (q/k/v), :192 (o_proj),
 * :46-48 (gate/up/down), :465 (lm_head); Multitask/model/projector.py:141-143; SenseVoice.py:66-67,109-110;
 * and their autograd dgrad/wgrad (called with the transposed resident copies).                          */
#define TASU_GEMM_OUT_BF16 0          /* C bf16  = round(acc + bias)                                    */
#define TASU_GEMM_OUT_F32 1           /* C fp32  = acc + bias                                           */
#define TASU_GEMM_OUT_F32_RESID_BF16R 2 /* C fp32 = resid fp32 + bf16_round(acc + bias); resid has ldc, may == C */
int tasu_gemm_nt_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* bias,
                      const float* resid, int M, int N, int K, int out_mode, void* stream);

/* Same GEMM with a caller-owned device workspace that lets small grids split the K range over several blocks per tile
 * (256 x 192 tiles; the last-arriving block of a tile sums the fp32 partials in split order: deterministic).
 * Workspace layout: TASU_GEMM_WS_COUNTERS int32 arrival counters, which the caller zeroes ONCE (every launch leaves
 * them zero), followed by the partial tiles; 64 MiB + 16 KiB covers every shape of the path.  Launches that share a
 * workspace must be ordered on one stream.  workspace may be NULL (identical to tasu_gemm_nt_bf16).                  */
#define TASU_GEMM_WS_COUNTERS 4096
int tasu_gemm_nt_bf16_ws(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* bias,
                         const float* resid, int M, int N, int K, int out_mode, void* workspace,
                         int64_t workspace_bytes, void* stream);
/* C = bf16(relu(bf16(A . B^T + bias))): PositionwiseFeedForward's w_1 followed by its ReLU (Multitask/model/SenseVoice.py:71-73) in
 * one launch -- the ReLU runs in the GEMM kernels' epilogue (max before the single bf16 rounding: the same bits as tasu_gemm_nt_bf16
 * + tasu_relu_fwd, which is what runs for the at-most-128-row shapes the small-tile kernels serve; those need ldc == N).  */
int tasu_gemm_bias_relu_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* bias, int M, int N, int K,
                             void* workspace, int64_t workspace_bytes, void* stream);

/* tasu_gemm_nt_bf16 on a NAMED kernel, regardless of the tile policy of tasu_gemm_nt_bf16_ws (tests compare the kernels with
 * each other bit for bit -- they accumulate every output element in the same K order -- and tuning runs time them side by side):
 *   PP256    256 x 256 tiles, eight MFMA waves in two groups one barrier apart, every wave stages its own share of the
 *            operands by LDS-DMA (csrc/gemm_pp.hip); needs K % 128 == 0, K >= 256
 *   PIPE128 / PIPE192 / PIPE96   256 x 128 / 128 x 192 / 256 x 96 tiles, four MFMA waves + four loader waves (csrc/gemm_pipe.hip) */
#define TASU_GEMM_KERNEL_PP256 1
#define TASU_GEMM_KERNEL_PIPE128 2
#define TASU_GEMM_KERNEL_PIPE192 3
#define TASU_GEMM_KERNEL_PIPE96 4
int tasu_gemm_nt_bf16_kernel(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* bias,
                             const float* resid, int M, int N, int K, int out_mode, int kernel, void* stream);

/* Qwen2MLP gate|up projection with the activation in its epilogue (modeling_qwen2.py Qwen2MLP.forward), training step:
 * gu[M, 2I] = bf16(A[M,K] . Wgu[2I,K]^T) (gate columns first; saved for the backward) and
 * act[M, I] = bf16( bf16(silu(g)) * u ) -- bit-identical to tasu_gemm_nt_bf16 + tasu_swiglu_fwd.  I % 4 == 0.        */
int tasu_gemm_gate_up_swiglu(const void* A, int lda, const void* Wgu, int ldw, void* gu, void* act, int M, int I, int K,
                             void* stream);
/* Qwen2Attention's q|k|v projection with bias and the rotary embedding of the q and k heads in one launch (training step and
 * prefill; transformers modeling_qwen2.py:91-135, 150-172):
 *   qkv[M, (H + 2G) * 128] = bf16(A[M, K] . Wqkv^T + bias), then for every q / k head and d < 64
 *   out[d] = bf16(x[d] * cos[m][d] - x[d+64] * sin[m][d]),  out[d+64] = bf16(x[d+64] * cos[m][d] + x[d] * sin[m][d])
 * (cos / sin: fp32 [M, 64] from tasu_rope_table) -- bit-identical to tasu_gemm_nt_bf16 + tasu_rope_fwd, which is what runs with
 * TASU_GEMM_QKV_ROPE=0.  The kernel's loader hands each MFMA wave both members of its rotation pairs, so the rotation costs
 * no exchange and qkv keeps its layout.  K % 64 == 0, head_dim 128, 16-byte aligned operands; bias may be NULL. */
int tasu_gemm_qkv_rope(const void* A, int lda, const void* Wqkv, int ldw, const void* bias, void* qkv, const float* cos_tab,
                       const float* sin_tab, int M, int H, int G, int K, void* workspace, int64_t workspace_bytes, void* stream);
/* Qwen2MLP backward, first half, in one call (the training step's d_down + SwiGLU backward):
 *   dact[M, I] = bf16(dy[M, K] . WdT[I, K]^T)   (WdT = down_proj.weight transposed once at load time: [I, K = hidden])
 *   dgu[m, n]     = bf16(d * u * sig(g) * (1 + g * (1 - sig(g)))),  dgu[m, I + n] = bf16(d * g * sig(g)),
 *   g = gu[m, n], u = gu[m, I + n] (the gate|up matrix tasu_gemm_gate_up_swiglu saved), d = dact[m, n]
 * = tasu_gemm_nt_bf16 (bf16 output) into dact_ws (bf16 [M, I], required) followed by tasu_swiglu_bwd.  (A form with the SwiGLU
 * backward in the GEMM kernels' epilogue -- dact never reaches memory, same bits -- was measured worth 0.3 % of the step and
 * lives in the lab build only: make -C ps_slm_amd/csrc lab.)  I % 8 == 0, K % 64 == 0, 16-byte aligned operands; workspace as
 * for tasu_gemm_nt_bf16_ws (may be NULL). */
int tasu_gemm_dswiglu(const void* dy, int lddy, const void* WdT, int ldw, const void* gu, void* dgu, void* dact_ws, int M, int I,
                      int K, void* workspace, int64_t workspace_bytes, void* stream);
/* ... with the workspace of tasu_gemm_nt_bf16_ws (or NULL): the 256 x 256 kernel may then cut its last rounds of tiles along K
 * (stream-K, below).  gu / act are bit-identical to the call without a workspace only where no tile is cut. */
int tasu_gemm_gate_up_swiglu_ws(const void* A, int lda, const void* Wgu, int ldw, void* gu, void* act, int M, int I, int K,
                                void* workspace, int64_t workspace_bytes, void* stream);
/* tasu_gemm_nt_bf16_ws on the 256 x 256 eight-wave kernel with the STREAM-K schedule wherever the output tiles do not fill
 * whole rounds of workgroups (one per CU), whatever tile the dispatcher's policy would pick: the trailing tiles are cut along K
 * into one contiguous range of K-tile pairs per workgroup; a range that does not begin its tile leaves an fp32 partial tile in
 * the workspace, and the workgroup that holds the tile's first K-tiles adds the partials IN K ORDER and runs the epilogue -- the
 * result is bitwise repeatable, and differs from the unsplit kernel's only in the association of the fp32 sum over K.
 * tasu_gemm_nt_bf16_ws takes this path by itself for the K-deep N = 1536 projections (96 tiles on 256 CUs) and for shapes
 * whose last round of tiles would be mostly empty.  K % 128 == 0, K >= 256; workspace >= 64 MiB + 16 KiB, zero-initialised
 * once (every launch leaves its flag words at zero); launches sharing a workspace must be ordered on one stream. */
int tasu_gemm_nt_bf16_streamk(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* bias,
                              const float* resid, int M, int N, int K, int out_mode, void* workspace, int64_t workspace_bytes,
                              void* stream);
/* The dispatcher's decision for C[M,N] = A[M,K] . B[N,K]^T in tasu_gemm_nt_bf16_ws (with_workspace != 0) / tasu_gemm_nt_bf16,
 * without launching anything (no GPU needed; tests pin the policy on the training step's shapes).  Returns one of: */
#define TASU_GEMM_PLAN_PP256 1                /* 256 x 256 eight-wave kernel, whole tiles */
#define TASU_GEMM_PLAN_PP256_PLUS_PIPE128 2   /* whole rounds of 256 x 256 tiles + the remaining columns on 256 x 128 tiles */
#define TASU_GEMM_PLAN_PP256_PLUS_PIPE192 3   /* ... + the remaining columns on 128 x 192 tiles */
#define TASU_GEMM_PLAN_PP256_STREAMK 4        /* 256 x 256 tiles cut along K, one range per workgroup */
#define TASU_GEMM_PLAN_PIPE128 5              /* loader-wave kernel, 256 x 128 tiles */
#define TASU_GEMM_PLAN_PIPE192 6              /* loader-wave kernel, 128 x 192 tiles */
#define TASU_GEMM_PLAN_PIPE96 7               /* loader-wave kernel, 256 x 96 tiles */
#define TASU_GEMM_PLAN_TILE192_SPLITK 8       /* 256 x 192 tiles of gemm.hip with split-K through the workspace */
#define TASU_GEMM_PLAN_TILES 9                /* 128-row tiles of gemm.hip (at most 64 rows, or a forced kernel) */
/* (-1: the arguments would be rejected) */
int tasu_gemm_plan(int M, int N, int K, int out_mode, int with_workspace);
/* Host-side restatement of that schedule through the kernel's own code (no GPU needed; tests): the work-item lists of `grid`
 * workgroups for `tiles` output tiles of `pairs` K-tile pairs (K = 128 * pairs).  items[w][i] = {tile, first K-tile, K-tiles,
 * role: 0 whole tile, 1 partial-tile producer, 2 tile owner}, max_items per workgroup; counts[w] = items of workgroup w.
 * Returns the number of (trailing) tiles cut along K, or -1 on bad arguments. */
int tasu_streamk_schedule(int tiles, int pairs, int grid, int32_t* items, int32_t* counts, int max_items);
/* Split-K form of the NT GEMM for outputs too small to fill the chip behind a very long K (the lm_head dgrad over
 * the labelled rows: [n_rows, 1536] outputs, K = padded vocabulary): partials[s] [M, ldc] fp32 = A[:, Ks] . B[:, Ks]^T
 * for the ksplit contiguous K ranges (K % (64 * ksplit) == 0), written as ksplit consecutive [M, ldc] matrices;
 * tasu_sum_slabs_bf16 adds them in slab order and rounds ONCE to bf16 (the rounding point of the unsplit GEMM). */
int tasu_gemm_nt_bf16_splitk(const void* A, int lda, const void* B, int ldb, float* partials, int ldc, int M, int N, int K,
                             int ksplit, void* stream);
int tasu_sum_slabs_bf16(const float* slabs, int n_slabs, int64_t slab_stride, void* out_bf16, int64_t n, void* stream);
/* The same K-range slabs from the 256 x 256 eight-wave kernel (csrc/gemm_pp.hip): for outputs that fill the chip neither with
 * 256 x 256 tiles nor (efficiently) with small ones behind a long K -- the decoder's N = 1536 projections at K = 8960 / 17920,
 * where a CU's L2 -> LDS ingest per FLOP, not the matrix pipe, bounds the 128 x 192 one-tile-per-CU grid.
 * K % (128 * ksplit) == 0 and K / ksplit >= 256. */
int tasu_gemm_nt_bf16_slabs(const void* A, int lda, const void* B, int ldb, float* partials, int ldc, int M, int N, int K,
                            int ksplit, void* stream);

/* Weight-streaming GEMM for M <= 64 rows (the decode step; transformers modeling_qwen2.py linears at one token per beam).
 * workspace: fp32 split-K slabs, 32 * 64 * round_up(N, 96) floats always suffice; may be NULL (no K split).  Launches
 * sharing a workspace must be ordered on one stream.  Operand rules and out_mode as tasu_gemm_nt_bf16.              */
int tasu_gemm_skinny_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* bias,
                          const float* resid, int M, int N, int K, int out_mode, float* workspace,
                          int64_t workspace_floats, void* stream);
/* Projection + residual add + the NEXT RMSNorm in one call (o / down projection of a Qwen2DecoderLayer at M <= 64):
 * C[M,N] fp32 (row stride N) = resid + bf16(A[M,K] . B[N,K]^T);  y[M,N] bf16 = norm_w * (C * rsqrt(mean(C^2) + eps)).   */
int tasu_gemm_skinny_norm(const void* A, int lda, const void* B, int ldb, float* C, const float* resid, int M, int N, int K,
                          const float* norm_w, void* y, float eps, int y_frag, float* workspace, int64_t workspace_floats,
                          void* stream);
/* q|k|v projection (+ bias) of a decode step with RoPE and the cache append behind it (Qwen2Attention.forward at one token
 * per beam): qkv[M, (H+2G)*128] bf16 rotated in place semantics of tasu_rope_append; k, v -> cache[row, pos[row]].     */
int tasu_gemm_skinny_qkv_rope(const void* A, int lda, const void* Wqkv, int ldw, const void* bias, void* qkv, int M, int H,
                              int G, int K, const float* cos_tab, const float* sin_tab, void* kcache, void* vcache,
                              const int32_t* pos, int ctx, float* workspace, int64_t workspace_floats, void* stream);

/* Second-generation decode-step GEMMs (csrc/gemm_stream.hip): activations in registers, weights global -> registers,
 * persistent column tiles, the consuming op's epilogue in the SAME launch (no split-K finish launch).  M <= 64;
 * K (per K range) in {256, 512, 1280, 1536, 1792}, ONE range of 3584 (Qwen2.5-7B: 14 k-steps per wave on row halves), or ksplit - 1
 * equal ranges and a shorter last one (the 7B's down projection: 18944 = 12 x 1536 + 512) -- tasu_stream_supported(K, ksplit) says
 * whether a shape is served; other
 * shapes stay on the tasu_gemm_skinny_* entry points, whose arithmetic and rounding points these reproduce:
 *   tasu_gemm_stream_bf16      C = bf16(A W^T + bias)  (out_mode TASU_GEMM_OUT_BF16; lm_head) or
 *                              C(fp32) = resid + bf16(A W^T)  (TASU_GEMM_OUT_F32_RESID_BF16R; o projection + residual)
 *   tasu_gemm_stream_swiglu    act = bf16(bf16(silu(g)) * u), g | u = A Wgu^T   (modeling_qwen2.py:41-48)
 *   tasu_gemm_stream_qkv_rope  qkv = rope(A Wqkv^T + bias), k and v appended to the cache at pos[m]  (:189-208, :91-135)
 *   tasu_gemm_stream_slabs     K split over workgroups (K = ksplit * {256..1792}: the down projection, 8960 = 5 x 1792):
 *                              fp32 partial results, row-major [ksplit][64][N] (slab_floats >= ksplit * 64 * N);
 *                              tasu_stream_finish_norm (N a multiple of 256, N / 256 in {1, 2, 6, 7, 14}) adds them in order,
 *                              then C(fp32) = resid + bf16(sum) and y = rmsnorm(C, norm_w) for the next layer       */
int tasu_stream_supported(int K, int ksplit);
/* a_frag / w_frag / out_frag / y_frag = 1: that operand is in FRAGMENT ORDER (the order mfma_f32_16x16x32_bf16 consumes it
 * in, so that every wave instruction reads 1 KiB contiguous instead of 16 rows x 64 B):
 *   activations [<= 64, K]:  X_f[K / 32][4 row tiles][64 lanes][8],  lane = 16 * ((k % 32) / 8) + row % 16, element k % 8
 *   weights: per 16-row column tile, in the row order of the tile's epilogue: tasu_to_fragment_order(kind = 0 plain,
 *   2 SwiGLU (8 gate + 8 up rows), 3 q|k|v (paired RoPE columns); N = output columns) -> [tile][K / 32][64][8].           */
int tasu_gemm_stream_bf16(const void* A, int lda, const void* W, int ldw, void* C, int ldc, const void* bias,
                          const float* resid, int M, int N, int K, int out_mode, int a_frag, int w_frag, void* stream);
int tasu_gemm_stream_swiglu(const void* A, int lda, const void* Wgu, int ldw, void* act, int ldact, int M, int I, int K,
                            int a_frag, int w_frag, int out_frag, void* stream);
int tasu_gemm_stream_qkv_rope(const void* A, int lda, const void* Wqkv, int ldw, const void* bias, void* qkv, int M, int H,
                              int G, int K, const float* cos_tab, const float* sin_tab, void* kcache, void* vcache,
                              const int32_t* pos, int ctx, int a_frag, int w_frag, void* stream);
int tasu_gemm_stream_slabs(const void* A, int lda, const void* W, int ldw, float* slabs, int64_t slab_floats, int M, int N,
                           int K, int ksplit, int a_frag, int w_frag, void* stream);
int tasu_stream_finish_norm(const float* slabs, int ksplit, float* C, const float* resid, int M, int N, const float* norm_w,
                            void* y, float eps, int y_frag, void* stream);
/* Round 5: the post-attention RMSNorm of a decode layer WITHOUT a launch of its own (Qwen2DecoderLayer, modeling_qwen2.py:269-298,
 * through /root/reference/Multitask/model/ps-slm.py:660-675).  rstd is a per-row scalar, so
 *     (norm_w . x . rstd) Wgu^T  =  rstd . ((norm_w . x) Wgu^T):
 *   tasu_gemm_stream_resid_prenorm  the o projection + residual (C fp32 = resid + bf16(A W^T), as tasu_gemm_stream_bf16's RESID form)
 *                                   also writes yw = bf16(norm_w . C) (row-major [M, N] or, yw_frag, fragment order) and, per
 *                                   16-column tile, the sum of squares of the tile's C values of every row: sumsq[N / 16][64] fp32
 *                                   (rows >= M are not written: keep the buffer zero-initialised);
 *   tasu_gemm_stream_swiglu_rstd    tasu_gemm_stream_swiglu on A = yw whose accumulators are scaled by
 *                                   rstd[m] = rsqrt(sum_t sumsq[t][m] / K + eps) (n_part = N / 16 partials) before the SwiGLU.
 * Against the separate norm kernel the bf16 rounding of the normed activation happens before the multiplication by rstd instead of
 * after it: the same size of error, not the same bits (tests: tolerance against the two-launch form, exact tokens on the
 * rounding-stable decode cases).  M <= 64, N % 16 == 0, tasu_stream_supported(K, 1) and K <= 2048 (at K = 3584 the kernels have no
 * register to spare: the 7B geometry keeps the norm launch).                                              */
int tasu_gemm_stream_resid_prenorm(const void* A, int lda, const void* W, int ldw, float* C, const float* resid, int M, int N, int K,
                                   const float* norm_w, void* yw, int yw_frag, float* sumsq, int a_frag, int w_frag, void* stream);
int tasu_gemm_stream_swiglu_rstd(const void* A, int lda, const void* Wgu, int ldw, void* act, int ldact, int M, int I, int K,
                                 const float* sumsq, int n_part, float eps, int a_frag, int w_frag, int out_frag, void* stream);
/* The same for a layer's INPUT norm behind the down projection's K-range slabs: tasu_stream_finish_prenorm is tasu_stream_finish_norm
 * without the norm's whole-row dependency (C = resid + bf16(sum of the slabs); yw = bf16(norm_w . C); sumsq[N / 16][64] partials; a wave
 * per 256 columns of a row instead of a wave per row), tasu_gemm_stream_qkv_rope_rstd is tasu_gemm_stream_qkv_rope on A = yw with the
 * accumulators scaled by rstd before bias and RoPE.  N % 256 == 0; K <= 2048.                                                        */
int tasu_stream_finish_prenorm(const float* slabs, int ksplit, float* C, const float* resid, int M, int N, const float* norm_w, void* yw,
                               int yw_frag, float* sumsq, void* stream);
int tasu_gemm_stream_qkv_rope_rstd(const void* A, int lda, const void* Wqkv, int ldw, const void* bias, void* qkv, int M, int H, int G,
                                   int K, const float* cos_tab, const float* sin_tab, void* kcache, void* vcache, const int32_t* pos,
                                   int ctx, const float* sumsq, int n_part, float eps, int a_frag, int w_frag, void* stream);
/* Round 5: the projection, the residual add AND the RMSNorm of the finished rows in ONE launch (csrc/stream_body.h: norm_tail) --
 * a decode layer's two norm launches (tasu_rmsnorm_fwd_frag behind the o projection, tasu_stream_finish_norm behind the down
 * projection's slabs: Qwen2DecoderLayer, modeling_qwen2.py:269-298, through /root/reference/Multitask/model/ps-slm.py:660-675) move into
 * their producers: the projection's workgroups store write-through, take a ticket, and the last min(workgroups, M) arrivers each
 * normalise rows of the complete result (agent-scope hand-off, no grid barrier).
 *   C [M, N] fp32 = resid + bf16(A W^T);  y = bf16(norm_w * C * rstd), row-major [M, N] or (y_frag) fragment order.
 *   ksplit = 1: K in one range (tasu_gemm_stream_bf16's RESID form + tasu_rmsnorm_fwd[_frag]); ksplit > 1: K-range slabs in
 *   `slabs` ([ksplit][64][N] fp32, slab_floats >= ksplit * 64 * N) summed in slab order (tasu_gemm_stream_slabs +
 *   tasu_stream_finish_norm).  N = 256 or 1536, M <= 64, tasu_stream_supported(K, ksplit).  `sync`: 8 bytes of device memory,
 *   zero before the FIRST call; every call (and hipGraph replay) leaves them zero.  The same bits as the two-launch forms. */
int tasu_gemm_stream_norm(const void* A, int lda, const void* W, int ldw, float* C, const float* resid, int M, int N, int K,
                          int ksplit, float* slabs, int64_t slab_floats, const float* norm_w, void* y, float eps, int a_frag,
                          int w_frag, int y_frag, void* sync, void* stream);
/* y_f = rmsnorm(x, w) written in fragment order (M <= 64, D % 32 == 0): the decode step's first / post-attention norm. */
int tasu_rmsnorm_fwd_frag(const float* x, const float* w, void* y_frag, int M, int D, float eps, void* stream);
int tasu_to_fragment_order(const void* W, int ldw, void* out, int kind, int N, int K, int H, int G, void* stream);
/* Qwen2MLP gate|up projection + activation in one launch (modeling_qwen2.py Qwen2MLP.forward, M <= 64):
 * act[M, I] = bf16( bf16(silu(g)) * u ),  g | u = bf16(A[M,K] . Wgu[2I,K]^T)  (gate rows first, then up rows).        */
int tasu_gemm_skinny_swiglu(const void* A, int lda, const void* Wgu, int ldw, void* act, int ldact, int M, int I, int K,
                            float* workspace, int64_t workspace_floats, void* stream);

/* Tiled transpose out[c][r] = in[r][c], bf16 (used to feed wgrad through the NT GEMM). rows<=R and
 * cols<=C outside [R,C) of `out` up to (Cpad, Rpad) are written as zero so K-padding stays exact.       */
int tasu_transpose_bf16(const void* in, int ld_in, void* out, int ld_out, int R, int C, int Rpad, int Cpad,
                        void* stream);
/* fp32 -> bf16 cast (same layout) and fp32 -> bf16 transposed copy; n / R x C as above.                */
int tasu_cast_f32_bf16(const void* in, void* out, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------- RMSNorm
 * Qwen2RMSNorm (modeling_qwen2.py:247-252): y = bf16(w * (x * rsqrt(mean(x^2) + eps))), x fp32 residual
 * stream.  rstd[M] is saved for backward.  Backward (frozen weight => dgrad only):
 * dx += rstd * (w.dy - xhat * mean(w.dy.xhat)).                                                         */
int tasu_rmsnorm_fwd(const float* x, const float* w, void* y_bf16, float* rstd, int M, int D, float eps,
                     void* stream);
/* accumulate != 0: dx += ...; else dx = ...   dx_bf16 (optional) receives bf16(dx) after the update: the
 * gradient the next dgrad GEMM consumes.                                                                */
int tasu_rmsnorm_bwd(const void* dy_bf16, const float* x, const float* w, const float* rstd, float* dx,
                     void* dx_bf16, int accumulate, int M, int D, void* stream);
/* Row-indexed forms for the training step's lm_head, which projects only the positions that carry a label
 * (transformers loss_utils.py:49-71 ignores the others; ps_slm_amd/model.py "labelled rows"):
 * fwd: y[i,:] = rmsnorm(x[src_rows[i],:]) for i < n_rows, a zero row (rstd 0) where src_rows[i] < 0;
 * bwd: dy / rstd are compact [n_rows]; for every m < M: s = slot[m]; dx[m,:] = dgrad(dy[s], x[m], rstd[s]) if
 * s >= 0 else 0 (no accumulation), dx_bf16 likewise.                                                        */
int tasu_rmsnorm_fwd_rows(const float* x, const int32_t* src_rows, const float* w, void* y, float* rstd, int n_rows, int D,
                          float eps, void* stream);
int tasu_rmsnorm_bwd_rows(const void* dy_compact, const float* x, const float* w, const float* rstd_compact,
                          const int32_t* slot, float* dx, void* dx_bf16, int M, int D, void* stream);
/* ... and with the rows' residual-stream gradient given COMPACT as well (the last decoder layer's MLP runs on the labelled rows
 * only: positions without a label feed nothing there): dx[m,:] = resid[s,:] + dgrad(dy[s], x[m], rstd[s]) if s = slot[m] >= 0
 * else 0 -- the scatter back to all M rows, for the layer's attention backward. */
int tasu_rmsnorm_bwd_rows_resid(const void* dy_compact, const float* x, const float* w, const float* rstd_compact,
                                const int32_t* slot, const float* resid_compact, float* dx, void* dx_bf16, int M, int D,
                                void* stream);

/* ---------------------------------------------------------------------------------------------- RoPE
 * cos/sin tables from position ids (modeling_qwen2.py:91-102): tab[m][i] = cos/sin(pos[m] * theta^(-2i/hd)),
 * i < hd/2, fp32.                                                                                       */
int tasu_rope_table(const int32_t* pos, float* cos_tab, float* sin_tab, int M, int head_dim, float theta,
                    void* stream);
/* Rotate q and k heads of the fused qkv activation IN PLACE (rotate-half convention, modeling_qwen2.py:
 * 113-135, fp32 math, bf16 result):
 *   qkv    [M, (H+2G)*128] bf16 : q heads | k heads | v heads
 *   qt     [B, H, 128, S], kt [B, G, 128, S], vt [B, G, 128, S]: OPTIONAL transposed copies (NULL = not written; the
 *   attention kernels no longer read them: since round 2 they transpose in LDS with ds_read_b64_tr_b16)       */
int tasu_rope_fwd(void* qkv, const float* cos_tab, const float* sin_tab, void* qt, void* kt, void* vt,
                  int B, int S, int H, int G, void* stream);
/* Autograd of the above on dqkv [M,(H+2G)*128]: the q block (dQ in rotated space, written by tasu_attn_bwd_dq)
 * is un-rotated in place; the k and v blocks are produced from the per-query-head fp32 partials of
 * tasu_attn_bwd_dkv (sum over the H/G heads of each kv group, un-rotate K, round to bf16).               */
int tasu_rope_bwd(void* dqkv, const float* dk_part, const float* dv_part, const float* cos_tab,
                  const float* sin_tab, int B, int S, int H, int G, void* stream);

/* ----------------------------------------------------------------------------------------- attention
 * Causal grouped-query attention with key padding, head_dim 128, bf16 in/out, fp32 online softmax
 * (modeling_qwen2.py:150-172 / SDPA).  q/k are read from the (rotated) fused qkv buffer, V through its
 * transposed copy.  key_mask [B,S] uint8 (1 = attend).  Rows with no visible key produce 0.
 *   out [M, H*128] bf16,  lse [B,H,S] fp32 (log-sum-exp of the scaled scores, for backward).
 * causal = 0 gives the bidirectional SANM attention of the encoder (SenseVoice.py:171-207).              */
int tasu_attn_fwd(const void* qkv, const void* vt, const uint8_t* key_mask, void* out, float* lse, int B, int S,
                  int H, int G, float scale, int causal, void* stream);
/* The vt / kt / qt / dout_t arguments of the attention entry points are UNUSED since round 2 (pass NULL): the kernels read
 * V^T, K^T, Q^T and dO^T out of the token-major tiles in LDS with the hardware transpose read.
 * Backward.  prep: delta[b,h,s] = sum_d dO.O (and, only if dout_t != NULL, dOt [B,H,128,Spad]);  dq: dQ (rotated space) into the q block
 * of dqkv;  dkv: fp32 partials dk_part / dv_part [M, (H/HPB)*128] (no atomics; see TASU_ATTN_DKV_HPB).  Spad = S rounded up to 64; key_mask is [B, Spad] (pad = 0); lse/delta are
 * [B, H, Spad]; qt/kt/vt/dOt are [B, heads, 128, Spad] with zero token padding.                           */
int tasu_attn_bwd_prep(const void* dout, const void* out, float* delta, void* dout_t, int B, int S, int H,
                       void* stream);
int tasu_attn_bwd_dq(const void* qkv, const void* kt, const uint8_t* key_mask, const void* dout, const float* lse,
                     const float* delta, void* dqkv, int B, int S, int H, int G, float scale, int causal,
                     void* stream);
/* query heads of one kv group handled (and summed in registers) per block of tasu_attn_bwd_dkv; dk_part / dv_part are
 * fp32 [M, (H / HPB) * 128]: one partial per HPB-head group, reduced over the remaining (H/G)/HPB by tasu_rope_bwd. */
#define TASU_ATTN_DKV_HPB(rep) (((rep) % 3 == 0) ? 3 : (((rep) % 2 == 0) ? 2 : 1))
int tasu_attn_bwd_dkv(const void* qkv, const void* qt, const uint8_t* key_mask, const void* dout, const void* dout_t,
                      const float* lse, const float* delta, float* dk_part, float* dv_part, int B, int S, int H, int G,
                      float scale, int causal, void* stream);
/* tasu_attn_bwd_dq + tasu_attn_bwd_dkv in ONE launch (same arguments, same results): the dQ and dK/dV blocks share the
 * grid, so neither kernel's tail leaves CUs idle. */
int tasu_attn_bwd(const void* qkv, const void* qt, const void* kt, const uint8_t* key_mask, const void* dout,
                  const void* dout_t, const float* lse, const float* delta, void* dqkv, float* dk_part, float* dv_part, int B,
                  int S, int H, int G, float scale, int causal, void* stream);
/* tasu_attn_bwd followed by tasu_rope_bwd (modeling_qwen2.py:113-135 reversed) behind one entry point: dqkv receives the
 * finished gradient of the UNROTATED q | k | v projection.  `kernel`:
 *   TASU_ATTN_KERNEL_PER_HEAD  the two launches above (dk_part / dv_part: their fp32 partials)
 *   TASU_ATTN_KERNEL_GQA       ONE launch of csrc/attention_gqa.hip (round 4; served when H / G >= 2 and S <= 4096:
 *                              tasu_attn_gqa_supported, else bad argument): the query heads of a GQA group share the K / V and
 *                              Q / dO tiles a workgroup stages (LDS-DMA, four deep), dK / dV are complete in their workgroup (no
 *                              partials: dk_part / dv_part untouched, may be NULL), the rotation runs in the epilogues.  dq: the
 *                              same bits as the per-head kernels; dk / dv: the same products in another fp32 association
 *   TASU_ATTN_KERNEL_POLICY    the GQA kernel wherever it is served (round 5: faster or equal at every measured shape once both
 *                              families got the cheaper softmax arithmetic), the per-head kernels otherwise (H == G;
 *                              dk_part / dv_part required) */
#define TASU_ATTN_KERNEL_POLICY 0
#define TASU_ATTN_KERNEL_PER_HEAD 1
#define TASU_ATTN_KERNEL_GQA 2
#define TASU_ATTN_KERNEL_SP 3
int tasu_attn_gqa_supported(int S, int H, int G);
int tasu_attn_bwd_rope(const void* qkv, const uint8_t* key_mask, const void* dout, const float* lse, const float* delta,
                       const float* cos_tab, const float* sin_tab, void* dqkv, float* dk_part, float* dv_part, int B, int S,
                       int H, int G, float scale, int causal, int kernel, void* stream);
/* Round 5: single-pass kernels for sequences of at most 256 (padded) positions (csrc/attention_sp.hip) -- the alignment step's
 * decoder attention (SDPA in Qwen2Attention.forward, modeling_qwen2.py:150-172, via /root/reference/Multitask/model/ps-slm.py:530).
 * The whole K / V (forward, dQ) or Q / dO (dK / dV) of one (batch, head) is resident in LDS (128 KiB, one LDS-DMA burst): QK^T
 * against every visible key, ONE softmax, ONE P.V -- no key-tile loop, no online rescale.  Same results as the tiled kernels up
 * to the association of the fp32 sums (softmax denominators, dK / dV over the heads of a group).
 *   tasu_attn_sp_supported  1 when Spad <= 256, H % G == 0 and G <= 16
 *   tasu_attn_fwd_kernel    tasu_attn_fwd on a chosen kernel: TASU_ATTN_KERNEL_PER_HEAD = the tiled kernel (any S), _SP = the
 *                           single-pass kernel (bad argument when unsupported), _POLICY = the tiled kernel (measured faster at
 *                           every shape tried, tools/bench_attn_sp.py); tasu_attn_fwd is the _POLICY form
 *   tasu_attn_bwd_fused     the WHOLE attention backward: delta = rowsum(dO . O) (tasu_attn_bwd_prep), dQ / dK / dV and the rotary
 *                           embedding's backward; `out` = the forward's output.  _SP: two launches (delta is computed inside the
 *                           kernel, `delta` is not touched; dk_part / dv_part = fp32 [M, H * 128] each: one partial per QUERY head,
 *                           summed over the group's heads, un-rotated and rounded by the second launch); _PER_HEAD / _GQA:
 *                           tasu_attn_bwd_prep + tasu_attn_bwd_rope with that kernel (`delta` [B, H, Spad] scratch required);
 *                           _POLICY: _SP where measured faster (Spad <= 256 and 3 B H <= 320: small batches), else
 *                           tasu_attn_bwd_rope's policy                                                                      */
int tasu_attn_sp_supported(int S, int H, int G);
int tasu_attn_fwd_kernel(const void* qkv, const uint8_t* key_mask, void* out, float* lse, int B, int S, int H, int G, float scale,
                         int causal, int kernel, void* stream);
int tasu_attn_bwd_fused(const void* qkv, const uint8_t* key_mask, const void* dout, const void* out, const float* lse, float* delta,
                        const float* cos_tab, const float* sin_tab, void* dqkv, float* dk_part, float* dv_part, int B, int S, int H,
                        int G, float scale, int causal, int kernel, void* stream);

/* -------------------------------------------------------------------------------------------- SwiGLU
 * act = bf16(bf16(silu(gate)) * up) on the fused [M, 2I] gate|up activation (modeling_qwen2.py:46-48),
 * and its backward dgu = [dact*up*silu'(gate) | dact*silu(gate)].                                        */
int tasu_swiglu_fwd(const void* gu, void* act, int M, int I, void* stream);
int tasu_swiglu_bwd(const void* dact, const void* gu, void* dgu, int M, int I, void* stream);
/* Plain SiLU (projector.py:142) fwd/bwd, bf16, and ReLU (SenseVoice.py:63) fwd.                         */
int tasu_silu_fwd(const void* x, void* y, int64_t n, void* stream);
int tasu_silu_bwd(const void* dy, const void* x, void* dx, int64_t n, void* stream);
int tasu_relu_fwd(const void* x, void* y, int64_t n, void* stream);
/* ReLU backward (EncoderProjectorConcat, projector.py:35): dx = dy where x > 0 else 0.                     */
int tasu_relu_bwd(const void* dy, const void* x, void* dx, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------ LoRA
 * The use_peft=true recipe (Multitask/model/ps-slm.py:114-117; PeftConfig r / lora_alpha / lora_dropout / target_modules at
 * Multitask/aispeech_asr_config.py:41-50).  peft 0.6.0 is not part of the reference tree; its lora.Linear.forward,
 *     result = base(x);  result += lora_B(lora_A(dropout(x))) * scaling,
 * is restated by the host (ps_slm_amd/lora.py) on the GEMM entry points above plus the ones of this section.
 * tasu_scale_bf16: dst = bf16(src * s).
 * Dropout: `rng` is int64[2] in device memory = {seed, step}; element idx of dropout stream `stream_id` is kept iff the
 * upper 32 bits of splitmix64-finalize((seed ^ step * 0x9E3779B97F4A7C15 ^ stream_id << 44) + idx * 0xD1B54A32D192ED03) are
 * >= p * 2^32; kept values are scaled by 1 / (1 - p) (fp32) and rounded to bf16 once.  Stateless: the backward calls
 * tasu_lora_dropout on the gradient with the same (stream_id, step) and gets the forward's mask.  The draws are this
 * library's own -- torch's Philox stream is not reproduced (PARITY UNPINNED for the mask; pinned for a GIVEN mask).
 * tasu_lora_dropout takes a [M, C] matrix with leading dimensions (C, ld % 8 == 0; element index m * C + c).
 * tasu_lora_dropout_norm: the same on the fp32 RMSNorm output g * (x * rstd) recomputed from the saved row scales.
 * tasu_rng_advance: step += 1 (a launch, so that a replayed hipGraph draws fresh masks).                                   */
/* Rank-sized GEMM (csrc/gemm_rank.hip): C[M, N] = A[M, K] . B[N, K]^T for N <= 64 -- the adapters' u = xd A^T and du = dy (sB),
 * and their weight gradients dB = dy^T u [out, r], dA = du^T xd [r, in] on transposed operands (K = the rows of the step).
 * 16 x 64 tiles, eight waves per workgroup split K and meet in LDS (no global partials; deterministic).  out_f32: C is fp32,
 * else bf16; transposed: C^T [N, M] is stored (ldc = its leading dimension).  K % 64 == 0, lda / ldb % 8 == 0, A / B 16-byte
 * aligned.                                                                                                                  */
int tasu_gemm_nt_rank(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, int out_f32,
                      int transposed, void* stream);
/* The same product (bf16 out, not transposed) for the 1..4 members of an adapted group in ONE launch: problem t = (A[t], lda[t],
 * B[t], ldb[t], C[t], K[t]); M, N and ldc are shared (ps_slm_amd/lora.py: the members' rank activations / their gradients). */
int tasu_gemm_nt_rank_group(int n_members, const void* const* A, const int* lda, const void* const* B, const int* ldb, void* const* C, int ldc,
                            int M, int N, const int* K, void* stream);
/* The same product with A given K-MAJOR, C[M, N] = At[K, M]^T . B[N, K]^T (fp32): a weight gradient straight from the step's
 * row-major tensors -- dB = dy^T u with At = dy [rows, out], dA^T = xd^T du with At = xd [rows, in] (torch autograd of
 * peft's lora_B(lora_A(dropout(x))), cf. Multitask/model/ps-slm.py:114-117) -- without a transposed copy of the big operand
 * (hardware transpose reads from LDS; 64 x 64 tiles).  M % 64 == 0, K % 64 == 0, ldat >= M, ldat / ldb % 8 == 0; sums in the order of
 * tasu_gemm_nt_rank up to the order of the 32 products inside one MFMA. */
int tasu_gemm_tn_rank(const void* At, int ldat, const void* B, int ldb, float* C, int ldc, int M, int N, int K, int transposed,
                      void* stream);
int tasu_scale_bf16(const void* src, void* dst, float s, int64_t n, void* stream);
/* tasu_lora_apply: y[M, N] = bf16(y + mask . bf16(s . bf16(u[M, R] W[N, R]^T))) and, with x_in / x_out (fp32, leading dimension
 * ldx), x_out = x_in + float(y): an adapter's rank-R GEMM with the accumulate into the base result fused (one read + one write of
 * y instead of the GEMM's result, a mask pass and an add pass).  Forward: u = the rank-sized activations, W = lora_B; backward:
 * u = du, W = A^T [in, R], y = the base path's input gradient and, with p > 0, mask = the dropout mask (p, rng, stream_id) the
 * forward applied to that input (element index m * N + n).  R % 64 == 0 (zero-padded rank), N % 8 == 0, ld* % 8 == 0.           */
int tasu_lora_apply(void* y, int ldy, const void* u, int ldu, const void* W, int ldw, int M, int N, int R, float s, float p,
                    const void* rng, int stream_id, const float* x_in, float* x_out, int ldx, void* stream);
/* tasu_lora_apply (R = 64, no residual output) for the 1..4 members of a group in ONE pass over y, members applied in the order
 * given: the roundings of the member-by-member launches, the same bits.  tasu_lora_dropout_norm_group: tasu_lora_dropout_norm for
 * every member of a group (each its own mask stream) from one evaluation of the norm. */
int tasu_lora_apply_group(void* y, int ldy, int n_members, const void* const* u, int ldu, const void* const* W, int ldw, const int* stream_ids,
                          int M, int N, int R, float s, float p, const void* rng, void* stream);
int tasu_lora_dropout_norm_group(const float* x, const float* w, const float* rstd, int n_members, void* const* dst, const int* stream_ids,
                                 int M, int D, float p, const void* rng, void* stream);
int tasu_lora_dropout(const void* src, int ld_src, void* dst, int ld_dst, int M, int C, float p, const void* rng, int stream_id,
                      void* stream);
int tasu_lora_dropout_norm(const float* x, const float* w, const float* rstd, void* dst, int M, int D, float p, const void* rng,
                           int stream_id, void* stream);
int tasu_rng_advance(void* rng, void* stream);
/* Pieces of the K-extended forward of the adapted Linears, y = [x | us] [W | B]^T (one GEMM = base + low-rank branch, the fused
 * epilogues of the frozen recipe kept): the producers of x write into the wider operand buffer.
 * tasu_rmsnorm_fwd_ld: tasu_rmsnorm_fwd with a leading dimension for y; tasu_gemm_gate_up_swiglu_ld: tasu_gemm_gate_up_swiglu_ws
 * with a leading dimension for act; tasu_copy_rows_bf16: dst[m, 0:C] = src[m, 0:C] (C, ld % 8 == 0).                         */
int tasu_rmsnorm_fwd_ld(const float* x, const float* w, void* y, int ldy, float* rstd, int M, int D, float eps, void* stream);
int tasu_gemm_gate_up_swiglu_ld(const void* A, int lda, const void* Wgu, int ldw, void* gu, void* act, int ld_act, int M, int I, int K,
                                void* workspace, int64_t workspace_bytes, void* stream);
int tasu_copy_rows_bf16(const void* src, int ld_src, void* dst, int ld_dst, int M, int C, void* stream);
/* tasu_lora_refresh: every working copy of every adapter in ONE launch (after a load / an optimizer step).  `pb` is the bucket's
 * bf16 image; `table` (device, int64 [n_entries][8]) lists 2-D copies out of it: {source offset in elements, destination address,
 * rows, cols, destination leading dimension, transpose (0 / 1), scale (float bits in the low 32), first tile}; destination =
 * bf16(scale * source) or its transpose.  Tiles are 64 x 64; entry e owns tiles [first_e, first_e+1); total_tiles = their sum.   */
int tasu_lora_refresh(const void* pb, const void* table, int n_entries, int total_tiles, void* stream);

/* ---------------------------------------------------------------------- cross entropy + token accuracy
 * transformers loss_utils.py:49-71 (shift, ignore_index -100, mean) + ps-slm.py:533-535 / utils/metric.py
 * fused over bf16 logits [M, ldv] (V valid columns).  shift_labels[m] = label the row must predict
 * (-100 = ignored).  Per row: row_loss[m] = logsumexp - logit[label] (0 if ignored), row_hit[m] = argmax ==
 * label.  If dlogits != NULL it receives (softmax - onehot) * (*inv_count) as bf16 (zeros for ignored rows
 * and for pad columns [V, ldv)); dlogits may alias logits.  inv_count is a DEVICE float.                  */
int tasu_ce_fwd_bwd(const void* logits, int ldv, const int32_t* shift_labels, int M, int V, float* row_loss,
                    int32_t* row_hit, int32_t* row_argmax, void* dlogits, const float* inv_count, void* stream);
/* out[0] = mean loss, out[1] = hits / count, out[2] = count, out[3] = 1/count (single block, deterministic) */
int tasu_ce_reduce(const float* row_loss, const int32_t* row_hit, const int32_t* shift_labels, int M, float* out,
                   void* stream);

/* ----------------------------------------------------------------------------------------- LayerNorm
 * fp32 LayerNorm over the first D of Dpad columns (projector.py:139 with D = 25055; SenseVoice.py:270-282):
 * y = bf16 or fp32 ((x - mean) * rstd * gamma + beta); columns [D, Dpad) of y are zero.
 * Backward for the (trainable) projector LN parameters only: dgamma[j] = sum_r dy[r,j]*xhat[r,j],
 * dbeta[j] = sum_r dy[r,j].  (The input is a frozen posterior: no dx.)                                   */
int tasu_layernorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, void* y, int ldy,
                       int y_is_f32, float* mean, float* rstd, int R, int D, float eps, void* stream);
#define TASU_LN_BWD_SPLIT 16 /* ws must hold 2 * TASU_LN_BWD_SPLIT * D floats (deterministic two-stage sum) */
int tasu_layernorm_bwd_params(const void* dy_bf16, int lddy, const float* x, int ldx, const float* mean,
                              const float* rstd, float* dgamma, float* dbeta, float* ws, int R, int D, void* stream);
/* column sums of a bf16 [R, C] matrix into fp32 (bias gradients).                                        */
int tasu_colsum_bf16(const void* x, int ld, float* out, int R, int C, void* stream);

/* ------------------------------------------------------------------ text pseudo-posterior (CPS) builder
 * ps-slm.py:337-358 / :360-409 on device: row r of out [R, V] fp32 = (1-alpha[r])*onehot(ids[r]) + alpha[r]/V,
 * ids[r] < 0 => zero row (padding); columns [V, ld) are zero.                                             */
int tasu_posterior_build(const int32_t* ids, const float* alpha, float* out, int ld, int R, int V, void* stream);

/* ------------------------------------------------------------------------------ embedding + merge
 * ps-slm.py:525 + :679-873.  The integer plan is made on the host (ps_slm_amd/merge.py):
 *   src_kind[m]: 0 = zero row (padding), 1 = token (src_idx = token id), 2 = audio (src_idx = row of proj).
 * x[m,:] (fp32 residual stream) = table[src_idx] | float(proj[src_idx]) | 0.                            */
int tasu_embed_merge_fwd(const float* table, const void* proj_bf16, const int32_t* src_kind, const int32_t* src_idx,
                         float* x, int M, int D, void* stream);
/* dproj[r,:] = bf16(dx[audio_rows[r],:]) for r < n_audio (gather of the gradient rows that hold audio).   */
int tasu_merge_bwd(const float* dx, const int32_t* audio_rows, void* dproj_bf16, int n_audio, int D, void* stream);

/* --------------------------------------------------------------------------------------------- AdamW
 * DeepSpeed FusedAdam(adam_w_mode) of Multitask/conf/ds_config.json:4-11 over one flat fp32 buffer:
 * g' = g*grad_scale; m,v update; p = p*(1-lr*wd) - lr/bc1 * m/(sqrt(v)/sqrt(bc2)+eps).  Optionally writes
 * the bf16 working copy of p.  Elementwise, so a bucket may be updated in several calls over disjoint ranges
 * (the engine updates each all-reduced chunk as it arrives); lr is passed by value per call.               */
int tasu_adamw(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1,
               float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);

/* -------------------------------------------------------------------------- SenseVoice encoder pieces
 * x[b,t,:] = x*scale + sinusoidal PE (SenseVoice.py:26-50,556-558), fp32 in/out, positions 1..T.         */
int tasu_sinusoid_pe(const float* x, float* y, int B, int T, int D, float scale, void* stream);
/* FSMN memory block (SenseVoice.py:124-140): depthwise conv k taps over time on masked v + residual, masked.
 * v: bf16 column block of the fused qkv activation (row stride ldv), w fp32 [D, k], lens int32 [B];
 * out fp32 [B*T, D] (added to the attention output by the caller through `accumulate`).                  */
int tasu_fsmn_fwd(const void* v, int ldv, const float* w, const int32_t* lens, float* out, int B, int T, int D,
                  int ksize, int accumulate, void* stream);
/* x += fsmn(v) followed by xn = LayerNorm(x) (bf16, pad columns [D, ldy) zero) in one launch where a wave owns whole rows
 * (D = 512, kernel 11: SenseVoiceSmall); bit-identical to tasu_fsmn_fwd(accumulate = 1) + tasu_layernorm_fwd, which other sizes run. */
int tasu_fsmn_ln_fwd(const void* v, int ldv, const float* w, const int32_t* lens, float* x, const float* gamma, const float* beta,
                     void* xn, int ldy, int B, int T, int D, int ksize, float eps, void* stream);
/* row softmax over V columns, fp32 or bf16 in / fp32 out, pad columns [V, ldy) zeroed (ps-slm.py:451).   */
int tasu_softmax_rows(const void* x, int x_is_bf16, int ldx, float* y, int ldy, int R, int V, void* stream);
/* PSD (ps-slm.py:237-317) on device, three launches: per-frame argmax + blank prob; per-utterance segment
 * plan (one thread per utterance; T is a few hundred); segment-mean gather into the padded output.       */
/* `post` row of (b, t) = b*bstride + t (bstride >= T lets the caller skip the 4 query frames, ps-slm.py:452). */
int tasu_psd_frame_stats(const float* post, int ldp, const int32_t* lens, int32_t* frame_id, float* frame_blank,
                         int B, int T, int bstride, int V, int blank_id, void* stream);
int tasu_psd_plan(const int32_t* frame_id, const float* frame_blank, const int32_t* lens, int32_t* seg_start,
                  int32_t* seg_len, int32_t* new_lens, int B, int T, int blank_id, float threshold, void* stream);
int tasu_psd_gather(const float* post, int ldp, const int32_t* seg_start, const int32_t* seg_len,
                    const int32_t* new_lens, float* out, int ldo, int B, int T, int bstride, int Tout, int V,
                    void* stream);
/* PSD straight from the CTC head's bf16 logits (round 4): the fp32 posterior of every frame is never materialised.
 * tasu_psd_logit_stats: per frame argmax id (-1 past the utterance's length), blank probability and the softmax statistics
 * frame_stat[bt] = (max, 1 / sum); tasu_psd_gather_softmax: out[b, j, :] = mean over the frames of kept segment j of
 * softmax(logits[frame]) evaluated from those statistics (rows j >= new_lens[b] and columns [V, ldo) zero).  logits bf16
 * [B * bstride, ld], ld % 8 == 0, 16-byte aligned; tasu_psd_plan sits between the two as before.                          */
int tasu_psd_logit_stats(const void* logits, int ld, const int32_t* lens, int32_t* frame_id, float* frame_blank, float* frame_stat,
                         int B, int T, int bstride, int V, int blank_id, void* stream);
int tasu_psd_gather_softmax(const void* logits, int ld, const float* frame_stat, const int32_t* seg_start, const int32_t* seg_len,
                            const int32_t* new_lens, float* out, int ldo, int B, int T, int bstride, int Tout, int V, void* stream);

/* ------------------------------------------------------------------------------------------- decode loop
 * ps-slm.py:660-675 -> HF GenerationMixin beam search (num_beams 4, max_new_tokens 200, greedy-beam) with a KV cache.
 * Cache layout: k/v cache [M = B*n_beams, ctx, G*128] bf16 (row = beam, position-major so that appending one token
 * is one contiguous write) plus a ROW INDEX [M, ctx] int32: index[m, i] = the cache row that physically holds position
 * i of beam m.  The prompt is stored once per utterance (row of its first beam); a beam reorder
 * (Cache.reorder_cache) permutes index rows -- 4 bytes per position for all layers -- instead of copying K/V.
 * kv_fill: copy the rotated K and V of a prefill qkv activation [B*S, (H+2G)*128] into cache row b*n_beams. */
int tasu_kv_fill(const void* qkv, void* kcache, void* vcache, int B, int S, int H, int G, int n_beams, int ctx,
                 void* stream);
/* kv_append: cache[row, pos[row]] = k|v of the single-token qkv activation [M, (H+2G)*128] (a beam appends to ITS row). */
int tasu_kv_append(const void* qkv, void* kcache, void* vcache, const int32_t* pos, int M, int H, int G, int ctx,
                   void* stream);
/* rope_append (decode step): rotate q and k of the single-token qkv activation [M, (H+2G)*128] in place (tables
 * [M, 64] from tasu_rope_table) and write the rotated k and v to cache[row, pos[row]] -- tasu_rope_fwd + kv_append. */
int tasu_rope_append(void* qkv, const float* cos_tab, const float* sin_tab, void* kcache, void* vcache, const int32_t* pos,
                     int M, int H, int G, int ctx, void* stream);
/* index[m, i] = (m / n_beams) * n_beams for i < S (shared prompt), m for i >= S. */
int tasu_kv_index_init(int32_t* index, int B, int n_beams, int S, int ctx, void* stream);
/* dst[m, :lens[m]] = src[src_row[m], :lens[m]] (src_row NULL = identity); dst != src; entries >= lens[m] untouched. */
int tasu_kv_index_reorder(const int32_t* src_index, int32_t* dst_index, const int32_t* src_row, const int32_t* lens, int M,
                          int ctx, void* stream);
/* Single-token GQA attention over the cache: keys [kstart[row], lens[row]) visible (left padding / current length),
 * key i read from cache row row_index[row, i] (row_index NULL: the row itself; all ctx entries of a row must be valid cache
 * rows -- they are read before lens is known); out [M, H*128] bf16, or the o projection's fragment-order A operand with
 * out_frag = 1 (64-row chunks).  Scores and P.V both run as 16x16x32 bf16 MFMAs (csrc/attn_decode_body.h).  ctx <= 2048 and
 * 4 * (H/G * ctx + H/G * round_up(ctx, 32) / 2 + ctx) + 64 KiB of LDS <= 160 KiB (H/G = 8: ctx <= ~1800); 1 <= lens - kstart. */
int tasu_attn_decode(const void* qkv, const void* kcache, const void* vcache, const int32_t* row_index,
                     const int32_t* kstart, const int32_t* lens, void* out, int M, int H, int G, int ctx, float scale, int out_frag,
                     void* stream);
/* log_softmax + top-k per row of bf16 logits [M, ld]: out_val[M,k] (descending log-probs), out_idx[M,k] (token ids;
 * ties: smaller id first); the n_banned ids in `banned` (device) score -inf after the softmax
 * (MinLengthLogitsProcessor).  k in {1,2,4,6,8,16}.  workspace: M * 16 * (2 + 2k) floats (column-part partials).     */
int tasu_logprob_topk(const void* logits, int ld, int M, int V, int k, const int32_t* banned, int n_banned,
                      float* out_val, int32_t* out_idx, float* workspace, int64_t workspace_floats, void* stream);
/* One generated position of the beam search behind slam_model_asr.generate (ps-slm.py:660-675 -> HF
 * GenerationMixin._beam_search, num_beams = n_beams, do_sample = False, early_stopping = False), entirely on the
 * device: candidates = per-row top-2*n_beams log-probs (tasu_logprob_topk) + running scores; selection, finished-
 * hypothesis heap (score / len^penalty via len_pow[t] = float32(t ** length_penalty)), HF's early-stop heuristic.
 * State [B, n_beams]: run_scores, fin_scores, fin_len, fin_par, fin_tok, is_fin; unsat [B]; back-pointers
 * bp_tok / bp_par [max_new, B, n_beams] (token and parent slot of every running beam per step; a finished
 * hypothesis is (fin_len, fin_par, fin_tok) + the walk up the back-pointers).  ctl[0] = positions generated,
 * ctl[1] = done (the call is then a no-op); done_host (optional, pinned host word) receives ctl[0] when done.
 * Also writes the next step's inputs: token ids, cache source rows (beam reorder), position ids (valid[b] +
 * t), cache slots S + t, lengths, and banned[0] = eos while the next position is below min_length else -1.
 * first = 1: vals / idx hold B rows (the prompt's last position); beams >= 1 do not exist yet.  B <= 256,
 * n_beams <= 5.                                                                                              */
int tasu_beam_update(const float* vals, const int32_t* idx, float* run_scores, float* fin_scores, int32_t* fin_len,
                     int32_t* fin_par, int32_t* fin_tok, int32_t* is_fin, int32_t* unsat, int32_t* bp_tok,
                     int32_t* bp_par, const float* len_pow, int32_t* ctl, int32_t* done_host, const int32_t* valid,
                     int32_t* next_ids, int32_t* next_src, int32_t* next_pos, int32_t* next_slot, int32_t* next_lens,
                     int32_t* banned, int B, int n_beams, int max_new, int eos, int min_length, int S, int first,
                     void* stream);
/* x[m,:] = table[ids[m],:] (fp32 embedding rows of the last generated tokens). */
int tasu_embed_rows(const float* table, const int32_t* ids, float* x, int M, int D, void* stream);


/* ------------------------------------------------------------------------------------------- audio front end
 * funasr WavFrontend (speech_dataset_large.py:133-146; third-party algorithm, see oracle/fbank_oracle.py): Kaldi log-mel
 * filterbank of a device waveform.  frames = 1 + (n_samples - win) / shift (snip_edges); every frame: x * scale, DC removal,
 * pre-emphasis, `window` [win], zero-pad to 512, power spectrum, `mel` [n_mels, 257] (dense, fp32), log(max(., FLT_EPSILON)).
 * out [frames, n_mels] fp32.  win <= 512. */
int tasu_fbank(const float* wave, int64_t n_samples, float scale, int win, int shift, const float* window, const float* mel,
               int n_mels, float preemph, float* out, void* stream);
/* Low-frame-rate stacking + CMVN: out[i, m*D + k] = (fb[clamp(i*lfr_n + m - (lfr_m-1)/2, 0, T-1), k] + means[m*D+k]) *
 * scales[m*D+k] for i < ceil(T / lfr_n); means may be NULL (no CMVN). */
int tasu_lfr_cmvn(const float* fb, int T, int D, int lfr_m, int lfr_n, const float* means, const float* scales, float* out,
                  void* stream);

/* Everything a generated position needs before its first decoder layer, in one launch (each piece also exists as its own
 * entry point: tasu_embed_rows, tasu_rmsnorm_fwd_frag, tasu_rope_table, tasu_kv_index_reorder; bit-identical results):
 *   x[m, :] = table[ids[m], :];  xn_frag = RMSNorm(x, norm_w) in fragment order (64-row chunks);  cos / sin [M, 64] of pos[m]
 *   (HD = 128);  and the beam reorder of the cache row index IN PLACE: index[m, :lens[m]] = index_before[src_row[m], :lens[m]],
 *   which requires src_row[m] to be a row of m's own utterance (rows (m / n_beams) * n_beams .. + n_beams - 1; HF beam search
 *   reorders within a batch item).  D % 256 == 0 (D / 256 in {1, 2, 6, 7, 14}), n_beams <= 5, ctx <= 2048. */
int tasu_decode_step_prologue(const float* table, const int32_t* ids, float* x, const float* norm_w, void* xn_frag, float eps,
                              const int32_t* pos, float* cos_tab, float* sin_tab, float theta, int32_t* index,
                              const int32_t* src_row, const int32_t* lens, int n_beams, int M, int D, int ctx, void* stream);

/* ------------------------------------------------------------------------------------------ cross-attention projector
 * EncoderProjectorCTCCA (Multitask/model/projector.py:104-126; model_config.encoder_projector = "cross-attention",
 * ps-slm.py:475-480): Q = W_q(posterior) (tasu_gemm_nt_bf16), then per head h of 8 (width d = llm_dim / 8, a multiple of 64):
 * scores = Q_h . E_h^T (tasu_gemm_nt_bf16 over the V2 rows of the LLM's embedding table, K = d), P below, z_h = P . E_h
 * (tasu_gemm_nt_bf16 against the transposed table, K = ld).  The row kernels between the two contractions:
 *   tasu_scale_softmax_rows_bf16   P[r, :V] = bf16(softmax(bf16(S[r, :V] / denom))), P[r, V:ld] = 0   (S, P bf16 [R, ld]);
 *                                  stats (optional, fp32 [R, 2]) receives the row's (max, 1 / sum) for the backward
 *   tasu_softmax_bwd_rows_bf16     dS[r, :V] = bf16(bf16(P32 o (dP - sum_c P32 o dP)) / denom), dS[r, V:ld] = 0, with P32 the fp32
 *                                  softmax output recomputed from the saved scores S and stats (what autograd saves; the bf16 P
 *                                  of the forward is only the einsum's operand)                                              */
int tasu_scale_softmax_rows_bf16(const void* s, void* p, float* stats, int R, int V, int ld, float denom, void* stream);
int tasu_softmax_bwd_rows_bf16(const void* s, const float* stats, const void* dp, void* ds, int R, int V, int ld, float denom,
                               void* stream);

/* ------------------------------------------------------------------------------------------ RCCL (one process per GPU)
 * Replaces the gradient exchange of the DeepSpeed engine (Multitask/finetune_deepspeed.py:147-149; ZeRO-2 reduce-scatter +
 * all-gather, Multitask/conf/ds_config.json:15-21) by what it amounts to for 54.5 M replicated parameters: an in-place SUM
 * all-reduce of ranges of the flat fp32 gradient bucket over RCCL / xGMI, asynchronous on the given HIP stream (the engine uses
 * a side stream chained to the wgrad kernels by events and divides by the world size inside tasu_adamw).
 * RCCL is bound at run time: the copy already mapped into the process (found with dl_iterate_phdr, whatever path the host
 * framework loaded it from), else TASU_RCCL_PATH, else the ROCm installation's.  Two copies in one process are refused:
 * tasu_comm_available returns 0 when more than one librccl is mapped or when TASU_RCCL_PATH names a file other than the mapped
 * one; tasu_comm_library writes the bound file's path (return 0) or the reason (return 2) into out[n].
 * Bootstrap: rank 0 calls tasu_comm_unique_id, the launcher's rendezvous carries the 128 bytes to every rank (the entrypoint
 * broadcasts them over the process group torch.distributed.run set up), every rank calls tasu_comm_init on ITS device
 * (hipSetDevice first).  HOST calls; a communicator belongs to the thread / device that created it.
 * tasu_allreduce_min_i32: the 1-int "every rank still has a batch" flag that replaces the reference's per-step gloo
 * monitored_barrier (Multitask/utils/deepspeed_utils.py:102-123,191). */
int tasu_comm_available(void);                                   /* 1 when RCCL could be bound */
int tasu_comm_unique_id(uint8_t* id128);
int tasu_comm_init(const uint8_t* id128, int rank, int world, void** comm);
int tasu_comm_destroy(void* comm);
int tasu_comm_library(char* out, int n);                         /* HOST string: bound file, or why binding failed */
int tasu_comm_count(void* comm, int* count);                     /* ncclCommCount: ranks RCCL itself sees in the communicator */
int tasu_allreduce_f32(void* comm, float* buf, int64_t n, void* stream);
int tasu_allreduce_min_i32(void* comm, int32_t* buf, int64_t n, void* stream);

/* HOST: kernel launches of the bf16 GEMM kernel families (gemm_nt / gemm_pipe / gemm_pp) since the library was loaded.  bench.py
 * takes the difference around one eager step: a GEMM CALL whose columns are split over two tile shapes is two launches, and
 * roofline.avg_launch_us is quoted per kernel launch, the unit of rocprofv3's per-kernel average (profiles/r06_bench_summary.md).  */
int64_t tasu_gemm_launch_count(void);

/* ------------------------------------------------------------------------------------------ fp32 arithmetic mode (decode)
 * train_config.use_fp16 = false: the reference decodes with fp32 weights and no autocast (Multitask/inference_batch.py:113-117,146:
 * `model.eval()`, no `.half()`, `model.generate(**batch)`; Multitask/model/ps-slm.py:660-675 -> HF generate on the fp32 Qwen2).
 * Every pointer below is fp32 device memory; nothing is rounded to bf16 (csrc/fp32.hip).  These replace, for that mode, the spans
 * of the bf16 entry points above: projector Linears (Multitask/model/projector.py:128-151), Qwen2DecoderLayer.forward
 * (transformers modeling_qwen2.py:41-48 RMSNorm, :91-135 RoPE, :150-172 attention, Qwen2MLP), lm_head + log_softmax + top-k of
 * GenerationMixin._beam_search.  The KV cache has the layout of tasu_kv_fill (rows = beams, position-major, fp32) and shares
 * tasu_kv_index_init / tasu_kv_index_reorder, tasu_rope_table, tasu_embed_rows, tasu_beam_update with the bf16 path.
 *
 * tasu_f32_gemm_nt: C[M, N] = [resid +] act(A[M, K] . W[N, K]^T + bias), act 0 = none, 1 = SiLU (x / (1 + exp(-x))), 2 = ReLU;
 * v_mfma_f32_16x16x4_f32, K % 32 == 0, lda / ldw % 4 == 0, 16-byte aligned operands; resid may alias C.  With a workspace,
 * outputs of fewer than 128 tiles of 64 x 64 are computed as up to 16 K-range slabs (ksplit * M * N floats) summed in ascending
 * order by a second launch -- deterministic.
 * At most 64 rows (the decode step's beam rows) against a matrix of 32 MB or more, K % 128 == 0: the weight-streaming kernel
 * (f32_stream_kernel: one 8-wave workgroup per CU walks 16-column weight tiles, each wave a K slice of 16 ks with its slice of
 * the activation rows in registers, the waves' partial tiles meet in LDS in wave order; K ranges of 128 ks as slabs).  The split
 * is a function of (N, K) only, so a row's bits do not depend on how many rows travel with it.
 * tasu_f32_gemm_stream: the same product forced onto that kernel with a given ks (1..6, K % (128 ks) == 0, K / (128 ks) <= 16
 * slabs in the workspace) -- tests and tools; TASU_ERR_ARG when the problem does not fit it.                                     */
int tasu_f32_gemm_nt(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias, const float* resid,
                     int M, int N, int K, int act, float* workspace, int64_t workspace_floats, void* stream);
int tasu_f32_gemm_stream(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias, const float* resid,
                         int M, int N, int K, int act, int ks, float* workspace, int64_t workspace_floats, void* stream);
/* Fragment-order weights for the streaming kernel: out[((t * K/16 + k16) * 64 + lane) * 4 + e] = W[16 t + (lane & 15)][16 k16 + 4 (lane >> 4) + e]
 * (rows >= N zero; out: ceil(N / 16) * 16 * K floats) -- a wave instruction of the kernel then reads 1 KiB contiguous instead of 16 rows x
 * 64 B.  Every tasu_f32_gemm_* entry point takes the copy with ldw = TASU_F32_LDW_FRAGMENT where the streaming kernel serves the
 * problem (tasu_f32_gemm_streams(M, N, K, workspace_floats) == 1), and returns TASU_ERR_ARG for it elsewhere.  Same bits as the
 * row-major matrix on that kernel.                                                                                               */
#define TASU_F32_LDW_FRAGMENT (-1)
int tasu_f32_to_fragment_order(const float* W, int ldw, float* out, int N, int K, void* stream);
int tasu_f32_gemm_streams(int M, int N, int K, int64_t workspace_floats);
/* The decode step's three GEMMs that carry the NEXT row-wise kernel in the launch that sums their K-range slabs (one launch less
 * each; the same sums in the same order as tasu_f32_gemm_nt followed by that kernel -- the same bits; problems that do not split,
 * i.e. the prompt pass, run the two kernels):
 *   tasu_f32_gemm_resid_rmsnorm   x = resid + A W^T [+ bias] (resid may alias x);  y = w * (x * rsqrt(mean(x^2) + eps))
 *   tasu_f32_gemm_swiglu          act[M, I] = silu(A Wg^T) * (A Wu^T), Wgu = [Wg; Wu]; gu [M, 2I]: scratch of the unsplit route
 *   tasu_f32_gemm_qkv_rope        qkv = A Wqkv^T + bias, rotary embedding on the q / k heads, k / v to cache[m, slot[m]] (kcache may be NULL) */
int tasu_f32_gemm_resid_rmsnorm(const float* A, int lda, const float* W, int ldw, float* x, int ldx, const float* bias, const float* resid,
                                const float* norm_w, float* y, int M, int N, int K, float eps, float* workspace, int64_t workspace_floats,
                                void* stream);
int tasu_f32_gemm_swiglu(const float* A, int lda, const float* Wgu, int ldw, float* gu, float* act, int M, int I, int K, float* workspace,
                         int64_t workspace_floats, void* stream);
int tasu_f32_gemm_qkv_rope(const float* A, int lda, const float* Wqkv, int ldw, const float* bias, float* qkv, const float* cos_tab,
                           const float* sin_tab, int M, int H, int G, int K, float* kcache, float* vcache, const int32_t* slot, int ctx,
                           float* workspace, int64_t workspace_floats, void* stream);
/* y = w * (x * rsqrt(mean(x^2) + eps)) per row of x [M, D] */
int tasu_f32_rmsnorm(const float* x, const float* w, float* y, int M, int D, float eps, void* stream);
/* q and k heads of qkv [M, (H+2G)*128] rotated in place (tables [M, 64] of tasu_rope_table; x*cos + rotate_half(x)*sin with the two
 * products rounded separately like torch eager); kcache != NULL: the rotated k and the v of row m also go to cache[m, slot[m]].  */
int tasu_f32_rope(float* qkv, const float* cos_tab, const float* sin_tab, int M, int H, int G, float* kcache, float* vcache,
                  const int32_t* slot, int ctx, int inverse, void* stream);   /* inverse = 1: the rotation's backward (sin negated) */
/* K / V of a (rotated) prefill activation [B*S, (H+2G)*128] -> cache row b * n_beams, positions 0 .. S-1 (cache [B*n_beams, ctx, G*128]) */
int tasu_f32_kv_fill(const float* qkv, float* kcache, float* vcache, int B, int S, int H, int G, int n_beams, int ctx, void* stream);
/* attention over a whole sequence, out [B*S, H*128], S <= 2048.  klen == NULL: the decoder's causal prompt pass, query s of batch
 * row b attends keys [kstart[b], s] (left padding masked; rows s < kstart[b] get zeros).  klen != NULL: bidirectional with key
 * padding (the SANM encoder, Multitask/model/SenseVoice.py:209-228): every query s < klen[b] attends keys [0, klen[b]).            */
int tasu_f32_attn_prefill(const float* qkv, const int32_t* kstart, const int32_t* klen, float* out, int B, int S, int H, int G,
                          float scale, void* stream);
/* FSMN memory block in fp32 (SenseVoice.py:124-140): out[b, t, :] += conv_k(masked v) + masked v for t < lens[b] (tasu_fsmn_fwd with
 * accumulate = 1 on an fp32 v of row stride ldv).                                                                                */
int tasu_f32_fsmn(const float* v, int ldv, const float* w, const int32_t* lens, float* out, int B, int T, int D, int ksize, void* stream);
/* single-token GQA attention over the fp32 cache: semantics of tasu_attn_decode (row_index required); out [M, H*128]; ctx <= 2048 */
int tasu_f32_attn_decode(const float* qkv, const float* kcache, const float* vcache, const int32_t* row_index, const int32_t* kstart,
                         const int32_t* lens, float* out, int M, int H, int G, int ctx, float scale, void* stream);
/* act[M, I] = silu(gu[:, :I]) * gu[:, I:] */
int tasu_f32_swiglu(const float* gu, float* act, int M, int I, void* stream);
/* tasu_embed_merge_fwd with an fp32 projector output proj [*, ldp] */
int tasu_f32_embed_merge(const float* table, const float* proj, int ldp, const int32_t* src_kind, const int32_t* src_idx, float* x,
                         int M, int D, void* stream);
/* Shifted CE of the eval-mode fp32 forward (Multitask/utils/deepspeed_utils.py:394-498 with use_fp16 = false; the loss of
 * transformers' ForCausalLMLoss, ignore_index -100): per row m with shift_labels[m] >= 0: row_loss = logsumexp(logits[m]) -
 * logits[m, label], row_hit = (argmax == label) (ties: first column); other rows 0.  row_argmax / row_lse optional.  Feed
 * tasu_ce_reduce with row_loss / row_hit.  dlogits (optional, may alias logits; inv_count = DEVICE float 1 / #labelled rows):
 * the mean loss's gradient, (softmax - onehot) * inv_count on labelled rows, 0 elsewhere and in the pad columns [V, ld).          */
int tasu_f32_ce(const float* logits, int ld, const int32_t* shift_labels, int M, int V, float* row_loss, int32_t* row_hit,
                int32_t* row_argmax, float* row_lse, float* dlogits, const float* inv_count, void* stream);

/* ------------------------------------------------------------------------------------------ fp32 training step (backward)
 * train_config.use_fp16 = false DURING TRAINING (the shipped recipe, Multitask/scripts/finetune_deespeed_sensevoice.sh:37: forward and
 * backward without autocast, Multitask/utils/deepspeed_utils.py:160,205-236).  csrc/fp32_train.hip: the backward of every non-GEMM
 * operator of the text-only step; the GEMMs are tasu_f32_gemm_nt on transposed fp32 weight copies (dgrad) and on transposed
 * activations (wgrad: tasu_f32_transpose).  All fp32, deterministic.
 *   tasu_f32_rmsnorm_bwd            dx (+)= rstd (w . dy) - x rstd^3 / D sum(w dy x)            (modeling_qwen2.py:41-48 differentiated)
 *   tasu_f32_swiglu_bwd             dgu[M, 2I] from dact[M, I] and the saved gate|up
 *   tasu_f32_silu                   out = silu(x) (dy NULL) or dy * silu'(x)
 *   tasu_f32_colsum                 out[c] = sum_r x[r, c]
 *   tasu_f32_layernorm_bwd_params   dgamma / dbeta of the projector's LayerNorm from an fp32 dy (tasu_layernorm_bwd_params)
 *   tasu_f32_transpose              dst[c, r] = src[r, c], rows [R, Rpad) zero
 *   tasu_f32_gather_rows            out[r, :] = dx[rows[r], :] (rows[r] < 0: zeros): tasu_merge_bwd in fp32
 *   tasu_f32_attn_bwd               causal GQA attention backward over the prompt from the saved (rotated) q|k|v and dO [B*S, H*128]:
 *                                   dqkv [B*S, (H+2G)*128] in the rotated space (tasu_f32_rope(inverse = 1) takes it back);
 *                                   lse_ws / delta_ws: B * H * S floats each; S <= 2048; probabilities are recomputed, no [S, S] buffers */
int tasu_f32_rmsnorm_bwd(const float* dy, const float* x, const float* w, float* dx, int M, int D, float eps, int accumulate, void* stream);
int tasu_f32_swiglu_bwd(const float* dact, const float* gu, float* dgu, int M, int I, void* stream);
int tasu_f32_silu(const float* x, const float* dy, float* out, int64_t n, void* stream);
int tasu_f32_colsum(const float* x, int ld, float* out, int R, int C, void* stream);
int tasu_f32_layernorm_bwd_params(const float* dy, int lddy, const float* x, int ldx, const float* mean, const float* rstd, float* dgamma,
                                  float* dbeta, int R, int D, void* stream);
int tasu_f32_transpose(const float* src, int lds, float* dst, int ldd, int R, int C, int Rpad, void* stream);
int tasu_f32_gather_rows(const float* dx, const int32_t* rows, float* out, int n, int D, void* stream);
int tasu_f32_attn_bwd(const float* qkv, const float* dout, const int32_t* kstart, float* dqkv, float* lse_ws, float* delta_ws, int B, int S,
                      int H, int G, float scale, void* stream);
/* tasu_logprob_topk on fp32 logits: out_val = (x - max) - log(sum exp(x - max)) of the k best selectable columns (value descending,
 * column ascending), k <= 16; fewer than k selectable columns: (-inf, 0x7fffffff).  workspace (M * 16 * (2 + 2 k) floats, may be NULL):
 * with it the row is split over 16 workgroups + a merge launch (a decode step's 64 rows are too few workgroups for one per row).        */
int tasu_f32_logprob_topk(const float* logits, int ld, int M, int V, int k, const int32_t* banned, int n_banned, float* out_val,
                          int32_t* out_idx, float* workspace, int64_t workspace_floats, void* stream);

/* ------------------------------------------------------------------------------------------ FLAC (host)
 * The reference reads ``.flac`` entries with torchaudio.load (speech_dataset_large.py:123-127: [C, T] float
 * in [-1, 1), channel mean).  HOST functions (no device work, no stream): tasu_flac_info parses STREAMINFO
 * (rate_channels_bps[3], total samples per channel); tasu_flac_decode decodes the whole stream into
 * mono_out[capacity] = mean over channels of sample / 2^(bps-1), verifying every frame's CRC-8 / CRC-16 and the
 * STREAMINFO MD5 of the decoded PCM.  Returns 0, 1 (bad argument / capacity too small) or 3 (corrupt or
 * unsupported stream).  Restated from the published format; parity unpinned against libFLAC (absent).      */
int tasu_flac_info(const uint8_t* data, int64_t n_bytes, int32_t* rate_channels_bps, int64_t* total_samples);
int tasu_flac_decode(const uint8_t* data, int64_t n_bytes, float* mono_out, int64_t capacity, int64_t* n_decoded);

#ifdef __cplusplus
}
#endif
#endif /* TASU_HIP_H_ */
