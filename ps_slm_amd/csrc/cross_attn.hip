// Row kernels of the cross-attention projector (EncoderProjectorCTCCA, Multitask/model/projector.py:104-126; selected by
// model_config.encoder_projector = "cross-attention", called at Multitask/model/ps-slm.py:475-480): every posterior row
// attends, per head, over ALL rows of the LLM's embedding table (V2 = 151,936 keys = values).  The two contractions of a head
// run on the NT GEMM kernels (scores = Q_h . E_h^T, K = head width; z = P . E_h, K = V2); in between, per (row, head):
//   P  = bf16( softmax_fp32( bf16( bf16(scores) / sqrt(d) ) ) )         -- the autocast rounding points of :119-123
//   dS = bf16( bf16( P32 o (dP - sum(P32 o dP)) ) / sqrt(d) )            -- softmax backward on the FP32 probabilities (autograd
//        saves the fp32 softmax output; only the einsum operand is cast to bf16), then the division's backward.  The backward
//        recomputes P32 from the saved bf16 scores and the row's (max, 1 / sum) instead of reading a bf16-rounded P.
// HBM-bound: a row is V2 * 2 B = 300 KB, read three times forward (max, sum, write) out of L2 after the first pass.
#include "common.h"
#include "../../include/tasu_hip.h"

namespace {

// one 256-thread block per row; columns [V, ld) of the outputs are zeroed (the following GEMM contracts over ld)
__global__ __launch_bounds__(256) void scale_softmax_rows_kernel(const bf16* __restrict__ s, bf16* __restrict__ p, int V, int ld,
                                                                  float denom, float* __restrict__ stats) {
  __shared__ float red[4];
  const bf16* sr = s + (size_t)blockIdx.x * ld;
  bf16* pr = p + (size_t)blockIdx.x * ld;
  const int V8 = (ld % 8 == 0 && ((uintptr_t)s & 15) == 0 && ((uintptr_t)p & 15) == 0) ? (V & ~7) : 0;
  float m = -__builtin_inff();
  for (int c = threadIdx.x * 8; c < V8; c += 2048) {
    const bf16x8 v = *(const bf16x8*)(sr + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, bf16_round((float)v[j] / denom));
  }
  for (int c = V8 + threadIdx.x; c < V; c += 256) m = fmaxf(m, bf16_round((float)sr[c] / denom));
  m = block_max<4>(m, red);
  float sum = 0.f;
  for (int c = threadIdx.x * 8; c < V8; c += 2048) {
    const bf16x8 v = *(const bf16x8*)(sr + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += expf(bf16_round((float)v[j] / denom) - m);
  }
  for (int c = V8 + threadIdx.x; c < V; c += 256) sum += expf(bf16_round((float)sr[c] / denom) - m);
  sum = block_sum<4>(sum, red);
  const float inv = 1.f / sum;
  if (stats && threadIdx.x == 0) stats[2 * blockIdx.x] = m, stats[2 * blockIdx.x + 1] = inv;
  for (int c = threadIdx.x * 8; c < V8; c += 2048) {
    const bf16x8 v = *(const bf16x8*)(sr + c);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)(expf(bf16_round((float)v[j] / denom) - m) * inv);
    *(bf16x8*)(pr + c) = o;
  }
  for (int c = V8 + threadIdx.x; c < ld; c += 256) pr[c] = c < V ? (bf16)(expf(bf16_round((float)sr[c] / denom) - m) * inv) : (bf16)0.f;
}

__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const bf16* __restrict__ sc, const float* __restrict__ stats,
                                                                const bf16* __restrict__ dp, bf16* __restrict__ ds, int V, int ld,
                                                                float denom) {
  __shared__ float red[4];
  const bf16* sr = sc + (size_t)blockIdx.x * ld;
  const bf16* dr = dp + (size_t)blockIdx.x * ld;
  bf16* or_ = ds + (size_t)blockIdx.x * ld;
  const float m = stats[2 * blockIdx.x], inv = stats[2 * blockIdx.x + 1];
  auto prob = [&](int c) { return expf(bf16_round((float)sr[c] / denom) - m) * inv; };      // the forward's fp32 softmax output
  float dot = 0.f;
  for (int c = threadIdx.x; c < V; c += 256) dot += prob(c) * (float)dr[c];
  dot = block_sum<4>(dot, red);
  // the softmax's input is a bf16 tensor: its gradient is rounded to bf16 before the division's backward
  for (int c = threadIdx.x; c < ld; c += 256)
    or_[c] = c < V ? (bf16)(bf16_round(prob(c) * ((float)dr[c] - dot)) / denom) : (bf16)0.f;
}

}  // namespace

extern "C" int tasu_scale_softmax_rows_bf16(const void* s, void* p, float* stats, int R, int V, int ld, float denom, void* stream) {
  if (!s || !p || R <= 0 || V <= 0 || ld < V || !(denom > 0.f)) return TASU_ERR_ARG;
  TASU_LAUNCH(scale_softmax_rows_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, (const bf16*)s, (bf16*)p, V, ld, denom, stats);
  return TASU_OK;
}

extern "C" int tasu_softmax_bwd_rows_bf16(const void* s, const float* stats, const void* dp, void* ds, int R, int V, int ld, float denom,
                                          void* stream) {
  if (!s || !stats || !dp || !ds || R <= 0 || V <= 0 || ld < V || !(denom > 0.f)) return TASU_ERR_ARG;
  TASU_LAUNCH(softmax_bwd_rows_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, (const bf16*)s, stats, (const bf16*)dp, (bf16*)ds, V,
              ld, denom);
  return TASU_OK;
}
