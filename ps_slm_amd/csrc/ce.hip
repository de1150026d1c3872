// Fused shifted cross-entropy + argmax + dlogits over bf16 logits rows.
// Reference: transformers loss/loss_utils.py:49-71 (fp32 upcast, shift, ignore_index=-100, mean) and the token
// accuracy of Multitask/model/ps-slm.py:533-535 + Multitask/utils/metric.py:3-20.
// HBM-bound: one 256-thread block per row; pass 1 = online (max, sum-exp, first-argmax) over 16-byte loads,
// pass 2 (optional) rewrites the row as (softmax - onehot) / count in bf16 (the row is L2-resident by then).
#include "common.h"
#include "../../include/tasu_hip.h"

namespace {

struct Stat {
  float m, s;
  float best;
  int arg;
};
__device__ __forceinline__ Stat combine(Stat a, Stat b) {
  Stat o;
  o.m = fmaxf(a.m, b.m);
  const float ea = a.m == o.m ? 1.f : __expf(a.m - o.m);
  const float eb = b.m == o.m ? 1.f : __expf(b.m - o.m);
  o.s = a.s * ea + b.s * eb;
  if (a.best > b.best || (a.best == b.best && a.arg < b.arg)) {
    o.best = a.best;
    o.arg = a.arg;
  } else {
    o.best = b.best;
    o.arg = b.arg;
  }
  return o;
}

// logits and dlogits may be the SAME buffer (include/tasu_hip.h; the training step overwrites the logits in place), so
// neither is __restrict__ and every read of a row element that pass 2 may overwrite happens before the block barrier.
__global__ __launch_bounds__(256) void ce_kernel(const bf16* logits, int ldv,
                                                 const int32_t* __restrict__ labels, int V, float* __restrict__ row_loss,
                                                 int32_t* __restrict__ row_hit, int32_t* __restrict__ row_argmax,
                                                 bf16* dlogits, const float* __restrict__ inv_count) {
  __shared__ Stat red[4];
  const int row = blockIdx.x;
  const int label = labels[row];
  const bf16* lr = logits + (size_t)row * ldv;
  const int nv = ldv / 8;
  const bool ignored = label < 0;
  if (ignored && !row_argmax) {
    // nothing to compute for the loss; still must define outputs
    if (threadIdx.x == 0) {
      row_loss[row] = 0.f;
      row_hit[row] = 0;
    }
    if (dlogits) {
      bf16x8 z;
#pragma unroll
      for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
      for (int v = threadIdx.x; v < nv; v += 256) *(bf16x8*)(dlogits + (size_t)row * ldv + v * 8) = z;
    }
    return;
  }
  Stat st;
  st.m = -__builtin_inff();
  st.s = 0.f;
  st.best = -__builtin_inff();
  st.arg = 0x7fffffff;
  for (int v = threadIdx.x; v < nv; v += 256) {
    const bf16x8 x = *(const bf16x8*)(lr + v * 8);
    float f[8];
    float cm = -__builtin_inff();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f[j] = (v * 8 + j < V) ? (float)x[j] : -__builtin_inff();
      cm = fmaxf(cm, f[j]);
    }
    if (cm == -__builtin_inff()) continue;  // chunk entirely in the pad columns [V, ldv)
    if (cm > st.m) {
      st.s *= __expf(st.m - cm);
      st.m = cm;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      st.s += __expf(f[j] - st.m);
      if (f[j] > st.best) {
        st.best = f[j];
        st.arg = v * 8 + j;
      }
    }
  }
  // wave then block reduction
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Stat other;
    other.m = __shfl_xor(st.m, o, 64);
    other.s = __shfl_xor(st.s, o, 64);
    other.best = __shfl_xor(st.best, o, 64);
    other.arg = __shfl_xor(st.arg, o, 64);
    st = combine(st, other);
  }
  // the label's logit is read BEFORE the barrier: with dlogits aliasing logits, other waves start overwriting the row
  // (pass 2) as soon as they have passed it
  const float label_logit = (threadIdx.x == 0 && !ignored) ? (float)lr[label] : 0.f;
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = st;
  __syncthreads();
  st = combine(combine(red[0], red[1]), combine(red[2], red[3]));
  const float lse = st.m + __logf(st.s);
  if (threadIdx.x == 0) {
    if (row_argmax) row_argmax[row] = st.arg;
    if (ignored) {
      row_loss[row] = 0.f;
      row_hit[row] = 0;
    } else {
      row_loss[row] = lse - label_logit;
      row_hit[row] = st.arg == label ? 1 : 0;
    }
  }
  if (dlogits) {
    bf16* dr = dlogits + (size_t)row * ldv;
    const float ic = ignored ? 0.f : *inv_count;
    for (int v = threadIdx.x; v < nv; v += 256) {
      const bf16x8 x = *(const bf16x8*)(lr + v * 8);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = v * 8 + j;
        float g = 0.f;
        if (c < V && !ignored) {
          g = __expf((float)x[j] - lse);
          if (c == label) g -= 1.f;
          g *= ic;
        }
        o[j] = (bf16)g;
      }
      *(bf16x8*)(dr + v * 8) = o;
    }
  }
}

// Round 5: the same result with ONE read of the row.  The row's 16-byte chunks live in registers between the statistics and the
// gradient pass (1024 threads per row, at most RI chunks each: 76 registers at V = 151,936), so the logits are read once and
// the gradient written once -- the two-pass kernel's second read of a 304-KB row came from beyond L2 (2048 rows of them are in
// flight), 1.87 GB per call instead of 1.24.  Exact statistics instead of the online rescale: block max (+ first argmax), then the
// block's sum of exp(x - max).  Rows wider than RI x 8192 columns, and calls without a gradient, stay on ce_kernel.
constexpr int RT = 1024, RI = 20;
__global__ __launch_bounds__(RT) void ce_reg_kernel(const bf16* logits, int ldv, const int32_t* __restrict__ labels, int V,
                                                   float* __restrict__ row_loss, int32_t* __restrict__ row_hit,
                                                   int32_t* __restrict__ row_argmax, bf16* dlogits, const float* __restrict__ inv_count) {
  __shared__ float red_f[RT / 64];
  __shared__ int red_i[RT / 64];
  __shared__ float s_label_logit;
  const int row = blockIdx.x;
  const int label = labels[row];
  const bf16* lr = logits + (size_t)row * ldv;
  bf16* dr = dlogits + (size_t)row * ldv;
  const int nv = ldv / 8;
  if (label < 0) {                                       // ignored position: zero gradient, nothing to sum
    if (threadIdx.x == 0) {
      row_loss[row] = 0.f;
      row_hit[row] = 0;
      if (row_argmax) row_argmax[row] = 0;
    }
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
    for (int v = threadIdx.x; v < nv; v += RT) *(bf16x8*)(dr + v * 8) = z;
    return;
  }
  bf16x8 x[RI];
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int v = threadIdx.x + i * RT;
    if (v < nv) x[i] = *(const bf16x8*)(lr + v * 8);
  }
  // maximum and its first position; the label's logit out of the register copy (dlogits may alias logits: no second look at memory)
  float best = -__builtin_inff();
  int arg = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int v = threadIdx.x + i * RT;
    if (v < nv) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (float)x[i][j];
        if (v * 8 + j < V && f > best) {
          best = f;
          arg = v * 8 + j;
        }
        if (v * 8 + j == label) s_label_logit = f;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oa = __shfl_xor(arg, o, 64);
    if (ob > best || (ob == best && oa < arg)) best = ob, arg = oa;
  }
  if ((threadIdx.x & 63) == 0) red_f[threadIdx.x >> 6] = best, red_i[threadIdx.x >> 6] = arg;
  __syncthreads();
  best = red_f[0], arg = red_i[0];
#pragma unroll
  for (int w = 1; w < RT / 64; ++w)
    if (red_f[w] > best || (red_f[w] == best && red_i[w] < arg)) best = red_f[w], arg = red_i[w];
  __syncthreads();
  // sum of exp(x - max): per thread in chunk order, waves by shuffles, the block in wave order (deterministic)
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int v = threadIdx.x + i * RT;
    if (v < nv) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (v * 8 + j < V) sum += __expf((float)x[i][j] - best);
    }
  }
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) red_f[threadIdx.x >> 6] = sum;
  __syncthreads();
  sum = 0.f;
#pragma unroll
  for (int w = 0; w < RT / 64; ++w) sum += red_f[w];
  const float lse = best + __logf(sum);
  if (threadIdx.x == 0) {
    if (row_argmax) row_argmax[row] = arg;
    row_loss[row] = lse - s_label_logit;                  // (written before the first barrier above)
    row_hit[row] = arg == label ? 1 : 0;
  }
  const float ic = *inv_count;
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int v = threadIdx.x + i * RT;
    if (v < nv) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = v * 8 + j;
        float g = 0.f;
        if (c < V) {
          g = __expf((float)x[i][j] - lse);
          if (c == label) g -= 1.f;
          g *= ic;
        }
        o[j] = (bf16)g;
      }
      *(bf16x8*)(dr + v * 8) = o;
    }
  }
}

// single block: loss = sum(row_loss)/count, acc = hits/count (fixed order => deterministic)
__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* __restrict__ row_loss, const int32_t* __restrict__ row_hit,
                                                        const int32_t* __restrict__ labels, int M, float* __restrict__ out) {
  __shared__ float red[4];
  float l = 0.f, h = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < M; i += 256) {
    l += row_loss[i];
    h += (float)row_hit[i];
    c += labels[i] >= 0 ? 1.f : 0.f;
  }
  l = block_sum<4>(l, red);
  h = block_sum<4>(h, red);
  c = block_sum<4>(c, red);
  if (threadIdx.x == 0) {
    out[0] = l / c;
    out[1] = h / c;
    out[2] = c;
    out[3] = 1.f / c;
  }
}

}  // namespace

extern "C" int tasu_ce_fwd_bwd(const void* logits, int ldv, const int32_t* shift_labels, int M, int V, float* row_loss,
                               int32_t* row_hit, int32_t* row_argmax, void* dlogits, const float* inv_count,
                               void* stream) {
  if (!logits || !shift_labels || !row_loss || !row_hit || M <= 0 || V <= 0 || ldv < V || ldv % 8) return TASU_ERR_ARG;
  if (dlogits && !inv_count) return TASU_ERR_ARG;
  if (dlogits && !row_argmax && ldv / 8 <= RT * RI && V >= 8192) {      // the training step's call: the row stays in registers between the passes
    TASU_LAUNCH(ce_reg_kernel, dim3(M), dim3(RT), 0, (hipStream_t)stream, (const bf16*)logits, ldv, shift_labels, V, row_loss, row_hit,
                row_argmax, (bf16*)dlogits, inv_count);
    return TASU_OK;
  }
  TASU_LAUNCH(ce_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (const bf16*)logits, ldv, shift_labels, V,
                     row_loss, row_hit, row_argmax, (bf16*)dlogits, inv_count);
  return TASU_OK;
}

extern "C" int tasu_ce_reduce(const float* row_loss, const int32_t* row_hit, const int32_t* shift_labels, int M, float* out,
                              void* stream) {
  if (!row_loss || !row_hit || !shift_labels || !out || M <= 0) return TASU_ERR_ARG;
  TASU_LAUNCH(ce_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, row_loss, row_hit, shift_labels, M, out);
  return TASU_OK;
}
