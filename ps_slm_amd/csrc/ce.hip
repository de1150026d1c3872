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
