// Text pseudo-posterior builder (Multitask/model/ps-slm.py:337-358, :360-409), embedding gather + audio/text
// merge (ps-slm.py:525, :679-873) and its backward, fused AdamW (DeepSpeed FusedAdam per
// Multitask/conf/ds_config.json:4-11).  All HBM-bound streaming kernels with 16-byte accesses.
#include "common.h"
#include "../../include/tasu_hip.h"

namespace {

// one block per row; row = (1-a)*onehot(id) + a/V ; id < 0 -> zeros.  Columns [V, ld) are zeroed too.
__global__ __launch_bounds__(256) void posterior_kernel(const int32_t* __restrict__ ids, const float* __restrict__ alpha,
                                                        float* __restrict__ out, int ld, int V) {
  const int r = blockIdx.x;
  const int id = ids[r];
  const float a = alpha ? alpha[r] : 0.f;
  const float base = id >= 0 ? a / (float)V : 0.f;
  const float peak = (1.f - a) + base;
  float* o = out + (size_t)r * ld;
  for (int c = threadIdx.x; c < ld; c += 256) o[c] = c < V ? (c == id ? peak : base) : 0.f;
}

// x[m,:] = table[idx] (kind 1) | float(proj[idx]) (kind 2) | 0 (kind 0).  One wave per row, D % 4 == 0.
__global__ __launch_bounds__(256) void embed_merge_kernel(const float* __restrict__ table, const bf16* __restrict__ proj,
                                                          const int32_t* __restrict__ kind, const int32_t* __restrict__ idx,
                                                          float* __restrict__ x, int M, int D) {
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (m >= M) return;
  const int k = kind[m];
  const size_t src = (size_t)idx[m] * D;
  float* xr = x + (size_t)m * D;
  for (int c = lane * 4; c < D; c += 256) {
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (k == 1)
      v = *(const f32x4*)(table + src + c);
    else if (k == 2)
      v = __builtin_convertvector(*(const bf16x4*)(proj + src + c), f32x4);
    *(f32x4*)(xr + c) = v;
  }
}

__global__ __launch_bounds__(256) void merge_bwd_kernel(const float* __restrict__ dx, const int32_t* __restrict__ rows,
                                                        bf16* __restrict__ dproj, int n, int D) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= n) return;
  const int m = rows[r];
  bf16* o = dproj + (size_t)r * D;
  for (int c = lane * 4; c < D; c += 256) {
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (m >= 0) v = *(const f32x4*)(dx + (size_t)m * D + c);
    *(bf16x4*)(o + c) = __builtin_convertvector(v, bf16x4);
  }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16* __restrict__ pb, int64_t n,
                                                    float lr, float b1, float b2, float eps, float wd,
                                                    float inv_bc1, float inv_sqrt_bc2, float gscale) {
  const int64_t nv = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 pp = *(const f32x4*)(p + i * 4);
    const f32x4 gg = *(const f32x4*)(g + i * 4);
    f32x4 mm = *(const f32x4*)(m + i * 4);
    f32x4 vv = *(const f32x4*)(v + i * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gr = gg[j] * gscale;
      mm[j] = b1 * mm[j] + (1.f - b1) * gr;
      vv[j] = b2 * vv[j] + (1.f - b2) * gr * gr;
      const float denom = sqrtf(vv[j]) * inv_sqrt_bc2 + eps;
      pp[j] = pp[j] * (1.f - lr * wd) - (lr * inv_bc1) * (mm[j] / denom);
    }
    *(f32x4*)(p + i * 4) = pp;
    *(f32x4*)(m + i * 4) = mm;
    *(f32x4*)(v + i * 4) = vv;
    if (pb) *(bf16x4*)(pb + i * 4) = __builtin_convertvector(pp, bf16x4);
  }
}

}  // namespace

extern "C" int tasu_abi_version(void) { return TASU_ABI_VERSION; }

extern "C" int tasu_posterior_build(const int32_t* ids, const float* alpha, float* out, int ld, int R, int V,
                                    void* stream) {
  if (!ids || !out || R <= 0 || V <= 0 || ld < V) return TASU_ERR_ARG;
  TASU_LAUNCH(posterior_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, ids, alpha, out, ld, V);
  return TASU_OK;
}

extern "C" int tasu_embed_merge_fwd(const float* table, const void* proj, const int32_t* src_kind, const int32_t* src_idx,
                                    float* x, int M, int D, void* stream) {
  if (!table || !proj || !src_kind || !src_idx || !x || M <= 0 || D <= 0 || D % 4) return TASU_ERR_ARG;
  TASU_LAUNCH(embed_merge_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, table, (const bf16*)proj,
                     src_kind, src_idx, x, M, D);
  return TASU_OK;
}

extern "C" int tasu_merge_bwd(const float* dx, const int32_t* audio_rows, void* dproj, int n_audio, int D, void* stream) {
  if (!dx || !audio_rows || !dproj || n_audio <= 0 || D <= 0 || D % 4) return TASU_ERR_ARG;
  TASU_LAUNCH(merge_bwd_kernel, dim3((n_audio + 3) / 4), dim3(256), 0, (hipStream_t)stream, dx, audio_rows,
                     (bf16*)dproj, n_audio, D);
  return TASU_OK;
}

extern "C" int tasu_adamw(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr,
                          float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                          void* stream) {
  if (!p || !g || !m || !v || n <= 0 || n % 4 || step < 1) return TASU_ERR_ARG;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  TASU_LAUNCH(adamw_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16*)p_bf16, n, lr,
                     beta1, beta2, eps, weight_decay, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), grad_scale);
  return TASU_OK;
}
