// Elementwise pieces of the LoRA recipe (use_peft=true: Multitask/model/ps-slm.py:114-117, PeftConfig at
// Multitask/aispeech_asr_config.py:41-50; peft 0.6.0 lora.Linear.forward, absent from the reference tree:
//     result = base(x);  result += lora_B(lora_A(dropout(x))) * scaling
// under torch.autocast(bfloat16): base(x), lora_A(.), lora_B(.) are bf16 GEMM results, `* scaling` and `+=` round to bf16).
// The GEMMs themselves run on the existing NT kernels (gemm.hip); this file holds what sits between them.  All HBM-bound.
#include "common.h"
#include "../../include/tasu_hip.h"

namespace {

inline int grid_for(int64_t nvec) {
  int64_t b = (nvec + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

// y = bf16(y + bf16(t * s));  RESID: x_out = x_in + float(y_new)  (the decoder's residual add: fp32 stream + bf16 branch)
template <bool RESID>
__global__ __launch_bounds__(256) void lora_add_kernel(bf16* __restrict__ y, const bf16* __restrict__ t, float s,
                                                       const float* __restrict__ x_in, float* __restrict__ x_out, int64_t nvec) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const bf16x8 a = *(const bf16x8*)(y + i * 8), b = *(const bf16x8*)(t + i * 8);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = (float)(bf16)((float)b[j] * s);
      o[j] = (bf16)((float)a[j] + d);
    }
    *(bf16x8*)(y + i * 8) = o;
    if constexpr (RESID) {
      const f32x4 r0 = *(const f32x4*)(x_in + i * 8), r1 = *(const f32x4*)(x_in + i * 8 + 4);
      f32x4 q0, q1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        q0[j] = r0[j] + (float)o[j];
        q1[j] = r1[j] + (float)o[4 + j];
      }
      *(f32x4*)(x_out + i * 8) = q0;
      *(f32x4*)(x_out + i * 8 + 4) = q1;
    }
  }
}

__global__ __launch_bounds__(256) void scale_bf16_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, float s, int64_t nvec) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const bf16x8 a = *(const bf16x8*)(src + i * 8);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)a[j] * s);
    *(bf16x8*)(dst + i * 8) = o;
  }
}

// Counter-based mask: element `idx` of dropout stream `sid` at optimizer micro-step `step` is kept iff the upper 32 bits of
// splitmix64(seed ^ step * GOLD ^ sid << 44, + idx * ODD) are >= p * 2^32.  Stateless, so backward regenerates the mask the
// forward used, and graph replays draw new masks because {seed, step} live in device memory (tasu_rng_advance bumps step).
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t mask_key(const int64_t* rng, int sid) {
  return (uint64_t)rng[0] ^ ((uint64_t)rng[1] * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(uint32_t)sid << 44);
}
__device__ __forceinline__ float keep_scale(uint64_t key, int64_t idx, uint32_t thr, float inv) {
  const uint64_t z = mix64(key + (uint64_t)idx * 0xD1B54A32D192ED03ull);
  return (uint32_t)(z >> 32) >= thr ? inv : 0.f;
}

__global__ __launch_bounds__(256) void dropout_bf16_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, int64_t nvec,
                                                           uint32_t thr, float inv, const int64_t* __restrict__ rng, int sid) {
  const uint64_t key = mask_key(rng, sid);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const bf16x8 a = *(const bf16x8*)(src + i * 8);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)a[j] * keep_scale(key, i * 8 + j, thr, inv));
    *(bf16x8*)(dst + i * 8) = o;
  }
}

// dropout applied to the fp32 RMSNorm output (recomputed from the saved row scale: g * (x * rstd), norm.hip's arithmetic),
// rounded to bf16 once -- the order the reference has (the norm's output is fp32; lora_A's autocast cast comes after the dropout)
__global__ __launch_bounds__(256) void dropout_norm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ rstd, bf16* __restrict__ dst, int M, int D,
                                                           uint32_t thr, float inv, const int64_t* __restrict__ rng, int sid) {
  const uint64_t key = mask_key(rng, sid);
  const int64_t nvec = (int64_t)M * D / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const int64_t e = i * 4;
    const int row = (int)(e / D), c = (int)(e - (int64_t)row * D);
    const f32x4 v = *(const f32x4*)(x + e), g = *(const f32x4*)(w + c);
    const float r = rstd[row];
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (bf16)((g[j] * (v[j] * r)) * keep_scale(key, e + j, thr, inv));
    *(bf16x4*)(dst + e) = o;
  }
}

__global__ void rng_advance_kernel(int64_t* rng) {
  if (threadIdx.x == 0 && blockIdx.x == 0) rng[1] += 1;
}

inline bool drop_args(float p, uint32_t* thr, float* inv) {
  if (!(p >= 0.f) || !(p < 1.f)) return false;
  const double t = (double)p * 4294967296.0;
  *thr = t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
  *inv = 1.0f / (1.0f - p);
  return true;
}

}  // namespace

extern "C" int tasu_lora_add(void* y, const void* t, float s, const float* x_in, float* x_out, int64_t n, void* stream) {
  if (!y || !t || n <= 0 || n % 8 || ((x_in == nullptr) != (x_out == nullptr))) return TASU_ERR_ARG;
  if (x_in)
    TASU_LAUNCH(lora_add_kernel<true>, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (bf16*)y, (const bf16*)t, s, x_in, x_out, n / 8);
  else
    TASU_LAUNCH(lora_add_kernel<false>, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (bf16*)y, (const bf16*)t, s, x_in, x_out, n / 8);
  return TASU_OK;
}

extern "C" int tasu_scale_bf16(const void* src, void* dst, float s, int64_t n, void* stream) {
  if (!src || !dst || n <= 0 || n % 8) return TASU_ERR_ARG;
  TASU_LAUNCH(scale_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, (bf16*)dst, s, n / 8);
  return TASU_OK;
}

extern "C" int tasu_lora_dropout(const void* src, void* dst, int64_t n, float p, const void* rng, int stream_id, void* stream) {
  uint32_t thr;
  float inv;
  if (!src || !dst || !rng || n <= 0 || n % 8 || stream_id < 0 || !drop_args(p, &thr, &inv)) return TASU_ERR_ARG;
  TASU_LAUNCH(dropout_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, (bf16*)dst, n / 8, thr, inv,
              (const int64_t*)rng, stream_id);
  return TASU_OK;
}

extern "C" int tasu_lora_dropout_norm(const float* x, const float* w, const float* rstd, void* dst, int M, int D, float p,
                                      const void* rng, int stream_id, void* stream) {
  uint32_t thr;
  float inv;
  if (!x || !w || !rstd || !dst || !rng || M <= 0 || D <= 0 || D % 4 || stream_id < 0 || !drop_args(p, &thr, &inv)) return TASU_ERR_ARG;
  TASU_LAUNCH(dropout_norm_kernel, dim3(grid_for((int64_t)M * D / 4)), dim3(256), 0, (hipStream_t)stream, x, w, rstd, (bf16*)dst, M, D, thr,
              inv, (const int64_t*)rng, stream_id);
  return TASU_OK;
}

extern "C" int tasu_rng_advance(void* rng, void* stream) {
  if (!rng) return TASU_ERR_ARG;
  TASU_LAUNCH(rng_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (int64_t*)rng);
  return TASU_OK;
}
