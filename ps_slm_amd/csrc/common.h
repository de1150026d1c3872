// Shared device helpers for the TASU gfx950 kernels.  gfx950 only: wave = 64 lanes, MFMA 16x16x32 bf16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// Tuning / A-B switches of the dispatchers (TASU_GEMM_*, TASU_SKINNY_*, TASU_ATTN_QW2_FROM) exist in the LAB build only
// (make -C ps_slm_amd/csrc lab -> ps_slm_amd/libtasu_hip_lab.so, loaded with TASU_LIB_PATH): the shipped library always takes
// the measured-fastest path and reads no TASU_* variable of its own except TASU_RCCL_PATH (comm.hip).
#include <stdlib.h>
inline const char* tasu_lab_env(const char* name) {
#ifdef TASU_LAB
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

#define TASU_OK 0
#define TASU_ERR_ARG 1
#define TASU_ERR_LAUNCH 2

// hipGetLastError() is process-wide and sticky across unrelated runtime calls (e.g. a pending hipEventQuery of
// the host framework leaves hipErrorNotReady behind), so clear it before the launch and read it right after.
#define TASU_LAUNCH(kernel, grid, block, shmem, stream, ...)                      \
  do {                                                                            \
    (void)hipGetLastError();                                                      \
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);          \
    if (hipGetLastError() != hipSuccess) return TASU_ERR_LAUNCH;                  \
  } while (0)

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  // D[row = (lane>>4)*4 + r][col = lane&15] += sum_k A[row'=lane&15 -> rows of D][k] * B[k][col]
  // lane l supplies A[l&15][8*(l>>4)+j] and B[8*(l>>4)+j][l&15], j = 0..7.
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x = NW*64 threads; red must hold NW floats. All threads get the result.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) t += red[i];
  return t;
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) t = fmaxf(t, red[i]);
  return t;
}

__device__ __forceinline__ float bf16_round(float x) { return (float)(bf16)x; }
__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }
// rotate-half RoPE of one pair (x1 = x[i], x2 = x[i + 64]): tasu_rope_fwd and the tasu_gemm_qkv_rope epilogue share this text
__device__ __forceinline__ void rope_pair_f(float x1, float x2, float c, float s, float& y1, float& y2) {
  // explicit FMAs: -ffp-contract=fast may otherwise fuse either product of a * b - c * d, differently at different call sites
  const float t1 = x2 * s, t2 = x1 * s;
  y1 = __builtin_fmaf(x1, c, -t1);
  y2 = __builtin_fmaf(x2, c, t2);
}
// SwiGLU backward of one element (tasu_swiglu_bwd and the tasu_gemm_dswiglu epilogue share this text: the same bits)
__device__ __forceinline__ void swiglu_bwd_f(float gf, float uf, float df, float& dg, float& du) {
  const float sg = sigmoid_f(gf);
  dg = df * uf * sg * (1.f + gf * (1.f - sg));
  du = df * gf * sg;
}
