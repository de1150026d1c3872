// HBM-bound elementwise kernels: SwiGLU (Qwen2MLP, modeling_qwen2.py:46-48), SiLU (projector.py:142), ReLU
// (SenseVoice.py:63), casts, transposes.  16-byte vector accesses per lane, grid-stride, <= 2048 blocks.
#include "common.h"
#include "../../include/tasu_hip.h"

namespace {


inline int grid_for(int64_t nvec) {
  int64_t b = (nvec + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

// act[m, i] = bf16( bf16(silu(g)) * u ),  gu = [M, 2I] = gate | up
__global__ void swiglu_fwd_kernel(const bf16* __restrict__ gu, bf16* __restrict__ act, int M, int I) {
  const int ic = I / 8;
  const int64_t total = (int64_t)M * ic;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = v / ic;
    const int c = (int)(v - m * ic);
    const bf16x8 g = *(const bf16x8*)(gu + m * 2 * I + c * 8);
    const bf16x8 u = *(const bf16x8*)(gu + m * 2 * I + I + c * 8);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float s = bf16_round(silu_f((float)g[j]));
      o[j] = (bf16)(s * (float)u[j]);
    }
    *(bf16x8*)(act + m * I + c * 8) = o;
  }
}

// dgate = dact * u * silu'(g),  dup = dact * silu(g);  silu'(g) = sig(g) * (1 + g*(1 - sig(g)))
__global__ void swiglu_bwd_kernel(const bf16* __restrict__ dact, const bf16* __restrict__ gu, bf16* __restrict__ dgu, int M,
                                  int I) {
  const int ic = I / 8;
  const int64_t total = (int64_t)M * ic;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = v / ic;
    const int c = (int)(v - m * ic);
    const bf16x8 g = *(const bf16x8*)(gu + m * 2 * I + c * 8);
    const bf16x8 u = *(const bf16x8*)(gu + m * 2 * I + I + c * 8);
    const bf16x8 d = *(const bf16x8*)(dact + m * I + c * 8);
    bf16x8 dg, du;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float dgf, duf;
      swiglu_bwd_f((float)g[j], (float)u[j], (float)d[j], dgf, duf);
      dg[j] = (bf16)dgf;
      du[j] = (bf16)duf;
    }
    *(bf16x8*)(dgu + m * 2 * I + c * 8) = dg;
    *(bf16x8*)(dgu + m * 2 * I + I + c * 8) = du;
  }
}

template <int OP>  // 0 silu fwd, 1 relu fwd
__global__ void unary_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int64_t n) {
  const int64_t nv = n / 8;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (int64_t)gridDim.x * blockDim.x) {
    const bf16x8 a = *(const bf16x8*)(x + v * 8);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = (float)a[j];
      o[j] = (bf16)(OP == 0 ? silu_f(f) : fmaxf(f, 0.f));
    }
    *(bf16x8*)(y + v * 8) = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const int64_t i = nv * 8 + threadIdx.x;
    const float f = (float)x[i];
    y[i] = (bf16)(OP == 0 ? silu_f(f) : fmaxf(f, 0.f));
  }
}

__global__ void silu_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x, bf16* __restrict__ dx, int64_t n) {
  const int64_t nv = n / 8;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (int64_t)gridDim.x * blockDim.x) {
    const bf16x8 a = *(const bf16x8*)(x + v * 8);
    const bf16x8 d = *(const bf16x8*)(dy + v * 8);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = (float)a[j];
      const float sg = sigmoid_f(f);
      o[j] = (bf16)((float)d[j] * sg * (1.f + f * (1.f - sg)));
    }
    *(bf16x8*)(dx + v * 8) = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const int64_t i = nv * 8 + threadIdx.x;
    const float f = (float)x[i];
    const float sg = sigmoid_f(f);
    dx[i] = (bf16)((float)dy[i] * sg * (1.f + f * (1.f - sg)));
  }
}

// dx = dy where x > 0 else 0 (ReLU of EncoderProjectorConcat, projector.py:35)
__global__ void relu_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x, bf16* __restrict__ dx, int64_t n) {
  const int64_t nv = n / 8;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (int64_t)gridDim.x * blockDim.x) {
    const bf16x8 a = *(const bf16x8*)(x + v * 8);
    const bf16x8 d = *(const bf16x8*)(dy + v * 8);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (float)a[j] > 0.f ? d[j] : (bf16)0.f;
    *(bf16x8*)(dx + v * 8) = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const int64_t i = nv * 8 + threadIdx.x;
    dx[i] = (float)x[i] > 0.f ? dy[i] : (bf16)0.f;
  }
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16* __restrict__ y, int64_t n) {
  const int64_t nv = n / 4;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 a = *(const f32x4*)(x + v * 4);
    *(bf16x4*)(y + v * 4) = __builtin_convertvector(a, bf16x4);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = nv * 4 + threadIdx.x;
    y[i] = (bf16)x[i];
  }
}

// y = bf16(slab_0 + slab_1 + ... ) in slab order (the finish of a split-K GEMM: fp32 partial sums, ONE rounding).  n % 4 == 0.
__global__ void sum_slabs_bf16_kernel(const float* __restrict__ slabs, int n_slabs, int64_t stride, bf16* __restrict__ y, int64_t n) {
  const int64_t nv = n / 4;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (int64_t)gridDim.x * blockDim.x) {
    f32x4 a = *(const f32x4*)(slabs + v * 4);
    for (int s = 1; s < n_slabs; ++s) a += *(const f32x4*)(slabs + (size_t)s * stride + v * 4);
    *(bf16x4*)(y + v * 4) = __builtin_convertvector(a, bf16x4);
  }
}

// out[c][r] = in[r][c] for r < R, c < C; zero for the padding region up to (Cpad rows, Rpad cols) of out.
// 64x64 tiles through LDS; grid (ceil(Rpad/64), ceil(Cpad/64)).
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16* __restrict__ in, int ld_in, bf16* __restrict__ out,
                                                             int ld_out, int R, int C, int Rpad, int Cpad) {
  __shared__ bf16 tile[64][64 + 2];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  // interior tiles of 16-byte-aligned matrices: two 16-B loads and two 16-B stores per thread (the element-wise path
  // below moved the 100-MB projector transposes at under 1 TB/s)
  const bool fast = r0 + 64 <= R && c0 + 64 <= C && !(ld_in & 7) && !(ld_out & 7) && !(((uintptr_t)in | (uintptr_t)out) & 15);
  if (fast) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = i * 256 + threadIdx.x, rl = idx >> 3, ch = idx & 7;
      const bf16x8 v = *(const bf16x8*)(in + (size_t)(r0 + rl) * ld_in + c0 + ch * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) tile[rl][ch * 8 + j] = v[j];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = i * 256 + threadIdx.x, cl = idx >> 3, ch = idx & 7;
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tile[ch * 8 + j][cl];
      *(bf16x8*)(out + (size_t)(c0 + cl) * ld_out + r0 + ch * 8) = v;
    }
    return;
  }
  // load: thread -> row tl = tid>>2, 16 columns
  {
    const int tl = threadIdx.x >> 2, part = threadIdx.x & 3;
    const int r = r0 + tl;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int c = c0 + part * 16 + j;
      tile[tl][part * 16 + j] = (r < R && c < C) ? in[(size_t)r * ld_in + c] : (bf16)0.f;
    }
  }
  __syncthreads();
  {
    const int cl = threadIdx.x >> 2, part = threadIdx.x & 3;
    const int c = c0 + cl;
    if (c < Cpad) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int r = r0 + part * 16 + j;
        if (r < Rpad) out[(size_t)c * ld_out + r] = tile[part * 16 + j][cl];
      }
    }
  }
}

}  // namespace

extern "C" int tasu_swiglu_fwd(const void* gu, void* act, int M, int I, void* stream) {
  if (!gu || !act || M <= 0 || I <= 0 || I % 8) return TASU_ERR_ARG;
  TASU_LAUNCH(swiglu_fwd_kernel, dim3(grid_for((int64_t)M * I / 8)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)gu, (bf16*)act, M, I);
  return TASU_OK;
}
extern "C" int tasu_swiglu_bwd(const void* dact, const void* gu, void* dgu, int M, int I, void* stream) {
  if (!dact || !gu || !dgu || M <= 0 || I <= 0 || I % 8) return TASU_ERR_ARG;
  TASU_LAUNCH(swiglu_bwd_kernel, dim3(grid_for((int64_t)M * I / 8)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)dact, (const bf16*)gu, (bf16*)dgu, M, I);
  return TASU_OK;
}
extern "C" int tasu_silu_fwd(const void* x, void* y, int64_t n, void* stream) {
  if (!x || !y || n <= 0) return TASU_ERR_ARG;
  TASU_LAUNCH(unary_kernel<0>, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y, n);
  return TASU_OK;
}
extern "C" int tasu_relu_fwd(const void* x, void* y, int64_t n, void* stream) {
  if (!x || !y || n <= 0) return TASU_ERR_ARG;
  TASU_LAUNCH(unary_kernel<1>, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y, n);
  return TASU_OK;
}
extern "C" int tasu_silu_bwd(const void* dy, const void* x, void* dx, int64_t n, void* stream) {
  if (!dy || !x || !dx || n <= 0) return TASU_ERR_ARG;
  TASU_LAUNCH(silu_bwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16*)dy,
                     (const bf16*)x, (bf16*)dx, n);
  return TASU_OK;
}
extern "C" int tasu_relu_bwd(const void* dy, const void* x, void* dx, int64_t n, void* stream) {
  if (!dy || !x || !dx || n <= 0) return TASU_ERR_ARG;
  TASU_LAUNCH(relu_bwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16*)dy, (const bf16*)x, (bf16*)dx, n);
  return TASU_OK;
}
extern "C" int tasu_cast_f32_bf16(const void* in, void* out, int64_t n, void* stream) {
  if (!in || !out || n <= 0) return TASU_ERR_ARG;
  TASU_LAUNCH(cast_f32_bf16_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)in,
                     (bf16*)out, n);
  return TASU_OK;
}
extern "C" int tasu_transpose_bf16(const void* in, int ld_in, void* out, int ld_out, int R, int C, int Rpad, int Cpad,
                                   void* stream) {
  if (!in || !out || R <= 0 || C <= 0 || Rpad < R || Cpad < C || ld_in < C || ld_out < Rpad) return TASU_ERR_ARG;
  dim3 grid((Rpad + 63) / 64, (Cpad + 63) / 64);
  TASU_LAUNCH(transpose_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)in, ld_in, (bf16*)out,
                     ld_out, R, C, Rpad, Cpad);
  return TASU_OK;
}
extern "C" int tasu_sum_slabs_bf16(const float* slabs, int n_slabs, int64_t slab_stride, void* out, int64_t n, void* stream) {
  if (!slabs || !out || n_slabs < 1 || n <= 0 || n % 4 || slab_stride < n || slab_stride % 4) return TASU_ERR_ARG;
  TASU_LAUNCH(sum_slabs_bf16_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, slabs, n_slabs, slab_stride,
              (bf16*)out, n);
  return TASU_OK;
}
