// RMSNorm (Qwen2RMSNorm, transformers modeling_qwen2.py:247-252) fwd + dgrad, LayerNorm (projector.py:139 over
// D = 25055; SenseVoice.py:270-282 over 560/512) fwd + parameter gradients, column sums (bias gradients).
// All HBM-bound: 16-byte accesses, wavefront-shuffle reductions, fp32 statistics.
#include "common.h"
#include "../../include/tasu_hip.h"

namespace {

// one wave per row, 4 rows per 256-thread block.  D % 4 == 0.
// ``src`` (optional): output row i is the norm of input row src[i]; src[i] < 0 gives a zero row and rstd 0 (the lm_head of
// the training step only projects the rows that carry a label).
// ``frag``: y is written in the MFMA fragment order of the decode step's streaming GEMMs (csrc/gemm_stream.hip: element
// (row, c) at [c / 32][row / 16][16 * ((c % 32) / 8) + row % 16][c % 8], rows < 64) instead of row-major -- same arithmetic.
__device__ __forceinline__ size_t frag_offset(int row, int c) {
  return ((((size_t)(c >> 5) * 4 + (row >> 4)) * 64 + ((c & 31) >> 3) * 16 + (row & 15)) << 3) + (c & 7);
}
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const float* __restrict__ x, const int32_t* __restrict__ src,
                                                          const float* __restrict__ w, bf16* __restrict__ y,
                                                          float* __restrict__ rstd, int M, int D, float eps, int frag, int ldy) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const int srow = src ? src[row] : row;
  if (srow < 0) {
    for (int c = lane * 4; c < D; c += 256) *(bf16x4*)(y + (size_t)row * ldy + c) = bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
    if (lane == 0 && rstd) rstd[row] = 0.f;
    return;
  }
  const float* xr = x + (size_t)srow * D;
  float ss = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 v = *(const f32x4*)(xr + c);
    ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  ss = wave_sum(ss);
  const float r = rsqrtf(ss / (float)D + eps);
  if (lane == 0 && rstd) rstd[row] = r;
  bf16* yr = y + (size_t)row * ldy;
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 v = *(const f32x4*)(xr + c);
    const f32x4 g = *(const f32x4*)(w + c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = g[j] * (v[j] * r);
    *(bf16x4*)(frag ? y + frag_offset(row, c) : yr + c) = __builtin_convertvector(o, bf16x4);
  }
}

// Register-resident forms for D = NG * 256 (1536 -> 6, 3584 -> 14): the row is loaded once, all loads in flight together;
// same arithmetic and summation order as the generic kernels.
template <int NG>
__global__ __launch_bounds__(256) void rmsnorm_fwd_reg_kernel(const float* __restrict__ x, const int32_t* __restrict__ src,
                                                              const float* __restrict__ w, bf16* __restrict__ y,
                                                              float* __restrict__ rstd, int M, float eps, int frag, int ldy) {
  constexpr int D = NG * 256;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const int srow = src ? src[row] : row;
  if (srow < 0) {
#pragma unroll
    for (int g = 0; g < NG; ++g)
      *(bf16x4*)(y + (size_t)row * ldy + lane * 4 + g * 256) = bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
    if (lane == 0 && rstd) rstd[row] = 0.f;
    return;
  }
  const float* xr = x + (size_t)srow * D + lane * 4;
  f32x4 v[NG], gw[NG];                // the norm weights travel with the row: one memory round trip before the stores
#pragma unroll
  for (int g = 0; g < NG; ++g) v[g] = *(const f32x4*)(xr + g * 256);
#pragma unroll
  for (int g = 0; g < NG; ++g) gw[g] = *(const f32x4*)(w + lane * 4 + g * 256);
  float ss = 0.f;
#pragma unroll
  for (int g = 0; g < NG; ++g) ss += v[g][0] * v[g][0] + v[g][1] * v[g][1] + v[g][2] * v[g][2] + v[g][3] * v[g][3];
  ss = wave_sum(ss);
  const float r = rsqrtf(ss / (float)D + eps);
  if (lane == 0 && rstd) rstd[row] = r;
  bf16* yr = y + (size_t)row * ldy + lane * 4;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = gw[g][j] * (v[g][j] * r);
    *(bf16x4*)(frag ? y + frag_offset(row, lane * 4 + g * 256) : yr + g * 256) = __builtin_convertvector(o, bf16x4);
  }
}


template <int NG>
// ``slot`` (optional): dy and rstd are COMPACT (one row per labelled position); slot[row] is row's compact index, or < 0
// for a row whose output gradient is zero (then dx = dxb = 0; accumulate must be 0 with a slot map).
__global__ __launch_bounds__(256) void rmsnorm_bwd_reg_kernel(const bf16* __restrict__ dy, const float* __restrict__ x,
                                                              const float* __restrict__ w, const float* __restrict__ rstd,
                                                              const int32_t* __restrict__ slot, float* __restrict__ dx,
                                                              bf16* __restrict__ dxb, int accumulate, int M,
                                                              const float* __restrict__ resid_c) {
  constexpr int D = NG * 256;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const size_t base = (size_t)row * D + lane * 4;
  const int crow = slot ? slot[row] : row;
  if (crow < 0) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      *(f32x4*)(dx + base + g * 256) = f32x4{0.f, 0.f, 0.f, 0.f};
      if (dxb) *(bf16x4*)(dxb + base + g * 256) = bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
    }
    return;
  }
  const size_t cbase = (size_t)crow * D + lane * 4;
  f32x4 v[NG], d[NG], o[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    v[g] = *(const f32x4*)(x + base + g * 256);
    d[g] = __builtin_convertvector(*(const bf16x4*)(dy + cbase + g * 256), f32x4);
    // resid_c (with a slot map): the COMPACT residual-stream gradient of the row, which this row's output starts from
    o[g] = accumulate ? *(const f32x4*)(dx + base + g * 256)
                      : (resid_c ? *(const f32x4*)(resid_c + cbase + g * 256) : f32x4{0.f, 0.f, 0.f, 0.f});
  }
  const float r = rstd[crow];
  float dot = 0.f;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const f32x4 gw = *(const f32x4*)(w + lane * 4 + g * 256);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      d[g][j] *= gw[j];                          // w * dy, reused below
      dot += d[g][j] * v[g][j] * r;
    }
  }
  dot = wave_sum(dot) / (float)D;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[g][j] += r * (d[g][j] - v[g][j] * r * dot);
    *(f32x4*)(dx + base + g * 256) = o[g];
    if (dxb) *(bf16x4*)(dxb + base + g * 256) = __builtin_convertvector(o[g], bf16x4);
  }
}

// dx += rstd * (w*dy - xhat * mean(w*dy*xhat)),  xhat = x * rstd
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ w, const float* __restrict__ rstd,
                                                          const int32_t* __restrict__ slot, float* __restrict__ dx,
                                                          bf16* __restrict__ dxb, int accumulate, int M, int D,
                                                          const float* __restrict__ resid_c) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const int crow = slot ? slot[row] : row;
  if (crow < 0) {
    for (int c = lane * 4; c < D; c += 256) {
      *(f32x4*)(dx + (size_t)row * D + c) = f32x4{0.f, 0.f, 0.f, 0.f};
      if (dxb) *(bf16x4*)(dxb + (size_t)row * D + c) = bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
    }
    return;
  }
  const float* xr = x + (size_t)row * D;
  const bf16* dr = dy + (size_t)crow * D;
  const float r = rstd[crow];
  float dot = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 v = *(const f32x4*)(xr + c);
    const f32x4 g = *(const f32x4*)(w + c);
    const f32x4 d = __builtin_convertvector(*(const bf16x4*)(dr + c), f32x4);
#pragma unroll
    for (int j = 0; j < 4; ++j) dot += g[j] * d[j] * v[j] * r;
  }
  dot = wave_sum(dot) / (float)D;
  float* dxr = dx + (size_t)row * D;
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 v = *(const f32x4*)(xr + c);
    const f32x4 g = *(const f32x4*)(w + c);
    const f32x4 d = __builtin_convertvector(*(const bf16x4*)(dr + c), f32x4);
    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
    if (accumulate) o = *(const f32x4*)(dxr + c);
    else if (resid_c) o = *(const f32x4*)(resid_c + (size_t)crow * D + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] += r * (g[j] * d[j] - v[j] * r * dot);
    *(f32x4*)(dxr + c) = o;
    if (dxb) *(bf16x4*)(dxb + (size_t)row * D + c) = __builtin_convertvector(o, bf16x4);
  }
}

// LayerNorm forward: one 256-thread block per row, two-pass statistics (mean, then centered variance).
template <bool OUT_F32>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, int ldx,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            void* __restrict__ y, int ldy, float* __restrict__ mean,
                                                            float* __restrict__ rstd, int D, float eps) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const float* xr = x + (size_t)row * ldx;
  const bool vec = ((ldx & 3) == 0) && ((((uintptr_t)x) & 15) == 0);
  const int D4 = vec ? (D & ~3) : 0;
  float s = 0.f;
  for (int c = threadIdx.x * 4; c < D4; c += 1024) {
    const f32x4 v = *(const f32x4*)(xr + c);
    s += v[0] + v[1] + v[2] + v[3];
  }
  for (int c = D4 + threadIdx.x; c < D; c += 256) s += xr[c];
  const float mu = block_sum<4>(s, red) / (float)D;
  float q = 0.f;
  for (int c = threadIdx.x * 4; c < D4; c += 1024) {
    const f32x4 v = *(const f32x4*)(xr + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) q += (v[j] - mu) * (v[j] - mu);
  }
  for (int c = D4 + threadIdx.x; c < D; c += 256) q += (xr[c] - mu) * (xr[c] - mu);
  const float var = block_sum<4>(q, red) / (float)D;
  const float r = rsqrtf(var + eps);
  if (threadIdx.x == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = r;
  }
  for (int c = threadIdx.x; c < ldy; c += 256) {
    const float o = c < D ? (xr[c] - mu) * r * gamma[c] + beta[c] : 0.f;
    if (OUT_F32)
      ((float*)y)[(size_t)row * ldy + c] = o;
    else
      ((bf16*)y)[(size_t)row * ldy + c] = (bf16)o;
  }
}

// Short rows (D <= 1024, D % 4 == 0: the SANM encoder's 512- and 560-wide norms): one wave per row, the row in registers,
// four rows per block.  Same two-pass statistics as above (the block-per-row kernel spends most of its 16 us per launch in
// barriers for 2 KB of data).
template <bool OUT_F32>
__global__ __launch_bounds__(256) void layernorm_fwd_wave_kernel(const float* __restrict__ x, int ldx,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 void* __restrict__ y, int ldy, float* __restrict__ mean,
                                                                 float* __restrict__ rstd, int R, int D, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= R) return;
  const float* xr = x + (size_t)row * ldx;
  f32x4 v[4];
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int c = lane * 4 + g * 256;
    v[g] = c < D ? *(const f32x4*)(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    s += v[g][0] + v[g][1] + v[g][2] + v[g][3];
  }
  const float mu = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (lane * 4 + g * 256 < D) {
#pragma unroll
      for (int j = 0; j < 4; ++j) q += (v[g][j] - mu) * (v[g][j] - mu);
    }
  }
  const float r = rsqrtf(wave_sum(q) / (float)D + eps);
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = r;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int c = lane * 4 + g * 256;
    if (c < D) {
      const f32x4 ga = *(const f32x4*)(gamma + c), be = *(const f32x4*)(beta + c);
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (v[g][j] - mu) * r * ga[j] + be[j];
      if (OUT_F32) *(f32x4*)((float*)y + (size_t)row * ldy + c) = o;
      else *(bf16x4*)((bf16*)y + (size_t)row * ldy + c) = __builtin_convertvector(o, bf16x4);
    }
  }
  // pad columns [D, ldy) are zero like in the block-per-row kernel
  for (int c = D + lane; c < ldy; c += 64) {
    if (OUT_F32) ((float*)y)[(size_t)row * ldy + c] = 0.f;
    else ((bf16*)y)[(size_t)row * ldy + c] = (bf16)0.f;
  }
}

// dgamma/dbeta partials: grid (ceil(D/256), RSPLIT); thread = one column; rows r = split, split+RSPLIT, ...
// ws layout: [RSPLIT][2][Dws] fp32.
__global__ __launch_bounds__(256) void layernorm_bwd_partial_kernel(const bf16* __restrict__ dy, int lddy,
                                                                    const float* __restrict__ x, int ldx,
                                                                    const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd, float* __restrict__ ws,
                                                                    int R, int D) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int split = blockIdx.y, nsplit = gridDim.y;
  if (c >= D) return;
  float dg = 0.f, db = 0.f;
  for (int r = split; r < R; r += nsplit) {
    const float d = (float)dy[(size_t)r * lddy + c];
    const float xh = (x[(size_t)r * ldx + c] - mean[r]) * rstd[r];
    dg += d * xh;
    db += d;
  }
  ws[((size_t)split * 2 + 0) * D + c] = dg;
  ws[((size_t)split * 2 + 1) * D + c] = db;
}
__global__ __launch_bounds__(256) void layernorm_bwd_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta, int nsplit, int D) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= D) return;
  float dg = 0.f, db = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    dg += ws[((size_t)s * 2 + 0) * D + c];
    db += ws[((size_t)s * 2 + 1) * D + c];
  }
  dgamma[c] = dg;
  dbeta[c] = db;
}

// column sums of bf16 [R, C] (bias gradients).  One block = 32 columns x 16 row-lanes: thread t owns the column
// pair cp = t & 15 (one 4-byte load per row) for rows rl, rl+16, ... (rl = t >> 4); the 16 partial sums of a column
// are combined through LDS in a fixed order (deterministic, no atomics).  grid = ceil(C/32).
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ x, int ld, float* __restrict__ out, int R,
                                                     int C) {
  __shared__ float red[16][33];
  const int cp = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 32 + cp * 2;
  float s0 = 0.f, s1 = 0.f;
  if (c + 1 < C) {
    for (int r = rl; r < R; r += 16) {
      const bf16x2 v = *(const bf16x2*)(x + (size_t)r * ld + c);
      s0 += (float)v[0];
      s1 += (float)v[1];
    }
  } else if (c < C) {
    for (int r = rl; r < R; r += 16) s0 += (float)x[(size_t)r * ld + c];
  }
  red[rl][cp * 2] = s0;
  red[rl][cp * 2 + 1] = s1;
  __syncthreads();
  if (threadIdx.x < 32) {
    const int cc = blockIdx.x * 32 + threadIdx.x;
    if (cc < C) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) t += red[i][threadIdx.x];
      out[cc] = t;
    }
  }
}

}  // namespace

static int rmsnorm_fwd_any(const float* x, const int32_t* src, const float* w, void* y, float* rstd, int M, int D, float eps,
                           void* stream, int frag = 0, int ldy = 0) {
  if (!x || !w || !y || M <= 0 || D <= 0 || D % 4) return TASU_ERR_ARG;
  if (ldy == 0) ldy = D;
  if (ldy < D || ldy % 4 || (frag && ldy != D)) return TASU_ERR_ARG;
  // one wave per row.  A decode step's <= 64 rows go one per workgroup (64 CUs instead of 16: the kernel is one memory round trip
  // and a workgroup's rows share a CU's load path); the training step's thousands of rows four per workgroup.  TASU_NORM_ROWS (lab build): A/B.
  static const int rows_env = [] { const char* e = tasu_lab_env("TASU_NORM_ROWS"); return e ? atoi(e) : 0; }();
  const int rows = rows_env == 1 || rows_env == 2 || rows_env == 4 ? rows_env : (M <= 64 ? 1 : 4);
  const dim3 grid((M + rows - 1) / rows), block(64 * rows);
  hipStream_t st = (hipStream_t)stream;
  if (D == 1536) TASU_LAUNCH(rmsnorm_fwd_reg_kernel<6>, grid, block, 0, st, x, src, w, (bf16*)y, rstd, M, eps, frag, ldy);
  else if (D == 3584) TASU_LAUNCH(rmsnorm_fwd_reg_kernel<14>, grid, block, 0, st, x, src, w, (bf16*)y, rstd, M, eps, frag, ldy);
  else if (D == 256) TASU_LAUNCH(rmsnorm_fwd_reg_kernel<1>, grid, block, 0, st, x, src, w, (bf16*)y, rstd, M, eps, frag, ldy);
  else TASU_LAUNCH(rmsnorm_fwd_kernel, grid, block, 0, st, x, src, w, (bf16*)y, rstd, M, D, eps, frag, ldy);
  return TASU_OK;
}
extern "C" int tasu_rmsnorm_fwd_ld(const float* x, const float* w, void* y, int ldy, float* rstd, int M, int D, float eps, void* stream) {
  return rmsnorm_fwd_any(x, nullptr, w, y, rstd, M, D, eps, stream, 0, ldy);
}
extern "C" int tasu_rmsnorm_fwd_frag(const float* x, const float* w, void* y_frag, int M, int D, float eps, void* stream) {
  if (M > 64 || D % 32) return TASU_ERR_ARG;
  return rmsnorm_fwd_any(x, nullptr, w, y_frag, nullptr, M, D, eps, stream, 1);
}
static int rmsnorm_bwd_any(const void* dy, const float* x, const float* w, const float* rstd, const int32_t* slot, float* dx,
                           void* dx_bf16, int accumulate, int M, int D, void* stream, const float* resid_c = nullptr) {
  if (!dy || !x || !w || !rstd || !dx || M <= 0 || D <= 0 || D % 4 || (slot && accumulate) || (resid_c && !slot)) return TASU_ERR_ARG;
  const dim3 grid((M + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  const bf16* d = (const bf16*)dy;
  bf16* db = (bf16*)dx_bf16;
  if (D == 1536) TASU_LAUNCH(rmsnorm_bwd_reg_kernel<6>, grid, dim3(256), 0, st, d, x, w, rstd, slot, dx, db, accumulate, M, resid_c);
  else if (D == 3584) TASU_LAUNCH(rmsnorm_bwd_reg_kernel<14>, grid, dim3(256), 0, st, d, x, w, rstd, slot, dx, db, accumulate, M, resid_c);
  else if (D == 256) TASU_LAUNCH(rmsnorm_bwd_reg_kernel<1>, grid, dim3(256), 0, st, d, x, w, rstd, slot, dx, db, accumulate, M, resid_c);
  else TASU_LAUNCH(rmsnorm_bwd_kernel, grid, dim3(256), 0, st, d, x, w, rstd, slot, dx, db, accumulate, M, D, resid_c);
  return TASU_OK;
}
extern "C" int tasu_rmsnorm_fwd(const float* x, const float* w, void* y, float* rstd, int M, int D, float eps,
                                void* stream) {
  return rmsnorm_fwd_any(x, nullptr, w, y, rstd, M, D, eps, stream);
}
extern "C" int tasu_rmsnorm_bwd(const void* dy, const float* x, const float* w, const float* rstd, float* dx, void* dx_bf16,
                                int accumulate, int M, int D, void* stream) {
  return rmsnorm_bwd_any(dy, x, w, rstd, nullptr, dx, dx_bf16, accumulate, M, D, stream);
}
extern "C" int tasu_rmsnorm_fwd_rows(const float* x, const int32_t* src_rows, const float* w, void* y, float* rstd, int n_rows,
                                     int D, float eps, void* stream) {
  if (!src_rows) return TASU_ERR_ARG;
  return rmsnorm_fwd_any(x, src_rows, w, y, rstd, n_rows, D, eps, stream);
}
extern "C" int tasu_rmsnorm_bwd_rows(const void* dy_compact, const float* x, const float* w, const float* rstd_compact,
                                     const int32_t* slot, float* dx, void* dx_bf16, int M, int D, void* stream) {
  if (!slot) return TASU_ERR_ARG;
  return rmsnorm_bwd_any(dy_compact, x, w, rstd_compact, slot, dx, dx_bf16, 0, M, D, stream);
}
extern "C" int tasu_rmsnorm_bwd_rows_resid(const void* dy_compact, const float* x, const float* w, const float* rstd_compact,
                                           const int32_t* slot, const float* resid_compact, float* dx, void* dx_bf16, int M, int D,
                                           void* stream) {
  if (!slot || !resid_compact) return TASU_ERR_ARG;
  return rmsnorm_bwd_any(dy_compact, x, w, rstd_compact, slot, dx, dx_bf16, 0, M, D, stream, resid_compact);
}
extern "C" int tasu_layernorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, void* y, int ldy,
                                  int y_is_f32, float* mean, float* rstd, int R, int D, float eps, void* stream) {
  if (!x || !gamma || !beta || !y || R <= 0 || D <= 0 || ldx < D || ldy < D) return TASU_ERR_ARG;
  const bool wave_ok = D <= 1024 && D % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 &&
                       !(((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15);
  if (wave_ok) {
    if (y_is_f32)
      TASU_LAUNCH(layernorm_fwd_wave_kernel<true>, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta,
                  y, ldy, mean, rstd, R, D, eps);
    else
      TASU_LAUNCH(layernorm_fwd_wave_kernel<false>, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta,
                  y, ldy, mean, rstd, R, D, eps);
    return TASU_OK;
  }
  if (y_is_f32)
    TASU_LAUNCH(layernorm_fwd_kernel<true>, dim3(R), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta, y, ldy,
                       mean, rstd, D, eps);
  else
    TASU_LAUNCH(layernorm_fwd_kernel<false>, dim3(R), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta, y, ldy,
                       mean, rstd, D, eps);
  return TASU_OK;
}
extern "C" int tasu_layernorm_bwd_params(const void* dy, int lddy, const float* x, int ldx, const float* mean,
                                         const float* rstd, float* dgamma, float* dbeta, float* ws, int R, int D,
                                         void* stream) {
  if (!dy || !x || !mean || !rstd || !dgamma || !dbeta || !ws || R <= 0 || D <= 0) return TASU_ERR_ARG;
  const int nsplit = TASU_LN_BWD_SPLIT;
  TASU_LAUNCH(layernorm_bwd_partial_kernel, dim3((D + 255) / 256, nsplit), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)dy, lddy, x, ldx, mean, rstd, ws, R, D);
  TASU_LAUNCH(layernorm_bwd_reduce_kernel, dim3((D + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws, dgamma,
                     dbeta, nsplit, D);
  return TASU_OK;
}
extern "C" int tasu_colsum_bf16(const void* x, int ld, float* out, int R, int C, void* stream) {
  if (!x || !out || R <= 0 || C <= 0) return TASU_ERR_ARG;
  if (ld % 2) return TASU_ERR_ARG;
  TASU_LAUNCH(colsum_kernel, dim3((C + 31) / 32), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ld, out, R, C);
  return TASU_OK;
}
