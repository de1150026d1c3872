// fp32 TRAINING step kernels (train_config.use_fp16 = false during training: the reference's shipped recipe,
// Multitask/scripts/finetune_deespeed_sensevoice.sh:37, runs forward AND backward without autocast).  The forward is the fp32 prompt
// pass of fp32.hip with its activations kept; this file holds the backward of every non-GEMM operator (the GEMMs are
// tasu_f32_gemm_nt on transposed fp32 weight copies) and the small reductions of the projector's weight gradients.  Correctness
// mode: one thread per element / one workgroup per row, fp32 everywhere, deterministic sums; 11x slower than the bf16 step.
#include "common.h"
#include "../../include/tasu_hip.h"

namespace tasu_f32t {

constexpr int HD = 128;

__device__ __forceinline__ float sigmoid_exact(float x) { return 1.f / (1.f + expf(-x)); }

// Qwen2RMSNorm backward: y = w * x * rstd, rstd = rsqrt(mean(x^2) + eps):
//   dx = rstd * (w . dy) - x * (rstd^3 / D) * sum_j(w_j dy_j x_j);   accumulate: dx += (the residual stream's gradient)
__global__ __launch_bounds__(1024) void rmsnorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ w,
                                                           float* __restrict__ dx, int D, float eps, int accumulate) {
  __shared__ float red[16];
  const size_t row = (size_t)blockIdx.x * D;
  float ss = 0.f, dot = 0.f;
  for (int c = threadIdx.x; c < D; c += 1024) {
    const float xv = x[row + c];
    ss += xv * xv;
    dot += w[c] * dy[row + c] * xv;
  }
  ss = block_sum<16>(ss, red);
  dot = block_sum<16>(dot, red);
  const float rs = rsqrtf(ss / (float)D + eps);
  const float k = rs * rs * rs / (float)D * dot;
  for (int c = threadIdx.x; c < D; c += 1024) {
    const float g = rs * (w[c] * dy[row + c]) - x[row + c] * k;
    dx[row + c] = accumulate ? dx[row + c] + g : g;
  }
}

// SwiGLU backward: act = silu(g) * u  ->  dg = d * u * sig(g) * (1 + g * (1 - sig(g))),  du = d * silu(g)
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const float* __restrict__ dact, const float* __restrict__ gu, float* __restrict__ dgu,
                                                         int M, int I) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)M * I) return;
  const size_t m = idx / I, c = idx - m * I;
  const float g = gu[m * 2 * I + c], u = gu[m * 2 * I + I + c], d = dact[idx];
  const float sg = sigmoid_exact(g);
  dgu[m * 2 * I + c] = d * u * sg * (1.f + g * (1.f - sg));
  dgu[m * 2 * I + I + c] = d * g * sg;
}

// y = silu(x) / dx = dy * silu'(x) elementwise (projector: Linear -> SiLU -> Linear)
__global__ __launch_bounds__(256) void silu_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = x[i], sg = sigmoid_exact(v);
  out[i] = dy ? dy[i] * sg * (1.f + v * (1.f - sg)) : v * sg;
}

// out[c] = sum_r x[r, c] (bias gradients), rows in ascending order: one thread per column, coalesced over the columns
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int ld, float* __restrict__ out, int R, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int r = 0; r < R; ++r) s += x[(size_t)r * ld + c];
  out[c] = s;
}

// LayerNorm parameter gradients (the projector's norm over the CTC vocabulary; the input is a frozen posterior: no dx):
// dgamma[j] = sum_r dy[r, j] * (x[r, j] - mean[r]) * rstd[r],  dbeta[j] = sum_r dy[r, j]
__global__ __launch_bounds__(256) void layernorm_bwd_params_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                   float* __restrict__ dgamma, float* __restrict__ dbeta, int R, int D) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= D) return;
  float dg = 0.f, db = 0.f;
  for (int r = 0; r < R; ++r) {
    const float d = dy[(size_t)r * lddy + c];
    dg += d * ((x[(size_t)r * ldx + c] - mean[r]) * rstd[r]);
    db += d;
  }
  dgamma[c] = dg;
  dbeta[c] = db;
}

// dst[c, r] = src[r, c] for r < R, 0 for R <= r < Rpad (operands of the weight-gradient GEMMs: tasu_f32_gemm_nt contracts rows)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, int R, int C,
                                                        int Rpad) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < C) ? src[(size_t)r * lds + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < C && r < Rpad) dst[(size_t)c * ldd + r] = tile[tx][i];
  }
}

// dproj[r, :] = dx[rows[r], :] (rows[r] < 0: zeros): the gradient rows that hold audio -> the projector output's gradient
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ dx, const int32_t* __restrict__ rows, float* __restrict__ out,
                                                          int n, int D) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)n * D) return;
  const int r = (int)(idx / D), c = (int)(idx - (size_t)r * D);
  out[idx] = rows[r] >= 0 ? dx[(size_t)rows[r] * D + c] : 0.f;
}

// ---------------------------------------------------------------------------------------------------------- attention backward
// Causal GQA attention backward over the prompt (keys [kstart[b], s] for query s), fp32, in two launches that recompute the
// probabilities from the saved q|k|v (rotated) instead of storing [S, S] matrices:
//   bwd_q   one workgroup per (batch row, query, KV head): scores -> softmax statistics (lse), dP = dO . V^T, delta = sum_j P dP,
//           dS = P (dP - delta);  dQ = scale * dS . K;  lse / delta are kept per (row, head) for the second launch
//   bwd_kv  one workgroup per (batch row, key, KV head): for every query >= the key and every head of the group:
//           P = exp(scale q.k - lse), dS = P (dO.v - delta);  dV = sum P dO,  dK = scale * sum dS q   (the group's heads summed)
// Same structure as fp32.hip's forward: phase 1 thread = the other index (whole 128-dim dot products), phase 2 lanes = dims.
constexpr int MAXK = 2048;
template <int REP>
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ dout, const int32_t* __restrict__ kstart,
                                                         float* __restrict__ dqkv, float* __restrict__ lse_out, float* __restrict__ delta_out, int B,
                                                         int S, int H, int G, float scale) {
  extern __shared__ float smem[];
  float* sq = smem;                                  // REP * 128 queries
  float* sdo = sq + REP * HD;                        // REP * 128 output gradients
  float* sp = sdo + REP * HD;                        // REP * MAXK probabilities, then dS
  float* sdp = sp + REP * MAXK;                      // REP * MAXK dP
  float* part = sdp + REP * MAXK;                    // 4 * REP * 128 partial dQ
  float* red = part + 4 * REP * HD;                  // 4 * REP
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int g = blockIdx.x % G;
  const long long bs = blockIdx.x / G;
  const int b = (int)(bs / S), s = (int)(bs - (long long)b * S);
  const int LD = (H + 2 * G) * HD;
  const float* base = qkv + (size_t)b * S * LD;
  float* dq = dqkv + (size_t)bs * LD + g * REP * HD;
  const int k_lo = kstart[b], nk = s + 1 - k_lo;
  if (nk <= 0) {                                     // a padding position
    for (int i = t; i < REP * HD; i += 256) dq[i] = 0.f;
    if (t < REP) lse_out[((size_t)b * H + g * REP + t) * S + s] = 0.f, delta_out[((size_t)b * H + g * REP + t) * S + s] = 0.f;
    return;
  }
  for (int i = t; i < REP * HD; i += 256) {
    sq[i] = base[(size_t)s * LD + g * REP * HD + i];
    sdo[i] = dout[(size_t)bs * (H * HD) + g * REP * HD + i];
  }
  __syncthreads();
  float mx[REP];
#pragma unroll
  for (int h = 0; h < REP; ++h) mx[h] = -__builtin_inff();
  for (int j = t; j < nk; j += 256) {
    const f32x4* kr = (const f32x4*)(base + (size_t)(k_lo + j) * LD + (H + g) * HD);
    const f32x4* vr = (const f32x4*)(base + (size_t)(k_lo + j) * LD + (H + G + g) * HD);
    float a[REP], dp[REP];
#pragma unroll
    for (int h = 0; h < REP; ++h) a[h] = dp[h] = 0.f;
#pragma unroll 4
    for (int c = 0; c < HD / 4; ++c) {
      const f32x4 kv = kr[c], vv = vr[c];
#pragma unroll
      for (int h = 0; h < REP; ++h) {
        const f32x4 qv = *(const f32x4*)(sq + h * HD + c * 4), dv = *(const f32x4*)(sdo + h * HD + c * 4);
        a[h] += kv[0] * qv[0] + kv[1] * qv[1] + kv[2] * qv[2] + kv[3] * qv[3];
        dp[h] += vv[0] * dv[0] + vv[1] * dv[1] + vv[2] * dv[2] + vv[3] * dv[3];
      }
    }
#pragma unroll
    for (int h = 0; h < REP; ++h) {
      const float sc = a[h] * scale;
      sp[h * MAXK + j] = sc;
      sdp[h * MAXK + j] = dp[h];
      mx[h] = fmaxf(mx[h], sc);
    }
  }
#pragma unroll
  for (int h = 0; h < REP; ++h) {
    mx[h] = wave_max(mx[h]);
    if (lane == 0) red[wave * REP + h] = mx[h];
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < REP; ++h) mx[h] = fmaxf(fmaxf(red[h], red[REP + h]), fmaxf(red[2 * REP + h], red[3 * REP + h]));
  __syncthreads();
  float inv[REP], lse[REP];
#pragma unroll
  for (int h = 0; h < REP; ++h) {
    float sum = 0.f;
    for (int j = t; j < nk; j += 256) {
      const float e = expf(sp[h * MAXK + j] - mx[h]);
      sp[h * MAXK + j] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wave * REP + h] = sum;
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < REP; ++h) {
    const float l = ((red[h] + red[REP + h]) + red[2 * REP + h]) + red[3 * REP + h];
    inv[h] = 1.f / l;
    lse[h] = mx[h] + logf(l);
  }
  __syncthreads();
  // delta_h = sum_j P dP
  float dl[REP];
#pragma unroll
  for (int h = 0; h < REP; ++h) {
    float sum = 0.f;
    for (int j = t; j < nk; j += 256) sum += sp[h * MAXK + j] * inv[h] * sdp[h * MAXK + j];
    sum = wave_sum(sum);
    if (lane == 0) red[wave * REP + h] = sum;
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < REP; ++h) dl[h] = ((red[h] + red[REP + h]) + red[2 * REP + h]) + red[3 * REP + h];
  if (t < REP) {
    lse_out[((size_t)b * H + g * REP + t) * S + s] = lse[t];
    delta_out[((size_t)b * H + g * REP + t) * S + s] = dl[t];
  }
  // dS (scaled) in place of P
#pragma unroll
  for (int h = 0; h < REP; ++h)
    for (int j = t; j < nk; j += 256) sp[h * MAXK + j] = sp[h * MAXK + j] * inv[h] * (sdp[h * MAXK + j] - dl[h]) * scale;
  __syncthreads();
  // dQ_h = sum_j dS[h][j] K_j: this wave's quarter of the keys, lanes = dims
  float o0[REP], o1[REP];
#pragma unroll
  for (int h = 0; h < REP; ++h) o0[h] = o1[h] = 0.f;
  const int q4 = (nk + 3) >> 2, j_lo = wave * q4, j_hi = min(nk, j_lo + q4);
  for (int j = j_lo; j < j_hi; ++j) {
    const float* kr = base + (size_t)(k_lo + j) * LD + (H + g) * HD;
    const float ka = kr[lane], kb = kr[lane + 64];
#pragma unroll
    for (int h = 0; h < REP; ++h) {
      const float d = sp[h * MAXK + j];
      o0[h] += d * ka;
      o1[h] += d * kb;
    }
  }
#pragma unroll
  for (int h = 0; h < REP; ++h) {
    part[(wave * REP + h) * HD + lane] = o0[h];
    part[(wave * REP + h) * HD + lane + 64] = o1[h];
  }
  __syncthreads();
  for (int i = t; i < REP * HD; i += 256) dq[i] = ((part[i] + part[REP * HD + i]) + part[2 * REP * HD + i]) + part[3 * REP * HD + i];
}

template <int REP>
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ dout, const int32_t* __restrict__ kstart,
                                                          const float* __restrict__ lse, const float* __restrict__ delta, float* __restrict__ dqkv,
                                                          int B, int S, int H, int G, float scale) {
  extern __shared__ float smem[];
  float* sk = smem;                                  // 128: this key
  float* sv = sk + HD;                               // 128: this value
  float* sp = sv + HD;                               // REP * MAXK: P[h][q]
  float* sds = sp + REP * MAXK;                      // REP * MAXK: dS[h][q] (scaled)
  float* part = sds + REP * MAXK;                    // 4 * 2 * 128 partial dK | dV
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int g = blockIdx.x % G;
  const long long bj = blockIdx.x / G;
  const int b = (int)(bj / S), j = (int)(bj - (long long)b * S);
  const int LD = (H + 2 * G) * HD;
  const float* base = qkv + (size_t)b * S * LD;
  float* dk = dqkv + (size_t)bj * LD + (H + g) * HD;
  float* dv = dqkv + (size_t)bj * LD + (H + G + g) * HD;
  if (j < kstart[b]) {                               // a masked (padding) key: nobody attends to it
    if (t < HD) dk[t] = 0.f, dv[t] = 0.f;
    return;
  }
  if (t < HD) sk[t] = base[(size_t)j * LD + (H + g) * HD + t], sv[t] = base[(size_t)j * LD + (H + G + g) * HD + t];
  __syncthreads();
  const int nq = S - j;                              // queries j .. S - 1 see this key
  for (int qi = t; qi < nq; qi += 256) {
    const int q = j + qi;
#pragma unroll
    for (int h = 0; h < REP; ++h) {
      const f32x4* qr = (const f32x4*)(base + (size_t)q * LD + (g * REP + h) * HD);
      const f32x4* dor = (const f32x4*)(dout + ((size_t)b * S + q) * (H * HD) + (g * REP + h) * HD);
      float a = 0.f, dp = 0.f;
#pragma unroll 4
      for (int c = 0; c < HD / 4; ++c) {
        const f32x4 qv = qr[c], dv4 = dor[c], kv = *(const f32x4*)(sk + c * 4), vv = *(const f32x4*)(sv + c * 4);
        a += kv[0] * qv[0] + kv[1] * qv[1] + kv[2] * qv[2] + kv[3] * qv[3];
        dp += vv[0] * dv4[0] + vv[1] * dv4[1] + vv[2] * dv4[2] + vv[3] * dv4[3];
      }
      const size_t sidx = ((size_t)b * H + g * REP + h) * S + q;
      const float p = expf(a * scale - lse[sidx]);
      sp[h * MAXK + qi] = p;
      sds[h * MAXK + qi] = p * (dp - delta[sidx]) * scale;
    }
  }
  __syncthreads();
  float k0 = 0.f, k1 = 0.f, v0 = 0.f, v1 = 0.f;
  const int q4 = (nq + 3) >> 2, q_lo = wave * q4, q_hi = min(nq, q_lo + q4);
  for (int qi = q_lo; qi < q_hi; ++qi) {
    const int q = j + qi;
#pragma unroll
    for (int h = 0; h < REP; ++h) {
      const float* qr = base + (size_t)q * LD + (g * REP + h) * HD;
      const float* dor = dout + ((size_t)b * S + q) * (H * HD) + (g * REP + h) * HD;
      const float p = sp[h * MAXK + qi], d = sds[h * MAXK + qi];
      k0 += d * qr[lane];
      k1 += d * qr[lane + 64];
      v0 += p * dor[lane];
      v1 += p * dor[lane + 64];
    }
  }
  part[(wave * 2 + 0) * HD + lane] = k0, part[(wave * 2 + 0) * HD + lane + 64] = k1;
  part[(wave * 2 + 1) * HD + lane] = v0, part[(wave * 2 + 1) * HD + lane + 64] = v1;
  __syncthreads();
  if (t < HD) {
    dk[t] = ((part[t] + part[2 * HD + t]) + part[4 * HD + t]) + part[6 * HD + t];
    dv[t] = ((part[HD + t] + part[3 * HD + t]) + part[5 * HD + t]) + part[7 * HD + t];
  }
}

}  // namespace tasu_f32t

using namespace tasu_f32t;

extern "C" int tasu_f32_rmsnorm_bwd(const float* dy, const float* x, const float* w, float* dx, int M, int D, float eps, int accumulate,
                                    void* stream) {
  if (!dy || !x || !w || !dx || M <= 0 || D <= 0) return TASU_ERR_ARG;
  TASU_LAUNCH(rmsnorm_bwd_kernel, dim3(M), dim3(1024), 0, (hipStream_t)stream, dy, x, w, dx, D, eps, accumulate);
  return TASU_OK;
}

extern "C" int tasu_f32_swiglu_bwd(const float* dact, const float* gu, float* dgu, int M, int I, void* stream) {
  if (!dact || !gu || !dgu || M <= 0 || I <= 0) return TASU_ERR_ARG;
  const size_t n = (size_t)M * I;
  TASU_LAUNCH(swiglu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dact, gu, dgu, M, I);
  return TASU_OK;
}

extern "C" int tasu_f32_silu(const float* x, const float* dy, float* out, int64_t n, void* stream) {
  if (!x || !out || n <= 0) return TASU_ERR_ARG;
  TASU_LAUNCH(silu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, dy, out, (size_t)n);
  return TASU_OK;
}

extern "C" int tasu_f32_colsum(const float* x, int ld, float* out, int R, int C, void* stream) {
  if (!x || !out || R <= 0 || C <= 0 || ld < C) return TASU_ERR_ARG;
  TASU_LAUNCH(colsum_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, ld, out, R, C);
  return TASU_OK;
}

extern "C" int tasu_f32_layernorm_bwd_params(const float* dy, int lddy, const float* x, int ldx, const float* mean, const float* rstd,
                                             float* dgamma, float* dbeta, int R, int D, void* stream) {
  if (!dy || !x || !mean || !rstd || !dgamma || !dbeta || R <= 0 || D <= 0) return TASU_ERR_ARG;
  TASU_LAUNCH(layernorm_bwd_params_kernel, dim3((D + 255) / 256), dim3(256), 0, (hipStream_t)stream, dy, lddy, x, ldx, mean, rstd, dgamma,
              dbeta, R, D);
  return TASU_OK;
}

extern "C" int tasu_f32_transpose(const float* src, int lds, float* dst, int ldd, int R, int C, int Rpad, void* stream) {
  if (!src || !dst || R <= 0 || C <= 0 || Rpad < R || ldd < Rpad || lds < C) return TASU_ERR_ARG;
  TASU_LAUNCH(transpose_kernel, dim3((C + 31) / 32, (Rpad + 31) / 32), dim3(256), 0, (hipStream_t)stream, src, lds, dst, ldd, R, C, Rpad);
  return TASU_OK;
}

extern "C" int tasu_f32_gather_rows(const float* dx, const int32_t* rows, float* out, int n, int D, void* stream) {
  if (!dx || !rows || !out || n <= 0 || D <= 0) return TASU_ERR_ARG;
  const size_t tot = (size_t)n * D;
  TASU_LAUNCH(gather_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dx, rows, out, n, D);
  return TASU_OK;
}

template <int REP>
static int attn_bwd_launch(const float* qkv, const float* dout, const int32_t* kstart, float* dqkv, float* lse, float* delta, int B, int S, int H,
                           int G, float scale, hipStream_t st) {
  const int lds_q = (2 * REP * HD + 2 * REP * MAXK + 4 * REP * HD + 4 * REP) * 4;
  const int lds_kv = (2 * HD + 2 * REP * MAXK + 8 * HD) * 4;
  static bool set = false;
  if (!set) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_q_kernel<REP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_q);
    (void)hipFuncSetAttribute((const void*)attn_bwd_kv_kernel<REP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
    set = true;
  }
  const unsigned grid = (unsigned)((long long)B * S * G);
  TASU_LAUNCH(attn_bwd_q_kernel<REP>, dim3(grid), dim3(256), lds_q, st, qkv, dout, kstart, dqkv, lse, delta, B, S, H, G, scale);
  TASU_LAUNCH(attn_bwd_kv_kernel<REP>, dim3(grid), dim3(256), lds_kv, st, qkv, dout, kstart, lse, delta, dqkv, B, S, H, G, scale);
  return TASU_OK;
}

extern "C" int tasu_f32_attn_bwd(const float* qkv, const float* dout, const int32_t* kstart, float* dqkv, float* lse_ws, float* delta_ws, int B,
                                 int S, int H, int G, float scale, void* stream) {
  if (!qkv || !dout || !kstart || !dqkv || !lse_ws || !delta_ws || B <= 0 || S <= 0 || S > MAXK || H <= 0 || G <= 0 || H % G) return TASU_ERR_ARG;
  // LDS: two [REP][MAXK] fp32 rows per workgroup: REP <= 8 fits (132 KiB)
  switch (H / G) {
#define BWD_CASE(R) \
  case R: return attn_bwd_launch<R>(qkv, dout, kstart, dqkv, lse_ws, delta_ws, B, S, H, G, scale, (hipStream_t)stream);
    BWD_CASE(1) BWD_CASE(2) BWD_CASE(3) BWD_CASE(4) BWD_CASE(5) BWD_CASE(6) BWD_CASE(7) BWD_CASE(8)
#undef BWD_CASE
    default: return TASU_ERR_ARG;
  }
}
