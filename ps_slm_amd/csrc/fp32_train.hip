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
// Same structure as fp32.hip's forward: phase 1 eight lanes per row of the other index (16 dims each, lane shuffles add the partial
// dots), phase 2 lanes = dims.
constexpr int MAXK = 2048;
template <int REP>
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ dout, const int32_t* __restrict__ kstart,
                                                         float* __restrict__ dqkv, float* __restrict__ lse_out, float* __restrict__ delta_out, int B,
                                                         int S, int H, int G, float scale, int kst) {
  extern __shared__ float smem[];
  float* sq = smem;                                  // REP * 128 queries
  float* sdo = sq + REP * HD;                        // REP * 128 output gradients
  float* sp = sdo + REP * HD;                        // REP * kst probabilities, then dS (kst = S rounded up to 64: LDS sized for the
  float* sdp = sp + REP * kst;                       // REP * kst dP             sequence, not for MAXK -- 117 KB held one workgroup per CU)
  float* part = sdp + REP * kst;                     // 4 * REP * 128 partial dQ
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
  // EIGHT LANES PER KEY, 16 dims each (a K / V row is 512 contiguous bytes over 8 adjacent lanes; one thread per key read 16 bytes of
  // a different row per lane), two keys per lane group and trip; the partial dots are added by lane shuffles (csrc/fp32.hip's forward)
  {
    const int seg = t & 7;
    auto dots = [&](const f32x4 (&kv)[4], const f32x4 (&vv)[4], int j) {
#pragma unroll
      for (int h = 0; h < REP; ++h) {
        float a = 0.f, dp = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 qv = *(const f32x4*)(sq + h * HD + seg * 16 + c * 4), dv = *(const f32x4*)(sdo + h * HD + seg * 16 + c * 4);
          a += kv[c][0] * qv[0] + kv[c][1] * qv[1] + kv[c][2] * qv[2] + kv[c][3] * qv[3];
          dp += vv[c][0] * dv[0] + vv[c][1] * dv[1] + vv[c][2] * dv[2] + vv[c][3] * dv[3];
        }
        a += __shfl_xor(a, 1, 64), dp += __shfl_xor(dp, 1, 64);
        a += __shfl_xor(a, 2, 64), dp += __shfl_xor(dp, 2, 64);
        a += __shfl_xor(a, 4, 64), dp += __shfl_xor(dp, 4, 64);
        if (j < nk) {
          const float sc = a * scale;
          if (seg == 0) sp[h * kst + j] = sc, sdp[h * kst + j] = dp;
          mx[h] = fmaxf(mx[h], sc);
        }
      }
    };
    for (int j0 = 0; j0 < nk; j0 += 64) {
      const int ja = j0 + (t >> 3), jb = ja + 32;
      const float* ra = base + (size_t)(k_lo + min(ja, nk - 1)) * LD;
      const float* rb = base + (size_t)(k_lo + min(jb, nk - 1)) * LD;
      f32x4 ka[4], va[4], kb[4], vb[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        ka[c] = *(const f32x4*)(ra + (H + g) * HD + seg * 16 + c * 4), va[c] = *(const f32x4*)(ra + (H + G + g) * HD + seg * 16 + c * 4);
        kb[c] = *(const f32x4*)(rb + (H + g) * HD + seg * 16 + c * 4), vb[c] = *(const f32x4*)(rb + (H + G + g) * HD + seg * 16 + c * 4);
      }
      dots(ka, va, ja);
      if (j0 + 32 < nk) dots(kb, vb, jb);                // (workgroup-uniform)
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
      const float e = expf(sp[h * kst + j] - mx[h]);
      sp[h * kst + j] = e;
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
    for (int j = t; j < nk; j += 256) sum += sp[h * kst + j] * inv[h] * sdp[h * kst + j];
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
    for (int j = t; j < nk; j += 256) sp[h * kst + j] = sp[h * kst + j] * inv[h] * (sdp[h * kst + j] - dl[h]) * scale;
  __syncthreads();
  // dQ_h = sum_j dS[h][j] K_j: this wave's quarter of the keys, lanes = dims
  float o0[REP], o1[REP];
#pragma unroll
  for (int h = 0; h < REP; ++h) o0[h] = o1[h] = 0.f;
  const int q4 = (nk + 3) >> 2, j_lo = wave * q4, j_hi = min(nk, j_lo + q4);
  constexpr int UNR = 8;                               // eight K rows in flight per trip (two dependent 4-byte loads per trip were
  int j = j_lo;                                        // what this phase waited on)
  for (; j + UNR <= j_hi; j += UNR) {
    float ka[UNR], kb[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const float* kr = base + (size_t)(k_lo + j + u) * LD + (H + g) * HD;
      ka[u] = kr[lane], kb[u] = kr[lane + 64];
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int h = 0; h < REP; ++h) {
        const float d = sp[h * kst + j + u];
        o0[h] += d * ka[u];
        o1[h] += d * kb[u];
      }
  }
  for (; j < j_hi; ++j) {
    const float* kr = base + (size_t)(k_lo + j) * LD + (H + g) * HD;
    const float ka = kr[lane], kb = kr[lane + 64];
#pragma unroll
    for (int h = 0; h < REP; ++h) {
      const float d = sp[h * kst + j];
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
                                                          int B, int S, int H, int G, float scale, int kst) {
  extern __shared__ float smem[];
  float* sk = smem;                                  // 128: this key
  float* sv = sk + HD;                               // 128: this value
  float* sp = sv + HD;                               // REP * kst: P[h][q]
  float* sds = sp + REP * kst;                       // REP * kst: dS[h][q] (scaled)
  float* part = sds + REP * kst;                     // 4 * 2 * 128 partial dK | dV
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
  // items = (query, head of the group); eight lanes per item, 16 dims each (a q / dO head row is 512 contiguous bytes over 8 lanes),
  // two items per lane group and trip; this key's / value's 16 dims of the lane stay in registers
  {
    const int seg = t & 7, items = nq * REP;
    f32x4 kseg[4], vseg[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) kseg[c] = *(const f32x4*)(sk + seg * 16 + c * 4), vseg[c] = *(const f32x4*)(sv + seg * 16 + c * 4);
    auto item = [&](const f32x4 (&qv)[4], const f32x4 (&dv4)[4], int it) {
      float a = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        a += kseg[c][0] * qv[c][0] + kseg[c][1] * qv[c][1] + kseg[c][2] * qv[c][2] + kseg[c][3] * qv[c][3];
        dp += vseg[c][0] * dv4[c][0] + vseg[c][1] * dv4[c][1] + vseg[c][2] * dv4[c][2] + vseg[c][3] * dv4[c][3];
      }
      a += __shfl_xor(a, 1, 64), dp += __shfl_xor(dp, 1, 64);
      a += __shfl_xor(a, 2, 64), dp += __shfl_xor(dp, 2, 64);
      a += __shfl_xor(a, 4, 64), dp += __shfl_xor(dp, 4, 64);
      if (it < items && seg == 0) {
        const int qi = it / REP, h = it - qi * REP;
        const size_t sidx = ((size_t)b * H + g * REP + h) * S + j + qi;
        const float p = expf(a * scale - lse[sidx]);
        sp[h * kst + qi] = p;
        sds[h * kst + qi] = p * (dp - delta[sidx]) * scale;
      }
    };
    for (int i0 = 0; i0 < items; i0 += 64) {
      const int ia = i0 + (t >> 3), ib = ia + 32;
      const int ca = min(ia, items - 1), cb = min(ib, items - 1);
      const int qa = ca / REP, ha = ca - qa * REP, qb = cb / REP, hb = cb - qb * REP;
      const float* qra = base + (size_t)(j + qa) * LD + (g * REP + ha) * HD + seg * 16;
      const float* dra = dout + ((size_t)b * S + j + qa) * (H * HD) + (g * REP + ha) * HD + seg * 16;
      const float* qrb = base + (size_t)(j + qb) * LD + (g * REP + hb) * HD + seg * 16;
      const float* drb = dout + ((size_t)b * S + j + qb) * (H * HD) + (g * REP + hb) * HD + seg * 16;
      f32x4 q1[4], d1[4], q2[4], d2[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        q1[c] = *(const f32x4*)(qra + c * 4), d1[c] = *(const f32x4*)(dra + c * 4);
        q2[c] = *(const f32x4*)(qrb + c * 4), d2[c] = *(const f32x4*)(drb + c * 4);
      }
      item(q1, d1, ia);
      if (i0 + 32 < items) item(q2, d2, ib);            // (workgroup-uniform)
    }
  }
  __syncthreads();
  float k0 = 0.f, k1 = 0.f, v0 = 0.f, v1 = 0.f;
  const int q4 = (nq + 3) >> 2, q_lo = wave * q4, q_hi = min(nq, q_lo + q4);
  int qi = q_lo;
  for (; qi + 2 <= q_hi; qi += 2) {                    // two queries (2 x REP x 4 loads) in flight per trip
    float qa[2][REP], qb[2][REP], da[2][REP], db[2][REP];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int h = 0; h < REP; ++h) {
        const float* qr = base + (size_t)(j + qi + u) * LD + (g * REP + h) * HD;
        const float* dor = dout + ((size_t)b * S + j + qi + u) * (H * HD) + (g * REP + h) * HD;
        qa[u][h] = qr[lane], qb[u][h] = qr[lane + 64], da[u][h] = dor[lane], db[u][h] = dor[lane + 64];
      }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int h = 0; h < REP; ++h) {
        const float p = sp[h * kst + qi + u], d = sds[h * kst + qi + u];
        k0 += d * qa[u][h];
        k1 += d * qb[u][h];
        v0 += p * da[u][h];
        v1 += p * db[u][h];
      }
  }
  for (; qi < q_hi; ++qi) {
    const int q = j + qi;
#pragma unroll
    for (int h = 0; h < REP; ++h) {
      const float* qr = base + (size_t)q * LD + (g * REP + h) * HD;
      const float* dor = dout + ((size_t)b * S + q) * (H * HD) + (g * REP + h) * HD;
      const float p = sp[h * kst + qi], d = sds[h * kst + qi];
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
  const int kst = (S + 63) & ~63;
  const int lds_q = (2 * REP * HD + 2 * REP * kst + 4 * REP * HD + 4 * REP) * 4;
  const int lds_kv = (2 * HD + 2 * REP * kst + 8 * HD) * 4;
  static bool set = false;
  if (!set) {                                         // (the attribute: the largest the kernels may ask for, S = MAXK)
    (void)hipFuncSetAttribute((const void*)attn_bwd_q_kernel<REP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (2 * REP * HD + 2 * REP * MAXK + 4 * REP * HD + 4 * REP) * 4);
    (void)hipFuncSetAttribute((const void*)attn_bwd_kv_kernel<REP>, hipFuncAttributeMaxDynamicSharedMemorySize, (2 * HD + 2 * REP * MAXK + 8 * HD) * 4);
    set = true;
  }
  const unsigned grid = (unsigned)((long long)B * S * G);
  TASU_LAUNCH(attn_bwd_q_kernel<REP>, dim3(grid), dim3(256), lds_q, st, qkv, dout, kstart, dqkv, lse, delta, B, S, H, G, scale, kst);
  TASU_LAUNCH(attn_bwd_kv_kernel<REP>, dim3(grid), dim3(256), lds_kv, st, qkv, dout, kstart, lse, delta, dqkv, B, S, H, G, scale, kst);
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
