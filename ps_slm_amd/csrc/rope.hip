// Rotary position embedding for the fused qkv activation + the transposed operand copies that the attention
// kernels stream.  Reference arithmetic: transformers modeling_qwen2.py:91-135 (fp32 cos/sin from position_ids,
// rotate-half, result cast to the attention dtype).  HBM-bound; one 64-token x 128-d tile per block, staged
// through LDS so that both the in-place row write and the [128, S] transposed write are coalesced.
#include "attn_tiles.h"
#include "../../include/tasu_hip.h"

namespace {
constexpr int HD = 128;

__global__ void rope_table_kernel(const int32_t* __restrict__ pos, float* __restrict__ ct, float* __restrict__ st, int M,
                                  int half, float theta) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * half) return;
  const int m = idx / half, i = idx - m * half;
  const float inv = 1.0f / powf(theta, (float)(2 * i) / (float)(2 * half));
  const float ang = (float)pos[m] * inv;
  float s, c;
  sincosf(ang, &s, &c);
  ct[idx] = c;
  st[idx] = s;
}

// grid (ceil(S/64), H+2G, B)
__global__ __launch_bounds__(256) void rope_fwd_kernel(bf16* __restrict__ qkv, const float* __restrict__ ct,
                                                       const float* __restrict__ st, bf16* __restrict__ qt,
                                                       bf16* __restrict__ kt, bf16* __restrict__ vt, int S, int Spad, int H,
                                                       int G) {
  __shared__ bf16 tile[64][HD + 2];
  const int t0 = blockIdx.x * 64, hh = blockIdx.y, b = blockIdx.z;
  const int LD = (H + 2 * G) * HD;
  const int tl = threadIdx.x >> 2, part = threadIdx.x & 3;
  const int tok = t0 + tl;
  const bool rotate = hh < H + G;
  if (tok < S) {
    bf16* row = qkv + ((size_t)b * S + tok) * LD + hh * HD;
    bf16x8 a0 = *(const bf16x8*)(row + part * 16), a1 = *(const bf16x8*)(row + part * 16 + 8);
    bf16x8 b0 = *(const bf16x8*)(row + 64 + part * 16), b1 = *(const bf16x8*)(row + 64 + part * 16 + 8);
    if (rotate) {
      const float* cr = ct + ((size_t)b * S + tok) * 64 + part * 16;
      const float* sr = st + ((size_t)b * S + tok) * 64 + part * 16;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float c0 = cr[j], s0 = sr[j], c1 = cr[8 + j], s1 = sr[8 + j];
        float r1, r2, r3, r4;
        rope_pair_f((float)a0[j], (float)b0[j], c0, s0, r1, r2);
        rope_pair_f((float)a1[j], (float)b1[j], c1, s1, r3, r4);
        a0[j] = (bf16)r1;
        b0[j] = (bf16)r2;
        a1[j] = (bf16)r3;
        b1[j] = (bf16)r4;
      }
      *(bf16x8*)(row + part * 16) = a0;
      *(bf16x8*)(row + part * 16 + 8) = a1;
      *(bf16x8*)(row + 64 + part * 16) = b0;
      *(bf16x8*)(row + 64 + part * 16 + 8) = b1;
    }
    if (!qt && !kt && !vt) return;                       // in-place rotation only (the attention kernels transpose in LDS)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      tile[tl][part * 16 + j] = a0[j];
      tile[tl][part * 16 + 8 + j] = a1[j];
      tile[tl][64 + part * 16 + j] = b0[j];
      tile[tl][64 + part * 16 + 8 + j] = b1[j];
    }
  } else {
    if (!qt && !kt && !vt) return;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      tile[tl][part * 16 + j] = (bf16)0.f;
      tile[tl][64 + part * 16 + j] = (bf16)0.f;
    }
  }
  bf16* dstbase;
  if (hh < H) {
    if (!qt) return;
    dstbase = qt + ((size_t)b * H + hh) * HD * Spad;
  } else if (hh < H + G) {
    if (!kt) return;
    dstbase = kt + ((size_t)b * G + (hh - H)) * HD * Spad;
  } else {
    if (!vt) return;
    dstbase = vt + ((size_t)b * G + (hh - H - G)) * HD * Spad;
  }
  __syncthreads();
  const int d = threadIdx.x >> 1, half = threadIdx.x & 1;
  bf16* dst = dstbase + (size_t)d * Spad + t0 + half * 32;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = tile[half * 32 + c * 8 + j][d];
    *(bf16x8*)(dst + c * 8) = v;
  }
}

// Backward: q block of dqkv holds dQ (rotated space, bf16) -> un-rotate in place.  k/v blocks are produced from
// the per-query-head fp32 partials: sum over the H/G heads of the group, un-rotate K, round to bf16.
// grid (ceil(M / TOK)), block 256: (H + 2G) * 8 threads per token, each owning the 8-wide chunk pair (x[c..c+7],
// x[64+c..64+c+7]) of one head: 16-byte accesses throughout (one lane per rotation pair and one 64-thread block per head
// moved 50 MB at 2.9 TB/s).
__global__ __launch_bounds__(256) void rope_bwd_kernel(bf16* __restrict__ dqkv, const float* __restrict__ dk_part,
                                                      const float* __restrict__ dv_part, const float* __restrict__ ct,
                                                      const float* __restrict__ st, int M, int H, int G) {
  const int upt = (H + 2 * G) * 8;                       // units per token
  const int tpb = upt <= 128 ? 256 / upt : 1;            // tokens per block
  const int tpt = 256 / tpb;                             // threads per token
  const int tl = threadIdx.x / tpt;
  const int m = blockIdx.x * tpb + tl;
  if (tl >= tpb || m >= M) return;
  const int LD = (H + 2 * G) * HD;
  const int np_g = (H / G) / TASU_ATTN_DKV_HPB(H / G);   // fp32 partials per kv head written by tasu_attn_bwd_dkv
  const int np = H / TASU_ATTN_DKV_HPB(H / G);
  for (int u = threadIdx.x - tl * tpt; u < upt; u += tpt) {
  const int hh = u >> 3, c = (u & 7) * 8;
  bf16* row = dqkv + (size_t)m * LD + hh * HD;
  float y1[8], y2[8];
  if (hh < H) {
    const bf16x8 lo = *(const bf16x8*)(row + c), hi = *(const bf16x8*)(row + 64 + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      y1[j] = (float)lo[j];
      y2[j] = (float)hi[j];
    }
  } else {
    const bool isk = hh < H + G;
    const int g = isk ? hh - H : hh - H - G;
    const float* src = (isk ? dk_part : dv_part) + (size_t)m * (np * HD) + (size_t)g * np_g * HD;
#pragma unroll
    for (int j = 0; j < 8; ++j) y1[j] = y2[j] = 0.f;
    for (int r = 0; r < np_g; ++r) {
      const f32x4 a0 = *(const f32x4*)(src + r * HD + c), a1 = *(const f32x4*)(src + r * HD + c + 4);
      const f32x4 b0 = *(const f32x4*)(src + r * HD + 64 + c), b1 = *(const f32x4*)(src + r * HD + 64 + c + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        y1[j] += a0[j];
        y1[4 + j] += a1[j];
        y2[j] += b0[j];
        y2[4 + j] += b1[j];
      }
    }
    if (!isk) {
      bf16x8 lo, hi;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        lo[j] = (bf16)y1[j];
        hi[j] = (bf16)y2[j];
      }
      *(bf16x8*)(row + c) = lo;
      *(bf16x8*)(row + 64 + c) = hi;
      continue;
    }
  }
  const f32x4 c0 = *(const f32x4*)(ct + (size_t)m * 64 + c), c1 = *(const f32x4*)(ct + (size_t)m * 64 + c + 4);
  const f32x4 s0 = *(const f32x4*)(st + (size_t)m * 64 + c), s1 = *(const f32x4*)(st + (size_t)m * 64 + c + 4);
  // forward: y1 = x1 c - x2 s ; y2 = x2 c + x1 s   =>   dx1 = dy1 c + dy2 s ; dx2 = dy2 c - dy1 s
  bf16x8 lo, hi;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float cs = j < 4 ? c0[j] : c1[j - 4], sn = j < 4 ? s0[j] : s1[j - 4];
    float d1, d2;
    tasu_attn::rope_pair_bwd_f(y1[j], y2[j], cs, sn, d1, d2);
    lo[j] = (bf16)d1;
    hi[j] = (bf16)d2;
  }
  *(bf16x8*)(row + c) = lo;
  *(bf16x8*)(row + 64 + c) = hi;
  }
}
}  // namespace

extern "C" int tasu_rope_table(const int32_t* pos, float* cos_tab, float* sin_tab, int M, int head_dim, float theta,
                               void* stream) {
  if (!pos || !cos_tab || !sin_tab || M <= 0 || head_dim != HD) return TASU_ERR_ARG;
  const int n = M * (HD / 2);
  TASU_LAUNCH(rope_table_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pos, cos_tab, sin_tab,
                     M, HD / 2, theta);
  return TASU_OK;
}

extern "C" int tasu_rope_fwd(void* qkv, const float* cos_tab, const float* sin_tab, void* qt, void* kt, void* vt, int B,
                             int S, int H, int G, void* stream) {
  if (!qkv || !cos_tab || !sin_tab || B <= 0 || S <= 0 || H <= 0 || G <= 0) return TASU_ERR_ARG;
  // without transposed copies only the q and k heads have work (the v heads are not rotated)
  dim3 grid((S + 63) / 64, (qt || kt || vt) ? H + 2 * G : H + G, B);
  TASU_LAUNCH(rope_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, (bf16*)qkv, cos_tab, sin_tab, (bf16*)qt,
                     (bf16*)kt, (bf16*)vt, S, (S + 63) & ~63, H, G);
  return TASU_OK;
}

extern "C" int tasu_rope_bwd(void* dqkv, const float* dk_part, const float* dv_part, const float* cos_tab,
                             const float* sin_tab, int B, int S, int H, int G, void* stream) {
  if (!dqkv || !dk_part || !dv_part || !cos_tab || !sin_tab || B <= 0 || S <= 0 || H <= 0 || G <= 0 || H % G)
    return TASU_ERR_ARG;
  const int upt = (H + 2 * G) * 8;
  const int tpb = upt <= 128 ? 256 / upt : 1, M = B * S;
  TASU_LAUNCH(rope_bwd_kernel, dim3((M + tpb - 1) / tpb), dim3(256), 0, (hipStream_t)stream, (bf16*)dqkv, dk_part, dv_part,
              cos_tab, sin_tab, M, H, G);
  return TASU_OK;
}
