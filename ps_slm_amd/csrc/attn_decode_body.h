// Body of the single-token cache attention, shared by tasu_attn_decode's kernel (decode.hip) and the persistent decode-layer
// kernel (decode_mega.hip).
#pragma once
#include "common.h"
#include "stream_body.h"

namespace tasu_attn_dec {
constexpr int HD = 128;
#ifdef TASU_ATTN_TRACE
// debug build (tools/attn_trace.py): wall-clock stamps of workgroup (0, 0) at the body's phase boundaries
__device__ unsigned long long g_attn_trace[16];
#define TASU_ATTN_STAMP(k) do { if (threadIdx.x == 0 && row == 0 && g == 0) g_attn_trace[k] = wall_clock64(); } while (0)
#else
#define TASU_ATTN_STAMP(k) do { } while (0)
#endif
// Single-token GQA attention over the cache: one 8-wave block per (row, kv group); memory-bound (every K / V byte of the
// row's cache is read once for all REP query heads of the group), so the design goal is wide loads and many of them in
// flight, not arithmetic:
//   phase 1  scores on the matrix cores: a wave takes 16 keys at a time; lane l loads 16 B of key (l & 15) for each of the
//            four 32-wide slices of the head dimension, which is exactly the B operand of mfma_f32_16x16x32_bf16
//            (B[k][n] = K[key n][dim k]); the A operand holds the REP query heads in rows 0..REP-1 (rows >= REP zero).
//            4 MFMAs per 16 keys, no shuffles; scaled scores -> LDS [REP][ctx]
//   phase 1b softmax statistics per head (wave h), probabilities (bf16-rounded like the prefill kernel) back to LDS
//   phase 2  P.V: lane owns 8 dims (one 16-B load covers them, 16 lanes a whole 256-B V row, a wave instruction 4 keys)
//            for all REP heads; 4 loads in flight; key quarters folded with two xor-shuffles, waves through LDS
//   phase 3  cross-wave sum, 1/l, bf16 store
// keys in [kstart[row], lens[row]) are visible.  ctx <= MAX_CTX.
constexpr int MAX_CTX = 2048;
constexpr int DEC_NW = 8;
// LDS floats the body needs for a context of ctx positions
__host__ __device__ constexpr int attn_decode_lds_floats(int rep, int ctx) {
  return rep * ctx + DEC_NW * rep * HD + rep + (rep & 1) + ctx;
}
// row, g: the (beam row, kv group) this workgroup serves; sp: LDS; WT: write-through output (stream_body.h)
template <int REP, bool WT>
__device__ __forceinline__ void attn_decode_body(float* sp, int row, int g, const bf16* __restrict__ qkv, const bf16* __restrict__ kc,
                                                 const bf16* __restrict__ vc, const int32_t* __restrict__ row_index,
                                                 const int32_t* __restrict__ kstart, const int32_t* __restrict__ lens,
                                                 bf16* __restrict__ out, int H, int G, int ctx, float scale, int out_frag) {
  // [REP][ctx] scores | [DEC_NW][REP][128] partial outputs | [REP] 1/l | [ctx] physical cache row of every visible key
  float* sc = sp;
  float* part = sp + (size_t)REP * ctx;
  float* linv = part + DEC_NW * REP * HD;
  int* prow = (int*)(linv + REP + (REP & 1));
  const int W = G * HD, LD = (H + 2 * G) * HD;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, lq = lane >> 4;
  TASU_ATTN_STAMP(0);
  // the row's index entries by ABSOLUTE position (all ctx of them: entries past lens[row] are never used), so that their loads
  // do not wait for kstart / lens: one memory round trip before the K / V loads instead of two
  for (int i = threadIdx.x; i < ctx; i += 64 * DEC_NW) prow[i] = row_index ? row_index[(size_t)row * ctx + i] : row;
  const int k0 = kstart[row], nk = lens[row] - k0;
  __syncthreads();
  TASU_ATTN_STAMP(1);
  prow += k0;                                              // prow[i]: physical cache row of visible key i
  // ---- V prefetch: the first PRE_IT x UN value rows of this thread's phase-2 walk are requested NOW, so that their latency
  // runs under phase 1 (K loads, score MFMAs) and the softmax instead of behind them (the phases are otherwise two dependent
  // memory round trips); contexts up to PRE_IT * 128 keys are covered entirely
  constexpr int UN = 4, PRE_IT = 3;
  const bf16* vbase = vc + (size_t)k0 * W + g * HD + l15 * 8;
  bf16x8 vpre[PRE_IT][UN];
#pragma unroll
  for (int it = 0; it < PRE_IT; ++it)
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int kcl = min(wave * 4 + it * (DEC_NW * 4 * UN) + u * DEC_NW * 4 + lq, nk - 1);
      vpre[it][u] = *(const bf16x8*)(vbase + ((size_t)prow[kcl] * ctx + kcl) * W);
    }
  // ---- phase 1: scores
  bf16x8 qf[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    qf[s4] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (l15 < REP) qf[s4] = *(const bf16x8*)(qkv + (size_t)row * LD + (g * REP + l15) * HD + s4 * 32 + lq * 8);
  }
  const bf16* kbase = kc + (size_t)k0 * W + g * HD + lq * 8;
  const int nchunk = (nk + 15) >> 4;
  // the wave's first KPRE key chunks (contexts up to KPRE * 128 keys entirely) are requested together: one round trip, not one
  // per chunk
  constexpr int KPRE = 3;
  bf16x8 kpre[KPRE][4];
#pragma unroll
  for (int it = 0; it < KPRE; ++it) {
    const int kcl = min((wave + it * DEC_NW) * 16 + l15, nk - 1);
    const bf16* kp = kbase + ((size_t)prow[kcl] * ctx + kcl) * W;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) kpre[it][s4] = *(const bf16x8*)(kp + s4 * 32);
  }
  TASU_ATTN_STAMP(2);
  auto scores = [&](const bf16x8 (&kf)[4], int c) {
    const int key = c * 16 + l15;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) acc = mfma16(qf[s4], kf[s4], acc);
    // acc[r] = score(head lq*4 + r, key c*16 + l15)
    if (key < nk) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (lq * 4 + r < REP) sc[(lq * 4 + r) * ctx + key] = acc[r] * scale;
    }
  };
#pragma unroll
  for (int it = 0; it < KPRE; ++it)
    if (wave + it * DEC_NW < nchunk) scores(kpre[it], wave + it * DEC_NW);
  for (int c = wave + KPRE * DEC_NW; c < nchunk; c += DEC_NW) {
    const int kcl = min(c * 16 + l15, nk - 1);
    const bf16* kp = kbase + ((size_t)prow[kcl] * ctx + kcl) * W;
    bf16x8 kf[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) kf[s4] = *(const bf16x8*)(kp + s4 * 32);
    scores(kf, c);
  }
  TASU_ATTN_STAMP(3);
  __syncthreads();
  TASU_ATTN_STAMP(4);
  // ---- phase 1b: softmax statistics of head h
  for (int h = wave; h < REP; h += DEC_NW) {
    float m = -__builtin_inff();
    for (int i = lane; i < nk; i += 64) m = fmaxf(m, sc[h * ctx + i]);
    m = wave_max(m);
    float l = 0.f;
    for (int i = lane; i < nk; i += 64) {
      const float p = __expf(sc[h * ctx + i] - m);
      sc[h * ctx + i] = (float)(bf16)p;
      l += p;
    }
    l = wave_sum(l);
    if (lane == 0) linv[h] = l > 0.f ? 1.f / l : 0.f;
  }
  __syncthreads();
  TASU_ATTN_STAMP(5);
  // ---- phase 2: P.V   (lane: dims 8*l15 .. +7, key quarter lq; wave: keys wave*4 + lq, stride 32)
  float o[REP][8];
#pragma unroll
  for (int h = 0; h < REP; ++h)
#pragma unroll
    for (int j = 0; j < 8; ++j) o[h][j] = 0.f;
  auto accumulate = [&](const bf16x8 (&v)[UN], int i0) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int key = i0 + u * DEC_NW * 4 + lq;
      if (key < nk) {
#pragma unroll
        for (int h = 0; h < REP; ++h) {
          const float p = sc[h * ctx + key];
#pragma unroll
          for (int j = 0; j < 8; ++j) o[h][j] += p * (float)v[u][j];
        }
      }
    }
  };
#pragma unroll
  for (int it = 0; it < PRE_IT; ++it)
    if (wave * 4 + it * (DEC_NW * 4 * UN) < nk) accumulate(vpre[it], wave * 4 + it * (DEC_NW * 4 * UN));
  for (int i0 = wave * 4 + PRE_IT * (DEC_NW * 4 * UN); i0 < nk; i0 += DEC_NW * 4 * UN) {
    bf16x8 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int kcl = min(i0 + u * DEC_NW * 4 + lq, nk - 1);
      v[u] = *(const bf16x8*)(vbase + ((size_t)prow[kcl] * ctx + kcl) * W);
    }
    accumulate(v, i0);
  }
#pragma unroll
  for (int h = 0; h < REP; ++h)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float x = o[h][j];
      x += __shfl_xor(x, 16, 64);
      x += __shfl_xor(x, 32, 64);
      o[h][j] = x;
    }
  TASU_ATTN_STAMP(6);
  if (lq == 0) {
#pragma unroll
    for (int h = 0; h < REP; ++h) {
      float* dst = part + ((wave * REP + h) * HD) + l15 * 8;
      *(f32x4*)dst = f32x4{o[h][0], o[h][1], o[h][2], o[h][3]};
      *(f32x4*)(dst + 4) = f32x4{o[h][4], o[h][5], o[h][6], o[h][7]};
    }
  }
  __syncthreads();
  TASU_ATTN_STAMP(7);
  for (int e = threadIdx.x; e < REP * HD; e += 64 * DEC_NW) {
    const int h = e / HD, d = e - h * HD;
    float s = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < DEC_NW; ++w2) s += part[(w2 * REP + h) * HD + d];
    const int n = (g * REP + h) * HD + d;                  // column of the [M, H * 128] attention output
    // out_frag: the o projection's A operand in fragment order (csrc/gemm_stream.hip); row = row % 64 of its 64-row chunk
    const size_t o = out_frag ? ((size_t)(row >> 6) * 64 * (H * HD)) +
                                    ((((size_t)(n >> 5) * 4 + ((row & 63) >> 4)) * 64 + ((n & 31) >> 3) * 16 + (row & 15)) << 3) + (n & 7)
                              : (size_t)row * (H * HD) + n;
    tasu_stream::st_out<WT>(out + o, (bf16)(s * linv[h]));
  }
  TASU_ATTN_STAMP(8);
}
}  // namespace tasu_attn_dec
