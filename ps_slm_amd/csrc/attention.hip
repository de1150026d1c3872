// Flash-style attention for gfx950, head_dim 128, bf16 I/O, fp32 online softmax, MFMA 16x16x32.
// Replaces the SDPA call inside Qwen2Attention.forward (transformers modeling_qwen2.py:150-172, causal GQA with
// key padding) and, with causal = 0, the softmax(QK^T)V of MultiHeadedAttentionSANM (SenseVoice.py:171-207).
//
// Design ("query index stays on lane&15"): every product is issued so that the index that needs row
// statistics (the query for fwd/dq, the key for dk/dv) is the MFMA *column* (lane & 15), and the reduced
// index of the NEXT product is what the accumulator registers enumerate.  A 16x16 score tile then feeds the
// following MFMA as a register operand with no LDS round trip: the two 16-wide score tiles of a 32-deep
// k-step are packed as k-slots {4q'+r | 16+4q'+r}, and the other operand is read from LDS in that same
// permuted order.  Operands whose reduced index is the TOKEN index (V^T for P.V, K^T for dQ, Q^T / dO^T for dK / dV)
// come out of the SAME token-major LDS tile the row reads use, through gfx950's hardware transpose read
// (ds_read_b64_tr_b16: a 16-lane group reads a 4-token x 16-d block and every lane receives one d column of it) --
// round 1 streamed pre-transposed [128, S] global copies of Q, K, V and dO into extra LDS tiles instead, which doubled the
// bytes a dK/dV block stages per step (64 KB) in a loop that is bound by exactly that staging latency.
//
// Tiles: 64 queries x 64 keys per step, 256 threads = 4 waves, each wave owns 16 of the 64 rows.
#include <stdlib.h>

#include "attn_tiles.h"
#include "../../include/tasu_hip.h"

namespace {

using namespace tasu_attn;


// ======================================================================================= forward
// QW = 16-query sub-tiles per wave (block = 4 waves = 64 * QW queries).  The loop is LDS-read bound, not MFMA bound: with one
// sub-tile per wave every K / V^T fragment read from LDS feeds ONE MFMA (1 KiB of LDS per 16 MFMA cycles, four waves on a
// 128 B/clk LDS: twice the MFMA time).  With QW = 2 each fragment feeds two MFMAs (the wave's two query sub-tiles), halving
// the LDS traffic per FLOP.
template <int QW>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ vt,
                                                          const uint8_t* __restrict__ kmask, bf16* __restrict__ out,
                                                          float* __restrict__ lse, int S, int Spad, int H, int G,
                                                          float scale, int causal) {
  __shared__ __attribute__((aligned(16))) char smem[2 * ROW_TILE_BYTES + 256];
  char* sK = smem;
  char* sV = smem + ROW_TILE_BYTES;
  float* sBias = (float*)(smem + 2 * ROW_TILE_BYTES);   // additive key bias of the staged tile (attn_tiles.h)
  const float scale2 = scale * LOG2E;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // grid (H, B, query tiles): the late query tiles, which meet the most key tiles under the causal mask, are dispatched
  // first, so that a multi-round grid ends on the short ones
  const int qt = (int)gridDim.z - 1 - (int)blockIdx.z, h = blockIdx.x, b = blockIdx.y;
  const int g = h / (H / G);
  const int LD = (H + 2 * G) * HD;
  const bf16* qbase = qkv + (size_t)b * S * LD + h * HD;
  const bf16* kbase = qkv + (size_t)b * S * LD + (H + g) * HD;
  const bf16* vbase = qkv + (size_t)b * S * LD + (H + G + g) * HD;
  const uint8_t* mrow = kmask + (size_t)b * Spad;

  const int q0 = qt * 64 * QW + wave * 16 * QW;        // first query of this wave
  const int qp = lane >> 4;
  int qpos[QW];
  bf16x8 qf[QW][4];
#pragma unroll
  for (int u = 0; u < QW; ++u) {
    qpos[u] = q0 + u * 16 + (lane & 15);
    load_row_frags(qf[u], qbase, LD, min(qpos[u], S - 1), lane);
  }

  f32x4 o[QW][8];
  float m_run[QW], l_run[QW];
#pragma unroll
  for (int u = 0; u < QW; ++u) {
#pragma unroll
    for (int i = 0; i < 8; ++i) o[u][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    m_run[u] = NEG_INF;
    l_run[u] = 0.f;
  }

  const int nkt_all = (S + 63) >> 6;
  const int nkt = causal ? min(nkt_all, (qt + 1) * QW) : nkt_all;
  TileRegs rK, rVt;
  uint8_t mb_next = 0;                            // threads 0..63: the key-mask byte of the tile's key, prefetched with the tile
  fetch_row_tile(rK, kbase, LD, 0, S);
  fetch_row_tile(rVt, vbase, LD, 0, S);
  if (threadIdx.x < 64) mb_next = mrow[threadIdx.x];
  int qfirst[QW];                                 // first query of the unit (wave-uniform): tiles whose keys all precede it need no causal compare
#pragma unroll
  for (int u = 0; u < QW; ++u) qfirst[u] = q0 + u * 16;
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    commit_row_tile(sK, rK);
    commit_row_tile(sV, rVt);
    if (threadIdx.x < 64) sBias[threadIdx.x] = mask_bias(mb_next);
    __syncthreads();
    if (kt + 1 < nkt) {
      fetch_row_tile(rK, kbase, LD, (kt + 1) * 64, S);
      fetch_row_tile(rVt, vbase, LD, (kt + 1) * 64, S);
      if (threadIdx.x < 64) mb_next = mrow[(kt + 1) * 64 + threadIdx.x];
    }

    f32x4 s[QW][4];
    float tmax[QW];
#pragma unroll
    for (int u = 0; u < QW; ++u) tmax[u] = NEG_INF;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      f32x4 a[QW];
#pragma unroll
      for (int u = 0; u < QW; ++u) a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 kfrag = frag_row(sK, st, ks, lane);          // one LDS read, QW MFMAs
#pragma unroll
        for (int u = 0; u < QW; ++u) a[u] = mfma16(kfrag, qf[u][ks], a[u]);
      }
      const int key0 = kt * 64 + st * 16 + 4 * qp;
      const f32x4 kb = *(const f32x4*)(sBias + st * 16 + 4 * qp);
#pragma unroll
      for (int u = 0; u < QW; ++u) {
#pragma unroll
        for (int r = 0; r < 4; ++r) a[u][r] = __builtin_fmaf(a[u][r], scale2, kb[r]);
        if (causal && kt * 64 + 63 > qfirst[u]) {        // the tile reaches past the unit's first query: per-element compare
#pragma unroll
          for (int r = 0; r < 4; ++r) a[u][r] = key0 + r <= qpos[u] ? a[u][r] : NEG_INF;
        }
        tmax[u] = fmaxf(tmax[u], fmaxf(fmaxf(a[u][0], a[u][1]), fmaxf(a[u][2], a[u][3])));
        s[u][st] = a[u];
      }
    }
    bf16x8 pf0[QW], pf1[QW];
#pragma unroll
    for (int u = 0; u < QW; ++u) {
      float tm = fmaxf(tmax[u], __shfl_xor(tmax[u], 16, 64));
      tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
      const float m_new = fmaxf(m_run[u], tm);
      const float m_use = (m_new == NEG_INF) ? 0.f : m_new;
      const float alpha = exp2_fast(m_run[u] - m_use);  // m_run = -inf -> 0
      float psum = 0.f;
#pragma unroll
      for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = exp2_fast(s[u][st][r] - m_use);
          s[u][st][r] = p;
          psum += p;
        }
      l_run[u] = l_run[u] * alpha + psum;
      m_run[u] = m_new;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) o[u][nt] *= alpha;
      pf0[u] = pack_pair(s[u][0], s[u][1]);
      pf1[u] = pack_pair(s[u][2], s[u][3]);
    }
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const bf16x8 v0 = frag_tr_row(sV, nt, 0, lane), v1 = frag_tr_row(sV, nt, 1, lane);   // read once, used by QW sub-tiles
#pragma unroll
      for (int u = 0; u < QW; ++u) {
        o[u][nt] = mfma16(v0, pf0[u], o[u][nt]);
        o[u][nt] = mfma16(v1, pf1[u], o[u][nt]);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < QW; ++u) {
    float l_tot = l_run[u] + __shfl_xor(l_run[u], 16, 64);
    l_tot += __shfl_xor(l_tot, 32, 64);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    if (qpos[u] < S) {
      bf16* orow = out + ((size_t)b * S + qpos[u]) * (H * HD) + h * HD;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const f32x4 v = o[u][nt] * inv;
        *(bf16x4*)(orow + nt * 16 + 4 * qp) = __builtin_convertvector(v, bf16x4);
      }
      if (qp == 0) lse[((size_t)b * H + h) * Spad + qpos[u]] = l_tot > 0.f ? (m_run[u] + __log2f(l_tot)) * LN2 : 0.f;   // (base-2 running max)
    }
  }
}

// =============================================================================== backward: prep
// delta[b,h,s] = sum_d dO*O ; dOt[b,h,d,s] = dO[b,s,h,d].  One block per (64 tokens, head, batch).
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const bf16* __restrict__ dout, const bf16* __restrict__ out,
                                                            float* __restrict__ delta, bf16* __restrict__ dout_t, int S,
                                                            int Spad, int H) {
  __shared__ bf16 tile[64][HD + 2];
  const int t0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const int tl = threadIdx.x >> 2, part = threadIdx.x & 3;  // 4 threads per token, 32 d each
  const int tok = t0 + tl;
  float acc = 0.f;
  if (tok < S) {
    const bf16* dr = dout + ((size_t)b * S + tok) * (H * HD) + h * HD + part * 32;
    const bf16* orr = out + ((size_t)b * S + tok) * (H * HD) + h * HD + part * 32;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bf16x8 dv = *(const bf16x8*)(dr + c * 8);
      const bf16x8 ov = *(const bf16x8*)(orr + c * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc += (float)dv[j] * (float)ov[j];
        tile[tl][part * 32 + c * 8 + j] = dv[j];
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 32; ++j) tile[tl][part * 32 + j] = (bf16)0.f;
  }
  acc += __shfl_xor(acc, 1, 64);
  acc += __shfl_xor(acc, 2, 64);
  // rows in [S, Spad) get an explicit 0: the dK/dV kernels read delta for the whole padded tile, and the buffer comes from an
  // uninitialised allocation (0 * NaN would poison dK)
  if (part == 0 && tok < Spad) delta[((size_t)b * H + h) * Spad + tok] = tok < S ? acc : 0.f;
  if (!dout_t) return;                                   // the backward kernels transpose in LDS (frag_tr_row): delta only
  __syncthreads();
  // transposed write: thread -> d = tid>>1, 32 tokens
  const int d = threadIdx.x >> 1, half = threadIdx.x & 1;
  bf16* dst = dout_t + (((size_t)b * H + h) * HD + d) * Spad + t0 + half * 32;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = tile[half * 32 + c * 8 + j][d];
    *(bf16x8*)(dst + c * 8) = v;
  }
}

// =============================================================================== backward: dQ
constexpr int DQ_LDS = 2 * ROW_TILE_BYTES + 768;      // K and V tiles + (at + 512, behind the dK / dV role's floats) the tile's key bias
__device__ __forceinline__ void attn_bwd_dq_body(const bf16* __restrict__ qkv, const bf16* __restrict__ kt_g,
                                                 const uint8_t* __restrict__ kmask, const bf16* __restrict__ dout,
                                                 const float* __restrict__ lse, const float* __restrict__ delta,
                                                 bf16* __restrict__ dqkv, int S, int Spad, int H, int G, float scale,
                                                 int causal, int bx, int by, int bz, char* smem) {
  char* sK = smem;
  char* sV = smem + ROW_TILE_BYTES;
  float* sBias = (float*)(smem + 2 * ROW_TILE_BYTES + 512);   // (behind the dK / dV role's lse / delta floats of the merged kernel)
  const float scale2 = scale * LOG2E;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qt = bx, h = by, b = bz;
  const int g = h / (H / G);
  const int LD = (H + 2 * G) * HD;
  const bf16* qbase = qkv + (size_t)b * S * LD + h * HD;
  const bf16* kbase = qkv + (size_t)b * S * LD + (H + g) * HD;
  const bf16* vbase = qkv + (size_t)b * S * LD + (H + G + g) * HD;
  const uint8_t* mrow = kmask + (size_t)b * Spad;
  const int qpos = qt * 64 + wave * 16 + (lane & 15);
  const int qc = min(qpos, S - 1);
  const int qp = lane >> 4;
  bf16x8 qf[4], dof[4];
  load_row_frags(qf, qbase, LD, qc, lane);
  load_row_frags(dof, dout + (size_t)b * S * (H * HD) + h * HD, H * HD, qc, lane);
  const float lse2_q = lse[((size_t)b * H + h) * Spad + qc] * LOG2E;
  const float dl_q = delta[((size_t)b * H + h) * Spad + qc];

  f32x4 dq[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) dq[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt = causal ? (qt + 1) : ((S + 63) >> 6);
  TileRegs rK, rV;
  uint8_t mb_next = 0;
  fetch_row_tile(rK, kbase, LD, 0, S);
  fetch_row_tile(rV, vbase, LD, 0, S);
  if (threadIdx.x < 64) mb_next = mrow[threadIdx.x];
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    commit_row_tile(sK, rK);
    commit_row_tile(sV, rV);
    if (threadIdx.x < 64) sBias[threadIdx.x] = mask_bias(mb_next);
    __syncthreads();
    if (kt + 1 < nkt) {
      fetch_row_tile(rK, kbase, LD, (kt + 1) * 64, S);
      fetch_row_tile(rV, vbase, LD, (kt + 1) * 64, S);
      if (threadIdx.x < 64) mb_next = mrow[(kt + 1) * 64 + threadIdx.x];
    }
    const bool diag = causal && kt == qt;                 // the only key tile that reaches past the block's first query
    f32x4 ds[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        a = mfma16(frag_row(sK, st, ks, lane), qf[ks], a);
        dp = mfma16(frag_row(sV, st, ks, lane), dof[ks], dp);
      }
      const int key0 = kt * 64 + st * 16 + 4 * qp;
      const f32x4 kb = *(const f32x4*)(sBias + st * 16 + 4 * qp);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = prob2(a[r], scale2, kb[r] - lse2_q);
        if (diag) p = key0 + r <= qpos ? p : 0.f;
        ds[st][r] = p * (dp[r] - dl_q);
      }
    }
    const bf16x8 f0 = pack_pair(ds[0], ds[1]);
    const bf16x8 f1 = pack_pair(ds[2], ds[3]);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      dq[nt] = mfma16(frag_tr_row(sK, nt, 0, lane), f0, dq[nt]);      // K^T out of the token-major K tile
      dq[nt] = mfma16(frag_tr_row(sK, nt, 1, lane), f1, dq[nt]);
    }
  }
  if (qpos < S) {
    bf16* drow = dqkv + ((size_t)b * S + qpos) * LD + h * HD;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const f32x4 v = dq[nt] * scale;
      *(bf16x4*)(drow + nt * 16 + 4 * qp) = __builtin_convertvector(v, bf16x4);
    }
  }
}

// =============================================================================== backward: dK, dV
// One block per (64 keys, group of HPB query heads that share a kv head, batch): K / V fragments stay in registers,
// dK^T / dV^T accumulate in registers over the HPB heads x the (causal) query tiles, and only H/HPB fp32 partials
// per kv head go to memory (HPB = TASU_ATTN_DKV_HPB(H/G): 3 for Qwen2.5-1.5B -> 2 partials per kv head; writing one
// partial per QUERY head cost 50 MB of fp32 stores per call and dominated the kernel).  tasu_rope_bwd sums them.
constexpr int DKV_LDS = 2 * ROW_TILE_BYTES + 128 * 4 + 256;   static_assert(DKV_LDS == DQ_LDS, "the merged kernel's two roles share one LDS layout");   // Q and dO tiles + lse[64] + delta[64] (+ the dQ role's key bias row)
__device__ __forceinline__ void attn_bwd_dkv_body(const bf16* __restrict__ qkv, const bf16* __restrict__ qt_g,
                                                  const uint8_t* __restrict__ kmask, const bf16* __restrict__ dout,
                                                  const bf16* __restrict__ dout_t, const float* __restrict__ lse,
                                                  const float* __restrict__ delta, float* __restrict__ dk_part,
                                                  float* __restrict__ dv_part, int S, int Spad, int H, int G, int hpb,
                                                  float scale, int causal, int bx, int by, int bz, char* smem) {
  char* sQ = smem;
  char* sdO = smem + ROW_TILE_BYTES;
  float* s_ld = (float*)(smem + 2 * ROW_TILE_BYTES);   // [0,64) lse, [64,128) delta of the q tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ktile = bx, hg = by, b = bz;      // hg: index of the HPB-head group
  const int h0 = hg * hpb;
  const int g = h0 / (H / G);
  const int LD = (H + 2 * G) * HD;
  const bf16* kbase = qkv + (size_t)b * S * LD + (H + g) * HD;
  const bf16* vbase = qkv + (size_t)b * S * LD + (H + G + g) * HD;
  const int kpos = ktile * 64 + wave * 16 + (lane & 15);
  const int kc = min(kpos, S - 1);
  const int qp = lane >> 4;
  const bool kvalid = kpos < S && kmask[(size_t)b * Spad + kc] != 0;
  const float kbias = kvalid ? 0.f : NEG_INF;
  const float scale2 = scale * LOG2E;
  bf16x8 kf[4], vf[4];
  load_row_frags(kf, kbase, LD, kc, lane);
  load_row_frags(vf, vbase, LD, kc, lane);

  f32x4 dk[8], dv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    dk[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    dv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int nqt = (S + 63) >> 6;
  const int q_first = causal ? ktile : 0;
  const int per_head = nqt - q_first;
  const int n_it = per_head * hpb;                       // flattened (head, query tile) iteration space
  float r_ld = 0.f;                                      // threads 0..63: lse[q], 64..127: delta[q] of the fetched tile
  auto fetch = [&](TileRegs& rQ, TileRegs& rdO, int it) {
    const int h = h0 + it / per_head, qtile = q_first + it % per_head;
    fetch_row_tile(rQ, qkv + (size_t)b * S * LD + h * HD, LD, qtile * 64, S);
    fetch_row_tile(rdO, dout + (size_t)b * S * (H * HD) + h * HD, H * HD, qtile * 64, S);
    if (threadIdx.x < 128) {
      r_ld = (threadIdx.x < 64 ? lse : delta)[((size_t)b * H + h) * Spad + qtile * 64 + (threadIdx.x & 63)];
      // lse in the base-2 domain; +inf for query rows past the sequence (their probabilities are exp2(-inf) = 0)
      // ... and delta 0 there: p = 0 times a stale NaN would still be NaN
      if (threadIdx.x < 64) r_ld = qtile * 64 + (int)threadIdx.x < S ? r_ld * LOG2E : __builtin_inff();
      else r_ld = qtile * 64 + (int)(threadIdx.x & 63) < S ? r_ld : 0.f;
    }
  };
  TileRegs rQ, rdO;
  if (n_it > 0) fetch(rQ, rdO, 0);
  for (int it = 0; it < n_it; ++it) {
    const int qtile = q_first + it % per_head;
    __syncthreads();
    commit_row_tile(sQ, rQ);
    commit_row_tile(sdO, rdO);
    if (threadIdx.x < 128) s_ld[threadIdx.x] = r_ld;
    __syncthreads();
    if (it + 1 < n_it) fetch(rQ, rdO, it + 1);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {  // 32 query rows at a time keeps the live set small
      f32x4 pv[2], ds[2];
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        const int qs = qb * 2 + q2;
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          a = mfma16(frag_row(sQ, qs, ks, lane), kf[ks], a);
          dp = mfma16(frag_row(sdO, qs, ks, lane), vf[ks], dp);
        }
        const int q0 = qtile * 64 + qs * 16 + 4 * qp;  // < Spad
        const f32x4 l4 = *(const f32x4*)(s_ld + qs * 16 + 4 * qp);
        const f32x4 d4 = *(const f32x4*)(s_ld + 64 + qs * 16 + 4 * qp);
        const bool diag = causal && qtile == ktile;     // the only query tile that does not lie wholly behind the keys
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = prob2(a[r], scale2, kbias - l4[r]);
          if (diag) p = kpos <= q0 + r ? p : 0.f;
          pv[q2][r] = p;
          ds[q2][r] = p * (dp[r] - d4[r]);
        }
      }
      const bf16x8 pf = pack_pair(pv[0], pv[1]);
      const bf16x8 sf = pack_pair(ds[0], ds[1]);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        dv[nt] = mfma16(frag_tr_row(sdO, nt, qb, lane), pf, dv[nt]);     // dO^T, Q^T out of the token-major tiles
        dk[nt] = mfma16(frag_tr_row(sQ, nt, qb, lane), sf, dk[nt]);
      }
    }
  }
  if (kpos < S) {
    const int np = H / hpb;                              // partials per token row
    float* dkr = dk_part + ((size_t)b * S + kpos) * (np * HD) + hg * HD;
    float* dvr = dv_part + ((size_t)b * S + kpos) * (np * HD) + hg * HD;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      *(f32x4*)(dkr + nt * 16 + 4 * qp) = dk[nt] * scale;
      *(f32x4*)(dvr + nt * 16 + 4 * qp) = dv[nt];
    }
  }
}

// ---- launchable forms.  attn_bwd_kernel runs BOTH halves of the attention backward in one grid: blocks [0, n_dkv) are
// dK/dV blocks (one per CU-slot, ~50 us each), the rest dQ blocks (three short rounds); launched one after the other the
// two kernels leave half of the chip idle in their tails (dQ: 768 blocks on 512 slots; dK/dV: 256 blocks, one round).
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ kt_g,
                                                             const uint8_t* __restrict__ kmask,
                                                             const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                             const float* __restrict__ delta, bf16* __restrict__ dqkv,
                                                             int S, int Spad, int H, int G, float scale, int causal) {
  __shared__ __attribute__((aligned(16))) char smem[DQ_LDS];
  attn_bwd_dq_body(qkv, kt_g, kmask, dout, lse, delta, dqkv, S, Spad, H, G, scale, causal, blockIdx.x, blockIdx.y, blockIdx.z,
                   smem);
}
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ qt_g,
                                                              const uint8_t* __restrict__ kmask,
                                                              const bf16* __restrict__ dout,
                                                              const bf16* __restrict__ dout_t,
                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                              float* __restrict__ dk_part, float* __restrict__ dv_part, int S,
                                                              int Spad, int H, int G, int hpb, float scale, int causal) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  attn_bwd_dkv_body(qkv, qt_g, kmask, dout, dout_t, lse, delta, dk_part, dv_part, S, Spad, H, G, hpb, scale, causal, blockIdx.x,
                    blockIdx.y, blockIdx.z, smem);
}
__global__ __launch_bounds__(256, 2) void attn_bwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ qt_g,
                                                          const bf16* __restrict__ kt_g, const uint8_t* __restrict__ kmask,
                                                          const bf16* __restrict__ dout, const bf16* __restrict__ dout_t,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          bf16* __restrict__ dqkv, float* __restrict__ dk_part,
                                                          float* __restrict__ dv_part, int S, int Spad, int H, int G, int hpb,
                                                          float scale, int causal, int B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nt = (S + 63) >> 6;
  const int n_dkv = nt * (H / hpb) * B;
  int id = blockIdx.x;
  if (id < n_dkv) {
    // longest blocks first: under the causal mask key tile 0 meets every query tile, the last key tile only one
    const int bx = id % nt, by = (id / nt) % (H / hpb), bz = id / (nt * (H / hpb));
    attn_bwd_dkv_body(qkv, qt_g, kmask, dout, dout_t, lse, delta, dk_part, dv_part, S, Spad, H, G, hpb, scale, causal, bx, by, bz,
                      smem);
  } else {
    id -= n_dkv;
    const int bx = nt - 1 - id % nt, by = (id / nt) % H, bz = id / (nt * H);      // long (late) query tiles first
    attn_bwd_dq_body(qkv, kt_g, kmask, dout, lse, delta, dqkv, S, Spad, H, G, scale, causal, bx, by, bz, smem);
  }
}

}  // namespace

static inline bool bad_geo(int B, int S, int H, int G) { return B <= 0 || S <= 0 || H <= 0 || G <= 0 || H % G != 0; }
static inline int spad_of(int S) { return (S + 63) & ~63; }

// attention_gqa.hip
extern "C" int tasu_attn_gqa_supported(int S, int H, int G);
int tasu_attn_bwd_gqa_launch(const void* qkv, const uint8_t* key_mask, const void* dout, const float* lse, const float* delta,
                             const float* cos_tab, const float* sin_tab, void* dqkv, int B, int S, int H, int G, float scale, int causal,
                             hipStream_t stream);

// attention_sp.hip
extern "C" int tasu_attn_sp_supported(int S, int H, int G);
int tasu_attn_sp_fwd_launch(const void* qkv, const uint8_t* key_mask, void* out, float* lse, int B, int S, int H, int G, float scale,
                            int causal, hipStream_t stream);
int tasu_attn_sp_bwd_launch(const void* qkv, const uint8_t* key_mask, const void* dout, const void* out, const float* lse,
                            const float* cos_tab, const float* sin_tab, void* dqkv, float* dk_part, float* dv_part, int B, int S, int H,
                            int G, float scale, int causal, hipStream_t stream);

static int attn_fwd_tiled(const void* qkv, const uint8_t* key_mask, void* out, float* lse, int B, int S, int H, int G, float scale,
                          int causal, void* stream);

extern "C" int tasu_attn_fwd_kernel(const void* qkv, const uint8_t* key_mask, void* out, float* lse, int B, int S, int H, int G,
                                    float scale, int causal, int kernel, void* stream) {
  if (!qkv || !key_mask || !out || !lse || bad_geo(B, S, H, G)) return TASU_ERR_ARG;
  if (kernel != TASU_ATTN_KERNEL_POLICY && kernel != TASU_ATTN_KERNEL_PER_HEAD && kernel != TASU_ATTN_KERNEL_SP) return TASU_ERR_ARG;
  const bool sp_ok = tasu_attn_sp_supported(S, H, G) != 0;
  if (kernel == TASU_ATTN_KERNEL_SP && !sp_ok) return TASU_ERR_ARG;
  // policy (tools/bench_attn_sp.py, us per layer, tiled / single pass, after the tiled kernel got the cheaper softmax arithmetic of
  // attn_tiles.h): 16 x 256 x 12 heads 17.3 / 18.0, 8 x 256 x 12 12.0 / 15.6, 32 x 256 x 12 32.3 / 35.2, 16 x 256 x 28 33.5 / 34.6 --
  // the tiled kernel everywhere; the single-pass forward stays reachable by name (TASU_ATTN_KERNEL_SP)
  const bool sp_take = false;
  if (kernel == TASU_ATTN_KERNEL_SP || (kernel == TASU_ATTN_KERNEL_POLICY && sp_take))
    return tasu_attn_sp_fwd_launch(qkv, key_mask, out, lse, B, S, H, G, scale, causal, (hipStream_t)stream);
  return attn_fwd_tiled(qkv, key_mask, out, lse, B, S, H, G, scale, causal, stream);
}

extern "C" int tasu_attn_fwd(const void* qkv, const void* vt, const uint8_t* key_mask, void* out, float* lse, int B,
                             int S, int H, int G, float scale, int causal, void* stream) {
  (void)vt;                                              // unused since round 2: V^T is read out of the V tile in LDS
  return tasu_attn_fwd_kernel(qkv, key_mask, out, lse, B, S, H, G, scale, causal, TASU_ATTN_KERNEL_POLICY, stream);
}

static int attn_fwd_tiled(const void* qkv, const uint8_t* key_mask, void* out, float* lse, int B, int S, int H, int G, float scale,
                          int causal, void* stream) {
  const void* vt = nullptr;
  static const int qw2_from = [] { const char* e = tasu_lab_env("TASU_ATTN_QW2_FROM"); return e ? atoi(e) : 1 << 30; }();
  if (S >= qw2_from) {
    // two query sub-tiles per wave (128-query blocks): half the LDS traffic per FLOP of the one-sub-tile form.  Measured at the
    // training shape (S = 256, causal): 20.7 us against 20.5 us, so the form is opt-in (TASU_ATTN_QW2_FROM=<S>) for long
    // non-causal sequences.
    dim3 grid(H, B, (S + 127) / 128);
    TASU_LAUNCH(attn_fwd_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)qkv, (const bf16*)vt, key_mask,
                (bf16*)out, lse, S, spad_of(S), H, G, scale, causal);
  } else {
    dim3 grid(H, B, (S + 63) / 64);
    TASU_LAUNCH(attn_fwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)qkv, (const bf16*)vt, key_mask,
                (bf16*)out, lse, S, spad_of(S), H, G, scale, causal);
  }
  return TASU_OK;
}

extern "C" int tasu_attn_bwd_prep(const void* dout, const void* out, float* delta, void* dout_t, int B, int S, int H,
                                  void* stream) {
  if (!dout || !out || !delta || B <= 0 || S <= 0 || H <= 0) return TASU_ERR_ARG;
  dim3 grid((S + 63) / 64, H, B);
  TASU_LAUNCH(attn_bwd_prep_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)dout,
                     (const bf16*)out, delta, (bf16*)dout_t, S, spad_of(S), H);
  return TASU_OK;
}

extern "C" int tasu_attn_bwd_dq(const void* qkv, const void* kt, const uint8_t* key_mask, const void* dout,
                                const float* lse, const float* delta, void* dqkv, int B, int S, int H, int G,
                                float scale, int causal, void* stream) {
  if (!qkv || !key_mask || !dout || !lse || !delta || !dqkv || bad_geo(B, S, H, G)) return TASU_ERR_ARG;
  dim3 grid((S + 63) / 64, H, B);
  TASU_LAUNCH(attn_bwd_dq_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)qkv, (const bf16*)kt,
                     key_mask, (const bf16*)dout, lse, delta, (bf16*)dqkv, S, spad_of(S), H, G, scale, causal);
  return TASU_OK;
}

extern "C" int tasu_attn_bwd_dkv(const void* qkv, const void* qt, const uint8_t* key_mask, const void* dout,
                                 const void* dout_t, const float* lse, const float* delta, float* dk_part,
                                 float* dv_part, int B, int S, int H, int G, float scale, int causal, void* stream) {
  if (!qkv || !key_mask || !dout || !lse || !delta || !dk_part || !dv_part || bad_geo(B, S, H, G)) return TASU_ERR_ARG;
  const int hpb = TASU_ATTN_DKV_HPB(H / G);
  dim3 grid((S + 63) / 64, H / hpb, B);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS);
    attr_set = true;
  }
  TASU_LAUNCH(attn_bwd_dkv_kernel, grid, dim3(256), DKV_LDS, (hipStream_t)stream, (const bf16*)qkv, (const bf16*)qt,
                     key_mask, (const bf16*)dout, (const bf16*)dout_t, lse, delta, dk_part, dv_part, S, spad_of(S), H, G,
                     hpb, scale, causal);
  return TASU_OK;
}

extern "C" int tasu_attn_bwd(const void* qkv, const void* qt, const void* kt, const uint8_t* key_mask, const void* dout,
                             const void* dout_t, const float* lse, const float* delta, void* dqkv, float* dk_part,
                             float* dv_part, int B, int S, int H, int G, float scale, int causal, void* stream) {
  if (!qkv || !key_mask || !dout || !lse || !delta || !dqkv || !dk_part || !dv_part || bad_geo(B, S, H, G)) return TASU_ERR_ARG;
  const int hpb = TASU_ATTN_DKV_HPB(H / G);
  const int nt = (S + 63) / 64;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS);
    attr_set = true;
  }
  TASU_LAUNCH(attn_bwd_kernel, dim3(nt * (H / hpb) * B + nt * H * B), dim3(256), DKV_LDS, (hipStream_t)stream, (const bf16*)qkv,
              (const bf16*)qt, (const bf16*)kt, key_mask, (const bf16*)dout, (const bf16*)dout_t, lse, delta, (bf16*)dqkv, dk_part,
              dv_part, S, spad_of(S), H, G, hpb, scale, causal, B);
  return TASU_OK;
}

// tasu_attn_bwd followed by tasu_rope_bwd (dqkv complete: rotated dq and dk, dv), behind one entry point.  With at least two
// query heads per kv head this is ONE launch of the GQA kernels (attention_gqa.hip: dK / dV complete in their workgroup, the
// rotation in the epilogues; dk_part / dv_part are not touched); otherwise the two per-head launches.  Same bits either way.
extern "C" int tasu_rope_bwd(void* dqkv, const float* dk_part, const float* dv_part, const float* cos_tab, const float* sin_tab, int B,
                             int S, int H, int G, void* stream);
extern "C" int tasu_attn_bwd_rope(const void* qkv, const uint8_t* key_mask, const void* dout, const float* lse, const float* delta,
                                  const float* cos_tab, const float* sin_tab, void* dqkv, float* dk_part, float* dv_part, int B, int S,
                                  int H, int G, float scale, int causal, int kernel, void* stream) {
  if (!qkv || !key_mask || !dout || !lse || !delta || !cos_tab || !sin_tab || !dqkv || bad_geo(B, S, H, G)) return TASU_ERR_ARG;
  if (kernel != TASU_ATTN_KERNEL_POLICY && kernel != TASU_ATTN_KERNEL_PER_HEAD && kernel != TASU_ATTN_KERNEL_GQA) return TASU_ERR_ARG;
  const bool gqa_ok = tasu_attn_gqa_supported(S, H, G) != 0;
  if (kernel == TASU_ATTN_KERNEL_GQA && !gqa_ok) return TASU_ERR_ARG;
  // policy (tools/bench_attn_sp.py, us per layer incl. tasu_attn_bwd_prep, per-head kernels + tasu_rope_bwd -> the GQA kernel;
  // round 5, both with the softmax arithmetic of attn_tiles.h -- the GQA kernel gained 9.6 us from it at the training shape, the
  // per-head kernels nothing): 16 x 256 x 12 / 2 heads 58.1 -> 51.1, 32 x 256 x 12 / 2 107.6 -> 78.3, 16 x 256 x 28 / 4 152.8 -> 92.2,
  // 8 x 256 x 12 / 2 47.5 -> 47.6: the GQA kernel wherever it is served (at least two query heads per kv head)
  const bool take = kernel == TASU_ATTN_KERNEL_GQA || (kernel == TASU_ATTN_KERNEL_POLICY && gqa_ok);
  if (take)
    return tasu_attn_bwd_gqa_launch(qkv, key_mask, dout, lse, delta, cos_tab, sin_tab, dqkv, B, S, H, G, scale, causal, (hipStream_t)stream);
  if (!dk_part || !dv_part) return TASU_ERR_ARG;
  const int rc = tasu_attn_bwd(qkv, nullptr, nullptr, key_mask, dout, nullptr, lse, delta, dqkv, dk_part, dv_part, B, S, H, G, scale,
                               causal, stream);
  return rc ? rc : tasu_rope_bwd(dqkv, dk_part, dv_part, cos_tab, sin_tab, B, S, H, G, stream);
}

// The whole attention backward behind one entry point: delta = rowsum(dO . O), dQ / dK / dV, the rotary embedding's backward.
// `kernel`: TASU_ATTN_KERNEL_SP = the single-pass kernels of attention_sp.hip (two launches, delta computed inside, `delta`
// untouched); _PER_HEAD / _GQA = tasu_attn_bwd_prep + tasu_attn_bwd_rope with that kernel; _POLICY = single-pass where it is
// measured faster (Spad <= 256 and 3 B H <= 320), else tasu_attn_bwd_rope's own policy.  dk_part / dv_part: fp32 [M, H * 128] each.
extern "C" int tasu_attn_bwd_fused(const void* qkv, const uint8_t* key_mask, const void* dout, const void* out, const float* lse,
                                   float* delta, const float* cos_tab, const float* sin_tab, void* dqkv, float* dk_part, float* dv_part,
                                   int B, int S, int H, int G, float scale, int causal, int kernel, void* stream) {
  if (!qkv || !key_mask || !dout || !out || !lse || !cos_tab || !sin_tab || !dqkv || bad_geo(B, S, H, G)) return TASU_ERR_ARG;
  if (kernel < TASU_ATTN_KERNEL_POLICY || kernel > TASU_ATTN_KERNEL_SP) return TASU_ERR_ARG;
  const bool sp_ok = tasu_attn_sp_supported(S, H, G) != 0;
  if (kernel == TASU_ATTN_KERNEL_SP && !sp_ok) return TASU_ERR_ARG;
  // policy (tools/bench_attn_sp.py, us per layer, prep + tiled + rotary -> single pass): 8 x 256 x 12 heads 47.1 -> 35.6, but
  // 16 x 256 x 12 58.1 -> 65.5 and 32 x 256 x 12 106.7 -> 131.6: the three roles are 3 B H workgroups of one per CU, each of which
  // spends ~10 us on its 190-320 KiB of operand ingest (~50 GB/s per CU) and runs its steps with one wave per SIMD; they win only
  // while they fit one round of the chip
  const bool sp_take = sp_ok && 3 * B * H <= 320;
  if (kernel == TASU_ATTN_KERNEL_SP || (kernel == TASU_ATTN_KERNEL_POLICY && sp_take)) {
    if (!dk_part || !dv_part) return TASU_ERR_ARG;
    return tasu_attn_sp_bwd_launch(qkv, key_mask, dout, out, lse, cos_tab, sin_tab, dqkv, dk_part, dv_part, B, S, H, G, scale, causal,
                                   (hipStream_t)stream);
  }
  if (!delta) return TASU_ERR_ARG;
  const int rc = tasu_attn_bwd_prep(dout, out, delta, nullptr, B, S, H, stream);
  return rc ? rc : tasu_attn_bwd_rope(qkv, key_mask, dout, lse, delta, cos_tab, sin_tab, dqkv, dk_part, dv_part, B, S, H, G, scale, causal,
                                      kernel, stream);
}
