// Decode-loop kernels (Multitask/model/ps-slm.py:660-675 -> HF GenerationMixin beam search, 4 beams): KV cache
// fill / append / beam reorder, single-token GQA attention over the cache, and the per-row log-softmax + top-k that
// feeds the beam bookkeeping.  All HBM-bound (weights and KV are read once per step).
#include "common.h"
#include "../../include/tasu_hip.h"

namespace {
constexpr int HD = 128;

// cache[(b*nb + j), s, :] = k|v block of qkv[(b*S + s), :]   for every beam j.   grid (S, B), block 256
__global__ __launch_bounds__(256) void kv_fill_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ kc, bf16* __restrict__ vc,
                                                      int S, int H, int G, int nb, int ctx) {
  const int s = blockIdx.x, b = blockIdx.y;
  const int LD = (H + 2 * G) * HD, W = G * HD;
  const bf16* src = qkv + ((size_t)b * S + s) * LD + H * HD;
  for (int c = threadIdx.x * 8; c < 2 * W; c += 256 * 8) {
    const bf16x8 v = *(const bf16x8*)(src + c);
    bf16* base = c < W ? kc : vc;
    const int cc = c < W ? c : c - W;
    for (int j = 0; j < nb; ++j) *(bf16x8*)(base + (((size_t)(b * nb + j)) * ctx + s) * W + cc) = v;
  }
}

// cache[row, pos[row], :] = k|v block of qkv[row, :]     grid (M), block 128
__global__ __launch_bounds__(128) void kv_append_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ kc, bf16* __restrict__ vc,
                                                        const int32_t* __restrict__ pos, int H, int G, int ctx) {
  const int row = blockIdx.x;
  const int LD = (H + 2 * G) * HD, W = G * HD;
  const bf16* src = qkv + (size_t)row * LD + H * HD;
  const int p = pos[row];
  for (int c = threadIdx.x * 8; c < 2 * W; c += 128 * 8) {
    const bf16x8 v = *(const bf16x8*)(src + c);
    if (c < W)
      *(bf16x8*)(kc + ((size_t)row * ctx + p) * W + c) = v;
    else
      *(bf16x8*)(vc + ((size_t)row * ctx + p) * W + (c - W)) = v;
  }
}

// dst[row, 0:len[row], :] = src[src_row[row], 0:len[row], :]  (beam reorder; ping-pong buffers)   grid (ctx_blocks, M)
__global__ __launch_bounds__(256) void kv_gather_kernel(const bf16* __restrict__ sk, const bf16* __restrict__ sv, bf16* __restrict__ dk,
                                                        bf16* __restrict__ dv, const int32_t* __restrict__ src_row,
                                                        const int32_t* __restrict__ lens, int W, int ctx) {
  const int row = blockIdx.y;
  const int n = lens[row];
  const int sr = src_row[row];
  const int per = (W / 8);                     // 16-byte chunks per position
  const size_t total = (size_t)n * per;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t off = i * 8;
    *(bf16x8*)(dk + (size_t)row * ctx * W + off) = *(const bf16x8*)(sk + (size_t)sr * ctx * W + off);
    *(bf16x8*)(dv + (size_t)row * ctx * W + off) = *(const bf16x8*)(sv + (size_t)sr * ctx * W + off);
  }
}

// Single-token GQA attention over the cache, flash-decoding style inside one block per (row, kv group):
// the 8 waves split the KEYS (not the heads), so every K / V row is read once for all REP query heads of the group
// and the serial chain per wave is nk/8 keys long.
//   phase 1  scores: 16 lanes per key (one 16-byte load each: a wave instruction reads 4 whole 256-B K rows), REP dot
//            products per key from q kept in registers, xor-shuffle reduction, scores -> LDS [REP][ctx]
//   phase 1b softmax statistics per head (wave h), probabilities (bf16-rounded like the prefill kernel) back to LDS
//   phase 2  P.V: lane owns dims 2*lane, 2*lane+1 of its wave's key range for all REP heads; partials -> LDS
//   phase 3  cross-wave sum, 1/l, bf16 store
// keys in [kstart[row], lens[row]) are visible.  ctx <= MAX_CTX.
constexpr int MAX_CTX = 2048;
constexpr int DEC_NW = 8;
template <int REP>
__global__ __launch_bounds__(64 * DEC_NW) void attn_decode_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ kc,
                                                                  const bf16* __restrict__ vc,
                                                                  const int32_t* __restrict__ kstart,
                                                                  const int32_t* __restrict__ lens, bf16* __restrict__ out,
                                                                  int H, int G, int ctx, float scale) {
  extern __shared__ float sp[];                         // [REP][ctx] scores | [DEC_NW][REP][128] partial outputs | [REP] 1/l
  float* sc = sp;
  float* part = sp + (size_t)REP * ctx;
  float* linv = part + DEC_NW * REP * HD;
  const int row = blockIdx.x, g = blockIdx.y;
  const int W = G * HD, LD = (H + 2 * G) * HD;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int k0 = kstart[row], nk = lens[row] - k0;
  const int chunk = ((nk + DEC_NW - 1) / DEC_NW + 3) & ~3;     // keys per wave, multiple of 4
  const int kb = wave * chunk, ke = min(nk, kb + chunk);
  const int sub = lane & 15, kq = lane >> 4;
  float q[REP][8];
#pragma unroll
  for (int h = 0; h < REP; ++h) {
    const bf16x8 v = *(const bf16x8*)(qkv + (size_t)row * LD + (g * REP + h) * HD + sub * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) q[h][j] = (float)v[j] * scale;
  }
  const bf16* kbase = kc + ((size_t)row * ctx + k0) * W + g * HD + sub * 8;
  for (int i0 = kb; i0 < ke; i0 += 4) {
    const int i = i0 + kq;
    float kf[8];
    if (i < ke) {
      const bf16x8 v = *(const bf16x8*)(kbase + (size_t)i * W);
#pragma unroll
      for (int j = 0; j < 8; ++j) kf[j] = (float)v[j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) kf[j] = 0.f;
    }
#pragma unroll
    for (int h = 0; h < REP; ++h) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += q[h][j] * kf[j];
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      s += __shfl_xor(s, 8, 64);
      if (sub == 0 && i < ke) sc[h * ctx + i] = s;
    }
  }
  __syncthreads();
  for (int h = wave; h < REP; h += DEC_NW) {               // softmax statistics of head h
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
  const bf16* vbase = vc + ((size_t)row * ctx + k0) * W + g * HD + 2 * lane;
  float o[REP][2];
#pragma unroll
  for (int h = 0; h < REP; ++h) o[h][0] = o[h][1] = 0.f;
  int i = kb;
  for (; i + 4 <= ke; i += 4) {
    bf16x2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *(const bf16x2*)(vbase + (size_t)(i + u) * W);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int h = 0; h < REP; ++h) {
        const float p = sc[h * ctx + i + u];
        o[h][0] += p * (float)v[u][0];
        o[h][1] += p * (float)v[u][1];
      }
  }
  for (; i < ke; ++i) {
    const bf16x2 v = *(const bf16x2*)(vbase + (size_t)i * W);
#pragma unroll
    for (int h = 0; h < REP; ++h) {
      const float p = sc[h * ctx + i];
      o[h][0] += p * (float)v[0];
      o[h][1] += p * (float)v[1];
    }
  }
#pragma unroll
  for (int h = 0; h < REP; ++h) *(f32x2*)(part + ((wave * REP + h) * HD) + 2 * lane) = f32x2{o[h][0], o[h][1]};
  __syncthreads();
  for (int e = threadIdx.x; e < REP * HD; e += 64 * DEC_NW) {
    const int h = e / HD, d = e - h * HD;
    float s = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < DEC_NW; ++w2) s += part[(w2 * REP + h) * HD + d];
    out[(size_t)row * (H * HD) + (g * REP + h) * HD + d] = (bf16)(s * linv[h]);
  }
}

// per row: lse over V columns, then the k best log-probs (value = logit - lse) with their column ids, descending;
// columns listed in `banned` (n_banned ids, e.g. EOS while cur_len < min_length) score -inf.   k <= 16.
constexpr int TOPK_MAX = 16;
template <int K>
__global__ __launch_bounds__(256) void logprob_topk_kernel(const bf16* __restrict__ logits, int ld, int V,
                                                           const int32_t* __restrict__ banned, int n_banned,
                                                           float* __restrict__ out_val, int32_t* __restrict__ out_idx) {
  __shared__ float red[4];
  __shared__ float cv[256 * K];
  __shared__ int ci[256 * K];
  __shared__ float bestv[4];
  __shared__ int besti[4], bestslot[4];
  const int row = blockIdx.x;
  const bf16* lr = logits + (size_t)row * ld;
  float tv[K];
  int ti[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    tv[j] = -__builtin_inff();
    ti[j] = 0x7fffffff;
  }
  // 16-byte loads (ld % 8 == 0, checked by the host entry); columns >= V are skipped
  const int nv = (V + 7) / 8;
  float m = -__builtin_inff();
  for (int cv8 = threadIdx.x; cv8 < nv; cv8 += 256) {
    const bf16x8 x = *(const bf16x8*)(lr + cv8 * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (cv8 * 8 + j < V) m = fmaxf(m, (float)x[j]);
  }
  m = block_max<4>(m, red);
  float s = 0.f;
  for (int cv8 = threadIdx.x; cv8 < nv; cv8 += 256) {
    const bf16x8 x = *(const bf16x8*)(lr + cv8 * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cv8 * 8 + j;
      if (c >= V) continue;
      const float f = (float)x[j];
      s += __expf(f - m);
      // thread-local sorted list (descending; on ties the smaller column, seen first, stays ahead)
      if (f > tv[K - 1]) {
        bool ban = false;
        for (int b = 0; b < n_banned; ++b) ban |= (banned[b] == c);
        if (ban) continue;
        tv[K - 1] = f;
        ti[K - 1] = c;
#pragma unroll
        for (int jj = K - 1; jj > 0; --jj) {
          if (tv[jj] > tv[jj - 1]) {
            const float a = tv[jj];
            tv[jj] = tv[jj - 1];
            tv[jj - 1] = a;
            const int b2 = ti[jj];
            ti[jj] = ti[jj - 1];
            ti[jj - 1] = b2;
          }
        }
      }
    }
  }
  s = block_sum<4>(s, red);
  const float lse = m + __logf(s);
#pragma unroll
  for (int j = 0; j < K; ++j) {
    cv[threadIdx.x * K + j] = tv[j];
    ci[threadIdx.x * K + j] = ti[j];
  }
  __syncthreads();
  // K rounds of block-wide argmax over the 256 list heads (each list is sorted, so only heads compete)
  int head = 0;                                   // this thread's next unconsumed entry
  for (int r = 0; r < K; ++r) {
    float v = head < K ? cv[threadIdx.x * K + head] : -__builtin_inff();
    int id = head < K ? ci[threadIdx.x * K + head] : 0x7fffffff;
    int slot = threadIdx.x;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(v, o, 64);
      const int oi = __shfl_xor(id, o, 64);
      const int os = __shfl_xor(slot, o, 64);
      if (ov > v || (ov == v && oi < id)) {
        v = ov;
        id = oi;
        slot = os;
      }
    }
    if ((threadIdx.x & 63) == 0) {
      bestv[threadIdx.x >> 6] = v;
      besti[threadIdx.x >> 6] = id;
      bestslot[threadIdx.x >> 6] = slot;
    }
    __syncthreads();
    float bv = bestv[0];
    int bi = besti[0], bs = bestslot[0];
    for (int w = 1; w < 4; ++w)
      if (bestv[w] > bv || (bestv[w] == bv && besti[w] < bi)) {
        bv = bestv[w];
        bi = besti[w];
        bs = bestslot[w];
      }
    if (threadIdx.x == bs) ++head;
    if (threadIdx.x == 0) {
      out_val[(size_t)row * K + r] = bv - lse;
      out_idx[(size_t)row * K + r] = bi;
    }
    __syncthreads();
  }
}

// x[m,:] = table[ids[m], :]  (fp32 embedding rows for the decode step)
__global__ __launch_bounds__(256) void embed_rows_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids,
                                                         float* __restrict__ x, int M, int D) {
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (m >= M) return;
  const float* src = table + (size_t)ids[m] * D;
  for (int c = lane * 4; c < D; c += 256) *(f32x4*)(x + (size_t)m * D + c) = *(const f32x4*)(src + c);
}

}  // namespace

extern "C" int tasu_kv_fill(const void* qkv, void* kcache, void* vcache, int B, int S, int H, int G, int n_beams, int ctx,
                            void* stream) {
  if (!qkv || !kcache || !vcache || B <= 0 || S <= 0 || S > ctx || H <= 0 || G <= 0 || n_beams <= 0) return TASU_ERR_ARG;
  TASU_LAUNCH(kv_fill_kernel, dim3(S, B), dim3(256), 0, (hipStream_t)stream, (const bf16*)qkv, (bf16*)kcache, (bf16*)vcache, S,
              H, G, n_beams, ctx);
  return TASU_OK;
}
extern "C" int tasu_kv_append(const void* qkv, void* kcache, void* vcache, const int32_t* pos, int M, int H, int G, int ctx,
                              void* stream) {
  if (!qkv || !kcache || !vcache || !pos || M <= 0 || H <= 0 || G <= 0) return TASU_ERR_ARG;
  TASU_LAUNCH(kv_append_kernel, dim3(M), dim3(128), 0, (hipStream_t)stream, (const bf16*)qkv, (bf16*)kcache, (bf16*)vcache, pos,
              H, G, ctx);
  return TASU_OK;
}
extern "C" int tasu_kv_gather(const void* src_k, const void* src_v, void* dst_k, void* dst_v, const int32_t* src_row,
                              const int32_t* lens, int M, int G, int ctx, void* stream) {
  if (!src_k || !src_v || !dst_k || !dst_v || !src_row || !lens || M <= 0 || G <= 0 || ctx <= 0) return TASU_ERR_ARG;
  TASU_LAUNCH(kv_gather_kernel, dim3(8, M), dim3(256), 0, (hipStream_t)stream, (const bf16*)src_k, (const bf16*)src_v,
              (bf16*)dst_k, (bf16*)dst_v, src_row, lens, G * HD, ctx);
  return TASU_OK;
}
extern "C" int tasu_attn_decode(const void* qkv, const void* kcache, const void* vcache, const int32_t* kstart,
                                const int32_t* lens, void* out, int M, int H, int G, int ctx, float scale, void* stream) {
  if (!qkv || !kcache || !vcache || !kstart || !lens || !out || M <= 0 || H <= 0 || G <= 0 || H % G || ctx <= 0 ||
      ctx > MAX_CTX)
    return TASU_ERR_ARG;
  const int rep = H / G;
  const size_t lds = ((size_t)rep * ctx + DEC_NW * rep * HD + rep) * sizeof(float);
  if (lds > 160 * 1024) return TASU_ERR_ARG;
#define DEC_CASE(R)                                                                                                   \
  case R: {                                                                                                           \
    static bool attr_set = false;                                                                                     \
    if (!attr_set) {                                                                                                  \
      (void)hipFuncSetAttribute((const void*)attn_decode_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize,        \
                                160 * 1024);                                                                          \
      attr_set = true;                                                                                                \
    }                                                                                                                 \
    TASU_LAUNCH(attn_decode_kernel<R>, dim3(M, G), dim3(64 * DEC_NW), lds, (hipStream_t)stream, (const bf16*)qkv,      \
                (const bf16*)kcache, (const bf16*)vcache, kstart, lens, (bf16*)out, H, G, ctx, scale);                 \
    return TASU_OK;                                                                                                   \
  }
  switch (rep) {
    DEC_CASE(1) DEC_CASE(2) DEC_CASE(4) DEC_CASE(6) DEC_CASE(7) DEC_CASE(8)
    default:
      return TASU_ERR_ARG;
  }
#undef DEC_CASE
}
extern "C" int tasu_logprob_topk(const void* logits, int ld, int M, int V, int k, const int32_t* banned, int n_banned,
                                 float* out_val, int32_t* out_idx, void* stream) {
  if (!logits || !out_val || !out_idx || M <= 0 || V <= 0 || ld < V || ld % 8 || k <= 0 || k > TOPK_MAX || n_banned < 0 ||
      (n_banned > 0 && !banned))
    return TASU_ERR_ARG;
#define TOPK_CASE(KK)                                                                                                \
  case KK:                                                                                                           \
    TASU_LAUNCH(logprob_topk_kernel<KK>, dim3(M), dim3(256), 0, (hipStream_t)stream, (const bf16*)logits, ld, V, banned, \
                n_banned, out_val, out_idx);                                                                         \
    return TASU_OK;
  switch (k) {
    TOPK_CASE(1) TOPK_CASE(2) TOPK_CASE(4) TOPK_CASE(6) TOPK_CASE(8) TOPK_CASE(16)
    default:
      return TASU_ERR_ARG;
  }
#undef TOPK_CASE
}
extern "C" int tasu_embed_rows(const float* table, const int32_t* ids, float* x, int M, int D, void* stream) {
  if (!table || !ids || !x || M <= 0 || D <= 0 || D % 4) return TASU_ERR_ARG;
  TASU_LAUNCH(embed_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, table, ids, x, M, D);
  return TASU_OK;
}
