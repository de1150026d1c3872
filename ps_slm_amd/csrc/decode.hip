// Decode-loop kernels (Multitask/model/ps-slm.py:660-675 -> HF GenerationMixin beam search, 4 beams): KV cache
// fill / append / beam reorder, single-token GQA attention over the cache, and the per-row log-softmax + top-k that
// feeds the beam bookkeeping.  All HBM-bound (weights and KV are read once per step).
#include "common.h"
#include "attn_decode_body.h"
#include "../../include/tasu_hip.h"

namespace {
constexpr int HD = 128;

// cache[b*nb, s, :] = k|v block of qkv[(b*S + s), :]: the prompt is stored ONCE per utterance, in the cache row of its
// first beam; the other beams reach it through the row index (kv_index_*).   grid (S, B), block 256
__global__ __launch_bounds__(256) void kv_fill_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ kc, bf16* __restrict__ vc,
                                                      int S, int H, int G, int nb, int ctx) {
  const int s = blockIdx.x, b = blockIdx.y;
  const int LD = (H + 2 * G) * HD, W = G * HD;
  const bf16* src = qkv + ((size_t)b * S + s) * LD + H * HD;
  for (int c = threadIdx.x * 8; c < 2 * W; c += 256 * 8) {
    const bf16x8 v = *(const bf16x8*)(src + c);
    bf16* base = c < W ? kc : vc;
    const int cc = c < W ? c : c - W;
    *(bf16x8*)(base + (((size_t)b * nb) * ctx + s) * W + cc) = v;
  }
}

// Decode-step RoPE + cache append: one block per row; thread t rotates 8-element chunk pairs (x[d], x[d+64]) of the H query
// and G key heads in place and copies the rotated keys and the values into cache[row, pos[row]].
__global__ __launch_bounds__(256) void rope_append_kernel(bf16* __restrict__ qkv, const float* __restrict__ ct,
                                                          const float* __restrict__ st, bf16* __restrict__ kc,
                                                          bf16* __restrict__ vc, const int32_t* __restrict__ pos, int H, int G,
                                                          int ctx) {
  const int row = blockIdx.x;
  const int LD = (H + 2 * G) * HD, W = G * HD;
  bf16* x = qkv + (size_t)row * LD;
  const size_t slot = ((size_t)row * ctx + pos[row]) * W;
  const float* cr = ct + (size_t)row * 64;
  const float* sr = st + (size_t)row * 64;
  for (int u = threadIdx.x; u < (H + G) * 8; u += 256) {       // unit = (head, 8-wide chunk of the low half)
    const int hh = u >> 3, c = (u & 7) * 8;
    bf16* hp = x + hh * HD;
    bf16x8 lo = *(const bf16x8*)(hp + c), hi = *(const bf16x8*)(hp + 64 + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float cs = cr[c + j], sn = sr[c + j];
      const float x1 = (float)lo[j], x2 = (float)hi[j];
      lo[j] = (bf16)(x1 * cs - x2 * sn);
      hi[j] = (bf16)(x2 * cs + x1 * sn);
    }
    *(bf16x8*)(hp + c) = lo;
    *(bf16x8*)(hp + 64 + c) = hi;
    if (hh >= H) {
      bf16* kd = kc + slot + (hh - H) * HD;
      *(bf16x8*)(kd + c) = lo;
      *(bf16x8*)(kd + 64 + c) = hi;
    }
  }
  for (int c = threadIdx.x * 8; c < W; c += 256 * 8) *(bf16x8*)(vc + slot + c) = *(const bf16x8*)(x + (H + G) * HD + c);
}

// Row index of the beam-shared cache: index[m, i] = physical cache row that holds position i of logical row (beam) m.
// init: prompt positions point at the utterance's first beam, generated positions at the row itself.
__global__ __launch_bounds__(256) void kv_index_init_kernel(int32_t* __restrict__ index, int nb, int S, int ctx, int total) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int m = e / ctx, i = e - m * ctx;
  index[e] = i < S ? (m / nb) * nb : m;
}
// beam reorder (HF Cache.reorder_cache) on the index instead of on the cache: dst[m, :lens[m]] = src[src_row[m], :lens[m]];
// positions >= lens[m] keep their value (the identity written by init: a row always appends into itself).  grid (M)
__global__ __launch_bounds__(256) void kv_index_reorder_kernel(const int32_t* __restrict__ src, int32_t* __restrict__ dst,
                                                               const int32_t* __restrict__ src_row,
                                                               const int32_t* __restrict__ lens, int ctx) {
  const int m = blockIdx.x;
  const int n = lens[m];
  const int32_t* s = src + (size_t)(src_row ? src_row[m] : m) * ctx;
  for (int i = threadIdx.x; i < n; i += 256) dst[(size_t)m * ctx + i] = s[i];
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

using tasu_attn_dec::DEC_NW;
using tasu_attn_dec::MAX_CTX;
// Single-token GQA attention over the cache: one 8-wave block per (row, kv group); body and design notes: attn_decode_body.h
template <int REP>
__global__ __launch_bounds__(64 * DEC_NW) void attn_decode_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ kc,
                                                                  const bf16* __restrict__ vc,
                                                                  const int32_t* __restrict__ row_index,
                                                                  const int32_t* __restrict__ kstart,
                                                                  const int32_t* __restrict__ lens, bf16* __restrict__ out,
                                                                  int H, int G, int ctx, float scale, int out_frag) {
  extern __shared__ float sp[];
  tasu_attn_dec::attn_decode_body<REP, false>(sp, blockIdx.x, blockIdx.y, qkv, kc, vc, row_index, kstart, lens, out, H, G, ctx, scale,
                                              out_frag);
}

// per row: lse over V columns, then the k best log-probs (value = logit - lse) with their column ids, descending;
// columns listed in `banned` (n_banned ids, e.g. EOS while cur_len < min_length) score -inf.   k <= 16.
constexpr int TOPK_MAX = 16;
constexpr int TOPK_PARTS = 16;       // column parts per row: 64 rows x 16 parts = 1024 blocks for the 152k-column vocabulary
// Stage 1, grid (M, TOPK_PARTS): block (row, part) scans its share of the row's columns with 16-byte loads and leaves
//   pm/ps = max and sum exp(x - max) over its columns, pv/pi = its K largest logits (descending; ties: smaller column).
// Stage 2, grid (M): merges the parts (lse = log sum exp over all columns; K rounds of argmax over the P sorted lists).
template <int K>
__device__ void topk_part_lists(const bf16* __restrict__ logits, int ld, int V,
                                                        const int32_t* __restrict__ banned, int n_banned,
                                                        float* __restrict__ pm, float* __restrict__ ps,
                                                        float* __restrict__ pv, int32_t* __restrict__ pi) {
  // The general form: per-thread sorted lists merged per wave and per block.  Used by topk_part_kernel when more than CAND_CAP
  // columns tie with or exceed its selection threshold (it re-does the whole part, the statistics included).
  __shared__ float red[4];
  __shared__ float cv[4 * K];
  __shared__ int ci[4 * K];
  const int row = blockIdx.x, part = blockIdx.y;
  const bf16* lr = logits + (size_t)row * ld;
  float tv[K];
  int ti[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    tv[j] = -__builtin_inff();
    ti[j] = 0x7fffffff;
  }
  // this part's 8-column chunks [v0, v1) (ld % 8 == 0, checked by the host entry); columns >= V are skipped
  const int nv = (V + 7) / 8;
  const int per = (nv + TOPK_PARTS - 1) / TOPK_PARTS;
  const int v0 = part * per, v1 = min(nv, v0 + per);
  // a thread's chunks are loaded ONCE, all loads in flight together (two dependent passes over global memory, five
  // sequential loads each, were most of this kernel's time); parts longer than 256 * MAXC chunks take the loop below too
  constexpr int MAXC = 6;
  bf16x8 xs[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int cv8 = v0 + threadIdx.x + i * 256;
    xs[i] = cv8 < v1 ? *(const bf16x8*)(lr + cv8 * 8) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
  const int vreg = min(v1, v0 + 256 * MAXC);      // chunks [v0, vreg) live in registers
  float m = -__builtin_inff();
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int cv8 = v0 + threadIdx.x + i * 256;
    if (cv8 < vreg) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (cv8 * 8 + j < V) m = fmaxf(m, (float)xs[i][j]);
    }
  }
  for (int cv8 = vreg + threadIdx.x; cv8 < v1; cv8 += 256) {
    const bf16x8 x = *(const bf16x8*)(lr + cv8 * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (cv8 * 8 + j < V) m = fmaxf(m, (float)x[j]);
  }
  m = block_max<4>(m, red);
  float s = 0.f;
  const int ban0 = n_banned > 0 ? banned[0] : -1, ban1 = n_banned > 1 ? banned[1] : -1;    // the usual case: EOS below min_length
  auto visit = [&](const bf16x8& x, int cv8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cv8 * 8 + j;
      if (c >= V) continue;
      const float f = (float)x[j];
      s += __expf(f - m);
      // thread-local sorted list (descending; on ties the smaller column, seen first, stays ahead)
      if (f > tv[K - 1]) {
        bool ban = c == ban0 || c == ban1;             // (a global load per candidate here cost ~0.5 us each, ~20 per thread)
        for (int b = 2; b < n_banned; ++b) ban |= (banned[b] == c);
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
  };
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int cv8 = v0 + threadIdx.x + i * 256;
    if (cv8 < vreg) visit(xs[i], cv8);
  }
  for (int cv8 = vreg + threadIdx.x; cv8 < v1; cv8 += 256) visit(*(const bf16x8*)(lr + cv8 * 8), cv8);
  s = block_sum<4>(s, red);
  const size_t slot0 = (size_t)row * TOPK_PARTS + part;
  if (threadIdx.x == 0) {
    pm[slot0] = m;
    ps[slot0] = s;
  }
  // Selection without block-wide rounds (K rounds of block argmax with two barriers each were 30 of this kernel's 38 us):
  //   each WAVE merges its 64 sorted lists by K rounds of wave argmax over the list heads (shuffles only; the winner shifts
  //   its register list up), leaving the wave's K best, sorted; the four waves' 4 K candidates then meet in LDS and wave 0
  //   ranks them (rank = number of candidates ahead in the (value desc, column asc) order).
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < K; ++r) {
    float v = tv[0];
    int id = ti[0];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(v, o, 64);
      const int oi = __shfl_xor(id, o, 64);
      if (ov > v || (ov == v && oi < id)) {
        v = ov;
        id = oi;
      }
    }
    if (id == ti[0] && v == tv[0]) {                // this lane's head won (columns are unique; exhausted lists hold 0x7fffffff)
#pragma unroll
      for (int j = 0; j + 1 < K; ++j) {
        tv[j] = tv[j + 1];
        ti[j] = ti[j + 1];
      }
      tv[K - 1] = -__builtin_inff();
      ti[K - 1] = 0x7fffffff;
    }
    if (lane == 0) {
      cv[wave * K + r] = v;
      ci[wave * K + r] = id;
    }
  }
  __syncthreads();
  if (wave == 0) {
    static_assert(4 * K <= 64, "the four waves' candidates fit one wave");
    const bool live = lane < 4 * K;
    const float v = live ? cv[lane] : -__builtin_inff();
    const int id = live ? ci[lane] : 0x7fffffff;
    int rank = 0;
#pragma unroll
    for (int d = 0; d < 4 * K; ++d) {
      const float dv = __shfl(v, d, 64);
      const int di = __shfl(id, d, 64);
      rank += (dv > v || (dv == v && (di < id || (di == id && d < lane)))) ? 1 : 0;
    }
    if (live && rank < K) {
      pv[slot0 * K + rank] = v;
      pi[slot0 * K + rank] = id;
    }
  }
}

// Threshold form (the common case; the per-thread insertion lists of topk_part_lists are instruction-bound: in a wave some lane
// inserts at almost every element, so every element pays the whole insertion -- 10 of the kernel's 21 us):
//   tau = the largest, over the four waves, of the wave's K-th largest per-thread maximum (banned columns excluded): at least K
//   columns are >= tau, so the part's K best all are; a second scan over the register-resident chunks collects the columns
//   >= tau (a handful) into LDS, and one wave ranks them by (value desc, column asc).  More than CAND_CAP such columns
//   (massive ties): the general form takes over.
constexpr int CAND_CAP = 64;
template <int K>
__global__ __launch_bounds__(256) void topk_part_kernel(const bf16* __restrict__ logits, int ld, int V,
                                                        const int32_t* __restrict__ banned, int n_banned,
                                                        float* __restrict__ pm, float* __restrict__ ps,
                                                        float* __restrict__ pv, int32_t* __restrict__ pi) {
  __shared__ float red[4];
  __shared__ float wtau[4];
  __shared__ float cand_v[CAND_CAP];
  __shared__ int cand_i[CAND_CAP];
  __shared__ int cand_n;
  const int row = blockIdx.x, part = blockIdx.y;
  [[maybe_unused]] const int g = part;                   // (TASU_ATTN_STAMP's workgroup test: row == 0 && g == 0)
  TASU_ATTN_STAMP(9);
  const bf16* lr = logits + (size_t)row * ld;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = (V + 7) / 8;
  const int per = (nv + TOPK_PARTS - 1) / TOPK_PARTS;
  const int v0 = part * per, v1 = min(nv, v0 + per);
  constexpr int MAXC = 6;
  if (v1 - v0 > 256 * MAXC) {                              // a part longer than the register window: general form
    topk_part_lists<K>(logits, ld, V, banned, n_banned, pm, ps, pv, pi);
    return;
  }
  if (threadIdx.x == 0) cand_n = 0;
  bf16x8 xs[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int cv8 = v0 + threadIdx.x + i * 256;
    xs[i] = cv8 < v1 ? *(const bf16x8*)(lr + cv8 * 8) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
  const int ban0 = n_banned > 0 ? banned[0] : -1, ban1 = n_banned > 1 ? banned[1] : -1;
  auto is_banned = [&](int c) {
    bool ban = c == ban0 || c == ban1;
    for (int b = 2; b < n_banned; ++b) ban |= (banned[b] == c);
    return ban;
  };
  // pass 1: the part's maximum (softmax statistics: all columns) and this thread's best selectable column value
  float m = -__builtin_inff(), msel = -__builtin_inff();
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int cv8 = v0 + threadIdx.x + i * 256;
    if (cv8 < v1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = cv8 * 8 + j;
        if (c < V) {
          const float f = (float)xs[i][j];
          m = fmaxf(m, f);
          if (f > msel && !is_banned(c)) msel = f;
        }
      }
    }
  }
  TASU_ATTN_STAMP(10);
  m = block_max<4>(m, red);
  TASU_ATTN_STAMP(11);
  // the wave's K-th largest per-thread best: K rounds of (wave max, retire one lane that holds it)
  float mine = msel, kth = -__builtin_inff();
#pragma unroll
  for (int r = 0; r < K; ++r) {
    kth = wave_max(mine);
    const unsigned long long holders = __ballot(mine == kth);
    if (lane == __ffsll((long long)holders) - 1) mine = -__builtin_inff();
  }
  if (lane == 0) wtau[wave] = kth;
  __syncthreads();
  const float tau = fmaxf(fmaxf(wtau[0], wtau[1]), fmaxf(wtau[2], wtau[3]));
  // pass 2: sum of exp, and the columns >= tau
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int cv8 = v0 + threadIdx.x + i * 256;
    if (cv8 < v1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = cv8 * 8 + j;
        if (c < V) {
          const float f = (float)xs[i][j];
          s += __expf(f - m);
          if (f >= tau && f > -__builtin_inff() && !is_banned(c)) {
            const int slot = atomicAdd(&cand_n, 1);
            if (slot < CAND_CAP) {
              cand_v[slot] = f;
              cand_i[slot] = c;
            }
          }
        }
      }
    }
  }
  TASU_ATTN_STAMP(12);
  s = block_sum<4>(s, red);                              // (its barriers also publish the candidates)
  const int n_cand = cand_n;
  if (n_cand > CAND_CAP) {                                 // block-uniform
    topk_part_lists<K>(logits, ld, V, banned, n_banned, pm, ps, pv, pi);
    return;
  }
  const size_t slot0 = (size_t)row * TOPK_PARTS + part;
  if (threadIdx.x == 0) {
    pm[slot0] = m;
    ps[slot0] = s;
  }
  TASU_ATTN_STAMP(13);
  if (wave == 0) {
    const bool live = lane < n_cand;
    const float v = live ? cand_v[lane] : -__builtin_inff();
    const int id = live ? cand_i[lane] : 0x7fffffff;
    int rank = 0;
    for (int d = 0; d < n_cand; ++d) {
      const float dv = __shfl(v, d, 64);
      const int di = __shfl(id, d, 64);
      rank += (dv > v || (dv == v && di < id)) ? 1 : 0;
    }
    if (live && rank < K) {
      pv[slot0 * K + rank] = v;
      pi[slot0 * K + rank] = id;
    }
    if (lane >= n_cand && lane < K) {                     // fewer than K selectable columns in this part
      pv[slot0 * K + lane] = -__builtin_inff();
      pi[slot0 * K + lane] = 0x7fffffff;
    }
  }
  TASU_ATTN_STAMP(14);
}

// one wave per row: lane p < TOPK_PARTS owns part p's sorted list
template <int K>
__global__ __launch_bounds__(64) void topk_merge_kernel(const float* __restrict__ pm, const float* __restrict__ ps,
                                                        const float* __restrict__ pv, const int32_t* __restrict__ pi,
                                                        float* __restrict__ out_val, int32_t* __restrict__ out_idx) {
  const int row = blockIdx.x, lane = threadIdx.x;
  const bool live = lane < TOPK_PARTS;
  const size_t slot0 = (size_t)row * TOPK_PARTS + (live ? lane : 0);
  const float mp = live ? pm[slot0] : -__builtin_inff();
  const float m = wave_max(mp);
  const float s = wave_sum(live && mp > -__builtin_inff() ? ps[slot0] * __expf(mp - m) : 0.f);
  const float lse = m + __logf(s);
  // the part's sorted list in registers (one round trip; a load per round made the K rounds K dependent round trips);
  // the winner of a round shifts its list up
  float lv[K];
  int li[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    lv[j] = live ? pv[slot0 * K + j] : -__builtin_inff();
    li[j] = live ? pi[slot0 * K + j] : 0x7fffffff;
  }
#pragma unroll
  for (int r = 0; r < K; ++r) {
    float v = lv[0];
    int id = li[0];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(v, o, 64);
      const int oi = __shfl_xor(id, o, 64);
      if (ov > v || (ov == v && oi < id)) {
        v = ov;
        id = oi;
      }
    }
    if (id == li[0] && v == lv[0]) {                // this part's head won (columns are unique; exhausted lists hold 0x7fffffff)
#pragma unroll
      for (int j = 0; j + 1 < K; ++j) {
        lv[j] = lv[j + 1];
        li[j] = li[j + 1];
      }
      lv[K - 1] = -__builtin_inff();
      li[K - 1] = 0x7fffffff;
    }
    if (lane == 0) {
      out_val[(size_t)row * K + r] = v - lse;
      out_idx[(size_t)row * K + r] = id;
    }
  }
}

// ---------------------------------------------------------------------------------------------- beam bookkeeping
// One generated position of HF ``generate(num_beams = nb, do_sample = False, early_stopping = False)`` (transformers
// generation/utils.py ``_beam_search``; restated in oracle/tasu_oracle.py::beam_search_generate and, vectorised, in
// ps_slm_amd/decode.py::BeamState, which this kernel reproduces decision for decision): one wave per utterance.
//   candidates  = per-row top-K log-probs (K = 2 nb) + the running score of their beam;
//   top K       by (score desc, beam asc, token asc) -- the order of a flattened [nb * V] top-k;
//   running     = the first nb candidates that neither are EOS nor reach max_new (their score otherwise + NEG);
//   finished    = candidates among the first nb that stop compete with the kept ones on score / len^penalty;
//   done        = no utterance can still improve (HF's heuristic on the best running score) or every candidate stopped.
// Sequences are not copied: every step records (token, parent slot) of the running beams in bp_tok / bp_par
// [step][b][slot], a finished hypothesis keeps (last step, parent slot, last token), and the host walks the back-pointers
// once at the end.  The kernel also writes the NEXT step's device inputs (token ids, cache source rows for the beam
// reorder, position ids, cache slots, lengths, the EOS ban while cur < min_length), so the whole decode step replays as one
// hipGraph with no host round trip; ctl[0] = positions generated so far, ctl[1] = done (then the kernel is a no-op).
constexpr float BEAM_NEG = -1.0e9f;
constexpr int BEAM_MAX_NB = 5;          // nb * 2 nb candidates must fit the 64-bit `used` mask below
struct BeamArgs {
  const float* vals;          // [B * nb, K] (first call: [B, K], beams >= 1 absent)
  const int32_t* idx;
  float* run_scores;          // [B, nb]
  float* fin_scores;          // [B, nb]
  int32_t* fin_len;           // [B, nb]
  int32_t* fin_par;           // [B, nb]   parent slot (running set of the previous step) of a kept finished hypothesis
  int32_t* fin_tok;           // [B, nb]   its last token
  int32_t* is_fin;            // [B, nb]
  int32_t* unsat;             // [B]
  int32_t* bp_tok;            // [max_new, B, nb]
  int32_t* bp_par;            // [max_new, B, nb]
  const float* len_pow;       // [max_new + 2]: float32(t ** length_penalty)
  int32_t* ctl;               // [0] cur, [1] done
  int32_t* done_host;         // optional pinned host word mirroring ctl[1]
  const int32_t* valid;       // [B] real (unpadded) prompt length
  int32_t* next_ids;          // [B * nb]
  int32_t* next_src;
  int32_t* next_pos;
  int32_t* next_slot;
  int32_t* next_lens;
  int32_t* banned;            // [1]: eos while the next position is still below min_length, else -1
  int B, nb, max_new, eos, min_length, S, first;
};

// One WAVE per utterance (16 utterances per pass of a 1024-thread block): lane c < nb * K holds candidate (beam c / K, k-th best
// of that beam); every selection is a rank computed with wave shuffles -- rank = number of candidates that precede this one in
// the (score desc, beam asc, token asc, index asc) order -- so there are no per-thread arrays and no serial scans (the first,
// thread-per-utterance form of this kernel spent 88 us per position in scratch-memory loops).
__global__ __launch_bounds__(1024) void beam_update_kernel(BeamArgs p) {
  __shared__ int s_unsat_any, s_stop_all;
  __shared__ float s_top_lp[16][2 * BEAM_MAX_NB];
  __shared__ int s_top_tok[16][2 * BEAM_MAX_NB], s_top_beam[16][2 * BEAM_MAX_NB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nb = p.nb, K = 2 * nb, NC = nb * K;
  const int cur = p.ctl[0];
  if (p.ctl[1]) return;                                   // finished earlier: leave every output as it is
  if (threadIdx.x == 0) {
    s_unsat_any = 0;
    s_stop_all = 1;
  }
  __syncthreads();
  const float lp_now = p.len_pow[cur + 1];
  for (int b = wave; b < p.B; b += 16) {                  // wave-uniform loop
    // ---- candidates
    const int j = lane / K, k = lane - j * K;
    float cs = -__builtin_inff();
    int ct = 0;
    if (lane < NC) {
      float v = BEAM_NEG;
      if (!p.first || j == 0) {
        const size_t row = p.first ? (size_t)b : (size_t)b * nb + j;
        v = p.vals[row * K + k];
        ct = p.idx[row * K + k];
      }
      cs = v + p.run_scores[b * nb + j];
    }
    // ---- top K by (score desc, beam asc, token asc, candidate index asc)
    int rank = 0;
    for (int d = 0; d < NC; ++d) {
      const float sd = __shfl(cs, d, 64);
      const int td = __shfl(ct, d, 64);
      const int jd = d / K;
      const bool before = sd > cs || (sd == cs && (jd < j || (jd == j && (td < ct || (td == ct && d < lane)))));
      rank += before ? 1 : 0;
    }
    if (lane < NC && rank < K) {
      s_top_lp[wave][rank] = cs;
      s_top_tok[wave][rank] = ct;
      s_top_beam[wave][rank] = j;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- lanes i < K: the K best in order
    const bool is_top = lane < K;
    const float top_lp = is_top ? s_top_lp[wave][lane] : 0.f;
    const int tok = is_top ? s_top_tok[wave][lane] : 0, beam = is_top ? s_top_beam[wave][lane] : 0;
    const bool stop = is_top && (tok == p.eos || cur + 1 >= p.max_new);
    const float run_lp = is_top ? top_lp + (stop ? BEAM_NEG : 0.f) : -__builtin_inff();
    const bool all_stop = __all(!is_top || stop);
    // running beams of the next step: stable top nb of run_lp
    int r_run = 0;
    for (int d = 0; d < K; ++d) {
      const float sd = __shfl(run_lp, d, 64);
      r_run += (sd > run_lp || (sd == run_lp && d < lane)) ? 1 : 0;
    }
    const bool runs = is_top && r_run < nb;
    if (runs) {
      const size_t o = ((size_t)cur * p.B + b) * nb + r_run;
      p.bp_tok[o] = tok;
      p.bp_par[o] = beam;
      const int m = b * nb + r_run;
      p.next_ids[m] = tok;
      p.next_src[m] = b * nb + beam;
      p.next_pos[m] = p.valid[b] + cur;
      p.next_slot[m] = p.S + cur;
      p.next_lens[m] = p.S + cur + 1;
    }
    const float best_run_lp = __shfl(run_lp, __ffsll((long long)__ballot(runs && r_run == 0)) - 1, 64);   // new running score of slot 0
    // ---- finished hypotheses: stable top nb of [kept (lanes 0..nb-1) | new (lanes nb..nb+K-1)]
    const bool unsat = p.unsat[b] != 0;
    const bool is_old = lane < nb, is_new = lane >= nb && lane < nb + K;
    const int src = lane - nb;                             // index into the K best for the new entries
    float m_sc = -__builtin_inff();
    int m_len = 0, m_par = 0, m_tok = 0, m_fin = 0;
    {
      // new entries read the top list through shuffles (lane src holds entry src)
      const float t_lp = __shfl(top_lp, src < 0 ? 0 : src, 64);
      const int t_tok = __shfl(tok, src < 0 ? 0 : src, 64), t_beam = __shfl(beam, src < 0 ? 0 : src, 64);
      const int t_stop = __shfl((int)stop, src < 0 ? 0 : src, 64);
      if (is_old) {
        m_sc = p.fin_scores[b * nb + lane];
        m_len = p.fin_len[b * nb + lane];
        m_par = p.fin_par[b * nb + lane];
        m_tok = p.fin_tok[b * nb + lane];
        m_fin = p.is_fin[b * nb + lane];
      } else if (is_new) {
        const bool just = t_stop && src < nb;
        float sc = t_lp / lp_now;
        sc = sc + (unsat ? 0.f : BEAM_NEG);
        sc = sc + (just ? 0.f : BEAM_NEG);
        m_sc = sc;
        m_len = cur + 1;
        m_par = t_beam;
        m_tok = t_tok;
        m_fin = just ? 1 : 0;
      }
    }
    int r_fin = 0;
    for (int d = 0; d < nb + K; ++d) {
      const float sd = __shfl(m_sc, d, 64);
      r_fin += (sd > m_sc || (sd == m_sc && d < lane)) ? 1 : 0;
    }
    const bool kept = (is_old || is_new) && r_fin < nb;
    // every lane has read its old entry: the wave is converged here, so the writes below cannot overtake those reads
    if (kept) {
      p.fin_scores[b * nb + r_fin] = m_sc;
      p.fin_len[b * nb + r_fin] = m_len;
      p.fin_par[b * nb + r_fin] = m_par;
      p.fin_tok[b * nb + r_fin] = m_tok;
      p.is_fin[b * nb + r_fin] = m_fin;
    }
    if (runs) p.run_scores[b * nb + r_run] = run_lp;
    // ---- can a running beam still beat the worst kept hypothesis?
    float min_fin = kept ? m_sc : __builtin_inff();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) min_fin = fminf(min_fin, __shfl_xor(min_fin, o, 64));
    const float best_run = best_run_lp / lp_now;           // the new cur is cur + 1: the same power
    const bool improve = __any(kept && best_run > (m_fin ? min_fin : BEAM_NEG));
    const bool still = unsat && improve;
    if (lane == 0) {
      p.unsat[b] = still ? 1 : 0;
      if (still) atomicOr(&s_unsat_any, 1);
      if (!all_stop) atomicAnd(&s_stop_all, 0);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int done = !(s_unsat_any && !s_stop_all);
    p.ctl[0] = cur + 1;
    p.ctl[1] = done;
    p.banned[0] = (cur + 1 < p.min_length) ? p.eos : -1;
    if (p.done_host) __hip_atomic_store(p.done_host, done ? cur + 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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

// Everything a generated position needs before its first layer, in ONE launch (block = beam row m; each piece used to be a
// launch of ~5 us that does microseconds of work):
//   wave 0  x[m, :] = table[ids[m], :] (fp32 embedding row) and xn = RMSNorm(x[m], norm_w) in fragment order (stream_body.h's
//           norm_row_frag: the arithmetic of tasu_rmsnorm_fwd_frag), 64-row chunks;
//   wave 1  RoPE factors cos / sin [m, 64] of position pos[m] (rope.hip's rope_table_kernel arithmetic);
//   all     if m is the first row of its utterance (m % nb == 0): the beam reorder of the utterance's nb rows of the cache row
//           index, IN PLACE -- dst[r, :lens[r]] = src[src_row[r], :lens[r]] -- staged through LDS; a beam's parent is always a
//           row of the same utterance (HF beam search reorders within a batch item), which is what makes one workgroup enough.
template <int NG>
__global__ __launch_bounds__(256) void decode_step_prologue_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids,
                                                                   float* __restrict__ x, const float* __restrict__ norm_w,
                                                                   bf16* __restrict__ xn, float eps, const int32_t* __restrict__ pos,
                                                                   float* __restrict__ ct, float* __restrict__ st, float theta,
                                                                   int32_t* __restrict__ index, const int32_t* __restrict__ src_row,
                                                                   const int32_t* __restrict__ lens, int nb, int M, int ctx) {
  extern __shared__ int stage[];                          // [nb][ctx]
  constexpr int D = NG * 256;
  const int m = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave == 0) {
    const float* src = table + (size_t)ids[m] * D;
    for (int c = lane * 4; c < D; c += 256) *(f32x4*)(x + (size_t)m * D + c) = *(const f32x4*)(src + c);
    tasu_stream::norm_row_frag_ptr<NG, false>(src, norm_w, xn + (size_t)(m >> 6) * 64 * D, m & 63, eps);
  } else if (wave == 1) {
    const int half = HD / 2, i = lane;
    const float inv = 1.0f / powf(theta, (float)(2 * i) / (float)(2 * half));
    const float ang = (float)pos[m] * inv;
    float sn, cs;
    sincosf(ang, &sn, &cs);
    ct[m * half + i] = cs;
    st[m * half + i] = sn;
  }
  if (m % nb) return;
  const int rows = min(nb, M - m);
  for (int r = 0; r < rows; ++r) {
    const int n = lens[m + r];
    const int32_t* s = index + (size_t)src_row[m + r] * ctx;
    for (int i = threadIdx.x; i < n; i += 256) stage[r * ctx + i] = s[i];
  }
  __syncthreads();
  for (int r = 0; r < rows; ++r) {
    const int n = lens[m + r];
    for (int i = threadIdx.x; i < n; i += 256) index[(size_t)(m + r) * ctx + i] = stage[r * ctx + i];
  }
}

}  // namespace

extern "C" int tasu_decode_step_prologue(const float* table, const int32_t* ids, float* x, const float* norm_w, void* xn_frag,
                                         float eps, const int32_t* pos, float* cos_tab, float* sin_tab, float theta, int32_t* index,
                                         const int32_t* src_row, const int32_t* lens, int n_beams, int M, int D, int ctx,
                                         void* stream) {
  if (!table || !ids || !x || !norm_w || !xn_frag || !pos || !cos_tab || !sin_tab || !index || !src_row || !lens) return TASU_ERR_ARG;
  if (M <= 0 || n_beams <= 0 || n_beams > BEAM_MAX_NB || ctx <= 0 || ctx > MAX_CTX || D % 256) return TASU_ERR_ARG;
  const size_t lds = (size_t)n_beams * ctx * sizeof(int);
#define TASU_PRO(NG)                                                                                                              \
  case NG:                                                                                                                        \
    TASU_LAUNCH(decode_step_prologue_kernel<NG>, dim3(M), dim3(256), lds, (hipStream_t)stream, table, ids, x, norm_w, (bf16*)xn_frag, \
                eps, pos, cos_tab, sin_tab, theta, index, src_row, lens, n_beams, M, ctx);                                       \
    return TASU_OK;
  switch (D / 256) {
    TASU_PRO(1) TASU_PRO(2) TASU_PRO(6) TASU_PRO(7) TASU_PRO(14)
    default: return TASU_ERR_ARG;
  }
#undef TASU_PRO
}

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
extern "C" int tasu_rope_append(void* qkv, const float* cos_tab, const float* sin_tab, void* kcache, void* vcache,
                                const int32_t* pos, int M, int H, int G, int ctx, void* stream) {
  if (!qkv || !cos_tab || !sin_tab || !kcache || !vcache || !pos || M <= 0 || H <= 0 || G <= 0 || ctx <= 0) return TASU_ERR_ARG;
  TASU_LAUNCH(rope_append_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (bf16*)qkv, cos_tab, sin_tab, (bf16*)kcache,
              (bf16*)vcache, pos, H, G, ctx);
  return TASU_OK;
}
extern "C" int tasu_kv_index_init(int32_t* index, int B, int n_beams, int S, int ctx, void* stream) {
  if (!index || B <= 0 || n_beams <= 0 || S <= 0 || ctx < S) return TASU_ERR_ARG;
  const int total = B * n_beams * ctx;
  TASU_LAUNCH(kv_index_init_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, index, n_beams, S, ctx, total);
  return TASU_OK;
}
extern "C" int tasu_kv_index_reorder(const int32_t* src_index, int32_t* dst_index, const int32_t* src_row, const int32_t* lens,
                                     int M, int ctx, void* stream) {
  if (!src_index || !dst_index || src_index == dst_index || !lens || M <= 0 || ctx <= 0) return TASU_ERR_ARG;
  TASU_LAUNCH(kv_index_reorder_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, src_index, dst_index, src_row, lens, ctx);
  return TASU_OK;
}
extern "C" int tasu_attn_decode(const void* qkv, const void* kcache, const void* vcache, const int32_t* row_index,
                                const int32_t* kstart, const int32_t* lens, void* out, int M, int H, int G, int ctx, float scale,
                                int out_frag, void* stream) {
  if (!qkv || !kcache || !vcache || !kstart || !lens || !out || M <= 0 || H <= 0 || G <= 0 || H % G || ctx <= 0 ||
      ctx > MAX_CTX)
    return TASU_ERR_ARG;
  const int rep = H / G;
  const size_t lds = (size_t)tasu_attn_dec::attn_decode_lds_floats(rep, ctx) * sizeof(float);
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
                (const bf16*)kcache, (const bf16*)vcache, row_index, kstart, lens, (bf16*)out, H, G, ctx, scale, out_frag);       \
    return TASU_OK;                                                                                                   \
  }
  switch (rep) {
    DEC_CASE(1) DEC_CASE(2) DEC_CASE(4) DEC_CASE(6) DEC_CASE(7) DEC_CASE(8)
    default:
      return TASU_ERR_ARG;
  }
#undef DEC_CASE
}
#ifdef TASU_ATTN_TRACE
extern "C" int tasu_attn_trace_read(uint64_t* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tasu_attn_dec::g_attn_trace), 16 * sizeof(uint64_t)) == hipSuccess ? 0 : 2;
}
#endif
extern "C" int tasu_logprob_topk(const void* logits, int ld, int M, int V, int k, const int32_t* banned, int n_banned,
                                 float* out_val, int32_t* out_idx, float* workspace, int64_t workspace_floats, void* stream) {
  if (!logits || !out_val || !out_idx || !workspace || M <= 0 || V <= 0 || ld < V || ld % 8 || k <= 0 || k > TOPK_MAX ||
      n_banned < 0 || (n_banned > 0 && !banned))
    return TASU_ERR_ARG;
  const size_t slots = (size_t)M * TOPK_PARTS;
  if ((size_t)workspace_floats < slots * (2 + 2 * (size_t)k)) return TASU_ERR_ARG;
  float* pm = workspace;
  float* ps = pm + slots;
  float* pv = ps + slots;
  int32_t* pi = (int32_t*)(pv + slots * k);
#define TOPK_CASE(KK)                                                                                                \
  case KK:                                                                                                           \
    TASU_LAUNCH(topk_part_kernel<KK>, dim3(M, TOPK_PARTS), dim3(256), 0, (hipStream_t)stream, (const bf16*)logits, ld, V, \
                banned, n_banned, pm, ps, pv, pi);                                                                   \
    TASU_LAUNCH(topk_merge_kernel<KK>, dim3(M), dim3(64), 0, (hipStream_t)stream, pm, ps, pv, pi, out_val, out_idx);  \
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

extern "C" int tasu_beam_update(const float* vals, const int32_t* idx, float* run_scores, float* fin_scores, int32_t* fin_len,
                                int32_t* fin_par, int32_t* fin_tok, int32_t* is_fin, int32_t* unsat, int32_t* bp_tok,
                                int32_t* bp_par, const float* len_pow, int32_t* ctl, int32_t* done_host, const int32_t* valid,
                                int32_t* next_ids, int32_t* next_src, int32_t* next_pos, int32_t* next_slot, int32_t* next_lens,
                                int32_t* banned, int B, int n_beams, int max_new, int eos, int min_length, int S, int first,
                                void* stream) {
  if (!vals || !idx || !run_scores || !fin_scores || !fin_len || !fin_par || !fin_tok || !is_fin || !unsat || !bp_tok || !bp_par ||
      !len_pow || !ctl || !valid || !next_ids || !next_src || !next_pos || !next_slot || !next_lens || !banned)
    return TASU_ERR_ARG;
  if (B <= 0 || B > 256 || n_beams <= 0 || n_beams > BEAM_MAX_NB || max_new <= 0 || S <= 0) return TASU_ERR_ARG;
  BeamArgs a{vals, idx, run_scores, fin_scores, fin_len, fin_par, fin_tok, is_fin, unsat, bp_tok, bp_par, len_pow, ctl,
             done_host, valid, next_ids, next_src, next_pos, next_slot, next_lens, banned, B, n_beams, max_new, eos,
             min_length, S, first};
  TASU_LAUNCH(beam_update_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
  return TASU_OK;
}
