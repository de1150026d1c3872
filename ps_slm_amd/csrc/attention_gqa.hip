// Backward of the decoder's causal GQA attention (Qwen2Attention, transformers modeling_qwen2.py:150-172, SDPA over 12 q / 2 kv
// heads at 1.5B, 28 / 4 at 7B) as ONE launch: dQ, dK, dV and the rotary embedding's backward (tasu_rope_bwd), with the K / V
// (dQ role) and Q / dO (dK / dV role) tiles staged by LDS-DMA four deep and shared by the query heads of a GQA group.
//
//   * tiles go global -> LDS by DMA (buffer_load_dwordx4 ... lds), four tile pairs deep (ring of NS = 4 x 32 KiB); the bank
//     swizzle of the "row" image (attn_tiles.h: 16-B chunk c of row r at c ^ swz(r)) is applied on the per-lane SOURCE offset,
//     rows past the end of the sequence are clipped by the buffer descriptor (zeros);
//   * dQ role: one workgroup = (batch, kv group, 64 queries, up to three query heads per pass), wave = (head, 16 query rows):
//     the arithmetic, order and rounding of attention.hip's dQ kernel + tasu_rope_bwd -- the same bits;
//   * dK / dV role: one workgroup = (batch, kv head, 64 keys) sweeps ALL query heads of the group x the causal query tiles with
//     K / V fragments in registers (waves = 16-key sub-tile x query half of every tile; the halves meet in LDS at the end): dK
//     and dV are complete in the workgroup -- no fp32 partials in memory (the per-head kernel writes H / HPB of them per token,
//     7 per kv head at Qwen2.5-7B) and no reduction pass; the same products in another fp32 association.
//
// MEASURED (tools/bench_attn_gqa.py, us per layer incl. tasu_attn_bwd_prep, per-head kernels + tasu_rope_bwd -> this kernel):
// 16 x 256, 28 / 4 heads 152.6 -> 103.4; 16 x 628, 12 / 2 heads 171.5 -> 156.0; 16 x 256, 12 / 2 heads 59.1 -> 61.2 -- a win
// where the per-head kernels' partial sums or their staging chain weigh most, a tie at the 1.5B training shape, which the
// per-head kernels keep (tasu_attn_bwd_rope's policy).  What bounds both families at S = 256 is neither staging nor the LDS port
// (rocprofv3 --pmc: SQ_LDS_IDX_ACTIVE is 19 % of the kernel's CU cycles, and removing every bank conflict -- attn_tiles.h's swz,
// SQ_LDS_BANK_CONFLICT 1.47 M -> 0 -- changed no launch time) but the dependent chain inside a wave: fragment read -> 16 MFMAs ->
// exp / pack -> transposed reads -> 16 MFMAs, with two or three waves per SIMD to cover it (phase stamps of an instrumented
// build: an MFMA phase takes 2 - 4x its matrix-pipe time; a key-tile step of the forward kernel 2.5 us for 0.25 us of MFMA).
// A forward kernel of this family (built, bit-identical, 20.5 us against 19.5) is therefore not in the tree; what would move
// both is a software-pipelined body (the next step's fragment reads issued under the current MFMAs), not another staging scheme.
#include "attn_tiles.h"
#include "../../include/tasu_hip.h"

namespace tasu_gqa {

using namespace tasu_attn;

constexpr int TILE = ROW_TILE_BYTES;     // 64 tokens x 128 d, bf16
constexpr int PAIR = 2 * TILE;           // K + V, or Q + dO
constexpr int NS = 4;                    // ring slots
constexpr int EXTRA = 2048;              // per slot: lse[64], delta[64] of the slot's query tile (+ the other issuing waves' duplicates)
constexpr int MASK_MAX = 4096;           // keys whose additive mask bias (one float each: attn_tiles.h) is kept in LDS (Spad <= MASK_MAX)
constexpr int DUMMY = 4096;              // where the non-issuing waves' zero-record DMAs land (every wave issues the same count)
constexpr int BIAS_BYTES = MASK_MAX * 4;
constexpr int LDS_BYTES = NS * PAIR + NS * EXTRA + BIAS_BYTES + DUMMY;
static_assert(LDS_BYTES <= 160 * 1024, "LDS of one CU");

typedef __attribute__((address_space(3))) void lds_void;


template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt immediate");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else static_assert(N == 0, "add the immediate");
}
// Every wait of the tile loops leaves exactly the NS - 2 = 2 youngest tile pairs of the wave's DMAs in flight: tiles past the
// end of a workgroup's list are issued all the same, through a descriptor of zero records (no memory access; zeros land in a
// slot nobody reads), so that the counts are compile-time constants -- with a data-dependent number of DMAs in flight the
// compiler's own wait for the register-resident Q / K fragments degrades to vmcnt(0) at the loop header, which serialises the
// whole prologue.  PER = DMA instructions per tile pair and wave.
template <int PER>
__device__ __forceinline__ void wait_tile() {
  wait_vm<(NS - 2) * PER>();
}

// "The value is needed here": placed after the DMA issue for every register loaded before it, the compiler waits for those loads
// with a COUNTED vmcnt that leaves the DMAs in flight.  Without it, it finds loads pending at the head of a loop that only uses
// them and flushes vmcnt(0) in the preheader (all prologue tiles must land before the first MFMA).
template <typename T>
__device__ __forceinline__ void need(const T& v) {
  asm volatile("" ::"v"(v));
}

// Per-lane source offsets of a wave's four 1-KiB pieces of a [64][128] tile (row stride ld_bytes): piece p = wave * 4 + i holds
// tile rows 4p .. 4p + 3; lane l writes LDS chunk l & 15 of row 4p + (l >> 4), which must hold global chunk (l & 15) ^ swz(row).
__device__ __forceinline__ void tile_offsets(int (&voff)[4], int ld_bytes, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave * 4 + i) * 4 + (lane >> 4);
    voff[i] = r * ld_bytes + (((lane & 15) ^ swz(r)) << 4);
  }
}
// DMA of one tile whose first row is `origin` (wave-uniform); rows_left = valid rows from there on (<= 0: all zeros)
__device__ __forceinline__ void dma_tile(const bf16* origin, int ld_bytes, int rows_left, char* lds_tile, int wave, const int (&voff)[4]) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int rows = rows_left < 64 ? rows_left : 64;
  const unsigned nrec = rows > 0 ? (unsigned)(rows - 1) * (unsigned)ld_bytes + 256u : 0u;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)origin, 0, nrec, 0x00020000);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds_tile + (wave * 4 + i) * 1024), 16, voff[i], 0, 0, 0);
#endif
}
// 64 floats (one per lane) -> LDS
__device__ __forceinline__ void dma_floats64(const float* src, bool valid, char* lds_dst, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, valid ? 256u : 0u, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lds_dst, 4, lane * 4, 0, 0, 0);
#endif
}

// Workgroup placement: the workgroups that stage the same (batch, kv group)'s tiles get linear ids that are equal mod 8 -- one
// XCD under the round-robin placement (speed only) -- and the long (late-query / early-key) tiles come first.
// id -> (seq: position in the long-first order, pg: batch * G + g)
__device__ __forceinline__ void place(int id, int npg, int& seq, int& pg) {
  if ((npg & 7) == 0) {
    const int per = npg >> 3;              // pairs per XCD
    const int x = id & 7, q = id >> 3;
    seq = q / per;
    pg = (q % per) * 8 + x;
  } else {
    seq = id / npg;
    pg = id % npg;
  }
}

struct Geo {
  int S, Spad, H, G, B, hp;      // hp: query heads per forward / dQ workgroup
  float scale;
  int causal;
};

// ======================================================================================= forward / dQ shared skeleton
// Workgroups are NWAVES = 12 waves: wave = (head of the pass: wave / 4, 16-row sub-tile: wave % 4) in the forward / dQ roles --
// three waves per SIMD cover each other's LDS and MFMA latencies, which ONE wave per SIMD holding three heads' state could not
// (measured: 23.8 us against the per-head kernels' 20.0 at the training shape, 41 exposed LDS waits and 240 register moves per
// tile step) -- and waves 0..7 issue the DMAs of a tile pair (waves 0..3 the first tile, 4..7 the second; four 1-KiB pieces each).
constexpr int NWAVES = 12;

// K / V ring of the workgroup (b, g, query tile qt): tile pair t -> slot t % NS.
struct KvRing {
  const bf16 *kbase, *vbase;
  int ld_bytes, S, wave, nkt;
  int voff[4];
  char* smem;
  __device__ __forceinline__ void issue(int t) const {     // t >= nkt: a dummy (zero records)
    // waves 8..11 issue four zero-record DMAs into the scratch area: the same count on every wave, no branch around a DMA
    const bool real = wave < 8;
    char* slot = real ? smem + (t % NS) * PAIR + (wave >> 2) * TILE : smem + NS * PAIR + NS * EXTRA + BIAS_BYTES - (wave & 3) * 4096;
    const int rows = (real && t < nkt) ? S - t * 64 : 0;
    dma_tile((wave < 4 ? kbase : vbase) + (size_t)t * 64 * (ld_bytes >> 1), ld_bytes, rows, slot, wave & 3, voff);
  }
};

// ======================================================================================= backward: dQ role
__device__ __forceinline__ void dq_body(const bf16* __restrict__ qkv, const uint8_t* __restrict__ kmask, const bf16* __restrict__ dout,
                                        const float* __restrict__ lse, const float* __restrict__ delta, const float* __restrict__ ct,
                                        const float* __restrict__ st_, bf16* __restrict__ dqkv, const Geo& p, int qt, int part, int g,
                                        int b, char* smem) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hj = wave >> 2, sub = wave & 3;
  const int S = p.S, Spad = p.Spad, H = p.H, G = p.G;
  const int rep = H / G, LD = (H + 2 * G) * HD;
  const int h_first = g * rep + part * p.hp;
  float* sBias = (float*)(smem + NS * PAIR + NS * EXTRA);
  const uint8_t* mrow = kmask + (size_t)b * Spad;
  for (int i = threadIdx.x; i < Spad; i += NWAVES * 64) sBias[i] = mask_bias(mrow[i]);
  KvRing ring;
  ring.kbase = qkv + (size_t)b * S * LD + (H + g) * HD;
  ring.vbase = qkv + (size_t)b * S * LD + (H + G + g) * HD;
  ring.ld_bytes = LD * 2, ring.S = S, ring.wave = wave, ring.smem = smem;
  tile_offsets(ring.voff, LD * 2, wave & 3, lane);
  const int nkt = p.causal ? (qt + 1) : ((S + 63) >> 6);
  __builtin_assume(nkt >= 1);
  ring.nkt = nkt;
  const int qpos = qt * 64 + sub * 16 + (lane & 15);
  const int qc = min(qpos, S - 1);
  const int qp = lane >> 4;
  const float scale = p.scale, scale2 = p.scale * LOG2E;
  const int causal = p.causal;

  for (int h0 = 0; h0 < p.hp; h0 += 3) {
    const bool active = h0 + hj < p.hp;
    const int h = h_first + min(h0 + hj, p.hp - 1);
    bf16x8 qf[4], dof[4];
    load_row_frags(qf, qkv + (size_t)b * S * LD + h * HD, LD, qc, lane);
    load_row_frags(dof, dout + (size_t)b * S * (H * HD) + h * HD, H * HD, qc, lane);
    const float lse_q = lse[((size_t)b * H + h) * Spad + qc] * LOG2E;         // base-2 domain (attn_tiles.h: prob2)
    const float dl_q = delta[((size_t)b * H + h) * Spad + qc];
    f32x4 dq[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) dq[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);    // (as in fwd_body)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) ring.issue(t);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) need(qf[ks]), need(dof[ks]);
    need(lse_q), need(dl_q);
    for (int kt = 0; kt < nkt; ++kt) {
      wait_tile<4>();
      __builtin_amdgcn_s_barrier();
      ring.issue(kt + NS - 1);
      if (!active) continue;
      const char* sK = smem + (kt % NS) * PAIR;
      const char* sV = sK + TILE;
      const bool diag = causal && kt == qt;              // the only key tile that reaches past the block's first query
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
        const f32x4 kb = *(const f32x4*)(sBias + key0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float pr = prob2(a[r], scale2, kb[r] - lse_q);        // the per-head kernel's arithmetic: the same bits
          if (diag) pr = key0 + r <= qpos ? pr : 0.f;
          ds[st][r] = pr * (dp[r] - dl_q);
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
    wait_vm<0>();                         // (trailing dummy DMAs)
    // epilogue: dq * scale rounded to bf16 (what the per-head kernel stores), then the rotary embedding's backward
    if (qpos < S && active) {
      const size_t m = (size_t)b * S + qpos;
      bf16* drow = dqkv + m * LD + h * HD;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const f32x4 c4 = *(const f32x4*)(ct + m * 64 + nt * 16 + 4 * qp);
        const f32x4 s4 = *(const f32x4*)(st_ + m * 64 + nt * 16 + 4 * qp);
        const bf16x4 y1 = __builtin_convertvector(dq[nt] * scale, bf16x4);
        const bf16x4 y2 = __builtin_convertvector(dq[nt + 4] * scale, bf16x4);
        bf16x4 lo, hi;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float d1, d2;
          rope_pair_bwd_f((float)y1[r], (float)y2[r], c4[r], s4[r], d1, d2);
          lo[r] = (bf16)d1;
          hi[r] = (bf16)d2;
        }
        *(bf16x4*)(drow + nt * 16 + 4 * qp) = lo;
        *(bf16x4*)(drow + 64 + nt * 16 + 4 * qp) = hi;
      }
    }
  }
}

// ======================================================================================= backward: dK / dV role
// Waves 0..7: wave = (16-key sub-tile: wave % 4, query half of every 64-query tile: wave / 4).  The two halves' accumulators meet
// in LDS after the sweep (the ring's memory: 4 waves x (dK + dV) x 8 KiB = 64 KiB).  Waves 8..11 only keep the barriers.
__device__ __forceinline__ void dkv_body(const bf16* __restrict__ qkv, const uint8_t* __restrict__ kmask, const bf16* __restrict__ dout,
                                         const float* __restrict__ lse, const float* __restrict__ delta, const float* __restrict__ ct,
                                         const float* __restrict__ st_, bf16* __restrict__ dqkv, const Geo& p, int ktile, int g, int b,
                                         char* smem) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ksub = wave & 3, qh = (wave >> 2) & 1;
  const bool worker = wave < 8;
  const int S = p.S, Spad = p.Spad, H = p.H, G = p.G;
  const int rep = H / G, LD = (H + 2 * G) * HD;
  const bf16* kbase = qkv + (size_t)b * S * LD + (H + g) * HD;
  const bf16* vbase = qkv + (size_t)b * S * LD + (H + G + g) * HD;
  const int kpos = ktile * 64 + ksub * 16 + (lane & 15);
  const int kc = min(kpos, S - 1);
  const int qp = lane >> 4;
  const bool kvalid = kpos < S && kmask[(size_t)b * Spad + kc] != 0;
  const float kbias = kvalid ? 0.f : NEG_INF;
  const float scale = p.scale, scale2 = p.scale * LOG2E;
  const int causal = p.causal;
  bf16x8 kf[4], vf[4];
  load_row_frags(kf, kbase, LD, kc, lane);
  load_row_frags(vf, vbase, LD, kc, lane);
  int voff[4];
  tile_offsets(voff, (wave < 4 ? LD : H * HD) * 2, wave & 3, lane);

  f32x4 dk[8], dv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) dk[i] = dv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nqt = (S + 63) >> 6;
  const int q_first = causal ? ktile : 0;
  const int per_head = nqt - q_first;
  const int n_it = per_head * rep;                       // flattened (head, query tile) iteration space, head-major
  __builtin_assume(n_it >= 1);
  auto issue = [&](int it) {                 // it >= n_it: a dummy (zero records); waves 0..3: Q tile, 4..7: dO tile, + lse / delta
    const bool valid = worker && it < n_it;  // waves 8..11: zero-record DMAs into the scratch area (same count on every wave)
    const int itc = it < n_it ? it : 0;
    const int h = g * rep + itc / per_head, qtile = q_first + itc % per_head;
    const int rows = valid ? S - qtile * 64 : 0;
    char* slot = worker ? smem + (it % NS) * PAIR + (wave >> 2) * TILE : smem + NS * PAIR + NS * EXTRA + BIAS_BYTES - (wave & 3) * 4096;
    const bf16* org = wave < 4 ? qkv + ((size_t)b * S + qtile * 64) * LD + h * HD
                               : dout + ((size_t)b * S + qtile * 64) * (H * HD) + h * HD;
    dma_tile(org, (wave < 4 ? LD : H * HD) * 2, rows, slot, wave & 3, voff);
    // lse / delta of the slot's 64 queries: waves 0 / 1 (the other waves write duplicates: every wave has 5 DMAs per slot)
    const float* src = ((wave & 1) ? delta : lse) + ((size_t)b * H + h) * Spad + qtile * 64;
    char* fdst = smem + NS * PAIR + (worker ? (it % NS) * EXTRA + (wave & 1) * 256 + (wave >> 1) * 512 : NS * EXTRA + BIAS_BYTES);
    dma_floats64(src, valid, fdst, lane);
  };
  __builtin_amdgcn_sched_barrier(0);      // K / V fragment loads first, the DMAs behind them
#pragma unroll
  for (int t = 0; t < NS - 1; ++t) issue(t);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) need(kf[ks]), need(vf[ks]);
  need((int)kvalid);
  for (int it = 0; it < n_it; ++it) {
    const int qtile = q_first + it % per_head;
    wait_tile<5>();
    __builtin_amdgcn_s_barrier();
    issue(it + NS - 1);
    if (!worker) continue;
    const char* sQ = smem + (it % NS) * PAIR;
    const char* sdO = sQ + TILE;
    // per-element checks only where they can fail: the diagonal query tile (causal) and a tile that holds rows past the sequence
    const bool edge = (causal && qtile == ktile) || qtile * 64 + 63 >= S;
    const float* s_ld = (const float*)(smem + NS * PAIR + (it % NS) * EXTRA);      // [0,64) lse, [64,128) delta
    f32x4 pv[2], ds[2];
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
      const int qs = qh * 2 + q2;
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
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pr = prob2(a[r], scale2, kbias - l4[r] * LOG2E);
        float dsr = pr * (dp[r] - d4[r]);
        if (edge) {                          // (select BOTH: lse / delta of rows past S are whatever the allocation held, 0 * NaN = NaN)
          const bool ok = q0 + r < S && (!causal || kpos <= q0 + r);
          pr = ok ? pr : 0.f;
          dsr = ok ? dsr : 0.f;
        }
        pv[q2][r] = pr;
        ds[q2][r] = dsr;
      }
    }
    const bf16x8 pf = pack_pair(pv[0], pv[1]);
    const bf16x8 sf = pack_pair(ds[0], ds[1]);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      dv[nt] = mfma16(frag_tr_row(sdO, nt, qh, lane), pf, dv[nt]);     // dO^T, Q^T out of the token-major tiles
      dk[nt] = mfma16(frag_tr_row(sQ, nt, qh, lane), sf, dk[nt]);
    }
  }
  wait_vm<0>();                           // (trailing dummy DMAs)
  __builtin_amdgcn_s_barrier();           // every wave is done with the ring: its memory carries the second half's sums now
  f32x4* red = (f32x4*)smem + (size_t)ksub * (16 * 64);           // per key sub-tile: dk[8], dv[8] x 64 lanes
  if (worker && qh == 1) {
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      red[nt * 64 + lane] = dk[nt];
      red[(8 + nt) * 64 + lane] = dv[nt];
    }
  }
  __syncthreads();
  if (worker && qh == 0 && kpos < S) {
    const size_t m = (size_t)b * S + kpos;
    bf16* krow = dqkv + m * LD + (H + g) * HD;
    bf16* vrow = dqkv + m * LD + (H + G + g) * HD;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      dk[nt] = (dk[nt] + red[nt * 64 + lane]) * scale;
      dv[nt] = dv[nt] + red[(8 + nt) * 64 + lane];
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const f32x4 c4 = *(const f32x4*)(ct + m * 64 + nt * 16 + 4 * qp);
      const f32x4 s4 = *(const f32x4*)(st_ + m * 64 + nt * 16 + 4 * qp);
      bf16x4 lo, hi;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float d1, d2;
        rope_pair_bwd_f(dk[nt][r], dk[nt + 4][r], c4[r], s4[r], d1, d2);
        lo[r] = (bf16)d1;
        hi[r] = (bf16)d2;
      }
      *(bf16x4*)(krow + nt * 16 + 4 * qp) = lo;
      *(bf16x4*)(krow + 64 + nt * 16 + 4 * qp) = hi;
    }
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) *(bf16x4*)(vrow + nt * 16 + 4 * qp) = __builtin_convertvector(dv[nt], bf16x4);
  }
}

// Both roles in one grid: blocks [0, n_dkv) are dK / dV workgroups (long: all heads of the group x up to 4 query tiles), the rest
// dQ workgroups.
__global__ __launch_bounds__(NWAVES * 64, 1) void attn_bwd_gqa_kernel(const bf16* __restrict__ qkv, const uint8_t* __restrict__ kmask,
                                                                      const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                                      const float* __restrict__ delta, const float* __restrict__ ct,
                                                                      const float* __restrict__ st_, bf16* __restrict__ dqkv, Geo p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nt = (p.S + 63) >> 6, npg = p.B * p.G;
  const int n_dkv = nt * npg;
  int id = blockIdx.x, seq, pg;
  if (id < n_dkv) {
    place(id, npg, seq, pg);                             // seq = key tile: early key tiles (most query tiles) first
    dkv_body(qkv, kmask, dout, lse, delta, ct, st_, dqkv, p, seq, pg % p.G, pg / p.G, smem);
  } else {
    const int parts = (p.H / p.G) / p.hp;
    place(id - n_dkv, npg, seq, pg);
    dq_body(qkv, kmask, dout, lse, delta, ct, st_, dqkv, p, nt - 1 - seq / parts, seq % parts, pg % p.G, pg / p.G, smem);
  }
}

// heads per forward / dQ workgroup for a group of `rep` query heads (a pass = up to three heads = 12 waves)
inline int heads_per_block(int rep) {
  if (rep % 3 == 0) return 3;
  if (rep % 2 == 0) return 2;
  return rep;                                            // 5, 7, ...: all heads in one workgroup, passes of three
}

template <typename K>
bool set_lds(K kernel) {
  return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) == hipSuccess;
}

}  // namespace tasu_gqa

// The GQA kernels serve causal or non-causal attention with at least two query heads per kv head and Spad <= 4096.
extern "C" int tasu_attn_gqa_supported(int S, int H, int G) {
  return (G > 0 && H % G == 0 && H / G >= 2 && ((S + 63) & ~63) <= tasu_gqa::MASK_MAX) ? 1 : 0;
}

int tasu_attn_bwd_gqa_launch(const void* qkv, const uint8_t* key_mask, const void* dout, const float* lse, const float* delta,
                             const float* cos_tab, const float* sin_tab, void* dqkv, int B, int S, int H, int G, float scale, int causal,
                             hipStream_t stream) {
  using namespace tasu_gqa;
  Geo p{S, (S + 63) & ~63, H, G, B, heads_per_block(H / G), scale, causal};
  const int nt = (S + 63) >> 6, parts = (H / G) / p.hp;
  const dim3 grid(B * G * nt + B * G * nt * parts);
  static const bool ok = set_lds(attn_bwd_gqa_kernel);
  if (!ok) return TASU_ERR_LAUNCH;
  TASU_LAUNCH(attn_bwd_gqa_kernel, grid, dim3(NWAVES * 64), LDS_BYTES, stream, (const bf16*)qkv, key_mask, (const bf16*)dout, lse, delta,
              cos_tab, sin_tab, (bf16*)dqkv, p);
  return TASU_OK;
}
