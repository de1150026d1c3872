// Single-pass attention for sequences of at most 256 tokens (the alignment step's S = 256): the WHOLE K / V (forward, dQ) or
// Q / dO (dK / dV) of one (batch, head) is resident in LDS -- 2 x [256][128] bf16 = 128 KiB, fetched by LDS-DMA in bursts at
// kernel start -- so there is no key-tile loop, no online-softmax rescale chain and no per-tile barrier: a wave computes
// QK^T against every key it may see, ONE softmax, ONE P.V.  Replaces, for Spad <= 256, the tiled kernels of attention.hip /
// attention_gqa.hip behind the same entry points (SDPA inside Qwen2Attention.forward, transformers modeling_qwen2.py:150-172,
// reached from /root/reference/Multitask/model/ps-slm.py:530; rotary backward: modeling_qwen2.py:113-135 reversed).
//
//   forward   one workgroup = (batch, query head), 8 waves; wave w owns the 16-query units {w, 15 - w}: 17 causal key sub-tiles
//             per wave, every wave the same (the softmax is a per-wave chain of vector instructions: its length, not the MFMA
//             count, is what a wave's time follows).
//   backward  ONE launch, three roles, 4 waves of up to 512 registers each:
//             dQ role    (batch, query head): K / V resident, a wave owns four 16-query units {w, 7 - w, 8 + w, 15 - w} (equal
//                        causal work); delta = rowsum(dO . O) is computed in the kernel (no tasu_attn_bwd_prep launch), dq gets
//                        its scale, bf16 rounding and the rotary embedding's backward in the epilogue (as attention_gqa.hip);
//             dK/dV role (batch, query head, key half): Q / dO resident, a wave owns two 16-key units {w, 7 - w} of its half with
//                        K / V fragments and the fp32 dK^T / dV^T accumulators in registers (four units per wave need 256
//                        accumulator + ~300 vector registers: the two classes are capped at 256 each and the kernel spilled);
//                        fp32 partials PER QUERY HEAD go to memory and
//             kv_reduce_rope_kernel sums the H / G heads of a group, un-rotates dK and rounds once (tasu_rope_bwd's arithmetic).
//
// MEASURED (tools/micro/attn_sp_phases.hip, phase ablation of the first version): what a workgroup's time goes to is not the
// arithmetic.  Fragment-shaped global accesses -- a wave instruction = 16 rows x 64 B, the MFMA operand layout -- cost 7 us per
// workgroup for the dQ role's 52 loads per lane and 4-7 us for the epilogues' stores, against 0.7 us for the whole 128-KiB DMA
// burst and ~3 us of matrix-pipe time; the forward's softmax took 7 us on the wave that held the two longest rows.  So every
// operand that must be a register fragment (Q, dO) comes through LDS as well: DMA of whole rows, then ds_read_b128 out of the row
// image; the results leave through LDS images and full-row 16-byte stores; O (for delta) is read with row-contiguous loads.  The
// softmax's vector chain: the key mask is an additive bias row in LDS, the causal compare runs on the diagonal sub-tile only,
// exp2 with the scale folded in, and every wave gets the same number of elements.
// The MFMA operand plumbing (query / key index on lane & 15, score tiles feeding the next product from registers, transposed
// operands by ds_read_b64_tr_b16 out of the same row image) is attention.hip's; see attn_tiles.h.
#include "attn_tiles.h"
#include "../../include/tasu_hip.h"

namespace tasu_sp {

using namespace tasu_attn;

constexpr int IMG = 256 * 256;           // [256 tokens][128 d] bf16 "row" image (attn_tiles.h), 64 KiB
constexpr int MAXS = 256;
constexpr int AUX = 2 * IMG;             // byte offset of the small arrays behind the two images
constexpr int FWD_LDS = 2 * IMG + 1024;  // K, V (Q before it), key bias[256]
constexpr int BWD_LDS = 2 * IMG + 4096 + 64;   // the two images + bias / lse / delta rows; the dK / dV role's padded output tiles: 4 x 33,792 B
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

typedef __attribute__((address_space(3))) void lds_void;

struct Geo {
  int S, Spad, H, G, B;
  float scale;
  int causal;
#ifdef TASU_SP_ABL            // tools/micro/attn_sp_phases.hip: phase ablation (timing only, results are wrong), role selection, stamps
  int abl, id0;
  long long* stamps;          // [blocks][32] shader-clock stamps of wave 0 (diagnostic build only)
#endif
};
#ifdef TASU_SP_ABL
#define SP_SKIP(bit) ((p.abl & (bit)) != 0)
#define SP_ID0 p.id0
#define SP_STAMP(i)                                                                                         \
  do {                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 32 + (i)] = __builtin_readcyclecounter(); \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
  } while (0)
#else
#define SP_SKIP(bit) false
#define SP_ID0 0
#define SP_STAMP(i) do {} while (0)
#endif

template <int N>
__device__ __forceinline__ void wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else static_assert(N == 0, "add the immediate");
}
// every wave has waited for its own DMAs of the stage (wait_vm) and for its LDS accesses; the barrier makes all of them visible
__device__ __forceinline__ void stage_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}
// "the value is needed here" (attention_gqa.hip): placed behind the DMA issue, the compiler waits for the register loads issued
// before it with a counted vmcnt that leaves the DMAs in flight
template <typename T>
__device__ __forceinline__ void need(const T& v) {
  asm volatile("" ::"v"(v));
}
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

// Rows [row0, row0 + 4 * NP * NW) of a token-major matrix (row stride ld_bytes, `nrows` valid rows: the rest read as zeros
// through the descriptor's range) -> the same rows of a "row" image.  NW waves issue NP 1-KiB pieces each: piece
// p = wave * NP + i holds four rows; lane l writes LDS chunk l & 15 of its row, i.e. global chunk (l & 15) ^ swz(row).  row0 is a
// multiple of 16 (the swizzle's period).
template <int NP, int NW>
__device__ __forceinline__ void dma_rows(const bf16* mat, int ld_bytes, int nrows, int row0, char* img, int wave, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(NP % 4 == 0, "the pieces of a wave keep the swizzle period of 16 rows");
  constexpr int SPAN = 4 * NP * NW;
  const int left = nrows - row0;
  const int rows = left < SPAN ? left : SPAN;
  const unsigned nrec = rows > 0 ? (unsigned)(rows - 1) * (unsigned)ld_bytes + 256u : 0u;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(mat + (size_t)row0 * (ld_bytes >> 1)), 0, nrec, 0x00020000);
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int pc = wave * NP + i;
    const int r = pc * 4 + (lane >> 4);
    const int voff = r * ld_bytes + (((lane & 15) ^ swz(r)) << 4);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(img + row0 * 256 + pc * 1024), 16, voff, 0, 0, 0);
  }
#endif
}
// byte offset of 16-byte chunk c of row r in a row image
__device__ __forceinline__ int img_off(int r, int c) { return r * 256 + ((c ^ swz(r)) << 4); }

// Workgroup -> (batch, head): the H / G query heads of one (batch, kv group) get linear ids that are equal mod 8 -- one XCD under
// the round-robin placement, so that their reads of the shared K / V meet in one L2 (speed only).
__device__ __forceinline__ void place(int id, const Geo& p, int& b, int& h, int& g) {
  const int npg = p.B * p.G, rep = p.H / p.G;
  int j, pg;
  if ((npg & 7) == 0) {
    const int per = npg >> 3, x = id & 7, q = id >> 3;
    j = q / per;
    pg = (q % per) * 8 + x;
  } else {
    j = id / npg;
    pg = id % npg;
  }
  b = pg / p.G;
  g = pg % p.G;
  h = g * rep + j;
}

// additive key bias row (0 = attend, -inf = padded key or past the sequence) from the [B, Spad] byte mask: threads 0..255
__device__ __forceinline__ void write_key_bias(float* s_bias, uint8_t mbyte, int S) {
  if (threadIdx.x < MAXS) s_bias[threadIdx.x] = (mbyte != 0 && (int)threadIdx.x < S) ? 0.f : NEG_INF;
}

// ======================================================================================= forward
// scores of query unit u (16 rows, query on lane & 15) against key sub-tile st: a[u][st][r] <-> key st * 16 + 4 (lane >> 4) + r
template <int ST0, int ST1>
__device__ __forceinline__ void fwd_qk(f32x4 (&a)[2][16], const bf16x8 (&qf)[2][4], const char* sK, const int (&nst)[2], int lane) {
#pragma unroll
  for (int st = ST0; st < ST1; ++st) {
    if (st < nst[1]) {
      if (st < nst[0]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 kf = frag_row(sK, st, ks, lane);
          a[1][st] = mfma16(kf, qf[1][ks], a[1][st]);
          a[0][st] = mfma16(kf, qf[0][ks], a[0][st]);
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) a[1][st] = mfma16(frag_row(sK, st, ks, lane), qf[1][ks], a[1][st]);
      }
    }
  }
}

__global__ __launch_bounds__(512) void attn_sp_fwd_kernel(const bf16* __restrict__ qkv, const uint8_t* __restrict__ kmask,
                                                          bf16* __restrict__ out, float* __restrict__ lse, Geo p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;
  char* sV = smem + IMG;                               // holds Q until every wave has its query fragments
  float* s_bias = (float*)(smem + AUX);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int b, h, g;
  place(blockIdx.x, p, b, h, g);
  const int S = p.S, Spad = p.Spad, H = p.H, G = p.G, causal = p.causal;
  const float scale2 = p.scale * LOG2E;
  const int LD = (H + 2 * G) * HD;
  const bf16* qbase = qkv + (size_t)b * S * LD + h * HD;
  const bf16* kbase = qkv + (size_t)b * S * LD + (H + g) * HD;
  const bf16* vbase = qkv + (size_t)b * S * LD + (H + G + g) * HD;
  const int nstS = (S + 15) >> 4;
  const int qp = lane >> 4;
  int un[2], qpos[2], nst[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    un[u] = u == 0 ? wave : 15 - wave;
    qpos[u] = un[u] * 16 + (lane & 15);
    nst[u] = causal ? min(un[u] + 1, nstS) : nstS;      // key sub-tiles of the unit: wave-uniform, nst[0] <= nst[1] (units past S are masked at the store)
  }
  const uint8_t mbyte = kmask[(size_t)b * Spad + min((int)threadIdx.x, Spad - 1)];
  __builtin_amdgcn_sched_barrier(0);
  const int Sd = SP_SKIP(8) ? 0 : S;
  dma_rows<4, 8>(qbase, LD * 2, Sd, 0, sV, wave, lane);
  dma_rows<4, 8>(qbase, LD * 2, Sd, 128, sV, wave, lane);
  dma_rows<4, 8>(kbase, LD * 2, Sd, 0, sK, wave, lane);
  dma_rows<4, 8>(kbase, LD * 2, Sd, 128, sK, wave, lane);
  need((int)mbyte);
  write_key_bias(s_bias, mbyte, S);
  wait_vm<8>();
  stage_barrier();                                    // Q (and the bias row)
  bf16x8 qf[2][4];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[u][ks] = frag_row(sV, un[u], ks, lane);
  stage_barrier();                                    // every wave holds its query fragments: V may overwrite Q
  dma_rows<4, 8>(vbase, LD * 2, Sd, 0, sV, wave, lane);
  dma_rows<4, 8>(vbase, LD * 2, Sd, 128, sV, wave, lane);

  f32x4 a[2][16];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int st = 0; st < 16; ++st) a[u][st] = f32x4{0.f, 0.f, 0.f, 0.f};
  wait_vm<12>();
  stage_barrier();                                    // K rows 0..127
  if (!SP_SKIP(1)) fwd_qk<0, 8>(a, qf, sK, nst, lane);
  wait_vm<8>();
  stage_barrier();                                    // K rows 128..255
  if (!SP_SKIP(1)) fwd_qk<8, 16>(a, qf, sK, nst, lane);

  // one softmax over all visible keys of the row, in the base-2 domain: t = s * scale * log2(e) + bias[key]
  bf16x8 pf[2][8];
  float l_tot[2], m_row[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    float tmax = NEG_INF;
    if (SP_SKIP(2)) {
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[u][j] = pack_pair(a[u][2 * j], a[u][2 * j + 1]);
      l_tot[u] = 1.f, m_row[u] = 0.f;
      continue;
    }
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      if (st < nst[u]) {
        const int key0 = st * 16 + 4 * qp;
        const f32x4 kb = *(const f32x4*)(s_bias + key0);
#pragma unroll
        for (int r = 0; r < 4; ++r) a[u][st][r] = __builtin_fmaf(a[u][st][r], scale2, kb[r]);
        if (causal && st == un[u]) {                   // the diagonal sub-tile: keys past the query
#pragma unroll
          for (int r = 0; r < 4; ++r) a[u][st][r] = key0 + r <= qpos[u] ? a[u][st][r] : NEG_INF;
        }
        tmax = fmaxf(tmax, fmaxf(fmaxf(a[u][st][0], a[u][st][1]), fmaxf(a[u][st][2], a[u][st][3])));
      }
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_use = (tmax == NEG_INF) ? 0.f : tmax;
    float psum = 0.f;
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      if (st < nst[u]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = exp2_fast(a[u][st][r] - m_use);
          a[u][st][r] = e;
          psum += e;
        }
      } else {
        a[u][st] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    psum += __shfl_xor(psum, 16, 64);
    psum += __shfl_xor(psum, 32, 64);
    l_tot[u] = psum;
    m_row[u] = tmax;
#pragma unroll
    for (int j = 0; j < 8; ++j) pf[u][j] = pack_pair(a[u][2 * j], a[u][2 * j + 1]);
  }

  f32x4 o[2][8];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int i = 0; i < 8; ++i) o[u][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  wait_vm<0>();
  stage_barrier();                                    // V; nobody reads K any more
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (2 * j < nst[1] && !SP_SKIP(4)) {
      if (2 * j < nst[0]) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
          const bf16x8 vf = frag_tr_row(sV, nt, j, lane);
          o[1][nt] = mfma16(vf, pf[1][j], o[1][nt]);
          o[0][nt] = mfma16(vf, pf[0][j], o[0][nt]);
        }
      } else {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) o[1][nt] = mfma16(frag_tr_row(sV, nt, j, lane), pf[1][j], o[1][nt]);
      }
    }
  }
  // the wave's 2 x 16 output rows through its own 8 KiB of the K image, then whole 256-byte rows to memory
  char* stg = sK + wave * 8192;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const float inv = l_tot[u] > 0.f ? 1.f / l_tot[u] : 0.f;
    const int r = u * 16 + (lane & 15);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const f32x4 v = o[u][nt] * inv;
      *(bf16x4*)(stg + img_off(r, nt * 2 + (qp >> 1)) + (qp & 1) * 8) = __builtin_convertvector(v, bf16x4);
    }
    if (qp == 0 && qpos[u] < S) lse[((size_t)b * H + h) * Spad + qpos[u]] = l_tot[u] > 0.f ? (m_row[u] + __log2f(l_tot[u])) * LN2 : 0.f;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the wave's own LDS writes; no other wave touches this region)
  __builtin_amdgcn_sched_barrier(0);
  if (!SP_SKIP(16)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = i * 4 + (lane >> 4);               // row of the wave's 32: unit i >> 2 (static), row r & 15
      const int q = un[i >> 2] * 16 + (r & 15);
      const bf16x8 v = *(const bf16x8*)(stg + img_off(r, lane & 15));
      if (q < S) *(bf16x8*)(out + ((size_t)b * S + q) * (H * HD) + h * HD + (lane & 15) * 8) = v;
    }
  }
}

// ======================================================================================= backward
// 16-row units of a wave in the dQ role: {w, 7 - w, 8 + w, 15 - w}, ascending.
__device__ __forceinline__ int unit_of(int wave, int k) {
  return k == 0 ? wave : k == 1 ? 7 - wave : k == 2 ? 8 + wave : 15 - wave;
}

// Sum over the 16 lanes of a DPP row (lanes 16 g .. 16 g + 15) by shifted adds inside the vector ALU (row_shr 1, 2, 4, 8 with zero
// fill): lane 15 of the row ends up with the total.  __shfl_xor compiles to ds_bpermute -- an LDS round trip per step, and the
// sixteen four-step chains of a workgroup's delta rows took 4.3 us that way (stamps of tools/micro/attn_sp_phases.hip).
__device__ __forceinline__ float row16_sum(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
#endif
  return v;
}

// delta = rowsum(dO . O) of a 16-row unit: dO out of its LDS row image, O by row-contiguous loads (ov: this lane's chunk
// l & 15 of rows 4 i + (l >> 4), i = 0..3).  The lane with chunk 15 stores the row's sum.
__device__ __forceinline__ void unit_delta(const char* s_dO, int unit, const bf16x8 (&ov)[4], float* s_dl, int S, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = unit * 16 + i * 4 + (lane >> 4);
    const bf16x8 dv = *(const bf16x8*)(s_dO + img_off(r, lane & 15));
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += (float)dv[e] * (float)ov[i][e];
    acc = row16_sum(acc);                               // lane 15 of the row's 16 lanes holds the sum
    if ((lane & 15) == 15) s_dl[r] = r < S ? acc : 0.f;
  }
}
__device__ __forceinline__ void load_o_rows(bf16x8 (&ov)[4], const bf16* obase, int ld, int unit, int S, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = min(unit * 16 + i * 4 + (lane >> 4), S - 1);
    ov[i] = *(const bf16x8*)(obase + (size_t)r * ld + (lane & 15) * 8);
  }
}

// ---- dQ role.  One 32-key step j for the query units k = KMIN..3 (the units whose causal range still reaches the step).
template <int KMIN>
__device__ __forceinline__ void dq_step(int j, f32x4 (&dq)[4][8], const bf16x8 (&qf)[4][4], const bf16x8 (&dof)[4][4], const float (&lse2)[4],
                                        const float (&dl_q)[4], const int (&qpos)[4], const int (&jdiag)[4], const char* sK, const char* sV,
                                        const float* s_bias, float scale2, int causal, int lane) {
  const int qp = lane >> 4;
  bf16x8 f[4];
  f32x4 a[4][2], dp[4][2];
#pragma unroll
  for (int k = KMIN; k < 4; ++k)
#pragma unroll
    for (int s = 0; s < 2; ++s) a[k][s] = dp[k][s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int st = 2 * j + s;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 kf = frag_row(sK, st, ks, lane);
      const bf16x8 vf = frag_row(sV, st, ks, lane);
#pragma unroll
      for (int k = KMIN; k < 4; ++k) {
        a[k][s] = mfma16(kf, qf[k][ks], a[k][s]);
        dp[k][s] = mfma16(vf, dof[k][ks], dp[k][s]);
      }
    }
  }
  f32x4 kb[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) kb[s] = *(const f32x4*)(s_bias + (2 * j + s) * 16 + 4 * qp);
#pragma unroll
  for (int k = KMIN; k < 4; ++k) {
    f32x4 ds[2];
    const bool diag = causal && j == jdiag[k];           // the step that holds the unit's diagonal (wave-uniform)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int key0 = (2 * j + s) * 16 + 4 * qp;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = __builtin_fmaf(a[k][s][r], scale2, kb[s][r] - lse2[k]);      // p = exp2(s * scale * log2 e + bias - lse * log2 e)
        if (diag) t = key0 + r <= qpos[k] ? t : NEG_INF;
        ds[s][r] = exp2_fast(t) * (dp[k][s][r] - dl_q[k]);
      }
    }
    f[k] = pack_pair(ds[0], ds[1]);
  }
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    const bf16x8 kt = frag_tr_row(sK, nt, j, lane);          // K^T out of the token-major K image
#pragma unroll
    for (int k = KMIN; k < 4; ++k) dq[k][nt] = mfma16(kt, f[k], dq[k][nt]);
  }
}

__device__ __forceinline__ void dq_role(const bf16* __restrict__ qkv, const uint8_t* __restrict__ kmask, const bf16* __restrict__ dout,
                                        const bf16* __restrict__ out, const float* __restrict__ lse, const float* __restrict__ ct,
                                        const float* __restrict__ st_, bf16* __restrict__ dqkv, const Geo& p, int b, int h, int g, char* smem) {
  if (SP_SKIP(64)) return;
  SP_STAMP(0);
  char* sK = smem;                                       // Q before K
  char* sV = smem + IMG;                                 // dO before V
  float* s_bias = (float*)(smem + AUX);
  float* s_dl = s_bias + MAXS;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int S = p.S, Spad = p.Spad, H = p.H, G = p.G, causal = p.causal;
  const float scale = p.scale, scale2 = p.scale * LOG2E;
  const int LD = (H + 2 * G) * HD;
  const bf16* qbase = qkv + (size_t)b * S * LD + h * HD;
  const bf16* kbase = qkv + (size_t)b * S * LD + (H + g) * HD;
  const bf16* vbase = qkv + (size_t)b * S * LD + (H + G + g) * HD;
  const bf16* dobase = dout + (size_t)b * S * (H * HD) + h * HD;
  const bf16* obase = out + (size_t)b * S * (H * HD) + h * HD;
  const int njS = (S + 31) >> 5;
  const int qp = lane >> 4;
  int un[4], qpos[4], jmax[4], jdiag[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    un[k] = unit_of(wave, k);
    qpos[k] = un[k] * 16 + (lane & 15);
    jdiag[k] = un[k] >> 1;
    jmax[k] = causal ? min(jdiag[k] + 1, njS) : njS;     // 32-key steps of the unit: wave-uniform, ascending in k (units past S are masked at the store)
  }
  // round 1: Q and dO as row images (for the fragments), O by row-contiguous loads (for delta), the key mask
  const uint8_t mbyte = kmask[(size_t)b * Spad + min((int)threadIdx.x, Spad - 1)];
  bf16x8 ov[4][4];
  float lse2[4], dl_q[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    load_o_rows(ov[k], obase, H * HD, un[k], S, lane);
    lse2[k] = lse[((size_t)b * H + h) * Spad + min(qpos[k], S - 1)];
  }
  __builtin_amdgcn_sched_barrier(0);
  SP_STAMP(1);
  const int Sd = SP_SKIP(8) ? 0 : S;
  dma_rows<8, 4>(qbase, LD * 2, Sd, 0, sK, wave, lane);
  dma_rows<8, 4>(qbase, LD * 2, Sd, 128, sK, wave, lane);
  dma_rows<8, 4>(dobase, H * HD * 2, Sd, 0, sV, wave, lane);
  dma_rows<8, 4>(dobase, H * HD * 2, Sd, 128, sV, wave, lane);
  SP_STAMP(2);
  need((int)mbyte);
  write_key_bias(s_bias, mbyte, S);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int i = 0; i < 4; ++i) need(ov[k][i]);
    need(lse2[k]);
    lse2[k] *= LOG2E;
  }
  SP_STAMP(3);
  wait_vm<0>();
  SP_STAMP(4);
  stage_barrier();                                       // Q, dO
  SP_STAMP(5);
  bf16x8 qf[4][4], dof[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[k][ks] = frag_row(sK, un[k], ks, lane);
      dof[k][ks] = frag_row(sV, un[k], ks, lane);
    }
    unit_delta(sV, un[k], ov[k], s_dl, S, lane);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) dl_q[k] = s_dl[qpos[k]];   // (written by this wave: LDS accesses of a wave are in order)
  SP_STAMP(6);
  stage_barrier();                                       // every wave holds its fragments: K / V may overwrite Q / dO
  SP_STAMP(7);
  // round 2: K and V
  dma_rows<8, 4>(kbase, LD * 2, Sd, 0, sK, wave, lane);
  dma_rows<8, 4>(vbase, LD * 2, Sd, 0, sV, wave, lane);
  dma_rows<8, 4>(kbase, LD * 2, Sd, 128, sK, wave, lane);
  dma_rows<8, 4>(vbase, LD * 2, Sd, 128, sV, wave, lane);
  f32x4 dq[4][8];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int i = 0; i < 8; ++i) dq[k][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto run = [&](int j0, int j1) {
    for (int j = j0; j < j1; ++j) {
      if (j < jmax[0]) dq_step<0>(j, dq, qf, dof, lse2, dl_q, qpos, jdiag, sK, sV, s_bias, scale2, causal, lane);
      else if (j < jmax[1]) dq_step<1>(j, dq, qf, dof, lse2, dl_q, qpos, jdiag, sK, sV, s_bias, scale2, causal, lane);
      else if (j < jmax[2]) dq_step<2>(j, dq, qf, dof, lse2, dl_q, qpos, jdiag, sK, sV, s_bias, scale2, causal, lane);
      else dq_step<3>(j, dq, qf, dof, lse2, dl_q, qpos, jdiag, sK, sV, s_bias, scale2, causal, lane);
    }
  };
  const int jend = jmax[3];                              // (the largest unit reaches furthest)
  SP_STAMP(8);
  wait_vm<16>();
  stage_barrier();                                       // rows 0..127 of K and V
  SP_STAMP(9);
  if (!SP_SKIP(1)) run(0, min(jend, 4));
  SP_STAMP(10);
  wait_vm<0>();
  stage_barrier();                                       // rows 128..255
  SP_STAMP(11);
  if (!SP_SKIP(1)) run(4, jend);
  SP_STAMP(12);
  stage_barrier();                                       // every wave is done with K / V: the images carry the output tiles now
  SP_STAMP(13);

  // epilogue: dq * scale rounded to bf16 (the tiled kernels' store) into the wave's 16 KiB of LDS, then per whole row the rotary
  // embedding's backward and 16-byte stores (lane = chunk c of a row; the rotation pairs chunk c with chunk c ^ 8)
  char* stg = smem + wave * 16384;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = k * 16 + (lane & 15);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
      *(bf16x4*)(stg + img_off(r, nt * 2 + (qp >> 1)) + (qp & 1) * 8) = __builtin_convertvector(dq[k][nt] * scale, bf16x4);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  if (!SP_SKIP(16)) {
    const int c = lane & 15, cl = c & 7;
#pragma unroll
    for (int k = 0; k < 4; ++k) {                        // one unit at a time: its 4 x 4 table loads are in flight together
      f32x4 c0[4], c1[4], s0[4], s1[4];
      bf16x8 own[4], oth[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = k * 16 + i * 4 + (lane >> 4);
        const size_t m = (size_t)b * S + min(un[k] * 16 + (r & 15), S - 1);
        c0[i] = *(const f32x4*)(ct + m * 64 + cl * 8), c1[i] = *(const f32x4*)(ct + m * 64 + cl * 8 + 4);
        s0[i] = *(const f32x4*)(st_ + m * 64 + cl * 8), s1[i] = *(const f32x4*)(st_ + m * 64 + cl * 8 + 4);
        own[i] = *(const bf16x8*)(stg + img_off(r, c));
        oth[i] = *(const bf16x8*)(stg + img_off(r, c ^ 8));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = un[k] * 16 + i * 4 + (lane >> 4);
        bf16x8 res;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float cs = e < 4 ? c0[i][e] : c1[i][e - 4], sn = e < 4 ? s0[i][e] : s1[i][e - 4];
          const float y1 = (float)(c < 8 ? own[i][e] : oth[i][e]), y2 = (float)(c < 8 ? oth[i][e] : own[i][e]);
          float d1, d2;
          rope_pair_bwd_f(y1, y2, cs, sn, d1, d2);
          res[e] = (bf16)(c < 8 ? d1 : d2);
        }
        if (q < S) *(bf16x8*)(dqkv + ((size_t)b * S + q) * LD + h * HD + c * 8) = res;
      }
    }
  }
  SP_STAMP(14);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SP_STAMP(15);
}

// ---- dK / dV role.  One 32-query step i for the key units k = 0..KN-1 (the units whose keys the step's queries may see).
template <int KN>
__device__ __forceinline__ void dkv_step(int i, f32x4 (&dk)[2][8], f32x4 (&dv)[2][8], const bf16x8 (&kf)[2][4], const bf16x8 (&vf)[2][4],
                                         const int (&kpos)[2], const float (&kbias)[2], const int (&imin)[2], const char* sQ, const char* sdO,
                                         const float* s_lse2, const float* s_dl, float scale2, int causal, int lane) {
  const int qp = lane >> 4;
  bf16x4 ph[2][2], sh[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int qs = 2 * i + s;
    f32x4 a[2], dp[2];
#pragma unroll
    for (int k = 0; k < KN; ++k) a[k] = dp[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 qfr = frag_row(sQ, qs, ks, lane);
      const bf16x8 dofr = frag_row(sdO, qs, ks, lane);
#pragma unroll
      for (int k = 0; k < KN; ++k) {
        a[k] = mfma16(qfr, kf[k][ks], a[k]);
        dp[k] = mfma16(dofr, vf[k][ks], dp[k]);
      }
    }
    const int q0 = qs * 16 + 4 * qp;
    const f32x4 l4 = *(const f32x4*)(s_lse2 + q0);       // lse * log2(e); +inf for rows past S (p = 0 there)
    const f32x4 d4 = *(const f32x4*)(s_dl + q0);
#pragma unroll
    for (int k = 0; k < KN; ++k) {
      const bool diag = causal && i == imin[k];            // the step that holds the unit's diagonal (wave-uniform)
      f32x4 pv, ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = __builtin_fmaf(a[k][r], scale2, kbias[k] - l4[r]);
        if (diag) t = kpos[k] <= q0 + r ? t : NEG_INF;
        pv[r] = exp2_fast(t);
        ds[r] = pv[r] * (dp[k][r] - d4[r]);
      }
      ph[k][s] = __builtin_convertvector(pv, bf16x4);
      sh[k][s] = __builtin_convertvector(ds, bf16x4);
    }
  }
  bf16x8 pf[2], sf[2];
#pragma unroll
  for (int k = 0; k < KN; ++k) {
    pf[k] = __builtin_shufflevector(ph[k][0], ph[k][1], 0, 1, 2, 3, 4, 5, 6, 7);
    sf[k] = __builtin_shufflevector(sh[k][0], sh[k][1], 0, 1, 2, 3, 4, 5, 6, 7);
  }
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    const bf16x8 dot = frag_tr_row(sdO, nt, i, lane);          // dO^T, Q^T out of the token-major images
    const bf16x8 qt = frag_tr_row(sQ, nt, i, lane);
#pragma unroll
    for (int k = 0; k < KN; ++k) {
      dv[k][nt] = mfma16(dot, pf[k], dv[k][nt]);
      dk[k][nt] = mfma16(qt, sf[k], dk[k][nt]);
    }
  }
}

// kh: key half of the head (keys [128 kh, 128 kh + 128)); the wave's two 16-key units: 8 kh + {w, 7 - w}.
__device__ __forceinline__ void dkv_role(const bf16* __restrict__ qkv, const uint8_t* __restrict__ kmask, const bf16* __restrict__ dout,
                                         const bf16* __restrict__ out, const float* __restrict__ lse, float* __restrict__ dk_part,
                                         float* __restrict__ dv_part, const Geo& p, int b, int h, int g, int kh, char* smem) {
  SP_STAMP(0);
  char* sQ = smem;
  char* sdO = smem + IMG;
  float* s_lse2 = (float*)(smem + AUX);
  float* s_dl = s_lse2 + MAXS;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int S = p.S, Spad = p.Spad, H = p.H, G = p.G, causal = p.causal;
  const float scale = p.scale, scale2 = p.scale * LOG2E;
  const int LD = (H + 2 * G) * HD;
  const bf16* qbase = qkv + (size_t)b * S * LD + h * HD;
  const bf16* kbase = qkv + (size_t)b * S * LD + (H + g) * HD;
  const bf16* vbase = qkv + (size_t)b * S * LD + (H + G + g) * HD;
  const bf16* dobase = dout + (size_t)b * S * (H * HD) + h * HD;
  const bf16* obase = out + (size_t)b * S * (H * HD) + h * HD;
  const int niS = (S + 31) >> 5;
  const int qp = lane >> 4;
  int un[2], kpos[2], imin[2];
  float kbias[2];
  bf16x8 kf[2][4], vf[2][4];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    un[k] = kh * 8 + (k == 0 ? wave : 7 - wave);
    kpos[k] = un[k] * 16 + (lane & 15);
    imin[k] = causal ? (un[k] >> 1) : 0;                 // first 32-query step that sees the unit: wave-uniform, ascending in k
    const uint8_t mb = kmask[(size_t)b * Spad + min(kpos[k], Spad - 1)];
    kbias[k] = (kpos[k] < S && mb != 0) ? 0.f : NEG_INF;
    const int kc = min(kpos[k], S - 1);
    load_row_frags(kf[k], kbase, LD, kc, lane);          // 16 fragment-shaped loads per lane: they run under the DMA burst
    load_row_frags(vf[k], vbase, LD, kc, lane);
  }
  // O by row-contiguous loads for delta (this wave: query rows 64 w .. 64 w + 63), lse; then the DMAs.  Under the causal mask the
  // upper key half never meets the first 128 queries: their rows are neither fetched nor reduced.
  const bool lo_rows = !(causal && kh == 1);
  const bool my_rows = lo_rows || wave >= 2;
  bf16x8 ov[4][4];
  float lse_r = 0.f;
  if (my_rows) {
#pragma unroll
    for (int k = 0; k < 4; ++k) load_o_rows(ov[k], obase, H * HD, wave * 4 + k, S, lane);
    lse_r = lse[((size_t)b * H + h) * Spad + min(wave * 64 + lane, S - 1)];
  }
  __builtin_amdgcn_sched_barrier(0);
  SP_STAMP(1);
  dma_rows<8, 4>(qbase, LD * 2, S, 128, sQ, wave, lane);                     // the late queries first: every key sees them
  dma_rows<8, 4>(dobase, H * HD * 2, S, 128, sdO, wave, lane);
  dma_rows<8, 4>(qbase, LD * 2, lo_rows ? S : 0, 0, sQ, wave, lane);
  dma_rows<8, 4>(dobase, H * HD * 2, lo_rows ? S : 0, 0, sdO, wave, lane);
  SP_STAMP(2);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) need(kf[k][ks]), need(vf[k][ks]);
    need(kbias[k]);
  }
  if (my_rows) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i) need(ov[k][i]);
    need(lse_r);
    const int row = wave * 64 + lane;
    s_lse2[row] = row < S ? lse_r * LOG2E : __builtin_inff();
  }
  f32x4 dk[2][8], dv[2][8];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int i = 0; i < 8; ++i) dk[k][i] = dv[k][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto run = [&](int i_hi, int i_lo) {                   // steps i_hi - 1 .. i_lo
    for (int i = i_hi - 1; i >= i_lo; --i) {
      if (i >= imin[1]) dkv_step<2>(i, dk, dv, kf, vf, kpos, kbias, imin, sQ, sdO, s_lse2, s_dl, scale2, causal, lane);
      else dkv_step<1>(i, dk, dv, kf, vf, kpos, kbias, imin, sQ, sdO, s_lse2, s_dl, scale2, causal, lane);
    }
  };
  const int ifirst = imin[0];                            // (the earliest unit is seen by the most steps)
  SP_STAMP(3);
  wait_vm<16>();
  SP_STAMP(4);
  stage_barrier();                                       // rows 128..255 of Q and dO
  SP_STAMP(5);
  if (my_rows && wave >= 2) {
#pragma unroll
    for (int k = 0; k < 4; ++k) unit_delta(sdO, wave * 4 + k, ov[k], s_dl, S, lane);
  }
  stage_barrier();                                       // their lse / delta rows
  SP_STAMP(6);
  if (!SP_SKIP(1)) run(niS, max(ifirst, 4));
  SP_STAMP(7);
  wait_vm<0>();
  stage_barrier();                                       // rows 0..127
  SP_STAMP(8);
  if (my_rows && wave < 2) {
#pragma unroll
    for (int k = 0; k < 4; ++k) unit_delta(sdO, wave * 4 + k, ov[k], s_dl, S, lane);
  }
  stage_barrier();
  SP_STAMP(9);
  if (!SP_SKIP(1)) run(min(niS, 4), ifirst);
  SP_STAMP(10);
  stage_barrier();                                       // every wave is done with Q / dO: the images carry the output tiles now
  SP_STAMP(11);

  // fp32 partials of this query head: [M, H * 128] (dk scaled), summed over the group's heads by kv_reduce_rope_kernel.  Through
  // LDS (a [16][528 B] tile per unit and tensor: rows padded by one 16-byte slot) and out as whole 512-byte rows.
  constexpr int TROW = 528, TILE = 16 * TROW;
  static_assert(4 * 4 * TILE <= BWD_LDS, "output tiles of the four waves");
  char* stg = smem + wave * (4 * TILE);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      *(f32x4*)(stg + (2 * k) * TILE + (lane & 15) * TROW + (nt * 16 + 4 * qp) * 4) = dk[k][nt] * scale;
      *(f32x4*)(stg + (2 * k + 1) * TILE + (lane & 15) * TROW + (nt * 16 + 4 * qp) * 4) = dv[k][nt];
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  if (!SP_SKIP(16)) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = i * 2 + (lane >> 5), c = lane & 31;
        const int key = un[k] * 16 + r;
        const f32x4 a = *(const f32x4*)(stg + (2 * k) * TILE + r * TROW + c * 16);
        const f32x4 v = *(const f32x4*)(stg + (2 * k + 1) * TILE + r * TROW + c * 16);
        if (key < S) {
          const size_t off = ((size_t)b * S + key) * (H * HD) + h * HD + c * 4;
          *(f32x4*)(dk_part + off) = a;
          *(f32x4*)(dv_part + off) = v;
        }
      }
    }
  }
  SP_STAMP(12);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SP_STAMP(13);
}

// grid: [0, n) dK / dV workgroups of the lower key halves (the longest), [n, 2n) dQ workgroups, [2n, 3n) dK / dV of the upper key
// halves (short under the causal mask: they fill the tail), n = B * H
__global__ __launch_bounds__(256, 1) void attn_sp_bwd_kernel(const bf16* __restrict__ qkv, const uint8_t* __restrict__ kmask,
                                                             const bf16* __restrict__ dout, const bf16* __restrict__ out,
                                                             const float* __restrict__ lse, const float* __restrict__ ct,
                                                             const float* __restrict__ st_, bf16* __restrict__ dqkv,
                                                             float* __restrict__ dk_part, float* __restrict__ dv_part, Geo p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = p.B * p.H;
  int id = blockIdx.x + SP_ID0, b, h, g;
  if (id < n) {
    place(id, p, b, h, g);
    dkv_role(qkv, kmask, dout, out, lse, dk_part, dv_part, p, b, h, g, 0, smem);
  } else if (id < 2 * n) {
    place(id - n, p, b, h, g);
    dq_role(qkv, kmask, dout, out, lse, ct, st_, dqkv, p, b, h, g, smem);
  } else {
    place(id - 2 * n, p, b, h, g);
    dkv_role(qkv, kmask, dout, out, lse, dk_part, dv_part, p, b, h, g, 1, smem);
  }
}

// k and v blocks of dqkv from the per-query-head fp32 partials [M, H * 128]: sum over the H / G heads of the group (in head
// order), un-rotate dK, round to bf16 once.  8 threads per (token, kv head, k | v): the chunk pair (c .. c + 7, 64 + c .. + 7).
__global__ __launch_bounds__(256) void kv_reduce_rope_kernel(bf16* __restrict__ dqkv, const float* __restrict__ dk_part,
                                                            const float* __restrict__ dv_part, const float* __restrict__ ct,
                                                            const float* __restrict__ st, int M, int H, int G) {
  const int upt = 2 * G * 8;                             // threads per token
  const int tpb = 256 / upt;                             // tokens per block (upt <= 256: G <= 16)
  const int tl = threadIdx.x / upt, u = threadIdx.x - tl * upt;
  const int m = blockIdx.x * tpb + tl;
  if (tl >= tpb || m >= M) return;
  const int rep = H / G, LD = (H + 2 * G) * HD;
  const int c = (u & 7) * 8, gk = u >> 3;
  const bool isk = gk < G;
  const int g = isk ? gk : gk - G;
  const float* src = (isk ? dk_part : dv_part) + (size_t)m * (H * HD) + (size_t)g * rep * HD;
  float y1[8], y2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) y1[j] = y2[j] = 0.f;
  for (int r = 0; r < rep; ++r) {
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
  bf16* row = dqkv + (size_t)m * LD + (H + gk) * HD;
  bf16x8 lo, hi;
  if (isk) {
    const f32x4 c0 = *(const f32x4*)(ct + (size_t)m * 64 + c), c1 = *(const f32x4*)(ct + (size_t)m * 64 + c + 4);
    const f32x4 s0 = *(const f32x4*)(st + (size_t)m * 64 + c), s1 = *(const f32x4*)(st + (size_t)m * 64 + c + 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float cs = j < 4 ? c0[j] : c1[j - 4], sn = j < 4 ? s0[j] : s1[j - 4];
      float d1, d2;
      rope_pair_bwd_f(y1[j], y2[j], cs, sn, d1, d2);
      lo[j] = (bf16)d1;
      hi[j] = (bf16)d2;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      lo[j] = (bf16)y1[j];
      hi[j] = (bf16)y2[j];
    }
  }
  *(bf16x8*)(row + c) = lo;
  *(bf16x8*)(row + 64 + c) = hi;
}

template <typename K>
bool set_lds(K kernel, int bytes) {
  return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
}

}  // namespace tasu_sp

#ifndef TASU_SP_ABL
// The single-pass kernels serve causal or bidirectional attention over at most 256 (padded) positions, up to 16 kv heads.
extern "C" int tasu_attn_sp_supported(int S, int H, int G) {
  return (S > 0 && G > 0 && G <= 16 && H % G == 0 && ((S + 63) & ~63) <= tasu_sp::MAXS) ? 1 : 0;
}

int tasu_attn_sp_fwd_launch(const void* qkv, const uint8_t* key_mask, void* out, float* lse, int B, int S, int H, int G, float scale,
                            int causal, hipStream_t stream) {
  using namespace tasu_sp;
  Geo p{S, (S + 63) & ~63, H, G, B, scale, causal};
  static const bool ok = set_lds(attn_sp_fwd_kernel, FWD_LDS);
  if (!ok) return TASU_ERR_LAUNCH;
  TASU_LAUNCH(attn_sp_fwd_kernel, dim3(B * H), dim3(512), FWD_LDS, stream, (const bf16*)qkv, key_mask, (bf16*)out, lse, p);
  return TASU_OK;
}

int tasu_attn_sp_bwd_launch(const void* qkv, const uint8_t* key_mask, const void* dout, const void* out, const float* lse,
                            const float* cos_tab, const float* sin_tab, void* dqkv, float* dk_part, float* dv_part, int B, int S, int H,
                            int G, float scale, int causal, hipStream_t stream) {
  using namespace tasu_sp;
  Geo p{S, (S + 63) & ~63, H, G, B, scale, causal};
  static const bool ok = set_lds(attn_sp_bwd_kernel, BWD_LDS);
  if (!ok) return TASU_ERR_LAUNCH;
  TASU_LAUNCH(attn_sp_bwd_kernel, dim3(3 * B * H), dim3(256), BWD_LDS, stream, (const bf16*)qkv, key_mask, (const bf16*)dout,
              (const bf16*)out, lse, cos_tab, sin_tab, (bf16*)dqkv, dk_part, dv_part, p);
  const int M = B * S, tpb = 256 / (2 * G * 8);
  TASU_LAUNCH(kv_reduce_rope_kernel, dim3((M + tpb - 1) / tpb), dim3(256), 0, stream, (bf16*)dqkv, (const float*)dk_part,
              (const float*)dv_part, cos_tab, sin_tab, M, H, G);
  return TASU_OK;
}
#endif
