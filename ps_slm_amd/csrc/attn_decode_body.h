// Body of the single-token cache attention of tasu_attn_decode's kernel (decode.hip).  (A header because round 2's persistent
// decode-layer kernel shared it; that kernel was measured slower than the per-GEMM launches and removed in round 3.)
#pragma once
#include "common.h"
#include "stream_body.h"

namespace tasu_attn_dec {
constexpr int HD = 128;
#ifdef TASU_ATTN_TRACE
// debug build (tools/attn_trace.py): wall-clock stamps of workgroup (0, 0) at the body's phase boundaries
__device__ unsigned long long g_attn_trace[16];
#define TASU_ATTN_STAMP(k) do { if (threadIdx.x == 0 && row == 0 && g == 0) tasu_attn_dec::g_attn_trace[k] = wall_clock64(); } while (0)
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
//   phase 1b softmax statistics per head (wave h), probabilities (bf16-rounded like the prefill kernel) back to LDS as bf16
//   phase 2  P.V on the matrix cores too: a wave takes 32 keys at a time; their V rows (prefetched into registers at the very
//            start, 16-byte loads, a wave instruction = 4 whole 256-B rows) go through a wave-private 8-KiB LDS row image and
//            come back as MFMA operands by hardware transpose reads (ds_read_b64_tr_b16: lane = head dim, k = keys) --
//            8 MFMAs per 32 keys give out[head][128 dims] with the heads on the lanes' l & 15: no shuffles.  (The first
//            form walked the keys with scalar FMAs, 8 dims x REP heads per lane, and folded the key quarters with 96
//            shuffles: 3.9 of the kernel's 10.5 us, measured with tools/attn_trace.py.)
//   phase 3  cross-wave sum through LDS, 1/l, 8-byte bf16 stores
// keys in [kstart[row], lens[row]) are visible.  ctx <= MAX_CTX.
constexpr int MAX_CTX = 2048;
constexpr int DEC_NW = 8;
// LDS floats the body needs for a context of ctx positions
constexpr int VSTAGE_FLOATS = DEC_NW * 2048;              // per wave: 32 keys x 256 B
__host__ __device__ constexpr int rup4(int x) { return (x + 3) & ~3; }
__host__ __device__ constexpr int attn_decode_lds_floats(int rep, int ctx) {
  // scores | 1/l | cache rows | bf16 probabilities [rep][round_up(ctx, 32)] | V row images (the partial outputs reuse them)
  return rup4(rep * ctx) + rup4(rep + 1) + rup4(ctx) + rup4(rep * ((ctx + 31) & ~31) / 2) + VSTAGE_FLOATS;
}
static_assert(DEC_NW * 8 * HD <= VSTAGE_FLOATS, "partial outputs fit the V staging area");

// MFMA operand (16 rows = head dims nt*16 + (lane&15); k-slot j of lane group q' = lane>>4 <-> key (j < 4 ? 4q'+j : 16+4q'+j-4)
// of the 32-key block) out of a key-major row image [32 keys][256 B], 16-byte chunks swizzled by (chunk ^ (key & 15)), with
// two hardware transpose reads (the scheme of attention.hip's frag_tr_row).  EXEC must be all ones.
typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ bf16x8 v_frag_tr(const char* img, int nt, int lane) {
  const int gq = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int r0 = 4 * gq + q, r1 = r0 + 16;
  const int ch = nt * 2 + (pp >> 1), inner = (pp & 1) * 8;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + r0 * 256 + ((ch ^ (r0 & 15)) << 4) + inner));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + r1 * 256 + ((ch ^ (r1 & 15)) << 4) + inner));
  union { s16x4 s[2]; bf16x8 b; } u;
  u.s[0] = lo;
  u.s[1] = hi;
  return u.b;
}
// row, g: the (beam row, kv group) this workgroup serves; sp: LDS; WT: write-through output (stream_body.h)
template <int REP, bool WT>
__device__ __forceinline__ void attn_decode_body(float* sp, int row, int g, const bf16* __restrict__ qkv, const bf16* __restrict__ kc,
                                                 const bf16* __restrict__ vc, const int32_t* __restrict__ row_index,
                                                 const int32_t* __restrict__ kstart, const int32_t* __restrict__ lens,
                                                 bf16* __restrict__ out, int H, int G, int ctx, float scale, int out_frag) {
  // [REP][ctx] scores | [REP] 1/l | [ctx] physical cache row of every position | bf16 probabilities | V row images / partials
  const int ctxp = (ctx + 31) & ~31;
  float* sc = sp;
  float* linv = sc + rup4(REP * ctx);
  int* prow = (int*)(linv + rup4(REP + 1));
  bf16* pb = (bf16*)(prow + rup4(ctx));
  float* part = (float*)pb + rup4(REP * ctxp / 2);         // [DEC_NW][REP][128] partial outputs, after the V images are consumed
  char* vimg = (char*)part + (threadIdx.x >> 6) * 8192;    // this wave's V row image
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
  // ---- V prefetch: the wave's first VPRE 32-key blocks (contexts up to VPRE * 256 keys entirely) are requested NOW, so that
  // their latency runs under phase 1 (K loads, score MFMAs) and the softmax instead of behind them.  Lane: key 4 * i + lq of
  // the block, 16-byte chunk l15 of its 256-B row.
  constexpr int VPRE = 2;
  const bf16* vbase = vc + (size_t)k0 * W + g * HD + l15 * 8;
  const int nblk32 = (nk + 31) >> 5;
  auto load_vblock = [&](bf16x8 (&v)[8], int blk) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kcl = min(blk * 32 + 4 * i + lq, nk - 1);
      v[i] = *(const bf16x8*)(vbase + ((size_t)prow[kcl] * ctx + kcl) * W);
    }
  };
  // (blocks / chunks past the end are not requested: a workgroup's loads are L1-bandwidth bound -- 64 B per clock, and eight
  // waves prefetching two V blocks and three K chunks each would move 230 KB for a 228-key context that has 118)
  bf16x8 vpre[VPRE][8];
#pragma unroll
  for (int it = 0; it < VPRE; ++it)
    if (wave + it * DEC_NW < nblk32) load_vblock(vpre[it], wave + it * DEC_NW);
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
    if (wave + it * DEC_NW < nchunk) {
      const int kcl = min((wave + it * DEC_NW) * 16 + l15, nk - 1);
      const bf16* kp = kbase + ((size_t)prow[kcl] * ctx + kcl) * W;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) kpre[it][s4] = *(const bf16x8*)(kp + s4 * 32);
    }
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
    // the first 256 keys of the head travel in registers (four independent LDS reads per lane instead of a dependent loop)
    constexpr int SU = 4;
    float x[SU];
    float m = -__builtin_inff();
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int i = lane + 64 * u;
      x[u] = i < nk ? sc[h * ctx + i] : -__builtin_inff();
      m = fmaxf(m, x[u]);
    }
    for (int i = lane + 64 * SU; i < nk; i += 64) m = fmaxf(m, sc[h * ctx + i]);
    m = wave_max(m);
    float l = 0.f;
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int i = lane + 64 * u;
      if (i < nblk32 * 32) {                                // keys past nk (to the 32-key boundary) weigh zero
        const float p = i < nk ? __expf(x[u] - m) : 0.f;
        pb[h * ctxp + i] = (bf16)p;
        l += p;
      }
    }
    for (int i = lane + 64 * SU; i < nblk32 * 32; i += 64) {
      const float p = i < nk ? __expf(sc[h * ctx + i] - m) : 0.f;
      pb[h * ctxp + i] = (bf16)p;
      l += p;
    }
    l = wave_sum(l);
    if (lane == 0) linv[h] = l > 0.f ? 1.f / l : 0.f;
  }
  __syncthreads();
  TASU_ATTN_STAMP(5);
  // ---- phase 2: P.V   o[nt][r] = out[head l15][dim nt*16 + 4*lq + r] over this wave's key blocks
  f32x4 o[8];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) o[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto block_pv = [&](const bf16x8 (&v)[8], int blk) {
    // the block's 32 V rows -> this wave's row image (key-major, swizzled), then back as transposed MFMA operands
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = 4 * i + lq;
      *(bf16x8*)(vimg + r * 256 + ((l15 ^ (r & 15)) << 4)) = v[i];
    }
    // probabilities of head l15 in the operands' k-slot order: keys 4 lq .. +3 and 16 + 4 lq .. +3 of the block
    bf16x8 pf = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (l15 < REP) {
      const bf16x4 lo = *(const bf16x4*)(pb + l15 * ctxp + blk * 32 + 4 * lq), hi = *(const bf16x4*)(pb + l15 * ctxp + blk * 32 + 16 + 4 * lq);
      pf = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the wave's own LDS writes have landed (wave-private image)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) o[nt] = mfma16(v_frag_tr(vimg, nt, lane), pf, o[nt]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // reads done before the next block overwrites the image
    __builtin_amdgcn_wave_barrier();
  };
#pragma unroll
  for (int it = 0; it < VPRE; ++it)
    if (wave + it * DEC_NW < nblk32) block_pv(vpre[it], wave + it * DEC_NW);
  for (int blk = wave + VPRE * DEC_NW; blk < nblk32; blk += DEC_NW) {
    bf16x8 v[8];
    load_vblock(v, blk);
    block_pv(v, blk);
  }
  TASU_ATTN_STAMP(6);
  __syncthreads();                                          // every wave is done with its V image: the partials reuse the area
  if (l15 < REP) {
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) *(f32x4*)(part + (wave * REP + l15) * HD + nt * 16 + 4 * lq) = o[nt];
  }
  __syncthreads();
  TASU_ATTN_STAMP(7);
  for (int e = threadIdx.x; e < REP * HD / 4; e += 64 * DEC_NW) {
    const int h = e / (HD / 4), d = (e - h * (HD / 4)) * 4;
    f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w2 = 0; w2 < DEC_NW; ++w2) sum += *(const f32x4*)(part + (w2 * REP + h) * HD + d);
    const int n = (g * REP + h) * HD + d;                  // first of 4 columns of the [M, H * 128] attention output
    // out_frag: the o projection's A operand in fragment order (csrc/gemm_stream.hip); row = row % 64 of its 64-row chunk
    const size_t off = out_frag ? ((size_t)(row >> 6) * 64 * (H * HD)) +
                                      ((((size_t)(n >> 5) * 4 + ((row & 63) >> 4)) * 64 + ((n & 31) >> 3) * 16 + (row & 15)) << 3) + (n & 7)
                                : (size_t)row * (H * HD) + n;
    const float li = linv[h];
    bf16x4 ov;
#pragma unroll
    for (int r = 0; r < 4; ++r) ov[r] = (bf16)(sum[r] * li);
    tasu_stream::st_out<WT>((bf16x4*)(out + off), ov);
  }
  TASU_ATTN_STAMP(8);
}
}  // namespace tasu_attn_dec
