// Body of the weight-streaming decode GEMM (see gemm_stream.hip for the design notes) of the one-GEMM kernels of
// gemm_stream.hip.  (A header because round 2's persistent decode-layer kernel shared it; removed in round 3)
#pragma once
#include "common.h"

#ifndef TASU_TOUCH_MASK
#define TASU_TOUCH_MASK 0
#endif

namespace tasu_stream {

enum { E_BF16 = 0, E_RESID = 1, E_SWIGLU = 2, E_QKV = 3, E_SLAB = 4 };
constexpr int NW = 8;                      // waves per workgroup
#ifdef TASU_STREAM_TRACE
// debug build (make trace; tools/stream_trace.py): wall-clock stamps of workgroup 0's first thread after every tile
__device__ unsigned long long g_stream_trace[32];
#define TASU_STREAM_STAMP(k) do { if (threadIdx.x == 0 && bx == 0 && by == 0 && bz == 0 && (k) < 32) g_stream_trace[k] = wall_clock64(); } while (0)
#else
#define TASU_STREAM_STAMP(k) do { } while (0)
#endif
int cu_count();                            // compute units of the current device (gemm_stream.hip)

struct Args {
  const bf16* A;          // [M, lda] activations
  const bf16* W;          // [*, ldw] weights, K contiguous
  void* C;                // output (bf16 or fp32, see epilogues); E_SLAB: fp32 slabs [ksplit][64 rows][N]
  const float* R;         // E_RESID: residual [M, ldc] fp32
  const bf16* bias;       // [N] or null
  int M, N, K, lda, ldw, ldc;
  int tiles;              // column tiles per K range
  int I;                  // E_SWIGLU: first "up" row of W
  int a_frag, w_frag;     // operands in fragment order (see the header comment)
  int out_frag;           // E_SWIGLU: act is written in fragment order (K of its consumer = N)
  // E_QKV
  int H, G, ctx;
  const float* cos_t;
  const float* sin_t;
  bf16* kc;
  bf16* vc;
  const int32_t* pos;
  int gx, gy, gz;         // the virtual grid [column-tile walkers][K ranges][row splits] (grid_position)
  // The post-attention RMSNorm without a launch (round 5): E_RESID with ssq_out also writes yw = bf16(nw . C) (the consumer's
  // operand; yw_frag: fragment order) and, per column tile, the sum of squares of its 16 values of every row: ssq_out[tile][64].
  // E_SWIGLU with ssq_in sums the n_part partials of a row, rstd = rsqrt(sum / K + eps), and scales its accumulators by it -- rstd
  // is a per-row scalar, so (nw . x . rstd) W^T = rstd . ((nw . x) W^T).
  const float* nw;
  bf16* yw;
  float* ssq_out;
  const float* ssq_in;
  int n_part, yw_frag;
  float eps;
  int kstep0, slab0;      // E_SLAB, ragged K (tasu_gemm_stream_slabs: K = n * range + a shorter last range, launched apart): the
                          // launch's first k-step (of 32) inside K and its first slab; 0 otherwise
};

// Position of linear workgroup `lin` in the virtual grid (gx, gy, gz); false: no work (the launch is rounded up).  Workgroups
// are placed round-robin on the 8 XCDs, each with its own L2.  The gz = 2 row halves of one (column walker, K range) read the
// SAME weight tiles: they are made neighbours on one XCD (lin and lin + 8), so that the second read of a tile hits that XCD's
// L2 instead of going to HBM again.
__device__ __forceinline__ bool grid_position(int lin, int gx, int gy, int gz, int& bx, int& by, int& bz) {
  const int xcd = lin & 7, q = lin >> 3;
  bz = q % gz;
  const int r = (q / gz) * 8 + xcd;
  bx = r % gx;
  by = r / gx;
  return r < gx * gy;
}

// element (row, c) of a [<= 64, K] activation in fragment order (gemm_stream.hip header; the same map as norm.hip's frag_offset)
__device__ __forceinline__ size_t frag_index(int row, int c) {
  return ((((size_t)(c >> 5) * 4 + (row >> 4)) * 64 + ((c & 31) >> 3) * 16 + (row & 15)) << 3) + (c & 7);
}

// first weight row (of 16) that lane group row r = l & 15 of tile t reads
template <int EPI>
__device__ __forceinline__ int weight_row(const Args& p, int t, int r) {
  if (EPI == E_SWIGLU) return (r < 8 ? 0 : p.I) + t * 8 + (r & 7);
  if (EPI == E_QKV) {
    const int rot_tiles = (p.H + p.G) * 8;                 // q and k heads: 8 tiles of (8 + 8) paired columns each
    if (t < rot_tiles) return (t >> 3) * 128 + (t & 7) * 8 + (r & 7) + (r >= 8 ? 64 : 0);
    return (p.H + p.G) * 128 + (t - rot_tiles) * 16 + r;   // v heads: 16 plain columns
  }
  return t * 16 + r;
}

// Store of a result another workgroup will read.  WT: write-through at agent scope (global_store ... sc1, what an agent-scope atomic
// store compiles to): the line does not stay dirty in this XCD's L2, so a workgroup on another XCD that reads the address for the
// first time after a grid barrier gets it from memory -- no L2 write-back / invalidate fences at the barrier.  The compiler does
// not count these stores in its vmcnt bookkeeping; the barrier code waits for vmcnt(0) explicitly.
template <bool WT, typename T>
__device__ __forceinline__ void st_out(T* dst, T v) {
  static_assert(sizeof(T) == 16 || sizeof(T) == 8 || sizeof(T) == 2, "st_out: 2-, 8- or 16-byte values");
  if constexpr (!WT) {
    *dst = v;
  } else if constexpr (sizeof(T) == 16) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");   // (s_nop: the store reads its data registers late)
  } else if constexpr (sizeof(T) == 8) {
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
  } else {
    const unsigned bits = __builtin_bit_cast(unsigned short, v);
    asm volatile("global_store_short %0, %1, off sc1" ::"v"(dst), "v"(bits) : "memory");
  }
}

// MT = 16-row tiles of the activations a workgroup owns (4 = all 64 rows; 2 / 1: the rows are split over the grid's z, for GEMMs
// with too few column tiles to occupy the chip -- a workgroup's traffic is its (MT * 16 + columns) x K operand bytes, and with
// 1-2 column tiles per workgroup the 64 activation rows dominate it).
// FRAG: both operands in fragment order (compile-time: a run-time layout test inside the load lambdas splits the ring loop
// into branches across which the compiler drains vmcnt).
// PN (E_RESID / E_SWIGLU / E_QKV, KS <= 8): a layer norm travels with this GEMM (Args::ssq_out / ssq_in); a template flag so that
// the other kernels carry none of it (at KS = 14 every register counts).
// WT: outputs are stored write-through at agent scope (st_out), for a consumer on another XCD inside the SAME launch (round 2's
// persistent layer-loop kernel); the kernels of gemm_stream.hip pass false.
// bx / nbx, by, bz: the workgroup's position in the (virtual) grid [column-tile walkers][K ranges][row splits];
// red: LDS, 2 * NW * MT * 256 floats -- partial tiles [2 buffers][NW waves][MT row tiles][64 lanes] f32x4 -- + NW * 64 floats (rsq).
template <int KS, int EPI, int MT, bool FRAG, bool WT, bool PN = false>
__device__ __forceinline__ void stream_gemm_body(const Args& p, float* __restrict__ red, int bx, int nbx, int by, int bz) {
  const int mt0 = bz * MT;                              // first row tile of this workgroup
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int krange = NW * KS * 32;
  const int k0 = p.kstep0 * 32 + by * krange + wave * (KS * 32) + lq * 8;      // this lane's first k of step 0

  // ---- activations: this wave's K slice of all 64 rows, as MFMA B operands (rows beyond M are clamped; masked at the store)
  bf16x8 a[MT][KS];
  const int cg0 = p.kstep0 + (by * NW + wave) * KS;        // this wave's first global k-step
  // FIRST TOUCH (compile-time experiment, -DTASU_TOUCH_MASK=<bit per EPI>): the ~32 workgroups of an XCD read the same
  // activation image at the same moment and the L2 does not merge concurrent misses of one line.  With the flag each workgroup first
  // pulls ITS share of the image through the XCD's L2 (requested here, ahead of the first weight tiles); the operand loads follow
  // once those have arrived (late_acts), and hit.
  constexpr bool TOUCH = FRAG && ((TASU_TOUCH_MASK >> EPI) & 1) != 0;
  if constexpr (TOUCH) {
    const int nper = ((int)gridDim.x + 7) >> 3, rank = (int)blockIdx.x >> 3;
    const int total16 = (p.K >> 5) * 4 * 64;              // 16-byte pieces of [K / 32][4 row tiles][64 lanes][8]
    const int per = (total16 + nper - 1) / nper;
    for (int i = rank * per + (int)threadIdx.x; i < min((rank + 1) * per, total16); i += 64 * NW) {
      const bf16x8 v = *(const bf16x8*)(p.A + (size_t)i * 8);
      asm volatile("" ::"v"(v));
    }
  }
  auto load_acts = [&]() {
    if (FRAG) {
#pragma unroll
      for (int c = 0; c < KS; ++c)
#pragma unroll
        for (int t = 0; t < MT; ++t) a[t][c] = *(const bf16x8*)(p.A + (((size_t)(cg0 + c) * 4 + mt0 + t) * 64 + lane) * 8);
    } else {
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const bf16* ar = p.A + (size_t)min((mt0 + t) * 16 + l15, p.M - 1) * p.lda + k0;
#pragma unroll
        for (int c = 0; c < KS; ++c) a[t][c] = *(const bf16x8*)(ar + c * 32);
      }
    }
  };
  auto late_acts = [&]() {
    if constexpr (TOUCH) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      load_acts();
    }
  };
  if constexpr (!TOUCH) load_acts();

  // E_SWIGLU behind a pre-normed o projection: the rows' sums of squares from the producer's per-tile partials.  K / 16 = 16 KS
  // partials per row; wave w requests tiles w, w + 8, ... for the workgroup's 16 MT rows (lane = row; MT = 2: two lane halves share the
  // tiles) NOW, ahead of
  // the weight ring, sums them once the first weights are in flight (rsq_publish) and leaves the sum in LDS; the finishing waves add
  // the eight wave sums in wave order after the first tile's barrier.
  constexpr int ROWS = MT * 16, LPR = 64 / ROWS;        // lanes per row: MT = 2 splits a wave's tiles over two lane halves
  constexpr int QN = KS * 2 / LPR;
  float qv[QN];
  float* rsq = red + 2 * NW * MT * 256;                 // [NW][64]: lane = half * ROWS + row
  if constexpr (PN && (EPI == E_SWIGLU || EPI == E_QKV)) {
    const float* sp = p.ssq_in + mt0 * 16 + (lane & (ROWS - 1));
    const int t0 = wave + NW * (lane / ROWS);
#pragma unroll
    for (int j = 0; j < QN; ++j) qv[j] = sp[(size_t)(t0 + NW * LPR * j) * 64];
  }
  auto rsq_publish = [&]() {
    if constexpr (PN && (EPI == E_SWIGLU || EPI == E_QKV)) {
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < QN; ++j) q += qv[j];
      rsq[wave * 64 + lane] = q;
    }
  };

  const int ntl = (p.tiles - bx + nbx - 1) / nbx;     // tiles this workgroup walks
  auto tile_of = [&](int i) { return bx + min(i, ntl - 1) * nbx; };   // clamped: loads past the end re-read
  const int ksteps_all = p.K >> 5;
  auto load_w = [&](bf16x8 (&w)[KS], int i) {
    if (FRAG) {
      const bf16* wr = p.W + (((size_t)tile_of(i) * ksteps_all + cg0) * 64 + lane) * 8;
#pragma unroll
      for (int c = 0; c < KS; ++c) w[c] = __builtin_nontemporal_load((const bf16x8*)(wr + c * 512));
    } else {
      const bf16* wr = p.W + (size_t)min(weight_row<EPI>(p, tile_of(i), l15), (EPI == E_SWIGLU ? 2 * p.I : p.N) - 1) * p.ldw + k0;
#pragma unroll
      for (int c = 0; c < KS; ++c) w[c] = __builtin_nontemporal_load((const bf16x8*)(wr + c * 32));
    }
  };

  // Epilogue operands that do not depend on the GEMM (residual tile; q|k|v bias, RoPE factors, cache position).  With one or two
  // tiles per workgroup (q|k|v, o) they are requested together with the weights, ahead of the MFMAs, so that the epilogue does
  // not start with another memory round trip; in the ring they are loaded when the tile is finished (a load issued behind
  // three tiles of weight loads would make its consumer wait for all of them: vmcnt counts in order).
  struct Epi {
    f32x4 r4, cs, sn;
    bf16x4 b4;
    int pos;
  };
  auto load_epi = [&](int i) {
    Epi ep{};
    if (wave >= MT || i >= ntl) return ep;
    const int t = tile_of(i), m = (mt0 + wave) * 16 + l15;
    if (EPI == E_RESID) {
      const int n = t * 16 + 4 * lq;
      if (m < p.M && n + 4 <= p.N) ep.r4 = *(const f32x4*)(p.R + (size_t)m * p.ldc + n);
      if (PN && n + 4 <= p.N) ep.cs = *(const f32x4*)(p.nw + n);          // (cs: unused by this epilogue otherwise)
    }
    if (EPI == E_QKV) {
      const int rot_tiles = (p.H + p.G) * 8;
      const int mc = min(m, p.M - 1);
      ep.pos = p.pos[mc];
      if (t < rot_tiles) {
        const int c0 = (t & 7) * 8 + 4 * (lq & 1);
        ep.cs = *(const f32x4*)(p.cos_t + (size_t)mc * 64 + c0);
        ep.sn = *(const f32x4*)(p.sin_t + (size_t)mc * 64 + c0);
        if (p.bias) ep.b4 = *(const bf16x4*)(p.bias + (t >> 3) * 128 + c0 + (lq >= 2 ? 64 : 0));
      } else if (p.bias) {
        ep.b4 = *(const bf16x4*)(p.bias + (p.H + p.G) * 128 + (t - rot_tiles) * 16 + 4 * lq);
      }
    }
    return ep;
  };

  auto finish = [&](int i, const Epi& ep) {
    // ---- cross-wave sum + epilogue of tile i (waves 0..MT-1: row tile = mt0 + wave); called after the barrier of tile i
    if (wave >= MT || i >= ntl) return;
    const int buf = i & 1, t = tile_of(i);
    f32x4 s = *(const f32x4*)(red + (((buf * NW + 0) * MT + wave) * 64 + lane) * 4);
#pragma unroll
    for (int w2 = 1; w2 < NW; ++w2) s += *(const f32x4*)(red + (((buf * NW + w2) * MT + wave) * 64 + lane) * 4);
    const int rt = mt0 + wave;                           // row tile (0..3) of this wave's results
    const int m = rt * 16 + l15;
    // s[r] = C[m][tile column 4 * lq + r]
    if (EPI == E_SLAB) {
      // row-major [K range][64 rows][N]: 64-byte pieces here (spread over ~250 workgroups), so that the row-wise finish, which
      // runs on few CUs, reads whole contiguous rows (against per-tile slabs a row is 16 bytes every 256: 13 of its 18 us)
      float* slab = (float*)p.C + (size_t)(p.slab0 + by) * 64 * p.N;
      st_out<WT>((f32x4*)(slab + (size_t)m * p.N + t * 16 + 4 * lq), s);
      return;
    }
    if (EPI == E_SWIGLU) {
      // lanes lq < 2 hold gate columns t*8 + 4*lq + r, lanes lq + 2 the up values of the same columns
      if constexpr (PN) {
        float q = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2)
#pragma unroll
          for (int h = 0; h < LPR; ++h) q += rsq[w2 * 64 + h * ROWS + wave * 16 + l15];
        s *= rsqrtf(q / (float)p.K + p.eps);
      }
      f32x4 u;
#pragma unroll
      for (int r = 0; r < 4; ++r) u[r] = __shfl_xor(s[r], 32, 64);
      if (lq < 2 && (m < p.M || p.out_frag)) {
        const int n = t * 8 + 4 * lq;
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16)(bf16_round(silu_f(bf16_round(s[r]))) * bf16_round(u[r]));
        if (p.out_frag) {
          // element (m, n .. n+3) of the consumer's A operand: k-step n / 32, lane group (n % 32) / 8, row tile = wave
          st_out<WT>((bf16x4*)((bf16*)p.C + ((((size_t)(n >> 5) * 4 + rt) * 64 + ((n & 31) >> 3) * 16 + l15) << 3) + (n & 7)), o);
        } else {
          bf16* dst = (bf16*)p.C + (size_t)m * p.ldc + n;
          if (n + 4 <= p.N) {
            *(bf16x4*)dst = o;
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (n + r < p.N) dst[r] = o[r];
          }
        }
      }
      return;
    }
    if (EPI == E_QKV) {
      if constexpr (PN) {                                  // the input norm's rstd (the slab finish left bf16(w . x) + partials)
        float q = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2)
#pragma unroll
          for (int h = 0; h < LPR; ++h) q += rsq[w2 * 64 + h * ROWS + wave * 16 + l15];
        s *= rsqrtf(q / (float)p.K + p.eps);
      }
      const int rot_tiles = (p.H + p.G) * 8;
      const int W = p.G * 128;
      bf16* out = (bf16*)p.C + (size_t)m * p.ldc;
      if (t < rot_tiles) {
        // lanes lq < 2: low-half columns c0 + 4*lq + r of head hh; lanes lq + 2: their partners (+64)
        const int hh = t >> 3, c0 = (t & 7) * 8 + 4 * (lq & 1);
        const int col = hh * 128 + c0 + (lq >= 2 ? 64 : 0);
        bf16x4 mine;
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[r] = (bf16)(s[r] + (p.bias ? (float)ep.b4[r] : 0.f));
        f32x4 x1, x2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float other = __shfl_xor((float)mine[r], 32, 64);
          x1[r] = lq < 2 ? (float)mine[r] : other;       // low half
          x2[r] = lq < 2 ? other : (float)mine[r];       // high half
        }
        if (m < p.M) {
          const f32x4 cs = ep.cs, sn = ep.sn;
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16)(lq < 2 ? x1[r] * cs[r] - x2[r] * sn[r] : x2[r] * cs[r] + x1[r] * sn[r]);
          st_out<WT>((bf16x4*)(out + col), o);
          if (hh >= p.H) {
            const size_t slot = ((size_t)m * p.ctx + ep.pos) * W;
            st_out<WT>((bf16x4*)(p.kc + slot + (hh - p.H) * 128 + c0 + (lq >= 2 ? 64 : 0)), o);
          }
        }
      } else if (m < p.M) {
        const int c = (t - rot_tiles) * 16 + 4 * lq;           // column inside the v block
        const int col = (p.H + p.G) * 128 + c;
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16)(s[r] + (p.bias ? (float)ep.b4[r] : 0.f));
        st_out<WT>((bf16x4*)(out + col), o);
        st_out<WT>((bf16x4*)(p.vc + ((size_t)m * p.ctx + ep.pos) * W + c), o);
      }
      return;
    }
    // E_BF16 / E_RESID
    const int n = t * 16 + 4 * lq;
    if (m >= p.M || n >= p.N) return;
    if (EPI == E_RESID) {
      float* dst = (float*)p.C + (size_t)m * p.ldc + n;
      const float* rs = p.R + (size_t)m * p.ldc + n;
      if (n + 4 <= p.N) {
        const f32x4 old = ep.r4;
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = old[r] + bf16_round(s[r]);
        st_out<WT>((f32x4*)dst, o);
        if constexpr (PN) {
          bf16x4 yv;
#pragma unroll
          for (int r = 0; r < 4; ++r) yv[r] = (bf16)(ep.cs[r] * o[r]);
          st_out<WT>((bf16x4*)(p.yw_frag ? p.yw + frag_index(m, n) : p.yw + (size_t)m * p.N + n), yv);
          // the row's 16 columns of this tile sit in the four lanes l15 + 16 * {0..3} (all active: N is a multiple of 16 here)
          float q = o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3];
          q += __shfl_xor(q, 16, 64);
          q += __shfl_xor(q, 32, 64);
          if (lq == 0) p.ssq_out[(size_t)t * 64 + (m & 63)] = q;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < p.N) dst[r] = rs[r] + bf16_round(s[r]);
      }
    } else {
      bf16* dst = (bf16*)p.C + (size_t)m * p.ldc + n;
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (bf16)(s[r] + (p.bias && n + r < p.N ? (float)p.bias[n + r] : 0.f));
      if (n + 4 <= p.N) {
        *(bf16x4*)dst = o;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < p.N) dst[r] = o[r];
      }
    }
  };

  auto compute = [&](const bf16x8 (&w)[KS], int i, const Epi* pre) {
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < KS; ++c)
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t] = mfma16(w[c], a[t][c], acc[t]);
    const int buf = i & 1;
#pragma unroll
    for (int t = 0; t < MT; ++t) *(f32x4*)(red + (((buf * NW + wave) * MT + t) * 64 + lane) * 4) = acc[t];
    // my partial tile is in LDS; everybody's is after the barrier.  Raw s_barrier: the weight loads of the next tiles stay
    // in flight across it (a __syncthreads() would drain vmcnt).  Buffer (i & 1) is written again at tile i + 2, which every
    // wave reaches only after the barrier of tile i + 1, i.e. after all reads of tile i.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (pre) finish(i, *pre);
    else finish(i, load_epi(i));
  };

  if constexpr (KS > 8) {
    // K = 3584 in ONE range (Qwen2.5-7B's q|k|v, o and gate|up: 14 k-steps per wave; MT = 2, one workgroup per CU = 256 registers
    // per lane, 112 of them activations).  A tile's 14 KiB per wave travel as two HALVES of 7 k-steps on a ring of three half-tile
    // register sets (84 registers): one tile per wave in flight behind the one being multiplied (8 waves x 14 KiB per CU), the
    // same branch-free ring as below with the tile boundary on every second slot -- three tiles per trip.
    static_assert(KS % 2 == 0, "two halves");
    constexpr int KH = KS / 2;
    auto load_h = [&](bf16x8 (&w)[KH], int hidx) {
      const int i = hidx >> 1, part = hidx & 1;
      if (FRAG) {
        const bf16* wr = p.W + (((size_t)tile_of(i) * ksteps_all + cg0 + part * KH) * 64 + lane) * 8;
#pragma unroll
        for (int c = 0; c < KH; ++c) w[c] = __builtin_nontemporal_load((const bf16x8*)(wr + c * 512));
      } else {
        const bf16* wr = p.W + (size_t)min(weight_row<EPI>(p, tile_of(i), l15), (EPI == E_SWIGLU ? 2 * p.I : p.N) - 1) * p.ldw + k0 + part * KH * 32;
#pragma unroll
        for (int c = 0; c < KH; ++c) w[c] = __builtin_nontemporal_load((const bf16x8*)(wr + c * 32));
      }
    };
    f32x4 acc[MT];
    auto mma_h = [&](const bf16x8 (&w)[KH], int part) {
      if (part == 0) {
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int c = 0; c < KH; ++c)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = mfma16(w[c], part ? a[t][KH + c] : a[t][c], acc[t]);
    };
    auto meet = [&](int i) {                                   // the tile's partial sums meet in LDS; waves 0..MT-1 finish it
      const int buf = i & 1;
#pragma unroll
      for (int t = 0; t < MT; ++t) *(f32x4*)(red + (((buf * NW + wave) * MT + t) * 64 + lane) * 4) = acc[t];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      finish(i, load_epi(i));
    };
    bf16x8 h0[KH], h1[KH], h2[KH];
    load_h(h0, 0);
    load_h(h1, 1);
    late_acts();
    if (ntl <= 2) {
      // one or two tiles (q|k|v, o): nothing is read twice, the epilogue operands travel with the weights
      if (ntl == 2) load_h(h2, 2);
      const Epi e0 = load_epi(0), e1 = load_epi(1);
      auto meet_pre = [&](int i, const Epi& ep) {
        const int buf = i & 1;
#pragma unroll
        for (int t = 0; t < MT; ++t) *(f32x4*)(red + (((buf * NW + wave) * MT + t) * 64 + lane) * 4) = acc[t];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        finish(i, ep);
      };
      mma_h(h0, 0);
      if (ntl == 2) load_h(h0, 3);
      mma_h(h1, 1);
      meet_pre(0, e0);
      if (ntl == 2) {
        mma_h(h2, 0);
        mma_h(h0, 1);
        meet_pre(1, e1);
      }
      return;
    }
    for (int i = 0; i < ntl; i += 3) {
      const int h = 2 * i;
      load_h(h2, h + 2);
      mma_h(h0, 0);
      load_h(h0, h + 3);
      mma_h(h1, 1);
      meet(i);
      load_h(h1, h + 4);
      mma_h(h2, 0);
      load_h(h2, h + 5);
      mma_h(h0, 1);
      meet(i + 1);
      load_h(h0, h + 6);
      mma_h(h1, 0);
      load_h(h1, h + 7);
      mma_h(h2, 1);
      meet(i + 2);
    }
    return;
  }
  bf16x8 w0[KS], w1[KS], w2[KS];
  TASU_STREAM_STAMP(0);
  load_w(w0, 0);
  if (ntl <= 2) {
    // one or two tiles (the q|k|v, o and down projections): no ring, nothing loaded twice
    if (ntl == 2) load_w(w1, 1);
    late_acts();
    rsq_publish();
    const Epi e0 = load_epi(0), e1 = load_epi(1);
    TASU_STREAM_STAMP(1);
    compute(w0, 0, &e0);
    TASU_STREAM_STAMP(2);
    if (ntl == 2) compute(w1, 1, &e1);
    TASU_STREAM_STAMP(3);
    return;
  }
  // three tiles in flight per wave.  The body is branch-free (the trip count is rounded up to a multiple of three: the spare
  // bodies re-read the last tile and skip their epilogue), so that the compiler's vmcnt bookkeeping sees one straight ring
  // and waits for the oldest tile only.
  load_w(w1, 1);
  late_acts();
  rsq_publish();
  TASU_STREAM_STAMP(1);
  for (int i = 0; i < ntl; i += 3) {
    load_w(w2, i + 2);
    compute(w0, i, nullptr);
    TASU_STREAM_STAMP(2 + i);
    load_w(w0, i + 3);
    compute(w1, i + 1, nullptr);
    TASU_STREAM_STAMP(3 + i);
    load_w(w1, i + 4);
    compute(w2, i + 2, nullptr);
    TASU_STREAM_STAMP(4 + i);
  }
}


// 16 bytes of a row another workgroup of the SAME launch may have written (write-through stores, st_out<true>): SC1 = read past
// this CU's L1 at agent scope (buffer_load_dwordx4 ... sc1 through a descriptor on the row: MI355X_MICROARCH.md, "Valid forms":
// every store of the handed-off bytes sc1 and drained before the ticket, every load of them an sc1 load behind the poll).
template <bool SC1>
__device__ __forceinline__ f32x4 ld_row_f32x4(const float* row_base, int elem) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (SC1) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)row_base, 0, 0x7fffffff, 0x00020000);
    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, elem * 4, 0, 16);
    return __builtin_bit_cast(f32x4, r);
  }
#endif
  return *(const f32x4*)(row_base + elem);
}

// One WAVE: y[row, :] = bf16(w * (x[row, :] * rstd)) in fragment order, D = NG * 256 -- the arithmetic and summation order of
// norm.hip's rmsnorm_fwd_reg_kernel (lane owns columns 4 * lane + 256 * g).
template <int NG, bool WT, bool SC1 = false>
__device__ __forceinline__ void norm_row_frag_ptr(const float* __restrict__ xrow, const float* __restrict__ w, bf16* __restrict__ y,
                                                  int row, float eps, int y_frag = 1) {
  // xrow: the row's D fp32 values (wave-uniform pointer); row (< 64): its position in the fragment-order output
  constexpr int D = NG * 256;
  const int lane = threadIdx.x & 63;
  f32x4 v[NG], gw[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) v[g] = ld_row_f32x4<SC1>(xrow, lane * 4 + g * 256);
#pragma unroll
  for (int g = 0; g < NG; ++g) gw[g] = *(const f32x4*)(w + lane * 4 + g * 256);
  float ss = 0.f;
#pragma unroll
  for (int g = 0; g < NG; ++g) ss += v[g][0] * v[g][0] + v[g][1] * v[g][1] + v[g][2] * v[g][2] + v[g][3] * v[g][3];
  ss = wave_sum(ss);
  const float r = rsqrtf(ss / (float)D + eps);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = gw[g][j] * (v[g][j] * r);
    bf16* dst = y_frag ? y + frag_index(row, lane * 4 + g * 256) : y + (size_t)row * D + lane * 4 + g * 256;
    st_out<WT>((bf16x4*)dst, __builtin_convertvector(o, bf16x4));
  }
}

template <int NG, bool WT>
__device__ __forceinline__ void norm_row_frag(const float* __restrict__ x, const float* __restrict__ w, bf16* __restrict__ y, int row,
                                              float eps) {
  norm_row_frag_ptr<NG, WT>(x + (size_t)row * (NG * 256), w, y, row, eps);
}

// One WAVE: row-wise finish of the E_SLAB partial tiles for a projection that feeds an RMSNorm (the down projection of a decode
// layer), N = NG * 256 columns:  C[m, :] = R[m, :] + bf16(sum of the slabs in slab order);  y[m, :] = bf16(w * (C[m, :] * rstd)),
// lane owns columns 4 * lane + 256 * g like norm_row_frag.  slabs: [ksplit][64 rows][N] (E_SLAB's layout).
template <int NG, bool WT, bool SC1 = false>
__device__ __forceinline__ void finish_norm_row(const float* __restrict__ slabs, int ksplit, float* __restrict__ C,
                                                const float* __restrict__ R, const float* __restrict__ nw, bf16* __restrict__ y,
                                                float eps, int y_frag, int m) {
  constexpr int N = NG * 256;
  const int lane = threadIdx.x & 63;
  // every load of the row is issued before the first sum (slabs in chunks of 4 per column group), residual and norm weight with
  // them: the row costs two memory round trips (loads; stores), not one per slab
  constexpr int KC = 4;
  f32x4 v[NG], w[NG], s[NG];
  size_t e[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int n = lane * 4 + g * 256;
    e[g] = (size_t)m * N + n;
    v[g] = *(const f32x4*)(R + (size_t)m * N + n);
    w[g] = *(const f32x4*)(nw + n);
    s[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int k0 = 0; k0 < ksplit; k0 += KC) {
    f32x4 t[NG][KC];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int j = 0; j < KC; ++j)
        t[g][j] = k0 + j < ksplit ? ld_row_f32x4<SC1>(slabs + (size_t)(k0 + j) * 64 * N + (size_t)m * N, lane * 4 + g * 256)
                                  : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int j = 0; j < KC; ++j) s[g] += t[g][j];
  }
  float ss = 0.f;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int n = lane * 4 + g * 256;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[g][q] = v[g][q] + bf16_round(s[g][q]);
    st_out<WT>((f32x4*)(C + (size_t)m * N + n), v[g]);
    ss += v[g][0] * v[g][0] + v[g][1] * v[g][1] + v[g][2] * v[g][2] + v[g][3] * v[g][3];
  }
  ss = wave_sum(ss);
  const float rs = rsqrtf(ss / (float)N + eps);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int n = lane * 4 + g * 256;
    f32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = w[g][q] * (v[g][q] * rs);
    bf16* dst = y_frag ? y + frag_index(m, n) : y + (size_t)m * N + n;
    st_out<WT>((bf16x4*)dst, __builtin_convertvector(o, bf16x4));
  }
}

// ---- Norm in the producer's launch (round 5).  The two RMSNorm launches of a decode layer are pure latency: a launch boundary,
// one memory round trip for < 1 MB, a store (4.9 + 5.1 us per layer of 55.6: profiles/r04_decode_kernel_stats.csv).  Here the
// workgroups of the projection store their tiles WRITE-THROUGH (st_out<true>), drain them, and ONE lane takes a ticket; the last
// F = min(workgroups, rows) arrivers are the finishers: they wait for the last arrival (one lane polls the counter with sc1
// loads), then finisher f normalises rows f, f + F, ... one per wave -- reading what the other workgroups wrote with sc1 loads
// only -- with exactly the arithmetic of the separate kernels (norm_row_frag_ptr / finish_norm_row): the same bits.  The hand-off
// is MI355X_MICROARCH.md's first measured row (one lane per storing workgroup adds to one counter behind every wave's
// vmcnt(0) + the workgroup barrier; the consumer polls that counter; sc1 stores and loads).  State: two words {arrivals, done},
// zero at allocation; the last finisher to leave resets both, so every launch and every hipGraph replay starts from zero.  No
// deadlock: only the <= 64 LAST arrivers ever wait, and what they wait for are workgroups that are running or can still be
// scheduled (a finisher holds one of >= 512 workgroup slots).
struct NormTail {
  const float* nw;         // norm weight [N]
  bf16* y;                 // normed output (row-major [M, N] or fragment order)
  float eps;
  int y_frag;
  const float* slabs;      // E_SLAB: the partial slabs [ksplit][64][N] (== Args::C), ksplit of them
  int ksplit;
  float* C;                // E_SLAB: fp32 output rows;  E_RESID: the rows the projection wrote (== Args::C)
  const float* R;          // E_SLAB: residual rows
  unsigned* sync;          // {arrivals, finishers done}
};

template <int NG, int EPI>
__device__ __forceinline__ void norm_tail(const Args& p, const NormTail& t, float* red, int n_wg) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave: its write-through stores have left
  __syncthreads();
  int* s_i = (int*)red;
  if (threadIdx.x == 0) s_i[0] = (int)__hip_atomic_fetch_add(t.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int ticket = s_i[0];
  const int F = n_wg < p.M ? n_wg : p.M;
  const int fi = ticket - (n_wg - F);
  if (fi < 0) return;
  if (threadIdx.x == 0) {                                // a finisher: wait for the last arrival
    unsigned spins = 0;
    while (__hip_atomic_load(t.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)n_wg && ++spins < (1u << 26))
      __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  for (int row = fi + wave * F; row < p.M; row += NW * F) {
    if constexpr (EPI == E_RESID) norm_row_frag_ptr<NG, false, true>((const float*)t.C + (size_t)row * (NG * 256), t.nw, t.y, row, t.eps, t.y_frag);
    else finish_norm_row<NG, false, true>(t.slabs, t.ksplit, t.C, t.R, t.nw, t.y, t.eps, t.y_frag, row);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned d = __hip_atomic_fetch_add(t.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (d == (unsigned)F - 1) {                          // every finisher is past its poll: back to zero for the next launch
      __hip_atomic_store(t.sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(t.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#endif
}

}  // namespace tasu_stream
