// Epilogues shared by the bf16 NT GEMM kernels of gemm_pipe.hip (4 MFMA waves + 4 loader waves) and gemm_pp.hip (8 MFMA waves):
// a wave owns MI x NI accumulator fragments of 16 x 16,
//   acc[i][j][r] = C[m][n],  m = row0 + wrow + i*16 + (lane & 15),  n = col0 + wcol + j*16 + (lane >> 4)*4 + r
// (the weight fragment is the MFMA's first operand, so a lane holds 4 CONSECUTIVE output columns of one row).
#pragma once
#include <type_traits>
#include <utility>

#include "common.h"
#include "../../include/tasu_hip.h"

namespace tasu_gemm {

constexpr int OUT_GU_SWIGLU = 3;    // internal epilogue of tasu_gemm_gate_up_swiglu (after the three TASU_GEMM_OUT_* modes)
constexpr int OUT_QKV_ROPE = 5;     // internal epilogue of tasu_gemm_qkv_rope: bias + rotary embedding of the q and k heads
constexpr int OUT_DSWIGLU = 4;      // internal epilogue of tasu_gemm_dswiglu: the SwiGLU backward of the down projection's dgrad

struct Args {
  const bf16* A;
  const bf16* B;
  void* C;
  const float* R;
  const bf16* bias;
  int M, N, K;
  int lda, ldb, ldc;
  int tiles_m, tiles_n;
  bf16* act;            // OUT_GU_SWIGLU: act[M, N] (N = I); C = gate|up [M, 2N]; B = Wgu [2N, K], gate rows first
  // split-K (TASU_GEMM_OUT_F32 only): work item s covers K range [ks * K/ksplit, +K/ksplit) of output tile s % (tiles_m *
  // tiles_n), ks = s / (tiles_m * tiles_n), and writes its fp32 partial tile into slab ks (C + ks * split_stride floats)
  int ksplit;
  long long split_stride;
  // column range of this launch: output columns [n0, N) (OUT_GU_SWIGLU: act columns) -- the dispatcher covers a problem whose
  // 256 x 256 tiles would end in a mostly empty round with two launches, whole rounds of big tiles + the rest on small ones
  int n0 = 0;
  int n1 = 0;           // host side only: the launch's tiles cover columns [n0, n1) (0 = N); n1 is tile-aligned or N
  // stream-K (gemm_pp.hip): the LAST sk_tiles output tiles are cut along K into one contiguous range of K-tile pairs per
  // workgroup; a range that does not begin with its tile's first pair leaves an fp32 partial tile in sk_partial[workgroup]
  // (256 KiB each, accumulator order) and raises sk_flags[workgroup]; the workgroup holding the tile's first pairs adds the
  // partials in K order and stores the tile.  0 = every tile whole (the flags are left at 0 by every launch).
  int sk_tiles = 0;
  float* sk_partial = nullptr;
  int* sk_flags = nullptr;
  double sk_rem = -2.0;  // host side only: sk_plan's max_rem for this launch (-2 = the library default / TASU_GEMM_SK*)
  int act_ld = 0;        // OUT_GU_SWIGLU: leading dimension of act (0 = N); tasu_gemm_gate_up_swiglu_ld writes act into a wider buffer
  int relu = 0;          // TASU_GEMM_OUT_BF16 only: C = bf16(max(acc + bias, 0)) (tasu_gemm_bias_relu_bf16: PositionwiseFeedForward w_1)
};

// kernel launches of the three GEMM kernel families since the library was loaded (tasu_gemm_launch_count: bench.py counts the
// launches behind its GEMM CALLS -- a column-split call is two -- so that roofline.avg_launch_us is per kernel launch, the unit
// rocprofv3's per-kernel average has)
long long& gemm_launches();
// set by tasu_gemm_bias_relu_bf16 around its call of the dispatcher (host; the dispatchers copy it into Args::relu)
int& relu_next();
// likewise for tasu_gemm_gate_up_swiglu_ld: the act leading dimension of the next gate|up launch (0 = I)
int& act_ld_next();

// The work-item list of a workgroup of the 256 x 256 kernel (gemm_pp.hip), as one piece of host / device code so that the
// schedule can be checked on the CPU (tasu_streamk_schedule, tests/test_cabi.py).  Whole tiles (and K-range slabs) are dealt
// round-robin: item s = wg + i * G.  The last sk_tiles tiles are cut along K instead (stream-K): sk_tiles * P K-tile pairs, one
// contiguous range [ub(w), ub(w+1)) per workgroup, visited BEFORE its whole tiles, so that a partial tile is in memory long
// before the workgroup that completes the tile asks for it.  Range ends within 4 pairs of a tile boundary snap to it.
struct PpSchedule {
  enum { FULL = 0, PART = 1, HEAD = 2 };
  struct Item {
    int tile, ks, k0t, nkt, kind;                    // output tile, slab, first K-tile, K-tiles (even), role
  };
  int G, wg, P, nk, base_tiles, dp_tiles, sk_tiles, u0, u1, nsk;
  unsigned sk_units;                                 // (the host keeps sk_units * G below 2^31)
  __host__ __device__ int ub(int w) const {
    unsigned b = (unsigned)w * sk_units / (unsigned)G;
    const unsigned r = b % (unsigned)P;
    if (r && r < 4) b -= r;
    else if (r && P - r < 4) b += P - r;
    return (int)b;
  }
  // K = 128 * P; nk = K-tiles (64 deep) of a whole work item = 2 * P / ksplit
  __host__ __device__ void init(int grid, int block, int pairs, int ksplit, int tiles, int sk) {
    G = grid, wg = block, P = pairs, nk = 2 * pairs / ksplit, base_tiles = tiles, sk_tiles = sk;
    dp_tiles = tiles * ksplit - sk;
    sk_units = (unsigned)sk * (unsigned)pairs;
    u0 = sk ? ub(wg) : 0, u1 = sk ? ub(wg + 1) : 0;
    nsk = u1 > u0 ? (u1 - 1) / P - u0 / P + 1 : 0;
  }
  __host__ __device__ bool item(int idx, Item& it) const {
    if (idx < nsk) {
      const int t = u0 / P + idx;
      const int a = idx == 0 ? u0 : t * P, e = (t + 1) * P, b = u1 < e ? u1 : e;
      it.tile = dp_tiles + t, it.ks = 0, it.k0t = (a - t * P) * 2, it.nkt = (b - a) * 2;
      it.kind = a != t * P ? PART : (b != e ? HEAD : FULL);
      return true;
    }
    const int s = wg + (idx - nsk) * G;
    if (s >= dp_tiles) return false;
    it.ks = s / base_tiles, it.tile = s - it.ks * base_tiles, it.k0t = it.ks * nk, it.nkt = nk, it.kind = FULL;
    return true;
  }
};

// Stream-K plan of the 256 x 256 kernel for T output tiles of P K-tile pairs on G workgroups: how many (trailing) tiles are
// cut along K.  0 = none: whole rounds, no workspace, ranges shorter than 8 pairs, or a last round that is nearly full.
inline int sk_plan(long T, int P, int G, bool have_ws, double max_rem) {
  if (!have_ws || max_rem < 0 || T <= 0 || T % G == 0) return 0;
  // (more than 3/4 of a round: whole tiles on T CUs run faster per CU -- 1.13-1.2 us per K-tile against 1.45 with all 256
  // streaming, the chip-wide ceiling -- than the cut saves: 4096 x 3584 x 18944, 224 tiles: 394 us whole, 435 us cut)
  // ... and only when the ranges' K offsets fall into at most two classes per XCD (16 T / G whole: workgroups w, w + 8, ... --
  // one XCD -- then stream the SAME K position of tiles that share operand panels, through that XCD's L2): 96 tiles 181 us and
  // 48 tiles 110 us at K = 17920, but 72 tiles 189 us and 84 tiles 222 us (more work in less time with 96)
  // (up to a quarter of a round the offsets do not matter: 24 / 36 tiles at K = 17920: 89 / 116 us against 187 on 128 x 192)
  if (T < G) return T * P / G >= 8 && (max_rem >= 1.0 || 4 * T <= (long)G || (4 * T <= 3 * (long)G && (16 * T) % G == 0)) ? (int)T : 0;
  const long rem = T % G;
  if (max_rem <= 0 || (double)rem > max_rem * G || P < 8) return 0;
  return (int)(rem + G);                       // the remainder and one whole round: 1 to 2 tiles per workgroup
}

// tile s of the virtual one-tile-per-block grid -> (tm, tn): XCD-aware (block b and tile s = b + r*gridDim share b % 8,
// i.e. the XCD, because gridDim is a multiple of 8), bijective, then a GROUP_M-row-group raster for L2 reuse of the B panel.
template <int GROUP_M = 4>
__device__ __forceinline__ void tile_coords(const Args& p, int s, int ntiles, int& tm, int& tn) {
  const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = s & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (s >> 3);
  const int per_group = GROUP_M * p.tiles_n;
  const int gid = logical / per_group;
  const int first_m = gid * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int in_g = logical - gid * per_group;
  tm = first_m + in_g % gsz;
  tn = in_g / gsz;
}

// a: lanes 16-31 / 48-63 receive b of lanes 0-15 / 32-47; b: lanes 0-15 / 32-47 receive a of lanes 16-31 / 48-63
__device__ __forceinline__ void swap16(unsigned& a, unsigned& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r[0];
  b = r[1];
#endif
}

// OUT_GU_SWIGLU epilogue: fragments j < NI/2 of a wave are gate columns, j >= NI/2 the up values of the SAME act columns
// acol0 .. acol0 + 8*NI - 1 (the weight rows of a wave column are laid out that way by the loader): writes gate|up and
// act = bf16(bf16(silu(gate)) * up).
template <int MI, int NI, int BM>
__device__ __forceinline__ void store_gu_swiglu(const Args& p, f32x4 (&acc)[MI][NI], int row0, int acol0, int wrow, int lane) {
  int l15 = lane & 15, l4 = (lane >> 4) * 4;
  asm volatile("" : "+v"(l15), "+v"(l4));
  bf16* gu = (bf16*)p.C;
  if constexpr (NI == 4) {
    // paired 16-byte stores (see store_tile): gate fragments (0, 1) and up fragments (2, 3) each form one pair
    if ((p.N & 7) == 0 && (((uintptr_t)gu | (uintptr_t)p.act) & 15) == 0) {
      int cpair = ((lane >> 4) & 1) * 16 + (lane >> 5) * 8;
      asm volatile("" : "+v"(cpair));
      auto rows = [&](auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;     // whole tile inside the matrix: no per-lane tests
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          asm volatile("" ::: "memory");
          const int m = row0 + wrow + i * 16 + l15;
          union { bf16x4 h; unsigned u[2]; } g0, g1, u0, u1, a0, a1;
          g0.h = __builtin_convertvector(acc[i][0], bf16x4), g1.h = __builtin_convertvector(acc[i][1], bf16x4);
          u0.h = __builtin_convertvector(acc[i][2], bf16x4), u1.h = __builtin_convertvector(acc[i][3], bf16x4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            a0.h[r] = (bf16)(bf16_round(silu_f((float)g0.h[r])) * (float)u0.h[r]);
            a1.h[r] = (bf16)(bf16_round(silu_f((float)g1.h[r])) * (float)u1.h[r]);
          }
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            swap16(g0.u[d], g1.u[d]);
            swap16(u0.u[d], u1.u[d]);
            swap16(a0.u[d], a1.u[d]);
          }
          const int n = acol0 + cpair;                  // act column of this lane's 8 values
          if (INTERIOR || (m < p.M && n < p.N)) {                   // N % 8 == 0: all eight or none
            *(u32x4*)(gu + (size_t)m * (2 * (size_t)p.N) + n) = u32x4{g0.u[0], g0.u[1], g1.u[0], g1.u[1]};
            *(u32x4*)(gu + (size_t)m * (2 * (size_t)p.N) + p.N + n) = u32x4{u0.u[0], u0.u[1], u1.u[0], u1.u[1]};
            *(u32x4*)(p.act + (size_t)m * (size_t)(p.act_ld ? p.act_ld : p.N) + n) = u32x4{a0.u[0], a0.u[1], a1.u[0], a1.u[1]};
          }
        }
      };
      if (row0 + BM <= p.M && acol0 + 32 <= p.N) rows(std::true_type{});
      else rows(std::false_type{});
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    asm volatile("" ::: "memory");
    const int m = row0 + wrow + i * 16 + l15;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < NI / 2; ++j) {
      const int n = acol0 + j * 16 + l4;              // act column; N % 4 == 0
      if (n >= p.N) continue;
      const bf16x4 g4 = __builtin_convertvector(acc[i][j], bf16x4), u4 = __builtin_convertvector(acc[i][j + NI / 2], bf16x4);
      bf16x4 a4;
#pragma unroll
      for (int r = 0; r < 4; ++r) a4[r] = (bf16)(bf16_round(silu_f((float)g4[r])) * (float)u4[r]);
      *(bf16x4*)(gu + (size_t)m * (2 * (size_t)p.N) + n) = g4;
      *(bf16x4*)(gu + (size_t)m * (2 * (size_t)p.N) + p.N + n) = u4;
      *(bf16x4*)(p.act + (size_t)m * (size_t)(p.act_ld ? p.act_ld : p.N) + n) = a4;
    }
  }
}

// OUT_QKV_ROPE epilogue (Qwen2Attention.forward: q/k/v projections with bias, then apply_rotary_pos_emb on q and k;
// transformers modeling_qwen2.py:91-135, 150-172).  The tile is one 128-wide head; the loader hands wave column wc the weight
// rows of head dims wc*32 .. +31 (fragments 0, 1) AND 64 + wc*32 .. +31 (fragments 2, 3), so that both members of every
// rotation pair (d, d + 64) sit in the same lane and register index of fragments j and j + 2:
//   q = bf16(acc + bias);  out[d] = bf16(q[d] * cos[m][d] - q[d+64] * sin[m][d]),  out[d+64] = bf16(q[d+64] * cos[m][d] + q[d] * sin[m][d])
// -- bit-identical to tasu_gemm_nt_bf16 (bf16 output, bias) followed by tasu_rope_fwd.  Heads at columns >= p.split_stride
// (the v heads) are stored unrotated.  p.R = cos [M, 64], p.act = sin [M, 64] (fp32), p.bias may be NULL.
template <int MI, int NI, int BM>
__device__ __forceinline__ void store_qkv_rope(const Args& p, f32x4 (&acc)[MI][NI], int row0, int col0, int wrow, int wc, int lane) {
  if constexpr (NI == 4) {
    int l15 = lane & 15, l4 = (lane >> 4) * 4;
    asm volatile("" : "+v"(l15), "+v"(l4));
    const float* ct = p.R;
    const float* st = (const float*)p.act;
    bf16* out = (bf16*)p.C;
    const bool rotate = col0 < (int)p.split_stride;             // wave-uniform
    bf16x4 blo[2], bhi[2];                                       // (kept packed: 8 registers across the row loop, not 16)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int d = col0 + wc * 32 + jj * 16 + l4;
      blo[jj] = p.bias ? *(const bf16x4*)(p.bias + d) : bf16x4{0, 0, 0, 0};
      bhi[jj] = p.bias ? *(const bf16x4*)(p.bias + d + 64) : bf16x4{0, 0, 0, 0};
    }
    f32x4 cq[2], sq[2];                                          // the next (row block, jj) step's factors are in flight
    auto fetch = [&](int step) {
      const int i = step >> 1, jj = step & 1;
      const int m = min(row0 + wrow + i * 16 + l15, p.M - 1);
      cq[step & 1] = *(const f32x4*)(ct + (size_t)m * 64 + wc * 32 + jj * 16 + l4);
      sq[step & 1] = *(const f32x4*)(st + (size_t)m * 64 + wc * 32 + jj * 16 + l4);
    };
    if (rotate) fetch(0);
#pragma unroll
    for (int step = 0; step < 2 * MI; ++step) {
      asm volatile("" ::: "memory");
      if (rotate && step + 1 < 2 * MI) fetch(step + 1);
      const int i = step >> 1, jj = step & 1;
      const int m = row0 + wrow + i * 16 + l15;
      f32x4 lo = acc[i][jj], hi = acc[i][jj + 2];
#pragma unroll
      for (int r = 0; r < 4; ++r) lo[r] += (float)blo[jj][r], hi[r] += (float)bhi[jj][r];
      bf16x4 lob = __builtin_convertvector(lo, bf16x4), hib = __builtin_convertvector(hi, bf16x4);
      if (rotate) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float y1, y2;
          rope_pair_f((float)lob[r], (float)hib[r], cq[step & 1][r], sq[step & 1][r], y1, y2);
          lob[r] = (bf16)y1;
          hib[r] = (bf16)y2;
        }
      }
      if (m < p.M) {
        bf16* c = out + (size_t)m * p.ldc + col0 + wc * 32 + jj * 16 + l4;
        *(bf16x4*)c = lob;
        *(bf16x4*)(c + 64) = hib;
      }
    }
  }
}

// OUT_DSWIGLU epilogue (Qwen2MLP backward, modeling_qwen2.py Qwen2MLP.forward differentiated): the tile is
// dact[M, I] = dy . Wd (the down projection's input gradient); with the saved gate|up [M, 2I] (p.act) it writes
//   dgu[m, n]     = bf16(d * u * sig(g) * (1 + g * (1 - sig(g))))      (gate gradient)
//   dgu[m, I + n] = bf16(d * g * sig(g))                               (up gradient),   d = bf16(dact[m, n])
// into p.C [M, 2I] -- bit-identical to the GEMM with bf16 output followed by tasu_swiglu_bwd, without dact's round trip.
// Fragment pairs trade halves as in store_tile's bf16 path, so a lane owns 8 consecutive columns of a row: 16-byte loads
// of g and u, 16-byte stores of dg and du; the next row block's g / u are in flight while this one is computed.
template <int MI, int NI, int BM, int BN>
__device__ __forceinline__ void store_dswiglu(const Args& p, f32x4 (&acc)[MI][NI], int row0, int col0, int wrow, int wcol, int lane) {
  if constexpr (NI % 2 == 0) {
    constexpr int NP = NI / 2;
    int l15 = lane & 15;
    int cpair = ((lane >> 4) & 1) * 16 + (lane >> 5) * 8;         // first of this lane's 8 columns inside a fragment pair
    asm volatile("" : "+v"(l15), "+v"(cpair));
    const int I = p.N;
    const bf16* gu = p.act;
    bf16* dgu = (bf16*)p.C;
    const bool interior = row0 + BM <= p.M && col0 + BN <= p.N;
    auto rows = [&](auto interior_tag) {
      constexpr bool INTERIOR = decltype(interior_tag)::value;
      u32x4 gq[2][NP], uq[2][NP];
      auto fetch = [&](int i, int buf) {
        // clamped addresses (rows / columns past the matrix read a valid element; their results are not stored)
        const int m = INTERIOR ? row0 + wrow + i * 16 + l15 : max(min(row0 + wrow + i * 16 + l15, p.M - 1), 0);   // (M = 0: stream-K PART items)
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
          int n = col0 + wcol + jp * 32 + cpair;
          if (!INTERIOR) n = min(n, I - 8);
          const bf16* q = gu + (size_t)m * (2 * (size_t)I) + n;
          gq[buf][jp] = *(const u32x4*)q;
          uq[buf][jp] = *(const u32x4*)(q + I);
        }
      };
      fetch(0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        asm volatile("" ::: "memory");
        if (i + 1 < MI) fetch(i + 1, (i + 1) & 1);
        const int m = row0 + wrow + i * 16 + l15;
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
          union { bf16x4 h; unsigned u[2]; } a, b;
          a.h = __builtin_convertvector(acc[i][2 * jp], bf16x4);
          b.h = __builtin_convertvector(acc[i][2 * jp + 1], bf16x4);
          swap16(a.u[0], b.u[0]);
          swap16(a.u[1], b.u[1]);
          union { u32x4 q; bf16 h[8]; } d, g, u, dg, du;
          d.q = u32x4{a.u[0], a.u[1], b.u[0], b.u[1]};
          g.q = gq[i & 1][jp];
          u.q = uq[i & 1][jp];
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            float dgf, duf;
            swiglu_bwd_f((float)g.h[r], (float)u.h[r], (float)d.h[r], dgf, duf);
            dg.h[r] = (bf16)dgf;
            du.h[r] = (bf16)duf;
          }
          const int n = col0 + wcol + jp * 32 + cpair;
          bf16* c = dgu + (size_t)m * (2 * (size_t)I) + n;
          if constexpr (INTERIOR) {
            *(u32x4*)c = dg.q;
            *(u32x4*)(c + I) = du.q;
          } else if (m < p.M && n + 8 <= I) {
            *(u32x4*)c = dg.q;
            *(u32x4*)(c + I) = du.q;
          }
        }
      }
    };
    if (interior) rows(std::true_type{});
    else rows(std::false_type{});
  }
}

template <int MI, int NI, int OUT_MODE, bool HAS_BIAS, int BM, int BN, bool WIDE_RESID>
__device__ __forceinline__ void store_tile(const Args& p, f32x4 (&acc)[MI][NI], int row0, int col0, int wrow, int wcol, int lane) {
  // opaque copies of the lane coordinates: keeps the 32 per-fragment output addresses from being hoisted out of the
  // tile loop into registers that the K loop needs (the kernel sits at the 256-VGPR limit of two waves per SIMD)
  int l15 = lane & 15, l4 = (lane >> 4) * 4;
  asm volatile("" : "+v"(l15), "+v"(l4));
  if constexpr (OUT_MODE == OUT_DSWIGLU) {
    store_dswiglu<MI, NI, BM, BN>(p, acc, row0, col0, wrow, wcol, lane);
    return;
  }
  // the bias depends on the column only: NI x 4 values per lane, loaded once per tile (not once per row block)
  [[maybe_unused]] float bv[NI][4];
  if (HAS_BIAS) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = col0 + wcol + j * 16 + l4;
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[j][r] = n + r < p.N ? (float)p.bias[n + r] : 0.f;
    }
  }
  const bool interior = row0 + BM <= p.M && col0 + BN <= p.N;     // wave-uniform: the whole tile lies inside the matrix
  if constexpr (OUT_MODE == TASU_GEMM_OUT_F32_RESID_BF16R && WIDE_RESID) {
    // residual add: the fp32 residual rows of IB row blocks are fetched together (clamped addresses, no branches)
    // before the first use -- one memory round trip per IB row blocks instead of one per fragment; with one tile per CU
    // (N = 1536: o and down projections) nothing else hides this latency
    if ((p.ldc & 3) == 0 && (p.N & 3) == 0 && (((uintptr_t)p.R | (uintptr_t)p.C) & 15) == 0) {
      // (IB = 4 -- all four row blocks of a 128-row tile in ONE round trip, 216 instead of 200 VGPRs -- was measured in round 6 with
      // -DTASU_RESID_IB=4: o 26.7 -> 26.4 us, down 106.0 -> 105.7 us on 28 rotating weight sets: the residual rows' round trips are
      // not what this epilogue costs; its 2 x 25 MB of fp32 traffic in lock-step on all CUs are)
#ifndef TASU_RESID_IB
#define TASU_RESID_IB 2
#endif
      constexpr int IB = TASU_RESID_IB;           // (the 256-row tiles have no registers to spare: fragment-wise path below)
      static_assert(MI % IB == 0, "row blocks are processed in groups of IB");
#pragma unroll
      for (int i0 = 0; i0 < MI; i0 += IB) {
        asm volatile("" ::: "memory");
        f32x4 old[IB][NI];
#pragma unroll
        for (int ii = 0; ii < IB; ++ii) {
          const int m = min(row0 + wrow + (i0 + ii) * 16 + l15, p.M - 1);
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const int n = min(col0 + wcol + j * 16 + l4, p.N - 4);
            old[ii][j] = *(const f32x4*)(p.R + (size_t)m * p.ldc + n);
          }
        }
#pragma unroll
        for (int ii = 0; ii < IB; ++ii) {
          const int m = row0 + wrow + (i0 + ii) * 16 + l15;
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const int n = col0 + wcol + j * 16 + l4;
            f32x4 v = acc[i0 + ii][j];
            if (HAS_BIAS) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += bv[j][r];
            }
            const f32x4 rr = __builtin_convertvector(__builtin_convertvector(v, bf16x4), f32x4);
            if (interior || (m < p.M && n < p.N)) *(f32x4*)((float*)p.C + (size_t)m * p.ldc + n) = old[ii][j] + rr;
          }
        }
      }
      return;
    }
  }
  if constexpr (OUT_MODE == TASU_GEMM_OUT_BF16 && NI % 2 == 0) {
    // bf16 output: the epilogue is store-ISSUE bound (16 rows x 32 B per dwordx2 instruction).  Fragment pairs (j, j+1)
    // trade halves between lanes l and l+16 (v_permlane16_swap: odd 16-lane rows of the first operand <-> even rows
    // of the second), after which every lane holds 8 consecutive columns: one 16-byte store per pair, 64 B per row.
    if ((p.ldc & 7) == 0 && ((uintptr_t)p.C & 15) == 0) {
      int cpair = ((lane >> 4) & 1) * 16 + (lane >> 5) * 8;       // first of this lane's 8 columns inside the pair
      asm volatile("" : "+v"(cpair));
      auto rows = [&](auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;   // straight-line code: no per-lane edge tests
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          asm volatile("" ::: "memory");
          const int m = row0 + wrow + i * 16 + l15;
#pragma unroll
          for (int j = 0; j < NI; j += 2) {
            f32x4 v0 = acc[i][j], v1 = acc[i][j + 1];
            if (HAS_BIAS) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v0[r] += bv[j][r], v1[r] += bv[j + 1][r];
            }
            if (p.relu) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v0[r] = fmaxf(v0[r], 0.f), v1[r] = fmaxf(v1[r], 0.f);
            }
            union { bf16x4 h; unsigned u[2]; } a, b;
            a.h = __builtin_convertvector(v0, bf16x4);
            b.h = __builtin_convertvector(v1, bf16x4);
            union { u32x4 q; bf16 h[8]; } o;
            swap16(a.u[0], b.u[0]);
            swap16(a.u[1], b.u[1]);
            o.q = u32x4{a.u[0], a.u[1], b.u[0], b.u[1]};
            const int n = col0 + wcol + j * 16 + cpair;
            bf16* c = (bf16*)p.C + (size_t)m * p.ldc + n;
            if constexpr (INTERIOR) {
              *(u32x4*)c = o.q;
            } else if (m < p.M) {
              if (n + 8 <= p.N) {
                *(u32x4*)c = o.q;
              } else {
#pragma unroll
                for (int r = 0; r < 8; ++r)
                  if (n + r < p.N) c[r] = o.h[r];
              }
            }
          }
        }
      };
      if (interior) rows(std::true_type{});
      else rows(std::false_type{});
      return;
    }
  }
  if constexpr (OUT_MODE != TASU_GEMM_OUT_BF16) {
    // fp32 outputs, interior tile, 16-byte aligned rows: straight-line code (the general loop below tests every fragment)
    if (interior && (p.ldc & 3) == 0 && ((uintptr_t)p.C & 15) == 0 &&
        (OUT_MODE != TASU_GEMM_OUT_F32_RESID_BF16R || ((uintptr_t)p.R & 15) == 0)) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        asm volatile("" ::: "memory");
        const size_t rowoff = (size_t)(row0 + wrow + i * 16 + l15) * p.ldc + (col0 + wcol + l4);
        [[maybe_unused]] f32x4 old[NI];
        if constexpr (OUT_MODE == TASU_GEMM_OUT_F32_RESID_BF16R) {
#pragma unroll
          for (int j = 0; j < NI; ++j) old[j] = *(const f32x4*)(p.R + rowoff + j * 16);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          f32x4 v = acc[i][j];
          if (HAS_BIAS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += bv[j][r];
          }
          if constexpr (OUT_MODE == TASU_GEMM_OUT_F32_RESID_BF16R)
            v = old[j] + __builtin_convertvector(__builtin_convertvector(v, bf16x4), f32x4);
          *(f32x4*)((float*)p.C + rowoff + j * 16) = v;
        }
      }
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    asm volatile("" ::: "memory");               // one row block at a time: bounds the loads the scheduler batches
    const int m = row0 + wrow + i * 16 + l15;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = col0 + wcol + j * 16 + l4;
      if (n >= p.N) continue;
      f32x4 v = acc[i][j];
      if (HAS_BIAS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bv[j][r];
      }
      const size_t off = (size_t)m * p.ldc + n;
      const bool full = (n + 4 <= p.N) && ((off & 3) == 0);
      if (OUT_MODE == TASU_GEMM_OUT_BF16) {
        if (p.relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        bf16* c = (bf16*)p.C + off;
        const bf16x4 o = __builtin_convertvector(v, bf16x4);
        if (full) {
          *(bf16x4*)c = o;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) c[r] = o[r];
        }
      } else if (OUT_MODE == TASU_GEMM_OUT_F32) {
        float* c = (float*)p.C + off;
        if (full) {
          *(f32x4*)c = v;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) c[r] = v[r];
        }
      } else {  // TASU_GEMM_OUT_F32_RESID_BF16R: C(fp32) = R(fp32) + bf16_round(result)
        float* c = (float*)p.C + off;
        const float* rs = p.R + off;
        const f32x4 rr = __builtin_convertvector(__builtin_convertvector(v, bf16x4), f32x4);
        if (full) {
          const f32x4 old = *(const f32x4*)rs;
          *(f32x4*)c = old + rr;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) c[r] = rs[r] + rr[r];
        }
      }
    }
  }
}

}  // namespace tasu_gemm
