// Weight-streaming bf16 NT GEMM for M <= 64 rows (the decode step: M = batch x beams): C[M,N] = A[M,K] . B[N,K]^T.
// HBM-bound (every weight byte is read once per step), and a decode step is ~300 such small launches, so the design goals
// are bytes in flight, whole rounds of one-block-per-CU grids, and as few launches as possible:
//   * grid = (N / BN column tiles) x (K splits); BN in {64, 96} and the split count are chosen so that the blocks fill one
//     round of the CUs as fully as possible (a 280-block grid on 256 CUs takes two rounds);
//   * a block is 4 INDEPENDENT waves: wave w owns the K-steps  w, w+4, ...  of the block's K range and a PRIVATE
//     double-buffered LDS region filled by its own global_load_lds_dwordx4 -- no block barrier in the main loop, only
//     the wave's counted s_waitcnt vmcnt (data a wave DMA'd itself needs no barrier);
//   * the 4 waves' 64 x BN fp32 partials are summed through LDS once, and the K splits through fp32 slabs + a small
//     reduce/epilogue kernel (bias, bf16 rounding, residual add): deterministic, no atomics.  (Tried: the last-arriving
//     block of a tile reduces in place, hand-off through sc1 stores/loads and an arrival counter -- 15-30 % SLOWER than
//     the second launch, because one block per tile then sums all slabs serially while the other CUs idle.)
//   * SWIGLU: the block's BN weight rows are BN/2 gate rows and the BN/2 up rows of the same columns, and the epilogue
//     writes act = bf16(bf16(silu(g)) * u) -- the Qwen2 MLP's gate|up projection and activation in one launch.
#include <stdlib.h>

#include <type_traits>

#include <type_traits>
#include <utility>

#include "common.h"
#include "../../include/tasu_hip.h"

namespace tasu_skinny {

constexpr int BM = 64, BK = 64;
constexpr int A_BYTES = BM * BK * 2;                 // 8 KiB = 8 LDS-DMA pieces

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

struct Args {
  const bf16* A;
  const bf16* B;
  float* slab;          // [ksplit][tile][64 * BN] fp32 partials (ksplit > 1)
  void* C;
  const float* R;
  const bf16* bias;
  int M, N, K, lda, ldb, ldc;
  int ksplit, out_mode;
  int up_row0;          // SWIGLU: first "up" row of B (= I); N = I
};

__device__ __forceinline__ void store_out(const Args& p, int m, int n, f32x4 v) {
  if (m >= p.M || n >= p.N) return;
  if (p.bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n + r < p.N) v[r] += (float)p.bias[n + r];
  }
  const size_t off = (size_t)m * p.ldc + n;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (n + r >= p.N) break;
    if (p.out_mode == TASU_GEMM_OUT_BF16)
      ((bf16*)p.C)[off + r] = (bf16)v[r];
    else if (p.out_mode == TASU_GEMM_OUT_F32)
      ((float*)p.C)[off + r] = v[r];
    else
      ((float*)p.C)[off + r] = p.R[off + r] + bf16_round(v[r]);
  }
}

template <int BN, bool SWIGLU>
__global__ __launch_bounds__(256, 1) void gemm_skinny_kernel(Args p) {
  constexpr int NI = BN / 16, PB = BN / 8;            // B fragments per wave row, B pieces per K-step
  constexpr int B_BYTES = BN * BK * 2, WAVE_STAGE = A_BYTES + B_BYTES, WAVE_LDS = 2 * WAVE_STAGE;
  static_assert(64 * BN * 4 <= WAVE_LDS, "the cross-wave reduction reuses a wave's staging region");
  constexpr int HALF = BN / 2;                         // SWIGLU: gate rows [0, HALF), up rows [HALF, BN) of the tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tn = blockIdx.x, ks = blockIdx.y;
  const int col0 = tn * (SWIGLU ? HALF : BN);          // first output column of the tile
  const int nk = p.K / BK;
  const int per = (nk + p.ksplit - 1) / p.ksplit;
  const int kbeg = ks * per, kend = min(nk, kbeg + per);
  char* my = smem + wave * WAVE_LDS;

  // per-lane source pointers of the 8 + PB pieces of one K-step (swizzled like gemm.hip: chunk c of row r at c ^ ((r>>1)&7))
  const bf16* ga[8];
  const bf16* gb[PB];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = i * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    ga[i] = p.A + (size_t)min(r, p.M - 1) * p.lda + c * 8 - (i & 3) * 512;   // - the immediate offset of stage()
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int r = i * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int brow;
    if (SWIGLU) brow = r < HALF ? min(col0 + r, p.N - 1) : p.up_row0 + min(col0 + r - HALF, p.N - 1);
    else brow = min(col0 + r, p.N - 1);
    gb[i] = p.B + (size_t)brow * p.ldb + c * 8 - (i & 3) * 512;
  }
  auto stage = [&](int buf, int kt) {
    char* base = my + buf * WAVE_STAGE;
    const int koff = kt * BK;
    // one M0 (LDS base) per four pieces: the immediate offset (i & 3) KiB moves the LDS address and the global address
    // alike; the per-lane pointers were lowered by the same amount (writing M0 per piece costs ~40 issue cycles each)
    auto one_a = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      __builtin_amdgcn_global_load_lds((glb_void*)(ga[i] + koff), (lds_void*)(base + (i & ~3) * 1024), 16, (i & 3) * 1024, 0);
    };
    auto one_b = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      __builtin_amdgcn_global_load_lds((glb_void*)(gb[i] + koff), (lds_void*)(base + A_BYTES + (i & ~3) * 1024), 16,
                                       (i & 3) * 1024, 0);
    };
    [&]<int... I>(std::integer_sequence<int, I...>) { (one_a(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, 8>{});
    [&]<int... I>(std::integer_sequence<int, I...>) { (one_b(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, PB>{});
  };
  const int sw = (lane >> 1) & 7;
  int roff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) roff[kk] = (lane & 15) * 128 + (((kk * 4 + (lane >> 4)) ^ sw) << 4);

  f32x4 acc[4][NI];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int kt = kbeg + wave;
  int buf = 0;
  if (kt < kend) stage(0, kt);
  for (; kt < kend; kt += 4) {
    const bool more = kt + 4 < kend;
    if (more) {
      stage(buf ^ 1, kt + 4);
      // current tile landed, next (8 + PB pieces) stays in flight
      if (PB == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const char* sa = my + buf * WAVE_STAGE;
    const char* sb = sa + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[4], fb[NI];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *(const bf16x8*)(sa + i * 16 * 128 + roff[kk]);
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[j] = *(const bf16x8*)(sb + j * 16 * 128 + roff[kk]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);
    }
    // the LDS reads above must have returned before this buffer is DMA'd again by the next iteration's stage()
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    buf ^= 1;
  }
  // ---- sum the 4 waves' partials through LDS (each wave's private region is free again)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float* red = (float*)(smem + wave * WAVE_LDS);              // [4 x NI fragments][64 lanes] f32x4, lane-linear
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) *(f32x4*)(red + ((i * NI + j) * 64 + lane) * 4) = acc[i][j];
  __syncthreads();
  // wave w finalises the (i = w) row block: acc[w][j] summed over the 4 regions.
  // sum[j][r] = C[m = wave*16 + (lane&15)][tile column j*16 + (lane>>4)*4 + r]
  f32x4 sum[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    sum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2)
      sum[j] += *(const f32x4*)((const float*)(smem + w2 * WAVE_LDS) + ((wave * NI + j) * 64 + lane) * 4);
  }
  if (p.ksplit > 1) {
    // partial tile -> slab [ks][tile] in fragment-image order; skinny_reduce_kernel finishes
    float* mine = p.slab + ((size_t)ks * gridDim.x + tn) * (64 * BN);
#pragma unroll
    for (int j = 0; j < NI; ++j) *(f32x4*)(mine + ((wave * NI + j) * 64 + lane) * 4) = sum[j];
    return;
  }
  const int m = wave * 16 + (lane & 15);
  if (SWIGLU) {
    // fragments j < NI/2 hold gate columns, j + NI/2 the up values of the same columns
#pragma unroll
    for (int j = 0; j < NI / 2; ++j) {
      const int n = col0 + j * 16 + (lane >> 4) * 4;
      if (m < p.M && n < p.N) {
        bf16* dst = (bf16*)p.C + (size_t)m * p.ldc + n;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r >= p.N) break;
          const float g = bf16_round(sum[j][r]), u = bf16_round(sum[j + NI / 2][r]);
          dst[r] = (bf16)(bf16_round(silu_f(g)) * u);
        }
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < NI; ++j) store_out(p, m, col0 + j * 16 + (lane >> 4) * 4, sum[j]);
  }
}

// out = epilogue(sum over splits of the slabs).  One thread per f32x4 of the fragment image: grid = tiles * 64 * BN / 4 / 256.
template <int BN, bool SWIGLU>
__global__ __launch_bounds__(256) void skinny_reduce_kernel(Args p, int tiles) {
  constexpr int NI = BN / 16, TILE_F = 64 * BN;
  const int idx = blockIdx.x * 256 + threadIdx.x;       // f32x4 index over [tile][wave][j][lane]
  if (idx >= tiles * (TILE_F / 4)) return;
  const int tn = idx / (TILE_F / 4), e = idx - tn * (TILE_F / 4);
  const int lane = e & 63, j = (e >> 6) % NI, wave = (e >> 6) / NI;
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int k = 0; k < p.ksplit; ++k) s += *(const f32x4*)(p.slab + ((size_t)k * tiles + tn) * TILE_F + e * 4);
  const int m = wave * 16 + (lane & 15);
  if (SWIGLU) {
    // the thread that owns gate fragment j also reads the matching up fragment j + NI/2
    if (j >= NI / 2) return;
    f32x4 u = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int k = 0; k < p.ksplit; ++k)
      u += *(const f32x4*)(p.slab + ((size_t)k * tiles + tn) * TILE_F + (e + (NI / 2) * 64) * 4);
    const int n = tn * (BN / 2) + j * 16 + (lane >> 4) * 4;
    if (m < p.M && n < p.N) {
      bf16* dst = (bf16*)p.C + (size_t)m * p.ldc + n;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (n + r >= p.N) break;
        dst[r] = (bf16)(bf16_round(silu_f(bf16_round(s[r]))) * bf16_round(u[r]));
      }
    }
  } else {
    store_out(p, m, tn * BN + j * 16 + (lane >> 4) * 4, s);
  }
}

// Row-wise finish of a split GEMM whose result feeds an RMSNorm (o / down projection of the decode step): block = row m;
// C[m, :] = R[m, :] + bf16(sum of the slabs), then y[m, :] = bf16(w * (C[m, :] * rstd)) -- tasu_rmsnorm_fwd's arithmetic --
// in the same launch (one launch less per norm: ~5 us of the ~9-us floor these small kernels run at).  N % 4 == 0.
template <int BN>
__global__ __launch_bounds__(256) void skinny_reduce_norm_kernel(Args p, int tiles, const float* __restrict__ nw,
                                                                 bf16* __restrict__ y, float eps, int y_frag) {
  constexpr int NI = BN / 16, TILE_F = 64 * BN;
  __shared__ float red[4];
  const int m = blockIdx.x;
  const int wave_r = m >> 4, l15 = m & 15;
  const int ngroups = p.N / 4;
  float* crow = (float*)p.C + (size_t)m * p.ldc;
  const float* rrow = p.R + (size_t)m * p.ldc;
  float ss = 0.f;
  for (int g = threadIdx.x; g < ngroups; g += 256) {
    const int n = g * 4, tn = n / BN, nin = n - tn * BN;
    const int e = ((wave_r * NI + (nin >> 4)) * 64 + ((nin & 15) >> 2) * 16 + l15) * 4;
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int k = 0; k < p.ksplit; ++k) s += *(const f32x4*)(p.slab + ((size_t)k * tiles + tn) * TILE_F + e);
    const f32x4 r = *(const f32x4*)(rrow + n);
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = r[q] + bf16_round(s[q]);
    *(f32x4*)(crow + n) = v;
    ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  ss = block_sum<4>(ss, red);
  const float rs = rsqrtf(ss / (float)p.N + eps);
  for (int g = threadIdx.x; g < ngroups; g += 256) {
    const int n = g * 4;
    const f32x4 v = *(const f32x4*)(crow + n);     // this thread's own store above
    const f32x4 w = *(const f32x4*)(nw + n);
    f32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = w[q] * (v[q] * rs);
    // y_frag: the MFMA fragment order of csrc/gemm_stream.hip's activations ([n / 32][m / 16][16 * ((n % 32) / 8) + m % 16][n % 8])
    bf16* dst = y_frag ? y + ((((size_t)(n >> 5) * 4 + (m >> 4)) * 64 + ((n & 31) >> 3) * 16 + (m & 15)) << 3) + (n & 7)
                       : y + (size_t)m * p.N + n;
    *(bf16x4*)dst = __builtin_convertvector(o, bf16x4);
  }
}

// The same finish with the row in registers (N <= 4096): every slab load of a thread is issued before the first sum (the loop
// above waits for its loads four at a time: three dependent round trips at ksplit = 10), the norm weight and the residual
// travel with them, and the second pass uses the registers instead of re-reading the thread's own store -- two memory round
// trips instead of six (9.3 -> ~5.5 us per decode layer).  Same sums in the same order.
template <int BN>
__global__ __launch_bounds__(256) void skinny_reduce_norm_reg_kernel(Args p, int tiles, const float* __restrict__ nw,
                                                                     bf16* __restrict__ y, float eps, int y_frag) {
  constexpr int NI = BN / 16, TILE_F = 64 * BN, MAXG = 4, KC = 8;
  __shared__ float red[4];
  const int m = blockIdx.x;
  const int wave_r = m >> 4, l15 = m & 15;
  const int ngroups = p.N / 4;
  float* crow = (float*)p.C + (size_t)m * p.ldc;
  const float* rrow = p.R + (size_t)m * p.ldc;
  f32x4 v[MAXG], w[MAXG], s[MAXG];
  size_t e[MAXG];
  bool on[MAXG];
#pragma unroll
  for (int i = 0; i < MAXG; ++i) {
    const int g = threadIdx.x + i * 256;
    on[i] = g < ngroups;
    const int n = (on[i] ? g : 0) * 4, tn = n / BN, nin = n - tn * BN;
    e[i] = (size_t)tn * TILE_F + ((wave_r * NI + (nin >> 4)) * 64 + ((nin & 15) >> 2) * 16 + l15) * 4;
    s[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    v[i] = on[i] ? *(const f32x4*)(rrow + n) : s[i];
    w[i] = on[i] ? *(const f32x4*)(nw + n) : s[i];
  }
  for (int k0 = 0; k0 < p.ksplit; k0 += KC) {
    f32x4 t[MAXG][KC];
#pragma unroll
    for (int i = 0; i < MAXG; ++i)
#pragma unroll
      for (int j = 0; j < KC; ++j)
        t[i][j] = on[i] && k0 + j < p.ksplit ? *(const f32x4*)(p.slab + (size_t)(k0 + j) * tiles * TILE_F + e[i])
                                            : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MAXG; ++i)
#pragma unroll
      for (int j = 0; j < KC; ++j) s[i] += t[i][j];
  }
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXG; ++i) {
    if (on[i]) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[i][q] = v[i][q] + bf16_round(s[i][q]);
      *(f32x4*)(crow + (threadIdx.x + i * 256) * 4) = v[i];
      ss += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
    }
  }
  ss = block_sum<4>(ss, red);
  const float rs = rsqrtf(ss / (float)p.N + eps);
#pragma unroll
  for (int i = 0; i < MAXG; ++i) {
    if (on[i]) {
      const int n = (threadIdx.x + i * 256) * 4;
      f32x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = w[i][q] * (v[i][q] * rs);
      bf16* dst = y_frag ? y + ((((size_t)(n >> 5) * 4 + (m >> 4)) * 64 + ((n & 31) >> 3) * 16 + (m & 15)) << 3) + (n & 7)
                         : y + (size_t)m * p.N + n;
      *(bf16x4*)dst = __builtin_convertvector(o, bf16x4);
    }
  }
}

// Row-wise finish of the split qkv projection of a decode step: block = row m; q|k|v[m, :] = bf16(sum of the slabs + bias),
// RoPE on the H query and G key heads (tables [M, 64]), rotated row written to qkv[m, :] and its k and v appended to
// cache[m, pos[m]] -- skinny_reduce + tasu_rope_append in one launch, same arithmetic and rounding points.
template <int BN>
__global__ __launch_bounds__(256) void skinny_reduce_rope_kernel(Args p, int tiles, const float* __restrict__ ct,
                                                                 const float* __restrict__ st, bf16* __restrict__ kc,
                                                                 bf16* __restrict__ vc, const int32_t* __restrict__ pos, int H,
                                                                 int G, int ctx) {
  constexpr int NI = BN / 16, TILE_F = 64 * BN, HD = 128;
  const int m = blockIdx.x;
  const int wave_r = m >> 4, l15 = m & 15;
  const int W = G * HD;
  // bf16-rounded (sum over the slabs + bias) of four 4-column groups of row m at once: 4 independent loads in flight per split
  auto cols4x4 = [&](const int (&n)[4], bf16x4 (&o)[4]) {
    size_t off[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int tn = n[g] / BN, nin = n[g] - tn * BN;
      off[g] = (size_t)tn * TILE_F + ((wave_r * NI + (nin >> 4)) * 64 + ((nin & 15) >> 2) * 16 + l15) * 4;
    }
    f32x4 s[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) s[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int k = 0; k < p.ksplit; ++k) {
      const float* base = p.slab + (size_t)k * tiles * TILE_F;
#pragma unroll
      for (int g = 0; g < 4; ++g) s[g] += *(const f32x4*)(base + off[g]);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int q = 0; q < 4; ++q) o[g][q] = (bf16)(s[g][q] + (p.bias ? (float)p.bias[n[g] + q] : 0.f));
  };
  bf16* out = (bf16*)p.C + (size_t)m * p.ldc;
  const size_t slot = ((size_t)m * ctx + pos[m]) * W;
  const int nrot = (H + G) * 8;                               // rotation units: (head, 8-wide chunk of the low half)
  for (int u = threadIdx.x; u < nrot + G * 8; u += 256) {
    bf16x4 v[4];
    if (u < nrot) {
      const int hh = u >> 3, c = (u & 7) * 8;
      const int n[4] = {hh * HD + c, hh * HD + c + 4, hh * HD + 64 + c, hh * HD + 64 + c + 4};
      cols4x4(n, v);
      const f32x4 c0 = *(const f32x4*)(ct + (size_t)m * 64 + c), c1 = *(const f32x4*)(ct + (size_t)m * 64 + c + 4);
      const f32x4 s0 = *(const f32x4*)(st + (size_t)m * 64 + c), s1 = *(const f32x4*)(st + (size_t)m * 64 + c + 4);
      bf16x8 lo, hi;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float cs = j < 4 ? c0[j] : c1[j - 4], sn = j < 4 ? s0[j] : s1[j - 4];
        const float x1 = (float)(j < 4 ? v[0][j] : v[1][j - 4]), x2 = (float)(j < 4 ? v[2][j] : v[3][j - 4]);
        lo[j] = (bf16)(x1 * cs - x2 * sn);
        hi[j] = (bf16)(x2 * cs + x1 * sn);
      }
      *(bf16x8*)(out + hh * HD + c) = lo;
      *(bf16x8*)(out + hh * HD + 64 + c) = hi;
      if (hh >= H) {
        bf16* kd = kc + slot + (hh - H) * HD;
        *(bf16x8*)(kd + c) = lo;
        *(bf16x8*)(kd + 64 + c) = hi;
      }
    } else {                                                   // V: 16-element units
      const int c = (u - nrot) * 16;
      const int nb = (H + G) * HD + c;
      const int n[4] = {nb, nb + 4, nb + 8, nb + 12};
      cols4x4(n, v);
      bf16x8 v0, v1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v0[j] = v[0][j];
        v0[4 + j] = v[1][j];
        v1[j] = v[2][j];
        v1[4 + j] = v[3][j];
      }
      *(bf16x8*)(out + nb) = v0;
      *(bf16x8*)(out + nb + 8) = v1;
      *(bf16x8*)(vc + slot + c) = v0;
      *(bf16x8*)(vc + slot + c + 8) = v1;
    }
  }
}

int cu_count() {
  static const int n = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    }
    return cus;
  }();
  return n;
}

struct NormArgs {
  const float* w = nullptr;   // RMSNorm weight [N]; null = no fused norm
  bf16* y = nullptr;          // normalized bf16 output [M, N]
  float eps = 0.f;
  int y_frag = 0;             // y in fragment order (needs N % 32 == 0)
  // qkv finish (RoPE + cache append) instead of a norm: cos != nullptr
  const float* cos = nullptr;
  const float* sin = nullptr;
  bf16* kc = nullptr;
  bf16* vc = nullptr;
  const int32_t* pos = nullptr;
  int H = 0, G = 0, ctx = 0;
};

template <int BN, bool SWIGLU>
int launch(Args a, int tiles, hipStream_t st, NormArgs na = NormArgs()) {
  constexpr int LDS = 4 * 2 * (A_BYTES + BN * BK * 2);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<BN, SWIGLU>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  TASU_LAUNCH((gemm_skinny_kernel<BN, SWIGLU>), dim3(tiles, a.ksplit), dim3(256), LDS, st, a);
  if (a.ksplit > 1) {
    if (na.w && a.N <= 4096)
      TASU_LAUNCH((skinny_reduce_norm_reg_kernel<BN>), dim3(a.M), dim3(256), 0, st, a, tiles, na.w, na.y, na.eps, na.y_frag);
    else if (na.w)
      TASU_LAUNCH((skinny_reduce_norm_kernel<BN>), dim3(a.M), dim3(256), 0, st, a, tiles, na.w, na.y, na.eps, na.y_frag);
    else if (na.cos)
      TASU_LAUNCH((skinny_reduce_rope_kernel<BN>), dim3(a.M), dim3(256), 0, st, a, tiles, na.cos, na.sin, na.kc, na.vc, na.pos,
                  na.H, na.G, na.ctx);
    else
      TASU_LAUNCH((skinny_reduce_kernel<BN, SWIGLU>), dim3((tiles * 16 * BN + 255) / 256), dim3(256), 0, st, a, tiles);
  } else if (na.w) {
    return na.y_frag ? tasu_rmsnorm_fwd_frag((const float*)a.C, na.w, na.y, a.M, a.N, na.eps, st)
                     : tasu_rmsnorm_fwd((const float*)a.C, na.w, na.y, nullptr, a.M, a.N, na.eps, st);   // unsplit: C is final
  } else if (na.cos) {
    return tasu_rope_append(a.C, na.cos, na.sin, na.kc, na.vc, na.pos, a.M, na.H, na.G, na.ctx, st);
  }
  return TASU_OK;
}

// Tile width and K split (one block per CU: 128-160 KiB of LDS).  Measured on MI355X, M = 64, cold weights, graph replay
// (tools/bench_skinny.py): a launch costs ~8.5 us however small, so splits only pay when the column tiles alone leave
// most CUs idle --
//   enough tiles (>= 60 % of the CUs): no split; the width that needs fewer rounds x rows
//       (gate|up 2 x 8960 x 1536: 96 -> 187 blocks, 18.4 us = 3.0 TB/s; 64 -> 280 blocks = two rounds, 23.2 us);
//   few tiles: 64-wide, K split to about one block per CU, at most 4 ways when K is short
//       (down 1536 x 8960: 24 tiles x 10 = 15.0 us vs 45.3 us unsplit; qkv / o, K = 1536: 8.5-9 us at any split).
template <bool SWIGLU>
int plan_and_launch(Args a, float* ws, size_t ws_floats, hipStream_t st, NormArgs na = NormArgs()) {
  const int cus = cu_count();
  const int nk = a.K / BK;
  static const int force_bn = [] { const char* e = tasu_lab_env("TASU_SKINNY_BN"); return e ? atoi(e) : 0; }();
  static const int force_ks = [] { const char* e = tasu_lab_env("TASU_SKINNY_KS"); return e ? atoi(e) : 0; }();
  auto tiles_of = [&](int bn) { const int cols = SWIGLU ? bn / 2 : bn; return (a.N + cols - 1) / cols; };
  int best_bn = 64, best_ks = 1;
  const int t64 = tiles_of(64), t96 = tiles_of(96);
  if (t96 * 10 >= cus * 6) {
    const long r64 = (t64 + cus - 1) / cus, r96 = (t96 + cus - 1) / cus;
    best_bn = r96 * 96 < r64 * 64 ? 96 : 64;
  } else if (ws) {
    int ks = cus / t64;
    const int max_ks = nk <= 32 ? 4 : 32;
    if (ks > max_ks) ks = max_ks;
    if (ks > nk / 4) ks = nk / 4;                             // keep >= 1 K-step per wave
    while (ks > 1 && (size_t)t64 * ks * 64 * 64 > ws_floats) --ks;
    best_ks = ks > 1 ? ks : 1;
  }
  if (force_bn == 64 || force_bn == 96) best_bn = force_bn;
  if (force_ks > 0 && ws) {
    best_ks = force_ks <= nk / 4 ? force_ks : (nk / 4 > 0 ? nk / 4 : 1);
    while (best_ks > 1 && (size_t)tiles_of(best_bn) * best_ks * 64 * best_bn > ws_floats) --best_ks;
  }
  const int cols = SWIGLU ? best_bn / 2 : best_bn;
  const int tiles = (a.N + cols - 1) / cols;
  a.ksplit = best_ks;
  a.slab = ws;
  return best_bn == 96 ? launch<96, SWIGLU>(a, tiles, st, na) : launch<64, SWIGLU>(a, tiles, st, na);
}

}  // namespace tasu_skinny

extern "C" int tasu_gemm_skinny_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* bias,
                                     const float* resid, int M, int N, int K, int out_mode, float* workspace,
                                     int64_t workspace_floats, void* stream) {
  using namespace tasu_skinny;
  if (!A || !B || !C || M <= 0 || M > 64 || N <= 0 || K <= 0 || K % 64 || lda % 8 || ldb % 8) return TASU_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)workspace & 15)) return TASU_ERR_ARG;
  if (out_mode < 0 || out_mode > 2 || (out_mode == TASU_GEMM_OUT_F32_RESID_BF16R && !resid)) return TASU_ERR_ARG;
  Args a;
  a.A = (const bf16*)A;
  a.B = (const bf16*)B;
  a.C = C;
  a.R = resid;
  a.bias = (const bf16*)bias;
  a.M = M;
  a.N = N;
  a.K = K;
  a.lda = lda;
  a.ldb = ldb;
  a.ldc = ldc;
  a.out_mode = out_mode;
  a.up_row0 = 0;
  return plan_and_launch<false>(a, workspace, workspace ? (size_t)workspace_floats : 0, (hipStream_t)stream);
}

extern "C" int tasu_gemm_skinny_swiglu(const void* A, int lda, const void* Wgu, int ldw, void* act, int ldact, int M, int I,
                                       int K, float* workspace, int64_t workspace_floats, void* stream) {
  using namespace tasu_skinny;
  if (!A || !Wgu || !act || M <= 0 || M > 64 || I <= 0 || K <= 0 || K % 64 || lda % 8 || ldw % 8) return TASU_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)Wgu & 15) || ((uintptr_t)workspace & 15)) return TASU_ERR_ARG;
  Args a;
  a.A = (const bf16*)A;
  a.B = (const bf16*)Wgu;
  a.C = act;
  a.R = nullptr;
  a.bias = nullptr;
  a.M = M;
  a.N = I;
  a.K = K;
  a.lda = lda;
  a.ldb = ldw;
  a.ldc = ldact;
  a.out_mode = TASU_GEMM_OUT_BF16;
  a.up_row0 = I;
  return plan_and_launch<true>(a, workspace, workspace ? (size_t)workspace_floats : 0, (hipStream_t)stream);
}

extern "C" int tasu_gemm_skinny_norm(const void* A, int lda, const void* B, int ldb, float* C, const float* resid, int M, int N,
                                     int K, const float* norm_w, void* y, float eps, int y_frag, float* workspace,
                                     int64_t workspace_floats, void* stream) {
  using namespace tasu_skinny;
  if (!A || !B || !C || !resid || !norm_w || !y || M <= 0 || M > 64 || N <= 0 || N % 4 || K <= 0 || K % 64 || lda % 8 || ldb % 8 ||
      (y_frag && N % 32))
    return TASU_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)workspace & 15)) return TASU_ERR_ARG;
  Args a;
  a.A = (const bf16*)A;
  a.B = (const bf16*)B;
  a.C = C;
  a.R = resid;
  a.bias = nullptr;
  a.M = M;
  a.N = N;
  a.K = K;
  a.lda = lda;
  a.ldb = ldb;
  a.ldc = N;
  a.out_mode = TASU_GEMM_OUT_F32_RESID_BF16R;
  a.up_row0 = 0;
  NormArgs na;
  na.w = norm_w;
  na.y = (bf16*)y;
  na.eps = eps;
  na.y_frag = y_frag;
  return plan_and_launch<false>(a, workspace, workspace ? (size_t)workspace_floats : 0, (hipStream_t)stream, na);
}

extern "C" int tasu_gemm_skinny_qkv_rope(const void* A, int lda, const void* Wqkv, int ldw, const void* bias, void* qkv, int M,
                                         int H, int G, int K, const float* cos_tab, const float* sin_tab, void* kcache,
                                         void* vcache, const int32_t* pos, int ctx, float* workspace, int64_t workspace_floats,
                                         void* stream) {
  using namespace tasu_skinny;
  if (!A || !Wqkv || !qkv || !cos_tab || !sin_tab || !kcache || !vcache || !pos || M <= 0 || M > 64 || H <= 0 || G <= 0 || K <= 0 ||
      K % 64 || lda % 8 || ldw % 8 || ctx <= 0)
    return TASU_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)Wqkv & 15) || ((uintptr_t)workspace & 15) || ((uintptr_t)qkv & 15)) return TASU_ERR_ARG;
  Args a;
  a.A = (const bf16*)A;
  a.B = (const bf16*)Wqkv;
  a.C = qkv;
  a.R = nullptr;
  a.bias = (const bf16*)bias;
  a.M = M;
  a.N = (H + 2 * G) * 128;
  a.K = K;
  a.lda = lda;
  a.ldb = ldw;
  a.ldc = a.N;
  a.out_mode = TASU_GEMM_OUT_BF16;
  a.up_row0 = 0;
  NormArgs na;
  na.cos = cos_tab;
  na.sin = sin_tab;
  na.kc = (bf16*)kcache;
  na.vc = (bf16*)vcache;
  na.pos = pos;
  na.H = H;
  na.G = G;
  na.ctx = ctx;
  return plan_and_launch<false>(a, workspace, workspace ? (size_t)workspace_floats : 0, (hipStream_t)stream, na);
}
