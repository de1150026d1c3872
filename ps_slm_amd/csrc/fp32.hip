// fp32 arithmetic mode of the decode path (train_config.use_fp16 = false at inference: the reference loads the LLM and the
// projector in fp32 and calls HF generate without autocast -- Multitask/inference_batch.py:113-117,146, Multitask/model/ps-slm.py:660-675).
// Every tensor here is fp32: weights, residual stream, q|k|v, KV cache, logits.  The step is HBM-bound on the fp32 weights
// (6.2 GB per generated position at Qwen2.5-1.5B), so the kernels are simple: one tiled MFMA GEMM
// (v_mfma_f32_16x16x4_f32, fp32 products and accumulation) with deterministic K-range slabs for the narrow projections, and
// wave-per-row VALU kernels for the rest.  Summation orders differ from the reference's CPU BLAS; nothing is rounded to bf16.
#include <algorithm>
#include <utility>
#include "common.h"
#include "../../include/tasu_hip.h"

namespace tasu_f32 {

constexpr int HD = 128;
constexpr int BM = 64, BN = 64, BK = 32, LDS_LD = BK + 4;     // row pitch 36 floats: 16-byte aligned rows, conflict-free b128 fragment reads

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float silu_exact(float x) { return x / (1.f + expf(-x)); }
// x * cos + rotate_half(x) * sin with the two products and the sum rounded separately, like torch eager (apply_rotary_pos_emb):
// -ffp-contract=fast would otherwise fuse one product into the sum, and not the same one at every call site
__device__ __forceinline__ void rope_pair_eager(float x1, float x2, float c, float s, float& y1, float& y2) {
  // (every product passes through an empty asm before it is used: a pragma `fp contract(off)` here did not survive inlining into
  // f32_sum_slabs_rope_kernel, which came out with a v_pk_fma_f32 where f32_rope_kernel has mul + add)
  float a1 = x1 * c, b1 = x2 * s, a2 = x2 * c, b2 = x1 * s;
  asm volatile("" : "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2));
  y1 = a1 - b1;
  y2 = a2 + b2;
}
__device__ __forceinline__ float act_f(float v, int act) { return act == 1 ? silu_exact(v) : (act == 2 ? fmaxf(v, 0.f) : v); }

// C[m, n] (+)= sum_k A[m, k] W[n, k] over the K range of blockIdx.z.  ksplit == 1: C = [resid +] act(acc + bias).
// ksplit > 1: the raw partial goes to slab blockIdx.z ([M, N] each, row stride N); f32_sum_slabs_kernel finishes.
// 256 threads = 2 x 2 waves of 32 x 32; the weight fragment is the MFMA's first operand, so a lane ends up with 4 consecutive
// output columns of one row: acc[i][j][r] = C[m0 + wm*32 + i*16 + (lane & 15)][n0 + wn*32 + j*16 + (lane >> 4)*4 + r].
__global__ __launch_bounds__(256) void f32_gemm_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                       float* __restrict__ C, int ldc, const float* __restrict__ bias,
                                                       const float* __restrict__ resid, int M, int N, int K, int kchunk, int act,
                                                       int ksplit) {
  __shared__ __attribute__((aligned(16))) float sA[2][BM][LDS_LD];
  __shared__ __attribute__((aligned(16))) float sW[2][BN][LDS_LD];
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM, z = blockIdx.z;
  const int k_lo = z * kchunk, k_hi = min(K, k_lo + kchunk);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
  const int lr = t >> 3, lc = (t & 7) * 4;                       // this thread's rows lr, lr + 32 and first column of the 64 x 32 tiles
  const float* ap0 = A + (size_t)min(m0 + lr, M - 1) * lda + lc;
  const float* ap1 = A + (size_t)min(m0 + lr + 32, M - 1) * lda + lc;
  const float* wp0 = W + (size_t)min(n0 + lr, N - 1) * ldw + lc;
  const float* wp1 = W + (size_t)min(n0 + lr + 32, N - 1) * ldw + lc;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 ra0 = *(const f32x4*)(ap0 + k_lo), ra1 = *(const f32x4*)(ap1 + k_lo);
  f32x4 rw0 = *(const f32x4*)(wp0 + k_lo), rw1 = *(const f32x4*)(wp1 + k_lo);
  int buf = 0;
  for (int k0 = k_lo; k0 < k_hi; k0 += BK) {
    *(f32x4*)&sA[buf][lr][lc] = ra0, *(f32x4*)&sA[buf][lr + 32][lc] = ra1;
    *(f32x4*)&sW[buf][lr][lc] = rw0, *(f32x4*)&sW[buf][lr + 32][lc] = rw1;
    __syncthreads();
    if (k0 + BK < k_hi) {                              // the next K-step's 16 KiB are in flight under this one's MFMAs
      ra0 = *(const f32x4*)(ap0 + k0 + BK), ra1 = *(const f32x4*)(ap1 + k0 + BK);
      rw0 = *(const f32x4*)(wp0 + k0 + BK), rw1 = *(const f32x4*)(wp1 + k0 + BK);
    }
    // fragments as 16-byte reads: lane group g = lane >> 4 takes k = h * 16 + 4 g .. + 3 of its row; MFMA (h, e) then contracts
    // k = h * 16 + 4 g + e of both operands -- every k of the step exactly once (4x fewer LDS instructions than one float per MFMA)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 fa[2], fw[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = *(const f32x4*)&sA[buf][wm * 32 + i * 16 + (lane & 15)][h * 16 + 4 * (lane >> 4)];
#pragma unroll
      for (int j = 0; j < 2; ++j) fw[j] = *(const f32x4*)&sW[buf][wn * 32 + j * 16 + (lane & 15)][h * 16 + 4 * (lane >> 4)];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma4(fw[j][e], fa[i][e], acc[i][j]);
    }
    buf ^= 1;                                         // (the other buffer was last read before the barrier above)
  }
  float* out = ksplit > 1 ? C + (size_t)z * M * N : C;
  const int ldo = ksplit > 1 ? N : ldc;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 32 + i * 16 + (lane & 15);
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 32 + j * 16 + (lane >> 4) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (n + r >= N) continue;
        float v = acc[i][j][r];
        if (ksplit == 1) {
          if (bias) v += bias[n + r];
          v = act_f(v, act);
          if (resid) v = resid[(size_t)m * ldc + n + r] + v;
        }
        out[(size_t)m * ldo + n + r] = v;
      }
    }
  }
}

// The decode step's GEMM: at most 64 rows (beam rows) against a weight matrix that is read exactly once.  The tile kernel above
// stages both operands through LDS and keeps one K-step in flight per workgroup; at 64 rows that leaves the matrix pipes at a
// third of their rate (1.8 TB/s of weights, where 64 rows of fp32 MFMA can take 4.9).  This kernel is the fp32 sibling of
// stream_body.h (the bf16 decode GEMMs):
//   * one 8-wave workgroup per CU walks 16-column weight tiles (tile bx, bx + nbx, ...) for ONE K range of 128 KS; wave w owns the
//     slice [16 KS w, 16 KS (w + 1)) of that range and keeps its slice of all 64 activation rows in registers (16 KS of them), so a
//     tile is KS 16-byte loads per lane -- lane (n = lane & 15, g = lane >> 4) reads W[16 tile + n][k + 4 g .. + 3], straight into
//     the MFMA's operand registers, three tiles in flight per wave on a branch-free ring -- and 16 KS MFMAs with no LDS read;
//   * the eight waves' partial tiles meet in LDS (two buffers, one raw s_barrier per tile: the loads stay in flight across it) and
//     waves 0 .. RB - 1 add them in wave order = ascending k: deterministic, like the K-range slabs the finishers add;
//   * K ranges (grid y) write slabs [range][M][N] for the same finishers as the tile kernel's; the balance unit is one 16-column
//     tile (gate|up: 1120 tiles x 2 ranges over 256 workgroups = 8.75 +- 0.25 tiles each).
// RB = row blocks of 16 (ceil(M / 16)): 32 beam rows pay half the MFMAs of 64.
constexpr int SW_NW = 8;
// tiles in flight per wave (register slots of 4 KS each).  Three: deeper rings measured SLOWER (KS = 6: 4 slots 392 us against 366 on
// the lm_head, KS = 5: 6 slots 34 us against 29 on the down projection) -- the step is not waiting on a latency a longer queue
// would cover: at 64 rows the matrix pipes (2.6 us per tile) and the memory system (2.5 us at 4.8 TB/s) are both near their rates.
template <int KS, int RB>
constexpr int SW_RING = 3;
template <int KS, int RB, int DBG = 0>     // DBG (tools/bench_f32_stream.py, TASU_F32_STREAM_DBG): 1 = no weight loads, 2 = no MFMAs
__global__ __launch_bounds__(64 * SW_NW) void f32_stream_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                                float* __restrict__ C, int ldc, const float* __restrict__ bias,
                                                                const float* __restrict__ resid, int M, int N, int act, int ksplit,
                                                                int tiles, int nbx, int kfrag) {
  // ldw < 0: W is in fragment order, [tile][kfrag = K / 16][64 lanes][4]
  extern __shared__ __attribute__((aligned(16))) float red[];                // [2 buffers][SW_NW waves][RB][64 lanes] f32x4
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nl = lane & 15, g = lane >> 4;
  const int bx = blockIdx.x, by = blockIdx.y;
  const int k0 = (by * SW_NW + wave) * (KS * 16) + 4 * g;                    // this lane's first k
  // Activations: this wave's K slice of every row as MFMA operands, a[rb][c] = x[16 rb + nl][slice + 16 c + 4 g .. + 3].  Read that
  // way from the row-major matrix a wave instruction touches 16 rows x 64 B and the 24 loads of a 64-row slice took ~10 us of every
  // workgroup (o projection, one tile: 16.9 us with the MFMAs compiled out).  So the slice is read in row order -- 16 KS floats of a
  // row are 4 KS consecutive lanes -- into a wave-private LDS image (32 rows at a time, pitch 16 KS + 4) and comes back as fragments.
  f32x4 a[RB][KS];
  auto stage_acts = [&]() {
    constexpr int P4 = 4 * KS, PITCH = 16 * KS + 4, HR = RB >= 2 ? 32 : 16, NH = (16 * RB) / HR, PER = HR * P4 / 64;   // 16-byte pieces per row; rows per half
    static_assert(HR * P4 % 64 == 0, "whole wave instructions");
    float* img = red + (size_t)wave * HR * PITCH;
    const float* as = A + (by * SW_NW + wave) * (KS * 16);
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      f32x4 v[PER];
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const int f = j * 64 + lane, row = f / P4, c4 = f - row * P4;
        v[j] = *(const f32x4*)(as + (size_t)min(h * HR + row, M - 1) * lda + c4 * 4);   // (rows beyond M: clamped here, masked at the store)
      }
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const int f = j * 64 + lane, row = f / P4, c4 = f - row * P4;
        *(f32x4*)(img + row * PITCH + c4 * 4) = v[j];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // wave-private image: its own writes have landed
#pragma unroll
      for (int rl = 0; rl < HR / 16; ++rl)
#pragma unroll
        for (int c = 0; c < KS; ++c) a[h * (HR / 16) + rl][c] = *(const f32x4*)(img + (rl * 16 + nl) * PITCH + c * 16 + 4 * g);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // ... and are read before the next half overwrites them
    }
    __syncthreads();                                                        // the images overlay the partial-tile buffers
  };
  const int ntl = (tiles - bx + nbx - 1) / nbx;                              // tiles this workgroup walks (>= 1: nbx <= tiles)
  auto tile_of = [&](int i) { return bx + min(i, ntl - 1) * nbx; };          // clamped: loads past the end re-read the last tile
  auto load_w = [&](f32x4 (&w)[KS], int i) {
    if constexpr (DBG == 1) {
#pragma unroll
      for (int c = 0; c < KS; ++c) w[c] = f32x4{1.f + i, 2.f, 3.f, 4.f};
      return;
    }
    // (inline asm: the compiler's own vmcnt bookkeeping drains the ring once per trip -- at the loop head it waits for the tile
    // behind the one it needs as well, 4-5 us per three tiles at the loaded-memory latency; the waits are counted by hand in arrive())
    if constexpr (DBG == 4) {                              // timing probe: real operand bits, no traffic (tiles 0 .. NR - 2 are loaded once)
      if (i >= SW_RING<KS, RB> - 1) {
#pragma unroll
        for (int c = 0; c < KS; ++c) asm volatile("s_nop 0" : "+v"(w[c]));
        return;
      }
    }
    // (inline asm: the compiler's own vmcnt bookkeeping drains the ring once per trip -- at the loop head it waits for the tile
    // behind the one it needs as well, 4-5 us per three tiles at the loaded-memory latency; the waits are counted by hand in arrive())
    if constexpr (DBG == 4) {                              // timing probe: real operand bits, no traffic (tiles 0 .. NR - 2 are loaded once)
      if (i >= SW_RING<KS, RB> - 1) {
#pragma unroll
        for (int c = 0; c < KS; ++c) asm volatile("s_nop 0" : "+v"(w[c]));
        return;
      }
    }
    if constexpr (DBG == 3) {                              // timing probe: the same bytes read as if W were in fragment order (wrong values)
      const float* wr = W + (((size_t)tile_of(i) * (ldw / 16) + (by * SW_NW + wave) * KS) * 64 + lane) * 4;
#pragma unroll
      for (int c = 0; c < KS; ++c) asm volatile("global_load_dwordx4 %0, %1, off offset:%2 nt" : "=v"(w[c]) : "v"(wr + (c >> 1) * 512), "n"((c & 1) * 1024) : "memory");
      return;
    }
    if (ldw < 0) {                                         // fragment order (tasu_f32_to_fragment_order): a wave instruction reads 1 KiB
      const float* wr = W + (((size_t)tile_of(i) * kfrag + (by * SW_NW + wave) * KS) * 64 + lane) * 4;
#pragma unroll
      for (int c = 0; c < KS; ++c)
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2 nt" : "=v"(w[c]) : "v"(wr + (c >> 1) * 512), "n"((c & 1) * 1024) : "memory");
      return;
    }
    const float* wr = W + (size_t)min(tile_of(i) * 16 + nl, N - 1) * ldw + k0;
#pragma unroll
    for (int c = 0; c < KS; ++c) asm volatile("global_load_dwordx4 %0, %1, off offset:%2 nt" : "=v"(w[c]) : "v"(wr), "n"(c * 64) : "memory");
  };
  // piece c of the tile that has `behind` tiles of loads issued after it: loads return in order, so it is there once at most
  // behind * KS + (KS - 1 - c) operations are outstanding (a finishing wave's store in between only makes the wait longer)
  auto arrive = [&](f32x4& wc, auto behind, auto c) {
    if constexpr (DBG == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(wc)::"memory");
    else if constexpr (DBG != 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(wc) : "n"(decltype(behind)::value * KS + KS - 1 - decltype(c)::value) : "memory");
  };
  auto finish = [&](int i) {
    if (wave >= RB) return;
    const float* src = red + ((size_t)((i & 1) * SW_NW) * RB + wave) * 256 + lane * 4;
    f32x4 s = *(const f32x4*)src;
#pragma unroll
    for (int w = 1; w < SW_NW; ++w) {
      const f32x4 v = *(const f32x4*)(src + (size_t)w * RB * 256);
#pragma unroll
      for (int r = 0; r < 4; ++r) s[r] += v[r];
    }
    const int m = wave * 16 + nl, n = tile_of(i) * 16 + 4 * g;
    if (m >= M || n >= N) return;
    if (ksplit > 1) {
      float* dst = C + ((size_t)by * M + m) * N + n;
      if (n + 4 <= N && !(N & 3)) {
        *(f32x4*)dst = s;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < N) dst[r] = s[r];
      }
      return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (n + r >= N) continue;
      float v = s[r];
      if (bias) v += bias[n + r];
      v = act_f(v, act);
      if (resid) v = resid[(size_t)m * ldc + n + r] + v;
      C[(size_t)m * ldc + n + r] = v;
    }
  };
  auto compute = [&](f32x4 (&w)[KS], int i, auto behind) {
    if (i >= ntl) return;                                  // (workgroup-uniform: a spare slot of the last trip; its loads were issued)
    f32x4 acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    [&]<int... C>(std::integer_sequence<int, C...>) {
      (
          [&] {
            arrive(w[C], behind, std::integral_constant<int, C>{});
            if constexpr (DBG == 2) {
#pragma unroll
              for (int rb = 0; rb < RB; ++rb) acc[rb] += w[C] * a[rb][C];
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma4(w[C][e], a[rb][C][e], acc[rb]);
            }
          }(),
          ...);
    }(std::make_integer_sequence<int, KS>{});
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) *(f32x4*)(red + ((size_t)(((i & 1) * SW_NW + wave) * RB + rb) * 64 + lane) * 4) = acc[rb];
    // my partial tile is in LDS; everybody's is after the barrier.  Raw s_barrier: the weight loads of the next tiles stay in
    // flight across it.  Buffer (i & 1) is written again at tile i + 2, which every wave reaches only after the barrier of tile
    // i + 1, i.e. after the finishing waves' reads of tile i.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    finish(i);
  };
  // NR tiles in flight per wave: slot s holds tile i + s while tile i is multiplied, and is refilled with tile i + NR right after.
  // The trip count is rounded up to a multiple of NR: spare slots re-read the last tile (so that the hand-counted waits stay true)
  // and skip their MFMAs.
  constexpr int NR = SW_RING<KS, RB>;
  f32x4 w[NR][KS];
  constexpr std::integral_constant<int, NR - 1> behind{};
#pragma unroll
  for (int sl = 0; sl < NR - 1; ++sl) load_w(w[sl], sl);
  stage_acts();                                            // (behind the first weight requests; its waits drain them, once)
  for (int i = 0; i < ntl; i += NR) {
#pragma unroll
    for (int sl = 0; sl < NR; ++sl) {
      load_w(w[(sl + NR - 1) % NR], i + sl + NR - 1);
      compute(w[sl], i + sl, behind);
    }
  }
  if constexpr (DBG != 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the spare tiles' loads land before the registers die)
}

// C = [resid +] act(bias + slab 0 + slab 1 + ...): the K ranges in ascending order, the same order on every run
__global__ __launch_bounds__(256) void f32_sum_slabs_kernel(const float* __restrict__ slabs, int ksplit, float* __restrict__ C, int ldc,
                                                            const float* __restrict__ bias, const float* __restrict__ resid, int M,
                                                            int N, int act) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)M * N) return;
  const int m = (int)(idx / N), n = (int)(idx - (size_t)m * N);
  float v = slabs[idx];
  for (int s = 1; s < ksplit; ++s) v += slabs[(size_t)s * M * N + idx];
  if (bias) v += bias[n];
  v = act_f(v, act);
  if (resid) v = resid[(size_t)m * ldc + n] + v;
  C[(size_t)m * ldc + n] = v;
}

// The three finishers of the decode step's K-range-slab GEMMs that carry the NEXT row-wise kernel with them (one launch instead
// of two; same sums in the same order as f32_sum_slabs_kernel followed by that kernel, so the same bits):
//   f32_sum_slabs_norm_kernel    x = resid + (slab 0 + slab 1 + ...) [+ bias];  y = w * (x * rsqrt(mean(x^2) + eps))   (block per row)
//   f32_sum_slabs_swiglu_kernel  act = silu(sum of the gate slabs) * (sum of the up slabs)                            (gu never stored)
//   f32_sum_slabs_rope_kernel    qkv = sum + bias, q / k heads rotated, k / v appended to the cache                    (thread per pair)
__global__ __launch_bounds__(1024) void f32_sum_slabs_norm_kernel(const float* __restrict__ slabs, int ksplit, float* __restrict__ x, int ldx,
                                                                  const float* __restrict__ bias, const float* __restrict__ resid,
                                                                  const float* __restrict__ w, float* __restrict__ y, int M, int N, float eps) {
  // 1024 threads per row: a thread owns one or two columns and requests all of a column's slabs together (with 256 threads and
  // the slabs one after the other this kernel took longer than the two it replaces: 64 rows are only 64 workgroups)
  __shared__ float red[16];
  const int m = blockIdx.x;
  float ss = 0.f;
  for (int n = threadIdx.x; n < N; n += 1024) {
    const size_t idx = (size_t)m * N + n;
    float part[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) part[s] = s < ksplit ? slabs[(size_t)s * M * N + idx] : 0.f;
    float v = part[0];
#pragma unroll
    for (int s = 1; s < 16; ++s)
      if (s < ksplit) v += part[s];                  // (ascending, like f32_sum_slabs_kernel: the same bits)
    if (bias) v += bias[n];
    if (resid) v = resid[(size_t)m * ldx + n] + v;
    x[(size_t)m * ldx + n] = v;
    ss += v * v;
  }
  ss = block_sum<16>(ss, red);
  const float rs = rsqrtf(ss / (float)N + eps);
  for (int n = threadIdx.x; n < N; n += 1024) y[(size_t)m * N + n] = w[n] * (x[(size_t)m * ldx + n] * rs);
}

__global__ __launch_bounds__(256) void f32_sum_slabs_swiglu_kernel(const float* __restrict__ slabs, int ksplit, float* __restrict__ act, int M,
                                                                   int I) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)M * I) return;
  const size_t m = idx / I, c = idx - m * I;
  const size_t gi = m * 2 * I + c, stride = (size_t)M * 2 * I;
  float g = slabs[gi], u = slabs[gi + I];
  for (int s = 1; s < ksplit; ++s) g += slabs[s * stride + gi], u += slabs[s * stride + gi + I];
  act[idx] = __fmul_rn(silu_exact(g), u);
}

__global__ __launch_bounds__(256) void f32_sum_slabs_rope_kernel(const float* __restrict__ slabs, int ksplit, float* __restrict__ qkv,
                                                                 const float* __restrict__ bias, const float* __restrict__ ct,
                                                                 const float* __restrict__ st, int M, int H, int G, float* __restrict__ kc,
                                                                 float* __restrict__ vc, const int32_t* __restrict__ slot, int ctx) {
  const int LD = (H + 2 * G) * HD, Wd = G * HD;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int per_row = (H + 2 * G) * 64;
  if (idx >= (size_t)M * per_row) return;
  const int m = (int)(idx / per_row), rest = (int)(idx - (size_t)m * per_row), hh = rest >> 6, d = rest & 63;
  const size_t i1 = (size_t)m * LD + hh * HD + d, i2 = i1 + 64, stride = (size_t)M * LD;
  float y1 = slabs[i1], y2 = slabs[i2];
  for (int s = 1; s < ksplit; ++s) y1 += slabs[s * stride + i1], y2 += slabs[s * stride + i2];
  if (bias) y1 += bias[hh * HD + d], y2 += bias[hh * HD + d + 64];
  if (hh < H + G) {
    const float c = ct[(size_t)m * 64 + d], sn = st[(size_t)m * 64 + d];
    rope_pair_eager(y1, y2, c, sn, y1, y2);
  }
  qkv[i1] = y1;
  qkv[i2] = y2;
  if (kc && hh >= H) {
    float* dst = (hh < H + G ? kc : vc) + ((size_t)m * ctx + slot[m]) * Wd + (hh - (hh < H + G ? H : H + G)) * HD;
    dst[d] = y1;
    dst[d + 64] = y2;
  }
}

// Qwen2RMSNorm in fp32 (modeling_qwen2.py:41-48): y = w * (x * rsqrt(mean(x^2) + eps)); one 1024-thread block per row
__global__ __launch_bounds__(1024) void f32_rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                           int D, float eps) {
  __shared__ float red[16];
  const float* xr = x + (size_t)blockIdx.x * D;
  float s = 0.f;
  for (int c = threadIdx.x; c < D; c += 1024) s += xr[c] * xr[c];          // (the order f32_sum_slabs_norm_kernel sums in)
  s = block_sum<16>(s, red);
  const float rs = rsqrtf(s / (float)D + eps);
  for (int c = threadIdx.x; c < D; c += 1024) y[(size_t)blockIdx.x * D + c] = w[c] * (xr[c] * rs);
}

// apply_rotary_pos_emb (modeling_qwen2.py:91-135) on the q and k heads of qkv [M, (H + 2G) * 128], in place:
// out = x * cos + rotate_half(x) * sin with the products rounded separately (torch eager); optionally the rotated k and v rows
// go to cache[row, slot[row]] (decode step: Cache.update).  One thread per rotation pair.
__global__ __launch_bounds__(256) void f32_rope_kernel(float* __restrict__ qkv, const float* __restrict__ ct, const float* __restrict__ st,
                                                       int M, int H, int G, float* __restrict__ kc, float* __restrict__ vc,
                                                       const int32_t* __restrict__ slot, int ctx, int inverse) {
  const int LD = (H + 2 * G) * HD, Wd = G * HD;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int per_row = (H + 2 * G) * 64;
  if (idx >= (size_t)M * per_row) return;
  const int m = (int)(idx / per_row), rest = (int)(idx - (size_t)m * per_row), hh = rest >> 6, d = rest & 63;
  float* row = qkv + (size_t)m * LD + hh * HD;
  float y1 = row[d], y2 = row[d + 64];
  if (hh < H + G) {
    const float c = ct[(size_t)m * 64 + d], s = inverse ? -st[(size_t)m * 64 + d] : st[(size_t)m * 64 + d];   // inverse: the backward
    rope_pair_eager(y1, y2, c, s, y1, y2);
    row[d] = y1;
    row[d + 64] = y2;
  }
  if (kc && hh >= H) {
    float* dst = (hh < H + G ? kc : vc) + ((size_t)m * ctx + slot[m]) * Wd + (hh - (hh < H + G ? H : H + G)) * HD;
    dst[d] = y1;
    dst[d + 64] = y2;
  }
}

// prompt K / V of a prefill qkv activation [B * S, LD] -> cache row b * n_beams, positions 0 .. S - 1
__global__ __launch_bounds__(256) void f32_kv_fill_kernel(const float* __restrict__ qkv, float* __restrict__ kc, float* __restrict__ vc,
                                                          int B, int S, int H, int G, int nb, int ctx) {
  const int LD = (H + 2 * G) * HD, Wd = G * HD;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)B * S * Wd) return;
  const int c = (int)(idx % Wd);
  const size_t bs = idx / Wd;
  const int b = (int)(bs / S), s = (int)(bs - (size_t)b * S);
  const size_t dst = ((size_t)(b * nb) * ctx + s) * Wd + c;
  kc[dst] = qkv[bs * LD + H * HD + c];
  vc[dst] = qkv[bs * LD + (H + G) * HD + c];
}

// softmax(q . K^T * scale + mask) . V for ONE query row and ONE KV head per workgroup, fp32 (eager attention of
// modeling_qwen2.py:150-172): the REP = H / G query heads of the group share every K / V row that is loaded, and the waves
// split the keys (a single wave's chain of dependent loads is what bounds this kernel: 56 -> 111 us per layer were measured for one
// wave per group / per head).  Keys come through functors (prefill: rows of the qkv activation; decode: cache rows through the beam
// index).  Phase 1: the 128-dim dot products of all REP heads, q from LDS; phase 2: wave = its share of the keys, lane = two
// output dims, partial outputs added in wave order (deterministic).
// LDS: sq REP * 128 | sp REP * MAX_KEYS | part NW * REP * 128 | red NW * REP floats.
// NW waves per workgroup: 4 for the prompt pass (one workgroup per query position: thousands of them), 16 for a generated position
// (one per beam row and KV head: 128 workgroups on 256 CUs -- with four waves each the step waited on 512 waves' dependent loads).
// Phase 1: EIGHT LANES PER KEY, 16 dims each (a key row is 512 contiguous bytes over 8 adjacent lanes; one thread per key read its
// row 16 bytes at a time, every lane of a load instruction in a different row), the eight partial dots added by lane shuffles.
constexpr int F32_ATTN_MAX_KEYS = 2048;
constexpr int F32_ATTN_MAX_REP = 8;
constexpr int F32_PREFILL_NW = 4, F32_DECODE_NW = 8, F32_DECODE_VPRE = 32;
// kst: the score rows' stride = the launch's longest key range rounded up to 64 (LDS sized for the sequence at hand: with rows of
// MAX_KEYS the prompt pass held two workgroups per CU)
template <int REP, int NW>
__host__ __device__ constexpr int f32_attn_lds_floats(int kst) { return REP * HD + REP * kst + NW * REP * HD + NW * REP; }
template <int REP, int NW, int VPRE, typename KeyAt, typename ValAt>
// The REP "heads" are rows q + h * qstride (query heads of a KV group: stride 128; or, bidirectional attention, REP consecutive QUERY
// positions of one head: stride = the activation's leading dimension); rows h >= nvalid are computed on a copy of the last valid one
// and stored as zeros; rows h >= nstore are not stored at all (a tile that sticks out of the sequence).
__device__ void f32_attn_group(const float* q, int k_lo, int k_hi, float scale, float* out, float* smem, int kst, KeyAt key_at, ValAt val_at,
                               int qstride = HD, int ostride = HD, int nvalid = REP, int nstore = REP) {
  constexpr int NT = 64 * NW;
  float* sq = smem;
  float* sp = sq + REP * HD;
  float* part = sp + REP * kst;
  float* red = part + NW * REP * HD;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nk = k_hi - k_lo;
  for (int i = t; i < REP * HD; i += NT) sq[i] = q[(size_t)min(i / HD, nvalid - 1) * qstride + (i & (HD - 1))];
  __syncthreads();
  float mx[REP];
#pragma unroll
  for (int h = 0; h < REP; ++h) mx[h] = -__builtin_inff();
  // this wave's share of the keys for phase 2; its first VPRE value rows are requested NOW, together with the key rows (they do
  // not depend on the scores: one round trip fewer on a generated position's critical path, which is made of round trips)
  const int qn = (nk + NW - 1) / NW, j_lo = min(nk, wave * qn), j_hi = min(nk, j_lo + qn);
  float pva[VPRE > 0 ? VPRE : 1], pvb[VPRE > 0 ? VPRE : 1];      // VPRE = 0 (the prompt pass: throughput, not round trips): none
#pragma unroll
  for (int u = 0; u < VPRE; ++u) {
    const float* vr = val_at(k_lo + min(j_lo + u, nk - 1));
    pva[u] = vr[lane], pvb[u] = vr[lane + 64];
  }
  {
    const int seg = t & 7;                               // this lane's 16 dims of every key it visits
    auto dots = [&](const f32x4 (&kv)[4], int j) {
      float a[REP];
#pragma unroll
      for (int h = 0; h < REP; ++h) {
        a[h] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 qv = *(const f32x4*)(sq + h * HD + seg * 16 + c * 4);
          a[h] += kv[c][0] * qv[0] + kv[c][1] * qv[1] + kv[c][2] * qv[2] + kv[c][3] * qv[3];
        }
        a[h] += __shfl_xor(a[h], 1, 64);
        a[h] += __shfl_xor(a[h], 2, 64);
        a[h] += __shfl_xor(a[h], 4, 64);
      }
      if (j < nk) {
#pragma unroll
        for (int h = 0; h < REP; ++h) {
          const float sc = a[h] * scale;
          if (seg == 0) sp[h * kst + j] = sc;
          mx[h] = fmaxf(mx[h], sc);
        }
      }
    };
    for (int j0 = 0; j0 < nk; j0 += NT / 4) {            // two keys per lane group and trip, both rows in flight together
      const int ja = j0 + (t >> 3), jb = ja + NT / 8;
      const f32x4* ka = (const f32x4*)key_at(k_lo + min(ja, nk - 1)) + seg * 4;
      const f32x4* kb = (const f32x4*)key_at(k_lo + min(jb, nk - 1)) + seg * 4;
      f32x4 kva[4], kvb[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) kva[c] = ka[c];
#pragma unroll
      for (int c = 0; c < 4; ++c) kvb[c] = kb[c];
      dots(kva, ja);
      if (j0 + NT / 8 < nk) dots(kvb, jb);               // (workgroup-uniform)
    }
  }
#pragma unroll
  for (int h = 0; h < REP; ++h) {
    mx[h] = wave_max(mx[h]);
    if (lane == 0) red[wave * REP + h] = mx[h];
  }
  __syncthreads();
  float inv[REP];
#pragma unroll
  for (int h = 0; h < REP; ++h) {
    float m = red[h];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = fmaxf(m, red[w * REP + h]);
    mx[h] = m;
  }
  __syncthreads();                                     // (red is rewritten below)
#pragma unroll
  for (int h = 0; h < REP; ++h) {
    float sum = 0.f;
    for (int j = t; j < nk; j += NT) {
      const float e = expf(sp[h * kst + j] - mx[h]);
      sp[h * kst + j] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wave * REP + h] = sum;
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < REP; ++h) {
    float sum = red[h];
#pragma unroll
    for (int w = 1; w < NW; ++w) sum += red[w * REP + h];          // (wave order: deterministic)
    inv[h] = 1.f / sum;
  }
  float o0[REP], o1[REP];
#pragma unroll
  for (int h = 0; h < REP; ++h) o0[h] = o1[h] = 0.f;
  // the prefetched value rows first (key order, like everything after them), then eight rows per trip
  int j = j_lo;
#pragma unroll
  for (int u = 0; u < VPRE; ++u) {
    if (j_lo + u < j_hi) {
#pragma unroll
      for (int h = 0; h < REP; ++h) {
        const float p = sp[h * kst + j_lo + u] * inv[h];
        o0[h] += p * pva[u];
        o1[h] += p * pvb[u];
      }
    }
  }
  j = min(j_hi, j_lo + VPRE);
  constexpr int UNR = 8;
  for (; j + UNR <= j_hi; j += UNR) {
    float va[UNR], vb[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const float* vr = val_at(k_lo + j + u);
      va[u] = vr[lane], vb[u] = vr[lane + 64];
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int h = 0; h < REP; ++h) {
        const float p = sp[h * kst + j + u] * inv[h];
        o0[h] += p * va[u];
        o1[h] += p * vb[u];
      }
  }
  for (; j < j_hi; ++j) {
    const float* vr = val_at(k_lo + j);
    const float va = vr[lane], vb = vr[lane + 64];
#pragma unroll
    for (int h = 0; h < REP; ++h) {
      const float p = sp[h * kst + j] * inv[h];
      o0[h] += p * va;
      o1[h] += p * vb;
    }
  }
#pragma unroll
  for (int h = 0; h < REP; ++h) {
    part[(wave * REP + h) * HD + lane] = o0[h];
    part[(wave * REP + h) * HD + lane + 64] = o1[h];
  }
  __syncthreads();
  for (int i = t; i < REP * HD; i += NT) {
    float v = part[i];
#pragma unroll
    for (int w = 1; w < NW; ++w) v += part[w * REP * HD + i];      // (wave order = ascending keys)
    if (i / HD < nstore) out[(size_t)(i / HD) * ostride + (i & (HD - 1))] = i / HD < nvalid ? v : 0.f;
  }
}

// prefill: one workgroup per (batch row, position, KV head); query s of batch row b sees keys [kstart[b], s] (causal, left padding masked)
template <int REP>
__global__ __launch_bounds__(256) void f32_attn_prefill_kernel(const float* __restrict__ qkv, const int32_t* __restrict__ kstart,
                                                               const int32_t* __restrict__ klen, float* __restrict__ out, int B, int S,
                                                               int H, int G, float scale, int kst) {
  extern __shared__ float smem[];
  const long long id = blockIdx.x;
  const int g = (int)(id % G);
  const long long bs = id / G;
  const int b = (int)(bs / S), s = (int)(bs - (long long)b * S);
  const int LD = (H + 2 * G) * HD;
  float* o = out + (size_t)bs * (H * HD) + g * REP * HD;
  // klen == NULL: causal decoder prompt, keys [kstart[b], s]; else bidirectional with key padding (SANM encoder,
  // SenseVoice.py:209-228): every query sees keys [0, klen[b])
  const int k_lo = klen ? 0 : kstart[b];
  const int k_hi = klen ? klen[b] : s + 1;
  if (s < k_lo || k_hi <= k_lo || (klen && s >= k_hi)) {   // a padding position: its output is never read
    for (int i = threadIdx.x; i < REP * HD; i += 256) o[i] = 0.f;
    return;
  }
  const float* base = qkv + (size_t)b * S * LD;
  f32_attn_group<REP, F32_PREFILL_NW, 0>(base + (size_t)s * LD + g * REP * HD, k_lo, k_hi, scale, o, smem, kst,
                      [&](int j) { return base + (size_t)j * LD + (H + g) * HD; },
                      [&](int j) { return base + (size_t)j * LD + (H + G + g) * HD; });
}

// bidirectional attention with key padding (the SANM encoder: SenseVoice.py:209-228, H = G): one workgroup per (batch row, tile of QT
// consecutive query positions, head) -- the QT queries see the same keys [0, klen[b]) and share every K / V row that is loaded, the
// way the query heads of a KV group do in the causal kernel (one workgroup per query position re-read the head's 504 K / V rows 504
// times per utterance: 16 GB of L2 traffic per layer at 16 x 504 frames, 672 us).  Positions >= klen[b] get zeros.
constexpr int F32_BIDIR_QT = 8;
__global__ __launch_bounds__(64 * F32_PREFILL_NW) void f32_attn_bidir_kernel(const float* __restrict__ qkv, const int32_t* __restrict__ klen,
                                                                             float* __restrict__ out, int B, int S, int H, float scale, int kst) {
  extern __shared__ float smem[];
  const int tiles = (S + F32_BIDIR_QT - 1) / F32_BIDIR_QT;
  const int head = blockIdx.x % H;
  const int tile = (blockIdx.x / H) % tiles, b = blockIdx.x / (H * tiles);
  const int LD = 3 * H * HD, s0 = tile * F32_BIDIR_QT;
  const int k_hi = min(klen[b], S), rows = min(F32_BIDIR_QT, S - s0), nvalid = max(0, min(rows, k_hi - s0));
  float* o = out + ((size_t)b * S + s0) * (H * HD) + head * HD;
  if (nvalid == 0) {                                   // a tile of padding positions
    for (int i = threadIdx.x; i < rows * HD; i += 64 * F32_PREFILL_NW) o[(size_t)(i / HD) * (H * HD) + (i & (HD - 1))] = 0.f;
    return;
  }
  const float* base = qkv + (size_t)b * S * LD;
  f32_attn_group<F32_BIDIR_QT, F32_PREFILL_NW, 0>(base + (size_t)s0 * LD + head * HD, 0, k_hi, scale, o, smem, kst,
                                                  [&](int j) { return base + (size_t)j * LD + (H + head) * HD; },
                                                  [&](int j) { return base + (size_t)j * LD + (2 * H + head) * HD; }, LD, H * HD, nvalid, rows);
}

// decode: one workgroup per (beam row, KV head); key i of row m lives in cache row index[m, i] (tasu_kv_index_*), keys
// [kstart[m], lens[m]); the row's index entries are staged in LDS first (one round trip in front of the K / V loads, not one per key)
template <int REP>
__global__ __launch_bounds__(64 * F32_DECODE_NW) void f32_attn_decode_kernel(const float* __restrict__ qkv, const float* __restrict__ kc,
                                                              const float* __restrict__ vc, const int32_t* __restrict__ index,
                                                              const int32_t* __restrict__ kstart, const int32_t* __restrict__ lens,
                                                              float* __restrict__ out, int M, int H, int G, int ctx, float scale, int kst) {
  extern __shared__ float smem[];
  const int g = blockIdx.x % G, m = blockIdx.x / G;
  const int LD = (H + 2 * G) * HD, Wd = G * HD;
  int* six = (int*)(smem + f32_attn_lds_floats<REP, F32_DECODE_NW>(kst));
  const int k_lo = kstart[m], k_hi = lens[m];
  for (int i = k_lo + threadIdx.x; i < k_hi; i += 64 * F32_DECODE_NW) six[i] = index[(size_t)m * ctx + i];
  __syncthreads();
  f32_attn_group<REP, F32_DECODE_NW, F32_DECODE_VPRE>(qkv + (size_t)m * LD + g * REP * HD, k_lo, k_hi, scale, out + (size_t)m * (H * HD) + g * REP * HD, smem, kst,
                      [&](int j) { return kc + ((size_t)six[j] * ctx + j) * Wd + g * HD; },
                      [&](int j) { return vc + ((size_t)six[j] * ctx + j) * Wd + g * HD; });
}

// Qwen2MLP: act = silu(gate) * up over gu [M, 2I] (gate columns first), fp32
__global__ __launch_bounds__(256) void f32_swiglu_kernel(const float* __restrict__ gu, float* __restrict__ act, int M, int I) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)M * I) return;
  const size_t m = idx / I, c = idx - m * I;
  act[idx] = __fmul_rn(silu_exact(gu[m * 2 * I + c]), gu[m * 2 * I + I + c]);
}

// x[m, :] = table[idx] | proj[idx] | 0 (tasu_embed_merge_fwd with an fp32 projector output)
__global__ __launch_bounds__(256) void f32_embed_merge_kernel(const float* __restrict__ table, const float* __restrict__ proj, int ldp,
                                                              const int32_t* __restrict__ kind, const int32_t* __restrict__ src,
                                                              float* __restrict__ x, int M, int D) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)M * D) return;
  const int m = (int)(idx / D), c = (int)(idx - (size_t)m * D);
  const int k = kind[m];
  x[idx] = k == 1 ? table[(size_t)src[m] * D + c] : (k == 2 ? proj[(size_t)src[m] * ldp + c] : 0.f);
}

// log_softmax + top-k of one fp32 logits row per 1024-thread block: (x - max) - log(sum exp(x - max)) like torch.log_softmax; the k
// best selectable columns in the order (value descending, column ascending); banned columns never qualify
// (MinLengthLogitsProcessor sets them to -inf after the softmax).  Threshold form (three passes over the row instead of 2 + k):
// tau = the k-th largest of the 1024 per-thread maxima over selectable columns -- at least k columns are >= tau, so the k best all
// are; the columns >= tau (a handful) are collected in LDS during the sum-of-exp pass and ranked.  More than CAND of them (massive
// ties): the round-by-round form below takes over.
constexpr int F32_TOPK_CAND = 512;
__global__ __launch_bounds__(1024) void f32_logprob_topk_kernel(const float* __restrict__ logits, int ld, int V, int k,
                                                                const int32_t* __restrict__ banned, int n_banned,
                                                                float* __restrict__ out_val, int32_t* __restrict__ out_idx) {
  __shared__ float red[16];
  __shared__ float bv[16];
  __shared__ int bi[16];
  __shared__ float cand_v[F32_TOPK_CAND];
  __shared__ int cand_i[F32_TOPK_CAND];
  __shared__ int cand_n;
  const float* x = logits + (size_t)blockIdx.x * ld;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  auto is_banned = [&](int c) {
    bool ban = false;
    for (int b = 0; b < n_banned; ++b) ban |= banned[b] == c;
    return ban;
  };
  if (t == 0) cand_n = 0;
  float m = -__builtin_inff(), msel = -__builtin_inff();
  for (int c = t; c < V; c += 1024) {
    const float v = x[c];
    m = fmaxf(m, v);
    if (v > msel && !is_banned(c)) msel = v;
  }
  m = block_max<16>(m, red);
  // tau: k rounds of (block maximum of the per-thread selectable maxima, retire one thread that holds it)
  float mine = msel, tau = -__builtin_inff();
  for (int r = 0; r < k; ++r) {
    float best = mine;
    int who = t;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int ow = __shfl_xor(who, o, 64);
      if (ov > best || (ov == best && ow < who)) best = ov, who = ow;
    }
    __syncthreads();
    if (lane == 0) bv[wave] = best, bi[wave] = who;
    __syncthreads();
    best = bv[0], who = bi[0];
#pragma unroll
    for (int w = 1; w < 16; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < who)) best = bv[w], who = bi[w];
    tau = best;
    if (t == who) mine = -__builtin_inff();
  }
  float s = 0.f;
  for (int c = t; c < V; c += 1024) {
    const float v = x[c];
    s += expf(v - m);
    if (v >= tau && v > -__builtin_inff() && !is_banned(c)) {
      const int slot = atomicAdd(&cand_n, 1);
      if (slot < F32_TOPK_CAND) cand_v[slot] = v, cand_i[slot] = c;
    }
  }
  s = block_sum<16>(s, red);                         // (its barriers also publish the candidates)
  const float lse = logf(s);
  const int n_cand = cand_n;
  if (n_cand <= F32_TOPK_CAND) {
    if (t < n_cand) {
      const float v = cand_v[t];
      const int id = cand_i[t];
      int rank = 0;
      for (int d = 0; d < n_cand; ++d) rank += (cand_v[d] > v || (cand_v[d] == v && cand_i[d] < id)) ? 1 : 0;
      if (rank < k) {
        out_val[(size_t)blockIdx.x * k + rank] = (v - m) - lse;
        out_idx[(size_t)blockIdx.x * k + rank] = id;
      }
    }
    if (t >= n_cand && t < k) {                      // fewer than k selectable columns
      out_val[(size_t)blockIdx.x * k + t] = -__builtin_inff();
      out_idx[(size_t)blockIdx.x * k + t] = 0x7fffffff;
    }
    return;
  }
  // general form: k rounds of "the best column after the previous pick"
  float pv = __builtin_inff();
  int pi = -1;
  for (int r = 0; r < k; ++r) {
    float best = -__builtin_inff();
    int bid = 0x7fffffff;
    for (int c = t; c < V; c += 1024) {
      const float v = x[c];
      const bool after = v < pv || (v == pv && c > pi);
      if (!after || v < best || (v == best && c > bid)) continue;
      if (!is_banned(c)) best = v, bid = c;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bid, o, 64);
      if (ov > best || (ov == best && oi < bid)) best = ov, bid = oi;
    }
    __syncthreads();
    if (lane == 0) bv[wave] = best, bi[wave] = bid;
    __syncthreads();
    best = bv[0], bid = bi[0];
#pragma unroll
    for (int w = 1; w < 16; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < bid)) best = bv[w], bid = bi[w];
    pv = best, pi = bid;
    if (t == 0) {
      out_val[(size_t)blockIdx.x * k + r] = bid == 0x7fffffff ? -__builtin_inff() : (best - m) - lse;
      out_idx[(size_t)blockIdx.x * k + r] = bid;
    }
    if (bid == 0x7fffffff) pv = -__builtin_inff();     // fewer than k selectable columns: the remaining picks are empty too
  }
}

// FSMN memory block of the SANM layer in fp32 (SenseVoice.py:124-140; tasu_fsmn_fwd with an fp32 v): out[b, t, :] += depthwise
// conv over time (ksize taps, centred) of the masked v + the masked v itself, frames t >= lens[b] untouched.  v: column block of
// the fused q|k|v activation (row stride ldv).
__global__ __launch_bounds__(256) void f32_fsmn_kernel(const float* __restrict__ v, int ldv, const float* __restrict__ w,
                                                       const int32_t* __restrict__ lens, float* __restrict__ out, int T, int D, int ksize,
                                                       long long total) {
  const int left = (ksize - 1) / 2;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int d = (int)(i % D);
  const long long bt = i / D;
  const int t = (int)(bt % T), b = (int)(bt / T), len = lens[b];
  if (t >= len) return;
  float r = 0.f;
  for (int j = 0; j < ksize; ++j) {
    const int tt = t + j - left;
    if (tt >= 0 && tt < len) r += w[d * ksize + j] * v[((size_t)b * T + tt) * ldv + d];
  }
  r += v[((size_t)b * T + t) * ldv + d];
  out[i] += r;
}

// The same selection with the row split over F32_TOPK_PARTS workgroups (a generated position's 64 rows are 64 workgroups for the
// kernel above: a quarter of the CUs, three dependent passes over 600 KB each -- 132 us per position).  Stage 1, grid (rows, parts),
// 256 threads: a part's columns live in registers (one read); its maximum and sum of exp(x - max), and its k best selectable columns
// by the threshold form (tau = the largest over the four waves of the wave's k-th largest per-thread maximum) or, on massive ties,
// by k rounds of block argmax over the registers.  Stage 2, one wave per row: lse = log sum_p s_p exp(m_p - M) and the k best of the
// parts' candidates, (value descending, column ascending) as everywhere.
constexpr int F32_TOPK_PARTS = 16, F32_TOPK_MAXC = 10, F32_TOPK_PCAND = 64;
__global__ __launch_bounds__(256) void f32_topk_part_kernel(const float* __restrict__ logits, int ld, int V, int k,
                                                            const int32_t* __restrict__ banned, int n_banned, float* __restrict__ pm,
                                                            float* __restrict__ ps, float* __restrict__ pv, int32_t* __restrict__ pi) {
  __shared__ float red[4];
  __shared__ float wtau[4];
  __shared__ float bv[4];
  __shared__ int bi[4];
  __shared__ float cand_v[F32_TOPK_PCAND];
  __shared__ int cand_i[F32_TOPK_PCAND];
  __shared__ int cand_n;
  const int row = blockIdx.x, part = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* x = logits + (size_t)row * ld;
  const int nv4 = (V + 3) >> 2, per4 = (nv4 + F32_TOPK_PARTS - 1) / F32_TOPK_PARTS;
  const int v0 = part * per4, v1 = min(nv4, v0 + per4);
  auto is_banned = [&](int c) {
    bool ban = false;
    for (int b = 0; b < n_banned; ++b) ban |= banned[b] == c;
    return ban;
  };
  if (t == 0) cand_n = 0;
  f32x4 xs[F32_TOPK_MAXC];
#pragma unroll
  for (int i = 0; i < F32_TOPK_MAXC; ++i) {
    const int c4 = v0 + t + i * 256;
    xs[i] = c4 < v1 ? *(const f32x4*)(x + (size_t)c4 * 4) : f32x4{-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c4 * 4 + j >= V) xs[i][j] = -__builtin_inff();         // (columns past V: padding of the leading dimension)
  }
  float m = -__builtin_inff(), msel = -__builtin_inff();
#pragma unroll
  for (int i = 0; i < F32_TOPK_MAXC; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v = xs[i][j];
      m = fmaxf(m, v);
      if (v > msel && !is_banned((v0 + t + i * 256) * 4 + j)) msel = v;
    }
  m = block_max<4>(m, red);
  float mine = msel, kth = -__builtin_inff();
  for (int r = 0; r < k; ++r) {
    kth = wave_max(mine);
    const unsigned long long holders = __ballot(mine == kth);
    if (lane == __ffsll((long long)holders) - 1) mine = -__builtin_inff();
  }
  if (lane == 0) wtau[wave] = kth;
  __syncthreads();
  const float tau = fmaxf(fmaxf(wtau[0], wtau[1]), fmaxf(wtau[2], wtau[3]));
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < F32_TOPK_MAXC; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v = xs[i][j];
      const int c = (v0 + t + i * 256) * 4 + j;
      if (v > -__builtin_inff()) {
        s += expf(v - m);
        if (v >= tau && !is_banned(c)) {
          const int slot = atomicAdd(&cand_n, 1);
          if (slot < F32_TOPK_PCAND) cand_v[slot] = v, cand_i[slot] = c;
        }
      }
    }
  s = block_sum<4>(s, red);                              // (its barriers also publish the candidates)
  const size_t slot0 = (size_t)row * F32_TOPK_PARTS + part;
  if (t == 0) pm[slot0] = m, ps[slot0] = s;
  const int n_cand = cand_n;
  if (n_cand <= F32_TOPK_PCAND) {
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
      if (live && rank < k) pv[slot0 * k + rank] = v, pi[slot0 * k + rank] = id;
      if (lane >= n_cand && lane < k) pv[slot0 * k + lane] = -__builtin_inff(), pi[slot0 * k + lane] = 0x7fffffff;
    }
    return;
  }
  // massive ties: k rounds of "the best column after the previous pick" over the registers
  float pvv = __builtin_inff();
  int pii = -1;
  for (int r = 0; r < k; ++r) {
    float best = -__builtin_inff();
    int bid = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < F32_TOPK_MAXC; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = xs[i][j];
        const int c = (v0 + t + i * 256) * 4 + j;
        if (!(v > -__builtin_inff())) continue;
        const bool after = v < pvv || (v == pvv && c > pii);
        if (!after || v < best || (v == best && c > bid)) continue;
        if (!is_banned(c)) best = v, bid = c;
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bid, o, 64);
      if (ov > best || (ov == best && oi < bid)) best = ov, bid = oi;
    }
    __syncthreads();
    if (lane == 0) bv[wave] = best, bi[wave] = bid;
    __syncthreads();
    best = bv[0], bid = bi[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < bid)) best = bv[w], bid = bi[w];
    if (t == 0) pv[slot0 * k + r] = bid == 0x7fffffff ? -__builtin_inff() : best, pi[slot0 * k + r] = bid;
    pvv = best, pii = bid;
    if (bid == 0x7fffffff) pvv = -__builtin_inff(), pii = 0x7fffffff;
  }
}
__global__ __launch_bounds__(256) void f32_topk_merge_kernel(const float* __restrict__ pm, const float* __restrict__ ps,
                                                             const float* __restrict__ pv, const int32_t* __restrict__ pi, int k,
                                                             float* __restrict__ out_val, int32_t* __restrict__ out_idx) {
  __shared__ float cv[F32_TOPK_PARTS * 16];
  __shared__ int ci[F32_TOPK_PARTS * 16];
  __shared__ float stat[2];
  __shared__ int nvalid;
  const int row = blockIdx.x, t = threadIdx.x, lane = t & 63;
  const int n = F32_TOPK_PARTS * k;                      // <= 256: one candidate per thread
  if (t == 0) nvalid = 0;
  float v = -__builtin_inff();
  int id = 0x7fffffff;
  if (t < n) v = pv[(size_t)row * n + t], id = pi[(size_t)row * n + t];
  cv[t] = v, ci[t] = id;
  if (t < 64) {                                          // wave 0: the row's log-sum-exp from the parts' (max, sum)
    const bool lp = lane < F32_TOPK_PARTS;
    const float mp = lp ? pm[(size_t)row * F32_TOPK_PARTS + lane] : -__builtin_inff();
    const float sp = lp ? ps[(size_t)row * F32_TOPK_PARTS + lane] : 0.f;
    const float M = wave_max(mp);
    const float S = wave_sum(lp && mp > -__builtin_inff() ? sp * expf(mp - M) : 0.f);
    if (lane == 0) stat[0] = M, stat[1] = logf(S);
  }
  __syncthreads();
  if (id != 0x7fffffff) {
    atomicAdd(&nvalid, 1);
    int rank = 0;
    for (int d = 0; d < n; ++d) rank += (cv[d] > v || (cv[d] == v && ci[d] < id)) ? 1 : 0;
    if (rank < k) out_val[(size_t)row * k + rank] = (v - stat[0]) - stat[1], out_idx[(size_t)row * k + rank] = id;
  }
  __syncthreads();
  if (t >= nvalid && t < k) out_val[(size_t)row * k + t] = -__builtin_inff(), out_idx[(size_t)row * k + t] = 0x7fffffff;   // fewer than k selectable columns
}

// Shifted cross entropy of one fp32 logits row per 1024-thread block (the eval-mode forward in fp32: loss_utils' CE over the rows
// whose shifted label is >= 0, ignore_index -100): lse = max + log(sum exp(x - max)), row_loss = lse - x[label], row_hit =
// (argmax == label), argmax ties -> the first column (torch.argmax).  Rows without a label: loss 0, hit 0 (lse / argmax still written).
__global__ __launch_bounds__(1024) void f32_ce_kernel(const float* __restrict__ logits, int ld, const int32_t* __restrict__ labels, int V,
                                                      float* __restrict__ row_loss, int32_t* __restrict__ row_hit,
                                                      int32_t* __restrict__ row_argmax, float* __restrict__ row_lse,
                                                      float* __restrict__ dlogits, const float* __restrict__ inv_count) {
  __shared__ float red[16];
  __shared__ float bv[16];
  __shared__ int bi[16];
  const int m = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* x = logits + (size_t)m * ld;
  float best = -__builtin_inff();
  int bid = 0x7fffffff;
  for (int c = t; c < V; c += 1024) {
    const float v = x[c];
    if (v > best) best = v, bid = c;                 // (ascending c per thread: the first maximum wins)
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bid, o, 64);
    if (ov > best || (ov == best && oi < bid)) best = ov, bid = oi;
  }
  if (lane == 0) bv[wave] = best, bi[wave] = bid;
  __syncthreads();
  best = bv[0], bid = bi[0];
#pragma unroll
  for (int w = 1; w < 16; ++w)
    if (bv[w] > best || (bv[w] == best && bi[w] < bid)) best = bv[w], bid = bi[w];
  float s = 0.f;
  for (int c = t; c < V; c += 1024) s += expf(x[c] - best);
  s = block_sum<16>(s, red);
  if (t == 0) {
    const float lse = best + logf(s);
    const int lab = labels[m];
    const bool on = lab >= 0 && lab < V;
    row_loss[m] = on ? lse - x[lab] : 0.f;
    row_hit[m] = on && bid == lab ? 1 : 0;
    if (row_argmax) row_argmax[m] = bid;
    if (row_lse) row_lse[m] = lse;
  }
  if (dlogits) {                                     // d(mean CE) / d logits = (softmax - onehot) / count on labelled rows, 0 elsewhere
    const float lse = best + logf(s);
    const int lab = labels[m];
    const bool on = lab >= 0 && lab < V;
    const float k = on ? inv_count[0] : 0.f;
    float* d = dlogits + (size_t)m * ld;
    for (int c = t; c < ld; c += 1024) d[c] = c < V ? k * (expf(x[c] - lse) - (c == lab ? 1.f : 0.f)) : 0.f;   // (may alias the logits: own row, own columns)
  }
}

}  // namespace tasu_f32

using namespace tasu_f32;

// (lab build only -- common.h tasu_lab_env: the shipped library reads no tuning variable)
static int f32_env_int(const char* name, int dflt) {
  const char* e = tasu_lab_env(name);
  return e ? atoi(e) : dflt;
}
// the K-range split of a problem (1 = none) on the tile kernel: outputs of fewer than 1024 tiles run as K-range slabs until ~1024
// workgroups stream (a workgroup keeps 16 KiB in flight; the chip needs a few per CU to reach the HBM rate), eight slabs at most
static int f32_tile_ksplit(int M, int N, int K, const float* workspace, int64_t workspace_floats) {
  const int tiles = ((N + BN - 1) / BN) * ((M + BM - 1) / BM);
  int ksplit = 1;
  if (workspace && tiles < 1024) {
    // at most 8 slabs: q|k|v and o of a decode step measured 4.41 / 4.41 / 4.28 / 4.31 / 4.45 ms per position with caps of 16 / 12 / 8 / 6 / 4
    // (fewer slabs for the finisher to read against fewer workgroups streaming; TASU_F32_TILE_KSPLIT_MAX in the lab build)
    static const int cap = std::max(1, std::min(16, f32_env_int("TASU_F32_TILE_KSPLIT_MAX", 8)));
    ksplit = (1024 + tiles - 1) / tiles;
    if (ksplit > cap) ksplit = cap;
    while (ksplit > 1 && ((K / BK) % ksplit || (int64_t)ksplit * M * N > workspace_floats)) --ksplit;
  }
  return ksplit;
}
// ... and on the streaming kernel (M <= 64, K a multiple of 128): a workgroup's K range is 128 KS, so ksplit = K / (128 KS).  The
// KS that costs least under a two-term model -- matrix pipes (a tile is KS x 4 x RB MFMAs of 32 cycles per wave, two waves per pipe;
// a workgroup walks ceil(tiles / nbx) tiles) against HBM (the weights once, every slab written and read once, ~5 TB/s).  Few slabs
// for the wide projections (gate|up: KS = 6, 2 slabs), more for the narrow ones.
constexpr int SW_KS_MAX = 6;
struct F32StreamPlan {
  int ks = 0, ksplit = 1, nbx = 1;
};
static int f32_stream_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
              ? prop.multiProcessorCount : 256;
  }
  return cus;
}
static F32StreamPlan f32_stream_plan(int M, int N, int K, const float* workspace, int64_t workspace_floats) {
  static const int off = f32_env_int("TASU_F32_STREAM", 1) == 0, forced = f32_env_int("TASU_F32_STREAM_KS", 0);   // (tools: A/B runs, KS sweeps)
  F32StreamPlan best;
  // (matrices under 32 MB -- q|k|v, o at 1.5B -- stay on the tile kernel: one or two tiles per workgroup do not repay the 64-row
  // activation slice every workgroup stages; measured 12.7 / 13.8 us there against 13-17 here, tools/prof_f32_stream.sh)
  static const int min_mb = f32_env_int("TASU_F32_STREAM_MIN_MB", 32);
  if (off || M > 64 || K % (16 * SW_NW) || (int64_t)N * K * 4 < ((int64_t)min_mb << 20)) return best;
  // (the plan is that of 64 rows whatever M is: a row's bits do not depend on how many rows travel with it)
  const int kq = K / (16 * SW_NW), tiles = (N + 15) / 16, cus = f32_stream_cus(), rb = 4;
  double best_cost = 1e30;
  for (int ks = 1; ks <= SW_KS_MAX; ++ks) {
    if (kq % ks || (forced > 0 && ks != forced)) continue;
    const int s = kq / ks;
    if (s > 16 || (s > 1 && (!workspace || (int64_t)s * 64 * N > workspace_floats))) continue;
    const int nbx = std::max(1, std::min(tiles, cus / s));
    const double t_mfma = (double)((tiles + nbx - 1) / nbx) * ks * rb * 0.107;                         // us
    const double t_hbm = ((double)N * K * 4 + (s > 1 ? 2.0 * s * 64 * N * 4 : 0.0)) / 5.0e6;          // us
    const double cost = (t_mfma > t_hbm ? t_mfma : t_hbm) + 0.05 * s;                                 // (ties: fewer slabs)
    if (cost < best_cost) best_cost = cost, best = F32StreamPlan{ks, s, nbx};
  }
  return best;
}
static int f32_ksplit(int M, int N, int K, const float* workspace, int64_t workspace_floats) {
  const F32StreamPlan sp = f32_stream_plan(M, N, K, workspace, workspace_floats);
  return sp.ks ? sp.ksplit : f32_tile_ksplit(M, N, K, workspace, workspace_floats);
}
// ldw == TASU_F32_LDW_FRAGMENT (-1): W is the fragment-order copy (tasu_f32_to_fragment_order) -- served by the streaming kernel only
static bool f32_gemm_args_ok(const float* A, int lda, const float* W, int ldw, const float* C, int ldc, int M, int N, int K) {
  return A && W && C && M > 0 && N > 0 && K > 0 && K % BK == 0 && lda % 4 == 0 && (ldw == TASU_F32_LDW_FRAGMENT || (ldw >= K && ldw % 4 == 0)) &&
         ldc >= N && !(((uintptr_t)A | (uintptr_t)W) & 15);
}
template <int KS, int RB>
static int f32_stream_launch(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias, const float* resid, int M,
                             int N, int act, const F32StreamPlan& sp, hipStream_t st) {
  const int lds = std::max(2 * SW_NW * RB * 256, SW_NW * (RB >= 2 ? 32 : 16) * (16 * KS + 4)) * 4;     // partial tiles | activation images
  static bool set = false;
  if (!set) {
    (void)hipFuncSetAttribute((const void*)f32_stream_kernel<KS, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    set = true;
  }
  TASU_LAUNCH((f32_stream_kernel<KS, RB>), dim3(sp.nbx, sp.ksplit), dim3(64 * SW_NW), lds, st, A, lda, W, ldw, C, ldc, bias, resid, M, N, act,
              sp.ksplit, (N + 15) / 16, sp.nbx, sp.ks * sp.ksplit * SW_NW);
  return TASU_OK;
}
template <int KS>
static int f32_stream_launch_rb(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias, const float* resid,
                                int M, int N, int act, const F32StreamPlan& sp, hipStream_t st) {
  switch ((M + 15) / 16) {
    case 1: return f32_stream_launch<KS, 1>(A, lda, W, ldw, C, ldc, bias, resid, M, N, act, sp, st);
    case 2: return f32_stream_launch<KS, 2>(A, lda, W, ldw, C, ldc, bias, resid, M, N, act, sp, st);
    default: return f32_stream_launch<KS, 4>(A, lda, W, ldw, C, ldc, bias, resid, M, N, act, sp, st);      // (33-48 rows run as 64)
  }
}
static int f32_stream_dispatch(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias, const float* resid, int M,
                               int N, int act, const F32StreamPlan& sp, hipStream_t st) {
  switch (sp.ks) {
    case 1: return f32_stream_launch_rb<1>(A, lda, W, ldw, C, ldc, bias, resid, M, N, act, sp, st);
    case 2: return f32_stream_launch_rb<2>(A, lda, W, ldw, C, ldc, bias, resid, M, N, act, sp, st);
    case 3: return f32_stream_launch_rb<3>(A, lda, W, ldw, C, ldc, bias, resid, M, N, act, sp, st);
    case 4: return f32_stream_launch_rb<4>(A, lda, W, ldw, C, ldc, bias, resid, M, N, act, sp, st);
    case 5: return f32_stream_launch_rb<5>(A, lda, W, ldw, C, ldc, bias, resid, M, N, act, sp, st);
    default: {
#ifdef TASU_LAB
      static const int dbg = f32_env_int("TASU_F32_STREAM_DBG", 0);          // (tools/prof_f32_stream.sh: timing probes, wrong results)
      if ((dbg == 1 || dbg == 2 || dbg == 4) && M > 48) {
        const int lds = std::max(2 * SW_NW * 4 * 256, SW_NW * 32 * (16 * 6 + 4)) * 4;
        auto probe = [&](auto kernel) {
          (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
          TASU_LAUNCH(kernel, dim3(sp.nbx, sp.ksplit), dim3(64 * SW_NW), lds, st, A, lda, W, ldw, C, ldc, bias, resid, M, N, act, sp.ksplit,
                      (N + 15) / 16, sp.nbx, sp.ks * sp.ksplit * SW_NW);
          return TASU_OK;
        };
        if (dbg == 1) return probe(f32_stream_kernel<6, 4, 1>);
        if (dbg == 2) return probe(f32_stream_kernel<6, 4, 2>);
        return probe(f32_stream_kernel<6, 4, 4>);
      }
#endif
      return f32_stream_launch_rb<6>(A, lda, W, ldw, C, ldc, bias, resid, M, N, act, sp, st);
    }
  }
}
// `ksplit` is what f32_ksplit returned for this problem (the caller chose its finisher by it)
static int f32_gemm_launch(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias, const float* resid, int M,
                           int N, int K, int act, int ksplit, const float* workspace, int64_t workspace_floats, hipStream_t st) {
  const F32StreamPlan sp = f32_stream_plan(M, N, K, workspace, workspace_floats);
  if (sp.ks && sp.ksplit == ksplit) return f32_stream_dispatch(A, lda, W, ldw, C, ldc, bias, resid, M, N, act, sp, st);
  if (ldw < 0) return TASU_ERR_ARG;                       // (fragment order: the streaming kernel does not serve this problem)
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, ksplit);
  TASU_LAUNCH(f32_gemm_kernel, grid, dim3(256), 0, st, A, lda, W, ldw, C, ldc, bias, resid, M, N, K, K / ksplit, act, ksplit);
  return TASU_OK;
}

extern "C" int tasu_f32_gemm_nt(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias, const float* resid,
                                int M, int N, int K, int act, float* workspace, int64_t workspace_floats, void* stream) {
  if (!f32_gemm_args_ok(A, lda, W, ldw, C, ldc, M, N, K) || act < 0 || act > 2) return TASU_ERR_ARG;
  const int ksplit = f32_ksplit(M, N, K, workspace, workspace_floats);
  const int rc = f32_gemm_launch(A, lda, W, ldw, ksplit > 1 ? workspace : C, ldc, bias, resid, M, N, K, act, ksplit, workspace, workspace_floats, (hipStream_t)stream);
  if (rc || ksplit == 1) return rc;
  const size_t n = (size_t)M * N;
  TASU_LAUNCH(f32_sum_slabs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, workspace, ksplit, C, ldc, bias,
              resid, M, N, act);
  return TASU_OK;
}

// out[((t * (K / 16) + k16) * 64 + lane) * 4 + e] = W[16 t + (lane & 15)][16 k16 + 4 (lane >> 4) + e], rows >= N zero: a wave's operand
// pieces of a 16-row tile, contiguous in the order the streaming kernel requests them
__global__ __launch_bounds__(256) void f32_to_fragment_order_kernel(const float* __restrict__ W, int ldw, float* __restrict__ out, int N, int K) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;                // one 16-byte piece per thread
  const int k16n = K / 16;
  const size_t total = (size_t)((N + 15) / 16) * k16n * 64;
  if (idx >= total) return;
  const int lane = (int)(idx & 63);
  const size_t tk = idx >> 6;
  const int k16 = (int)(tk % k16n), t = (int)(tk / k16n);
  const int n = t * 16 + (lane & 15);
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
  if (n < N) v = *(const f32x4*)(W + (size_t)n * ldw + k16 * 16 + 4 * (lane >> 4));
  *(f32x4*)(out + idx * 4) = v;
}
extern "C" int tasu_f32_to_fragment_order(const float* W, int ldw, float* out, int N, int K, void* stream) {
  if (!W || !out || N <= 0 || K <= 0 || K % 16 || ldw < K || ldw % 4 || (((uintptr_t)W | (uintptr_t)out) & 15)) return TASU_ERR_ARG;
  const size_t total = (size_t)((N + 15) / 16) * (K / 16) * 64;
  TASU_LAUNCH(f32_to_fragment_order_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, ldw, out, N, K);
  return TASU_OK;
}
// 1 if a product of M rows with an [N, K] matrix runs on the streaming kernel (and may therefore be given the fragment-order copy)
extern "C" int tasu_f32_gemm_streams(int M, int N, int K, int64_t workspace_floats) {
  static float dummy;
  return f32_stream_plan(M, N, K, workspace_floats > 0 ? &dummy : nullptr, workspace_floats).ks > 0 ? 1 : 0;
}

// tasu_f32_gemm_nt on the streaming kernel with a given K slice per wave (tests and tools: the dispatcher above only sends matrices
// of 32 MB and more there); TASU_ERR_ARG when the problem does not fit it (M > 64, K not a multiple of 128 ks, slabs > workspace)
extern "C" int tasu_f32_gemm_stream(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias, const float* resid,
                                    int M, int N, int K, int act, int ks, float* workspace, int64_t workspace_floats, void* stream) {
  if (!f32_gemm_args_ok(A, lda, W, ldw, C, ldc, M, N, K) || act < 0 || act > 2 || M > 64 || ks < 1 || ks > SW_KS_MAX || K % (16 * SW_NW * ks))
    return TASU_ERR_ARG;
  F32StreamPlan sp;
  sp.ks = ks;
  sp.ksplit = K / (16 * SW_NW * ks);
  const int tiles = (N + 15) / 16;
  sp.nbx = std::max(1, std::min(tiles, f32_stream_cus() / sp.ksplit));
  if (sp.ksplit > 16 || (sp.ksplit > 1 && (!workspace || (int64_t)sp.ksplit * M * N > workspace_floats))) return TASU_ERR_ARG;
  const int rc = f32_stream_dispatch(A, lda, W, ldw, sp.ksplit > 1 ? workspace : C, ldc, bias, resid, M, N, act, sp, (hipStream_t)stream);
  if (rc || sp.ksplit == 1) return rc;
  const size_t n = (size_t)M * N;
  TASU_LAUNCH(f32_sum_slabs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, workspace, sp.ksplit, C, ldc, bias,
              resid, M, N, act);
  return TASU_OK;
}

// x = resid + A W^T [+ bias];  y = RMSNorm(x, norm_w): the o / down projection of a decoder layer with the norm that consumes it
extern "C" int tasu_f32_gemm_resid_rmsnorm(const float* A, int lda, const float* W, int ldw, float* x, int ldx, const float* bias,
                                           const float* resid, const float* norm_w, float* y, int M, int N, int K, float eps,
                                           float* workspace, int64_t workspace_floats, void* stream) {
  if (!f32_gemm_args_ok(A, lda, W, ldw, x, ldx, M, N, K) || !norm_w || !y) return TASU_ERR_ARG;
  const int ksplit = f32_ksplit(M, N, K, workspace, workspace_floats);
  if (ksplit == 1) {                                  // (whole-K tiles: the prompt pass) GEMM, then the norm
    const int rc = f32_gemm_launch(A, lda, W, ldw, x, ldx, bias, resid, M, N, K, 0, 1, workspace, workspace_floats, (hipStream_t)stream);
    if (rc) return rc;
    if (ldx != N) return TASU_ERR_ARG;
    TASU_LAUNCH(f32_rmsnorm_kernel, dim3(M), dim3(1024), 0, (hipStream_t)stream, x, norm_w, y, N, eps);
    return TASU_OK;
  }
  const int rc = f32_gemm_launch(A, lda, W, ldw, workspace, ldx, bias, resid, M, N, K, 0, ksplit, workspace, workspace_floats, (hipStream_t)stream);
  if (rc) return rc;
  TASU_LAUNCH(f32_sum_slabs_norm_kernel, dim3(M), dim3(1024), 0, (hipStream_t)stream, workspace, ksplit, x, ldx, bias, resid, norm_w, y, M, N,
              eps);
  return TASU_OK;
}

// act = silu(A Wg^T) * (A Wu^T) with Wgu = [Wg; Wu] ([2I, K]); gu: [M, 2I] scratch for the unsplit route (may be NULL when the
// problem splits: M <= 64 rows at the decoder's widths)
extern "C" int tasu_f32_gemm_swiglu(const float* A, int lda, const float* Wgu, int ldw, float* gu, float* act, int M, int I, int K,
                                    float* workspace, int64_t workspace_floats, void* stream) {
  if (!f32_gemm_args_ok(A, lda, Wgu, ldw, act, 2 * I, M, 2 * I, K) || I <= 0) return TASU_ERR_ARG;
  const int ksplit = f32_ksplit(M, 2 * I, K, workspace, workspace_floats);
  const size_t n = (size_t)M * I;
  if (ksplit == 1) {
    if (!gu) return TASU_ERR_ARG;
    const int rc = f32_gemm_launch(A, lda, Wgu, ldw, gu, 2 * I, nullptr, nullptr, M, 2 * I, K, 0, 1, workspace, workspace_floats, (hipStream_t)stream);
    if (rc) return rc;
    TASU_LAUNCH(f32_swiglu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gu, act, M, I);
    return TASU_OK;
  }
  const int rc = f32_gemm_launch(A, lda, Wgu, ldw, workspace, 2 * I, nullptr, nullptr, M, 2 * I, K, 0, ksplit, workspace, workspace_floats, (hipStream_t)stream);
  if (rc) return rc;
  TASU_LAUNCH(f32_sum_slabs_swiglu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, workspace, ksplit, act, M, I);
  return TASU_OK;
}

extern "C" int tasu_f32_rmsnorm(const float* x, const float* w, float* y, int M, int D, float eps, void* stream) {
  if (!x || !w || !y || M <= 0 || D <= 0) return TASU_ERR_ARG;
  TASU_LAUNCH(f32_rmsnorm_kernel, dim3(M), dim3(1024), 0, (hipStream_t)stream, x, w, y, D, eps);
  return TASU_OK;
}

extern "C" int tasu_f32_rope(float* qkv, const float* cos_tab, const float* sin_tab, int M, int H, int G, float* kcache, float* vcache,
                             const int32_t* slot, int ctx, int inverse, void* stream) {
  if (!qkv || !cos_tab || !sin_tab || M <= 0 || H <= 0 || G <= 0 || H % G) return TASU_ERR_ARG;
  if (kcache && (!vcache || !slot || ctx <= 0)) return TASU_ERR_ARG;
  const size_t n = (size_t)M * (H + 2 * G) * 64;
  TASU_LAUNCH(f32_rope_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, qkv, cos_tab, sin_tab, M, H, G, kcache,
              vcache, slot, ctx, inverse);
  return TASU_OK;
}

// qkv = A Wqkv^T + bias, q / k heads rotated, k / v appended to the cache (kcache may be NULL: the prompt pass stores with tasu_f32_kv_fill)
extern "C" int tasu_f32_gemm_qkv_rope(const float* A, int lda, const float* Wqkv, int ldw, const float* bias, float* qkv, const float* cos_tab,
                                      const float* sin_tab, int M, int H, int G, int K, float* kcache, float* vcache, const int32_t* slot,
                                      int ctx, float* workspace, int64_t workspace_floats, void* stream) {
  const int N = (H + 2 * G) * HD;
  if (H <= 0 || G <= 0 || H % G || !f32_gemm_args_ok(A, lda, Wqkv, ldw, qkv, N, M, N, K) || !cos_tab || !sin_tab) return TASU_ERR_ARG;
  if (kcache && (!vcache || !slot || ctx <= 0)) return TASU_ERR_ARG;
  const int ksplit = f32_ksplit(M, N, K, workspace, workspace_floats);
  const size_t n = (size_t)M * (H + 2 * G) * 64;
  if (ksplit == 1) {
    const int rc = f32_gemm_launch(A, lda, Wqkv, ldw, qkv, N, bias, nullptr, M, N, K, 0, 1, workspace, workspace_floats, (hipStream_t)stream);
    if (rc) return rc;
    TASU_LAUNCH(f32_rope_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, qkv, cos_tab, sin_tab, M, H, G, kcache,
                vcache, slot, ctx, 0);
    return TASU_OK;
  }
  const int rc = f32_gemm_launch(A, lda, Wqkv, ldw, workspace, N, nullptr, nullptr, M, N, K, 0, ksplit, workspace, workspace_floats, (hipStream_t)stream);
  if (rc) return rc;
  TASU_LAUNCH(f32_sum_slabs_rope_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, workspace, ksplit, qkv, bias,
              cos_tab, sin_tab, M, H, G, kcache, vcache, slot, ctx);
  return TASU_OK;
}

extern "C" int tasu_f32_kv_fill(const float* qkv, float* kcache, float* vcache, int B, int S, int H, int G, int n_beams, int ctx,
                                void* stream) {
  if (!qkv || !kcache || !vcache || B <= 0 || S <= 0 || S > ctx || n_beams <= 0) return TASU_ERR_ARG;
  const size_t n = (size_t)B * S * G * HD;
  TASU_LAUNCH(f32_kv_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, qkv, kcache, vcache, B, S, H, G,
              n_beams, ctx);
  return TASU_OK;
}

template <int REP>
static int f32_attn_launch(bool decode, const float* qkv, const float* kc, const float* vc, const int32_t* index, const int32_t* kstart,
                           const int32_t* lens, float* out, int rows, int S, int H, int G, int ctx, float scale, hipStream_t st) {
  // (prefill: `lens` = klen [B] or NULL)
  // score rows as long as this launch's longest key range (decode: the cache's ctx; prefill: S), rounded up to 64
  const int kst = ((decode ? ctx : S) + 63) & ~63;
  const int lds = decode ? (f32_attn_lds_floats<REP, F32_DECODE_NW>(kst) + kst) * 4 : f32_attn_lds_floats<REP, F32_PREFILL_NW>(kst) * 4;
  static bool set[2] = {false, false};
  if (!set[decode]) {                                 // (the attribute: the largest the kernels may ask for, MAX_KEYS keys)
    if (decode)
      (void)hipFuncSetAttribute((const void*)f32_attn_decode_kernel<REP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (f32_attn_lds_floats<REP, F32_DECODE_NW>(F32_ATTN_MAX_KEYS) + F32_ATTN_MAX_KEYS) * 4);
    else
      (void)hipFuncSetAttribute((const void*)f32_attn_prefill_kernel<REP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                f32_attn_lds_floats<REP, F32_PREFILL_NW>(F32_ATTN_MAX_KEYS) * 4);
    set[decode] = true;
  }
  if (decode) {
    TASU_LAUNCH(f32_attn_decode_kernel<REP>, dim3(rows * G), dim3(64 * F32_DECODE_NW), lds, st, qkv, kc, vc, index, kstart, lens, out, rows, H, G, ctx, scale, kst);
  } else if (REP == 1 && lens) {                       // bidirectional with key padding, H = G: tiles of F32_BIDIR_QT queries per workgroup
    const int lds_b = f32_attn_lds_floats<F32_BIDIR_QT, F32_PREFILL_NW>(kst) * 4;
    static bool set_b = false;
    if (!set_b) {
      (void)hipFuncSetAttribute((const void*)f32_attn_bidir_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                f32_attn_lds_floats<F32_BIDIR_QT, F32_PREFILL_NW>(F32_ATTN_MAX_KEYS) * 4);
      set_b = true;
    }
    const int Bn = rows / S, tiles = (S + F32_BIDIR_QT - 1) / F32_BIDIR_QT;
    TASU_LAUNCH(f32_attn_bidir_kernel, dim3((unsigned)((long long)Bn * tiles * H)), dim3(64 * F32_PREFILL_NW), lds_b, st, qkv, lens, out, Bn, S, H,
                scale, kst);
  } else {
    TASU_LAUNCH(f32_attn_prefill_kernel<REP>, dim3((unsigned)((long long)rows * G)), dim3(256), lds, st, qkv, kstart, lens, out, rows / S, S,
                H, G, scale, kst);
  }
  return TASU_OK;
}
static int f32_attn_dispatch(bool decode, const float* qkv, const float* kc, const float* vc, const int32_t* index, const int32_t* kstart,
                             const int32_t* lens, float* out, int rows, int S, int H, int G, int ctx, float scale, hipStream_t st) {
  switch (H / G) {
#define F32_ATTN_CASE(R) \
  case R: return f32_attn_launch<R>(decode, qkv, kc, vc, index, kstart, lens, out, rows, S, H, G, ctx, scale, st);
    F32_ATTN_CASE(1) F32_ATTN_CASE(2) F32_ATTN_CASE(3) F32_ATTN_CASE(4) F32_ATTN_CASE(5) F32_ATTN_CASE(6) F32_ATTN_CASE(7) F32_ATTN_CASE(8)
#undef F32_ATTN_CASE
    default: return TASU_ERR_ARG;
  }
}

extern "C" int tasu_f32_attn_prefill(const float* qkv, const int32_t* kstart, const int32_t* klen, float* out, int B, int S, int H, int G,
                                     float scale, void* stream) {
  if (!qkv || (!kstart && !klen) || !out || B <= 0 || S <= 0 || S > F32_ATTN_MAX_KEYS || H <= 0 || G <= 0 || H % G ||
      H / G > F32_ATTN_MAX_REP)
    return TASU_ERR_ARG;
  return f32_attn_dispatch(false, qkv, nullptr, nullptr, nullptr, kstart, klen, out, B * S, S, H, G, 0, scale, (hipStream_t)stream);
}

extern "C" int tasu_f32_attn_decode(const float* qkv, const float* kcache, const float* vcache, const int32_t* row_index,
                                    const int32_t* kstart, const int32_t* lens, float* out, int M, int H, int G, int ctx, float scale,
                                    void* stream) {
  if (!qkv || !kcache || !vcache || !row_index || !kstart || !lens || !out || M <= 0 || H <= 0 || G <= 0 || H % G ||
      H / G > F32_ATTN_MAX_REP || ctx <= 0 || ctx > F32_ATTN_MAX_KEYS)
    return TASU_ERR_ARG;
  return f32_attn_dispatch(true, qkv, kcache, vcache, row_index, kstart, lens, out, M, 1, H, G, ctx, scale, (hipStream_t)stream);
}

extern "C" int tasu_f32_fsmn(const float* v, int ldv, const float* w, const int32_t* lens, float* out, int B, int T, int D, int ksize,
                             void* stream) {
  if (!v || !w || !lens || !out || B <= 0 || T <= 0 || D <= 0 || ksize <= 0 || ldv < D) return TASU_ERR_ARG;
  const long long total = (long long)B * T * D;
  TASU_LAUNCH(f32_fsmn_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, v, ldv, w, lens, out, T, D, ksize,
              total);
  return TASU_OK;
}

extern "C" int tasu_f32_swiglu(const float* gu, float* act, int M, int I, void* stream) {
  if (!gu || !act || M <= 0 || I <= 0) return TASU_ERR_ARG;
  const size_t n = (size_t)M * I;
  TASU_LAUNCH(f32_swiglu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gu, act, M, I);
  return TASU_OK;
}

extern "C" int tasu_f32_embed_merge(const float* table, const float* proj, int ldp, const int32_t* src_kind, const int32_t* src_idx,
                                    float* x, int M, int D, void* stream) {
  if (!table || !proj || !src_kind || !src_idx || !x || M <= 0 || D <= 0 || ldp < D) return TASU_ERR_ARG;
  const size_t n = (size_t)M * D;
  TASU_LAUNCH(f32_embed_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, table, proj, ldp, src_kind,
              src_idx, x, M, D);
  return TASU_OK;
}

extern "C" int tasu_f32_ce(const float* logits, int ld, const int32_t* shift_labels, int M, int V, float* row_loss, int32_t* row_hit,
                           int32_t* row_argmax, float* row_lse, float* dlogits, const float* inv_count, void* stream) {
  if (!logits || !shift_labels || !row_loss || !row_hit || M <= 0 || V <= 0 || ld < V || (dlogits && !inv_count)) return TASU_ERR_ARG;
  TASU_LAUNCH(f32_ce_kernel, dim3(M), dim3(1024), 0, (hipStream_t)stream, logits, ld, shift_labels, V, row_loss, row_hit, row_argmax, row_lse,
              dlogits, inv_count);
  return TASU_OK;
}

extern "C" int tasu_f32_logprob_topk(const float* logits, int ld, int M, int V, int k, const int32_t* banned, int n_banned, float* out_val,
                                     int32_t* out_idx, float* workspace, int64_t workspace_floats, void* stream) {
  if (!logits || !out_val || !out_idx || M <= 0 || V <= 0 || ld < V || k <= 0 || k > 16 || n_banned < 0 || (n_banned > 0 && !banned))
    return TASU_ERR_ARG;
  // with a workspace of M * 16 * (2 + 2 k) floats, 16-byte aligned rows and a vocabulary the parts hold in registers: two launches,
  // the row split over 16 workgroups; otherwise one workgroup per row
  const size_t slots = (size_t)M * F32_TOPK_PARTS;
  const int per4 = (((V + 3) >> 2) + F32_TOPK_PARTS - 1) / F32_TOPK_PARTS;
  if (workspace && (size_t)workspace_floats >= slots * (2 + 2 * (size_t)k) && ld % 4 == 0 && !((uintptr_t)logits & 15) &&
      per4 <= 256 * F32_TOPK_MAXC && M > 1) {
    float* pm = workspace;
    float* ps = pm + slots;
    float* pv = ps + slots;
    int32_t* pi = (int32_t*)(pv + slots * k);
    TASU_LAUNCH(f32_topk_part_kernel, dim3(M, F32_TOPK_PARTS), dim3(256), 0, (hipStream_t)stream, logits, ld, V, k, banned, n_banned, pm, ps, pv,
                pi);
    TASU_LAUNCH(f32_topk_merge_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, pm, ps, pv, pi, k, out_val, out_idx);
    return TASU_OK;
  }
  TASU_LAUNCH(f32_logprob_topk_kernel, dim3(M), dim3(1024), 0, (hipStream_t)stream, logits, ld, V, k, banned, n_banned, out_val, out_idx);
  return TASU_OK;
}
