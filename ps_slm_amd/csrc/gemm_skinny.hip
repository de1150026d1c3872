// Weight-streaming bf16 NT GEMM for M <= 64 rows (the decode step: M = batch x beams): C[M,N] = A[M,K] . B[N,K]^T.
// HBM-bound (every weight byte is read once per step), so the design goal is bytes in flight, not MFMA rate:
//   * grid = (N / 64 column tiles) x (K splits), sized to cover the 256 CUs even for N = 1536;
//   * a block is 4 INDEPENDENT waves: wave w owns the K-steps  w, w+4, ...  of the block's K range and a PRIVATE
//     double-buffered LDS region filled by its own global_load_lds_dwordx4 -- no block barrier in the main loop, only
//     the wave's counted s_waitcnt vmcnt (data a wave DMA'd itself needs no barrier);
//   * the 4 waves' 64x64 fp32 partials are summed through LDS once, and the K splits through fp32 slabs + a small
//     reduce/epilogue kernel (bias, bf16 rounding, residual add): deterministic, no atomics.
#include <type_traits>

#include "common.h"
#include "../../include/tasu_hip.h"

namespace tasu_skinny {

constexpr int BM = 64, BN = 64, BK = 64;
constexpr int TILE_BYTES = 64 * BK * 2;              // one 64 x 64 bf16 operand tile = 8 KiB = 8 LDS-DMA pieces
constexpr int WAVE_STAGE = 2 * TILE_BYTES;           // A + B
constexpr int WAVE_LDS = 2 * WAVE_STAGE;             // double buffered: 32 KiB per wave, 128 KiB per block

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

struct Args {
  const bf16* A;
  const bf16* B;
  float* slab;          // [ksplit][64][ldn] fp32 partials (ksplit > 1) ...
  void* C;              // ... or the final output (ksplit == 1)
  const float* R;
  const bf16* bias;
  int M, N, K, lda, ldb, ldc, ldn;
  int ksplit, out_mode;
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

__global__ __launch_bounds__(256, 1) void gemm_skinny_kernel(Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tn = blockIdx.x, ks = blockIdx.y;
  const int col0 = tn * BN;
  const int nk = p.K / BK;
  const int per = (nk + p.ksplit - 1) / p.ksplit;
  const int kbeg = ks * per, kend = min(nk, kbeg + per);
  char* my = smem + wave * WAVE_LDS;

  // per-lane source pointers of the 8 + 8 pieces of one K-step (swizzled like gemm.hip: chunk c of row r at c ^ ((r>>1)&7))
  const bf16* ga[8];
  const bf16* gb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = i * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    ga[i] = p.A + (size_t)min(r, p.M - 1) * p.lda + c * 8;
    gb[i] = p.B + (size_t)min(col0 + r, p.N - 1) * p.ldb + c * 8;
  }
  auto stage = [&](int buf, int kt) {
    char* base = my + buf * WAVE_STAGE;
    const int koff = kt * BK;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((glb_void*)(ga[i] + koff), (lds_void*)(base + i * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((glb_void*)(gb[i] + koff), (lds_void*)(base + TILE_BYTES + i * 1024), 16, 0, 0);
  };
  const int sw = (lane >> 1) & 7;
  int roff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) roff[kk] = (lane & 15) * 128 + (((kk * 4 + (lane >> 4)) ^ sw) << 4);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int kt = kbeg + wave;
  int buf = 0;
  if (kt < kend) stage(0, kt);
  for (; kt < kend; kt += 4) {
    const bool more = kt + 4 < kend;
    if (more) {
      stage(buf ^ 1, kt + 4);
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // current tile landed, next stays in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const char* sa = my + buf * WAVE_STAGE;
    const char* sb = sa + TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *(const bf16x8*)(sa + i * 16 * 128 + roff[kk]);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = *(const bf16x8*)(sb + j * 16 * 128 + roff[kk]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);
    }
    // the LDS reads above must have returned before this buffer is DMA'd again by the next iteration's stage()
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    buf ^= 1;
  }
  // ---- sum the 4 waves' partials through LDS (each wave's private region is free again: 16 KiB of fp32 per wave)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float* red = (float*)(smem + wave * WAVE_LDS);              // [64 m][64 n] fp32, lane-linear chunks
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) *(f32x4*)(red + ((i * 4 + j) * 64 + lane) * 4) = acc[i][j];
  __syncthreads();
  // wave w finalises the (i = w) row block: acc[i][j] summed over the 4 regions
  f32x4 sum[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2)
      sum[j] += *(const f32x4*)((const float*)(smem + w2 * WAVE_LDS) + ((wave * 4 + j) * 64 + lane) * 4);
  }
  const int m = wave * 16 + (lane & 15);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = col0 + j * 16 + (lane >> 4) * 4;
    if (p.ksplit == 1) {
      store_out(p, m, n, sum[j]);
    } else if (n < p.ldn) {
      *(f32x4*)(p.slab + ((size_t)ks * BM + m) * p.ldn + n) = sum[j];
    }
  }
}

// out = epilogue(sum over splits of slab)      grid = ceil(M * ldn/4 / 256)
__global__ __launch_bounds__(256) void skinny_reduce_kernel(Args p) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int n4 = p.ldn / 4;
  if (idx >= p.M * n4) return;
  const int m = idx / n4, n = (idx - m * n4) * 4;
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < p.ksplit; ++k) s += *(const f32x4*)(p.slab + ((size_t)k * BM + m) * p.ldn + n);
  store_out(p, m, n, s);
}

}  // namespace tasu_skinny

// workspace: ksplit * 64 * round_up(N, 64) floats.  Returns TASU_ERR_ARG when the shape is not a skinny one.
static int tasu_gemm_skinny_dispatch(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* bias,
                              const float* resid, int M, int N, int K, int out_mode, float* ws, size_t ws_floats,
                              hipStream_t st) {
  using namespace tasu_skinny;
  if (M > BM) return TASU_ERR_ARG;
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
  a.ldn = (N + 63) / 64 * 64;
  a.out_mode = out_mode;
  const int tiles = (N + BN - 1) / BN, nk = K / BK;
  int ks = 1;
  if (tiles < 256) {
    ks = (320 + tiles - 1) / tiles;            // aim at >= 320 blocks ...
    if (ks > nk / 4) ks = nk / 4 > 0 ? nk / 4 : 1;   // ... but keep >= 1 K-step per wave
    if (ks < 1) ks = 1;
  }
  while (ks > 1 && (size_t)ks * BM * a.ldn > ws_floats) --ks;
  if (ks > 1 && !ws) ks = 1;
  a.ksplit = ks;
  a.slab = ws;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * WAVE_LDS);
    attr_set = true;
  }
  TASU_LAUNCH(gemm_skinny_kernel, dim3(tiles, ks), dim3(256), 4 * WAVE_LDS, st, a);
  if (ks > 1) TASU_LAUNCH(skinny_reduce_kernel, dim3((M * (a.ldn / 4) + 255) / 256), dim3(256), 0, st, a);
  return TASU_OK;
}

extern "C" int tasu_gemm_skinny_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* bias,
                                     const float* resid, int M, int N, int K, int out_mode, float* workspace,
                                     int64_t workspace_floats, void* stream) {
  if (!A || !B || !C || M <= 0 || M > 64 || N <= 0 || K <= 0 || K % 64 || lda % 8 || ldb % 8) return TASU_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return TASU_ERR_ARG;
  if (out_mode < 0 || out_mode > 2 || (out_mode == TASU_GEMM_OUT_F32_RESID_BF16R && !resid)) return TASU_ERR_ARG;
  return tasu_gemm_skinny_dispatch(A, lda, B, ldb, C, ldc, bias, resid, M, N, K, out_mode, workspace,
                                   workspace ? (size_t)workspace_floats : 0, (hipStream_t)stream);
}
