// C[M, N] = A[M, K] . B[N, K]^T for N <= 64: the rank-sized GEMMs of the LoRA recipe (ps_slm_amd/lora.py) --
//   u  = xd A^T          [rows, r]     K = in        (forward)
//   du = dy (sB)         [rows, r]     K = out       (backward, through lora_B)
//   dB = dy^T u          [out, r]      K = rows      (weight gradient, fp32: rank_gemm_tn_kernel below, dy K-major as it lies)
//   dA = (xd^T du)^T     [r, in]       K = rows      (weight gradient, fp32, stored transposed; xd K-major)
// On the tile policy of gemm.hip these are ONE column of 128-row tiles: 32 of 256 CUs walk the whole K range at one
// latency-bound K-step (~0.7 us) after the other -- 16-77 us.  What was tried first (measured on MI355X, kept here as the reason
// for this design): (a) fragments straight from global memory, 16 waves splitting K: adjacent lanes hold different ROWS of an
// MFMA operand, so every 64-lane load is 64 separate 16-byte accesses -- bound by the L1's line rate, 11-50 us; (b) the 64 x 64
// tiles of gemm.hip with its global split-K: every splitting workgroup pays an agent-scope release fence (the XCDs' L2s are not
// coherent), 47-93 us.
// This kernel: 16 rows x 64 columns per workgroup, EIGHT waves that split K (64-wide chunks round-robin), each wave staging its
// own chunk -- 2 KiB of A, 8 KiB of B -- into its private 10-KiB LDS region with LDS-DMA (whole 128-byte lines, no barriers: the
// region is wave-private) and reading MFMA fragments back with the XOR swizzle of gemm.hip; two workgroups per CU = sixteen
// chunks in flight per CU, which is what hides the load latency.  Partial tiles meet in LDS and are summed in wave order
// (deterministic, no global traffic).
#include "common.h"
#include "../../include/tasu_hip.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int NW = 8;
constexpr int REGION = 10240;                           // per wave: A 16 x 128 B, then B 64 x 128 B

template <bool F32OUT, bool TSTORE>
__device__ __forceinline__ void rank_gemm_body(const bf16* __restrict__ A, int lda, const bf16* __restrict__ B, int ldb,
                                               void* __restrict__ Cv, int ldc, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m0 = blockIdx.x * 16;
  const int ntn = (N + 15) >> 4;
  char* mine = smem + wave * REGION;
  // per-lane global sources of the 1-KiB pieces (8 rows x 128 B): lane -> row pc * 8 + (lane >> 3), LDS chunk lane & 7, which
  // must hold global chunk (lane & 7) ^ ((row >> 1) & 7)  (the LDS image of a DMA is lane-linear: the swizzle sits on the source)
  const bf16* ga[2];
  const bf16* gb[8];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = i * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
    ga[i] = A + (size_t)min(m0 + r, M - 1) * lda + c * 8;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = i * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
    gb[i] = B + (size_t)min(r, N - 1) * ldb + c * 8;
  }
  const int sw = (lane >> 1) & 7;
  int roff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) roff[kk] = (lane & 15) * 128 + (((kk * 4 + (lane >> 4)) ^ sw) << 4);
  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nchunks = K >> 6;
  for (int c = wave; c < nchunks; c += NW) {
    const int koff = c << 6;
#pragma unroll
    for (int i = 0; i < 2; ++i) __builtin_amdgcn_global_load_lds((glb_void*)(ga[i] + koff), (lds_void*)(mine + i * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i * 8 < N) __builtin_amdgcn_global_load_lds((glb_void*)(gb[i] + koff), (lds_void*)(mine + 2048 + i * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bf16x8 fa[2], fb[2][4];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      fa[kk] = *(const bf16x8*)(mine + roff[kk]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < ntn) fb[kk][j] = *(const bf16x8*)(mine + 2048 + j * 2048 + roff[kk]);
    }
    // the weight-side fragment goes in as the MFMA's A operand: acc[j][r] = C[m0 + (lane & 15)][j * 16 + (lane >> 4) * 4 + r]
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < ntn) acc[j] = mfma16(fb[kk][j], fa[kk], acc[j]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this chunk's fragment reads are done before the next DMA lands on them
  }
  // partial tiles -> LDS (own region, free now), summed by waves 0..3 (one 16-column group each) in wave order
  f32x4* part = (f32x4*)mine;
#pragma unroll
  for (int j = 0; j < 4; ++j) part[j * 64 + lane] = acc[j];
  __syncthreads();
  const int j = wave;
  if (j >= 4 || j >= ntn) return;
  f32x4 s = ((const f32x4*)smem)[j * 64 + lane];
#pragma unroll
  for (int w = 1; w < NW; ++w) {
    const f32x4 v = ((const f32x4*)(smem + w * REGION))[j * 64 + lane];
    s[0] += v[0], s[1] += v[1], s[2] += v[2], s[3] += v[3];
  }
  const int m = m0 + (lane & 15), n0 = j * 16 + (lane >> 4) * 4;
  if (m >= M) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n0 + r;
    if (n >= N) break;
    const size_t off = TSTORE ? (size_t)n * ldc + m : (size_t)m * ldc + n;
    if constexpr (F32OUT) ((float*)Cv)[off] = s[r];
    else ((bf16*)Cv)[off] = (bf16)s[r];
  }
}

template <bool F32OUT, bool TSTORE>
__global__ __launch_bounds__(64 * NW, 2) void rank_gemm_kernel(const bf16* __restrict__ A, int lda, const bf16* __restrict__ B, int ldb,
                                                               void* __restrict__ Cv, int ldc, int M, int N, int K) {
  rank_gemm_body<F32OUT, TSTORE>(A, lda, B, ldb, Cv, ldc, M, N, K);
}
// the members of an adapted group in one launch (grid y = member): same M, N and output leading dimension, each member its own
// operands and K -- bf16 out, the arithmetic of rank_gemm_kernel<false, false> per member
constexpr int RANK_GROUP_MAX = 4;
struct RankGroup {
  const bf16* A[RANK_GROUP_MAX];
  const bf16* B[RANK_GROUP_MAX];
  bf16* C[RANK_GROUP_MAX];
  int lda[RANK_GROUP_MAX], ldb[RANK_GROUP_MAX], K[RANK_GROUP_MAX];
};
__global__ __launch_bounds__(64 * NW, 2) void rank_gemm_group_kernel(RankGroup g, int ldc, int M, int N) {
  const int t = blockIdx.y;
  rank_gemm_body<false, false>(g.A[t], g.lda[t], g.B[t], g.ldb[t], g.C[t], ldc, M, N, g.K[t]);
}

template <bool F32OUT, bool TSTORE>
int launch(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, hipStream_t st) {
  constexpr int LDS = NW * REGION;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)rank_gemm_kernel<F32OUT, TSTORE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  TASU_LAUNCH((rank_gemm_kernel<F32OUT, TSTORE>), dim3((M + 15) / 16), dim3(64 * NW), LDS, st, (const bf16*)A, lda, (const bf16*)B, ldb, C,
              ldc, M, N, K);
  return TASU_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------
// The weight gradients' form: C[M, N] = At[K, M]^T . B[N, K]^T with the big operand K-MAJOR (dy [rows, out] / xd [rows, in] as the
// step leaves them; no transposed copy).  64 x 64 tiles (a k-row of the tile is one whole 128-byte line -- the first version took
// 16 columns = 32-byte pieces of a line per k-row and ran at a third of the speed), eight waves split K as above; a wave stages
// its chunk's 64 k-rows of At and 64 rows of B into its private 16-KiB region with LDS-DMA and reads At back as MFMA operands
// through the hardware transpose read (ds_read_tr16_b64: a 16-lane group reads 4 k-rows x 16 columns, every lane receives one
// column's 4 k-values) -- k-slots {4q..4q+3, 16+4q..16+4q+3} of a 32-block for lane group q, so B's fragment is read as the two
// matching 8-byte halves.  16-byte chunk c of k-row r sits at chunk c ^ (r & 7) (swizzle on the DMA's source side).
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
constexpr int TN_REGION = 16384;                        // per wave: At 64 k-rows x 128 B, then B 64 x 128 B

template <bool TSTORE>
__global__ __launch_bounds__(64 * NW, 1) void rank_gemm_tn_kernel(const bf16* __restrict__ At, int ldat, const bf16* __restrict__ B, int ldb,
                                                                  float* __restrict__ C, int ldc, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m0 = blockIdx.x * 64;                       // M % 64 == 0
  const int ntn = (N + 15) >> 4;
  char* mine = smem + wave * TN_REGION;
  const bf16* ga[8];
  const bf16* gb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int kr = i * 8 + (lane >> 3), c = (lane & 7) ^ (kr & 7);
    ga[i] = At + (size_t)kr * ldat + m0 + c * 8;
    const int r = i * 8 + (lane >> 3), cb = (lane & 7) ^ ((r >> 1) & 7);
    gb[i] = B + (size_t)min(r, N - 1) * ldb + cb * 8;
  }
  const int l15 = lane & 15, q = lane >> 4, pp = lane & 3;
  const int sw = (lane >> 1) & 7;                       // B rows: the swizzle of the kernel above
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nchunks = K >> 6;
  for (int c = wave; c < nchunks; c += NW) {
    const size_t koff = (size_t)c << 6;
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_amdgcn_global_load_lds((glb_void*)(ga[i] + koff * ldat), (lds_void*)(mine + i * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i * 8 < N) __builtin_amdgcn_global_load_lds((glb_void*)(gb[i] + koff), (lds_void*)(mine + 8192 + i * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[4], fb[4];
      const int r0 = kk * 32 + 4 * q + ((lane >> 2) & 3), r1 = r0 + 16;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ch = i * 2 + (pp >> 1), inner = (pp & 1) * 8;
        union { s16x4 h[2]; bf16x8 b; } u;
        u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(mine + r0 * 128 + ((ch ^ (r0 & 7)) << 4) + inner));
        u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(mine + r1 * 128 + ((ch ^ (r1 & 7)) << 4) + inner));
        fa[i] = u.b;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < ntn) {
          const char* row = mine + 8192 + j * 2048 + l15 * 128;
          union { bf16x4 h[2]; bf16x8 b; } v;
          v.h[0] = *(const bf16x4*)(row + (((kk * 4 + (q >> 1)) ^ sw) << 4) + (q & 1) * 8);
          v.h[1] = *(const bf16x4*)(row + (((kk * 4 + 2 + (q >> 1)) ^ sw) << 4) + (q & 1) * 8);
          fb[j] = v.b;
        }
      // acc[i][j][r] = C[m0 + i * 16 + (lane & 15)][j * 16 + (lane >> 4) * 4 + r]
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < ntn) acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this chunk's fragment reads are done before the next DMA lands on them
  }
  // partial tiles -> LDS (own region), summed in wave order: wave w finishes fragments 2 w and 2 w + 1
  f32x4* part = (f32x4*)mine;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) part[(i * 4 + j) * 64 + lane] = acc[i][j];
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int f = wave * 2 + h, i = f >> 2, j = f & 3;
    if (j >= ntn) continue;
    f32x4 s = ((const f32x4*)smem)[f * 64 + lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const f32x4 v = ((const f32x4*)(smem + w * TN_REGION))[f * 64 + lane];
      s[0] += v[0], s[1] += v[1], s[2] += v[2], s[3] += v[3];
    }
    const int m = m0 + i * 16 + l15, n0 = j * 16 + q * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + r;
      if (n >= N) break;
      C[TSTORE ? (size_t)n * ldc + m : (size_t)m * ldc + n] = s[r];
    }
  }
}

template <bool TSTORE>
int launch_tn(const void* At, int ldat, const void* B, int ldb, float* C, int ldc, int M, int N, int K, hipStream_t st) {
  constexpr int LDS = NW * TN_REGION;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)rank_gemm_tn_kernel<TSTORE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  TASU_LAUNCH((rank_gemm_tn_kernel<TSTORE>), dim3(M / 64), dim3(64 * NW), LDS, st, (const bf16*)At, ldat, (const bf16*)B, ldb, C, ldc, M, N, K);
  return TASU_OK;
}

}  // namespace

extern "C" int tasu_gemm_nt_rank(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, int out_f32,
                                 int transposed, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || N > 64 || K <= 0 || K % 64 || lda % 8 || ldb % 8 || lda < K || ldb < K) return TASU_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return TASU_ERR_ARG;
  if (ldc < (transposed ? M : N)) return TASU_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (out_f32) return transposed ? launch<true, true>(A, lda, B, ldb, C, ldc, M, N, K, st) : launch<true, false>(A, lda, B, ldb, C, ldc, M, N, K, st);
  return transposed ? launch<false, true>(A, lda, B, ldb, C, ldc, M, N, K, st) : launch<false, false>(A, lda, B, ldb, C, ldc, M, N, K, st);
}

// tasu_gemm_nt_rank (bf16 out) for n_members problems that share M, N and ldc, in one launch
extern "C" int tasu_gemm_nt_rank_group(int n_members, const void* const* A, const int* lda, const void* const* B, const int* ldb, void* const* C,
                                       int ldc, int M, int N, const int* K, void* stream) {
  if (n_members < 1 || n_members > RANK_GROUP_MAX || !A || !lda || !B || !ldb || !C || !K || M <= 0 || N <= 0 || N > 64 || ldc < N)
    return TASU_ERR_ARG;
  RankGroup g{};
  for (int t = 0; t < n_members; ++t) {
    if (!A[t] || !B[t] || !C[t] || K[t] <= 0 || K[t] % 64 || lda[t] % 8 || ldb[t] % 8 || lda[t] < K[t] || ldb[t] < K[t] ||
        (((uintptr_t)A[t] | (uintptr_t)B[t]) & 15))
      return TASU_ERR_ARG;
    g.A[t] = (const bf16*)A[t], g.B[t] = (const bf16*)B[t], g.C[t] = (bf16*)C[t], g.lda[t] = lda[t], g.ldb[t] = ldb[t], g.K[t] = K[t];
  }
  constexpr int LDS = NW * REGION;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)rank_gemm_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  TASU_LAUNCH(rank_gemm_group_kernel, dim3((M + 15) / 16, n_members), dim3(64 * NW), LDS, (hipStream_t)stream, g, ldc, M, N);
  return TASU_OK;
}

// C[M, N] = At[K, M]^T . B[N, K]^T, A given K-major (include/tasu_hip.h): the adapters' weight gradients from the row-major dy / xd.
extern "C" int tasu_gemm_tn_rank(const void* At, int ldat, const void* B, int ldb, float* C, int ldc, int M, int N, int K,
                                 int transposed, void* stream) {
  if (!At || !B || !C || M <= 0 || M % 64 || N <= 0 || N > 64 || K <= 0 || K % 64 || ldat % 8 || ldb % 8 || ldat < M || ldb < K)
    return TASU_ERR_ARG;
  if (((uintptr_t)At & 15) || ((uintptr_t)B & 15)) return TASU_ERR_ARG;
  if (ldc < (transposed ? M : N)) return TASU_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  return transposed ? launch_tn<true>(At, ldat, B, ldb, C, ldc, M, N, K, st) : launch_tn<false>(At, ldat, B, ldb, C, ldc, M, N, K, st);
}
