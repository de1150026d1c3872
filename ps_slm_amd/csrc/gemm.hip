// bf16 "NT" GEMM for gfx950:  C[M,N] (+)= A[M,K] . B[N,K]^T (+ bias[N])
//
// Both operands are K-contiguous (A = activations, token-major; B = an nn.Linear weight [out,in] or its
// pre-transposed copy for dgrad), so the SAME kernel serves forward, dgrad (against the resident W^T copy)
// and wgrad (against transposed activations).  Replaces every torch.nn.functional.linear on the reference
// hot path (transformers modeling_qwen2.py q/k/v/o/gate/up/down/lm_head; Multitask/model/projector.py:141-143).
//
// Structure: 128x128 tile, BK = 64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 tiles.
// A/B tiles go global -> LDS with global_load_lds_dwordx4 (no VGPR round trip), double-buffered, one barrier
// per K-step.  The LDS image is lane-linear (a glds constraint), so the bank-conflict swizzle is applied on
// the per-lane SOURCE address and again on the ds_read_b128 address (16-B chunk c of row r lives at chunk
// c ^ ((r>>1)&7): conflict-free for the 16x16x32 operand read).  The MFMA is issued with the weight fragment
// as the A operand so that every lane ends up with 4 CONSECUTIVE output columns of one row (8-/16-byte
// stores).  blockIdx is remapped so that each XCD (private L2) works on a contiguous band of tiles.
#include "common.h"
#include "../../include/tasu_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int A_BYTES = BM * BK * 2;  // 16 KiB
constexpr int B_BYTES = BN * BK * 2;  // 16 KiB
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;

struct GemmArgs {
  const bf16* A;
  const bf16* B;
  void* C;
  const float* R;
  const bf16* bias;
  int M, N, K;
  int lda, ldb, ldc;
  int tiles_m, tiles_n;
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <int OUT_MODE, bool HAS_BIAS>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  // ---- XCD-aware, bijective block -> tile map, then an 8-row-group raster for L2 reuse of the B panel.
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * p.tiles_n;
  const int gid = logical / per_group;
  const int first_m = gid * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int in_g = logical - gid * per_group;
  const int tm = first_m + in_g % gsz;
  const int tn = in_g / gsz;
  const int row0 = tm * BM, col0 = tn * BN;

  // ---- per-lane global source pointers for the pieces this wave stages (4 of A, 4 of B per K-step).
  // piece pc = 8 tile rows x 128 B; lane l -> tile row pc*8 + (l>>3), LDS chunk l&7, which must hold
  // global chunk (l&7) ^ ((row>>1)&7).
  const bf16* ga[4];
  const bf16* gb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pc = wave * 4 + i;
    const int r = pc * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    const int ra = min(row0 + r, p.M - 1);
    const int rb = min(col0 + r, p.N - 1);
    ga[i] = p.A + (size_t)ra * p.lda + c * 8;
    gb[i] = p.B + (size_t)rb * p.ldb + c * 8;
  }

  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE_BYTES;
    const int koff = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pc = wave * 4 + i;
      __builtin_amdgcn_global_load_lds((glb_void*)(ga[i] + koff), (lds_void*)(base + pc * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pc = wave * 4 + i;
      __builtin_amdgcn_global_load_lds((glb_void*)(gb[i] + koff), (lds_void*)(base + A_BYTES + pc * 1024), 16, 0, 0);
    }
  };

  // ---- per-lane LDS read offsets (bytes) for the two 32-deep k sub-steps.
  const int sw = (lane >> 1) & 7;
  int roff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) roff[kk] = (lane & 15) * 128 + (((kk * 4 + (lane >> 4)) ^ sw) << 4);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  stage(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* sa = smem + cur * STAGE_BYTES + (wr * 64) * 128;
    const char* sb = smem + cur * STAGE_BYTES + A_BYTES + (wc * 64) * 128;
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
    __syncthreads();
  }

  // ---- epilogue.  acc[i][j][r] = C[m][n], m = row0 + wr*64 + i*16 + (lane&15),
  //                                        n = col0 + wc*64 + j*16 + (lane>>4)*4 + r.
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = row0 + wr * 64 + i * 16 + (lane & 15);
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = col0 + wc * 64 + j * 16 + (lane >> 4) * 4;
      if (n >= p.N) continue;
      f32x4 v = acc[i][j];
      if (HAS_BIAS) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < p.N) v[r] += (float)p.bias[n + r];
      }
      const size_t off = (size_t)m * p.ldc + n;
      const bool full = (n + 4 <= p.N);
      if (OUT_MODE == TASU_GEMM_OUT_BF16) {
        bf16* c = (bf16*)p.C + off;
        bf16x4 o = __builtin_convertvector(v, bf16x4);
        if (full && ((off & 3) == 0)) {
          *(bf16x4*)c = o;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) c[r] = o[r];
        }
      } else if (OUT_MODE == TASU_GEMM_OUT_F32) {
        float* c = (float*)p.C + off;
        if (full && ((off & 3) == 0)) {
          *(f32x4*)c = v;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) c[r] = v[r];
        }
      } else {  // TASU_GEMM_OUT_F32_RESID_BF16R: C(fp32) = R(fp32) + bf16_round(result)  (autocast residual add)
        float* c = (float*)p.C + off;
        const float* rs = p.R + off;
        f32x4 rr = __builtin_convertvector(__builtin_convertvector(v, bf16x4), f32x4);
        if (full && ((off & 3) == 0)) {
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

template <int OUT_MODE, bool HAS_BIAS>
int launch(const GemmArgs& a, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<OUT_MODE, HAS_BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                        2 * STAGE_BYTES);
    attr_set = true;
  }
  const int nwg = a.tiles_m * a.tiles_n;
  TASU_LAUNCH((gemm_nt_kernel<OUT_MODE, HAS_BIAS>), dim3(nwg), dim3(256), 2 * STAGE_BYTES, st, a);
  return TASU_OK;
}

}  // namespace

extern "C" int tasu_gemm_nt_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* bias,
                                 const float* resid, int M, int N, int K, int out_mode, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return TASU_ERR_ARG;
  if (K % BK != 0 || lda % 8 != 0 || ldb % 8 != 0) return TASU_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return TASU_ERR_ARG;
  if (out_mode == TASU_GEMM_OUT_F32_RESID_BF16R && !resid) return TASU_ERR_ARG;
  GemmArgs a;
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
  a.tiles_m = (M + BM - 1) / BM;
  a.tiles_n = (N + BN - 1) / BN;
  hipStream_t st = (hipStream_t)stream;
  const bool hb = bias != nullptr;
  switch (out_mode) {
    case TASU_GEMM_OUT_BF16:
      return hb ? launch<TASU_GEMM_OUT_BF16, true>(a, st) : launch<TASU_GEMM_OUT_BF16, false>(a, st);
    case TASU_GEMM_OUT_F32:
      return hb ? launch<TASU_GEMM_OUT_F32, true>(a, st) : launch<TASU_GEMM_OUT_F32, false>(a, st);
    case TASU_GEMM_OUT_F32_RESID_BF16R:
      return hb ? launch<TASU_GEMM_OUT_F32_RESID_BF16R, true>(a, st) : launch<TASU_GEMM_OUT_F32_RESID_BF16R, false>(a, st);
    default:
      return TASU_ERR_ARG;
  }
}
