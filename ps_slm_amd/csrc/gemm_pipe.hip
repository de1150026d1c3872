// Software-pipelined bf16 NT GEMM for gfx950: 256 x BN block tile (BN = 128 or 96), BK = 64, 4 waves (2 x 2), wave tile
// 128 x BN/2, ONE wave per SIMD (1 block per CU), 3-stage LDS ring filled by global_load_lds_dwordx4.
//
// Why a second kernel next to gemm.hip: with 64x64 wave tiles the 128x128 kernel spends an LDS read for every two
// MFMAs and hides latency only through the second wave of each SIMD.  Here a wave owns 128 x 64 (2.7 MFMAs per read,
// a third less LDS traffic per FLOP) and hides its OWN latencies: each K-step is two phases of 32 MFMAs, and every
// phase carries the 12 fragment reads of the NEXT phase and (first phase) the 12 LDS-DMA issues of the tile two steps
// ahead, interleaved between the MFMAs with sched_group_barrier.  One raw s_barrier per K-step sits between the two
// phases; tile t+1 is awaited with a COUNTED s_waitcnt vmcnt(12) that leaves tile t+2 in flight (a __syncthreads would
// drain it).  Ring safety: buffer (t+2)%3 == (t-1)%3 is refilled only after barrier(t-1), before which every wave
// waited for its last reads of tile t-1 (lgkmcnt(0)).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "../../include/tasu_hip.h"

namespace tasu_pipe {

constexpr int BM = 256, BK = 64;

struct Args {
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

//
// LOADERS: the block gets four more waves (4..7) that do nothing but issue the LDS-DMA of the ring and wait for it; the
// four MFMA waves then carry fragment reads and MFMAs only.  Measured on this kernel (8192^3, warm): MFMA + barrier alone
// 1.94 PFLOP/s, + fragment reads 1.60, + LDS-DMA issue 1.10 -- a global_load_lds costs its issuing wave ~100 cycles of
// address processing during which that wave's MFMA queue drains, so the issue is moved to waves that have none.
template <int BN, int OUT_MODE, bool HAS_BIAS, bool LOADERS>
__global__ __launch_bounds__(LOADERS ? 512 : 256, LOADERS ? 1 : 2) void gemm_pipe_kernel(Args p) {
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int WM = 128, WN = BN / 2, MI = WM / 16, NI = WN / 16;
  constexpr int PA = BM / 32, PB = BN / 32;       // LDS-DMA pieces per wave per tile: 8 + 4 (or 3)
  constexpr int NG = PA + PB;                      // vmcnt units per tile per wave
  constexpr int NR = MI + NI;                      // fragment reads per phase
  constexpr int NM = MI * NI;                      // MFMAs per phase
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave_id = threadIdx.x >> 6;
  const bool is_loader = LOADERS && wave_id >= 4;
  const int wave = wave_id & 3;                    // staging share (loader) / tile quadrant (MFMA wave)
  const int wr = wave >> 1, wc = wave & 1;

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  constexpr int GROUP_M = 4;
  const int per_group = GROUP_M * p.tiles_n;
  const int gid = logical / per_group;
  const int first_m = gid * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int in_g = logical - gid * per_group;
  const int tm = first_m + in_g % gsz, tn = in_g / gsz;
  const int row0 = tm * BM, col0 = tn * BN;

  const bf16* ga[PA];
  const bf16* gb[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int r = (wave * PA + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    ga[i] = p.A + (size_t)min(row0 + r, p.M - 1) * p.lda + c * 8;
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int r = (wave * PB + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    gb[i] = p.B + (size_t)min(col0 + r, p.N - 1) * p.ldb + c * 8;
  }
  // Loader waves address through buffer descriptors (base = the block's first A / B row, per-lane 32-bit byte offset
  // fixed for the whole K loop, the K offset in an SGPR): no per-K-step address VALU and a cheaper issue than the
  // 64-bit-pointer global form (buffer_load_dwordx4 ... offen lds).
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer builtins exist in the device pass only
  const __amdgpu_buffer_rsrc_t rsA =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (size_t)row0 * p.lda), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + (size_t)col0 * p.ldb), 0, 0x7fffffff, 0x00020000);
#endif
  int voa[PA], vob[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) voa[i] = (int)((ga[i] - (p.A + (size_t)row0 * p.lda)) * 2);
#pragma unroll
  for (int i = 0; i < PB; ++i) vob[i] = (int)((gb[i] - (p.B + (size_t)col0 * p.ldb)) * 2);
  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE;
    const int koff = kt * BK;
    if constexpr (LOADERS) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
      for (int i = 0; i < PA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(base + (wave * PA + i) * 1024), 16, voa[i], koff * 2, 0, 0);
#pragma unroll
      for (int i = 0; i < PB; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(base + A_BYTES + (wave * PB + i) * 1024), 16, vob[i],
                                                 koff * 2, 0, 0);
#endif
    } else {
#pragma unroll
      for (int i = 0; i < PA; ++i)
        __builtin_amdgcn_global_load_lds((glb_void*)(ga[i] + koff), (lds_void*)(base + (wave * PA + i) * 1024), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < PB; ++i)
        __builtin_amdgcn_global_load_lds((glb_void*)(gb[i] + koff), (lds_void*)(base + A_BYTES + (wave * PB + i) * 1024), 16,
                                         0, 0);
    }
  };

  const int nk = p.K / BK;
  if (LOADERS && is_loader) {
    stage(0, 0);
    if (nk > 1) {
      stage(1, 1);
      if (NG == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    int slot2 = 2;                                 // ring slot of tile t+2
    for (int t = 0; t < nk; ++t) {
      if (t + 2 < nk) {
        stage(slot2, t + 2);
        if (NG == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();                // tile t+1 is in LDS; every MFMA wave is done with tile t-1... (see below)
      slot2 = slot2 == 2 ? 0 : slot2 + 1;
    }
    return;
  }

  const int sw = (lane >> 1) & 7;
  int roff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) roff[kk] = (lane & 15) * 128 + (((kk * 4 + (lane >> 4)) ^ sw) << 4);
  const int a_base = (wr * WM) * 128, b_base = A_BYTES + (wc * WN) * 128;

  auto read_frags = [&](bf16x8 (&fa)[MI], bf16x8 (&fb)[NI], int buf, int kk) {
    const char* sa = smem + buf * STAGE + a_base + roff[kk];
    const char* sb = smem + buf * STAGE + b_base + roff[kk];
#pragma unroll
    for (int j = 0; j < NI; ++j) fb[j] = *(const bf16x8*)(sb + j * 16 * 128);
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[i] = *(const bf16x8*)(sa + i * 16 * 128);
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mma = [&](bf16x8 (&fa)[MI], bf16x8 (&fb)[NI]) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);
  };

  bf16x8 fa0[MI], fb0[NI], fa1[MI], fb1[NI];
  // ---- prologue: tiles 0 and 1 in flight, wait for tile 0, first fragments
  if (!LOADERS) {
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    if (nk > 1) {
      if (NG == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_frags(fa0, fb0, 0, 0);

  // One K-step.  MORE2: tile t+2 exists (its LDS-DMA is issued here); MORE1: tile t+1 exists (awaited + first reads).
  // Straight-line code (no branches inside) so that sched_group_barrier can interleave across the whole phase.
  auto kstep = [&](auto more2_tag, auto more1_tag, int t, int cur) {
    constexpr bool MORE2 = decltype(more2_tag)::value, MORE1 = decltype(more1_tag)::value;
    const int nxt = cur == 2 ? 0 : cur + 1;      // slot of tile t+1
    const int nx2 = nxt == 2 ? 0 : nxt + 1;      // slot of tile t+2 (== slot of tile t-1)
    // ---------------- phase 1: MFMA(t, k-half 0)  ||  reads (t, k-half 1)  ||  LDS-DMA of tile t+2
    if constexpr (MORE2 && !LOADERS) stage(nx2, t + 2);
    read_frags(fa1, fb1, cur, 1);
    mma(fa0, fb0);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                     // 2 MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                     // 1 DS read
      if constexpr (MORE2 && !LOADERS) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);  // 1 VMEM (LDS-DMA issue)
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NM - 2 * NR, 0);
    // ---------------- hand-over: my reads of tile t are done; tile t+1 has landed (tile t+2 may stay in flight)
    if constexpr (MORE1 && !LOADERS) {
      if constexpr (MORE2) {
        if constexpr (NG == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // ---------------- phase 2: MFMA(t, k-half 1)  ||  reads (t+1, k-half 0)
    if constexpr (MORE1) read_frags(fa0, fb0, nxt, 0);
    mma(fa1, fb1);
    if constexpr (MORE1) {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NM - 2 * NR, 1);
    }
    return nxt;
  };
  using T = std::true_type;
  using F = std::false_type;
  int cur = 0;                                   // ring slot of tile t
  int t = 0;
  for (; t + 2 < nk; ++t) cur = kstep(T{}, T{}, t, cur);
  if (t + 1 < nk) {
    cur = kstep(F{}, T{}, t, cur);
    ++t;
  }
  kstep(F{}, F{}, t, cur);

  // ---- epilogue: acc[i][j][r] = C[m][n], m = row0 + wr*128 + i*16 + (lane&15), n = col0 + wc*WN + j*16 + (lane>>4)*4 + r
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = row0 + wr * WM + i * 16 + (lane & 15);
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = col0 + wc * WN + j * 16 + (lane >> 4) * 4;
      if (n >= p.N) continue;
      f32x4 v = acc[i][j];
      if (HAS_BIAS) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < p.N) v[r] += (float)p.bias[n + r];
      }
      const size_t off = (size_t)m * p.ldc + n;
      const bool full = (n + 4 <= p.N) && ((off & 3) == 0);
      if (OUT_MODE == TASU_GEMM_OUT_BF16) {
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
      } else {
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

int loaders_enabled() {
  static const int v = [] {
    const char* e = getenv("TASU_PIPE_LOADERS");
    return e ? atoi(e) : 1;
  }();
  return v;
}

template <int BN, int OUT_MODE, bool HAS_BIAS, bool LOADERS>
int launch_v(Args a, hipStream_t st) {
  constexpr int LDS = 3 * (BM + BN) * BK * 2;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_pipe_kernel<BN, OUT_MODE, HAS_BIAS, LOADERS>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  TASU_LAUNCH((gemm_pipe_kernel<BN, OUT_MODE, HAS_BIAS, LOADERS>), dim3(a.tiles_m * a.tiles_n),
              dim3(LOADERS ? 512 : 256), LDS, st, a);
  return TASU_OK;
}

template <int BN, int OUT_MODE, bool HAS_BIAS>
int launch(Args a, hipStream_t st) {
  return loaders_enabled() ? launch_v<BN, OUT_MODE, HAS_BIAS, true>(a, st) : launch_v<BN, OUT_MODE, HAS_BIAS, false>(a, st);
}

template <int OUT_MODE, bool HAS_BIAS>
int launch_bn(const Args& a, int bn, hipStream_t st) {
  return bn == 96 ? launch<96, OUT_MODE, HAS_BIAS>(a, st) : launch<128, OUT_MODE, HAS_BIAS>(a, st);
}

}  // namespace tasu_pipe

// called from gemm.hip's dispatcher
int tasu_gemm_pipe_dispatch(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* bias,
                            const float* resid, int M, int N, int K, int out_mode, int bn, hipStream_t st) {
  using namespace tasu_pipe;
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
  a.tiles_m = a.tiles_n = 0;
  const bool hb = bias != nullptr;
  switch (out_mode) {
    case TASU_GEMM_OUT_BF16:
      return hb ? launch_bn<TASU_GEMM_OUT_BF16, true>(a, bn, st) : launch_bn<TASU_GEMM_OUT_BF16, false>(a, bn, st);
    case TASU_GEMM_OUT_F32:
      return hb ? launch_bn<TASU_GEMM_OUT_F32, true>(a, bn, st) : launch_bn<TASU_GEMM_OUT_F32, false>(a, bn, st);
    case TASU_GEMM_OUT_F32_RESID_BF16R:
      return hb ? launch_bn<TASU_GEMM_OUT_F32_RESID_BF16R, true>(a, bn, st)
                : launch_bn<TASU_GEMM_OUT_F32_RESID_BF16R, false>(a, bn, st);
    default:
      return TASU_ERR_ARG;
  }
}
