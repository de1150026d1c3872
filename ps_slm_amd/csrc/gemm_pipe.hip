// Persistent, software-pipelined bf16 NT GEMM for gfx950.
//
// One 512-thread workgroup per CU walks a static list of output tiles (256 x 128, 256 x 96 or 128 x 192; BK = 64):
//   * waves 0..3 ("MFMA waves", one per SIMD, 2 x 2 over the tile, wave tile BM/2 x BN/2) carry fragment reads and MFMAs
//     only.  Each K-step is two phases of 32 (24) MFMAs; every phase carries the 12 (11) ds_read_b128 of the NEXT phase,
//     interleaved with sched_group_barrier, so a wave hides its own LDS latency; one raw s_barrier per K-step sits
//     between the phases.
//   * waves 4..7 ("loader waves") do nothing but fill an NS-slot LDS ring NS - 1 K-steps ahead (NS = 3, ring_slots) with
//     buffer_load_dwordx4 ... lds (base = the tile's first A / B row in a buffer descriptor, per-lane 32-bit offset fixed
//     per tile, K offset in an SGPR) and wait for it with a counted vmcnt that leaves the newest slot in flight.
//     Why separate waves: an LDS-DMA costs its issuing wave ~100 cycles of address processing during which that wave's
//     MFMA queue drains.  Measured on the 4-wave predecessor of this kernel (8192^3, warm): MFMA + barrier alone
//     1.94 PFLOP/s, + fragment reads 1.60, + LDS-DMA issue 1.10; with loader waves 1.30, with descriptor addressing 1.37.
//   * the ring does not stop at tile boundaries: while the MFMA waves convert and store a finished tile, the first two
//     K-steps of the workgroup's next tile are already landing, and the stores drain under the next tile's MFMAs
//     (the MFMA waves never wait on vmcnt).  Measured against the same kernel launched one tile per workgroup: equal within
//     noise on the decoder shapes (the per-tile cost that remains, ~4 us of a 25-us K = 1536 tile, is the epilogue's own
//     issue time plus the write of C itself: time = 35 us + K * 0.115 us for 4096 x 17920, i.e. 1.27 PFLOP/s asymptotic).
//
// Ring safety: slot (q+NS-1)%NS == (q-1)%NS is refilled after barrier(q-1), before which every MFMA wave waited for its last
// reads of step q-1 (lgkmcnt(0)); step q+1 is complete in LDS before barrier(q) because every loader waited for its own
// share first (a counted vmcnt that leaves only the YOUNGER steps' pieces in flight: loads complete in order).  The LDS image is lane-linear per 1-KiB piece (8 rows x 128 B), so the bank swizzle (16-B chunk c of row r
// at chunk c ^ ((r>>1)&7)) is applied on the per-lane SOURCE offset and again on the ds_read_b128 address.
#include <stdlib.h>

#include "gemm_epilogue.h"

namespace tasu_pipe {

using namespace tasu_gemm;

constexpr int BK = 64;

// Slots of the LDS ring: the loader waves run NS - 1 K-steps ahead of the MFMA waves.  Three everywhere.  A fourth slot fits the
// CU's 160 KiB for the 128 x 192 tile (4 x 40 KiB) and was measured in round 6 (-DTASU_PIPE_NS4): one more K-step of lookahead for
// the cold weight panels changes nothing -- o 26.7 -> 27.0-27.9 us, down 106.8 -> 107.2 us (tools/lab_resid_gemms.py, 28 rotating
// weight sets), the step 28.27 / 28.32 / 28.34 -> 28.43 / 28.35 / 28.41 ms (alternating processes, one box).  So the K loop is
// not waiting for operands that a deeper ring would have brought earlier; the ring depth stays a parameter.
__host__ __device__ constexpr int ring_slots(int bm, int bn) {
#ifdef TASU_PIPE_NS4
  return 4 * (bm + bn) * BK * 2 <= 160 * 1024 ? 4 : 3;
#else
  return 3;
#endif
}

typedef __attribute__((address_space(3))) void lds_void;

template <int BM, int BN, int OUT_MODE, bool HAS_BIAS>
__global__ __launch_bounds__(512, 1) void gemm_pipe_kernel(Args p) {
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int NS = ring_slots(BM, BN);           // LDS ring slots (3; see ring_slots)
  constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 16, NI = WN / 16;
  constexpr int PA = BM / 32, PB = BN / 32;       // LDS-DMA pieces per loader wave per K-step: 8 + 4 (or 3)
  constexpr int NG = PA + PB;                      // vmcnt units per K-step per loader wave
  constexpr int NR = MI + NI;                      // fragment reads per phase
  constexpr int NM = MI * NI;                      // MFMAs per phase
  static_assert(NG >= 10 && NG <= 12, "the counted vmcnt immediates below cover 10, 11 or 12 pieces per K-step");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave_id = threadIdx.x >> 6;
  const int wave = wave_id & 3;                    // staging share (loader) / tile quadrant (MFMA wave)
  const int nk = p.K / BK / p.ksplit;              // K-steps per work item
  const int base_tiles = p.tiles_m * p.tiles_n;
  const int ntiles = base_tiles * p.ksplit;
  const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  [[maybe_unused]] const int total = my_tiles * nk;  // K-steps this workgroup walks, across all its tiles

  if (wave_id >= 4) {
    // ================================================================ loader waves
#if defined(__HIP_DEVICE_COMPILE__)                 // the buffer builtins exist in the device pass only
    __amdgpu_buffer_rsrc_t rsA, rsB;
    int voa[PA], vob[PB];
    int ld_tile = blockIdx.x, ld_k = 0;            // load cursor (runs two K-steps ahead of the MFMA waves)
    auto setup = [&](int s) {
      int tm, tn;
      const int ks = s / base_tiles;
      tile_coords(p, s - ks * base_tiles, base_tiles, tm, tn);
      const int row0 = tm * BM, col0 = p.n0 + tn * BN;
      const size_t kbase = (size_t)ks * nk * BK;    // first K element of this work item
      // both descriptors start 3 KiB below the tile: the per-lane offsets carry +3 KiB minus the immediate offset of their
      // piece (issue()), which keeps every register offset non-negative
      rsA = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)(p.A + (size_t)row0 * p.lda + kbase) - 3072), 0, 0x7fffffff, 0x00020000);
      // OUT_GU_SWIGLU: the tile's 128 weight rows are, per 64-row half (= one MFMA wave column), 32 gate rows and the 32 up
      // rows of the same output columns; the descriptor then starts at the weight matrix itself
      const int brow0 = OUT_MODE == OUT_GU_SWIGLU ? 0 : col0;
      rsB = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)(p.B + (size_t)brow0 * p.ldb + kbase) - 3072), 0, 0x7fffffff, 0x00020000);
      // piece pc = 8 tile rows x 128 B; lane l -> tile row pc*8 + (l>>3), LDS chunk l&7 <- global chunk (l&7)^((row>>1)&7)
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int r = (wave * PA + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        voa[i] = (min(row0 + r, p.M - 1) - row0) * p.lda * 2 + c * 16 + (3 - (i & 3)) * 1024;   // see issue(): immediate offset, base 3 KiB low
      }
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int r = (wave * PB + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        if (OUT_MODE == OUT_GU_SWIGLU) {
          const int half = r >> 6, rr = r & 63;                       // wave column, row inside it
          const int ocol = min(p.n0 + tn * 64 + half * 32 + (rr & 31), p.N - 1);   // output (act) column
          vob[i] = (ocol + (rr >= 32 ? p.N : 0)) * p.ldb * 2 + c * 16 + (3 - (i & 3)) * 1024;
        } else if (OUT_MODE == OUT_QKV_ROPE) {
          // one head per tile; wave column `half` gets head dims half*32 .. +31 and 64 + half*32 .. +31 (store_qkv_rope)
          const int half = r >> 6, rr = r & 63;
          const int d = half * 32 + (rr & 31) + (rr >= 32 ? 64 : 0);
          vob[i] = d * p.ldb * 2 + c * 16 + (3 - (i & 3)) * 1024;
        } else {
          vob[i] = (min(col0 + r, p.N - 1) - col0) * p.ldb * 2 + c * 16 + (3 - (i & 3)) * 1024;
        }
      }
    };
    auto issue = [&](int slot) {
      char* base = smem + slot * STAGE;
      const int koff = ld_k * (BK * 2);
      // One M0 (LDS base) per group of four pieces: the instruction's 12-bit immediate offset is added to the LDS address
      // AND to the global address, so piece i of a group uses offset (i & 3) KiB and a per-lane global offset that was
      // lowered by the same amount in setup().  Writing M0 per piece costs the issuing wave ~40 cycles more per DMA
      // (measured: gate|up 235 -> 227 us, 8192^3 1300 -> 1348 TFLOP/s).
      auto one_a = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(base + (wave * PA + (i & ~3)) * 1024), 16,
                                                 voa[i], koff, (i & 3) * 1024, 0);
      };
      auto one_b = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(base + A_BYTES + (wave * PB + (i & ~3)) * 1024), 16,
                                                 vob[i], koff, (i & 3) * 1024, 0);
      };
      [&]<int... I>(std::integer_sequence<int, I...>) { (one_a(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, PA>{});
      [&]<int... I>(std::integer_sequence<int, I...>) { (one_b(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, PB>{});
      if (++ld_k == nk) {
        ld_k = 0;
        ld_tile += gridDim.x;
        if (ld_tile < ntiles) setup(ld_tile);
      }
    };
    // "all but the newest KEEP K-steps' pieces have landed" (counted vmcnt: KEEP * NG pieces stay in flight)
    auto wait_keep = [&](auto keep_tag) {
      constexpr int N = decltype(keep_tag)::value * NG;
      if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
      else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
      else if constexpr (N == 22) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
      else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else static_assert(N == 0, "add the immediate");
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    // before barrier(q) step q + 1 must be complete; the steps behind it that exist (at most NS - 2) stay in flight
    auto wait_for_step = [&](int newest_issued, int needed) {
      const int keep = newest_issued - needed;       // wave-uniform
      if (NS == 4 && keep >= 2) wait_keep(K2{});
      else if (keep >= 1) wait_keep(K1{});
      else wait_keep(K0{});
    };
    setup(ld_tile);
    int issued = 0;                                  // K-steps issued so far (the stream position of the next issue)
    for (; issued < NS - 1 && issued < total; ++issued) issue(issued);
    wait_for_step(issued - 1, 0);
    __builtin_amdgcn_s_barrier();
    int slot_new = NS - 1;                           // ring slot of step q + NS - 1
    for (int q = 0; q < total; ++q) {
      if (issued < total) {
        issue(slot_new);
        ++issued;
      }
      if (q + 1 < total) wait_for_step(issued - 1, q + 1);
      else wait_keep(K0{});
      __builtin_amdgcn_s_barrier();                // step q+1 is in LDS; slot (q-1) % NS is free (see "Ring safety" above)
      slot_new = slot_new == NS - 1 ? 0 : slot_new + 1;
    }
#endif
    return;
  }

  // ================================================================== MFMA waves
  const int wr = wave >> 1, wc = wave & 1;
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
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto mma = [&](bf16x8 (&fa)[MI], bf16x8 (&fb)[NI]) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);
  };

  bf16x8 fa0[MI], fb0[NI], fa1[MI], fb1[NI];
  zero_acc();
  __builtin_amdgcn_s_barrier();                    // step 0 is in LDS
  asm volatile("" ::: "memory");
  read_frags(fa0, fb0, 0, 0);

  // One K-step.  MORE: a next step exists (possibly the first of the next tile): its first fragments are read in phase 2.
  // Straight-line code (no branches inside) so that sched_group_barrier can interleave across the whole phase.
  auto kstep = [&](auto more_tag, int cur) {
    constexpr bool MORE = decltype(more_tag)::value;
    const int nxt = cur == NS - 1 ? 0 : cur + 1;
    // ---------------- phase 1: MFMA(q, k-half 0)  ||  reads (q, k-half 1)
    read_frags(fa1, fb1, cur, 1);
    mma(fa0, fb0);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                     // 2 MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                     // 1 DS read
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NM - 2 * NR, 0);
    // ---------------- hand-over: my reads of step q are done (slot reusable); step q+1 has landed (loaders waited)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // ---------------- phase 2: MFMA(q, k-half 1)  ||  reads (q+1, k-half 0)
    if constexpr (MORE) read_frags(fa0, fb0, nxt, 0);
    mma(fa1, fb1);
    if constexpr (MORE) {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NM - 2 * NR, 1);
    }
    return nxt;
  };

  // epilogues: gemm_epilogue.h (acc[i][j][r] = C[m][n], m = row0 + wr*WM + i*16 + (lane&15), n = col0 + wc*WN + j*16 + (lane>>4)*4 + r)
  auto store_gu = [&](int row0, int tn) { store_gu_swiglu<MI, NI, BM>(p, acc, row0, p.n0 + tn * 64 + wc * 32, wr * WM, lane); };
  auto store_c = [&](int row0, int col0) {
    store_tile<MI, NI, OUT_MODE, HAS_BIAS, BM, BN, BM == 128>(p, acc, row0, col0, wr * WM, wc * WN, lane);
  };

  using T = std::true_type;
  using F = std::false_type;
  int cur = 0;                                     // ring slot of the current step
  void* const c_first = p.C;
  for (int s = blockIdx.x; s < ntiles; s += gridDim.x) {
    int tm, tn;
    const int ks = s / base_tiles;
    tile_coords(p, s - ks * base_tiles, base_tiles, tm, tn);
    if constexpr (OUT_MODE == TASU_GEMM_OUT_F32) p.C = (float*)c_first + (size_t)ks * p.split_stride;   // slab of this K range
    for (int kt = 0; kt + 1 < nk; ++kt) cur = kstep(T{}, cur);
    cur = kstep(F{}, cur);                         // no read-ahead into the next tile: the fragment registers are free
    if constexpr (OUT_MODE == OUT_GU_SWIGLU) store_gu(tm * BM, tn);
    else if constexpr (OUT_MODE == OUT_QKV_ROPE) store_qkv_rope<MI, NI, BM>(p, acc, tm * BM, p.n0 + tn * BN, wr * WM, wc, lane);
    else store_c(tm * BM, p.n0 + tn * BN);      // for the epilogue, whose stores then drain under the next tile
    zero_acc();
    if (s + (int)gridDim.x < ntiles) read_frags(fa0, fb0, cur, 0);   // landed before the barrier of the step just done
  }
}

int cu_count() {
  static const int n = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    }
    return cus >= 8 ? (cus & ~7) : 8;              // a multiple of 8 keeps a workgroup's tiles on one XCD
  }();
  return n;
}

template <int BM, int BN, int OUT_MODE, bool HAS_BIAS>
int launch(Args a, hipStream_t st) {
  constexpr int LDS = ring_slots(BM, BN) * (BM + BN) * BK * 2;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_pipe_kernel<BM, BN, OUT_MODE, HAS_BIAS>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  a.tiles_m = (a.M + BM - 1) / BM;
  const int n_end = a.n1 > 0 ? a.n1 : a.N;
  a.tiles_n = OUT_MODE == OUT_GU_SWIGLU ? (n_end - a.n0 + 63) / 64 : (n_end - a.n0 + BN - 1) / BN;
  const int ntiles = a.tiles_m * a.tiles_n * a.ksplit;
  const int grid = ntiles < cu_count() ? ntiles : cu_count();
  ++tasu_gemm::gemm_launches();
  TASU_LAUNCH((gemm_pipe_kernel<BM, BN, OUT_MODE, HAS_BIAS>), dim3(grid), dim3(512), LDS, st, a);
  return TASU_OK;
}

// tile = 256 x 128, 256 x 96, or (bn == 192) 128 x 192
template <int OUT_MODE, bool HAS_BIAS>
int launch_bn(const Args& a, int bn, hipStream_t st) {
  if (bn == 192) return launch<128, 192, OUT_MODE, HAS_BIAS>(a, st);
  return bn == 96 ? launch<256, 96, OUT_MODE, HAS_BIAS>(a, st) : launch<256, 128, OUT_MODE, HAS_BIAS>(a, st);
}

}  // namespace tasu_pipe

// called from gemm.hip's dispatcher
int tasu_gemm_pipe_dispatch(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* bias,
                            const float* resid, int M, int N, int K, int out_mode, int bn, hipStream_t st, int n0, int n1) {
  using namespace tasu_pipe;
  Args a;
  a.n0 = n0;
  a.n1 = n1;
  a.A = (const bf16*)A;
  a.B = (const bf16*)B;
  a.C = C;
  a.R = resid;
  a.bias = (const bf16*)bias;
  a.relu = out_mode == TASU_GEMM_OUT_BF16 ? tasu_gemm::relu_next() : 0;
  a.M = M;
  a.N = N;
  a.K = K;
  a.lda = lda;
  a.ldb = ldb;
  a.ldc = ldc;
  a.tiles_m = a.tiles_n = 0;
  a.act = nullptr;
  a.ksplit = 1;
  a.split_stride = 0;
  const bool hb = bias != nullptr;
  switch (out_mode) {
    case TASU_GEMM_OUT_BF16:
      return hb ? launch_bn<TASU_GEMM_OUT_BF16, true>(a, bn, st) : launch_bn<TASU_GEMM_OUT_BF16, false>(a, bn, st);
    case TASU_GEMM_OUT_F32:
      return hb ? launch_bn<TASU_GEMM_OUT_F32, true>(a, bn, st) : launch_bn<TASU_GEMM_OUT_F32, false>(a, bn, st);
    case TASU_GEMM_OUT_F32_RESID_BF16R:
      return hb ? launch_bn<TASU_GEMM_OUT_F32_RESID_BF16R, true>(a, bn, st)
                : launch_bn<TASU_GEMM_OUT_F32_RESID_BF16R, false>(a, bn, st);
#ifdef TASU_LAB
    case OUT_DSWIGLU:                                // `resid` = the saved gate|up matrix (bf16 [M, 2N]); C = dgu [M, 2N]
      if (hb || !resid || N % 8 || ldc != 2 * N || bn == 96) return TASU_ERR_ARG;
      a.act = (bf16*)resid;
      a.R = nullptr;
      return launch_bn<OUT_DSWIGLU, false>(a, bn, st);
#endif
    default:
      return TASU_ERR_ARG;
  }
}

int tasu_gemm_pp_gu_dispatch(const void* A, int lda, const void* Wgu, int ldw, void* gu, void* act, int M, int I, int K, hipStream_t st,
                             int n0, int n1, void* ws, size_t ws_bytes);
namespace tasu_pp {
double sk_max_rem();
}

// workspace (that of tasu_gemm_nt_bf16_ws) or nullptr: with it the 256 x 256 kernel may cut its last rounds along K (stream-K)
extern "C" int tasu_gemm_gate_up_swiglu_ws(const void* A, int lda, const void* Wgu, int ldw, void* gu, void* act, int M, int I,
                                           int K, void* workspace, int64_t workspace_bytes, void* stream) {
  using namespace tasu_pipe;
  if (((uintptr_t)workspace & 15) || workspace_bytes < 0) return TASU_ERR_ARG;
  void* const ws = workspace;
  const size_t ws_bytes = workspace ? (size_t)workspace_bytes : 0;
  if (!A || !Wgu || !gu || !act || M <= 0 || I <= 0 || I % 4 || K <= 0 || K % BK || lda % 8 || ldw % 8) return TASU_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)Wgu & 15) || ((uintptr_t)gu & 7) || ((uintptr_t)act & 7)) return TASU_ERR_ARG;
  static const int forced = [] {                  // TASU_GEMM_GU_KERNEL=pipe|pp: A/B runs and tests of either kernel
    const char* e = tasu_lab_env("TASU_GEMM_GU_KERNEL");
    return !e ? 0 : (e[0] == 'p' && e[1] == 'p' ? 2 : 1);
  }();
  auto pipe_range = [&](int n0) {                 // 256 x 128 tiles (64 act columns) of this file over act columns [n0, I)
    Args a;
    a.A = (const bf16*)A;
    a.B = (const bf16*)Wgu;
    a.C = gu;
    a.R = nullptr;
    a.bias = nullptr;
    a.M = M;
    a.N = I;
    a.K = K;
    a.lda = lda;
    a.ldb = ldw;
    a.ldc = 2 * I;
    a.tiles_m = a.tiles_n = 0;
    a.act = (bf16*)act;
    a.act_ld = tasu_gemm::act_ld_next();
    a.ksplit = 1;
    a.split_stride = 0;
    a.n0 = n0;
    return launch<256, 128, OUT_GU_SWIGLU, false>(a, (hipStream_t)stream);
  };
  const bool pp_ok = I % 128 == 0 && K >= 256 && K % 128 == 0;
  if (forced == 2 && pp_ok) return tasu_gemm_pp_gu_dispatch(A, lda, Wgu, ldw, gu, act, M, I, K, (hipStream_t)stream, 0, 0, ws, ws_bytes);
  if (forced == 0 && pp_ok) {
    // tile policy as in tasu_gemm_nt_bf16_ws (gemm.hip): 256 x 256 tiles (128 act columns, gemm_pp.hip) where their coarser
    // rounds cost less than the per-FLOP efficiency they bring (4096 x 17920 x 1536: 257 -> 218 us); and when the last round
    // of big tiles would be mostly empty (1120 tiles on 256 CUs: 4.375 rounds), whole rounds on the big tiles + the remaining
    // columns on the small ones in a second launch (4 rounds + 192 tiles of 256 x 128)
    static const bool pp_on = [] {
      const char* e = tasu_lab_env("TASU_GEMM_PP");
      return !(e && e[0] == '0');
    }();
    static const bool split_on = [] {
      const char* e = tasu_lab_env("TASU_GEMM_NSPLIT");
      return !(e && e[0] == '0');
    }();
    const long tm = (M + 255) / 256, cus = cu_count(), tn = (I + 127) / 128;
    auto rounds = [&](long tiles) { return (double)((tiles + cus - 1) / cus); };
    const double c128 = rounds(tm * ((I + 63) / 64)) * 0.5;
    // stream-K (workspace given): the big tiles fill fractional rounds, for the price of the partial tiles' round trip (~35 us)
    const bool sk = tasu_gemm::sk_plan(tm * tn, K / 128, (int)cus, ws_bytes >= TASU_GEMM_WS_COUNTERS * sizeof(int) + (size_t)cus * 262144,
                                       tasu_pp::sk_max_rem()) > 0;
    const double c256_whole = rounds(tm * tn) / 1.26;
    const double c256_sk = sk ? ((double)(tm * tn) / cus) / 1.26 + 1.0e8 / K / 52012.0 : 1e30;
    const double c256 = c256_sk < c256_whole ? c256_sk : c256_whole;
    if (pp_on && c256 < c128) {
      if (c256_sk < c256_whole)
        return tasu_gemm_pp_gu_dispatch(A, lda, Wgu, ldw, gu, act, M, I, K, (hipStream_t)stream, 0, 0, ws, ws_bytes);
      const long full = (tm * tn) / cus;                          // whole rounds of big tiles
      const long tn_main = full * cus / tm;                       // column tiles they cover
      if (split_on && full >= 1 && tn_main < tn && tn_main > 0) {
        const double c_split = (double)full / 1.26 + rounds(tm * (tn - tn_main) * 2) * 0.5 + 0.05;   // + the second launch's ramp
        if (c_split < c256) {
          const int rc = tasu_gemm_pp_gu_dispatch(A, lda, Wgu, ldw, gu, act, M, I, K, (hipStream_t)stream, 0, (int)tn_main * 128, nullptr, 0);
          return rc ? rc : pipe_range((int)tn_main * 128);
        }
      }
      return tasu_gemm_pp_gu_dispatch(A, lda, Wgu, ldw, gu, act, M, I, K, (hipStream_t)stream, 0, 0, nullptr, 0);
    }
  }
  return pipe_range(0);
}

// act with a leading dimension (the LoRA recipe keeps [act | rank activations] side by side as the down projection's operand)
extern "C" int tasu_gemm_gate_up_swiglu_ld(const void* A, int lda, const void* Wgu, int ldw, void* gu, void* act, int ld_act, int M,
                                           int I, int K, void* workspace, int64_t workspace_bytes, void* stream) {
  if (ld_act < I || ld_act % 8) return TASU_ERR_ARG;
  tasu_gemm::act_ld_next() = ld_act == I ? 0 : ld_act;
  const int rc = tasu_gemm_gate_up_swiglu_ws(A, lda, Wgu, ldw, gu, act, M, I, K, workspace, workspace_bytes, stream);
  tasu_gemm::act_ld_next() = 0;
  return rc;
}

extern "C" int tasu_gemm_gate_up_swiglu(const void* A, int lda, const void* Wgu, int ldw, void* gu, void* act, int M, int I,
                                        int K, void* stream) {
  return tasu_gemm_gate_up_swiglu_ws(A, lda, Wgu, ldw, gu, act, M, I, K, nullptr, 0, stream);
}

// q|k|v projection + bias + rotary embedding of the q and k heads in one launch (include/tasu_hip.h): 256 x 128 tiles, one
// head per tile column.  TASU_GEMM_QKV_ROPE=0 (read per call) selects the two-kernel form.
extern "C" int tasu_gemm_nt_bf16_ws(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* bias,
                                    const float* resid, int M, int N, int K, int out_mode, void* workspace,
                                    int64_t workspace_bytes, void* stream);
extern "C" int tasu_rope_fwd(void* qkv, const float* cos_tab, const float* sin_tab, void* qt, void* kt, void* vt, int B, int S, int H,
                             int G, void* stream);
extern "C" int tasu_gemm_qkv_rope(const void* A, int lda, const void* Wqkv, int ldw, const void* bias, void* qkv, const float* cos_tab,
                                  const float* sin_tab, int M, int H, int G, int K, void* workspace, int64_t workspace_bytes,
                                  void* stream) {
  using namespace tasu_pipe;
  if (!A || !Wqkv || !qkv || !cos_tab || !sin_tab || M <= 0 || H <= 0 || G <= 0 || K <= 0 || K % BK || lda % 8 || ldw % 8)
    return TASU_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)Wqkv & 15) || ((uintptr_t)qkv & 15) || ((uintptr_t)cos_tab & 15) || ((uintptr_t)sin_tab & 15))
    return TASU_ERR_ARG;
  const int N = (H + 2 * G) * 128;
  const char* const e = tasu_lab_env("TASU_GEMM_QKV_ROPE");
  if (e && e[0] == '0') {
    const int rc = tasu_gemm_nt_bf16_ws(A, lda, Wqkv, ldw, qkv, N, bias, nullptr, M, N, K, TASU_GEMM_OUT_BF16, workspace, workspace_bytes,
                                        stream);
    return rc ? rc : tasu_rope_fwd(qkv, cos_tab, sin_tab, nullptr, nullptr, nullptr, 1, M, H, G, stream);
  }
  Args a;
  a.A = (const bf16*)A;
  a.B = (const bf16*)Wqkv;
  a.C = qkv;
  a.R = cos_tab;
  a.bias = (const bf16*)bias;
  a.M = M;
  a.N = N;
  a.K = K;
  a.lda = lda;
  a.ldb = ldw;
  a.ldc = N;
  a.tiles_m = a.tiles_n = 0;
  a.act = (bf16*)sin_tab;
  a.ksplit = 1;
  a.split_stride = (long long)(H + G) * 128;       // first column that is not rotated (the v heads)
  return launch<256, 128, OUT_QKV_ROPE, false>(a, (hipStream_t)stream);
}

// ---- split-K form for grids that would leave most CUs idle behind a very long K (the lm_head dgrad over the labelled rows
// of a batch: 2048 x 1536 outputs, K = 151,936): ksplit fp32 partial matrices, summed by tasu_sum_slabs_bf16.
extern "C" int tasu_gemm_nt_bf16_splitk(const void* A, int lda, const void* B, int ldb, float* partials, int ldc, int M, int N,
                                        int K, int ksplit, void* stream) {
  using namespace tasu_pipe;
  if (!A || !B || !partials || M <= 0 || N <= 0 || K <= 0 || ksplit < 1 || ksplit > 16 || K % (BK * ksplit) || lda % 8 || ldb % 8 ||
      ldc < N)
    return TASU_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)partials & 15)) return TASU_ERR_ARG;
  if ((((size_t)K / ksplit) * 2) % 16) return TASU_ERR_ARG;     // every K range starts 16-byte aligned
  Args a;
  a.A = (const bf16*)A;
  a.B = (const bf16*)B;
  a.C = partials;
  a.R = nullptr;
  a.bias = nullptr;
  a.M = M;
  a.N = N;
  a.K = K;
  a.lda = lda;
  a.ldb = ldb;
  a.ldc = ldc;
  a.tiles_m = a.tiles_n = 0;
  a.act = nullptr;
  a.ksplit = ksplit;
  a.split_stride = (long long)M * ldc;
  return launch<128, 192, TASU_GEMM_OUT_F32, false>(a, (hipStream_t)stream);
}
