// Persistent, software-pipelined bf16 NT GEMM for gfx950.
//
// One 512-thread workgroup per CU walks a static list of output tiles (256 x 128, 256 x 96 or 128 x 192; BK = 64):
//   * waves 0..3 ("MFMA waves", one per SIMD, 2 x 2 over the tile, wave tile BM/2 x BN/2) carry fragment reads and MFMAs
//     only.  Each K-step is two phases of 32 (24) MFMAs; every phase carries the 12 (11) ds_read_b128 of the NEXT phase,
//     interleaved with sched_group_barrier, so a wave hides its own LDS latency; one raw s_barrier per K-step sits
//     between the phases.
//   * waves 4..7 ("loader waves") do nothing but fill a 3-slot LDS ring two K-steps ahead with
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
// Ring safety: slot (q+2)%3 == (q-1)%3 is refilled after barrier(q-1), before which every MFMA wave waited for its last
// reads of step q-1 (lgkmcnt(0)); step q+1 is complete in LDS before barrier(q) because every loader waited for its own
// share first.  The LDS image is lane-linear per 1-KiB piece (8 rows x 128 B), so the bank swizzle (16-B chunk c of row r
// at chunk c ^ ((r>>1)&7)) is applied on the per-lane SOURCE offset and again on the ds_read_b128 address.
#include <type_traits>
#include <utility>

#include "common.h"
#include "../../include/tasu_hip.h"

#if defined(TASU_EXP_B_DIRECT)                      // prototype (tools/bench_gemm_bdirect.py): B = fragment-order weights, global -> registers
#define TASU_EXP_NO_B_READS
#define TASU_EXP_NO_B_DMA
#endif

namespace tasu_pipe {

constexpr int BK = 64;
constexpr int OUT_GU_SWIGLU = 3;    // internal epilogue of tasu_gemm_gate_up_swiglu (after the three TASU_GEMM_OUT_* modes)

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
};

typedef __attribute__((address_space(3))) void lds_void;

// tile s of the virtual one-tile-per-block grid -> (tm, tn): XCD-aware (block b and tile s = b + r*gridDim share b % 8,
// i.e. the XCD, because gridDim is a multiple of 8), bijective, then a 4-row-group raster for L2 reuse of the B panel.
__device__ __forceinline__ void tile_coords(const Args& p, int s, int ntiles, int& tm, int& tn) {
  const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = s & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (s >> 3);
  constexpr int GROUP_M = 4;
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

template <int BM, int BN, int OUT_MODE, bool HAS_BIAS>
__global__ __launch_bounds__(512, 1) void gemm_pipe_kernel(Args p) {
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
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
      const int row0 = tm * BM, col0 = tn * BN;
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
          const int ocol = min(tn * 64 + half * 32 + (rr & 31), p.N - 1);   // output (act) column
          vob[i] = (ocol + (rr >= 32 ? p.N : 0)) * p.ldb * 2 + c * 16 + (3 - (i & 3)) * 1024;
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
#if !defined(TASU_EXP_NO_B_DMA)                    // (timing experiment, wrong results: the B tile never reaches LDS)
      [&]<int... I>(std::integer_sequence<int, I...>) { (one_b(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, PB>{});
#else
      (void)one_b;
#endif
      if (++ld_k == nk) {
        ld_k = 0;
        ld_tile += gridDim.x;
        if (ld_tile < ntiles) setup(ld_tile);
      }
    };
    auto wait_keep_one_step = [&]() {              // all but the newest K-step's pieces have landed
#if defined(TASU_EXP_NO_B_DMA)
      if constexpr (PA == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      return;
#endif
      if constexpr (NG == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if constexpr (NG == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    };
    setup(ld_tile);
    issue(0);
    if (total > 1) {
      issue(1);
      wait_keep_one_step();
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    int slot2 = 2;                                 // ring slot of step q+2
    for (int q = 0; q < total; ++q) {
      if (q + 2 < total) {
        issue(slot2);
        wait_keep_one_step();
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();                // step q+1 is in LDS; slot (q-1)%3 is free (see "Ring safety" above)
      slot2 = slot2 == 2 ? 0 : slot2 + 1;
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
#if !defined(TASU_EXP_NO_B_READS)                  // (timing experiment, wrong results: the B fragments are never refreshed)
#pragma unroll
    for (int j = 0; j < NI; ++j) fb[j] = *(const bf16x8*)(sb + j * 16 * 128);
#else
    (void)sb;
#endif
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
#if defined(TASU_EXP_NO_B_READS)
#pragma unroll
  for (int j = 0; j < NI; ++j) fb0[j] = fb1[j] = bf16x8{1, 1, 1, 1, 1, 1, 1, 1};
#endif
#if defined(TASU_EXP_B_DIRECT)
  // B comes from a fragment-order copy [tile_n][wave column][K-step][k-half][j][64 lanes][8] (p.B), one K-step ahead:
  // gbn0 / gbn1 = the next step's fragments, requested at the top of a K-step and moved into fb0 / fb1 at its end
  bf16x8 gbn0[NI], gbn1[NI];
  constexpr size_t BSTEP = (size_t)2 * NI * 512;
  auto b_tile = [&](int tn) { return p.B + (size_t)(tn * 2 + wc) * (p.K / BK) * BSTEP + lane * 8; };
  auto load_b = [&](bf16x8 (&g0)[NI], bf16x8 (&g1)[NI], const bf16* src) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      g0[j] = *(const bf16x8*)(src + j * 512);
      g1[j] = *(const bf16x8*)(src + (NI + j) * 512);
    }
  };
  const bf16* bnext = nullptr;
#endif
  zero_acc();
  __builtin_amdgcn_s_barrier();                    // step 0 is in LDS
  asm volatile("" ::: "memory");
  read_frags(fa0, fb0, 0, 0);

  // One K-step.  MORE: a next step exists (possibly the first of the next tile): its first fragments are read in phase 2.
  // Straight-line code (no branches inside) so that sched_group_barrier can interleave across the whole phase.
  auto kstep = [&](auto more_tag, int cur) {
    constexpr bool MORE = decltype(more_tag)::value;
    const int nxt = cur == 2 ? 0 : cur + 1;
#if defined(TASU_EXP_B_DIRECT)
    load_b(gbn0, gbn1, bnext);
#endif
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
#if defined(TASU_EXP_B_DIRECT)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      fb0[j] = gbn0[j];
      fb1[j] = gbn1[j];
    }
#endif
    return nxt;
  };

  // acc[i][j][r] = C[m][n], m = row0 + wr*WM + i*16 + (lane&15), n = col0 + wc*WN + j*16 + (lane>>4)*4 + r
  // OUT_GU_SWIGLU epilogue: fragments j = 0, 1 of a wave are gate columns, j = 2, 3 the up values of the same columns
  auto store_gu_swiglu = [&](int row0, int tn) {
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
            const int m = row0 + wr * WM + i * 16 + l15;
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
            const int n = tn * 64 + wc * 32 + cpair;                  // act column of this lane's 8 values
            if (INTERIOR || (m < p.M && n < p.N)) {                   // N % 8 == 0: all eight or none
              *(u32x4*)(gu + (size_t)m * (2 * (size_t)p.N) + n) = u32x4{g0.u[0], g0.u[1], g1.u[0], g1.u[1]};
              *(u32x4*)(gu + (size_t)m * (2 * (size_t)p.N) + p.N + n) = u32x4{u0.u[0], u0.u[1], u1.u[0], u1.u[1]};
              *(u32x4*)(p.act + (size_t)m * p.N + n) = u32x4{a0.u[0], a0.u[1], a1.u[0], a1.u[1]};
            }
          }
        };
        if (row0 + BM <= p.M && tn * 64 + 64 <= p.N) rows(std::true_type{});
        else rows(std::false_type{});
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      asm volatile("" ::: "memory");
      const int m = row0 + wr * WM + i * 16 + l15;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < NI / 2; ++j) {
        const int n = tn * 64 + wc * 32 + j * 16 + l4;              // act column; N % 4 == 0
        if (n >= p.N) continue;
        const bf16x4 g4 = __builtin_convertvector(acc[i][j], bf16x4), u4 = __builtin_convertvector(acc[i][j + NI / 2], bf16x4);
        bf16x4 a4;
#pragma unroll
        for (int r = 0; r < 4; ++r) a4[r] = (bf16)(bf16_round(silu_f((float)g4[r])) * (float)u4[r]);
        *(bf16x4*)(gu + (size_t)m * (2 * (size_t)p.N) + n) = g4;
        *(bf16x4*)(gu + (size_t)m * (2 * (size_t)p.N) + p.N + n) = u4;
        *(bf16x4*)(p.act + (size_t)m * p.N + n) = a4;
      }
    }
  };

  auto store_tile = [&](int row0, int col0) {
    // opaque copies of the lane coordinates: keeps the 32 per-fragment output addresses from being hoisted out of the
    // tile loop into registers that the K loop needs (the kernel sits at the 256-VGPR limit of two waves per SIMD)
    int l15 = lane & 15, l4 = (lane >> 4) * 4;
    asm volatile("" : "+v"(l15), "+v"(l4));
    // the bias depends on the column only: NI x 4 values per lane, loaded once per tile (not once per row block)
    [[maybe_unused]] float bv[NI][4];
    if (HAS_BIAS) {
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = col0 + wc * WN + j * 16 + l4;
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[j][r] = n + r < p.N ? (float)p.bias[n + r] : 0.f;
      }
    }
    const bool interior = row0 + BM <= p.M && col0 + BN <= p.N;     // wave-uniform: the whole tile lies inside the matrix
    if constexpr (OUT_MODE == TASU_GEMM_OUT_F32_RESID_BF16R && BM == 128) {
      // residual add: the fp32 residual rows of IB row blocks are fetched together (clamped addresses, no branches)
      // before the first use -- one memory round trip per IB row blocks instead of one per fragment; with one tile per CU
      // (N = 1536: o and down projections) nothing else hides this latency
      if ((p.ldc & 3) == 0 && (p.N & 3) == 0 && (((uintptr_t)p.R | (uintptr_t)p.C) & 15) == 0) {
        constexpr int IB = 2;                       // (the 256-row tiles have no registers to spare: fragment-wise path below)
        static_assert(MI % IB == 0, "row blocks are processed in groups of IB");
#pragma unroll
        for (int i0 = 0; i0 < MI; i0 += IB) {
          asm volatile("" ::: "memory");
          f32x4 old[IB][NI];
#pragma unroll
          for (int ii = 0; ii < IB; ++ii) {
            const int m = min(row0 + wr * WM + (i0 + ii) * 16 + l15, p.M - 1);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
              const int n = min(col0 + wc * WN + j * 16 + l4, p.N - 4);
              old[ii][j] = *(const f32x4*)(p.R + (size_t)m * p.ldc + n);
            }
          }
#pragma unroll
          for (int ii = 0; ii < IB; ++ii) {
            const int m = row0 + wr * WM + (i0 + ii) * 16 + l15;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
              const int n = col0 + wc * WN + j * 16 + l4;
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
            const int m = row0 + wr * WM + i * 16 + l15;
#pragma unroll
            for (int j = 0; j < NI; j += 2) {
              f32x4 v0 = acc[i][j], v1 = acc[i][j + 1];
              if (HAS_BIAS) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v0[r] += bv[j][r], v1[r] += bv[j + 1][r];
              }
              union { bf16x4 h; unsigned u[2]; } a, b;
              a.h = __builtin_convertvector(v0, bf16x4);
              b.h = __builtin_convertvector(v1, bf16x4);
              union { u32x4 q; bf16 h[8]; } o;
              swap16(a.u[0], b.u[0]);
              swap16(a.u[1], b.u[1]);
              o.q = u32x4{a.u[0], a.u[1], b.u[0], b.u[1]};
              const int n = col0 + wc * WN + j * 16 + cpair;
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
          const size_t rowoff = (size_t)(row0 + wr * WM + i * 16 + l15) * p.ldc + (col0 + wc * WN + l4);
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
      const int m = row0 + wr * WM + i * 16 + l15;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = col0 + wc * WN + j * 16 + l4;
        if (n >= p.N) continue;
        f32x4 v = acc[i][j];
        if (HAS_BIAS) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += bv[j][r];
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
#if defined(TASU_EXP_B_DIRECT)
    const bf16* btile = b_tile(tn);
    if (s == (int)blockIdx.x) load_b(fb0, fb1, btile);            // the workgroup's first tile: step 0 synchronously
    for (int kt = 0; kt + 1 < nk; ++kt) {
      bnext = btile + (size_t)(kt + 1) * BSTEP;
      cur = kstep(T{}, cur);
    }
    bnext = btile;                                 // (no next tile: a harmless re-read)
    if (s + (int)gridDim.x < ntiles) {
      int tm2, tn2;
      tile_coords(p, s + (int)gridDim.x, base_tiles, tm2, tn2);
      bnext = b_tile(tn2);
    }
    cur = kstep(F{}, cur);
#else
    for (int kt = 0; kt + 1 < nk; ++kt) cur = kstep(T{}, cur);
    cur = kstep(F{}, cur);                         // no read-ahead into the next tile: the fragment registers are free
#endif
    if constexpr (OUT_MODE == OUT_GU_SWIGLU) store_gu_swiglu(tm * BM, tn);
    else store_tile(tm * BM, tn * BN);             // for the epilogue, whose stores then drain under the next tile
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
  constexpr int LDS = 3 * (BM + BN) * BK * 2;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_pipe_kernel<BM, BN, OUT_MODE, HAS_BIAS>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = OUT_MODE == OUT_GU_SWIGLU ? (a.N + 63) / 64 : (a.N + BN - 1) / BN;
  const int ntiles = a.tiles_m * a.tiles_n * a.ksplit;
  const int grid = ntiles < cu_count() ? ntiles : cu_count();
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
    default:
      return TASU_ERR_ARG;
  }
}

extern "C" int tasu_gemm_gate_up_swiglu(const void* A, int lda, const void* Wgu, int ldw, void* gu, void* act, int M, int I,
                                        int K, void* stream) {
  using namespace tasu_pipe;
  if (!A || !Wgu || !gu || !act || M <= 0 || I <= 0 || I % 4 || K <= 0 || K % BK || lda % 8 || ldw % 8) return TASU_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)Wgu & 15) || ((uintptr_t)gu & 7) || ((uintptr_t)act & 7)) return TASU_ERR_ARG;
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
  a.ksplit = 1;
  a.split_stride = 0;
  return launch<256, 128, OUT_GU_SWIGLU, false>(a, (hipStream_t)stream);
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
