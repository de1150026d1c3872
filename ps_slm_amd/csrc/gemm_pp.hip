// bf16 NT GEMM for gfx950, 256 x 256 x 64 tiles, eight MFMA waves in two staggered groups ("ping-pong").
//
// Why a second kernel beside gemm_pipe.hip: what bounds that kernel on the wide grids is a CU's ingest from L2 into LDS
// (48 KB per 4.2 MFLOP K-step of its 256 x 128 tile); a 256 x 256 tile takes in 64 KB per 8.4 MFLOP -- 2/3 of
// the bytes per FLOP -- but needs 8 accumulating waves (128 accumulator registers each), so nobody is left over to be a
// loader wave.  Here every wave stages its share of the operands itself, two 1-KiB LDS-DMA pieces per phase, and the two
// wave groups (waves 0-3 = tile rows 0-127, waves 4-7 = rows 128-255; one wave of each group per SIMD) run ONE barrier
// apart: while a group issues its 16 MFMAs of a phase, its SIMD partners read fragments and issue DMA.
//
// One K-tile (64 deep) of a wave = 4 phases, one quadrant (64 x 32) of its 128 x 64 output each, 16 MFMAs per phase:
//     phase 0: (rows 0-63,  cols 0-31)   reads A rows 0-63   (8 x ds_read_b128)      stages A1(q+1)
//     phase 1: (rows 0-63,  cols 32-63)  reads B cols 32-63  (4)                     stages B0(q+2)
//     phase 2: (rows 64-127, cols 32-63) reads A rows 64-127 (8)                     stages A0(q+2)
//     phase 3: (rows 64-127, cols 0-31)  reads B cols 0-31 of K-tile q+1 (4)         stages B1(q+2)
//     phase  = reads ; 2 DMA pieces ; s_waitcnt vmcnt(10) ; s_barrier ; s_waitcnt lgkmcnt(0) ; 16 MFMA ; s_barrier
// (the two B fragment sets swap roles every K-tile, so the loop body exists for both parities of the stream position q).
//
// Staging units (16 KiB = 16 pieces, two per wave): A0 / A1 = the phase-0 / phase-2 rows of BOTH groups, B0 / B1 = the
// cols-0-31 / cols-32-63 slices of all four wave columns; two LDS buffers of 4 units (128 KiB), K-tile q lives in buffer
// q & 1.  Hazards, with group 1 one barrier behind group 0 (phase p of group 0 lies between barriers 2p-1 .. 2p+1, of
// group 1 between 2p .. 2p+2):
//   WAR  a unit read in phase r is overwritten by DMA issued in phase r + 2 or later: every wave's reads of phase r have
//        returned (its lgkmcnt(0)) before it arrives at barrier 2r + 2, which precedes both groups' phase r + 2 issue.
//   RAW  the unit read in phase r was waited for (counted vmcnt, own pieces) by every wave before the FIRST barrier of its
//        phase r - 1, i.e. before barrier 2r - 1 at the latest, which precedes both groups' reads of phase r.
//   The DMA order is the consumption order B0(q), A0(q), B1(q), A1(q), B0(q+1) ..., one unit per phase, six phases ahead of
//   its first read, so "the unit the NEXT phase reads has landed" is always "all but my 10 youngest DMAs are done".
// The stream of K-tiles does not stop at tile boundaries (persistent workgroups, static tile list): the first units of a
// workgroup's next tile land while it converts and stores the finished one.
#include "gemm_epilogue.h"

namespace tasu_pp {

using namespace tasu_gemm;

[[maybe_unused]] constexpr int BK = 64, BM = 256;
constexpr int UNIT = 16384, BUF = 4 * UNIT;        // LDS: [buffer][A0 | A1 | B0 | B1], 128 KiB

typedef __attribute__((address_space(3))) void lds_void;

template <int OUT_MODE, bool HAS_BIAS>
__global__ __launch_bounds__(512, 1) void gemm_pp_kernel(Args p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BN = 256, MI = 8, NI = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 2, wc = wave & 3;          // group (tile rows wr*128 ..), wave column (tile cols wc*64 ..)
  // split-K (TASU_GEMM_OUT_F32 only, gemm_epilogue.h): work item s = K range s / base_tiles of output tile s % base_tiles
  const int base_tiles = p.tiles_m * p.tiles_n;
  // ---- the workgroup's list of work items (PpSchedule, gemm_epilogue.h): stream-K range pieces first, then whole tiles
  using Item = PpSchedule::Item;
  constexpr int PART = PpSchedule::PART, HEAD = PpSchedule::HEAD;
  PpSchedule sched;
  sched.init((int)gridDim.x, (int)blockIdx.x, p.K / (2 * BK), p.ksplit, base_tiles, p.sk_tiles);
  const int G = sched.G, wg = sched.wg, P = sched.P, dp_tiles = sched.dp_tiles;
  auto ub = [&](int w) { return sched.ub(w); };
  auto get_item = [&](int idx, Item& it) { return sched.item(idx, it); };

  // ------------------------------------------------------------------ staging
  // Per work item only wave-uniform values change: the operand origins (1 KiB below the first A / B row of the tile at the
  // item's first K element: piece 1 of a pair is addressed with immediate offset 1 KiB, which moves the LDS address and the
  // global address alike, and a per-lane offset lowered by the same amount) and the byte counts up to the end of the matrices.
  // Rows past the end of a matrix (edge tiles) are beyond the descriptor's range: the hardware returns zeros for them, nothing
  // is clamped per lane, and the per-lane offsets below are the same for every item.
  struct Src {
    const char* a;
    const char* b;
    unsigned na, nb;
  };
  auto setup = [&](Src& d, const Item& it) {
    int tm, tn;
    tile_coords<4>(p, it.tile, base_tiles, tm, tn);
    const int row0 = tm * BM;
    const int brow0 = p.n0 + (OUT_MODE == OUT_GU_SWIGLU ? tn * 128 : tn * BN);   // OUT_GU_SWIGLU: first gate row = first act column
    const size_t k0 = (size_t)it.k0t * BK;           // first K element of this work item
    d.a = (const char*)(p.A + (size_t)row0 * p.lda + k0) - 1024;
    d.b = (const char*)(p.B + (size_t)brow0 * p.ldb + k0) - 1024;
    // (the byte counts run from the descriptor base, which a K range moves into the first row: rows past the end are still
    // past the count, and the last row's K range ends inside it)
    const size_t ra = (size_t)(p.M - row0) * p.lda * 2 + 1024 - k0 * 2;
    // OUT_GU_SWIGLU: the up rows lie N rows behind the gate rows; the tile's 128 act columns exist (N % 128 == 0 is required)
    const size_t rb = (size_t)(OUT_MODE == OUT_GU_SWIGLU ? 2 * p.N - brow0 : p.N - brow0) * p.ldb * 2 + 1024 - k0 * 2;
    d.na = (unsigned)(ra < 0x7ffffff0ull ? ra : 0x7ffffff0ull);
    d.nb = (unsigned)(rb < 0x7ffffff0ull ? rb : 0x7ffffff0ull);
  };
  int voa[2][2], vob[2][2];                          // [unit half][piece] per-lane byte offsets, the same for every tile
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // A unit h, wave w: tile rows (w>>2)*128 + h*64 + (w&3)*16 + i*8 + (lane>>3); LDS chunk lane&7 <- global chunk ^ swizzle
      const int ra = (wave >> 2) * 128 + h * 64 + (wave & 3) * 16 + i * 8 + (lane >> 3);
      const int ca = (lane & 7) ^ ((ra >> 1) & 7);
      voa[h][i] = ra * p.lda * 2 + ca * 16 + (1 - i) * 1024;
      // B unit h, wave w: tile cols (w>>1)*64 + h*32 + (w&1)*16 + i*8 + (lane>>3)
      const int rb = (wave >> 1) * 64 + h * 32 + (wave & 1) * 16 + i * 8 + (lane >> 3);
      const int cb = (lane & 7) ^ ((rb >> 1) & 7);
      // OUT_GU_SWIGLU: a wave column's 64 weight rows = 32 gate rows + the 32 up rows of the same act columns
      const int wrow = OUT_MODE == OUT_GU_SWIGLU ? (rb >> 6) * 32 + (rb & 31) + ((rb & 32) ? p.N : 0) : rb;
      vob[h][i] = wrow * p.ldb * 2 + cb * 16 + (1 - i) * 1024;
    }
  // this wave's pair of pieces inside a unit: pieces 2w, 2w+1 (2 KiB contiguous)
  char* const pair_base = smem + wave * 2048;
  // stage<U, BUFI>(src, kt): unit U (0 = B0, 1 = A0, 2 = B1, 3 = A1: the DMA order) of K-tile kt of tile src into buffer BUFI
  auto stage = [&](const Src& d, auto unit_tag, auto buf_tag, int kt) {
    constexpr int U = decltype(unit_tag)::value, BUFI = decltype(buf_tag)::value;
    constexpr bool IS_A = U & 1;
    constexpr int H = U >> 1;
    char* dst = pair_base + BUFI * BUF + (IS_A ? H * UNIT : 2 * UNIT + H * UNIT);
    const int koff = kt * (BK * 2);
    if constexpr (IS_A) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)d.a, 0, (int)d.na, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, voa[H][0], koff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, voa[H][1], koff, 1024, 0);
    } else {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)d.b, 0, (int)d.nb, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, vob[H][0], koff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, vob[H][1], koff, 1024, 0);
    }
  };
  using Z = std::integral_constant<int, 0>;
  using O = std::integral_constant<int, 1>;
  using U0 = std::integral_constant<int, 0>;
  using U1 = std::integral_constant<int, 1>;
  using U2 = std::integral_constant<int, 2>;
  using U3 = std::integral_constant<int, 3>;

  // ------------------------------------------------------------------ fragment addressing
  // LDS rows are 128 B (one K-tile of a row), 16-B chunk c of row r at chunk c ^ ((r >> 1) & 7); the rows a lane reads are
  // base (a multiple of 16) + (lane & 15), so the swizzle term is ((lane >> 1) & 7) for all of them
  const int sw = (lane >> 1) & 7;
  // A unit h holds this group's rows h*64 .. at unit rows wr*64 ..; B unit h holds this wave column's cols h*32 .. at unit rows wc*32 ..
  const char* pa[2][2];                              // [buffer][k half]
  const char* pb[2][2];
#pragma unroll
  for (int bi = 0; bi < 2; ++bi)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ro = (lane & 15) * 128 + (((kk * 4 + (lane >> 4)) ^ sw) << 4);
      pa[bi][kk] = smem + bi * BUF + wr * 64 * 128 + ro;
      pb[bi][kk] = smem + bi * BUF + 2 * UNIT + wc * 32 * 128 + ro;
    }

  bf16x8 fa[4][2], fbx[2][2], fby[2][2];
  f32x4 acc[MI][NI];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto read_a = [&](auto buf_tag, auto h_tag) {
    constexpr int BI = decltype(buf_tag)::value, H = decltype(h_tag)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) fa[i][kk] = *(const bf16x8*)(pa[BI][kk] + H * UNIT + i * 16 * 128);
  };
  auto read_b = [&](bf16x8 (&fb)[2][2], auto buf_tag, auto h_tag) {
    constexpr int BI = decltype(buf_tag)::value, H = decltype(h_tag)::value;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) fb[j][kk] = *(const bf16x8*)(pb[BI][kk] + H * UNIT + j * 16 * 128);
  };
  auto mma = [&](bf16x8 (&fb)[2][2], auto mh_tag, auto nh_tag) {
    constexpr int MH = decltype(mh_tag)::value, NH = decltype(nh_tag)::value;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[MH * 4 + i][NH * 2 + j] = mfma16(fb[j][kk], fa[i][kk], acc[MH * 4 + i][NH * 2 + j]);
  };
  auto fence = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
  };
  // the two halves of a phase around its first barrier
  auto before_mma = [&]() {
    fence();
    // all but my 10 youngest pieces have landed: 5 units were staged after the one the next phase reads
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
  };
  auto after_mma = [&]() {
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    fence();
  };

  Src cur, nxt;
  int nk_cur = 0;                                    // K-tiles of the current work item
  // Two K-tiles kt, kt + 1 of the current tile (stream parities 0 and 1 = buffers 0 and 1; the cols-0-31 fragments of the first
  // are in fbx).  What is staged two K-tiles ahead belongs to the workgroup's next work item once kt + 2 == nk_cur (even).
  auto kpair = [&](int kt) {
    const bool wrap = kt + 2 >= nk_cur;
    Src ahead;
    ahead.a = wrap ? nxt.a : cur.a;
    ahead.b = wrap ? nxt.b : cur.b;
    ahead.na = wrap ? nxt.na : cur.na;
    ahead.nb = wrap ? nxt.nb : cur.nb;
    const int k2 = wrap ? 0 : kt + 2;
    // ---- K-tile kt (parity 0)
    read_a(Z{}, Z{});
    stage(cur, U3{}, O{}, kt + 1);                   // A1(kt+1): always this tile
    before_mma();
    mma(fbx, Z{}, Z{});
    after_mma();
    read_b(fby, Z{}, O{});
    stage(ahead, U0{}, Z{}, k2);
    before_mma();
    mma(fby, Z{}, O{});
    after_mma();
    read_a(Z{}, O{});
    stage(ahead, U1{}, Z{}, k2);
    before_mma();
    mma(fby, O{}, O{});
    after_mma();
    read_b(fby, O{}, Z{});                           // cols 0-31 of K-tile kt+1 (buffer 1)
    stage(ahead, U2{}, Z{}, k2);
    before_mma();
    mma(fbx, O{}, Z{});
    after_mma();
    // ---- K-tile kt + 1 (parity 1)
    read_a(O{}, Z{});
    stage(ahead, U3{}, Z{}, k2);
    before_mma();
    mma(fby, Z{}, Z{});
    after_mma();
    read_b(fbx, O{}, O{});
    stage(ahead, U0{}, O{}, k2 + 1);
    before_mma();
    mma(fbx, Z{}, O{});
    after_mma();
    read_a(O{}, O{});
    stage(ahead, U1{}, O{}, k2 + 1);
    before_mma();
    mma(fbx, O{}, O{});
    after_mma();
    read_b(fbx, Z{}, Z{});                           // cols 0-31 of the next K-tile (buffer 0; possibly of the next tile)
    stage(ahead, U2{}, O{}, k2 + 1);
    before_mma();
    mma(fby, O{}, Z{});
    after_mma();
  };

  // ------------------------------------------------------------------ prologue: the DMA sequence up to phase 0 of K-tile 0
  Item ci, ni;
  if (!get_item(0, ci)) return;
  setup(cur, ci);
  nxt = cur;
  stage(cur, U0{}, Z{}, 0);
  stage(cur, U1{}, Z{}, 0);
  stage(cur, U2{}, Z{}, 0);
  stage(cur, U3{}, Z{}, 0);
  stage(cur, U0{}, O{}, 1);
  stage(cur, U1{}, O{}, 1);
  stage(cur, U2{}, O{}, 1);
  zero_acc();
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  // B0(0) and A0(0), the two oldest of the 7 units
  __builtin_amdgcn_s_barrier();
  fence();
  read_b(fbx, Z{}, Z{});

  auto store_gu = [&](int row0, int tn, bool off) {
    Args q = p;
    if (off) q.M = 0;
    store_gu_swiglu<MI, NI, BM>(q, acc, row0, p.n0 + tn * 128 + wc * 32, wr * 128, lane);
  };
  auto store_c = [&](int row0, int col0, int ks, bool off) {
    Args q = p;
    if (off) q.M = 0;
    if constexpr (OUT_MODE == TASU_GEMM_OUT_F32) q.C = (float*)p.C + (size_t)ks * p.split_stride;   // slab of this K range
    store_tile<MI, NI, OUT_MODE, HAS_BIAS, BM, BN, false>(q, acc, row0, col0, wr * 128, wc * 64, lane);
  };
  // stream-K partial tiles (256 KiB per workgroup): fragment f = i * NI + j of wave w at float4 index (w * 32 + f) * 64 + lane,
  // i.e. a wave's fragments are 1 KiB apart (immediate offsets from one address per accumulator row)
  auto part_row = [&](int w, int i) {
    int slot = wave * (32 * 64) + lane;              // opaque: recomputed where it is used, not kept in a register (or in
    asm volatile("" : "+v"(slot));                   // scratch) across the K loop
    return (f32x4*)(p.sk_partial + (size_t)w * (BM * BN)) + (slot + i * NI * 64);
  };

  for (int idx = 0;; ++idx) {
    // the workgroup's next work item; after the last one the stream re-stages this item's first units (never read) so that
    // the DMA count behind every wait stays the same
    const bool more = get_item(idx + 1, ni);
    if (more) setup(nxt, ni);
    nk_cur = ci.nkt;
    if (wr == 1) {                                   // group 1 runs one barrier behind group 0 through the K loop
      fence();
      __builtin_amdgcn_s_barrier();
      fence();
    }
    for (int kt = 0; kt < nk_cur; kt += 2) kpair(kt);
    if (wr == 0) {                                   // ... and both run the epilogue together (a group that stored alone would
      fence();                                       // hold its partners at their next barrier for the whole epilogue, twice)
      __builtin_amdgcn_s_barrier();
      fence();
    }
    {
      // a K range that does not begin its tile (PART): the accumulators go to this workgroup's partial tile, then the flag.
      // The stores are issued for EVERY item, through a buffer descriptor that is empty unless the item is such a range (the
      // hardware drops them): no branch around code that reads all 128 accumulator registers.  The partial tiles are written
      // and read with device-scope (sc1) accesses, which are coherent between the XCDs' L2s by themselves: no cache-wide
      // write-back / invalidate (a release / acquire FENCE would flush the operand panels the other workgroups of the XCD
      // are sharing through that L2).
      const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.sk_partial + (size_t)wg * (BM * BN)), 0,
                                                                           ci.kind == PART ? BM * BN * 4 : 0, 0x00020000);
      int slot = (wave * (32 * 64) + lane) * 16;
      asm volatile("" : "+v"(slot));
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          union { f32x4 f; u32x4 u; } v;
          v.f = acc[i][j];
          __builtin_amdgcn_raw_buffer_store_b128(v.u, prs, slot + (i * NI + j) * 1024, 0, /*sc1*/ 16);
        }
    }
    if (ci.kind == PART) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (threadIdx.x == 0) __hip_atomic_store(p.sk_flags + wg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (ci.kind == HEAD) {
      // the tile's first pairs: the later ranges belong to the next workgroups (each one's FIRST item, done long ago);
      // their partial tiles are added in K order, so the sum does not depend on who finished when
      const int tile_end = (ci.tile - dp_tiles + 1) * P;
#pragma nounroll
      for (int j = wg + 1; j < G && ub(j) < tile_end; ++j) {
        if (threadIdx.x == 0) {
          // bounded: the producer of range j is workgroup j's FIRST item, so on a healthy launch the flag is up within
          // microseconds.  If it is not there after ~2 s (a grid that was not co-resident and got dispatched out of order, a
          // second GEMM sharing the workspace from another stream), abort the launch -- the host sees a launch failure --
          // instead of hanging the GPU silently.
          long long spins = 0;
          while (__hip_atomic_load(p.sk_flags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1ll << 25)) __builtin_trap();
          }
          __hip_atomic_store(p.sk_flags + j, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // left at 0 for the next launch
        }
        __builtin_amdgcn_s_barrier();
        // 32 KiB per wave, read one accumulator row (4 loads) at a time with the next row already in flight: the round trip
        // to the memory side (sc1: past the L2) is paid once per partial tile, not once per row
        f32x4 ta[NI], tb[NI];
        auto issue = [&](f32x4 (&t)[NI], int i) {
          const f32x4* q = part_row(j, i);
          asm volatile(
              "global_load_dwordx4 %0, %4, off sc1\n"
              "global_load_dwordx4 %1, %4, off offset:1024 sc1\n"
              "global_load_dwordx4 %2, %4, off offset:2048 sc1\n"
              "global_load_dwordx4 %3, %4, off offset:3072 sc1"
              : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])
              : "v"(q)
              : "memory");
        };
        auto add = [&](f32x4 (&t)[NI], int i, auto left_tag) {     // waits until all but the `left` youngest loads are back
          constexpr int LEFT = decltype(left_tag)::value;
          if constexpr (LEFT == 4) asm volatile("s_waitcnt vmcnt(4)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3])::"memory");
          else asm volatile("s_waitcnt vmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3])::"memory");
#pragma unroll
          for (int jj = 0; jj < NI; ++jj) acc[i][jj] += t[jj];
        };
        using W4 = std::integral_constant<int, 4>;
        using W0 = std::integral_constant<int, 0>;
        issue(ta, 0);
#pragma unroll
        for (int i = 0; i < MI; i += 2) {
          issue(tb, i + 1);
          add(ta, i, W4{});
          if (i + 2 < MI) {
            issue(ta, i + 2);
            add(tb, i + 1, W4{});
          } else {
            add(tb, i + 1, W0{});
          }
        }
      }
    }
    {
      // the tile's epilogue -- for EVERY item, PART ones with an empty matrix (every store predicated off): see above
      int tm, tn;
      tile_coords<4>(p, ci.tile, base_tiles, tm, tn);
      if constexpr (OUT_MODE == OUT_GU_SWIGLU) store_gu(tm * BM, tn, ci.kind == PART);
      else store_c(tm * BM, p.n0 + tn * BN, ci.ks, ci.kind == PART);
    }
    zero_acc();
    if (!more) break;
    cur = nxt;
    ci = ni;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the re-staged units: no DMA may be in flight into a released LDS
#endif
}

int cu_count() {
  static const int n = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    }
    return cus >= 8 ? (cus & ~7) : 8;
  }();
  return n;
}

// Library default of sk_plan's max_rem.  Measured (profiles/r03_gemm_streamk.txt): the round trip of the partial tiles costs
// ~35 us per launch; it pays for fewer tiles than CUs behind a long K (d_gate_up: 96 tiles, K = 17920: 203 -> 181 us), not for
// shapes that only lose a fraction of their last round (d_down 2.19 rounds, gate|up 4.4, lm_head 18.6: slower by 5-25 us) --
// so those keep whole tiles unless TASU_GEMM_SK_MAXREM says otherwise.  TASU_GEMM_SK=0 disables the schedule altogether.
double sk_max_rem() {
  static const double v = [] {
    const char* off = tasu_lab_env("TASU_GEMM_SK");
    if (off && off[0] == '0') return -1.0;
    const char* e = tasu_lab_env("TASU_GEMM_SK_MAXREM");
    return e ? atof(e) : 0.0;
  }();
  return v;
}

template <int OUT_MODE, bool HAS_BIAS>
int launch(Args a, hipStream_t st) {
  constexpr int LDS = 2 * BUF;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<OUT_MODE, HAS_BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  a.tiles_m = (a.M + BM - 1) / BM;
  const int n_end = a.n1 > 0 ? a.n1 : a.N;
  a.tiles_n = OUT_MODE == OUT_GU_SWIGLU ? (n_end - a.n0 + 127) / 128 : (n_end - a.n0 + 255) / 256;
  const int ntiles = a.tiles_m * a.tiles_n * a.ksplit;
  const int G = cu_count();
  // stream-K over the last (partial + one whole) round when the workspace is there (sk_plan, gemm_epilogue.h)
  a.sk_tiles = a.ksplit == 1 && (long long)ntiles * (a.K / 128) < (1 << 22) ? sk_plan(ntiles, a.K / 128, G, a.sk_flags && a.sk_partial, a.sk_rem > -1.5 ? a.sk_rem : sk_max_rem()) : 0;
  if (!a.sk_tiles) a.sk_flags = nullptr, a.sk_partial = nullptr;
  const int grid = a.sk_tiles || ntiles >= G ? G : ntiles;
  ++tasu_gemm::gemm_launches();
  TASU_LAUNCH((gemm_pp_kernel<OUT_MODE, HAS_BIAS>), dim3(grid), dim3(512), LDS, st, a);
  return TASU_OK;
}

// workspace of tasu_gemm_nt_bf16_ws: TASU_GEMM_WS_COUNTERS ints (all zero between launches: the flags), then the partial tiles
void set_sk_workspace(Args& a, void* ws, size_t ws_bytes, double sk_rem) {
  const size_t need = TASU_GEMM_WS_COUNTERS * sizeof(int) + (size_t)cu_count() * BM * 256 * sizeof(float);
  a.sk_rem = sk_rem;
  if (!ws || ws_bytes < need || cu_count() > TASU_GEMM_WS_COUNTERS) return;
  a.sk_flags = (int*)ws;
  a.sk_partial = (float*)((char*)ws + TASU_GEMM_WS_COUNTERS * sizeof(int));
}

}  // namespace tasu_pp

// C[M,N] = A[M,K] . B[N,K]^T (+ bias) with the 256 x 256 ping-pong kernel; same contract as tasu_gemm_nt_bf16_ws
// (K % 64 == 0, lda / ldb % 8 == 0, 16-byte aligned operands).  Called from gemm.hip's dispatcher.
int tasu_gemm_pp_dispatch(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* bias, const float* resid,
                          int M, int N, int K, int out_mode, hipStream_t st, int n0, int n1, void* ws, size_t ws_bytes, double sk_rem) {
  using namespace tasu_pp;
  Args a;
  set_sk_workspace(a, ws, ws_bytes, sk_rem);
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
      return hb ? launch<TASU_GEMM_OUT_BF16, true>(a, st) : launch<TASU_GEMM_OUT_BF16, false>(a, st);
    case TASU_GEMM_OUT_F32:
      return hb ? launch<TASU_GEMM_OUT_F32, true>(a, st) : launch<TASU_GEMM_OUT_F32, false>(a, st);
    case TASU_GEMM_OUT_F32_RESID_BF16R:
      return hb ? launch<TASU_GEMM_OUT_F32_RESID_BF16R, true>(a, st) : launch<TASU_GEMM_OUT_F32_RESID_BF16R, false>(a, st);
#ifdef TASU_LAB
    case OUT_DSWIGLU:                                // `resid` = the saved gate|up matrix (bf16 [M, 2N]); C = dgu [M, 2N]
      if (hb || !resid || N % 8 || ldc != 2 * N) return TASU_ERR_ARG;
      a.act = (bf16*)resid;
      a.R = nullptr;
      return launch<OUT_DSWIGLU, false>(a, st);
#endif
    default:
      return TASU_ERR_ARG;
  }
}

// gate|up projection + SwiGLU epilogue on 256 x 256 tiles (128 act columns); called from tasu_gemm_gate_up_swiglu (gemm_pipe.hip)
int tasu_gemm_pp_gu_dispatch(const void* A, int lda, const void* Wgu, int ldw, void* gu, void* act, int M, int I, int K, hipStream_t st,
                             int n0, int n1, void* ws, size_t ws_bytes) {
  using namespace tasu_pp;
  if (I % 128 || K < 256 || K % 128) return TASU_ERR_ARG;
  Args a;
  set_sk_workspace(a, ws, ws_bytes, -2.0);
  a.n0 = n0;
  a.n1 = n1;
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
  return launch<OUT_GU_SWIGLU, false>(a, st);
}

// K range slabs on 256 x 256 tiles: partials[ks][M, ldc] (fp32) = A[:, ks-th K range] . B[:, ks-th K range]^T, ks < ksplit.
// For outputs too narrow to fill the chip with whole tiles behind a long K (N = 1536 at M = 4096: 96 tiles): the consumer sums
// the slabs (tasu_rmsnorm_*_slabs, tasu_sum_slabs_bf16).  K % (128 * ksplit) == 0, K / ksplit >= 256.
extern "C" int tasu_gemm_nt_bf16_slabs(const void* A, int lda, const void* B, int ldb, float* partials, int ldc, int M, int N, int K,
                                       int ksplit, void* stream) {
  using namespace tasu_pp;
  if (!A || !B || !partials || M <= 0 || N <= 0 || K <= 0 || ksplit < 1 || ksplit > 16 || K % (128 * ksplit) || K / ksplit < 256 ||
      lda % 8 || ldb % 8 || ldc < N)
    return TASU_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)partials & 15)) return TASU_ERR_ARG;
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
  return launch<TASU_GEMM_OUT_F32, false>(a, (hipStream_t)stream);
}

int tasu_gemm_pipe_dispatch(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* bias,
                            const float* resid, int M, int N, int K, int out_mode, int bn, hipStream_t st, int n0, int n1);

// The stream-K work-item lists of tasu_gemm_nt_bf16_streamk for `tiles` output tiles of `pairs` K-tile pairs on `grid`
// workgroups (host restatement through the kernel's own PpSchedule; no GPU): items[w][i] = {tile, first K-tile, K-tiles, role
// (0 whole tile, 1 partial-tile producer, 2 tile owner)}, up to max_items per workgroup; counts[w] = items of workgroup w.
// Returns the number of tiles the plan cuts along K (0: whole tiles only), or -1 on bad arguments.
extern "C" int tasu_streamk_schedule(int tiles, int pairs, int grid, int32_t* items, int32_t* counts, int max_items) {
  using namespace tasu_gemm;
  if (tiles <= 0 || pairs <= 0 || grid <= 0 || !items || !counts || max_items <= 0) return -1;
  const int sk = (long long)tiles * pairs < (1 << 22) ? sk_plan(tiles, pairs, grid, true, 1.0) : 0;
  for (int w = 0; w < grid; ++w) {
    PpSchedule s;
    s.init(grid, w, pairs, 1, tiles, sk);
    PpSchedule::Item it;
    int n = 0;
    for (; s.item(n, it); ++n) {
      if (n >= max_items) return -1;
      int32_t* o = items + ((size_t)w * max_items + n) * 4;
      o[0] = it.tile, o[1] = it.k0t, o[2] = it.nkt, o[3] = it.kind;
    }
    counts[w] = n;
  }
  return sk;
}

// tasu_gemm_nt_bf16_ws on the 256 x 256 kernel with the stream-K schedule wherever the tiles do not fill whole rounds of
// workgroups, whatever the dispatcher's policy would choose (tests, tuning runs).  Same contract; K % 128 == 0, K >= 256.
extern "C" int tasu_gemm_nt_bf16_streamk(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* bias,
                                         const float* resid, int M, int N, int K, int out_mode, void* workspace,
                                         int64_t workspace_bytes, void* stream) {
  if (!A || !B || !C || !workspace || M <= 0 || N <= 0 || K < 256 || K % 128 || lda % 8 || ldb % 8) return TASU_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)workspace & 15)) return TASU_ERR_ARG;
  if (out_mode == TASU_GEMM_OUT_F32_RESID_BF16R && !resid) return TASU_ERR_ARG;
  return tasu_gemm_pp_dispatch(A, lda, B, ldb, C, ldc, bias, resid, M, N, K, out_mode, (hipStream_t)stream, 0, 0, workspace,
                               (size_t)workspace_bytes, 1.0);
}

// tasu_gemm_nt_bf16 on a NAMED kernel, regardless of the dispatcher's tile policy (tests, tuning runs; include/tasu_hip.h)
extern "C" int tasu_gemm_nt_bf16_kernel(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* bias,
                                        const float* resid, int M, int N, int K, int out_mode, int kernel, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || K % 64 || lda % 8 || ldb % 8) return TASU_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return TASU_ERR_ARG;
  if (out_mode == TASU_GEMM_OUT_F32_RESID_BF16R && !resid) return TASU_ERR_ARG;
  switch (kernel) {
    case TASU_GEMM_KERNEL_PP256:
      if (K < 256 || K % 128) return TASU_ERR_ARG;            // an even number (>= 4) of 64-deep K-tiles
      return tasu_gemm_pp_dispatch(A, lda, B, ldb, C, ldc, bias, resid, M, N, K, out_mode, (hipStream_t)stream, 0, 0, nullptr, 0, -2.0);
    case TASU_GEMM_KERNEL_PIPE128:
      return tasu_gemm_pipe_dispatch(A, lda, B, ldb, C, ldc, bias, resid, M, N, K, out_mode, 128, (hipStream_t)stream, 0, 0);
    case TASU_GEMM_KERNEL_PIPE192:
      return tasu_gemm_pipe_dispatch(A, lda, B, ldb, C, ldc, bias, resid, M, N, K, out_mode, 192, (hipStream_t)stream, 0, 0);
    case TASU_GEMM_KERNEL_PIPE96:
      return tasu_gemm_pipe_dispatch(A, lda, B, ldb, C, ldc, bias, resid, M, N, K, out_mode, 96, (hipStream_t)stream, 0, 0);
    default:
      return TASU_ERR_ARG;
  }
}
