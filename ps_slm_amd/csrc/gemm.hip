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
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"
#include "gemm_epilogue.h"
#include "../../include/tasu_hip.h"

namespace {

constexpr int BK = 64;

struct GemmArgs {
  const bf16* A;
  const bf16* B;
  void* C;
  const float* R;
  const bf16* bias;
  int M, N, K;
  int lda, ldb, ldc;
  int tiles_m, tiles_n;
  int ksplit;           // > 1: blockIdx covers tiles x ksplit, every block reduces K-steps [ks*per, (ks+1)*per)
  float* partial;       // [tiles][ksplit][BM*BN] fp32 partial tiles (register-image order)
  int* counters;        // [tiles] arrival counters, zero between launches (the last arriver resets its tile's)
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// BM x BN block tile (BM, BN multiples of 32), 2x2 waves, wave tile (BM/2) x (BN/2) = MI x NI MFMA tiles.
// BN = 96 exists for N = 1536 at M = 4096: 32 x 16 = 512 tiles = exactly two per CU (128x128 would give 384 tiles and
// leave half of the CUs with one block).
// 256 x 256 / 8 waves (2 x 4, wave tile 128 x 64, one block per CU) halves the L2 -> LDS bytes per FLOP of the 128-wide
// tiles ((BM+BN)/(BM*BN): 1/128 vs 1/64); the 128-wide tiles run at roughly the chip's L2 bandwidth (2 x 32 KB per
// K-step per CU), which is what caps them near 1 PFLOP/s.
//
// SPLITK: grids that cannot fill the chip with whole tiles (M = 4096 x N = 1536 is 128 tiles of 256 x 192) split the K
// range over `ksplit` blocks per tile.  Every block writes its fp32 partial tile to the workspace, fences at agent
// scope (the 8 XCD L2s are not coherent with each other without it) and takes a ticket from the tile's arrival counter;
// the LAST arriver adds the partials in split order (ksplit == 2: own + other, which is order-independent) and runs the
// normal epilogue.  No spinning, so no co-residency requirement; bitwise deterministic.
template <int BM, int BN, int NWM, int NWN, int OUT_MODE, bool HAS_BIAS, int SCHED, bool SPLITK = false>
__global__ __launch_bounds__(64 * NWM * NWN, 2) void gemm_nt_kernel(GemmArgs p) {
  constexpr int NW = NWM * NWN, NT = 64 * NW;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int WM = BM / NWM, WN = BN / NWN, MI = WM / 16, NI = WN / 16;
  constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW;   // 1-KiB pieces (8 rows x 128 B) staged per wave and K-step
  static_assert(PA * NW * 8 == BM && PB * NW * 8 == BN, "tile rows must split evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave / NWN, wc = wave % NWN;
  (void)NT;

  // ---- XCD-aware, bijective block -> tile map, then an 8-row-group raster for L2 reuse of the B panel.
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  int ks = 0;
  if (SPLITK) {
    ks = logical % p.ksplit;
    logical /= p.ksplit;
  }
  const int tile_id = logical;
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
  const bf16* ga[PA];
  const bf16* gb[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int r = (wave * PA + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    ga[i] = p.A + (size_t)min(row0 + r, p.M - 1) * p.lda + c * 8 - (i & 3) * 512;   // - the immediate offset of stage()
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int r = (wave * PB + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    gb[i] = p.B + (size_t)min(col0 + r, p.N - 1) * p.ldb + c * 8 - (i & 3) * 512;
  }

  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE_BYTES;
    const int koff = kt * BK;
    // one M0 (LDS base) per four pieces: the immediate offset (i & 3) KiB moves the LDS address and the global address
    // alike, and the per-lane pointers were lowered by the same amount (see gemm_pipe.hip)
    auto one_a = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      __builtin_amdgcn_global_load_lds((glb_void*)(ga[i] + koff), (lds_void*)(base + (wave * PA + (i & ~3)) * 1024), 16,
                                       (i & 3) * 1024, 0);
    };
    auto one_b = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      __builtin_amdgcn_global_load_lds((glb_void*)(gb[i] + koff), (lds_void*)(base + A_BYTES + (wave * PB + (i & ~3)) * 1024), 16,
                                       (i & 3) * 1024, 0);
    };
    [&]<int... I>(std::integer_sequence<int, I...>) { (one_a(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, PA>{});
    [&]<int... I>(std::integer_sequence<int, I...>) { (one_b(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, PB>{});
  };

  // ---- per-lane LDS read offsets (bytes) for the two 32-deep k sub-steps.
  const int sw = (lane >> 1) & 7;
  int roff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) roff[kk] = (lane & 15) * 128 + (((kk * 4 + (lane >> 4)) ^ sw) << 4);

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int kbeg = 0, nk = p.K / BK;
  if (SPLITK) {
    const int per = (nk + p.ksplit - 1) / p.ksplit;
    kbeg = ks * per;
    nk = min(nk, kbeg + per);
  }
  if (kbeg < nk) stage(0, kbeg);
  __syncthreads();
  for (int kt = kbeg; kt < nk; ++kt) {
    const int cur = (kt - kbeg) & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* sa = smem + cur * STAGE_BYTES + (wr * WM) * 128;
    const char* sb = smem + cur * STAGE_BYTES + A_BYTES + (wc * WN) * 128;
    // all fragment reads of the K-step are issued up front (both 32-deep sub-steps): the second half lands under
    // the first half's MFMAs instead of stalling the wave between the two halves.
    bf16x8 fa[2][MI], fb[2][NI];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[kk][i] = *(const bf16x8*)(sa + i * 16 * 128 + roff[kk]);
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[kk][j] = *(const bf16x8*)(sb + j * 16 * 128 + roff[kk]);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = mfma16(fb[kk][j], fa[kk][i], acc[i][j]);
    if (SCHED == 1) {
      // pin the issue order (hipcc otherwise re-serialises reads -> wait -> MFMAs per half): first half's reads, then
      // the second half's reads interleaved one per two MFMAs of the first half, then the remaining MFMAs.
      __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);
#pragma unroll
      for (int r = 0; r < MI + NI; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * MI * NI - 2 * (MI + NI), 0);
    }
    __syncthreads();
  }

  if (SPLITK) {
    constexpr int TILE_F = BM * BN;
    float* mine = p.partial + ((size_t)tile_id * p.ksplit + ks) * TILE_F;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) *(f32x4*)(mine + ((i * NI + j) * NT + tid) * 4) = acc[i][j];
    __threadfence();                       // release: partial tile visible device-wide before the ticket
    __syncthreads();
    int* flag = (int*)smem;
    if (tid == 0) flag[0] = atomicAdd(p.counters + tile_id, 1);
    __syncthreads();
    if (flag[0] != p.ksplit - 1) return;   // not the last arriver of this tile
    __threadfence();                       // acquire: the other blocks' partials
    if (tid == 0) p.counters[tile_id] = 0; // leave the counter ready for the next launch
    const float* base = p.partial + (size_t)tile_id * p.ksplit * TILE_F;
    if (p.ksplit == 2) {
      const float* other = base + (size_t)(1 - ks) * TILE_F;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] += *(const f32x4*)(other + ((i * NI + j) * NT + tid) * 4);
    } else {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          f32x4 s = *(const f32x4*)(base + ((i * NI + j) * NT + tid) * 4);
          for (int k2 = 1; k2 < p.ksplit; ++k2) s += *(const f32x4*)(base + (size_t)k2 * TILE_F + ((i * NI + j) * NT + tid) * 4);
          acc[i][j] = s;
        }
    }
  }

  // ---- epilogue.  acc[i][j][r] = C[m][n], m = row0 + wr*WM + i*16 + (lane&15),
  //                                        n = col0 + wc*WN + j*16 + (lane>>4)*4 + r.
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

template <int BM, int BN, int NWM, int NWN, int OUT_MODE, bool HAS_BIAS, int SCHED, bool SPLITK = false>
int launch(GemmArgs a, hipStream_t st) {
  constexpr int LDS = 2 * (BM + BN) * BK * 2;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<BM, BN, NWM, NWN, OUT_MODE, HAS_BIAS, SCHED, SPLITK>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  ++tasu_gemm::gemm_launches();
  TASU_LAUNCH((gemm_nt_kernel<BM, BN, NWM, NWN, OUT_MODE, HAS_BIAS, SCHED, SPLITK>),
              dim3(a.tiles_m * a.tiles_n * (SPLITK ? a.ksplit : 1)), dim3(64 * NWM * NWN), LDS, st, a);
  return TASU_OK;
}

int sched_variant() {
  static const int v = [] {
    const char* e = tasu_lab_env("TASU_GEMM_SCHED");
    return e ? atoi(e) : 0;
  }();
  return v;
}

template <int OUT_MODE, bool HAS_BIAS>
int launch_tiled(const GemmArgs& a, int bn, hipStream_t st) {
  if (bn == 256) return launch<256, 256, 2, 4, OUT_MODE, HAS_BIAS, 1>(a, st);
  if (bn == 192)
    return a.ksplit > 1 ? launch<256, 192, 2, 4, OUT_MODE, HAS_BIAS, 1, true>(a, st)
                        : launch<256, 192, 2, 4, OUT_MODE, HAS_BIAS, 1>(a, st);
  if (sched_variant() == 1)
    return bn == 96 ? launch<128, 96, 2, 2, OUT_MODE, HAS_BIAS, 1>(a, st) : launch<128, 128, 2, 2, OUT_MODE, HAS_BIAS, 1>(a, st);
  return bn == 96 ? launch<128, 96, 2, 2, OUT_MODE, HAS_BIAS, 0>(a, st) : launch<128, 128, 2, 2, OUT_MODE, HAS_BIAS, 0>(a, st);
}

// Tile choice: both configurations run 2 blocks per CU (512 slots on 256 CUs).  When the grid is at most two waves of
// blocks, the tail efficiency tiles / (waves * 512) decides (N = 1536 at M = 4096: 384 tiles of 128x128 fill 75 % of
// the slots, 512 tiles of 128x96 fill all of them: measured +11...+17 %); larger grids keep the wider tile, whose MFMA
// per LDS read ratio is better (measured: N = 8960 loses 10 % with the narrow tile).
int pick_bn(int M, int N) {
  static const int forced = [] {
    const char* e = tasu_lab_env("TASU_GEMM_BN");
    return e ? atoi(e) : 0;
  }();
  if (forced == 96 || forced == 128 || forced == 192 || forced == 256) return forced;
  // 256 x 256 (one block per CU): worth it when the grid is many rounds of 256 blocks, or (almost) exactly one round
  // (measured on MI355X, M = 4096 / 8192: gate_up +15 %, lm_head +8 %, M = 8192 x N = 1536..2048 +10...18 %;
  //  560- and 784-tile grids lose 2...3 % against the 128-wide tiles and stay there).
  const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256);
  if (t256 >= 1024 || (t256 >= 192 && t256 <= 256)) return 256;
  const long slots = 512;
  const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128), t96 = (long)((M + 127) / 128) * ((N + 95) / 96);
  const long w128 = (t128 + slots - 1) / slots, w96 = (t96 + slots - 1) / slots;
  if (w128 > 2) return 128;
  const double e128 = (double)t128 / (double)(w128 * slots), e96 = (double)t96 / (double)(w96 * slots) / 1.08;
  return e96 > e128 ? 96 : 128;
}

// 0 = heuristic, 1 = always the 128-wide 2-blocks-per-CU kernel of this file, 2 = always gemm_pipe.hip
int kernel_choice() {
  static const int v = [] {
    const char* e = tasu_lab_env("TASU_GEMM_KERNEL");
    if (!e) return 0;
    return e[0] == 'p' ? 2 : (e[0] == 'v' ? 1 : 0);
  }();
  return v;
}

}  // namespace

int tasu_gemm_pipe_dispatch(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* bias,
                            const float* resid, int M, int N, int K, int out_mode, int bn, hipStream_t st, int n0, int n1);
int tasu_gemm_pp_dispatch(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* bias, const float* resid,
                          int M, int N, int K, int out_mode, hipStream_t st, int n0, int n1, void* ws, size_t ws_bytes, double sk_rem);
namespace tasu_pp {
double sk_max_rem();
int cu_count();
}

// Split-K plan for the 256 x 192 tile: ksplit blocks per tile so that tiles * ksplit is (close to) one round of 256
// blocks, every split keeping >= 16 K-steps.  Returns 1 when the workspace is missing or too small.
static int plan_ksplit(int M, int N, int K, size_t ws_bytes) {
  static const int forced = [] {
    const char* e = tasu_lab_env("TASU_GEMM_KSPLIT");
    return e ? atoi(e) : 0;
  }();
  const long tiles = (long)((M + 255) / 256) * ((N + 191) / 192);
  int ks = forced > 0 ? forced : (int)(256 / tiles);
  const int nk = K / BK;
  if (ks > nk / 16) ks = nk / 16;
  if (ks > 8) ks = 8;
  if (ks < 1) ks = 1;
  if (tiles > TASU_GEMM_WS_COUNTERS) return 1;
  while (ks > 1 && TASU_GEMM_WS_COUNTERS * sizeof(int) + (size_t)tiles * ks * 256 * 192 * 4 > ws_bytes) --ks;
  return ks;
}

// out_mode: TASU_GEMM_OUT_* or tasu_gemm::OUT_DSWIGLU (`resid` is then the saved gate|up matrix, bf16 [M, 2N], C = dgu [M, 2N]:
// served by the gemm_pipe / gemm_pp kernels only -- kUnsupported otherwise, and tasu_gemm_dswiglu runs the two-kernel form)
constexpr int kUnsupported = -1000;
// plan (optional): the decision only -- *plan = one of TASU_GEMM_PLAN_* (include/tasu_hip.h), nothing is launched
static int gemm_policy(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* bias, const float* resid, int M,
                       int N, int K, int out_mode, void* workspace, int64_t workspace_bytes, void* stream, int* plan = nullptr) {
  const bool dsw = out_mode == tasu_gemm::OUT_DSWIGLU;
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return TASU_ERR_ARG;
  if (K % BK != 0 || lda % 8 != 0 || ldb % 8 != 0) return TASU_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)workspace & 15)) return TASU_ERR_ARG;
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
  a.tiles_m = a.tiles_n = 0;
  a.ksplit = 1;
  a.partial = nullptr;
  a.counters = nullptr;
  hipStream_t st = (hipStream_t)stream;
  const bool hb = bias != nullptr;
  const size_t ws_bytes = workspace ? (size_t)workspace_bytes : 0;
  static const int forced_bn = [] {
    const char* e = tasu_lab_env("TASU_GEMM_BN");
    return e ? atoi(e) : 0;
  }();
  // ---- kernel / tile policy (MI355X, cold weight operands as inside the training step; tools/bench_gemm.py --cold):
  //  * the pipelined kernel with loader waves (gemm_pipe.hip; tiles 256 x 128, 128 x 192, 256 x 96) is the fastest on
  //    every decoder, lm_head and projector shape (qkv 700 -> 822, gate_up 780 -> 837, d_down 690 -> 772, d_lm_head
  //    975 -> 1149 TFLOP/s ...); the tile is the one that fills whole rounds of 256 one-per-CU blocks at the least cost;
  //  * grids that cover less than half of the CUs behind K >= 16384 and too few K-tile pairs per CU for the stream-K schedule
  //    (below) split K over the 256 x 192 tiles of this file instead (256 / 512 x 1536 x 17920: 130 / 150 us against 190 on
  //    the loader-wave tiles; at K = 8960 the loader-wave tiles win: 1024 rows 94 against 162 us);
  //  * problems of at most 64 rows keep the 128-row tiles of this file (128 x 1536 x 8960: 73 us on 256 x 96 tiles, 122 here).
  static const bool pp_on = [] {
    const char* e = tasu_lab_env("TASU_GEMM_PP");
    return !(e && e[0] == '0');
  }();
  int use_pipe_bn = 0;
  if (kernel_choice() == 2) {
    use_pipe_bn = (forced_bn == 96 || forced_bn == 128 || forced_bn == 192) ? forced_bn : -1;
  } else if (kernel_choice() == 0 && forced_bn == 0 && M > 64) {
    use_pipe_bn = -1;
  }
  if (use_pipe_bn != 0) {
    const long tm = (M + 255) / 256;
    const long t128 = tm * ((N + 127) / 128), t96 = tm * ((N + 95) / 96);
    const long t256 = tm * ((N + 255) / 256);
    const int cus = tasu_pp::cu_count();
    const bool sk = pp_on && kernel_choice() == 0 && K >= 256 && K % 128 == 0 &&
                    tasu_gemm::sk_plan(t256, K / 128, cus, ws_bytes >= TASU_GEMM_WS_COUNTERS * sizeof(int) + (size_t)cus * 262144,
                                       tasu_pp::sk_max_rem()) > 0;
    if (!dsw && !sk && kernel_choice() == 0 && M > 128 && t96 < 128 && K >= 16384 && plan_ksplit(M, N, K, ws_bytes) > 1) {
      use_pipe_bn = 0;                              // falls through to the split-K tile below
    } else {
      if (use_pipe_bn < 0) {
        // time ~ rounds of one-block-per-CU grids x tile area / per-FLOP efficiency of the tile (8192^3, cold: 256 x 128
        // 1300, 128 x 192 1123, 256 x 96 ~1040 TFLOP/s).  N = 1536 at M = 4096 -> 256 tiles of 128 x 192 (+7 % over
        // 256 x 96: fewer staged bytes and fragment reads per FLOP); wide grids -> 256 x 128.
        auto cost = [&](long tiles, double area, double eff) { return (double)((tiles + 255) / 256) * area / eff; };
        const long t192 = (long)((M + 127) / 128) * ((N + 191) / 192);
        const double c128 = cost(t128, 256.0 * 128, 1.00), c192 = cost(t192, 128.0 * 192, 0.86), c96 = cost(t96, 256.0 * 96, 0.80);
        use_pipe_bn = c128 <= c192 && c128 <= c96 ? 128 : (c192 <= c96 ? 192 : 96);
        // the 256 x 256 eight-wave kernel (gemm_pp.hip): 2/3 of the L2 -> LDS bytes per FLOP of the 256 x 128 tile.  Measured on
        // whole rounds at K = 1536 (4096 x 16384: 1232 vs 995 TFLOP/s) its per-FLOP efficiency is 1.24 x that tile's, so it wins
        // wherever its coarser rounds do not eat that up (gate|up, lm_head; d_down's 3 rounds against 5: a tie on paper, +0.5 %
        // on the step measured with TASU_GEMM_PP_EFF = 1.26 against 1.19 on one box; not the one-round N = 1536 grids)
        if (pp_on && kernel_choice() == 0 && K >= 256 && K % 128 == 0) {
          static const double pp_eff = [] {            // TASU_GEMM_PP_EFF: tuning runs
            const char* e = tasu_lab_env("TASU_GEMM_PP_EFF");
            return e ? atof(e) : 1.26;
          }();
          // stream-K (gemm_pp.hip; needs the workspace): the 256 x 256 tiles fill FRACTIONAL rounds -- every workgroup gets the
          // same number of K-tile pairs -- for the price of the partial tiles' round trip (~35 us per launch whatever K is:
          // 1e8 / K in the units of this model).  That serves d_gate_up (96 tiles on 256 CUs, K = 17920: 203 -> 181 us); at
          // K = 8960 (down) the 128 x 192 one-round grid still wins (100 vs 108 us).
          const double c256_whole = cost(t256, 256.0 * 256, pp_eff);
          const double c256_sk = sk ? (double)t256 / cus * 256.0 * 256 / pp_eff + 1.0e8 / K : 1e30;
          const double c256 = c256_sk < c256_whole ? c256_sk : c256_whole;
          const double best = use_pipe_bn == 128 ? c128 : (use_pipe_bn == 192 ? c192 : c96);
          if (c256 < best) {
            if (c256_sk < c256_whole) {
              if (plan) return *plan = TASU_GEMM_PLAN_PP256_STREAMK, TASU_OK;
              return tasu_gemm_pp_dispatch(A, lda, B, ldb, C, ldc, bias, resid, M, N, K, out_mode, st, 0, 0, workspace, ws_bytes, -2.0);
            }
            // a mostly empty last round of big tiles (d_down: 560 tiles = 2.19 rounds): whole rounds on the big tiles, the
            // remaining columns on the small tiles in a second launch (TASU_GEMM_NSPLIT=0 disables)
            static const bool split_on = [] {
              const char* e = tasu_lab_env("TASU_GEMM_NSPLIT");
              return !(e && e[0] == '0');
            }();
            const long tn = (N + 255) / 256, full = (tm * tn) / 256, tn_main = full * 256 / tm;
            if (split_on && full >= 1 && tn_main > 0 && tn_main < tn) {
              const int n_main = (int)tn_main * 256, n_tail = N - n_main;
              const long u128 = tm * ((n_tail + 127) / 128), u192 = (long)((M + 127) / 128) * ((n_tail + 191) / 192);
              const double t128 = cost(u128, 256.0 * 128, 1.00), t192 = cost(u192, 128.0 * 192, 0.86);
              const double c_split = (double)full * 256.0 * 256 / pp_eff + (t128 < t192 ? t128 : t192) + 0.05 * 256.0 * 256;
              if (c_split < c256) {
                if (plan) return *plan = t128 < t192 ? TASU_GEMM_PLAN_PP256_PLUS_PIPE128 : TASU_GEMM_PLAN_PP256_PLUS_PIPE192, TASU_OK;
                const int rc = tasu_gemm_pp_dispatch(A, lda, B, ldb, C, ldc, bias, resid, M, N, K, out_mode, st, 0, n_main, nullptr, 0, -2.0);
                if (rc) return rc;
                return tasu_gemm_pipe_dispatch(A, lda, B, ldb, C, ldc, bias, resid, M, N, K, out_mode, t128 < t192 ? 128 : 192, st,
                                               n_main, 0);
              }
            }
            if (plan) return *plan = TASU_GEMM_PLAN_PP256, TASU_OK;
            return tasu_gemm_pp_dispatch(A, lda, B, ldb, C, ldc, bias, resid, M, N, K, out_mode, st, 0, 0, nullptr, 0, -2.0);
          }
        }
      }
      if (plan) return *plan = use_pipe_bn == 128 ? TASU_GEMM_PLAN_PIPE128 : (use_pipe_bn == 192 ? TASU_GEMM_PLAN_PIPE192 : TASU_GEMM_PLAN_PIPE96), TASU_OK;
      return tasu_gemm_pipe_dispatch(A, lda, B, ldb, C, ldc, bias, resid, M, N, K, out_mode, use_pipe_bn, st, 0, 0);
    }
  }
  if (dsw) return kUnsupported;
  int bn;
  if (kernel_choice() == 0 && forced_bn == 0 && M > 128) {
    bn = 192;                                       // the deep small-grid case above
  } else {
    bn = pick_bn(M, N);
  }
  if (bn == 192) {
    a.ksplit = plan_ksplit(M, N, K, ws_bytes);
    if (a.ksplit > 1) {
      a.counters = (int*)workspace;
      a.partial = (float*)((char*)workspace + TASU_GEMM_WS_COUNTERS * sizeof(int));
    }
  }
  if (plan) return *plan = a.ksplit > 1 ? TASU_GEMM_PLAN_TILE192_SPLITK : TASU_GEMM_PLAN_TILES, TASU_OK;
  switch (out_mode) {
    case TASU_GEMM_OUT_BF16:
      return hb ? launch_tiled<TASU_GEMM_OUT_BF16, true>(a, bn, st) : launch_tiled<TASU_GEMM_OUT_BF16, false>(a, bn, st);
    case TASU_GEMM_OUT_F32:
      return hb ? launch_tiled<TASU_GEMM_OUT_F32, true>(a, bn, st) : launch_tiled<TASU_GEMM_OUT_F32, false>(a, bn, st);
    case TASU_GEMM_OUT_F32_RESID_BF16R:
      return hb ? launch_tiled<TASU_GEMM_OUT_F32_RESID_BF16R, true>(a, bn, st)
                : launch_tiled<TASU_GEMM_OUT_F32_RESID_BF16R, false>(a, bn, st);
    default:
      return TASU_ERR_ARG;
  }
}

namespace tasu_gemm {
long long& gemm_launches() {
  static long long n = 0;
  return n;
}
int& relu_next() {
  static thread_local int flag = 0;
  return flag;
}
int& act_ld_next() {
  static int v = 0;
  return v;
}
}  // namespace tasu_gemm

extern "C" int64_t tasu_gemm_launch_count(void) { return (int64_t)tasu_gemm::gemm_launches(); }

extern "C" int tasu_relu_fwd(const void* x, void* y, int64_t n, void* stream);
// C = bf16(relu(bf16(A . B^T + bias))): PositionwiseFeedForward's w_1 + ReLU (SenseVoice.py:71-73) in one launch -- the ReLU in
// the GEMM kernels' epilogue (max before the one bf16 rounding: the same bits as rounding first); shapes the small-tile kernels
// of gemm.hip serve (at most 128 rows) run the GEMM and tasu_relu_fwd in place.
extern "C" int tasu_gemm_bias_relu_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* bias, int M, int N,
                                        int K, void* workspace, int64_t workspace_bytes, void* stream) {
  int plan = 0;
  int rc = gemm_policy(A, lda, B, ldb, C, ldc, bias, nullptr, M, N, K, TASU_GEMM_OUT_BF16, workspace, workspace_bytes, stream, &plan);
  if (rc) return rc;
  const bool fused = plan != TASU_GEMM_PLAN_TILES && plan != TASU_GEMM_PLAN_TILE192_SPLITK;
  tasu_gemm::relu_next() = fused ? 1 : 0;
  rc = gemm_policy(A, lda, B, ldb, C, ldc, bias, nullptr, M, N, K, TASU_GEMM_OUT_BF16, workspace, workspace_bytes, stream);
  tasu_gemm::relu_next() = 0;
  if (rc || fused) return rc;
  if (ldc != N) return TASU_ERR_ARG;                       // (the in-place ReLU walks a dense matrix)
  return tasu_relu_fwd(C, C, (int64_t)M * N, stream);
}

extern "C" int tasu_gemm_nt_bf16_ws(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* bias,
                                    const float* resid, int M, int N, int K, int out_mode, void* workspace,
                                    int64_t workspace_bytes, void* stream) {
  if (out_mode != TASU_GEMM_OUT_BF16 && out_mode != TASU_GEMM_OUT_F32 && out_mode != TASU_GEMM_OUT_F32_RESID_BF16R) return TASU_ERR_ARG;
  return gemm_policy(A, lda, B, ldb, C, ldc, bias, resid, M, N, K, out_mode, workspace, workspace_bytes, stream);
}

// The dispatcher's decision for a problem, without launching anything (no GPU needed; include/tasu_hip.h)
extern "C" int tasu_gemm_plan(int M, int N, int K, int out_mode, int with_workspace) {
  if (out_mode != TASU_GEMM_OUT_BF16 && out_mode != TASU_GEMM_OUT_F32 && out_mode != TASU_GEMM_OUT_F32_RESID_BF16R) return -1;
  alignas(16) static char dummy[16];
  int plan = -1;
  const int rc = gemm_policy(dummy, K, dummy, K, dummy, N, nullptr, (const float*)dummy, M, N, K, out_mode, with_workspace ? dummy : nullptr,
                             with_workspace ? ((int64_t)64 << 20) + 16384 : 0, nullptr, &plan);
  return rc == TASU_OK ? plan : -1;
}

extern "C" int tasu_swiglu_bwd(const void* dact, const void* gu, void* dgu, int M, int I, void* stream);

// Down projection's input gradient with the SwiGLU backward in the GEMM's epilogue (include/tasu_hip.h)
extern "C" int tasu_gemm_dswiglu(const void* dy, int lddy, const void* WdT, int ldw, const void* gu, void* dgu, void* dact_ws, int M,
                                 int I, int K, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!dy || !WdT || !gu || !dgu || M <= 0 || I <= 0 || I % 8 || K <= 0 || K % BK || lddy % 8 || ldw % 8) return TASU_ERR_ARG;
  if (((uintptr_t)dy & 15) || ((uintptr_t)WdT & 15) || ((uintptr_t)gu & 15) || ((uintptr_t)dgu & 15)) return TASU_ERR_ARG;
  // The form with the SwiGLU backward in the GEMM's epilogue (dact never reaches memory) exists in the LAB build only
  // (TASU_GEMM_DSWIGLU=1 there): measured in the step it was worth 0.09 ms of 29 -- the epilogue's gate|up reads and dgu writes
  // (294 MB per call) run while the CU's MFMA pipes idle, with every CU in its epilogue at the same time, so the GEMM grows by
  // what the separate kernel took (55 us at 6.6 TB/s).  It pays once an epilogue runs under the next tile's K loop.
#ifdef TASU_LAB
  const char* const fused_env = tasu_lab_env("TASU_GEMM_DSWIGLU");   // (read per call: A/B runs in one process)
  if (fused_env && fused_env[0] == '1') {
    const int rc = gemm_policy(dy, lddy, WdT, ldw, dgu, 2 * I, nullptr, (const float*)gu, M, I, K, tasu_gemm::OUT_DSWIGLU, workspace,
                               workspace_bytes, stream);
    if (rc != kUnsupported) return rc;
  }
#endif
  if (!dact_ws || ((uintptr_t)dact_ws & 15)) return TASU_ERR_ARG;
  const int rc = gemm_policy(dy, lddy, WdT, ldw, dact_ws, I, nullptr, nullptr, M, I, K, TASU_GEMM_OUT_BF16, workspace, workspace_bytes,
                             stream);
  return rc ? rc : tasu_swiglu_bwd(dact_ws, gu, dgu, M, I, stream);
}

extern "C" int tasu_gemm_nt_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* bias,
                                 const float* resid, int M, int N, int K, int out_mode, void* stream) {
  return tasu_gemm_nt_bf16_ws(A, lda, B, ldb, C, ldc, bias, resid, M, N, K, out_mode, nullptr, 0, stream);
}
