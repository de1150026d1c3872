// Weight-streaming bf16 NT GEMM for the decode step (M <= 64 rows = batch x beams), second generation:
//     C[M, N] = A[M, K] . W[N, K]^T     with the epilogue of the consuming op fused and NO second launch.
//
// A decode step streams every weight byte once (HBM-bound), but its GEMMs are so short (6 MB of weights for the q|k|v and o
// projections) that what a layer costs is launches and their tails.  gemm_skinny.hip fills the chip by splitting K over
// workgroups and pays a finish launch per GEMM (bias / RoPE / residual need the complete sum): 2 x ~8 us.  Here:
//   * ACTIVATIONS LIVE IN REGISTERS.  A workgroup is 8 waves; wave w owns the K slice [w, w+1) * K/8 of the workgroup's K
//     range and loads its slice of all 64 activation rows ONCE, as MFMA operands (KS k-steps x 4 row tiles x 16 B per lane).
//   * WEIGHTS GO GLOBAL -> REGISTERS, no LDS staging: a lane's 16-byte load is exactly its MFMA operand
//     (mfma_f32_16x16x32_bf16: weight row l&15, 8 k-values of lane group l>>4), three column tiles in flight per wave
//     (3 x KS KiB), counted by the compiler's own vmcnt bookkeeping on an unrolled-by-3 register ring.
//   * a workgroup walks its column tiles persistently; per tile the 8 waves' partial 16(32) x 64 results meet in LDS
//     (double-buffered: ONE raw s_barrier per tile, the loads of the next tiles stay in flight across it), waves 0..3 sum
//     them in wave order (deterministic) and run the epilogue:
//        E_BF16    C = bf16(sum + bias)                                  (lm_head)
//        E_RESID   C(fp32) = R + bf16(sum)                               (o projection + residual add)
//        E_SWIGLU  tile = 8 gate + 8 up rows of the same columns:  act = bf16(bf16(silu(g)) * u)      (MLP in)
//        E_QKV     tile = 8 + 8 columns (j, j + 64) of one head: bias, RoPE, rotated row to qkv[m] and k / v appended to
//                  the cache at pos[m]   (q|k|v projection + tasu_rope_append)
//        E_SLAB    fp32 partial tile into slab[blockIdx.y] (K split over workgroups for K too long for registers: the down
//                  projection, K = 8960 = 5 x 1792; skinny_reduce_norm finishes it together with residual and next norm)
//     -- the same arithmetic and rounding points as gemm_skinny.hip's kernels + finish kernels.
// K range per workgroup = 8 waves x KS x 32 (KS = 1, 2, 6, 7: 256, 512, 1536, 1792); anything else stays on gemm_skinny.
//
// FRAGMENT-ORDER OPERANDS.  A lane's MFMA operand is 8 k-values of ONE matrix row, and the 16 lanes of a lane group hold 16
// different rows: read straight from a row-major matrix, one wave instruction touches 16 rows x 64 B -- half-used cache lines,
// 16 of them per 16-lane group (measured: the q|k|v projection spent ~10 of its 13 us loading its 196 KB of activations this
// way).  Both operands therefore also exist in the order the MFMA consumes them, in which every wave instruction reads 1 KiB
// contiguous:   X_f[k-step c = k / 32][row tile t][lane = 16 * ((k % 32) / 8) + row % 16][k % 8]
//   * weights are static: tasu_to_fragment_order re-lays them out once at load time, per column tile in the row order of the
//     tile's epilogue (w_frag = 1: W points at [tile][K / 32][64 lanes][8]);
//   * activations [<= 64, K] are written in this order by their producers -- the SwiGLU epilogue here (out_frag), the norm
//     kernels (tasu_rmsnorm_fwd_frag, tasu_stream_finish_norm) and the cache attention (tasu_attn_decode_frag) -- and read
//     with a_frag = 1: [K / 32][4 row tiles][64 lanes][8].
// Row-major operands (a_frag = w_frag = 0) remain supported (tests, first use before the layouts are registered).
#include "common.h"
#include "../../include/tasu_hip.h"

namespace tasu_stream {

enum { E_BF16 = 0, E_RESID = 1, E_SWIGLU = 2, E_QKV = 3, E_SLAB = 4 };
constexpr int NW = 8;                      // waves per workgroup

struct Args {
  const bf16* A;          // [M, lda] activations
  const bf16* W;          // [*, ldw] weights, K contiguous
  void* C;                // output (bf16 or fp32, see epilogues); E_SLAB: fp32 slabs [ksplit][tiles][16 x 64]
  const float* R;         // E_RESID: residual [M, ldc] fp32
  const bf16* bias;       // [N] or null
  int M, N, K, lda, ldw, ldc;
  int tiles;              // column tiles per K range
  int I;                  // E_SWIGLU: first "up" row of W
  int a_frag, w_frag;     // operands in fragment order (see the header comment)
  int out_frag;           // E_SWIGLU: act is written in fragment order (K of its consumer = N)
  // E_QKV
  int H, G, ctx;
  const float* cos_t;
  const float* sin_t;
  bf16* kc;
  bf16* vc;
  const int32_t* pos;
};

// first weight row (of 16) that lane group row r = l & 15 of tile t reads
template <int EPI>
__device__ __forceinline__ int weight_row(const Args& p, int t, int r) {
  if (EPI == E_SWIGLU) return (r < 8 ? 0 : p.I) + t * 8 + (r & 7);
  if (EPI == E_QKV) {
    const int rot_tiles = (p.H + p.G) * 8;                 // q and k heads: 8 tiles of (8 + 8) paired columns each
    if (t < rot_tiles) return (t >> 3) * 128 + (t & 7) * 8 + (r & 7) + (r >= 8 ? 64 : 0);
    return (p.H + p.G) * 128 + (t - rot_tiles) * 16 + r;   // v heads: 16 plain columns
  }
  return t * 16 + r;
}

// MT = 16-row tiles of the activations a workgroup owns (4 = all 64 rows; 2 / 1: the rows are split over blockIdx.z, for GEMMs
// with too few column tiles to occupy the chip -- a workgroup's traffic is its (MT * 16 + columns) x K operand bytes, and with
// 1-2 column tiles per workgroup the 64 activation rows dominate it).
// FRAG: both operands in fragment order (compile-time: a run-time layout test inside the load lambdas splits the ring loop
// into branches across which the compiler drains vmcnt).
template <int KS, int EPI, int MT, bool FRAG>
__global__ __launch_bounds__(64 * NW, 2) void stream_gemm_kernel(Args p) {
  // partial tiles: [2 buffers][NW waves][MT row tiles][64 lanes] f32x4
  __shared__ __attribute__((aligned(16))) float red[2][NW][MT][64][4];
  const int mt0 = (int)blockIdx.z * MT;                 // first row tile of this workgroup
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int krange = NW * KS * 32;
  const int k0 = blockIdx.y * krange + wave * (KS * 32) + lq * 8;      // this lane's first k of step 0

  // ---- activations: this wave's K slice of all 64 rows, as MFMA B operands (rows beyond M are clamped; masked at the store)
  bf16x8 a[MT][KS];
  const int cg0 = ((int)blockIdx.y * NW + wave) * KS;                   // this wave's first global k-step
  if (FRAG) {
#pragma unroll
    for (int c = 0; c < KS; ++c)
#pragma unroll
      for (int t = 0; t < MT; ++t) a[t][c] = *(const bf16x8*)(p.A + (((size_t)(cg0 + c) * 4 + mt0 + t) * 64 + lane) * 8);
  } else {
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const bf16* ar = p.A + (size_t)min((mt0 + t) * 16 + l15, p.M - 1) * p.lda + k0;
#pragma unroll
      for (int c = 0; c < KS; ++c) a[t][c] = *(const bf16x8*)(ar + c * 32);
    }
  }

  const int ntl = (p.tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;     // tiles this workgroup walks
  auto tile_of = [&](int i) { return (int)blockIdx.x + min(i, ntl - 1) * (int)gridDim.x; };   // clamped: loads past the end re-read
  const int ksteps_all = p.K >> 5;
  auto load_w = [&](bf16x8 (&w)[KS], int i) {
    if (FRAG) {
      const bf16* wr = p.W + (((size_t)tile_of(i) * ksteps_all + cg0) * 64 + lane) * 8;
#pragma unroll
      for (int c = 0; c < KS; ++c) w[c] = __builtin_nontemporal_load((const bf16x8*)(wr + c * 512));
    } else {
      const bf16* wr = p.W + (size_t)min(weight_row<EPI>(p, tile_of(i), l15), (EPI == E_SWIGLU ? 2 * p.I : p.N) - 1) * p.ldw + k0;
#pragma unroll
      for (int c = 0; c < KS; ++c) w[c] = __builtin_nontemporal_load((const bf16x8*)(wr + c * 32));
    }
  };

  auto finish = [&](int i) {
    // ---- cross-wave sum + epilogue of tile i (waves 0..MT-1: row tile = mt0 + wave); called after the barrier of tile i
    if (wave >= MT || i >= ntl) return;
    const int buf = i & 1, t = tile_of(i);
    f32x4 s = *(const f32x4*)red[buf][0][wave][lane];
#pragma unroll
    for (int w2 = 1; w2 < NW; ++w2) s += *(const f32x4*)red[buf][w2][wave][lane];
    const int rt = mt0 + wave;                           // row tile (0..3) of this wave's results
    const int m = rt * 16 + l15;
    // s[r] = C[m][tile column 4 * lq + r]
    if (EPI == E_SLAB) {
      float* slab = (float*)p.C + ((size_t)blockIdx.y * p.tiles + t) * 1024;     // [16 columns][64 rows]: fragment order
      *(f32x4*)(slab + (rt * 64 + lane) * 4) = s;
      return;
    }
    if (EPI == E_SWIGLU) {
      // lanes lq < 2 hold gate columns t*8 + 4*lq + r, lanes lq + 2 the up values of the same columns
      f32x4 u;
#pragma unroll
      for (int r = 0; r < 4; ++r) u[r] = __shfl_xor(s[r], 32, 64);
      if (lq < 2 && (m < p.M || p.out_frag)) {
        const int n = t * 8 + 4 * lq;
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16)(bf16_round(silu_f(bf16_round(s[r]))) * bf16_round(u[r]));
        if (p.out_frag) {
          // element (m, n .. n+3) of the consumer's A operand: k-step n / 32, lane group (n % 32) / 8, row tile = wave
          *(bf16x4*)((bf16*)p.C + ((((size_t)(n >> 5) * 4 + rt) * 64 + ((n & 31) >> 3) * 16 + l15) << 3) + (n & 7)) = o;
        } else {
          bf16* dst = (bf16*)p.C + (size_t)m * p.ldc + n;
          if (n + 4 <= p.N) {
            *(bf16x4*)dst = o;
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (n + r < p.N) dst[r] = o[r];
          }
        }
      }
      return;
    }
    if (EPI == E_QKV) {
      const int rot_tiles = (p.H + p.G) * 8;
      const int W = p.G * 128;
      bf16* out = (bf16*)p.C + (size_t)m * p.ldc;
      if (t < rot_tiles) {
        // lanes lq < 2: low-half columns c0 + 4*lq + r of head hh; lanes lq + 2: their partners (+64)
        const int hh = t >> 3, c0 = (t & 7) * 8 + 4 * (lq & 1);
        const int col = hh * 128 + c0 + (lq >= 2 ? 64 : 0);
        bf16x4 mine;
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[r] = (bf16)(s[r] + (p.bias ? (float)p.bias[col + r] : 0.f));
        f32x4 x1, x2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float other = __shfl_xor((float)mine[r], 32, 64);
          x1[r] = lq < 2 ? (float)mine[r] : other;       // low half
          x2[r] = lq < 2 ? other : (float)mine[r];       // high half
        }
        if (m < p.M) {
          const f32x4 cs = *(const f32x4*)(p.cos_t + (size_t)m * 64 + c0), sn = *(const f32x4*)(p.sin_t + (size_t)m * 64 + c0);
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16)(lq < 2 ? x1[r] * cs[r] - x2[r] * sn[r] : x2[r] * cs[r] + x1[r] * sn[r]);
          *(bf16x4*)(out + col) = o;
          if (hh >= p.H) {
            const size_t slot = ((size_t)m * p.ctx + p.pos[m]) * W;
            *(bf16x4*)(p.kc + slot + (hh - p.H) * 128 + c0 + (lq >= 2 ? 64 : 0)) = o;
          }
        }
      } else if (m < p.M) {
        const int c = (t - rot_tiles) * 16 + 4 * lq;           // column inside the v block
        const int col = (p.H + p.G) * 128 + c;
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16)(s[r] + (p.bias ? (float)p.bias[col + r] : 0.f));
        *(bf16x4*)(out + col) = o;
        *(bf16x4*)(p.vc + ((size_t)m * p.ctx + p.pos[m]) * W + c) = o;
      }
      return;
    }
    // E_BF16 / E_RESID
    const int n = t * 16 + 4 * lq;
    if (m >= p.M || n >= p.N) return;
    if (EPI == E_RESID) {
      float* dst = (float*)p.C + (size_t)m * p.ldc + n;
      const float* rs = p.R + (size_t)m * p.ldc + n;
      if (n + 4 <= p.N) {
        const f32x4 old = *(const f32x4*)rs;
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = old[r] + bf16_round(s[r]);
        *(f32x4*)dst = o;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < p.N) dst[r] = rs[r] + bf16_round(s[r]);
      }
    } else {
      bf16* dst = (bf16*)p.C + (size_t)m * p.ldc + n;
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (bf16)(s[r] + (p.bias && n + r < p.N ? (float)p.bias[n + r] : 0.f));
      if (n + 4 <= p.N) {
        *(bf16x4*)dst = o;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < p.N) dst[r] = o[r];
      }
    }
  };

  auto compute = [&](const bf16x8 (&w)[KS], int i) {
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < KS; ++c)
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t] = mfma16(w[c], a[t][c], acc[t]);
    const int buf = i & 1;
#pragma unroll
    for (int t = 0; t < MT; ++t) *(f32x4*)red[buf][wave][t][lane] = acc[t];
    // my partial tile is in LDS; everybody's is after the barrier.  Raw s_barrier: the weight loads of the next tiles stay
    // in flight across it (a __syncthreads() would drain vmcnt).  Buffer (i & 1) is written again at tile i + 2, which every
    // wave reaches only after the barrier of tile i + 1, i.e. after all reads of tile i.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    finish(i);
  };

  bf16x8 w0[KS], w1[KS], w2[KS];
  load_w(w0, 0);
  if (ntl <= 2) {
    // one or two tiles (the q|k|v, o and down projections): no ring, nothing loaded twice
    if (ntl == 2) load_w(w1, 1);
    compute(w0, 0);
    if (ntl == 2) compute(w1, 1);
    return;
  }
  // three tiles in flight per wave.  The body is branch-free (the trip count is rounded up to a multiple of three: the spare
  // bodies re-read the last tile and skip their epilogue), so that the compiler's vmcnt bookkeeping sees one straight ring
  // and waits for the oldest tile only.
  load_w(w1, 1);
  for (int i = 0; i < ntl; i += 3) {
    load_w(w2, i + 2);
    compute(w0, i);
    load_w(w0, i + 3);
    compute(w1, i + 1);
    load_w(w1, i + 4);
    compute(w2, i + 2);
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

template <int EPI, int MT, bool FRAG>
int launch_ks(const Args& a, int ksplit, dim3 grid, hipStream_t st) {
  const int ks = a.K / ksplit / (NW * 32);
  switch (ks) {
    case 1: TASU_LAUNCH((stream_gemm_kernel<1, EPI, MT, FRAG>), grid, dim3(64 * NW), 0, st, a); return TASU_OK;
    case 2: TASU_LAUNCH((stream_gemm_kernel<2, EPI, MT, FRAG>), grid, dim3(64 * NW), 0, st, a); return TASU_OK;
    case 6: TASU_LAUNCH((stream_gemm_kernel<6, EPI, MT, FRAG>), grid, dim3(64 * NW), 0, st, a); return TASU_OK;
    case 7: TASU_LAUNCH((stream_gemm_kernel<7, EPI, MT, FRAG>), grid, dim3(64 * NW), 0, st, a); return TASU_OK;
    default: return TASU_ERR_ARG;
  }
}
template <int EPI, int MT>
int launch_mt(const Args& a, int ksplit, dim3 grid, hipStream_t st) {
  if (a.a_frag != a.w_frag) return TASU_ERR_ARG;         // the operands travel in fragment order together or not at all
  return a.a_frag ? launch_ks<EPI, MT, true>(a, ksplit, grid, st) : launch_ks<EPI, MT, false>(a, ksplit, grid, st);
}

template <int EPI>
int launch(const Args& a, int ksplit, hipStream_t st) {
  // Rows per workgroup: all 64 when the column tiles alone give every CU work; otherwise the row tiles are split over
  // blockIdx.z (2 x 32 rows) so that twice as many workgroups each load half of the activations.
  const int cus = cu_count();
  const int row_tiles = (a.M + 15) / 16;
  // (K-range slabs: always -- 7 k-steps x 4 row tiles of activations per wave would not leave registers for the weight ring)
  const bool split_rows = row_tiles > 2 && (EPI == E_SLAB || a.tiles * ksplit * 2 <= cus + cus / 4);
  const int zs = split_rows ? 2 : 1;
  const int per_split = cus / (ksplit * zs) > 0 ? cus / (ksplit * zs) : 1;
  const dim3 grid(a.tiles < per_split ? a.tiles : per_split, ksplit, zs);
  return split_rows ? launch_mt<EPI, 2>(a, ksplit, grid, st) : launch_mt<EPI, 4>(a, ksplit, grid, st);
}

bool k_supported(int K, int ksplit) {
  if (ksplit < 1 || K % ksplit) return false;
  const int kr = K / ksplit;
  return kr == 256 || kr == 512 || kr == 1536 || kr == 1792;
}

}  // namespace tasu_stream

using tasu_stream::Args;

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" int tasu_stream_supported(int K, int ksplit) { return tasu_stream::k_supported(K, ksplit) ? 1 : 0; }

extern "C" int tasu_gemm_stream_bf16(const void* A, int lda, const void* W, int ldw, void* C, int ldc, const void* bias,
                                     const float* resid, int M, int N, int K, int out_mode, int a_frag, int w_frag, void* stream) {
  using namespace tasu_stream;
  if (!A || !W || !C || M <= 0 || M > 64 || N <= 0 || !k_supported(K, 1) || lda % 8 || ldw % 8) return TASU_ERR_ARG;
  if (!aligned16(A) || !aligned16(W)) return TASU_ERR_ARG;
  if (out_mode != TASU_GEMM_OUT_BF16 && out_mode != TASU_GEMM_OUT_F32_RESID_BF16R) return TASU_ERR_ARG;
  if (out_mode == TASU_GEMM_OUT_F32_RESID_BF16R && (!resid || bias || ldc % 4 || !aligned16(C) || !aligned16(resid))) return TASU_ERR_ARG;
  if (out_mode == TASU_GEMM_OUT_BF16 && (ldc % 4 || ((uintptr_t)C & 7))) return TASU_ERR_ARG;
  Args a{};
  a.A = (const bf16*)A;
  a.W = (const bf16*)W;
  a.C = C;
  a.R = resid;
  a.bias = (const bf16*)bias;
  a.M = M, a.N = N, a.K = K, a.lda = lda, a.ldw = ldw, a.ldc = ldc;
  a.tiles = (N + 15) / 16;
  a.a_frag = a_frag, a.w_frag = w_frag;
  return out_mode == TASU_GEMM_OUT_BF16 ? launch<E_BF16>(a, 1, (hipStream_t)stream) : launch<E_RESID>(a, 1, (hipStream_t)stream);
}

extern "C" int tasu_gemm_stream_swiglu(const void* A, int lda, const void* Wgu, int ldw, void* act, int ldact, int M, int I, int K,
                                       int a_frag, int w_frag, int out_frag, void* stream) {
  using namespace tasu_stream;
  if (!A || !Wgu || !act || M <= 0 || M > 64 || I <= 0 || I % 8 || !k_supported(K, 1) || lda % 8 || ldw % 8 || ldact % 4)
    return TASU_ERR_ARG;
  if (!aligned16(A) || !aligned16(Wgu) || ((uintptr_t)act & 7)) return TASU_ERR_ARG;
  Args a{};
  a.A = (const bf16*)A;
  a.W = (const bf16*)Wgu;
  a.C = act;
  a.M = M, a.N = I, a.K = K, a.lda = lda, a.ldw = ldw, a.ldc = ldact;
  a.I = I;
  a.tiles = I / 8;
  a.a_frag = a_frag, a.w_frag = w_frag, a.out_frag = out_frag;
  if (out_frag && I % 32) return TASU_ERR_ARG;
  return launch<E_SWIGLU>(a, 1, (hipStream_t)stream);
}

extern "C" int tasu_gemm_stream_qkv_rope(const void* A, int lda, const void* Wqkv, int ldw, const void* bias, void* qkv, int M,
                                         int H, int G, int K, const float* cos_tab, const float* sin_tab, void* kcache,
                                         void* vcache, const int32_t* pos, int ctx, int a_frag, int w_frag, void* stream) {
  using namespace tasu_stream;
  if (!A || !Wqkv || !qkv || !cos_tab || !sin_tab || !kcache || !vcache || !pos || M <= 0 || M > 64 || H <= 0 || G <= 0 ||
      !k_supported(K, 1) || lda % 8 || ldw % 8 || ctx <= 0)
    return TASU_ERR_ARG;
  if (!aligned16(A) || !aligned16(Wqkv) || !aligned16(qkv) || !aligned16(cos_tab) || !aligned16(sin_tab) || !aligned16(kcache) ||
      !aligned16(vcache))
    return TASU_ERR_ARG;
  Args a{};
  a.A = (const bf16*)A;
  a.W = (const bf16*)Wqkv;
  a.C = qkv;
  a.bias = (const bf16*)bias;
  a.M = M, a.N = (H + 2 * G) * 128, a.K = K, a.lda = lda, a.ldw = ldw, a.ldc = a.N;
  a.H = H, a.G = G, a.ctx = ctx;
  a.cos_t = cos_tab, a.sin_t = sin_tab;
  a.kc = (bf16*)kcache, a.vc = (bf16*)vcache, a.pos = pos;
  a.tiles = (H + 2 * G) * 8;
  a.a_frag = a_frag, a.w_frag = w_frag;
  return launch<E_QKV>(a, 1, (hipStream_t)stream);
}

// K split over workgroups: fp32 partial tiles [ksplit][N/16][16 x 64] in `slabs` (fragment order: element ((w * 64 + l) * 4 + r)
// of a tile = C[m = 16 w + (l & 15)][column 4 (l >> 4) + r]); tasu_stream_finish_norm sums them.
extern "C" int tasu_gemm_stream_slabs(const void* A, int lda, const void* W, int ldw, float* slabs, int64_t slab_floats, int M,
                                      int N, int K, int ksplit, int a_frag, int w_frag, void* stream) {
  using namespace tasu_stream;
  if (!A || !W || !slabs || M <= 0 || M > 64 || N <= 0 || N % 16 || !k_supported(K, ksplit) || lda % 8 || ldw % 8) return TASU_ERR_ARG;
  if (!aligned16(A) || !aligned16(W) || !aligned16(slabs)) return TASU_ERR_ARG;
  if ((int64_t)ksplit * (N / 16) * 1024 > slab_floats) return TASU_ERR_ARG;
  Args a{};
  a.A = (const bf16*)A;
  a.W = (const bf16*)W;
  a.C = slabs;
  a.M = M, a.N = N, a.K = K, a.lda = lda, a.ldw = ldw, a.ldc = N;
  a.tiles = N / 16;
  a.a_frag = a_frag, a.w_frag = w_frag;
  return launch<E_SLAB>(a, ksplit, (hipStream_t)stream);
}

namespace tasu_stream {
// Row-wise finish of tasu_gemm_stream_slabs for a projection that feeds an RMSNorm (the down projection of a decode layer):
// block = row m;  C[m, :] = R[m, :] + bf16(sum of the slabs in slab order);  y[m, :] = bf16(w * (C[m, :] * rstd)) -- the
// arithmetic of gemm_skinny.hip's skinny_reduce_norm_kernel on this kernel's slab layout.  N % 16 == 0.
__global__ __launch_bounds__(256) void stream_finish_norm_kernel(const float* __restrict__ slabs, int ksplit, int tiles,
                                                                 float* __restrict__ C, const float* __restrict__ R, int N,
                                                                 const float* __restrict__ nw, bf16* __restrict__ y, float eps,
                                                                 int y_frag) {
  __shared__ float red[4];
  const int m = blockIdx.x;
  const int row_tile = m >> 4, l15 = m & 15;
  float* crow = C + (size_t)m * N;
  const float* rrow = R + (size_t)m * N;
  float ss = 0.f;
  for (int g = threadIdx.x; g < N / 4; g += 256) {
    const int n = g * 4, t = n >> 4, lq = (n & 15) >> 2;
    const size_t e = (size_t)t * 1024 + ((row_tile * 64 + lq * 16 + l15) << 2);
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int k = 0; k < ksplit; ++k) s += *(const f32x4*)(slabs + (size_t)k * tiles * 1024 + e);
    const f32x4 r = *(const f32x4*)(rrow + n);
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = r[q] + bf16_round(s[q]);
    *(f32x4*)(crow + n) = v;
    ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  ss = block_sum<4>(ss, red);
  const float rs = rsqrtf(ss / (float)N + eps);
  for (int g = threadIdx.x; g < N / 4; g += 256) {
    const int n = g * 4;
    const f32x4 v = *(const f32x4*)(crow + n);     // this thread's own store above
    const f32x4 w = *(const f32x4*)(nw + n);
    f32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = w[q] * (v[q] * rs);
    bf16* dst = y_frag ? y + ((((size_t)(n >> 5) * 4 + row_tile) * 64 + ((n & 31) >> 3) * 16 + l15) << 3) + (n & 7)
                       : y + (size_t)m * N + n;
    *(bf16x4*)dst = __builtin_convertvector(o, bf16x4);
  }
}

// Re-lays a row-major weight matrix out in fragment order, one 16-row column tile at a time in the row order of the tile's
// epilogue (kind: E_BF16 plain, E_SWIGLU 8 gate + 8 up rows, E_QKV paired RoPE columns).  Load-time work.
template <int EPI>
__global__ __launch_bounds__(256) void to_fragment_order_kernel(Args p, bf16* __restrict__ out) {
  const int t = blockIdx.x;
  const int ksteps = p.K >> 5;
  const int limit = (EPI == E_SWIGLU ? 2 * p.I : p.N) - 1;
  for (int e = threadIdx.x; e < ksteps * 64; e += 256) {
    const int c = e >> 6, lane = e & 63;
    const int row = min(weight_row<EPI>(p, t, lane & 15), limit);
    *(bf16x8*)(out + (((size_t)t * ksteps + c) * 64 + lane) * 8) = *(const bf16x8*)(p.W + (size_t)row * p.ldw + c * 32 + (lane >> 4) * 8);
  }
}
}  // namespace tasu_stream

extern "C" int tasu_stream_finish_norm(const float* slabs, int ksplit, float* C, const float* resid, int M, int N,
                                       const float* norm_w, void* y, float eps, int y_frag, void* stream) {
  if (!slabs || !C || !resid || !norm_w || !y || ksplit < 1 || M <= 0 || M > 64 || N <= 0 || N % 16 || (y_frag && N % 32))
    return TASU_ERR_ARG;
  TASU_LAUNCH(tasu_stream::stream_finish_norm_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, slabs, ksplit, N / 16, C, resid, N,
              norm_w, (bf16*)y, eps, y_frag);
  return TASU_OK;
}

extern "C" int tasu_to_fragment_order(const void* W, int ldw, void* out, int kind, int N, int K, int H, int G, void* stream) {
  using namespace tasu_stream;
  if (!W || !out || N <= 0 || K <= 0 || K % 32 || ldw % 8 || !aligned16(W) || !aligned16(out)) return TASU_ERR_ARG;
  Args a{};
  a.W = (const bf16*)W;
  a.K = K, a.ldw = ldw;
  hipStream_t st = (hipStream_t)stream;
  if (kind == E_BF16) {
    a.N = N;
    TASU_LAUNCH(to_fragment_order_kernel<E_BF16>, dim3((N + 15) / 16), dim3(256), 0, st, a, (bf16*)out);
  } else if (kind == E_SWIGLU) {                         // N = I (output columns); W holds 2 I rows
    if (N % 8) return TASU_ERR_ARG;
    a.N = N, a.I = N;
    TASU_LAUNCH(to_fragment_order_kernel<E_SWIGLU>, dim3(N / 8), dim3(256), 0, st, a, (bf16*)out);
  } else if (kind == E_QKV) {
    if (H <= 0 || G <= 0 || N != (H + 2 * G) * 128) return TASU_ERR_ARG;
    a.N = N, a.H = H, a.G = G;
    TASU_LAUNCH(to_fragment_order_kernel<E_QKV>, dim3((H + 2 * G) * 8), dim3(256), 0, st, a, (bf16*)out);
  } else {
    return TASU_ERR_ARG;
  }
  return TASU_OK;
}
