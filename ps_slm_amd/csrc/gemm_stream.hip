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
// K range per workgroup = 8 waves x KS x 32 (KS = 1, 2, 5, 6, 7: 256, 512, 1280, 1536, 1792); anything else stays on gemm_skinny.
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
#include "stream_body.h"
#include "../../include/tasu_hip.h"

namespace tasu_stream {
bool k_ranges(int K, int ksplit, int& kr, int& rem);

template <int KS, int EPI, int MT, bool FRAG, bool PN = false>
__global__ __launch_bounds__(64 * NW, KS > 8 ? 1 : 2) void stream_gemm_kernel(Args p) {
  __shared__ __attribute__((aligned(16))) float red[2 * NW * MT * 256 + (PN ? NW * 64 : 0)];
  int bx, by, bz;
  if (!grid_position((int)blockIdx.x, p.gx, p.gy, p.gz, bx, by, bz)) return;
  stream_gemm_body<KS, EPI, MT, FRAG, false, PN>(p, red, bx, p.gx, by, bz);
}

// The projection and, in the same launch, the RMSNorm of its complete rows (norm_tail, stream_body.h): E_RESID (o projection +
// residual -> post-attention norm) or E_SLAB (down projection's K-range slabs -> sum + residual + the next layer's input norm).
template <int KS, int EPI, int MT, bool FRAG, int NG>
__global__ __launch_bounds__(64 * NW, 2) void stream_gemm_norm_kernel(Args p, NormTail t) {
  __shared__ __attribute__((aligned(16))) float red[2 * NW * MT * 256];
  int bx, by, bz;
  if (!grid_position((int)blockIdx.x, p.gx, p.gy, p.gz, bx, by, bz)) return;
  stream_gemm_body<KS, EPI, MT, FRAG, true>(p, red, bx, p.gx, by, bz);
  norm_tail<NG, EPI>(p, t, red, p.gx * p.gy * p.gz);
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
int launch_ks(Args a, int kr, dim3 grid3, hipStream_t st) {
  a.gx = grid3.x, a.gy = grid3.y, a.gz = grid3.z;
  const int per = 8 * a.gz;                              // whole groups of (8 XCDs x row splits): grid_position
  const dim3 grid((a.gx * a.gy * a.gz + per - 1) / per * per);
  const int ks = kr / (NW * 32);
  if (a.ssq_out || a.ssq_in) {                           // the post-attention norm inside (PN): one range of K <= 2048
    if constexpr (EPI == E_RESID || EPI == E_SWIGLU || EPI == E_QKV) {
      switch (ks) {
        case 1: TASU_LAUNCH((stream_gemm_kernel<1, EPI, MT, FRAG, true>), grid, dim3(64 * NW), 0, st, a); return TASU_OK;
        case 2: TASU_LAUNCH((stream_gemm_kernel<2, EPI, MT, FRAG, true>), grid, dim3(64 * NW), 0, st, a); return TASU_OK;
        case 5: TASU_LAUNCH((stream_gemm_kernel<5, EPI, MT, FRAG, true>), grid, dim3(64 * NW), 0, st, a); return TASU_OK;
        case 6: TASU_LAUNCH((stream_gemm_kernel<6, EPI, MT, FRAG, true>), grid, dim3(64 * NW), 0, st, a); return TASU_OK;
        case 7: TASU_LAUNCH((stream_gemm_kernel<7, EPI, MT, FRAG, true>), grid, dim3(64 * NW), 0, st, a); return TASU_OK;
        default: return TASU_ERR_ARG;
      }
    }
    return TASU_ERR_ARG;
  }
  switch (ks) {
    case 1: TASU_LAUNCH((stream_gemm_kernel<1, EPI, MT, FRAG>), grid, dim3(64 * NW), 0, st, a); return TASU_OK;
    case 2: TASU_LAUNCH((stream_gemm_kernel<2, EPI, MT, FRAG>), grid, dim3(64 * NW), 0, st, a); return TASU_OK;
    case 5: TASU_LAUNCH((stream_gemm_kernel<5, EPI, MT, FRAG>), grid, dim3(64 * NW), 0, st, a); return TASU_OK;
    case 6: TASU_LAUNCH((stream_gemm_kernel<6, EPI, MT, FRAG>), grid, dim3(64 * NW), 0, st, a); return TASU_OK;
    case 7: TASU_LAUNCH((stream_gemm_kernel<7, EPI, MT, FRAG>), grid, dim3(64 * NW), 0, st, a); return TASU_OK;
    case 14:                                              // K = 3584 in one range: row halves only (registers; stream_body.h)
      if constexpr (MT == 2 && EPI != E_SLAB) {
        TASU_LAUNCH((stream_gemm_kernel<14, EPI, 2, FRAG>), grid, dim3(64 * NW), 0, st, a);
        return TASU_OK;
      }
      return TASU_ERR_ARG;
    default: return TASU_ERR_ARG;
  }
}
template <int EPI, int MT>
int launch_mt(const Args& a, int kr, dim3 grid, hipStream_t st) {
  if (a.a_frag != a.w_frag) return TASU_ERR_ARG;         // the operands travel in fragment order together or not at all
  return a.a_frag ? launch_ks<EPI, MT, true>(a, kr, grid, st) : launch_ks<EPI, MT, false>(a, kr, grid, st);
}

// ksplit ranges of kr_ (0: K / ksplit) starting at a.kstep0 / a.slab0
template <int EPI>
int launch(const Args& a, int ksplit, hipStream_t st, int kr_ = 0) {
  // Rows per workgroup: all 64 when the column tiles alone give every CU work; otherwise the row tiles are split over
  // blockIdx.z (2 x 32 rows) so that twice as many workgroups each load half of the activations.
  const int cus = cu_count();
  const int row_tiles = (a.M + 15) / 16;
  // (K-range slabs of 1792: always -- 7 k-steps x 4 row tiles of activations per wave would not leave registers for the weight
  //  ring, so the two row halves each stream the weights (the second read comes from the XCD's L2).  Slabs of 1280 (5 k-steps,
  //  K = 8960 as 7 ranges) keep all 64 rows in one workgroup: every weight byte is read once.)
  const int kr = kr_ ? kr_ : a.K / ksplit;
  // (one range of 3584 = 14 k-steps per wave: two row tiles of activations are all the registers hold, for any row count)
  const bool split_rows = kr > 1792 || (row_tiles > 2 && ((EPI == E_SLAB && kr > 1280) || (EPI != E_SLAB && a.tiles * ksplit * 2 <= cus + cus / 4)));
  const int zs = split_rows ? 2 : 1;
  const int per_split = cus / (ksplit * zs) > 0 ? cus / (ksplit * zs) : 1;
  const dim3 grid(a.tiles < per_split ? a.tiles : per_split, ksplit, zs);
  return split_rows ? launch_mt<EPI, 2>(a, kr, grid, st) : launch_mt<EPI, 4>(a, kr, grid, st);
}

// launch geometry of launch<EPI> for the fused-norm kernels (the same decisions)
template <int EPI, int NG>
int launch_norm(Args a, const NormTail& t, int ksplit, hipStream_t st) {
  const int cus = cu_count();
  const int row_tiles = (a.M + 15) / 16;
  const int kr = a.K / ksplit;
  const bool split_rows = row_tiles > 2 && ((EPI == E_SLAB && kr > 1280) || (EPI != E_SLAB && a.tiles * ksplit * 2 <= cus + cus / 4));
  const int zs = split_rows ? 2 : 1;
  const int per_split = cus / (ksplit * zs) > 0 ? cus / (ksplit * zs) : 1;
  a.gx = a.tiles < per_split ? a.tiles : per_split, a.gy = ksplit, a.gz = zs;
  const int per = 8 * a.gz;
  const dim3 grid((a.gx * a.gy * a.gz + per - 1) / per * per);
  if (a.a_frag != a.w_frag) return TASU_ERR_ARG;
  const int ks = kr / (NW * 32);
#define TASU_NL(KSV, MTV, FR)                                                                                              \
  do {                                                                                                                     \
    TASU_LAUNCH((stream_gemm_norm_kernel<KSV, EPI, MTV, FR, NG>), grid, dim3(64 * NW), 0, st, a, t);                       \
    return TASU_OK;                                                                                                        \
  } while (0)
#define TASU_NL_KS(KSV)                                                                                                    \
  case KSV:                                                                                                                \
    if (split_rows) { if (a.a_frag) TASU_NL(KSV, 2, true); else TASU_NL(KSV, 2, false); }                                   \
    else { if (a.a_frag) TASU_NL(KSV, 4, true); else TASU_NL(KSV, 4, false); }
  switch (ks) {
    TASU_NL_KS(1) TASU_NL_KS(2) TASU_NL_KS(5) TASU_NL_KS(6) TASU_NL_KS(7)
    default: return TASU_ERR_ARG;
  }
#undef TASU_NL_KS
#undef TASU_NL
}

static bool range_ok(int kr) { return kr == 256 || kr == 512 || kr == 1280 || kr == 1536 || kr == 1792; }

// The K ranges of a split: ksplit equal ranges, or (K not a multiple) ksplit - 1 ranges of `kr` and a shorter last one of `rem`,
// e.g. Qwen2.5-7B's down projection K = 18944 = 12 x 1536 + 512.  The ragged form is unique: the largest served range size whose
// remainder is a served size too.  One range (ksplit = 1) may also be 3584 (14 k-steps per wave, row halves).
bool k_ranges(int K, int ksplit, int& kr, int& rem) {
  kr = rem = 0;
  if (ksplit < 1 || K <= 0) return false;
  if (K % ksplit == 0) {
    kr = K / ksplit;
    return range_ok(kr) || (ksplit == 1 && kr == 3584);
  }
  if (ksplit < 2) return false;
  static const int sizes[] = {1792, 1536, 1280, 512};
  for (int c : sizes) {
    const int r = K - (ksplit - 1) * c;
    if (r > 0 && r < c && range_ok(r)) {
      kr = c, rem = r;
      return true;
    }
  }
  return false;
}

bool k_supported(int K, int ksplit) {
  int kr, rem;
  return k_ranges(K, ksplit, kr, rem);
}

}  // namespace tasu_stream

using tasu_stream::Args;

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

#ifdef TASU_STREAM_TRACE
extern "C" int tasu_stream_trace_read(uint64_t* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tasu_stream::g_stream_trace), 32 * sizeof(uint64_t)) == hipSuccess ? 0 : 2;
}
#endif
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

// Round 5: the post-attention RMSNorm without a launch of its own (Args::ssq_out / ssq_in, stream_body.h).
extern "C" int tasu_gemm_stream_resid_prenorm(const void* A, int lda, const void* W, int ldw, float* C, const float* resid, int M, int N,
                                              int K, const float* norm_w, void* yw, int yw_frag, float* sumsq, int a_frag, int w_frag,
                                              void* stream) {
  using namespace tasu_stream;
  if (!A || !W || !C || !resid || !norm_w || !yw || !sumsq || M <= 0 || M > 64 || N <= 0 || N % 16 || !k_supported(K, 1) || K > 2048 ||
      lda % 8 || ldw % 8)
    return TASU_ERR_ARG;
  if (!aligned16(A) || !aligned16(W) || !aligned16(C) || !aligned16(resid) || !aligned16(norm_w) || ((uintptr_t)yw & 7)) return TASU_ERR_ARG;
  Args a{};
  a.A = (const bf16*)A;
  a.W = (const bf16*)W;
  a.C = C;
  a.R = resid;
  a.M = M, a.N = N, a.K = K, a.lda = lda, a.ldw = ldw, a.ldc = N;
  a.tiles = N / 16;
  a.a_frag = a_frag, a.w_frag = w_frag;
  a.nw = norm_w, a.yw = (bf16*)yw, a.yw_frag = yw_frag, a.ssq_out = sumsq;
  return launch<E_RESID>(a, 1, (hipStream_t)stream);
}

extern "C" int tasu_gemm_stream_swiglu_rstd(const void* A, int lda, const void* Wgu, int ldw, void* act, int ldact, int M, int I, int K,
                                            const float* sumsq, int n_part, float eps, int a_frag, int w_frag, int out_frag,
                                            void* stream) {
  using namespace tasu_stream;
  if (!A || !Wgu || !act || !sumsq || n_part != K / 16 || M <= 0 || M > 64 || I <= 0 || I % 8 || !k_supported(K, 1) || K > 2048 || lda % 8 ||
      ldw % 8 || ldact % 4)
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
  a.ssq_in = sumsq, a.n_part = n_part, a.eps = eps;
  if (out_frag && I % 32) return TASU_ERR_ARG;
  return launch<E_SWIGLU>(a, 1, (hipStream_t)stream);
}

// tasu_gemm_stream_qkv_rope on A = bf16(norm_w . x) (tasu_stream_finish_prenorm): the accumulators are scaled by the rows' rstd from the
// K / 16 partial sums of squares before bias and RoPE.
extern "C" int tasu_gemm_stream_qkv_rope_rstd(const void* A, int lda, const void* Wqkv, int ldw, const void* bias, void* qkv, int M, int H,
                                              int G, int K, const float* cos_tab, const float* sin_tab, void* kcache, void* vcache,
                                              const int32_t* pos, int ctx, const float* sumsq, int n_part, float eps, int a_frag, int w_frag,
                                              void* stream) {
  using namespace tasu_stream;
  if (!A || !Wqkv || !qkv || !cos_tab || !sin_tab || !kcache || !vcache || !pos || !sumsq || n_part != K / 16 || M <= 0 || M > 64 || H <= 0 ||
      G <= 0 || !k_supported(K, 1) || K > 2048 || lda % 8 || ldw % 8 || ctx <= 0)
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
  a.ssq_in = sumsq, a.n_part = n_part, a.eps = eps;
  return launch<E_QKV>(a, 1, (hipStream_t)stream);
}

// K split over workgroups: fp32 partial results, row-major [ksplit][64 rows][N] in `slabs` (rows >= M: unspecified);
// tasu_stream_finish_norm sums them.
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
  int kr, rem;
  k_ranges(K, ksplit, kr, rem);
  if (!rem) return launch<E_SLAB>(a, ksplit, (hipStream_t)stream);
  // ragged: ksplit - 1 ranges of kr, then the short last range into the last slab (two launches back to back, each over the whole chip; they write different slabs)
  const int rc = launch<E_SLAB>(a, ksplit - 1, (hipStream_t)stream, kr);
  if (rc != TASU_OK) return rc;
  a.kstep0 = (ksplit - 1) * (kr / 32), a.slab0 = ksplit - 1;
  return launch<E_SLAB>(a, 1, (hipStream_t)stream, rem);
}

// Round 5: projection + residual + RMSNorm of the result in ONE launch (norm_tail, stream_body.h).  C (fp32) = resid + bf16(A W^T);
// y = rmsnorm(C, norm_w) (bf16; y_frag: fragment order).  ksplit = 1: the o projection's form (K in one range); ksplit > 1: the
// down projection's K-range slabs, summed by the finishing workgroups (`slabs`: [ksplit][64][N] fp32 scratch).  N = 256 or 1536;
// `sync`: two zero-initialised words (the kernel leaves them zero).  Same bits as tasu_gemm_stream_bf16(RESID) + tasu_rmsnorm_fwd[_frag]
// resp. tasu_gemm_stream_slabs + tasu_stream_finish_norm.
extern "C" int tasu_gemm_stream_norm(const void* A, int lda, const void* W, int ldw, float* C, const float* resid, int M, int N, int K,
                                     int ksplit, float* slabs, int64_t slab_floats, const float* norm_w, void* y, float eps, int a_frag,
                                     int w_frag, int y_frag, void* sync, void* stream) {
  using namespace tasu_stream;
  if (!A || !W || !C || !resid || !norm_w || !y || !sync || M <= 0 || M > 64 || (N != 256 && N != 1536) || ksplit < 1 || K % ksplit || !k_supported(K, ksplit) || lda % 8 ||
      ldw % 8)
    return TASU_ERR_ARG;
  if (!aligned16(A) || !aligned16(W) || !aligned16(C) || !aligned16(resid) || ((uintptr_t)y & 7)) return TASU_ERR_ARG;
  if (ksplit > 1 && (!slabs || !aligned16(slabs) || (int64_t)ksplit * (N / 16) * 1024 > slab_floats)) return TASU_ERR_ARG;
  Args a{};
  a.A = (const bf16*)A;
  a.W = (const bf16*)W;
  a.M = M, a.N = N, a.K = K, a.lda = lda, a.ldw = ldw, a.ldc = N;
  a.tiles = N / 16;
  a.a_frag = a_frag, a.w_frag = w_frag;
  NormTail t{};
  t.nw = norm_w, t.y = (bf16*)y, t.eps = eps, t.y_frag = y_frag, t.sync = (unsigned*)sync, t.ksplit = ksplit;
  hipStream_t st = (hipStream_t)stream;
  if (ksplit == 1) {
    a.C = C, a.R = resid;
    t.C = C;
    return N == 1536 ? launch_norm<E_RESID, 6>(a, t, 1, st) : launch_norm<E_RESID, 1>(a, t, 1, st);
  }
  a.C = slabs;
  t.slabs = slabs, t.C = C, t.R = resid;
  return N == 1536 ? launch_norm<E_SLAB, 6>(a, t, ksplit, st) : launch_norm<E_SLAB, 1>(a, t, ksplit, st);
}

namespace tasu_stream {
// Row-wise finish of tasu_gemm_stream_slabs (finish_norm_row, stream_body.h): one wave per row, blockDim.x / 64 rows per block.
template <int NG>
__global__ __launch_bounds__(256) void stream_finish_norm_kernel(const float* __restrict__ slabs, int ksplit, float* __restrict__ C,
                                                                 const float* __restrict__ R, int M, const float* __restrict__ nw,
                                                                 bf16* __restrict__ y, float eps, int y_frag) {
  const int m = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (m < M) finish_norm_row<NG, false>(slabs, ksplit, C, R, nw, y, eps, y_frag, m);
}

// The slab finish WITHOUT the norm's row dependency (round 5): C = R + bf16(sum of the slabs), yw = bf16(nw . C) in the consumer's
// operand order, and one sum of squares per (16-column tile, row) -- the consuming q|k|v projection applies rstd to its accumulators
// (tasu_gemm_stream_qkv_rope_rstd).  No whole-row reduction, so a wave takes 256 columns of one row: M * N / 256 waves.
__global__ __launch_bounds__(256) void stream_finish_prenorm_kernel(const float* __restrict__ slabs, int ksplit, float* __restrict__ C,
                                                                    const float* __restrict__ R, int M, int N, const float* __restrict__ nw,
                                                                    bf16* __restrict__ yw, int yw_frag, float* __restrict__ ssq) {
  const int ng = N >> 8;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wid >= M * ng) return;
  const int m = wid / ng, g = wid - m * ng;
  const int n = g * 256 + lane * 4;
  const size_t e = (size_t)m * N + n;
  const f32x4 v = *(const f32x4*)(R + e), w = *(const f32x4*)(nw + n);
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int KC = 8;
  for (int k0 = 0; k0 < ksplit; k0 += KC) {
    f32x4 t[KC];
#pragma unroll
    for (int j = 0; j < KC; ++j) t[j] = k0 + j < ksplit ? *(const f32x4*)(slabs + (size_t)(k0 + j) * 64 * N + e) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KC; ++j) s += t[j];
  }
  f32x4 c;
  bf16x4 y;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    c[q] = v[q] + bf16_round(s[q]);
    y[q] = (bf16)(w[q] * c[q]);
  }
  *(f32x4*)(C + e) = c;
  *(bf16x4*)(yw_frag ? yw + frag_index(m, n) : yw + e) = y;
  float q2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3];   // a 16-column tile = the 4 lanes 4j .. 4j + 3
  q2 += __shfl_xor(q2, 1, 64);
  q2 += __shfl_xor(q2, 2, 64);
  if ((lane & 3) == 0) ssq[(size_t)(g * 16 + (lane >> 2)) * 64 + m] = q2;
}

// N = 2 * NGH * 256 columns (3584: Qwen2.5-7B's down projection): a row's slab chunks in flight do not fit one wave's registers, so
// TWO waves share a row -- wave half h owns the column groups h * NGH .. + NGH - 1 -- and the sum of squares passes through LDS in
// the one-wave kernels' order (per lane the groups 0 .. NG - 1 in sequence, then the wave reduction).  blockDim.x / 128 rows per block (1 or 2).
template <int NGH>
__global__ __launch_bounds__(256) void stream_finish_norm_pair_kernel(const float* __restrict__ slabs, int ksplit, float* __restrict__ C,
                                                                      const float* __restrict__ R, int M, const float* __restrict__ nw,
                                                                      bf16* __restrict__ y, float eps, int y_frag) {
  constexpr int N = 2 * NGH * 256, KC = 4;
  __shared__ float xch[2][64];
  __shared__ float rsx[2];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rp = wave >> 1, h = wave & 1;
  const int m = blockIdx.x * (blockDim.x >> 7) + rp;
  const bool live = m < M;
  const int mm = live ? m : M - 1;                       // (a block's spare row pair repeats the last row and stores nothing)
  f32x4 v[NGH], s[NGH];
#pragma unroll
  for (int g = 0; g < NGH; ++g) {
    v[g] = *(const f32x4*)(R + (size_t)mm * N + lane * 4 + (h * NGH + g) * 256);
    s[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int k0 = 0; k0 < ksplit; k0 += KC) {
    f32x4 t[NGH][KC];
#pragma unroll
    for (int g = 0; g < NGH; ++g)
#pragma unroll
      for (int j = 0; j < KC; ++j)
        t[g][j] = k0 + j < ksplit ? *(const f32x4*)(slabs + (size_t)(k0 + j) * 64 * N + (size_t)mm * N + lane * 4 + (h * NGH + g) * 256)
                                  : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < NGH; ++g)
#pragma unroll
      for (int j = 0; j < KC; ++j) s[g] += t[g][j];
  }
  float ss = 0.f;
#pragma unroll
  for (int g = 0; g < NGH; ++g) {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[g][q] = v[g][q] + bf16_round(s[g][q]);
    if (live) *(f32x4*)(C + (size_t)m * N + lane * 4 + (h * NGH + g) * 256) = v[g];
  }
  if (h == 1) ss = 0.f;
  if (h == 0) {
#pragma unroll
    for (int g = 0; g < NGH; ++g) ss += v[g][0] * v[g][0] + v[g][1] * v[g][1] + v[g][2] * v[g][2] + v[g][3] * v[g][3];
    xch[rp][lane] = ss;
  }
  __syncthreads();
  if (h == 1) {
    ss = xch[rp][lane];
#pragma unroll
    for (int g = 0; g < NGH; ++g) ss += v[g][0] * v[g][0] + v[g][1] * v[g][1] + v[g][2] * v[g][2] + v[g][3] * v[g][3];
    ss = wave_sum(ss);
    if (lane == 0) rsx[rp] = rsqrtf(ss / (float)N + eps);
  }
  __syncthreads();
  const float rs = rsx[rp];
  if (!live) return;
#pragma unroll
  for (int g = 0; g < NGH; ++g) {
    const int n = lane * 4 + (h * NGH + g) * 256;
    const f32x4 w = *(const f32x4*)(nw + n);
    f32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = w[q] * (v[g][q] * rs);
    bf16* dst = y_frag ? y + frag_index(m, n) : y + (size_t)m * N + n;
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
  if (!slabs || !C || !resid || !norm_w || !y || ksplit < 1 || M <= 0 || M > 64 || N <= 0 || N % 256) return TASU_ERR_ARG;
  // rows per block: the finish is bound by what ONE CU ingests (ksplit slabs x N x 4 B per row), so the rows are spread over as
  // many CUs as there are rows (TASU_FINISH_ROWS in the lab build: A/B runs; round 4 ran 4 rows per block)
  static const int rows_env = [] { const char* e = tasu_lab_env("TASU_FINISH_ROWS"); return e ? atoi(e) : 0; }();
  const int rows = rows_env == 1 || rows_env == 2 || rows_env == 4 ? rows_env : 1;
  const dim3 grid((M + rows - 1) / rows);
  hipStream_t st = (hipStream_t)stream;
#define TASU_FIN(NG)                                                                                                              \
  case NG:                                                                                                                        \
    TASU_LAUNCH(tasu_stream::stream_finish_norm_kernel<NG>, grid, dim3(64 * rows), 0, st, slabs, ksplit, C, resid, M, norm_w, (bf16*)y, \
                eps, y_frag);                                                                                                     \
    return TASU_OK;
  switch (N / 256) {
    TASU_FIN(1) TASU_FIN(2) TASU_FIN(6) TASU_FIN(7)
    case 14: {
      const int pr = rows >= 2 ? 2 : 1;
      TASU_LAUNCH(tasu_stream::stream_finish_norm_pair_kernel<7>, dim3((M + pr - 1) / pr), dim3(128 * pr), 0, st, slabs, ksplit, C, resid, M,
                  norm_w, (bf16*)y, eps, y_frag);
      return TASU_OK;
    }
    default: return TASU_ERR_ARG;
  }
#undef TASU_FIN
}

extern "C" int tasu_stream_finish_prenorm(const float* slabs, int ksplit, float* C, const float* resid, int M, int N, const float* norm_w,
                                          void* yw, int yw_frag, float* sumsq, void* stream) {
  if (!slabs || !C || !resid || !norm_w || !yw || !sumsq || ksplit < 1 || M <= 0 || M > 64 || N <= 0 || N % 256) return TASU_ERR_ARG;
  const int waves = M * (N / 256);
  TASU_LAUNCH(tasu_stream::stream_finish_prenorm_kernel, dim3((waves + 3) / 4), dim3(256), 0, (hipStream_t)stream, slabs, ksplit, C, resid,
              M, N, norm_w, (bf16*)yw, yw_frag, sumsq);
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
