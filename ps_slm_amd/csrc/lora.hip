// Pieces of the LoRA recipe (use_peft=true: Multitask/model/ps-slm.py:114-117, PeftConfig at
// Multitask/aispeech_asr_config.py:41-50; peft 0.6.0 lora.Linear.forward, absent from the reference tree:
//     result = base(x);  result += lora_B(lora_A(dropout(x))) * scaling
// under torch.autocast(bfloat16): base(x), lora_A(.), lora_B(.) are bf16 GEMM results, `* scaling` and `+=` round to bf16).
// The rank-sized GEMMs are in gemm_rank.hip, the base GEMMs in gemm*.hip; this file holds the fused low-rank accumulate
// (tasu_lora_apply), the counter-based dropout, the one-launch refresh of the working copies and two layout helpers.  All
// HBM-bound.
#include "common.h"
#include "../../include/tasu_hip.h"

namespace {

inline int grid_for(int64_t nvec) {
  int64_t b = (nvec + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

__global__ __launch_bounds__(256) void scale_bf16_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, float s, int64_t nvec) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const bf16x8 a = *(const bf16x8*)(src + i * 8);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)a[j] * s);
    *(bf16x8*)(dst + i * 8) = o;
  }
}

// Counter-based mask: element `idx` of dropout stream `sid` at optimizer micro-step `step` is kept iff the upper 32 bits of
// splitmix64(seed ^ step * GOLD ^ sid << 44, + idx * ODD) are >= p * 2^32.  Stateless, so backward regenerates the mask the
// forward used, and graph replays draw new masks because {seed, step} live in device memory (tasu_rng_advance bumps step).
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t mask_key(const int64_t* rng, int sid) {
  return (uint64_t)rng[0] ^ ((uint64_t)rng[1] * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(uint32_t)sid << 44);
}
__device__ __forceinline__ float keep_scale(uint64_t key, int64_t idx, uint32_t thr, float inv) {
  const uint64_t z = mix64(key + (uint64_t)idx * 0xD1B54A32D192ED03ull);
  return (uint32_t)(z >> 32) >= thr ? inv : 0.f;
}

__global__ __launch_bounds__(256) void dropout_bf16_kernel(const bf16* __restrict__ src, int ld_src, bf16* __restrict__ dst, int ld_dst,
                                                           int M, int C8, uint32_t thr, float inv, const int64_t* __restrict__ rng, int sid) {
  const uint64_t key = mask_key(rng, sid);
  const int64_t nvec = (int64_t)M * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const int m = (int)(i / C8), c = (int)(i - (int64_t)m * C8) * 8;
    const bf16x8 a = *(const bf16x8*)(src + (size_t)m * ld_src + c);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)a[j] * keep_scale(key, i * 8 + j, thr, inv));   // element index m * C + c + j
    *(bf16x8*)(dst + (size_t)m * ld_dst + c) = o;
  }
}

// dropout applied to the fp32 RMSNorm output (recomputed from the saved row scale: g * (x * rstd), norm.hip's arithmetic),
// rounded to bf16 once -- the order the reference has (the norm's output is fp32; lora_A's autocast cast comes after the dropout)
__global__ __launch_bounds__(256) void dropout_norm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ rstd, bf16* __restrict__ dst, int M, int D,
                                                           uint32_t thr, float inv, const int64_t* __restrict__ rng, int sid) {
  const uint64_t key = mask_key(rng, sid);
  const int64_t nvec = (int64_t)M * D / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const int64_t e = i * 4;
    const int row = (int)(e / D), c = (int)(e - (int64_t)row * D);
    const f32x4 v = *(const f32x4*)(x + e), g = *(const f32x4*)(w + c);
    const float r = rstd[row];
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (bf16)((g[j] * (v[j] * r)) * keep_scale(key, e + j, thr, inv));
    *(bf16x4*)(dst + e) = o;
  }
}

// y[M, N] = bf16(y + mask . bf16(s . bf16(u[M, R] W[N, R]^T)))  [, x_out = x_in + float(y)]: an adapter's second GEMM (K = R = the
// padded rank, 64) with the accumulate into the base result fused -- forward: u = the rank-sized activations, W = lora_B, y = the
// base Linear's output; backward: u = du, W = A^T, y = the base path's input gradient, mask = the forward's dropout mask of that
// input.  HBM-bound on y (read + write); everything about the kernel serves that pass: the 64 x 256 tile's low-rank product goes
// through MFMA into an LDS image (bf16, the rounding the unfused GEMM had), then every thread handles 8 consecutive columns --
// 16-byte loads and stores of whole 512-byte row segments.  W (256 rows x 128 B) and u (64 rows) reach LDS by LDS-DMA, XOR-swizzled
// on the source like the GEMMs' tiles.
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
constexpr int AP_BM = 64, AP_BN = 256, AP_VLD = 528;   // v image row stride in bytes (512 + 16: 8-byte writes of 16 rows spread over the banks)

template <bool MASK, bool RESID>
__global__ __launch_bounds__(256, 2) void lora_apply_kernel(bf16* __restrict__ y, int ldy, const bf16* __restrict__ u, int ldu,
                                                            const bf16* __restrict__ W, int ldw, int M, int N, int R, float s,
                                                            uint32_t thr, float inv, const int64_t* __restrict__ rng, int sid,
                                                            const float* __restrict__ x_in, float* __restrict__ x_out, int ldx) {
  __shared__ __attribute__((aligned(16))) char smem[AP_BN * 128 + AP_BM * 128];   // 40 KiB: W and u tiles, then (aliased) the v image
  static_assert(AP_BM * AP_VLD <= AP_BN * 128 + AP_BM * 128, "the v image reuses the operand tiles' space");
  char* wt = smem;
  char* ut = smem + AP_BN * 128;
  char* vt = smem;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m0 = blockIdx.x * AP_BM, n0 = blockIdx.y * AP_BN;
  f32x4 acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int sw = (lane >> 1) & 7;
  for (int kc = 0; kc < R; kc += 64) {
    // W: 32 pieces of 8 rows (8 per wave); u: 8 pieces (2 per wave)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int pc = wave * 8 + i, r = pc * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
      const bf16* src = W + (size_t)min(n0 + r, N - 1) * ldw + kc + c * 8;
      __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(wt + pc * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pc = wave * 2 + i, r = pc * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
      const bf16* src = u + (size_t)min(m0 + r, M - 1) * ldu + kc + c * 8;
      __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(ut + pc * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int roff = (lane & 15) * 128 + (((kk * 4 + (lane >> 4)) ^ sw) << 4);
      const bf16x8 fa = *(const bf16x8*)(ut + wave * 2048 + roff);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const bf16x8 fb = *(const bf16x8*)(wt + j * 2048 + roff);
        acc[j] = mfma16(fb, fa, acc[j]);               // acc[j][r] = v[m0 + wave*16 + (lane & 15)][n0 + j*16 + (lane >> 4)*4 + r]
      }
    }
    __syncthreads();
  }
  {
    char* row = vt + (wave * 16 + (lane & 15)) * AP_VLD + (lane >> 4) * 8;
#pragma unroll
    for (int j = 0; j < 16; ++j) *(bf16x4*)(row + j * 32) = __builtin_convertvector(acc[j], bf16x4);
  }
  __syncthreads();
  uint64_t key = 0;
  if constexpr (MASK) key = mask_key(rng, sid);
  const int cch = (tid & 31) * 8, n = n0 + cch;
  if (n >= N) return;
#pragma unroll
  for (int pass = 0; pass < 8; ++pass) {
    const int r = pass * 8 + (tid >> 5), m = m0 + r;
    if (m >= M) break;
    bf16* yp = y + (size_t)m * ldy + n;
    const bf16x8 yv = *(const bf16x8*)yp;
    const bf16x8 vv = *(const bf16x8*)(vt + r * AP_VLD + cch * 2);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float d = (float)(bf16)((float)vv[e] * s);
      if constexpr (MASK) d = (float)(bf16)(d * keep_scale(key, (int64_t)m * N + n + e, thr, inv));
      o[e] = (bf16)((float)yv[e] + d);
    }
    *(bf16x8*)yp = o;
    if constexpr (RESID) {
      const float* xi = x_in + (size_t)m * ldx + n;
      float* xo = x_out + (size_t)m * ldx + n;
      const f32x4 a0 = *(const f32x4*)xi, a1 = *(const f32x4*)(xi + 4);
      f32x4 q0, q1;
#pragma unroll
      for (int e = 0; e < 4; ++e) q0[e] = a0[e] + (float)o[e], q1[e] = a1[e] + (float)o[4 + e];
      *(f32x4*)xo = q0;
      *(f32x4*)(xo + 4) = q1;
    }
  }
}

// All working copies of all adapters in one launch (after every optimizer step): entry e of the device table describes one 2-D
// copy out of the bucket's bf16 image -- dst = bf16(scale * src) or its transpose -- as eight int64: {src offset (elements),
// dst address, rows, cols, dst leading dimension, transpose, scale (float bits), first tile}.  Work unit: a 64 x 64 tile.
__global__ __launch_bounds__(256) void lora_refresh_kernel(const bf16* __restrict__ pb, const int64_t* __restrict__ table, int n_entries) {
  __shared__ bf16 tile[64][64 + 2];
  const int b = blockIdx.x;
  int lo = 0, hi = n_entries - 1;                      // last entry whose first tile is <= b
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid * 8 + 7] <= b) lo = mid;
    else hi = mid - 1;
  }
  const int64_t* e = table + lo * 8;
  const bf16* src = pb + e[0];
  bf16* dst = (bf16*)(uintptr_t)e[1];
  const int rows = (int)e[2], cols = (int)e[3], ld = (int)e[4], tr = (int)e[5];
  const float scale = __int_as_float((int)e[6]);
  const int ti = b - (int)e[7], tiles_c = (cols + 63) >> 6;
  const int r0 = (ti / tiles_c) * 64, c0 = (ti % tiles_c) * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    bf16 v = (bf16)0.f;
    if (r0 + r < rows && c0 + c < cols) v = (bf16)((float)src[(size_t)(r0 + r) * cols + c0 + c] * scale);
    tile[r][c] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    if (tr) {
      const int c = i >> 6, r = i & 63;                // consecutive threads -> consecutive rows of src = consecutive columns of dst
      if (r0 + r < rows && c0 + c < cols) dst[(size_t)(c0 + c) * ld + r0 + r] = tile[r][c];
    } else {
      const int r = i >> 6, c = i & 63;
      if (r0 + r < rows && c0 + c < cols) dst[(size_t)(r0 + r) * ld + c0 + c] = tile[r][c];
    }
  }
}

// dst[m, 0:C] = src[m, 0:C] for two row-major bf16 matrices with their own leading dimensions (C % 8 == 0)
__global__ __launch_bounds__(256) void copy_rows_kernel(const bf16* __restrict__ src, int lds_, bf16* __restrict__ dst, int ldd, int M, int C8) {
  const int64_t nvec = (int64_t)M * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const int m = (int)(i / C8), c = (int)(i - (int64_t)m * C8) * 8;
    *(bf16x8*)(dst + (size_t)m * ldd + c) = *(const bf16x8*)(src + (size_t)m * lds_ + c);
  }
}

// ---- one launch per adapted GROUP (q|k|v: three members, gate|up: two) instead of one per member --------------------------------
constexpr int GROUP_MAX = 4;
struct ApplyGroup {
  const bf16* u[GROUP_MAX];
  const bf16* W[GROUP_MAX];
  int sid[GROUP_MAX];
  int n;
};
// tasu_lora_apply for every member of a group in ONE pass over y: y = bf16(y + mask_t . bf16(s . bf16(u_t W_t^T))) for t = 0, 1, ...
// in that order -- the roundings of the member-by-member launches, the same bits -- with the 64 x 256 tile of y held in registers
// across the members (one read and one write of y instead of one per member).  R = 64.
template <bool MASK>
__global__ __launch_bounds__(256, 2) void lora_apply_group_kernel(bf16* __restrict__ y, int ldy, ApplyGroup g, int ldu, int ldw, int M, int N,
                                                                  float s, uint32_t thr, float inv, const int64_t* __restrict__ rng) {
  __shared__ __attribute__((aligned(16))) char smem[AP_BN * 128 + AP_BM * 128];
  char* wt = smem;
  char* ut = smem + AP_BN * 128;
  char* vt = smem;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m0 = blockIdx.x * AP_BM, n0 = blockIdx.y * AP_BN;
  const int sw = (lane >> 1) & 7;
  const int cch = (tid & 31) * 8, n = n0 + cch;
  const bool live = n < N;
  bf16x8 yv[8];
#pragma unroll
  for (int pass = 0; pass < 8; ++pass) {
    const int m = m0 + pass * 8 + (tid >> 5);
    yv[pass] = (live && m < M) ? *(const bf16x8*)(y + (size_t)m * ldy + n) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
  for (int t = 0; t < g.n; ++t) {
    f32x4 acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int pc = wave * 8 + i, r = pc * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
      const bf16* src = g.W[t] + (size_t)min(n0 + r, N - 1) * ldw + c * 8;
      __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(wt + pc * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pc = wave * 2 + i, r = pc * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
      const bf16* src = g.u[t] + (size_t)min(m0 + r, M - 1) * ldu + c * 8;
      __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(ut + pc * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int roff = (lane & 15) * 128 + (((kk * 4 + (lane >> 4)) ^ sw) << 4);
      const bf16x8 fa = *(const bf16x8*)(ut + wave * 2048 + roff);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const bf16x8 fb = *(const bf16x8*)(wt + j * 2048 + roff);
        acc[j] = mfma16(fb, fa, acc[j]);
      }
    }
    __syncthreads();
    {
      char* row = vt + (wave * 16 + (lane & 15)) * AP_VLD + (lane >> 4) * 8;
#pragma unroll
      for (int j = 0; j < 16; ++j) *(bf16x4*)(row + j * 32) = __builtin_convertvector(acc[j], bf16x4);
    }
    __syncthreads();
    uint64_t key = 0;
    if constexpr (MASK) key = mask_key(rng, g.sid[t]);
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int r = pass * 8 + (tid >> 5), m = m0 + r;
      const bf16x8 vv = *(const bf16x8*)(vt + r * AP_VLD + cch * 2);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float d = (float)(bf16)((float)vv[e] * s);
        if constexpr (MASK) d = (float)(bf16)(d * keep_scale(key, (int64_t)m * N + n + e, thr, inv));
        yv[pass][e] = (bf16)((float)yv[pass][e] + d);
      }
    }
    __syncthreads();                                     // the v image is the next member's operand space
  }
  if (!live) return;
#pragma unroll
  for (int pass = 0; pass < 8; ++pass) {
    const int m = m0 + pass * 8 + (tid >> 5);
    if (m < M) *(bf16x8*)(y + (size_t)m * ldy + n) = yv[pass];
  }
}

struct DropGroup {
  bf16* dst[GROUP_MAX];
  int sid[GROUP_MAX];
  int n;
};
// tasu_lora_dropout_norm for every member of a group: the norm's fp32 output is computed once, each member's mask applied to it
__global__ __launch_bounds__(256) void dropout_norm_group_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ rstd, DropGroup g, int M, int D, uint32_t thr,
                                                                 float inv, const int64_t* __restrict__ rng) {
  uint64_t key[GROUP_MAX];
#pragma unroll
  for (int t = 0; t < GROUP_MAX; ++t) key[t] = t < g.n ? mask_key(rng, g.sid[t]) : 0;
  const int64_t nvec = (int64_t)M * D / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const int64_t e = i * 4;
    const int row = (int)(e / D), c = (int)(e - (int64_t)row * D);
    const f32x4 v = *(const f32x4*)(x + e), gw = *(const f32x4*)(w + c);
    const float r = rstd[row];
    float nv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) nv[j] = gw[j] * (v[j] * r);
#pragma unroll
    for (int t = 0; t < GROUP_MAX; ++t) {
      if (t >= g.n) break;
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (bf16)(nv[j] * keep_scale(key[t], e + j, thr, inv));
      *(bf16x4*)(g.dst[t] + e) = o;
    }
  }
}

__global__ void rng_advance_kernel(int64_t* rng) {
  if (threadIdx.x == 0 && blockIdx.x == 0) rng[1] += 1;
}

inline bool drop_args(float p, uint32_t* thr, float* inv) {
  if (!(p >= 0.f) || !(p < 1.f)) return false;
  const double t = (double)p * 4294967296.0;
  *thr = t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
  *inv = 1.0f / (1.0f - p);
  return true;
}

}  // namespace

extern "C" int tasu_scale_bf16(const void* src, void* dst, float s, int64_t n, void* stream) {
  if (!src || !dst || n <= 0 || n % 8) return TASU_ERR_ARG;
  TASU_LAUNCH(scale_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, (bf16*)dst, s, n / 8);
  return TASU_OK;
}

extern "C" int tasu_lora_dropout(const void* src, int ld_src, void* dst, int ld_dst, int M, int C, float p, const void* rng, int stream_id,
                                 void* stream) {
  uint32_t thr;
  float inv;
  if (!src || !dst || !rng || M <= 0 || C <= 0 || C % 8 || ld_src < C || ld_dst < C || ld_src % 8 || ld_dst % 8 || stream_id < 0 ||
      !drop_args(p, &thr, &inv))
    return TASU_ERR_ARG;
  TASU_LAUNCH(dropout_bf16_kernel, dim3(grid_for((int64_t)M * C / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, ld_src, (bf16*)dst,
              ld_dst, M, C / 8, thr, inv, (const int64_t*)rng, stream_id);
  return TASU_OK;
}

extern "C" int tasu_lora_dropout_norm(const float* x, const float* w, const float* rstd, void* dst, int M, int D, float p,
                                      const void* rng, int stream_id, void* stream) {
  uint32_t thr;
  float inv;
  if (!x || !w || !rstd || !dst || !rng || M <= 0 || D <= 0 || D % 4 || stream_id < 0 || !drop_args(p, &thr, &inv)) return TASU_ERR_ARG;
  TASU_LAUNCH(dropout_norm_kernel, dim3(grid_for((int64_t)M * D / 4)), dim3(256), 0, (hipStream_t)stream, x, w, rstd, (bf16*)dst, M, D, thr,
              inv, (const int64_t*)rng, stream_id);
  return TASU_OK;
}

extern "C" int tasu_rng_advance(void* rng, void* stream) {
  if (!rng) return TASU_ERR_ARG;
  TASU_LAUNCH(rng_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (int64_t*)rng);
  return TASU_OK;
}

extern "C" int tasu_lora_apply(void* y, int ldy, const void* u, int ldu, const void* W, int ldw, int M, int N, int R, float s, float p,
                               const void* rng, int stream_id, const float* x_in, float* x_out, int ldx, void* stream) {
  if (!y || !u || !W || M <= 0 || N <= 0 || N % 8 || R <= 0 || R % 64 || ldy % 8 || ldu % 8 || ldw % 8 || ldu < R || ldw < R || ldy < N)
    return TASU_ERR_ARG;
  if ((((uintptr_t)y | (uintptr_t)u | (uintptr_t)W) & 15) || ((x_in == nullptr) != (x_out == nullptr)) || (x_in && (ldx % 4 || ldx < N)))
    return TASU_ERR_ARG;
  const bool mask = p > 0.f;
  uint32_t thr = 0;
  float inv = 1.f;
  if (mask && (!rng || stream_id < 0 || !drop_args(p, &thr, &inv))) return TASU_ERR_ARG;
  if (p < 0.f) return TASU_ERR_ARG;
  const dim3 grid((M + AP_BM - 1) / AP_BM, (N + AP_BN - 1) / AP_BN), block(256);
  hipStream_t st = (hipStream_t)stream;
  bf16* yy = (bf16*)y;
  const bf16 *uu = (const bf16*)u, *ww = (const bf16*)W;
  const int64_t* rg = (const int64_t*)rng;
  if (mask && x_in) TASU_LAUNCH((lora_apply_kernel<true, true>), grid, block, 0, st, yy, ldy, uu, ldu, ww, ldw, M, N, R, s, thr, inv, rg, stream_id, x_in, x_out, ldx);
  else if (mask) TASU_LAUNCH((lora_apply_kernel<true, false>), grid, block, 0, st, yy, ldy, uu, ldu, ww, ldw, M, N, R, s, thr, inv, rg, stream_id, x_in, x_out, ldx);
  else if (x_in) TASU_LAUNCH((lora_apply_kernel<false, true>), grid, block, 0, st, yy, ldy, uu, ldu, ww, ldw, M, N, R, s, thr, inv, rg, stream_id, x_in, x_out, ldx);
  else TASU_LAUNCH((lora_apply_kernel<false, false>), grid, block, 0, st, yy, ldy, uu, ldu, ww, ldw, M, N, R, s, thr, inv, rg, stream_id, x_in, x_out, ldx);
  return TASU_OK;
}

extern "C" int tasu_lora_apply_group(void* y, int ldy, int n_members, const void* const* u, int ldu, const void* const* W, int ldw,
                                     const int* stream_ids, int M, int N, int R, float s, float p, const void* rng, void* stream) {
  if (!y || !u || !W || !stream_ids || n_members < 1 || n_members > GROUP_MAX || M <= 0 || N <= 0 || N % 8 || R != 64 || ldy % 8 || ldu % 8 ||
      ldw % 8 || ldu < R || ldw < R || ldy < N || ((uintptr_t)y & 15) || p < 0.f)
    return TASU_ERR_ARG;
  ApplyGroup g{};
  g.n = n_members;
  for (int t = 0; t < n_members; ++t) {
    if (!u[t] || !W[t] || stream_ids[t] < 0 || (((uintptr_t)u[t] | (uintptr_t)W[t]) & 15)) return TASU_ERR_ARG;
    g.u[t] = (const bf16*)u[t], g.W[t] = (const bf16*)W[t], g.sid[t] = stream_ids[t];
  }
  const bool mask = p > 0.f;
  uint32_t thr = 0;
  float inv = 1.f;
  if (mask && (!rng || !drop_args(p, &thr, &inv))) return TASU_ERR_ARG;
  const dim3 grid((M + AP_BM - 1) / AP_BM, (N + AP_BN - 1) / AP_BN), block(256);
  if (mask) TASU_LAUNCH((lora_apply_group_kernel<true>), grid, block, 0, (hipStream_t)stream, (bf16*)y, ldy, g, ldu, ldw, M, N, s, thr, inv, (const int64_t*)rng);
  else TASU_LAUNCH((lora_apply_group_kernel<false>), grid, block, 0, (hipStream_t)stream, (bf16*)y, ldy, g, ldu, ldw, M, N, s, thr, inv, (const int64_t*)rng);
  return TASU_OK;
}

extern "C" int tasu_lora_dropout_norm_group(const float* x, const float* w, const float* rstd, int n_members, void* const* dst,
                                            const int* stream_ids, int M, int D, float p, const void* rng, void* stream) {
  uint32_t thr;
  float inv;
  if (!x || !w || !rstd || !dst || !stream_ids || !rng || n_members < 1 || n_members > GROUP_MAX || M <= 0 || D <= 0 || D % 4 ||
      !drop_args(p, &thr, &inv))
    return TASU_ERR_ARG;
  DropGroup g{};
  g.n = n_members;
  for (int t = 0; t < n_members; ++t) {
    if (!dst[t] || stream_ids[t] < 0) return TASU_ERR_ARG;
    g.dst[t] = (bf16*)dst[t], g.sid[t] = stream_ids[t];
  }
  TASU_LAUNCH(dropout_norm_group_kernel, dim3(grid_for((int64_t)M * D / 4)), dim3(256), 0, (hipStream_t)stream, x, w, rstd, g, M, D, thr, inv,
              (const int64_t*)rng);
  return TASU_OK;
}

extern "C" int tasu_lora_refresh(const void* pb, const void* table, int n_entries, int total_tiles, void* stream) {
  if (!pb || !table || n_entries <= 0 || total_tiles <= 0) return TASU_ERR_ARG;
  TASU_LAUNCH(lora_refresh_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, (const bf16*)pb, (const int64_t*)table, n_entries);
  return TASU_OK;
}

extern "C" int tasu_copy_rows_bf16(const void* src, int ld_src, void* dst, int ld_dst, int M, int C, void* stream) {
  if (!src || !dst || M <= 0 || C <= 0 || C % 8 || ld_src < C || ld_dst < C || ld_src % 8 || ld_dst % 8) return TASU_ERR_ARG;
  TASU_LAUNCH(copy_rows_kernel, dim3(grid_for((int64_t)M * C / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, ld_src, (bf16*)dst,
              ld_dst, M, C / 8);
  return TASU_OK;
}
