// The decoder layers of one generated position as ONE launch (decode step of slam_model_asr.generate, Multitask/model/ps-slm.py:
// 660-675 -> HF generate: 28 x [q|k|v + RoPE + cache append, cache attention, o + residual, RMSNorm, gate|up + SwiGLU, down,
// residual + next RMSNorm], M <= 64 beam rows).
//
// Why: a layer was seven dependent launches of 5-15 us each (58 us), of which the weight stream needs 17: every launch starts
// with an idle fabric, pays the trip to its first weight tile (2-5 us with 256 workgroups asking at once) and ends on a tail
// (rocprofv3: q|k|v 5.4 us for 6.3 MB, o 5.1 us for 4.7 MB, two norm kernels of 5 us for < 1 MB each).  Here the seven steps
// are ROLES of one grid: workgroup b of the launch serves role (b / per_layer, position of b % per_layer in the layer's role
// list), and a role's workgroup
//      1. requests what does not depend on the step before it -- its first three weight tiles, RoPE factors, bias, cache rows
//         of the old positions -- BEFORE
//      2. waiting for its producers' completion counter (one lane polls, device scope, bounded), then
//      3. loads the activations its producers wrote (write-through stores, sc1 loads: MI355X_MICROARCH "valid forms"), computes
//         (the bodies of gemm_stream.hip / decode.hip: same arithmetic, order and rounding -- bit-identical results), stores
//         write-through, drains, and adds 1 to its own role's counter.
// So the weight stream of step k+1 is in flight while step k computes: what a launch boundary cannot do.
//
// Progress: a workgroup only waits for workgroups with LOWER linear ids (roles are laid out in dependency order), and the
// hardware dispatches workgroups in id order: the lowest unfinished workgroup always has all its producers done.  HIP does not
// promise that order, so every wait is bounded (50 ms): a timeout raises the launch's error word, every other waiter sees it and
// leaves, and the host raises -- never a hang.
//
// Every inter-workgroup buffer is a per-layer slice written exactly once per launch (no address is read before it is written
// inside the launch, so no cache can hold an older copy of a line it is about to be handed), stored sc1 and loaded sc1.
#include <stdlib.h>

#ifndef TASU_ROLES_LD_AUX
#define TASU_ROLES_LD_AUX 0
#endif
#include "attn_decode_body.h"
#include "stream_body.h"
#include "../../include/tasu_hip.h"

namespace tasu_roles {

using namespace tasu_stream;

enum { R_QKV = 0, R_ATTN, R_O, R_NORM, R_GU, R_DOWN, R_FIN, R_KINDS };
constexpr int SC1 = TASU_ROLES_LD_AUX;      // cache policy of the loads of handed-off data (lab: 16 = sc1, 0 = plain)
constexpr long long TIMEOUT_TICKS = 5000000;   // 50 ms of the 100 MHz wall clock

struct Role {
  Args a;                                  // GEMM roles (stream_body.h); gx / gy / gz = the role's virtual grid
  // attention
  const bf16* qkv;                         // [M, (H + 2G) * 128] of this layer
  const bf16 *kc, *vc;                     // this layer's cache [M, ctx, G * 128]
  const int32_t *index, *kstart, *lens;
  bf16* ao;                                // attention output (fragment order)
  // row roles (post-attention norm; slab finish + next norm)
  const float* x_in;                       // R_NORM: fp32 rows to normalise; R_FIN: residual rows
  const float* nw;                         // norm weight
  bf16* y;                                 // normed rows, fragment order
  float* c_out;                            // R_FIN: residual stream out (next layer's input)
  const float* slabs;                      // R_FIN: K-range slabs [ksplit][64][N]
  int ksplit;
  float eps, scale;
  int H, G, ctx, M;
  int dep, dep_target, sig;                // counter indices (-1: none) and the count that means "all producers done"
  int work;                                // workgroups of the role that have work (the rest of its blocks exit at once)
};

#ifdef TASU_ROLES_TRACE
// instrumented build (make trace; tools/roles_trace.py): per workgroup of ONE layer, wall-clock stamps (10 ns) at entry, after
// the dependency wait, before the signal and at exit
constexpr int TRACE_LAYER = 10;
__device__ unsigned long long g_roles_trace[4 * 2048];
#define TASU_ROLES_STAMP(k) do { if (threadIdx.x == 0 && layer == TRACE_LAYER && w < 2048) g_roles_trace[w * 4 + (k)] = wall_clock64(); } while (0)
#else
#define TASU_ROLES_STAMP(k) do { } while (0)
#endif

struct Launch {
  const Role* roles;                       // [layers][R_KINDS]
  int cum[R_KINDS + 1];                    // block offsets of the roles inside a layer
  int* counters;
  int* err;                                // 0, or (1 + role index) << 8 | kind of the wait that timed out
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mkrs(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0xffffffffu, 0x00020000);
}
template <typename T, int AUX>
__device__ __forceinline__ T ldb_c(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  static_assert(sizeof(T) == 16 || sizeof(T) == 8 || sizeof(T) == 4, "4-, 8- or 16-byte buffer loads");
  if constexpr (sizeof(T) == 16) return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUX));
  else if constexpr (sizeof(T) == 8) return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, AUX));
  else return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, AUX));
}
// aux: 0 (plain) or SC1 (device scope: what this launch's producers wrote); wave-uniform at every call site
template <typename T>
__device__ __forceinline__ T ldb(__amdgpu_buffer_rsrc_t rs, unsigned off, int aux) {
  return aux ? ldb_c<T, 16>(rs, off) : ldb_c<T, 0>(rs, off);
}

// Consumer side: thread 0 polls the producers' counter (relaxed device-scope loads, s_sleep between polls), everybody else parks
// at the barrier.  Returns false when the launch has failed (here or elsewhere): the caller skips its work but still signals.
__device__ __forceinline__ bool wait_dep(const Launch& L, int dep, int target, int code) {
  __shared__ int s_ok;                       // one decision for the whole workgroup (the error word may change between two reads)
  if (threadIdx.x == 0) {
    int ok = 1;
    if (dep >= 0) {
      const long long t0 = wall_clock64();
      while (__hip_atomic_load(L.counters + dep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (__hip_atomic_load(L.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        if (wall_clock64() - t0 > TIMEOUT_TICKS) {
          int expected = 0;
          __hip_atomic_compare_exchange_strong(L.err, &expected, code, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    if (__hip_atomic_load(L.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ok = 0;
    s_ok = ok;
  }
  __syncthreads();
  return s_ok != 0;
}
// Producer side: every wave's stores have left (write-through stores are not in the compiler's vmcnt bookkeeping), then one add.
__device__ __forceinline__ void signal(const Launch& L, int sig) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (sig >= 0 && threadIdx.x == 0) __hip_atomic_fetch_add(L.counters + sig, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------------------------------ GEMM roles
// stream_gemm_body (fragment-order operands, write-through outputs) with the weight ring started BEFORE the dependency wait and
// the activations / residual tile loaded device-scope after it.  Same tiles, K split over the 8 waves, cross-wave sum in wave
// order and epilogues: the results equal the kernels of gemm_stream.hip bit for bit.
template <int KS, int EPI, int MT, typename Wait>
__device__ __forceinline__ void role_gemm(const Args& p, float* __restrict__ red, int bx, int nbx, int by, int bz, Wait&& wait) {
  const int mt0 = bz * MT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int cg0 = (by * NW + wave) * KS;                   // this wave's first global k-step
  const int ntl = (p.tiles - bx + nbx - 1) / nbx;          // tiles this workgroup walks
  auto tile_of = [&](int i) { return bx + min(i, ntl - 1) * nbx; };
  const int ksteps_all = p.K >> 5;
  auto load_w = [&](bf16x8 (&w)[KS], int i) {
    const bf16* wr = p.W + (((size_t)tile_of(i) * ksteps_all + cg0) * 64 + lane) * 8;
#pragma unroll
    for (int c = 0; c < KS; ++c) w[c] = __builtin_nontemporal_load((const bf16x8*)(wr + c * 512));
  };
  struct Epi {
    f32x4 r4, cs, sn;
    bf16x4 b4;
    int pos;
  };
  const __amdgpu_buffer_rsrc_t rsR = mkrs(p.R);
  auto load_epi = [&](int i) {
    Epi ep{};
    if (wave >= MT || i >= ntl) return ep;
    const int t = tile_of(i), m = (mt0 + wave) * 16 + l15;
    if (EPI == E_RESID) {
      const int n = t * 16 + 4 * lq;
      if (m < p.M && n + 4 <= p.N) ep.r4 = ldb<f32x4>(rsR, (unsigned)(((size_t)m * p.ldc + n) * 4), SC1);
    }
    if (EPI == E_QKV) {
      const int rot_tiles = (p.H + p.G) * 8;
      const int mc = min(m, p.M - 1);
      ep.pos = p.pos[mc];
      if (t < rot_tiles) {
        const int c0 = (t & 7) * 8 + 4 * (lq & 1);
        ep.cs = *(const f32x4*)(p.cos_t + (size_t)mc * 64 + c0);
        ep.sn = *(const f32x4*)(p.sin_t + (size_t)mc * 64 + c0);
        if (p.bias) ep.b4 = *(const bf16x4*)(p.bias + (t >> 3) * 128 + c0 + (lq >= 2 ? 64 : 0));
      } else if (p.bias) {
        ep.b4 = *(const bf16x4*)(p.bias + (p.H + p.G) * 128 + (t - rot_tiles) * 16 + 4 * lq);
      }
    }
    return ep;
  };
  auto finish = [&](int i, const Epi& ep) {
    if (wave >= MT || i >= ntl) return;
    const int buf = i & 1, t = tile_of(i);
    f32x4 s = *(const f32x4*)(red + (((buf * NW + 0) * MT + wave) * 64 + lane) * 4);
#pragma unroll
    for (int w2 = 1; w2 < NW; ++w2) s += *(const f32x4*)(red + (((buf * NW + w2) * MT + wave) * 64 + lane) * 4);
    const int rt = mt0 + wave;
    const int m = rt * 16 + l15;
    if (EPI == E_SLAB) {
      float* slab = (float*)p.C + (size_t)by * 64 * p.N;
      st_out<true>((f32x4*)(slab + (size_t)m * p.N + t * 16 + 4 * lq), s);
      return;
    }
    if (EPI == E_SWIGLU) {
      f32x4 u;
#pragma unroll
      for (int r = 0; r < 4; ++r) u[r] = __shfl_xor(s[r], 32, 64);
      if (lq < 2) {
        const int n = t * 8 + 4 * lq;
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16)(bf16_round(silu_f(bf16_round(s[r]))) * bf16_round(u[r]));
        st_out<true>((bf16x4*)((bf16*)p.C + ((((size_t)(n >> 5) * 4 + rt) * 64 + ((n & 31) >> 3) * 16 + l15) << 3) + (n & 7)), o);
      }
      return;
    }
    if (EPI == E_QKV) {
      const int rot_tiles = (p.H + p.G) * 8;
      const int W = p.G * 128;
      bf16* out = (bf16*)p.C + (size_t)m * p.ldc;
      if (t < rot_tiles) {
        const int hh = t >> 3, c0 = (t & 7) * 8 + 4 * (lq & 1);
        const int col = hh * 128 + c0 + (lq >= 2 ? 64 : 0);
        bf16x4 mine;
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[r] = (bf16)(s[r] + (p.bias ? (float)ep.b4[r] : 0.f));
        f32x4 x1, x2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float other = __shfl_xor((float)mine[r], 32, 64);
          x1[r] = lq < 2 ? (float)mine[r] : other;
          x2[r] = lq < 2 ? other : (float)mine[r];
        }
        if (m < p.M) {
          const f32x4 cs = ep.cs, sn = ep.sn;
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16)(lq < 2 ? x1[r] * cs[r] - x2[r] * sn[r] : x2[r] * cs[r] + x1[r] * sn[r]);
          st_out<true>((bf16x4*)(out + col), o);
          if (hh >= p.H) {
            const size_t slot = ((size_t)m * p.ctx + ep.pos) * W;
            st_out<true>((bf16x4*)(p.kc + slot + (hh - p.H) * 128 + c0 + (lq >= 2 ? 64 : 0)), o);
          }
        }
      } else if (m < p.M) {
        const int c = (t - rot_tiles) * 16 + 4 * lq;
        const int col = (p.H + p.G) * 128 + c;
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16)(s[r] + (p.bias ? (float)ep.b4[r] : 0.f));
        st_out<true>((bf16x4*)(out + col), o);
        st_out<true>((bf16x4*)(p.vc + ((size_t)m * p.ctx + ep.pos) * W + c), o);
      }
      return;
    }
    // E_RESID (N % 16 == 0 is a condition of the launch)
    const int n = t * 16 + 4 * lq;
    if (m >= p.M || n >= p.N) return;
    float* dst = (float*)p.C + (size_t)m * p.ldc + n;
    const f32x4 old = ep.r4;
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = old[r] + bf16_round(s[r]);
    st_out<true>((f32x4*)dst, o);
  };
  bf16x8 a[MT][KS];
  auto compute = [&](const bf16x8 (&w)[KS], int i, const Epi* pre) {
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < KS; ++c)
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t] = mfma16(w[c], a[t][c], acc[t]);
    const int buf = i & 1;
#pragma unroll
    for (int t = 0; t < MT; ++t) *(f32x4*)(red + (((buf * NW + wave) * MT + t) * 64 + lane) * 4) = acc[t];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (pre) finish(i, *pre);
    else finish(i, load_epi(i));
  };

  // ---- 1. what does not depend on the producers: the head of the weight ring (+ the q|k|v epilogue's RoPE factors / bias / slot)
  bf16x8 w0[KS], w1[KS], w2[KS];
  load_w(w0, 0);
  load_w(w1, 1);
  if (ntl > 2) load_w(w2, 2);
  Epi e0{}, e1{};
  if (EPI == E_QKV) e0 = load_epi(0), e1 = load_epi(1);
  // ---- 2. the producers
  const bool ok = wait();
  // ---- 3. their output: this wave's K slice of the workgroup's rows, as MFMA operands (fragment order: 1 KiB per wave instruction)
  const __amdgpu_buffer_rsrc_t rsA = mkrs(p.A);
#pragma unroll
  for (int c = 0; c < KS; ++c)
#pragma unroll
    for (int t = 0; t < MT; ++t) a[t][c] = ldb<bf16x8>(rsA, (unsigned)((((size_t)(cg0 + c) * 4 + mt0 + t) * 64 + lane) * 16), SC1);
  if (!ok) return;
  if (ntl <= 2) {
    if (EPI == E_RESID) e0 = load_epi(0), e1 = load_epi(1);
    compute(w0, 0, &e0);
    if (ntl == 2) compute(w1, 1, &e1);
    return;
  }
  // three tiles in flight per wave, branch-free body (the spare bodies of the last round re-read the last tile and skip their
  // epilogue) -- the ring of stream_gemm_body with its first three loads moved in front of the wait
  for (int i = 0; i < ntl; i += 3) {
    compute(w0, i, nullptr);
    load_w(w0, i + 3);
    compute(w1, i + 1, nullptr);
    load_w(w1, i + 4);
    compute(w2, i + 2, nullptr);
    load_w(w2, i + 5);
  }
}

// ------------------------------------------------------------------------------------------------------------------ row roles
// norm_row_frag_ptr / finish_norm_row (stream_body.h) with device-scope loads of what this launch's producers wrote.
template <int NG>
__device__ __forceinline__ void role_norm_row(const float* __restrict__ x, const float* __restrict__ w, bf16* __restrict__ y, int row,
                                              float eps) {
  constexpr int D = NG * 256;
  const int lane = threadIdx.x & 63;
  const __amdgpu_buffer_rsrc_t rs = mkrs(x);
  f32x4 v[NG], gw[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) v[g] = ldb<f32x4>(rs, (unsigned)(((size_t)row * D + lane * 4 + g * 256) * 4), SC1);
#pragma unroll
  for (int g = 0; g < NG; ++g) gw[g] = *(const f32x4*)(w + lane * 4 + g * 256);
  float ss = 0.f;
#pragma unroll
  for (int g = 0; g < NG; ++g) ss += v[g][0] * v[g][0] + v[g][1] * v[g][1] + v[g][2] * v[g][2] + v[g][3] * v[g][3];
  ss = wave_sum(ss);
  const float r = rsqrtf(ss / (float)D + eps);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = gw[g][j] * (v[g][j] * r);
    st_out<true>((bf16x4*)(y + frag_index(row, lane * 4 + g * 256)), __builtin_convertvector(o, bf16x4));
  }
}

template <int NG>
__device__ __forceinline__ void role_finish_row(const float* __restrict__ slabs, int ksplit, float* __restrict__ C,
                                                const float* __restrict__ R, const float* __restrict__ nw, bf16* __restrict__ y,
                                                float eps, int m) {
  constexpr int N = NG * 256;
  const int lane = threadIdx.x & 63;
  const __amdgpu_buffer_rsrc_t rsS = mkrs(slabs), rsR = mkrs(R);
  constexpr int KC = 4;
  f32x4 v[NG], w[NG], s[NG];
  unsigned e[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int n = lane * 4 + g * 256;
    e[g] = (unsigned)(((size_t)m * N + n) * 4);
    v[g] = ldb<f32x4>(rsR, e[g], SC1);
    w[g] = *(const f32x4*)(nw + n);
    s[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int k0 = 0; k0 < ksplit; k0 += KC) {
    f32x4 t[NG][KC];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int j = 0; j < KC; ++j)
        t[g][j] = k0 + j < ksplit ? ldb<f32x4>(rsS, (unsigned)((size_t)(k0 + j) * 64 * N * 4) + e[g], SC1) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int j = 0; j < KC; ++j) s[g] += t[g][j];
  }
  float ss = 0.f;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int n = lane * 4 + g * 256;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[g][q] = v[g][q] + bf16_round(s[g][q]);
    st_out<true>((f32x4*)(C + (size_t)m * N + n), v[g]);
    ss += v[g][0] * v[g][0] + v[g][1] * v[g][1] + v[g][2] * v[g][2] + v[g][3] * v[g][3];
  }
  ss = wave_sum(ss);
  const float rs = rsqrtf(ss / (float)N + eps);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int n = lane * 4 + g * 256;
    f32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = w[g][q] * (v[g][q] * rs);
    st_out<true>((bf16x4*)(y + frag_index(m, n)), __builtin_convertvector(o, bf16x4));
  }
}

// ------------------------------------------------------------------------------------------------------------------ attention role
// attn_decode_body (attn_decode_body.h) with everything that belongs to OLD positions -- the row index, the K chunks and V blocks
// of the cache -- requested before the wait for this position's q|k|v, and the newest position's K chunk / V block and the query
// rows loaded device-scope after it.  Same phases, arithmetic and output layout: the same bits.
template <int REP, typename Wait>
__device__ __forceinline__ void role_attn(float* sp, int row, int g, const Role& r, Wait&& wait) {
  using namespace tasu_attn_dec;
  const int H = r.H, G = r.G, ctx = r.ctx;
  const float scale = r.scale;
  const int ctxp = (ctx + 31) & ~31;
  float* sc = sp;
  float* linv = sc + rup4(REP * ctx);
  int* prow = (int*)(linv + rup4(REP + 1));
  bf16* pb = (bf16*)(prow + rup4(ctx));
  float* part = (float*)pb + rup4(REP * ctxp / 2);
  char* vimg = (char*)part + (threadIdx.x >> 6) * 8192;
  const int W = G * HD, LD = (H + 2 * G) * HD;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, lq = lane >> 4;
  for (int i = threadIdx.x; i < ctx; i += 64 * DEC_NW) prow[i] = r.index ? r.index[(size_t)row * ctx + i] : row;
  const int k0 = r.kstart[row], nk = r.lens[row] - k0;
  __syncthreads();
  prow += k0;
  const __amdgpu_buffer_rsrc_t rsK = mkrs(r.kc), rsV = mkrs(r.vc), rsQ = mkrs(r.qkv);
  const int c_new = (nk - 1) >> 4, b_new = (nk - 1) >> 5;       // the chunk / block that holds this position's own K / V
  constexpr int VPRE = 2;
  const int nblk32 = (nk + 31) >> 5;
  auto load_vblock = [&](bf16x8 (&v)[8], int blk, int aux) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kcl = min(blk * 32 + 4 * i + lq, nk - 1);
      v[i] = ldb<bf16x8>(rsV, (unsigned)((((size_t)prow[kcl] * ctx + k0 + kcl) * W + g * HD + l15 * 8) * 2), aux);
    }
  };
  bf16x8 vpre[VPRE][8];
#pragma unroll
  for (int it = 0; it < VPRE; ++it) {
    const int blk = wave + it * DEC_NW;
    if (blk < nblk32 && blk != b_new) load_vblock(vpre[it], blk, 0);
  }
  const int nchunk = (nk + 15) >> 4;
  constexpr int KPRE = 3;
  auto load_kchunk = [&](bf16x8 (&kf)[4], int c, int aux) {
    const int kcl = min(c * 16 + l15, nk - 1);
    const unsigned off = (unsigned)((((size_t)prow[kcl] * ctx + k0 + kcl) * W + g * HD + lq * 8) * 2);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) kf[s4] = ldb<bf16x8>(rsK, off + s4 * 64, aux);
  };
  bf16x8 kpre[KPRE][4];
#pragma unroll
  for (int it = 0; it < KPRE; ++it) {
    const int c = wave + it * DEC_NW;
    if (c < nchunk && c != c_new) load_kchunk(kpre[it], c, 0);
  }
  // ---- this position's q|k|v
  const bool ok = wait();
  bf16x8 qf[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    qf[s4] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (l15 < REP) qf[s4] = ldb<bf16x8>(rsQ, (unsigned)((((size_t)row * LD + (g * REP + l15) * HD + s4 * 32 + lq * 8)) * 2), SC1);
  }
#pragma unroll
  for (int it = 0; it < KPRE; ++it) {
    const int c = wave + it * DEC_NW;
    if (c < nchunk && c == c_new) load_kchunk(kpre[it], c, SC1);
  }
#pragma unroll
  for (int it = 0; it < VPRE; ++it) {
    const int blk = wave + it * DEC_NW;
    if (blk < nblk32 && blk == b_new) load_vblock(vpre[it], blk, SC1);
  }
  if (!ok) return;
  // ---- phase 1: scores
  auto scores = [&](const bf16x8 (&kf)[4], int c) {
    const int key = c * 16 + l15;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) acc = mfma16(qf[s4], kf[s4], acc);
    if (key < nk) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        if (lq * 4 + rr < REP) sc[(lq * 4 + rr) * ctx + key] = acc[rr] * scale;
    }
  };
#pragma unroll
  for (int it = 0; it < KPRE; ++it)
    if (wave + it * DEC_NW < nchunk) scores(kpre[it], wave + it * DEC_NW);
  for (int c = wave + KPRE * DEC_NW; c < nchunk; c += DEC_NW) {
    bf16x8 kf[4];
    load_kchunk(kf, c, c == c_new ? SC1 : 0);
    scores(kf, c);
  }
  __syncthreads();
  // ---- phase 1b: softmax statistics of head h
  for (int h = wave; h < REP; h += DEC_NW) {
    constexpr int SU = 4;
    float x[SU];
    float m = -__builtin_inff();
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int i = lane + 64 * u;
      x[u] = i < nk ? sc[h * ctx + i] : -__builtin_inff();
      m = fmaxf(m, x[u]);
    }
    for (int i = lane + 64 * SU; i < nk; i += 64) m = fmaxf(m, sc[h * ctx + i]);
    m = wave_max(m);
    float l = 0.f;
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int i = lane + 64 * u;
      if (i < nblk32 * 32) {
        const float pp = i < nk ? __expf(x[u] - m) : 0.f;
        pb[h * ctxp + i] = (bf16)pp;
        l += pp;
      }
    }
    for (int i = lane + 64 * SU; i < nblk32 * 32; i += 64) {
      const float pp = i < nk ? __expf(sc[h * ctx + i] - m) : 0.f;
      pb[h * ctxp + i] = (bf16)pp;
      l += pp;
    }
    l = wave_sum(l);
    if (lane == 0) linv[h] = l > 0.f ? 1.f / l : 0.f;
  }
  __syncthreads();
  // ---- phase 2: P.V
  f32x4 o[8];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) o[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto block_pv = [&](const bf16x8 (&v)[8], int blk) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rr = 4 * i + lq;
      *(bf16x8*)(vimg + rr * 256 + ((l15 ^ (rr & 15)) << 4)) = v[i];
    }
    bf16x8 pf = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (l15 < REP) {
      const bf16x4 lo = *(const bf16x4*)(pb + l15 * ctxp + blk * 32 + 4 * lq), hi = *(const bf16x4*)(pb + l15 * ctxp + blk * 32 + 16 + 4 * lq);
      pf = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) o[nt] = mfma16(v_frag_tr(vimg, nt, lane), pf, o[nt]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
#pragma unroll
  for (int it = 0; it < VPRE; ++it)
    if (wave + it * DEC_NW < nblk32) block_pv(vpre[it], wave + it * DEC_NW);
  for (int blk = wave + VPRE * DEC_NW; blk < nblk32; blk += DEC_NW) {
    bf16x8 v[8];
    load_vblock(v, blk, blk == b_new ? SC1 : 0);
    block_pv(v, blk);
  }
  __syncthreads();
  if (l15 < REP) {
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) *(f32x4*)(part + (wave * REP + l15) * HD + nt * 16 + 4 * lq) = o[nt];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < REP * HD / 4; e += 64 * DEC_NW) {
    const int h = e / (HD / 4), d = (e - h * (HD / 4)) * 4;
    f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w2 = 0; w2 < DEC_NW; ++w2) sum += *(const f32x4*)(part + (w2 * REP + h) * HD + d);
    const int n = (g * REP + h) * HD + d;
    const size_t off = ((((size_t)(n >> 5) * 4 + ((row & 63) >> 4)) * 64 + ((n & 31) >> 3) * 16 + (row & 15)) << 3) + (n & 7);
    const float li = linv[h];
    bf16x4 ov;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) ov[rr] = (bf16)(sum[rr] * li);
    st_out<true>((bf16x4*)(r.ao + off), ov);
  }
}

// ------------------------------------------------------------------------------------------------------------------ the kernel
// KSD: k-steps per wave of the hidden-size contractions (D = H * 128 = 256 * KSD); KSDOWN: of one K range of the down projection
// (I = ksplit * 256 * KSDOWN); NG = D / 256 column groups of the row roles; REP = query heads per kv head.
template <int KSD, int KSDOWN, int NG, int REP>
__global__ __launch_bounds__(64 * NW, 2) void decode_roles_kernel(Launch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int per_layer = L.cum[R_KINDS];
  const int layer = (int)blockIdx.x / per_layer, w = (int)blockIdx.x - layer * per_layer;
  int kind = 0;
#pragma unroll
  for (int k = 1; k < R_KINDS; ++k) kind += w >= L.cum[k];
  const int local = w - L.cum[kind];
  TASU_ROLES_STAMP(0);
  const Role& r = L.roles[layer * R_KINDS + kind];
  // blocks of a role beyond its work (grids are rounded up to whole XCD groups) leave at once and are not counted
  int bx = local, by = 0, bz = 0;
  const bool gemm = kind == R_QKV || kind == R_O || kind == R_DOWN;
  if (gemm ? !grid_position(local, r.a.gx, r.a.gy, r.a.gz, bx, by, bz) : local >= r.work) return;
  const int code = ((layer * R_KINDS + kind + 1) << 8) | kind;
  auto wait = [&]() {
    const bool ok = wait_dep(L, r.dep, r.dep_target, code);
    TASU_ROLES_STAMP(1);
    return ok;
  };
  switch (kind) {
    case R_QKV:
      role_gemm<KSD, E_QKV, 2>(r.a, smem, bx, r.a.gx, by, bz, wait);
      break;
    case R_ATTN:
      role_attn<REP>(smem, local / r.G, local % r.G, r, wait);
      break;
    case R_O:
      role_gemm<KSD, E_RESID, 2>(r.a, smem, bx, r.a.gx, by, bz, wait);
      break;
    case R_NORM: {
      const bool ok = wait();
      const int row = local * NW + (threadIdx.x >> 6);
      if (ok && row < r.M) role_norm_row<NG>(r.x_in, r.nw, r.y, row, r.eps);
      break;
    }
    case R_GU:
      role_gemm<KSD, E_SWIGLU, 4>(r.a, smem, local, r.a.gx, 0, 0, wait);
      break;
    case R_DOWN:
      role_gemm<KSDOWN, E_SLAB, 2>(r.a, smem, bx, r.a.gx, by, bz, wait);
      break;
    default: {                                            // R_FIN
      const bool ok = wait();
      const int row = local * NW + (threadIdx.x >> 6);
      if (ok && row < r.M) role_finish_row<NG>(r.slabs, r.ksplit, r.c_out, r.x_in, r.nw, r.y, r.eps, row);
      break;
    }
  }
  TASU_ROLES_STAMP(2);
  signal(L, r.sig);
  TASU_ROLES_STAMP(3);
}

// virtual grid of a GEMM role on `cus` compute units (launch() of gemm_stream.hip with the row split fixed per role)
static void plan(Args& a, int ksplit, int zs, int cus, int& blocks, int& work) {
  const int per_split = cus / (ksplit * zs) > 0 ? cus / (ksplit * zs) : 1;
  a.gx = a.tiles < per_split ? a.tiles : per_split;
  a.gy = ksplit, a.gz = zs;
  work = a.gx * a.gy * a.gz;
  const int per = 8 * zs;
  blocks = (work + per - 1) / per * per;                  // whole groups of (8 XCDs x row halves): grid_position
}

}  // namespace tasu_roles

namespace tasu_stream {
int cu_count();
}

extern "C" int tasu_decode_roles_supported(int M, int D, int I, int H, int G) {
  if (M <= 32 || M > 64 || G <= 0 || H % G || H * 128 != D) return 0;
  const int rep = H / G;
  const bool geo15 = D == 1536 && I == 5 * 1792 && rep == 6;      // Qwen2.5-1.5B
  const bool geot = D == 256 && I == 5 * 256 && rep == 2;         // the small geometry of tests/test_gpu_decode_roles.py
  return (geo15 || geot) ? 1 : 0;
}

extern "C" int64_t tasu_decode_roles_table_bytes(int layers) { return (int64_t)layers * tasu_roles::R_KINDS * sizeof(tasu_roles::Role); }
// Builds the role table of `layers` decoder layers in HOST memory `table` (tasu_decode_roles_table_bytes), which the caller
// uploads once per (model, workspace): every pointer below is a DEVICE address that stays valid for the table's life.
extern "C" int tasu_decode_roles_build(void* table, int layers, const tasu_decode_layer_t* lw, const tasu_decode_ws_t* ws, int M, int D,
                                       int I, int H, int G, int ctx, float eps, float scale, int32_t* cum_out) {
  using namespace tasu_roles;
  if (!table || !lw || !ws || !cum_out || layers <= 0 || ctx <= 0 || ctx > tasu_attn_dec::MAX_CTX) return TASU_ERR_ARG;
  if (!tasu_decode_roles_supported(M, D, I, H, G)) return TASU_ERR_ARG;
  // A role's grid is sized for HALF of the compute units: with one 512-thread workgroup per CU (the GEMM roles hold their
  // operands in ~240 registers) the next role's workgroups are only resident -- and their weight tiles in flight -- while the
  // current role computes if that role leaves room for them.  (Sized for all CUs every role filled the chip by itself, the roles
  // ran one after the other like launches, and the position took 2.56 ms against 1.78 for the launches.)
  int cus = tasu_stream::cu_count() / 2;
  if (const char* e = getenv("TASU_ROLES_CUS")) cus = atoi(e) > 0 ? atoi(e) : cus;     // (lab)
  const int LDQ = (H + 2 * G) * 128;
  Role* out = (Role*)table;
  const int64_t sx = (int64_t)64 * D, sq = (int64_t)64 * LDQ, sa = (int64_t)64 * I, ss = (int64_t)5 * 64 * D;   // per-layer slices
  int cum[R_KINDS + 1] = {0};
  for (int l = 0; l < layers; ++l) {
    const tasu_decode_layer_t& w = lw[l];
    const bf16 *xn_l = (const bf16*)ws->xn + l * sx, *xn_n = (const bf16*)ws->xn + (l + 1) * sx, *xn2_l = (const bf16*)ws->xn2 + l * sx;
    const float *x_l = ws->x + l * sx, *x_n = ws->x + (l + 1) * sx, *x2_l = ws->x2 + l * sx;
    const bf16 *qkv_l = (const bf16*)ws->qkv + l * sq, *ao_l = (const bf16*)ws->ao + l * sx, *act_l = (const bf16*)ws->act + l * sa;
    const float* slabs_l = ws->slabs + l * ss;
    Role* r = out + (size_t)l * R_KINDS;
    for (int k = 0; k < R_KINDS; ++k) {
      r[k] = Role{};
      r[k].eps = eps, r[k].scale = scale, r[k].H = H, r[k].G = G, r[k].ctx = ctx, r[k].M = M;
      r[k].dep = k == 0 ? (l == 0 ? -1 : (l - 1) * R_KINDS + R_FIN) : l * R_KINDS + k - 1;
      r[k].sig = l * R_KINDS + k;
    }
    int blocks[R_KINDS];
    // q|k|v + bias + RoPE + cache append
    Args& q = r[R_QKV].a;
    q.A = xn_l, q.W = (const bf16*)w.wqkv, q.C = (void*)qkv_l, q.bias = (const bf16*)w.bqkv;
    q.M = M, q.N = LDQ, q.K = D, q.ldc = LDQ, q.H = H, q.G = G, q.ctx = ctx;
    q.cos_t = ws->cos_tab, q.sin_t = ws->sin_tab, q.kc = (bf16*)w.kcache, q.vc = (bf16*)w.vcache, q.pos = ws->slot;
    q.tiles = (H + 2 * G) * 8, q.a_frag = q.w_frag = 1;
    plan(q, 1, 2, cus, blocks[R_QKV], r[R_QKV].work);
    // cache attention
    Role& at = r[R_ATTN];
    at.qkv = qkv_l, at.kc = (const bf16*)w.kcache, at.vc = (const bf16*)w.vcache, at.index = ws->index, at.kstart = ws->kstart;
    at.lens = ws->lens, at.ao = (bf16*)ao_l;
    at.work = M * G, blocks[R_ATTN] = (at.work + 7) / 8 * 8;
    // o projection + residual
    Args& o = r[R_O].a;
    o.A = ao_l, o.W = (const bf16*)w.wo, o.C = (void*)x2_l, o.R = x_l, o.M = M, o.N = D, o.K = H * 128, o.ldc = D;
    o.tiles = D / 16, o.a_frag = o.w_frag = 1;
    plan(o, 1, 2, cus, blocks[R_O], r[R_O].work);
    // post-attention norm
    Role& nm = r[R_NORM];
    nm.x_in = x2_l, nm.nw = (const float*)w.ln2, nm.y = (bf16*)xn2_l;
    nm.work = (M + 7) / 8, blocks[R_NORM] = 8;
    // gate|up + SwiGLU
    Args& gu = r[R_GU].a;
    gu.A = xn2_l, gu.W = (const bf16*)w.wgu, gu.C = (void*)act_l, gu.M = M, gu.N = I, gu.K = D, gu.ldc = I, gu.I = I;
    gu.tiles = I / 8, gu.a_frag = gu.w_frag = gu.out_frag = 1;
    plan(gu, 1, 1, cus, blocks[R_GU], r[R_GU].work);
    // down projection: K-range slabs
    Args& dn = r[R_DOWN].a;
    dn.A = act_l, dn.W = (const bf16*)w.wd, dn.C = (void*)slabs_l, dn.M = M, dn.N = D, dn.K = I, dn.ldc = D;
    dn.tiles = D / 16, dn.a_frag = dn.w_frag = 1;
    plan(dn, 5, 2, cus, blocks[R_DOWN], r[R_DOWN].work);
    // slab sum + residual + the next layer's (or the final) norm
    Role& fn = r[R_FIN];
    fn.slabs = slabs_l, fn.ksplit = 5, fn.c_out = (float*)x_n, fn.x_in = x2_l, fn.nw = (const float*)w.next_norm, fn.y = (bf16*)xn_n;
    fn.work = (M + 7) / 8, blocks[R_FIN] = 8;
    for (int k = 0; k < R_KINDS; ++k) {
      r[k].dep_target = r[k].dep < 0 ? 0 : out[r[k].dep].work;
      if (l == 0) cum[k + 1] = cum[k] + blocks[k];
    }
  }
  for (int k = 0; k <= R_KINDS; ++k) cum_out[k] = cum[k];
  return TASU_OK;
}

extern "C" int tasu_decode_roles_run(const void* table_dev, int layers, const int32_t* cum, int32_t* counters, int32_t* err, int M, int D,
                                     int I, int H, int G, int ctx, void* stream) {
  using namespace tasu_roles;
  if (!table_dev || !cum || !counters || !err || layers <= 0 || !tasu_decode_roles_supported(M, D, I, H, G)) return TASU_ERR_ARG;
  Launch L{};
  L.roles = (const Role*)table_dev;
  for (int k = 0; k <= R_KINDS; ++k) L.cum[k] = cum[k];
  L.counters = counters, L.err = err;
  const int rep = H / G;
  const int attn_bytes = tasu_attn_dec::attn_decode_lds_floats(rep, ctx) * 4;
  const int lds = attn_bytes > 65536 ? attn_bytes : 65536;
  if (lds > 160 * 1024) return TASU_ERR_ARG;
  const dim3 grid(layers * cum[R_KINDS]);
  hipStream_t st = (hipStream_t)stream;
#define TASU_ROLES_LAUNCH(KSD, KSDOWN, NG, REP)                                                                                       \
  do {                                                                                                                                 \
    static int lds_set = 0;                                                                                                            \
    if (lds > lds_set) {                                                                                                               \
      if (hipFuncSetAttribute((const void*)decode_roles_kernel<KSD, KSDOWN, NG, REP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != \
          hipSuccess)                                                                                                                  \
        return TASU_ERR_LAUNCH;                                                                                                        \
      lds_set = lds;                                                                                                                   \
    }                                                                                                                                  \
    TASU_LAUNCH((decode_roles_kernel<KSD, KSDOWN, NG, REP>), grid, dim3(64 * NW), lds, st, L);                                         \
  } while (0)
  if (D == 1536) TASU_ROLES_LAUNCH(6, 7, 6, 6);
  else TASU_ROLES_LAUNCH(1, 1, 1, 2);
#undef TASU_ROLES_LAUNCH
  return TASU_OK;
}

#ifdef TASU_ROLES_TRACE
extern "C" int tasu_roles_trace_read(uint64_t* host_out, int n) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tasu_roles::g_roles_trace), (size_t)n * sizeof(uint64_t)) == hipSuccess ? 0 : 2;
}
#endif
