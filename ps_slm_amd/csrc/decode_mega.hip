// The decode step's layer loop as ONE persistent launch (M <= 64 beam rows).
//
// Launched one GEMM at a time, a generated position of the 1.5B decoder is ~200 launches of 5-17 us, and a launch on this chip
// costs ~4.7 us before it does anything (measured: the empty-handed kernels of the step -- embedding gather, RoPE table, index
// reorder -- all take 4.7-5.0 us in a hipGraph replay); the seven dependent launches of a layer put ~33 us of that under 19 us
// of weight streaming.  Here one workgroup per CU stays resident for the whole layer loop and the dependent steps of a layer
// become PHASES separated by a grid barrier:
//     qkv (+bias, RoPE, cache append) | cache attention | o (+residual) | RMSNorm | gate|up (+SwiGLU) | down (K-range slabs)
//     | slab sum + residual + next RMSNorm
// with the bodies of the one-GEMM kernels (stream_body.h, attn_decode_body.h: same arithmetic, same rounding points, same
// summation orders -- tests/test_gpu_ops.py compares the two paths bit for bit).
//
// GRID BARRIER (tools/micro/grid_barrier*.hip measured the alternatives on MI355X): one counter for 256 workgroups costs 3.7 us
// (the 256 atomics serialise at the memory side), every workgroup polling 256 flags 4.6 us; a TWO-LEVEL counter -- 16 groups of
// 16 arrive on their group's counter, the last arriver of a group on the top counter, the last of those publishes the epoch word
// everybody polls -- costs 1.6-1.7 us.  Epochs and counters only ever grow (no reset between launches: a launch reads the epoch
// it starts from).
//
// VISIBILITY WITHOUT FENCES.  The XCDs' L2s are not coherent with each other; the release/acquire fences that make plain stores
// visible (buffer_wbl2 / buffer_inv) cost 2 us per barrier when one wave issues them and 15+ us when every wave does.  Instead:
//   * every result another workgroup reads is stored WRITE-THROUGH at agent scope (st_out<true>, global_store ... sc1) and the
//     barrier waits for those stores (vmcnt(0)) before it arrives;
//   * every such address is written exactly ONCE per launch and only read after the barrier that follows its write: the
//     workspace has a slice per layer, and a cache slot belongs to one position -- so no cache can hold an older copy of a line
//     when it is first read (caches start invalidated at kernel launch), and the readers use plain loads that the L2 of their
//     XCD then serves to its other workgroups (the 64 x K activations are read by every workgroup of a GEMM phase);
//   * buffers of different phases never share a 128-byte line.
#include "common.h"
#include "stream_body.h"
#include "attn_decode_body.h"
#include "../../include/tasu_hip.h"

namespace tasu_mega {
using namespace tasu_stream;
using tasu_attn_dec::attn_decode_body;
using tasu_attn_dec::attn_decode_lds_floats;

constexpr int GROUPS = 16;           // barrier groups
constexpr int SYNC_WORDS = 64 + 32 * GROUPS;   // [0] epoch, [1] timeouts, [32] top counter, [64 + 32 g] group counters (own lines)

struct Args {
  const tasu_decode_layer* layers;   // device array [L]
  int L, M, H, G, ksplit, ctx;
  float eps, scale;
  const float* x0;                   // [M, D] fp32: the embeddings of this position's tokens
  const float* final_norm;           // [D]
  bf16* xn_out;                      // fragment order [D / 32][4][64][8]: final-normed hidden state (lm_head's A operand)
  unsigned char* ws;                 // workspace: L slices (ws_layout)
  const float* cos_t;
  const float* sin_t;
  const int32_t* slot;               // [M] cache position this step appends at
  const int32_t* row_index;          // [M, ctx] beam row index of the cache
  const int32_t* kstart;
  const int32_t* lens;
  unsigned* sync;                    // SYNC_WORDS words, zeroed once at allocation
  unsigned long long* trace;         // optional (tasu_decode_layers_set_trace): [barrier][workgroup][enter, exit] wall-clock ticks
  int trace_barriers;
};

__host__ __device__ constexpr size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
// one layer's slice of the workspace (byte offsets)
struct Layout {
  size_t xn, qkv, ao, x2, xn2, act, slabs, x, total;
};
__host__ __device__ inline Layout ws_layout(int D, int H, int G, int I, int ksplit) {
  Layout o{};
  size_t p = 0;
  o.xn = p, p += align256((size_t)64 * D * 2);
  o.qkv = p, p += align256((size_t)64 * (H + 2 * G) * 128 * 2);
  o.ao = p, p += align256((size_t)64 * H * 128 * 2);
  o.x2 = p, p += align256((size_t)64 * D * 4);
  o.xn2 = p, p += align256((size_t)64 * D * 2);
  o.act = p, p += align256((size_t)64 * I * 2);
  o.slabs = p, p += align256((size_t)ksplit * (D / 16) * 1024 * 4);
  o.x = p, p += align256((size_t)64 * D * 4);
  o.total = p;
  return o;
}

// Two-level counter barrier over the nblk resident workgroups.  `dead`: a workgroup that once timed out (a workgroup of the grid
// is not resident -- cannot happen when the grid is no larger than the CU count and nothing else runs; the guard turns a
// would-be hang into an error the host reports) stops waiting.
__device__ __forceinline__ void grid_barrier(unsigned* st, unsigned& epoch, int nblk, int* dead, unsigned long long* trace = nullptr,
                                             int trace_slot = 0) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's write-through stores are acknowledged
  __syncthreads();
  ++epoch;
  if (trace && threadIdx.x == 0) trace[((size_t)trace_slot * nblk + blockIdx.x) * 2] = wall_clock64();
  if (threadIdx.x == 0 && !*dead) {
    const int b = blockIdx.x, g = b % GROUPS, gsize = (nblk - g + GROUPS - 1) / GROUPS, groups = nblk < GROUPS ? nblk : GROUPS;
    if (__hip_atomic_fetch_add(&st[64 + 32 * g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch * gsize - 1) {
      if (__hip_atomic_fetch_add(&st[32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch * groups - 1)
        __hip_atomic_store(&st[0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(&st[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) {
      if (wall_clock64() - t0 > 30000000ll) {            // 0.3 s of the 100 MHz wall clock
        __hip_atomic_fetch_add(&st[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *dead = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  if (trace && threadIdx.x == 0) trace[((size_t)trace_slot * nblk + blockIdx.x) * 2 + 1] = wall_clock64();
  __syncthreads();
}

// REGISTER LIVENESS.  Inlined naively, the compiler hoists every phase's loop-invariant addresses and kernel arguments out of
// the layer loop and keeps them alive across all phases: 256 VGPRs + 225 spilled (SGPR spills land in VGPR lanes).  As real
// function calls the arguments travel in VGPRs, i.e. stop being wave-uniform, and the GEMM bodies slow down 1.6x.  So the phases
// are inlined, but each one re-reads what it needs from the kernel-argument segment through a pointer the compiler cannot see
// through (fresh()): nothing a phase computes can be hoisted above its own start, and each phase gets the register allocation
// of its stand-alone kernel.  Pointers loaded from the layer table are made wave-uniform again with readfirstlane (uni()).
typedef const __attribute__((address_space(4))) Args KArgs;
__device__ __forceinline__ KArgs* fresh(KArgs* q) {
  asm volatile("" : "+s"(q));
  return q;
}
template <typename T>
__device__ __forceinline__ T* uni(T* ptr) {
  const unsigned long long v = (unsigned long long)ptr;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
}

// D = KSD * 256 (= H * 128), I = ksplit * KSI * 256, REP = H / G
template <int KSD, int KSI, int REP>
__global__ __launch_bounds__(64 * NW) void decode_layers_kernel(Args p_unused) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int dead;
  constexpr int D = KSD * 256;
  KArgs* const kargs = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();      // the Args above, at offset 0 of the segment
  const int nblk = gridDim.x, b = blockIdx.x;
  const int wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) dead = 0;
  unsigned epoch = __hip_atomic_load(&kargs->sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // nobody publishes before all arrived
  const int L = kargs->L;
  int n_bar = 0;
  auto barrier = [&]() {
    unsigned long long* tr = kargs->trace;
    grid_barrier(kargs->sync, epoch, nblk, &dead, tr && n_bar < kargs->trace_barriers ? tr : nullptr, n_bar);
    ++n_bar;
  };
  __syncthreads();

  // ---- phase 0: xn_0 = norm(x0) with layer 0's input norm (one row per workgroup: the rows' loads run on 64 CUs)
  {
    KArgs* p = fresh(kargs);
    if (wave == 0 && b < p->M) norm_row_frag<KSD, true>(p->x0, uni(p->layers[0].ln1), (bf16*)p->ws /* lay.xn == 0 */, b, p->eps);
  }
  barrier();

  for (int l = 0; l < L; ++l) {
    // ---- q|k|v projection + bias + RoPE + cache append
    {
      KArgs* p = fresh(kargs);
      const int I = p->ksplit * KSI * 256;
      const Layout lay = ws_layout(D, p->H, p->G, I, p->ksplit);
      unsigned char* wl = p->ws + (size_t)l * lay.total;
      const tasu_decode_layer* w = p->layers + l;
      const int zs = p->M > 32 ? 2 : 1;                    // row splits of the MT = 2 phases
      tasu_stream::Args a{};
      a.A = (const bf16*)(wl + lay.xn), a.W = uni((const bf16*)w->wqkv), a.C = wl + lay.qkv, a.bias = uni((const bf16*)w->bqkv);
      a.M = p->M, a.N = (p->H + 2 * p->G) * 128, a.K = D, a.ldc = a.N;
      a.H = p->H, a.G = p->G, a.ctx = p->ctx;
      a.cos_t = p->cos_t, a.sin_t = p->sin_t, a.kc = uni((bf16*)w->kcache), a.vc = uni((bf16*)w->vcache), a.pos = p->slot;
      a.tiles = (p->H + 2 * p->G) * 8;
      const int nbx = min(a.tiles, (nblk & ~15) / zs);
      int bx, by, bz;
      if (grid_position(b, nbx, 1, zs, bx, by, bz)) stream_gemm_body<KSD, E_QKV, 2, true, true>(a, lds, bx, nbx, 0, bz);
    }
    barrier();

    // ---- cache attention: one (row, kv group) per workgroup
    {
      KArgs* p = fresh(kargs);
      const int I = p->ksplit * KSI * 256;
      const Layout lay = ws_layout(D, p->H, p->G, I, p->ksplit);
      unsigned char* wl = p->ws + (size_t)l * lay.total;
      const tasu_decode_layer* w = p->layers + l;
      const bf16* kc = uni((const bf16*)w->kcache);
      const bf16* vc = uni((const bf16*)w->vcache);
      for (int item = b; item < p->M * p->G; item += nblk) {
        attn_decode_body<REP, true>(lds, item / p->G, item % p->G, (const bf16*)(wl + lay.qkv), kc, vc, p->row_index, p->kstart,
                                    p->lens, (bf16*)(wl + lay.ao), p->H, p->G, p->ctx, p->scale, 1);
        __syncthreads();
      }
    }
    barrier();

    // ---- o projection + residual
    {
      KArgs* p = fresh(kargs);
      const int I = p->ksplit * KSI * 256;
      const Layout lay = ws_layout(D, p->H, p->G, I, p->ksplit);
      unsigned char* wl = p->ws + (size_t)l * lay.total;
      const tasu_decode_layer* w = p->layers + l;
      const int zs = p->M > 32 ? 2 : 1;
      tasu_stream::Args a{};
      a.A = (const bf16*)(wl + lay.ao), a.W = uni((const bf16*)w->wo), a.C = wl + lay.x2;
      a.R = l == 0 ? p->x0 : (const float*)(wl + lay.x);
      a.M = p->M, a.N = D, a.K = D, a.ldc = D;
      a.tiles = D / 16;
      const int nbx = min(a.tiles, (nblk & ~15) / zs);
      int bx, by, bz;
      if (grid_position(b, nbx, 1, zs, bx, by, bz)) stream_gemm_body<KSD, E_RESID, 2, true, true>(a, lds, bx, nbx, 0, bz);
    }
    barrier();

    // ---- post-attention RMSNorm
    {
      KArgs* p = fresh(kargs);
      const int I = p->ksplit * KSI * 256;
      const Layout lay = ws_layout(D, p->H, p->G, I, p->ksplit);
      unsigned char* wl = p->ws + (size_t)l * lay.total;
      if (wave == 0 && b < p->M)
        norm_row_frag<KSD, true>((const float*)(wl + lay.x2), uni(p->layers[l].ln2), (bf16*)(wl + lay.xn2), b, p->eps);
    }
    barrier();

    // ---- gate|up projection + SwiGLU
    {
      KArgs* p = fresh(kargs);
      const int I = p->ksplit * KSI * 256;
      const Layout lay = ws_layout(D, p->H, p->G, I, p->ksplit);
      unsigned char* wl = p->ws + (size_t)l * lay.total;
      tasu_stream::Args a{};
      a.A = (const bf16*)(wl + lay.xn2), a.W = uni((const bf16*)p->layers[l].wgu), a.C = wl + lay.act;
      a.M = p->M, a.N = I, a.K = D, a.ldc = I, a.I = I;
      a.tiles = I / 8;
      a.out_frag = 1;
      const int nbx = min(a.tiles, nblk);
      if (b < nbx) stream_gemm_body<KSD, E_SWIGLU, 4, true, true>(a, lds, b, nbx, 0, 0);
    }
    barrier();

    // ---- down projection: fp32 partial tiles per K range
    {
      KArgs* p = fresh(kargs);
      const int ksplit = p->ksplit, I = ksplit * KSI * 256;
      const Layout lay = ws_layout(D, p->H, p->G, I, ksplit);
      unsigned char* wl = p->ws + (size_t)l * lay.total;
      const int zs = p->M > 32 ? 2 : 1;
      tasu_stream::Args a{};
      a.A = (const bf16*)(wl + lay.act), a.W = uni((const bf16*)p->layers[l].wd), a.C = wl + lay.slabs;
      a.M = p->M, a.N = D, a.K = I, a.ldc = D;
      a.tiles = D / 16;
      const int per_split = max((nblk & ~15) / (ksplit * zs), 1);
      const int nbx = min(a.tiles, per_split);
      int bx, by, bz;
      if (grid_position(b, nbx, ksplit, zs, bx, by, bz)) stream_gemm_body<KSI, E_SLAB, 2, true, true>(a, lds, bx, nbx, by, bz);
    }
    barrier();

    // ---- slab sum + residual + the next layer's input norm (final norm after the last layer); what the layer hands on goes
    // into the next layer's slice, after the last layer into the caller's buffer (x: layer 0's otherwise unused x field)
    {
      KArgs* p = fresh(kargs);
      const int ksplit = p->ksplit, I = ksplit * KSI * 256;
      const Layout lay = ws_layout(D, p->H, p->G, I, ksplit);
      unsigned char* wl = p->ws + (size_t)l * lay.total;
      const bool last = l + 1 == L;
      unsigned char* wn = last ? p->ws : wl + lay.total;
      const float* nw = last ? p->final_norm : uni(p->layers[last ? l : l + 1].ln1);
      bf16* xn_next = last ? p->xn_out : (bf16*)(wn + lay.xn);
      if (wave == 0 && b < p->M)
        finish_norm_row<KSD, true>((const float*)(wl + lay.slabs), ksplit, (float*)(wn + lay.x), (const float*)(wl + lay.x2), nw, xn_next,
                                   p->eps, 1, b);
    }
    barrier();
  }
}

template <int KSD, int KSI, int REP>
int launch(const Args& a, int D, int I, hipStream_t st) {
  const int cus = cu_count();
  auto kern = decode_layers_kernel<KSD, KSI, REP>;
  const size_t lds = sizeof(float) * (size_t)max(2 * NW * 4 * 256, attn_decode_lds_floats(REP, a.ctx));
  static size_t lds_set = 0;
  static int resident = -1;
  if (lds > lds_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return TASU_ERR_LAUNCH;
    lds_set = lds;
    resident = -1;
  }
  if (resident < 0) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64 * NW, lds) != hipSuccess) return TASU_ERR_LAUNCH;
    resident = per_cu;
  }
  if (resident < 1) return TASU_ERR_ARG;                 // the grid barrier needs every workgroup resident
  TASU_LAUNCH(kern, dim3(cus), dim3(64 * NW), lds, st, a);
  return TASU_OK;
}

bool supported(int M, int D, int H, int G, int I, int ctx, int* ksplit) {
  if (M <= 0 || M > 64 || H <= 0 || G <= 0 || H % G || H * 128 != D || ctx <= 0 || ctx > tasu_attn_dec::MAX_CTX) return false;
  int ks = 0;
  if (D == 1536 && H / G == 6 && I % 1792 == 0) ks = I / 1792;
  else if (D == 256 && H / G == 2 && I % 512 == 0) ks = I / 512;
  if (ks < 1 || ks > 8) return false;
  if (ksplit) *ksplit = ks;
  return true;
}

}  // namespace tasu_mega

// debug facility (tools/mega_trace.py): while set, every launch records per workgroup the wall-clock tick (100 MHz) at which it
// entered and left each grid barrier, [barrier][workgroup][2], for as many barriers as the buffer holds
static unsigned long long* g_trace = nullptr;
static int64_t g_trace_words = 0;
extern "C" int tasu_decode_layers_set_trace(uint64_t* device_buf, int64_t words) {
  g_trace = (unsigned long long*)device_buf;
  g_trace_words = device_buf ? words : 0;
  return TASU_OK;
}

extern "C" int tasu_decode_layers_supported(int M, int D, int H, int G, int I, int ctx) {
  return tasu_mega::supported(M, D, H, G, I, ctx, nullptr) ? 1 : 0;
}

extern "C" int64_t tasu_decode_layers_ws_bytes(int L, int D, int H, int G, int I) {
  int ks = 0;
  if (L <= 0 || !tasu_mega::supported(1, D, H, G, I, 1, &ks)) return -1;
  return (int64_t)tasu_mega::ws_layout(D, H, G, I, ks).total * L;
}

extern "C" int tasu_decode_layers_sync_words(void) { return tasu_mega::SYNC_WORDS; }

extern "C" int tasu_decode_layers(const tasu_decode_layer* layers, int L, const float* x0, const float* final_norm, void* xn_out,
                                  void* ws, int64_t ws_bytes, uint32_t* sync, int M, int D, int H, int G, int I,
                                  const float* cos_tab, const float* sin_tab, const int32_t* slot, const int32_t* row_index,
                                  const int32_t* kstart, const int32_t* lens, int ctx, float eps, float scale, void* stream) {
  using namespace tasu_mega;
  int ks = 0;
  if (!layers || L <= 0 || !x0 || !final_norm || !xn_out || !ws || !sync || !cos_tab || !sin_tab || !slot || !kstart || !lens)
    return TASU_ERR_ARG;
  if (!supported(M, D, H, G, I, ctx, &ks)) return TASU_ERR_ARG;
  if (ws_bytes < (int64_t)ws_layout(D, H, G, I, ks).total * L || ((uintptr_t)ws & 255) || ((uintptr_t)xn_out & 255)) return TASU_ERR_ARG;
  tasu_mega::Args a{};
  a.layers = layers, a.L = L, a.M = M, a.H = H, a.G = G, a.ksplit = ks, a.ctx = ctx;
  a.eps = eps, a.scale = scale;
  a.x0 = x0, a.final_norm = final_norm, a.xn_out = (bf16*)xn_out, a.ws = (unsigned char*)ws;
  a.cos_t = cos_tab, a.sin_t = sin_tab, a.slot = slot, a.row_index = row_index, a.kstart = kstart, a.lens = lens;
  a.sync = sync;
  a.trace = g_trace;
  a.trace_barriers = a.trace ? (int)(g_trace_words / (2 * (int64_t)cu_count())) : 0;
  hipStream_t st = (hipStream_t)stream;
  if (D == 1536) return launch<6, 7, 6>(a, D, I, st);
  if (D == 256) return launch<1, 2, 2>(a, D, I, st);
  return TASU_ERR_ARG;
}
