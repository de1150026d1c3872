// Audio-path helpers around the SenseVoice SANM encoder (Multitask/model/SenseVoice.py) and the PSD
// down-sampler (Multitask/model/ps-slm.py:237-317).  GEMMs and attention come from gemm.hip / attention.hip.
#include "common.h"
#include "../../include/tasu_hip.h"

namespace {

// y = x*scale + PE;  PE(t, d) = sin((t+1) * exp(-d*inc)) for d < D/2, cos(...) for d >= D/2,
// inc = ln(1e4) / (D/2 - 1)          (SenseVoice.py:26-50, :556-558)
__global__ void sinusoid_pe_kernel(const float* __restrict__ x, float* __restrict__ y, int T, int D, float scale,
                                   int64_t total) {
  const int half = D / 2;
  const float inc = logf(10000.f) / (float)(half - 1);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const int t = (int)((i / D) % T);
    const int k = d < half ? d : d - half;
    const float ang = (float)(t + 1) * expf(-(float)k * inc);
    y[i] = x[i] * scale + (d < half ? sinf(ang) : cosf(ang));
  }
}

// FSMN memory (SenseVoice.py:124-140): out[b,t,d] (+)= mask_t * (sum_j w[d][j] * vm[t+j-left] + vm[t]), vm = v*mask
__global__ void fsmn_kernel(const bf16* __restrict__ v, int ldv, const float* __restrict__ w, const int32_t* __restrict__ lens,
                            float* __restrict__ out, int T, int D, int ksize, int accumulate, int64_t total) {
  const int left = (ksize - 1) / 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const int64_t bt = i / D;
    const int t = (int)(bt % T);
    const int b = (int)(bt / T);
    const int len = lens[b];
    float r = 0.f;
    if (t < len) {
      for (int j = 0; j < ksize; ++j) {
        const int tt = t + j - left;
        if (tt >= 0 && tt < len) r += w[d * ksize + j] * (float)v[((size_t)b * T + tt) * ldv + d];
      }
      r += (float)v[((size_t)b * T + t) * ldv + d];
    }
    out[i] = accumulate ? out[i] + r : r;
  }
}

// Vector form for D % 8 == 0: a thread owns 8 consecutive channels (16-byte loads of v, its 8 x KS filter taps in registers:
// they are contiguous in w) and FR = 4 consecutive frames, whose KS + FR - 1 input rows it loads once -- 3.5 loads per output
// frame instead of 12, no division per element (the first form: one frame per thread, 64-bit div / mod per unit, taps
// transposed through LDS by every block: 29.6 us for the encoder's 8064 x 512 rows, i.e. 1.4 TB/s; this one runs at the rate
// of its 33 MB of fp32 read-modify-write).  Block = (16 frames of one utterance) x (D / 8 channel chunks, 64 per wave row).
template <int KS>
__global__ __launch_bounds__(256) void fsmn_rows_kernel(const bf16* __restrict__ v, int ldv, const float* __restrict__ w,
                                                        const int32_t* __restrict__ lens, float* __restrict__ out, int T, int D,
                                                        int accumulate) {
  constexpr int FR = 4, LEFT = (KS - 1) / 2, NR = KS + FR - 1;
  const int b = blockIdx.y, len = lens[b];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t0 = blockIdx.x * (4 * FR) + wave * FR;             // this thread's first frame
  if (t0 >= T) return;
  for (int c = lane * 8; c < D; c += 512) {
    float wt[8][KS];                                             // taps of channels c .. c + 7: 8 * KS contiguous floats
    {
      const float* wp = w + (size_t)c * KS;
      float flat[8 * KS];
#pragma unroll
      for (int i = 0; i < 8 * KS / 4; ++i) *(f32x4*)(flat + 4 * i) = *(const f32x4*)(wp + 4 * i);
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int j = 0; j < KS; ++j) wt[q][j] = flat[q * KS + j];
    }
    const bf16* base = v + ((size_t)b * T) * ldv + c;
    bf16x8 x[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int tt = t0 - LEFT + i;
      x[i] = (tt >= 0 && tt < len) ? *(const bf16x8*)(base + (size_t)tt * ldv) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
#pragma unroll
    for (int f = 0; f < FR; ++f) {
      const int t = t0 + f;
      if (t >= T) break;
      float r[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) r[q] = 0.f;
      if (t < len) {
#pragma unroll
        for (int j = 0; j < KS; ++j) {
          const int tt = t + j - LEFT;
          if (tt < 0 || tt >= len) continue;                     // (the same terms, in the same order, as the scalar form)
#pragma unroll
          for (int q = 0; q < 8; ++q) r[q] += wt[q][j] * (float)x[f + j][q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] += (float)x[f + LEFT][q];
      }
      float* o = out + ((size_t)b * T + t) * D + c;
      f32x4 o0 = f32x4{r[0], r[1], r[2], r[3]}, o1 = f32x4{r[4], r[5], r[6], r[7]};
      if (accumulate) {
        o0 += *(const f32x4*)o;
        o1 += *(const f32x4*)(o + 4);
      }
      *(f32x4*)o = o0;
      *(f32x4*)(o + 4) = o1;
    }
  }
}

// FSMN memory block + the LayerNorm that follows it (SANM layer: x += fsmn(v); xn = norm2(x)) for D = 512, one wave per 4 frames of
// one utterance: the wave owns whole rows, so the norm's statistics are wave reductions over values still in registers and the
// separate LayerNorm launch (one more read of the 16.5-MB fp32 stream per layer) disappears.  A lane owns channels lane*4..+3 and
// 256+lane*4..+3 -- the element-to-lane map of layernorm_fwd_wave_kernel (norm.hip) -- and the sums run in that kernel's order:
// bit-identical to tasu_fsmn_fwd followed by tasu_layernorm_fwd.
template <int KS>
__global__ __launch_bounds__(256) void fsmn_ln_rows_kernel(const bf16* __restrict__ v, int ldv, const float* __restrict__ w,
                                                           const int32_t* __restrict__ lens, float* __restrict__ out,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           bf16* __restrict__ y, int ldy, int T, float eps) {
  constexpr int D = 512, FR = 4, LEFT = (KS - 1) / 2, NR = KS + FR - 1;
  const int b = blockIdx.y, len = lens[b];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t0 = blockIdx.x * (4 * FR) + wave * FR;
  if (t0 >= T) return;
  f32x4 res[FR][2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int c = lane * 4 + g * 256;
    float wt[4][KS];
    {
      const float* wp = w + (size_t)c * KS;
      float flat[4 * KS];
#pragma unroll
      for (int i = 0; i < 4 * KS / 4; ++i) *(f32x4*)(flat + 4 * i) = *(const f32x4*)(wp + 4 * i);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < KS; ++j) wt[q][j] = flat[q * KS + j];
    }
    const bf16* base = v + ((size_t)b * T) * ldv + c;
    bf16x4 x[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int tt = t0 - LEFT + i;
      x[i] = (tt >= 0 && tt < len) ? *(const bf16x4*)(base + (size_t)tt * ldv) : bf16x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int f = 0; f < FR; ++f) {
      const int t = t0 + f;
      float r[4] = {0.f, 0.f, 0.f, 0.f};
      if (t < T && t < len) {
#pragma unroll
        for (int j = 0; j < KS; ++j) {
          const int tt = t + j - LEFT;
          if (tt < 0 || tt >= len) continue;                     // (the same terms, in the same order, as fsmn_rows_kernel)
#pragma unroll
          for (int q = 0; q < 4; ++q) r[q] += wt[q][j] * (float)x[f + j][q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] += (float)x[f + LEFT][q];
      }
      f32x4 o = f32x4{r[0], r[1], r[2], r[3]};
      if (t < T) {
        float* op = out + ((size_t)b * T + t) * D + c;
        o += *(const f32x4*)op;                                  // x += fsmn(v)  (accumulate form only)
        *(f32x4*)op = o;
      }
      res[f][g] = o;
    }
  }
  const f32x4 ga0 = *(const f32x4*)(gamma + lane * 4), ga1 = *(const f32x4*)(gamma + 256 + lane * 4);
  const f32x4 be0 = *(const f32x4*)(beta + lane * 4), be1 = *(const f32x4*)(beta + 256 + lane * 4);
#pragma unroll
  for (int f = 0; f < FR; ++f) {
    const int t = t0 + f;
    if (t >= T) break;                                            // (wave-uniform)
    float sm = 0.f;
#pragma unroll
    for (int g = 0; g < 2; ++g) sm += res[f][g][0] + res[f][g][1] + res[f][g][2] + res[f][g][3];
    const float mu = wave_sum(sm) / (float)D;
    float q2 = 0.f;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) q2 += (res[f][g][j] - mu) * (res[f][g][j] - mu);
    const float rs = rsqrtf(wave_sum(q2) / (float)D + eps);
    bf16* yr = y + ((size_t)b * T + t) * ldy;
    f32x4 o0, o1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o0[j] = (res[f][0][j] - mu) * rs * ga0[j] + be0[j];
      o1[j] = (res[f][1][j] - mu) * rs * ga1[j] + be1[j];
    }
    *(bf16x4*)(yr + lane * 4) = __builtin_convertvector(o0, bf16x4);
    *(bf16x4*)(yr + 256 + lane * 4) = __builtin_convertvector(o1, bf16x4);
    for (int c = D + lane; c < ldy; c += 64) yr[c] = (bf16)0.f;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const T* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                           int V) {
  __shared__ float red[4];
  const T* xr = x + (size_t)blockIdx.x * ldx;
  float* yr = y + (size_t)blockIdx.x * ldy;
  float m = -__builtin_inff();
  for (int c = threadIdx.x; c < V; c += 256) m = fmaxf(m, (float)xr[c]);
  m = block_max<4>(m, red);
  float s = 0.f;
  for (int c = threadIdx.x; c < V; c += 256) s += expf((float)xr[c] - m);
  s = block_sum<4>(s, red);
  const float inv = 1.f / s;
  for (int c = threadIdx.x; c < ldy; c += 256) yr[c] = c < V ? expf((float)xr[c] - m) * inv : 0.f;
}

// per frame: argmax id (first index on ties) and blank probability.  grid B*T blocks.
__global__ __launch_bounds__(256) void psd_frame_stats_kernel(const float* __restrict__ post, int ldp,
                                                              const int32_t* __restrict__ lens, int32_t* __restrict__ fid,
                                                              float* __restrict__ fblank, int T, int bstride, int V,
                                                              int blank_id) {
  __shared__ float rv[4];
  __shared__ int ri[4];
  const int bt = blockIdx.x;
  const int b = bt / T, t = bt - b * T;
  if (t >= lens[b]) {
    if (threadIdx.x == 0) {
      fid[bt] = -1;
      fblank[bt] = 0.f;
    }
    return;
  }
  const float* pr = post + ((size_t)b * bstride + t) * ldp;
  float best = -__builtin_inff();
  int arg = 0x7fffffff;
  for (int c = threadIdx.x; c < V; c += 256) {
    const float f = pr[c];
    if (f > best) {
      best = f;
      arg = c;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oa = __shfl_xor(arg, o, 64);
    if (ob > best || (ob == best && oa < arg)) {
      best = ob;
      arg = oa;
    }
  }
  if ((threadIdx.x & 63) == 0) {
    rv[threadIdx.x >> 6] = best;
    ri[threadIdx.x >> 6] = arg;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; ++i)
      if (rv[i] > best || (rv[i] == best && ri[i] < arg)) {
        best = rv[i];
        arg = ri[i];
      }
    fid[bt] = arg;
    fblank[bt] = pr[blank_id];
  }
}

// PSD straight from the CTC head's bf16 LOGITS (round 4): the fp32 posterior of every frame ([B x 504, 25055] = 808 MB written by
// the softmax, read again by the frame statistics) is never materialised -- PSD keeps ~100 of 500 frames of a trained encoder.
// Per frame: argmax (of the logits = of the posterior; first index on ties), the softmax statistics (max, 1 / sum) and the blank
// probability; the gather below evaluates the softmax only for the frames PSD keeps.  grid B*T blocks.
__global__ __launch_bounds__(256) void psd_logit_stats_kernel(const bf16* __restrict__ logits, int ld, const int32_t* __restrict__ lens,
                                                              int32_t* __restrict__ fid, float* __restrict__ fblank,
                                                              float* __restrict__ fstat, int T, int bstride, int V, int blank_id) {
  __shared__ float rv[4];
  __shared__ int ri[4];
  __shared__ float red[4];
  const int bt = blockIdx.x;
  const int b = bt / T, t = bt - b * T;
  if (t >= lens[b]) {
    if (threadIdx.x == 0) {
      fid[bt] = -1;
      fblank[bt] = 0.f;
      fstat[2 * bt] = 0.f, fstat[2 * bt + 1] = 0.f;
    }
    return;
  }
  const bf16* xr = logits + ((size_t)b * bstride + t) * ld;
  const int V8 = V & ~7;
  // ONE pass over the row (401 MB of logits per 16 x 500 frames; a second pass went to the fabric again, profiles/
  // r04_new_kernels_pmc.json): per thread a running (max, sum of exp) pair rescaled when the max moves, and the argmax
  float best = -__builtin_inff(), sum = 0.f;
  int arg = 0x7fffffff;
  for (int c = threadIdx.x * 8; c < V8; c += 2048) {
    const bf16x8 v = *(const bf16x8*)(xr + c);
    float f[8], cm = -__builtin_inff();
    int ca = 0;
#pragma unroll
    for (int j2 = 0; j2 < 8; ++j2) {
      f[j2] = (float)v[j2];
      if (f[j2] > cm) cm = f[j2], ca = c + j2;
    }
    if (cm > best) {
      sum *= expf(best - cm);                          // (best = -inf: sum is 0)
      best = cm, arg = ca;
    }
#pragma unroll
    for (int j2 = 0; j2 < 8; ++j2) sum += expf(f[j2] - best);
  }
  for (int c = V8 + threadIdx.x; c < V; c += 256) {
    const float f = (float)xr[c];
    if (f > best) {
      sum *= expf(best - f);
      best = f, arg = c;
    }
    sum += expf(f - best);
  }
  // block reduction of (max, argmax [first index on ties], sum at that max)
  float tmax = best;
  int targ = arg;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(tmax, o, 64);
    const int oa = __shfl_xor(targ, o, 64);
    if (ob > tmax || (ob == tmax && oa < targ)) tmax = ob, targ = oa;
  }
  if ((threadIdx.x & 63) == 0) rv[threadIdx.x >> 6] = tmax, ri[threadIdx.x >> 6] = targ;
  __syncthreads();
  float m = rv[0];
  int am = ri[0];
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (rv[i] > m || (rv[i] == m && ri[i] < am)) m = rv[i], am = ri[i];
  sum = best == -__builtin_inff() ? 0.f : sum * expf(best - m);        // this thread's sum at the row's max
  sum = block_sum<4>(sum, red);
  if (threadIdx.x == 0) {
    const float inv = 1.f / sum;
    fid[bt] = am;
    fstat[2 * bt] = m, fstat[2 * bt + 1] = inv;
    fblank[bt] = expf((float)xr[blank_id] - m) * inv;
  }
}

// out[b, j, :] = mean over the frames of kept segment j of softmax(logits[frame]) (j < new_lens[b]) else 0.   grid (Tout, B)
__global__ __launch_bounds__(256) void psd_gather_softmax_kernel(const bf16* __restrict__ logits, int ld, const float* __restrict__ fstat,
                                                                 const int32_t* __restrict__ seg_start, const int32_t* __restrict__ seg_len,
                                                                 const int32_t* __restrict__ new_lens, float* __restrict__ out, int ldo,
                                                                 int T, int bstride, int Tout, int V) {
  const int j = blockIdx.x, b = blockIdx.y;
  float* o = out + ((size_t)b * Tout + j) * ldo;
  if (j >= new_lens[b]) {
    for (int c = threadIdx.x * 4; c < ldo; c += 1024) *(f32x4*)(o + c) = f32x4{0.f, 0.f, 0.f, 0.f};
    return;
  }
  const int s0 = seg_start[(size_t)b * T + j], len = seg_len[(size_t)b * T + j];
  const bf16* x0 = logits + ((size_t)b * bstride + s0) * ld;
  const float* st = fstat + 2 * ((size_t)b * T + s0);
  for (int c = threadIdx.x * 8; c < ldo; c += 2048) {
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int t = 0; t < len; ++t) {
      const bf16x8 v = *(const bf16x8*)(x0 + (size_t)t * ld + c);
      const float m = st[2 * t], inv = st[2 * t + 1];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += expf((float)v[e] - m) * inv;
    }
    f32x4 o0, o1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o0[e] = c + e < V ? (len > 1 ? acc[e] / (float)len : acc[e]) : 0.f;
      o1[e] = c + 4 + e < V ? (len > 1 ? acc[4 + e] / (float)len : acc[4 + e]) : 0.f;
    }
    *(f32x4*)(o + c) = o0;
    *(f32x4*)(o + c + 4) = o1;
  }
}

// one 256-thread block per utterance: run-length segments of equal non-blank ids (blank frames stay single), blank filter.
// A frame STARTS a segment iff it is the first, differs from its predecessor, or is blank (blank_id < 0: every frame); the thread
// of a start frame walks its run (same summation order as a sequential scan), and the kept segments are numbered by a block-wide
// prefix count per chunk of 256 frames.  (Was one THREAD per utterance: 213 us of dependent global loads for 500 frames.)
__global__ __launch_bounds__(256) void psd_plan_kernel(const int32_t* __restrict__ fid, const float* __restrict__ fblank,
                                                       const int32_t* __restrict__ lens, int32_t* __restrict__ seg_start,
                                                       int32_t* __restrict__ seg_len, int32_t* __restrict__ new_lens, int B, int T,
                                                       int blank_id, float thr) {
  __shared__ int wsum[4];
  __shared__ int base_s;
  const int b = blockIdx.x;
  const int L = lens[b];
  const int32_t* id = fid + (size_t)b * T;
  const float* bp = fblank + (size_t)b * T;
  if (threadIdx.x == 0) base_s = 0;
  __syncthreads();
  for (int c0 = 0; c0 < L; c0 += 256) {
    const int t = c0 + threadIdx.x;
    int keep = 0, len = 0;
    if (t < L) {
      const int me = id[t];
      const bool start = t == 0 || blank_id < 0 || me == blank_id || me != id[t - 1];
      if (start) {
        int end = t + 1;
        if (!(blank_id < 0 || me == blank_id))
          while (end < L && id[end] == me) ++end;
        len = end - t;
        float sm = 0.f;
        for (int u = t; u < end; ++u) sm += bp[u];
        const float mean = len == 1 ? sm : sm / (float)len;
        keep = mean < thr ? 1 : 0;
      }
    }
    const unsigned long long mask = __ballot(keep);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int before = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[w] = __popcll(mask);
    __syncthreads();
    int off = base_s;
    for (int i = 0; i < w; ++i) off += wsum[i];
    if (keep) {
      seg_start[(size_t)b * T + off + before] = t;
      seg_len[(size_t)b * T + off + before] = len;
    }
    __syncthreads();
    if (threadIdx.x == 0) base_s += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) new_lens[b] = base_s;
}

// out[b, j, :] = mean of post rows seg_start..+len (j < new_lens[b]) else 0.   grid (Tout, B)
__global__ __launch_bounds__(256) void psd_gather_kernel(const float* __restrict__ post, int ldp,
                                                         const int32_t* __restrict__ seg_start,
                                                         const int32_t* __restrict__ seg_len,
                                                         const int32_t* __restrict__ new_lens, float* __restrict__ out, int ldo,
                                                         int T, int bstride, int Tout, int V) {
  const int j = blockIdx.x, b = blockIdx.y;
  float* o = out + ((size_t)b * Tout + j) * ldo;
  if (j >= new_lens[b]) {
    for (int c = threadIdx.x; c < ldo; c += 256) o[c] = 0.f;
    return;
  }
  const int s0 = seg_start[(size_t)b * T + j], len = seg_len[(size_t)b * T + j];
  const float* p0 = post + ((size_t)b * bstride + s0) * ldp;
  for (int c = threadIdx.x; c < ldo; c += 256) {
    float s = 0.f;
    if (c < V) {
      for (int t = 0; t < len; ++t) s += p0[(size_t)t * ldp + c];
      if (len > 1) s /= (float)len;
    }
    o[c] = s;
  }
}

inline int grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}
}  // namespace

extern "C" int tasu_sinusoid_pe(const float* x, float* y, int B, int T, int D, float scale, void* stream) {
  if (!x || !y || B <= 0 || T <= 0 || D < 4 || D % 2) return TASU_ERR_ARG;
  const int64_t total = (int64_t)B * T * D;
  TASU_LAUNCH(sinusoid_pe_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, T, D, scale, total);
  return TASU_OK;
}
extern "C" int tasu_fsmn_fwd(const void* v, int ldv, const float* w, const int32_t* lens, float* out, int B, int T, int D,
                             int ksize, int accumulate, void* stream) {
  if (!v || !w || !lens || !out || B <= 0 || T <= 0 || D <= 0 || ksize <= 0) return TASU_ERR_ARG;
  const int64_t total = (int64_t)B * T * D;
  if (D % 8 == 0 && ldv % 8 == 0 && ksize == 11 && !(((uintptr_t)v | (uintptr_t)out | (uintptr_t)w) & 15)) {      // SenseVoiceSmall's kernel_size
    TASU_LAUNCH(fsmn_rows_kernel<11>, dim3((T + 15) / 16, B), dim3(256), 0, (hipStream_t)stream, (const bf16*)v, ldv, w, lens, out,
                T, D, accumulate);
    return TASU_OK;
  }
  TASU_LAUNCH(fsmn_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16*)v, ldv, w, lens,
                     out, T, D, ksize, accumulate, total);
  return TASU_OK;
}
extern "C" int tasu_softmax_rows(const void* x, int x_is_bf16, int ldx, float* y, int ldy, int R, int V, void* stream) {
  if (!x || !y || R <= 0 || V <= 0 || ldx < V || ldy < V) return TASU_ERR_ARG;
  if (x_is_bf16)
    TASU_LAUNCH(softmax_rows_kernel<bf16>, dim3(R), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx, y, ldy, V);
  else
    TASU_LAUNCH(softmax_rows_kernel<float>, dim3(R), dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, y, ldy, V);
  return TASU_OK;
}
extern "C" int tasu_psd_frame_stats(const float* post, int ldp, const int32_t* lens, int32_t* frame_id, float* frame_blank,
                                    int B, int T, int bstride, int V, int blank_id, void* stream) {
  if (!post || !lens || !frame_id || !frame_blank || B <= 0 || T <= 0 || bstride < T || V <= 0 || blank_id < 0 ||
      blank_id >= V)
    return TASU_ERR_ARG;
  TASU_LAUNCH(psd_frame_stats_kernel, dim3(B * T), dim3(256), 0, (hipStream_t)stream, post, ldp, lens, frame_id,
                     frame_blank, T, bstride, V, blank_id);
  return TASU_OK;
}
extern "C" int tasu_psd_plan(const int32_t* frame_id, const float* frame_blank, const int32_t* lens, int32_t* seg_start,
                             int32_t* seg_len, int32_t* new_lens, int B, int T, int blank_id, float threshold,
                             void* stream) {
  if (!frame_id || !frame_blank || !lens || !seg_start || !seg_len || !new_lens || B <= 0 || T <= 0) return TASU_ERR_ARG;
  TASU_LAUNCH(psd_plan_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, frame_id, frame_blank, lens,
                     seg_start, seg_len, new_lens, B, T, blank_id, threshold);
  return TASU_OK;
}
extern "C" int tasu_psd_gather(const float* post, int ldp, const int32_t* seg_start, const int32_t* seg_len,
                               const int32_t* new_lens, float* out, int ldo, int B, int T, int bstride, int Tout, int V,
                               void* stream) {
  if (!post || !seg_start || !seg_len || !new_lens || !out || B <= 0 || T <= 0 || bstride < T || Tout <= 0 || V <= 0 || ldo < V)
    return TASU_ERR_ARG;
  TASU_LAUNCH(psd_gather_kernel, dim3(Tout, B), dim3(256), 0, (hipStream_t)stream, post, ldp, seg_start, seg_len,
                     new_lens, out, ldo, T, bstride, Tout, V);
  return TASU_OK;
}

extern "C" int tasu_psd_logit_stats(const void* logits, int ld, const int32_t* lens, int32_t* frame_id, float* frame_blank,
                                    float* frame_stat, int B, int T, int bstride, int V, int blank_id, void* stream) {
  if (!logits || !lens || !frame_id || !frame_blank || !frame_stat || B <= 0 || T <= 0 || bstride < T || V <= 0 || blank_id < 0 ||
      blank_id >= V || ld < V || ld % 8 || ((uintptr_t)logits & 15))
    return TASU_ERR_ARG;
  TASU_LAUNCH(psd_logit_stats_kernel, dim3(B * T), dim3(256), 0, (hipStream_t)stream, (const bf16*)logits, ld, lens, frame_id,
              frame_blank, frame_stat, T, bstride, V, blank_id);
  return TASU_OK;
}
extern "C" int tasu_psd_gather_softmax(const void* logits, int ld, const float* frame_stat, const int32_t* seg_start,
                                       const int32_t* seg_len, const int32_t* new_lens, float* out, int ldo, int B, int T, int bstride,
                                       int Tout, int V, void* stream) {
  if (!logits || !frame_stat || !seg_start || !seg_len || !new_lens || !out || B <= 0 || T <= 0 || bstride < T || Tout <= 0 || V <= 0 ||
      ldo < V || ldo % 8 || ld < ldo || ld % 8 || (((uintptr_t)logits | (uintptr_t)out) & 15))
    return TASU_ERR_ARG;
  TASU_LAUNCH(psd_gather_softmax_kernel, dim3(Tout, B), dim3(256), 0, (hipStream_t)stream, (const bf16*)logits, ld, frame_stat, seg_start,
              seg_len, new_lens, out, ldo, T, bstride, Tout, V);
  return TASU_OK;
}

extern "C" int tasu_layernorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, void* y, int ldy, int y_is_f32,
                                  float* mean, float* rstd, int R, int D, float eps, void* stream);
// x += fsmn(v) and xn = LayerNorm(x) (bf16) -- one launch for the SANM layer's D = 512 / kernel 11; other sizes: the two kernels
extern "C" int tasu_fsmn_ln_fwd(const void* v, int ldv, const float* w, const int32_t* lens, float* x, const float* gamma,
                                const float* beta, void* xn, int ldy, int B, int T, int D, int ksize, float eps, void* stream) {
  if (!v || !w || !lens || !x || !gamma || !beta || !xn || B <= 0 || T <= 0 || D <= 0 || ksize <= 0 || ldy < D) return TASU_ERR_ARG;
  const bool fused = D == 512 && ksize == 11 && ldv % 4 == 0 && ldy % 4 == 0 &&
                     !(((uintptr_t)v | (uintptr_t)xn) & 7) && !(((uintptr_t)x | (uintptr_t)w | (uintptr_t)gamma | (uintptr_t)beta) & 15);
  if (fused) {
    TASU_LAUNCH(fsmn_ln_rows_kernel<11>, dim3((T + 15) / 16, B), dim3(256), 0, (hipStream_t)stream, (const bf16*)v, ldv, w, lens, x, gamma,
                beta, (bf16*)xn, ldy, T, eps);
    return TASU_OK;
  }
  const int rc = tasu_fsmn_fwd(v, ldv, w, lens, x, B, T, D, ksize, 1, stream);
  return rc ? rc : tasu_layernorm_fwd(x, D, gamma, beta, xn, ldy, 0, nullptr, nullptr, B * T, D, eps, stream);
}
