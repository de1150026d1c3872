// LDS tile images and MFMA operand reads shared by the attention kernels (attention.hip: one head per workgroup;
// attention_gqa.hip: the query heads of a GQA group share the staged K / V tiles).  gfx950 only.
#pragma once
#include "common.h"

namespace tasu_attn {

constexpr int HD = 128;

// ---- LDS tile images ---------------------------------------------------------------------------------
// "row" image : [64 tokens][128 d] bf16, 256-B rows, 16-B chunk c of row r stored at chunk position c ^ swz(r).
// swz (round 4): rows 0..7 -> 0, 2, .., 14, rows 8..15 -> 9, 11, 13, 15, 1, 3, 5, 7 (linear over GF(2): bit i of the row
// contributes 2, 4, 8, 9).  Rounds 1-3 used swz(r) = r & 15, which serves the ROW reads (ds_read_b128: 16 rows x one chunk per
// lane group) without bank conflicts but not the TRANSPOSE reads: the 32 lanes a ds_read_b64_tr_b16 is served in touch 8
// consecutive rows x the two chunks of one 16-d block, and r -> r ^ 1 maps chunk 2n of row r onto chunk 2n + 1 of row r ^ 1 --
// every bank pair is hit twice.  With swz(r) >> 1 distinct over 8 consecutive rows both read forms are conflict-free
// (tools/lab/lds_swizzle_search.py: the lane groups of MI355X_MICROARCH.md's LDS table).  rocprofv3 --pmc on the training shape
// (tools/lab/attn_lds_pmc.sh): SQ_LDS_BANK_CONFLICT 491,520 -> 0 (forward), 1,474,560 -> 0 (backward), SQ_LDS_IDX_ACTIVE
// -24 %; launch times unchanged -- these kernels are not bound by the LDS port (attention_gqa.hip, MEASURED).
constexpr int ROW_TILE_BYTES = 64 * 256;
__device__ __forceinline__ int swz(int r) { return ((r & 7) << 1) ^ ((r & 8) ? 9 : 0); }

// Tile staging is split (issue-early / write-late): fetch_* issues the 4 global loads of a tile into registers, the
// MFMA work of the previous tile runs while they are in flight, and commit_* writes them to LDS after the barrier.
struct TileRegs {
  bf16x8 v[4];
};
// [64][128] tile whose rows are tokens tok0.. of a token-major matrix (row stride ld elements); rows >= nrows are
// clamped (callers mask them).
__device__ __forceinline__ void fetch_row_tile(TileRegs& t, const bf16* g, int ld, int tok0, int nrows) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = i * 256 + threadIdx.x;
    const int r = idx >> 4, c = idx & 15;
    const int tok = min(tok0 + r, nrows - 1);
    t.v[i] = *(const bf16x8*)(g + (size_t)tok * ld + c * 8);
  }
}
__device__ __forceinline__ void commit_row_tile(char* lds, const TileRegs& t) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = i * 256 + threadIdx.x;
    const int r = idx >> 4, c = idx & 15;
    *(bf16x8*)(lds + r * 256 + ((c ^ swz(r)) << 4)) = t.v[i];
  }
}
// MFMA operand (16 rows = tile rows sub*16 + (lane&15), k = d in [32ks + 8q', +8)) from a "row" image.
__device__ __forceinline__ bf16x8 frag_row(const char* lds, int sub, int ks, int lane) {
  const int r = sub * 16 + (lane & 15);
  const int c = ks * 4 + (lane >> 4);
  return *(const bf16x8*)(lds + r * 256 + ((c ^ swz(lane & 15)) << 4));
}
// MFMA operand (16 rows = d in nt*16 + (lane&15), k-slots of token block tb (32 tokens): element j <-> token tb*32 + (j<4 ? 4q'+j :
// 16+4q'+j-4), q' = lane>>4) read out of a token-major "row" image with two hardware transpose reads.  A 16-lane group g reads
// the 4-token x 16-d block (tokens T0 + 4g .. +3, d = nt*16 .. +15): lane 4q+p of the group supplies the address of token
// T0 + 4g + q, columns nt*16 + 4p .. +3, and lane i receives column nt*16 + i of the four tokens.  EXEC must be all ones.
typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ bf16x8 frag_tr_row(const char* lds, int nt, int tb, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int r0 = tb * 32 + 4 * g + q, r1 = r0 + 16;                 // this lane's address rows for the two reads
  const int ch = nt * 2 + (p >> 1), inner = (p & 1) * 8;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + r0 * 256 + ((ch ^ swz(r0)) << 4) + inner));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + r1 * 256 + ((ch ^ swz(r1)) << 4) + inner));
  union { s16x4 s[2]; bf16x8 b; } u;
  u.s[0] = lo;
  u.s[1] = hi;
  return u.b;
}
// pack two 16-wide score tiles (fp32 accumulators) into the k-slot order frag_tr / frag_tr_row use.
__device__ __forceinline__ bf16x8 pack_pair(f32x4 a, f32x4 b) {
  bf16x8 o;
  o[0] = (bf16)a[0]; o[1] = (bf16)a[1]; o[2] = (bf16)a[2]; o[3] = (bf16)a[3];
  o[4] = (bf16)b[0]; o[5] = (bf16)b[1]; o[6] = (bf16)b[2]; o[7] = (bf16)b[3];
  return o;
}
// operand straight from global: row `tok` of a token-major matrix, d in [32ks + 8q', +8), ks = 0..3
__device__ __forceinline__ void load_row_frags(bf16x8 f[4], const bf16* g, int ld, int tok, int lane) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) f[ks] = *(const bf16x8*)(g + (size_t)tok * ld + ks * 32 + (lane >> 4) * 8);
}

constexpr float NEG_INF = -__builtin_inff();

// Softmax arithmetic shared by the tiled kernels (attention.hip), the GQA backward (attention_gqa.hip) and the single-pass kernels
// (attention_sp.hip), round 5: scores live in the base-2 domain, t = s * (scale * log2 e) + bias, with the key-padding mask as an
// ADDITIVE bias (0 = attend, -inf = masked: exp2(-inf) = 0 exactly) -- one fused multiply-add per element where rounds 1-4 spent a
// byte extract, two compares, a select, a multiply and expf's own multiply (the tiled kernels' softmax was ~1200 vector-ALU cycles
// per wave and key tile against 512 matrix-pipe cycles).  The causal compare runs only on tiles that straddle the diagonal.
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float mask_bias(unsigned byte) { return byte ? 0.f : NEG_INF; }
// probability of one score in the backward kernels: exp2(s * scale2 + (bias - lse * log2 e))
__device__ __forceinline__ float prob2(float s, float scale2, float bias_minus_lse2) { return exp2_fast(__builtin_fmaf(s, scale2, bias_minus_lse2)); }

// rotate-half RoPE backward of one pair (forward: y1 = x1 c - x2 s, y2 = x2 c + x1 s  =>  dx1 = dy1 c + dy2 s, dx2 = dy2 c - dy1 s).
// Explicit FMAs (as rope_pair_f): tasu_rope_bwd and the epilogues of the GQA backward kernel share this text -- the same bits.
__device__ __forceinline__ void rope_pair_bwd_f(float dy1, float dy2, float c, float s, float& dx1, float& dx2) {
  const float t1 = dy2 * s, t2 = dy1 * s;
  dx1 = __builtin_fmaf(dy1, c, t1);
  dx2 = __builtin_fmaf(dy2, c, -t2);
}

}  // namespace tasu_attn
