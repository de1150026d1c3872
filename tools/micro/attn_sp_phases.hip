// Micro-benchmark: where does a workgroup of the single-pass attention kernels (ps_slm_amd/csrc/attention_sp.hip) spend its time?
// The kernels are compiled here with TASU_SP_ABL: Geo carries an ablation mask (timing only -- the results of an ablated run are
// wrong) and a block-id offset that restricts the backward launch to one role.
//   forward bits : 1 skip QK^T, 2 skip the softmax, 4 skip P.V, 8 zero-record DMAs (no memory fetch), 16 skip the output stores
//   backward     : role 0 all / 1 lower key halves (dK, dV) / 2 dQ / 3 upper key halves; bit 1 = skip the step loops, 16 no epilogue
//                  stores; dQ role only: 8 zero-record DMAs, 64 return at once
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=fast -I../../ps_slm_amd/csrc -o attn_sp_phases attn_sp_phases.hip && ./attn_sp_phases
#define TASU_SP_ABL 1
#include "../../ps_slm_amd/csrc/attention_sp.hip"
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  using namespace tasu_sp;
  const int B = argc > 1 ? atoi(argv[1]) : 16, S = 256, H = 12, G = 2, L = 8;     // L rotating buffer sets
  const int M = B * S, LD = (H + 2 * G) * 128, Spad = 256;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<bf16> hq((size_t)M * LD), hd((size_t)M * H * 128);
  for (auto& v : hq) v = (bf16)nd(rng);
  for (auto& v : hd) v = (bf16)nd(rng);
  bf16 *qkv[L], *dout[L], *out[L], *dqkv;
  float *lse[L], *ct, *st, *dkp, *dvp;
  uint8_t* km;
  for (int l = 0; l < L; ++l) {
    CK(hipMalloc(&qkv[l], hq.size() * 2)); CK(hipMemcpy(qkv[l], hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&dout[l], hd.size() * 2)); CK(hipMemcpy(dout[l], hd.data(), hd.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&out[l], hd.size() * 2)); CK(hipMemset(out[l], 0, hd.size() * 2));
    CK(hipMalloc(&lse[l], (size_t)B * H * Spad * 4)); CK(hipMemset(lse[l], 0, (size_t)B * H * Spad * 4));
  }
  CK(hipMalloc(&dqkv, hq.size() * 2));
  CK(hipMalloc(&ct, (size_t)M * 64 * 4)); CK(hipMalloc(&st, (size_t)M * 64 * 4));
  CK(hipMemset(ct, 0, (size_t)M * 64 * 4)); CK(hipMemset(st, 0, (size_t)M * 64 * 4));
  CK(hipMalloc(&dkp, (size_t)M * H * 128 * 4)); CK(hipMalloc(&dvp, (size_t)M * H * 128 * 4));
  CK(hipMalloc(&km, (size_t)B * Spad)); CK(hipMemset(km, 1, (size_t)B * Spad));
  CK(hipFuncSetAttribute((const void*)attn_sp_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FWD_LDS));
  CK(hipFuncSetAttribute((const void*)attn_sp_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const float scale = 0.0883883f;
  auto time_it = [&](const char* name, auto launch) -> int {
    for (int i = 0; i < 3; ++i) launch(i % L);
    CK(hipDeviceSynchronize());
    const int reps = 40;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch(i % L);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.2f us\n", name, ms * 1e3f / reps);
    return 0;
  };
  // real forward first so that out / lse are sane for the backward
  for (int l = 0; l < L; ++l) {
    Geo p{S, Spad, H, G, B, scale, 1, 0, 0, nullptr};
    hipLaunchKernelGGL(attn_sp_fwd_kernel, dim3(B * H), dim3(512), FWD_LDS, 0, qkv[l], km, out[l], lse[l], p);
  }
  CK(hipDeviceSynchronize());
  struct { const char* name; int abl; } fv[] = {{"fwd full", 0}, {"fwd no stores", 16}, {"fwd no PV", 4}, {"fwd no softmax", 2}, {"fwd no QK", 1},
                                               {"fwd no QK/softmax/PV", 7}, {"fwd nothing but DMA + barriers", 23}, {"fwd zero-record DMA, no compute", 31},
                                               {"fwd zero-record DMA, full compute", 8}};
  for (auto& v : fv) {
    if (time_it(v.name, [&](int l) {
          Geo p{S, Spad, H, G, B, scale, 1, v.abl, 0, nullptr};
          hipLaunchKernelGGL(attn_sp_fwd_kernel, dim3(B * H), dim3(512), FWD_LDS, 0, qkv[l], km, out[(l + 1) % L], lse[(l + 1) % L], p);
        })) return 1;
  }
  const int n = B * H;
  struct { const char* name; int abl, id0, grid; } bv[] = {{"bwd all roles", 0, 0, 3 * n}, {"bwd lower key halves only", 0, 0, n}, {"bwd dQ only", 0, n, n},
                                                          {"bwd upper key halves only", 0, 2 * n, n}, {"bwd all roles, no steps", 1, 0, 3 * n},
                                                          {"bwd lower key halves, no steps", 1, 0, n}, {"bwd dQ, no steps", 1, n, n},
                                                          {"bwd upper key halves, no steps", 1, 2 * n, n},
                                                          {"bwd dQ, returns at once", 64, n, n}, {"bwd dQ, no steps, no epilogue stores", 17, n, n},
                                                          {"bwd lower key halves, no steps, no stores", 17, 0, n},
                                                          {"bwd dQ, no steps/stores, zero-record DMA", 25, n, n}};
  for (auto& v : bv) {
    if (time_it(v.name, [&](int l) {
          Geo p{S, Spad, H, G, B, scale, 1, v.abl, v.id0, nullptr};
          hipLaunchKernelGGL(attn_sp_bwd_kernel, dim3(v.grid), dim3(256), BWD_LDS, 0, qkv[l], km, dout[l], out[l], lse[l], ct, st, dqkv, dkp, dvp, p);
        })) return 1;
  }
  // in-kernel stamps of wave 0 (shader clock), one launch of each role alone and one of all roles; median over workgroups
  {
    long long* st_d;
    const int nb = 3 * n;
    CK(hipMalloc(&st_d, (size_t)nb * 32 * 8));
    std::vector<long long> hs((size_t)nb * 32);
    struct { const char* name; int id0, grid, nst; } sv[] = {{"dK/dV lower halves alone", 0, n, 14}, {"dQ alone", n, n, 16}, {"dK/dV upper halves alone", 2 * n, n, 14},
                                                            {"all roles: lower", 0, 3 * n, 14}, {"all roles: dQ", 0, 3 * n, 16}, {"all roles: upper", 0, 3 * n, 14}};
    for (int vi = 0; vi < 6; ++vi) {
      auto& v = sv[vi];
      CK(hipMemset(st_d, 0, (size_t)nb * 32 * 8));
      for (int rep = 0; rep < 3; ++rep) {
        Geo p{S, Spad, H, G, B, scale, 1, 0, v.id0, st_d};
        hipLaunchKernelGGL(attn_sp_bwd_kernel, dim3(v.grid), dim3(256), BWD_LDS, 0, qkv[rep], km, dout[rep], out[rep], lse[rep], ct, st, dqkv, dkp, dvp, p);
      }
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(hs.data(), st_d, (size_t)nb * 32 * 8, hipMemcpyDeviceToHost));
      const int b0 = vi < 3 ? 0 : (vi - 3) * n, b1 = vi < 3 ? v.grid : (vi - 2) * n;
      printf("%-26s stamps (us at 2.1 GHz, median over %d workgroups; last column = first start to last end of the role):\n  ", v.name, b1 - b0);
      for (int i = 1; i < v.nst; ++i) {
        std::vector<double> d;
        for (int bb = b0; bb < b1; ++bb) d.push_back((double)(hs[(size_t)bb * 32 + i] - hs[(size_t)bb * 32 + i - 1]) / 2100.0);
        std::sort(d.begin(), d.end());
        printf("%d:%5.2f ", i, d[d.size() / 2]);
      }
      long long lo = hs[(size_t)b0 * 32], hi = 0;
      for (int bb = b0; bb < b1; ++bb) { lo = std::min(lo, hs[(size_t)bb * 32]); hi = std::max(hi, hs[(size_t)bb * 32 + v.nst - 1]); }
      printf(" | span %.2f\n", (double)(hi - lo) / 2100.0);
    }
  }
  if (time_it("kv reduce + rope", [&](int l) {
        hipLaunchKernelGGL(kv_reduce_rope_kernel, dim3((M + 7) / 8), dim3(256), 0, 0, dqkv, dkp, dvp, ct, st, M, H, G);
      })) return 1;
  return 0;
}
