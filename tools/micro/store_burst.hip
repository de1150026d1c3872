// Micro-benchmark: how long does a CU wait for the write acknowledgements of a 256 x 256 bf16 output tile (128 KiB) when all 256
// CUs store their tiles at the same moment -- the GEMM epilogue's burst (s_waitcnt vmcnt counts stores on gfx9: the K loop's
// counted waits behind an epilogue cannot pass before the stores are acknowledged) -- and does the store pattern matter?
//   pattern 0: one wave instruction = 16 rows x 64 B (half cache lines; the MFMA accumulator layout after pairing fragments)
//   pattern 1: one wave instruction = 8 rows x 128 B (whole lines)
//   pattern 2: one wave instruction = 2 rows x 512 B
//   pattern 0 + nt: the same with the non-temporal hint
//   spread: workgroup w starts (w % 8) * gap microseconds late (a staggered grid: do fewer simultaneous tiles drain faster?)
//   hipcc --offload-arch=gfx950 -O3 -o store_burst store_burst.hip && ./store_burst
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int PAT, bool NT>
__global__ __launch_bounds__(512) void burst(char* c, int ldc_bytes, int tiles_n, int reps, int stagger_clk, long long* cycles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 2, wc = wave & 3;     // the GEMM's wave grid: rows wr*128, cols wc*64 (128 B per row)
  long long spent = 0;
  if (stagger_clk) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (long long)(blockIdx.x % 8) * stagger_clk) __builtin_amdgcn_s_sleep(8);
  }
  for (int r = 0; r < reps; ++r) {
    const int t = blockIdx.x + r * gridDim.x;
    const int tm = t / tiles_n, tn = t % tiles_n;
    char* base = c + (size_t)(tm * 256 + wr * 128) * ldc_bytes + (size_t)tn * 512 + wc * 128;
    __syncthreads();
    const long long t0 = wall_clock64();
    u32x4 v = {(unsigned)r, (unsigned)lane, 3u, 4u};
#pragma unroll
    for (int i = 0; i < 16; ++i) {             // 16 instructions x 1 KiB = the wave's 128 x 64 bf16 block
      char* q;
      if (PAT == 0) q = base + (size_t)((i >> 1) * 16 + (lane & 15)) * ldc_bytes + (i & 1) * 64 + (lane >> 4) * 16;
      else if (PAT == 1) q = base + (size_t)(i * 8 + (lane >> 3)) * ldc_bytes + (lane & 7) * 16;
      else q = base + (size_t)(i * 8 + (lane >> 3)) * ldc_bytes + (lane & 7) * 16;   // (a wave's block is only 128 B wide)
      if (NT) __builtin_nontemporal_store(v, (u32x4*)q);
      else *(u32x4*)q = v;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    spent += wall_clock64() - t0;
    // "compute" between two epilogues: ~34 us like 24 K-tiles, so that consecutive bursts do not overlap
    const long long t1 = wall_clock64();
    while (wall_clock64() - t1 < 3400) __builtin_amdgcn_s_sleep(4);
  }
  if (threadIdx.x == 0) cycles[blockIdx.x] = spent;
}

int main() {
  const int M = 4096, N = 16384, reps = 4;
  char* c;
  long long* cyc;
  CK(hipMalloc(&c, (size_t)M * N * 2));
  CK(hipMalloc(&cyc, 256 * sizeof(long long)));
  std::vector<long long> h(256);
  auto run = [&](const char* name, auto kern, int stagger) -> int {
    for (int it = 0; it < 3; ++it) {
      hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, c, N * 2, N / 256, reps, stagger, cyc);
      CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h.data(), cyc, 256 * sizeof(long long), hipMemcpyDeviceToHost));
    double s = 0, mx = 0;
    for (auto x : h) s += x, mx = x > mx ? x : mx;
    printf("%-44s: stores issued -> all acknowledged, per tile: mean %.2f us, slowest workgroup %.2f us (100 MHz clock64)\n", name,
           s / 256 / reps / 100.0, mx / reps / 100.0);
    return 0;
  };
  if (run("16 rows x 64 B per instruction", burst<0, false>, 0)) return 1;
  if (run("8 rows x 128 B per instruction", burst<1, false>, 0)) return 1;
  if (run("16 rows x 64 B, non-temporal", burst<0, true>, 0)) return 1;
  if (run("8 rows x 128 B, non-temporal", burst<1, true>, 0)) return 1;
  if (run("16 rows x 64 B, starts staggered by 4 us", burst<0, false>, 400)) return 1;
  if (run("8 rows x 128 B, starts staggered by 4 us", burst<1, false>, 400)) return 1;
  return 0;
}
