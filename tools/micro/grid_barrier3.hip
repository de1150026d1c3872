// Micro-benchmark 3: two-level counter barrier.  Workgroup b arrives on the counter of its group (b % groups: round-robin placement
// puts consecutive workgroups on consecutive XCDs, so a group = the workgroups of one XCD when groups == 8); the last arriver of a
// group arrives on the top counter; everybody polls one "epoch" word that the last top arriver publishes.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(512) void barrier_kernel(unsigned* st, int iters, int groups, int sleep, unsigned* timeout) {
  // st[0] = epoch word, st[32] = top counter, st[64 + 32 g] = group counters (separate 128-byte lines)
  const int nb = gridDim.x, b = blockIdx.x;
  const int g = b % groups, gsize = (nb - g + groups - 1) / groups;
  unsigned epoch = 0;
  for (int i = 0; i < iters; ++i) {
    __syncthreads();
    ++epoch;
    if (threadIdx.x == 0) {
      bool publish = false;
      if (groups == 1) {
        publish = __hip_atomic_fetch_add(&st[32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch * nb - 1;
      } else if (__hip_atomic_fetch_add(&st[64 + 32 * g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch * gsize - 1) {
        publish = __hip_atomic_fetch_add(&st[32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch * groups - 1;
      }
      if (publish) __hip_atomic_store(&st[0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      long long t0 = clock64();
      while (__hip_atomic_load(&st[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
        if (clock64() - t0 > 400000000ll) { atomicAdd(timeout, 1u); break; }
        if (sleep) __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();
  }
}

int run(int grid, int groups, int sleep, int iters) {
  unsigned *st, *to;
  CK(hipMalloc(&st, 4096 * 4)); CK(hipMalloc(&to, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f; unsigned hto = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(st, 0, 4096 * 4)); CK(hipMemset(to, 0, 4));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(barrier_kernel, dim3(grid), dim3(512), 0, 0, st, iters, groups, sleep, to);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
    CK(hipMemcpy(&hto, to, 4, hipMemcpyDeviceToHost));
  }
  printf("grid %3d groups %2d sleep %d: %.2f us per barrier, timeouts %u\n", grid, groups, sleep, best * 1e3 / iters, hto);
  return 0;
}

int main() {
  for (int grid : {8, 32, 128, 256})
    for (int groups : {1, 8, 16, 32})
      for (int sleep : {0, 1}) if (groups <= grid) run(grid, groups, sleep, 4000);
  return 0;
}
