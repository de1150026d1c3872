// Micro-benchmark 2: flag-based grid barrier (every workgroup publishes its epoch in its own word; every workgroup polls all words
// with one load per thread) and the price of the fences that make plain stores visible across XCDs.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier2 grid_barrier2.hip && ./grid_barrier2
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// MODE 0: coherent (agent-scope) data stores and loads, no fences       -> raw barrier latency
// MODE 1: plain stores; release fence (all threads) before the flag; acquire fence after; plain loads
// MODE 2: write-through data stores (agent-scope atomic store); acquire fence only; plain loads
// MODE 3: plain stores; release fence only; coherent loads
template <int MODE>
__global__ __launch_bounds__(512) void barrier_kernel(unsigned* flags, unsigned* slots, int iters, unsigned* bad, unsigned* timeout) {
  const int nb = gridDim.x, b = blockIdx.x;
  unsigned wrong = 0, epoch = 0;
  auto sync = [&](bool rel, bool acq) {
    if (rel) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++epoch;
    if (threadIdx.x == 0) __hip_atomic_store(&flags[b], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)threadIdx.x < nb) {
      long long t0 = clock64();
      while (__hip_atomic_load(&flags[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
        if (clock64() - t0 > 400000000ll) { atomicAdd(timeout, 1u); break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (acq) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  };
  for (int i = 1; i <= iters; ++i) {
    if (MODE == 0 || MODE == 2) __hip_atomic_store(&slots[b * 512 + threadIdx.x], (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else slots[b * 512 + threadIdx.x] = i;
    sync(MODE == 1 || MODE == 3, MODE == 1 || MODE == 2);
    const int other = (b + 37) % nb;
    unsigned got;
    if (MODE == 0 || MODE == 3) got = __hip_atomic_load(&slots[other * 512 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else got = slots[other * 512 + threadIdx.x];
    wrong += got != (unsigned)i;
    sync(false, false);
  }
  if (wrong) atomicAdd(bad, wrong);
}

template <int MODE>
int run(const char* name, int grid, int iters) {
  unsigned *flags, *slots, *bad, *to;
  CK(hipMalloc(&flags, 1024 * 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&to, 4)); CK(hipMalloc(&slots, grid * 512 * 4));
  CK(hipMemset(slots, 0, grid * 512 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  unsigned hbad = 0, hto = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(flags, 0, 1024 * 4)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(to, 0, 4));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(barrier_kernel<MODE>, dim3(grid), dim3(512), 0, 0, flags, slots, iters, bad, to);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
    CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hto, to, 4, hipMemcpyDeviceToHost));
  }
  printf("%-46s grid %3d: %.2f us per pair of barriers, stale reads %u, timeouts %u\n", name, grid, best * 1e3 / iters, hbad, hto);
  return 0;
}

int main() {
  for (int grid : {128, 256}) {
    run<0>("coherent stores+loads, no fences", grid, 2000);
    run<1>("plain stores, release + acquire fences", grid, 2000);
    run<2>("write-through stores, acquire fence, plain loads", grid, 2000);
    run<3>("plain stores, release fence, coherent loads", grid, 2000);
  }
  return 0;
}
