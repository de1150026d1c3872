// Micro-benchmark: cost of a grid-wide barrier among co-resident workgroups on MI355X (one workgroup per CU), and whether data written
// before the barrier by one workgroup is visible after it to a workgroup on another XCD, for three flavours of fencing.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip && ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void barrier_kernel(unsigned* ctr, unsigned* slots, int iters, unsigned* bad, unsigned* timeout) {
  unsigned target = 0;
  const int nb = gridDim.x, b = blockIdx.x;
  unsigned wrong = 0;
  for (int i = 1; i <= iters; ++i) {
    // every thread publishes one word
    if (MODE == 0) slots[b * 512 + threadIdx.x] = i;                                             // plain store
    else if (MODE == 1) slots[b * 512 + threadIdx.x] = i;                                        // plain store + fences below
    else __hip_atomic_store(&slots[b * 512 + threadIdx.x], (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through
    __syncthreads();
    if (threadIdx.x == 0) {
      target += nb;
      if (MODE == 1) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      else {
        if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      long long t0 = clock64();
      while (true) {
        unsigned v = (MODE == 1) ? __hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
                                 : __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= target) break;
        if (clock64() - t0 > 400000000ll) { atomicAdd(timeout, 1u); break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    const int other = (b + 37) % nb;      // a workgroup on another XCD (round-robin placement)
    unsigned got;
    if (MODE == 2) got = __hip_atomic_load(&slots[other * 512 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else got = slots[other * 512 + threadIdx.x];
    wrong += got != (unsigned)i;
    __syncthreads();                       // (second barrier below keeps iteration i+1's stores behind these loads)
    if (threadIdx.x == 0) {
      target += nb;
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      long long t0 = clock64();
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (clock64() - t0 > 400000000ll) { atomicAdd(timeout, 1u); break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
  }
  if (wrong) atomicAdd(bad, wrong);
}

template <int MODE>
int run(const char* name, int grid, int iters) {
  unsigned *ctr, *slots, *bad, *to;
  CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&to, 4)); CK(hipMalloc(&slots, grid * 512 * 4));
  CK(hipMemset(slots, 0, grid * 512 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  unsigned hbad = 0, hto = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(ctr, 0, 4)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(to, 0, 4));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(barrier_kernel<MODE>, dim3(grid), dim3(512), 0, 0, ctr, slots, iters, bad, to);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
    CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hto, to, 4, hipMemcpyDeviceToHost));
  }
  printf("%-34s grid %3d: %.2f us per barrier pair (%.2f per barrier), stale reads %u, timeouts %u\n", name, grid, best * 1e3 / iters,
         best * 1e3 / iters / 2, hbad, hto);
  return 0;
}

int main() {
  for (int grid : {64, 256}) {
    run<0>("plain stores, relaxed atomics", grid, 2000);
    run<1>("plain stores, release/acquire", grid, 2000);
    run<2>("agent-scope stores+loads, relaxed", grid, 2000);
  }
  return 0;
}
