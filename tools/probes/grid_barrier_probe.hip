// grid_barrier_probe.hip — what a device-side grid barrier costs on MI355X (8 XCDs, one L2 each): the price a persistent
// "one kernel per LM step" design would pay at every phase boundary instead of a kernel launch.
// One workgroup per CU (all resident: the spin cannot starve anybody), NB barriers in a row; a barrier = every workgroup's thread 0
// does a device-scope release, an atomic add on one counter, and spins on a device-scope load until the counter reaches the
// round's target; the other threads wait at a workgroup barrier.  Two payload variants between barriers: none, and a small
// produce/consume (each workgroup writes 1 KiB write-through, and after the barrier reads its neighbour's 1 KiB): the hand-off a
// real phase boundary needs.  Every spin is bounded (the kernel gives up and reports it instead of hanging).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool PAYLOAD>
__global__ __launch_bounds__(256) void barrier_kernel(unsigned int* counter, int nb, double* buf, unsigned int* gave_up, double* sink) {
  __shared__ int ok;
  double acc = 0;
  for (int b = 0; b < nb; ++b) {
    if (PAYLOAD) {   // produce: 1 KiB per workgroup, write-through to device scope
      double* mine = buf + (size_t(b & 1) * gridDim.x + blockIdx.x) * 128;
      if (threadIdx.x < 128) asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(mine + threadIdx.x), "v"(double(b + threadIdx.x)) : "memory");
      asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      atomicAdd(counter, 1u);
      const unsigned int target = unsigned(b + 1) * gridDim.x;
      int spins = 0;
      ok = 1;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > 2000000) { ok = 0; atomicAdd(gave_up, 1u); break; }
      }
    }
    __syncthreads();
    if (!ok) return;
    if (PAYLOAD) {   // consume the neighbour's block
      __threadfence();
      const double* theirs = buf + (size_t(b & 1) * gridDim.x + (blockIdx.x + 37) % gridDim.x) * 128;
      if (threadIdx.x < 128) acc += theirs[threadIdx.x];
    }
  }
  if (acc == 1.2345e-300) sink[0] = acc;
}

// Two cheaper barriers for comparison: `sleep` = the same single counter with s_sleep in the spin loop (less traffic against the
// atomics), `tree` = per-XCD counters (workgroup b arrives at counter b % 8: the round-robin of workgroups over XCDs), the last
// arrival of an XCD arrives at the global counter, the last of those raises a flag everybody spins on (with s_sleep).
template <int KIND>   // 1 sleep, 2 tree
__global__ __launch_bounds__(256) void barrier2_kernel(unsigned int* c, int nb, unsigned int* gave_up) {
  __shared__ int ok;
  unsigned int* xcd = c + 16 * (1 + blockIdx.x % 8);   // counters 64 bytes apart
  unsigned int* global = c, *flag = c + 16 * 9;
  const unsigned per_xcd = gridDim.x / 8;
  for (int b = 0; b < nb; ++b) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      ok = 1;
      int spins = 0;
      if (KIND == 1) {
        atomicAdd(global, 1u);
        const unsigned target = unsigned(b + 1) * gridDim.x;
        while (__hip_atomic_load(global, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > 1000000) { ok = 0; atomicAdd(gave_up, 1u); break; }
        }
      } else {
        if (atomicAdd(xcd, 1u) == per_xcd * unsigned(b + 1) - 1u)
          if (atomicAdd(global, 1u) == 8u * unsigned(b + 1) - 1u) __hip_atomic_store(flag, unsigned(b + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < unsigned(b + 1)) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > 1000000) { ok = 0; atomicAdd(gave_up, 1u); break; }
        }
      }
    }
    __syncthreads();
    if (!ok) return;
  }
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int grid = prop.multiProcessorCount;   // one workgroup per CU
  unsigned int *counter = nullptr, *gave_up = nullptr; double *buf = nullptr, *sink = nullptr;
  CK(hipMalloc(&counter, 4)); CK(hipMalloc(&gave_up, 4)); CK(hipMalloc(&buf, size_t(2) * grid * 128 * 8)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 0, size_t(2) * grid * 128 * 8)); CK(hipMemset(gave_up, 0, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int payload = 0; payload < 2; ++payload) {
    for (int nb : {1, 101, 1001}) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(counter, 0, 4));
        CK(hipEventRecord(e0));
        if (payload) hipLaunchKernelGGL((barrier_kernel<true>), dim3(grid), dim3(256), 0, 0, counter, nb, buf, gave_up, sink);
        else hipLaunchKernelGGL((barrier_kernel<false>), dim3(grid), dim3(256), 0, 0, counter, nb, buf, gave_up, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      static float first = 0;
      if (nb == 1) first = best;
      printf("%s %4d barriers of %d workgroups: %.1f us per kernel", payload ? "produce/consume" : "bare           ", nb, grid, best * 1e3);
      if (nb > 1) printf(", %.2f us per barrier beyond the first", (best - first) * 1e3 / (nb - 1));
      printf("\n");
      fflush(stdout);
    }
  }
  {
    unsigned int* c2 = nullptr; CK(hipMalloc(&c2, 4096));
    for (int kind = 1; kind <= 2; ++kind) {
      float t1 = 0;
      for (int nb : {1, 1001}) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipMemset(c2, 0, 4096));
          CK(hipEventRecord(e0));
          if (kind == 1) hipLaunchKernelGGL((barrier2_kernel<1>), dim3(grid), dim3(256), 0, 0, c2, nb, gave_up);
          else hipLaunchKernelGGL((barrier2_kernel<2>), dim3(grid), dim3(256), 0, 0, c2, nb, gave_up);
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
          best = ms < best ? ms : best;
        }
        if (nb == 1) t1 = best;
        else printf("%s: %.2f us per barrier (bare, %d workgroups)\n", kind == 1 ? "one counter + s_sleep in the spin" : "per-XCD counters -> global counter -> flag, s_sleep", (best - t1) * 1e3 / (nb - 1), grid);
        fflush(stdout);
      }
    }
  }
  unsigned int g = 0; CK(hipMemcpy(&g, gave_up, 4, hipMemcpyDeviceToHost));
  printf("workgroups that gave up spinning: %u\n", g);
  // for comparison: an empty kernel launched back to back
  {
    CK(hipEventRecord(e0));
    for (int r = 0; r < 1000; ++r) hipLaunchKernelGGL((barrier_kernel<false>), dim3(grid), dim3(256), 0, 0, counter, 0, buf, gave_up, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("1000 empty launches of %d workgroups back to back: %.2f us each\n", grid, ms);
  }
  return 0;
}
