// xcd_atomic_probe.hip — can camera-space sums that do not fit in LDS be accumulated in the XCDs' L2s?
// The many-camera regime (BASELINE.json configs[4]: 50 000 cameras x 9 doubles = 3.6 MB of accumulators) spills F^T z rows to a ring
// and sums them in a second pass, because device-scope global_atomic_add_f64 runs memory-side at 9-15 G atomics/s (design/03 §3.1).
// An XCD's 4 MB L2 would hold a PRIVATE copy of the accumulators: atomics of workgroups on that XCD to that copy need no coherence
// with any other XCD, so they can be issued at a scope below `agent` (no sc1 bit: the L2 executes them and keeps the line) — one
// accumulator copy per XCD, selected by the hardware's XCC id, summed by a small kernel afterwards.  This probe measures the rate:
//   mode 0  agent-scope atomics, one shared copy (what design/03 measured)
//   mode 1  agent-scope atomics, one copy per XCD
//   mode 2  atomics without scope bits (inline asm), one copy per XCD
//   mode 3  mode 2 while the same waves stream a tile-shaped read (12 x 16 B per lane and iteration, non-temporal) through the L2;
//           SPILL of every 1000 lanes do the nine atomics (the hybrid plan keeps the rest in LDS)
//   mode 4  the stream of mode 3 alone
// Every atomic adds 1.0: the sum over all copies must equal the number of atomics issued (checked).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
  return v;
}
__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ void atomic_add_noscope(double* p, double v) {
  asm volatile("global_atomic_add_f64 %0, %1, off" : : "v"(p), "v"(v) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(512) void probe_kernel(double* acc, long n_per_copy, int n_cams, int iters, const double2* stream, long stream_tiles,
                                                    int spill_per_1000, unsigned long long* issued, double* sink) {
  const unsigned xcc = xcc_id() & 7u;
  double* mine = (MODE == 0) ? acc : acc + long(xcc) * n_per_copy;
  const unsigned gid = blockIdx.x * 512 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const long wave = long(blockIdx.x) * 8 + (threadIdx.x >> 6), nwaves = long(gridDim.x) * 8;
  unsigned long long cnt = 0;
  double s = 0;
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 3) {
      const long tile = (wave + long(it) * nwaves) % stream_tiles;
      const double2* p = stream + tile * (12 * 64) + lane;
      typedef int v4 __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const v4 r = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p + j * 64));
        s += double(r.x ^ r.y ^ r.z ^ r.w);
      }
    }
    if (MODE == 4) continue;
    const unsigned h = hash32(gid * 2654435761u + unsigned(it) * 40503u + 17u);
    if (MODE == 3 && int(h % 1000u) >= spill_per_1000) continue;
    const int cam = int(hash32(h + 0x9e3779b9u) % unsigned(n_cams));
    double* base = mine + 9 * long(cam);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      if (MODE <= 1) unsafeAtomicAdd(base + k, 1.0);
      else atomic_add_noscope(base + k, 1.0);
    }
    cnt += 9;
  }
  if (MODE != 4) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) cnt += __shfl_xor(cnt, m, 64);
    if (lane == 0) atomicAdd(issued, cnt);
  }
  if (s == 1.2345e-300) sink[0] = s;
}

__global__ void sum_kernel(const double* acc, long n, double* out) {
  double s = 0;
  for (long i = blockIdx.x * 256 + threadIdx.x; i < n; i += long(gridDim.x) * 256) s += acc[i];
  atomicAdd(out, s);
}

int main(int argc, char** argv) {
  const int n_cams = argc > 1 ? atoi(argv[1]) : 50000;
  const long n_per_copy = 9L * n_cams;
  const int grid = 256, iters = 229;   // 256 x 512 lanes x 229 = 30 M "observations"
  const long stream_tiles = 262144;    // x 12 KiB = 3 GiB
  double *acc = nullptr, *out = nullptr, *sink = nullptr; double2* stream = nullptr; unsigned long long* issued = nullptr;
  CK(hipMalloc(&acc, 8 * n_per_copy * sizeof(double)));
  CK(hipMalloc(&out, 8)); CK(hipMalloc(&sink, 8)); CK(hipMalloc(&issued, 8));
  CK(hipMalloc(&stream, size_t(stream_tiles) * 12 * 64 * sizeof(double2)));
  CK(hipMemset(stream, 0, size_t(stream_tiles) * 12 * 64 * sizeof(double2)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("cameras %d (%.2f MB of accumulators per copy), %d workgroups x 512 lanes x %d iterations\n", n_cams, n_per_copy * 8 / 1e6, grid, iters);
  for (int mode = 0; mode <= 4; ++mode) {
    for (int spill : {440, 1000}) {
      if (mode != 3 && spill != 440) continue;
      float best = 1e30f; double total = 0; unsigned long long n_issued = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(acc, 0, 8 * n_per_copy * sizeof(double))); CK(hipMemset(out, 0, 8)); CK(hipMemset(issued, 0, 8));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        switch (mode) {
          case 0: hipLaunchKernelGGL((probe_kernel<0>), dim3(grid), dim3(512), 0, 0, acc, n_per_copy, n_cams, iters, stream, stream_tiles, spill, issued, sink); break;
          case 1: hipLaunchKernelGGL((probe_kernel<1>), dim3(grid), dim3(512), 0, 0, acc, n_per_copy, n_cams, iters, stream, stream_tiles, spill, issued, sink); break;
          case 2: hipLaunchKernelGGL((probe_kernel<2>), dim3(grid), dim3(512), 0, 0, acc, n_per_copy, n_cams, iters, stream, stream_tiles, spill, issued, sink); break;
          case 3: hipLaunchKernelGGL((probe_kernel<3>), dim3(grid), dim3(512), 0, 0, acc, n_per_copy, n_cams, iters, stream, stream_tiles, spill, issued, sink); break;
          case 4: hipLaunchKernelGGL((probe_kernel<4>), dim3(grid), dim3(512), 0, 0, acc, n_per_copy, n_cams, iters, stream, stream_tiles, spill, issued, sink); break;
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        hipLaunchKernelGGL(sum_kernel, dim3(256), dim3(256), 0, 0, acc, 8 * n_per_copy, out);
        CK(hipMemcpy(&total, out, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&n_issued, issued, 8, hipMemcpyDeviceToHost));
      }
      const double gb = (mode >= 3) ? double(grid) * 8 * iters * 12 * 1024 / 1e9 : 0.0;
      printf("mode %d%s: %.3f ms, %llu atomics (%.1f G atomics/s), sum of the copies %.0f (%s)", mode, mode == 3 ? (spill == 440 ? " spill 44 %" : " spill 100 %") : "",
             best, n_issued, n_issued / (best * 1e-3) / 1e9, total, (mode == 4 || total == double(n_issued)) ? "exact" : "WRONG");
      if (gb > 0) printf(", stream %.2f GB = %.0f GB/s", gb, gb / (best * 1e-3));
      printf("\n");
    }
  }
  return 0;
}
