// gather_probe.hip — what a camera-major gather of 144-byte cells costs, and what FETCH_SIZE reports for it.
// N cells of CELL bytes (16-byte aligned, random positions in a buffer of BUF bytes, each read exactly once) are read by
// wavefronts whose lanes take one cell each (nine 16-byte loads or eighteen 8-byte loads per lane), as
// bal_camera_items_kernel reads F.  Known useful bytes = N * CELL; compare with rocprofv3 --pmc FETCH_SIZE of the same run
// (tools/gpu_r02.sh gather_pmc) to calibrate the counter for this access pattern.  Also: the same with the cells in
// ascending address order (what a camera's observation list looks like: increasing row ids, far apart).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <numeric>
#include <cmath>
#include <string>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int W>  // W = 8 or 16 bytes per load
__global__ __launch_bounds__(256) void gather_kernel(const double* __restrict__ buf, const int* __restrict__ pos, int n, int cell_doubles, double* out) {
  double a = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const double* p = buf + pos[i];
    if (W == 16) { const double2* q = reinterpret_cast<const double2*>(p); for (int k = 0; k < cell_doubles / 2; ++k) { const double2 t = q[k]; a += t.x + t.y; } }
    else for (int k = 0; k < cell_doubles; ++k) a += p[k];
  }
  if (a == 1.2345e-300) out[0] = a;
}

int main(int argc, char** argv) {
  const int n = 5001946, cell = 18;                  // Venice: 5 M observations, F cells of 18 doubles
  const size_t buf_doubles = size_t(n) * 24;         // the caller's value array: 24 doubles per observation
  std::vector<int> pos(n);
  for (int i = 0; i < n; ++i) pos[i] = 6 * n + 18 * i;  // E|F-split layout: F cells behind the E cells, row order
  double *buf = nullptr, *out = nullptr; int* dpos = nullptr;
  CK(hipMalloc(&buf, buf_doubles * 8)); CK(hipMemset(buf, 0, buf_doubles * 8)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&dpos, n * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::mt19937 rng(7);
  const char* only = argc > 1 ? argv[1] : "";
  for (int order = 0; order < 3; ++order) {
    // 0: row order (a sequential stream, the ceiling); 1: random permutation; 2: camera-major (1778 cameras, skewed, ascending rows inside a camera)
    std::vector<int> p = pos;
    if (order == 1) std::shuffle(p.begin(), p.end(), rng);
    if (order == 2) {
      std::vector<double> w(1778); for (int c = 0; c < 1778; ++c) w[c] = std::pow(c + 1.0, -0.6);
      std::discrete_distribution<int> d(w.begin(), w.end());
      std::vector<int> cam(n); for (int i = 0; i < n; ++i) cam[i] = d(rng);
      std::vector<int> idx(n); std::iota(idx.begin(), idx.end(), 0);
      std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return cam[a] < cam[b]; });
      for (int i = 0; i < n; ++i) p[i] = pos[idx[i]];
    }
    CK(hipMemcpy(dpos, p.data(), n * 4, hipMemcpyHostToDevice));
    for (int w : {8, 16}) {
      char tag[64]; snprintf(tag, sizeof(tag), "order%d_w%d", order, w);
      if (only[0] && std::string(only) != tag) continue;
      const int grid = 256 * 8;
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < 10; ++r) {
          if (w == 8) hipLaunchKernelGGL((gather_kernel<8>), dim3(grid), dim3(256), 0, 0, buf, dpos, n, cell, out);
          else hipLaunchKernelGGL((gather_kernel<16>), dim3(grid), dim3(256), 0, 0, buf, dpos, n, cell, out);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      }
      const double useful = double(n) * cell * 8;
      printf("%s  order=%s  load=%2d B  %8.1f us   useful %.1f MB  -> %.0f GB/s of useful bytes\n", tag,
             order == 0 ? "row (stream)" : order == 1 ? "random" : "camera-major", w, 1e3 * ms / 10, useful / 1e6, useful / (ms / 10 * 1e-3) / 1e9);
    }
  }
  return 0;
}
