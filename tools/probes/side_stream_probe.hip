// side_stream_probe.hip — the camera-major {F_o, M_o} side stream the round-2 review asked to MEASURE: what
// it costs to WRITE 176-byte records (144 B of F + 32 B of M_o) of 5 M observations to camera-major positions from a pass that
// walks the observations in point order, and what it gains to READ them back coalesced instead of gathering 144 + 32-byte
// pieces from the caller's rows.  Three writers:
//   lane     every lane stores its own record as eleven 16-byte stores (a 176-byte stride across the wavefront),
//   coop     the wavefront packs its 64 records through LDS and stores them so that eleven consecutive lanes cover one record
//            (whole-record cooperative stores, as the review proposed), plain stores,
//   coop_nt  the same with non-temporal stores;
// and two readers: `gather` (what bal_camera_items_kernel does today: 18 doubles from the caller's 24-double rows + a 32-byte
// record, in camera-major order) and `stream` (the 176-byte records back to back).  Positions: 1778 cameras of skewed popularity,
// ascending rows inside a camera (the Venice-shaped workload of bench.py).  Prints ms per pass and GB/s on the useful bytes.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kRec = 22;   // doubles per record: 18 of F + 4 of M_o

__global__ __launch_bounds__(256) void write_lane_kernel(const double2* __restrict__ src, const int* __restrict__ dst_of, int n, double2* __restrict__ dst) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const double2* s = src + int64_t(i) * 12;          // the observation's 24 doubles in point order (a tile stream stands in for them)
    double2* d = dst + int64_t(dst_of[i]) * (kRec / 2);
#pragma unroll
    for (int k = 0; k < kRec / 2; ++k) d[k] = s[k];
  }
}

template <bool NT>
__global__ __launch_bounds__(256) void write_coop_kernel(const double2* __restrict__ src, const int* __restrict__ dst_of, int n, double2* __restrict__ dst) {
  __shared__ double2 stage[4][64 * (kRec / 2)];
  __shared__ int where[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int base = (blockIdx.x * 4 + wv) * 64; base < n; base += gridDim.x * 256) {
    const int i = base + lane;
    const bool in = i < n;
    if (in) {
      const double2* s = src + int64_t(i) * 12;
#pragma unroll
      for (int k = 0; k < kRec / 2; ++k) stage[wv][lane * (kRec / 2) + k] = s[k];
      where[wv][lane] = dst_of[i];
    }
    __builtin_amdgcn_wave_barrier();
    const int cnt = min(64, n - base);
    // element e of the wave's 64 x 11 pairs: record e / 11, pair e % 11 -> eleven consecutive lanes write one record's 176 bytes
#pragma unroll
    for (int j = 0; j < kRec / 2; ++j) {
      const int e = j * 64 + lane, r = e / (kRec / 2), k = e - r * (kRec / 2);
      if (r < cnt) {
        double2* p = dst + int64_t(where[wv][r]) * (kRec / 2) + k;
        const double2 v = stage[wv][e];
        if (NT) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); }
        else *p = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ __launch_bounds__(256) void read_gather_kernel(const double* __restrict__ rows, const double2* __restrict__ mo, const int* __restrict__ row_of, int n, double* out) {
  double a = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int r = row_of[i];
    const double* f = rows + int64_t(r) * 24 + 6;
#pragma unroll
    for (int k = 0; k < 18; ++k) a += f[k];
    const double2 m0 = mo[2 * int64_t(r)], m1 = mo[2 * int64_t(r) + 1];
    a += m0.x + m0.y + m1.x;
  }
  if (a == 1.2345e-300) out[0] = a;
}

__global__ __launch_bounds__(256) void read_stream_kernel(const double2* __restrict__ rec, int64_t n_pairs, double* out) {
  double a = 0;
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n_pairs; i += int64_t(gridDim.x) * 256) { const double2 t = rec[i]; a += t.x + t.y; }
  if (a == 1.2345e-300) out[0] = a;
}

int main() {
  const int n = 5001946, ncam = 1778;
  std::mt19937 rng(7);
  std::vector<double> w(ncam); for (int c = 0; c < ncam; ++c) w[c] = std::pow(c + 1.0, -0.6);
  std::discrete_distribution<int> d(w.begin(), w.end());
  std::vector<int> cam(n); for (int i = 0; i < n; ++i) cam[i] = d(rng);
  std::vector<int> idx(n); std::iota(idx.begin(), idx.end(), 0);
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return cam[a] < cam[b]; });   // camera-major list of rows
  std::vector<int> dst_of(n); for (int q = 0; q < n; ++q) dst_of[idx[q]] = q;             // row -> its place in the camera-major stream
  double2 *src = nullptr, *dst = nullptr, *mo = nullptr; double *rows = nullptr, *out = nullptr; int *d_dst = nullptr, *d_row = nullptr;
  CK(hipMalloc(&src, size_t(n) * 12 * 16)); CK(hipMemset(src, 0, size_t(n) * 12 * 16));
  CK(hipMalloc(&dst, size_t(n) * kRec * 8)); CK(hipMemset(dst, 0, size_t(n) * kRec * 8));
  CK(hipMalloc(&rows, size_t(n) * 24 * 8)); CK(hipMemset(rows, 0, size_t(n) * 24 * 8));
  CK(hipMalloc(&mo, size_t(n) * 32)); CK(hipMemset(mo, 0, size_t(n) * 32));
  CK(hipMalloc(&out, 64)); CK(hipMalloc(&d_dst, n * 4)); CK(hipMalloc(&d_row, n * 4));
  CK(hipMemcpy(d_dst, dst_of.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_row, idx.data(), n * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * 8;
  auto timed = [&](const char* name, double useful_bytes, auto launch) -> int {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      for (int r = 0; r < 10; ++r) launch();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms / 10);
    }
    printf("%-28s %.4f ms  %.0f GB/s on %.0f MB useful\n", name, best, useful_bytes / best / 1e6, useful_bytes / 1e6);
    return 0;
  };
  const double wr = double(n) * (192 + 176), rd_g = double(n) * 176, rd_s = double(n) * 176;
  if (timed("write lane (read 192 + 176)", wr, [&] { hipLaunchKernelGGL(write_lane_kernel, dim3(grid), dim3(256), 0, 0, src, d_dst, n, dst); })) return 1;
  if (timed("write coop", wr, [&] { hipLaunchKernelGGL((write_coop_kernel<false>), dim3(grid), dim3(256), 0, 0, src, d_dst, n, dst); })) return 1;
  if (timed("write coop_nt", wr, [&] { hipLaunchKernelGGL((write_coop_kernel<true>), dim3(grid), dim3(256), 0, 0, src, d_dst, n, dst); })) return 1;
  if (timed("read-only of the source", double(n) * 192, [&] { hipLaunchKernelGGL(read_stream_kernel, dim3(grid), dim3(256), 0, 0, src, int64_t(n) * 12, out); })) return 1;
  if (timed("read gather (today)", rd_g, [&] { hipLaunchKernelGGL(read_gather_kernel, dim3(grid), dim3(256), 0, 0, rows, mo, d_row, n, out); })) return 1;
  if (timed("read stream (side stream)", rd_s, [&] { hipLaunchKernelGGL(read_stream_kernel, dim3(grid), dim3(256), 0, 0, dst, int64_t(n) * (kRec / 2), out); })) return 1;
  return 0;
}
