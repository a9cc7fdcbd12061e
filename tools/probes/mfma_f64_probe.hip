// mfma_f64_probe.hip — what v_mfma_f64_16x16x4_f64 sustains on this part: the ceiling DENSE_SCHUR's trailing update is measured against
// (MI355X_MICROARCH.md has no f64 row; AMD's datasheet says FP64 matrix = FP64 vector = 78.6 TFLOP/s).  Every wavefront runs ACC
// independent accumulator chains back to back, no memory traffic; waves per SIMD = 1, 2.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int ACC>
__global__ __launch_bounds__(256) void mfma_kernel(double* out, int iters, double a0, double b0) {
  v4f64 acc[ACC];
#pragma unroll
  for (int i = 0; i < ACC; ++i) acc[i] = v4f64{0.0, 0.0, 0.0, 0.0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < ACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 1.2345e-300) out[0] = s;
}
// v_mfma_f64_4x4x4_4b_f64: four 4 x 4 x 4 blocks per instruction (512 flop), one double of accumulator per lane
template <int ACC>
__global__ __launch_bounds__(256) void mfma4_kernel(double* out, int iters, double a0, double b0) {
  double acc[ACC];
#pragma unroll
  for (int i = 0; i < ACC; ++i) acc[i] = 0.0;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < ACC; ++i) s += acc[i];
  if (s == 1.2345e-300) out[0] = s;
}
// both pipes at once: 8 MFMA chains and 16 FMA chains interleaved in one wave (does the matrix pipe run beside the vector pipe?)
__global__ __launch_bounds__(256) void mixed_kernel(double* out, int iters, double a0, double b0) {
  v4f64 acc[8];
  double f[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < 16; ++i) f[i] = i;
  const double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      f[2 * i] = fma(a, f[2 * i], b);
      f[2 * i + 1] = fma(a, f[2 * i + 1], b);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += f[i];
  if (s == 1.2345e-300) out[0] = s;
}
__global__ __launch_bounds__(256) void fma_kernel(double* out, int iters, double a0, double b0) {   // the vector pipe, for comparison
  double acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = i;
  const double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = fma(a, acc[i], b);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  if (s == 1.2345e-300) out[0] = s;
}

template <typename K>
int run(const char* name, K kernel, int waves_per_simd, int per_iter_flops_per_wave, double* d_out) {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int grid = p.multiProcessorCount * waves_per_simd;   // 256 threads = 4 waves per workgroup = one per SIMD
  const int iters = 20000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, d_out, 100, 1.0, 1e-3);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, d_out, iters, 1.0, 1e-3);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double flops = double(grid) * 4 * iters * per_iter_flops_per_wave;
  const double cyc = ms * 1e-3 * p.clockRate * 1e3 / (double(iters) * waves_per_simd);   // cycles per iteration per SIMD at the nominal clock
  printf("%-28s waves/SIMD %d: %8.3f ms  %7.2f TFLOP/s  (%.1f nominal cycles per wave-iteration; clock %d MHz, %d CUs)\n", name, waves_per_simd, ms,
         flops / ms / 1e9, cyc, p.clockRate / 1000, p.multiProcessorCount);
  return 0;
}

int main() {
  double* d_out = nullptr;
  CK(hipMalloc(&d_out, 64));
  // round 4: more chains, more waves, the 4x4x4 form, both pipes at once; "nominal cycles per wave-iteration" / chains
  // = issue interval of one instruction on one SIMD
  for (int w = 1; w <= 8; w *= 2) {
    if (run("mfma_f64_16x16x4 x32 chains", mfma_kernel<32>, w, 32 * 2048, d_out)) return 1;
    if (run("mfma_f64_4x4x4_4b x16 chains", mfma4_kernel<16>, w, 16 * 512, d_out)) return 1;
    if (run("mfma_f64_4x4x4_4b x64 chains", mfma4_kernel<64>, w, 64 * 512, d_out)) return 1;
    if (run("8 mfma16 + 16 v_fma per iter", mixed_kernel, w, 8 * 2048 + 16 * 64 * 2, d_out)) return 1;
    if (run("v_fma_f64 x16 chains", fma_kernel, w, 16 * 64 * 2, d_out)) return 1;
  }
  for (int w = 1; w <= 2; ++w) {
    if (run("mfma_f64_16x16x4 x16 chains", mfma_kernel<16>, w, 16 * 2048, d_out)) return 1;
    if (run("mfma_f64_16x16x4 x4 chains", mfma_kernel<4>, w, 4 * 2048, d_out)) return 1;
    if (run("mfma_f64_16x16x4 x1 chain", mfma_kernel<1>, w, 1 * 2048, d_out)) return 1;
    if (run("v_fma_f64 x16 chains", fma_kernel, w, 16 * 64 * 2, d_out)) return 1;
  }
  return 0;
}
