// the evaluator side of the boundary for bundle adjustment in BAL form.
//
// What the reference does per LM iteration outside LinearSolver::Solve (ProgramEvaluator,
// I/program_evaluator.h:137-300, with autodiff Jets of examples/snavely_reprojection_error.h:
// 53-105 and BlockJacobianWriter, I/block_jacobian_writer.cc:68-167) is, for this one cost
// function, an embarrassingly parallel map over the observations: one thread per observation
// evaluates the Snavely reprojection residual and its ANALYTIC 2x9 / 2x3 Jacobian and writes
// both straight into the BlockSparseMatrix value layout the solver reads (E cells, then F cells:
// I/block_jacobian_writer.cc:141-162), already Jacobi-scaled (TrustRegionMinimizer,
// I/trust_region_minimizer.cc:263-279) so that no separate ScaleColumns pass exists.
// HBM-bound by its 208 B/observation of output; the arithmetic (~400 flop) is register resident.
#include <hip/hip_runtime.h>

#include "device.h"

namespace chip {

namespace {

__device__ __forceinline__ double wave_sum_e(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ double wave_max_e(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, 64));
  return v;
}

// Residual (and, JAC, the Jacobian) of one observation.  cam = [angle-axis(3) t(3) f k1 k2].
// jc = d res / d cam (2x9 row-major), jp = d res / d point (2x3 row-major).
template <bool JAC>
__device__ __forceinline__ void snavely(const double (&cam)[9], const double (&X)[3], double ox, double oy,
                                        double (&res)[2], double (&jc)[18], double (&jp)[6]) {
  const double a0 = cam[0], a1 = cam[1], a2 = cam[2];
  const double theta2 = a0 * a0 + a1 * a1 + a2 * a2;
  double P[3];
  double R[9];      // d P / d X
  double dPa[9];    // d P / d angle-axis, column j = derivative w.r.t. a_j, stored [row * 3 + j]
  if (theta2 != 0.0) {
    // Rodrigues: P = X cos + (w x X) sin + w (w.X)(1 - cos), w = a / theta   (include/ceres/rotation.h:864-905)
    const double theta = sqrt(theta2);
    const double c = cos(theta), s = sin(theta), inv = 1.0 / theta;
    const double w[3] = {a0 * inv, a1 * inv, a2 * inv};
    const double wxX[3] = {w[1] * X[2] - w[2] * X[1], w[2] * X[0] - w[0] * X[2], w[0] * X[1] - w[1] * X[0]};
    const double wdX = w[0] * X[0] + w[1] * X[1] + w[2] * X[2];
    const double omc = 1.0 - c;
    const double tmp = wdX * omc;
#pragma unroll
    for (int i = 0; i < 3; ++i) P[i] = X[i] * c + wxX[i] * s + w[i] * tmp;
    if constexpr (JAC) {
      // R = c I + s [w]x + (1 - c) w w^T
      R[0] = c + omc * w[0] * w[0];         R[1] = -s * w[2] + omc * w[0] * w[1];  R[2] = s * w[1] + omc * w[0] * w[2];
      R[3] = s * w[2] + omc * w[1] * w[0];  R[4] = c + omc * w[1] * w[1];          R[5] = -s * w[0] + omc * w[1] * w[2];
      R[6] = -s * w[1] + omc * w[2] * w[0]; R[7] = s * w[0] + omc * w[2] * w[1];   R[8] = c + omc * w[2] * w[2];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        // d theta / d a_j = w_j ;  d w / d a_j = (e_j - w w_j) / theta
        const double wj = w[j];
        double dw[3] = {-w[0] * wj * inv, -w[1] * wj * inv, -w[2] * wj * inv};
        dw[j] += inv;
        const double dwxX[3] = {dw[1] * X[2] - dw[2] * X[1], dw[2] * X[0] - dw[0] * X[2], dw[0] * X[1] - dw[1] * X[0]};
        const double dwdX = dw[0] * X[0] + dw[1] * X[1] + dw[2] * X[2];
        const double dtmp = dwdX * omc + wdX * s * wj;
#pragma unroll
        for (int i = 0; i < 3; ++i)
          dPa[i * 3 + j] = -X[i] * s * wj + dwxX[i] * s + wxX[i] * c * wj + dw[i] * tmp + w[i] * dtmp;
      }
    }
  } else {
    // first-order Taylor branch: P = X + a x X
    P[0] = X[0] + (a1 * X[2] - a2 * X[1]);
    P[1] = X[1] + (a2 * X[0] - a0 * X[2]);
    P[2] = X[2] + (a0 * X[1] - a1 * X[0]);
    if constexpr (JAC) {
      R[0] = 1.0; R[1] = -a2; R[2] = a1;
      R[3] = a2;  R[4] = 1.0; R[5] = -a0;
      R[6] = -a1; R[7] = a0;  R[8] = 1.0;
      // d (a x X) / d a_j = e_j x X
      dPa[0] = 0.0;   dPa[1] = X[2];  dPa[2] = -X[1];
      dPa[3] = -X[2]; dPa[4] = 0.0;   dPa[5] = X[0];
      dPa[6] = X[1];  dPa[7] = -X[0]; dPa[8] = 0.0;
    }
  }
  const double p0 = P[0] + cam[3], p1 = P[1] + cam[4], p2 = P[2] + cam[5];
  const double iz = 1.0 / p2;
  const double xp = -p0 * iz, yp = -p1 * iz;
  const double f = cam[6], k1 = cam[7], k2 = cam[8];
  const double r2 = xp * xp + yp * yp;
  const double dist = 1.0 + r2 * (k1 + k2 * r2);
  res[0] = f * dist * xp - ox;
  res[1] = f * dist * yp - oy;
  if constexpr (JAC) {
    const double g = k1 + 2.0 * k2 * r2;  // d dist / d r2
    // A = d res / d (xp, yp)
    const double A00 = f * (dist + 2.0 * g * xp * xp), A01 = f * 2.0 * g * xp * yp;
    const double A10 = A01, A11 = f * (dist + 2.0 * g * yp * yp);
    // d (xp, yp) / d p = [-1/z 0 x/z^2 ; 0 -1/z y/z^2] = [-iz 0 -xp iz ; 0 -iz -yp iz]
    const double J00 = -A00 * iz, J01 = -A01 * iz, J02 = -(A00 * xp + A01 * yp) * iz;
    const double J10 = -A10 * iz, J11 = -A11 * iz, J12 = -(A10 * xp + A11 * yp) * iz;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      jp[j] = J00 * R[j] + J01 * R[3 + j] + J02 * R[6 + j];
      jp[3 + j] = J10 * R[j] + J11 * R[3 + j] + J12 * R[6 + j];
      jc[j] = J00 * dPa[j] + J01 * dPa[3 + j] + J02 * dPa[6 + j];
      jc[9 + j] = J10 * dPa[j] + J11 * dPa[3 + j] + J12 * dPa[6 + j];
    }
    jc[3] = J00; jc[4] = J01; jc[5] = J02;
    jc[12] = J10; jc[13] = J11; jc[14] = J12;
    jc[6] = dist * xp;          jc[15] = dist * yp;
    jc[7] = f * r2 * xp;        jc[16] = f * r2 * yp;
    jc[8] = f * r2 * r2 * xp;   jc[17] = f * r2 * r2 * yp;
  }
}

template <bool JAC>
__global__ __launch_bounds__(kVecBlock) void bal_evaluate_kernel(BalEvalArgs A) {
  __shared__ double sh[4];
  double cost = 0.0;
  for (int64_t r = int64_t(blockIdx.x) * kVecBlock + threadIdx.x; r < A.n_rows; r += int64_t(gridDim.x) * kVecBlock) {
    const int c = A.row_cam[r], p = A.row_pt[r];
    const double2 o = A.row_obs[r];
    double cam[9], X[3], res[2], jc[18], jp[6];
    const double* cs = A.state + A.cam_base + 9 * int64_t(c);
    const double* ps = A.state + 3 * int64_t(p);
#pragma unroll
    for (int i = 0; i < 9; ++i) cam[i] = cs[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) X[i] = ps[i];
    snavely<JAC>(cam, X, o.x, o.y, res, jc, jp);
    cost += 0.5 * (res[0] * res[0] + res[1] * res[1]);
    if (A.residuals) reinterpret_cast<double2*>(A.residuals)[r] = make_double2(res[0], res[1]);
    if constexpr (JAC) {
      if (A.scale) {
        const double* sc = A.scale + A.cam_base + 9 * int64_t(c);
        const double* sp = A.scale + 3 * int64_t(p);
#pragma unroll
        for (int j = 0; j < 9; ++j) { const double v = sc[j]; jc[j] *= v; jc[9 + j] *= v; }
#pragma unroll
        for (int j = 0; j < 3; ++j) { const double v = sp[j]; jp[j] *= v; jp[3 + j] *= v; }
      }
      double2* e = reinterpret_cast<double2*>(A.values + 6 * r);                // 48 B per row, 16-byte aligned
      double2* fo = reinterpret_cast<double2*>(A.values + 6 * A.n_rows + 18 * r);  // 144 B per row
#pragma unroll
      for (int j = 0; j < 3; ++j) e[j] = make_double2(jp[2 * j], jp[2 * j + 1]);
#pragma unroll
      for (int j = 0; j < 9; ++j) fo[j] = make_double2(jc[2 * j], jc[2 * j + 1]);
    }
  }
  cost = wave_sum_e(cost);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = cost;
  __syncthreads();
  if (threadIdx.x == 0) A.partials[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// delta = step .* scale ; candidate = x + delta ; partials: |x|^2 at [b], |delta|^2 at [grid + b]
__global__ __launch_bounds__(kVecBlock) void bal_candidate_kernel(const double* x, const double* step, const double* scale,
                                                                  double* delta, double* cand, int64_t n, double* partials) {
  __shared__ double sh[8];
  double xn = 0, dn = 0;
  for (int64_t i = int64_t(blockIdx.x) * kVecBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kVecBlock) {
    const double d = scale ? step[i] * scale[i] : step[i];
    const double xi = x[i];
    delta[i] = d;
    cand[i] = xi + d;
    xn += xi * xi;
    dn += d * d;
  }
  xn = wave_sum_e(xn); dn = wave_sum_e(dn);
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = xn; sh[4 + (threadIdx.x >> 6)] = dn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    partials[gridDim.x + blockIdx.x] = (sh[4] + sh[5]) + (sh[6] + sh[7]);
  }
}

// max_i |g_i / scale_i|: the gradient of the UNSCALED problem from J^T f of the scaled Jacobian
__global__ __launch_bounds__(kVecBlock) void bal_gradient_max_kernel(const double* g, const double* scale, int64_t n, double* partials) {
  __shared__ double sh[4];
  double m = 0;
  for (int64_t i = int64_t(blockIdx.x) * kVecBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kVecBlock)
    m = fmax(m, fabs(scale ? g[i] / scale[i] : g[i]));
  m = wave_max_e(m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}

// scale = 1 / (1 + sqrt(squared column norm))      I/trust_region_minimizer.cc:263-270
__global__ __launch_bounds__(kVecBlock) void bal_jacobi_scale_kernel(const double* colnorm2, double* scale, int64_t n) {
  for (int64_t i = int64_t(blockIdx.x) * kVecBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kVecBlock)
    scale[i] = 1.0 / (1.0 + sqrt(colnorm2[i]));
}

int grid_for(int64_t n) {
  int64_t g = (n + kVecBlock - 1) / kVecBlock;
  return int(g < 1 ? 1 : (g > kMaxVecGrid ? kMaxVecGrid : g));
}

}  // namespace

hipError_t LaunchBalEvaluate(const BalEvalArgs& A, bool jacobian, int* nparts, hipStream_t stream) {
  // one observation per thread and several per thread at scale: 2048 workgroups keep 256 CUs busy
  int64_t g = (A.n_rows + kVecBlock - 1) / kVecBlock;
  const int grid = int(g < 1 ? 1 : (g > 2048 ? 2048 : g));
  *nparts = grid;
  if (jacobian) hipLaunchKernelGGL(bal_evaluate_kernel<true>, dim3(grid), dim3(kVecBlock), 0, stream, A);
  else hipLaunchKernelGGL(bal_evaluate_kernel<false>, dim3(grid), dim3(kVecBlock), 0, stream, A);
  return hipGetLastError();
}

hipError_t LaunchBalCandidate(const double* x, const double* step, const double* scale, double* delta, double* cand, int64_t n,
                              double* partials, int* nparts, hipStream_t stream) {
  const int grid = grid_for(n);
  *nparts = grid;
  hipLaunchKernelGGL(bal_candidate_kernel, dim3(grid), dim3(kVecBlock), 0, stream, x, step, scale, delta, cand, n, partials);
  return hipGetLastError();
}

hipError_t LaunchBalGradientMax(const double* g, const double* scale, int64_t n, double* partials, int* nparts, hipStream_t stream) {
  const int grid = grid_for(n);
  *nparts = grid;
  hipLaunchKernelGGL(bal_gradient_max_kernel, dim3(grid), dim3(kVecBlock), 0, stream, g, scale, n, partials);
  return hipGetLastError();
}

hipError_t LaunchBalJacobiScale(const double* colnorm2, double* scale, int64_t n, hipStream_t stream) {
  hipLaunchKernelGGL(bal_jacobi_scale_kernel, dim3(grid_for(n)), dim3(kVecBlock), 0, stream, colnorm2, scale, n);
  return hipGetLastError();
}

}  // namespace chip
