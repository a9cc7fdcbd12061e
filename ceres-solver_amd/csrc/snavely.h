// snavely.h — the Snavely reprojection residual and its analytic Jacobian for one observation, shared by the
// evaluator kernels (kernels_evaluator.hip) and the camera-major preconditioner pass that evaluates its F cells on the fly
// (kernels_bal.inc, <2,3,9> shape).  Formula: examples/snavely_reprojection_error.h:53-105 with AngleAxisRotatePoint,
// include/ceres/rotation.h:864-905; the oracle differentiates the same formula with dual numbers (oracle/bal_harness.cc).
#pragma once
#include <hip/hip_runtime.h>

namespace chip {

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

}  // namespace chip
