// kernels_cg.hip — device-resident preconditioned conjugate gradients.
//
// Restates ConjugateGradientsSolver<V> (I/conjugate_gradients_solver.h:108-306) and the
// BLAS-1 it is written against (Norm/Dot/Axpby/SetZero/Copy, I/eigen_vector_ops.h:47-101)
// as a fixed sequence of stream-ordered kernels that never need the host inside an
// iteration:
//
//   precondition   z = M^-1 r ; partial r.z                     (:162-167)
//   direction      rho, beta ; p = z + beta p                    (:167-191)
//   <operator>     q = A p            (q lives in z, as in the reference :193-195)
//   dot_pq         partial p.q                                   (:196)
//   step           pq, alpha ; x += alpha p ; r -= alpha q ; partial Q1, |r|^2   (:208-249)
//                  (every residual_reset_period-th iteration instead: <operator> tmp = A x,
//                   residual_reset r = rhs - tmp, :235-239)
//   finalize       zeta / |r| / max-iteration tests in the reference's order, iter++ (:273-302)
//
// All scalars (rho, alpha, beta, Q0, Q1, the status word) stay in HBM in CgScalars; every
// inner product is a two-stage reduction — wavefront shuffles (__shfl_xor over 64 lanes),
// one partial per workgroup, and the CONSUMER kernel re-sums the <= 512 partials in a
// fixed order, so results are deterministic and no atomics or grid barriers are needed.
// Once the status word is non-zero every kernel returns at once, which lets the host
// enqueue a batch of iterations and poll the word only every few iterations.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "device.h"
#include "p2p.h"

namespace chip {

namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// Sum over the workgroup; every thread gets the result.  kVecBlock = 256 = 4 waves.
__device__ __forceinline__ double block_sum(double v, double* sh /* >= 4 */) {
  v = wave_sum(v);
  __syncthreads();  // protect sh from a previous use
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// Three sums at once (one pair of barriers instead of three).
__device__ __forceinline__ void block_sum3(double& a, double& b, double& c, double* sh /* >= 12 */) {
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; sh[w] = a; sh[4 + w] = b; sh[8 + w] = c; }
  __syncthreads();
  a = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  b = (sh[4] + sh[5]) + (sh[6] + sh[7]);
  c = (sh[8] + sh[9]) + (sh[10] + sh[11]);
}

struct Span { int64_t i, end, step; };

// Element range of this thread: workgroups [0, grid_e) stride over the shard
// [0, n_local), the others over [n_local, n).
__device__ __forceinline__ Span my_span(const CgBuffers& B) {
  Span s;
  if (int(blockIdx.x) < B.grid_e) {
    s.i = int64_t(blockIdx.x) * kVecBlock + threadIdx.x;
    s.end = B.n_local;
    s.step = int64_t(B.grid_e) * kVecBlock;
  } else {
    s.i = B.n_local + int64_t(blockIdx.x - B.grid_e) * kVecBlock + threadIdx.x;
    s.end = B.n;
    s.step = int64_t(B.grid - B.grid_e) * kVecBlock;
  }
  return s;
}

// Full inner product from a slot of workgroup partials (+ the all-reduced shard share).
__device__ __forceinline__ double total_of(const CgBuffers& B, int slot, double* sh) {
  const double* p = B.partials + slot * kMaxVecGrid;
  double v = 0;
  for (int k = B.grid_e + threadIdx.x; k < B.grid; k += kVecBlock) v += p[k];
  double t = block_sum(v, sh);
  if (B.grid_e > 0) t += B.comm[slot];
  return t;
}

__device__ __forceinline__ bool zero_or_inf(double v) { return v == 0.0 || isinf(v); }

__global__ __launch_bounds__(kVecBlock) void set_kernel(double* x, double v, int64_t n) {
  for (int64_t i = int64_t(blockIdx.x) * kVecBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kVecBlock) x[i] = v;
}
__global__ __launch_bounds__(kVecBlock) void axpby_kernel(double a, const double* x, double b, const double* y, double* z, int64_t n) {
  for (int64_t i = int64_t(blockIdx.x) * kVecBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kVecBlock) z[i] = a * x[i] + b * y[i];
}
template <bool ADD>
__global__ __launch_bounds__(kVecBlock) void square_scale_kernel(const double* D, const double* x, double* y, int64_t n, const int* status) {
  if (status && *status != 0) return;
  for (int64_t i = int64_t(blockIdx.x) * kVecBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kVecBlock) {
    const double d = D ? D[i] : 0.0;
    if (ADD) y[i] += d * d * x[i]; else y[i] = d * d * x[i];
  }
}
__global__ __launch_bounds__(kVecBlock) void dot_partial_kernel(const double* x, const double* y, int64_t n, double* partials) {
  __shared__ double sh[4];
  double v = 0;
  for (int64_t i = int64_t(blockIdx.x) * kVecBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kVecBlock) v += x[i] * y[i];
  v = block_sum(v, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = v;
}
__global__ __launch_bounds__(kVecBlock) void dot_final_kernel(const double* partials, int n, double* out) {
  __shared__ double sh[4];
  double v = 0;
  for (int k = threadIdx.x; k < n; k += kVecBlock) v += partials[k];
  v = block_sum(v, sh);
  if (threadIdx.x == 0) out[0] = v;
}
// the packed upper triangle of a point's ne x ne block (row by row, `pitch` doubles per point) -> the dense block
__global__ void expand_sym_kernel(const double* packed, int ne, int pitch, double* dense, const int64_t* off, int n) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const double* a = packed + int64_t(p) * pitch;
  double* o = dense + (off ? off[p] : int64_t(ne) * ne * p);
  for (int i = 0; i < ne; ++i)
    for (int j = 0; j < ne; ++j) {
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      o[ne * i + j] = a[lo * (2 * ne + 1 - lo) / 2 + (hi - lo)];
    }
}

// out[off[p] + k] = in[9 p + k]: the dense 3x3 point blocks of a CGNR solve (internal point order) into the caller's block order
__global__ void scatter_blocks9_kernel(const double* in, double* out, const int64_t* off, int n) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= int64_t(9) * n) return;
  const int p = int(i / 9), k = int(i - int64_t(9) * p);
  out[off[p] + k] = in[i];
}

__global__ __launch_bounds__(kVecBlock) void lm_diagonal_kernel(double* diag, double lo, double hi, double radius, double* D, int64_t n) {
  for (int64_t i = int64_t(blockIdx.x) * kVecBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kVecBlock) {
    const double d = fmin(fmax(diag[i], lo), hi);
    diag[i] = d;
    D[i] = sqrt(d / radius);
  }
}
__global__ __launch_bounds__(kVecBlock) void negate_and_check_kernel(double* x, int64_t n, int* nonfinite) {
  int bad = 0;
  for (int64_t i = int64_t(blockIdx.x) * kVecBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kVecBlock) {
    const double v = x[i];
    if (!isfinite(v)) ++bad;
    x[i] = -v;
  }
  if (bad) atomicAdd(nonfinite, bad);
}

// Partial sums of  y.g + y.r + |D y|^2  over [begin, end) (one partial per workgroup).  perm (CGNR on internally numbered points,
// solver.hip): y, g, r are in the internal order, D's point part comes from perm.D_e (internal) and -y goes out in the caller's order.
__global__ __launch_bounds__(kVecBlock) void cgnr_model_cost_kernel(const double* y, const double* g, const double* r, const double* D,
                                                                     int64_t begin, int64_t end, double* partials,
                                                                     double* neg_out, int* nonfinite, const int* gate, PointPerm perm) {
  __shared__ double sh[4];
  if (gate && !CgStatusAllowsSolution(*gate)) return;
  double v = 0;
  int bad = 0;
  for (int64_t i = begin + int64_t(blockIdx.x) * kVecBlock + threadIdx.x; i < end; i += int64_t(gridDim.x) * kVecBlock) {
    const bool pt = perm.pt_pos && i < perm.n_e;
    const double yi = y[i], d = D ? (pt ? perm.D_e[i] : D[i]) * yi : 0.0;
    v += yi * (g[i] + r[i]) + d * d;
    if (neg_out) { neg_out[pt ? perm.pt_pos[i / 3] + int(i % 3) : i] = -yi; if (!isfinite(yi)) ++bad; }
  }
  if (bad) atomicAdd(nonfinite, bad);
  v = block_sum(v, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = v;
}

// ---- CG ---------------------------------------------------------------------
__global__ __launch_bounds__(kVecBlock) void cg_rhs_norm_kernel(CgBuffers B) {
  __shared__ double sh[4];
  double v = 0;
  for (Span s = my_span(B); s.i < s.end; s.i += s.step) v += B.rhs[s.i] * B.rhs[s.i];
  v = block_sum(v, sh);
  if (threadIdx.x == 0) B.partials[blockIdx.x] = v;
}

__global__ __launch_bounds__(kVecBlock) void cg_init_kernel(CgBuffers B, double q_tol, double r_tol, int min_it, int max_it) {
  __shared__ double sh[4];
  const double nn = total_of(B, 0, sh);
  for (Span s = my_span(B); s.i < s.end; s.i += s.step) { B.x[s.i] = 0.0; B.r[s.i] = B.rhs[s.i]; }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    CgScalars& S = *B.S;
    S.norm_rhs = sqrt(nn);
    S.tol_r = r_tol * S.norm_rhs;
    S.q_tol = q_tol;
    S.rho = 1.0; S.rho_new = 1.0; S.beta = 0; S.pq = 0; S.alpha = 0;
    S.Q0 = 0.0;  // -x.(rhs + r) with x = 0
    S.Q0_pp[0] = 0.0; S.Q0_pp[1] = 0.0; S.rho_pp[0] = 1.0; S.rho_pp[1] = 1.0;
    S.Q1 = 0; S.zeta = 0; S.norm_r = S.norm_rhs; S.norm_p = 0; S.norm_q = 0;
    S.iter = 1; S.min_it = min_it; S.max_it = max_it;
    S.status = kCgRunning;
    S.fail_dir = 0; S.fail_step = 0;
    if (S.norm_rhs == 0.0) S.status = kCgZeroRhs;
    else if (min_it == 0 && S.norm_r <= S.tol_r) S.status = kCgInitialResidual;
    if (B.setup_fail && *B.setup_fail != 0) S.status = kCgSetupFailed;
  }
}

__global__ __launch_bounds__(kVecBlock) void cg_dot_slot_kernel(CgBuffers B, const double* x, const double* y, int slot) {
  __shared__ double sh[4];
  if (B.S->status != 0) return;
  double v = 0;
  for (Span s = my_span(B); s.i < s.end; s.i += s.step) v += x[s.i] * y[s.i];
  v = block_sum(v, sh);
  if (threadIdx.x == 0) B.partials[slot * kMaxVecGrid + blockIdx.x] = v;
}

// CG from a non-zero initial guess (:138-159 with x != 0): r = rhs - A x (tmp = A x), Q0 = -x.(rhs + r).
__global__ __launch_bounds__(kVecBlock) void cg_init_from_guess_kernel(CgBuffers B, const double* tmp) {
  __shared__ double sh[4];
  double q = 0, rr = 0;
  for (Span s = my_span(B); s.i < s.end; s.i += s.step) {
    const double r = B.rhs[s.i] - tmp[s.i];
    B.r[s.i] = r;
    q -= B.x[s.i] * (B.rhs[s.i] + r);
    rr += r * r;
  }
  q = block_sum(q, sh);
  rr = block_sum(rr, sh);
  if (threadIdx.x == 0) { B.partials[2 * kMaxVecGrid + blockIdx.x] = q; B.partials[3 * kMaxVecGrid + blockIdx.x] = rr; }
}
__global__ __launch_bounds__(kVecBlock) void cg_init_from_guess_finish_kernel(CgBuffers B, double q_tol, double r_tol, int min_it, int max_it) {
  __shared__ double sh[4];
  const double nn = total_of(B, 0, sh);
  const double Q0 = total_of(B, 2, sh);
  const double rr = total_of(B, 3, sh);
  if (threadIdx.x != 0) return;
  CgScalars& S = *B.S;
  S.norm_rhs = sqrt(nn);
  S.tol_r = r_tol * S.norm_rhs;
  S.q_tol = q_tol;
  S.rho = 1.0; S.rho_new = 1.0; S.beta = 0; S.pq = 0; S.alpha = 0;
  S.Q0 = Q0; S.Q1 = 0; S.zeta = 0; S.norm_r = sqrt(rr); S.norm_p = 0; S.norm_q = 0;
  S.Q0_pp[0] = Q0; S.Q0_pp[1] = Q0; S.rho_pp[0] = 1.0; S.rho_pp[1] = 1.0;
  S.iter = 1; S.min_it = min_it; S.max_it = max_it;
  S.status = kCgRunning;
  S.fail_dir = 0; S.fail_step = 0;
  if (S.norm_rhs == 0.0) S.status = kCgZeroRhs;  // the host zeroes x in that case
  else if (min_it == 0 && S.norm_r <= S.tol_r) S.status = kCgInitialResidual;
  if (B.setup_fail && *B.setup_fail != 0) S.status = kCgSetupFailed;
}

// One thread per COLUMN BLOCK (not per scalar): the block's metadata is read once, its
// n x n inverse and the n entries of r are contiguous reads, and 3- and 9-wide blocks (the
// bundle-adjustment case) are fully unrolled.  Workgroups [0, grid_e) take the shard's
// blocks [0, n_local_blocks), the others the replicated blocks.
template <int N>
__device__ __forceinline__ double precondition_block(const double* __restrict__ m, const double* __restrict__ r, double* __restrict__ z) {
  double rr[N], v = 0;
#pragma unroll
  for (int c = 0; c < N; ++c) rr[c] = r[c];
#pragma unroll
  for (int a = 0; a < N; ++a) {
    double s = 0;
#pragma unroll
    for (int c = 0; c < N; ++c) s += m[a * N + c] * rr[c];
    z[a] = s;
    v += rr[a] * s;
  }
  return v;
}

__global__ __launch_bounds__(kVecBlock) void cg_precondition_kernel(CgBuffers B, GenStructure G, int first_block, int col_begin,
                                                                    int nblocks, int n_local_blocks,
                                                                    const int64_t* diag_off, const double* blocks) {
  __shared__ double sh[4];
  if (B.S->status != 0) return;
  double v = 0;
  if (!blocks) {  // IDENTITY
    for (Span s = my_span(B); s.i < s.end; s.i += s.step) { const double r = B.r[s.i]; B.z[s.i] = r; v += r * r; }
  } else {
    int64_t q, q_end, q_step;
    if (int(blockIdx.x) < B.grid_e) {
      q = int64_t(blockIdx.x) * kVecBlock + threadIdx.x; q_end = n_local_blocks; q_step = int64_t(B.grid_e) * kVecBlock;
    } else {
      q = n_local_blocks + int64_t(blockIdx.x - B.grid_e) * kVecBlock + threadIdx.x; q_end = nblocks;
      q_step = int64_t(B.grid - B.grid_e) * kVecBlock;
    }
    for (; q < q_end; q += q_step) {
      const int j = first_block + int(q);
      const int n = G.csz[j];
      const int64_t pos = G.cpos[j] - col_begin;
      const double* m = blocks + (diag_off[q] - diag_off[0]);
      if (n == 3) v += precondition_block<3>(m, B.r + pos, B.z + pos);
      else if (n == 9) v += precondition_block<9>(m, B.r + pos, B.z + pos);
      else {
        for (int a = 0; a < n; ++a) {
          double s = 0;
          for (int c = 0; c < n; ++c) s += m[a * n + c] * B.r[pos + c];
          B.z[pos + a] = s;
          v += B.r[pos + a] * s;
        }
      }
    }
  }
  v = block_sum(v, sh);
  if (threadIdx.x == 0) B.partials[0 * kMaxVecGrid + blockIdx.x] = v;
}

__global__ __launch_bounds__(kVecBlock) void cg_direction_kernel(CgBuffers B) {
  __shared__ double sh[4];
  if (B.S->status != 0) return;
  const double rho = total_of(B, 0, sh);
  const double last_rho = B.S->rho;
  const int iter = B.S->iter;
  int fail = 0;
  double beta = 0.0;
  if (zero_or_inf(rho)) fail = kCgFailRho;
  else if (iter > 1) { beta = rho / last_rho; if (zero_or_inf(beta)) fail = kCgFailBeta; }
  if (!fail) {
    if (iter == 1) { for (Span s = my_span(B); s.i < s.end; s.i += s.step) B.p[s.i] = B.z[s.i]; }
    else { for (Span s = my_span(B); s.i < s.end; s.i += s.step) B.p[s.i] = B.z[s.i] + beta * B.p[s.i]; }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    B.S->rho_new = rho;
    B.S->rho_pp[iter & 1] = rho;  // where the fused iteration kernels look for rho_iter
    B.S->beta = beta;
    // The status word itself is written only by cg_finalize (one workgroup): a
    // workgroup of THIS launch that starts late must not see it change.  Failures are
    // staged in fail_dir / fail_step and committed there.
    B.S->fail_dir = fail;
  }
}

__global__ __launch_bounds__(kVecBlock) void cg_dot_pq_kernel(CgBuffers B) {
  __shared__ double sh[4];
  if (B.S->status != 0) return;
  if (B.S->fail_dir != 0) return;  // direction failed: cg_step forwards it, cg_finalize commits it
  double v = 0;
  for (Span s = my_span(B); s.i < s.end; s.i += s.step) v += B.p[s.i] * B.z[s.i];
  v = block_sum(v, sh);
  if (threadIdx.x == 0) B.partials[1 * kMaxVecGrid + blockIdx.x] = v;
}

__global__ __launch_bounds__(kVecBlock) void cg_step_kernel(CgBuffers B, int reset) {
  __shared__ double sh[4];
  if (B.S->status != 0) return;
  const int staged = B.S->fail_dir;
  const double pq = staged ? 1.0 : total_of(B, 1, sh);
  const double rho = B.S->rho_new;
  int fail = staged;
  double alpha = 0;
  if (!fail) {
    if (pq <= 0 || isinf(pq)) fail = kCgIndefinite;
    else { alpha = rho / pq; if (isinf(alpha)) fail = kCgFailAlpha; }
  }
  double q1 = 0, rr = 0;
  if (!fail) {
    if (reset) {
      for (Span s = my_span(B); s.i < s.end; s.i += s.step) B.x[s.i] += alpha * B.p[s.i];
    } else {
      for (Span s = my_span(B); s.i < s.end; s.i += s.step) {
        const double x = B.x[s.i] + alpha * B.p[s.i];
        const double r = B.r[s.i] - alpha * B.z[s.i];
        B.x[s.i] = x;
        B.r[s.i] = r;
        q1 -= x * (B.rhs[s.i] + r);
        rr += r * r;
      }
    }
  }
  if (!reset) {
    q1 = block_sum(q1, sh);
    rr = block_sum(rr, sh);
    if (threadIdx.x == 0) { B.partials[2 * kMaxVecGrid + blockIdx.x] = q1; B.partials[3 * kMaxVecGrid + blockIdx.x] = rr; }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    B.S->pq = pq;
    B.S->alpha = alpha;
    B.S->fail_step = fail;
  }
}

__global__ __launch_bounds__(kVecBlock) void cg_residual_reset_kernel(CgBuffers B, const double* tmp) {
  __shared__ double sh[4];
  if (B.S->status != 0) return;
  if (B.S->fail_step != 0) return;
  double q1 = 0, rr = 0;
  for (Span s = my_span(B); s.i < s.end; s.i += s.step) {
    const double r = B.rhs[s.i] - tmp[s.i];
    B.r[s.i] = r;
    q1 -= B.x[s.i] * (B.rhs[s.i] + r);
    rr += r * r;
  }
  q1 = block_sum(q1, sh);
  rr = block_sum(rr, sh);
  if (threadIdx.x == 0) { B.partials[2 * kMaxVecGrid + blockIdx.x] = q1; B.partials[3 * kMaxVecGrid + blockIdx.x] = rr; }
}

// One workgroup.  The only writer of the status word while iterating.
__global__ __launch_bounds__(kVecBlock) void cg_finalize_kernel(CgBuffers B) {
  __shared__ double sh[4];
  CgScalars& S = *B.S;
  if (S.status != 0) return;
  const int fail = S.fail_step;
  if (fail != 0) {
    if (threadIdx.x == 0) S.status = fail;
    return;
  }
  const double Q1 = total_of(B, 2, sh);
  const double rr = total_of(B, 3, sh);
  if (threadIdx.x != 0) return;
  const double norm_r = sqrt(rr);
  const double zeta = S.iter * (Q1 - S.Q0) / Q1;
  S.Q1 = Q1;
  S.zeta = zeta;
  S.norm_r = norm_r;
  S.rho = S.rho_new;
  if (zeta < S.q_tol && S.iter >= S.min_it) { S.status = kCgConvergedZeta; return; }
  S.Q0 = Q1;
  if (norm_r <= S.tol_r && S.iter >= S.min_it) { S.status = kCgConvergedResidual; return; }
  if (S.iter >= S.max_it) { S.status = kCgMaxIterations; return; }
  S.iter += 1;
}

// ---- fused iteration (two kernels after the operator instead of five) ---------------------------------
// Second half of iteration `it`: what cg_step_kernel does (pq, alpha, x += alpha p, r -= alpha q, partial Q1 and
// |r|^2) and, in the same pass over r, what cg_precondition_kernel does for the NEXT iteration (z = M^-1 r, partial
// r.z): a column block's r is in registers when its update is done.  q arrives in z and is replaced by M^-1 r.
template <int N, bool START>
__device__ __forceinline__ void update_block(const double* __restrict__ m, double alpha, const double* __restrict__ p,
                                             const double* __restrict__ rhs, double* __restrict__ x, double* __restrict__ r,
                                             double* __restrict__ z, double& q1, double& rr, double& rz) {
  double rn[N];
#pragma unroll
  for (int c = 0; c < N; ++c) {
    const double xv = START ? 0.0 : x[c] + alpha * p[c];
    const double rv = START ? rhs[c] : r[c] - alpha * z[c];
    x[c] = xv;
    r[c] = rv;
    rn[c] = rv;
    q1 -= xv * (rhs[c] + rv);
    rr += rv * rv;
  }
#pragma unroll
  for (int a = 0; a < N; ++a) {
    double t = 0;
#pragma unroll
    for (int c = 0; c < N; ++c) t += m[a * N + c] * rn[c];
    z[a] = t;
    rz += rn[a] * t;
  }
}

// Column blocks [nine_from, nblocks) are all 9 wide (the camera blocks): workgroups [grid_a, gridDim.x) take them with NINE
// LANES PER BLOCK (7 blocks per wavefront, lane a of a group owns row a: its 72-byte row of M and the block's 648 bytes
// are contiguous across the group, r travels by shuffle) instead of one thread per block walking 81 strided entries —
// with 1778 cameras that thread-per-block tail alone took 15 us.  Blocks [0, nine_from) stay one thread per block.
// START: the same pass as the first kernel of a solve with x0 = 0 (:130-167): x = 0, r = rhs, z = M^-1 r, partial |rhs|^2 -> slot 3
// and r.z -> slot 0; cg_begin_kernel turns those into the scalars and p = z.
template <bool START>
__global__ __launch_bounds__(kVecBlock) void cg_update_kernel(CgBuffers B, GenStructure G, int first_block, int col_begin, int nblocks,
                                                              const int64_t* diag_off, const double* blocks, int reset, int it,
                                                              int nine_from, int grid_a) {
  __shared__ double sh[12];
  CgScalars& S = *B.S;
  double pq = 0, rho = 0, alpha = 0;
  int fail = 0;
  if constexpr (!START) {
    if (S.status != 0) return;
    const int staged = S.fail_dir;  // the direction of this iteration failed: forwarded to cg_finalize_direction_kernel
    double v = 0;
    if (!staged) for (int k = threadIdx.x; k < B.n_pq; k += kVecBlock) v += B.pq_parts[k];
    pq = staged ? 1.0 : block_sum(v, sh) + (B.pq_extra ? *B.pq_extra : 0.0);
    rho = S.rho_pp[it & 1];
    fail = staged;
    if (!fail) {
      if (pq <= 0 || isinf(pq)) fail = kCgIndefinite;
      else { alpha = rho / pq; if (isinf(alpha)) fail = kCgFailAlpha; }
    }
  }
  double q1 = 0, rr = 0, rz = 0;
  if (!fail) {
    const int64_t t0 = int64_t(blockIdx.x) * kVecBlock + threadIdx.x, step = int64_t(gridDim.x) * kVecBlock;
    if (reset) {  // x only: r comes from the operator (cg_residual_reset_kernel), then cg_precondition_kernel
      for (int64_t i = t0; i < B.n; i += step) B.x[i] += alpha * B.p[i];
    } else if (!blocks) {  // IDENTITY: z = r   (my_span: a sharded run's partial sums must split at the shard boundary)
      const Span sp = my_span(B);
      for (int64_t i = sp.i; i < sp.end; i += sp.step) {
        const double xv = START ? 0.0 : B.x[i] + alpha * B.p[i];
        const double rv = START ? B.rhs[i] : B.r[i] - alpha * B.z[i];
        B.x[i] = xv; B.r[i] = rv; B.z[i] = rv;
        q1 -= xv * (B.rhs[i] + rv);
        rr += rv * rv;
      }
      rz = rr;
    } else if (int(blockIdx.x) >= grid_a) {
      const int lane = threadIdx.x & 63;
      const int g = lane / 9, a = lane - 9 * g;
      const int n9 = nblocks - nine_from;
      const int64_t wave = int64_t(int(blockIdx.x) - grid_a) * (kVecBlock / 64) + (threadIdx.x >> 6);
      const int64_t nwaves = int64_t(int(gridDim.x) - grid_a) * (kVecBlock / 64);
      for (int64_t base = wave * 7; base < n9; base += nwaves * 7) {
        const bool active = g < 7 && base + g < n9;
        const int64_t q = nine_from + (active ? base + g : base);
        const int64_t i = G.cpos[first_block + q] - col_begin + (active ? a : 0);
        const double* m = blocks + (diag_off[q] - diag_off[0]) + (active ? 9 * a : 0);
        double mr[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) mr[c] = m[c];
        const double xv = START ? 0.0 : B.x[i] + alpha * B.p[i];
        const double rv = START ? B.rhs[i] : B.r[i] - alpha * B.z[i];
        double t = 0;
#pragma unroll
        for (int c = 0; c < 9; ++c) t += mr[c] * __shfl(rv, 9 * (g < 7 ? g : 0) + c, 64);
        if (active) {
          B.x[i] = xv; B.r[i] = rv; B.z[i] = t;
          q1 -= xv * (B.rhs[i] + rv);
          rr += rv * rv;
          rz += rv * t;
        }
      }
    } else {
      const int64_t stepa = int64_t(grid_a) * kVecBlock;
      for (int64_t q = t0; q < nine_from; q += stepa) {
        const int j = first_block + int(q);
        const int n = G.csz[j];
        const int64_t pos = G.cpos[j] - col_begin;
        const double* m = blocks + (diag_off[q] - diag_off[0]);
        if (n == 3) update_block<3, START>(m, alpha, B.p + pos, B.rhs + pos, B.x + pos, B.r + pos, B.z + pos, q1, rr, rz);
        else if (n == 9) update_block<9, START>(m, alpha, B.p + pos, B.rhs + pos, B.x + pos, B.r + pos, B.z + pos, q1, rr, rz);
        else {
          double rn[kMaxGenericBlock];
          for (int c = 0; c < n; ++c) {
            const double xv = START ? 0.0 : B.x[pos + c] + alpha * B.p[pos + c];
            const double rv = START ? B.rhs[pos + c] : B.r[pos + c] - alpha * B.z[pos + c];
            B.x[pos + c] = xv; B.r[pos + c] = rv; rn[c] = rv;
            q1 -= xv * (B.rhs[pos + c] + rv);
            rr += rv * rv;
          }
          for (int a = 0; a < n; ++a) {
            double t = 0;
            for (int c = 0; c < n; ++c) t += m[a * n + c] * rn[c];
            B.z[pos + a] = t;
            rz += rn[a] * t;
          }
        }
      }
    }
  }
  if (!reset) {
    block_sum3(q1, rr, rz, sh);
    if (threadIdx.x == 0) {
      B.partials[2 * kMaxVecGrid + blockIdx.x] = q1;
      B.partials[3 * kMaxVecGrid + blockIdx.x] = rr;
      B.partials[0 * kMaxVecGrid + blockIdx.x] = rz;
    }
  }
  if (!START && blockIdx.x == 0 && threadIdx.x == 0) {
    S.pq = pq;
    S.alpha = alpha;
    S.rho_new = rho;
    S.fail_step = fail;
  }
}

// Second and last kernel of the start of a solve (x0 = 0): the scalars cg_init_kernel sets, from the partial sums cg_update_kernel<START>
// left (|rhs|^2 in slot 3, r.z in slot 0), and the direction of iteration 1, p = z (:175-177).  Same discipline as
// cg_finalize_direction_kernel: every workgroup derives the same decision, workgroup 0 records it.
__global__ __launch_bounds__(kVecBlock) void cg_begin_kernel(CgBuffers B, double q_tol, double r_tol, int min_it, int max_it) {
  __shared__ double sh[12];
  double nn = 0, rho = 0, unused = 0;
  for (int k = B.grid_e + threadIdx.x; k < B.grid; k += kVecBlock) { nn += B.partials[3 * kMaxVecGrid + k]; rho += B.partials[0 * kMaxVecGrid + k]; }
  block_sum3(nn, rho, unused, sh);
  if (B.grid_e > 0) { nn += B.comm[3]; rho += B.comm[0]; }  // the shard's share, summed over ranks (cg_collapse_kernel + all-reduce)
  const double norm_rhs = sqrt(nn);
  const double tol_r = r_tol * norm_rhs;
  int status = kCgRunning;
  if (norm_rhs == 0.0) status = kCgZeroRhs;
  else if (min_it == 0 && norm_rhs <= tol_r) status = kCgInitialResidual;
  if (B.setup_fail && *B.setup_fail != 0) status = kCgSetupFailed;
  int fail_dir = 0;
  if (status == kCgRunning) {
    if (zero_or_inf(rho)) fail_dir = kCgFailRho;
    else
      for (int64_t i = int64_t(blockIdx.x) * kVecBlock + threadIdx.x; i < B.n; i += int64_t(gridDim.x) * kVecBlock) B.p[i] = B.z[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    CgScalars& S = *B.S;
    S.norm_rhs = norm_rhs; S.tol_r = tol_r; S.q_tol = q_tol;
    S.rho = 1.0; S.rho_new = rho; S.beta = 0; S.pq = 0; S.alpha = 0;
    S.Q0 = 0.0; S.Q1 = 0; S.zeta = 0; S.norm_r = norm_rhs; S.norm_p = 0; S.norm_q = 0;
    S.Q0_pp[0] = 0.0; S.Q0_pp[1] = 0.0; S.rho_pp[0] = 1.0; S.rho_pp[1] = rho;
    S.iter = 1; S.min_it = min_it; S.max_it = max_it;
    S.fail_dir = fail_dir; S.fail_step = 0;
    S.status = status;
  }
}

// End of iteration `it`: cg_finalize_kernel's tests and, if CG continues, cg_direction_kernel for iteration it + 1.
// Every workgroup derives the same decision from the same partial sums; workgroup 0 records it.  Scalars this
// kernel both needs and produces (rho, Q0) are double-buffered by iteration parity, the iteration number is a launch
// argument, and a workgroup that starts after the status word went terminal just returns (CG is over, p is dead).
__global__ __launch_bounds__(kVecBlock) void cg_finalize_direction_kernel(CgBuffers B, int it) {
  __shared__ double sh[12];
  CgScalars& S = *B.S;
  if (S.status != 0) return;
  const int fail = S.fail_step;
  if (fail != 0) {
    if (blockIdx.x == 0 && threadIdx.x == 0) S.status = fail;
    return;
  }
  double Q1 = 0, rr = 0, rho = 0;
  for (int k = B.grid_e + threadIdx.x; k < B.grid; k += kVecBlock) {  // same order in every workgroup: all agree bit for bit
    Q1 += B.partials[2 * kMaxVecGrid + k];
    rr += B.partials[3 * kMaxVecGrid + k];
    rho += B.partials[0 * kMaxVecGrid + k];
  }
  block_sum3(Q1, rr, rho, sh);
  if (B.grid_e > 0) { Q1 += B.comm[2]; rr += B.comm[3]; rho += B.comm[0]; }  // sharded: + the all-reduced shard sums (identical on every rank)
  const double Q0 = S.Q0_pp[it & 1], rho_prev = S.rho_pp[it & 1];
  const double norm_r = sqrt(rr);
  const double zeta = it * (Q1 - Q0) / Q1;
  int status = kCgRunning;
  if (zeta < S.q_tol && it >= S.min_it) status = kCgConvergedZeta;
  else if (norm_r <= S.tol_r && it >= S.min_it) status = kCgConvergedResidual;
  else if (it >= S.max_it) status = kCgMaxIterations;
  int fail_dir = 0;
  double beta = 0.0;
  if (status == kCgRunning) {
    if (zero_or_inf(rho)) fail_dir = kCgFailRho;
    else { beta = rho / rho_prev; if (zero_or_inf(beta)) fail_dir = kCgFailBeta; }
    if (!fail_dir)
      for (int64_t i = int64_t(blockIdx.x) * kVecBlock + threadIdx.x; i < B.n; i += int64_t(gridDim.x) * kVecBlock)
        B.p[i] = B.z[i] + beta * B.p[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    S.Q1 = Q1; S.zeta = zeta; S.norm_r = norm_r;
    if (status == kCgConvergedZeta) { S.status = status; return; }   // Q0 keeps its value, as in the reference (:273-284)
    S.Q0 = Q1;
    S.Q0_pp[(it + 1) & 1] = Q1;
    if (status != kCgRunning) { S.status = status; return; }
    S.rho = rho_prev; S.rho_new = rho; S.beta = beta;
    S.rho_pp[(it + 1) & 1] = rho;
    S.fail_dir = fail_dir;
    S.iter = it + 1;
  }
}

__global__ __launch_bounds__(kVecBlock) void cg_collapse_kernel(CgBuffers B, int first_slot, int count) {
  __shared__ double sh[4];
  for (int k = 0; k < count; ++k) {
    const double* p = B.partials + (first_slot + k) * kMaxVecGrid;
    double v = 0;
    for (int b = threadIdx.x; b < B.grid_e; b += kVecBlock) v += p[b];
    v = block_sum(v, sh);
    if (threadIdx.x == 0) B.comm[first_slot + k] = v;
  }
}

// the same with the sum over ranks inside (p2p.h; slot index = k, one chunk: the numbering of the all-reduce of `count` doubles)
__global__ __launch_bounds__(kVecBlock) void cg_collapse_exchange_kernel(CgBuffers B, int first_slot, int count, P2pComm C) {
  __shared__ double sh[4];
  __shared__ double tot[64];
  for (int k = 0; k < count; ++k) {
    const double* p = B.partials + (first_slot + k) * kMaxVecGrid;
    double v = 0;
    for (int b = threadIdx.x; b < B.grid_e; b += kVecBlock) v += p[b];
    v = block_sum(v, sh);
    if (threadIdx.x == 0) tot[k] = v;
  }
  __syncthreads();
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  const double t = p2p_exchange_wave(C, 0, lane, lane < count, lane < count ? tot[lane] : 0.0);
  if (lane < count) B.comm[first_slot + lane] = t;
}

inline int vec_grid(int64_t n) {
  int64_t g = (n + kVecBlock * 4 - 1) / (kVecBlock * 4);
  if (g < 1) g = 1;
  if (g > kMaxVecGrid) g = kMaxVecGrid;
  return int(g);
}

}  // namespace

hipError_t LaunchSet(double* x, double v, int64_t n, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(set_kernel, dim3(vec_grid(n)), dim3(kVecBlock), 0, s, x, v, n);
  return hipGetLastError();
}
hipError_t LaunchAxpby(double a, const double* x, double b, const double* y, double* z, int64_t n, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(axpby_kernel, dim3(vec_grid(n)), dim3(kVecBlock), 0, s, a, x, b, y, z, n);
  return hipGetLastError();
}
hipError_t LaunchSquareScale(const double* D, const double* x, double* y, int64_t n, const int* status, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL((square_scale_kernel<false>), dim3(vec_grid(n)), dim3(kVecBlock), 0, s, D, x, y, n, status);
  return hipGetLastError();
}
hipError_t LaunchAddSquareScale(const double* D, const double* x, double* y, int64_t n, const int* status, hipStream_t s) {
  if (n > 0 && D) hipLaunchKernelGGL((square_scale_kernel<true>), dim3(vec_grid(n)), dim3(kVecBlock), 0, s, D, x, y, n, status);
  return hipGetLastError();
}
hipError_t LaunchDot(const double* x, const double* y, int64_t n, double* partials, double* out, hipStream_t s) {
  const int g = vec_grid(n);
  hipLaunchKernelGGL(dot_partial_kernel, dim3(g), dim3(kVecBlock), 0, s, x, y, n, partials);
  hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(kVecBlock), 0, s, partials, g, out);
  return hipGetLastError();
}
hipError_t LaunchCgnrModelCost(const double* y, const double* g, const double* r, const double* D, int64_t begin, int64_t end,
                               double* partials, int* nparts, hipStream_t s, double* neg_out, int* nonfinite, const int* gate,
                               const PointPerm& perm) {
  const int grid = vec_grid(end - begin);
  *nparts = grid;
  if (end > begin) hipLaunchKernelGGL(cgnr_model_cost_kernel, dim3(grid), dim3(kVecBlock), 0, s, y, g, r, D, begin, end, partials,
                                      nonfinite ? neg_out : nullptr, nonfinite, gate, perm);
  return hipGetLastError();
}
// out (caller's point order) <- in (internal order), or the other way round, over the 3-wide point blocks [0, n_e); [n_e, n) is copied
__global__ __launch_bounds__(kVecBlock) void permute_points_kernel(const double* in, double* out, const int32_t* pt_pos, int64_t n_e, int64_t n, int to_internal) {
  for (int64_t i = int64_t(blockIdx.x) * kVecBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kVecBlock) {
    const int64_t o = i < n_e ? pt_pos[i / 3] + int(i % 3) : i;
    if (to_internal) out[i] = in[o];
    else out[o] = in[i];
  }
}
hipError_t LaunchPermutePoints(const double* in, double* out, const int32_t* pt_pos, int64_t n_e, int64_t n, bool to_internal, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(permute_points_kernel, dim3(vec_grid(n)), dim3(kVecBlock), 0, s, in, out, pt_pos, n_e, n, to_internal ? 1 : 0);
  return hipGetLastError();
}
hipError_t LaunchLmDiagonal(double* diag, double lo, double hi, double radius, double* D, int64_t n, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(lm_diagonal_kernel, dim3(vec_grid(n)), dim3(kVecBlock), 0, s, diag, lo, hi, radius, D, n);
  return hipGetLastError();
}
hipError_t LaunchNegateAndCheck(double* x, int64_t n, int* nonfinite, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(negate_and_check_kernel, dim3(vec_grid(n)), dim3(kVecBlock), 0, s, x, n, nonfinite);
  return hipGetLastError();
}
hipError_t LaunchExpandSym(const double* packed, int ne, int pitch, double* dense, const int64_t* off, int n, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(expand_sym_kernel, dim3((n + 255) / 256), dim3(256), 0, s, packed, ne, pitch, dense, off, n);
  return hipGetLastError();
}

hipError_t LaunchScatterBlocks9(const double* in, double* out, const int64_t* off, int n, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(scatter_blocks9_kernel, dim3(unsigned((int64_t(9) * n + 255) / 256)), dim3(256), 0, s, in, out, off, n);
  return hipGetLastError();
}

hipError_t LaunchCgRhsNorm(const CgBuffers& B, hipStream_t s) {
  hipLaunchKernelGGL(cg_rhs_norm_kernel, dim3(B.grid), dim3(kVecBlock), 0, s, B);
  return hipGetLastError();
}
hipError_t LaunchCgInit(const CgBuffers& B, double q_tol, double r_tol, int min_it, int max_it, hipStream_t s) {
  hipLaunchKernelGGL(cg_init_kernel, dim3(B.grid), dim3(kVecBlock), 0, s, B, q_tol, r_tol, min_it, max_it);
  return hipGetLastError();
}
hipError_t LaunchCgPrecondition(const CgBuffers& B, const GenStructure& G, int first_block, int col_begin, int nblocks,
                                int n_local_blocks, const int64_t* diag_off, const double* blocks, hipStream_t s) {
  hipLaunchKernelGGL(cg_precondition_kernel, dim3(B.grid), dim3(kVecBlock), 0, s, B, G, first_block, col_begin, nblocks,
                     n_local_blocks, diag_off, blocks);
  return hipGetLastError();
}
hipError_t LaunchCgDotSlot(const CgBuffers& B, const double* x, const double* y, int slot, hipStream_t s) {
  hipLaunchKernelGGL(cg_dot_slot_kernel, dim3(B.grid), dim3(kVecBlock), 0, s, B, x, y, slot);
  return hipGetLastError();
}
hipError_t LaunchCgInitFromGuess(const CgBuffers& B, const double* tmp, double q_tol, double r_tol, int min_it, int max_it,
                                 hipStream_t s) {
  hipLaunchKernelGGL(cg_init_from_guess_kernel, dim3(B.grid), dim3(kVecBlock), 0, s, B, tmp);
  hipLaunchKernelGGL(cg_init_from_guess_finish_kernel, dim3(1), dim3(kVecBlock), 0, s, B, q_tol, r_tol, min_it, max_it);
  return hipGetLastError();
}
hipError_t LaunchCgDirection(const CgBuffers& B, hipStream_t s) {
  hipLaunchKernelGGL(cg_direction_kernel, dim3(B.grid), dim3(kVecBlock), 0, s, B);
  return hipGetLastError();
}
hipError_t LaunchCgDotPq(const CgBuffers& B, hipStream_t s) {
  hipLaunchKernelGGL(cg_dot_pq_kernel, dim3(B.grid), dim3(kVecBlock), 0, s, B);
  return hipGetLastError();
}
hipError_t LaunchCgStep(const CgBuffers& B, int reset, hipStream_t s) {
  hipLaunchKernelGGL(cg_step_kernel, dim3(B.grid), dim3(kVecBlock), 0, s, B, reset);
  return hipGetLastError();
}
hipError_t LaunchCgResidualReset(const CgBuffers& B, const double* tmp, hipStream_t s) {
  hipLaunchKernelGGL(cg_residual_reset_kernel, dim3(B.grid), dim3(kVecBlock), 0, s, B, tmp);
  return hipGetLastError();
}
hipError_t LaunchCgFinalize(const CgBuffers& B, hipStream_t s) {
  hipLaunchKernelGGL(cg_finalize_kernel, dim3(1), dim3(kVecBlock), 0, s, B);
  return hipGetLastError();
}
hipError_t LaunchCgUpdate(const CgBuffers& B, const GenStructure& G, int first_block, int col_begin, int nblocks,
                          const int64_t* diag_off, const double* blocks, int reset, int it, int nine_from, hipStream_t s) {
  nine_from = std::max(0, std::min(nine_from, nblocks));
  const int n9 = nblocks - nine_from;
  int grid_a = B.grid;  // workgroups of the one-thread-per-block part
  if (n9 > 0) grid_a = nine_from == 0 ? 0 : std::max(1, B.grid - std::max(1, std::min(B.grid / 2, (n9 + 111) / 112)));
  if (n9 > 0 && grid_a >= B.grid) { nine_from = nblocks; grid_a = B.grid; }  // a single workgroup: no room to split
  if (B.grid_e > 0) grid_a = B.grid_e;  // sharded: the shard's blocks [0, nine_from) are exactly what workgroups [0, grid_e) own
  if (it == 0)  // start of a solve
    hipLaunchKernelGGL((cg_update_kernel<true>), dim3(B.grid), dim3(kVecBlock), 0, s, B, G, first_block, col_begin, nblocks, diag_off, blocks, 0, 0,
                       nine_from, grid_a);
  else
    hipLaunchKernelGGL((cg_update_kernel<false>), dim3(B.grid), dim3(kVecBlock), 0, s, B, G, first_block, col_begin, nblocks, diag_off, blocks, reset, it,
                       nine_from, grid_a);
  return hipGetLastError();
}
hipError_t LaunchCgBegin(const CgBuffers& B, double q_tol, double r_tol, int min_it, int max_it, hipStream_t s) {
  hipLaunchKernelGGL(cg_begin_kernel, dim3(B.grid), dim3(kVecBlock), 0, s, B, q_tol, r_tol, min_it, max_it);
  return hipGetLastError();
}
hipError_t LaunchCgFinalizeDirection(const CgBuffers& B, int it, hipStream_t s) {
  hipLaunchKernelGGL(cg_finalize_direction_kernel, dim3(B.grid), dim3(kVecBlock), 0, s, B, it);
  return hipGetLastError();
}
hipError_t LaunchCgCollapse(const CgBuffers& B, int first_slot, int count, hipStream_t s) {
  hipLaunchKernelGGL(cg_collapse_kernel, dim3(1), dim3(kVecBlock), 0, s, B, first_slot, count);
  return hipGetLastError();
}

}  // namespace chip

namespace chip {
namespace {
// out[0] = (*flag != 0), out[1] = sum(parts[0 .. n)) in a fixed order: what a sharded LM step all-reduces at its end
// (finite-step flag, this rank's share of the model cost change) without a host round trip in front of the collective.
__global__ __launch_bounds__(kVecBlock) void collect_scalars_kernel(const int* flag, const double* parts, int n, double* out) {
  __shared__ double sh[4];
  double v = 0;
  for (int k = threadIdx.x; k < n; k += kVecBlock) v += parts[k];
  v = block_sum(v, sh);
  if (threadIdx.x == 0) { out[0] = (*flag != 0) ? 1.0 : 0.0; out[1] = v; }
}
}  // namespace
namespace {
// The same with the sum over ranks inside (p2p.h): out = sum over ranks of {flag != 0, sum(parts)}.
// image != nullptr: the kernel is also the read-back of the poll that follows (mailbox_kernel's job: image[0 .. n_image) to mapped host
// memory, then the stamp) — the sharded speculative tail ends in ONE launch instead of two.
__global__ __launch_bounds__(kVecBlock) void collect_scalars_exchange_kernel(const int* flag, const double* parts, int n, double* out, P2pComm C,
                                                                             const double* image, int n_image, double* host_dst,
                                                                             unsigned long long* host_stamp, unsigned long long stamp) {
  __shared__ double sh[4];
  double v = 0;
  for (int k = threadIdx.x; k < n; k += kVecBlock) v += parts[k];
  v = block_sum(v, sh);
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const double mine = lane == 0 ? ((*flag != 0) ? 1.0 : 0.0) : v;
    const double t = p2p_exchange_wave(C, 0, lane, lane < 2, mine);
    if (lane < 2) out[lane] = t;
  }
  if (image == nullptr) return;
  __syncthreads();   // the sums are in the image
  for (int i = threadIdx.x; i < n_image; i += kVecBlock) host_dst[i] = image[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(host_stamp, stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace
hipError_t LaunchCollectScalarsExchange(const int* flag, const double* parts, int n, double* out, const P2pComm& comm, hipStream_t s,
                                        const double* image, int n_image, double* host_dst, unsigned long long* host_stamp, unsigned long long stamp) {
  hipLaunchKernelGGL(collect_scalars_exchange_kernel, dim3(1), dim3(kVecBlock), 0, s, flag, parts, n, out, comm, image, n_image, host_dst, host_stamp, stamp);
  return hipGetLastError();
}
hipError_t LaunchCgCollapseExchange(const CgBuffers& B, int first_slot, int count, const P2pComm& comm, hipStream_t s) {
  hipLaunchKernelGGL(cg_collapse_exchange_kernel, dim3(1), dim3(kVecBlock), 0, s, B, first_slot, count, comm);
  return hipGetLastError();
}
hipError_t LaunchCollectScalars(const int* flag, const double* parts, int n, double* out, hipStream_t s) {
  hipLaunchKernelGGL(collect_scalars_kernel, dim3(1), dim3(kVecBlock), 0, s, flag, parts, n, out);
  return hipGetLastError();
}
}  // namespace chip

// ---- one-shot peer-to-peer all-reduce (the <= few-MB camera-space sums of a sharded solve) ----
// RCCL's small-message all-reduce costs tens of microseconds; the vectors summed here (9 or 81 doubles per camera) are latency-bound,
// and a step issues several.  The protocol is p2p.h's (round 6): a chunk = 64 consecutive elements = one wavefront, slot index =
// element index, flag index = chunk index — the SAME numbering the producers' own exchanges use for a vector (bal_reduce_exchange_kernel),
// so a rank that has to take this stand-alone kernel for a sum (rows outside its tiles, a layout its fused kernels do not take) and a
// rank that exchanges inside its reduction still meet in the same slots.  A wavefront takes K chunks per round trip (K = 8 for long
// vectors: the step's merged sum is 99 doubles per camera), a workgroup four wavefronts, every kP2pMaxGrid-th group of them.
namespace chip {
namespace {
template <int K>
__global__ __launch_bounds__(256) void p2p_allreduce_kernel(const double* __restrict__ in, double* __restrict__ out, int64_t n, P2pComm C) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t n_rounds = (n + 64 * K - 1) / (64 * K);
  for (int64_t r = int64_t(blockIdx.x) * 4 + wv; r < n_rounds; r += int64_t(gridDim.x) * 4) {
    const int64_t i0 = r * 64 * K;
    double v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { const int64_t i = i0 + 64 * k + lane; v[k] = i < n ? in[i] : 0.0; }
    p2p_exchange_wave_multi<K>(C, int(r * K), i0, 64, 64, n, v);
#pragma unroll
    for (int k = 0; k < K; ++k) { const int64_t i = i0 + 64 * k + lane; if (i < n) out[i] = v[k]; }
  }
}
}  // namespace

hipError_t LaunchP2pAllReduce(const double* in, double* out, int64_t n, const P2pComm& comm, int grid_cap, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  const int cap = std::max(1, std::min(kP2pMaxGrid, grid_cap));
  if (n > 64 * 1024) {
    const int64_t rounds = (n + 64 * 8 - 1) / (64 * 8);
    hipLaunchKernelGGL((p2p_allreduce_kernel<8>), dim3(int(std::min<int64_t>((rounds + 3) / 4, cap))), dim3(256), 0, stream, in, out, n, comm);
  } else {
    const int64_t rounds = (n + 63) / 64;
    hipLaunchKernelGGL((p2p_allreduce_kernel<1>), dim3(int(std::min<int64_t>((rounds + 3) / 4, cap))), dim3(256), 0, stream, in, out, n, comm);
  }
  return hipGetLastError();
}

// Read-back without a copy command and without hipStreamSynchronize: the image goes to MAPPED host memory by plain stores, followed by a
// stamp (system-scope release) the host spins on — see wait_mailbox, solver.hip.
namespace {
__global__ __launch_bounds__(256) void mailbox_kernel(const double* __restrict__ src, int n, double* __restrict__ host_dst,
                                                      unsigned long long* host_stamp, unsigned long long stamp) {
  for (int i = threadIdx.x; i < n; i += 256) host_dst[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(host_stamp, stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace
hipError_t LaunchMailbox(const double* src, int n, double* host_dst, unsigned long long* host_stamp, unsigned long long stamp, hipStream_t stream) {
  hipLaunchKernelGGL(mailbox_kernel, dim3(1), dim3(256), 0, stream, src, n, host_dst, host_stamp, stamp);
  return hipGetLastError();
}
}  // namespace chip
