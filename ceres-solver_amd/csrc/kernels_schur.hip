// kernels_schur.hip — explicit Schur complement solvers, any block sizes up to kMaxGenericBlock.
//
//   SparseSchurComplementSolver with use_explicit_schur_complement (ITERATIVE_SCHUR on a stored S):
//     S lives in the block-sparse storage of BlockRandomAccessSparseMatrix (I/block_random_access_sparse_matrix.cc:51-110),
//     the block pairs SparseSchurComplementSolver::InitStorage derives from the chunks (I/schur_complement_solver.cc:224-290);
//     SchurEliminator::Eliminate (I/schur_eliminator_impl.h:184-311) fills it as a GATHER: one wavefront per stored block walks the
//     (chunk, cell, cell) contributions plan.cc listed for it — no mutex per cell as in the reference, no atomics, bit-reproducible;
//     CG applies it with SymmetricRightMultiplyAndAccumulate (:125-163): each stored (upper-triangle) block serves y_i += M x_j and
//     y_j += M^T x_i, here one thread per output scalar through the row list and the transposed list of its block.
//   DenseSchurComplementSolver (DENSE_SCHUR, I/schur_complement_solver.cc:163-222): dense S, Cholesky factorisation of its
//     upper triangle (DenseCholesky, I/dense_cholesky.cc), two triangular solves — a right-looking blocked factorisation with
//     32-wide panels (diagonal block in LDS, panel solve one thread per row, LDS-tiled symmetric rank-32 update).
#include <hip/hip_runtime.h>

#include "device.h"

namespace chip {
namespace {

constexpr int kB = 256;

__device__ __forceinline__ int find_pair(const int64_t* off, int n, int64_t e) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

// ---- Eliminate into the block-sparse lhs -------------------------------------------------------------------------
// One wavefront per stored block (i, j), i <= j; lane l owns entries l, l + 64, ... (a, b) = (entry / n_j, entry % n_j).
// Contribution of a triple (chunk e, cell k1 of block i in row r1, cell k2 of block j in row r2):
//     [r1 == r2] F1^T F2  -  (E_r1^T F1)^T (E^T E + D_e^2)^-1 (E_r2^T F2)
// and for an E-free row just F1^T F2.  D_f^2 joins the diagonal of (i, i).
__global__ __launch_bounds__(kB) void schur_sparse_eliminate_kernel(GenStructure G, SchurPairs P, const double* __restrict__ v,
                                                                    const double* __restrict__ ete_inv, const double* __restrict__ D,
                                                                    double* __restrict__ S) {
  const int pair = blockIdx.x * (kB / 64) + (threadIdx.x >> 6);
  if (pair >= P.npairs) return;
  const int lane = threadIdx.x & 63;
  const int bi = P.pair_i[pair], bj = P.pair_j[pair];
  const int ji = G.nelim + bi, jj = G.nelim + bj;
  const int n1 = G.csz[ji], n2 = G.csz[jj];
  const int nent = n1 * n2;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};  // nent <= 256
  for (int64_t t = P.trip_ptr[pair]; t < P.trip_ptr[pair + 1]; ++t) {
    const int e = P.trip_e[t], k1 = P.trip_k1[t], k2 = P.trip_k2[t];
    const int r1 = P.cell_row[k1], r2 = P.cell_row[k2];
    const int rs1 = G.rsz[r1], rs2 = G.rsz[r2];
    const double* f1 = v + G.cval[k1];
    const double* f2 = v + G.cval[k2];
    const double* E1 = nullptr;
    const double* E2 = nullptr;
    const double* inv = nullptr;
    int es = 0;
    if (e >= 0) {
      es = G.csz[e];
      E1 = v + G.cval[G.rptr[r1]];
      E2 = v + G.cval[G.rptr[r2]];
      inv = ete_inv + G.diag_off_e[e];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ent = lane + 64 * q;
      if (ent >= nent) break;
      const int a = ent / n2, b = ent - a * n2;
      double s = 0.0;
      if (r1 == r2)
        for (int r = 0; r < rs1; ++r) s += f1[r * n1 + a] * f2[r * n2 + b];
      if (e >= 0) {
        // u1 = E_r1^T F1[:, a], u2 = E_r2^T F2[:, b]; es <= 16
        double w = 0.0;
        for (int p = 0; p < es; ++p) {
          double u1 = 0.0;
          for (int r = 0; r < rs1; ++r) u1 += E1[r * es + p] * f1[r * n1 + a];
          double iv = 0.0;
          for (int c = 0; c < es; ++c) {
            double u2 = 0.0;
            for (int r = 0; r < rs2; ++r) u2 += E2[r * es + c] * f2[r * n2 + b];
            iv += inv[p * es + c] * u2;
          }
          w += u1 * iv;
        }
        s -= w;
      }
      acc[q] += s;
    }
  }
  double* out = S + P.pair_off[pair];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int ent = lane + 64 * q;
    if (ent >= nent) break;
    double s = acc[q];
    if (D && bi == bj) {
      const int a = ent / n2, b = ent - a * n2;
      if (a == b) { const double d = D[G.cpos[ji] + a]; s += d * d; }
    }
    out[ent] = s;
  }
}

// y = S x with only the upper block triangle stored: one thread per scalar of y.
__global__ __launch_bounds__(kB) void schur_sparse_symv_kernel(GenStructure G, SchurPairs P, const double* __restrict__ S,
                                                               const double* __restrict__ x, double* __restrict__ y,
                                                               const int* __restrict__ status, int accumulate) {
  if (status && *status != 0) return;
  const int g = blockIdx.x * kB + threadIdx.x;
  if (g >= G.ncf) return;
  const int jb = G.col_block_of[G.nce + g];
  const int i = jb - G.nelim;
  const int a = G.nce + g - G.cpos[jb];
  const int ni = G.csz[jb];
  double s = 0.0;
  for (int p = P.row_ptr[i]; p < P.row_ptr[i + 1]; ++p) {  // cells (i, j >= i): row a of M
    const int jc = G.nelim + P.pair_j[p];
    const int nj = G.csz[jc];
    const double* M = S + P.pair_off[p] + int64_t(a) * nj;
    const double* xj = x + (G.cpos[jc] - G.nce);
    for (int b = 0; b < nj; ++b) s += M[b] * xj[b];
  }
  for (int q = P.col_ptr[i]; q < P.col_ptr[i + 1]; ++q) {  // cells (j < i, i): column a of M, i.e. row a of M^T
    const int p = P.col_pair[q];
    const int jc = G.nelim + P.pair_i[p];
    const int nj = G.csz[jc];
    const double* M = S + P.pair_off[p] + a;
    const double* xj = x + (G.cpos[jc] - G.nce);
    for (int b = 0; b < nj; ++b) s += M[int64_t(b) * ni] * xj[b];
  }
  if (accumulate) y[g] += s; else y[g] = s;
}

// blocks[i] = S(i, i): the first cell of block row i
__global__ __launch_bounds__(kB) void schur_sparse_diag_kernel(GenStructure G, SchurPairs P, const double* __restrict__ S,
                                                               const int64_t* __restrict__ diag_off_f, double* __restrict__ blocks) {
  const int nf = G.ncb - G.nelim;
  const int64_t e = int64_t(blockIdx.x) * kB + threadIdx.x;
  if (e >= diag_off_f[nf]) return;
  const int q = find_pair(diag_off_f, nf, e);
  blocks[e] = S[P.pair_off[P.row_ptr[q]] + (e - diag_off_f[q])];
}

// ---- dense Cholesky (upper triangle authoritative, result L in the LOWER triangle, row-major n x n) --------------
constexpr int kNb = 32;

// Mirror the stored upper triangle into the lower one (the factorisation below works on the lower triangle).
__global__ __launch_bounds__(kB) void dense_mirror_upper_kernel(double* __restrict__ A, int n) {
  const int64_t e = int64_t(blockIdx.x) * kB + threadIdx.x;
  if (e >= int64_t(n) * n) return;
  const int r = int(e / n), c = int(e % n);
  if (r > c) A[e] = A[int64_t(c) * n + r];
}

// Factor the diagonal block A[k0 : k0 + nb, k0 : k0 + nb] in LDS (one workgroup); fail_flag on a non-positive pivot.
__global__ __launch_bounds__(kNb * kNb) void dense_potrf_diag_kernel(double* __restrict__ A, int n, int k0, int nb, int* fail_flag) {
  __shared__ double T[kNb][kNb + 1];
  const int r = threadIdx.x / kNb, c = threadIdx.x % kNb;
  T[r][c] = (r < nb && c < nb) ? A[int64_t(k0 + r) * n + (k0 + c)] : (r == c ? 1.0 : 0.0);
  __syncthreads();
  for (int j = 0; j < nb; ++j) {
    if (threadIdx.x == 0) {
      double d = T[j][j];
      if (!(d > 0.0)) { atomicExch(fail_flag, 1); d = 1.0; }
      T[j][j] = sqrt(d);
    }
    __syncthreads();
    if (c == j && r > j && r < nb) T[r][j] /= T[j][j];
    __syncthreads();
    if (r > j && c > j && c <= r && r < nb) T[r][c] -= T[r][j] * T[c][j];
    __syncthreads();
  }
  if (r < nb && c < nb && c <= r) A[int64_t(k0 + r) * n + (k0 + c)] = T[r][c];
}

// Panel solve: rows i > k0 + nb - 1: A[i, k0 : k0 + nb] <- A[i, k0 : k0 + nb] L_kk^-T (one thread per row, L_kk in LDS)
__global__ __launch_bounds__(kB) void dense_trsm_panel_kernel(double* __restrict__ A, int n, int k0, int nb) {
  __shared__ double L[kNb][kNb + 1];
  for (int t = threadIdx.x; t < kNb * kNb; t += kB) {
    const int r = t / kNb, c = t % kNb;
    L[r][c] = (r < nb && c <= r) ? A[int64_t(k0 + r) * n + (k0 + c)] : 0.0;
  }
  __syncthreads();
  const int i = k0 + nb + blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
  double* row = A + int64_t(i) * n + k0;
  double x[kNb];
#pragma unroll
  for (int c = 0; c < kNb; ++c) x[c] = c < nb ? row[c] : 0.0;
#pragma unroll
  for (int c = 0; c < kNb; ++c) {
    if (c < nb) {
      double s = x[c];
#pragma unroll
      for (int p = 0; p < kNb; ++p) if (p < c) s -= x[p] * L[c][p];
      x[c] = s / L[c][c];
    }
  }
#pragma unroll
  for (int c = 0; c < kNb; ++c) if (c < nb) row[c] = x[c];
}

// Trailing update of the LOWER triangle: A[i, j] -= sum_p P[i, p] P[j, p], P = A[:, k0 : k0 + nb], i >= j >= k0 + nb.
// 32 x 32 output tiles, panel rows staged in LDS.
__global__ __launch_bounds__(kB) void dense_syrk_kernel(double* __restrict__ A, int n, int k0, int nb) {
  const int first = k0 + nb;
  const int ti = blockIdx.y, tj = blockIdx.x;
  if (tj > ti) return;
  __shared__ double Pi[kNb][kNb + 1], Pj[kNb][kNb + 1];
  const int i0 = first + ti * kNb, j0 = first + tj * kNb;
  for (int t = threadIdx.x; t < kNb * kNb; t += kB) {
    const int r = t / kNb, c = t % kNb;
    Pi[r][c] = (i0 + r < n && c < nb) ? A[int64_t(i0 + r) * n + (k0 + c)] : 0.0;
    Pj[r][c] = (j0 + r < n && c < nb) ? A[int64_t(j0 + r) * n + (k0 + c)] : 0.0;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < kNb * kNb; t += kB) {
    const int r = t / kNb, c = t % kNb;
    const int i = i0 + r, j = j0 + c;
    if (i >= n || j >= n || j > i) continue;
    double s = 0.0;
#pragma unroll
    for (int p = 0; p < kNb; ++p) s += Pi[r][p] * Pj[c][p];
    A[int64_t(i) * n + j] -= s;
  }
}

// Triangular solves with the factor, one right-hand side, blocked by kNb: x <- L^-1 x then x <- L^-T x.
__global__ __launch_bounds__(kNb) void dense_trsv_diag_kernel(const double* __restrict__ A, int n, int k0, int nb, double* __restrict__ x, int transposed) {
  __shared__ double xs[kNb];
  const int t = threadIdx.x;
  if (t < nb) xs[t] = x[k0 + t];
  __syncthreads();
  if (t == 0) {
    if (!transposed) {
      for (int r = 0; r < nb; ++r) {
        double s = xs[r];
        for (int c = 0; c < r; ++c) s -= A[int64_t(k0 + r) * n + (k0 + c)] * xs[c];
        xs[r] = s / A[int64_t(k0 + r) * n + (k0 + r)];
      }
    } else {
      for (int r = nb - 1; r >= 0; --r) {
        double s = xs[r];
        for (int c = r + 1; c < nb; ++c) s -= A[int64_t(k0 + c) * n + (k0 + r)] * xs[c];
        xs[r] = s / A[int64_t(k0 + r) * n + (k0 + r)];
      }
    }
  }
  __syncthreads();
  if (t < nb) x[k0 + t] = xs[t];
}
// forward: x[i] -= sum_c L[i, k0 + c] x[k0 + c] for i >= k0 + nb;  backward (transposed): x[i] -= sum_c L[k0 + c, i] x[k0 + c] for i < k0
__global__ __launch_bounds__(kB) void dense_trsv_update_kernel(const double* __restrict__ A, int n, int k0, int nb, double* __restrict__ x, int transposed) {
  __shared__ double xs[kNb];
  if (threadIdx.x < kNb) xs[threadIdx.x] = int(threadIdx.x) < nb ? x[k0 + threadIdx.x] : 0.0;
  __syncthreads();
  const int i = (transposed ? 0 : k0 + nb) + blockIdx.x * kB + threadIdx.x;
  if (transposed ? i >= k0 : i >= n) return;
  double s = 0.0;
  if (!transposed) { for (int c = 0; c < nb; ++c) s += A[int64_t(i) * n + (k0 + c)] * xs[c]; }
  else { for (int c = 0; c < nb; ++c) s += A[int64_t(k0 + c) * n + i] * xs[c]; }
  x[i] -= s;
}

inline unsigned blocks_for(int64_t n) { return unsigned((n + kB - 1) / kB); }

}  // namespace

hipError_t LaunchSchurSparseEliminate(const GenStructure& G, const SchurPairs& P, const double* values, const double* ete_inv,
                                      const double* D, double* S, hipStream_t s) {
  if (P.npairs > 0) hipLaunchKernelGGL(schur_sparse_eliminate_kernel, dim3((P.npairs + kB / 64 - 1) / (kB / 64)), dim3(kB), 0, s, G, P, values, ete_inv, D, S);
  return hipGetLastError();
}
hipError_t LaunchSchurSparseSymv(const GenStructure& G, const SchurPairs& P, const double* S, const double* x, double* y,
                                 const int* status, int accumulate, hipStream_t s) {
  if (G.ncf > 0) hipLaunchKernelGGL(schur_sparse_symv_kernel, dim3(blocks_for(G.ncf)), dim3(kB), 0, s, G, P, S, x, y, status, accumulate);
  return hipGetLastError();
}
hipError_t LaunchSchurSparseDiag(const GenStructure& G, const SchurPairs& P, const double* S, const int64_t* diag_off_f, double* blocks,
                                 int64_t total, hipStream_t s) {
  if (total > 0) hipLaunchKernelGGL(schur_sparse_diag_kernel, dim3(blocks_for(total)), dim3(kB), 0, s, G, P, S, diag_off_f, blocks);
  return hipGetLastError();
}

// In-place Cholesky of the symmetric n x n matrix whose UPPER triangle is authoritative; L is left in the lower triangle.
hipError_t LaunchDenseCholesky(double* A, int n, int* fail_flag, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(dense_mirror_upper_kernel, dim3(blocks_for(int64_t(n) * n)), dim3(kB), 0, s, A, n);
  for (int k0 = 0; k0 < n; k0 += kNb) {
    const int nb = n - k0 < kNb ? n - k0 : kNb;
    hipLaunchKernelGGL(dense_potrf_diag_kernel, dim3(1), dim3(kNb * kNb), 0, s, A, n, k0, nb, fail_flag);
    const int rest = n - k0 - nb;
    if (rest <= 0) break;
    hipLaunchKernelGGL(dense_trsm_panel_kernel, dim3(blocks_for(rest)), dim3(kB), 0, s, A, n, k0, nb);
    const unsigned tiles = unsigned((rest + kNb - 1) / kNb);
    hipLaunchKernelGGL(dense_syrk_kernel, dim3(tiles, tiles), dim3(kB), 0, s, A, n, k0, nb);
  }
  return hipGetLastError();
}
// x <- (L L^T)^-1 x
hipError_t LaunchDenseCholeskySolve(const double* A, int n, double* x, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  for (int k0 = 0; k0 < n; k0 += kNb) {
    const int nb = n - k0 < kNb ? n - k0 : kNb;
    hipLaunchKernelGGL(dense_trsv_diag_kernel, dim3(1), dim3(kNb), 0, s, A, n, k0, nb, x, 0);
    const int rest = n - k0 - nb;
    if (rest > 0) hipLaunchKernelGGL(dense_trsv_update_kernel, dim3(blocks_for(rest)), dim3(kB), 0, s, A, n, k0, nb, x, 0);
  }
  const int last = ((n - 1) / kNb) * kNb;
  for (int k0 = last; k0 >= 0; k0 -= kNb) {
    const int nb = n - k0 < kNb ? n - k0 : kNb;
    hipLaunchKernelGGL(dense_trsv_diag_kernel, dim3(1), dim3(kNb), 0, s, A, n, k0, nb, x, 1);
    if (k0 > 0) hipLaunchKernelGGL(dense_trsv_update_kernel, dim3(blocks_for(k0)), dim3(kB), 0, s, A, n, k0, nb, x, 1);
  }
  return hipGetLastError();
}

}  // namespace chip
