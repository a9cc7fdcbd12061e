// kernels_generic.hip — block-sparse kernels for ARBITRARY block sizes and cell layouts.
//
// These mirror the reference's operator decomposition one pass at a time (they are the
// correctness path for every structure that is not the static <2,3,9> case, and the
// path the known-answer problems of linear_least_squares_problems.cc run through):
//   BlockSparseMatrix::Right/LeftMultiplyAndAccumulate   I/block_sparse_matrix.cc:239-349
//   PartitionedMatrixView E/F products                   I/partitioned_matrix_view_impl.h:112-375
//   UpdateBlockDiagonalEtE / FtF, BlockJacobi blocks     :446-658, I/block_jacobi_preconditioner.cc:59-115
//   BlockRandomAccessDiagonalMatrix::Invert / apply      I/block_random_access_diagonal_matrix.cc:90-116
//   SchurEliminator::Eliminate (diagonal or dense lhs)   I/schur_eliminator_impl.h:184-311
// Parallelisation: one thread per OUTPUT scalar (row, column, or block entry), reading
// through the row structure or the transpose structure, so there are no atomics and every
// result is deterministic.  `cell_value_pos` is honoured everywhere: nothing assumes a
// value layout.
#include <hip/hip_runtime.h>

#include "device.h"

namespace chip {

namespace {

constexpr int kB = 256;

// Which cells of row block i belong to `part` (kAll / kE / kF): [k0, k1)
__device__ __forceinline__ void row_cells(const GenStructure& G, int i, int part, int& k0, int& k1) {
  k0 = G.rptr[i];
  k1 = G.rptr[i + 1];
  if (part == kE) { if (i < G.nrbe) k1 = k0 + 1; else k1 = k0; }
  else if (part == kF) { if (i < G.nrbe) k0 += 1; }
}
// Is cell k of row i (reached through the transpose of column block j) in `part`?
__device__ __forceinline__ bool cell_in_part(const GenStructure& G, int i, int k, int part) {
  if (part == kAll) return true;
  const bool is_e = (i < G.nrbe) && (k == G.rptr[i]);
  return part == kE ? is_e : !is_e;
}

__global__ __launch_bounds__(kB) void gen_right_multiply_kernel(GenStructure G, const double* __restrict__ v, int part,
                                                                const double* __restrict__ x, double* __restrict__ y,
                                                                const int* status, int first_row) {
  if (status && *status != 0) return;
  const int row = first_row + blockIdx.x * kB + threadIdx.x;
  if (row >= G.num_rows) return;
  const int i = G.row_block_of[row];
  const int r = row - G.rpos[i];
  int k0, k1;
  row_cells(G, i, part, k0, k1);
  const int xoff = (part == kF) ? G.nce : 0;
  double s = 0;
  for (int k = k0; k < k1; ++k) {
    const int j = G.ccol[k];
    const int cs = G.csz[j];
    const double* a = v + G.cval[k] + int64_t(r) * cs;
    const double* xx = x + G.cpos[j] - xoff;
    for (int c = 0; c < cs; ++c) s += a[c] * xx[c];
  }
  y[row] += s;
}

__global__ __launch_bounds__(kB) void gen_left_multiply_kernel(GenStructure G, const double* __restrict__ v, int part,
                                                               const double* __restrict__ x, double* __restrict__ y,
                                                               const int* status) {
  if (status && *status != 0) return;
  const int begin = (part == kF) ? G.nce : 0;
  const int end = (part == kE) ? G.nce : G.num_cols;
  const int col = begin + blockIdx.x * kB + threadIdx.x;
  if (col >= end) return;
  const int j = G.col_block_of[col];
  const int c = col - G.cpos[j];
  const int cs = G.csz[j];
  double s = 0;
  for (int t = G.tptr[j]; t < G.tptr[j + 1]; ++t) {
    const int i = G.trow[t], k = G.tcell[t];
    if (!cell_in_part(G, i, k, part)) continue;
    const double* a = v + G.cval[k] + c;
    const double* xx = x + G.rpos[i];
    const int rs = G.rsz[i];
    for (int r = 0; r < rs; ++r) s += a[int64_t(r) * cs] * xx[r];
  }
  y[col - begin] += s;
}

// Locate the block that owns diagonal-store entry e: largest q with off[q] <= e.
__device__ __forceinline__ int find_block(const int64_t* off, int n, int64_t e) {
  int lo = 0, hi = n;  // off has n+1 entries
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(kB) void gen_block_diagonal_kernel(GenStructure G, const double* __restrict__ v, int part,
                                                                const double* __restrict__ D, double* __restrict__ blocks) {
  const int64_t* off = part == kAll ? G.diag_off_all : (part == kE ? G.diag_off_e : G.diag_off_f);
  const int nb = part == kAll ? G.ncb : (part == kE ? G.nelim : G.ncb - G.nelim);
  const int first = part == kF ? G.nelim : 0;
  const int64_t e = int64_t(blockIdx.x) * kB + threadIdx.x;
  if (e >= off[nb]) return;
  const int q = find_block(off, nb, e);
  const int j = first + q;
  const int n = G.csz[j];
  const int a = int((e - off[q]) / n), b = int((e - off[q]) % n);
  double s = 0;
  for (int t = G.tptr[j]; t < G.tptr[j + 1]; ++t) {
    const int i = G.trow[t], k = G.tcell[t];
    if (!cell_in_part(G, i, k, part)) continue;
    const double* m = v + G.cval[k];
    const int rs = G.rsz[i];
    for (int r = 0; r < rs; ++r) s += m[int64_t(r) * n + a] * m[int64_t(r) * n + b];
  }
  if (D && a == b) { const double d = D[G.cpos[j] + a]; s += d * d; }
  blocks[e] = s;
}

__global__ __launch_bounds__(kB) void gen_squared_column_norm_kernel(GenStructure G, const double* __restrict__ v, double* __restrict__ x) {
  const int col = blockIdx.x * kB + threadIdx.x;
  if (col >= G.num_cols) return;
  const int j = G.col_block_of[col];
  const int c = col - G.cpos[j];
  const int cs = G.csz[j];
  double s = 0;
  for (int t = G.tptr[j]; t < G.tptr[j + 1]; ++t) {
    const int i = G.trow[t], k = G.tcell[t];
    const double* a = v + G.cval[k] + c;
    for (int r = 0; r < G.rsz[i]; ++r) s += a[int64_t(r) * cs] * a[int64_t(r) * cs];
  }
  x[col] = s;
}

// One thread per scalar row of a cell would need a cell index; instead one thread per scalar ROW of the
// matrix walks that row's cells (each value is touched exactly once).
__global__ __launch_bounds__(kB) void gen_scale_columns_kernel(GenStructure G, double* __restrict__ v, const double* __restrict__ scale) {
  const int row = blockIdx.x * kB + threadIdx.x;
  if (row >= G.num_rows) return;
  const int i = G.row_block_of[row];
  const int r = row - G.rpos[i];
  for (int k = G.rptr[i]; k < G.rptr[i + 1]; ++k) {
    const int j = G.ccol[k];
    const int cs = G.csz[j];
    double* a = v + G.cval[k] + int64_t(r) * cs;
    const double* sc = scale + G.cpos[j];
    for (int c = 0; c < cs; ++c) a[c] *= sc[c];
  }
}

// In-place inverse of SPD block from its upper triangle: Cholesky + solves against I.
// One thread per block; n <= kMaxGenericBlock.
__global__ __launch_bounds__(64) void gen_invert_blocks_kernel(GenStructure G, int first_block, int nblocks,
                                                               const int64_t* __restrict__ off, double* __restrict__ blocks,
                                                               int* fail_flag) {
  const int q = blockIdx.x * 64 + threadIdx.x;
  if (q >= nblocks) return;
  const int n = G.csz[first_block + q];
  double* a = blocks + (off[q] - off[0]);
  double L[kMaxGenericBlock * kMaxGenericBlock];
  double col[kMaxGenericBlock];
  bool ok = true;
  for (int j = 0; j < n; ++j) {
    double d = a[j * n + j];
    for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
    if (!(d > 0.0)) { ok = false; d = 1.0; }
    d = sqrt(d);
    L[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = a[j * n + i];
      for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = s / d;
    }
  }
  if (!ok && fail_flag) atomicExch(fail_flag, 1);
  for (int e = 0; e < n; ++e) {
    for (int i = 0; i < n; ++i) {
      double s = (i == e) ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) s -= L[i * n + k] * col[k];
      col[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = col[i];
      for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * col[k];
      col[i] = s / L[i * n + i];
    }
    for (int i = 0; i < n; ++i) a[i * n + e] = col[i];
  }
}

__global__ __launch_bounds__(kB) void gen_block_diagonal_apply_kernel(GenStructure G, int first_block, int col_begin, int ncols,
                                                                      const int64_t* __restrict__ off, const double* __restrict__ blocks,
                                                                      const double* __restrict__ x, double* __restrict__ y,
                                                                      const int* status) {
  if (status && *status != 0) return;
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= ncols) return;
  const int j = G.col_block_of[col_begin + i];
  const int n = G.csz[j];
  const int pos = G.cpos[j] - col_begin;
  const double* m = blocks + (off[j - first_block] - off[0]) + int64_t(i - pos) * n;
  double s = 0;
  for (int c = 0; c < n; ++c) s += m[c] * x[pos + c];
  y[i] += s;
}

// Entry (a,b) of S(q1,q2), q1 <= q2, of the Schur complement (without the D_f^2 term):
//   sum over rows holding both blocks of F1[:,a] . F2[:,b]                (E rows and E-free rows alike)
// - sum over chunks holding both of (E^T F1)[:,a]^T (E^T E)^-1 (E^T F2)[:,b]
// which is what ChunkDiagonalBlockAndGradient + ChunkOuterProduct + NoEBlockRowsUpdate
// accumulate (I/schur_eliminator_impl.h:449-721).  Walks the cells of column block q1 in
// row order (rows of one chunk are contiguous) and, once per chunk, the rows of that chunk
// (= the cells of the E column block in the transpose structure).
__device__ double schur_entry(const GenStructure& G, const double* __restrict__ v, const double* __restrict__ ete_inv,
                              int j1, int a, int j2, int b) {
  const int n1 = G.csz[j1], n2 = G.csz[j2];
  double s = 0;
  double u1[kMaxGenericBlock], u2[kMaxGenericBlock];
  int last_e = -1;
  for (int t = G.tptr[j1]; t < G.tptr[j1 + 1]; ++t) {
    const int i = G.trow[t], k1 = G.tcell[t];
    const int rs = G.rsz[i];
    const double* f1 = v + G.cval[k1];
    // (1) F1^T F2 over rows holding both blocks
    int k2 = -1;
    if (j1 == j2) k2 = k1;
    else for (int k = G.rptr[i]; k < G.rptr[i + 1]; ++k) if (G.ccol[k] == j2) { k2 = k; break; }
    if (k2 >= 0) {
      const double* f2 = v + G.cval[k2];
      for (int r = 0; r < rs; ++r) s += f1[int64_t(r) * n1 + a] * f2[int64_t(r) * n2 + b];
    }
    // (2) once per chunk that holds j1: -(E^T F1)[:,a]^T (E^T E)^-1 (E^T F2)[:,b]
    const int e = G.row_e_block[i];
    if (e < 0 || e == last_e) continue;
    last_e = e;
    const int es = G.csz[e];
    for (int p = 0; p < es; ++p) { u1[p] = 0; u2[p] = 0; }
    bool any2 = false;
    for (int tt = G.tptr[e]; tt < G.tptr[e + 1]; ++tt) {  // the rows of the chunk, in order
      const int ii = G.trow[tt];
      if (G.tcell[tt] != G.rptr[ii]) continue;
      const int rsi = G.rsz[ii];
      const double* E = v + G.cval[G.rptr[ii]];
      for (int k = G.rptr[ii] + 1; k < G.rptr[ii + 1]; ++k) {
        const int jj = G.ccol[k];
        if (jj != j1 && jj != j2) continue;
        const double* f = v + G.cval[k];
        if (jj == j1) {
          for (int p = 0; p < es; ++p) {
            double w = 0;
            for (int r = 0; r < rsi; ++r) w += E[int64_t(r) * es + p] * f[int64_t(r) * n1 + a];
            u1[p] += w;
          }
        }
        if (jj == j2) {
          any2 = true;
          for (int p = 0; p < es; ++p) {
            double w = 0;
            for (int r = 0; r < rsi; ++r) w += E[int64_t(r) * es + p] * f[int64_t(r) * n2 + b];
            u2[p] += w;
          }
        }
      }
    }
    if (any2) {
      const double* inv = ete_inv + G.diag_off_e[e];
      double tsum = 0;
      for (int p = 0; p < es; ++p) {
        double w = 0;
        for (int q = 0; q < es; ++q) w += inv[p * es + q] * u2[q];
        tsum += u1[p] * w;
      }
      s -= tsum;
    }
  }
  return s;
}

__global__ __launch_bounds__(kB) void gen_schur_jacobi_kernel(GenStructure G, const double* __restrict__ v,
                                                              const double* __restrict__ ete_inv, const double* __restrict__ D,
                                                              int add_f_diag, double* __restrict__ blocks) {
  const int nf = G.ncb - G.nelim;
  const int64_t e = int64_t(blockIdx.x) * kB + threadIdx.x;
  if (e >= G.diag_off_f[nf]) return;
  const int q = find_block(G.diag_off_f, nf, e);
  const int j = G.nelim + q;
  const int n = G.csz[j];
  const int a = int((e - G.diag_off_f[q]) / n), b = int((e - G.diag_off_f[q]) % n);
  double s = schur_entry(G, v, ete_inv, j, a, j, b);
  if (D && add_f_diag && a == b) { const double d = D[G.cpos[j] + a]; s += d * d; }
  blocks[e] = s;
}

// Dense lhs, num_cols_f x num_cols_f row-major; only block1 <= block2 is written, the
// rest is zeroed (the reference leaves those cells untouched after SetZero).
__global__ __launch_bounds__(kB) void gen_schur_dense_kernel(GenStructure G, const double* __restrict__ v,
                                                             const double* __restrict__ ete_inv, const double* __restrict__ D,
                                                             double* __restrict__ lhs) {
  const int64_t e = int64_t(blockIdx.x) * kB + threadIdx.x;
  const int64_t n = G.ncf;
  if (e >= n * n) return;
  const int row = int(e / n), col = int(e % n);
  const int j1 = G.col_block_of[G.nce + row], j2 = G.col_block_of[G.nce + col];
  if (j1 > j2) { lhs[e] = 0.0; return; }
  const int a = G.nce + row - G.cpos[j1], b = G.nce + col - G.cpos[j2];
  double s = schur_entry(G, v, ete_inv, j1, a, j2, b);
  if (D && row == col) { const double d = D[G.nce + row]; s += d * d; }
  lhs[e] = s;
}

// blocks[j](a,a) += D[cpos[j] + a]^2 for column blocks [first_block, first_block + nblocks)
__global__ __launch_bounds__(kB) void gen_add_diag_squares_kernel(GenStructure G, int first_block, int nblocks,
                                                                  const int64_t* __restrict__ off, const double* __restrict__ D,
                                                                  double* __restrict__ blocks) {
  const int col0 = G.cpos[first_block];
  const int last = first_block + nblocks - 1;
  const int ncols = G.cpos[last] + G.csz[last] - col0;
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= ncols) return;
  const int j = G.col_block_of[col0 + i];
  const int n = G.csz[j];
  const int a = col0 + i - G.cpos[j];
  const double d = D[col0 + i];
  blocks[(off[j - first_block] - off[0]) + int64_t(a) * n + a] += d * d;
}

// ---- explicit Schur complement, dense storage for small reduced systems ----
// The eliminator fills block1 <= block2 only (as the reference does); mirror it so that S x is a plain
// dense product (BlockRandomAccessSparseMatrix::SymmetricRightMultiplyAndAccumulate,
// I/block_random_access_sparse_matrix.cc:125-163, uses each stored block for both triangles).
__global__ __launch_bounds__(kB) void gen_symmetrize_dense_kernel(GenStructure G, double* __restrict__ lhs) {
  const int64_t e = int64_t(blockIdx.x) * kB + threadIdx.x;
  const int64_t n = G.ncf;
  if (e >= n * n) return;
  const int row = int(e / n), col = int(e % n);
  if (G.col_block_of[G.nce + row] > G.col_block_of[G.nce + col]) lhs[e] = lhs[int64_t(col) * n + row];
}

// y = S x, one wavefront per row (S row-major n x n, symmetric)
__global__ __launch_bounds__(kB) void gen_dense_symv_kernel(const double* __restrict__ S, int n, const double* __restrict__ x,
                                                            double* __restrict__ y, const int* __restrict__ status) {
  if (status && *status != 0) return;
  const int row = blockIdx.x * (kB / 64) + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  const double* r = S + int64_t(row) * n;
  double v = 0.0;
  for (int j = lane; j < n; j += 64) v += r[j] * x[j];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  if (lane == 0) y[row] = v;
}

// blocks[j] = S(j, j) for the F column blocks (the SCHUR_JACOBI preconditioner of the explicit solver,
// I/schur_complement_solver.cc:361-381); one thread per row of S
__global__ __launch_bounds__(kB) void gen_extract_diag_blocks_kernel(GenStructure G, const double* __restrict__ S,
                                                                     const int64_t* __restrict__ off, double* __restrict__ blocks) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= G.ncf) return;
  const int j = G.col_block_of[G.nce + i];
  const int nj = G.csz[j];
  const int c0 = G.cpos[j] - G.nce;
  const int a = i - c0;
  double* o = blocks + (off[j - G.nelim] - off[0]) + int64_t(a) * nj;
  const double* r = S + int64_t(i) * G.ncf + c0;
  for (int b = 0; b < nj; ++b) o[b] = r[b];
}

inline unsigned blocks_for(int64_t n) { return unsigned((n + kB - 1) / kB); }

// ---------------------------------------------------------------------------------------------------------------------------
// GROUPED kernels: L lanes (a power of two, 4 .. 64) per column block or chunk, striding over its cells / rows; every lane
// accumulates the block's whole (small) result in registers, the group combines by butterfly shuffles (a fixed tree:
// deterministic).  The thread-per-output-scalar kernels above walk a column block's cells ALONE: on the Ladybug shape a
// camera's 394 cells are a 394-step dependent chain per thread with 15 507 threads in flight — S.x took 20.5 ms, 0.0009 of the
// HBM peak (profiles/r03t_*).  MAXC / MAXE bound the block sizes a kernel is compiled for (predicated, fully unrolled register
// arrays); larger blocks keep the old kernels.
// ---------------------------------------------------------------------------------------------------------------------------
template <int L>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
  for (int m = L / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// y[cpos(j) - ybase + c] += sum over the cells of column block j (in `part`) of sum_r A[r][c] x[rpos(i) + r],  j in [j0, j0 + nb)
// (x == nullptr: the squared column norms instead, y[col] += sum of A[r][c]^2 — BlockSparseMatrix::SquaredColumnNorm)
template <int L, int MAXC>
__global__ __launch_bounds__(kB) void gen_left_multiply_grouped_kernel(GenStructure G, const double* __restrict__ v, int part, int j0, int nb,
                                                                       int ybase, const double* __restrict__ x, double* __restrict__ y,
                                                                       const int* status) {
  if (status && *status != 0) return;
  const int64_t grp = (int64_t(blockIdx.x) * kB + threadIdx.x) / L;
  const int gl = threadIdx.x & (L - 1);
  if (grp >= nb) return;
  const int j = j0 + int(grp);
  const int cs = G.csz[j];
  double s[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) s[c] = 0.0;
  for (int t = G.tptr[j] + gl; t < G.tptr[j + 1]; t += L) {
    const int i = G.trow[t], k = G.tcell[t];
    if (!cell_in_part(G, i, k, part)) continue;
    const double* a = v + G.cval[k];
    const int rs = G.rsz[i];
    for (int r = 0; r < rs; ++r) {
      const double xr = x ? x[G.rpos[i] + r] : 0.0;
#pragma unroll
      for (int c = 0; c < MAXC; ++c) if (c < cs) { const double av = a[r * cs + c]; s[c] += av * (x ? xr : av); }
    }
  }
  double* out = y + G.cpos[j] - ybase;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const double t = group_sum<L>(s[c]);
    if (c < cs && (c & (L - 1)) == gl) out[c] += t;
  }
}

// One group per chunk: w = ete_inv(e) * sum over the chunk's rows of E_r^T t_r; then t_r -= E_r w (update_t) and / or x_e = w.
template <int L, int MAXE>
__global__ __launch_bounds__(kB) void gen_chunk_project_kernel(GenStructure G, const double* __restrict__ v, const double* __restrict__ ete_inv,
                                                               double* __restrict__ t_rows, int update_t, double* __restrict__ x_e,
                                                               const int* status) {
  if (status && *status != 0) return;
  const int64_t grp = (int64_t(blockIdx.x) * kB + threadIdx.x) / L;
  const int gl = threadIdx.x & (L - 1);
  if (grp >= G.nelim) return;
  const int e = int(grp);
  const int es = G.csz[e];
  const int i0 = G.chunk_start[e], i1 = i0 + G.chunk_size[e];
  double u[MAXE];
#pragma unroll
  for (int p = 0; p < MAXE; ++p) u[p] = 0.0;
  for (int i = i0 + gl; i < i1; i += L) {
    const double* E = v + G.cval[G.rptr[i]];
    const double* tt = t_rows + G.rpos[i];
    const int rs = G.rsz[i];
    for (int r = 0; r < rs; ++r) {
      const double tr = tt[r];
#pragma unroll
      for (int p = 0; p < MAXE; ++p) if (p < es) u[p] += E[r * es + p] * tr;
    }
  }
#pragma unroll
  for (int p = 0; p < MAXE; ++p) u[p] = group_sum<L>(u[p]);
  const double* inv = ete_inv + G.diag_off_e[e];
  double w[MAXE];
#pragma unroll
  for (int p = 0; p < MAXE; ++p) {
    double a = 0.0;
#pragma unroll
    for (int q = 0; q < MAXE; ++q) if (p < es && q < es) a += inv[p * es + q] * u[q];
    w[p] = a;
  }
  if (x_e) {
#pragma unroll
    for (int p = 0; p < MAXE; ++p) if (p < es && (p & (L - 1)) == gl) x_e[G.cpos[e] + p] = w[p];
  }
  if (update_t) {
    for (int i = i0 + gl; i < i1; i += L) {
      const double* E = v + G.cval[G.rptr[i]];
      double* tt = t_rows + G.rpos[i];
      const int rs = G.rsz[i];
      for (int r = 0; r < rs; ++r) {
        double a = 0.0;
#pragma unroll
        for (int p = 0; p < MAXE; ++p) if (p < es) a += E[r * es + p] * w[p];
        tt[r] -= a;
      }
    }
  }
}

// The row-space half of S x in ONE launch, a group per chunk: z_r = t_r - E_r (E^T E)^-1 sum_r' E_r'^T t_r' with t_r = sum over the
// row's F cells of F x_f — what RightMultiplyF, LeftMultiplyE, the block-diagonal solve and RightMultiplyE do in
// ImplicitSchurComplement::RightMultiplyAndAccumulate (I/implicit_schur_complement.cc:106-144); z goes to `z_rows`, F^T z follows
// (LaunchGenLeftMultiply).  Rows up to MAXR high, E up to MAXE wide, F cells up to MAXC wide: every load of a cell is issued
// unconditionally (clamped), see gen_left_multiply_items_kernel.  A lane whose chunk has more than L rows recomputes t in the second sweep.
template <int L, int MAXR, int MAXE, int MAXC>
__global__ __launch_bounds__(kB) void gen_chunk_sx_kernel(GenStructure G, const double* __restrict__ v, const double* __restrict__ ete_inv,
                                                          const double* __restrict__ x_f, double* __restrict__ z_rows, const int* status) {
  if (status && *status != 0) return;
  const int64_t grp = (int64_t(blockIdx.x) * kB + threadIdx.x) / L;
  const int gl = threadIdx.x & (L - 1);
  if (grp >= G.nelim) return;
  const int e = int(grp);
  const int es = G.csz[e];
  const int i0 = G.chunk_start[e], i1 = i0 + G.chunk_size[e];
  auto f_times_x = [&](int i, int rs, double (&t)[MAXR]) {
#pragma unroll
    for (int r = 0; r < MAXR; ++r) t[r] = 0.0;
    for (int k = G.rptr[i] + 1; k < G.rptr[i + 1]; ++k) {
      const int j = G.ccol[k];
      const int cs = G.csz[j];
      const double* a = v + G.cval[k];
      const double* xx = x_f + (G.cpos[j] - G.nce);
      const int last = rs * cs - 1;
      double av[MAXR][MAXC], xv[MAXC];
#pragma unroll
      for (int c = 0; c < MAXC; ++c) xv[c] = xx[c < cs ? c : 0];
#pragma unroll
      for (int r = 0; r < MAXR; ++r)
#pragma unroll
        for (int c = 0; c < MAXC; ++c) { const int q = r * cs + c; av[r][c] = a[q < last ? q : last]; }
#pragma unroll
      for (int r = 0; r < MAXR; ++r)
#pragma unroll
        for (int c = 0; c < MAXC; ++c) t[r] += (c < cs && r < rs) ? av[r][c] * xv[c] : 0.0;
    }
  };
  auto load_e = [&](int i, int rs, double (&E)[MAXR][MAXE]) {
    const double* m = v + G.cval[G.rptr[i]];
    const int last = rs * es - 1;
#pragma unroll
    for (int r = 0; r < MAXR; ++r)
#pragma unroll
      for (int p = 0; p < MAXE; ++p) { const int q = r * es + p; const double val = m[q < last ? q : last]; E[r][p] = (r < rs && p < es) ? val : 0.0; }
  };
  double u[MAXE], t_first[MAXR], E_first[MAXR][MAXE];
#pragma unroll
  for (int p = 0; p < MAXE; ++p) u[p] = 0.0;
  for (int i = i0 + gl; i < i1; i += L) {
    const int rs = G.rsz[i];
    double t[MAXR], E[MAXR][MAXE];
    f_times_x(i, rs, t);
    load_e(i, rs, E);
#pragma unroll
    for (int r = 0; r < MAXR; ++r)
#pragma unroll
      for (int p = 0; p < MAXE; ++p) u[p] += E[r][p] * t[r];
    if (i == i0 + gl) {
#pragma unroll
      for (int r = 0; r < MAXR; ++r) { t_first[r] = t[r];
#pragma unroll
        for (int p = 0; p < MAXE; ++p) E_first[r][p] = E[r][p]; }
    }
  }
#pragma unroll
  for (int p = 0; p < MAXE; ++p) u[p] = group_sum<L>(u[p]);
  const double* inv = ete_inv + G.diag_off_e[e];
  double w[MAXE];
#pragma unroll
  for (int p = 0; p < MAXE; ++p) {
    double a = 0.0;
#pragma unroll
    for (int q = 0; q < MAXE; ++q) if (p < es && q < es) a += inv[p * es + q] * u[q];
    w[p] = a;
  }
  for (int i = i0 + gl; i < i1; i += L) {
    const int rs = G.rsz[i];
    double t[MAXR], E[MAXR][MAXE];
    if (i == i0 + gl) {
#pragma unroll
      for (int r = 0; r < MAXR; ++r) { t[r] = t_first[r];
#pragma unroll
        for (int p = 0; p < MAXE; ++p) E[r][p] = E_first[r][p]; }
    } else {
      f_times_x(i, rs, t);
      load_e(i, rs, E);
    }
    double* zz = z_rows + G.rpos[i];
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
      double a = t[r];
#pragma unroll
      for (int p = 0; p < MAXE; ++p) a -= E[r][p] * w[p];
      if (r < rs) zz[r] = a;
    }
  }
}

// blocks(j) = sum over the cells of column block j (in `part`) of A^T A (+ D^2), j in [j0, j0 + nb); off = the offsets of the store
// the blocks live in, indexed by j - first (first = the store's first block)
template <int L, int MAXC>
__global__ __launch_bounds__(kB) void gen_block_diagonal_grouped_kernel(GenStructure G, const double* __restrict__ v, int part, int j0, int nb,
                                                                        const int64_t* __restrict__ off, int first, const double* __restrict__ D,
                                                                        double* __restrict__ blocks) {
  constexpr int NT = MAXC * (MAXC + 1) / 2;
  const int64_t grp = (int64_t(blockIdx.x) * kB + threadIdx.x) / L;
  const int gl = threadIdx.x & (L - 1);
  if (grp >= nb) return;
  const int j = j0 + int(grp);
  const int n = G.csz[j];
  double acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = 0.0;
  for (int t = G.tptr[j] + gl; t < G.tptr[j + 1]; t += L) {
    const int i = G.trow[t], k = G.tcell[t];
    if (!cell_in_part(G, i, k, part)) continue;
    const double* m = v + G.cval[k];
    const int rs = G.rsz[i];
    for (int r = 0; r < rs; ++r) {
      double row[MAXC];
#pragma unroll
      for (int c = 0; c < MAXC; ++c) row[c] = c < n ? m[r * n + c] : 0.0;
      int idx = 0;
#pragma unroll
      for (int a = 0; a < MAXC; ++a)
#pragma unroll
        for (int b = a; b < MAXC; ++b) acc[idx++] += row[a] * row[b];
    }
  }
  double* o = blocks + (off[j - first] - off[0]);
  int idx = 0;
#pragma unroll
  for (int a = 0; a < MAXC; ++a)
#pragma unroll
    for (int b = a; b < MAXC; ++b) {
      double s = group_sum<L>(acc[idx]);
      if (a < n && b < n && (idx & (L - 1)) == gl) {
        if (D && a == b) { const double d = D[G.cpos[j] + a]; s += d * d; }
        o[a * n + b] = s;
        o[b * n + a] = s;
      }
      ++idx;
    }
}

// Diagonal blocks of the Schur complement, one group per F block j: its cells in row order; the cells of j inside ONE chunk are
// adjacent in the transpose list, and the lane that meets the first of them takes the whole run:
//   S_jj += sum over the run of F^T F - G^T (E^T E)^-1 G,  G = sum over the run of E_r^T F_r       (an E-free row: F^T F alone)
// — ChunkDiagonalBlockAndGradient + ChunkOuterProduct + NoEBlockRowsUpdate restricted to the cell (j, j)
// (I/schur_eliminator_impl.h:449-721).
template <int L, int MAXE, int MAXC>
__global__ __launch_bounds__(kB) void gen_schur_jacobi_grouped_kernel(GenStructure G, const double* __restrict__ v, const double* __restrict__ ete_inv,
                                                                      const double* __restrict__ D, int add_f_diag, double* __restrict__ blocks) {
  constexpr int NT = MAXC * (MAXC + 1) / 2;
  const int nf = G.ncb - G.nelim;
  const int64_t grp = (int64_t(blockIdx.x) * kB + threadIdx.x) / L;
  const int gl = threadIdx.x & (L - 1);
  if (grp >= nf) return;
  const int j = G.nelim + int(grp);
  const int n = G.csz[j];
  double acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = 0.0;
  const int t0 = G.tptr[j], t1 = G.tptr[j + 1];
  for (int t = t0 + gl; t < t1; t += L) {
    const int i = G.trow[t];
    const int e = G.row_e_block[i];
    if (e >= 0 && t > t0 && G.row_e_block[G.trow[t - 1]] == e) continue;   // not the first cell of j in its chunk
    const int es = e >= 0 ? G.csz[e] : 0;
    double g[MAXE][MAXC];
#pragma unroll
    for (int p = 0; p < MAXE; ++p)
#pragma unroll
      for (int c = 0; c < MAXC; ++c) g[p][c] = 0.0;
    for (int tt = t; tt < t1; ++tt) {   // the run of j's cells in this chunk (an E-free row: itself alone)
      const int ii = G.trow[tt];
      if (tt > t && (e < 0 || G.row_e_block[ii] != e)) break;
      const double* f = v + G.cval[G.tcell[tt]];
      const double* E = e >= 0 ? v + G.cval[G.rptr[ii]] : nullptr;
      const int rs = G.rsz[ii];
      for (int r = 0; r < rs; ++r) {
        double row[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) row[c] = c < n ? f[r * n + c] : 0.0;
        int idx = 0;
#pragma unroll
        for (int a = 0; a < MAXC; ++a)
#pragma unroll
          for (int b = a; b < MAXC; ++b) acc[idx++] += row[a] * row[b];
        if (e >= 0) {
#pragma unroll
          for (int p = 0; p < MAXE; ++p) {
            const double ep = p < es ? E[r * es + p] : 0.0;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) g[p][c] += ep * row[c];
          }
        }
      }
    }
    if (e >= 0) {
      const double* inv = ete_inv + G.diag_off_e[e];
      double w[MAXE][MAXC];   // (E^T E)^-1 G
#pragma unroll
      for (int p = 0; p < MAXE; ++p)
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
          double a = 0.0;
#pragma unroll
          for (int q = 0; q < MAXE; ++q) if (p < es && q < es) a += inv[p * es + q] * g[q][c];
          w[p][c] = a;
        }
      int idx = 0;
#pragma unroll
      for (int a = 0; a < MAXC; ++a)
#pragma unroll
        for (int b = a; b < MAXC; ++b) {
          double sub = 0.0;
#pragma unroll
          for (int p = 0; p < MAXE; ++p) sub += g[p][a] * w[p][b];
          acc[idx++] -= sub;
        }
    }
  }
  double* o = blocks + G.diag_off_f[grp];
  int idx = 0;
#pragma unroll
  for (int a = 0; a < MAXC; ++a)
#pragma unroll
    for (int b = a; b < MAXC; ++b) {
      double s = group_sum<L>(acc[idx]);
      if (a < n && b < n && (idx & (L - 1)) == gl) {
        if (D && add_f_diag && a == b) { const double d = D[G.cpos[j] + a]; s += d * d; }
        o[a * n + b] = s;
        o[b * n + a] = s;
      }
      ++idx;
    }
}


// ---- remainder rows of the fused path (rows without a point cell; R = their own GenStructure, compact row space) ----
// Templated on the camera width NF (the widths kernels_bal.inc is compiled for: 2, 3, 4, 6, 8, 9, 10; round 5 — the round-3 kernels
// were 9-wide and plan.cc kept every other shape with such rows off the fused path).
// out[NF NF c + NF a + b] = sum over the remainder cells of camera c of (F^T F)(a, b): what SchurEliminator::NoEBlockRowsUpdate adds to
// the diagonal cell of S (I/schur_eliminator_impl.h:574-666) and UpdateBlockDiagonalFtF's second loop to blockdiag(F^T F)
// (I/partitioned_matrix_view_impl.h:617-658).
// One WAVEFRONT per camera, one LANE per cell (64 cells per round): a lane forms its cell's F^T F — the upper-triangle entries, every
// load of a row in flight at once — and the wave adds the lanes' results up by shuffles.  (Thread per entry, as this kernel first was,
// and then wave per camera with every lane walking all the cells, are one long chain of dependent loads: 131 / 140 us for 50 k prior rows
// on 1778 cameras.)
template <int NF, int RS>   // RS: rows of a cell known at compile time (NF: priors on whole cameras), or 0: read from the structure
__device__ __forceinline__ void cell_ftf_upper(const double* __restrict__ m, int rs, double (&u)[NF * (NF + 1) / 2]) {
  const int n = RS > 0 ? RS : rs;
#pragma unroll
  for (int r = 0; r < (RS > 0 ? RS : 1); ++r) {
    for (int rr = r; rr < n; rr += (RS > 0 ? RS : 1)) {
      double f[NF];
#pragma unroll
      for (int k = 0; k < NF; ++k) f[k] = m[rr * NF + k];
      int idx = 0;
#pragma unroll
      for (int a = 0; a < NF; ++a)
#pragma unroll
        for (int b2 = a; b2 < NF; ++b2) u[idx++] += f[a] * f[b2];
    }
  }
}
template <int NF>
__global__ __launch_bounds__(kB) void rem_camera_blocks_kernel(GenStructure R, const double* __restrict__ v, const int32_t* __restrict__ cam_block,
                                                               int n_cameras, double* __restrict__ out) {
  constexpr int NU = NF * (NF + 1) / 2;
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * (kB / 64) + (threadIdx.x >> 6);
  if (c >= n_cameras) return;
  const int j = cam_block[c];
  double u[NU];
#pragma unroll
  for (int i = 0; i < NU; ++i) u[i] = 0.0;
  for (int t = R.tptr[j] + lane; t < R.tptr[j + 1]; t += 64) {
    const int rs = R.rsz[R.trow[t]];
    const double* m = v + R.cval[R.tcell[t]];
    if (rs == NF) cell_ftf_upper<NF, NF>(m, rs, u);
    else cell_ftf_upper<NF, 0>(m, rs, u);
  }
  double* o = out + int64_t(NF * NF) * c;
  int idx = 0;
#pragma unroll
  for (int a = 0; a < NF; ++a)
#pragma unroll
    for (int b2 = a; b2 < NF; ++b2) {
      double x = u[idx++];
#pragma unroll
      for (int w = 32; w >= 1; w >>= 1) x += __shfl_xor(x, w, 64);
      if (lane == 0) { o[NF * a + b2] = x; o[NF * b2 + a] = x; }
    }
}
// y_f[pos(c) + k] += sum over the remainder cells of camera c of sum_r F[r][k] t[row + r]: F_R^T t on the NF-wide camera blocks, one
// WAVEFRONT per camera — lane = (cell slot 0..CS-1, column k 0..NF-1) with CS = 64 / NF cells per round (9 wide: seven), the CS partial
// sums of a column combined through a fixed shuffle tree (deterministic); a cell of NF rows (a prior on a whole camera) has all its loads
// issued before the first is used.  (gen_left_multiply_kernel's thread per output scalar walks a camera's cells alone: 111 us per call on
// a Venice-shaped problem with 1 % prior rows, three calls per solve: 1.32x the pure problem.)
template <int NF>
__global__ __launch_bounds__(kB) void rem_left_multiply_kernel(GenStructure R, const double* __restrict__ v, const int32_t* __restrict__ cam_block,
                                                               const int32_t* __restrict__ cam_pos, int cam_base, int n_cameras,
                                                               const double* __restrict__ t_rows, double* __restrict__ y_f, const int* __restrict__ status) {
  if (status && *status != 0) return;
  constexpr int CS = 64 / NF;   // cell slots of a wavefront (lanes CS NF .. 63 idle)
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * (kB / 64) + (threadIdx.x >> 6);
  if (c >= n_cameras) return;
  const int j = cam_block[c];
  const int cs = lane / NF, k = lane - NF * cs;
  double acc = 0;
  const int t0 = R.tptr[j], t1 = R.tptr[j + 1];
  for (int t = t0 + cs; t < t1 && cs < CS; t += CS) {
    const int i = R.trow[t];
    const double* m = v + R.cval[R.tcell[t]] + k;
    const double* tr = t_rows + R.rpos[i];
    const int rs = R.rsz[i];
    if (rs == NF) {
      double a[NF], x[NF];
#pragma unroll
      for (int r = 0; r < NF; ++r) { a[r] = m[r * NF]; x[r] = tr[r]; }
#pragma unroll
      for (int r = 0; r < NF; ++r) acc += a[r] * x[r];
    } else {
      for (int r = 0; r < rs; ++r) acc += m[r * NF] * tr[r];
    }
  }
  // slot s (lanes [s NF, (s + 1) NF)) adds slot s + h for h = the powers of two below CS, largest first: slot 0 ends with every slot's sum
  constexpr int H0 = CS > 16 ? 16 : (CS > 8 ? 8 : (CS > 4 ? 4 : (CS > 2 ? 2 : 1)));
#pragma unroll
  for (int h = H0; h >= 1; h >>= 1) {
    const double u = __shfl_down(acc, h * NF, 64);
    if (cs + h < CS && cs < h) acc += u;
  }
  if (lane < NF) y_f[(cam_pos ? cam_pos[c] : cam_base + NF * c) + k] += acc;   // (cam_base: where camera 0 sits in the F-space vectors when they are back to back)
}
// y[pos(c) + k] += blocks[nf nf c + (nf + 1) k]: the remainder's share of the camera columns' squared norms
__global__ __launch_bounds__(kB) void rem_add_diag_kernel(const double* __restrict__ blocks, const int32_t* __restrict__ cam_pos, int cam_base, int n_cameras,
                                                          int nf, double* __restrict__ y) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= nf * n_cameras) return;
  const int c = i / nf, k = i % nf;
  y[(cam_pos ? cam_pos[c] : cam_base + nf * c) + k] += blocks[int64_t(nf) * nf * c + (nf + 1) * k];
}
// out[r] = in[map[r]]: the remainder rows' residuals into their compact row space (rows anywhere among the observation rows)
__global__ __launch_bounds__(kB) void gather_rows_kernel(const double* __restrict__ in, const int32_t* __restrict__ map, int n, double* __restrict__ out) {
  const int r = blockIdx.x * kB + threadIdx.x;
  if (r < n) out[r] = in[map[r]];
}
// One workgroup: *out = sum_r m_r (f_r - m_r / 2)   (mode 0: m = J x of the un-negated solution, the back-substitution kernel's
// convention) or  -sum_r m_r (f_r + m_r / 2)  (mode 1: m = J step) — the remainder rows' share of the model cost change,
// I/trust_region_minimizer.cc:420-438.  gate: optional CG status word, as in the fused kernels.
__global__ __launch_bounds__(kB) void rem_model_cost_kernel(const double* __restrict__ m, const double* __restrict__ f, int n, int mode,
                                                            double* __restrict__ out) {
  __shared__ double sh[kB / 64];
  double acc = 0;
  for (int r = threadIdx.x; r < n; r += kB) acc += mode == 0 ? m[r] * (f[r] - 0.5 * m[r]) : -m[r] * (f[r] + 0.5 * m[r]);
#pragma unroll
  for (int w = 32; w >= 1; w >>= 1) acc += __shfl_xor(acc, w, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) { double t = 0; for (int i = 0; i < kB / 64; ++i) t += sh[i]; *out = t; }
}

}  // namespace

// one instantiation per camera width kernels_bal.inc is compiled for (common.h: BalShapeCompiled)
// (every camera width the fused path is compiled for: build.py BAL_SHAPES — a width missing here made every pass over rows without a point cell fail
// with hipErrorInvalidValue for cameras 5 and 7 wide, found by tools/fuzz_parity.py)
#define CERES_HIP_REM_WIDTHS(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10)
hipError_t LaunchRemCameraBlocks(const GenStructure& R, const double* values, const int32_t* cam_block, int n_cameras, int nf, double* out, hipStream_t s) {
  if (n_cameras <= 0) return hipSuccess;
  const dim3 grid((n_cameras + kB / 64 - 1) / (kB / 64));
  switch (nf) {
#define X(W) case W: hipLaunchKernelGGL(rem_camera_blocks_kernel<W>, grid, dim3(kB), 0, s, R, values, cam_block, n_cameras, out); break;
    CERES_HIP_REM_WIDTHS(X)
#undef X
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t LaunchRemLeftMultiply(const GenStructure& R, const double* values, const int32_t* cam_block, const int32_t* cam_pos, int cam_base, int n_cameras,
                                 int nf, const double* t_rows, double* y_f, const int* status, hipStream_t s) {
  if (n_cameras <= 0) return hipSuccess;
  const dim3 grid((n_cameras + kB / 64 - 1) / (kB / 64));
  switch (nf) {
#define X(W) case W: hipLaunchKernelGGL(rem_left_multiply_kernel<W>, grid, dim3(kB), 0, s, R, values, cam_block, cam_pos, cam_base, n_cameras, t_rows, y_f, status); break;
    CERES_HIP_REM_WIDTHS(X)
#undef X
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t LaunchRemAddDiag(const double* blocks, const int32_t* cam_pos, int cam_base, int n_cameras, int nf, double* y, hipStream_t s) {
  if (n_cameras > 0) hipLaunchKernelGGL(rem_add_diag_kernel, dim3(blocks_for(int64_t(nf) * n_cameras)), dim3(kB), 0, s, blocks, cam_pos, cam_base, n_cameras, nf, y);
  return hipGetLastError();
}
hipError_t LaunchGatherRows(const double* in, const int32_t* map, int n, double* out, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks_for(n)), dim3(kB), 0, s, in, map, n, out);
  return hipGetLastError();
}
hipError_t LaunchRemModelCost(const double* m, const double* f, int n, int mode, double* out, hipStream_t s) {
  hipLaunchKernelGGL(rem_model_cost_kernel, dim3(1), dim3(kB), 0, s, m, f, n, mode, out);
  return hipGetLastError();
}

hipError_t LaunchGenRightMultiply(const GenStructure& G, const double* values, int part, const double* x, double* y,
                                  const int* status, hipStream_t s) {
  if (G.num_rows > 0) hipLaunchKernelGGL(gen_right_multiply_kernel, dim3(blocks_for(G.num_rows)), dim3(kB), 0, s, G, values, part, x, y, status, 0);
  return hipGetLastError();
}
// ---- itemized forms: one wave per ITEM (device.h: GenItems), partial results to scratch, a second kernel adds a block's items up ----
// is transpose entry t (info word) in `part`?
__device__ __forceinline__ bool info_in_part(int info, int part) { return part == kAll || ((info >> 8) & 1) == (part == kE ? 1 : 0); }

template <int MAXC, int MAXR>
__global__ __launch_bounds__(kB) void gen_left_multiply_items_kernel(GenStructure G, const double* __restrict__ v, int part, const double* __restrict__ x,
                                                                     const int* status) {
  if (status && *status != 0) return;
  const int item = blockIdx.x * (kB / 64) + (threadIdx.x >> 6);
  if (item >= G.items.count) return;
  const int lane = threadIdx.x & 63;
  const int j = G.items.block[item];
  const int cs = G.csz[j];
  double s[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) s[c] = 0.0;
  const int t1 = G.items.t1[item];
  // two cells per lane and round: their loads are independent and all in flight together (rows up to MAXR high: fully unrolled, predicated)
  for (int t = G.items.t0[item] + lane; t < t1; t += 128) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int tt = t + 64 * h;
      const bool on = tt < t1;
      const int ts = on ? tt : t;
      const int info = G.tinfo[ts];
      const int rs = (on && info_in_part(info, part)) ? (info & 0xff) : 0;
      const double* a = v + G.tval[ts];
      const double* xx = x ? x + G.trpos[ts] : a;   // (x == nullptr: the squared column norms; xx is then a valid address that is not used)
      if constexpr (MAXR <= 4) {
        // every load of the cell is issued unconditionally, back to back (addresses clamped into the cell; what lies outside is
        // selected away afterwards): loads under a predicate are separated by waits, and a 144-byte cell read by eighteen loads that
        // are far apart in time is fetched from the L2 eighteen times — 64 lanes x 3 lines x the waves of a CU do not fit its L1
        const int last = rs > 0 ? rs * cs - 1 : 0;
        double av[MAXR][MAXC], xv[MAXR];
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
          xv[r] = xx[r < rs ? r : 0];
#pragma unroll
          for (int c = 0; c < MAXC; ++c) { const int e = r * cs + c; av[r][c] = a[e < last ? e : last]; }
        }
#pragma unroll
        for (int r = 0; r < MAXR; ++r)
#pragma unroll
          for (int c = 0; c < MAXC; ++c) s[c] += (c < cs && r < rs) ? av[r][c] * (x ? xv[r] : av[r][c]) : 0.0;
      } else {
        for (int r = 0; r < rs; ++r) {
          const double xr = xx[r];
#pragma unroll
          for (int c = 0; c < MAXC; ++c) if (c < cs) { const double av = a[r * cs + c]; s[c] += av * (x ? xr : av); }
        }
      }
    }
  }
  double* out = G.items.scratch + int64_t(item) * kGenItemValues;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const double t = group_sum<64>(s[c]);
    if (lane == c) out[c] = t;
  }
}
// y[cpos(j) - ybase + c] += the items' partial sums: a wave per block, lane l adds the items l, l + 64, ... (fixed order), the lanes'
// sums combine by butterfly; lane c stores column c
__global__ __launch_bounds__(kB) void gen_left_multiply_combine_kernel(GenStructure G, int ybase, double* __restrict__ y, const int* status) {
  if (status && *status != 0) return;
  const int q = blockIdx.x * (kB / 64) + (threadIdx.x >> 6);
  if (q >= G.items.nblocks) return;
  const int lane = threadIdx.x & 63;
  const int j = G.items.first_block + q;
  const int cs = G.csz[j];
  double s[kMaxGenericBlock];
#pragma unroll
  for (int c = 0; c < kMaxGenericBlock; ++c) s[c] = 0.0;
  for (int it = G.items.block_ptr[q] + lane; it < G.items.block_ptr[q + 1]; it += 64) {
    const double* p = G.items.scratch + int64_t(it) * kGenItemValues;
#pragma unroll
    for (int c = 0; c < kMaxGenericBlock; ++c) if (c < cs) s[c] += p[c];
  }
#pragma unroll
  for (int c = 0; c < kMaxGenericBlock; ++c) {
    const double t = group_sum<64>(s[c]);
    if (lane == c && c < cs) y[G.cpos[j] - ybase + c] += t;
  }
}

// Per item: the packed upper triangle (10 x 10 layout) of sum A^T A over the item's cells (schur == 0), or of the item's share of the
// Schur complement's diagonal block (schur != 0: see gen_schur_jacobi_grouped_kernel; an item never cuts a chunk's run of cells).
template <int MAXE, int MAXC>
__global__ __launch_bounds__(kB) void gen_block_items_kernel(GenStructure G, const double* __restrict__ v, int part, int schur,
                                                             const double* __restrict__ ete_inv) {
  constexpr int NT = MAXC * (MAXC + 1) / 2;
  static_assert(NT <= kGenItemValues, "scratch row");
  const int item = blockIdx.x * (kB / 64) + (threadIdx.x >> 6);
  if (item >= G.items.count) return;
  const int lane = threadIdx.x & 63;
  const int j = G.items.block[item];
  const int n = G.csz[j];
  double acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = 0.0;
  const int t0 = G.items.t0[item], t1 = G.items.t1[item];
  for (int t = t0 + lane; t < t1; t += 64) {
    const int i = schur ? G.trow[t] : 0;
    if (!schur) {
      const int info = G.tinfo[t];
      if (!info_in_part(info, part)) continue;
      const double* m = v + G.tval[t];
      const int rs = info & 0xff;
      if (rs <= 2) {   // (the common case) both rows' loads issued together, unconditionally: see gen_left_multiply_items_kernel
        const int last = rs * n - 1;
        double row0[MAXC], row1[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) { row0[c] = m[c < last ? c : last]; row1[c] = m[n + c < last ? n + c : last]; }
#pragma unroll
        for (int c = 0; c < MAXC; ++c) { row0[c] = c < n ? row0[c] : 0.0; row1[c] = (c < n && rs > 1) ? row1[c] : 0.0; }
        int idx = 0;
#pragma unroll
        for (int a = 0; a < MAXC; ++a)
#pragma unroll
          for (int b = a; b < MAXC; ++b) acc[idx++] += row0[a] * row0[b] + row1[a] * row1[b];
        continue;
      }
      for (int r = 0; r < rs; ++r) {
        double row[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) row[c] = c < n ? m[r * n + c] : 0.0;
        int idx = 0;
#pragma unroll
        for (int a = 0; a < MAXC; ++a)
#pragma unroll
          for (int b = a; b < MAXC; ++b) acc[idx++] += row[a] * row[b];
      }
      continue;
    }
    const int e = G.row_e_block[i];
    if (e >= 0 && t > t0 && G.row_e_block[G.trow[t - 1]] == e) continue;   // not the first cell of j in its chunk
    const int es = e >= 0 ? G.csz[e] : 0;
    double g[MAXE][MAXC];
#pragma unroll
    for (int p = 0; p < MAXE; ++p)
#pragma unroll
      for (int c = 0; c < MAXC; ++c) g[p][c] = 0.0;
    for (int tt = t; tt < t1; ++tt) {   // the run of j's cells in this chunk (an E-free row: itself alone)
      const int ii = G.trow[tt];
      if (tt > t && (e < 0 || G.row_e_block[ii] != e)) break;
      const double* f = v + G.cval[G.tcell[tt]];
      const double* E = e >= 0 ? v + G.cval[G.rptr[ii]] : nullptr;
      const int rs = G.rsz[ii];
      for (int r = 0; r < rs; ++r) {
        double row[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) row[c] = c < n ? f[r * n + c] : 0.0;
        int idx = 0;
#pragma unroll
        for (int a = 0; a < MAXC; ++a)
#pragma unroll
          for (int b = a; b < MAXC; ++b) acc[idx++] += row[a] * row[b];
        if (e >= 0) {
#pragma unroll
          for (int p = 0; p < MAXE; ++p) {
            const double ep = p < es ? E[r * es + p] : 0.0;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) g[p][c] += ep * row[c];
          }
        }
      }
    }
    if (e >= 0) {
      const double* inv = ete_inv + G.diag_off_e[e];
      int idx = 0;
#pragma unroll
      for (int a = 0; a < MAXC; ++a) {
        double wa[MAXE];   // column a of (E^T E)^-1 G ... by symmetry of the inverse: row a of G^T (E^T E)^-1
#pragma unroll
        for (int p = 0; p < MAXE; ++p) {
          double x = 0.0;
#pragma unroll
          for (int q = 0; q < MAXE; ++q) if (p < es && q < es) x += inv[q * es + p] * g[q][a];
          wa[p] = x;
        }
#pragma unroll
        for (int b = a; b < MAXC; ++b) {
          double sub = 0.0;
#pragma unroll
          for (int p = 0; p < MAXE; ++p) sub += wa[p] * g[p][b];
          acc[idx++] -= sub;
        }
      }
    }
  }
  double* out = G.items.scratch + int64_t(item) * kGenItemValues;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const double t = group_sum<64>(acc[i]);
    if (lane == (i & 63)) out[i] = t;
  }
}
// blocks(j) = the items' partial triangles (+ D^2), mirrored: a wave per block, lane l adds the items l, l + 64, ... entry by entry,
// the lanes' sums combine by butterfly (a fixed tree: deterministic)
template <int MAXC>
__global__ __launch_bounds__(kB) void gen_block_items_combine_kernel(GenStructure G, const int64_t* __restrict__ off, int first, const double* __restrict__ D,
                                                                     double* __restrict__ blocks) {
  constexpr int NT = MAXC * (MAXC + 1) / 2;
  const int q = blockIdx.x * (kB / 64) + (threadIdx.x >> 6);
  if (q >= G.items.nblocks) return;
  const int lane = threadIdx.x & 63;
  const int j = G.items.first_block + q;
  const int n = G.csz[j];
  double acc[NT];
#pragma unroll
  for (int e = 0; e < NT; ++e) acc[e] = 0.0;
  for (int it = G.items.block_ptr[q] + lane; it < G.items.block_ptr[q + 1]; it += 64) {
    const double* p = G.items.scratch + int64_t(it) * kGenItemValues;
#pragma unroll
    for (int e = 0; e < NT; ++e) acc[e] += p[e];
  }
  double* o = blocks + (off[j - first] - off[0]);
  int e = 0;
#pragma unroll
  for (int a = 0; a < MAXC; ++a)
#pragma unroll
    for (int b = a; b < MAXC; ++b) {
      double s = group_sum<64>(acc[e]);
      if (a < n && b < n && lane == (e & 63)) {
        if (D && a == b) { const double d = D[G.cpos[j] + a]; s += d * d; }
        o[a * n + b] = s;
        o[b * n + a] = s;
      }
      ++e;
    }
}

// L lanes per group: dispatch of a grouped kernel over the compiled group widths
#define GEN_DISPATCH_L(L_, CALL)            \
  switch (L_) {                             \
    case 4: { constexpr int L = 4; CALL; } break;   \
    case 8: { constexpr int L = 8; CALL; } break;   \
    case 16: { constexpr int L = 16; CALL; } break; \
    case 32: { constexpr int L = 32; CALL; } break; \
    default: { constexpr int L = 64; CALL; } break; \
  }
template <int L>
static void launch_left_grouped(const GenStructure& G, const double* values, int part, int j0, int nb, int ybase, int max_c, const double* x,
                                double* y, const int* status, hipStream_t s) {
  const dim3 grid(blocks_for(int64_t(nb) * L));
  if (max_c <= 4) hipLaunchKernelGGL((gen_left_multiply_grouped_kernel<L, 4>), grid, dim3(kB), 0, s, G, values, part, j0, nb, ybase, x, y, status);
  else if (max_c <= 10) hipLaunchKernelGGL((gen_left_multiply_grouped_kernel<L, 10>), grid, dim3(kB), 0, s, G, values, part, j0, nb, ybase, x, y, status);
  else hipLaunchKernelGGL((gen_left_multiply_grouped_kernel<L, kMaxGenericBlock>), grid, dim3(kB), 0, s, G, values, part, j0, nb, ybase, x, y, status);
}
hipError_t LaunchGenLeftMultiply(const GenStructure& G, const double* values, int part, const double* x, double* y,
                                 const int* status, hipStream_t s) {
  const int n = part == kAll ? G.num_cols : (part == kE ? G.nce : G.ncf);
  if (n <= 0) return hipSuccess;
  if (G.lanes_all == 0) {   // no hints: the thread-per-scalar kernel
    hipLaunchKernelGGL(gen_left_multiply_kernel, dim3(blocks_for(n)), dim3(kB), 0, s, G, values, part, x, y, status);
    return hipGetLastError();
  }
  // the E blocks (many, a few cells each) and the F blocks (few, many cells each) get their own group widths
  const int ybase = part == kF ? G.nce : 0;
  if (part != kF && G.nelim > 0) { GEN_DISPATCH_L(G.lanes_e, (launch_left_grouped<L>(G, values, part, 0, G.nelim, ybase, G.max_csz_e, x, y, status, s))) }
  if (part != kE && G.ncb > G.nelim) {
    if (G.items.count > 0 && G.items.first_block == G.nelim) {   // heavy blocks: a wave per item, then the items of a block in order
      const dim3 grid((G.items.count + kB / 64 - 1) / (kB / 64));
      const int mc = G.nelim > 0 ? G.max_csz_f : G.max_csz;
      if (mc <= 4 && G.max_rsz <= 4) hipLaunchKernelGGL((gen_left_multiply_items_kernel<4, 4>), grid, dim3(kB), 0, s, G, values, part, x, status);
      else if (mc <= 10 && G.max_rsz <= 2) hipLaunchKernelGGL((gen_left_multiply_items_kernel<10, 2>), grid, dim3(kB), 0, s, G, values, part, x, status);
      else if (mc <= 10) hipLaunchKernelGGL((gen_left_multiply_items_kernel<10, kMaxGenericBlock>), grid, dim3(kB), 0, s, G, values, part, x, status);
      else hipLaunchKernelGGL((gen_left_multiply_items_kernel<kMaxGenericBlock, kMaxGenericBlock>), grid, dim3(kB), 0, s, G, values, part, x, status);
      hipLaunchKernelGGL(gen_left_multiply_combine_kernel, dim3((G.items.nblocks + kB / 64 - 1) / (kB / 64)), dim3(kB), 0, s, G, ybase, y, status);
    } else if (G.nelim > 0) { GEN_DISPATCH_L(G.lanes_f, (launch_left_grouped<L>(G, values, part, G.nelim, G.ncb - G.nelim, ybase, G.max_csz_f, x, y, status, s))) }
    else { GEN_DISPATCH_L(G.lanes_all, (launch_left_grouped<L>(G, values, part, 0, G.ncb, ybase, G.max_csz, x, y, status, s))) }
  }
  return hipGetLastError();
}
template <int L>
static void launch_chunk_project(const GenStructure& G, const double* values, const double* ete_inv, double* t_rows, int update_t, double* x_e,
                                 const int* status, hipStream_t s) {
  const dim3 grid(blocks_for(int64_t(G.nelim) * L));
  if (G.max_csz_e <= 4) hipLaunchKernelGGL((gen_chunk_project_kernel<L, 4>), grid, dim3(kB), 0, s, G, values, ete_inv, t_rows, update_t, x_e, status);
  else hipLaunchKernelGGL((gen_chunk_project_kernel<L, 10>), grid, dim3(kB), 0, s, G, values, ete_inv, t_rows, update_t, x_e, status);
}
hipError_t LaunchGenChunkProject(const GenStructure& G, const double* values, const double* ete_inv, double* t_rows, int update_t,
                                 double* x_e, const int* status, hipStream_t s) {
  if (G.nelim <= 0) return hipSuccess;
  if (G.lanes_chunk == 0 || !G.chunk_start || G.max_csz_e > 10) return hipErrorNotSupported;
  GEN_DISPATCH_L(G.lanes_chunk, (launch_chunk_project<L>(G, values, ete_inv, t_rows, update_t, x_e, status, s)))
  return hipGetLastError();
}
template <int L>
static void launch_chunk_sx(const GenStructure& G, const double* values, const double* ete_inv, const double* x_f, double* z_rows, const int* status,
                            hipStream_t s) {
  const dim3 grid(blocks_for(int64_t(G.nelim) * L));
  if (G.max_rsz <= 2) hipLaunchKernelGGL((gen_chunk_sx_kernel<L, 2, 4, 10>), grid, dim3(kB), 0, s, G, values, ete_inv, x_f, z_rows, status);
  else hipLaunchKernelGGL((gen_chunk_sx_kernel<L, 4, 4, 10>), grid, dim3(kB), 0, s, G, values, ete_inv, x_f, z_rows, status);
}
hipError_t LaunchGenChunkSx(const GenStructure& G, const double* values, const double* ete_inv, const double* x_f, double* z_rows,
                            const int* status, hipStream_t s) {
  if (G.nelim <= 0) return hipSuccess;
  if (G.lanes_chunk == 0 || !G.chunk_start || G.max_csz_e > 4 || G.max_csz_f > 10 || G.max_rsz > 4) return hipErrorNotSupported;
  GEN_DISPATCH_L(G.lanes_chunk, (launch_chunk_sx<L>(G, values, ete_inv, x_f, z_rows, status, s)))
  return hipGetLastError();
}
hipError_t LaunchGenRightMultiplyFrom(const GenStructure& G, const double* values, int part, int first_row, const double* x, double* y,
                                      const int* status, hipStream_t s) {
  const int n = G.num_rows - first_row;
  if (n > 0) hipLaunchKernelGGL(gen_right_multiply_kernel, dim3(blocks_for(n)), dim3(kB), 0, s, G, values, part, x, y, status, first_row);
  return hipGetLastError();
}
template <int L>
static void launch_block_diagonal_grouped(const GenStructure& G, const double* values, int part, int j0, int nb, const int64_t* off, int first,
                                          int max_c, const double* D, double* blocks, hipStream_t s) {
  const dim3 grid(blocks_for(int64_t(nb) * L));
  if (max_c <= 4) hipLaunchKernelGGL((gen_block_diagonal_grouped_kernel<L, 4>), grid, dim3(kB), 0, s, G, values, part, j0, nb, off, first, D, blocks);
  else hipLaunchKernelGGL((gen_block_diagonal_grouped_kernel<L, 10>), grid, dim3(kB), 0, s, G, values, part, j0, nb, off, first, D, blocks);
}
hipError_t LaunchGenSquaredColumnNorm(const GenStructure& G, const double* values, double* x, hipStream_t s) {
  if (G.num_cols <= 0) return hipSuccess;
  if (G.lanes_all == 0) {
    hipLaunchKernelGGL(gen_squared_column_norm_kernel, dim3(blocks_for(G.num_cols)), dim3(kB), 0, s, G, values, x);
    return hipGetLastError();
  }
  // the grouped / itemized left-multiply kernels with the values in place of the vector (x == nullptr): x[col] = sum of A[r][col]^2
  if (hipError_t e = hipMemsetAsync(x, 0, sizeof(double) * size_t(G.num_cols), s); e != hipSuccess) return e;
  return LaunchGenLeftMultiply(G, values, kAll, nullptr, x, nullptr, s);
}
hipError_t LaunchGenScaleColumns(const GenStructure& G, double* values, const double* scale, hipStream_t s) {
  if (G.num_rows > 0) hipLaunchKernelGGL(gen_scale_columns_kernel, dim3(blocks_for(G.num_rows)), dim3(kB), 0, s, G, values, scale);
  return hipGetLastError();
}
hipError_t LaunchGenInvertBlocks(const GenStructure& G, int first_block, int nblocks, const int64_t* diag_off, double* blocks,
                                 int* fail_flag, hipStream_t s) {
  if (nblocks > 0) hipLaunchKernelGGL(gen_invert_blocks_kernel, dim3((nblocks + 63) / 64), dim3(64), 0, s, G, first_block, nblocks, diag_off, blocks, fail_flag);
  return hipGetLastError();
}
hipError_t LaunchGenBlockDiagonalApply(const GenStructure& G, int first_block, int nblocks, const int64_t* diag_off,
                                       const double* blocks, const double* x, double* y, const int* status, hipStream_t s) {
  (void)nblocks;
  // scalar range covered by the blocks: from cpos[first_block] to the end of the E part
  // (first_block == 0 with the E list), of the F part, or of everything.
  int col_begin, ncols;
  if (first_block == 0 && diag_off == G.diag_off_e) { col_begin = 0; ncols = G.nce; }
  else if (first_block == G.nelim && diag_off == G.diag_off_f) { col_begin = G.nce; ncols = G.ncf; }
  else { col_begin = 0; ncols = G.num_cols; }
  if (ncols > 0) hipLaunchKernelGGL(gen_block_diagonal_apply_kernel, dim3(blocks_for(ncols)), dim3(kB), 0, s, G, first_block, col_begin, ncols, diag_off, blocks, x, y, status);
  return hipGetLastError();
}
hipError_t LaunchGenSchurDense(const GenStructure& G, const double* values, const double* ete_inv, const double* D, double* lhs,
                               hipStream_t s) {
  const int64_t n = int64_t(G.ncf) * G.ncf;
  if (n > 0) hipLaunchKernelGGL(gen_schur_dense_kernel, dim3(blocks_for(n)), dim3(kB), 0, s, G, values, ete_inv, D, lhs);
  return hipGetLastError();
}

hipError_t LaunchGenSymmetrizeDense(const GenStructure& G, double* lhs, hipStream_t s) {
  const int64_t n = int64_t(G.ncf) * G.ncf;
  if (n > 0) hipLaunchKernelGGL(gen_symmetrize_dense_kernel, dim3(blocks_for(n)), dim3(kB), 0, s, G, lhs);
  return hipGetLastError();
}
hipError_t LaunchGenDenseSymv(const double* S, int n, const double* x, double* y, const int* status, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(gen_dense_symv_kernel, dim3((n + kB / 64 - 1) / (kB / 64)), dim3(kB), 0, s, S, n, x, y, status);
  return hipGetLastError();
}
hipError_t LaunchGenExtractDiagBlocks(const GenStructure& G, const double* S, const int64_t* diag_off_f, double* blocks, hipStream_t s) {
  if (G.ncf > 0) hipLaunchKernelGGL(gen_extract_diag_blocks_kernel, dim3(blocks_for(G.ncf)), dim3(kB), 0, s, G, S, diag_off_f, blocks);
  return hipGetLastError();
}

hipError_t LaunchAddBlockDiagonalSquares(const GenStructure& G, int first_block, int nblocks, const int64_t* diag_off,
                                         const double* D, double* blocks, hipStream_t s) {
  if (nblocks > 0) {
    const int ncols = first_block == 0 && nblocks == G.ncb ? G.num_cols : (first_block == G.nelim ? G.ncf : G.nce);
    hipLaunchKernelGGL(gen_add_diag_squares_kernel, dim3(blocks_for(ncols)), dim3(kB), 0, s, G, first_block, nblocks, diag_off, D, blocks);
  }
  return hipGetLastError();
}

// The two launchers below need the total entry count, which lives on the host side of
// the structure; solver.hip passes it.
hipError_t LaunchGenBlockDiagonal(const GenStructure& G, const double* values, int part, const double* D, double* blocks,
                                   int64_t total, hipStream_t s) {
  if (total <= 0) return hipSuccess;
  const int max_c = part == kE ? G.max_csz_e : (part == kF ? G.max_csz_f : G.max_csz);
  if (G.lanes_all == 0 || max_c > 10) {   // no hints, or blocks wider than the grouped kernel's register triangle: one thread per entry
    hipLaunchKernelGGL(gen_block_diagonal_kernel, dim3(blocks_for(total)), dim3(kB), 0, s, G, values, part, D, blocks);
    return hipGetLastError();
  }
  const int64_t* off = part == kAll ? G.diag_off_all : (part == kE ? G.diag_off_e : G.diag_off_f);
  const int first = part == kF ? G.nelim : 0;
  if (part != kF && G.nelim > 0) { GEN_DISPATCH_L(G.lanes_e, (launch_block_diagonal_grouped<L>(G, values, part, 0, G.nelim, off, first, G.max_csz_e, D, blocks, s))) }
  if (part != kE && G.ncb > G.nelim) {
    if (G.items.count > 0 && G.items.first_block == G.nelim) {
      hipLaunchKernelGGL((gen_block_items_kernel<4, 10>), dim3((G.items.count + kB / 64 - 1) / (kB / 64)), dim3(kB), 0, s, G, values, part, 0, nullptr);
      hipLaunchKernelGGL((gen_block_items_combine_kernel<10>), dim3((G.items.nblocks + kB / 64 - 1) / (kB / 64)), dim3(kB), 0, s, G, off, first, D, blocks);
    } else if (G.nelim > 0) { GEN_DISPATCH_L(G.lanes_f, (launch_block_diagonal_grouped<L>(G, values, part, G.nelim, G.ncb - G.nelim, off, first, G.max_csz_f, D, blocks, s))) }
    else { GEN_DISPATCH_L(G.lanes_all, (launch_block_diagonal_grouped<L>(G, values, part, 0, G.ncb, off, first, G.max_csz, D, blocks, s))) }
  }
  return hipGetLastError();
}
template <int L>
static void launch_schur_jacobi_grouped(const GenStructure& G, const double* values, const double* ete_inv, const double* D, int add_f_diag,
                                        double* blocks, hipStream_t s) {
  const int nf = G.ncb - G.nelim;
  hipLaunchKernelGGL((gen_schur_jacobi_grouped_kernel<L, 4, 10>), dim3(blocks_for(int64_t(nf) * L)), dim3(kB), 0, s, G, values, ete_inv, D, add_f_diag, blocks);
}
hipError_t LaunchGenSchurJacobi(const GenStructure& G, const double* values, const double* ete_inv, const double* D,
                                 int add_f_diag, double* blocks, int64_t total, hipStream_t s) {
  if (total <= 0) return hipSuccess;
  if (G.lanes_f == 0 || G.max_csz_e > 4 || G.max_csz_f > 10) {
    hipLaunchKernelGGL(gen_schur_jacobi_kernel, dim3(blocks_for(total)), dim3(kB), 0, s, G, values, ete_inv, D, add_f_diag, blocks);
    return hipGetLastError();
  }
  if (G.items.count > 0 && G.items.first_block == G.nelim) {
    hipLaunchKernelGGL((gen_block_items_kernel<4, 10>), dim3((G.items.count + kB / 64 - 1) / (kB / 64)), dim3(kB), 0, s, G, values, int(kF), 1, ete_inv);
    hipLaunchKernelGGL((gen_block_items_combine_kernel<10>), dim3((G.items.nblocks + kB / 64 - 1) / (kB / 64)), dim3(kB), 0, s, G, G.diag_off_f, G.nelim,
                       add_f_diag ? D : nullptr, blocks);
    return hipGetLastError();
  }
  GEN_DISPATCH_L(G.lanes_f, (launch_schur_jacobi_grouped<L>(G, values, ete_inv, D, add_f_diag, blocks, s)))
  return hipGetLastError();
}

}  // namespace chip
