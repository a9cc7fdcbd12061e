// ceres_oracle.cc — CPU restatement of the reference's LM linear-solve path.
//
// TEST INFRASTRUCTURE ONLY (see ceres_oracle.h for the rules and the parity
// status).  Citations: "I/" = /root/reference/internal/ceres/.
//
// Build: see oracle/Makefile (g++ -O3 -march=x86-64-v3 -fopenmp -shared).

#include "ceres_oracle.h"

#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <vector>

namespace {

int g_threads = 1;

// ParallelSetZero / ParallelAssign (I/parallel_vector_ops.h: the reference splits vectors above a minimum block size
// over its thread pool; used by ImplicitSchurComplement for every temporary, I/implicit_schur_complement.cc:109-158, :212-241)
constexpr long kMinParallelVectorSize = 1 << 16;
void parallel_set_zero(double* x, long n) {
  if (g_threads == 1 || n < kMinParallelVectorSize) { std::fill(x, x + n, 0.0); return; }
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (long i = 0; i < n; ++i) x[i] = 0.0;
}
void parallel_negate(double* x, long n) {
  if (g_threads == 1 || n < kMinParallelVectorSize) { for (long i = 0; i < n; ++i) x[i] = -x[i]; return; }
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (long i = 0; i < n; ++i) x[i] = -x[i];
}
// y = a - y
void parallel_subtract_from(const double* a, double* y, long n) {
  if (g_threads == 1 || n < kMinParallelVectorSize) { for (long i = 0; i < n; ++i) y[i] = a[i] - y[i]; return; }
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (long i = 0; i < n; ++i) y[i] = a[i] - y[i];
}

using Vec = std::vector<double>;

// --------------------------------------------------------------------------
// Tiny dense kernels.  Semantics of I/small_blas.h:164-555: row-major blocks,
// "sign" = kOperation (+1: +=, -1: -=, 0: =).  One generic loop nest each; the
// reference's static-size/unrolled variants compute the same sums.
// --------------------------------------------------------------------------
// Each kernel exists once as a loop nest over compile-time sizes (R, C, ... > 0) or run-time sizes
// (template argument 0); the public entry dispatches the sizes bundle adjustment produces (2x3, 2x9,
// 3x3, 9x9, ...) to the compile-time instances, as the reference does with its template
// specialisations (I/small_blas.h, I/schur_eliminator.cc).  Same loop order, same sums.
template <int R, int C>
inline void mat_vec_t(const double* A, int r_, int c_, const double* x, double* y, int sign) {
  const int r = R ? R : r_, c = C ? C : c_;
  for (int i = 0; i < r; ++i) {
    double s = 0;
    for (int j = 0; j < c; ++j) s += A[i * c + j] * x[j];
    if (sign > 0) y[i] += s; else if (sign < 0) y[i] -= s; else y[i] = s;
  }
}
inline void mat_vec(const double* A, int r, int c, const double* x, double* y, int sign) {
  if (r == 2 && c == 9) return mat_vec_t<2, 9>(A, r, c, x, y, sign);
  if (r == 2 && c == 3) return mat_vec_t<2, 3>(A, r, c, x, y, sign);
  if (r == 9 && c == 9) return mat_vec_t<9, 9>(A, r, c, x, y, sign);
  if (r == 3 && c == 3) return mat_vec_t<3, 3>(A, r, c, x, y, sign);
  mat_vec_t<0, 0>(A, r, c, x, y, sign);
}
template <int R, int C>
inline void mat_t_vec_t(const double* A, int r_, int c_, const double* x, double* y, int sign) {
  const int r = R ? R : r_, c = C ? C : c_;
  for (int j = 0; j < c; ++j) {
    double s = 0;
    for (int i = 0; i < r; ++i) s += A[i * c + j] * x[i];
    if (sign > 0) y[j] += s; else if (sign < 0) y[j] -= s; else y[j] = s;
  }
}
inline void mat_t_vec(const double* A, int r, int c, const double* x, double* y, int sign) {
  if (r == 2 && c == 9) return mat_t_vec_t<2, 9>(A, r, c, x, y, sign);
  if (r == 2 && c == 3) return mat_t_vec_t<2, 3>(A, r, c, x, y, sign);
  mat_t_vec_t<0, 0>(A, r, c, x, y, sign);
}
// C[r0.., c0..] (op)= A^T B, A is ra x ca, B is ra x cb, C has row stride ldc.
template <int RA, int CA, int CB>
inline void mat_t_mat_t(const double* A, int ra_, int ca_, const double* B, int cb_, double* C, int r0, int c0, int ldc,
                        int sign) {
  const int ra = RA ? RA : ra_, ca = CA ? CA : ca_, cb = CB ? CB : cb_;
  for (int i = 0; i < ca; ++i)
    for (int j = 0; j < cb; ++j) {
      double s = 0;
      for (int k = 0; k < ra; ++k) s += A[k * ca + i] * B[k * cb + j];
      double& d = C[(r0 + i) * ldc + c0 + j];
      if (sign > 0) d += s; else if (sign < 0) d -= s; else d = s;
    }
}
inline void mat_t_mat(const double* A, int ra, int ca, const double* B, int cb, double* C, int r0,
                      int c0, int ldc, int sign) {
  if (ra == 2) {
    if (ca == 9 && cb == 9) return mat_t_mat_t<2, 9, 9>(A, ra, ca, B, cb, C, r0, c0, ldc, sign);
    if (ca == 3 && cb == 3) return mat_t_mat_t<2, 3, 3>(A, ra, ca, B, cb, C, r0, c0, ldc, sign);
    if (ca == 3 && cb == 9) return mat_t_mat_t<2, 3, 9>(A, ra, ca, B, cb, C, r0, c0, ldc, sign);
    if (ca == 9 && cb == 3) return mat_t_mat_t<2, 9, 3>(A, ra, ca, B, cb, C, r0, c0, ldc, sign);
  }
  if (ra == 3 && ca == 9 && cb == 9) return mat_t_mat_t<3, 9, 9>(A, ra, ca, B, cb, C, r0, c0, ldc, sign);
  mat_t_mat_t<0, 0, 0>(A, ra, ca, B, cb, C, r0, c0, ldc, sign);
}
// C[r0.., c0..] (op)= A B, A is ra x ca, B is ca x cb.
template <int RA, int CA, int CB>
inline void mat_mat_t(const double* A, int ra_, int ca_, const double* B, int cb_, double* C, int r0, int c0, int ldc,
                      int sign) {
  const int ra = RA ? RA : ra_, ca = CA ? CA : ca_, cb = CB ? CB : cb_;
  for (int i = 0; i < ra; ++i)
    for (int j = 0; j < cb; ++j) {
      double s = 0;
      for (int k = 0; k < ca; ++k) s += A[i * ca + k] * B[k * cb + j];
      double& d = C[(r0 + i) * ldc + c0 + j];
      if (sign > 0) d += s; else if (sign < 0) d -= s; else d = s;
    }
}
inline void mat_mat(const double* A, int ra, int ca, const double* B, int cb, double* C, int r0,
                    int c0, int ldc, int sign) {
  if (ra == 3 && ca == 3 && cb == 9) return mat_mat_t<3, 3, 9>(A, ra, ca, B, cb, C, r0, c0, ldc, sign);
  if (ra == 3 && ca == 3 && cb == 3) return mat_mat_t<3, 3, 3>(A, ra, ca, B, cb, C, r0, c0, ldc, sign);
  if (ra == 2 && ca == 3 && cb == 3) return mat_mat_t<2, 3, 3>(A, ra, ca, B, cb, C, r0, c0, ldc, sign);
  if (ra == 2 && ca == 3 && cb == 9) return mat_mat_t<2, 3, 9>(A, ra, ca, B, cb, C, r0, c0, ldc, sign);
  if (ra == 9 && ca == 3 && cb == 9) return mat_mat_t<9, 3, 9>(A, ra, ca, B, cb, C, r0, c0, ldc, sign);
  mat_mat_t<0, 0, 0>(A, ra, ca, B, cb, C, r0, c0, ldc, sign);
}

// In-place inverse of an SPD matrix from its UPPER triangle via Cholesky and a
// solve against the identity: selfadjointView<Upper>().llt().solve(I) of
// I/invert_psd_matrix.h:51-83, I/block_random_access_diagonal_matrix.cc:90-100,
// I/implicit_schur_complement.cc:179-204.
int invert_spd_upper(int n, double* a) {
  std::vector<double> L(n * n, 0.0), inv(n * n, 0.0), col(n);
  for (int j = 0; j < n; ++j) {
    double d = a[j * n + j];
    for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
    if (!(d > 0)) return 1;
    d = std::sqrt(d);
    L[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = a[j * n + i];  // upper triangle entry (j,i) == (i,j)
      for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = s / d;
    }
  }
  for (int e = 0; e < n; ++e) {
    for (int i = 0; i < n; ++i) {  // L w = e_e
      double s = (i == e) ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) s -= L[i * n + k] * col[k];
      col[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {  // L^T v = w
      double s = col[i];
      for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * col[k];
      col[i] = s / L[i * n + i];
    }
    for (int i = 0; i < n; ++i) inv[i * n + e] = col[i];
  }
  std::memcpy(a, inv.data(), sizeof(double) * n * n);
  return 0;
}

// General inverse of a small matrix by cofactors: what Eigen's fixed-size
// inverse() does for sizes < 5, which SchurEliminator uses when the E block size
// is static (I/invert_psd_matrix.h:62-64).  Sizes 1..3 closed form; larger
// sizes fall through to the Cholesky route (differs only in rounding).
int invert_small(int n, double* a) {
  if (n == 1) { a[0] = 1.0 / a[0]; return 0; }
  if (n == 2) {
    const double det = a[0] * a[3] - a[1] * a[2];
    const double i0 = a[3] / det, i1 = -a[1] / det, i2 = -a[2] / det, i3 = a[0] / det;
    a[0] = i0; a[1] = i1; a[2] = i2; a[3] = i3;
    return 0;
  }
  if (n == 3) {
    double c[9];
    c[0] = a[4] * a[8] - a[5] * a[7];
    c[1] = a[2] * a[7] - a[1] * a[8];
    c[2] = a[1] * a[5] - a[2] * a[4];
    c[3] = a[5] * a[6] - a[3] * a[8];
    c[4] = a[0] * a[8] - a[2] * a[6];
    c[5] = a[2] * a[3] - a[0] * a[5];
    c[6] = a[3] * a[7] - a[4] * a[6];
    c[7] = a[1] * a[6] - a[0] * a[7];
    c[8] = a[0] * a[4] - a[1] * a[3];
    const double det = a[0] * c[0] + a[1] * c[3] + a[2] * c[6];
    for (int i = 0; i < 9; ++i) a[i] = c[i] / det;
    return 0;
  }
  return invert_spd_upper(n, a);
}

}  // namespace

// --------------------------------------------------------------------------
// Structure (I/block_structure.h:52-182) with the derived data the reference
// computes in BlockSparseMatrix's ctor (transpose structure,
// I/block_sparse_matrix.cc:178-216,784-808), PartitionedMatrixView's ctor
// (num_row_blocks_e, I/partitioned_matrix_view_impl.h:47-105) and
// SchurEliminator::Init (chunks, I/schur_eliminator_impl.h:87-181).
// --------------------------------------------------------------------------
struct oracle_matrix {
  int nrb = 0, ncb = 0, nelim = 0;
  std::vector<int> rsz, rpos, csz, cpos, rptr, ccol, cval;
  int num_rows = 0, num_cols = 0, num_cols_e = 0, num_cols_f = 0, num_row_blocks_e = 0;
  int64_t nnz = 0;
  // transpose: for column block j, entries tptr[j]..tptr[j+1): (row block, cell index)
  std::vector<int> tptr, trow, tcell;
  // chunks
  struct Chunk { int start, size; std::map<int, int> layout; };
  std::vector<Chunk> chunks;
  int buffer_size = 1, uneliminated_row_begins = 0;
  std::vector<int> lhs_row_layout;  // position of each F block in the reduced system
  std::vector<int> diag_offset_f;   // offset of each F block's dense block in a block-diagonal store
  std::vector<int> diag_offset_all; // same over all column blocks
  std::vector<int> diag_offset_e;
  // F column blocks cut into contiguous ranges of nearly equal cell counts: the work units of the column-major loops
  // (camera degrees are heavily skewed; the reference balances its ParallelFor the same way, by cumulative nnz,
  // I/partitioned_matrix_view_impl.h:89-103)
  std::vector<int> f_ranges;
  std::vector<int> col_ranges;  // all column blocks cut into runs of about equal numbers of non-zeros (as the reference's
                                // ParallelFor over the transpose structure partitions by cumulative_nnz, I/block_sparse_matrix.cc:293-325)
};

extern "C" {

void oracle_set_num_threads(int n) { g_threads = std::max(1, n); }
int oracle_get_num_threads(void) { return g_threads; }

oracle_matrix* oracle_matrix_create(const oracle_block_structure* bs, int num_eliminate_blocks) {
  auto* m = new oracle_matrix;
  m->nrb = bs->num_row_blocks;
  m->ncb = bs->num_col_blocks;
  m->nelim = num_eliminate_blocks;
  m->rsz.assign(bs->row_block_size, bs->row_block_size + m->nrb);
  m->rpos.assign(bs->row_block_pos, bs->row_block_pos + m->nrb);
  m->csz.assign(bs->col_block_size, bs->col_block_size + m->ncb);
  m->cpos.assign(bs->col_block_pos, bs->col_block_pos + m->ncb);
  m->rptr.assign(bs->row_cell_ptr, bs->row_cell_ptr + m->nrb + 1);
  const int ncells = m->rptr[m->nrb];
  m->ccol.assign(bs->cell_col_block, bs->cell_col_block + ncells);
  m->cval.assign(bs->cell_value_pos, bs->cell_value_pos + ncells);
  for (int i = 0; i < m->nrb; ++i) {
    m->num_rows += m->rsz[i];
    for (int k = m->rptr[i]; k < m->rptr[i + 1]; ++k)
      m->nnz += int64_t(m->rsz[i]) * m->csz[m->ccol[k]];
  }
  int off_all = 0, off_e = 0, off_f = 0, lhs_rows = 0;
  for (int j = 0; j < m->ncb; ++j) {
    m->num_cols += m->csz[j];
    m->diag_offset_all.push_back(off_all);
    off_all += m->csz[j] * m->csz[j];
    if (j < m->nelim) {
      m->num_cols_e += m->csz[j];
      m->diag_offset_e.push_back(off_e);
      off_e += m->csz[j] * m->csz[j];
    } else {
      m->num_cols_f += m->csz[j];
      m->diag_offset_f.push_back(off_f);
      off_f += m->csz[j] * m->csz[j];
      m->lhs_row_layout.push_back(lhs_rows);
      lhs_rows += m->csz[j];
    }
  }
  m->diag_offset_all.push_back(off_all);
  m->diag_offset_e.push_back(off_e);
  m->diag_offset_f.push_back(off_f);
  // Rows whose first cell is an E block.
  for (int i = 0; i < m->nrb; ++i)
    if (m->rptr[i] < m->rptr[i + 1] && m->ccol[m->rptr[i]] < m->nelim) ++m->num_row_blocks_e;
  // Transpose structure (counting sort by column block keeps row order).
  m->tptr.assign(m->ncb + 1, 0);
  for (int k = 0; k < ncells; ++k) ++m->tptr[m->ccol[k] + 1];
  for (int j = 0; j < m->ncb; ++j) m->tptr[j + 1] += m->tptr[j];
  m->trow.resize(ncells);
  m->tcell.resize(ncells);
  std::vector<int> cur(m->tptr.begin(), m->tptr.end() - 1);
  for (int i = 0; i < m->nrb; ++i)
    for (int k = m->rptr[i]; k < m->rptr[i + 1]; ++k) {
      const int p = cur[m->ccol[k]]++;
      m->trow[p] = i;
      m->tcell[p] = k;
    }
  {
    const int nf = m->ncb - m->nelim;
    const int want = std::max(1, std::min(nf, 1024));
    const int64_t total = m->tptr[m->ncb] - m->tptr[m->nelim];
    m->f_ranges.assign(1, m->nelim);
    for (int k = 1; k < want; ++k) {
      const int64_t target = m->tptr[m->nelim] + total * k / want;
      int j = int(std::lower_bound(m->tptr.begin() + m->nelim, m->tptr.begin() + m->ncb, target) - m->tptr.begin());
      j = std::max(j, m->f_ranges.back());
      if (j > m->f_ranges.back() && j < m->ncb) m->f_ranges.push_back(j);
    }
    m->f_ranges.push_back(m->ncb);
  }
  {
    std::vector<int64_t> nnz(m->ncb + 1, 0);
    for (int j = 0; j < m->ncb; ++j) {
      int64_t n = 0;
      for (int t = m->tptr[j]; t < m->tptr[j + 1]; ++t) n += int64_t(m->rsz[m->trow[t]]) * m->csz[j];
      nnz[j + 1] = nnz[j] + n;
    }
    const int want = std::max(1, std::min(m->ncb, 4096));
    m->col_ranges.assign(1, 0);
    for (int k = 1; k < want; ++k) {
      const int64_t target = nnz[m->ncb] * k / want;
      int j = int(std::lower_bound(nnz.begin(), nnz.end(), target) - nnz.begin());
      j = std::max(j, m->col_ranges.back());
      if (j > m->col_ranges.back() && j < m->ncb) m->col_ranges.push_back(j);
    }
    m->col_ranges.push_back(m->ncb);
  }
  // Chunks: maximal runs of rows sharing their first (E) block.
  if (m->nelim > 0) {
    int r = 0;
    while (r < m->nrb) {
      if (m->rptr[r] == m->rptr[r + 1]) break;
      const int e_id = m->ccol[m->rptr[r]];
      if (e_id >= m->nelim) break;
      oracle_matrix::Chunk ch{r, 0, {}};
      int buffer = 0;
      const int es = m->csz[e_id];
      while (r + ch.size < m->nrb) {
        const int row = r + ch.size;
        if (m->rptr[row] == m->rptr[row + 1] || m->ccol[m->rptr[row]] != e_id) break;
        for (int k = m->rptr[row] + 1; k < m->rptr[row + 1]; ++k) {
          if (ch.layout.emplace(m->ccol[k], buffer).second) buffer += es * m->csz[m->ccol[k]];
        }
        m->buffer_size = std::max(m->buffer_size, buffer);
        ++ch.size;
      }
      r += ch.size;
      m->chunks.push_back(std::move(ch));
    }
    m->uneliminated_row_begins = m->chunks.empty() ? 0 : m->chunks.back().start + m->chunks.back().size;
  }
  return m;
}

void oracle_matrix_destroy(oracle_matrix* m) { delete m; }
int oracle_matrix_num_rows(const oracle_matrix* m) { return m->num_rows; }
int oracle_matrix_num_cols(const oracle_matrix* m) { return m->num_cols; }
int oracle_matrix_num_cols_e(const oracle_matrix* m) { return m->num_cols_e; }
int oracle_matrix_num_cols_f(const oracle_matrix* m) { return m->num_cols_f; }
int oracle_matrix_num_row_blocks_e(const oracle_matrix* m) { return m->num_row_blocks_e; }
int64_t oracle_matrix_num_nonzeros(const oracle_matrix* m) { return m->nnz; }

// I/detect_structure.cc:39-121: sizes taken from rows with an E block only; a
// size becomes -1 (Eigen::Dynamic) as soon as two rows disagree.
void oracle_detect_structure(const oracle_matrix* m, int* row, int* e, int* f) {
  *row = 0; *e = 0; *f = 0;
  for (int i = 0; i < m->nrb; ++i) {
    if (m->rptr[i] == m->rptr[i + 1]) continue;
    const int first = m->ccol[m->rptr[i]];
    if (first >= m->nelim) break;
    auto upd = [](int* cur, int v) { if (*cur == 0) *cur = v; else if (*cur != -1 && *cur != v) *cur = -1; };
    upd(row, m->rsz[i]);
    upd(e, m->csz[first]);
    if (m->rptr[i + 1] - m->rptr[i] > 1) {
      if (*f == 0) *f = m->csz[m->ccol[m->rptr[i] + 1]];
      for (int k = m->rptr[i] + 1; k < m->rptr[i + 1] && *f != -1; ++k)
        if (m->csz[m->ccol[k]] != *f) *f = -1;
    }
    if (*row == -1 && *e == -1 && *f == -1) break;
  }
}

// y += A x.  I/block_sparse_matrix.cc:239-274 (parallel over row blocks).
void oracle_right_multiply(const oracle_matrix* m, const double* v, const double* x, double* y) {
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int i = 0; i < m->nrb; ++i)
    for (int k = m->rptr[i]; k < m->rptr[i + 1]; ++k) {
      const int j = m->ccol[k];
      mat_vec(v + m->cval[k], m->rsz[i], m->csz[j], x + m->cpos[j], y + m->rpos[i], 1);
    }
}

// y += A^T x.  Single thread: row order scatter (I/block_sparse_matrix.cc:327-349);
// multi-thread: column-block order through the transpose structure (:278-325).
void oracle_left_multiply(const oracle_matrix* m, const double* v, const double* x, double* y) {
  if (g_threads == 1) {
    for (int i = 0; i < m->nrb; ++i)
      for (int k = m->rptr[i]; k < m->rptr[i + 1]; ++k) {
        const int j = m->ccol[k];
        mat_t_vec(v + m->cval[k], m->rsz[i], m->csz[j], x + m->rpos[i], y + m->cpos[j], 1);
      }
    return;
  }
  const int n_ranges = int(m->col_ranges.size()) - 1;
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 1)
  for (int rg = 0; rg < n_ranges; ++rg)
    for (int j = m->col_ranges[rg]; j < m->col_ranges[rg + 1]; ++j)
      for (int t = m->tptr[j]; t < m->tptr[j + 1]; ++t) {
        const int i = m->trow[t], k = m->tcell[t];
        mat_t_vec(v + m->cval[k], m->rsz[i], m->csz[j], x + m->rpos[i], y + m->cpos[j], 1);
      }
}

// I/block_sparse_matrix.cc:351-401.
void oracle_squared_column_norm(const oracle_matrix* m, const double* v, double* x) {
  parallel_set_zero(x, m->num_cols);
  const int n_ranges = int(m->col_ranges.size()) - 1;
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 1)
  for (int rg = 0; rg < n_ranges; ++rg)
    for (int j = m->col_ranges[rg]; j < m->col_ranges[rg + 1]; ++j)
      for (int t = m->tptr[j]; t < m->tptr[j + 1]; ++t) {
        const int i = m->trow[t], k = m->tcell[t];
        const double* a = v + m->cval[k];
        for (int r = 0; r < m->rsz[i]; ++r)
          for (int c = 0; c < m->csz[j]; ++c) x[m->cpos[j] + c] += a[r * m->csz[j] + c] * a[r * m->csz[j] + c];
      }
}

// I/block_sparse_matrix.cc:403-450.
void oracle_scale_columns(const oracle_matrix* m, double* v, const double* scale) {
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int i = 0; i < m->nrb; ++i)
    for (int k = m->rptr[i]; k < m->rptr[i + 1]; ++k) {
      const int j = m->ccol[k];
      double* a = v + m->cval[k];
      for (int r = 0; r < m->rsz[i]; ++r)
        for (int c = 0; c < m->csz[j]; ++c) a[r * m->csz[j] + c] *= scale[m->cpos[j] + c];
    }
}

void oracle_to_dense(const oracle_matrix* m, const double* v, double* d) {
  std::fill(d, d + int64_t(m->num_rows) * m->num_cols, 0.0);
  for (int i = 0; i < m->nrb; ++i)
    for (int k = m->rptr[i]; k < m->rptr[i + 1]; ++k) {
      const int j = m->ccol[k];
      for (int r = 0; r < m->rsz[i]; ++r)
        for (int c = 0; c < m->csz[j]; ++c)
          d[int64_t(m->rpos[i] + r) * m->num_cols + m->cpos[j] + c] = v[m->cval[k] + r * m->csz[j] + c];
    }
}

// ---- PartitionedMatrixView -------------------------------------------------
// E = first cell of the first num_row_blocks_e rows; F = everything else; F-space
// vectors are indexed at col_pos - num_cols_e.  I/partitioned_matrix_view_impl.h.

// :112-137
void oracle_right_multiply_e(const oracle_matrix* m, const double* v, const double* x, double* y) {
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int i = 0; i < m->num_row_blocks_e; ++i) {
    const int k = m->rptr[i], j = m->ccol[k];
    mat_vec(v + m->cval[k], m->rsz[i], m->csz[j], x + m->cpos[j], y + m->rpos[i], 1);
  }
}
// :139-191
void oracle_right_multiply_f(const oracle_matrix* m, const double* v, const double* x, double* y) {
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int i = 0; i < m->nrb; ++i) {
    const int k0 = m->rptr[i] + (i < m->num_row_blocks_e ? 1 : 0);
    for (int k = k0; k < m->rptr[i + 1]; ++k) {
      const int j = m->ccol[k];
      mat_vec(v + m->cval[k], m->rsz[i], m->csz[j], x + m->cpos[j] - m->num_cols_e, y + m->rpos[i], 1);
    }
  }
}
// :206-264
void oracle_left_multiply_e(const oracle_matrix* m, const double* v, const double* x, double* y) {
  if (g_threads == 1) {
    for (int i = 0; i < m->num_row_blocks_e; ++i) {
      const int k = m->rptr[i], j = m->ccol[k];
      mat_t_vec(v + m->cval[k], m->rsz[i], m->csz[j], x + m->rpos[i], y + m->cpos[j], 1);
    }
    return;
  }
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 64)
  for (int j = 0; j < m->nelim; ++j)
    for (int t = m->tptr[j]; t < m->tptr[j + 1]; ++t) {
      const int i = m->trow[t], k = m->tcell[t];
      if (i >= m->num_row_blocks_e || k != m->rptr[i]) continue;
      mat_t_vec(v + m->cval[k], m->rsz[i], m->csz[j], x + m->rpos[i], y + m->cpos[j], 1);
    }
}
// :278-375
void oracle_left_multiply_f(const oracle_matrix* m, const double* v, const double* x, double* y) {
  if (g_threads == 1) {
    for (int i = 0; i < m->nrb; ++i) {
      const int k0 = m->rptr[i] + (i < m->num_row_blocks_e ? 1 : 0);
      for (int k = k0; k < m->rptr[i + 1]; ++k) {
        const int j = m->ccol[k];
        mat_t_vec(v + m->cval[k], m->rsz[i], m->csz[j], x + m->rpos[i], y + m->cpos[j] - m->num_cols_e, 1);
      }
    }
    return;
  }
  const int n_ranges = int(m->f_ranges.size()) - 1;
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 1)
  for (int rg = 0; rg < n_ranges; ++rg)
    for (int j = m->f_ranges[rg]; j < m->f_ranges[rg + 1]; ++j)
      for (int t = m->tptr[j]; t < m->tptr[j + 1]; ++t) {
        const int i = m->trow[t], k = m->tcell[t];
        mat_t_vec(v + m->cval[k], m->rsz[i], m->csz[j], x + m->rpos[i], y + m->cpos[j] - m->num_cols_e, 1);
      }
}
// UpdateBlockDiagonalEtE, :446-523
void oracle_block_diagonal_ete(const oracle_matrix* m, const double* v, double* blocks) {
  std::fill(blocks, blocks + m->diag_offset_e.back(), 0.0);
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 64)
  for (int j = 0; j < m->nelim; ++j)
    for (int t = m->tptr[j]; t < m->tptr[j + 1]; ++t) {
      const int i = m->trow[t], k = m->tcell[t];
      if (i >= m->num_row_blocks_e || k != m->rptr[i]) continue;
      mat_t_mat(v + m->cval[k], m->rsz[i], m->csz[j], v + m->cval[k], m->csz[j],
                blocks + m->diag_offset_e[j], 0, 0, m->csz[j], 1);
    }
}
// UpdateBlockDiagonalFtF, :530-658
void oracle_block_diagonal_ftf(const oracle_matrix* m, const double* v, double* blocks) {
  std::fill(blocks, blocks + m->diag_offset_f.back(), 0.0);
  const int n_ranges = int(m->f_ranges.size()) - 1;
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 1)
  for (int rg = 0; rg < n_ranges; ++rg)
    for (int j = m->f_ranges[rg]; j < m->f_ranges[rg + 1]; ++j)
      for (int t = m->tptr[j]; t < m->tptr[j + 1]; ++t) {
        const int i = m->trow[t], k = m->tcell[t];
        mat_t_mat(v + m->cval[k], m->rsz[i], m->csz[j], v + m->cval[k], m->csz[j],
                  blocks + m->diag_offset_f[j - m->nelim], 0, 0, m->csz[j], 1);
      }
}

int oracle_invert_psd(int n, double* a) { return invert_spd_upper(n, a); }

// y += blockdiag x.  I/block_random_access_diagonal_matrix.cc:102-116.
void oracle_block_diagonal_apply(int nb, const int32_t* bsz, const double* blocks, const double* x,
                                 double* y) {
  std::vector<int64_t> boff(nb + 1, 0), voff(nb + 1, 0);
  for (int i = 0; i < nb; ++i) { boff[i + 1] = boff[i] + int64_t(bsz[i]) * bsz[i]; voff[i + 1] = voff[i] + bsz[i]; }
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int i = 0; i < nb; ++i) mat_vec(blocks + boff[i], bsz[i], bsz[i], x + voff[i], y + voff[i], 1);
}

}  // extern "C"

// --------------------------------------------------------------------------
// ImplicitSchurComplement.  I/implicit_schur_complement.cc.
// --------------------------------------------------------------------------
struct oracle_isc {
  const oracle_matrix* m;
  const double *values = nullptr, *D = nullptr, *b = nullptr;
  Vec ete_inv, ftf_inv, rhs, tmp_rows, tmp_e, tmp_e2, tmp_f;
  std::vector<int32_t> e_sizes, f_sizes;
  bool f_diagonal_in_sx = true;  // sharded runs add D_f^2 x after the all-reduce
};

namespace {

void add_diagonal_and_invert(const oracle_matrix* m, int first_block, int num_blocks,
                             const std::vector<int>& offsets, const double* D, double* blocks) {
  // I/implicit_schur_complement.cc:179-204
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int q = 0; q < num_blocks; ++q) {
    const int j = first_block + q, n = m->csz[j];
    double* blk = blocks + offsets[q];
    if (D) for (int i = 0; i < n; ++i) blk[i * n + i] += D[m->cpos[j] + i] * D[m->cpos[j] + i];
    invert_spd_upper(n, blk);
  }
}

void isc_update_rhs(oracle_isc* s) {
  // I/implicit_schur_complement.cc:251-276
  const oracle_matrix* m = s->m;
  parallel_set_zero(s->tmp_e.data(), long(s->tmp_e.size()));
  oracle_left_multiply_e(m, s->values, s->b, s->tmp_e.data());
  parallel_set_zero(s->tmp_e2.data(), long(s->tmp_e2.size()));
  oracle_block_diagonal_apply(m->nelim, s->e_sizes.data(), s->ete_inv.data(), s->tmp_e.data(), s->tmp_e2.data());
  parallel_set_zero(s->tmp_rows.data(), long(s->tmp_rows.size()));
  oracle_right_multiply_e(m, s->values, s->tmp_e2.data(), s->tmp_rows.data());
  parallel_subtract_from(s->b, s->tmp_rows.data(), m->num_rows);
  std::fill(s->rhs.begin(), s->rhs.end(), 0.0);
  oracle_left_multiply_f(m, s->values, s->tmp_rows.data(), s->rhs.data());
}

}  // namespace

extern "C" {

oracle_isc* oracle_isc_create(const oracle_matrix* m) {
  auto* s = new oracle_isc;
  s->m = m;
  s->ete_inv.assign(m->diag_offset_e.back(), 0.0);
  s->ftf_inv.assign(m->diag_offset_f.back(), 0.0);
  s->rhs.assign(m->num_cols_f, 0.0);
  s->tmp_rows.assign(m->num_rows, 0.0);
  s->tmp_e.assign(m->num_cols_e, 0.0);
  s->tmp_e2.assign(m->num_cols_e, 0.0);
  s->tmp_f.assign(m->num_cols_f, 0.0);
  for (int j = 0; j < m->ncb; ++j) (j < m->nelim ? s->e_sizes : s->f_sizes).push_back(m->csz[j]);
  return s;
}
void oracle_isc_destroy(oracle_isc* s) { delete s; }

// Init, :49-97 (the F^T F inverse is only built for the JACOBI preconditioner).
void oracle_isc_init(oracle_isc* s, const double* values, const double* D, const double* b) {
  s->values = values; s->D = D; s->b = b;
  oracle_block_diagonal_ete(s->m, values, s->ete_inv.data());
  add_diagonal_and_invert(s->m, 0, s->m->nelim, s->m->diag_offset_e, D, s->ete_inv.data());
  isc_update_rhs(s);
}

// RightMultiplyAndAccumulate, :106-144.  Note: ASSIGNS y.
void oracle_isc_sx(oracle_isc* s, const double* x, double* y) {
  const oracle_matrix* m = s->m;
  parallel_set_zero(s->tmp_rows.data(), long(s->tmp_rows.size()));
  oracle_right_multiply_f(m, s->values, x, s->tmp_rows.data());
  parallel_set_zero(s->tmp_e.data(), long(s->tmp_e.size()));
  oracle_left_multiply_e(m, s->values, s->tmp_rows.data(), s->tmp_e.data());
  parallel_set_zero(s->tmp_e2.data(), long(s->tmp_e2.size()));
  oracle_block_diagonal_apply(m->nelim, s->e_sizes.data(), s->ete_inv.data(), s->tmp_e.data(), s->tmp_e2.data());
  parallel_negate(s->tmp_e2.data(), long(s->tmp_e2.size()));
  oracle_right_multiply_e(m, s->values, s->tmp_e2.data(), s->tmp_rows.data());
  if (s->D && s->f_diagonal_in_sx) {
    const double* Df = s->D + m->num_cols_e;
    for (int i = 0; i < m->num_cols_f; ++i) y[i] = Df[i] * Df[i] * x[i];
  } else {
    std::fill(y, y + m->num_cols_f, 0.0);
  }
  oracle_left_multiply_f(m, s->values, s->tmp_rows.data(), y);
}

// InversePowerSeriesOperatorRightMultiplyAccumulate, :146-174:  y += (F'F)^-1 F'E (E'E)^-1 E'F x.
// Needs block_diagonal_FtF_inverse (oracle_isc_compute_ftf_inverse).
void oracle_isc_power_series_operator(oracle_isc* s, const double* x, double* y) {
  const oracle_matrix* m = s->m;
  std::fill(s->tmp_rows.begin(), s->tmp_rows.end(), 0.0);
  oracle_right_multiply_f(m, s->values, x, s->tmp_rows.data());
  std::fill(s->tmp_e.begin(), s->tmp_e.end(), 0.0);
  oracle_left_multiply_e(m, s->values, s->tmp_rows.data(), s->tmp_e.data());
  std::fill(s->tmp_e2.begin(), s->tmp_e2.end(), 0.0);
  oracle_block_diagonal_apply(m->nelim, s->e_sizes.data(), s->ete_inv.data(), s->tmp_e.data(), s->tmp_e2.data());
  std::fill(s->tmp_rows.begin(), s->tmp_rows.end(), 0.0);
  oracle_right_multiply_e(m, s->values, s->tmp_e2.data(), s->tmp_rows.data());
  std::fill(s->tmp_f.begin(), s->tmp_f.end(), 0.0);
  oracle_left_multiply_f(m, s->values, s->tmp_rows.data(), s->tmp_f.data());
  oracle_block_diagonal_apply(m->ncb - m->nelim, s->f_sizes.data(), s->ftf_inv.data(), s->tmp_f.data(), y);
}

// block_diagonal_FtF_inverse_ of ImplicitSchurComplement::Init (:57-91): blockdiag(F'F + D_f^2)^-1.
void oracle_isc_compute_ftf_inverse(oracle_isc* s) {
  const oracle_matrix* m = s->m;
  oracle_block_diagonal_ftf(m, s->values, s->ftf_inv.data());
  add_diagonal_and_invert(m, m->nelim, m->ncb - m->nelim, m->diag_offset_f, s->D, s->ftf_inv.data());
}

// PowerSeriesExpansionPreconditioner::RightMultiplyAndAccumulate,
// I/power_series_expansion_preconditioner.cc:57-84 (y is ASSIGNED, as there).
void oracle_isc_spse_apply(oracle_isc* s, const double* x, double* y, int max_num_spse_iterations, double spse_tolerance) {
  const oracle_matrix* m = s->m;
  const int n = m->num_cols_f;
  Vec series(n), previous(n);
  std::fill(y, y + n, 0.0);
  oracle_block_diagonal_apply(m->ncb - m->nelim, s->f_sizes.data(), s->ftf_inv.data(), x, y);
  std::copy(y, y + n, previous.begin());
  double ny = 0;
  for (int i = 0; i < n; ++i) ny += y[i] * y[i];
  const double threshold = spse_tolerance * std::sqrt(ny);
  for (int i = 1;; ++i) {
    std::fill(series.begin(), series.end(), 0.0);
    oracle_isc_power_series_operator(s, previous.data(), series.data());
    double nt = 0;
    for (int k = 0; k < n; ++k) { y[k] += series[k]; nt += series[k] * series[k]; }
    if (i >= max_num_spse_iterations || std::sqrt(nt) < threshold) break;
    std::swap(previous, series);
  }
}

void oracle_isc_rhs(const oracle_isc* s, double* rhs) { std::copy(s->rhs.begin(), s->rhs.end(), rhs); }
void oracle_isc_ete_inverse(const oracle_isc* s, double* blocks) { std::copy(s->ete_inv.begin(), s->ete_inv.end(), blocks); }

// BackSubstitute, :208-243.  z may be NULL when there are no F blocks.
void oracle_isc_back_substitute(oracle_isc* s, const double* z, double* x) {
  const oracle_matrix* m = s->m;
  parallel_set_zero(s->tmp_rows.data(), long(s->tmp_rows.size()));
  if (m->num_cols_f > 0) oracle_right_multiply_f(m, s->values, z, s->tmp_rows.data());
  parallel_subtract_from(s->b, s->tmp_rows.data(), m->num_rows);
  parallel_set_zero(s->tmp_e.data(), long(s->tmp_e.size()));
  oracle_left_multiply_e(m, s->values, s->tmp_rows.data(), s->tmp_e.data());
  parallel_set_zero(x, m->num_cols);
  oracle_block_diagonal_apply(m->nelim, s->e_sizes.data(), s->ete_inv.data(), s->tmp_e.data(), x);
  for (int i = 0; i < m->num_cols_f; ++i) x[m->num_cols_e + i] = z[i];
}

}  // extern "C"

// --------------------------------------------------------------------------
// SchurEliminator.  I/schur_eliminator_impl.h.
// The lhs is reached through "get cell" exactly as BlockRandomAccessMatrix:
// diagonal store: only (i,i) cells exist (I/block_random_access_diagonal_matrix.cc:62-81);
// dense store: every cell exists, stride = num_cols_f.
// --------------------------------------------------------------------------
namespace {

// The reference guards every lhs cell with a mutex (I/schur_eliminator_impl.h:557,689,710).  With a block-diagonal lhs
// (SCHUR_JACOBI) every thread hammers the same few thousand 9x9 cells and the locks — and the cache lines behind them —
// stop the loop from scaling past ~16 threads.  `private_copy` gives each thread its own zeroed image of the diagonal
// store (a few MB), no locks, and the images are added up afterwards: same sums, a fair multi-core baseline.
struct Lhs {
  const oracle_matrix* m;
  bool diagonal;
  double* data;
  bool use_locks = true;
  std::vector<omp_lock_t> locks;
  Lhs(const oracle_matrix* mm, bool d, double* p, bool with_locks = true) : m(mm), diagonal(d), data(p), use_locks(with_locks), locks(with_locks ? 1021 : 1) {
    for (auto& l : locks) omp_init_lock(&l);
  }
  ~Lhs() { for (auto& l : locks) omp_destroy_lock(&l); }
  // returns pointer to storage base and (r, c, stride) or nullptr
  double* cell(int b1, int b2, int* r, int* c, int* stride, omp_lock_t** lock) {
    if (diagonal) {
      if (b1 != b2) return nullptr;
      *r = 0; *c = 0; *stride = m->csz[m->nelim + b1];
      *lock = use_locks ? &locks[b1 % locks.size()] : nullptr;
      return data + m->diag_offset_f[b1];
    }
    *r = m->lhs_row_layout[b1]; *c = m->lhs_row_layout[b2]; *stride = m->num_cols_f;
    *lock = &locks[(size_t(b1) * 7919u + b2) % locks.size()];
    return data;
  }
};

// S += F_i^T F_j over the F cells of one row, i <= j.  first = index of first F cell.
// EBlockRowOuterProduct :672-721 / NoEBlockRowOuterProduct :617-666.
void row_outer_product(const oracle_matrix* m, const double* v, int row, int first, Lhs* lhs) {
  for (int a = first; a < m->rptr[row + 1]; ++a) {
    const int b1 = m->ccol[a] - m->nelim, s1 = m->csz[m->ccol[a]];
    int r, c, stride; omp_lock_t* lock;
    if (double* p = lhs->cell(b1, b1, &r, &c, &stride, &lock)) {
      if (lock) omp_set_lock(lock);
      mat_t_mat(v + m->cval[a], m->rsz[row], s1, v + m->cval[a], s1, p, r, c, stride, 1);
      if (lock) omp_unset_lock(lock);
    }
    for (int bb = a + 1; bb < m->rptr[row + 1]; ++bb) {
      const int b2 = m->ccol[bb] - m->nelim, s2 = m->csz[m->ccol[bb]];
      if (double* p = lhs->cell(b1, b2, &r, &c, &stride, &lock)) {
        if (lock) omp_set_lock(lock);
        mat_t_mat(v + m->cval[a], m->rsz[row], s1, v + m->cval[bb], s2, p, r, c, stride, 1);
        if (lock) omp_unset_lock(lock);
      }
    }
  }
}

// Eliminate :184-311.  add_f_diagonal=false leaves D_f^2 out (sharded runs add it
// once after the all-reduce).
void schur_eliminate(const oracle_matrix* m, const double* v, const double* b, const double* D,
                     bool diagonal_only, bool add_f_diagonal, double* lhs_data, double* rhs) {
  const int nf = m->ncb - m->nelim;
  const int64_t lhs_len = diagonal_only ? m->diag_offset_f.back() : int64_t(m->num_cols_f) * m->num_cols_f;
  std::fill(lhs_data, lhs_data + lhs_len, 0.0);
  if (rhs) std::fill(rhs, rhs + m->num_cols_f, 0.0);
  Lhs lhs(m, diagonal_only, lhs_data);
  int e_static, row_static, f_static;
  oracle_detect_structure(m, &row_static, &e_static, &f_static);

  if (D && add_f_diagonal) {  // :198-219
    for (int q = 0; q < nf; ++q) {
      int r, c, stride; omp_lock_t* lock;
      double* p = lhs.cell(q, q, &r, &c, &stride, &lock);
      const int j = m->nelim + q;
      for (int i = 0; i < m->csz[j]; ++i) p[(r + i) * stride + c + i] += D[m->cpos[j] + i] * D[m->cpos[j] + i];
    }
  }
  std::vector<omp_lock_t> rhs_locks(nf);
  for (auto& l : rhs_locks) omp_init_lock(&l);

  // thread-private images of the (block-diagonal) lhs and of the rhs: see struct Lhs
  const bool private_copy = diagonal_only && g_threads > 1;
  std::vector<Vec> lhs_priv, rhs_priv;
  if (private_copy) {
    lhs_priv.assign(g_threads, Vec());
    rhs_priv.assign(g_threads, Vec());
  }
  Lhs* const shared_lhs = &lhs;
#pragma omp parallel num_threads(g_threads)
  {
    Vec buffer(m->buffer_size), scratch(m->buffer_size), ete, g, inv_g, sj;
    Lhs* lhs_ptr = shared_lhs;
    double* rhs_t = rhs;
    std::unique_ptr<Lhs> mine;
    if (private_copy) {
      const int tid = omp_get_thread_num();
      lhs_priv[tid].assign(size_t(lhs_len), 0.0);
      mine.reset(new Lhs(m, true, lhs_priv[tid].data(), false));
      lhs_ptr = mine.get();
      if (rhs) { rhs_priv[tid].assign(size_t(m->num_cols_f), 0.0); rhs_t = rhs_priv[tid].data(); }
    }
    Lhs& lhs = *lhs_ptr;
#pragma omp for schedule(dynamic, 16)
    for (int ci = 0; ci < int(m->chunks.size()); ++ci) {
      const auto& ch = m->chunks[ci];
      const int e_id = m->ccol[m->rptr[ch.start]], es = m->csz[e_id];
      std::fill(buffer.begin(), buffer.end(), 0.0);
      ete.assign(es * es, 0.0);
      if (D) for (int i = 0; i < es; ++i) ete[i * es + i] = D[m->cpos[e_id] + i] * D[m->cpos[e_id] + i];
      g.assign(es, 0.0);
      // ChunkDiagonalBlockAndGradient :449-512
      for (int j = 0; j < ch.size; ++j) {
        const int row = ch.start + j, k0 = m->rptr[row], rs = m->rsz[row];
        if (m->rptr[row + 1] - k0 > 1) row_outer_product(m, v, row, k0 + 1, &lhs);
        const double* E = v + m->cval[k0];
        mat_t_mat(E, rs, es, E, es, ete.data(), 0, 0, es, 1);
        if (b) mat_t_vec(E, rs, es, b + m->rpos[row], g.data(), 1);
        for (int k = k0 + 1; k < m->rptr[row + 1]; ++k) {
          const int fs = m->csz[m->ccol[k]];
          mat_t_mat(E, rs, es, v + m->cval[k], fs, buffer.data() + ch.layout.at(m->ccol[k]), 0, 0, fs, 1);
        }
      }
      // InvertPSDMatrix<kEBlockSize>(assume_full_rank_ete = true, ete) :265-266
      if (e_static > 0 && e_static < 5) invert_small(es, ete.data()); else invert_spd_upper(es, ete.data());
      if (rhs) {  // UpdateRhs :386-427
        inv_g.assign(es, 0.0);
        mat_vec(ete.data(), es, es, g.data(), inv_g.data(), 0);
        for (int j = 0; j < ch.size; ++j) {
          const int row = ch.start + j, k0 = m->rptr[row], rs = m->rsz[row];
          sj.assign(b + m->rpos[row], b + m->rpos[row] + rs);
          mat_vec(v + m->cval[k0], rs, es, inv_g.data(), sj.data(), -1);
          for (int k = k0 + 1; k < m->rptr[row + 1]; ++k) {
            const int blk = m->ccol[k] - m->nelim;
            if (!private_copy) omp_set_lock(&rhs_locks[blk]);
            mat_t_vec(v + m->cval[k], rs, m->csz[m->ccol[k]], sj.data(), rhs_t + m->lhs_row_layout[blk], 1);
            if (!private_copy) omp_unset_lock(&rhs_locks[blk]);
          }
        }
      }
      // ChunkOuterProduct :519-568: S(i,j) -= b_i^T ete^-1 b_j for i <= j in layout order
      for (auto it1 = ch.layout.begin(); it1 != ch.layout.end(); ++it1) {
        const int b1 = it1->first - m->nelim, s1 = m->csz[it1->first];
        mat_t_mat(buffer.data() + it1->second, es, s1, ete.data(), es, scratch.data(), 0, 0, es, 0);
        for (auto it2 = it1; it2 != ch.layout.end(); ++it2) {
          const int b2 = it2->first - m->nelim, s2 = m->csz[it2->first];
          int r, c, stride; omp_lock_t* lock;
          if (double* p = lhs.cell(b1, b2, &r, &c, &stride, &lock)) {
            if (lock) omp_set_lock(lock);
            mat_mat(scratch.data(), s1, es, buffer.data() + it2->second, s2, p, r, c, stride, -1);
            if (lock) omp_unset_lock(lock);
          }
        }
      }
    }
    if (private_copy) {  // add the images up (implicit barrier of the omp for above: every image is complete)
#pragma omp for schedule(static)
      for (int64_t i = 0; i < lhs_len; ++i) {
        double t = 0;
        for (int th = 0; th < g_threads; ++th) if (!lhs_priv[th].empty()) t += lhs_priv[th][size_t(i)];
        lhs_data[i] += t;
      }
      if (rhs) {
#pragma omp for schedule(static)
        for (int i = 0; i < m->num_cols_f; ++i) {
          double t = 0;
          for (int th = 0; th < g_threads; ++th) if (!rhs_priv[th].empty()) t += rhs_priv[th][size_t(i)];
          rhs[i] += t;
        }
      }
    }
  }
  // NoEBlockRowsUpdate :574-600
  for (int row = m->uneliminated_row_begins; row < m->nrb; ++row) {
    row_outer_product(m, v, row, m->rptr[row], &lhs);
    if (!rhs) continue;
    for (int k = m->rptr[row]; k < m->rptr[row + 1]; ++k)
      mat_t_vec(v + m->cval[k], m->rsz[row], m->csz[m->ccol[k]], b + m->rpos[row],
                rhs + m->lhs_row_layout[m->ccol[k] - m->nelim], 1);
  }
  for (auto& l : rhs_locks) omp_destroy_lock(&l);
}

}  // namespace

extern "C" {

void oracle_schur_eliminate(const oracle_matrix* m, const double* v, const double* b, const double* D,
                            int diagonal_only, double* lhs, double* rhs) {
  schur_eliminate(m, v, b, D, diagonal_only != 0, true, lhs, rhs);
}

// BackSubstitute :314-380
void oracle_schur_back_substitute(const oracle_matrix* m, const double* v, const double* b,
                                  const double* D, const double* z, double* y) {
  int e_static, row_static, f_static;
  oracle_detect_structure(m, &row_static, &e_static, &f_static);
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 16)
  for (int ci = 0; ci < int(m->chunks.size()); ++ci) {
    const auto& ch = m->chunks[ci];
    const int e_id = m->ccol[m->rptr[ch.start]], es = m->csz[e_id];
    double* yp = y + m->cpos[e_id];
    Vec ete(es * es, 0.0), sj, acc(es, 0.0);
    if (D) for (int i = 0; i < es; ++i) ete[i * es + i] = D[m->cpos[e_id] + i] * D[m->cpos[e_id] + i];
    for (int j = 0; j < ch.size; ++j) {
      const int row = ch.start + j, k0 = m->rptr[row], rs = m->rsz[row];
      sj.assign(b + m->rpos[row], b + m->rpos[row] + rs);
      for (int k = k0 + 1; k < m->rptr[row + 1]; ++k)
        mat_vec(v + m->cval[k], rs, m->csz[m->ccol[k]], z + m->lhs_row_layout[m->ccol[k] - m->nelim], sj.data(), -1);
      mat_t_vec(v + m->cval[k0], rs, es, sj.data(), acc.data(), 1);
      mat_t_mat(v + m->cval[k0], rs, es, v + m->cval[k0], es, ete.data(), 0, 0, es, 1);
    }
    if (e_static > 0 && e_static < 5) invert_small(es, ete.data()); else invert_spd_upper(es, ete.data());
    mat_vec(ete.data(), es, es, acc.data(), yp, 0);
  }
  // The F part of y is the reduced solution itself (what the callers copy in).
  for (int i = 0; i < m->num_cols_f; ++i) y[m->num_cols_e + i] = z[i];
}

}  // extern "C"

// --------------------------------------------------------------------------
// Preconditioners.
// --------------------------------------------------------------------------
namespace {

// BlockSparseJacobiPreconditioner::UpdateImpl, I/block_jacobi_preconditioner.cc:59-115.
// shared_first: column blocks >= shared_first are replicated over ranks and their
// raw blocks are summed with `ar` before the diagonal is added (sharded runs).
void block_jacobi(const oracle_matrix* m, const double* v, const double* D, double* inv, double* raw,
                  oracle_allreduce_fn ar, void* ctx) {
  const int64_t len = m->diag_offset_all.back();
  std::fill(inv, inv + len, 0.0);
  const int n_ranges = int(m->col_ranges.size()) - 1;
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 1)
  for (int rg = 0; rg < n_ranges; ++rg)
    for (int j = m->col_ranges[rg]; j < m->col_ranges[rg + 1]; ++j)
      for (int t = m->tptr[j]; t < m->tptr[j + 1]; ++t) {
        const int i = m->trow[t], k = m->tcell[t];
        mat_t_mat(v + m->cval[k], m->rsz[i], m->csz[j], v + m->cval[k], m->csz[j], inv + m->diag_offset_all[j], 0, 0, m->csz[j], 1);
      }
  if (ar) ar(ctx, inv + m->diag_offset_all[m->nelim], len - m->diag_offset_all[m->nelim]);
  if (D)
    for (int j = 0; j < m->ncb; ++j)
      for (int i = 0; i < m->csz[j]; ++i) inv[m->diag_offset_all[j] + i * m->csz[j] + i] += D[m->cpos[j] + i] * D[m->cpos[j] + i];
  if (raw) std::copy(inv, inv + len, raw);
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 256)
  for (int j = 0; j < m->ncb; ++j) invert_spd_upper(m->csz[j], inv + m->diag_offset_all[j]);
}

// SchurJacobiPreconditioner::UpdateImpl, I/schur_jacobi_preconditioner.cc:87-97.
void schur_jacobi(const oracle_matrix* m, const double* v, const double* D, double* inv, double* raw,
                  oracle_allreduce_fn ar, void* ctx) {
  const int nf = m->ncb - m->nelim;
  schur_eliminate(m, v, nullptr, D, true, ar == nullptr, inv, nullptr);
  if (ar) {
    ar(ctx, inv, m->diag_offset_f.back());
    if (D)
      for (int q = 0; q < nf; ++q) {
        const int j = m->nelim + q;
        for (int i = 0; i < m->csz[j]; ++i) inv[m->diag_offset_f[q] + i * m->csz[j] + i] += D[m->cpos[j] + i] * D[m->cpos[j] + i];
      }
  }
  if (raw) std::copy(inv, inv + m->diag_offset_f.back(), raw);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int q = 0; q < nf; ++q) invert_spd_upper(m->csz[m->nelim + q], inv + m->diag_offset_f[q]);
}

// --------------------------------------------------------------------------
// ConjugateGradientsSolver, I/conjugate_gradients_solver.h:108-306, with the
// vector ops of I/eigen_vector_ops.h:47-101 behind `dot` so that a sharded run
// can sum partial inner products.  The order of the termination tests is the
// reference's.
// --------------------------------------------------------------------------
using Op = std::function<void(const double*, double*)>;  // y += A x
using DotFn = std::function<double(const double*, const double*)>;

void cg_solve(int n, const Op& lhs, const double* rhs, const Op& precond, const DotFn& dot, int min_it,
              int max_it, int reset_period, double q_tol, double r_tol, double* x, oracle_summary* sum) {
  auto is_zero_or_inf = [](double v) { return v == 0.0 || std::isinf(v); };
  Vec p(n), r(n), z(n), tmp(n);
  sum->termination_type = 1;
  std::snprintf(sum->message, sizeof(sum->message), "Maximum number of iterations reached.");
  sum->num_iterations = 0;
  sum->residual_norm = -1;
  const double norm_rhs = std::sqrt(dot(rhs, rhs));
  if (norm_rhs == 0.0) {
    std::fill(x, x + n, 0.0);
    sum->termination_type = 0;
    std::snprintf(sum->message, sizeof(sum->message), "Convergence. |b| = 0.");
    return;
  }
  const double tol_r = r_tol * norm_rhs;
  std::fill(tmp.begin(), tmp.end(), 0.0);
  lhs(x, tmp.data());
  for (int i = 0; i < n; ++i) r[i] = rhs[i] - tmp[i];
  double norm_r = std::sqrt(dot(r.data(), r.data()));
  if (min_it == 0 && norm_r <= tol_r) {
    sum->termination_type = 0;
    std::snprintf(sum->message, sizeof(sum->message), "Convergence. |r| = %e <= %e.", norm_r, tol_r);
    return;
  }
  double rho = 1.0;
  for (int i = 0; i < n; ++i) tmp[i] = rhs[i] + r[i];
  double Q0 = -dot(x, tmp.data());
  for (sum->num_iterations = 1;; ++sum->num_iterations) {
    std::fill(z.begin(), z.end(), 0.0);
    precond(r.data(), z.data());
    const double last_rho = rho;
    rho = dot(r.data(), z.data());
    if (is_zero_or_inf(rho)) {
      sum->termination_type = 2;
      std::snprintf(sum->message, sizeof(sum->message), "Numerical failure. rho = r'z = %e.", rho);
      break;
    }
    if (sum->num_iterations == 1) {
      p = z;
    } else {
      const double beta = rho / last_rho;
      if (is_zero_or_inf(beta)) {
        sum->termination_type = 2;
        std::snprintf(sum->message, sizeof(sum->message),
                      "Numerical failure. beta = rho_n / rho_{n-1} = %e, rho_n = %e, rho_{n-1} = %e", beta, rho, last_rho);
        break;
      }
      for (int i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
    }
    Vec& q = z;
    std::fill(q.begin(), q.end(), 0.0);
    lhs(p.data(), q.data());
    const double pq = dot(p.data(), q.data());
    if (pq <= 0 || std::isinf(pq)) {
      sum->termination_type = 1;
      std::snprintf(sum->message, sizeof(sum->message),
                    "Matrix is indefinite, no more progress can be made. p'q = %e. |p| = %e, |q| = %e", pq,
                    std::sqrt(dot(p.data(), p.data())), std::sqrt(dot(q.data(), q.data())));
      break;
    }
    const double alpha = rho / pq;
    if (std::isinf(alpha)) {
      sum->termination_type = 2;
      std::snprintf(sum->message, sizeof(sum->message),
                    "Numerical failure. alpha = rho / pq = %e, rho = %e, pq = %e.", alpha, rho, pq);
      break;
    }
    for (int i = 0; i < n; ++i) x[i] = x[i] + alpha * p[i];
    if (sum->num_iterations % reset_period == 0) {
      std::fill(tmp.begin(), tmp.end(), 0.0);
      lhs(x, tmp.data());
      for (int i = 0; i < n; ++i) r[i] = rhs[i] - tmp[i];
    } else {
      for (int i = 0; i < n; ++i) r[i] = r[i] - alpha * q[i];
    }
    for (int i = 0; i < n; ++i) tmp[i] = rhs[i] + r[i];
    const double Q1 = -dot(x, tmp.data());
    const double zeta = sum->num_iterations * (Q1 - Q0) / Q1;
    if (zeta < q_tol && sum->num_iterations >= min_it) {
      sum->termination_type = 0;
      std::snprintf(sum->message, sizeof(sum->message), "Iteration: %d Convergence: zeta = %e < %e. |r| = %e",
                    sum->num_iterations, zeta, q_tol, std::sqrt(dot(r.data(), r.data())));
      break;
    }
    Q0 = Q1;
    norm_r = std::sqrt(dot(r.data(), r.data()));
    if (norm_r <= tol_r && sum->num_iterations >= min_it) {
      sum->termination_type = 0;
      std::snprintf(sum->message, sizeof(sum->message), "Iteration: %d Convergence. |r| = %e <= %e.",
                    sum->num_iterations, norm_r, tol_r);
      break;
    }
    if (sum->num_iterations >= max_it) break;
  }
}

double plain_dot(const double* a, const double* b, int n) {
  double s = 0;
#pragma omp parallel for num_threads(g_threads) reduction(+ : s) schedule(static)
  for (int i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

}  // namespace

extern "C" {

void oracle_block_jacobi(const oracle_matrix* m, const double* v, const double* D, double* inv, double* raw) {
  block_jacobi(m, v, D, inv, raw, nullptr, nullptr);
}
void oracle_schur_jacobi(const oracle_matrix* m, const double* v, const double* D, double* inv, double* raw) {
  schur_jacobi(m, v, D, inv, raw, nullptr, nullptr);
}

void oracle_cg_dense(int n, const double* A, const double* b, const double* Minv, int min_it, int max_it,
                     int reset_period, double q_tol, double r_tol, double* x, oracle_summary* summary) {
  Op lhs = [&](const double* in, double* out) { mat_vec(A, n, n, in, out, 1); };
  Op pre = [&](const double* in, double* out) {
    if (Minv) mat_vec(Minv, n, n, in, out, 1); else for (int i = 0; i < n; ++i) out[i] += in[i];
  };
  DotFn dot = [&](const double* a, const double* c) { return plain_dot(a, c, n); };
  cg_solve(n, lhs, b, pre, dot, min_it, max_it, reset_period, q_tol, r_tol, x, summary);
}

// CgnrSolver::SolveImpl, I/cgnr_solver.cc:146-207, and its linear operator :85-114.
void oracle_cgnr_solve_sharded(const oracle_matrix* m, const double* v, const double* b, const double* D,
                               int preconditioner, int min_it, int max_it, int reset_period, double q_tol,
                               double r_tol, double* x, oracle_summary* summary, oracle_allreduce_fn ar,
                               void* ctx) {
  const int n = m->num_cols;
  // Sharded convention: the first nelim column blocks are local, the rest replicated.
  const int shared_pos = ar ? m->num_cols_e : 0;
  std::vector<int32_t> sizes(m->csz.begin(), m->csz.end());
  Vec minv;
  if (preconditioner == 1) {
    minv.resize(m->diag_offset_all.back());
    block_jacobi(m, v, D, minv.data(), nullptr, ar, ctx);
  }
  Vec zrows(m->num_rows);
  Op lhs = [&](const double* in, double* out) {
    std::fill(zrows.begin(), zrows.end(), 0.0);
    oracle_right_multiply(m, v, in, zrows.data());
    if (!ar) {
      oracle_left_multiply(m, v, zrows.data(), out);
    } else {
      Vec part(n, 0.0);
      oracle_left_multiply(m, v, zrows.data(), part.data());
      ar(ctx, part.data() + shared_pos, n - shared_pos);
      for (int i = 0; i < n; ++i) out[i] += part[i];
    }
    if (D) for (int i = 0; i < n; ++i) out[i] += D[i] * D[i] * in[i];
  };
  Op pre = [&](const double* in, double* out) {
    if (preconditioner == 1) oracle_block_diagonal_apply(m->ncb, sizes.data(), minv.data(), in, out);
    else for (int i = 0; i < n; ++i) out[i] += in[i];
  };
  DotFn dot = [&](const double* a, const double* c) {
    if (!ar) return plain_dot(a, c, n);
    double local = plain_dot(a, c, shared_pos);
    ar(ctx, &local, 1);
    return local + plain_dot(a + shared_pos, c + shared_pos, n - shared_pos);
  };
  Vec rhs(n, 0.0);
  oracle_left_multiply(m, v, b, rhs.data());
  if (ar) ar(ctx, rhs.data() + shared_pos, n - shared_pos);
  std::fill(x, x + n, 0.0);
  cg_solve(n, lhs, rhs.data(), pre, dot, min_it, max_it, reset_period, q_tol, r_tol, x, summary);
}

void oracle_cgnr_solve(const oracle_matrix* m, const double* v, const double* b, const double* D,
                       int preconditioner, int min_it, int max_it, int reset_period, double q_tol,
                       double r_tol, double* x, oracle_summary* summary) {
  oracle_cgnr_solve_sharded(m, v, b, D, preconditioner, min_it, max_it, reset_period, q_tol, r_tol, x, summary,
                            nullptr, nullptr);
}

// IterativeSchurComplementSolver::SolveImpl, I/iterative_schur_complement_solver.cc:64-157.
void oracle_iterative_schur_solve_sharded(const oracle_matrix* m, const double* v, const double* b,
                                          const double* D, int preconditioner, int min_it, int max_it,
                                          int reset_period, double q_tol, double r_tol, double* x,
                                          oracle_summary* summary, oracle_allreduce_fn ar, void* ctx) {
  std::unique_ptr<oracle_isc, void (*)(oracle_isc*)> isc(oracle_isc_create(m), oracle_isc_destroy);
  isc->f_diagonal_in_sx = (ar == nullptr);
  oracle_isc_init(isc.get(), v, D, b);
  if (ar) ar(ctx, isc->rhs.data(), m->num_cols_f);
  const int nf = m->num_cols_f;
  if (m->ncb - m->nelim == 0) {  // :88-95
    summary->num_iterations = 0;
    summary->termination_type = 0;
    summary->residual_norm = -1;
    summary->message[0] = 0;
    oracle_isc_back_substitute(isc.get(), nullptr, x);
    return;
  }
  Vec minv;
  if (preconditioner == 2) {
    minv.resize(m->diag_offset_f.back());
    schur_jacobi(m, v, D, minv.data(), nullptr, ar, ctx);
  } else if (preconditioner == 1) {  // blockdiag(F^T F + D_f^2)^-1, :173-177 + ISC::Init
    minv.resize(m->diag_offset_f.back());
    oracle_block_diagonal_ftf(m, v, minv.data());
    if (ar) ar(ctx, minv.data(), minv.size());
    add_diagonal_and_invert(m, m->nelim, m->ncb - m->nelim, m->diag_offset_f, D, minv.data());
  }
  const double* Df = D ? D + m->num_cols_e : nullptr;
  Op lhs = [&](const double* in, double* out) {
    Vec y(nf);
    oracle_isc_sx(isc.get(), in, y.data());
    if (ar) {
      ar(ctx, y.data(), nf);
      if (Df) for (int i = 0; i < nf; ++i) y[i] += Df[i] * Df[i] * in[i];
    }
    // CG zeroes `out` first, so adding reproduces the reference's assignment.
    for (int i = 0; i < nf; ++i) out[i] += y[i];
  };
  Op pre = [&](const double* in, double* out) {
    if (preconditioner == 0) for (int i = 0; i < nf; ++i) out[i] += in[i];
    else oracle_block_diagonal_apply(m->ncb - m->nelim, isc->f_sizes.data(), minv.data(), in, out);
  };
  // ITERATIVE_SCHUR never sets cg_options.num_threads (:124-129): serial BLAS-1.
  DotFn dot = [&](const double* a, const double* c) { double s = 0; for (int i = 0; i < nf; ++i) s += a[i] * c[i]; return s; };
  Vec sol(nf, 0.0);
  cg_solve(nf, lhs, isc->rhs.data(), pre, dot, min_it, max_it, reset_period, q_tol, r_tol, sol.data(), summary);
  if (summary->termination_type != 2 && summary->termination_type != 3)
    oracle_isc_back_substitute(isc.get(), sol.data(), x);
}

// The same solver with the SCHUR_POWER_SERIES_EXPANSION options of LinearSolver::Options
// (I/linear_solver.h:168-185): preconditioner = 3 uses PowerSeriesExpansionPreconditioner with
// tolerance 0 (fixed max_num_spse_iterations terms, :178-186 of the solver), and
// use_spse_initialization starts CG from the power-series estimate of S^-1 rhs (:97-111).
void oracle_iterative_schur_solve_spse(const oracle_matrix* m, const double* v, const double* b, const double* D,
                                       int preconditioner, int min_it, int max_it, int reset_period, double q_tol,
                                       double r_tol, int use_spse_initialization, int max_num_spse_iterations,
                                       double spse_tolerance, double* x, oracle_summary* summary) {
  std::unique_ptr<oracle_isc, void (*)(oracle_isc*)> isc(oracle_isc_create(m), oracle_isc_destroy);
  oracle_isc_init(isc.get(), v, D, b);
  const int nf = m->num_cols_f;
  if (m->ncb - m->nelim == 0) {
    summary->num_iterations = 0; summary->termination_type = 0; summary->residual_norm = -1; summary->message[0] = 0;
    oracle_isc_back_substitute(isc.get(), nullptr, x);
    return;
  }
  if (use_spse_initialization || preconditioner == 1 || preconditioner == 3) oracle_isc_compute_ftf_inverse(isc.get());
  Vec sol(nf, 0.0);
  if (use_spse_initialization)
    oracle_isc_spse_apply(isc.get(), isc->rhs.data(), sol.data(), max_num_spse_iterations, spse_tolerance);
  Vec minv;
  if (preconditioner == 2) { minv.resize(m->diag_offset_f.back()); schur_jacobi(m, v, D, minv.data(), nullptr, nullptr, nullptr); }
  Op lhs = [&](const double* in, double* out) {
    Vec y(nf);
    oracle_isc_sx(isc.get(), in, y.data());
    for (int i = 0; i < nf; ++i) out[i] += y[i];
  };
  Op pre = [&](const double* in, double* out) {
    if (preconditioner == 0) for (int i = 0; i < nf; ++i) out[i] += in[i];
    else if (preconditioner == 1) oracle_block_diagonal_apply(m->ncb - m->nelim, isc->f_sizes.data(), isc->ftf_inv.data(), in, out);
    else if (preconditioner == 2) oracle_block_diagonal_apply(m->ncb - m->nelim, isc->f_sizes.data(), minv.data(), in, out);
    else { Vec y(nf); oracle_isc_spse_apply(isc.get(), in, y.data(), max_num_spse_iterations, 0.0); for (int i = 0; i < nf; ++i) out[i] += y[i]; }
  };
  DotFn dot = [&](const double* a, const double* c) { double s = 0; for (int i = 0; i < nf; ++i) s += a[i] * c[i]; return s; };
  cg_solve(nf, lhs, isc->rhs.data(), pre, dot, min_it, max_it, reset_period, q_tol, r_tol, sol.data(), summary);
  if (summary->termination_type != 2 && summary->termination_type != 3) oracle_isc_back_substitute(isc.get(), sol.data(), x);
}

void oracle_iterative_schur_solve(const oracle_matrix* m, const double* v, const double* b, const double* D,
                                  int preconditioner, int min_it, int max_it, int reset_period, double q_tol,
                                  double r_tol, double* x, oracle_summary* summary) {
  oracle_iterative_schur_solve_sharded(m, v, b, D, preconditioner, min_it, max_it, reset_period, q_tol, r_tol, x,
                                       summary, nullptr, nullptr);
}

int oracle_solver_callback(void* vctx, const double* values, const double* b, const double* D, double q_tol,
                           double r_tol, double* x, oracle_summary* summary) {
  auto* c = static_cast<oracle_solver_ctx*>(vctx);
  if (c->solver_type == 6)
    oracle_cgnr_solve(c->m, values, b, D, c->preconditioner, c->min_it, c->max_it, c->reset_period, q_tol, r_tol, x, summary);
  else
    oracle_iterative_schur_solve(c->m, values, b, D, c->preconditioner, c->min_it, c->max_it, c->reset_period, q_tol, r_tol, x, summary);
  return 0;
}

}  // extern "C"
