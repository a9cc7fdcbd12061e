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

// Right-looking blocked Cholesky with kPanel-wide panels (DenseCholesky::FactorAndSolve for DENSE_SCHUR's reduced system,
// I/schur_complement_solver.cc:163-222, I/dense_cholesky.cc).  Per panel three kernels:
//   dense_potrf_panel_kernel   the kw x kw diagonal block, in LDS, one workgroup: 16-wide steps — a 16 x 16 factorisation by ONE
//                              wavefront on registers and shuffles, the rows below by substitution, the block's own trailing update;
//   dense_trsm_panel_kernel    rows below the block: X L_kk^T = A_panel by substitution, one thread per row, L_kk in LDS;
//   dense_syrk_mfma_kernel     the trailing update C -= X X^T, the n^3 / 3 of the factorisation: v_mfma_f64_16x16x4_f64, one wavefront
//                              per 64 x 64 output tile (16 accumulator tiles), K = the panel width.
// A rank-32 update (the first version) moves 16 B per 64 flops of every trailing entry: HBM-bound at a quarter of the fp64 rate; at
// K = 128 the update is compute-bound and sits on the matrix pipe.  gfx950's fp64 MFMA peak equals its fp64 vector peak (78.6 TFLOP/s:
// 32 flop / clk / SIMD); what MFMA buys is the operand traffic — 8 eight-byte loads per lane feed 16 instructions of 2048 flops.
constexpr int kPanel = 128;
constexpr int kPanelPitch = kPanel + 1;   // LDS row pitch (doubles): column accesses hit distinct banks
constexpr int kSub = 16;                  // step width inside the diagonal block
typedef double v4f64 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void dense_potrf_panel_kernel(double* __restrict__ A, int n, int k0, int kw, int* fail_flag) {
  extern __shared__ double T[];   // [kwp][kPanelPitch], lower triangle; kwp = kw rounded up to the step width, padded with the identity
  const int tid = threadIdx.x, lane = tid & 63;
  const int kwp = (kw + kSub - 1) / kSub * kSub;
  for (int e = tid; e < kwp * kwp; e += 256) {
    const int r = e / kwp, c = e - r * kwp;
    if (c <= r) T[r * kPanelPitch + c] = (r < kw) ? A[int64_t(k0 + r) * n + (k0 + c)] : (r == c ? 1.0 : 0.0);
  }
  __syncthreads();
  for (int s0 = 0; s0 < kwp; s0 += kSub) {
    if (tid < 64) {
      // lane r < 16 owns row r of the 16 x 16 diagonal sub-block; column by column, the pivot row's entries travel by shuffle
      double row[kSub];
#pragma unroll
      for (int c = 0; c < kSub; ++c) row[c] = (lane < kSub && c <= lane) ? T[(s0 + lane) * kPanelPitch + s0 + c] : (c == lane ? 1.0 : 0.0);
      bool ok = true;
#pragma unroll
      for (int j = 0; j < kSub; ++j) {
        double d = __shfl(row[j], j, 64);   // T[j][j]
        if (!(d > 0.0)) { ok = false; d = 1.0; }
        d = sqrt(d);
        if (lane == j) row[j] = d;
        else if (lane > j) row[j] /= d;
#pragma unroll
        for (int c = j + 1; c < kSub; ++c) {
          const double tcj = __shfl(row[j], c, 64);   // T[c][j]
          if (lane >= c) row[c] -= row[j] * tcj;
        }
      }
      if (!ok && lane == 0) atomicExch(fail_flag, 1);
#pragma unroll
      for (int c = 0; c < kSub; ++c) if (lane < kSub && c <= lane) T[(s0 + lane) * kPanelPitch + s0 + c] = row[c];
    }
    __syncthreads();
    const int below = kwp - s0 - kSub;   // rows of the block under the sub-block (a multiple of 16)
    if (tid < below) {                   // x L_ss^T = T[r, s0 : s0 + 16], one thread per row, L_ss read as broadcasts
      double* tr = T + (s0 + kSub + tid) * kPanelPitch + s0;
      const double* ls = T + s0 * kPanelPitch + s0;
      double x[kSub];
#pragma unroll
      for (int c = 0; c < kSub; ++c) x[c] = tr[c];
#pragma unroll
      for (int c = 0; c < kSub; ++c) {
        double v = x[c];
#pragma unroll
        for (int q = 0; q < c; ++q) v -= x[q] * ls[c * kPanelPitch + q];
        x[c] = v / ls[c * kPanelPitch + c];
      }
#pragma unroll
      for (int c = 0; c < kSub; ++c) tr[c] = x[c];
    }
    __syncthreads();
    // the block's own trailing update, lower triangle, in 4 x 4 patches: patch (pr, pc), pc <= pr, of the (below / 4)^2 grid
    const int P = below / 4, n_patches = P * (P + 1) / 2;
    for (int e = tid; e < n_patches; e += 256) {
      int pr = int((sqrtf(8.0f * float(e) + 1.0f) - 1.0f) * 0.5f);
      while (pr * (pr + 1) / 2 > e) --pr;
      while ((pr + 1) * (pr + 2) / 2 <= e) ++pr;
      const int pc = e - pr * (pr + 1) / 2;
      const double* ra = T + (s0 + kSub + 4 * pr) * kPanelPitch + s0;
      const double* rb = T + (s0 + kSub + 4 * pc) * kPanelPitch + s0;
      double acc[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
#pragma unroll
      for (int q = 0; q < kSub; ++q) {
        double a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = ra[i * kPanelPitch + q]; b[i] = rb[i * kPanelPitch + q]; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
      }
      double* out = T + (s0 + kSub + 4 * pr) * kPanelPitch + s0 + kSub + 4 * pc;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) out[i * kPanelPitch + j] -= acc[i][j];   // (entries above the diagonal inside a diagonal patch are never read)
    }
    __syncthreads();
  }
  for (int e = tid; e < kw * kw; e += 256) {
    const int r = e / kw, c = e - r * kw;
    if (c <= r) A[int64_t(k0 + r) * n + (k0 + c)] = T[r * kPanelPitch + c];
  }
}

// Rows i >= k0 + kw: A[i, k0 : k0 + kw] <- A[i, k0 : k0 + kw] L_kk^-T.  One thread per row; L_kk in LDS (every lane reads the same
// entry: a broadcast); the row is solved 32 columns at a time in registers, finished columns are re-read from the row itself.
__global__ __launch_bounds__(64) void dense_trsm_panel_kernel(double* __restrict__ A, int n, int k0, int kw) {
  extern __shared__ double L[];   // [kw][kPanelPitch], lower triangle
  for (int e = threadIdx.x; e < kw * kw; e += 64) {
    const int r = e / kw, c = e - r * kw;
    if (c <= r) L[r * kPanelPitch + c] = A[int64_t(k0 + r) * n + (k0 + c)];
  }
  __syncthreads();
  const int i = k0 + kw + blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  double* row = A + int64_t(i) * n + k0;
  for (int c0 = 0; c0 < kw; c0 += 32) {
    double x[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) x[c] = c0 + c < kw ? row[c0 + c] : 0.0;
    for (int q0 = 0; q0 < c0; q0 += 32) {   // columns finished in earlier rounds
      double xq[32];
#pragma unroll
      for (int q = 0; q < 32; ++q) xq[q] = row[q0 + q];
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        if (c0 + c < kw) {
          const double* lc = L + (c0 + c) * kPanelPitch + q0;
          double v = 0.0;
#pragma unroll
          for (int q = 0; q < 32; ++q) v += xq[q] * lc[q];
          x[c] -= v;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      if (c0 + c < kw) {
        const double* lc = L + (c0 + c) * kPanelPitch + c0;
        double v = x[c];
#pragma unroll
        for (int q = 0; q < 32; ++q) if (q < c) v -= x[q] * lc[q];
        x[c] = v / lc[c];
      }
    }
#pragma unroll
    for (int c = 0; c < 32; ++c) if (c0 + c < kw) row[c0 + c] = x[c];
  }
}

// Trailing update of the LOWER triangle: A[i, j] -= sum_p X[i, p] X[j, p], X = A[:, k0 : k0 + kw], i >= j >= k0 + kw.
// Workgroup = 128 x 128 outputs, wavefront = 64 x 64 = 4 x 4 MFMA tiles.  v_mfma_f64_16x16x4_f64: lane l supplies A[l & 15][l >> 4] and
// B[l >> 4][l & 15] — for X X^T both are "row (l & 15) of the operand's 16 rows, column p + (l >> 4) of the panel" — and holds
// C[(l >> 4) + 4 r][l & 15], r = 0..3.  Operands come straight from memory (the panel is L2-resident: n x kw doubles), one k-step ahead.
__global__ __launch_bounds__(256) void dense_syrk_mfma_kernel(double* __restrict__ A, int n, int k0) {
  constexpr int kw = kPanel;   // a trailing update follows full panels only (a narrower last panel has nothing below it)
  const int first = k0 + kw;
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj > bi) return;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i0 = first + 128 * bi + 64 * (wv >> 1), j0 = first + 128 * bj + 64 * (wv & 1);
  if (i0 >= n || j0 >= n || j0 > i0 + 63) return;   // outside the matrix, or entirely above the diagonal
  const int lr = lane & 15, lk = lane >> 4;
  const double *ap[4], *bp[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {   // rows past the end are clamped: their results are never stored
    ap[t] = A + int64_t(min(i0 + 16 * t + lr, n - 1)) * n + k0 + lk;
    bp[t] = A + int64_t(min(j0 + 16 * t + lr, n - 1)) * n + k0 + lk;
  }
  v4f64 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = v4f64{0.0, 0.0, 0.0, 0.0};
  // two k-steps (8 panel columns) per round, the operands of the NEXT round in flight while this one multiplies: a load has two
  // steps = 32 MFMAs of 64 cycles to arrive
  double a0[4], b0[4], a1[4], b1[4], c0[4], d0[4], c1[4], d1[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) { a0[t] = ap[t][0]; b0[t] = bp[t][0]; a1[t] = ap[t][4]; b1[t] = bp[t][4]; }
  for (int p = 0; p < kw; p += 8) {
    const int pn = p + 8 < kw ? p + 8 : p;   // the last round re-loads itself: no branch in the loop
#pragma unroll
    for (int t = 0; t < 4; ++t) { c0[t] = ap[t][pn]; d0[t] = bp[t][pn]; c1[t] = ap[t][pn + 4]; d1[t] = bp[t][pn + 4]; }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[a], b0[b], acc[a][b], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[a], b1[b], acc[a][b], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) { a0[t] = c0[t]; b0[t] = d0[t]; a1[t] = c1[t]; b1[t] = d1[t]; }
  }
  // C -= acc, one row of tiles at a time: 16 loads in flight, then 16 predicated stores (addresses clamped into the matrix)
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    double c[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = min(i0 + 16 * a + lk + 4 * r, n - 1), j = min(j0 + 16 * b + lr, n - 1);
        c[b][r] = A[int64_t(i) * n + j];
      }
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + 16 * a + lk + 4 * r, j = j0 + 16 * b + lr;
        if (i < n && j <= i) A[int64_t(i) * n + j] = c[b][r] - acc[a][b][r];
      }
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
  static const hipError_t lds_ok = [] {   // the panel kernels keep a kPanel x kPanel block in LDS (132 KB)
    const int bytes = kPanel * kPanelPitch * int(sizeof(double));
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dense_potrf_panel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(dense_trsm_panel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  }();
  if (lds_ok != hipSuccess) return lds_ok;
  hipLaunchKernelGGL(dense_mirror_upper_kernel, dim3(blocks_for(int64_t(n) * n)), dim3(kB), 0, s, A, n);
  for (int k0 = 0; k0 < n; k0 += kPanel) {
    const int kw = n - k0 < kPanel ? n - k0 : kPanel;
    const size_t lds = size_t((kw + kSub - 1) / kSub * kSub) * kPanelPitch * sizeof(double);
    hipLaunchKernelGGL(dense_potrf_panel_kernel, dim3(1), dim3(256), lds, s, A, n, k0, kw, fail_flag);
    const int rest = n - k0 - kw;
    if (rest <= 0) break;
    hipLaunchKernelGGL(dense_trsm_panel_kernel, dim3((rest + 63) / 64), dim3(64), lds, s, A, n, k0, kw);
    const unsigned blocks = unsigned((rest + 127) / 128);
    hipLaunchKernelGGL(dense_syrk_mfma_kernel, dim3(blocks, blocks), dim3(256), 0, s, A, n, k0);
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
